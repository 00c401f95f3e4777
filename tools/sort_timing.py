"""tools/sort_timing.py (GPU box): time the in-tree radix sort on the BASELINE sparse-update shape (1.7 M pairs, 25-bit keys);
DIR_RS_DBG masks parts of the pass kernel (results wrong, timing only)."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dir_amd
lib = dir_amd.load_library()
n = int(os.environ.get("N", 65536 * 26)); bits = int(os.environ.get("BITS", 25))
gen = torch.Generator(device="cuda").manual_seed(0)
k = torch.randint(0, 1 << bits, (n,), generator=gen, device="cuda", dtype=torch.int64).to(torch.int32)
v = torch.arange(n, device="cuda", dtype=torch.int32)
need = int(lib.dir_debug_radix_sort_workspace_bytes(n, bits))
ws = torch.empty(need, dtype=torch.uint8, device="cuda")
ko, vo = torch.empty_like(k), torch.empty_like(v)
ki, vi = k.clone(), v.clone()
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run():
    assert lib.dir_debug_radix_sort_pairs_u32(p(ki), p(vi), n, bits, p(ko), p(vo), p(ws), need, st) == 0
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print("DIR_RS_DBG=%s n=%d bits=%d: %.1f us per sort" % (os.environ.get("DIR_RS_DBG", "0"), n, bits, e0.elapsed_time(e1) * 1e3 / 50))
