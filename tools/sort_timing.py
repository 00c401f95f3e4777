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

# the slot-major sort of ids [B, F] (what the sorted sparse updates run at B >= 4096)
B, F, V = int(os.environ.get("B", 65536)), int(os.environ.get("F", 26)), int(os.environ.get("V", 1000000))
ids = torch.randint(0, V, (B, F), generator=gen, device="cuda")
row_base = torch.arange(F, device="cuda", dtype=torch.int64) * V
need2 = int(lib.dir_debug_slot_sort_workspace_bytes(B, F, V * F))
ws2 = torch.empty(need2, dtype=torch.uint8, device="cuda")
ko2, vo2 = torch.empty(B * F, dtype=torch.int32, device="cuda"), torch.empty(B * F, dtype=torch.int32, device="cuda")
def run2():
    assert lib.dir_debug_slot_sort_entries(p(ids), ids.stride(0), ids.stride(1), F, B, p(row_base), V * F, p(ko2), p(vo2), p(ws2), need2, st) == 0
for _ in range(5): run2()
torch.cuda.synchronize()
e0.record()
for _ in range(50): run2()
e1.record(); torch.cuda.synchronize()
print("DIR_RS_DBG=%s slot-major B=%d F=%d V=%d: %.1f us per sort (keys from ids included)" % (os.environ.get("DIR_RS_DBG", "0"), B, F, V, e0.elapsed_time(e1) * 1e3 / 50))
