"""tools/din_bwd_probe.py (GPU box) -- the two DIN backward implementations timed alone at the cfg-4 shape (development tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dir_amd
from dir_amd import ops
dir_amd.load_library()
g = torch.Generator(device="cuda").manual_seed(0)
B, T, K, H1, H2, V = 65536, 50, 64, 80, 40, int(os.environ.get("V", 10000000))
table = torch.randn((V, K), generator=g, device="cuda") * 0.1
hist = torch.randint(0, V, (B, T), generator=g, device="cuda")
hl = torch.randint(1, T + 1, (B,), generator=g, device="cuda").to(torch.int32)
cand = torch.randint(0, V, (B,), generator=g, device="cuda")
Ws = [torch.randn((4 * K, H1), generator=g, device="cuda") * 0.1, torch.zeros(H1, device="cuda"), torch.randn((H1, H2), generator=g, device="cuda") * 0.2,
      torch.zeros(H2, device="cuda"), torch.randn(H2, generator=g, device="cuda") * 0.5, torch.zeros(1, device="cuda")]
gout = torch.randn((B, K), generator=g, device="cuda")
out, scores = ops.din_attention_pool(table, hist, hl, cand, *Ws, normalize=True, want_scores=True)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
t_old = timeit(lambda: ops.din_attention_pool_backward(table, hist, hl, cand, *Ws, gout, normalize=True))
t_new = timeit(lambda: ops.din_attention_pool_backward(table, hist, hl, cand, *Ws, gout, normalize=True, scores=scores))
t_fwd = timeit(lambda: ops.din_attention_pool(table, hist, hl, cand, *Ws, normalize=True))
print("forward %.3f ms | backward single kernel %.3f ms | rows + wgrad %.3f ms  (wrappers included)" % (t_fwd, t_old, t_new))
