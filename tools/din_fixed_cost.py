"""tools/din_fixed_cost.py (GPU box): the DIN kernels' fixed cost per launch (weight images built by every workgroup, queue, tail): device time of the
forward, the saving forward and the saved-activation backward at B = 2 048 ... 65 536 with the same per-sample work -- the intercept of time over B."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dir_amd
from dir_amd import ops
T, K, V, H1, H2 = 50, 64, 10_000_000, 80, 40
g = torch.Generator(device="cuda").manual_seed(4)
table = torch.randn((V, K), generator=g, device="cuda") * 0.125
Ws = [torch.randn((4 * K, H1), generator=g, device="cuda") * 0.1, torch.randn((H1,), generator=g, device="cuda") * 0.1,
      torch.randn((H1, H2), generator=g, device="cuda") * 0.2, torch.randn((H2,), generator=g, device="cuda") * 0.1,
      torch.randn((H2,), generator=g, device="cuda") * 0.5, torch.randn((1,), generator=g, device="cuda")]


def dev_us(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for B in (2048, 4096, 8192, 16384, 32768, 65536):
    hist = torch.randint(0, V, (B, T), generator=g, device="cuda")
    hl = torch.randint(1, T + 1, (B,), generator=g, device="cuda", dtype=torch.int32)
    cand = torch.randint(0, V, (B,), generator=g, device="cuda")
    gout = torch.randn((B, K), generator=g, device="cuda")
    t_f = dev_us(lambda: ops.din_attention_pool(table, hist, hl, cand, *Ws, normalize=True))
    plan = ops.DinTrainPlan(hist, hl)
    out, sc, saved = ops.din_attention_pool_save(table, hist, hl, cand, *Ws, normalize=True, plan=plan)
    t_s = dev_us(lambda: ops.din_attention_pool_save(table, hist, hl, cand, *Ws, normalize=True, plan=plan))
    t_b = dev_us(lambda: ops.din_attention_pool_backward(table, hist, hl, cand, *Ws, gout, normalize=True, scores=sc, saved=saved), n=10)
    print("B %6d: forward %7.1f us, saving forward %7.1f us, backward (all kernels + torch glue) %8.1f us" % (B, t_f, t_s, t_b), flush=True)
