"""tools/din_pack_probe.py (GPU box): the packed DIN unit at config 4's batch against the table size (HBM vs cache-resident rows), the
length distribution and the sample weight of the partition -- where its time goes.  Prints us per launch (HIP events, median of 30)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dir_amd import ops

def run(V, lens, B=65536, T=50, reps=30, **kw):
    g = torch.Generator(device="cuda").manual_seed(1)
    K, H1, H2 = 64, 80, 40
    table = torch.randn((V, K), generator=g, device="cuda") * 0.125
    hist = torch.randint(0, V, (B, T), generator=g, device="cuda")
    if lens == "uniform":
        hl = torch.randint(1, T + 1, (B,), generator=g, device="cuda", dtype=torch.int32)
    else:
        hl = torch.full((B,), int(lens), device="cuda", dtype=torch.int32)
    cand = torch.randint(0, V, (B,), generator=g, device="cuda")
    W1 = torch.randn((4 * K, H1), generator=g, device="cuda") * 0.05
    W2 = torch.randn((H1, H2), generator=g, device="cuda") * 0.1
    W3 = torch.randn((H2,), generator=g, device="cuda") * 0.1
    b1, b2, b3 = torch.zeros(H1, device="cuda"), torch.zeros(H2, device="cuda"), torch.zeros(1, device="cuda")
    f = lambda: ops.din_attention_pool(table, hist, hl, cand, W1, b1, W2, b2, W3, b3, normalize=True, arith="f16x2", **kw)
    for _ in range(3):
        f()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]

if __name__ == "__main__":
    for V in (10000000, 1000000, 100000, 2000):
        print("V %9d lengths U{1..50}: %7.1f us" % (V, run(V, "uniform")))
    for L in (1, 8, 16, 17, 32, 48, 50):
        print("V  10000000 length %2d:        %7.1f us  (%.1f ns per row)" % (L, run(10000000, L), run(10000000, L) * 1e3 / (65536 * L)))
    for B in (256, 2048, 16384):
        print("B %6d length 1 (prologue + launch): %7.1f us" % (B, run(10000000, 1, B=B)))
