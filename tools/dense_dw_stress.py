"""tools/dense_dw_stress.py (GPU box) -- repeated calls of the weight-gradient kernels (dir_dense_dw_bf16x3_f32, dir_dense_dw_small_f32) at
the tower shapes: every rerun bitwise equal to the first, and within the float64 bar."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dir_amd  # noqa: E402,F401
from dir_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(11)
bad = 0
for M, N, K, arith, reps in [(65536, 400, 416, "bf16x3", 60), (65536, 1024, 432, "bf16x3", 30), (65521, 360, 416, "bf16x3", 40), (12288, 400, 400, "bf16x3", 60),
                             (65536, 80, 64, "small", 60), (65536, 80, 200, "small", 40), (40000, 128, 128, "small", 40)]:
    G = torch.randn((M, N), generator=gen, device=dev) * 0.5
    X = torch.randn((M, K), generator=gen, device=dev)
    w0, b0 = ops.dense_dw(G, X, arith=arith, want_bias=True)
    ref = G.double().t() @ X.double()
    err = float((w0.double() - ref).abs().max()) / (1 + M ** 0.5 * 0.5)
    errb = float((b0.double() - G.double().sum(0)).abs().max()) / (1 + M ** 0.5 * 0.5)
    diff = 0
    for _ in range(reps):
        w, b = ops.dense_dw(G, X, arith=arith, want_bias=True)
        diff += int(not torch.equal(w, w0)) + int(not torch.equal(b, b0))
    bad += diff + int(err > 1e-5) + int(errb > 1e-5)
    print("M=%6d N=%5d K=%5d %-7s reruns not bitwise equal %d / %d, scaled err dW %.2e db %.2e" % (M, N, K, arith, diff, reps, err, errb), flush=True)
print("FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
