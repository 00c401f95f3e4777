"""tools/dense_bf3_stress.py (GPU box) -- repeated dir_dense_bf16x3_f32 calls against the fp32-MFMA kernel's result: counts outputs that are off
by more than 1e-4 (a wrong piece product, not rounding) and checks run-to-run bitwise equality."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dir_amd
from dir_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
for M, Kd, N, gated in [(12288, 400, 416, False), (12288, 416, 400, False), (65536, 416, 400, False), (12300, 1024, 432, True), (65536, 400, 416, True)]:
    x = torch.randn((M, Kd), generator=g, device="cuda")
    w = torch.randn((N, Kd), generator=g, device="cuda") / Kd ** 0.5
    gate = torch.randn((M, N), generator=g, device="cuda")
    f = (lambda a: ops.dense_gated(x, w, gate, arith=a)) if gated else (lambda a: ops.dense(x, w, None, arith=a))
    ref = f("f32")
    first = f("bf16x3")
    bad_runs, neq_runs, worst = 0, 0, 0.0
    for it in range(30):
        y = f("bf16x3")
        err = ((y - ref).abs() / (1 + ref.abs()))
        nb = int((err > 1e-4).sum())
        bad_runs += nb > 0
        neq_runs += not torch.equal(y, first)
        worst = max(worst, float(err.max()))
    print("M %d Kd %d N %d gated %d: runs with wrong outputs %d / 30, runs not bitwise equal to the first %d / 30, worst scaled err %.2e"
          % (M, Kd, N, gated, bad_runs, neq_runs, worst), flush=True)
