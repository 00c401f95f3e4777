"""tools/dense_bf3_probe.py (GPU box) -- the bf16 x 3 dense layer (dir_dense_bf16x3_f32, csrc/dense_bf3.hip) against the fp32-MFMA kernel
(dir_dense_f32): scaled error against float64 and time per layer at the tower shapes of the BASELINE models."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dir_amd  # noqa: E402,F401
from dir_amd import ops  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


g = torch.Generator(device="cuda").manual_seed(0)
for M, Kd, N in [(65536, 416, 400), (65536, 400, 400), (65536, 432, 1024), (65536, 1024, 1024), (65536, 416, 360), (65536, 360, 200),
                 (65536, 200, 80), (16384, 416, 400), (4096, 416, 400)]:
    x = torch.randn((M, Kd), generator=g, device="cuda")
    w = torch.randn((N, Kd), generator=g, device="cuda") / Kd ** 0.5
    b = torch.randn((N,), generator=g, device="cuda") * 0.1
    y32 = ops.dense(x, w, b, relu=True, arith="f32")
    y3 = ops.dense(x, w, b, relu=True, arith="bf16x3")
    sel = torch.arange(0, M, max(1, M // 512), device="cuda")
    ref = (x[sel].double() @ w.double().t() + b.double()).clamp(min=0)
    e32 = float(((y32[sel].double() - ref).abs() / (1 + ref.abs())).max())
    e3 = float(((y3[sel].double() - ref).abs() / (1 + ref.abs())).max())
    t32 = timeit(lambda: ops.dense(x, w, b, relu=True, out=y32, arith="f32"))
    t3 = timeit(lambda: ops.dense(x, w, b, relu=True, out=y3, arith="bf16x3"))
    fl = 2.0 * M * Kd * N
    print("M %6d Kd %4d N %4d | fp32 MFMA %7.1f us (%.2f of 157.3 TF) err %.2e | bf16x3 %7.1f us (%.1f TF fp32-equivalent) err %.2e | x%.2f | auto: %s"
          % (M, Kd, N, t32, fl / t32 / 1e6 / 157.3, e32, t3, fl / t3 / 1e6, e3, t32 / t3, ops.dense_auto_arith(M, Kd, N)), flush=True)
