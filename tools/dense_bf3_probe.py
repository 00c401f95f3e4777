"""tools/dense_bf3_probe.py (GPU box) -- the bf16 x 3 split-operand dense layer (dir_dense_bf16x3_f32) against the fp32-MFMA kernel
(dir_dense_f32): error vs float64 on the acceptance shapes, and time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dir_amd
from dir_amd import ops
dir_amd.load_library()
g = torch.Generator(device="cuda").manual_seed(0)
def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, Kd, N in ((65536, 416, 400), (65536, 400, 400), (65536, 1024, 1024), (65536, 432, 1024), (4096, 64, 80)):
    x = torch.randn((M, Kd), generator=g, device="cuda")
    w = torch.randn((N, Kd), generator=g, device="cuda") / Kd ** 0.5
    b = torch.randn((N,), generator=g, device="cuda") * 0.1
    planes = ops.dense_bf3_planes(w)
    y3 = ops.dense_bf3(x, planes, Kd, b, relu=True)
    y1 = ops.dense(x, w, b, relu=True)
    sel = torch.arange(0, M, max(1, M // 512), device="cuda")
    ref = (x[sel].double() @ w.double().t() + b.double()).clamp(min=0)
    e3 = ((y3[sel].double() - ref).abs() / (1 + ref.abs())).max().item()
    e1 = ((y1[sel].double() - ref).abs() / (1 + ref.abs())).max().item()
    t3 = timeit(lambda: ops.dense_bf3(x, planes, Kd, b, relu=True, out=y3))
    t1 = timeit(lambda: ops.dense(x, w, b, relu=True, out=y1))
    fl = 2.0 * M * Kd * N
    print("M %6d Kd %4d N %4d | fp32 MFMA %7.1f us (%.2f of 157.3 TF) err %.2e | bf16x3 %7.1f us (%.1f TF fp32-equivalent) err %.2e | x%.2f"
          % (M, Kd, N, t1, fl / t1 / 1e6 / 157.3, e1, t3, fl / t3 / 1e6, e3, t1 / t3))
