"""tools/dense_shapes_probe.py (GPU box): dir_dense_bf16x3_f32 on the layer shapes of the models (device time per call, share of the bf16 pipe with
the x 6 accounting)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dir_amd
from dir_amd import ops
M = 65536
g = torch.Generator(device="cuda").manual_seed(0)
for Kd, N in ((432, 1024), (1024, 1024), (416, 400), (400, 400), (416, 360), (360, 200), (200, 80), (416, 1024), (1024, 512)):
    x = torch.randn((M, Kd), generator=g, device="cuda") * 0.3
    w = torch.randn((N, Kd), generator=g, device="cuda") / Kd ** 0.5
    b = torch.zeros(N, device="cuda")
    y = torch.empty((M, N), device="cuda")
    f = lambda: ops.dense(x, w, b, relu=True, out=y, arith="bf16x3")
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30
    print("M %d  Kd %4d -> N %4d: %7.1f us, %.2f of the bf16 pipe (x 6), in + out %.0f MB" % (M, Kd, N, us, 2.0 * M * Kd * N * 6 / (us * 1e-6) / 2.5e15, M * (Kd + N) * 4 / 1e6), flush=True)
