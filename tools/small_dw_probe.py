"""tools/small_dw_probe.py (GPU box): the narrow layers of the towers in training -- dW = g^T x for an 80 x 200 layer at 65 536 rows on
dir_dense_dw_small_f32 / dir_dense_dw_bf16x3_f32 / the library's batched GEMM + sum, and the units = 1 forward on dir_units1_f32 / the library."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dir_amd  # noqa: F401
from dir_amd import ops


def t(f, n=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


M = 65536
shapes = ((80, 200), (80, 360), (128, 256), (40, 80), (64, 416), (80, 64), (80, 128), (128, 128), (128, 416), (128, 1024), (200, 200), (16, 416), (32, 200),
          (64, 64), (200, 80), (360, 80), (1024, 128), (256, 256), (320, 320))
for N, K in shapes:
    g = torch.randn((M, N), device="cuda") * 1e-3
    x = torch.randn((M, K), device="cuda")
    line = "dW %3d x %3d:" % (N, K)
    gb = ops.grad_bits(g)
    for a in ("small", "bf16x3", "f16x2", "f32"):
        try:
            line += "  %s %.1f us" % (a, t(lambda: ops.dense_dw(g, x, arith=a, want_bias=True, g_bits=gb[1] if a == "f16x2" else None)))
        except Exception as e:
            line += "  %s n/a" % a
    print(line + "   (auto: %s)" % ops.dense_dw_auto_arith(M, N, K))
for N in (80, 400, 429, 1024, 384):
    x = torch.randn((M, N), device="cuda")
    w = torch.randn((1, N), device="cuda")
    b = torch.zeros(1, device="cuda")
    print("units1 forward N=%4d:  dir_units1_f32 %.1f us   x @ w.t() + b %.1f us   F.linear %.1f us" % (
        N, t(lambda: ops.units1(x, w, b)), t(lambda: x @ w.t() + b), t(lambda: torch.nn.functional.linear(x, w, b))))
