"""tools/head_bwd_probe.py (GPU box): dir_units1_relu_backward(_bits)_f32 at the towers' head widths, 65 536 rows."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


B = 65536
for N in (80, 400, 1024, 2048):
    y = torch.relu(torch.randn((B, N), device="cuda"))
    g = torch.randn((B, 1), device="cuda") * 1e-3
    w = torch.randn((1, N), device="cuda")
    a = t(lambda: ops.units1_relu_backward(g, w, y))
    b = t(lambda: ops.units1_relu_backward(g, w, y, want_bits=True))
    print("N=%4d: %.1f us (%.2f TB/s of y + gx), with the scales %.1f us" % (N, a, 2 * B * N * 4 / a / 1e6, b))
