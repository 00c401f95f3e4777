"""tools/dense_dw_probe.py (GPU box) -- dir_dense_dw_bf16x3_f32 vs the library's formulation of the weight-gradient GEMM g^T x: error
against float64 and time at the tower shapes (M = 65 536 rows)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dir_amd  # noqa: E402,F401
from dir_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(5)


def timed(f, n=10):
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for M, N, K in [(300, 40, 52), (4096, 400, 416), (65536, 400, 416), (65536, 400, 400), (65536, 1024, 432), (65536, 1024, 1024), (65536, 200, 360),
                (65536, 128, 128), (65536, 512, 256), (12288, 400, 416), (65521, 400, 416)]:
    G = torch.randn((M, N), generator=g, device=dev) * 0.5
    X = torch.randn((M, K), generator=g, device=dev)
    ref = (G.double().t() @ X.double())
    a = ops.dense_dw(G, X, arith="bf16x3")
    b = ops.dense_dw(G, X, arith="f32")
    scale = 1 + ref.abs()
    ea, eb = float(((a.double() - ref).abs() / scale).max()), float(((b.double() - ref).abs() / scale).max())
    same = torch.equal(ops.dense_dw(G, X, arith="bf16x3"), a)
    ta = timed(lambda: ops.dense_dw(G, X, arith="bf16x3"))
    tb = timed(lambda: ops.dense_dw(G, X, arith="f32"))
    print("M=%6d N=%5d K=%5d  bf16x3 %8.1f us (err %.2e, rerun equal %s)   library %8.1f us (err %.2e)   auto=%s" %
          (M, N, K, ta, ea, same, tb, eb, ops.dense_dw_auto_arith(M, N, K)), flush=True)
