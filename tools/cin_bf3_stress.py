"""tools/cin_bf3_stress.py (GPU box) -- repeated dir_cin_layer_bf16x3_f32 / dir_cin_layer_dot_bf16x3_f32 calls at the BASELINE shape: every
rerun must be bitwise equal to the first, and within 1e-5 of the fp32-MFMA kernels."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dir_amd
from dir_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
B, m, D = 65536, 26, 16
x0 = torch.randn((B, m, D), generator=g, device="cuda") * 0.5
for Hp, H, n in [(128, 128, 40), (26, 128, 40), (200, 200, 10)]:
    xk = torch.randn((B, Hp, D), generator=g, device="cuda") * 0.5
    W = torch.randn((H, Hp * m), generator=g, device="cuda") / (Hp * m) ** 0.5
    G = torch.randn((B, H, D), generator=g, device="cuda") * 0.5
    rx, rp = ops.cin_layer(x0, xk, W, arith="f32")
    fx, fp = ops.cin_layer(x0, xk, W, arith="bf16x3")
    e = float(((fx - rx).abs() / (1 + rx.abs())).max())
    neq = 0
    for _ in range(n):
        x, p = ops.cin_layer(x0, xk, W, arith="bf16x3")
        neq += (not torch.equal(x, fx)) or (not torch.equal(p, fp))
    d0, dk, _ = ops.cin_layer_backward(x0, xk, W, G, need_w=False, arith="f32")
    bk, b0 = ops.cin_dx_bf16x3(x0, xk, W, G)
    e2 = max(float(((bk - dk).abs() / (1 + dk.abs())).max()), float(((b0 - d0).abs() / (1 + d0.abs())).max()))
    neq2 = 0
    for _ in range(n):
        k2, z2 = ops.cin_dx_bf16x3(x0, xk, W, G)
        neq2 += (not torch.equal(k2, bk)) or (not torch.equal(z2, b0))
    w32 = ops.cin_dw(x0, xk, G, arith="f32")
    w3 = ops.cin_dw(x0, xk, G, arith="bf16x3")
    e3 = float(((w3 - w32).abs() / (129.0 + w32.abs())).max())
    neq3 = 0
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        neq3 += not torch.equal(ops.cin_dw(x0, xk, G, arith="bf16x3"), w3)
    t1.record(); torch.cuda.synchronize()
    print("   weight gradient vs fp32 kernel %.2e (scaled by 129 + |ref|), reruns not equal %d / %d, %.2f ms per call (incl. the compare)" % (e3, neq3, n, t0.elapsed_time(t1) / n))
    print("Hp %d H %d: forward vs fp32 kernel %.2e, reruns not bitwise equal %d / %d; data gradients vs fp32 kernel %.2e, reruns not equal %d / %d"
          % (Hp, H, e, neq, n, e2, neq2, n), flush=True)
    del xk, W, G, rx, rp, fx, fp, d0, dk, bk, b0
