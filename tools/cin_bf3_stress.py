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

# the first layer of a stack (xk IS x0): the pair-form forward (dir_cin_layer1_bf16x3_f32), the pair-form weight gradient
# (dir_cin_dw_sym_bf16x3_f32) and the whole-stack backward (ops.cin_stack_backward) at the BASELINE shape
H, n = 128, 40
W = torch.randn((H, m * m), generator=g, device="cuda") / m
G = torch.randn((B, H, D), generator=g, device="cuda") * 0.5
rx, rp = ops.cin_layer(x0, x0.clone(), W, arith="bf16x3")            # a copy: the general kernel
fx, fp = ops.cin_layer(x0, x0, W, arith="bf16x3")
e = float(((fx - rx).abs() / (1 + rx.abs())).max())
neq = sum(int((not torch.equal(a, fx)) or (not torch.equal(b, fp))) for a, b in (ops.cin_layer(x0, x0, W, arith="bf16x3") for _ in range(n)))
w32 = ops.cin_dw(x0, x0, G, arith="f32")
ws = ops.cin_dw(x0, x0, G, arith="bf16x3_sym")
e3 = float(((ws - w32).abs() / (129.0 + w32.abs())).max())
neq3 = sum(int(not torch.equal(ops.cin_dw(x0, x0, G, arith="bf16x3_sym"), ws)) for _ in range(n))
print("first layer (xk is x0): pair forward vs the general bf16x3 kernel %.2e, reruns not bitwise equal %d / %d; pair weight gradient vs fp32 kernel "
      "%.2e, reruns not equal %d / %d" % (e, neq, n, e3, neq3, n), flush=True)
Hs = (128, 128, 128)
Ws, hp = [], m
for h in Hs:
    Ws.append(torch.randn((h, hp * m), generator=g, device="cuda") / (hp * m) ** 0.5)
    hp = h
xks, xk = [x0], x0
for Wk in Ws[:-1]:
    xk, _ = ops.cin_layer(x0, xk, Wk)
    xks.append(xk)
gp = torch.randn((B, sum(Hs)), generator=g, device="cuda") * 0.1
d0, dws = ops.cin_stack_backward(x0, xks, Ws, gp)
neq4 = 0
for _ in range(10):
    d1, dw1 = ops.cin_stack_backward(x0, xks, Ws, gp)
    neq4 += int((not torch.equal(d1, d0)) or any(not torch.equal(a, b) for a, b in zip(dw1, dws)))
print("whole-stack backward (3 x 128): reruns not bitwise equal %d / 10" % neq4, flush=True)
