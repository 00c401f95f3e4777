"""tools/cin_pooled_probe.py (GPU box) -- the LAST layer of a CIN stack only feeds its pooled sums, and
  pooled[b,h] = sum_d xout[b,h,d] = sum_{i,j} W[h,i,j] * Z[b,i,j],   Z[b,i,j] = sum_d xk[b,i,d] x0[b,j,d]:
the sum over d can be taken BEFORE the contraction with W -- 1/D of the matrix work.  Times and checks the formulation built from a batched
product and the dense kernel against the layer kernel, forward and backward."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dir_amd
from dir_amd import ops
B, m, D, Hp, H = 65536, 26, 16, 128, 128
g = torch.Generator(device="cuda").manual_seed(3)
x0 = torch.randn((B, m, D), generator=g, device="cuda") * 0.25
xk = torch.randn((B, Hp, D), generator=g, device="cuda") * 0.25
W = torch.randn((H, Hp * m), generator=g, device="cuda") / (Hp * m) ** 0.5
gp = torch.randn((B, H), generator=g, device="cuda") * 0.1
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
_, ref = ops.cin_layer(x0, xk, W, want_xout=False)
def zform(): return torch.bmm(xk, x0.transpose(1, 2))
def fwd():
    Z = zform()
    return ops.dense(Z.view(B, Hp * m), W), Z
p, Z = fwd()
print("pooled: max diff %.2e (scale %.2f)" % (float((p - ref).abs().max()), float(ref.abs().max())))
print("layer kernel (pooled only) %.3f ms | Z by bmm %.3f ms, Z + dense %.3f ms" % (t(lambda: ops.cin_layer(x0, xk, W, want_xout=False)), t(zform), t(fwd)))
# backward of the top layer given g_pooled
G = gp.reshape(B, H, 1).expand(B, H, D).contiguous()
r0, rk, rw = ops.cin_layer_backward(x0, xk, W, G)
def bwd():
    dW = ops.dense_dw(gp, Z.view(B, Hp * m))                       # [H, Hp*m]
    dZ = ops.dense(gp, W.t()).view(B, Hp, m)                       # [B, Hp, m]
    dxk = torch.bmm(dZ, x0)                                        # [B, Hp, D]
    dx0 = torch.bmm(dZ.transpose(1, 2), xk)                        # [B, m, D]
    return dx0, dxk, dW
b0, bk, bw = bwd()
for nme, a, r in (("dx0", b0, r0), ("dxk", bk, rk), ("dW", bw, rw)):
    print("%s: max diff %.2e (scale %.2f)" % (nme, float((a - r).abs().max()), float(r.abs().max())))
print("layer backward kernels %.3f ms | pooled form %.3f ms (dense_dw %.3f, dense %.3f, two bmm %.3f)" % (
    t(lambda: ops.cin_layer_backward(x0, xk, W, G)), t(bwd), t(lambda: ops.dense_dw(gp, Z.view(B, Hp * m))), t(lambda: ops.dense(gp, W.t())),
    t(lambda: (torch.bmm(Z, x0), torch.bmm(Z.transpose(1, 2), xk)))))
