import sys, torch
sys.path.insert(0, "/root/repo")
import dir_amd
from dir_amd import ops
B, m, D, H = 65536, 26, 16, 128
g = torch.Generator(device="cuda").manual_seed(1)
x0 = torch.randn((B, m, D), generator=g, device="cuda") * 0.25
W = torch.randn((H, m * m), generator=g, device="cuda") / m
G = torch.randn((B, H, D), generator=g, device="cuda") * 0.1
W3 = W.view(H, m, m)
Wsym = (W3 + W3.transpose(1, 2)).permute(1, 0, 2).reshape(m, H * m).contiguous()      # [i, h*m + j]
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
a = lambda: ops.cin_dx_bf16x3(x0, x0, W, G)
b = lambda: ops.cin_layer(x0, G, Wsym)
dxk, dx0 = a()
tot, _ = b()
ref = dxk + dx0
print("max diff", float((tot - ref).abs().max()), "scale", float(ref.abs().max()))
print("dot form %.3f ms   forward form on symmetrised W %.3f ms" % (t(a), t(b)))
