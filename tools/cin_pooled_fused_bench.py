import sys, torch
sys.path.insert(0, "/root/repo")
from dir_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
B, m, Hp, H, D = 65536, 26, 128, 128, 16
x0 = torch.randn((B, m, D), generator=g, device="cuda") * 0.5
xk = torch.randn((B, Hp, D), generator=g, device="cuda") * 0.5
W = torch.randn((H, Hp * m), generator=g, device="cuda") / (Hp * m) ** 0.5
for _ in range(4): ops.cin_layer(x0, xk, W, want_xout=False)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): ops.cin_layer(x0, xk, W, want_xout=False)
b.record(); torch.cuda.synchronize()
print("%.1f us" % (a.elapsed_time(b) * 100))
