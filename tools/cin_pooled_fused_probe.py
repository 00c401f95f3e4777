import os, sys, torch
sys.path.insert(0, "/root/repo")
from dir_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
for (B, m, Hp, H) in ((1000, 26, 128, 128), (37, 26, 7, 40), (513, 32, 50, 128), (129, 5, 3, 16), (4099, 13, 64, 100), (65536, 26, 128, 128)):
    D = 16
    x0 = torch.randn((B, m, D), generator=g, device="cuda") * 0.5
    xk = torch.randn((B, Hp, D), generator=g, device="cuda") * 0.5
    W = torch.randn((H, Hp * m), generator=g, device="cuda") / (Hp * m) ** 0.5
    ops.CIN_POOLED_FUSED = True
    _, p1 = ops.cin_layer(x0, xk, W, want_xout=False)
    _, p1b = ops.cin_layer(x0, xk, W, want_xout=False)
    ops.CIN_POOLED_FUSED = False
    _, p0 = ops.cin_layer(x0, xk, W, want_xout=False)
    Z = torch.einsum("bid,bjd->bij", xk[:2048].double(), x0[:2048].double()).reshape(min(B, 2048), -1)
    ref = Z @ W.double().t()
    e1 = float(((p1[:2048].double() - ref).abs() / (1 + ref.abs())).max()); e0 = float(((p0[:2048].double() - ref).abs() / (1 + ref.abs())).max())
    print(B, m, Hp, H, "fused err %.2e two-pass err %.2e  bitwise rerun %s" % (e1, e0, torch.equal(p1, p1b)))
    if B == 65536:
        for name, flag in (("fused", True), ("two-pass", False)):
            ops.CIN_POOLED_FUSED = flag
            for _ in range(3): ops.cin_layer(x0, xk, W, want_xout=False)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20): ops.cin_layer(x0, xk, W, want_xout=False)
            b.record(); torch.cuda.synchronize()
            print(name, "%.1f us" % (a.elapsed_time(b) * 1e3 / 20))
