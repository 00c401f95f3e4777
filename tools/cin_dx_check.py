"""Development check: cin_dx against a float64 einsum on the GPU (which output, which rows/cols are off)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dir_amd
from dir_amd import ops
B, m, D, Hp, H = [int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (64, 26, 16, 26, 128))]
g = torch.Generator().manual_seed(0)
x0 = (torch.randn(B, m, D, generator=g) * 0.5).cuda(); xk = (torch.randn(B, Hp, D, generator=g) * 0.5).cuda()
W = (torch.randn(H, Hp * m, generator=g) / (Hp * m) ** 0.5).cuda(); G = (torch.randn(B, H, D, generator=g) * 0.5).cuda()
dxk, dx0 = ops.cin_dx(x0, xk, W, G)
W3 = W.double().view(H, Hp, m)
rxk = torch.einsum("hij,bhd,bjd->bid", W3, G.double(), x0.double())
rx0 = torch.einsum("hij,bhd,bid->bjd", W3, G.double(), xk.double())
for name, got, ref in (("dxk", dxk, rxk), ("dx0", dx0, rx0)):
    err = (got.double() - ref).abs() / (1 + ref.abs())
    print(name, "max err %.3e" % err.max().item())
    if err.max() > 1e-4:
        bad = (err > 1e-4).nonzero()
        print("  bad count", bad.shape[0], "of", err.numel(), "first", bad[:6].tolist())
        print("  bad d values", sorted(set(bad[:, 2].tolist())), " bad channel values", sorted(set(bad[:, 1].tolist()))[:40])
