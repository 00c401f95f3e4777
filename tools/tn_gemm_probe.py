"""tn_gemm_probe.py -- library formulations of the weight-gradient GEMM g^T x (reduction over the 65 536 batch rows): single TN GEMM vs
batched row slices + sum (development tool; result quoted in details-in-recommendation_amd/dense.py)."""
import torch
M,N,Kd=65536,400,416
g=torch.randn(M,N,device='cuda'); x=torch.randn(M,Kd,device='cuda')
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
def run(name,f):
    for _ in range(3): r=f()
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): r=f()
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/10
    print("%-40s %8.1f us  %6.1f TF" % (name, us, 2*M*N*Kd/us/1e6))
    return r
ref=run("g.t() @ x", lambda: g.t() @ x)
run("(x.t() @ g).t()", lambda: (x.t() @ g).t())
gt=g.t().contiguous()
run("g.t().contiguous() @ x (excl. transpose)", lambda: gt @ x)
for S in (8,16,32,64,128):
    r=run("bmm split S=%d + sum"%S, lambda: torch.bmm(g.view(S,M//S,N).transpose(1,2), x.view(S,M//S,Kd)).sum(0))
    print("   max diff", float((r-ref).abs().max()))
# other BLAS back ends / formulations (round 2)
for lib in ("hipblaslt", "cublas"):
    try:
        torch.backends.cuda.preferred_blas_library(lib)
    except Exception as ex:
        print("preferred_blas_library(%s): %r" % (lib, ex)); continue
    run("[%s] g.t() @ x" % lib, lambda: g.t() @ x)
    for S in (8, 16, 32):
        run("[%s] bmm split S=%d + sum" % (lib, S), lambda: torch.bmm(g.view(S, M // S, N).transpose(1, 2), x.view(S, M // S, Kd)).sum(0))
    out = torch.empty(N, Kd, device='cuda')
    run("[%s] baddbmm chain of 16 into one output" % lib, lambda: [out.zero_()] + [torch.addmm(out, g[i * 4096:(i + 1) * 4096].t(), x[i * 4096:(i + 1) * 4096], out=out) for i in range(16)])
# half the shapes of the towers
for (N2, K2) in ((1024, 432), (1024, 1024), (200, 360), (80, 200)):
    g2 = torch.randn(M, N2, device='cuda'); x2 = torch.randn(M, K2, device='cuda')
    for S in (1, 16):
        e0.record()
        for _ in range(5):
            r = (g2.t() @ x2) if S == 1 else torch.bmm(g2.view(S, M // S, N2).transpose(1, 2), x2.view(S, M // S, K2)).sum(0)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 5
        print("N=%d Kd=%d S=%d: %.1f us, %.1f TF" % (N2, K2, S, us, 2 * M * N2 * K2 / us / 1e6))
