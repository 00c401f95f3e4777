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
