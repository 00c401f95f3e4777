import torch
M,K=65536,400
h=torch.randn(M,K,device='cuda'); g=torch.randn(M,1,device='cuda'); w=torch.randn(1,K,device='cuda')
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
def run(name,f):
    for _ in range(3): r=f()
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): r=f()
    e1.record(); torch.cuda.synchronize()
    print("%-44s %8.1f us" % (name, e0.elapsed_time(e1)*1e3/20)); return r
ref=run("g.t() @ h  (linear's dW)", lambda: g.t() @ h)
r=run("torch.mv(h.t(), g[:,0])", lambda: torch.mv(h.t(), g[:,0])); print("   diff", float((r-ref[0]).abs().max()))
r=run("(g*h).sum(0)", lambda: (g*h).sum(0)); print("   diff", float((r-ref[0]).abs().max()))
r=run("bmm 16 slices + sum", lambda: torch.bmm(g.view(16,M//16,1).transpose(1,2), h.view(16,M//16,K)).sum(0)); print("   diff", float((r[0]-ref[0]).abs().max()))
r=run("einsum('mk,m->k')", lambda: torch.einsum('mk,m->k', h, g[:,0]))
run("g @ w  (linear's dX)", lambda: g @ w)
run("g * w  (broadcast)", lambda: g * w)
run("forward h @ w.t()", lambda: h @ w.t())
run("forward torch.mv(h, w[0])", lambda: torch.mv(h, w[0]))
