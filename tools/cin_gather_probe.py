"""tools/cin_gather_probe.py (GPU box): the CIN stack at config 5 (B 65 536, m 26, D 16, 3 x 128) on a materialised x0 against x0 read through
inverse positions of a shuffled row list (ops.cin_stack_gather), whole batch and as two half batches (the sharded lookup's micro-batches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dir_amd  # noqa: F401
from dir_amd import ops

B, m, D, Hs = 65536, 26, 16, (128, 128, 128)
g = torch.Generator(device="cuda").manual_seed(1)
n = B * m
rows = torch.randn((n, D), generator=g, device="cuda") * 0.25
mode = os.environ.get("INV", "shuffled")
if mode == "shuffled":
    inv = torch.randperm(n, generator=g, device="cuda").view(B, m)
elif mode == "slot_major":      # what the owner gather leaves at world 1: slab order = (slot, sample)?  here: field-major
    inv = (torch.arange(m, device="cuda")[None, :] * B + torch.arange(B, device="cuda")[:, None]).contiguous()
else:
    inv = torch.arange(n, device="cuda").view(B, m)
x0 = rows[inv.reshape(-1)].view(B, m, D).contiguous()
Ws, hp = [], m
for h in Hs:
    Ws.append(torch.randn((h, hp * m), generator=g, device="cuda") * (1.0 / (hp * m) ** 0.5))
    hp = h
pooled = torch.empty((B, sum(Hs)), device="cuda")
pooled2 = torch.empty_like(pooled)


def plain(x, out):
    xk, off = x, 0
    for k, (W, h) in enumerate(zip(Ws, Hs)):
        xk, _ = ops.cin_layer(x, xk, W, pooled=out[:, off:off + h], want_xout=k + 1 < len(Hs))
        off += h


def t(fn, iters=20, warm=40):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


h = B // 2
cases = {
    "plain, whole batch": lambda: plain(x0, pooled),
    "gather, whole batch": lambda: ops.cin_stack_gather(rows, inv, Ws, pooled2),
    "plain, two halves": lambda: (plain(x0[:h], pooled[:h]), plain(x0[h:], pooled[h:])),
    "gather, two halves": lambda: (ops.cin_stack_gather(rows, inv[:h], Ws, pooled2[:h]), ops.cin_stack_gather(rows, inv[h:], Ws, pooled2[h:])),
}
only = os.environ.get("ONLY")
for name, fn in cases.items():
    if only and only not in name:
        continue
    print("%-22s %.4f ms" % (name, t(fn)), flush=True)
if not only:
    print("bitwise equal:", bool(torch.equal(pooled, pooled2)), " inv order:", mode)
