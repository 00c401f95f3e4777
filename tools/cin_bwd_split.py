"""tools/cin_bwd_split.py (GPU box) -- dir_cin_dw_f32 and dir_cin_dx_f32 timed separately per shape (development tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, dir_amd
from dir_amd import ops
dir_amd.load_library()
g = torch.Generator(device="cuda").manual_seed(0)
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B = 65536
for m, D, Hp, H in [(26, 16, 200, 200), (26, 16, 128, 128), (26, 16, 100, 100), (26, 16, 64, 64), (26, 16, 128, 32)]:
    x0 = torch.randn((B, m, D), generator=g, device="cuda") * 0.25
    xk = torch.randn((B, Hp, D), generator=g, device="cuda") * 0.25
    W = torch.randn((H, Hp * m), generator=g, device="cuda") / (Hp * m) ** 0.5
    G = torch.randn((B, H, D), generator=g, device="cuda") * 0.1
    fl = 2.0 * B * D * Hp * m * H
    tw = timeit(lambda: ops.cin_dw(x0, xk, G)); tx = timeit(lambda: ops.cin_dx(x0, xk, W, G)); tf = timeit(lambda: ops.cin_layer(x0, xk, W))
    print("m %d D %d Hp %d H %d | fwd %.2f ms %.1f TF | dW %.2f ms %.1f TF | dx %.2f ms %.1f TF" % (m, D, Hp, H, tf, fl / tf / 1e9, tw, fl / tw / 1e9, tx, fl / tx / 1e9))
    del x0, xk, W, G
