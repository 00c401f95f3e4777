"""tools/cin_power_probe.py (GPU box): is the row-scaled fp16 x 2 CIN layer slower than the unscaled one because of what it computes, or
because of what its operands look like?  The UNSCALED kernel on the same xk multiplied by 2^k: at k = 0 (embedding scale) the second fp16
piece of most elements is a subnormal with one or two significant bits; at k = 12 it is a normal number with eleven."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dir_amd  # noqa: E402
from dir_amd import ops  # noqa: E402

dir_amd.load_library()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
B, m, D, H = 65536, 26, 16, 128
x0 = torch.randn((B, m, D), generator=g, device=dev) * 0.25
W2 = torch.randn((H, H * m), generator=g, device=dev) / (H * m) ** 0.5
xk = torch.randn((B, H, D), generator=g, device=dev) * 0.3


def t(name, fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-70s %9.1f us" % (name, e0.elapsed_time(e1) * 1e3 / n), flush=True)


for k in (0, 4, 8, 12, 0):
    xs = xk * 2.0 ** k
    Ws = W2 * 2.0 ** (k if k else 0)
    t("unscaled fp16 x 2, xk * 2^%d, W * 2^%d" % (k, k), lambda: ops.cin_layer(x0, xs, Ws, arith="f16x2"))
    t("row-scaled fp16 x 2, same operands", lambda: ops.cin_layer(x0, xs, Ws, arith="f16x2_grad"))
xz = xk.clone()
xz.view(torch.int32).bitwise_and_(-8192)              # xk with 13 low mantissa bits cleared: exact in ONE fp16 piece at embedding scale
t("unscaled, xk representable in one fp16 piece (second piece zero)", lambda: ops.cin_layer(x0, xz, W2, arith="f16x2"))
t("row-scaled, the same", lambda: ops.cin_layer(x0, xz, W2, arith="f16x2_grad"))
