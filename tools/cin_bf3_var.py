"""tools/cin_bf3_var.py (GPU box) -- A/B of the cin_bf3_k build variants (DIR_BF3_VAR) in one process, interleaved, on the 128 x 128 layer
of the BASELINE stack; prints ms per variant per round (and a zero-operand run: the clock-bound ceiling of the same instruction stream)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dir_amd  # noqa: E402
from dir_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
dir_amd.load_library()
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3".split(","))]
B, m, D, Hp, H = 65536, 26, 16, 128, 128
gen = torch.Generator(device=dev).manual_seed(1)
x0 = torch.randn((B, m, D), generator=gen, device=dev) * 0.5
xk = torch.randn((B, Hp, D), generator=gen, device=dev) * 0.5
W = torch.randn((H, Hp * m), generator=gen, device=dev) / (Hp * m) ** 0.5


def timed(arith, n=8, **kw):
    for _ in range(2):
        ops.cin_layer(kw.get("x0", x0), kw.get("xk", xk), kw.get("W", W), arith=arith)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.cin_layer(kw.get("x0", x0), kw.get("xk", xk), kw.get("W", W), arith=arith)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


ref = None
for rnd in range(3):
    line = ["round %d: f32 %.3f" % (rnd, timed("f32", n=4))]
    for v in variants:
        os.environ["DIR_BF3_VAR"] = str(v)
        line.append("var%d %.3f" % (v, timed("bf16x3")))
        out, _ = ops.cin_layer(x0, xk, W, arith="bf16x3")
        if ref is None:
            ref = out
        else:
            line.append("(maxdiff %.1e)" % float((out - ref).abs().max()))
    print("  ".join(line), flush=True)
z = torch.zeros_like(xk)
for v in variants:
    os.environ["DIR_BF3_VAR"] = str(v)
    print("zeros var%d %.3f ms" % (v, timed("bf16x3", xk=z, W=torch.zeros_like(W))), flush=True)
