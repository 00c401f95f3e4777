"""tools/cin_bf3_probe.py (GPU box) -- dir_cin_layer_bf16x3_f32 vs dir_cin_layer_f32: error against the double-accumulating
oracle on a set of shapes, then time per layer of the BASELINE stack (B = 65 536, m = 26, D = 16, H = 128)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dir_amd  # noqa: E402
from dir_amd import ops  # noqa: E402
from oracle import oracle as O  # noqa: E402

dev = torch.device("cuda:0")
dir_amd.load_library()
out = {"errors": [], "timing": []}


def scaled(got, ref):
    return float((np.abs(got.cpu().double().numpy() - ref) / (1 + np.abs(ref))).max())


if "--no-check" not in sys.argv:
    shapes = [(64, 26, 16, 26, 128), (37, 26, 16, 128, 128), (16, 26, 16, 7, 40), (33, 39, 16, 39, 128), (17, 39, 16, 64, 96),
              (10, 22, 16, 50, 128), (20, 26, 16, 10, 128), (10, 26, 16, 200, 200), (6, 26, 16, 8, 190), (5, 26, 16, 4, 270),
              (7, 40, 16, 9, 128), (9, 15, 4, 6, 7), (5, 16, 32, 3, 64), (300, 26, 8, 33, 65), (129, 30, 16, 100, 130)]
    for B, m, D, Hp, H in shapes:
        rng = np.random.default_rng(Hp * 13 + H)
        x0 = (rng.standard_normal((B, m, D)) * 0.5).astype(np.float32)
        xk = (rng.standard_normal((B, Hp, D)) * 0.5).astype(np.float32)
        W = (rng.standard_normal((H, Hp * m)) * (1.0 / np.sqrt(Hp * m))).astype(np.float32)
        ref_x, ref_p = O.cin_layer(x0, xk, W, acc64=True)
        t = [torch.from_numpy(a).to(dev) for a in (x0, xk, W)]
        fx, fp = ops.cin_layer(*t, arith="f32")
        bx, bp = ops.cin_layer(*t, arith="bf16x3")
        torch.cuda.synchronize()
        rec = {"shape": [B, m, D, Hp, H], "f32_x": scaled(fx, ref_x), "bf3_x": scaled(bx, ref_x), "f32_p": scaled(fp, ref_p),
               "bf3_p": scaled(bp, ref_p)}
        out["errors"].append(rec)
        print(rec, flush=True)

B, m, D = 65536, 26, 16
gen = torch.Generator(device=dev).manual_seed(1)
x0 = torch.randn((B, m, D), generator=gen, device=dev) * 0.5
for Hp, H in [(26, 128), (128, 128), (200, 200), (128, 32)]:
    xk = torch.randn((B, Hp, D), generator=gen, device=dev) * 0.5
    W = torch.randn((H, Hp * m), generator=gen, device=dev) / (Hp * m) ** 0.5
    rec = {"Hp": Hp, "H": H}
    for arith in ("f32", "bf16x3"):
        for _ in range(2):
            ops.cin_layer(x0, xk, W, arith=arith)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        e0.record()
        for _ in range(n):
            ops.cin_layer(x0, xk, W, arith=arith)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        rec[arith + "_ms"] = ms
        rec[arith + "_TF_fp32_equiv"] = 2.0 * B * D * Hp * m * H / ms / 1e9
    if "--no-check" not in sys.argv:
        a, _ = ops.cin_layer(x0, xk, W, arith="f32")
        b, _ = ops.cin_layer(x0, xk, W, arith="bf16x3")
        rec["max_scaled_diff_f32_vs_bf3"] = float(((a - b).abs() / (1 + a.abs())).max())
    out["timing"].append(rec)
    print(rec, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "cin_bf3_probe.json"), "w"), indent=1)
