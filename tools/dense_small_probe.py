"""tools/dense_small_probe.py (GPU box): one dense layer at small M -- dir_dense_small_f32 against dir_dense_f32 (128-row workgroups) and
the library (torch.nn.functional.linear + relu), per launch from a HIP graph of 20 launches (no host gaps), and the error against float64."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dir_amd  # noqa: E402
from dir_amd import ops  # noqa: E402

dir_amd.load_library()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)


def graph_us(fn, n=20, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n):
            fn()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (n * reps)


for Kd, N in ((416, 400), (400, 400), (432, 1024), (1024, 1024), (360, 200), (200, 80)):
    W = torch.randn((N, Kd), generator=g, device=dev) * 0.05
    b = torch.randn((N,), generator=g, device=dev) * 0.1
    for M in (100, 256, 512, 1024, 2048, 4096, 6144, 8192, 12288):
        x = torch.randn((M, Kd), generator=g, device=dev) * 0.25
        out = torch.empty((M, N), device=dev)
        ref = torch.relu(x.double() @ W.double().t() + b.double())
        res = {}
        for name, small_rows, mid_rows in (("small", 1 << 30, 0), ("mid", 0, 1 << 30), ("dense_k", 0, 0)):
            ops.DENSE_SMALL_ROWS, ops.DENSE_MID_ROWS = small_rows, mid_rows
            fn = lambda: ops.dense(x, W, b, relu=True, out=out, arith="f32")      # noqa: E731
            fn()
            err = float(((out.double() - ref).abs() / (1 + ref.abs())).max())
            res[name] = (graph_us(fn), err)
        lib = lambda: torch.relu_(torch.nn.functional.linear(x, W, b))          # noqa: E731
        res["library"] = (graph_us(lib), float(((lib().double() - ref).abs() / (1 + ref.abs())).max()))
        print("%4d x %4d  M %5d   small %6.2f us (%.1e)   mid %6.2f us (%.1e)   dense_k %6.2f us (%.1e)   library %6.2f us (%.1e)" %
              (Kd, N, M, res["small"][0], res["small"][1], res["mid"][0], res["mid"][1], res["dense_k"][0], res["dense_k"][1], res["library"][0],
               res["library"][1]), flush=True)
