"""tools/gather_out_window_probe.py (GPU box) -- why does the fused gather run 10-12 % slower per row at 4 x B than at B?

Hypothesis: the concat OUTPUT.  At B = 65 536 the [B, 416] fp32 output is 109 MB and is overwritten by every launch: it fits the 256 MiB
Infinity Cache (behind L2, invisible to the L2 <-> fabric PMC counters), so most of its write-back never has to reach HBM before the next
launch overwrites it.  At 4 x B the output is 436 MB and every byte goes to HBM.  The probe separates the two: the same kernel at B with
the output ROTATING through n buffers (n x 109 MB written before a line is reused) and at 4 x B writing one buffer; plus the table-side
check (the id batches are fresh each launch in every case).  Prints one line per case."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dir_amd  # noqa: E402
from dir_amd import ops  # noqa: E402

dir_amd.load_library()
dev = torch.device("cuda:0")
F, V, K = 26, 1000000, 16
gen = torch.Generator(device=dev).manual_seed(1)
slab = torch.randn((F * V, K), generator=gen, device=dev).mul_(0.25)
ts = ops.TableSet([slab[f * V:(f + 1) * V] for f in range(F)])


def run(B, n_out, iters=60, fm=True):
    ids = [torch.randint(0, V, (B, F), generator=gen, device=dev) for _ in range(4)]
    outs = [torch.empty((B, F * K), dtype=torch.float32, device=dev) for _ in range(n_out)]
    fmo = torch.empty((B, 1), dtype=torch.float32, device=dev)
    step = (lambda i: ops.gather_fm(ts, ids[i % 4], out=outs[i % n_out], fm=fmo)) if fm else (lambda i: ops.embedding_bag(ts, ids[i % 4], out=outs[i % n_out]))
    for i in range(8):
        step(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        step(i)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    alg = B * (F * (8 + 8 * K) + 4)
    print("B %7d  out buffers %2d (%5.0f MB written before reuse)  %8.2f us  %7.2f us per 65536 rows  frac %.3f" %
          (B, n_out, n_out * B * F * K * 4 / 1e6, us, us * 65536 / B, alg / (us * 1e-6) / 8e12), flush=True)


for B, n in ((65536, 1), (65536, 2), (65536, 4), (65536, 8), (131072, 1), (131072, 4), (262144, 1), (262144, 2), (32768, 1), (32768, 16)):
    run(B, n)
