"""tools/host_launch_probe.py (GPU box): host time of one ops.gather_fm call (Python + ctypes + hipLaunchKernel), measured on a tiny batch
so that the GPU never back-pressures the host."""
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, ".")
import dir_amd  # noqa: E402
from dir_amd import ops  # noqa: E402

dir_amd.load_library()
F, K, V, B = 26, 16, 1000, 64
ts = ops.TableSet([torch.randn((V, K), device="cuda") for _ in range(F)])
ids = torch.randint(0, V, (B, F), device="cuda")
out = torch.empty((B, F * K), device="cuda")
fm = torch.empty((B, 1), device="cuda")
for _ in range(100):
    ops.gather_fm(ts, ids, out=out, fm=fm)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 2000
for _ in range(n):
    ops.gather_fm(ts, ids, out=out, fm=fm)
t1 = time.perf_counter()
torch.cuda.synchronize()
print("ops.gather_fm host time per call: %.1f us" % ((t1 - t0) / n * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    ops.gather_fm(ts, ids, out=out, fm=fm)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
