"""tools/dedup_probe.py (GPU box): time of dir_shard_bucket_cap vs dir_shard_bucket_cap_dedup at 65 536 x 26 ids (uniform and Zipf(1.05),
P = 8 owners) and the share of the entries that still travels after the per-tile de-duplication."""
import sys

import torch

sys.path.insert(0, ".")
import dir_amd  # noqa: E402
from dir_amd import ops  # noqa: E402

dir_amd.load_library()
gen = torch.Generator(device="cuda").manual_seed(3)
B, F, V, P = 65536, 26, 1000000, 8
vdev = torch.full((F,), V, dtype=torch.int64, device="cuda")


def us(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for dist in ("uniform", "zipf"):
    if dist == "uniform":
        ids = torch.randint(0, V, (B, F), generator=gen, device="cuda")
    else:
        u = torch.rand((B, F), generator=gen, device="cuda", dtype=torch.float64)
        a = 1.05
        ids = ((V ** (1 - a) - 1) * u + 1).pow(1 / (1 - a)).floor().long().clamp_(1, V) - 1
    n = B * F
    cap = n // P * 2 if dist == "zipf" else n // P + 4096
    cap = max(cap, n)  # roomy: nothing overflows
    payload = torch.empty(P * (cap + 1), dtype=torch.int64, device="cuda")
    inv = torch.empty(n, dtype=torch.int64, device="cuda")
    counts = torch.empty(P, dtype=torch.int64, device="cuda")
    over = torch.zeros(1, dtype=torch.int32, device="cuda")
    ws = torch.zeros(128, dtype=torch.int32, device="cuda")
    flat = ids.reshape(-1)
    t0 = us(lambda: ops.shard_bucket_cap(flat, vdev, P, cap, payload, inv, counts, over, ws))
    t1 = us(lambda: ops.shard_bucket_cap_dedup(ids, vdev, P, cap, payload, inv, counts, over, ws))
    sent = int(counts.sum())
    exact = sum(int(torch.unique(ids[:, f]).numel()) for f in range(F))
    fb = ids.t().contiguous().t()
    t2 = us(lambda: ops.shard_bucket_cap_dedup(fb, vdev, P, cap, payload, inv, counts, over, ws))
    print("%-8s bucket_cap %.1f us | bucket_cap_dedup %.1f us ([B,F] ids) / %.1f us ([F,B] ids) | travel %.3f of the entries (an exact unique: %.3f)"
          % (dist, t0, t1, t2, sent / n, exact / n))
