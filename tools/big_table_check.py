"""Development check: a 10^8-row x 16 table (6.4 GB, BASELINE configs[4]'s logical table) shared by 26 slots on one GPU --
ids near the top of the table exercise > 2^32-byte offsets in the gather, the fused FM and the multi-hot bag kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import dir_amd  # noqa: E402,F401
from dir_amd import ops  # noqa: E402

V, K, F, B = 100_000_000, 16, 26, 8192
g = torch.Generator(device="cuda").manual_seed(0)
table = torch.empty((V, K), device="cuda")
for s in range(0, V, 10_000_000):                      # fill in slices (keeps peak memory at one table)
    table[s:s + 10_000_000].normal_(0, 0.25, generator=g)
ts = ops.TableSet([table] * F)
ids = torch.randint(V - 5_000_000, V, (B, F), generator=g, device="cuda")
ids[0, 0] = V - 1
ids[1, 1] = 0
emb, fm = ops.gather_fm(ts, ids)
ref = table[ids.reshape(-1)].reshape(B, F * K)
assert torch.equal(emb, ref), "gather mismatch"
e = ref.view(B, F, K)
fm_ref = 0.5 * ((e.sum(1) ** 2 - (e ** 2).sum(1)).sum(1, keepdim=True))
assert torch.allclose(fm, fm_ref, rtol=1e-4, atol=1e-4), "fm mismatch"
offs = torch.arange(0, B * F * 2 + 1, 2, device="cuda", dtype=torch.int64)
vals = torch.randint(V - 1000, V, (B * F * 2,), generator=g, device="cuda")
bag = ops.embedding_bag(ts, vals, offs, None, combiner="sum")
ref_bag = (table[vals[0::2]] + table[vals[1::2]]).reshape(B, F * K)
assert torch.equal(bag, ref_bag), "bag mismatch"
print("big table ok: %.1f GB table, ids up to %d" % (table.numel() * 4 / 1e9, int(ids.max())))
