"""E673 debug build: out[b][lane] = the lane's own sum of the per-sample term pieces it holds (pass 0, row tile 0).  Which lanes differ between runs?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dir_amd
from dir_amd import ops
B, T, K, V, H1, H2 = 65536, 50, 64, 10_000_000, 80, 40
g = torch.Generator(device="cuda").manual_seed(4)
table = torch.randn((V, K), generator=g, device="cuda") * 0.125
hist = torch.randint(0, V, (B, T), generator=g, device="cuda")
hl = torch.randint(1, T + 1, (B,), generator=g, device="cuda", dtype=torch.int32)
cand = torch.randint(0, V, (B,), generator=g, device="cuda")
Ws = [torch.randn((4 * K, H1), generator=g, device="cuda") * 0.1, torch.randn((H1,), generator=g, device="cuda") * 0.1,
      torch.randn((H1, H2), generator=g, device="cuda") * 0.2, torch.randn((H2,), generator=g, device="cuda") * 0.1,
      torch.randn((H2,), generator=g, device="cuda") * 0.5, torch.randn((1,), generator=g, device="cuda")]
runs = []
for it in range(12):
    o, s = ops.din_attention_pool(table, hist, hl, cand, *Ws, normalize=False, want_scores=True)
    runs.append(o.clone())
    if it == 0: S = []
    S.append(s[:, 0].clone())
R = torch.stack(runs)            # [12, B, 64]
S = torch.stack(S)
print("row-0 score of samples 0..9 over the runs:")
for b in range(10):
    vals, cnt = torch.unique(S[:, b], return_counts=True)
    print("  sample %d: %s" % (b, ", ".join("%.7f x%d" % (float(v), int(c)) for v, c in zip(vals, cnt))))
import numpy as np
W1 = Ws[0].double().cpu().numpy(); b1 = Ws[1].double().cpu().numpy()
NL = -1.4426950408889634
for b in range(10):
    a = table[cand[b]].double().cpu().numpy()
    c = ((a @ (W1[K:2 * K] - W1[2 * K:3 * K])) + b1) * NL
    want = np.array([sum(c[16 * mt + 4 * kk + g] for mt in range(5) for g in range(4)) for kk in range(4)])
    stage = os.environ.get("DW_STAGE", "0")
    if stage == "1": want = np.array([want[0] + want[1], want[0] + want[1], want[2] + want[3], want[2] + want[3]])
    if stage == "2": want = np.full(4, want.sum())
    print("sample %d len %d: expected per lane group kk: %s" % (b, int(hl[b]), np.round(want, 6).tolist()))
    seen = set()
    for it in range(12):
        v = R[it, b].cpu().numpy().reshape(4, 16)
        key = v.tobytes()
        if key in seen: continue
        seen.add(key)
        d = v - want[:, None]
        print("   run %d: max |lane - expected| per kk group %s; lanes off by > 1e-4: %s" % (it, np.round(np.abs(d).max(1), 6).tolist(),
              [(int(kk), int(r), round(float(d[kk, r]), 5)) for kk, r in zip(*np.nonzero(np.abs(d) > 1e-4))][:20]))
