"""Distinct values of one sample's score over reruns (debug builds of din_wave.hip whose 'score' is an intermediate sum)."""
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
mode = os.environ.get("PROBE_MODE", "")
if mode == "zero_hist_rows":            # every history row all zeros: the MFMA results are exact zeros
    table_h = torch.zeros_like(table)
    hist_t = table_h
elif mode == "small_rows":
    hist_t = table * 1e-3
else:
    hist_t = table
if mode:                                # candidate rows must stay as they are: give the history its own table copy appended behind
    table = torch.cat([table[:1000000], hist_t[:1000000]])
    hist = hist % 1000000 + 1000000
    cand = cand % 1000000
runs = []
for it in range(30):
    o, s = ops.din_attention_pool(table, hist, hl, cand, *Ws, normalize=False, want_scores=True)
    runs.append(s[:, :33:16].clone())            # row 0 of tiles 0, 1, 2
R = torch.stack(runs)                               # [30, B, 3]
for b in range(12):
    for t in range(3):
        vals, cnt = torch.unique(R[:, b, t], return_counts=True)
        if vals.numel() > 1:
            print("sample %d len %d tile %d: %s" % (b, int(hl[b]), t, ", ".join("%.7f x%d" % (float(v), int(c)) for v, c in zip(vals, cnt))))
nd = (R != R[0]).any(0)
print("samples with any differing tile-0 row: %d, tile 1: %d, tile 2: %d" % (int(nd[:, 0].sum()), int(nd[:, 1].sum()), int(nd[:, 2].sum())))
both = (nd[:, 0] & nd[:, 1]).sum()
print("tile 0 and tile 1 both: %d" % int(both))
# per-sample number of distinct values in tile 0
k = torch.tensor([torch.unique(R[:, b, 0]).numel() for b in range(2000)])
print("distinct tile-0 values per sample (first 2000 samples): histogram", torch.bincount(k).tolist())
# which part of the per-sample term is missing in the alternative value?  (E161 builds: score = sum_m c[m] + b3, c scaled by -log2 e)
import numpy as np
W1 = Ws[0].double().cpu().numpy(); b1 = Ws[1].double().cpu().numpy(); b3 = float(Ws[5])
NL = -1.4426950408889634
for b in range(8):
    a = table[cand[b]].double().cpu().numpy()
    c = ((a @ (W1[K:2 * K] - W1[2 * K:3 * K])) + b1) * NL
    true = c.sum() + b3
    vals = torch.unique(R[:, b, 0]).tolist()
    msg = "sample %d: float64 sum %.7f; seen %s" % (b, true, ["%.7f" % v for v in vals])
    for v in vals:
        d = v - true
        if abs(d) < 1e-4: continue
        best = None
        for mt in range(5):
            for g in range(4):
                part = sum(c[16 * mt + 4 * kk + g] for kk in range(4))
                for mt2 in list(range(5)) + [None]:
                    alt = -part + (sum(c[16 * mt2 + 4 * kk + g] for kk in range(4)) if mt2 is not None else 0.0)
                    if best is None or abs(alt - d) < best[0]: best = (abs(alt - d), mt, g, mt2)
            # whole float4 of one mt replaced
            part4 = sum(c[16 * mt + 4 * kk + g] for kk in range(4) for g in range(4))
            for mt2 in list(range(5)) + [None]:
                alt = -part4 + (sum(c[16 * mt2 + 4 * kk + g] for kk in range(4) for g in range(4)) if mt2 is not None else 0.0)
                if abs(alt - d) < best[0]: best = (abs(alt - d), mt, "all4", mt2)
        msg += "; value %.7f is off by %.7f: best explanation (residual %.2e): tile mt=%s element %s replaced by %s" % (v, d, best[0], best[1], best[2], "zero" if best[3] is None else "mt=%s" % best[3])
    print(msg)
# regression: is (alternative - true) a fixed linear combination of the sample's 80 c values?
nS = 4000
A = table[cand[:nS]].double().cpu().numpy()
C = ((A @ (W1[K:2 * K] - W1[2 * K:3 * K])) + b1) * NL            # [nS, 80]
rows, dl = [], []
for b in range(nS):
    vals = torch.unique(R[:, b, 0]).tolist()
    if len(vals) != 2: continue
    true = C[b].sum() + b3
    alt = vals[0] if abs(vals[1] - true) < abs(vals[0] - true) else vals[1]
    rows.append(b); dl.append(alt - true)
X = np.concatenate([C[rows], np.ones((len(rows), 1))], 1)
w, res, rk, sv = np.linalg.lstsq(X, np.array(dl), rcond=None)
pred = X @ w
print("regression over %d samples: residual rms %.3e (delta rms %.3e)" % (len(rows), np.sqrt(np.mean((pred - dl) ** 2)), np.sqrt(np.mean(np.array(dl) ** 2))))
print("weights on c[m] (rounded):", np.round(w[:80], 2).tolist())
print("intercept %.5f" % w[80])
# delta as a linear function of the candidate row a (64 values, unique), then a sparse fit over the columns of the (Wa - Wd) image
Xa = np.concatenate([A[rows], np.ones((len(rows), 1))], 1)
v, *_ = np.linalg.lstsq(Xa, np.array(dl), rcond=None)
print("regression on the candidate row: residual rms %.3e" % np.sqrt(np.mean((Xa @ v - dl) ** 2)))
D = np.concatenate([(W1[K:2 * K] - W1[2 * K:3 * K]) * NL, (b1 * NL)[None, :]], 0)      # [65, 80]: column m = how c[m] depends on (a, 1)
resid = v.copy(); chosen = []
for step in range(8):
    # best single column (with free coefficient)
    coef = (D * resid[:, None]).sum(0) / (D * D).sum(0)
    gain = coef ** 2 * (D * D).sum(0)
    m = int(np.argmax(gain))
    chosen.append(m)
    sol, *_ = np.linalg.lstsq(D[:, chosen], v, rcond=None)
    resid = v - D[:, chosen] @ sol
    print("  columns %s coefficients %s -> residual norm %.3e (of %.3e)" % (chosen, np.round(sol, 4).tolist(), np.linalg.norm(resid), np.linalg.norm(v)))
    if np.linalg.norm(resid) < 1e-6 * np.linalg.norm(v) + 1e-7: break
