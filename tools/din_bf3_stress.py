"""tools/din_bf3_stress.py (GPU box) -- run-to-run bitwise equality of the DIN forward at BASELINE.json configs[3] size (65536 samples, T 50,
K 64, 10 M-row table): R reruns on identical inputs, samples whose pooled output or scores differ from the first run are counted and the
first few are described (which row tiles, how far off).  DIR_HIP_LIBRARY selects an A/B build of the library, DIR_DIN_ARITH the arithmetic."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dir_amd
from dir_amd import ops
R = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B, T, K, V, H1, H2 = 65536, 50, 64, 10_000_000, 80, 40
g = torch.Generator(device="cuda").manual_seed(4)
table = torch.randn((V, K), generator=g, device="cuda") * 0.125
hist = torch.randint(0, V, (B, T), generator=g, device="cuda")
hl = torch.randint(1, T + 1, (B,), generator=g, device="cuda", dtype=torch.int32)
cand = torch.randint(0, V, (B,), generator=g, device="cuda")
Ws = [torch.randn((4 * K, H1), generator=g, device="cuda") * 0.1, torch.randn((H1,), generator=g, device="cuda") * 0.1,
      torch.randn((H1, H2), generator=g, device="cuda") * 0.2, torch.randn((H2,), generator=g, device="cuda") * 0.1,
      torch.randn((H2,), generator=g, device="cuda") * 0.5, torch.randn((1,), generator=g, device="cuda")]
for normalize in (False, True):
    f = lambda: ops.din_attention_pool(table, hist, hl, cand, *Ws, normalize=normalize, want_scores=True)
    o0, s0 = f()
    o0, s0 = o0.clone(), s0.clone()
    runs, samples, shown, consec = 0, 0, 0, 0
    prev = s0
    for it in range(R):
        o, s = f()
        consec += int((s != prev).any(1).sum())
        prev = s.clone()
        bad = ((o != o0).any(1) | (s != s0).any(1)).nonzero().flatten()
        if bad.numel():
            runs += 1
            samples += int(bad.numel())
            for b in bad[:3].tolist():
                if shown < 6:
                    d = (s[b] - s0[b])
                    rows = (d != 0).nonzero().flatten().tolist()
                    print("  run %d sample %d len %d: rows differing %s diffs %s" % (it, b, int(hl[b]), rows[:20], [float("%.3g" % d[r]) for r in rows[:6]]), flush=True)
                    shown += 1
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        f()
    t1.record(); torch.cuda.synchronize()
    print("  samples differing between consecutive reruns: %d in total" % consec)
    print("lib %s arith %s normalize %s: %d / %d reruns differ from the first, %d differing samples in total; %.3f ms per call"
          % (os.path.basename(dir_amd.library_path()), os.environ.get("DIR_DIN_ARITH", "bf16x3"), normalize, runs, R, samples, t0.elapsed_time(t1) / 20), flush=True)
