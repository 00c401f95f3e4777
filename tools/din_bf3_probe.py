"""A/B of the DIN forward arithmetics (DIR_DIN_ARITH=bf16x3|f32) on the inputs of test_din_rows_backward_matches_single_kernel:
each mode in its own process (the switch is read once), outputs compared here, both against the float64 oracle."""
import os, subprocess, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def inputs(B=300, T=50, H1=80, H2=40):
    import torch
    K, V = 64, 5000
    g = torch.Generator().manual_seed(B + T)
    table = (torch.randn(V, K, generator=g) * 0.3)
    Ws = [(torch.randn(4 * K, H1, generator=g) * 0.1), (torch.randn(H1, generator=g) * 0.1),
          (torch.randn(H1, H2, generator=g) * 0.2), (torch.randn(H2, generator=g) * 0.1),
          (torch.randn(H2, generator=g) * 0.5), torch.randn(1, generator=g)]
    hist = torch.randint(0, V, (B, T), generator=g)
    hist[torch.rand((B, T), generator=g) < 0.1] = -1
    hl = torch.randint(0, T + 1, (B,), generator=g).to(torch.int32)
    hl[0], hl[1 % B] = T, 0
    cand = torch.randint(0, V, (B,), generator=g)
    cand[3 % B] = -1
    return table, hist, hl, cand, Ws

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    import dir_amd
    from dir_amd import ops
    table, hist, hl, cand, Ws = inputs()
    for normalize in (False, True):
        out, sc = ops.din_attention_pool(table.cuda(), hist.cuda(), hl.cuda(), cand.cuda(), *[w.cuda() for w in Ws], normalize=normalize, want_scores=True)
        np.save(f"/tmp/din_{os.environ['DIR_DIN_ARITH']}_{int(normalize)}_out.npy", out.cpu().numpy())
        np.save(f"/tmp/din_{os.environ['DIR_DIN_ARITH']}_{int(normalize)}_sc.npy", sc.cpu().numpy())
    sys.exit(0)

for a in ("bf16x3", "f32"):
    subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, DIR_DIN_ARITH=a), check=True)
from oracle import oracle
oracle.build()
table, hist, hl, cand, Ws = inputs()
for normalize in (False, True):
    ro, rs = oracle.din_attention_pool(table.numpy(), hist.numpy(), hl.numpy(), cand.numpy(), *[w.numpy() for w in Ws], normalize=normalize, acc64=True)
    for a in ("bf16x3", "f32"):
        o = np.load(f"/tmp/din_{a}_{int(normalize)}_out.npy"); s = np.load(f"/tmp/din_{a}_{int(normalize)}_sc.npy")
        es = np.abs(s - rs); eo = np.abs(o - ro)
        print(f"normalize={normalize} {a}: scores max err {es.max():.3e} (max |s| {np.abs(rs).max():.3f}) at {np.unravel_index(es.argmax(), es.shape)}; out max err {eo.max():.3e} (max |o| {np.abs(ro).max():.3f})")
        bad = np.argwhere(es > 1e-4)
        if len(bad):
            print("  bad rows (sample, j):", bad[:20].tolist(), "lens", [int(hl[b]) for b, _ in bad[:20]])
