"""tools/din_len_sweep.py (GPU box) -- dir_din_attention_pool_f32 at FIXED history lengths: time = per-sample overhead + per-row-tile cost.
DIR_DIN_WAVE=0 selects the round-1 kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dir_amd
from dir_amd import ops
dir_amd.load_library()
B, T, K, V, H1, H2 = 65536, 64, 64, 10_000_000, 80, 40
g = torch.Generator(device="cuda").manual_seed(1)
table = torch.randn((V, K), generator=g, device="cuda") * 0.125
hist = torch.randint(0, V, (B, T), generator=g, device="cuda")
cand = torch.randint(0, V, (B,), generator=g, device="cuda")
W1 = torch.randn((4 * K, H1), generator=g, device="cuda") * 0.05
W2 = torch.randn((H1, H2), generator=g, device="cuda") * 0.1
W3 = torch.randn((H2,), generator=g, device="cuda") * 0.1
b1, b2, b3 = torch.zeros(H1, device="cuda"), torch.zeros(H2, device="cuda"), torch.zeros(1, device="cuda")
for L in (1, 16, 17, 32, 33, 48, 49, 64):
    hl = torch.full((B,), L, dtype=torch.int32, device="cuda")
    for _ in range(3):
        ops.din_attention_pool(table, hist, hl, cand, W1, b1, W2, b2, W3, b3, normalize=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.din_attention_pool(table, hist, hl, cand, W1, b1, W2, b2, W3, b3, normalize=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tiles = (L + 15) // 16
    mf = B * tiles * 220 * 2048 / (ms * 1e-3) / 1e12
    print("len %2d tiles %d: %.3f ms  (%.1f TFLOP/s executed = %.2f of the fp32 MFMA peak)" % (L, tiles, ms, mf, mf / 157.3))
