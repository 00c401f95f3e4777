"""tools/tower_probe.py (GPU box): device time of dir_tower_bf16x3_f32 at the DeepFM tower shape vs the per-layer kernels, and where the
host time of DeepFM.forward_ids goes (cProfile)."""
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, ".")
import dir_amd  # noqa: E402
from dir_amd import ops  # noqa: E402

dir_amd.load_library()
gen = torch.Generator(device="cuda").manual_seed(1)
M, dims = 65536, [416, 400, 400, 400]
x = torch.randn((M, dims[0]), generator=gen, device="cuda") * 0.25
Ws = [torch.randn((dims[i + 1], dims[i]), generator=gen, device="cuda") / dims[i] ** 0.5 for i in range(3)]
bs = [torch.randn((dims[i + 1],), generator=gen, device="cuda") * 0.1 for i in range(3)]
hw, hb = torch.randn((1, 400), generator=gen, device="cuda") / 20, torch.full((1,), 0.1, device="cuda")


def dev_us(fn, n=30):
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


out = torch.empty((M, 1), device="cuda")
print("tower + head      : %.1f us" % dev_us(lambda: ops.tower(x, Ws, bs, head=(hw, hb), out=out)))
y = torch.empty((M, 400), device="cuda")
print("tower, no head    : %.1f us" % dev_us(lambda: ops.tower(x, Ws, bs, out=y)))
print("tower, 1 layer    : %.1f us" % dev_us(lambda: ops.tower(x, Ws[:1], bs[:1], out=y)))
print("tower, 2 layers   : %.1f us" % dev_us(lambda: ops.tower(x, Ws[:2], bs[:2], out=y)))
from dir_amd.dense import pack_weight  # noqa: E402
Wp = [pack_weight(w) for w in Ws]
ys = [torch.empty((M, 400), device="cuda") for _ in range(3)]


def layers():
    h = x
    for l in range(3):
        h = ops.dense(h, Wp[l], bs[l], relu=True, out=ys[l])


print("3 x dense_bf3_k   : %.1f us" % dev_us(layers))

from dir_amd.deepfm import DeepFM  # noqa: E402
from dir_amd import feature_column as fc  # noqa: E402
F, K, V = 26, 16, 100000
cats = [fc.categorical_column_with_identity("C%d" % i, V) for i in range(F)]
model = DeepFM(linear_feature_columns=cats, dnn_feature_columns=[fc.embedding_column(c, K) for c in cats], dnn_hidden_units=[400, 400, 400],
               fm_embedding_size=K).cuda()
ids = torch.randint(0, V, (M, F), device="cuda")
with torch.no_grad():
    for _ in range(3):
        model.forward_ids(ids, ids)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        model.forward_ids(ids, ids)
    host = (time.perf_counter() - t0) / 50 * 1e6
    torch.cuda.synchronize()
    print("forward_ids host issue time: %.1f us; device: %.1f us" % (host, dev_us(lambda: model.forward_ids(ids, ids))))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        model.forward_ids(ids, ids)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
