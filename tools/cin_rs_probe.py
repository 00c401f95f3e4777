"""tools/cin_rs_probe.py (GPU box): the CIN forward kernels of cfg 5 one by one, same process -- what the scaled fp16 x 2 forms of round 5
cost against round 4's unscaled ones: layer 1 (pairs), layer 2 (unscaled / row-scaled with the rows scanned / with the producer's row
maxima), the pooled last layer's Z with and without its row maxima + the dense product behind a max pass or behind those maxima."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dir_amd  # noqa: E402
from dir_amd import ops  # noqa: E402

dir_amd.load_library()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
B, m, D, H = 65536, 26, 16, 128
x0 = torch.randn((B, m, D), generator=g, device=dev) * 0.25
W1 = torch.randn((H, m * m), generator=g, device=dev) / m
W2 = torch.randn((H, H * m), generator=g, device=dev) / (H * m) ** 0.5
x1, _ = ops.cin_layer(x0, x0, W1)
x1p = x1.clone()                     # no row maxima attached


def t(name, fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-70s %9.1f us" % (name, e0.elapsed_time(e1) * 1e3 / n), flush=True)


pooled = torch.empty((B, H), device=dev)
for rep in range(2):
    t("layer 1 pairs, scaled fp16 x 2 (+ row maxima out)", lambda: ops.cin_layer(x0, x0, W1))
    t("layer 1 pairs, bf16 x 3", lambda: ops.cin_layer(x0, x0, W1, arith="bf16x3"))
    t("layer 2 unscaled fp16 x 2", lambda: ops.cin_layer(x0, x1p, W2, arith="f16x2"))
    t("layer 2 row-scaled, rows scanned", lambda: ops.cin_layer(x0, x1p, W2, arith="f16x2_grad"))
    t("layer 2 row-scaled, producer's row maxima", lambda: ops.cin_layer(x0, x1, W2, arith="f16x2_grad"))
    t("layer 2 bf16 x 3", lambda: ops.cin_layer(x0, x1p, W2, arith="bf16x3"))
    t("layer 3 pooled (Z + bits, dense rows)", lambda: ops.cin_layer(x0, x1, W2, pooled=pooled, want_xout=False))
    t("  Z alone", lambda: ops.cin_pool_z(x0, x1))
    t("  Z + row maxima", lambda: ops.cin_pool_z(x0, x1, want_bits=True))
    Z, zb = ops.cin_pool_z(x0, x1, want_bits=True)
    t("  dense(Z) behind its own max pass", lambda: ops.dense(Z, W2, out=pooled))
    t("  dense(Z, row_bits)", lambda: ops.dense(Z, W2, out=pooled, row_bits=zb))
    t("  dense(Z) unscaled fp16 x 2", lambda: ops.dense(Z, W2, out=pooled, arith="f16x2"))

# ---- the four combinations of the row-scaled layer: rows scanned | producer's maxima  x  output maxima written | not
import ctypes
from dir_amd import _lib
lib = _lib.load()
nbytes = int(lib.dir_cin_bf16x3_workspace_bytes(m, H, H))
ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
xo = torch.empty((B, H, D), device=dev)
ob = torch.empty(B * D, dtype=torch.int32, device=dev)
ib = ops._row_bits_hint(x1, B * D)
p = lambda t_: ctypes.c_void_p(t_.data_ptr()) if t_ is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, i_, o_ in (("scan, no out bits", None, None), ("scan, out bits", None, ob), ("hint, no out bits", ib, None), ("hint, out bits", ib, ob)):
    t("rows kernel: " + name, lambda: lib.dir_cin_layer_rows_f16x2_f32(p(x0), p(x1), p(W2), m, H, H, D, B, p(xo), p(pooled), pooled.stride(0), p(ws), nbytes,
                                                                      p(i_), p(o_), st))
t("layer 2 unscaled fp16 x 2 (again)", lambda: ops.cin_layer(x0, x1p, W2, arith="f16x2"))
