"""tools/f16x2_probe.py (CPU, NumPy): the representation / product error of the "fp16 x 2" split (two fp16 pieces per operand, the three
piece products of weight >= 2^-11) against "bf16 x 3" (three bf16 pieces, six products) on the operand scales of the kernels that use
them -- piece products and their sums taken in float64, so the numbers are the SPLITS' errors, not the fp32 accumulation's.
-> profiles/r04_f16x2_probe.txt"""
import numpy as np

rng = np.random.default_rng(0)


def split_f16x2(x):
    x = x.astype(np.float32)
    h0 = x.astype(np.float16)
    h1 = (x - h0.astype(np.float32)).astype(np.float32).astype(np.float16)
    return h0, h1


def bf16(v):
    u = v.astype(np.float32).view(np.uint32)
    return ((u + (((u >> 16) & 1) + 0x7fff)) & 0xffff0000).view(np.float32)


def split_bf16x3(x):
    p0 = bf16(x)
    r = (x - p0).astype(np.float32)
    p1 = bf16(r)
    return p0, p1, bf16((r - p1).astype(np.float32))


def mm(a, b):
    return a.astype(np.float64) @ b.astype(np.float64)


def mm_f16x2(A, B):
    a0, a1 = split_f16x2(A)
    b0, b1 = split_f16x2(B)
    return mm(a1, b0) + mm(a0, b1) + mm(a0, b0)


def mm_bf16x3(A, B):
    a, b = split_bf16x3(A), split_bf16x3(B)
    return mm(a[0], b[2]) + mm(a[2], b[0]) + mm(a[1], b[1]) + mm(a[0], b[1]) + mm(a[1], b[0]) + mm(a[0], b[0])


def report(name, A, B):
    ref = mm(A, B)
    for nm, fn in (("f16x2", mm_f16x2), ("bf16x3", mm_bf16x3)):
        got = fn(A, B)
        scaled = np.abs(got - ref) / (1 + np.abs(ref))
        rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)
        print("%-44s %-7s max |err| / (1 + |ref|) %.2e   median rel %.2e   median |ref| %.2e" % (name, nm, scaled.max(), np.median(rel), np.median(np.abs(ref))))


h = rng.standard_normal((4096, 64)).astype(np.float32) * 0.125
a = rng.standard_normal((4096, 64)).astype(np.float32) * 0.125
U = np.concatenate([h, a, h - a, h * a], 1)
W1 = (rng.standard_normal((256, 80)) * 0.05).astype(np.float32)
report("DIN layer 1 (table sigma .125, W .05)", U, W1)
report("DIN layer 1, table x 1e-3", U * 1e-3, W1)
report("DIN layer 1, table x 1e+3", U * 1e3, W1)
xk = rng.standard_normal((2048, 128)).astype(np.float32) * 0.25
x0 = rng.standard_normal((2048, 26)).astype(np.float32) * 0.25
Z = (xk[:, :, None] * x0[:, None, :]).reshape(2048, -1).astype(np.float32)
W = (rng.standard_normal((3328, 128)) / np.sqrt(3328)).astype(np.float32)
report("CIN layer as Z x W (reduction 3328)", Z, W)
X = np.maximum(rng.standard_normal((4096, 416)).astype(np.float32), 0) * 0.5
Wd = (rng.standard_normal((416, 400)) / np.sqrt(416)).astype(np.float32)
report("tower layer 416 -> 400 (ReLU-scale input)", X, Wd)
G = (rng.standard_normal((4096, 128)) * 1e-5).astype(np.float32)
report("gradient-scale operand (1e-5) x W", G, (rng.standard_normal((128, 400)) * 0.05).astype(np.float32))
big = X.copy()
big[:, 3] = 99999.0
with np.errstate(over="ignore", invalid="ignore"):
    print("an input column of 99 999 (adult-census capital_gain): f16x2 finite = %s, bf16x3 finite = %s" %
          (bool(np.isfinite(mm_f16x2(big, Wd)).all()), bool(np.isfinite(mm_bf16x3(big, Wd)).all())))
