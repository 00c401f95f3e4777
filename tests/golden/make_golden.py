#!/usr/bin/env python3
"""make_golden.py -- writes tests/golden/golden_v1.npz and golden_v2_backward.npz.

WHAT THESE VECTORS ARE: seeded inputs (np.random.default_rng(20240607)) and the outputs of oracle/ on
them, plus the hand known-answer cases of SURVEY.md 8(c).  The reference repository holds no golden
vectors for this path and cannot be executed here (TensorFlow 1.x is not installable), so nothing in this
file comes from running the reference: the fixture freezes the ORACLE (regression anchor for oracle/ and
target for the GPU path), it does not pin the oracle to the reference.  Parity stays "unpinned".

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from oracle import np_ref as R  # noqa: E402


def main():
    rng = np.random.default_rng(20240607)
    g = {}
    # ---- DeepFM-shaped: 26 fields, K = 16, B = 64 ---------------------------------------------------
    B, F, K, V = 64, 26, 16, 199
    tables = (rng.standard_normal((F, V, K)) * 0.25).astype(np.float32)
    ids = rng.integers(-1, V, size=(B, F)).astype(np.int64)
    emb = O.embedding_bag(list(tables), ids)
    g["fm_tables"], g["fm_ids"], g["fm_emb"] = tables, ids, emb
    g["fm_logit_f32"] = O.fm_second_order(emb, F, K)
    g["fm_logit_f64"] = R.fm_logit(emb, F, K, np.float64)[:, 0]
    lin_w = (rng.standard_normal((F, V)) * 0.05).astype(np.float32)
    g["lin_w"], g["lin_bias"] = lin_w, np.array([0.125], np.float32)
    g["lin_logit"] = O.linear_sparse_sum(list(lin_w), ids, bias=g["lin_bias"])
    # ---- ragged weighted bags, all three combiners ------------------------------------------------------
    Fb, Kb, Vb, Bb = 4, 8, 61, 33
    btab = rng.standard_normal((Fb, Vb, Kb)).astype(np.float32)
    lens = rng.integers(0, 7, size=Bb * Fb)
    lens[::9] = 0
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    bids = rng.integers(-1, Vb, size=offs[-1]).astype(np.int64)
    bw = rng.uniform(-0.25, 2.0, size=offs[-1]).astype(np.float32)
    g["bag_tables"], g["bag_offsets"], g["bag_ids"], g["bag_weights"] = btab, offs, bids, bw
    for c, name in [(0, "sum"), (1, "mean"), (2, "sqrtn")]:
        g["bag_out_%s" % name] = O.embedding_bag(list(btab), bids, offsets=offs, combiner=c, B=Bb)
        g["bag_out_%s_w" % name] = O.embedding_bag(list(btab), bids, offsets=offs, weights=bw, combiner=c, B=Bb)
    # ---- DCN cross: d = 416 (vector path) and d = 51 (the adult-census schema width, scalar path) --------
    for d in (416, 51):
        x0 = (rng.standard_normal((B, d)) * 0.25).astype(np.float32)
        w = np.clip(rng.standard_normal((3, d)) * 0.1, -0.2, 0.2).astype(np.float32)
        b = np.clip(rng.standard_normal((3, d)) * 0.1, -0.2, 0.2).astype(np.float32)
        g["cross%d_x0" % d], g["cross%d_w" % d], g["cross%d_b" % d] = x0, w, b
        g["cross%d_out" % d] = O.dcn_cross(x0, w, b, acc64=True)
    # ---- DIN (paper-derived) ---------------------------------------------------------------------------
    Vd, Kd, T, H1, H2, Bd = 300, 64, 50, 80, 40, 16
    table = (rng.standard_normal((Vd, Kd)) * 0.125).astype(np.float32)
    hist = rng.integers(-1, Vd, size=(Bd, T)).astype(np.int64)
    hl = rng.integers(0, T + 1, size=Bd).astype(np.int32)
    cand = rng.integers(0, Vd, size=Bd).astype(np.int64)
    W1 = (rng.standard_normal((4 * Kd, H1)) * 0.05).astype(np.float32)
    b1 = (rng.standard_normal(H1) * 0.05).astype(np.float32)
    W2 = (rng.standard_normal((H1, H2)) * 0.1).astype(np.float32)
    b2 = (rng.standard_normal(H2) * 0.05).astype(np.float32)
    W3 = (rng.standard_normal(H2) * 0.2).astype(np.float32)
    b3 = np.array([0.01], np.float32)
    for k, v in dict(table=table, hist=hist, len=hl, cand=cand, W1=W1, b1=b1, W2=W2, b2=b2, W3=W3, b3=b3).items():
        g["din_" + k] = v
    for norm in (0, 1):
        o, s = O.din_attention_pool(table, hist, hl, cand, W1, b1, W2, b2, W3, b3, normalize=bool(norm), acc64=True)
        g["din_out_n%d" % norm], g["din_scores_n%d" % norm] = o, s
    # ---- CIN (paper-derived) ---------------------------------------------------------------------------
    Bc, m, D = 8, 26, 16
    x0 = (rng.standard_normal((Bc, m, D)) * 0.5).astype(np.float32)
    Wa = (rng.standard_normal((32, m * m)) / np.sqrt(m * m)).astype(np.float32)
    Wb = (rng.standard_normal((64, 32 * m)) / np.sqrt(32 * m)).astype(np.float32)
    x1, p1 = O.cin_layer(x0, x0, Wa, acc64=True)
    x2, p2 = O.cin_layer(x0, x1, Wb, acc64=True)
    g["cin_x0"], g["cin_W1"], g["cin_W2"] = x0, Wa, Wb
    g["cin_x1"], g["cin_p1"], g["cin_x2"], g["cin_p2"] = x1, p1, x2, p2
    # ---- integer paths ---------------------------------------------------------------------------------
    keys = np.concatenate([rng.integers(-10**9, 10**9, 200), [0, -1, 7, 10**15, -10**17, 2**63 - 1]]).astype(np.int64)
    g["hash_keys"], g["hash_out_1000"] = keys, R.hash_bucket_int(keys, 1000)
    strs = ["Hello", "TensorFlow", "2.x", "Private", "Self-emp-not-inc", "", "a" * 40, "b" * 100]
    g["hash_strs"] = np.array(strs)
    g["hash_strs_fp64"] = np.array([R.fingerprint64(s.encode()) for s in strs], np.uint64)
    x = rng.uniform(-1, 11, 300).astype(np.float32)
    bd = np.array([0, 1, 2.5, 5, 10], np.float32)
    g["bkt_x"], g["bkt_bd"], g["bkt_out"] = x, bd, R.bucketize(x, bd)
    # ---- hand KATs ----------------------------------------------------------------------
    g["kat_fm_in"] = np.array([[1, 2, 3, 4, -1, .5]], np.float32)
    g["kat_fm_out"] = np.array([10.0], np.float32)
    g["kat_cross_x0"] = np.array([[1, 2]], np.float32)
    g["kat_cross_w"] = np.array([[.5, -1], [.25, .5]], np.float32)
    g["kat_cross_b"] = np.array([[.1, .2], [0, -.1]], np.float32)
    g["kat_cross_out"] = np.array([[-0.9, -1.9]], np.float32)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_v1.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, "%.1f KB" % (os.path.getsize(out) / 1024))
    make_backward()


def make_backward():
    """golden_v2_backward.npz: seeded inputs and the float64 backward / optimiser restatements of oracle/ (np_ref.py,
    dir_oracle.c orc_cin_backward).  Same status as v1: a regression anchor for the oracle and a target for the HIP backward
    kernels, not reference output."""
    rng = np.random.default_rng(20240608)
    g = {}
    B, F, K = 48, 7, 16
    emb = (rng.standard_normal((B, F * K)) * 0.3).astype(np.float32)
    gfm = rng.standard_normal((B, 1)).astype(np.float32)
    gdnn = rng.standard_normal((B, F * K)).astype(np.float32)
    g["fmb_emb"], g["fmb_g"], g["fmb_add"] = emb, gfm, gdnn
    g["fmb_out"] = R.fm_logit_backward(emb, gfm, F, K, gdnn)
    for d, L in ((416, 3), (51, 2)):
        x0 = (rng.standard_normal((B, d)) * 0.25).astype(np.float32)
        w = np.clip(rng.standard_normal((L, d)) * 0.1, -0.2, 0.2).astype(np.float32)
        b = np.clip(rng.standard_normal((L, d)) * 0.1, -0.2, 0.2).astype(np.float32)
        go = rng.standard_normal((B, d)).astype(np.float32)
        gx0, gw, gb = R.cross_network_backward(x0, w, b, go)
        for k, v in dict(x0=x0, w=w, b=b, gout=go, gx0=gx0, gw=gw, gb=gb).items():
            g["crossb%d_%s" % (d, k)] = v
    Bc, m, D, Hp, H = 6, 26, 16, 26, 40
    x0 = (rng.standard_normal((Bc, m, D)) * 0.5).astype(np.float32)
    xk = (rng.standard_normal((Bc, Hp, D)) * 0.5).astype(np.float32)
    W = (rng.standard_normal((H, Hp * m)) / np.sqrt(Hp * m)).astype(np.float32)
    G = (rng.standard_normal((Bc, H, D)) * 0.5).astype(np.float32)
    dW, dxk, dx0 = O.cin_backward(x0, xk, W, G)
    for k, v in dict(x0=x0, xk=xk, W=W, G=G, dW=dW, dxk=dxk, dx0=dx0).items():
        g["cinb_" + k] = v
    Bo, Fo, Ko, Vo = 200, 3, 8, 17
    tabs = [rng.standard_normal((Vo, Ko)).astype(np.float32) for _ in range(Fo)]
    ids = rng.integers(-1, Vo, size=(Bo, Fo)).astype(np.int64)
    grad = (rng.standard_normal((Bo, Fo * Ko)) * 0.5).astype(np.float32)
    w64 = [t.astype(np.float64) for t in tabs]
    a64 = [np.full((Vo, Ko), 0.1) for _ in range(Fo)]
    R.sparse_adagrad_step(w64, a64, ids, grad, 0.05)
    g["opt_tables"], g["opt_ids"], g["opt_grad"] = np.stack(tabs), ids, grad
    g["opt_adagrad_w"], g["opt_adagrad_acc"] = np.stack(w64), np.stack(a64)
    w64 = [t.astype(np.float64) for t in tabs]
    n64 = [np.full((Vo, Ko), 0.1) for _ in range(Fo)]
    z64 = [np.zeros((Vo, Ko)) for _ in range(Fo)]
    R.sparse_ftrl_step(w64, n64, z64, ids, grad, 0.2, l1=0.01, l2=0.05)
    g["opt_ftrl_w"], g["opt_ftrl_n"], g["opt_ftrl_z"] = np.stack(w64), np.stack(n64), np.stack(z64)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_v2_backward.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, "%.1f KB" % (os.path.getsize(out) / 1024))


if __name__ == "__main__":
    main()
