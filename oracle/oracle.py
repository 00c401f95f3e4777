"""oracle.py -- ctypes binding of oracle/libdir_oracle.so (dir_oracle.c) over NumPy arrays.

TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see dir_oracle.c header).  Function names mirror the
product C ABI (include/dir_hip.h) with host NumPy arrays in place of device pointers.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdir_oracle.so")
SUM, MEAN, SQRTN = 0, 1, 2
PRUNE_NONPOSITIVE_WEIGHTS = 1


def build(force=False):
    src = os.path.join(_HERE, "dir_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libdir_oracle.so"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct)) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


def _i64(a):
    return np.ascontiguousarray(a, np.int64)


def _ptr_array(arrs):
    arr = (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    return arr


def embedding_bag(tables, ids, offsets=None, weights=None, stride_b=None, stride_f=None,
                  combiner=MEAN, flags=0, B=None, out_ld=None, vocab=None, max_norm=0.0, out=None):
    """tables: list of F arrays [V_f,K].  One-hot: ids [B,F] (strides default to its layout).
    combiner: one code or a sequence of F codes; vocab: True (the tables' row counts) / sequence / None;
    out: a preallocated [B, out_ld] fp32 array to write into (the CPU-baseline leg reuses one)."""
    tables = [_f32(t) for t in tables]
    F, K = len(tables), tables[0].shape[1]
    ids = _i64(ids)
    if offsets is None:
        if B is None:
            B = ids.shape[0]
        if stride_b is None:
            stride_b, stride_f = F, 1
    else:
        offsets = _i64(offsets)
        if B is None:
            B = (offsets.size - 1) // F
        if stride_b is None:
            stride_b, stride_f = F, 1
    weights = _f32(weights) if weights is not None else None
    out_ld = out_ld or F * K
    if out is None:
        out = np.zeros((B, out_ld), np.float32)
    assert out.dtype == np.float32 and out.flags.c_contiguous and out.shape == (B, out_ld)
    slot = None
    if not isinstance(combiner, (int, np.integer)):
        slot = np.ascontiguousarray(combiner, np.int32)
        assert slot.size == F
        combiner = int(slot[0])
    if vocab is True:
        vocab = [t.shape[0] for t in tables]
    voc = _i64(vocab) if vocab is not None else None
    slot_mn = None
    if max_norm is not None and not isinstance(max_norm, (int, float, np.floating, np.integer)):   # one max_norm per slot (None / 0: none)
        slot_mn = np.ascontiguousarray([float(m or 0.0) for m in max_norm], np.float32)
        assert slot_mn.size == F
        max_norm = 0.0
    rc = lib().orc_embedding_bag_ex2_f32(_ptr_array(tables), _p(voc, ctypes.c_int64), F, K, _p(ids, ctypes.c_int64),
                                         _p(offsets, ctypes.c_int64), _p(weights, ctypes.c_float),
                                         ctypes.c_int64(stride_b), ctypes.c_int64(stride_f), _p(slot, ctypes.c_int32),
                                         int(combiner), _p(slot_mn, ctypes.c_float), ctypes.c_float(max_norm or 0.0), flags,
                                         ctypes.c_int64(B), _p(out, ctypes.c_float), ctypes.c_int64(out_ld))
    assert rc == 0, rc
    return out


def fm_second_order(emb, F, K, acc64=False):
    emb = _f32(emb)
    B = emb.shape[0]
    out = np.zeros(B, np.float32)
    rc = lib().orc_fm_second_order_f32(_p(emb, ctypes.c_float), ctypes.c_int64(emb.shape[1]),
                                       ctypes.c_int64(B), F, K, _p(out, ctypes.c_float), int(acc64))
    assert rc == 0, rc
    return out


def linear_sparse_sum(wts, ids, offsets=None, entry_weights=None, stride_b=None, stride_f=None,
                      combiner=SUM, bias=None, B=None, out=None):
    wts = [_f32(w).reshape(-1) for w in wts]
    F = len(wts)
    ids = _i64(ids)
    if offsets is not None:
        offsets = _i64(offsets)
    if B is None:
        B = ids.shape[0] if offsets is None else (offsets.size - 1) // F
    if stride_b is None:
        stride_b, stride_f = F, 1
    entry_weights = _f32(entry_weights) if entry_weights is not None else None
    bias_a = _f32(np.asarray(bias).reshape(1)) if bias is not None else None
    accumulate = out is not None
    out = _f32(out).copy() if accumulate else np.zeros(B, np.float32)
    rc = lib().orc_linear_sparse_sum_f32(_ptr_array(wts), F, _p(ids, ctypes.c_int64),
                                         _p(offsets, ctypes.c_int64), _p(entry_weights, ctypes.c_float),
                                         ctypes.c_int64(stride_b), ctypes.c_int64(stride_f), combiner,
                                         _p(bias_a, ctypes.c_float), int(accumulate), ctypes.c_int64(B),
                                         _p(out, ctypes.c_float))
    assert rc == 0, rc
    return out


def dcn_cross(x0, w, b, acc64=False):
    x0, w, b = _f32(x0), _f32(w), _f32(b)
    B, d = x0.shape
    L = w.shape[0]
    out = np.zeros((B, d), np.float32)
    rc = lib().orc_dcn_cross_f32(_p(x0, ctypes.c_float), ctypes.c_int64(d), _p(w, ctypes.c_float),
                                 _p(b, ctypes.c_float), L, ctypes.c_int64(B), d, _p(out, ctypes.c_float),
                                 ctypes.c_int64(d), int(acc64))
    assert rc == 0, rc
    return out


DIN_ACTIVATIONS = {"sigmoid": 0, "prelu": 1, "dice": 2}


def din_attention_pool(table, hist, hist_len, cand, W1, b1, W2, b2, W3, b3, normalize=False,
                       acc64=False, activation="sigmoid", act_params=None):
    """activation "prelu" / "dice" (paper-derived, arXiv:1706.06978 section 5.3): act_params float32 [3 H1 + 3 H2] = alpha1, scale1, shift1,
    alpha2, scale2, shift2 (Dice in its inference form: p = sigmoid(scale s + shift))."""
    table = _f32(table)
    K = table.shape[1]
    hist = _i64(hist)
    B, T = hist.shape
    hist_len = np.ascontiguousarray(hist_len, np.int32)
    cand = _i64(cand)
    W1, b1, W2, b2, W3, b3 = map(_f32, (W1, b1, W2, b2, W3, b3))
    H1, H2 = W1.shape[1], W2.shape[1]
    out = np.zeros((B, K), np.float32)
    scores = np.zeros((B, T), np.float32)
    f = ctypes.c_float
    ap = None
    if activation != "sigmoid":
        ap = _f32(np.asarray(act_params).reshape(-1))
        assert ap.size == 3 * H1 + 3 * H2, "act_params: [3 H1 + 3 H2]"
    rc = lib().orc_din_attention_pool_act_f32(_p(table, f), K, _p(hist, ctypes.c_int64),
                                              _p(hist_len, ctypes.c_int32), _p(cand, ctypes.c_int64), T,
                                              _p(W1, f), _p(b1, f), H1, _p(W2, f), _p(b2, f), H2, _p(W3, f),
                                              _p(b3, f), int(normalize), DIN_ACTIVATIONS[activation], _p(ap, f) if ap is not None else None,
                                              ctypes.c_int64(B), _p(out, f), _p(scores, f), int(acc64))
    assert rc == 0, rc
    return out, scores


def cin_layer(x0, xk, W, acc64=False):
    x0, xk, W = _f32(x0), _f32(xk), _f32(W)
    B, m, D = x0.shape
    Hp = xk.shape[1]
    H = W.shape[0]
    xout = np.zeros((B, H, D), np.float32)
    pooled = np.zeros((B, H), np.float32)
    f = ctypes.c_float
    rc = lib().orc_cin_layer_f32(_p(x0, f), _p(xk, f), _p(W, f), m, Hp, H, D, ctypes.c_int64(B),
                                 _p(xout, f), _p(pooled, f), ctypes.c_int64(H), int(acc64))
    assert rc == 0, rc
    return xout, pooled


def cin_backward(x0, xk, W, G):
    """-> (dW float64 [H, Hp*m], dxk float32 [B,Hp,D], dx0 float32 [B,m,D]); double accumulation."""
    x0, xk, W, G = _f32(x0), _f32(xk), _f32(W), _f32(G)
    B, m, D = x0.shape
    Hp = xk.shape[1]
    H = W.shape[0]
    dW = np.zeros((H, Hp * m), np.float64)
    dxk = np.zeros((B, Hp, D), np.float32)
    dx0 = np.zeros((B, m, D), np.float32)
    f = ctypes.c_float
    rc = lib().orc_cin_backward(_p(x0, f), _p(xk, f), _p(W, f), _p(G, f), m, Hp, H, D, ctypes.c_int64(B),
                                _p(dW, ctypes.c_double), _p(dxk, f), _p(dx0, f))
    assert rc == 0, rc
    return dW, dxk, dx0


def bucketize(x, boundaries):
    x = _f32(x).reshape(-1)
    bd = _f32(boundaries)
    out = np.zeros(x.size, np.int64)
    rc = lib().orc_bucketize_f32(_p(x, ctypes.c_float), ctypes.c_int64(x.size), _p(bd, ctypes.c_float),
                                 bd.size, _p(out, ctypes.c_int64))
    assert rc == 0, rc
    return out


def shard_div_owner(ids, vocab, P):
    ids = _i64(ids).reshape(-1)
    owner = np.zeros(ids.size, np.int64)
    local = np.zeros(ids.size, np.int64)
    o = ctypes.c_int()
    l = ctypes.c_int64()
    fn = lib().orc_shard_div_owner
    fn.restype = None
    for i, v in enumerate(ids):
        fn(ctypes.c_int64(int(v)), ctypes.c_int64(vocab), P, ctypes.byref(o), ctypes.byref(l))
        owner[i], local[i] = o.value, l.value
    return owner, local
