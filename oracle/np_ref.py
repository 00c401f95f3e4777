"""np_ref.py -- independent NumPy / pure-Python restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY (see oracle/dir_oracle.c header): imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product package.

PARITY UNPINNED by the reference: it holds no tests or golden vectors for this path and needs
TensorFlow 1.x, which cannot run here.  What pins this file: hand known-answer tests derived from the
cited lines, published FarmHash Fingerprint64 known answers (BigQuery FARM_FINGERPRINT documentation
examples, the TensorFlow to_hash_bucket_fast documentation example), and agreement with the separate C
restatement in dir_oracle.c.

Citations are relative to /root/reference.  [TF-upstream] = TensorFlow 1.x library semantics the
reference calls but does not contain.
"""
import numpy as np

SUM, MEAN, SQRTN = 0, 1, 2


# ---------------------------------------------------------------------------------------------
# A2: one embedding bag, written as a Python loop on purpose (small cases only).
# [TF-upstream] safe_embedding_lookup_sparse / embedding_lookup_sparse; call site
# models/DeepFM/deepFM.py:387-390.
# ---------------------------------------------------------------------------------------------
def clip_by_norm(row, max_norm):
    """[TF-upstream] clip_ops.clip_by_norm (r1.10+) on one looked-up row (embedding_lookup(max_norm=)), fp32."""
    row = row.astype(np.float32)
    l2sum = np.float32(0)
    for v in row:
        l2sum = np.float32(l2sum + v * v)
    l2norm = np.sqrt(l2sum) if l2sum > 0 else l2sum
    return ((row * np.float32(max_norm)) / np.maximum(np.float32(l2norm), np.float32(max_norm))).astype(np.float32)


def bag(table, ids, weights=None, combiner=MEAN, prune_nonpositive_weights=False, max_norm=None):
    K = table.shape[1]
    acc = np.zeros(K, np.float32)
    wsum = np.float32(0)
    w2sum = np.float32(0)
    cnt = 0
    for e, i in enumerate(ids):
        if i < 0:
            continue
        w = np.float32(1) if weights is None else np.float32(weights[e])
        if weights is not None and prune_nonpositive_weights and not (w > 0):
            continue
        if i >= table.shape[0]:
            continue                     # out of range: no contribution (TF GPU lookups return zeros)
        row = table[int(i)].astype(np.float32)
        if max_norm:
            row = clip_by_norm(row, max_norm)
        acc = (acc + w * row).astype(np.float32) if weights is not None else (acc + row).astype(np.float32)
        wsum = np.float32(wsum + w)
        w2sum = np.float32(w2sum + w * w)
        cnt += 1
    if cnt == 0:
        return np.zeros(K, np.float32)
    if combiner == MEAN:
        den = wsum if weights is not None else np.float32(cnt)
        return (acc / den).astype(np.float32)
    if combiner == SQRTN:
        den = np.sqrt(w2sum) if weights is not None else np.sqrt(np.float32(cnt))
        return (acc / np.float32(den)).astype(np.float32)
    return acc


def embedding_bag_onehot(tables, ids_bf):
    """ids_bf [B,F] -> [B, F*K]; id < 0 -> zeros (empty bag).  deepFM.py:383-393."""
    B, F = ids_bf.shape
    K = tables[0].shape[1]
    out = np.zeros((B, F, K), np.float32)
    for f in range(F):
        ok = ids_bf[:, f] >= 0
        out[ok, f] = tables[f][ids_bf[ok, f]]
    return out.reshape(B, F * K)


# ---------------------------------------------------------------------------------------------
# A4: fm_logit_fn, models/DeepFM/deepFM.py:329-334, in the reference's four-op form.
# ---------------------------------------------------------------------------------------------
def fm_logit(net, F, K, dtype=np.float32):
    emb = net.reshape(-1, F, K).astype(dtype)                       # :329
    summed_squared = np.square(emb.sum(axis=-2, dtype=dtype))       # :331
    squared_summed = np.square(emb).sum(axis=-2, dtype=dtype)       # :332
    logits = dtype(0.5) * (summed_squared - squared_summed).sum(axis=-1, dtype=dtype)  # :333
    return logits[:, None]                                          # :334


# ---------------------------------------------------------------------------------------------
# A8: _cross_op / _cross_architecture, models/DeepCrossNetwork/DeepCrossNetwork.py:345-346,361-365
# ---------------------------------------------------------------------------------------------
def cross_op(x0, x, w, b):
    x_w = np.tensordot(x, w, axes=1)                                # :345
    return x0 * x_w[:, None] + b + x                                # :346


def cross_network(x0, w, b):
    xl = x0
    for l in range(w.shape[0]):                                     # :363-365
        xl = cross_op(x0, xl, w[l], b[l])
    return xl


# ---------------------------------------------------------------------------------------------
# A5 / A9: MLPs.  dnn_logit_fn deepFM.py:284-319 (dense -> [dropout] -> [BN]);
# _deep_architecture DeepCrossNetwork.py:370-410 (dense(act) -> BN on all but last).
# Inference-mode BN [TF-upstream]: y = (x - mean) * rsqrt(var + eps) * gamma + beta, eps 1e-3;
# contrib.layers.batch_norm default has no gamma (scale=False).
# ---------------------------------------------------------------------------------------------
def relu(x):
    return np.maximum(x, 0)


def batch_norm_infer(x, mean, var, gamma=None, beta=None, eps=1e-3):
    inv = 1.0 / np.sqrt(var + eps)
    if gamma is not None:
        inv = inv * gamma
    y = x * inv + ((beta if beta is not None else 0.0) - mean * inv)
    return y.astype(x.dtype)


def dnn_logit(net, layers, logits_layer, bn=None, act=relu):
    """layers: [(W[in,out], b[out])]; bn: None or [(mean,var,gamma,beta)] per hidden layer."""
    for i, (W, b) in enumerate(layers):
        net = act(net @ W + b)                                      # deepFM.py:295-300
        if bn is not None:
            net = batch_norm_infer(net, *bn[i])                     # deepFM.py:303-308
    W, b = logits_layer
    return net @ W + b                                              # deepFM.py:312-317


def deep_architecture(net, layers, bn=None, act=relu):
    n = len(layers)
    for i, (W, b) in enumerate(layers):
        net = act(net @ W + b)                                      # DeepCrossNetwork.py:394-399
        if bn is not None and i < n - 1:                            # :401
            mean, var, beta = bn[i]
            net = batch_norm_infer(net, mean, var, None, beta)      # :403, :418-419
    return net


def predictions(logits):
    """_create_estimator_spec predictions, DeepCrossNetwork.py:153-165."""
    logistic = 1.0 / (1.0 + np.exp(-logits))                        # :156
    two = np.concatenate([np.zeros_like(logits), logits], axis=-1)  # :157
    e = np.exp(two - two.max(axis=-1, keepdims=True))
    prob = e / e.sum(axis=-1, keepdims=True)                        # :159
    class_ids = np.argmax(two, axis=-1)[:, None]                    # :160-161
    return {"logits": logits, "logistic": logistic, "probabilities": prob, "class_ids": class_ids}


# ---------------------------------------------------------------------------------------------
# A13 / A14: paper restatements (no reference code; README.md:27-28).
# ---------------------------------------------------------------------------------------------
def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def din_unit_activation(s, kind, params):
    """The unit's hidden activation on [n, H] pre-activations: "sigmoid" | "prelu" (params[0] = alpha [H]) | "dice" (params = alpha,
    scale, shift rows: the inference form of arXiv:1706.06978 section 5.3's Dice, p = sigmoid(scale s + shift), f = p s + (1 - p) alpha s).
    Paper-derived: the reference holds no DIN code (README.md:27)."""
    if kind == "prelu":
        return np.where(s > 0, s, params[0] * s)
    if kind == "dice":
        pg = sigmoid(params[1] * s + params[2])
        return pg * s + (1.0 - pg) * params[0] * s
    return sigmoid(s)


def din_attention_pool(table, hist, hist_len, cand, W1, b1, W2, b2, W3, b3, normalize=False,
                       dtype=np.float64, activation="sigmoid", act_params=None):
    """activation "prelu" / "dice": act_params [3 H1 + 3 H2] = alpha1, scale1, shift1, alpha2, scale2, shift2."""
    B, T = hist.shape
    K = table.shape[1]
    H1, H2 = W1.shape[1], W2.shape[1]
    ap1 = ap2 = None
    if activation != "sigmoid":
        ap = np.asarray(act_params, dtype).reshape(-1)
        ap1, ap2 = ap[:3 * H1].reshape(3, H1), ap[3 * H1:].reshape(3, H2)
    out = np.zeros((B, K), dtype)
    scores = np.zeros((B, T), dtype)
    for b in range(B):
        a = table[cand[b]].astype(dtype) if cand[b] >= 0 else np.zeros(K, dtype)
        valid = [j for j in range(T) if j < hist_len[b] and hist[b, j] >= 0]
        if not valid:
            continue
        h = table[hist[b, valid]].astype(dtype)                               # [n,K]
        u = np.concatenate([h, np.broadcast_to(a, h.shape), h - a, h * a], axis=1)
        z1 = din_unit_activation(u @ W1.astype(dtype) + b1.astype(dtype), activation, ap1)
        z2 = din_unit_activation(z1 @ W2.astype(dtype) + b2.astype(dtype), activation, ap2)
        s = z2 @ W3.astype(dtype) + dtype(b3[0])
        if normalize:
            s = s / np.sqrt(dtype(K))
            e = np.exp(s - s.max())
            s = e / e.sum()
        out[b] = s @ h
        scores[b, valid] = s
    return out, scores


def din_model_logits(x_cols, table, hist, hist_len, cand, unit, mlp, head, normalize=False, activation="sigmoid", act_params=None,
                     dtype=np.float64):
    """The whole DIN forward (arXiv:1706.06978 figure 2; paper-derived, README.md:27 has no code): concat([profile / context block x_cols
    [B, W] or None, pooled interest vector, candidate embedding]) -> hidden layers -> logit [B, 1].
    unit = (W1, b1, W2, b2, W3, b3); mlp = [(W [out, in], b [out], kind, params)] with kind "relu" | "sigmoid" | "prelu" | "dice" and
    params as din_unit_activation takes them; head = (w [1, in], b [1])."""
    pooled, _ = din_attention_pool(table, hist, hist_len, cand, *unit, normalize=normalize, dtype=dtype, activation=activation,
                                   act_params=act_params)
    ce = np.where((np.asarray(cand) >= 0)[:, None], table[np.maximum(cand, 0)].astype(dtype), 0.0)
    parts = ([np.asarray(x_cols, dtype)] if x_cols is not None else []) + [pooled, ce]
    net = np.concatenate(parts, axis=1)
    for W, b, kind, params in mlp:
        pre = net @ np.asarray(W, dtype).T + np.asarray(b, dtype)
        net = np.maximum(pre, 0) if kind == "relu" else din_unit_activation(pre, kind, params)
    return net @ np.asarray(head[0], dtype).T + np.asarray(head[1], dtype)


def cin_layer(x0, xk, W, dtype=np.float64):
    """x0 [B,m,D], xk [B,Hp,D], W [H, Hp*m] -> (xout [B,H,D], pooled [B,H]).  arXiv:1803.05170 eq.6"""
    B, m, D = x0.shape
    Hp = xk.shape[1]
    H = W.shape[0]
    Wr = W.reshape(H, Hp, m).astype(dtype)
    xout = np.einsum("hij,bid,bjd->bhd", Wr, xk.astype(dtype), x0.astype(dtype), optimize=True)
    return xout, xout.sum(axis=-1)


# ---------------------------------------------------------------------------------------------
# A3: FarmHash Fingerprint64 (farmhashna::Hash64, FarmHash 1.1), what [TF-upstream]
# string_to_hash_bucket_fast computes before `mod num_buckets`; reference call site
# models/DeepCrossNetwork/train.py:85-86.  Pure integer arithmetic, restated from the public source.
# Pinned for lengths <= 16 by published known answers (tests/test_oracle_kat.py); the 17-32, 33-64 and
# > 64 byte branches have no published vector known to us and are pinned only by C <-> Python agreement.
# ---------------------------------------------------------------------------------------------
_M = (1 << 64) - 1
_K0, _K1, _K2 = 0xC3A5C85C97CB3127, 0xB492B66FBE98F273, 0x9AE16A3B2F90404F


def _f64(s, i):
    return int.from_bytes(s[i:i + 8], "little")


def _f32(s, i):
    return int.from_bytes(s[i:i + 4], "little")


def _rot(v, n):
    return ((v >> n) | (v << (64 - n))) & _M if n else v


def _smix(v):
    return v ^ (v >> 47)


def _h16(u, v, mul):
    a = ((u ^ v) * mul) & _M
    a ^= a >> 47
    b = ((v ^ a) * mul) & _M
    b ^= b >> 47
    return (b * mul) & _M


def _weak(s, i, a, b):
    w, x, y, z = _f64(s, i), _f64(s, i + 8), _f64(s, i + 16), _f64(s, i + 24)
    a = (a + w) & _M
    b = _rot((b + a + z) & _M, 21)
    c = a
    a = (a + x) & _M
    a = (a + y) & _M
    b = (b + _rot(a, 44)) & _M
    return (a + z) & _M, (b + c) & _M


def fingerprint64(s: bytes) -> int:
    n = len(s)
    if n <= 16:
        if n >= 8:
            mul = (_K2 + n * 2) & _M
            a = (_f64(s, 0) + _K2) & _M
            b = _f64(s, n - 8)
            c = (_rot(b, 37) * mul + a) & _M
            d = ((_rot(a, 25) + b) * mul) & _M
            return _h16(c, d, mul)
        if n >= 4:
            mul = (_K2 + n * 2) & _M
            return _h16((n + (_f32(s, 0) << 3)) & _M, _f32(s, n - 4), mul)
        if n > 0:
            y = (s[0] + (s[n >> 1] << 8)) & 0xFFFFFFFF
            z = (n + (s[n - 1] << 2)) & 0xFFFFFFFF
            return (_smix((y * _K2 ^ z * _K0) & _M) * _K2) & _M
        return _K2
    if n <= 32:
        mul = (_K2 + n * 2) & _M
        a = (_f64(s, 0) * _K1) & _M
        b = _f64(s, 8)
        c = (_f64(s, n - 8) * mul) & _M
        d = (_f64(s, n - 16) * _K2) & _M
        return _h16((_rot((a + b) & _M, 43) + _rot(c, 30) + d) & _M,
                    (a + _rot((b + _K2) & _M, 18) + c) & _M, mul)
    if n <= 64:
        mul = (_K2 + n * 2) & _M
        a = (_f64(s, 0) * _K2) & _M
        b = _f64(s, 8)
        c = (_f64(s, n - 8) * mul) & _M
        d = (_f64(s, n - 16) * _K2) & _M
        y = (_rot((a + b) & _M, 43) + _rot(c, 30) + d) & _M
        z = _h16(y, (a + _rot((b + _K2) & _M, 18) + c) & _M, mul)
        e = (_f64(s, 16) * mul) & _M
        f = _f64(s, 24)
        g = ((y + _f64(s, n - 32)) * mul) & _M
        h = ((z + _f64(s, n - 24)) * mul) & _M
        return _h16((_rot((e + f) & _M, 43) + _rot(g, 30) + h) & _M,
                    (e + _rot((f + a) & _M, 18) + g) & _M, mul)
    x = 81
    y = (81 * _K1 + 113) & _M
    z = (_smix((y * _K2 + 113) & _M) * _K2) & _M
    v = (0, 0)
    w = (0, 0)
    x = (x * _K2 + _f64(s, 0)) & _M
    end = ((n - 1) // 64) * 64
    last64 = end + ((n - 1) & 63) - 63
    p = 0
    while True:
        x = (_rot((x + y + v[0] + _f64(s, p + 8)) & _M, 37) * _K1) & _M
        y = (_rot((y + v[1] + _f64(s, p + 48)) & _M, 42) * _K1) & _M
        x ^= w[1]
        y = (y + v[0] + _f64(s, p + 40)) & _M
        z = (_rot((z + w[0]) & _M, 33) * _K1) & _M
        v = _weak(s, p, (v[1] * _K1) & _M, (x + w[0]) & _M)
        w = _weak(s, p + 32, (z + w[1]) & _M, (y + _f64(s, p + 16)) & _M)
        z, x = x, z
        p += 64
        if p == end:
            break
    mul = (_K1 + ((z & 0xFF) << 1)) & _M
    p = last64
    w = ((w[0] + ((n - 1) & 63)) & _M, w[1])
    v = ((v[0] + w[0]) & _M, v[1])
    w = ((w[0] + v[0]) & _M, w[1])
    x = (_rot((x + y + v[0] + _f64(s, p + 8)) & _M, 37) * mul) & _M
    y = (_rot((y + v[1] + _f64(s, p + 48)) & _M, 42) * mul) & _M
    x ^= (w[1] * 9) & _M
    y = (y + v[0] * 9 + _f64(s, p + 40)) & _M
    z = (_rot((z + w[0]) & _M, 33) * mul) & _M
    v = _weak(s, p, (v[1] * mul) & _M, (x + w[0]) & _M)
    w = _weak(s, p + 32, (z + w[1]) & _M, (y + _f64(s, p + 16)) & _M)
    z, x = x, z
    return _h16((_h16(v[0], w[0], mul) + (_smix(y) * _K0) + z) & _M,
                (_h16(v[1], w[1], mul) + x) & _M, mul)


def hash_bucket_fast(strings, num_buckets):
    """[TF-upstream] string_to_hash_bucket_fast: Fingerprint64(s) mod num_buckets (unsigned)."""
    out = np.empty(len(strings), np.int64)
    for i, s in enumerate(strings):
        if isinstance(s, str):
            s = s.encode("utf-8")
        out[i] = fingerprint64(bytes(s)) % num_buckets
    return out


def hash_bucket_int(keys, num_buckets):
    """[TF-upstream] integer keys are as_string()-ed (decimal) before hashing."""
    return hash_bucket_fast([str(int(k)) for k in keys], num_buckets)


def bucketize(x, boundaries):
    """[TF-upstream] bucketized_column: number of boundaries <= x."""
    return np.searchsorted(np.asarray(boundaries, np.float32), np.asarray(x, np.float32),
                           side="right").astype(np.int64)


def shard_div_owner(ids, vocab, P):
    """'div' partition strategy (deepFM.py:163-167 partitioner + [TF-upstream] embedding_lookup)."""
    ids = np.asarray(ids, np.int64)
    q, r = divmod(int(vocab), int(P))
    thr = r * (q + 1)
    owner = np.where(ids < thr, ids // (q + 1), r + (ids - thr) // max(q, 1))
    start = np.where(owner < r, owner * (q + 1), thr + (owner - r) * q)
    return owner.astype(np.int64), (ids - start).astype(np.int64)


# ---- backward / optimiser restatements (float64; derivatives of the expressions above and the [TF-upstream] update rules) ------
def fm_logit_backward(emb, g, F, K, add_in=None):
    """d/d emb of sum_b g[b] * fm_logit(emb)[b] (+ add_in): g[b] * (sum_f' e[b,f',:] - e[b,f,:])   (deepFM.py:329-334)."""
    e = np.asarray(emb, np.float64).reshape(-1, F, K)
    d = np.asarray(g, np.float64).reshape(-1, 1, 1) * (e.sum(1, keepdims=True) - e)
    d = d.reshape(e.shape[0], F * K)
    return d if add_in is None else d + np.asarray(add_in, np.float64)


def cross_network_backward(x0, w, b, gout):
    """Backward of cross_network (DeepCrossNetwork.py:345-346,361-365): -> (gx0 [B,d], gw [L,d], gb [L,d])."""
    x0 = np.asarray(x0, np.float64)
    w, b = np.asarray(w, np.float64), np.asarray(b, np.float64)
    L = w.shape[0]
    xs, s = [x0], []
    for l in range(L):
        s.append(xs[l] @ w[l])
        xs.append(x0 * s[l][:, None] + b[l] + xs[l])
    g = np.asarray(gout, np.float64).copy()
    gx0 = np.zeros_like(x0)
    gw, gb = np.zeros_like(w), np.zeros_like(b)
    for l in range(L - 1, -1, -1):
        t = (g * x0).sum(1)                       # dL/ds_l
        gb[l] = g.sum(0)
        gw[l] = (t[:, None] * xs[l]).sum(0)
        gx0 += g * s[l][:, None]
        g = g + t[:, None] * w[l]
    return gx0 + g, gw, gb


def _dedup_sum(ids_f, grad_f, V):
    gs = np.zeros((V, grad_f.shape[1]))
    ok = ids_f >= 0
    np.add.at(gs, ids_f[ok], np.asarray(grad_f, np.float64)[ok])
    touched = np.zeros(V, bool)
    touched[ids_f[ok]] = True
    return gs, touched


def sparse_adagrad_step(tables, accums, ids, grad, lr):
    """[TF-upstream] AdagradOptimizer on IndexedSlices (duplicates summed first): accum += g^2; w -= lr*g/sqrt(accum).
    tables / accums: lists of float64 [V,K] arrays updated in place; ids [B,F]; grad [B,F*K]."""
    K = tables[0].shape[1]
    for f, (w, a) in enumerate(zip(tables, accums)):
        gs, t = _dedup_sum(ids[:, f], grad[:, f * K:(f + 1) * K], w.shape[0])
        a[t] += gs[t] ** 2
        w[t] -= lr * gs[t] / np.sqrt(a[t])


def sparse_ftrl_step(tables, accums, linears, ids, grad, lr, l1=0.0, l2=0.0):
    """[TF-upstream] FtrlOptimizer (learning_rate_power -0.5) on IndexedSlices; grad [B,F*K] or [B,K] (shared by the slots)."""
    K = tables[0].shape[1]
    for f, (w, n, z) in enumerate(zip(tables, accums, linears)):
        gf = grad if grad.shape[1] == K else grad[:, f * K:(f + 1) * K]
        gs, t = _dedup_sum(ids[:, f], gf, w.shape[0])
        n_new = n[t] + gs[t] ** 2
        sigma = (np.sqrt(n_new) - np.sqrt(n[t])) / lr
        z_new = z[t] + gs[t] - sigma * w[t]
        quad = np.sqrt(n_new) / lr + 2 * l2
        w[t] = np.where(np.abs(z_new) > l1, (np.sign(z_new) * l1 - z_new) / quad, 0.0)
        n[t], z[t] = n_new, z_new
