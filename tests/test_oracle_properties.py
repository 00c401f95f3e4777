"""Property tests (hypothesis) of the oracle: algebraic identities that follow from the cited reference expressions and
[TF-upstream] bag semantics, checked on BOTH restatements (C via oracle.py, NumPy via np_ref.py).  Inputs are small
integers stored as fp32, so every identity below is exact in floating point.  The reference holds no vectors for this
path (parity unpinned): these properties, the hand KATs and the published FarmHash answers are what pin the oracle."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from oracle import np_ref as R

SET = dict(max_examples=40, deadline=None)


def _int_table(rng, v, k):
    return rng.integers(-8, 9, size=(v, k)).astype(np.float32)


@settings(**SET)
@given(st.integers(0, 2**31 - 1), st.integers(1, 12), st.integers(1, 9), st.lists(st.integers(-1, 11), min_size=0, max_size=9))
def test_bag_semantics(oracle, seed, V, K, raw_ids):
    rng = np.random.default_rng(seed)
    tab = _int_table(rng, V, K)
    ids = np.array([i if i < V else -1 for i in raw_ids], np.int64)          # some pruned (< 0)
    kept = ids[ids >= 0]

    def bag(ids_, combiner, weights=None):
        offs = np.array([0, len(ids_)], np.int64)
        src = ids_ if len(ids_) else np.zeros(1, np.int64)
        w = None if weights is None else (weights if len(ids_) else np.ones(1, np.float32))
        return oracle.embedding_bag([tab], src, offsets=offs, weights=w, combiner=combiner, B=1)[0]

    s = bag(ids, oracle.SUM)
    # pruned ids contribute nothing; the empty bag is the zero vector ([TF-upstream] safe_embedding_lookup_sparse)
    np.testing.assert_array_equal(s, tab[kept].sum(0) if len(kept) else np.zeros(K, np.float32))
    np.testing.assert_array_equal(s, bag(kept, oracle.SUM))
    np.testing.assert_array_equal(s, R.bag(tab, ids, combiner=R.SUM))
    # sum is additive over a split of the bag (integers: exact)
    cut = len(ids) // 2
    np.testing.assert_array_equal(s, bag(ids[:cut], oracle.SUM) + bag(ids[cut:], oracle.SUM))
    # mean = sum / count, sqrtn = sum / sqrt(count); a bag repeating one id is that row under mean
    if len(kept):
        np.testing.assert_allclose(bag(ids, oracle.MEAN), s / np.float32(len(kept)), rtol=1e-6)
        np.testing.assert_allclose(bag(ids, oracle.SQRTN), s / np.sqrt(np.float32(len(kept))), rtol=1e-6)
        one = np.full(3, kept[0], np.int64)
        np.testing.assert_array_equal(bag(one, oracle.MEAN), tab[kept[0]])
        # weights: mean is invariant to scaling all weights by a power of two; sum scales with them
        w = rng.integers(1, 5, size=len(ids)).astype(np.float32)
        np.testing.assert_array_equal(bag(ids, oracle.MEAN, w), bag(ids, oracle.MEAN, w * 4))
        np.testing.assert_array_equal(bag(ids, oracle.SUM, w * 2), bag(ids, oracle.SUM, w) * 2)
        np.testing.assert_allclose(bag(ids, oracle.MEAN, w), R.bag(tab, ids, weights=w, combiner=R.MEAN), rtol=1e-6)


@settings(**SET)
@given(st.integers(0, 2**31 - 1), st.integers(1, 6), st.integers(1, 5), st.integers(1, 4))
def test_fm_is_the_sum_of_pairwise_dots(oracle, seed, F, K, B):
    # deepFM.py:329-334: 0.5 * sum_k[(sum_f e)^2 - sum_f e^2] == sum_{i<j} e_i . e_j
    rng = np.random.default_rng(seed)
    e = rng.integers(-4, 5, size=(B, F, K)).astype(np.float32)
    want = np.zeros(B, np.float32)
    for i in range(F):
        for j in range(i + 1, F):
            want += (e[:, i] * e[:, j]).sum(1)
    np.testing.assert_array_equal(oracle.fm_second_order(e.reshape(B, F * K), F, K), want)
    np.testing.assert_array_equal(R.fm_logit(e.reshape(B, F * K), F, K)[:, 0], want)
    # scaling the embeddings by 2 scales the logit by 4
    np.testing.assert_array_equal(oracle.fm_second_order(2 * e.reshape(B, F * K), F, K), 4 * want)


@settings(**SET)
@given(st.integers(0, 2**31 - 1), st.integers(1, 9), st.integers(0, 4), st.integers(1, 4))
def test_cross_identities(oracle, seed, d, L, B):
    # DeepCrossNetwork.py:345-346: x_{l+1} = x0 * (x_l . w_l) + b_l + x_l
    rng = np.random.default_rng(seed)
    x0 = rng.integers(-3, 4, size=(B, d)).astype(np.float32)
    w = rng.integers(-2, 3, size=(L, d)).astype(np.float32)
    b = rng.integers(-2, 3, size=(L, d)).astype(np.float32)
    # w = 0: every layer only adds its bias
    np.testing.assert_array_equal(oracle.dcn_cross(x0, np.zeros_like(w), b), x0 + b.sum(0))
    # recurrence, step by step, both restatements
    x = x0
    for l in range(L):
        x = x0 * (x @ w[l])[:, None] + b[l] + x
    np.testing.assert_array_equal(oracle.dcn_cross(x0, w, b), x)
    np.testing.assert_array_equal(R.cross_network(x0, w, b), x)
    # every layer output stays in span{x0} + x0 + biases: with b = 0 each row of the output is a multiple of its x0 row
    out = oracle.dcn_cross(x0, w, np.zeros_like(b))
    for r in range(B):
        nz = np.flatnonzero(x0[r])
        if len(nz):
            ratio = out[r, nz[0]] / x0[r, nz[0]]
            np.testing.assert_array_equal(out[r], ratio * x0[r])


@settings(**SET)
@given(st.integers(1, 5000), st.integers(1, 17))
def test_div_sharding_is_a_contiguous_balanced_partition(oracle, V, P):
    # [TF-upstream] partition_strategy='div' under deepFM.py:163-167
    ids = np.arange(V)
    own, loc = oracle.shard_div_owner(ids, V, P)
    own2, loc2 = R.shard_div_owner(ids, V, P)
    assert own.tolist() == own2.tolist() and loc.tolist() == loc2.tolist()
    assert (np.diff(own) >= 0).all() and own.min() == 0 and own.max() <= P - 1
    sizes = np.bincount(own, minlength=P)
    assert sizes.max() - sizes.min() <= 1 and (np.diff(sizes) <= 0).all()        # the first V % P shards hold one extra row
    for p in range(P):
        np.testing.assert_array_equal(loc[own == p], np.arange(sizes[p]))       # local ids are 0..size-1, in order


@settings(max_examples=60, deadline=None)
@given(st.binary(min_size=0, max_size=200), st.integers(1, 10**6))
def test_farmhash_c_equals_python_and_bucket_in_range(built_lib, data, buckets):
    """The C library's host FarmHash64 (dir_fingerprint64, the product's hashed-column path) against the independent pure-Python
    restatement, over every length branch (0-16, 17-32, 33-64, > 64 bytes)."""
    from dir_amd import ops
    got = ops.fingerprint64(data)                      # host code of libdir_hip.so: no GPU involved
    assert got == R.fingerprint64(data)
    assert 0 <= got % buckets < buckets
