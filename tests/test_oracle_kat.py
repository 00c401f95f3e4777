"""CPU tests of the oracle itself: hand known-answer tests derivable from the cited reference lines
(SURVEY.md 8c), published FarmHash known answers, and C <-> NumPy agreement of the two restatements.
The reference holds no golden vectors for this path (parity unpinned); these are what pin the oracle."""
import numpy as np
import pytest

from oracle import np_ref as R


def test_fm_hand_kat(oracle):
    # deepFM.py:329-334: two fields -> e1.e2 ; three fields -> sum_{i<j} ei.ej
    e = np.array([[1, 2, 3, 4]], np.float32)
    assert oracle.fm_second_order(e, 2, 2)[0] == 11.0
    assert R.fm_logit(e, 2, 2)[0, 0] == 11.0
    e = np.array([[1, 2, 3, 4, -1, .5]], np.float32)
    assert oracle.fm_second_order(e, 3, 2)[0] == 10.0
    assert oracle.fm_second_order(e, 3, 2, acc64=True)[0] == 10.0


def test_cross_hand_kat(oracle):
    # DeepCrossNetwork.py:345-346,361-365
    x0 = np.array([[1, 2]], np.float32)
    w = np.array([[.5, -1], [.25, .5]], np.float32)
    b = np.array([[.1, .2], [0, -.1]], np.float32)
    np.testing.assert_allclose(oracle.dcn_cross(x0, w[:1], b[:1]), [[-0.4, -0.8]], rtol=1e-6)
    np.testing.assert_allclose(oracle.dcn_cross(x0, w, b), [[-0.9, -1.9]], rtol=1e-6)
    np.testing.assert_allclose(R.cross_network(x0, w, b), [[-0.9, -1.9]], rtol=1e-6)


def test_bag_hand_kat(oracle):
    tab = np.arange(40, dtype=np.float32).reshape(10, 4)
    # ids [3,-1,3] mean -> table[3]; empty bag -> zeros
    ids = np.array([3, -1, 3], np.int64)
    offs = np.array([0, 3, 3], np.int64)  # B=2, F=1: second bag empty
    out = oracle.embedding_bag([tab], ids, offsets=offs, combiner=oracle.MEAN, B=2)
    np.testing.assert_array_equal(out[0], tab[3])
    np.testing.assert_array_equal(out[1], np.zeros(4, np.float32))
    np.testing.assert_array_equal(R.bag(tab, ids, combiner=R.MEAN), tab[3])
    # sum of two rows; sqrtn divides by sqrt(2)
    ids = np.array([1, 2], np.int64)
    offs = np.array([0, 2], np.int64)
    np.testing.assert_array_equal(oracle.embedding_bag([tab], ids, offsets=offs, combiner=oracle.SUM, B=1)[0], tab[1] + tab[2])
    np.testing.assert_allclose(oracle.embedding_bag([tab], ids, offsets=offs, combiner=oracle.SQRTN, B=1)[0],
                               (tab[1] + tab[2]) / np.sqrt(np.float32(2)), rtol=1e-7)
    # weighted mean: (2*r1 + 6*r2) / 8
    w = np.array([2, 6], np.float32)
    np.testing.assert_allclose(oracle.embedding_bag([tab], ids, offsets=offs, weights=w, combiner=oracle.MEAN, B=1)[0],
                               (2 * tab[1] + 6 * tab[2]) / 8, rtol=1e-7)
    # one-hot: id < 0 -> zeros
    out = oracle.embedding_bag([tab, tab], np.array([[2, -1]], np.int64))
    np.testing.assert_array_equal(out[0], np.concatenate([tab[2], np.zeros(4, np.float32)]))


def test_shard_div_kat(oracle):
    exp = [0, 0, 0, 1, 1, 1, 2, 2, 3, 3]  # V=10, P=4
    own, loc = oracle.shard_div_owner(np.arange(10), 10, 4)
    assert own.tolist() == exp
    assert loc.tolist() == [0, 1, 2, 0, 1, 2, 0, 1, 0, 1]
    own2, loc2 = R.shard_div_owner(np.arange(10), 10, 4)
    assert own2.tolist() == exp and loc2.tolist() == loc.tolist()
    for V, P in [(1000000, 8), (7, 8), (100000000, 8), (13, 2)]:
        ids = np.unique(np.concatenate([np.arange(0, min(V, 50)), np.random.default_rng(1).integers(0, V, 200), [V - 1]]))
        o1, l1 = oracle.shard_div_owner(ids, V, P)
        o2, l2 = R.shard_div_owner(ids, V, P)
        np.testing.assert_array_equal(o1, o2)
        np.testing.assert_array_equal(l1, l2)
        assert o1.max() < P and (l1 >= 0).all()


# Published FarmHash Fingerprint64 known answers: BigQuery FARM_FINGERPRINT documentation examples
# ("1footrue", "2applefalse", "3true"), the widely quoted FARM_FINGERPRINT("foo"), Fingerprint64("") = k2,
# and the TensorFlow tf.strings.to_hash_bucket_fast(["Hello","TensorFlow","2.x"], 3) -> [0,2,2] doc example.
FARM_KATS = [(b"1footrue", -1541654101129638711), (b"2applefalse", 2794438866806483259),
             (b"3true", -4880158226897771312), (b"foo", 6150913649986995171), (b"", 0x9AE16A3B2F90404F - (1 << 64))]


def _s64(u):
    return u - (1 << 64) if u >= (1 << 63) else u


def test_farmhash_published_kats():
    for s, exp in FARM_KATS:
        assert _s64(R.fingerprint64(s)) == exp
    assert R.hash_bucket_fast(["Hello", "TensorFlow", "2.x"], 3).tolist() == [0, 2, 2]


def test_bucketize(oracle):
    bd = [0.5, 1.0, 2.0]
    x = np.array([-1, 0.5, 0.75, 1.0, 3.0, 2.0], np.float32)
    assert oracle.bucketize(x, bd).tolist() == [0, 1, 1, 2, 3, 3]
    assert R.bucketize(x, bd).tolist() == [0, 1, 1, 2, 3, 3]


@pytest.mark.parametrize("combiner", [0, 1, 2])
@pytest.mark.parametrize("weighted", [False, True])
def test_bag_c_vs_numpy(oracle, combiner, weighted):
    rng = np.random.default_rng(20240607 + combiner)
    F, K, B, V = 3, 8, 17, 50
    tables = [rng.standard_normal((V, K)).astype(np.float32) for _ in range(F)]
    lens = rng.integers(0, 6, size=B * F)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ids = rng.integers(-1, V, size=offs[-1]).astype(np.int64)
    w = rng.uniform(-0.5, 2.0, size=offs[-1]).astype(np.float32) if weighted else None
    for flags in ([0, 1] if weighted else [0]):
        out = oracle.embedding_bag(tables, ids, offsets=offs, weights=w, combiner=combiner, flags=flags, B=B)
        for b in range(B):
            for f in range(F):
                bag = b * F + f
                sl = slice(offs[bag], offs[bag + 1])
                ref = R.bag(tables[f], ids[sl], None if w is None else w[sl], combiner, bool(flags))
                np.testing.assert_array_equal(out[b, f * K:(f + 1) * K], ref)


def test_fm_cross_c_vs_numpy(oracle):
    rng = np.random.default_rng(7)
    B, F, K = 64, 26, 16
    emb = (rng.standard_normal((B, F * K)) * 0.25).astype(np.float32)
    a = oracle.fm_second_order(emb, F, K, acc64=True)
    b = R.fm_logit(emb, F, K, np.float64)[:, 0]
    np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-6)
    c = oracle.fm_second_order(emb, F, K)
    np.testing.assert_allclose(c, b, rtol=1e-5, atol=1e-5)
    d = F * K
    w = np.clip(rng.standard_normal((3, d)) * 0.1, -0.2, 0.2).astype(np.float32)
    bb = np.clip(rng.standard_normal((3, d)) * 0.1, -0.2, 0.2).astype(np.float32)
    x = oracle.dcn_cross(emb, w, bb, acc64=True)
    y = R.cross_network(emb.astype(np.float64), w.astype(np.float64), bb.astype(np.float64))
    np.testing.assert_allclose(x, y, rtol=1e-5, atol=1e-6)


def test_din_cin_c_vs_numpy(oracle):
    rng = np.random.default_rng(11)
    V, K, B, T, H1, H2 = 200, 8, 9, 7, 12, 8
    table = (rng.standard_normal((V, K)) * 0.3).astype(np.float32)
    hist = rng.integers(-1, V, size=(B, T)).astype(np.int64)
    hl = rng.integers(0, T + 1, size=B).astype(np.int32)
    cand = rng.integers(0, V, size=B).astype(np.int64)
    W1 = (rng.standard_normal((4 * K, H1)) * 0.2).astype(np.float32); b1 = (rng.standard_normal(H1) * 0.1).astype(np.float32)
    W2 = (rng.standard_normal((H1, H2)) * 0.2).astype(np.float32); b2 = (rng.standard_normal(H2) * 0.1).astype(np.float32)
    W3 = (rng.standard_normal(H2) * 0.3).astype(np.float32); b3 = np.array([0.05], np.float32)
    for norm in (False, True):
        o, s = oracle.din_attention_pool(table, hist, hl, cand, W1, b1, W2, b2, W3, b3, normalize=norm, acc64=True)
        o2, s2 = R.din_attention_pool(table, hist, hl, cand, W1, b1, W2, b2, W3, b3, normalize=norm)
        np.testing.assert_allclose(o, o2, rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(s, s2, rtol=2e-5, atol=2e-6)
    m, D, Hp, H = 5, 4, 6, 7
    x0 = rng.standard_normal((B, m, D)).astype(np.float32)
    xk = rng.standard_normal((B, Hp, D)).astype(np.float32)
    W = (rng.standard_normal((H, Hp * m)) * 0.2).astype(np.float32)
    xo, p = oracle.cin_layer(x0, xk, W, acc64=True)
    xo2, p2 = R.cin_layer(x0, xk, W)
    np.testing.assert_allclose(xo, xo2, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(p, p2, rtol=1e-5, atol=1e-5)
    xo3, _ = oracle.cin_layer(x0, xk, W, acc64=False)
    np.testing.assert_allclose(xo3, xo2, rtol=1e-4, atol=1e-5)
