"""The PACKED DIN unit (round 6: csrc/din_pack.hip, dir_din_attention_pool_packed_f32) against the double-accumulating oracle
(oracle/dir_oracle.c: paper-derived, README.md:27 -> arXiv:1706.06978; no reference code) at the 1e-5 bar of tests/test_gpu_parity.py, on
the cases packing makes interesting: rows of several samples in one MFMA tile, samples spanning passes and tiles, empty samples (alone,
in runs longer than a block, at the ends of the batch), masked positions inside the length, T > 64, one-sample batches, no length vector."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _close(got, ref, tol=1e-5):
    err = np.abs(got.astype(np.float64) - ref) / (1.0 + np.abs(ref))
    assert err.max() <= tol, "max scaled err %.3e" % err.max()


def _weights(rng, K, H1, H2):
    W1 = (rng.standard_normal((4 * K, H1)) * 0.1).astype(np.float32); b1 = (rng.standard_normal(H1) * 0.1).astype(np.float32)
    W2 = (rng.standard_normal((H1, H2)) * 0.2).astype(np.float32); b2 = (rng.standard_normal(H2) * 0.1).astype(np.float32)
    W3 = (rng.standard_normal(H2) * 0.3).astype(np.float32); b3 = np.array([0.05], np.float32)
    return W1, b1, W2, b2, W3, b3


def _lengths(rng, B, T, kind):
    if kind == "uniform":
        hl = rng.integers(0, T + 1, size=B)
    elif kind == "short":                      # many samples per tile
        hl = rng.integers(0, 4, size=B)
    elif kind == "full":
        hl = np.full(B, T)
    elif kind == "empty_runs":                 # runs of empty samples longer than a block of 16, also at both ends of the batch
        hl = rng.integers(1, T + 1, size=B)
        hl[:min(B, 21)] = 0
        hl[B // 2:B // 2 + 40] = 0
        hl[-19:] = 0
    elif kind == "all_empty":
        hl = np.zeros(B)
    elif kind == "multiples":                  # block rows a multiple of 32: the last pass ends exactly at the block's end
        hl = np.full(B, 32 if T >= 32 else T)
        hl[1::2] = 0
    else:
        raise ValueError(kind)
    return hl.astype(np.int32)


@pytest.fixture(scope="module")
def ops(built_lib):
    from dir_amd import ops as o
    assert o.DIN_PACKED
    return o


@pytest.mark.parametrize("B,T,H1,H2,kind", [
    (33, 50, 80, 40, "uniform"), (130, 64, 80, 40, "uniform"), (20, 17, 72, 36, "uniform"), (9, 1, 80, 40, "uniform"),
    (40, 50, 64, 32, "uniform"), (40, 33, 16, 4, "uniform"), (25, 40, 48, 48, "uniform"), (1, 50, 80, 40, "full"), (2, 7, 80, 40, "uniform"),
    (17, 100, 80, 40, "uniform"), (70, 300, 80, 40, "uniform"), (1000, 50, 80, 40, "short"), (1500, 50, 80, 40, "empty_runs"),
    (64, 50, 80, 40, "all_empty"), (96, 50, 80, 40, "multiples"), (4099, 50, 80, 40, "uniform"), (513, 16, 80, 40, "full"),
    (700, 33, 80, 40, "multiples")])
@pytest.mark.parametrize("normalize", [False, True])
def test_packed_din_matches_oracle(ops, oracle, B, T, H1, H2, kind, normalize):
    K, V = 64, 3000
    assert ops.din_pack_covers(K, T, H1, H2)
    rng = np.random.default_rng(B * 131 + T * 7 + H1)
    table = (rng.standard_normal((V, K)) * 0.3).astype(np.float32)
    hist = rng.integers(-1, V, size=(B, T)).astype(np.int64)           # -1: a masked position inside the length
    hl = _lengths(rng, B, T, kind)
    cand = rng.integers(0, V, size=B).astype(np.int64)
    ws = _weights(rng, K, H1, H2)
    ref_o, ref_s = oracle.din_attention_pool(table, hist, hl, cand, *ws, normalize=normalize, acc64=True)
    args = [_dev(table), _dev(hist), _dev(hl), _dev(cand)] + [_dev(w) for w in ws]
    got_o, got_s = ops.din_attention_pool(*args, normalize=normalize, want_scores=True, arith="f16x2")
    _close(got_s.cpu().numpy(), ref_s)
    _close(got_o.cpu().numpy(), ref_o)
    only_o = ops.din_attention_pool(*args, normalize=normalize, arith="f16x2")          # the kernel without the scores output
    assert torch.equal(only_o, got_o)
    again = ops.din_attention_pool(*args, normalize=normalize, arith="f16x2")           # static partition: bitwise the same on every run
    assert torch.equal(again, got_o)
    empty = hl == 0
    if empty.any():
        assert not got_o.cpu().numpy()[empty].any()


@pytest.mark.parametrize("activation", ["prelu", "dice"])
@pytest.mark.parametrize("normalize", [False, True])
def test_packed_din_prelu_dice(ops, oracle, activation, normalize):
    """The paper's own hidden activations (arXiv:1706.06978 section 5.3) on the packed kernel."""
    from oracle import np_ref as R
    K, V, B, T, H1, H2 = 64, 2000, 300, 50, 80, 40
    rng = np.random.default_rng(77)
    table = (rng.standard_normal((V, K)) * 0.3).astype(np.float32)
    hist = rng.integers(-1, V, size=(B, T)).astype(np.int64)
    hl = _lengths(rng, B, T, "uniform")
    cand = rng.integers(0, V, size=B).astype(np.int64)
    ws = _weights(rng, K, H1, H2)
    a1 = rng.uniform(-0.2, 0.5, H1).astype(np.float32); a2 = rng.uniform(-0.2, 0.5, H2).astype(np.float32)
    s1 = rng.uniform(0.5, 2.0, H1).astype(np.float32); t1 = rng.standard_normal(H1).astype(np.float32) * 0.3
    s2 = rng.uniform(0.5, 2.0, H2).astype(np.float32); t2 = rng.standard_normal(H2).astype(np.float32) * 0.3
    ap = np.concatenate([a1, s1, t1, a2, s2, t2]).astype(np.float32)
    args = [_dev(table), _dev(hist), _dev(hl), _dev(cand)] + [_dev(w) for w in ws]
    got_o, got_s = ops.din_attention_pool(*args, normalize=normalize, want_scores=True, activation=activation, act_params=_dev(ap), arith="f16x2")
    import dir_amd.ops as O
    old = O.DIN_PACKED
    O.DIN_PACKED = False                         # the wave-per-sample kernel on the same inputs (itself held to the oracle in test_gpu_din_model.py)
    try:
        ref_o, ref_s = ops.din_attention_pool(*args, normalize=normalize, want_scores=True, activation=activation, act_params=_dev(ap), arith="f16x2")
    finally:
        O.DIN_PACKED = old
    _close(got_s.cpu().numpy(), ref_s.cpu().numpy().astype(np.float64), 2e-6)
    _close(got_o.cpu().numpy(), ref_o.cpu().numpy().astype(np.float64), 2e-6)


def test_packed_din_without_a_length_vector(ops, oracle):
    K, V, B, T, H1, H2 = 64, 1000, 77, 23, 80, 40
    rng = np.random.default_rng(5)
    table = (rng.standard_normal((V, K)) * 0.3).astype(np.float32)
    hist = rng.integers(-1, V, size=(B, T)).astype(np.int64)
    cand = rng.integers(0, V, size=B).astype(np.int64)
    ws = _weights(rng, K, H1, H2)
    ref_o, ref_s = oracle.din_attention_pool(table, hist, np.full(B, T, np.int32), cand, *ws, normalize=True, acc64=True)
    got_o, got_s = ops.din_attention_pool(_dev(table), _dev(hist), None, _dev(cand), *[_dev(w) for w in ws], normalize=True, want_scores=True,
                                          arith="f16x2")
    _close(got_s.cpu().numpy(), ref_s)
    _close(got_o.cpu().numpy(), ref_o)


def test_packed_din_full_batch_against_the_wave_kernel(ops):
    """BASELINE configs[3]'s batch (65 536 x T 50, lengths U{1..50}) on a smaller table: the packed kernel against the wave-per-sample kernel
    (same arithmetic, another summation order: 2e-6), bitwise equal reruns, and every sample written (the partition covers the batch)."""
    import dir_amd.ops as O
    K, V, B, T, H1, H2 = 64, 200000, 65536, 50, 80, 40
    g = torch.Generator(device="cuda").manual_seed(9)
    table = torch.randn((V, K), generator=g, device="cuda") * 0.125
    hist = torch.randint(0, V, (B, T), generator=g, device="cuda")
    hl = torch.randint(1, T + 1, (B,), generator=g, device="cuda", dtype=torch.int32)
    cand = torch.randint(0, V, (B,), generator=g, device="cuda")
    W1 = torch.randn((4 * K, H1), generator=g, device="cuda") * 0.05
    W2 = torch.randn((H1, H2), generator=g, device="cuda") * 0.1
    W3 = torch.randn((H2,), generator=g, device="cuda") * 0.1
    b1, b2, b3 = torch.zeros(H1, device="cuda"), torch.zeros(H2, device="cuda"), torch.zeros(1, device="cuda")
    for normalize in (True, False):
        got, sc = ops.din_attention_pool(table, hist, hl, cand, W1, b1, W2, b2, W3, b3, normalize=normalize, want_scores=True, arith="f16x2")
        again = ops.din_attention_pool(table, hist, hl, cand, W1, b1, W2, b2, W3, b3, normalize=normalize, arith="f16x2")
        assert torch.equal(got, again)
        old = O.DIN_PACKED
        O.DIN_PACKED = False
        try:
            ref, rsc = ops.din_attention_pool(table, hist, hl, cand, W1, b1, W2, b2, W3, b3, normalize=normalize, want_scores=True, arith="f16x2")
        finally:
            O.DIN_PACKED = old
        assert float(((got.double() - ref.double()).abs() / (1 + ref.double().abs())).max()) <= 2e-6
        assert float(((sc.double() - rsc.double()).abs() / (1 + rsc.double().abs())).max()) <= 2e-6


def test_packed_din_argument_errors(ops, built_lib):
    import ctypes
    lib = built_lib
    assert lib.dir_din_pack_workspace_bytes(65536, 0) >= 64 * 4 and lib.dir_din_pack_workspace_bytes(65536, 1) >= 65536 * 8
    assert lib.dir_din_pack_image_bytes() > 79872
    p = ctypes.c_void_p(256)
    rc = lib.dir_din_attention_pool_packed_f32(p, 32, p, p, p, 50, p, p, 80, p, p, 40, p, p, 0, 0, None, None, 4, p, None, p, 1 << 20, None)
    assert rc == -4 and b"covers K = 64" in lib.dir_last_error()
    rc = lib.dir_din_attention_pool_packed_f32(p, 64, p, p, p, 50, p, p, 80, p, p, 40, p, p, 0, 0, None, None, 4, p, None, p, 1, None)
    assert rc == -1 and b"workspace needs" in lib.dir_last_error()
    rc = lib.dir_din_attention_pool_packed_f32(p, 64, p, p, p, 50, p, p, 80, p, p, 40, p, p, 0, 0, None, None, 4, p, None, None, 1 << 20, None)
    assert rc == -1 and b"null pointer" in lib.dir_last_error()
    rc = lib.dir_din_attention_pool_packed_f32(p, 64, p, p, p, 50, None, p, 80, p, p, 40, p, p, 0, 0, None, None, 4, p, None, p, 1 << 20, None)
    assert rc == -1 and b"null pointer" in lib.dir_last_error()          # neither an image nor the weights to build one from
    rc = lib.dir_din_pack_weights_f32(p, p, 84, p, p, 40, p, 0, None, p, None)
    assert rc == -4 and b"H1 <= 80" in lib.dir_last_error()


def test_packed_din_image_follows_the_weights(ops):
    """The weight image is cached per version of the weights: an in-place update (a torch op, or a raw write reported through
    ops.mark_written) rebuilds it; calling with the entry's own image (NULL) gives the same bits."""
    import ctypes
    from dir_amd import _lib
    K, V, B, T, H1, H2 = 64, 5000, 257, 30, 80, 40
    g = torch.Generator(device="cuda").manual_seed(3)
    table = torch.randn((V, K), generator=g, device="cuda") * 0.2
    hist = torch.randint(0, V, (B, T), generator=g, device="cuda")
    hl = torch.randint(0, T + 1, (B,), generator=g, device="cuda", dtype=torch.int32)
    cand = torch.randint(0, V, (B,), generator=g, device="cuda")
    W1 = torch.randn((4 * K, H1), generator=g, device="cuda") * 0.1
    W2 = torch.randn((H1, H2), generator=g, device="cuda") * 0.2
    W3 = torch.randn((H2,), generator=g, device="cuda") * 0.3
    b1, b2, b3 = torch.randn(H1, generator=g, device="cuda") * 0.1, torch.randn(H2, generator=g, device="cuda") * 0.1, torch.zeros(1, device="cuda")
    a = ops.din_attention_pool(table, hist, hl, cand, W1, b1, W2, b2, W3, b3, normalize=True, arith="f16x2")
    n_img = len(ops._DIN_PACK_IMAGES)
    a2 = ops.din_attention_pool(table, hist, hl, cand, W1, b1, W2, b2, W3, b3, normalize=True, arith="f16x2")
    assert torch.equal(a, a2) and len(ops._DIN_PACK_IMAGES) == n_img
    # the entry with no image: the same bits
    lib = _lib.load()
    ws = torch.empty(int(lib.dir_din_pack_workspace_bytes(B, 0)), dtype=torch.uint8, device="cuda")
    out = torch.empty((B, K), device="cuda")
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    _lib.check(lib.dir_din_attention_pool_packed_f32(ptr(table), K, ptr(hist), ptr(hl), ptr(cand), T, ptr(W1), ptr(b1), H1, ptr(W2), ptr(b2), H2, ptr(W3),
                                                     ptr(b3), 1, 0, None, None, B, ptr(out), None, ptr(ws), ws.numel(), None))
    torch.cuda.synchronize()
    assert torch.equal(out, a)
    W2.mul_(1.5)                                  # a torch op bumps the version: a new image
    b = ops.din_attention_pool(table, hist, hl, cand, W1, b1, W2, b2, W3, b3, normalize=True, arith="f16x2")
    assert not torch.equal(a, b)
    W2.data.div_(1.5)                             # a raw write ...
    ops.mark_written(W2)                          # ... reported as the fused updaters do
    c = ops.din_attention_pool(table, hist, hl, cand, W1, b1, W2, b2, W3, b3, normalize=True, arith="f16x2")
    assert float((c - a).abs().max()) <= 1e-6
