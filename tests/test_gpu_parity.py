"""GPU parity tests: every HIP op, called through the C ABI (dir_amd.ops -> libdir_hip.so), against the
CPU oracle on the same seeded inputs.  Bar: bit-exact for index / copy / ordered-sum paths (gather, bags,
FM, linear term, hashing, bucketising, routing); |err| <= 1e-5 * (1 + |ref|) for the fp32 dot-product
paths (cross, DIN, CIN), the tolerance BASELINE.json's north_star states."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import np_ref as R  # noqa: E402

TOL = 1e-5


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _close(got, ref, tol=TOL):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    err = np.abs(got - ref) / (1.0 + np.abs(ref))
    assert err.max() <= tol, "max scaled err %.3e at %s" % (err.max(), np.unravel_index(err.argmax(), err.shape))


@pytest.fixture(scope="module")
def ops(built_lib):
    assert torch.cuda.is_available()
    from dir_amd import ops as _ops
    return _ops


def _tables(rng, F, V, K, scale=0.25):
    return [(rng.standard_normal((V, K)) * scale).astype(np.float32) for _ in range(F)]


@pytest.mark.parametrize("B,F,K,V", [(1, 1, 16, 10), (64, 26, 16, 1000), (1000, 26, 8, 10000), (4099, 26, 16, 5000),
                                      (333, 3, 64, 100), (257, 5, 4, 50), (130, 7, 12, 64), (100, 4, 1, 30),
                                      (77, 3, 6, 40), (65, 2, 128, 33), (19, 2, 256, 9),
                                      (300, 60, 16, 200), (129, 100, 8, 50), (64, 52, 16, 77), (200, 39, 16, 300)])   # wide slot counts
def test_gather_onehot_bit_exact(ops, oracle, B, F, K, V):
    rng = np.random.default_rng(B * 31 + K)
    tables = _tables(rng, F, V, K)
    ids = rng.integers(-1, V, size=(B, F)).astype(np.int64)  # includes pruned ids
    ts = ops.TableSet([_dev(t) for t in tables])
    ref = oracle.embedding_bag(tables, ids)
    got = ops.embedding_bag(ts, _dev(ids)).cpu().numpy()
    np.testing.assert_array_equal(got, ref)
    # field-major ids ([F,B] storage) through strides
    ids_fb = _dev(ids.T.copy()).t()
    np.testing.assert_array_equal(ops.embedding_bag(ts, ids_fb).cpu().numpy(), ref)
    # fused gather + FM: both outputs bit-exact
    emb, fm = ops.gather_fm(ts, _dev(ids))
    np.testing.assert_array_equal(emb.cpu().numpy(), ref)
    np.testing.assert_array_equal(fm.cpu().numpy()[:, 0], oracle.fm_second_order(ref, F, K))
    _, fm2 = ops.gather_fm(ts, ids_fb, want_emb=False)
    np.testing.assert_array_equal(fm2.cpu().numpy(), fm.cpu().numpy())
    # standalone FM on the materialised matrix
    np.testing.assert_array_equal(ops.fm_logit(emb, F, K).cpu().numpy()[:, 0], oracle.fm_second_order(ref, F, K))


@pytest.mark.parametrize("uf", ["4", "8", "13", "26"])
def test_gather_unroll_variants(ops, oracle, uf, monkeypatch):
    # the UF env knob is read once per process; exercise the variants through a subprocess-free path:
    # they are separate template instantiations selected by DIR_GATHER_UF at first launch.
    import subprocess, sys, os
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np, torch\n"
        "from dir_amd import ops\nfrom oracle import oracle as O\n"
        "rng=np.random.default_rng(5); F,K,V,B=26,16,777,1031\n"
        "t=[(rng.standard_normal((V,K))*0.25).astype(np.float32) for _ in range(F)]\n"
        "ids=rng.integers(-1,V,size=(B,F)).astype(np.int64)\n"
        "ts=ops.TableSet([torch.from_numpy(x).cuda() for x in t])\n"
        "e,f=ops.gather_fm(ts, torch.from_numpy(ids).cuda())\n"
        "r=O.embedding_bag(t,ids)\n"
        "assert np.array_equal(e.cpu().numpy(), r)\n"
        "assert np.array_equal(f.cpu().numpy()[:,0], O.fm_second_order(r,F,K))\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, DIR_GATHER_UF=uf)
    subprocess.run([sys.executable, "-c", code], check=True, env=env, timeout=300)


@pytest.mark.parametrize("combiner", ["sum", "mean", "sqrtn"])
@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("K", [16, 8, 6, 64])
def test_bag_csr_bit_exact(ops, oracle, combiner, weighted, K):
    rng = np.random.default_rng(99 + K)
    F, B, V = 5, 301, 200
    tables = _tables(rng, F, V, K)
    lens = rng.integers(0, 9, size=B * F)  # ragged, includes empty bags
    lens[::17] = 0
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ids = rng.integers(-1, V, size=offs[-1]).astype(np.int64)
    w = rng.uniform(-0.5, 2.0, size=offs[-1]).astype(np.float32) if weighted else None
    ts = ops.TableSet([_dev(t) for t in tables])
    comb = {"sum": 0, "mean": 1, "sqrtn": 2}[combiner]
    for flags in ([0, 1] if weighted else [0]):
        ref = oracle.embedding_bag(tables, ids, offsets=offs, weights=w, combiner=comb, flags=flags, B=B)
        got = ops.embedding_bag(ts, _dev(ids), offsets=_dev(offs), weights=None if w is None else _dev(w),
                                combiner=combiner, flags=flags).cpu().numpy()
        np.testing.assert_array_equal(got, ref)
    # field-major bag order (f*B + b): same bags, permuted CSR
    order = np.array([b * F + f for f in range(F) for b in range(B)])
    lens_fm = lens[order]
    offs_fm = np.concatenate([[0], np.cumsum(lens_fm)]).astype(np.int64)
    ids_fm = np.concatenate([ids[offs[i]:offs[i + 1]] for i in order]) if offs[-1] else ids
    w_fm = np.concatenate([w[offs[i]:offs[i + 1]] for i in order]) if weighted else None
    ref = oracle.embedding_bag(tables, ids, offsets=offs, weights=w, combiner=comb, B=B)
    got = ops.embedding_bag(ts, _dev(ids_fm), offsets=_dev(offs_fm), weights=None if w_fm is None else _dev(w_fm),
                            combiner=combiner, field_major=True).cpu().numpy()
    np.testing.assert_array_equal(got, ref)


def test_bag_edge_cases(ops, oracle):
    rng = np.random.default_rng(1)
    tab = _tables(rng, 1, 10, 16)
    ts = ops.TableSet([_dev(tab[0])])
    # all bags empty
    offs = np.zeros(5, np.int64)
    got = ops.embedding_bag(ts, _dev(np.zeros(1, np.int64)), offsets=_dev(offs)).cpu().numpy()
    np.testing.assert_array_equal(got, np.zeros((4, 16), np.float32))
    # all ids pruned; duplicates average to the row itself (SURVEY 8c KAT)
    ids = np.array([-1, -1, 3, -1, 3], np.int64)
    offs = np.array([0, 2, 5], np.int64)
    got = ops.embedding_bag(ts, _dev(ids), offsets=_dev(offs), combiner="mean").cpu().numpy()
    np.testing.assert_array_equal(got[0], np.zeros(16, np.float32))
    np.testing.assert_array_equal(got[1], tab[0][3])
    # B = 0
    assert ops.embedding_bag(ts, torch.zeros((0, 1), dtype=torch.int64, device="cuda")).shape == (0, 16)
    # id range check
    with pytest.raises(Exception, match="DIR_E_RANGE"):
        ops.check_ids(ts, _dev(np.array([[3], [10]], np.int64)))
    ops.check_ids(ts, _dev(np.array([[3], [-1], [9]], np.int64)))


@pytest.mark.parametrize("B,F,V", [(1, 1, 5), (1000, 39, 10000), (4097, 26, 1000)])
def test_linear_term_bit_exact(ops, oracle, B, F, V):
    rng = np.random.default_rng(B + F)
    wts = [(rng.standard_normal(V) * 0.1).astype(np.float32) for _ in range(F)]
    ids = rng.integers(-1, V, size=(B, F)).astype(np.int64)
    bias = np.array([0.3], np.float32)
    ts = ops.TableSet([_dev(w) for w in wts])
    ref = oracle.linear_sparse_sum(wts, ids, bias=bias)
    got = ops.linear_logit(ts, _dev(ids), bias=_dev(bias)).cpu().numpy()[:, 0]
    np.testing.assert_array_equal(got, ref)
    # accumulate into an existing logit
    base = rng.standard_normal(B).astype(np.float32)
    out = _dev(base.reshape(B, 1).copy())
    ops.linear_logit(ts, _dev(ids), out=out, accumulate=True)
    np.testing.assert_array_equal(out.cpu().numpy()[:, 0], oracle.linear_sparse_sum(wts, ids, out=base))
    # multi-hot weighted
    lens = rng.integers(0, 5, size=B * F)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    mids = rng.integers(-1, V, size=max(int(offs[-1]), 1)).astype(np.int64)
    ew = rng.uniform(0.1, 2, size=mids.size).astype(np.float32)
    for comb, name in [(0, "sum"), (1, "mean"), (2, "sqrtn")]:
        ref = oracle.linear_sparse_sum(wts, mids, offsets=offs, entry_weights=ew, combiner=comb, bias=bias, B=B)
        got = ops.linear_logit(ts, _dev(mids), offsets=_dev(offs), entry_weights=_dev(ew), combiner=name,
                               bias=_dev(bias)).cpu().numpy()[:, 0]
        np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("B,d,L", [(1, 4, 1), (64, 416, 3), (1000, 429, 3), (513, 51, 2), (300, 64, 6), (129, 1024, 2),
                                    (50, 2048, 1), (77, 10, 0)])
def test_cross_network(ops, oracle, B, d, L):
    rng = np.random.default_rng(d * 7 + L)
    x0 = (rng.standard_normal((B, d)) * 0.25).astype(np.float32)
    w = np.clip(rng.standard_normal((max(L, 1), d)) * 0.1, -0.2, 0.2).astype(np.float32)[:L]
    b = np.clip(rng.standard_normal((max(L, 1), d)) * 0.1, -0.2, 0.2).astype(np.float32)[:L]
    ref = oracle.dcn_cross(x0, w.reshape(L, d), b.reshape(L, d), acc64=True)
    got = ops.cross_network(_dev(x0), _dev(w.reshape(L, d)), _dev(b.reshape(L, d))).cpu().numpy()
    _close(got, ref)
    if L >= 1:  # the literal one-layer _cross_op on a distinct x
        x = (rng.standard_normal((B, d)) * 0.25).astype(np.float32)
        ref1 = R.cross_op(x0.astype(np.float64), x.astype(np.float64), w[0].astype(np.float64), b[0].astype(np.float64))
        _close(ops.cross_op(_dev(x0), _dev(x), _dev(w[0]), _dev(b[0])).cpu().numpy(), ref1)


def test_cross_hand_kat(ops):
    x0 = _dev(np.array([[1, 2]], np.float32))
    w = _dev(np.array([[.5, -1], [.25, .5]], np.float32))
    b = _dev(np.array([[.1, .2], [0, -.1]], np.float32))
    np.testing.assert_allclose(ops.cross_network(x0, w, b).cpu().numpy(), [[-0.9, -1.9]], rtol=1e-6)


@pytest.mark.parametrize("B,T,K,H1,H2", [(33, 50, 64, 80, 40), (7, 5, 8, 12, 8), (100, 13, 16, 20, 12), (4, 70, 32, 36, 16),
                                            (65, 64, 32, 40, 16), (130, 64, 64, 80, 40), (20, 17, 64, 72, 36), (9, 1, 64, 80, 40),
                                            # hidden layers narrower than the instantiated tile counts (zero-padded columns)
                                            (40, 50, 64, 64, 32), (40, 33, 64, 16, 4), (40, 50, 32, 20, 8), (40, 21, 16, 12, 4),
                                            (25, 40, 64, 48, 48)])
@pytest.mark.parametrize("normalize", [False, True])
@pytest.mark.parametrize("arith", ["f16x2", "bf16x3", "f32"])
def test_din_attention_pool(ops, oracle, B, T, K, H1, H2, normalize, arith, monkeypatch):
    """The three arithmetics (fp16 x 2: the default since round 4; bf16 x 3; fp32 MFMA) of the (K 64, H1 <= 80, H2 <= 48) wave-per-sample kernel (DIR_DIN_ARITH, read per call) against the
    double-accumulating oracle at the same 1e-5 bar; the other shape classes have one kernel and ignore the switch."""
    monkeypatch.setenv("DIR_DIN_ARITH", arith)
    rng = np.random.default_rng(T * 3 + K)
    V = 500
    table = (rng.standard_normal((V, K)) * 0.3).astype(np.float32)
    hist = rng.integers(-1, V, size=(B, T)).astype(np.int64)
    hl = rng.integers(0, T + 1, size=B).astype(np.int32)
    hl[0] = 0  # a sample with no valid history -> zeros
    cand = rng.integers(0, V, size=B).astype(np.int64)
    W1 = (rng.standard_normal((4 * K, H1)) * 0.1).astype(np.float32); b1 = (rng.standard_normal(H1) * 0.1).astype(np.float32)
    W2 = (rng.standard_normal((H1, H2)) * 0.2).astype(np.float32); b2 = (rng.standard_normal(H2) * 0.1).astype(np.float32)
    W3 = (rng.standard_normal(H2) * 0.3).astype(np.float32); b3 = np.array([0.05], np.float32)
    ref_o, ref_s = oracle.din_attention_pool(table, hist, hl, cand, W1, b1, W2, b2, W3, b3, normalize=normalize, acc64=True)
    got_o, got_s = ops.din_attention_pool(_dev(table), _dev(hist), _dev(hl), _dev(cand), _dev(W1), _dev(b1), _dev(W2),
                                          _dev(b2), _dev(W3), _dev(b3), normalize=normalize, want_scores=True)
    _close(got_s.cpu().numpy(), ref_s)
    _close(got_o.cpu().numpy(), ref_o)
    assert not got_o.cpu().numpy()[0].any()


@pytest.mark.parametrize("B,m,D,Hp,H", [(64, 26, 16, 26, 128), (37, 26, 16, 128, 128), (16, 26, 16, 7, 40), (21, 8, 8, 5, 32),
                                         (9, 5, 4, 6, 7), (5, 5, 32, 3, 64), (3, 8, 16, 9, 200),
                                         # field counts that are not an instantiated size (padded with zero operands)
                                         (33, 39, 16, 39, 128), (17, 39, 16, 64, 96), (12, 13, 16, 13, 64), (8, 3, 8, 3, 32),
                                         (10, 22, 16, 50, 128), (6, 40, 16, 2, 16), (5, 1, 4, 1, 8),
                                         # tails: half a chunk (fast clamped staging + half burst sequence) and others
                                         (20, 26, 16, 6, 128), (20, 26, 16, 10, 128), (11, 16, 16, 6, 64), (9, 8, 8, 14, 32),
                                         (7, 40, 16, 3, 128), (7, 40, 16, 9, 128), (20, 26, 16, 5, 128), (20, 26, 16, 7, 128),
                                         # column-block splits (csrc/cin.hip dir_cin_layer_f32): 3 | 3+2 | 4+2 | 4+3 | 4+3+2 | 4+4+3 tiles
                                         (19, 26, 16, 26, 96), (19, 26, 16, 12, 70), (10, 26, 16, 8, 160), (10, 26, 16, 200, 200),
                                         (6, 26, 16, 8, 190), (6, 16, 8, 8, 224), (5, 26, 16, 4, 270), (5, 8, 16, 6, 350), (7, 39, 16, 4, 90)])
def test_cin_layer(ops, oracle, B, m, D, Hp, H):
    """Both arithmetics of the layer against the double-accumulating oracle at the same 1e-5 bar: dir_cin_layer_f32 (fp32 MFMA) on
    every shape, dir_cin_layer_bf16x3_f32 (three-way bf16 split, csrc/cin_bf3.hip) on the shapes it covers -- which is also what
    the default arith="auto" runs there."""
    rng = np.random.default_rng(Hp * 13 + H)
    x0 = (rng.standard_normal((B, m, D)) * 0.5).astype(np.float32)
    xk = (rng.standard_normal((B, Hp, D)) * 0.5).astype(np.float32)
    W = (rng.standard_normal((H, Hp * m)) * (1.0 / np.sqrt(Hp * m))).astype(np.float32)
    ref_x, ref_p = oracle.cin_layer(x0, xk, W, acc64=True)
    got_x, got_p = ops.cin_layer(_dev(x0), _dev(xk), _dev(W), arith="f32")
    _close(got_x.cpu().numpy(), ref_x)
    _close(got_p.cpu().numpy(), ref_p)
    assert ops.cin_bf16x3_covers(m, D)
    bx, bp = ops.cin_layer(_dev(x0), _dev(xk), _dev(W), arith="bf16x3")
    _close(bx.cpu().numpy(), ref_x)
    _close(bp.cpu().numpy(), ref_p)
    # round 4: the fp16 x 2 forward (two fp16 pieces per operand, three products: dir_cin_layer_f16x2_f32) at the same bar
    fx, fp_ = ops.cin_layer(_dev(x0), _dev(xk), _dev(W), arith="f16x2")
    _close(fx.cpu().numpy(), ref_x)
    _close(fp_.cpu().numpy(), ref_p)
    fx2, _ = ops.cin_layer(_dev(x0), _dev(xk), _dev(W), arith="f16x2")
    assert torch.equal(fx, fx2)                                      # rerun: bitwise equal
    ax, ap = ops.cin_layer(_dev(x0), _dev(xk), _dev(W))            # "auto" is one of them, bitwise: since round 5 the ROW-SCALED fp16 x 2 form
    rx, rp = ops.cin_layer(_dev(x0), _dev(xk), _dev(W), arith="f16x2_grad")      # (rows of xk and the tensor W scaled by exact powers of two)
    split = (rx, rp) if ops.CIN_FWD_SPLIT == "f16x2" else ((fx, fp_) if ops.CIN_FWD_SPLIT == "f16x2_unscaled" else (bx, bp))
    want = split if ops.cin_auto_arith(m, D, Hp, H) == "bf16x3" else (got_x, got_p)
    assert torch.equal(ax, want[0]) and torch.equal(ap, want[1])
    if H <= 128 and ops.CIN_FWD_SPLIT == "f16x2" and ops.cin_auto_arith(m, D, Hp, H) == "bf16x3":
        # DIR_CIN_ROW_BITS_CARRY: the layer leaves its output's row maxima on the tensor (the next layer's row scales): exact, and consumed
        # bit for bit like a scan of the rows
        keep = ops.CIN_ROW_BITS_CARRY
        ops.CIN_ROW_BITS_CARRY = True
        try:
            cx, cp = ops.cin_layer(_dev(x0), _dev(xk), _dev(W))
            bits = ops._row_bits_hint(cx, B * D)
            assert torch.equal(cx, ax) and torch.equal(cp, ap)
            assert bits is not None and torch.equal(bits.view(B, D), cx.abs().amax(dim=1).view(torch.int32))
            if Hp == H:
                # the next layer takes its device-side verdict from them (dir_cin_layer_auto_f16x2_f32): inside the magnitude window the
                # plain fp16 x 2 kernel's bits, outside it the row-scaled kernel's
                nx, np_ = ops.cin_layer(_dev(x0), cx, _dev(W))
                xmax = float(cx.abs().max())
                if 2.0 ** -4 <= xmax < 2.0 ** 15:        # the plain kernel on the tensor-scaled W image: its own bits, at the same bar
                    _close(nx.cpu().numpy(), oracle.cin_layer(x0, cx.cpu().numpy(), W, acc64=True)[0])
                    nx2, _ = ops.cin_layer(_dev(x0), cx, _dev(W))
                    assert torch.equal(nx, nx2)
                else:
                    ex, ep = ops.cin_layer(_dev(x0), cx.clone(), _dev(W), arith="f16x2_grad")
                    assert torch.equal(nx, ex) and torch.equal(np_, ep)
                big = cx * 2.0 ** 20                                        # outside the window: the row-scaled kernel, the same rows' maxima scaled
                big._dir_row_bits = (bits + (20 << 23), big._version)
                bx, _ = ops.cin_layer(_dev(x0), big, _dev(W))
                rx2, _ = ops.cin_layer(_dev(x0), big.clone(), _dev(W), arith="f16x2_grad")
                assert torch.equal(bx, rx2) and bool(torch.isfinite(bx).all())
        finally:
            ops.CIN_ROW_BITS_CARRY = keep
    # a gradient as the left operand never takes the PLAIN fp16 split: it runs the row-scaled form (dir_cin_layer_grad_f16x2_f32: every row
    # of xk times a power of two inside the kernel) or bf16 x 3.  Same bar; rows that differ by powers of two over 40 binades give the same
    # bits times those powers; tiny (1e-30-scale) and large (1e+6-scale) operands are as exact as O(1) ones
    sx, sp = ops.cin_layer(_dev(x0), _dev(xk), _dev(W), arith="f16x2_grad")
    _close(sx.cpu().numpy(), ref_x)
    _close(sp.cpu().numpy(), ref_p)
    pw = torch.from_numpy(np.ldexp(1.0, rng.integers(-80, 20, size=(B, 1, 1))).astype(np.float32)).cuda()
    px, _ = ops.cin_layer(_dev(x0), _dev(xk) * pw, _dev(W), arith="f16x2_grad")
    assert torch.equal(px, sx * pw)
    gx, gp = ops.cin_layer(_dev(x0), _dev(xk), _dev(W), grad_operand=True)
    want_g = ((sx, sp) if ops.CIN_BWD_SPLIT == "f16x2" else (bx, bp)) if ops.cin_auto_arith(m, D, Hp, H) == "bf16x3" else (got_x, got_p)
    assert torch.equal(gx, want_g[0]) and torch.equal(gp, want_g[1])


def test_bf16x3_kernels_on_empty_and_single_row_batches(ops):
    """B = 0 is a no-op that returns empty results (dW: zeros) and B = 1 (a single partial workgroup) is exact to the bar, on every
    bf16x3 entry point: CIN forward, data gradients, weight gradient, dense layer."""
    m, D, Hp, H = 26, 16, 128, 128
    g = torch.Generator(device="cuda").manual_seed(11)
    W = torch.randn((H, Hp * m), generator=g, device="cuda") / (Hp * m) ** 0.5
    for B in (0, 1):
        x0 = torch.randn((B, m, D), generator=g, device="cuda") * 0.5
        xk = torch.randn((B, Hp, D), generator=g, device="cuda") * 0.5
        G = torch.randn((B, H, D), generator=g, device="cuda") * 0.5
        xo, po = ops.cin_layer(x0, xk, W, arith="bf16x3")
        dxk, dx0 = ops.cin_dx_bf16x3(x0, xk, W, G)
        dW = ops.cin_dw(x0, xk, G, arith="bf16x3")
        assert xo.shape == (B, H, D) and po.shape == (B, H) and dxk.shape == (B, Hp, D) and dx0.shape == (B, m, D) and dW.shape == (H, Hp * m)
        if B == 0:
            assert float(dW.abs().max()) == 0.0
        else:
            fo, fp_ = ops.cin_layer(x0, xk, W, arith="f32")
            f0, fk, fW = ops.cin_layer_backward(x0, xk, W, G, arith="f32")
            for a, b in ((xo, fo), (po, fp_), (dxk, fk), (dx0, f0), (dW, fW)):          # two fp32-accurate kernels against each other
                assert float(((a - b).abs() / (1 + b.abs())).max()) <= 2e-5
    w = torch.randn((400, 416), generator=g, device="cuda") / 20.0
    assert ops.dense(torch.zeros((0, 416), device="cuda"), w, arith="bf16x3").shape == (0, 400)
    x1 = torch.randn((1, 416), generator=g, device="cuda")
    assert float((ops.dense(x1, w, arith="bf16x3") - ops.dense(x1, w, arith="f32")).abs().max()) <= 1e-5


def test_cin_bf16x3_refuses_uncovered_shapes(ops):
    x0 = torch.zeros((4, 41, 16), device="cuda"); xk = torch.zeros((4, 3, 16), device="cuda"); W = torch.zeros((8, 3 * 41), device="cuda")
    assert not ops.cin_bf16x3_covers(41, 16) and ops.cin_auto_arith(41, 16, 3, 8) == "f32"
    with pytest.raises(ValueError):
        ops.cin_layer(x0, xk, W, arith="bf16x3")
    assert ops.cin_auto_arith(26, 16, 128, 128) == "bf16x3" and ops.cin_auto_arith(26, 16, 26, 128) == "bf16x3"
    assert ops.cin_auto_arith(26, 16, 128, 32) == "bf16x3" and ops.cin_auto_arith(26, 16, 200, 200) == "bf16x3"
    assert ops.cin_auto_arith(26, 16, 7, 128) == "f32" and ops.cin_auto_arith(26, 16, 128, 10) == "f32"


@pytest.mark.parametrize("B,m,D,Hp,H", [(300, 26, 16, 128, 128), (257, 15, 8, 9, 33), (64, 17, 4, 24, 129), (31, 40, 32, 8, 256),
                                         (130, 16, 16, 1, 5), (65, 33, 16, 17, 100), (1, 26, 16, 26, 128), (513, 26, 16, 100, 64),
                                         (40, 1, 8, 70, 20), (77, 3, 4, 33, 130), (19, 8, 32, 64, 128), (260, 13, 16, 65, 16),
                                         # narrow last column blocks (2 / 4 / 6 tiles) and short last halves (Hp % 64 in 1..32)
                                         (100, 26, 16, 200, 200), (70, 26, 16, 96, 160), (33, 20, 8, 72, 224), (50, 26, 16, 130, 32), (64, 16, 16, 160, 288)])
def test_cin_layer_bf16x3_shapes(ops, oracle, B, m, D, Hp, H):
    """csrc/cin_bf3.hip on its own edge shapes: one field, odd field counts, Hp on both sides of the 32 / 64-wide i blocks, several
    column blocks, partial last workgroups, D = 4 .. 32, and operands of very different magnitudes (the split keeps fp32's
    exponent range: the error bar is relative to the terms, not to 1)."""
    rng = np.random.default_rng(B * 7 + Hp)
    for scale in (1.0, 1e-3, 64.0):
        x0 = (rng.standard_normal((B, m, D)) * 0.5 * scale).astype(np.float32)
        xk = (rng.standard_normal((B, Hp, D)) * 0.5).astype(np.float32)
        W = (rng.standard_normal((H, Hp * m)) * (1.0 / np.sqrt(Hp * m))).astype(np.float32)
        ref_x, ref_p = oracle.cin_layer(x0, xk, W, acc64=True)
        bx, bp = ops.cin_layer(_dev(x0), _dev(xk), _dev(W), arith="bf16x3")
        err_x = np.abs(bx.cpu().double().numpy() - ref_x) / (scale + np.abs(ref_x))
        err_p = np.abs(bp.cpu().double().numpy() - ref_p) / (scale + np.abs(ref_p))
        assert err_x.max() <= 1e-5 and err_p.max() <= 1e-5, (scale, err_x.max(), err_p.max())
    # pooled-only call: the pooled form (sum over d first, csrc/cin_pool.hip + the dense kernel) at the same bar; on the layer kernel
    # (DIR_CIN_POOLED_LAST=0) bitwise the pooled sums of the full call
    none, only = ops.cin_layer(_dev(x0), _dev(xk), _dev(W), want_xout=False, arith="bf16x3")
    assert none is None
    err_o = np.abs(only.cpu().double().numpy() - ref_p) / (scale + np.abs(ref_p))
    assert err_o.max() <= 1e-5, err_o.max()
    old = ops.CIN_POOLED_LAST
    ops.CIN_POOLED_LAST = False
    try:
        none, only = ops.cin_layer(_dev(x0), _dev(xk), _dev(W), want_xout=False, arith="bf16x3")
    finally:
        ops.CIN_POOLED_LAST = old
    assert none is None and torch.equal(only, bp)


def test_cin_pooled_only(ops, monkeypatch):
    """xout = NULL (last layer of a stack) on the layer kernels: the pooled sums are bit-identical to the full call, and a
    call with neither output is a BADARG.  (The default pooled-only path is the pooled form: test_cin_pooled_form_of_the_last_layer.)"""
    monkeypatch.setattr(ops, "CIN_POOLED_LAST", False)
    rng = np.random.default_rng(77)
    for B, m, D, Hp, H in [(300, 26, 16, 26, 128), (37, 10, 8, 7, 40), (5, 3, 4, 2, 3)]:
        x0 = _dev((rng.standard_normal((B, m, D)) * 0.5).astype(np.float32))
        xk = _dev((rng.standard_normal((B, Hp, D)) * 0.5).astype(np.float32))
        W = _dev((rng.standard_normal((H, Hp * m)) / np.sqrt(Hp * m)).astype(np.float32))
        _, full = ops.cin_layer(x0, xk, W)
        none, only = ops.cin_layer(x0, xk, W, want_xout=False)
        assert none is None
        np.testing.assert_array_equal(only.cpu().numpy(), full.cpu().numpy())


def test_cin_identity_layout(ops):
    """A = I style check with an ASYMMETRIC weight: catches a transposed C/D or operand map."""
    B, m, D, Hp, H = 2, 8, 16, 8, 32
    x0 = np.zeros((B, m, D), np.float32); xk = np.zeros((B, Hp, D), np.float32)
    x0[:, 3, :] = np.arange(1, D + 1)          # only field j=3 is non-zero, value d+1
    xk[0, 5, :] = 2.0; xk[1, 2, :] = 3.0       # sample 0 uses i=5, sample 1 uses i=2
    W = np.arange(H * Hp * m, dtype=np.float32).reshape(H, Hp * m) / 64.0
    got, _ = ops.cin_layer(_dev(x0), _dev(xk), _dev(W))
    got = got.cpu().numpy()
    for b, (i, s) in enumerate([(5, 2.0), (2, 3.0)]):
        exp = W[:, i * m + 3][:, None] * s * np.arange(1, D + 1)[None, :]
        np.testing.assert_allclose(got[b], exp, rtol=1e-6)


def test_id_paths_bit_exact(ops, oracle):
    rng = np.random.default_rng(2)
    keys = np.concatenate([rng.integers(-10**6, 10**6, 500), rng.integers(-2**62, 2**62, 500),
                           [0, -1, 9, 10, 99999999, 10**15, 10**16, 10**17, -10**17, 2**63 - 1, -2**63]]).astype(np.int64)
    for nb in (1000, 3, 10**6 + 3):
        got = ops.hash_bucket_ints(_dev(keys), nb).cpu().numpy()
        np.testing.assert_array_equal(got, R.hash_bucket_int(keys, nb))
    # per-field bucket counts over a [B, F] key matrix
    km = rng.integers(-2**40, 2**40, size=(257, 5)).astype(np.int64)
    nbf = np.array([1000, 3, 10**6 + 3, 1, 977], np.int64)
    got = ops.hash_bucket_ints_fields(_dev(km), _dev(nbf)).cpu().numpy()
    for f in range(5):
        np.testing.assert_array_equal(got[:, f], R.hash_bucket_int(km[:, f], int(nbf[f])))
    # byte strings on the device: every FarmHash length branch (0..16, 17..32, 33..64, > 64)
    strs = [bytes(rng.integers(0, 256, size=n, dtype=np.uint8).tolist()) for n in list(range(0, 70)) + [64, 65, 127, 128, 129, 200, 777]]
    strs += [b"Hello", b"TensorFlow", b"2.x", b"1footrue"]
    offs = np.concatenate([[0], np.cumsum([len(x) for x in strs])]).astype(np.int64)
    buf = np.frombuffer(b"".join(strs) or b"\0", dtype=np.uint8).copy()
    got = ops.hash_bucket_bytes(_dev(buf), _dev(offs), 1000003).cpu().numpy()
    np.testing.assert_array_equal(got, np.array([R.fingerprint64(x) % 1000003 for x in strs], np.int64))
    assert ops.hash_bucket_bytes(_dev(buf), _dev(offs), 3).cpu().numpy()[-4:-1].tolist() == [0, 2, 2]   # TF doc example
    x = rng.uniform(-2, 12, 5000).astype(np.float32)
    bd = np.array([0.0, 1.0, 2.5, 2.5, 7.0, 10.0], np.float32)
    x[:6] = bd
    np.testing.assert_array_equal(ops.bucketize(_dev(x), _dev(bd)).cpu().numpy(), oracle.bucketize(x, bd))
    for V, P in [(10, 4), (1000000, 8), (100000000, 8), (7, 8), (999983, 3)]:
        ids = np.concatenate([rng.integers(0, V, 3000), [0, V - 1, -1, -5]]).astype(np.int64)
        own, loc = ops.shard_route(_dev(ids), _dev(np.array([V], np.int64)), P)
        ro, rl = R.shard_div_owner(np.maximum(ids, 0), V, P)
        ok = ids >= 0
        np.testing.assert_array_equal(own.cpu().numpy()[ok], ro[ok])
        np.testing.assert_array_equal(loc.cpu().numpy()[ok], rl[ok])
        assert (loc.cpu().numpy()[~ok] == -1).all()


def test_full_size_properties(ops, oracle):
    """BASELINE config 2 sizes (B=65536, F=26, V=1e6, K=16) through size-independent properties:
    (1) the gather is a pure copy: a checksum of row checksums equals the one computed from the ids on the
        host; (2) FM linearity in scale: fm(c*E) == c^2 fm(E) for c a power of two (exact in fp32);
    (3) the fused kernel's two outputs agree with gather-then-FM bit for bit; (4) a strided sample of
        rows matches the oracle bit for bit."""
    B, F, K, V = 65536, 26, 16, 1000000
    g = torch.Generator(device="cuda").manual_seed(1234)
    tables = [torch.randn((V, K), generator=g, device="cuda") * 0.25 for _ in range(F)]
    ids = torch.randint(0, V, (B, F), generator=g, device="cuda")
    ts = ops.TableSet(tables)
    emb, fm = ops.gather_fm(ts, ids)
    emb2 = ops.embedding_bag(ts, ids)
    assert torch.equal(emb, emb2)
    assert torch.equal(fm, ops.fm_logit(emb2, F, K))
    # (1) checksum of checksums: per-table row sums in fp64, gathered by id, vs row sums of the output
    rs = torch.stack([t.double().sum(1) for t in tables])                       # [F, V]
    want = rs.gather(1, ids.t()).t()                                              # [B, F]
    have = emb.view(B, F, K).double().sum(2)
    assert torch.equal(want, have)
    # (2) exact scaling
    ts4 = ops.TableSet([t * 4.0 for t in tables])
    _, fm4 = ops.gather_fm(ts4, ids, want_emb=False)
    assert torch.equal(fm4, fm * 16.0)
    # (4) sampled rows against the oracle
    sel = torch.arange(0, B, 997, device="cuda")
    ids_s = ids[sel].cpu().numpy()
    sub = [t.cpu().numpy() for t in tables]
    ref = oracle.embedding_bag(sub, ids_s)
    np.testing.assert_array_equal(emb[sel].cpu().numpy(), ref)
    np.testing.assert_array_equal(fm[sel].cpu().numpy()[:, 0], oracle.fm_second_order(ref, F, K))


@pytest.mark.parametrize("B,F,K,V", [(1, 1, 16, 7), (513, 26, 16, 1000), (300, 5, 8, 64), (129, 3, 64, 50), (77, 4, 32, 33), (64, 2, 4, 9)])
def test_packed_rows_gather_fm_linear(ops, oracle, B, F, K, V):
    """Packed serving layout: emb / fm / lin bit-identical to the reference-layout path and to the oracle."""
    rng = np.random.default_rng(B + 3 * K)
    tables = _tables(rng, F, V, K)
    lws = [(rng.standard_normal(V) * 0.1).astype(np.float32) for _ in range(F)]
    ids = rng.integers(-1, V, size=(B, F)).astype(np.int64)
    bias = np.array([0.25], np.float32)
    pt = ops.PackedTables([_dev(t) for t in tables], [_dev(w) for w in lws])
    assert all(r.data_ptr() % 128 == 0 for r in pt.rows) and pt.ld * 4 % 128 == 0
    emb, fm, lin = ops.gather_fm_linear(pt, _dev(ids), bias=_dev(bias))
    ref = oracle.embedding_bag(tables, ids)
    np.testing.assert_array_equal(emb.cpu().numpy(), ref)
    np.testing.assert_array_equal(fm.cpu().numpy()[:, 0], oracle.fm_second_order(ref, F, K))
    np.testing.assert_array_equal(lin.cpu().numpy()[:, 0], oracle.linear_sparse_sum(lws, ids, bias=bias))
    # field-major ids, no concat output
    _, fm2, lin2 = ops.gather_fm_linear(pt, _dev(ids.T.copy()).t(), bias=_dev(bias), want_emb=False)
    assert torch.equal(fm2, fm) and torch.equal(lin2, lin)


def test_full_size_properties_cross_cin_din(ops, oracle):
    """BASELINE configs 3-5 at full size (B = 65536) through size-independent properties + sampled rows against the oracle.
    cross: with w = 0 every layer is x + b (exact); scaling x0 and b by 2 scales... only the affine part, so use w = 0 there and
    sampled rows for the general case.  CIN: xout is linear in xk -- scaling xk by a power of two scales xout exactly; pooled is
    the d-sum of xout.  DIN (normalize): the weights of a sample sum to 1 and out is that convex combination of its history rows."""
    B = 65536
    g = torch.Generator(device="cuda").manual_seed(99)
    # ---- cross, d = 416, L = 3 ----------------------------------------------------------------------------------------
    d, L = 416, 3
    x0 = torch.randn((B, d), generator=g, device="cuda") * 0.25
    w = (torch.randn((L, d), generator=g, device="cuda") * 0.1).clamp_(-0.2, 0.2)
    b = (torch.randn((L, d), generator=g, device="cuda") * 0.1).clamp_(-0.2, 0.2)
    out0 = ops.cross_network(x0, torch.zeros_like(w), b)
    want = x0
    for l in range(L):
        want = ((x0 * 0.0) + b[l]) + want
    assert torch.equal(out0, want)
    out = ops.cross_network(x0, w, b)
    sel = torch.arange(0, B, 1499, device="cuda")
    ref = oracle.dcn_cross(x0[sel].cpu().numpy(), w.cpu().numpy(), b.cpu().numpy(), acc64=True)
    err = np.abs(out[sel].cpu().double().numpy() - ref) / (1 + np.abs(ref))
    assert err.max() <= 1e-5
    # ---- CIN, m = 26, D = 16, Hp = H = 128 ------------------------------------------------------------------------------
    m, D, H = 26, 16, 128
    c0 = torch.randn((B, m, D), generator=g, device="cuda") * 0.25
    xk = torch.randn((B, H, D), generator=g, device="cuda") * 0.25
    W = torch.randn((H, H * m), generator=g, device="cuda") * (1.0 / (H * m) ** 0.5)
    xo, po = ops.cin_layer(c0, xk, W)
    xo2, po2 = ops.cin_layer(c0, xk * 2.0, W)
    # linearity in xk under a power of two: EXACT for the bf16 x 3 split (the pieces of 2 x are twice the pieces of x: fp32's exponent
    # range), and within the fp16 x 2 split's error for the default forward arithmetic of round 4 (a second piece that is an fp16
    # subnormal rounds differently after the scaling): checked at 2e-6 there, bitwise for arith="bf16x3"
    assert float(((xo2 - xo * 2.0).abs() / (1 + xo2.abs())).max()) <= 2e-6 and float(((po2 - po * 2.0).abs() / (1 + po2.abs())).max()) <= 2e-6
    bo, bp_ = ops.cin_layer(c0, xk, W, arith="bf16x3")
    bo2, bp2 = ops.cin_layer(c0, xk * 2.0, W, arith="bf16x3")
    assert torch.equal(bo2, bo * 2.0) and torch.equal(bp2, bp_ * 2.0)
    assert float(((bo - xo).abs() / (1 + bo.abs())).max()) <= 2e-6       # fp16 x 2 against bf16 x 3, every element of the full-size layer
    del bo, bp_, bo2, bp2
    assert torch.allclose(po, xo.sum(2), rtol=1e-5, atol=1e-6)
    sel = torch.arange(0, B, 4099, device="cuda")
    rx, rp = oracle.cin_layer(c0[sel].cpu().numpy(), xk[sel].cpu().numpy(), W.cpu().numpy(), acc64=True)
    assert (np.abs(xo[sel].cpu().double().numpy() - rx) / (1 + np.abs(rx))).max() <= 1e-5
    assert (np.abs(po[sel].cpu().double().numpy() - rp) / (1 + np.abs(rp))).max() <= 1e-5
    # (the default "auto" ran the fp16 x 2 kernel above; the fp32-MFMA kernel at the same size, same checks)
    fo, fp_ = ops.cin_layer(c0, xk, W, arith="f32")
    assert (np.abs(fo[sel].cpu().double().numpy() - rx) / (1 + np.abs(rx))).max() <= 1e-5
    assert (np.abs(fp_[sel].cpu().double().numpy() - rp) / (1 + np.abs(rp))).max() <= 1e-5
    assert float(((fo - xo).abs() / (1 + fo.abs())).max()) <= 1e-5       # the two arithmetics against each other, every element
    for _ in range(3):                                                    # run-to-run bitwise reproducibility of the bf16x3 kernel at full size
        xr, pr = ops.cin_layer(c0, xk, W)
        assert torch.equal(xr, xo) and torch.equal(pr, po)
    del xr, pr
    del xo, xo2, po2, xk, fo, fp_
    # ---- DIN, T = 50, K = 64, 80-40-1, normalised -----------------------------------------------------------------------
    T, K, V, H1, H2 = 50, 64, 1000000, 80, 40
    table = torch.randn((V, K), generator=g, device="cuda") * 0.125
    hist = torch.randint(0, V, (B, T), generator=g, device="cuda")
    hl = torch.randint(0, T + 1, (B,), generator=g, device="cuda", dtype=torch.int32)
    cand = torch.randint(0, V, (B,), generator=g, device="cuda")
    W1 = torch.randn((4 * K, H1), generator=g, device="cuda") * 0.05
    W2 = torch.randn((H1, H2), generator=g, device="cuda") * 0.1
    W3 = torch.randn((H2,), generator=g, device="cuda") * 0.1
    b1, b2, b3 = torch.zeros(H1, device="cuda"), torch.zeros(H2, device="cuda"), torch.zeros(1, device="cuda")
    out, sc = ops.din_attention_pool(table, hist, hl, cand, W1, b1, W2, b2, W3, b3, normalize=True, want_scores=True)
    ssum = sc.sum(1)
    nonempty = hl > 0
    assert torch.allclose(ssum[nonempty], torch.ones_like(ssum[nonempty]), atol=1e-5)
    assert (sc[~nonempty] == 0).all() and (out[~nonempty] == 0).all()
    mask = torch.arange(T, device="cuda").unsqueeze(0) < hl.unsqueeze(1)
    assert (sc[~mask] == 0).all()
    sel = torch.arange(0, B, 257, device="cuda")
    comb = (sc[sel].unsqueeze(2) * table[hist[sel]]).sum(1)
    assert torch.allclose(out[sel], comb, rtol=1e-5, atol=1e-6)
    ro, rs = oracle.din_attention_pool(table.cpu().numpy(), hist[sel[:64]].cpu().numpy(), hl[sel[:64]].cpu().numpy(), cand[sel[:64]].cpu().numpy(),
                                       W1.cpu().numpy(), b1.cpu().numpy(), W2.cpu().numpy(), b2.cpu().numpy(), W3.cpu().numpy(),
                                       b3.cpu().numpy(), normalize=True, acc64=True)
    assert (np.abs(out[sel[:64]].cpu().double().numpy() - ro) / (1 + np.abs(ro))).max() <= 1e-5


def test_table_beyond_4gib_offsets(ops):
    """BASELINE configs[4]'s logical table: 10^8 rows x 16 floats = 6.4 GB shared by the slots; ids at the top of the table
    need > 2^32-byte offsets in the gather, the fused FM and the multi-hot bag kernels."""
    V, K, F, B = 100_000_000, 16, 6, 2048
    g = torch.Generator(device="cuda").manual_seed(0)
    table = torch.empty((V, K), device="cuda")
    for s in range(0, V, 25_000_000):
        table[s:s + 25_000_000].normal_(0, 0.25, generator=g)
    ts = ops.TableSet([table] * F)
    ids = torch.randint(V - 5_000_000, V, (B, F), generator=g, device="cuda")
    ids[0, 0] = V - 1
    ids[1, 1] = 0
    emb, fm = ops.gather_fm(ts, ids)
    ref = table[ids.reshape(-1)].reshape(B, F * K)
    assert torch.equal(emb, ref)
    assert torch.equal(fm, ops.fm_logit(ref, F, K))
    offs = torch.arange(0, B * F * 2 + 1, 2, device="cuda", dtype=torch.int64)
    vals = torch.randint(V - 1000, V, (B * F * 2,), generator=g, device="cuda")
    bag = ops.embedding_bag(ts, vals, offs, None, combiner="sum")
    assert torch.equal(bag, (table[vals[0::2]] + table[vals[1::2]]).reshape(B, F * K))


# ---- A5 / A9 hidden layers: dir_dense_f32 (fp32 MFMA GEMM + bias + ReLU) ------------------------------------------------------
@pytest.mark.parametrize("M,Kd,N", [(300, 416, 400), (129, 400, 400), (1000, 64, 16), (77, 1024, 1024), (5, 4, 200), (128, 36, 80),
                                    (256, 416, 1024), (1, 16, 17), (1025, 428, 160)])
@pytest.mark.parametrize("relu", [False, True])
def test_dense_matches_float64(built_lib, M, Kd, N, relu):
    import torch
    from dir_amd import ops
    g = torch.Generator().manual_seed(M + Kd + N)
    xfull = torch.randn(M, Kd + 4, generator=g)
    x = xfull.cuda()[:, :Kd]                                   # row stride Kd + 4
    w = torch.randn(N, Kd, generator=g) / Kd ** 0.5
    b = torch.randn(N, generator=g) * 0.1 if (M + N) % 2 else None
    assert ops.dense_supported(x, w.cuda())
    ref = xfull[:, :Kd].double() @ w.double().t()
    if b is not None:
        ref = ref + b.double()
    if relu:
        ref = ref.clamp(min=0)
    got = ops.dense(x, w.cuda(), None if b is None else b.cuda(), relu=relu, arith="f32")
    err = (got.cpu().double() - ref).abs() / (1 + ref.abs())
    assert err.max() <= 1e-5, float(err.max())
    out = torch.full((M, N + 3), -7.0).cuda()                  # strided output, untouched padding
    ops.dense(x, w.cuda(), None if b is None else b.cuda(), relu=relu, out=out[:, :N], arith="f32")
    assert torch.equal(out[:, :N], got) and float(out[:, N:].max()) == -7.0 and float(out[:, N:].min()) == -7.0


def test_dense_limits(built_lib):
    import torch
    from dir_amd import ops
    from dir_amd._lib import DirError
    x = torch.randn(8, 10).cuda()
    assert not ops.dense_supported(x, torch.randn(32, 10).cuda())          # in_features not a multiple of 4
    with pytest.raises(DirError):
        ops.dense(x, torch.randn(32, 10).cuda())
    assert not ops.dense_supported(torch.randn(8, 16).cuda(), torch.randn(1, 16).cuda())   # units = 1: library matrix-vector product


# ---- dir_embedding_bag_ex_f32: per-column combiners, max_norm, vocabulary bound (round 2) --------------------------------------
@pytest.mark.parametrize("K,F,V,B", [(16, 5, 97, 257), (8, 3, 41, 130), (6, 4, 33, 65), (64, 2, 50, 40), (1, 3, 20, 77)])
def test_bag_ex_slot_combiners_max_norm_vocab_bit_exact(ops, oracle, K, F, V, B):
    rng = np.random.default_rng(K * 7 + F)
    tables = _tables(rng, F, V, K, scale=1.0)
    lens = rng.integers(0, 6, size=B * F)
    lens[::5] = 0
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ids = rng.integers(-1, V + 3, size=offs[-1]).astype(np.int64)          # -1 pruned, V..V+2 out of range
    w = rng.uniform(0.1, 2.0, size=offs[-1]).astype(np.float32)
    combs = [(f * 2 + 1) % 3 for f in range(F)]                            # mean, sum, sqrtn, ... per column
    names = [["sum", "mean", "sqrtn"][c] for c in combs]
    ts = ops.TableSet([_dev(t) for t in tables])
    for mn in (None, 0.75):
        for wts in (None, w):
            ref = oracle.embedding_bag(tables, ids, offsets=offs, weights=wts, combiner=combs, B=B, vocab=True, max_norm=mn or 0.0)
            got = ops.embedding_bag(ts, _dev(ids), _dev(offs), None if wts is None else _dev(wts), combiner=names, max_norm=mn)
            np.testing.assert_array_equal(got.cpu().numpy(), ref)
    # one-hot ids: out-of-range ids give zero rows in every gather flavour; max_norm clips one-hot rows too
    oh = rng.integers(-1, V + 2, size=(B, F)).astype(np.int64)
    ref = oracle.embedding_bag(tables, oh, vocab=True)
    assert (ref[(oh >= V).repeat(K, axis=1)] == 0).all()
    np.testing.assert_array_equal(ops.embedding_bag(ts, _dev(oh)).cpu().numpy(), ref)
    emb, fm = ops.gather_fm(ts, _dev(oh))
    np.testing.assert_array_equal(emb.cpu().numpy(), ref)
    np.testing.assert_array_equal(fm.cpu().numpy()[:, 0], oracle.fm_second_order(ref, F, K))
    refc = oracle.embedding_bag(tables, oh, vocab=True, max_norm=0.5)
    np.testing.assert_array_equal(ops.embedding_bag(ts, _dev(oh), max_norm=0.5).cpu().numpy(), refc)
    nrm = np.linalg.norm(refc.reshape(B, F, K), axis=2)
    assert nrm.max() <= 0.5 * (1 + 1e-6)
    if K == 1:                                                             # the linear term prunes out-of-range ids as well
        lts = ops.TableSet([_dev(t[:, 0].copy()) for t in tables])
        refl = oracle.embedding_bag(tables, oh, vocab=True).sum(axis=1)
        _close(ops.linear_logit(lts, _dev(oh)).cpu().numpy()[:, 0], refl)


def test_bag_per_slot_max_norm_bit_exact_and_deepfm_columns_may_differ(ops, oracle):
    """One max_norm PER COLUMN (every embedding_column carries its own; dir_embedding_bag_ex2_f32's slot_max_norm array): multi-hot and
    one-hot bags bit-exact against the oracle, which clips each slot with its own value; a DeepFM whose embedding columns differ in
    max_norm runs (forward and backward) and equals the per-column composition."""
    rng = np.random.default_rng(23)
    F, K, V, B = 5, 16, 60, 300
    tables = _tables(rng, F, V, K, scale=1.0)
    mns = [0.5, None, 2.0, 0.0, 1.25]
    lens = rng.integers(0, 5, size=B * F)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    ids = rng.integers(-1, V + 2, size=offs[-1]).astype(np.int64)
    w = rng.uniform(0.1, 2.0, size=offs[-1]).astype(np.float32)
    ts = ops.TableSet([_dev(t) for t in tables])
    for wts in (None, w):
        ref = oracle.embedding_bag(tables, ids, offsets=offs, weights=wts, combiner=1, B=B, vocab=True, max_norm=mns)
        got = ops.embedding_bag(ts, _dev(ids), _dev(offs), None if wts is None else _dev(wts), combiner="mean", max_norm=mns)
        np.testing.assert_array_equal(got.cpu().numpy(), ref)
    oh = rng.integers(-1, V + 1, size=(B, F)).astype(np.int64)
    ref = oracle.embedding_bag(tables, oh, vocab=True, max_norm=mns)
    np.testing.assert_array_equal(ops.embedding_bag(ts, _dev(oh), max_norm=mns).cpu().numpy(), ref)
    per = np.concatenate([oracle.embedding_bag([tables[f]], oh[:, f:f + 1], vocab=True, max_norm=mns[f] or 0.0) for f in range(F)], 1)
    np.testing.assert_array_equal(ref, per)
    nrm = np.linalg.norm(ref.reshape(B, F, K), axis=2)
    assert nrm[:, 0].max() <= 0.5 * (1 + 1e-6) and nrm[:, 1].max() > 2.0          # slot 0 clipped, slot 1 untouched
    # DeepFM: columns with different max_norm (round 2 raised NotImplementedError here)
    from dir_amd.deepfm import DeepFM
    from dir_amd import feature_column as fc
    torch.manual_seed(2)
    cats = [fc.categorical_column_with_identity("C%d" % i, V) for i in range(3)]
    cols = [fc.embedding_column(cats[0], K, max_norm=0.3), fc.embedding_column(cats[1], K), fc.embedding_column(cats[2], K, max_norm=1.0)]
    model = DeepFM(linear_feature_columns=cats, dnn_feature_columns=cols, dnn_hidden_units=[32, 16], fm_embedding_size=K).cuda()
    feats = {"C%d" % i: _dev(oh[:, i].clip(0, V - 1).copy()) for i in range(3)}
    logits = model(feats)
    logits.sum().backward()
    assert bool(torch.isfinite(logits).all()) and all(p.grad is not None for p in model.embedding_weights)
    with torch.no_grad():
        emb = ops.embedding_bag(ops.TableSet([p.data for p in model.embedding_weights]), torch.stack([feats["C%d" % i] for i in range(3)], 1),
                                max_norm=[0.3, None, 1.0])
    n3 = emb.view(B, 3, K).norm(dim=2)
    assert float(n3[:, 0].max()) <= 0.3 * (1 + 1e-6) and float(n3[:, 2].max()) <= 1.0 * (1 + 1e-6)


def test_np_ref_bag_agrees_with_c_oracle_on_max_norm(oracle):
    rng = np.random.default_rng(5)
    table = rng.standard_normal((30, 8)).astype(np.float32)
    for _ in range(20):
        n = int(rng.integers(0, 6))
        ids = rng.integers(-1, 32, size=n).astype(np.int64)
        w = rng.uniform(0.1, 2, size=n).astype(np.float32)
        for comb in (0, 1, 2):
            a = R.bag(table, ids, w, comb, max_norm=0.9)
            b = oracle.embedding_bag([table], ids, offsets=np.array([0, n]), weights=w, combiner=comb, B=1, vocab=True, max_norm=0.9)[0]
            np.testing.assert_array_equal(a, b)


def test_din_strided_candidate_and_lengths(ops, oracle):
    """cand = ids[:, j] / hist_len = lens[:, 0] are strided views; the C ABI takes dense [B] arrays."""
    rng = np.random.default_rng(11)
    V, K, T, H1, H2, B = 200, 64, 20, 80, 40, 37
    table = (rng.standard_normal((V, K)) * 0.125).astype(np.float32)
    both = rng.integers(0, V, size=(B, 3)).astype(np.int64)
    lens2 = rng.integers(0, T + 1, size=(B, 2)).astype(np.int32)
    hist = rng.integers(-1, V, size=(B, T)).astype(np.int64)
    W = [(rng.standard_normal(s) * 0.1).astype(np.float32) for s in ((4 * K, H1), (H1,), (H1, H2), (H2,), (H2,), (1,))]
    cand_v, len_v = _dev(both)[:, 1], _dev(lens2)[:, 0]
    assert not cand_v.is_contiguous() and not len_v.is_contiguous()
    out = ops.din_attention_pool(_dev(table), _dev(hist), len_v, cand_v, *[_dev(x) for x in W], normalize=True)
    ref, _ = oracle.din_attention_pool(table, hist, lens2[:, 0].copy(), both[:, 1].copy(), *W, normalize=True, acc64=True)
    _close(out.cpu().numpy(), ref)


def test_full_size_config4_din_on_the_10m_row_table(ops, oracle):
    """BASELINE.json configs[3] at its own size: 10 000 000-item vocabulary x dim 64 (2.56 GB, ten times the Infinity Cache),
    B = 65 536, T = 50, 256-80-40-1 unit, masked softmax.  Size-independent properties on the whole batch + 64 sampled samples
    against the double-accumulating oracle."""
    B, T, K, V, H1, H2 = 65536, 50, 64, 10_000_000, 80, 40
    g = torch.Generator(device="cuda").manual_seed(4)
    table = torch.empty((V, K), device="cuda")
    for s in range(0, V, 2_500_000):
        table[s:s + 2_500_000].normal_(0, 0.125, generator=g)
    hist = torch.randint(0, V, (B, T), generator=g, device="cuda")
    hist[:, 0] = torch.randint(V - 1000, V, (B,), generator=g, device="cuda")        # rows at the very top of the table
    hl = torch.randint(0, T + 1, (B,), generator=g, device="cuda", dtype=torch.int32)
    cand = torch.randint(0, V, (B,), generator=g, device="cuda")
    W1 = torch.randn((4 * K, H1), generator=g, device="cuda") * 0.05
    W2 = torch.randn((H1, H2), generator=g, device="cuda") * 0.1
    W3 = torch.randn((H2,), generator=g, device="cuda") * 0.1
    b1 = torch.randn(H1, generator=g, device="cuda") * 0.05
    b2 = torch.randn(H2, generator=g, device="cuda") * 0.05
    b3 = torch.zeros(1, device="cuda")
    for norm in (True, False):
        out, sc = ops.din_attention_pool(table, hist, hl, cand, W1, b1, W2, b2, W3, b3, normalize=norm, want_scores=True)
        nonempty = hl > 0
        mask = torch.arange(T, device="cuda").unsqueeze(0) < hl.unsqueeze(1)
        assert (sc[~mask] == 0).all() and (out[~nonempty] == 0).all()
        if norm:
            ssum = sc.sum(1)
            assert torch.allclose(ssum[nonempty], torch.ones_like(ssum[nonempty]), atol=1e-5)
        sel = torch.arange(0, B, 1021, device="cuda")[:64]
        comb = (sc[sel].unsqueeze(2) * table[hist[sel]]).sum(1)                       # out = sum_j w_j h_j with the kernel's own weights
        assert torch.allclose(out[sel], comb, rtol=1e-5, atol=1e-6)
        # the oracle on a compact copy of exactly the rows these 64 samples touch (the 2.56 GB table stays on the device)
        rows = torch.unique(torch.cat([hist[sel].reshape(-1), cand[sel]]))
        small = table[rows].cpu().numpy()
        remap = {int(r): i for i, r in enumerate(rows.cpu().tolist())}
        h_s = np.vectorize(remap.get)(hist[sel].cpu().numpy()).astype(np.int64)
        c_s = np.vectorize(remap.get)(cand[sel].cpu().numpy()).astype(np.int64)
        ro, rs = oracle.din_attention_pool(small, h_s, hl[sel].cpu().numpy(), c_s, W1.cpu().numpy(), b1.cpu().numpy(),
                                           W2.cpu().numpy(), b2.cpu().numpy(), W3.cpu().numpy(), b3.cpu().numpy(),
                                           normalize=norm, acc64=True)
        _close(out[sel].cpu().numpy(), ro)
        _close(sc[sel].cpu().numpy(), rs)
    # run-to-run bitwise reproducibility at full size: 100 reruns of the shipped (bf16x3) arithmetic with both sample schedules (the
    # round-2 build of this kernel came out different in 30-150 samples per launch: a gfx950 hazard between packed fp32 VALU
    # instructions and 16x16x32 MFMAs -- dir_amd/isa_check.py, tools/pk_mfma_probe.hip), and the fp32 kernel
    import os
    saved = {k: os.environ.get(k) for k in ("DIR_DIN_ARITH", "DIR_DIN_STATIC")}
    ref_static = {}
    try:
        for arith, static, reruns in (("f16x2", "1", 100), ("bf16x3", "1", 100), ("bf16x3", "0", 100), ("f32", "0", 10)):
            os.environ["DIR_DIN_ARITH"], os.environ["DIR_DIN_STATIC"] = arith, static
            for norm in (True, False):
                o0, s0 = ops.din_attention_pool(table, hist, hl, cand, W1, b1, W2, b2, W3, b3, normalize=norm, want_scores=True)
                o0, s0 = o0.clone(), s0.clone()
                for _ in range(reruns):
                    o1, s1 = ops.din_attention_pool(table, hist, hl, cand, W1, b1, W2, b2, W3, b3, normalize=norm, want_scores=True)
                    assert torch.equal(o1, o0) and torch.equal(s1, s0), (arith, static, norm)
                if arith == "bf16x3" and static == "0":            # the two schedules give the same bits (no cross-sample reduction)
                    assert torch.equal(o0, ref_static[norm][0]) and torch.equal(s0, ref_static[norm][1])
                if arith == "bf16x3" and static == "1":
                    ref_static[norm] = (o0, s0)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("M,Kd,N", [(300, 416, 400), (129, 400, 400), (1000, 64, 80), (77, 1024, 1024), (128, 40, 80), (256, 432, 1024), (1, 16, 32),
                                    (513, 36, 208), (257, 100, 260), (700, 8, 128), (31, 416, 200), (2100, 360, 200)])
@pytest.mark.parametrize("relu", [False, True])
def test_dense_bf16x3_matches_float64(built_lib, M, Kd, N, relu):
    """dir_dense_bf16x3_f32 (csrc/dense_bf3.hip): fp32 operands split into three bf16 pieces, six piece products on the bf16 matrix
    pipe, fp32 accumulation -- the same 1e-5 bar as the fp32-MFMA kernel against float64; all three column-block widths, k tails,
    row tails, strided x / out, the affine and gated epilogues, operands of very different magnitudes."""
    from dir_amd import ops
    g = torch.Generator().manual_seed(M + Kd + N)
    xfull = torch.randn(M, Kd + 4, generator=g).cuda()
    x = xfull[:, :Kd]                                              # row stride Kd + 4
    w = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).cuda()
    b = (torch.randn(N, generator=g) * 0.1).cuda()
    assert ops.dense_bf16x3_covers(x, w)
    ref = x.cpu().double() @ w.cpu().double().t() + b.cpu().double()
    if relu:
        ref = ref.clamp(min=0)
    got = ops.dense(x, w, b, relu=relu, arith="bf16x3")
    err = (got.cpu().double() - ref).abs() / (1 + ref.abs())
    assert err.max() <= 1e-5, float(err.max())
    f32 = ops.dense(x, w, b, relu=relu, arith="f32")                # the two arithmetics against each other
    assert float(((got - f32).abs() / (1 + f32.abs())).max()) <= 1e-5
    ps, psh = torch.rand(N).cuda() + 0.5, torch.randn(N).cuda()
    got2 = ops.dense(x, w, b, relu=relu, post_scale=ps, post_shift=psh, arith="bf16x3")
    assert torch.equal(got2, got * ps + psh)
    out = torch.full((M, N + 4), -7.0).cuda()                       # strided output, untouched padding
    ops.dense(x, w, b, relu=relu, out=out[:, :N], arith="bf16x3")
    assert torch.equal(out[:, :N], got) and float(out[:, N:].max()) == -7.0 and float(out[:, N:].min()) == -7.0
    gate = torch.randn(M, N, generator=g).cuda()
    gg = ops.dense_gated(x, w, gate, arith="bf16x3")
    lin = ops.dense(x, w, None, relu=False, arith="bf16x3")
    assert torch.equal(gg, torch.where(gate > 0, lin, torch.zeros_like(lin)))
    for scale in (1e-4, 300.0):                                     # fp32 exponent range: the bar is relative to the terms
        ys = ops.dense(x * scale, w, None, arith="bf16x3")
        refs = (x.cpu().double() * scale) @ w.cpu().double().t()
        assert float(((ys.cpu().double() - refs).abs() / (scale + refs.abs())).max()) <= 1e-5
    # a weight modified in place is re-packed (the image cache follows tensor._version)
    w.mul_(2.0)
    assert torch.equal(ops.dense(x, w, None, arith="bf16x3"), lin * 2.0)


def test_dense_bf16x3_limits_and_auto(built_lib):
    from dir_amd import ops
    x = torch.randn(64, 18).cuda()
    w = torch.randn(32, 18).cuda()
    assert not ops.dense_bf16x3_covers(x, w)                       # Kd not a multiple of 4
    with pytest.raises(ValueError):
        ops.dense(x, w, arith="bf16x3")
    assert ops.dense_auto_arith(65536, 416, 400) == "bf16x3" and ops.dense_auto_arith(65536, 1024, 1024) == "bf16x3"
    assert ops.dense_auto_arith(65536, 360, 200) == "bf16x3" and ops.dense_auto_arith(65536, 200, 80) == "f32"
    assert ops.dense_auto_arith(4096, 416, 400) == "f32" and ops.dense_auto_arith(65536, 400, 16) == "f32"


@pytest.mark.parametrize("B,d,L", [(300, 416, 3), (77, 432, 3), (1000, 51, 2), (5, 16, 1), (129, 429, 3)])
def test_cross_network_head_epilogue(ops, oracle, B, d, L):
    """dir_dcn_cross_head_f32: the cross stack's output is bit for bit dir_dcn_cross_f32's, and x_L . head_w (the cross branch's share of
    the final dense(1) over concat([cross, deep]), DeepCrossNetwork.py:136-137) is within 1e-6-class rounding of the float64 dot; with
    out = NULL the head value is the same bits."""
    g = torch.Generator().manual_seed(B + d)
    x0 = (torch.randn(B, d, generator=g) * 0.5).cuda()
    w = (torch.randn(L, d, generator=g) * 0.1).cuda()
    b = (torch.randn(L, d, generator=g) * 0.1).cuda()
    hw = (torch.randn(d, generator=g) * 0.2).cuda()
    ref_x = ops.cross_network(x0, w, b)
    h, x = ops.cross_network_head(x0, w, b, hw, want_x=True)
    assert torch.equal(x, ref_x)
    h_only = ops.cross_network_head(x0, w, b, hw)
    assert torch.equal(h_only, h)
    ref = (ref_x.double() @ hw.double()).reshape(B, 1)
    err = ((h.double() - ref).abs() / (1 + ref.abs())).max().item()
    assert err < 2e-6, err


@pytest.mark.parametrize("M,Kd,N,relu", [(16384, 1024, 1024, True), (12300, 432, 1024, True), (20000, 400, 416, False), (13000, 64, 80, True)])
def test_dense_head_epilogue(ops, M, Kd, N, relu):
    """dir_dense_bf16x3_head_f32: act(x W^T + b) . head_w with the activation never written, against the same layer followed by the dot
    product in float64 (the layer itself is the bf16x3 kernel's: 1e-5 class), rerun bitwise."""
    g = torch.Generator().manual_seed(M + N)
    x = (torch.randn(M, Kd, generator=g) * 0.5).cuda()
    w = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).cuda()
    b = (torch.randn(N, generator=g) * 0.1).cuda()
    hw = (torch.randn(N, generator=g) * 0.1).cuda()
    got = ops.dense_head(x, w, b, hw, relu=relu)
    assert got is not None and tuple(got.shape) == (M, 1)
    y = x.double() @ w.double().t() + b.double()
    if relu:
        y = torch.relu(y)
    ref = (y @ hw.double()).reshape(M, 1)
    err = ((got.double() - ref).abs() / (1 + ref.abs())).max().item()
    assert err < 1e-5, err
    assert torch.equal(ops.dense_head(x, w, b, hw, relu=relu), got)


@pytest.mark.parametrize("B,m,D,H", [(64, 26, 16, 128), (300, 26, 16, 128), (33, 40, 8, 70), (129, 8, 32, 200), (4097, 12, 4, 130), (50, 28, 16, 64),
                                     (17, 9, 16, 16)])
def test_cin_first_layer_over_field_pairs(ops, oracle, B, m, D, H):
    """dir_cin_layer1_bf16x3_f32 (the first layer of a stack, xk IS x0: the kernel multiplies the m (m + 1) / 2 unordered field pairs with
    the symmetrised weights) against the double-accumulating oracle at the general kernel's bar, against the general bf16x3 kernel on a
    copy of x0, pooled sums, the pooled-only form, reruns bitwise equal."""
    import torch
    rng = np.random.default_rng(B * 5 + m)
    x0n = (rng.standard_normal((B, m, D)) * 0.5).astype(np.float32)
    Wn = (rng.standard_normal((H, m * m)) / m).astype(np.float32)
    ref_x, ref_p = oracle.cin_layer(x0n, x0n, Wn)
    x0, W = torch.from_numpy(x0n).cuda(), torch.from_numpy(Wn).cuda()
    xo, po = ops.cin_layer(x0, x0, W, arith="bf16x3")                       # xk IS x0: the pair kernel
    scale = 1.0 + np.abs(ref_x)
    assert (np.abs(xo.cpu().double().numpy() - ref_x) / scale).max() <= 1e-5
    assert (np.abs(po.cpu().double().numpy() - ref_p) / (1.0 + np.abs(ref_p))).max() <= 1e-5 * np.sqrt(D)
    xg, pg = ops.cin_layer(x0, x0.clone(), W, arith="bf16x3")               # a copy: the general kernel
    assert float((xo - xg).abs().max()) <= 2e-6 * (1 + float(xg.abs().max()))
    x2, p2 = ops.cin_layer(x0, x0, W, arith="bf16x3")
    assert torch.equal(x2, xo) and torch.equal(p2, po)
    wide = torch.zeros((B, H + 8), device="cuda")
    none, p3 = ops.cin_layer(x0, x0, W, pooled=wide[:, 4:4 + H], want_xout=False, arith="bf16x3")
    assert none is None and float(wide[:, :4].abs().max()) == 0.0          # (pooled-only: the pooled form, another summation order)
    assert float((wide[:, 4:4 + H] - po).abs().max()) <= 1e-5 * (1 + float(po.abs().max())) * np.sqrt(D)


@pytest.mark.parametrize("B,m,D,Hp,H", [(300, 26, 16, 128, 128), (37, 10, 8, 8, 40), (5, 3, 4, 4, 3), (129, 26, 16, 300, 64), (64, 64, 32, 5, 36),
                                         (4099, 12, 16, 40, 128), (20000, 26, 16, 128, 128)])
def test_cin_pooled_form_of_the_last_layer(ops, oracle, B, m, D, Hp, H):
    """A layer whose map only feeds its pooled sums (want_xout=False): pooled = Z W^T with Z = sum_d xk x0 (dir_cin_pool_z_f32 + the dense
    kernels) against the double-accumulating oracle; Z itself against float64; the backward pieces (dir_cin_pool_dx_f32 with the pooled
    gradient of the layer below added, accumulation into dx0) against float64; reruns bitwise equal."""
    rng = np.random.default_rng(B + Hp)
    x0n = (rng.standard_normal((B, m, D)) * 0.5).astype(np.float32)
    xkn = (rng.standard_normal((B, Hp, D)) * 0.5).astype(np.float32)
    Wn = (rng.standard_normal((H, Hp * m)) / np.sqrt(Hp * m)).astype(np.float32)
    x0, xk, W = _dev(x0n), _dev(xkn), _dev(Wn)
    zl = []
    none, p = ops.cin_layer(x0, xk, W, want_xout=False, z_out=zl)
    assert none is None and len(zl) == 1
    _, ref_p = oracle.cin_layer(x0n, xkn, Wn, acc64=True)
    assert (np.abs(p.cpu().double().numpy() - ref_p) / (1.0 + np.abs(ref_p))).max() <= 1e-5
    z64 = torch.einsum("bid,bjd->bij", torch.from_numpy(xkn).double(), torch.from_numpy(x0n).double()).reshape(B, Hp * m)
    assert float((zl[0].cpu().double() - z64).abs().max()) <= 2e-6 * (1 + float(z64.abs().max()))
    assert torch.equal(ops.cin_pool_z(x0, xk), zl[0])
    # inference (nobody asks for Z): since round 6 the two passes are ONE kernel where it covers the shape (dir_cin_pooled_last_bf16x3_f32: Z formed
    # in registers, never written) -- against the oracle at the same bar, bitwise equal reruns; DIR_CIN_POOLED_FUSED=0 is the two-pass form, bit for bit
    pf = ops.cin_layer(x0, xk, W, want_xout=False)[1]
    assert (np.abs(pf.cpu().double().numpy() - ref_p) / (1.0 + np.abs(ref_p))).max() <= 1e-5
    assert torch.equal(ops.cin_layer(x0, xk, W, want_xout=False)[1], pf)
    fused, ops.CIN_POOLED_FUSED = ops.CIN_POOLED_FUSED, False
    try:
        assert torch.equal(ops.cin_layer(x0, xk, W, want_xout=False)[1], p)
    finally:
        ops.CIN_POOLED_FUSED = fused
    assert ops.cin_pooled_fused_covers(m, Hp, H, D) == (D == 16 and m <= 32 and H <= 128 and H % 4 == 0)
    if not ops.cin_pooled_fused_covers(m, Hp, H, D):
        assert torch.equal(pf, p)                                    # (an uncovered shape keeps the two-pass form)
    wide = torch.zeros((B, H + 5), device="cuda")                    # a pooled view that is not 16-byte aligned: through a copy
    ops.cin_layer(x0, xk, W, pooled=wide[:, 3:3 + H], want_xout=False)
    assert torch.equal(wide[:, 3:3 + H], p) and float(wide[:, :3].abs().max()) == 0.0
    dZn = (rng.standard_normal((B, Hp * m)) * 0.1).astype(np.float32)
    addn = (rng.standard_normal((B, Hp)) * 0.1).astype(np.float32)
    accn = (rng.standard_normal((B, m, D)) * 0.1).astype(np.float32)
    dZ, add, acc = _dev(dZn), _dev(addn), _dev(accn)
    dxk, dx0 = ops.cin_pool_dx(x0, xk, dZ, add_pooled=add, dx0=acc)
    assert dx0 is acc
    d3 = torch.from_numpy(dZn).double().view(B, Hp, m)
    rk = torch.einsum("bij,bjd->bid", d3, torch.from_numpy(x0n).double()) + torch.from_numpy(addn).double().unsqueeze(2)
    r0 = torch.einsum("bij,bid->bjd", d3, torch.from_numpy(xkn).double()) + torch.from_numpy(accn).double()
    assert float((dxk.cpu().double() - rk).abs().max()) <= 2e-6 * (1 + float(rk.abs().max()))
    assert float((dx0.cpu().double() - r0).abs().max()) <= 1e-5 * (1 + float(r0.abs().max()))
    k2, z2 = ops.cin_pool_dx(x0, xk, dZ, add_pooled=add)
    k3, z3 = ops.cin_pool_dx(x0, xk, dZ, add_pooled=add)
    assert torch.equal(k2, dxk) and torch.equal(k3, k2) and torch.equal(z3, z2)


def test_cin_pooled_fused_entry_checks(ops):
    """dir_cin_pooled_last_bf16x3_f32 / dir_cin_pooled_pack_f32 (round 6): shapes outside D = 16, m <= 32, H <= 128 are refused, a pooled view
    into a wider buffer is written in place, an odd channel count and a partial last row tile are handled."""
    import ctypes
    from dir_amd import _lib
    lib = _lib.load()
    assert lib.dir_cin_pooled_image_bytes(26, 128, 128, 16) == 128 * 8 * 3 * 1024 and lib.dir_cin_pooled_image_bytes(26, 128, 128, 8) == 0
    assert lib.dir_cin_pooled_image_bytes(40, 128, 128, 16) == 0 and lib.dir_cin_pooled_image_bytes(26, 128, 132, 16) == 0
    p = ctypes.c_void_p(256)
    assert lib.dir_cin_pooled_last_bf16x3_f32(p, p, p, 26, 128, 128, 8, 4, p, 128, None) == -4
    assert lib.dir_cin_pooled_last_bf16x3_f32(p, p, None, 26, 128, 128, 16, 4, p, 128, None) == -1
    assert lib.dir_cin_pooled_last_bf16x3_f32(p, p, p, 26, 128, 128, 16, 4, p, 126, None) == -1
    g = torch.Generator(device="cuda").manual_seed(4)
    B, m, Hp, H = 1037, 26, 7, 36                                    # an odd channel count, H not a multiple of 16, a partial row tile
    x0 = torch.randn((B, m, 16), generator=g, device="cuda") * 0.5
    xk = torch.randn((B, Hp, 16), generator=g, device="cuda") * 0.5
    W = torch.randn((H, Hp * m), generator=g, device="cuda") / (Hp * m) ** 0.5
    wide = torch.full((B, 3 * H + 4), 7.0, device="cuda")
    ops.cin_layer(x0, xk, W, pooled=wide[:, H:2 * H], want_xout=False)
    ref = torch.einsum("bid,bjd->bij", xk.double(), x0.double()).reshape(B, -1) @ W.double().t()
    assert float(((wide[:, H:2 * H].double() - ref).abs() / (1 + ref.abs())).max()) <= 1e-5
    assert bool((wide[:, :H] == 7.0).all()) and bool((wide[:, 2 * H:] == 7.0).all())
