"""Golden-fixture tests.  tests/golden/golden_v1.npz is written by tests/golden/make_golden.py (seeded
inputs + oracle outputs + the hand KATs; nothing in it comes from the reference, which cannot run here).
CPU: the oracle still reproduces the frozen outputs.  GPU: the HIP path reproduces them (bit-exact for
copy / ordered-sum / integer paths, 1e-5 scaled for the dot-product paths)."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v1.npz"))


def _close(got, ref, tol=1e-5):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    err = np.abs(got - ref) / (1 + np.abs(ref))
    assert err.max() <= tol, err.max()


def test_oracle_reproduces_golden(oracle):
    F, K = G["fm_ids"].shape[1], G["fm_tables"].shape[2]
    emb = oracle.embedding_bag(list(G["fm_tables"]), G["fm_ids"])
    np.testing.assert_array_equal(emb, G["fm_emb"])
    np.testing.assert_array_equal(oracle.fm_second_order(emb, F, K), G["fm_logit_f32"])
    _close(G["fm_logit_f32"], G["fm_logit_f64"])
    np.testing.assert_array_equal(oracle.linear_sparse_sum(list(G["lin_w"]), G["fm_ids"], bias=G["lin_bias"]), G["lin_logit"])
    Bb = (G["bag_offsets"].size - 1) // G["bag_tables"].shape[0]
    for c, name in [(0, "sum"), (1, "mean"), (2, "sqrtn")]:
        np.testing.assert_array_equal(oracle.embedding_bag(list(G["bag_tables"]), G["bag_ids"], offsets=G["bag_offsets"],
                                                           weights=G["bag_weights"], combiner=c, B=Bb), G["bag_out_%s_w" % name])
    for d in (416, 51):
        np.testing.assert_array_equal(oracle.dcn_cross(G["cross%d_x0" % d], G["cross%d_w" % d], G["cross%d_b" % d], acc64=True),
                                      G["cross%d_out" % d])
    assert oracle.fm_second_order(G["kat_fm_in"], 3, 2)[0] == G["kat_fm_out"][0]
    np.testing.assert_allclose(oracle.dcn_cross(G["kat_cross_x0"], G["kat_cross_w"], G["kat_cross_b"]), G["kat_cross_out"], rtol=1e-6)
    from oracle import np_ref as R
    np.testing.assert_array_equal(R.hash_bucket_int(G["hash_keys"], 1000), G["hash_out_1000"])
    np.testing.assert_array_equal(oracle.bucketize(G["bkt_x"], G["bkt_bd"]), G["bkt_out"])


def test_host_hash_reproduces_golden(built_lib):
    from dir_amd import ops
    for s, fp in zip(G["hash_strs"], G["hash_strs_fp64"]):
        assert ops.fingerprint64(str(s)) == int(fp)


@pytest.mark.gpu
def test_gpu_reproduces_golden(built_lib):
    import torch
    from dir_amd import ops

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).cuda()

    F, K = G["fm_ids"].shape[1], G["fm_tables"].shape[2]
    ts = ops.TableSet([dev(t) for t in G["fm_tables"]])
    emb, fm = ops.gather_fm(ts, dev(G["fm_ids"]))
    np.testing.assert_array_equal(emb.cpu().numpy(), G["fm_emb"])
    np.testing.assert_array_equal(fm.cpu().numpy()[:, 0], G["fm_logit_f32"])
    lts = ops.TableSet([dev(w) for w in G["lin_w"]])
    np.testing.assert_array_equal(ops.linear_logit(lts, dev(G["fm_ids"]), bias=dev(G["lin_bias"])).cpu().numpy()[:, 0], G["lin_logit"])
    bts = ops.TableSet([dev(t) for t in G["bag_tables"]])
    for name in ("sum", "mean", "sqrtn"):
        got = ops.embedding_bag(bts, dev(G["bag_ids"]), offsets=dev(G["bag_offsets"]), combiner=name)
        np.testing.assert_array_equal(got.cpu().numpy(), G["bag_out_%s" % name])
        got = ops.embedding_bag(bts, dev(G["bag_ids"]), offsets=dev(G["bag_offsets"]), weights=dev(G["bag_weights"]), combiner=name)
        np.testing.assert_array_equal(got.cpu().numpy(), G["bag_out_%s_w" % name])
    for d in (416, 51):
        _close(ops.cross_network(dev(G["cross%d_x0" % d]), dev(G["cross%d_w" % d]), dev(G["cross%d_b" % d])).cpu().numpy(),
               G["cross%d_out" % d])
    for norm in (0, 1):
        o, s = ops.din_attention_pool(dev(G["din_table"]), dev(G["din_hist"]), dev(G["din_len"]), dev(G["din_cand"]),
                                      dev(G["din_W1"]), dev(G["din_b1"]), dev(G["din_W2"]), dev(G["din_b2"]), dev(G["din_W3"]),
                                      dev(G["din_b3"]), normalize=bool(norm), want_scores=True)
        _close(o.cpu().numpy(), G["din_out_n%d" % norm])
        _close(s.cpu().numpy(), G["din_scores_n%d" % norm])
    x0 = dev(G["cin_x0"])
    x1, p1 = ops.cin_layer(x0, x0, dev(G["cin_W1"]))
    _close(x1.cpu().numpy(), G["cin_x1"]); _close(p1.cpu().numpy(), G["cin_p1"])
    x2, p2 = ops.cin_layer(x0, x1, dev(G["cin_W2"]))
    _close(x2.cpu().numpy(), G["cin_x2"]); _close(p2.cpu().numpy(), G["cin_p2"])
    np.testing.assert_array_equal(ops.hash_bucket_ints(dev(G["hash_keys"]), 1000).cpu().numpy(), G["hash_out_1000"])
    np.testing.assert_array_equal(ops.bucketize(dev(G["bkt_x"]), dev(G["bkt_bd"])).cpu().numpy(), G["bkt_out"])
    np.testing.assert_allclose(ops.cross_network(dev(G["kat_cross_x0"]), dev(G["kat_cross_w"]), dev(G["kat_cross_b"])).cpu().numpy(),
                               G["kat_cross_out"], rtol=1e-6)
    assert ops.fm_logit(dev(G["kat_fm_in"]), 3, 2).cpu().numpy()[0, 0] == G["kat_fm_out"][0]


# ---- v2: backward / optimiser fixture (tests/golden/golden_v2_backward.npz, make_golden.make_backward) ---------------------
G2 = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v2_backward.npz"))


def _opt_state(kind):
    from oracle import np_ref as R
    tabs = G2["opt_tables"]
    Fo, Vo, Ko = tabs.shape
    w = [t.astype(np.float64) for t in tabs]
    n = [np.full((Vo, Ko), 0.1) for _ in range(Fo)]
    if kind == "adagrad":
        R.sparse_adagrad_step(w, n, G2["opt_ids"], G2["opt_grad"], 0.05)
        return np.stack(w), np.stack(n), None
    z = [np.zeros((Vo, Ko)) for _ in range(Fo)]
    R.sparse_ftrl_step(w, n, z, G2["opt_ids"], G2["opt_grad"], 0.2, l1=0.01, l2=0.05)
    return np.stack(w), np.stack(n), np.stack(z)


def test_oracle_reproduces_backward_golden(oracle):
    from oracle import np_ref as R
    F, K = 7, 16
    np.testing.assert_allclose(R.fm_logit_backward(G2["fmb_emb"], G2["fmb_g"], F, K, G2["fmb_add"]), G2["fmb_out"], rtol=1e-12)
    for d in (416, 51):
        got = R.cross_network_backward(G2["crossb%d_x0" % d], G2["crossb%d_w" % d], G2["crossb%d_b" % d], G2["crossb%d_gout" % d])
        for a, k in zip(got, ("gx0", "gw", "gb")):
            np.testing.assert_allclose(a, G2["crossb%d_%s" % (d, k)], rtol=1e-12, atol=1e-14)
    dW, dxk, dx0 = oracle.cin_backward(G2["cinb_x0"], G2["cinb_xk"], G2["cinb_W"], G2["cinb_G"])
    np.testing.assert_array_equal(dW, G2["cinb_dW"])
    np.testing.assert_array_equal(dxk, G2["cinb_dxk"])
    np.testing.assert_array_equal(dx0, G2["cinb_dx0"])
    w, n, _ = _opt_state("adagrad")
    np.testing.assert_allclose(w, G2["opt_adagrad_w"], rtol=1e-12); np.testing.assert_allclose(n, G2["opt_adagrad_acc"], rtol=1e-12)
    w, n, z = _opt_state("ftrl")
    for a, k in ((w, "w"), (n, "n"), (z, "z")):
        np.testing.assert_allclose(a, G2["opt_ftrl_" + k], rtol=1e-12, atol=1e-14)


def test_backward_restatements_are_derivatives():
    """The float64 backward restatements in oracle/np_ref.py against central differences of the forward restatements (so the
    v2 fixture is anchored on the forward expressions, which cite the reference, and not on a second hand derivation)."""
    from oracle import np_ref as R
    rng = np.random.default_rng(3)
    B, F, K = 5, 4, 3
    emb = rng.standard_normal((B, F * K)); g = rng.standard_normal((B, 1))
    ana = R.fm_logit_backward(emb, g, F, K)
    eps = 1e-6
    num = np.zeros_like(emb)
    for i in range(emb.shape[1]):
        ep, em = emb.copy(), emb.copy()
        ep[:, i] += eps; em[:, i] -= eps
        num[:, i] = ((R.fm_logit(ep, F, K, np.float64) - R.fm_logit(em, F, K, np.float64)).reshape(B) * g[:, 0]) / (2 * eps)
    np.testing.assert_allclose(ana, num, rtol=1e-6, atol=1e-8)
    d, L = 6, 3
    x0 = rng.standard_normal((B, d)); w = rng.standard_normal((L, d)) * 0.3; b = rng.standard_normal((L, d)) * 0.3
    go = rng.standard_normal((B, d))
    gx0, gw, gb = R.cross_network_backward(x0, w, b, go)
    loss = lambda x0_, w_, b_: float((R.cross_network(x0_, w_, b_) * go).sum())
    for arr, ana_g, idx in ((x0, gx0, 0), (w, gw, 1), (b, gb, 2)):
        num = np.zeros_like(arr)
        for i in np.ndindex(arr.shape):
            args_p, args_m = [x0.copy(), w.copy(), b.copy()], [x0.copy(), w.copy(), b.copy()]
            args_p[idx][i] += eps; args_m[idx][i] -= eps
            num[i] = (loss(*args_p) - loss(*args_m)) / (2 * eps)
        np.testing.assert_allclose(ana_g, num, rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
def test_gpu_reproduces_backward_golden(built_lib):
    import torch
    from dir_amd import ops

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).cuda()

    got = ops.fm_logit_backward(dev(G2["fmb_emb"]), dev(G2["fmb_g"]), 7, 16, add_in=dev(G2["fmb_add"]))
    _close(got.cpu().numpy(), G2["fmb_out"])
    for d in (416, 51):
        gx0, gw, gb = ops.cross_network_backward(dev(G2["crossb%d_x0" % d]), dev(G2["crossb%d_w" % d]), dev(G2["crossb%d_b" % d]),
                                                 dev(G2["crossb%d_gout" % d]))
        for a, k in ((gx0, "gx0"), (gw, "gw"), (gb, "gb")):
            ref = G2["crossb%d_%s" % (d, k)]
            assert np.abs(a.cpu().numpy() - ref).max() <= 1e-5 * (1 + np.abs(ref).max()), k
    x0, xk, W, Gd = (dev(G2["cinb_" + k]) for k in ("x0", "xk", "W", "G"))
    for fwd_form in (False, True):
        dx0, dxk, dW = ops.cin_layer_backward(x0, xk, W, Gd, force_forward_form=fwd_form)
        for a, k in ((dW, "dW"), (dxk, "dxk"), (dx0, "dx0")):
            ref = G2["cinb_" + k]
            assert np.abs(a.cpu().numpy() - ref).max() <= 2e-5 * (1 + np.abs(ref).max()), (k, fwd_form)
    ids, grad = dev(G2["opt_ids"]), dev(G2["opt_grad"])
    ts = ops.TableSet([dev(t) for t in G2["opt_tables"]])
    opt = ops.SparseAdagrad(ts, 0.05, method="sorted")      # the fixture has pruned (-1) ids: the sorted form's path
    opt.step(ids, grad)
    _close(torch.stack(ts.tables).cpu().numpy(), G2["opt_adagrad_w"], 2e-6)
    _close(torch.stack(opt.accums).cpu().numpy(), G2["opt_adagrad_acc"], 2e-6)
    ts = ops.TableSet([dev(t) for t in G2["opt_tables"]])
    opt = ops.SparseFtrl(ts, lr=0.2, l1=0.01, l2=0.05)
    opt.step(ids, grad)
    _close(torch.stack(ts.tables).cpu().numpy(), G2["opt_ftrl_w"], 5e-6)
    _close(torch.stack(opt.accums).cpu().numpy(), G2["opt_ftrl_n"], 2e-6)
    _close(torch.stack(opt.linears).cpu().numpy(), G2["opt_ftrl_z"], 5e-6)
