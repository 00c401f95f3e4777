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
