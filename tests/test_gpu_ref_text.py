"""GPU side of tests/test_ref_text.py: the HIP path (through the C ABI) and the model modules against
tests/golden/ref_text_v1.npz -- vectors obtained by executing the REFERENCE'S OWN TEXT under a NumPy stand-in for
tensorflow ("stubbed tf": the reference's op order / axes / concat orders / scopes; the stub's primitive arithmetic).
Only the arrays travel to the GPU box; nothing here reads /root/reference.

Bars: the gathered inputs bit-exact (copies and in-order fp32 bag sums); logits / FM / cross within 1e-5 * (1 + |ref|)
of the reference text's float64 run."""
import numpy as np
import pytest
import torch

from tests.test_ref_text import G, G2, VARIANTS, _build_dcn, _build_deepfm, _build_esmm, _close, dcn_vars, deepfm_csr, deepfm_vars, esmm_vars

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _load(model, vars_):
    """checkpoint.load_npz's mapping, from the fixture's dict of TF-named arrays."""
    from dir_amd.checkpoint import tf_variable_map
    with torch.no_grad():
        for name, (p, lay) in tf_variable_map(model).items():
            a = vars_[name]
            a = a.T if lay == "T" else a.reshape(-1) if lay == "col" else a
            assert tuple(a.shape) == tuple(p.shape), name
            p.copy_(torch.from_numpy(np.ascontiguousarray(a)).to(p.device))
    return model


def _deepfm_features(variant, fc):
    ids, extra = G["deepfm_ids"], G["deepfm_ids_extra"]
    feats = {"C%d" % f: _dev(ids[:, f]) for f in range(6)}
    feats.update({"L%d" % j: _dev(extra[:, j]) for j in range(2)})
    if variant == "ragged":
        for f in (1, 4):
            feats["C%d" % f] = fc.Ragged(_dev(G["deepfm_rag%d_values" % f]), _dev(G["deepfm_rag%d_offsets" % f]),
                                         _dev(G["deepfm_rag%d_weights" % f]))
    return feats


def test_fm_kernels_match_reference_text(built_lib):
    from dir_amd import ops
    emb = _dev(G["fm26_emb"])
    _close(ops.fm_logit(emb, 26, 16).cpu().numpy(), G["fm26_logit_f64"])
    V = deepfm_vars()
    tabs = [_dev(V["dnn_fm_inputs/myself_input_layer/C%d_embedding/embedding_weights" % f]) for f in range(6)]
    ts = ops.TableSet(tabs)
    emb, fm = ops.gather_fm(ts, _dev(G["deepfm_ids"]))
    np.testing.assert_array_equal(emb.cpu().numpy(), G["deepfm_onehot_inputs_f32"])       # myself_input_layer, deepFM.py:363-400
    _close(fm.cpu().numpy(), G["deepfm_onehot_fm_f64"])                                   # fm_logit_fn, deepFM.py:321-335
    # ragged columns with per-column combiners through ONE launch (dir_embedding_bag_ex_f32)
    vals, offs, wts, combs = deepfm_csr("ragged")
    got = ops.embedding_bag(ts, _dev(vals), _dev(offs), _dev(wts), combiner=combs, field_major=True)
    np.testing.assert_array_equal(got.cpu().numpy(), G["deepfm_ragged_inputs_f32"])
    _close(ops.fm_logit(got, 6, 8).cpu().numpy(), G["deepfm_ragged_fm_f64"])


def test_linear_kernel_matches_reference_text(built_lib):
    from dir_amd import ops
    V = deepfm_vars()
    names = ["C%d" % f for f in range(6)] + ["L0", "L1"]
    ts = ops.TableSet([_dev(V["linear/linear_model/%s/weights" % n][:, 0].copy()) for n in names])
    ids = _dev(np.concatenate([G["deepfm_ids"], G["deepfm_ids_extra"]], axis=1))
    got = ops.linear_logit(ts, ids, bias=_dev(V["linear/linear_model/bias_weights"]))
    _close(got.cpu().numpy(), G["deepfm_onehot_linear_f64"])


@pytest.mark.parametrize("d", [51, 416, 429])
def test_cross_kernels_match_reference_text(built_lib, d):
    from dir_amd import ops
    x0, x, w, b = (_dev(G["cross%d_%s" % (d, n)]) for n in ("x0", "x", "w", "b"))
    _close(ops.cross_network(x0, w, b).cpu().numpy(), G["cross%d_arch_f64" % d])           # _cross_architecture, :350-367
    _close(ops.cross_op(x0, x, w[1], b[1]).cpu().numpy(), G["cross%d_op_f64" % d])         # _cross_op, :336-347


@pytest.mark.parametrize("variant", VARIANTS)
def test_deepfm_module_matches_reference_model_fn(built_lib, variant):
    """DeepFM.forward == _DeepFM_model_fn's logits (deepFM.py:143-252 executed under the stub) with the same weights loaded by
    their TensorFlow variable names."""
    from dir_amd import feature_column as fc
    from dir_amd.deepfm import DeepFM
    model = _load(_build_deepfm(variant, fc, DeepFM).cuda(), deepfm_vars()).eval()
    feats = _deepfm_features(variant, fc)
    with torch.no_grad():
        logits = model(feats)
        _close(logits.cpu().numpy(), G["deepfm_%s_logits_f64" % variant])
        _close(model.linear_logit_fn(feats, logits.device).cpu().numpy(), G["deepfm_%s_linear_f64" % variant])
    # the same forward with autograd recording (the training graph) gives the same logits
    model.train()
    model.hparams["dnn_dropout"] = None
    if variant != "onehot_bn":            # batch-norm in TRAIN mode uses batch statistics: a different function by design
        _close(model(feats).detach().cpu().numpy(), G["deepfm_%s_logits_f64" % variant])


def test_dcn_module_matches_reference_model_fn(built_lib):
    """DeepCrossNetwork.forward / .predict / .create_loss == the reference's _model_fn (DeepCrossNetwork.py:96-243 executed
    under the stub): name-sorted input_layer, cross || deep, concat, dense(1), predictions dict, MEAN-reduced loss."""
    from dir_amd import feature_column as fc
    from dir_amd.dcn import DeepCrossNetwork
    model = _load(_build_dcn(fc, DeepCrossNetwork).cuda(), dcn_vars()).eval()
    feats = {k: _dev(G["dcn_feat:" + k]) for k in ("age", "hours", "gain", "workclass", "marital", "occupation")}
    feats["native"] = fc.Ragged(_dev(G["dcn_feat:native_values"]), _dev(G["dcn_feat:native_offsets"]))
    with torch.no_grad():
        x0 = model.input_layer(feats)
        _close(x0.cpu().numpy(), G["dcn_x0_f64"], 1e-6)
        logits = model(feats)
        _close(logits.cpu().numpy(), G["dcn_logits_f64"])
        p = model.predict(feats)
        _close(p["logistic"].cpu().numpy(), G["dcn_logistic_f64"])
        _close(p["probabilities"].cpu().numpy(), G["dcn_probabilities_f64"])
        np.testing.assert_array_equal(p["class_ids"].cpu().numpy(), G["dcn_class_ids_f32"])
        wl, ul = model.create_loss(feats, logits, _dev(G["dcn_labels"]))
        _close(wl.cpu().numpy(), G["dcn_loss_f64"])
        _close(ul.cpu().numpy(), G["dcn_unweighted_loss_f64"])


def test_esmm_module_matches_reference_model_fn(built_lib):
    """ESMM.forward / .predict / .get_loss == the reference's _model_fn (ESMM.py:62-175 executed under the stub, PREDICT and EVAL):
    two towers with their own variables, name-sorted input_layer (bit-exact), ctcvr = sigmoid(ctr) * sigmoid(cvr) turned back into a
    logit through clip(p, 1e-7, 1 - 1e-7), the predictions dict, and the CTR (weight column) + CTCVR MEAN-reduced losses."""
    from dir_amd import feature_column as fc
    from dir_amd.esmm import ESMM
    model = _load(_build_esmm(fc, ESMM).cuda(), esmm_vars()).eval()
    feats = {k: _dev(G2["esmm_feat:" + k]) for k in ("price", "age", "user", "item", "w_click")}
    feats["tags"] = fc.Ragged(_dev(G2["esmm_feat:tags_values"]), _dev(G2["esmm_feat:tags_offsets"]), _dev(G2["esmm_feat:tags_weights"]))
    with torch.no_grad():
        np.testing.assert_array_equal(model.cvr_model.input_layer(feats).cpu().numpy(), G2["esmm_cvr_inputs_f32"])
        out = model(feats)
        _close(out["cvr_logits"].cpu().numpy(), G2["esmm_cvr_logits_f64"])
        _close(out["ctr_logits"].cpu().numpy(), G2["esmm_ctr_logits_f64"])
        _close(out["ctcvr_logits"].cpu().numpy(), G2["esmm_ctcvr_logits_f64"])
        p = model.predict(feats)
        _close(p["logistic"].cpu().numpy(), G2["esmm_logistic_f64"])
        _close(p["probabilities"].cpu().numpy(), G2["esmm_probabilities_f64"])
        np.testing.assert_array_equal(p["class_ids"].cpu().numpy(), G2["esmm_class_ids_f32"])
        labels = {"click_label": _dev(G2["esmm_click_label"]), "convert_label": _dev(G2["esmm_convert_label"])}
        wl, ul = model.get_loss(feats, labels, out)
        _close(wl.cpu().numpy(), G2["esmm_loss_f64"])
        _close(ul.cpu().numpy(), G2["esmm_unweighted_loss_f64"])

