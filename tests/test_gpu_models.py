"""GPU end-to-end tests of the modules that keep the reference constructor kwargs: logits against a NumPy
float64 restatement of the reference graph with the SAME parameters (oracle/np_ref.py), tolerance
1e-5 * (1 + |ref|) as BASELINE.json's north_star states.  Also the sharded lookup on one GPU under RCCL."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import np_ref as R  # noqa: E402


def _init_single_rank():
    """One-rank RCCL process group over a FileStore: no TCP port to clash on (a probed-free port was taken again before the rendezvous
    bound it on a shared GPU host: EADDRINUSE)."""
    import tempfile
    import torch.distributed as dist
    d = tempfile.mkdtemp(prefix="dir_pg_")
    dist.init_process_group("nccl", init_method="file://" + os.path.join(d, "store"), rank=0, world_size=1, device_id=torch.device("cuda", 0))


def _close(got, ref, tol=1e-5):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    err = np.abs(got - ref) / (1 + np.abs(ref))
    assert err.max() <= tol, "max scaled err %.3e" % err.max()


def _np(p):
    return p.detach().cpu().numpy().astype(np.float64)


@pytest.fixture(params=[1, 6144], ids=["hip_layers", "product_routing"])
def min_rows(request, monkeypatch):
    """The dense-layer routing threshold: 1 = every covered layer on the HIP kernels (what tests/conftest.py sets for the suite: the kernels
    are under test), 6144 = the PRODUCT default (details-in-recommendation_amd/dense.py: MIN_ROWS) -- at the reference's own batch sizes (100 / 256 / 1000)
    the towers' layers then run on nn.Linear (rocBLAS) between the HIP lookup / FM / cross kernels: what a user of the modules gets."""
    from dir_amd import dense as D
    monkeypatch.setattr(D, "MIN_ROWS", request.param)
    D.reset_routing()
    return request.param


def _check_routing(min_rows, batch=None):
    """product routing: a batch ops.dense_small_covers() accepts -- up to 256 rows always, the reference's own 100 / 256 -- runs
    dir_dense_small_f32 (round 5), and what lies between that and dense.MIN_ROWS runs dir_dense_mid_f32 (round 6): at no batch size does a
    covered layer go to nn.Linear any more (only layers under 16 units stay library code)."""
    from dir_amd import dense as D
    assert D.ROUTING["hip"], dict(D.ROUTING)
    if min_rows != 1:
        assert all(int(k.split("x")[1]) < 16 for k in D.ROUTING["library"]), dict(D.ROUTING)


def test_deepfm_config1_forward(built_lib, oracle, min_rows):
    """BASELINE configs[0] shape: 1k rows, 13 dense (bucketised, linear part only) + 26 sparse, dim 8."""
    from dir_amd.deepfm import DeepFM
    from dir_amd import feature_column as fc
    rng = np.random.default_rng(11)
    B, F, K, V = 1000, 26, 8, 10000
    cats = [fc.categorical_column_with_identity("C%d" % i, V) for i in range(F)]
    nums = [fc.bucketized_column(fc.numeric_column("I%d" % i), [0.1 * j for j in range(1, 10)]) for i in range(13)]
    model = DeepFM(linear_feature_columns=cats + nums, dnn_feature_columns=[fc.embedding_column(c, K) for c in cats],
                   dnn_hidden_units=[400, 400, 400], fm_embedding_size=K, batch_norm=True).cuda()
    with torch.no_grad():
        for w in model.linear_weights:
            w.normal_(0, 0.05)
        model.linear_bias.fill_(0.2)
        for bn in model.bns:
            bn.moving_mean.normal_(0, 0.1); bn.moving_variance.uniform_(0.5, 1.5); bn.gamma.uniform_(0.8, 1.2); bn.beta.normal_(0, 0.1)
    ids = rng.integers(0, V, size=(B, F)).astype(np.int64)
    dense = rng.uniform(0, 1, size=(B, 13)).astype(np.float32)
    feats = {"C%d" % i: torch.from_numpy(ids[:, i].copy()).cuda() for i in range(F)}
    feats.update({"I%d" % i: torch.from_numpy(dense[:, i].copy()).cuda() for i in range(13)})
    with torch.no_grad():
        pred = model.predict(feats)
    _check_routing(min_rows)
    # reference graph in float64 (deepFM.py:169-223)
    tabs = [p.detach().cpu().numpy() for p in model.embedding_weights]
    emb = oracle.embedding_bag(tabs, ids).astype(np.float64)
    fm = R.fm_logit(emb, F, K, np.float64)
    layers = [(_np(l.weight).T, _np(l.bias)) for l in model.hidden]
    bn = [(_np(b.moving_mean), _np(b.moving_variance), _np(b.gamma), _np(b.beta)) for b in model.bns]
    dnn = R.dnn_logit(emb, layers, (_np(model.logits_layer.weight).T, _np(model.logits_layer.bias)), bn=bn)
    lin_ids = np.concatenate([ids, np.stack([R.bucketize(dense[:, i], nums[i].boundaries) for i in range(13)], 1)], 1)
    lin = sum(_np(w)[lin_ids[:, f]] for f, w in enumerate(model.linear_weights)) + 0.2
    ref = fm + dnn + lin[:, None]
    _close(pred["logits"].cpu().numpy(), ref)
    rp = R.predictions(ref)
    _close(pred["logistic"].cpu().numpy(), rp["logistic"])
    _close(pred["probabilities"].cpu().numpy(), rp["probabilities"])
    np.testing.assert_array_equal(pred["class_ids"].cpu().numpy(), rp["class_ids"])
    # forward_ids fast path == dict path
    with torch.no_grad():
        l2 = model.forward_ids(torch.from_numpy(ids).cuda())
        lin_only = pred["logits"] - l2
    assert torch.isfinite(lin_only).all()


@pytest.mark.parametrize("batch_norm", [False, True])
def test_deepfm_forward_at_a_batch_that_takes_the_bf16x3_towers(built_lib, oracle, batch_norm):
    """DeepFM (cfg-2 style schema: 26 sparse x dim 16, 400-400-400) at 12 288 rows, where dense.dense_act runs the hidden layers on
    dir_dense_bf16x3_f32 (with the folded inference batch-norm in its epilogue): logits against the float64 reference graph
    (deepFM.py:169-223), and against the same model with the towers forced onto the fp32-MFMA kernel."""
    from dir_amd.deepfm import DeepFM
    from dir_amd import feature_column as fc
    from dir_amd import ops
    rng = np.random.default_rng(21)
    B, F, K, V = 12288, 26, 16, 5000
    assert ops.dense_auto_arith(B, F * K, 400) == "bf16x3"
    cats = [fc.categorical_column_with_identity("C%d" % i, V) for i in range(F)]
    model = DeepFM(linear_feature_columns=cats, dnn_feature_columns=[fc.embedding_column(c, K) for c in cats],
                   dnn_hidden_units=[400, 400, 400], fm_embedding_size=K, batch_norm=batch_norm).cuda()
    with torch.no_grad():
        for w in model.linear_weights:
            w.normal_(0, 0.05)
        model.linear_bias.fill_(-0.1)
        for bn in model.bns:
            bn.moving_mean.normal_(0, 0.1); bn.moving_variance.uniform_(0.5, 1.5); bn.gamma.uniform_(0.8, 1.2); bn.beta.normal_(0, 0.1)
    ids = rng.integers(0, V, size=(B, F)).astype(np.int64)
    feats = {"C%d" % i: torch.from_numpy(ids[:, i].copy()).cuda() for i in range(F)}
    with torch.no_grad():
        logits = model(feats)
        old = ops.DENSE_ARITH
        ops.DENSE_ARITH = "f32"
        try:
            logits32 = model(feats)
        finally:
            ops.DENSE_ARITH = old
    tabs = [p.detach().cpu().numpy() for p in model.embedding_weights]
    emb = oracle.embedding_bag(tabs, ids).astype(np.float64)
    fm = R.fm_logit(emb, F, K, np.float64)
    layers = [(_np(l.weight).T, _np(l.bias)) for l in model.hidden]
    bn = [(_np(b.moving_mean), _np(b.moving_variance), _np(b.gamma), _np(b.beta)) for b in model.bns] if batch_norm else None
    dnn = R.dnn_logit(emb, layers, (_np(model.logits_layer.weight).T, _np(model.logits_layer.bias)), bn=bn)
    lin = sum(_np(w)[ids[:, f]] for f, w in enumerate(model.linear_weights)) - 0.1
    ref = fm + dnn + lin[:, None]
    _close(logits.cpu().numpy(), ref)
    _close(logits32.cpu().numpy(), ref)
    assert not torch.equal(logits, logits32)          # two different kernels did run


def test_deepfm_multihot_weighted(built_lib, oracle):
    """The multi-hot path the reference advertises (deepFM.py:53,77; SequenceTensorFlowDataset/test4.py:50-59)."""
    from dir_amd.deepfm import DeepFM
    from dir_amd import feature_column as fc
    rng = np.random.default_rng(5)
    B, K, V = 257, 8, 50
    hist = fc.weighted_categorical_column(fc.categorical_column_with_identity("hist", V), "hist_w")
    item = fc.categorical_column_with_identity("item", V)
    cols = [fc.embedding_column(hist, K, combiner="mean"), fc.embedding_column(item, K)]
    model = DeepFM(linear_feature_columns=[item], dnn_feature_columns=cols, dnn_hidden_units=[16], fm_embedding_size=K).cuda()
    lens = rng.integers(0, 6, size=B)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    vals = rng.integers(-1, V, size=offs[-1]).astype(np.int64)
    w = rng.uniform(0.1, 2, size=offs[-1]).astype(np.float32)
    it = rng.integers(0, V, size=B).astype(np.int64)
    feats = {"hist": fc.Ragged(torch.from_numpy(vals).cuda(), torch.from_numpy(offs).cuda()),
             "hist_w": torch.from_numpy(w).cuda(), "item": torch.from_numpy(it).cuda()}
    with torch.no_grad():
        got = model(feats).cpu().numpy()
    t0, t1 = [p.detach().cpu().numpy() for p in model.embedding_weights]
    e0 = oracle.embedding_bag([t0], vals, offsets=offs, weights=w, combiner=oracle.MEAN, B=B)
    e1 = oracle.embedding_bag([t1], it.reshape(B, 1))
    emb = np.concatenate([e0, e1], 1).astype(np.float64)
    ref = R.fm_logit(emb, 2, K, np.float64) + R.dnn_logit(emb, [(_np(model.hidden[0].weight).T, _np(model.hidden[0].bias))],
                                                          (_np(model.logits_layer.weight).T, _np(model.logits_layer.bias)))
    ref = ref + (_np(model.linear_weights[0])[it] + _np(model.linear_bias))[:, None]
    _close(got, ref)


def test_dcn_adult_schema_forward(built_lib, oracle, min_rows):
    """build_model_columns of DeepCrossNetwork/train.py:54-101: 5 numeric, 4 indicator, 1 hashed embedding (dim 8)
    -> d = 51, concatenated in NAME-SORTED order (DeepCrossNetwork.py:126)."""
    from dir_amd.dcn import DeepCrossNetwork
    from dir_amd import feature_column as fc
    rng = np.random.default_rng(3)
    B = 256
    nums = ["age", "education_num", "capital_gain", "capital_loss", "hours_per_week"]
    vocabs = {"education": ["Bachelors", "HS-grad", "11th", "Masters", "9th", "Some-college", "Assoc-acdm", "Assoc-voc", "7th-8th",
                            "Doctorate", "Prof-school", "5th-6th", "10th", "1st-4th", "Preschool", "12th"],
              "marital_status": ["Married-civ-spouse", "Divorced", "Married-spouse-absent", "Never-married", "Separated",
                                 "Married-AF-spouse", "Widowed"],
              "relationship": ["Husband", "Not-in-family", "Wife", "Own-child", "Unmarried", "Other-relative"],
              "workclass": ["Self-emp-not-inc", "Private", "State-gov", "Federal-gov", "Local-gov", "?", "Self-emp-inc", "Without-pay",
                            "Never-worked"]}
    occupation = fc.categorical_column_with_hash_bucket("occupation", hash_bucket_size=1000)
    columns = [fc.numeric_column(n) for n in nums] + \
              [fc.indicator_column(fc.categorical_column_with_vocabulary_list(k, v)) for k, v in
               [("workclass", vocabs["workclass"]), ("education", vocabs["education"]), ("marital_status", vocabs["marital_status"]),
                ("relationship", vocabs["relationship"])]] + [fc.embedding_column(occupation, dimension=8)]
    model = DeepCrossNetwork(columns=columns, cross_layer_num=2, dnn_hidden_units=[32, 16, 8], batch_norm=True).cuda()
    assert model.column_num == 51
    with torch.no_grad():
        for bn in model.bns:
            bn.moving_mean.normal_(0, 0.1); bn.moving_variance.uniform_(0.5, 1.5); bn.beta.normal_(0, 0.1)
    occ_strings = ["Tech-support", "Craft-repair", "Other-service", "Sales", "Exec-managerial", "Prof-specialty", "?"]
    feats, raw = {}, {}
    for n in nums:
        raw[n] = rng.uniform(0, 2, size=B).astype(np.float32)
        feats[n] = torch.from_numpy(raw[n]).cuda()
    for k, v in vocabs.items():
        raw[k] = [v[i] if i < len(v) else "OOV" for i in rng.integers(0, len(v) + 1, size=B)]
        feats[k] = raw[k]
    raw["occupation"] = [occ_strings[i] for i in rng.integers(0, len(occ_strings), size=B)]
    feats["occupation"] = raw["occupation"]
    with torch.no_grad():
        got = model.predict(feats)
    _check_routing(min_rows, B)
    # NumPy restatement with name-sorted concat
    parts = {}
    for n in nums:
        parts[n] = raw[n].astype(np.float64)[:, None]
    for k, v in vocabs.items():
        ind = np.zeros((B, len(v)))
        for b, s in enumerate(raw[k]):
            if s in v:
                ind[b, v.index(s)] = 1.0
        parts[k + "_indicator"] = ind
    occ_ids = R.hash_bucket_fast(raw["occupation"], 1000)
    parts["occupation_embedding"] = _np(model.embedding_weights[0])[occ_ids]
    x0 = np.concatenate([parts[k] for k in sorted(parts)], axis=1)
    assert [c.name for c in model.columns] == sorted(parts)
    cross = R.cross_network(x0, _np(model.cross_w), _np(model.cross_b))
    layers = [(_np(l.weight).T, _np(l.bias)) for l in model.hidden]
    bn = [(_np(b.moving_mean), _np(b.moving_variance), _np(b.beta)) for b in model.bns]
    deep = R.deep_architecture(x0, layers, bn=bn)
    ref = np.concatenate([cross, deep], -1) @ _np(model.logits_layer.weight).T + _np(model.logits_layer.bias)
    _close(got["logits"].cpu().numpy(), ref)
    rp = R.predictions(ref)
    _close(got["probabilities"].cpu().numpy(), rp["probabilities"])
    np.testing.assert_array_equal(got["class_ids"].cpu().numpy(), rp["class_ids"])


def test_esmm_two_towers(built_lib, oracle):
    """models/ESMM: two towers with their own embedding variables over the same columns (ESMM.py:62-78,130-147)."""
    from dir_amd.esmm import ESMM
    from dir_amd import feature_column as fc
    rng = np.random.default_rng(21)
    B = 300
    cols = [fc.numeric_column("age"), fc.numeric_column("hours"),
            fc.indicator_column(fc.categorical_column_with_vocabulary_list("rel", ["a", "b", "c", "d"])),
            fc.embedding_column(fc.categorical_column_with_hash_bucket("occupation", 1000), dimension=8),
            fc.embedding_column(fc.categorical_column_with_identity("item", 500), dimension=8)]
    model = ESMM(columns=cols, dnn_hidden_units=[32, 16]).cuda()
    occ = [["Sales", "Tech", "?", "Exec"][i] for i in rng.integers(0, 4, size=B)]
    rel = [["a", "b", "c", "d", "zz"][i] for i in rng.integers(0, 5, size=B)]
    item = rng.integers(0, 500, size=B).astype(np.int64)
    age = rng.uniform(0, 1, B).astype(np.float32); hours = rng.uniform(0, 1, B).astype(np.float32)
    feats = {"age": torch.from_numpy(age).cuda(), "hours": torch.from_numpy(hours).cuda(), "rel": rel, "occupation": occ,
             "item": torch.from_numpy(item).cuda()}
    with torch.no_grad():
        out = model(feats)
        pred = model.predict(feats)
    occ_ids = R.hash_bucket_fast(occ, 1000)

    def tower(t):
        il = t.input_layer
        parts = {"age": age.astype(np.float64)[:, None], "hours": hours.astype(np.float64)[:, None]}
        ind = np.zeros((B, 4))
        for b, s_ in enumerate(rel):
            if s_ in "abcd":
                ind[b, "abcd".index(s_)] = 1.0
        parts["rel_indicator"] = ind
        names = [c.name for c in il.emb_cols]
        parts["occupation_embedding"] = _np(il.embedding_weights[names.index("occupation_embedding")])[occ_ids]
        parts["item_embedding"] = _np(il.embedding_weights[names.index("item_embedding")])[item]
        assert [c.name for c in il.columns] == sorted(parts)
        net = np.concatenate([parts[k] for k in sorted(parts)], 1)
        for l in t.hidden:
            net = R.relu(net @ _np(l.weight).T + _np(l.bias))
        return net @ _np(t.logits.weight).T + _np(t.logits.bias)

    ctr, cvr = tower(model.ctr_model), tower(model.cvr_model)
    _close(out["ctr_logits"].cpu().numpy(), ctr)
    p = np.clip(R.sigmoid(ctr) * R.sigmoid(cvr), 1e-7, 1 - 1e-7)
    _close(out["ctcvr_logits"].cpu().numpy(), np.log(p / (1 - p)), tol=2e-5)
    _close(pred["logistic"].cpu().numpy(), R.sigmoid(ctr) * R.sigmoid(cvr))
    assert model.ctr_model.input_layer.embedding_weights[0].data_ptr() != model.cvr_model.input_layer.embedding_weights[0].data_ptr()


def test_esmm_wide_deep(built_lib, oracle):
    """models/ESMM/ESMM_wide_deep.py: every sub-model = dnn tower + linear model over its own variables, logits summed (:194-270);
    predictions as ESMM (:127-150).  Forward against a float64 restatement, a training step with finite sparse / dense gradients,
    the reference's argument checks and default learning rates (:30,33,72-73)."""
    from dir_amd.esmm import ESMM_W_D
    from dir_amd import feature_column as fc
    rng = np.random.default_rng(33)
    B = 300
    item = fc.categorical_column_with_identity("item", 500)
    occupation = fc.categorical_column_with_hash_bucket("occupation", 1000)
    rel = fc.categorical_column_with_vocabulary_list("rel", ["a", "b", "c", "d"])
    dnn_cols = [fc.numeric_column("age"), fc.embedding_column(occupation, dimension=8), fc.embedding_column(item, dimension=8)]
    lin_cols = [item, occupation, rel]
    model = ESMM_W_D(linear_feature_columns=lin_cols, dnn_feature_columns=dnn_cols, dnn_hidden_units=[32, 16]).cuda()
    assert model.learning_rates() == (min(0.2, 1.0 / np.sqrt(3)), 0.05)
    with torch.no_grad():
        for sub in (model.ctr_model, model.cvr_model):
            for w in sub.linear.weights:
                w.normal_(0, 0.1)
            sub.linear.bias.fill_(0.3)
    occ = [["Sales", "Tech", "?", "Exec"][i] for i in rng.integers(0, 4, size=B)]
    rl = [["a", "b", "c", "d", "zz"][i] for i in rng.integers(0, 5, size=B)]
    it = rng.integers(0, 500, size=B).astype(np.int64)
    age = rng.uniform(0, 1, B).astype(np.float32)
    feats = {"age": torch.from_numpy(age).cuda(), "rel": rl, "occupation": occ, "item": torch.from_numpy(it).cuda()}
    with torch.no_grad():
        out = model(feats)
        pred = model.predict(feats)
    occ_ids = R.hash_bucket_fast(occ, 1000)
    rel_ids = np.array([("abcd".index(s_) if s_ in "abcd" else -1) for s_ in rl])

    def sub_logits(t):
        il = t.dnn.input_layer
        names = [c.name for c in il.emb_cols]
        parts = {"age": age.astype(np.float64)[:, None],
                 "occupation_embedding": _np(il.embedding_weights[names.index("occupation_embedding")])[occ_ids],
                 "item_embedding": _np(il.embedding_weights[names.index("item_embedding")])[it]}
        assert [c.name for c in il.columns] == sorted(parts)
        net = np.concatenate([parts[k] for k in sorted(parts)], 1)
        for l in t.dnn.hidden:
            net = R.relu(net @ _np(l.weight).T + _np(l.bias))
        dnn = net @ _np(t.dnn.logits.weight).T + _np(t.dnn.logits.bias)
        w_item, w_occ, w_rel = [_np(w) for w in t.linear.weights]
        lin = w_item[it] + w_occ[occ_ids] + np.where(rel_ids >= 0, w_rel[np.maximum(rel_ids, 0)], 0.0) + 0.3      # OOV: no weight
        return dnn + lin[:, None]

    ctr, cvr = sub_logits(model.ctr_model), sub_logits(model.cvr_model)
    _close(out["ctr_logits"].cpu().numpy(), ctr)
    _close(out["cvr_logits"].cpu().numpy(), cvr)
    p = np.clip(R.sigmoid(ctr) * R.sigmoid(cvr), 1e-7, 1 - 1e-7)
    _close(out["ctcvr_logits"].cpu().numpy(), np.log(p / (1 - p)), tol=2e-5)
    _close(pred["logistic"].cpu().numpy(), R.sigmoid(ctr) * R.sigmoid(cvr))
    # a training step: the linear weights receive (sparse) gradients, the towers dense ones; loss of ESMM_wide_deep.py:283-309
    labels = {"click_label": torch.from_numpy((rng.random(B) < 0.3).astype(np.float32)).cuda(),
              "convert_label": torch.from_numpy((rng.random(B) < 0.1).astype(np.float32)).cuda()}
    lg = model(feats)
    loss, _ = model.get_loss(feats, labels, lg)
    loss.backward()
    assert bool(torch.isfinite(loss))
    for sub in (model.ctr_model, model.cvr_model):
        assert all(w.grad is not None for w in sub.linear.weights) and sub.linear.bias.grad is not None
        assert all(l.weight.grad is not None and bool(torch.isfinite(l.weight.grad).all()) for l in sub.dnn.hidden)
    # the reference's argument checks
    with pytest.raises(ValueError):
        ESMM_W_D(dnn_hidden_units=[8])
    with pytest.raises(ValueError):
        ESMM_W_D(dnn_feature_columns=dnn_cols)
    only_wide = ESMM_W_D(linear_feature_columns=lin_cols).cuda()
    with torch.no_grad():
        z = only_wide(feats)
    assert float(z["ctr_logits"].abs().max()) == 0.0                       # zero-initialised linear model


def test_xdeepfm_forward(built_lib, oracle):
    from dir_amd.xdeepfm import XDeepFM
    from dir_amd import feature_column as fc
    rng = np.random.default_rng(9)
    B, m, D, V = 130, 26, 16, 500
    cats = [fc.categorical_column_with_identity("C%d" % i, V) for i in range(m)]
    model = XDeepFM(linear_feature_columns=cats, dnn_feature_columns=[fc.embedding_column(c, D) for c in cats],
                    cin_layer_sizes=(64, 32), dnn_hidden_units=(64, 32)).cuda()
    with torch.no_grad():
        for w in model.linear_weights:
            w.normal_(0, 0.05)
    ids = rng.integers(0, V, size=(B, m)).astype(np.int64)
    feats = {"C%d" % i: torch.from_numpy(ids[:, i].copy()).cuda() for i in range(m)}
    with torch.no_grad():
        got = model(feats).cpu().numpy()
    tabs = [p.detach().cpu().numpy() for p in model.embedding_weights]
    emb = oracle.embedding_bag(tabs, ids).astype(np.float64)
    x0 = emb.reshape(B, m, D)
    xk, pooled = x0, []
    for W in model.cin_W:
        xk, p = R.cin_layer(x0, xk, _np(W))
        pooled.append(p)
    logit = np.concatenate(pooled, 1) @ _np(model.cin_out.weight).T + _np(model.cin_out.bias)
    net = emb
    for l in model.hidden:
        net = R.relu(net @ _np(l.weight).T + _np(l.bias))
    logit = logit + net @ _np(model.dnn_out.weight).T + _np(model.dnn_out.bias)
    logit = logit + (sum(_np(w)[ids[:, f]] for f, w in enumerate(model.linear_weights)) + _np(model.linear_bias))[:, None]
    _close(got, logit)


def test_din_module(built_lib, oracle):
    from dir_amd.din import DINAttentionPool
    rng = np.random.default_rng(2)
    V, K, B, T = 1000, 64, 40, 50
    mod = DINAttentionPool(V, K, (80, 40), normalize=True).cuda()
    with torch.no_grad():
        mod.b1.normal_(0, 0.05); mod.b2.normal_(0, 0.05); mod.b3.fill_(0.01)
    hist = rng.integers(0, V, size=(B, T)).astype(np.int64)
    hl = rng.integers(1, T + 1, size=B).astype(np.int32)
    cand = rng.integers(0, V, size=B).astype(np.int64)
    args = (torch.from_numpy(hist).cuda(), torch.from_numpy(hl).cuda(), torch.from_numpy(cand).cuda())
    out = mod(*args)                                   # grad enabled: the autograd wrapper around the same kernel
    assert out.requires_grad
    with torch.no_grad():            # inference: the packed kernel since round 6 (the training forward keeps the wave-per-sample kernel): another summation order
        inf = mod(*args)
        assert float((inf - out.detach()).abs().max()) <= 2e-6 * (1 + float(out.detach().abs().max()))
        assert torch.equal(mod(*args), inf)
    _close(inf.cpu().numpy(), R.din_attention_pool(mod.table.detach().cpu().numpy(), hist, hl, cand, _np(mod.W1), _np(mod.b1), _np(mod.W2), _np(mod.b2),
                                                    _np(mod.W3), _np(mod.b3), normalize=True)[0])
    got = out.detach().cpu().numpy()
    ref, _ = R.din_attention_pool(mod.table.detach().cpu().numpy(), hist, hl, cand, _np(mod.W1), _np(mod.b1), _np(mod.W2), _np(mod.b2),
                                  _np(mod.W3), _np(mod.b3), normalize=True)
    _close(got, ref)


def test_gather_rows_and_sharded_lookup_single_gpu(built_lib, oracle):
    """The sharded path end to end on ONE GPU under the nccl (RCCL) backend with world_size 1: the HIP route
    and gather_rows kernels, the bucketing, both all_to_all calls' code path and the un-permute."""
    import torch.distributed as dist
    from dir_amd import ops
    from dir_amd.shard import ShardedTables
    rng = np.random.default_rng(8)
    F, K, B = 7, 16, 513
    vocab = [100, 17, 64, 1000, 5, 333, 64]
    full = [rng.standard_normal((v, K)).astype(np.float32) for v in vocab]
    ids = np.stack([rng.integers(-1, v, size=B) for v in vocab], 1).astype(np.int64)
    # flat row gather
    slot = rng.integers(0, F, size=300).astype(np.int32)
    row = np.array([rng.integers(-1, vocab[s]) for s in slot], np.int64)
    got = ops.gather_rows([torch.from_numpy(t).cuda() for t in full], torch.from_numpy(slot).cuda(), torch.from_numpy(row).cuda())
    ref = np.stack([full[s][r] if r >= 0 else np.zeros(K, np.float32) for s, r in zip(slot, row)])
    np.testing.assert_array_equal(got.cpu().numpy(), ref)
    created = False
    if not dist.is_initialized():
        _init_single_rank()
        created = True
    try:
        st = ShardedTables.from_full([torch.from_numpy(t).cuda() for t in full], force_collective=True)
        got, fm = st.lookup(torch.from_numpy(ids).cuda(), want_fm=True)
        ref = R.embedding_bag_onehot(full, ids)
        np.testing.assert_array_equal(got.cpu().numpy(), ref)
        np.testing.assert_array_equal(fm.cpu().numpy()[:, 0], oracle.fm_second_order(ref, F, K))
        np.testing.assert_array_equal(st.lookup(torch.from_numpy(ids).cuda()).cpu().numpy(), ref)
        # the device bucketing against a NumPy counting sort: same buckets, inv is the exact inverse
        flat = torch.from_numpy(ids.reshape(-1)).cuda()
        for P in (1, 2, 3, 8):
            vdev = torch.tensor(vocab, dtype=torch.int64, device="cuda")
            payload, inv, counts, starts = ops.shard_bucket(flat, vdev, P)
            payload, inv, counts, starts = [t.cpu().numpy() for t in (payload, inv, counts, starts)]
            a = ids.reshape(-1)
            own = np.empty(a.size, np.int64); loc = np.empty(a.size, np.int64)
            for f in range(F):
                sel = np.arange(f, a.size, F)
                o, l = R.shard_div_owner(np.maximum(a[sel], 0), vocab[f], P)
                own[sel] = np.where(a[sel] < 0, sel % P, o); loc[sel] = np.where(a[sel] < 0, -1, l)
            np.testing.assert_array_equal(counts, np.bincount(own, minlength=P))
            np.testing.assert_array_equal(starts, np.concatenate([[0], np.cumsum(counts)[:-1]]))
            assert sorted(inv.tolist()) == list(range(a.size))
            want = np.where(loc < 0, -1, loc * F + np.arange(a.size) % F)
            np.testing.assert_array_equal(payload[inv], want)
            assert ((inv >= starts[own]) & (inv < starts[own] + counts[own])).all()
    finally:
        if created:
            dist.destroy_process_group()


def test_graphed_forward_matches_eager(built_lib):
    """HIP-graph capture of a DeepFM forward (serving.GraphedForward): replays equal the eager result bit for bit,
    also for new inputs copied into the static buffers."""
    from dir_amd.deepfm import DeepFM
    from dir_amd import feature_column as fc
    from dir_amd.serving import GraphedForward
    torch.manual_seed(1)
    B, F, K, V = 256, 26, 16, 1000
    cats = [fc.categorical_column_with_identity("C%d" % i, V) for i in range(F)]
    model = DeepFM(linear_feature_columns=cats, dnn_feature_columns=[fc.embedding_column(c, K) for c in cats],
                   dnn_hidden_units=[64, 32], fm_embedding_size=K).cuda()
    with torch.no_grad():
        for w in model.linear_weights:
            w.normal_(0, 0.1)
    ids = torch.randint(0, V, (B, F), device="cuda")
    fwd = lambda x: model.forward_ids(x, x)
    g = GraphedForward(fwd, ids)
    for _ in range(3):
        ids = torch.randint(0, V, (B, F), device="cuda")
        with torch.no_grad():
            ref = fwd(ids)
        assert torch.equal(g(ids), ref)


def test_deepfm_packed_serving_equals_reference_layout(built_lib):
    from dir_amd.deepfm import DeepFM
    from dir_amd import feature_column as fc
    torch.manual_seed(2)
    B, F, K, V = 700, 26, 16, 500
    cats = [fc.categorical_column_with_identity("C%d" % i, V) for i in range(F)]
    nums = [fc.bucketized_column(fc.numeric_column("I%d" % i), [0.25, 0.5, 0.75]) for i in range(3)]
    model = DeepFM(linear_feature_columns=cats + nums, dnn_feature_columns=[fc.embedding_column(c, K) for c in cats],
                   dnn_hidden_units=[64, 32], fm_embedding_size=K).cuda()
    with torch.no_grad():
        for w in model.linear_weights:
            w.normal_(0, 0.1)
        model.linear_bias.fill_(-0.3)
    ids = torch.randint(-1, V, (B, F), device="cuda")
    lin_ids = torch.cat([ids, torch.randint(0, 4, (B, 3), device="cuda")], dim=1)
    with torch.no_grad():
        ref = model.forward_ids(ids, lin_ids)
        model.pack_for_serving()
        got = model.forward_ids(ids, lin_ids)
    # same kernels' arithmetic for emb / fm / first-order sums; only the final additions associate differently
    assert float((got - ref).abs().max()) <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("collective", [False, True])
def test_lookup_consume_feeds_the_tower_without_a_finish_pass(built_lib, collective):
    """ShardedTables.lookup_consume (round 5): the DeepFM tower kernel reads the rows where the exchange left them, through the inverse
    positions (ops.tower(gather=(rows_as_tables(rows, F), inv, ...)): lookups, FM term, hidden layers and head in one launch per micro-batch)
    -- bit for bit the logit of lookup(want_fm=True) + tower(adds=(fm,)), with no [B, F*K] concatenation written.  Pruned and
    out-of-range ids, a batch that is no multiple of the tile, one rank with and without the exchange code path."""
    import torch.distributed as dist
    from dir_amd import ops
    from dir_amd.shard import ShardedTables, rows_as_tables
    rng = np.random.default_rng(28)
    F, K, B = 26, 16, 1000
    vocab = [int(v) for v in rng.integers(5, 3000, size=F)]
    full = [torch.from_numpy((rng.standard_normal((v, K)) * 0.3).astype(np.float32)).cuda() for v in vocab]
    ids = np.stack([rng.integers(-1, v + 3, size=B) for v in vocab], 1).astype(np.int64)       # -1 pruned, >= vocab out of range
    ids_t = torch.from_numpy(ids).cuda()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    Ws = [torch.randn((400, F * K), generator=g, device=dev) * 0.05, torch.randn((400, 400), generator=g, device=dev) * 0.05]
    bs = [torch.randn((400,), generator=g, device=dev) * 0.1 for _ in Ws]
    hw, hb = torch.randn((400,), generator=g, device=dev) * 0.05, torch.randn((1,), generator=g, device=dev)
    created = False
    if collective and not dist.is_initialized():
        _init_single_rank()
        created = True
    try:
        st = ShardedTables.from_full(full, force_collective=collective)
        emb, fm = st.lookup(ids_t, want_fm=True)
        ref = ops.tower(emb, Ws, bs, head=(hw, hb), adds=(fm,), split="f16x2")
        out = torch.full((B, 1), float("nan"), device=dev)
        calls = []

        def consumer(s, e, rows, inv):
            calls.append((s, e))
            ops.tower(None, Ws, bs, head=(hw, hb), gather=(rows_as_tables(rows, F), inv, None, True), out=out[s:e], split="f16x2")
        st.lookup_consume(ids_t, consumer)
        torch.cuda.synchronize()
        assert calls and calls[0][0] == 0 and calls[-1][1] == B
        assert torch.equal(out, ref)
    finally:
        if created:
            dist.destroy_process_group()


def test_sharded_trainer_predict_runs_the_tower_over_the_received_rows(built_lib):
    """ShardedDeepFMTrainer.predict on a DeepFM the one-launch tower kernel covers (K = 16, 26 slots, 400-400 ReLU tower, units = 1): the
    lookup without its finish pass + the tower in gather form, bit for bit lookup(want_fm=True) + dnn_logit_fn; and ShardedTables.absmax
    is the tables' largest magnitude."""
    from dir_amd import feature_column as fc
    from dir_amd.deepfm import DeepFM
    from dir_amd.shard import ShardedTables, ShardedDeepFMTrainer
    torch.manual_seed(11)
    F, K, V, B = 26, 16, 300, 2048
    cats = [fc.categorical_column_with_identity("C%d" % i, V) for i in range(F)]
    model = DeepFM(linear_feature_columns=[], dnn_feature_columns=[fc.embedding_column(c, K) for c in cats], dnn_hidden_units=[400, 400],
                   fm_embedding_size=K).cuda()
    full = [p.detach().clone() for p in model.embedding_weights]
    st = ShardedTables.from_full(full)
    tr = ShardedDeepFMTrainer(model, st, lr_sparse=0.05, dense_optimizer=torch.optim.SGD([p for n, p in model.named_parameters()
                                                                                           if n.startswith(("hidden", "logits"))], lr=0.0))
    ids = torch.randint(-1, V + 2, (B, F), device="cuda")
    assert abs(st.absmax() - max(float(t.abs().max()) for t in full)) == 0.0
    seen = []
    orig = st.lookup_consume
    st.lookup_consume = lambda i, c: (seen.append(1), orig(i, c))[1]
    got = tr.predict(ids)
    assert seen, "predict() did not take the lookup_consume route"
    model.eval()
    with torch.no_grad():
        emb, fm = st.lookup(ids, want_fm=True)
        ref = model.dnn_logit_fn(emb, adds=(fm,))
    assert torch.equal(got, ref)


def test_sharded_training_step_single_gpu(built_lib):
    """ShardedTables.enable_training + lookup_train + backward on ONE GPU under nccl (RCCL) with world_size 1: the row-gradient
    exchange code path and the owner-side dir_sparse_adagrad_sorted_payload_f32 against a float64 dedup-sum Adagrad."""
    import torch.distributed as dist
    from dir_amd.shard import ShardedTables
    rng = np.random.default_rng(18)
    F, K, B = 5, 16, 700
    vocab = [50, 7, 300, 3, 64]
    full = [rng.standard_normal((v, K)).astype(np.float32) for v in vocab]
    ids = np.stack([rng.integers(-1, v, size=B) for v in vocab], 1).astype(np.int64)
    gout = (rng.standard_normal((B, F * K)) * 0.5).astype(np.float32)
    created = False
    if not dist.is_initialized():
        _init_single_rank()
        created = True
    try:
        tabs = [torch.from_numpy(t.copy()).cuda() for t in full]
        st = ShardedTables.from_full(tabs, force_collective=True).enable_training(lr=0.05, initial_accumulator_value=0.1)
        for step in range(2):
            emb = st.lookup_train(torch.from_numpy(ids).cuda())
            if step == 0:
                np.testing.assert_array_equal(emb.detach().cpu().numpy(), R.embedding_bag_onehot(full, ids))
            (emb * torch.from_numpy(gout).cuda()).sum().backward()
        ref_w = [t.astype(np.float64) for t in full]
        ref_a = [np.full((v, K), 0.1) for v in vocab]
        for step in range(2):
            for f, v in enumerate(vocab):
                gsum = np.zeros((v, K))
                ok = ids[:, f] >= 0
                np.add.at(gsum, ids[ok, f], gout[ok, f * K:(f + 1) * K].astype(np.float64))
                t = np.zeros(v, bool); t[ids[ok, f]] = True
                ref_a[f][t] += gsum[t] ** 2
                ref_w[f][t] -= 0.05 * gsum[t] / np.sqrt(ref_a[f][t])
        for f in range(F):
            _close(st.local_tables[f].cpu().numpy(), ref_w[f].astype(np.float32))
    finally:
        if created:
            dist.destroy_process_group()


# ---- round 2: API gaps named by the round-1 review -------------------------------------------------------------------------------
def _f64_deepfm_reference(model, emb_cols_feats, lin_feats, B, units):
    """float64 torch restatement of _DeepFM_model_fn (deepFM.py:169-223) on leaf copies of the model's parameters; returns
    (logits, {name: leaf})."""
    P = {n: p.detach().double().cpu().requires_grad_(True) for n, p in model.named_parameters()}
    embs = []
    for f, (kind, payload) in enumerate(emb_cols_feats):
        tab = P["embedding_weights.%d" % f]
        if kind == "onehot":
            ok = (payload >= 0).double().unsqueeze(1)
            embs.append(tab[payload.clamp(min=0)] * ok)
        else:
            vals, offs, w, comb = payload
            rows = []
            for b in range(B):
                sl = slice(int(offs[b]), int(offs[b + 1]))
                v, ww = vals[sl], w[sl]
                ok = v >= 0
                if ok.sum() == 0:
                    rows.append(torch.zeros(tab.shape[1], dtype=torch.float64))
                    continue
                r = (tab[v[ok]] * ww[ok].double().unsqueeze(1)).sum(0)
                den = ww[ok].double().sum() if comb == "mean" else (ww[ok].double() ** 2).sum().sqrt() if comb == "sqrtn" else 1.0
                rows.append(r / den)
            embs.append(torch.stack(rows))
    emb = torch.cat(embs, 1)
    F = len(embs)
    K = emb.shape[1] // F
    e3 = emb.view(B, F, K)
    fm = 0.5 * ((e3.sum(1) ** 2) - (e3 ** 2).sum(1)).sum(1, keepdim=True)
    net = emb
    i = 0
    while "hidden.%d.weight" % i in P:
        net = torch.relu(net @ P["hidden.%d.weight" % i].t() + P["hidden.%d.bias" % i])
        i += 1
    dnn = net @ P["logits_layer.weight"].t() + P["logits_layer.bias"]
    lin = P["linear_bias"].reshape(1, -1).expand(B, units)
    for f, (kind, payload) in enumerate(lin_feats):
        wt = P["linear_weights.%d" % f].reshape(-1, units)
        if kind == "onehot":
            lin = lin + wt[payload.clamp(min=0)] * (payload >= 0).double().unsqueeze(1)
        else:
            vals, offs, w, comb = payload
            rows = []
            for b in range(B):
                sl = slice(int(offs[b]), int(offs[b + 1]))
                v, ww = vals[sl], w[sl]
                ok = v >= 0
                rows.append((wt[v[ok]] * ww[ok].double().unsqueeze(1)).sum(0))       # sparse_combiner = 'sum'
            lin = lin + torch.stack(rows)
    return fm + dnn + lin, P


def test_deepfm_ragged_linear_column_trains(built_lib):
    """with a Ragged (multi-hot) linear column the first-order weights and the bias must receive
    gradients; checked against float64 autograd of the reference graph."""
    from dir_amd.deepfm import DeepFM
    from dir_amd import feature_column as fc
    g = torch.Generator().manual_seed(3)
    B, K, V = 65, 8, 31
    tags = fc.categorical_column_with_identity("tags", V)
    item = fc.categorical_column_with_identity("item", V)
    model = DeepFM(linear_feature_columns=[tags, item],
                   dnn_feature_columns=[fc.embedding_column(tags, K, combiner="sqrtn"), fc.embedding_column(item, K)],
                   dnn_hidden_units=[16, 16], fm_embedding_size=K).cuda()
    with torch.no_grad():
        for w in model.linear_weights:
            w.normal_(0, 0.1, generator=None)
        model.linear_bias.fill_(0.1)
    lens = torch.randint(0, 5, (B,), generator=g)
    offs = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(lens, 0)])
    vals = torch.randint(-1, V, (int(offs[-1]),), generator=g)
    it = torch.randint(0, V, (B,), generator=g)
    feats = {"tags": fc.Ragged(vals.cuda(), offs.cuda()), "item": it.cuda()}
    labels = (torch.rand(B, 1, generator=g) > 0.5).float()
    model.train()
    logits = model(feats)
    assert logits.grad_fn is not None
    loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, labels.cuda(), reduction="sum")
    loss.backward()
    ones = torch.ones(vals.numel())
    ref, P = _f64_deepfm_reference(model, [("ragged", (vals, offs, ones, "sqrtn")), ("onehot", it)],
                                   [("ragged", (vals, offs, ones, "sum")), ("onehot", it)], B, 1)
    _close(logits.detach().cpu().numpy(), ref.detach().numpy())
    torch.nn.functional.binary_cross_entropy_with_logits(ref, labels.double(), reduction="sum").backward()
    for n, p in model.named_parameters():
        assert p.grad is not None, "%s got no gradient" % n
        gp = p.grad.to_dense() if p.grad.is_sparse else p.grad
        err = (gp.cpu().double() - P[n].grad.reshape(gp.shape)).abs().max() / (1 + P[n].grad.abs().max())
        assert float(err) <= 2e-5, (n, float(err))
    assert float(model.linear_weights[0].grad.to_dense().abs().sum()) > 0 and float(model.linear_bias.grad.abs().sum()) > 0


def test_deepfm_forward_ids_is_differentiable(built_lib):
    """forward_ids() under autograd must give the tables gradients (it used the no-grad op)."""
    from dir_amd.deepfm import DeepFM
    from dir_amd import feature_column as fc
    B, F, K, V = 96, 5, 8, 40
    cats = [fc.categorical_column_with_identity("C%d" % i, V) for i in range(F)]
    model = DeepFM(linear_feature_columns=cats, dnn_feature_columns=[fc.embedding_column(c, K) for c in cats],
                   dnn_hidden_units=[16], fm_embedding_size=K).cuda()
    ids = torch.randint(0, V, (B, F), device="cuda")
    a = model.forward_ids(ids, ids)
    b = model({"C%d" % i: ids[:, i] for i in range(F)})
    assert torch.allclose(a, b, atol=1e-6)
    a.sum().backward()
    assert all(p.grad is not None for p in model.embedding_weights) and all(p.grad is not None for p in model.linear_weights)


def test_deepfm_multi_class_head(built_lib, oracle):
    """n_classes > 2 (deepFM.py:112-117): logits_dimension = n_classes; fm [B,1] broadcasts onto the dnn logits (deepFM.py:337-338),
    linear_model has units = n_classes; softmax head predictions and SUM-reduced sparse softmax cross entropy."""
    from dir_amd.deepfm import DeepFM
    from dir_amd import feature_column as fc
    g = torch.Generator().manual_seed(9)
    B, F, K, V, C = 130, 4, 8, 29, 5
    cats = [fc.categorical_column_with_identity("C%d" % i, V) for i in range(F)]
    model = DeepFM(linear_feature_columns=cats, dnn_feature_columns=[fc.embedding_column(c, K) for c in cats],
                   dnn_hidden_units=[32], fm_embedding_size=K, n_classes=C).cuda()
    with torch.no_grad():
        for w in model.linear_weights:
            w.normal_(0, 0.1)
        model.linear_bias.normal_(0, 0.1)
    assert tuple(model.linear_weights[0].shape) == (V, C) and model.logits_layer.out_features == C
    ids = torch.randint(-1, V, (B, F), generator=g)
    feats = {"C%d" % i: ids[:, i].cuda() for i in range(F)}
    labels = torch.randint(0, C, (B,), generator=g)
    ref, P = _f64_deepfm_reference(model, [("onehot", ids[:, i]) for i in range(F)], [("onehot", ids[:, i]) for i in range(F)], B, C)
    with torch.no_grad():
        p = model.predict(feats)
    _close(p["logits"].cpu().numpy(), ref.detach().numpy())
    _close(p["probabilities"].cpu().numpy(), torch.softmax(ref, -1).detach().numpy())
    np.testing.assert_array_equal(p["class_ids"].cpu().numpy(), ref.argmax(-1, keepdim=True).numpy())
    model.train()
    logits = model(feats)
    loss, unweighted = model.create_loss(feats, logits, labels.cuda())
    ref_loss = torch.nn.functional.cross_entropy(ref, labels, reduction="sum")
    _close(loss.detach().cpu().numpy(), ref_loss.detach().numpy())
    assert tuple(unweighted.shape) == (B, 1)
    loss.backward()
    ref_loss.backward()
    for n, p_ in model.named_parameters():
        gp = p_.grad.to_dense() if p_.grad.is_sparse else p_.grad
        err = (gp.cpu().double() - P[n].grad.reshape(gp.shape)).abs().max() / (1 + P[n].grad.abs().max())
        assert float(err) <= 2e-5, (n, float(err))
    from dir_amd.checkpoint import tf_variable_map
    assert tf_variable_map(model)["linear/linear_model/C0/weights"][1] is None        # [V, units] as TensorFlow stores it


def test_identity_column_range_check(built_lib):
    """categorical_column_with_identity without default_value raises on id >= num_buckets like TensorFlow's
    InvalidArgument; with a default_value the id is replaced; -1 stays the 'missing' marker."""
    from dir_amd.deepfm import DeepFM
    from dir_amd import feature_column as fc
    V, K, B = 10, 4, 6
    cols = [fc.categorical_column_with_identity("a", V), fc.categorical_column_with_identity("b", V, default_value=0)]
    model = DeepFM(linear_feature_columns=cols, dnn_feature_columns=[fc.embedding_column(c, K) for c in cols],
                   dnn_hidden_units=[8], fm_embedding_size=K).cuda()
    ok = {"a": torch.tensor([0, 9, -1, 3, 4, 5]).cuda(), "b": torch.tensor([0, 99, -5, 3, 4, 5]).cuda()}
    with torch.no_grad():
        assert torch.isfinite(model(ok)).all()                     # b's out-of-range ids take default_value
    bad = dict(ok, a=torch.tensor([0, 10, 2, 3, 4, 5]).cuda())
    from dir_amd import _input
    with pytest.raises(ValueError, match="num_buckets=10"):            # default: TensorFlow's timing, the offending forward raises
        with torch.no_grad():
            model(bad)
    with torch.no_grad():
        assert torch.isfinite(model(ok)).all()                         # a clean forward afterwards does not raise
    old = _input.CHECK_MODE
    _input.CHECK_MODE = "deferred"                                     # DIR_CHECK_IDS=deferred: never waits; the verdict is read on demand
    try:
        with pytest.raises(ValueError, match="num_buckets=10"):
            with torch.no_grad():
                out = model(bad)                                       # kernels treat the id as pruned: finite output, no fault
                assert torch.isfinite(out).all()
                torch.cuda.synchronize()
                _input.raise_pending(block=True)
    finally:
        _input.CHECK_MODE = old
    with pytest.raises(ValueError):
        fc.categorical_column_with_identity("c", 10, default_value=10)


def test_dcn_odd_width_padded_inference_path(built_lib, oracle):
    """d = 26 x 4 + 13 = 117 (not a multiple of 4): the inference forward writes x0 with row stride pad4(d) and zero pad columns,
    runs cross / first deep layer on the 16-byte paths and never materialises the concat; it must equal the plain formulation
    (the training-graph forward) and a float64 restatement of dcn_logits_fn (DeepCrossNetwork.py:118-141)."""
    from dir_amd.dcn import DeepCrossNetwork
    from dir_amd import feature_column as fc
    g = torch.Generator().manual_seed(2)
    B, F, K, V = 300, 26, 4, 50
    cols = [fc.embedding_column(fc.categorical_column_with_identity("C%02d" % i, V), K) for i in range(F)]
    cols += [fc.numeric_column("I%02d" % i) for i in range(13)]
    model = DeepCrossNetwork(columns=cols, cross_layer_num=3, dnn_hidden_units=[64, 32], batch_norm=True).cuda().eval()
    assert model.column_num == 117
    ids = torch.randint(-1, V, (B, F), generator=g)
    dense = torch.rand(B, 13, generator=g)
    feats = {"C%02d" % i: ids[:, i].cuda() for i in range(F)}
    feats.update({"I%02d" % i: dense[:, i].cuda() for i in range(13)})
    with torch.no_grad():
        x0p = model.input_layer(feats, pad_to=4)
        assert tuple(x0p.shape) == (B, 120) and float(x0p[:, 117:].abs().max()) == 0.0
        x0 = model.input_layer(feats)
        assert torch.equal(x0p[:, :117], x0)
        fast = model(feats)                              # padded path
        plain = model(x0)                                # tensor input: the plain formulation
        cr = __import__("dir_amd").ops.cross_network_padded(x0p, model.cross_w.data, model.cross_b.data)
        assert float(cr[:, 117:].abs().max()) == 0.0     # pad columns stay exactly zero through the layers
    _close(fast.cpu().numpy(), plain.cpu().numpy(), 2e-6)
    xr = x0.cpu().double().numpy()
    cross = R.cross_network(xr, _np(model.cross_w), _np(model.cross_b))
    layers = [(_np(l.weight).T, _np(l.bias)) for l in model.hidden]
    bn = [(_np(b.moving_mean), _np(b.moving_variance), _np(b.beta)) for b in model.bns]
    deep = R.deep_architecture(xr, layers, bn)
    ref = np.concatenate([cross, deep], -1) @ _np(model.logits_layer.weight).T + _np(model.logits_layer.bias)
    _close(fast.cpu().numpy(), ref)


@pytest.mark.parametrize("P,parts", [(1, None), (2, None), (3, None), (8, None), (8, [1, 8, 3, 2, 1, 5, 8]), (4, [4, 1, 2, 2, 3, 1, 4])])
def test_fixed_capacity_bucket_and_slab_gather(built_lib, P, parts):
    """dir_shard_bucket_cap / dir_gather_slabs_f32 (the sync-free requester / owner steps) against a NumPy restatement: per-owner
    counts, slab headers, inv = the exact inverse into the slab layout, overflow flag, self-cleaning workspace; with and without
    the reference partitioner's per-table slice counts."""
    from dir_amd import ops
    from dir_amd.shard import place_slices, local_slice
    rng = np.random.default_rng(P * 7 + (len(parts) if parts else 0))
    F, K, B = 7, 16, 1500
    vocab = [100, 17, 64, 1000, 5, 333, 64]
    pl = parts or [P] * F
    first = place_slices(pl, P) if parts else [0] * F
    ids = np.stack([rng.integers(-1, v + 2, size=B) for v in vocab], 1).astype(np.int64)      # pruned and out-of-range ids too
    a = ids.reshape(-1)
    n = a.size
    own = np.full(n, -1, np.int64); loc = np.full(n, -1, np.int64)
    for f in range(F):
        sel = np.arange(f, n, F)
        ok = (a[sel] >= 0) & (a[sel] < vocab[f])
        o, l = R.shard_div_owner(np.where(ok, a[sel], 0), vocab[f], pl[f])
        own[sel] = np.where(ok, (np.asarray(o) + first[f]) % P, -1); loc[sel] = np.where(ok, l, -1)
    true_counts = np.bincount(own[own >= 0], minlength=P)
    vdev = torch.tensor(vocab, dtype=torch.int64, device="cuda")
    pdev = torch.tensor(pl, dtype=torch.int32, device="cuda") if parts else None
    fdev = torch.tensor(first, dtype=torch.int32, device="cuda") if parts else None
    ws = torch.zeros(128, dtype=torch.int32, device="cuda")
    flat = torch.from_numpy(a).cuda()
    for cap in (int(true_counts.max()) + 5, max(1, int(true_counts.max()) // 2)):          # roomy, then overflowing
        payload = torch.full((P * (cap + 1),), -7, dtype=torch.int64, device="cuda")
        inv = torch.empty(n, dtype=torch.int64, device="cuda")
        counts = torch.empty(P, dtype=torch.int64, device="cuda")
        over = torch.full((1,), 9, dtype=torch.int32, device="cuda")
        stat = torch.full((3,), -1, dtype=torch.int64, device="cuda")
        ops.shard_bucket_cap(flat, vdev, P, cap, payload, inv, counts, over, ws, parts=pdev, first=fdev, stat=stat)
        assert stat.tolist()[:2] == [int((true_counts > cap).any()), int(true_counts.max())]
        pay = payload.cpu().numpy().reshape(P, cap + 1)
        iv = inv.cpu().numpy()
        np.testing.assert_array_equal(counts.cpu().numpy(), true_counts)
        np.testing.assert_array_equal(pay[:, 0] >> 32, np.full(P, true_counts.max()))        # header: this sender's largest demand |
        pay[:, 0] &= 0xffffffff                                                           # valid slots
        np.testing.assert_array_equal(pay[:, 0], np.minimum(true_counts, cap))
        assert int(over.item()) == int((true_counts > cap).any())
        assert int(ws.abs().sum()) == 0                                                   # left zero for the next call
        assert (iv[own < 0] == -1).all()
        kept = iv >= 0
        if (true_counts <= cap).all():
            assert kept[own >= 0].all()
        assert len(set(iv[kept].tolist())) == int(kept.sum())                             # distinct slots
        o_k, pos_k = iv[kept] // cap, iv[kept] % cap
        np.testing.assert_array_equal(o_k, own[kept])
        assert (pos_k < pay[o_k, 0]).all()
        np.testing.assert_array_equal(pay[o_k, 1 + pos_k], loc[kept] * F + (np.nonzero(kept)[0] % F))
        for o in range(P):                                                                # every valid slot is somebody's
            assert np.bincount(pos_k[o_k == o], minlength=1).max(initial=0) <= 1 and (o_k == o).sum() == pay[o, 0]
    # owner side: slabs addressed to "rank 0" of a P-way split hold local rows of rank 0's slices
    cap = int(true_counts.max()) + 3
    payload = torch.empty(P * (cap + 1), dtype=torch.int64, device="cuda")
    ops.shard_bucket_cap(flat, vdev, P, cap, payload, inv, counts, over, ws, parts=pdev, first=fdev)
    full = [rng.standard_normal((v, K)).astype(np.float32) for v in vocab]
    for r in range(min(P, 3)):
        local = []
        for f, v in enumerate(vocab):
            s, e = local_slice(v, pl[f], first[f], P, r)
            local.append(torch.from_numpy(full[f][s:e].copy()).cuda() if e > s else torch.zeros((0, K), device="cuda"))
        # pretend every peer sent rank r the slab this requester built for it
        recv = payload.view(P, cap + 1)[r].repeat(P).contiguous()
        out = torch.full((P * cap, K), 5.0, device="cuda")
        ops.gather_slabs(ops.TableSet(local), recv, P, cap, out, sanitize=True)
        o = out.cpu().numpy().reshape(P, cap, K)
        hdr = int(payload.view(P, cap + 1)[r, 0]) & 0xffffffff
        slots = payload.view(P, cap + 1)[r, 1:1 + hdr].cpu().numpy()
        want = np.stack([local[p % F].cpu().numpy()[p // F] for p in slots]) if hdr else np.zeros((0, K), np.float32)
        for sl in range(P):
            np.testing.assert_array_equal(o[sl, :hdr], want)
            assert (o[sl, hdr:] == 5.0).all()                                              # rows behind the header: untouched
        assert (recv.view(P, cap + 1)[:, 1 + hdr:] == -1).all()                            # sanitised


@pytest.mark.parametrize("P,B,layout", [(1, 700, "bf"), (2, 5000, "bf"), (8, 5000, "fb"), (3, 41000, "bf")])
def test_fixed_capacity_bucket_dedup(built_lib, P, B, layout):
    """dir_shard_bucket_cap_dedup against a NumPy restatement: inv (field-major) is an exact inverse into the slabs, a payload value
    appears once per (owner, slot, tile of 2048 / 4096 samples), the demand in counts / headers / stat is the de-duplicated one,
    an overflowing slab drops whole values (every duplicate gets -1); dir_shard_slab_stat reads the verdict back off the headers."""
    from dir_amd import ops
    rng = np.random.default_rng(P * 100 + B)
    F, vocab = 5, [50, 7, 3000, 64, 100000]
    ids = np.stack([np.minimum(rng.zipf(1.3, size=B) - 1, v + 1) - (rng.random(B) < 0.03) for v in vocab], 1).astype(np.int64)
    tile = 2048 if B * F <= (1 << 20) else 4096
    own = np.full((B, F), -1, np.int64); pay_ref = np.full((B, F), -1, np.int64)
    for f, v in enumerate(vocab):
        ok = (ids[:, f] >= 0) & (ids[:, f] < v)
        o, l = R.shard_div_owner(np.where(ok, ids[:, f], 0), v, P)
        own[:, f] = np.where(ok, np.asarray(o), -1); pay_ref[:, f] = np.where(ok, np.asarray(l) * F + f, -1)
    uniq = np.zeros(P, np.int64)                          # distinct payloads per owner, counted per (slot, tile)
    for f in range(F):
        for t0 in range(0, B, tile):
            sl = slice(t0, min(B, t0 + tile))
            for o in range(P):
                uniq[o] += np.unique(pay_ref[sl, f][own[sl, f] == o]).size
    idd = torch.from_numpy(ids).cuda()
    if layout == "fb":
        idd = idd.t().contiguous().t()                    # [B, F] view of field-major storage
    vdev = torch.tensor(vocab, dtype=torch.int64, device="cuda")
    ws = torch.zeros(128, dtype=torch.int32, device="cuda")
    for cap in (int(uniq.max()) + 3, max(1, int(uniq.max()) // 2)):
        payload = torch.full((P * (cap + 1),), -7, dtype=torch.int64, device="cuda")
        inv = torch.full((F * B,), -9, dtype=torch.int64, device="cuda")
        counts = torch.empty(P, dtype=torch.int64, device="cuda")
        over = torch.full((1,), 9, dtype=torch.int32, device="cuda")
        stat = torch.full((2,), -1, dtype=torch.int64, device="cuda")
        ops.shard_bucket_cap_dedup(idd, vdev, P, cap, payload, inv, counts, over, ws, stat=stat)
        np.testing.assert_array_equal(counts.cpu().numpy(), uniq)
        assert stat.tolist() == [int((uniq > cap).any()), int(uniq.max())] and int(over.item()) == int((uniq > cap).any())
        assert int(ws.abs().sum()) == 0
        pay = payload.cpu().numpy().reshape(P, cap + 1)
        np.testing.assert_array_equal(pay[:, 0] >> 32, np.full(P, uniq.max()))
        hdr = pay[:, 0] & 0xffffffff
        np.testing.assert_array_equal(hdr, np.minimum(uniq, cap))
        iv = inv.cpu().numpy().reshape(F, B).T                                   # [B, F]
        assert (iv[own < 0] == -1).all()
        kept = iv >= 0
        if (uniq <= cap).all():
            assert kept[own >= 0].all()
        o_k, pos_k = iv[kept] // cap, iv[kept] % cap
        np.testing.assert_array_equal(o_k, own[kept])
        assert (pos_k < hdr[o_k]).all()
        np.testing.assert_array_equal(pay[o_k, 1 + pos_k], pay_ref[kept])       # the slot holds this entry's payload
        assert np.unique(iv[kept]).size == sum(int(min(u, cap)) for u in uniq)    # every valid slot is somebody's
        # duplicates of a value inside a tile share ONE slot (or all miss together when it did not fit)
        for f in range(F):
            sl = slice(0, min(B, tile))
            key = np.where(own[sl, f] >= 0, ids[sl, f], -1)                        # the row id identifies a value (owner included)
            v_, first_idx, invu = np.unique(key, return_index=True, return_inverse=True)
            np.testing.assert_array_equal(iv[sl, f], iv[sl, f][first_idx][invu])
        st2 = torch.full((2,), -1, dtype=torch.int64, device="cuda")
        ops.shard_slab_stat(payload, P, cap, st2)
        assert st2.tolist() == [int((uniq > cap).any()), int(uniq.max())]


def test_sharded_lookup_dedup_zipf_full_size(built_lib):
    """ShardedTables(dedup=True) on Zipf(1.05) ids at the BASELINE shape (65 536 x 26, 1 M rows per table): bit-identical to the plain
    gather, and the exchange carries well under two thirds of the entries."""
    from dir_amd import ops
    from dir_amd.shard import ShardedTables
    gen = torch.Generator(device="cuda").manual_seed(5)
    B, F, V, K = 65536, 26, 1000000, 16
    tabs = [torch.randn((V, K), generator=gen, device="cuda") for _ in range(F)]
    u = torch.rand((B, F), generator=gen, device="cuda", dtype=torch.float64)
    a = 1.05
    ids = ((V ** (1 - a) - 1) * u + 1).pow(1 / (1 - a)).floor().long().clamp_(1, V) - 1
    ref = ops.embedding_bag(ops.TableSet(tabs), ids)
    st = ShardedTables(tabs, [V] * F, dedup=True)
    out, fm = st.lookup(ids, want_fm=True)
    assert torch.equal(out, ref) and torch.equal(fm, ops.fm_logit(ref, F, K))
    sent = int(st._plans[next(iter(st._plans))].counts.sum())
    assert 0 < sent < 0.62 * B * F, sent
    plain = ShardedTables(tabs, [V] * F)
    assert torch.equal(plain.lookup(ids), ref)


def test_sharded_lookup_fixed_capacity_paths_single_gpu(built_lib, oracle):
    """The pipelined fixed-capacity lookup with the HIP backend under nccl (RCCL) at world size 1 (collectives issued):
    chunked, preallocated outputs, overflow -> exact fallback -> grown capacity, dedup, lazy check."""
    import torch.distributed as dist
    from dir_amd.shard import ShardedTables
    rng = np.random.default_rng(18)
    F, K, B = 6, 16, 2000
    vocab = [100, 17, 64, 1000, 5, 333]
    full = [rng.standard_normal((v, K)).astype(np.float32) for v in vocab]
    ids = np.stack([rng.integers(-1, v, size=B) for v in vocab], 1).astype(np.int64)
    ref = R.embedding_bag_onehot(full, ids)
    ref_fm = oracle.fm_second_order(ref, F, K)
    created = False
    if not dist.is_initialized():
        _init_single_rank()
        created = True
    try:
        tabs = [torch.from_numpy(t).cuda() for t in full]
        idd = torch.from_numpy(ids).cuda()
        for kw in ({}, {"chunks": 1}, {"chunks": 7}, {"dedup": True}, {"dedup": True, "chunks": 2}, {"check": "lazy"}, {"mode": "exact"}):
            st = ShardedTables.from_full(tabs, force_collective=True, **kw)
            for _ in range(3):
                out = torch.full((B, F * K), 3.0, device="cuda")
                fm = torch.full((B, 1), 3.0, device="cuda")
                e, f = st.lookup(idd, want_fm=True, out=out, fm=fm)
                assert e.data_ptr() == out.data_ptr()
                np.testing.assert_array_equal(out.cpu().numpy(), ref)
                np.testing.assert_array_equal(fm.cpu().numpy()[:, 0], ref_fm)
            np.testing.assert_array_equal(st.lookup(idd).cpu().numpy(), ref)
            assert st.stats["fallbacks"] == 0
        st = ShardedTables.from_full(tabs, force_collective=True, slack=0.25, mode="fixed")     # slabs a quarter of the demand
        np.testing.assert_array_equal(st.lookup(idd).cpu().numpy(), ref)
        assert st.stats["fallbacks"] == 1
        np.testing.assert_array_equal(st.lookup(idd).cpu().numpy(), ref)                          # capacity grown: fixed path again
        assert st.stats["fallbacks"] == 1 and st.stats["cap"] >= B * F // 4
        lazy = ShardedTables.from_full(tabs, force_collective=True, slack=0.25, mode="fixed", check="lazy")
        lazy.lookup(idd)
        with pytest.raises(RuntimeError, match="overflowed"):
            lazy.lookup(idd)
    finally:
        if created:
            dist.destroy_process_group()


def test_esmm_head_entry_matches_the_reference_op_sequence(built_lib):
    """dir_esmm_head_f32 (ESMM.py:67-77 in one launch) against the same ops in float64, incl. saturated logits where the clip to [1e-7, 1 - 1e-7] decides."""
    from dir_amd import ops
    g = torch.Generator().manual_seed(12)
    ctr = torch.cat([torch.randn(5000, generator=g) * 3, torch.tensor([40.0, -40.0, 0.0, 25.0, 90.0, -90.0])]).reshape(-1, 1).cuda()
    cvr = torch.cat([torch.randn(5000, generator=g) * 3, torch.tensor([40.0, 40.0, 0.0, -25.0, 90.0, -90.0])]).reshape(-1, 1).cuda()
    got = ops.esmm_head(ctr, cvr, 1e-7)
    p = (torch.sigmoid(ctr.double()) * torch.sigmoid(cvr.double())).clamp(1e-7, 1 - 1e-7)
    ref = torch.log(p / (1 - p))
    p32 = (torch.sigmoid(ctr) * torch.sigmoid(cvr)).clamp(1e-7, 1 - 1e-7)
    lib_form = torch.log(p32 / (1 - p32))                                     # the library's fp32 op sequence (esmm.ESMM.forward under autograd)
    assert got.shape == ctr.shape and bool(torch.isfinite(got).all())
    # fp32 evaluates 1 - p at p near 1 with one bit to spare: hold the kernel to the fp32 library form's own distance from float64
    err = float(((got.double() - ref).abs() / (1 + ref.abs())).max())
    err_lib = float(((lib_form.double() - ref).abs() / (1 + ref.abs())).max())
    assert err <= max(2e-6, 2 * err_lib), (err, err_lib)
    assert built_lib.dir_esmm_head_f32(None, None, 8, 1e-7, None, None) != 0 and b"null pointer" in built_lib.dir_last_error()


def test_graphed_forward_with_frozen_weights_takes_the_cached_images(built_lib):
    """serving.GraphedForward(..., frozen_weights=True): the capture reads the weight images the warm-up calls built (no pack launch per replay)
    -- the same logits as the eager forward and as the default capture, bit for bit; and, as documented, a replay after an in-place weight
    update still runs the capture-time images while the default capture follows the update."""
    from dir_amd.deepfm import DeepFM
    from dir_amd import feature_column as fc
    from dir_amd.serving import GraphedForward
    torch.manual_seed(3)
    B, F, K, V = 2048, 26, 16, 3000
    cats = [fc.categorical_column_with_identity("C%d" % i, V) for i in range(F)]
    model = DeepFM(linear_feature_columns=cats, dnn_feature_columns=[fc.embedding_column(c, K) for c in cats],
                   dnn_hidden_units=[400, 400, 400], fm_embedding_size=K).cuda().eval()
    ids = torch.randint(0, V, (B, F), device="cuda")
    f = lambda x: model.forward_ids(x, x)      # noqa: E731
    with torch.no_grad():
        ref = f(ids).clone()
    g_default = GraphedForward(f, ids)
    g_frozen = GraphedForward(f, ids, frozen_weights=True)
    assert torch.equal(g_default(ids), ref) and torch.equal(g_frozen(ids), ref)
    with torch.no_grad():
        model.hidden[1].weight.mul_(0.5)
        new = f(ids).clone()
    assert not torch.equal(new, ref)
    assert torch.equal(g_default(ids), new)        # re-packs inside the graph: follows the update
    assert torch.equal(g_frozen(ids), ref)         # frozen: the images of capture time (build a new GraphedForward after loading weights)
    assert torch.equal(GraphedForward(f, ids, frozen_weights=True)(ids), new)
