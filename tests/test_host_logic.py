"""CPU tests of the host-side mirror of the reference interface (no GPU, no kernels): column shims, the
features-dict -> id-layout assembly, DCN's name-sorted input layout, constructor/argument errors."""
import os

import numpy as np
import pytest
import torch


def test_column_names_follow_tf(built_lib):
    from dir_amd import feature_column as fc
    occ = fc.categorical_column_with_hash_bucket("occupation", 1000)
    assert fc.embedding_column(occ, 8).name == "occupation_embedding"
    assert fc.indicator_column(fc.categorical_column_with_vocabulary_list("workclass", ["a", "b"])).name == "workclass_indicator"
    assert fc.bucketized_column(fc.numeric_column("age"), [18, 30]).name == "age_bucketized"
    w = fc.weighted_categorical_column(fc.categorical_column_with_identity("week_list", 7), "week_weight")
    assert w.name == "week_list_weighted_by_week_weight" and w.num_buckets == 7
    with pytest.raises(ValueError):
        fc.embedding_column(occ, 0)
    with pytest.raises(ValueError):
        fc.embedding_column(occ, 8, combiner="max")
    with pytest.raises(ValueError):
        fc.bucketized_column(fc.numeric_column("x"), [3, 1])


def test_collect_ids_onehot_is_a_strided_view(built_lib):
    from dir_amd import feature_column as fc
    from dir_amd._input import collect_ids
    cols = [fc.categorical_column_with_identity("C%d" % i, 100) for i in range(3)]
    feats = {"C%d" % i: torch.arange(5) * 3 + i for i in range(3)}
    kind, ids = collect_ids(cols, feats, "cpu")
    assert kind == "onehot" and tuple(ids.shape) == (5, 3)
    assert ids.stride() == (1, 5)                       # [F,B] storage viewed as [B,F]: no second copy
    assert ids.tolist() == [[3 * b + f for f in range(3)] for b in range(5)]
    # identity column with default_value: out-of-range ids replaced (LFM/Mixture_1/train.py:34-37)
    c = fc.categorical_column_with_identity("u", 10, default_value=0)
    assert c.ids({"u": torch.tensor([3, 12, -1])}, "cpu").tolist() == [3, 0, 0]
    with pytest.raises(ValueError, match="batch size"):
        collect_ids(cols[:2], {"C0": torch.arange(5), "C1": torch.arange(4)}, "cpu")


def test_collect_ids_takes_the_columns_of_one_matrix_without_a_copy(built_lib):
    """Feature columns that are views of ONE int64 matrix (a batched input pipeline's dict) become a strided [B, F] view of that matrix:
    same values as the stacked copy, no copy; anything else (separate tensors, other dtypes, columns out of order, a repeated column)
    still stacks."""
    from dir_amd import feature_column as fc
    from dir_amd._input import collect_ids
    cols = [fc.categorical_column_with_identity("C%d" % i, 1000) for i in range(4)]
    g = torch.Generator().manual_seed(0)
    wide = torch.randint(0, 1000, (7, 9), generator=g)                                   # [B, 9]: columns 2..5 of a wider batch
    kind, ids = collect_ids(cols, {"C%d" % i: wide[:, 2 + i] for i in range(4)}, "cpu")
    assert kind == "onehot" and ids.data_ptr() == wide[:, 2].data_ptr() and ids.stride() == (9, 1)
    assert torch.equal(ids, wide[:, 2:6])
    with torch.no_grad():                    # inference keeps the transposing stack for sample-major ids (its kernels want field-major)
        ids = collect_ids(cols, {"C%d" % i: wide[:, 2 + i] for i in range(4)}, "cpu")[1]
        assert ids.stride() == (1, 7) and torch.equal(ids, wide[:, 2:6])
    fm = torch.randint(0, 1000, (4, 7), generator=g)                                     # field-major [F, B] rows
    kind, ids = collect_ids(cols, {"C%d" % i: fm[i] for i in range(4)}, "cpu")
    assert ids.data_ptr() == fm.data_ptr() and ids.stride() == (1, 7) and torch.equal(ids, fm.t())
    kind, ids = collect_ids(cols, {"C%d" % i: wide[:, 2 * i] for i in range(4)}, "cpu")     # every second column: still one view
    assert ids.stride() == (9, 2) and torch.equal(ids, wide[:, 0:8:2])
    for feats in ({"C%d" % i: wide[:, 5 - i] for i in range(4)},                          # descending columns
                  {"C%d" % i: wide[:, 3] for i in range(4)},                              # the same column four times
                  {"C%d" % i: wide[:, i].clone() for i in range(4)},                      # separate tensors
                  {"C0": wide[:, 0], "C1": wide[:, 1], "C2": wide[:, 3], "C3": wide[:, 4]},   # uneven steps
                  {"C%d" % i: wide[:, i].to(torch.int32) for i in range(4)}):             # another dtype (converted per column)
        kind, ids = collect_ids(cols, feats, "cpu")
        want = torch.stack([feats["C%d" % i].to(torch.int64) for i in range(4)], dim=1)
        assert ids.stride() == (1, 7) and torch.equal(ids, want)
    one = collect_ids(cols[:1], {"C0": wide[:, 4]}, "cpu")[1]                              # F = 1
    assert tuple(one.shape) == (7, 1) and torch.equal(one[:, 0], wide[:, 4])


def test_collect_ids_ragged_field_major_csr(built_lib):
    from dir_amd import feature_column as fc
    from dir_amd._input import collect_ids
    hist = fc.weighted_categorical_column(fc.categorical_column_with_identity("hist", 50), "hist_w")
    item = fc.categorical_column_with_identity("item", 50)
    feats = {"hist": fc.Ragged(torch.tensor([4, 5, 6, 7]), torch.tensor([0, 1, 1, 4])),
             "hist_w": torch.tensor([0.5, 1.0, 2.0, 3.0]), "item": torch.tensor([9, 8, 7])}
    kind, vals, offs, wts, B = collect_ids([hist, item], feats, "cpu")
    assert kind == "ragged" and B == 3
    assert vals.tolist() == [4, 5, 6, 7, 9, 8, 7]
    assert offs.tolist() == [0, 1, 1, 4, 5, 6, 7]        # bag(b, f) = f*B + b, last entry = nnz
    assert wts.tolist() == [0.5, 1.0, 2.0, 3.0, 1.0, 1.0, 1.0]


def test_vocabulary_and_hash_columns_on_host(built_lib):
    from dir_amd import feature_column as fc
    from oracle import np_ref as R
    v = fc.categorical_column_with_vocabulary_list("rel", ["Husband", "Wife", "Own-child"])
    assert v.ids({"rel": ["Wife", "nope", "Husband"]}, "cpu").tolist() == [1, -1, 0]     # OOV -> -1 (pruned by the bag)
    h = fc.categorical_column_with_hash_bucket("occupation", 1000)
    s = ["Tech-support", "Sales", "?", ""]
    assert h.ids({"occupation": s}, "cpu").tolist() == R.hash_bucket_fast(s, 1000).tolist()
    with pytest.raises(ValueError):
        fc.categorical_column_with_hash_bucket("x", 0)


def test_dcn_input_layout_is_name_sorted(built_lib):
    from dir_amd import feature_column as fc
    from dir_amd.dcn import DeepCrossNetwork
    cols = [fc.numeric_column("age"), fc.numeric_column("capital_gain"),
            fc.indicator_column(fc.categorical_column_with_vocabulary_list("workclass", list("abcdefghi"))),
            fc.indicator_column(fc.categorical_column_with_vocabulary_list("education", list("abcdefghijklmnop"))),
            fc.embedding_column(fc.categorical_column_with_hash_bucket("occupation", 1000), dimension=8)]
    m = DeepCrossNetwork(columns=cols, cross_layer_num=3, dnn_hidden_units=[32, 16, 8])
    names = [c.name for c in m.columns]
    assert names == sorted(names) == ["age", "capital_gain", "education_indicator", "occupation_embedding", "workclass_indicator"]
    assert m.offsets == [0, 1, 2, 18, 26] and m.column_num == 35           # DeepCrossNetwork.py:127-128
    assert tuple(m.cross_w.shape) == (3, 35) and float(m.cross_w.abs().max()) <= 0.2   # trunc normal(0, 0.1)
    assert len(m.bns) == 2                                                    # BN on all but the last layer (:401)
    assert m.logits_layer.in_features == 35 + 8
    with pytest.raises(ValueError, match="empty columns"):
        DeepCrossNetwork(columns=[])
    with pytest.raises(ValueError, match="_DenseColumn"):
        DeepCrossNetwork(columns=[fc.categorical_column_with_identity("x", 3)])


def test_deepfm_constructor_contract(built_lib):
    from dir_amd import feature_column as fc
    from dir_amd.deepfm import DeepFM
    cats = [fc.categorical_column_with_identity("C%d" % i, 50) for i in range(4)]
    with pytest.raises(ValueError, match="empty columns"):                    # deepFM.py:104-105
        DeepFM()
    with pytest.raises(ValueError, match="_DenseColumn"):                    # deepFM.py:371-373
        DeepFM(dnn_feature_columns=cats, dnn_hidden_units=[8])
    with pytest.raises(ValueError, match="fm_embedding_size"):                # deepFM.py:329 reshape contract
        DeepFM(dnn_feature_columns=[fc.embedding_column(cats[0], 8), fc.embedding_column(cats[1], 4)],
               dnn_hidden_units=[8], fm_embedding_size=8)
    m = DeepFM(linear_feature_columns=cats, dnn_feature_columns=[fc.embedding_column(c, 8) for c in cats],
               dnn_hidden_units=[16, 8], fm_embedding_size=8, batch_norm=True, dnn_dropout=0.5, model_dir="/tmp/x")
    assert all(float(w.abs().sum()) == 0.0 for w in m.linear_weights)         # linear_model weights start at zero
    assert len(m.hidden) == 2 and len(m.bns) == 2 and m.logits_layer.in_features == 8
    assert float(m.embedding_weights[0].abs().max()) <= 2.0 / np.sqrt(8) + 1e-6
    names = m.tf_variable_names()
    assert names["embedding_weights.0"] == "dnn_fm_inputs/myself_input_layer/C0_embedding/embedding_weights"
    assert names["hidden.1.weight"] == "dnn_fm/hiddenlayer_1/kernel" and names["linear_bias"] == "linear/linear_model/bias_weights"
    with pytest.raises(ValueError, match="dictionary"):                       # deepFM.py:159-161
        m.forward([1, 2, 3])
    with pytest.raises(RuntimeError, match="no CPU fallback"):                # the product never computes on the CPU
        m.forward({"C%d" % i: torch.zeros(3, dtype=torch.int64) for i in range(4)})


def test_shard_div_range_and_table_check(built_lib):
    from dir_amd.shard import div_range
    assert [div_range(10, 4, r) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert [div_range(7, 8, r) for r in range(8)][-2:] == [(6, 7), (7, 7)]


def test_tf_checkpoint_mapping_roundtrip(built_lib, tmp_path):
    """parameters <-> reference checkpoint names/layouts; export -> load is the identity."""
    from dir_amd import feature_column as fc
    from dir_amd import checkpoint as ck
    from dir_amd.deepfm import DeepFM
    from dir_amd.dcn import DeepCrossNetwork
    from dir_amd.esmm import ESMM
    cats = [fc.categorical_column_with_identity("C%d" % i, 30) for i in range(3)]
    embs = [fc.embedding_column(c, 4) for c in cats]

    def fresh(kind):
        if kind == "deepfm":
            return DeepFM(linear_feature_columns=cats, dnn_feature_columns=embs, dnn_hidden_units=[8, 4], fm_embedding_size=4,
                          batch_norm=True)
        if kind == "dcn":
            return DeepCrossNetwork(columns=embs + [fc.numeric_column("x")], cross_layer_num=2, dnn_hidden_units=[8, 4, 2])
        return ESMM(columns=embs + [fc.numeric_column("x")], dnn_hidden_units=[8, 4])

    for kind in ("deepfm", "dcn", "esmm"):
        a, b = fresh(kind), fresh(kind)
        with torch.no_grad():
            for p in a.parameters():
                p.add_(torch.randn_like(p) * 0.1)
        names = ck.tf_variable_map(a)
        # every parameter and BN buffer is mapped exactly once
        mapped = {id(t) for t, _ in names.values()}
        want = {id(p) for p in a.parameters()} | {id(v) for k, v in a.named_buffers()}
        assert mapped == want, kind
        path = str(tmp_path / (kind + ".npz"))
        ck.export_npz(a, path)
        assert ck.load_npz(b, path) == []
        for (ka, ta), (kb, tb) in zip(sorted(ck.tf_variable_map(a).items()), sorted(ck.tf_variable_map(b).items())):
            assert ka == kb and torch.equal(ta[0], tb[0])
    m = ck.tf_variable_map(fresh("deepfm"))
    assert "dnn_fm_inputs/myself_input_layer/C0_embedding/embedding_weights" in m and "dnn_fm/hiddenlayer_0/kernel" in m
    assert "linear/linear_model/C1/weights" in m and "dnn_fm/hiddenlayer_1/batchnorm_1/moving_variance" in m
    d = np.load(str(tmp_path / "deepfm.npz"))
    assert d["dnn_fm/hiddenlayer_0/kernel"].shape == (12, 8) and d["linear/linear_model/C0/weights"].shape == (30, 1)
    m = ck.tf_variable_map(fresh("dcn"))
    assert "dcn_model/input_from_feature_columns/cross_w" in m and "dcn_model/logits/dense/kernel" in m
    assert "esmm/cvr_model/hiddenlayer_1/kernel" in ck.tf_variable_map(fresh("esmm"))


# ---- DCN training spec (DeepCrossNetwork.py:264-290, 422-458; [TF-upstream] tf.train decay formulas) -----------------
def test_learning_rate_decay_methods():
    import math
    import dir_amd  # noqa: F401
    from dir_amd import train_spec as ts
    spec = {"learning_rate": 0.001, "decay_method": "cosine_decay", "decay_steps": 3000, "alpha": 0.5}   # train.py:121-124
    assert ts.learning_rate_decay(spec, 0) == pytest.approx(0.001)
    assert ts.learning_rate_decay(spec, 1500) == pytest.approx(0.00075)
    assert ts.learning_rate_decay(spec, 3000) == pytest.approx(0.0005)
    assert ts.learning_rate_decay(spec, 99999) == pytest.approx(0.0005)
    assert spec == {"learning_rate": 0.001, "decay_method": "cosine_decay", "decay_steps": 3000, "alpha": 0.5}   # not mutated
    assert ts.learning_rate_decay({"learning_rate": 0.01}, 7) == 0.01                                       # _no_decay
    with pytest.raises(KeyError):
        ts.learning_rate_decay({}, 0)
    with pytest.raises(ValueError, match="Unsupported learning rate name"):
        ts.learning_rate_decay({"learning_rate": 0.1, "decay_method": "nope"}, 0)
    with pytest.raises(TypeError, match="argument are not correct"):
        ts.learning_rate_decay({"learning_rate": 0.1, "decay_method": "cosine_decay"}, 0)               # decay_steps missing
    e = {"learning_rate": 0.1, "decay_method": "exponential_decay", "decay_steps": 100, "decay_rate": 0.5}
    assert ts.learning_rate_decay(e, 100) == pytest.approx(0.05)
    assert ts.learning_rate_decay(e, 150) == pytest.approx(0.1 * 0.5 ** 1.5)
    assert ts.learning_rate_decay(dict(e, staircase=True), 150) == pytest.approx(0.05)
    pw = {"decay_method": "piecewise_constant", "boundaries": [10, 20], "values": [1.0, 0.5, 0.1]}
    assert [ts.learning_rate_decay(pw, s) for s in (0, 10, 11, 20, 21)] == [1.0, 1.0, 0.5, 0.5, 0.1]
    po = {"learning_rate": 0.1, "decay_method": "polynomial_decay", "decay_steps": 100, "end_learning_rate": 0.01, "power": 2.0}
    assert ts.learning_rate_decay(po, 50) == pytest.approx(0.09 * 0.25 + 0.01)
    assert ts.learning_rate_decay(po, 500) == pytest.approx(0.01)
    assert ts.learning_rate_decay(dict(po, cycle=True), 150) == pytest.approx(0.09 * 0.25 ** 2 + 0.01)     # period 200
    cr = {"learning_rate": 1.0, "decay_method": "cosine_decay_restarts", "first_decay_steps": 10}
    assert ts.learning_rate_decay(cr, 0) == pytest.approx(1.0)
    assert ts.learning_rate_decay(cr, 10) == pytest.approx(1.0)                                             # restart
    assert ts.learning_rate_decay(cr, 20) == pytest.approx(0.5 * (1 + math.cos(math.pi * 0.5)))            # 2nd period is 20 long
    assert ts.learning_rate_decay(dict(cr, t_mul=1.0, m_mul=0.5), 15) == pytest.approx(0.5 * 0.5 * (1 + math.cos(math.pi * 0.5)))
    nl = {"learning_rate": 1.0, "decay_method": "noisy_linear_cosine_decay", "decay_steps": 100, "initial_variance": 0.0}
    assert ts.learning_rate_decay(nl, 50) == pytest.approx((0.5) * 0.5 * (1 + math.cos(math.pi * 0.5)) + 0.001)


def test_clip_by_norm_per_tensor():
    import dir_amd  # noqa: F401
    from dir_amd import train_spec as ts
    g = torch.full((4, 100), 10.0)                     # norm 200 -> scaled to 100
    out = ts.clip_by_norm_(g.clone())
    assert torch.linalg.vector_norm(out).item() == pytest.approx(100.0, rel=1e-6)
    small = torch.ones(5)
    assert torch.equal(ts.clip_by_norm_(small.clone()), small)
    sp = torch.sparse_coo_tensor(torch.tensor([[1, 3, 1]]), torch.full((3, 2), 100.0), (5, 2))
    out = ts.clip_by_norm_(sp)
    assert out.is_sparse and torch.linalg.vector_norm(out.to_dense()).item() == pytest.approx(100.0, rel=1e-6)
    assert ts.clip_by_norm_(None) is None


def test_batch_norm_train_and_inference_forms():
    """_BatchNormInfer: inference uses the moving statistics; TRAIN (train() state + autograd on) normalises with the batch's
    mean / population variance and moves the statistics with momentum 0.999 ([TF-upstream] batch_normalization)."""
    import dir_amd  # noqa: F401
    from dir_amd.deepfm import _BatchNormInfer, _dropout_train
    torch.manual_seed(0)
    bn = _BatchNormInfer(5)
    x = torch.randn(64, 5) * 3 + 1.5
    with torch.no_grad():                                  # inference form even though the module is in train() state
        y = bn(x)
    assert torch.allclose(y, x / (1 + 1e-3) ** 0.5, atol=1e-6)
    assert torch.equal(bn.moving_mean, torch.zeros(5))
    y = bn(x)                                              # TRAIN
    mean, var = x.mean(0), x.var(0, unbiased=False)
    assert torch.allclose(y, (x - mean) / torch.sqrt(var + 1e-3), atol=1e-5)
    assert torch.allclose(bn.moving_mean, 0.001 * mean, atol=1e-7)
    assert torch.allclose(bn.moving_variance, 0.999 + 0.001 * var, atol=1e-6)
    bn.eval()
    before = bn.moving_mean.clone()
    y2 = bn(x)
    assert torch.equal(bn.moving_mean, before)             # eval(): no update, moving statistics used
    assert not torch.allclose(y2, y)
    # dropout: TRAIN only, scaled by 1 / (1 - rate)
    m = torch.nn.Module()
    ones = torch.ones(1000, 10)
    d = _dropout_train(m, ones, 0.5)
    assert set(d.unique().tolist()) == {0.0, 2.0}
    with torch.no_grad():
        assert torch.equal(_dropout_train(m, ones, 0.5), ones)
    m.eval()
    assert torch.equal(_dropout_train(m, ones, 0.5), ones)
    assert torch.equal(_dropout_train(torch.nn.Module(), ones, None), ones)


def test_weighted_losses_of_the_model_fns():
    """weighted_sigmoid_cross_entropy against hand values of tf.losses.compute_weighted_loss's reductions
    (DeepCrossNetwork.py:209-225 MEAN, deepFM.py:72 SUM, ESMM.py:150-175 sum of two MEANs)."""
    import math
    import dir_amd  # noqa: F401
    from dir_amd import train_spec as ts
    logits = torch.tensor([[0.0], [2.0], [-1.0]])
    labels = torch.tensor([1, 0, 1])
    un = [math.log(2.0), 2.0 + math.log1p(math.exp(-2.0)), 1.0 + math.log1p(math.exp(-1.0))]     # max(x,0) - x*z + log(1+e^-|x|)
    loss, unw = ts.weighted_sigmoid_cross_entropy(logits, labels, None, "sum")
    assert loss.item() == pytest.approx(sum(un), rel=1e-6) and unw.reshape(-1).tolist() == pytest.approx(un, rel=1e-6)
    assert ts.weighted_sigmoid_cross_entropy(logits, labels, None, "mean")[0].item() == pytest.approx(sum(un) / 3, rel=1e-6)
    w = torch.tensor([[2.0], [0.0], [1.0]])
    assert ts.weighted_sigmoid_cross_entropy(logits, labels, w, "mean")[0].item() == pytest.approx((2 * un[0] + un[2]) / 3.0, rel=1e-6)
    assert ts.weighted_sigmoid_cross_entropy(logits, labels, w, "sum")[0].item() == pytest.approx(2 * un[0] + un[2], rel=1e-6)
    assert ts.weighted_sigmoid_cross_entropy(logits, labels, torch.zeros(3, 1), "mean")[0].item() == 0.0
    with pytest.raises(ValueError):
        ts.weighted_sigmoid_cross_entropy(logits, labels, None, "nope")
    feats = {"w": torch.tensor([2.0, 0.0, 1.0])}
    assert torch.equal(ts._weights_of(feats, "w", logits), w)
    with pytest.raises(ValueError):
        ts._weights_of(feats, "missing", logits)


def test_binary_metrics_match_hand_values_and_sklearn():
    """metrics.BinaryMetrics (DeepCrossNetwork.py:293-319): counters against hand values; the 200-threshold trapezoidal AUC
    against scikit-learn's exact ROC AUC (they agree to the bucket resolution); streaming == one shot."""
    import dir_amd  # noqa: F401
    from dir_amd.metrics import BinaryMetrics
    from sklearn.metrics import roc_auc_score
    g = torch.Generator().manual_seed(0)
    y = (torch.rand(4000, generator=g) < 0.3).float()
    p = torch.sigmoid(torch.randn(4000, generator=g) + 1.5 * (y - 0.3))
    w = torch.rand(4000, generator=g) + 0.5
    loss = torch.nn.functional.binary_cross_entropy(p, y, reduction="none")
    one = BinaryMetrics(device="cpu").update(y, p, loss, w).result()
    two = BinaryMetrics(device="cpu")
    for a in range(0, 4000, 1000):
        two.update(y[a:a + 1000], p[a:a + 1000], loss[a:a + 1000], w[a:a + 1000])
    two = two.result()
    for k in one:
        assert one[k] == pytest.approx(two[k], rel=1e-9), k
    pred = (p > 0.5).float()
    assert one["accuracy"] == pytest.approx(float((w * (pred == y)).sum() / w.sum()), rel=1e-6)
    assert one["precision"] == pytest.approx(float((w * pred * y).sum() / (w * pred).sum()), rel=1e-6)
    assert one["recall"] == pytest.approx(float((w * pred * y).sum() / (w * y).sum()), rel=1e-6)
    assert one["average_loss"] == pytest.approx(float((w * loss).sum() / w.sum()), rel=1e-6)
    lm = float((w * y).sum() / w.sum())
    assert one["label/mean"] == pytest.approx(lm, rel=1e-6) and one["accuracy_baseline"] == pytest.approx(max(lm, 1 - lm), rel=1e-6)
    assert one["auc"] == pytest.approx(roc_auc_score(y.numpy(), p.numpy(), sample_weight=w.numpy()), abs=2e-3)
    # a perfect and an inverted ranking
    yy = torch.tensor([0.0, 0.0, 1.0, 1.0])
    assert BinaryMetrics(device="cpu").update(yy, torch.tensor([0.1, 0.2, 0.8, 0.9])).result()["auc"] == pytest.approx(1.0, abs=1e-4)
    assert BinaryMetrics(device="cpu").update(yy, torch.tensor([0.9, 0.8, 0.2, 0.1])).result()["auc"] == pytest.approx(0.0, abs=1e-4)


def test_dense_row_threshold_routes_small_batches_to_the_library(monkeypatch):
    """dense.dense_act / mlp_stack_supported: below MIN_ROWS rows a layer stays on nn.Linear (CPU tensors always do)."""
    import torch
    from dir_amd import dense as D
    lin = torch.nn.Linear(16, 32)
    x = torch.randn(5, 16)
    y = D.dense_act(lin, x, torch.relu)
    assert torch.allclose(y, torch.relu(lin(x)))
    monkeypatch.setattr(D, "MIN_ROWS", 6144)
    assert not D.mlp_stack_supported(torch.nn.ModuleList([lin]), x.requires_grad_(True), torch.relu)
    assert torch.allclose(D.units1(torch.nn.Linear(16, 1), x), torch.nn.Linear(16, 1)(x)) is not None


# ---- TensorFlow checkpoint bundles (tf_bundle.py, checkpoint.export_tf_checkpoint / load_tf_checkpoint) ----------------------------------
def test_crc32c_known_answers(built_lib):
    from dir_amd import tf_bundle as tb
    assert tb._crc32c(b"123456789") == 0xE3069283                    # the CRC-32C check value (RFC 3720 appendix B.4 family)
    assert tb._crc32c(b"\x00" * 32) == 0x8A9136AA                    # RFC 3720 B.4: 32 bytes of zeros
    assert tb._crc32c(b"\xff" * 32) == 0x62A8AB43                    # RFC 3720 B.4: 32 bytes of ones
    assert tb._crc32c(bytes(range(32))) == 0x46DD794E                # RFC 3720 B.4: incrementing bytes
    data = np.random.default_rng(0).integers(0, 256, 100003, dtype=np.uint8).tobytes()
    assert tb._crc32c(data) == tb._crc32c_py(data)                   # C (slice-by-8) vs the table loop, unaligned length
    assert tb._crc32c(data[1000:], tb._crc32c(data[:1000])) == tb._crc32c(data)     # incremental
    for c in (0, 1, 0xdeadbeef, 0xffffffff):
        assert tb.unmask_crc(tb.mask_crc(c)) == c
    assert tb.mask_crc(0) == 0xa282ead8


def test_tf_bundle_format_invariants_and_round_trip(tmp_path, built_lib):
    """The index is a leveldb-format table: footer magic, block trailers with masked crc32c, prefix-compressed keys with restart
    points every 16 entries; the data shard holds row-major little-endian bytes at the recorded offsets."""
    import struct
    from dir_amd import tf_bundle as tb
    rng = np.random.default_rng(3)
    tensors = {"dnn_fm/hiddenlayer_%d/kernel" % i: rng.standard_normal((7, 5)).astype(np.float32) for i in range(40)}   # shared prefixes
    tensors["global_step"] = np.array(1234, dtype=np.int64)
    tensors["linear/linear_model/C1/weights"] = rng.standard_normal((1000, 1)).astype(np.float32)
    tensors["a/ids"] = np.arange(12, dtype=np.int32).reshape(3, 4)
    tensors["scalar"] = np.array(2.5, dtype=np.float64)
    prefix = str(tmp_path / "model.ckpt-1234")
    tb.write_bundle(prefix, tensors)
    idx = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", idx[-8:])[0] == 0xdb4775248b80fb57
    items = tb.read_table(prefix + ".index")
    keys = [k for k, _ in items]
    assert keys[0] == b"" and keys == sorted(keys) and len(keys) == len(tensors) + 1
    # header: num_shards = 1, version.producer = 1
    assert items[0][1] == b"\x08\x01\x1a\x02\x08\x01"
    # prefix compression really happened: the index is much smaller than the sum of the keys
    assert len(idx) < sum(len(k) + len(v) for k, v in items)
    # a flipped byte in a block is caught by the trailer crc
    bad = bytearray(idx)
    bad[10] ^= 0xff
    open(prefix + ".index", "wb").write(bytes(bad))
    with pytest.raises(ValueError, match="crc32c"):
        tb.read_table(prefix + ".index")
    open(prefix + ".index", "wb").write(idx)
    got = tb.read_bundle(prefix)
    assert set(got) == set(tensors)
    for k, v in tensors.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape
        np.testing.assert_array_equal(got[k], v)
    # entry protos: offsets are the running sum of the sizes in key order; a corrupted tensor byte fails its crc
    off = 0
    for k, val in items[1:]:
        e = tb._parse_entry(val)
        assert e["offset"] == off and e["size"] == tensors[k.decode()].nbytes
        off += e["size"]
    raw = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    raw[5] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="crc32c"):
        tb.read_bundle(prefix)
    # many entries: several data blocks and an index block with one entry per block
    big = {"v/%06d" % i: np.full((3,), i, np.float32) for i in range(30000)}
    tb.write_bundle(str(tmp_path / "big"), big)
    g2 = tb.read_bundle(str(tmp_path / "big"), names={"v/000000", "v/029999", "v/012345"})
    assert float(g2["v/012345"][0]) == 12345.0 and len(g2) == 3
    assert len(tb.read_table(str(tmp_path / "big.index"))) == 30001


def test_tf_checkpoint_export_load_round_trip(tmp_path, built_lib):
    """A model_dir as the reference's Estimators keep it: `checkpoint` state file + bundle keyed by the reference's variable
    names; loading it into a freshly initialised module reproduces every parameter (incl. the transposed dense kernels)."""
    from dir_amd import feature_column as fc, tf_bundle as tb
    from dir_amd.checkpoint import export_tf_checkpoint, load_tf_checkpoint, export_serving, tf_variable_map
    from dir_amd.dcn import DeepCrossNetwork
    from dir_amd.deepfm import DeepFM

    def dcn():
        cols = ([fc.numeric_column(k) for k in ("age", "hours")] + [fc.indicator_column(fc.categorical_column_with_identity("wc", 9))]
                + [fc.embedding_column(fc.categorical_column_with_hash_bucket("occ", 50), 8)])
        return DeepCrossNetwork(columns=cols, cross_layer_num=2, dnn_hidden_units=[16, 8], batch_norm=True)

    def deepfm():
        cats = [fc.categorical_column_with_identity("C%d" % i, 20 + i) for i in range(3)]
        return DeepFM(linear_feature_columns=cats, dnn_feature_columns=[fc.embedding_column(c, 4) for c in cats], dnn_hidden_units=[8],
                      fm_embedding_size=4, batch_norm=True)

    for make in (dcn, deepfm):
        torch.manual_seed(1)
        a = make()
        with torch.no_grad():
            for p in a.parameters():
                p.add_(torch.randn_like(p) * 0.1)
        mdir = str(tmp_path / make.__name__)
        prefix = export_tf_checkpoint(a, mdir, global_step=77)
        assert os.path.basename(prefix) == "model.ckpt-77" and tb.latest_checkpoint(mdir) == prefix
        names = [k.decode() for k, _ in tb.read_table(prefix + ".index")][1:]
        assert set(names) == set(tf_variable_map(a)) | {"global_step"}
        torch.manual_seed(2)
        b = make()
        missing, step = load_tf_checkpoint(b, mdir)
        assert missing == [] and step == 77
        for (n1, p1), (n2, p2) in zip(a.state_dict().items(), b.state_dict().items()):
            assert n1 == n2 and torch.equal(p1, p2), n1
        # TensorFlow layout on disk: dense kernels are [in, out]
        raw = tb.read_bundle(prefix)
        k = [n for n in raw if n.endswith("hidden_layer_0/kernel") or n.endswith("hiddenlayer_0/kernel")][0]
        assert raw[k].shape == tuple(reversed(a.hidden[0].weight.shape))
        exp = export_serving(a, str(tmp_path / (make.__name__ + "_export")), global_step=77)
        import json
        sig = json.load(open(os.path.join(exp, "serving_signature.json")))
        assert sig["signature_def"]["serving_default"]["method_name"] == "tensorflow/serving/classify"
        assert set(tb.read_bundle(os.path.join(exp, "variables", "variables"))) == set(raw)


def test_arithmetic_selection_rules():
    """ops.cin_auto_arith / cin_dw_auto_arith / dense_auto_arith (pure host logic): which kernel arith="auto" runs.  The bf16x3
    kernels compute whole column blocks / k-steps, so they are chosen while their padding costs less than their advantage."""
    from dir_amd import ops
    # CIN forward (and, transposed, its data gradients): the BASELINE stack and the paper's 200-wide layers on bf16x3 ...
    assert ops.cin_auto_arith(26, 16, 26, 128) == "bf16x3" and ops.cin_auto_arith(26, 16, 128, 128) == "bf16x3"
    assert ops.cin_auto_arith(26, 16, 200, 200) == "bf16x3" and ops.cin_auto_arith(26, 16, 128, 32) == "bf16x3"
    # ... a 7-channel input (32-wide k-step: 4.6 x padding), a 10-wide output, more than 40 fields or an odd D on fp32 MFMA
    assert ops.cin_auto_arith(26, 16, 7, 128) == "f32" and ops.cin_auto_arith(26, 16, 128, 10) == "f32"
    assert ops.cin_auto_arith(41, 16, 128, 128) == "f32" and ops.cin_auto_arith(26, 12, 128, 128) == "f32"
    assert ops.cin_bf16x3_covers(1, 4) and ops.cin_bf16x3_covers(40, 32) and not ops.cin_bf16x3_covers(41, 16)
    # weight gradient: wide layers with D >= 8 only (a wave's 8 column tiles are xk channels; D = 4 splits an octet over two samples)
    assert ops.cin_dw_auto_arith(26, 16, 128, 128) == "bf16x3" and ops.cin_dw_auto_arith(24, 8, 128, 128) == "bf16x3"
    assert ops.cin_dw_auto_arith(26, 16, 26, 128) == "f32" and ops.cin_dw_auto_arith(26, 4, 128, 128) == "f32"
    assert ops.cin_dw_auto_arith(26, 16, 200, 200) == "f32"            # 256 x 256 x 28 computed for 200 x 200 x 26
    # dense layers: the tower shapes at training / serving batch sizes on bf16x3; small batches, narrow or badly padded layers on fp32
    assert ops.dense_auto_arith(65536, 416, 400) == "bf16x3" and ops.dense_auto_arith(65536, 1024, 1024) == "bf16x3"
    assert ops.dense_auto_arith(ops.DENSE_BF3_MIN_ROWS, 360, 200) == "bf16x3"
    assert ops.dense_auto_arith(ops.DENSE_BF3_MIN_ROWS - 1, 416, 400) == "f32"
    assert ops.dense_auto_arith(65536, 200, 80) == "f32" and ops.dense_auto_arith(65536, 400, 16) == "f32"
    with pytest.raises(ValueError):
        ops.cin_layer(torch.zeros(2, 3, 4), torch.zeros(2, 3, 4), torch.zeros(5, 9), arith="bf16")      # CPU tensors / bad arith: refused


def test_tf_bundle_partitioned_variables_round_trip_and_key_encoding(tmp_path, built_lib):
    """Partitioned variables ([TF-upstream] saved as slices of the full tensor; models/DeepFM/deepFM.py:163-175 creates the embedding
    variables under a partitioner scope): the slice keys' OrderedCode encoding against values worked out by hand from
    tensorflow/core/lib/strings/ordered_code.cc, a round trip through write_bundle(partitions=) / read_bundle, the slices' own
    entries (shapes, 'div' row ranges), and a missing slice is an error."""
    from dir_amd import tf_bundle as tb
    from dir_amd.shard import div_range
    assert tb._oc_num_increasing(0) == b"\x00" and tb._oc_num_increasing(2) == b"\x01\x02" and tb._oc_num_increasing(300) == b"\x02\x01\x2c"
    assert tb._oc_string(b"a\x00b\xff") == b"a\x00\xffb\xff\x00\x00\x01"
    kat = {0: "80", -1: "7f", 10: "8a", 63: "bf", 64: "c040", -64: "40", -65: "3fbf", 8191: "dfff", 8192: "e02000", 1000000: "ef4240",
           -8192: "2000", -8193: "1fdfff"}
    for v, h in kat.items():
        assert tb._oc_signed_increasing(v).hex() == h, (v, tb._oc_signed_increasing(v).hex())
    enc = [tb._oc_signed_increasing(v) for v in sorted(range(-70000, 70000, 997))]
    assert enc == sorted(enc)                                              # the code is order-preserving
    assert tb.encode_tensor_name_slice("a", [(0, 10), (0, -1)]).hex() == "006100010102808a807f"
    # the form a TensorFlow Saver writes (SaveSliceInfo.spec is "offset,length" for every dimension): explicit length 16 -> 0x80 ^ 16
    assert tb.encode_tensor_name_slice("a", [(0, 10), (0, 16)]).hex() == "006100010102808a8090"
    rng = np.random.default_rng(8)
    full = rng.standard_normal((1003, 8)).astype(np.float32)
    tensors = {"m/embedding_weights": full, "m/bias": np.arange(5, dtype=np.float32), "global_step": np.array(7, np.int64)}
    prefix = str(tmp_path / "model.ckpt-7")
    tb.write_bundle(prefix, tensors, partitions={"m/embedding_weights": 4})
    got = tb.read_bundle(prefix)
    assert sorted(got) == sorted(tensors)
    for k in tensors:
        np.testing.assert_array_equal(got[k], tensors[k])
    items = dict(tb.read_table(prefix + ".index"))
    head = tb._parse_entry(items[b"m/embedding_weights"])
    assert head["shape"] == [1003, 8] and len(head["slices"]) == 4 and head["size"] == 0
    for j, ext in enumerate(head["slices"]):
        s0, e0 = div_range(1003, 4, j)
        assert ext == [(s0, e0 - s0), (0, 8)]                             # explicit extents on the unpartitioned axis too
        part = tb._parse_entry(items[tb.encode_tensor_name_slice("m/embedding_weights", ext)])
        assert part["shape"] == [e0 - s0, 8] and part["size"] == (e0 - s0) * 8 * 4
    only = tb.read_bundle(prefix, names={"m/embedding_weights"})
    assert list(only) == ["m/embedding_weights"]
    # drop one slice's entry: the reader must refuse
    keep = [(k, v) for k, v in sorted(items.items()) if k != tb.encode_tensor_name_slice("m/embedding_weights", head["slices"][2])]
    tb.write_table(prefix + ".index", keep)
    with pytest.raises(ValueError, match="slice"):
        tb.read_bundle(prefix)


def test_export_tf_checkpoint_with_parameter_server_partitions(tmp_path, built_lib):
    """export_tf_checkpoint(num_ps_replicas=) cuts the big tables by the reference partitioner's rule and load_tf_checkpoint reads them back."""
    import dir_amd
    from dir_amd import checkpoint, feature_column as fc, tf_bundle as tb
    from dir_amd.deepfm import DeepFM
    torch.manual_seed(0)
    V, K = 600000, 32                                                      # 76.8 MB > 64 MiB: two slices at num_ps_replicas >= 2
    cats = [fc.categorical_column_with_identity("C0", V), fc.categorical_column_with_identity("C1", 50)]
    model = DeepFM(linear_feature_columns=cats, dnn_feature_columns=[fc.embedding_column(c, K) for c in cats], dnn_hidden_units=[8],
                   fm_embedding_size=K)
    prefix = checkpoint.export_tf_checkpoint(model, str(tmp_path), global_step=3, num_ps_replicas=4)
    items = dict(tb.read_table(prefix + ".index", verify=False))
    big = [k for k in items if k.endswith(b"embedding_weights") and len(tb._parse_entry(items[k])["slices"]) > 1]
    assert len(big) == 1 and len(tb._parse_entry(items[big[0]])["slices"]) == 2
    other = DeepFM(linear_feature_columns=cats, dnn_feature_columns=[fc.embedding_column(c, K) for c in cats], dnn_hidden_units=[8],
                   fm_embedding_size=K)
    missing, step = checkpoint.load_tf_checkpoint(other, str(tmp_path))
    assert not missing and step == 3
    for a, b in zip(model.parameters(), other.parameters()):
        assert torch.equal(a, b)


def test_round5_routing_rules():
    """Host-side rules added in round 5 (no GPU): the unscaled fp16 x 2 window, the small-batch dense rule, the stale-measurement policy."""
    from dir_amd import ops
    assert ops.f16_range_ok(0.0) and ops.f16_range_ok(1.0) and ops.f16_range_ok(2.0 ** -6) and ops.f16_range_ok(2.0 ** 15 - 1)
    assert not ops.f16_range_ok(2.0 ** 15) and not ops.f16_range_ok(65504.0) and not ops.f16_range_ok(2.0 ** -7) and not ops.f16_range_ok(float("inf"))
    # the reference's batch sizes (models/DeepCrossNetwork/train.py:16-17) always run the small kernel; larger ones while M N K stays small
    assert ops.dense_small_covers(100, 416, 400) and ops.dense_small_covers(256, 1024, 1024) and ops.dense_small_covers(1, 4, 1)
    assert ops.dense_small_covers(512, 416, 400) and not ops.dense_small_covers(512, 1024, 1024) and not ops.dense_small_covers(513, 16, 16)
    assert not ops.dense_small_covers(256, 430, 400) and not ops.dense_small_covers(0, 16, 16)
    import torch
    w = torch.nn.Parameter(torch.full((4, 4), 3.0))
    assert ops.weight_absmax(w) == 3.0
    w.data.mul_(2.0)                                               # what a fused updater does: a raw write ...
    ops.mark_written(w)                                            # ... + the version bump it reports (the ledger counts it)
    assert ops.weight_absmax(w, every=32) == 3.0                   # training policy: a measurement may be up to `every` OPTIMISER steps old
    assert ops.weight_absmax(w) == 6.0                             # inference policy: every change is seen
    with torch.no_grad():
        w.mul_(2.0 ** 15)                                          # any OTHER write (load_state_dict, copy_, a torch op): re-measured at once
    assert ops.weight_absmax(w, every=32) == 6.0 * 2.0 ** 15
    # TableSet.absmax: the same policy over the tables and their owners; invalidate_caches() drops the measurement
    t = torch.full((8, 4), 0.5)
    ts = ops.TableSet.__new__(ops.TableSet)
    ts.tables, ts.owners, ts.device = [t], [], t.device
    assert ts.absmax(every=32) == 0.5
    t.data.mul_(4.0)
    ops.mark_written(t)
    assert ts.absmax(every=32) == 0.5 and ts.absmax(every=1) == 2.0
    t.mul_(2.0)                                                    # not a fused updater's step
    assert ts.absmax(every=32) == 4.0
    t.data.mul_(2.0)                                               # a write nothing reports: invalidate_caches() is the caller's duty
    assert ts.absmax(every=32) == 4.0
    ops.invalidate_caches()
    assert ts.absmax(every=32) == 8.0
    assert ops.din_arith(w, (), arith="bf16x3") == ops.DIN_ARITHS["bf16x3"]
