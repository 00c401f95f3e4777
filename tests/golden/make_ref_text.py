#!/usr/bin/env python3
"""make_ref_text.py -- writes tests/golden/ref_text_v1.npz by EXECUTING THE REFERENCE'S OWN SOURCE TEXT
(/root/reference/models/DeepFM/deepFM.py, models/DeepCrossNetwork/DeepCrossNetwork.py and models/ESMM/ESMM.py -> ref_text_v2_esmm.npz)
under oracle/tf_stub.py, a NumPy
stand-in for the tensorflow symbols those files touch.  Build container only: /root/reference does not exist on the GPU
box and no reference file travels -- only the arrays written here do.

LABEL: "stubbed tf".  The op order, axes, constants, concat orders, scope names and control flow of
    fm_logit_fn / dnn_logit_fn / dnn_fm_logit_fn        deepFM.py:284-338
    _DeepFM_model_fn (inputs, add_n of the logits)      deepFM.py:143-252
    myself_input_layer                                  deepFM.py:363-400
    _cross_op / _cross_architecture                     DeepCrossNetwork.py:336-367
    _deep_architecture / _dcn_logit_fn_builder          DeepCrossNetwork.py:118-141,370-419
    _create_estimator_spec / _create_loss               DeepCrossNetwork.py:143-243
are the reference's (its text ran); the arithmetic of every primitive underneath (concat, reduce_sum, dense, batch norm,
embedding lookup, linear_model, input_layer) is the stub's NumPy restatement of TensorFlow 1.x's documented behaviour.  So
these vectors pin the oracle's *formulas* to the reference text; they do not make the TF-upstream numerics "verified".

Every case is run twice: fp32 (what TF would compute in) and fp64 (the tolerance anchor).  Variable names recorded in
`*_created` are the names the reference text itself asked for, in creation order -- checkpoint.py's name map is tested
against them.

Run from the repo root:  python tests/golden/make_ref_text.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from oracle import tf_stub as S  # noqa: E402


def _closure(fn, name):
    return fn.__closure__[fn.__code__.co_freevars.index(name)].cell_contents


class _Head:
    """Stands in for head_lib._binary_logistic_head...: the model_fn hands it the assembled logits (deepFM.py:247-252)."""
    logits_dimension = 1

    def create_estimator_spec(self, features, mode, labels, train_op_fn, logits):
        return logits


def _glorot(rng, fi, fo):
    lim = np.sqrt(6.0 / (fi + fo))
    return rng.uniform(-lim, lim, size=(fi, fo)).astype(np.float32)


def deepfm_cases(g, rng):
    mod = S.load_reference(os.path.join(REF, "models/DeepFM/deepFM.py"), "ref_deepfm")
    B, K = 48, 8
    vocab = [37, 41, 11, 53, 29, 17]
    F = len(vocab)
    lin_extra = [7, 13]                        # two more linear-only categorical columns (the bucketised numerics' place)
    hidden = [32, 16]
    ids = np.stack([rng.integers(-1, v, size=B) for v in vocab], axis=1).astype(np.int64)          # -1: pruned
    ids_extra = np.stack([rng.integers(0, v, size=B) for v in lin_extra], axis=1).astype(np.int64)
    # ragged weighted stand-ins for columns 1 ('mean') and 4 ('sum')
    rag = {}
    for f in (1, 4):
        lens = rng.integers(0, 5, size=B)
        lens[::7] = 0
        offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        vals = rng.integers(-1, vocab[f], size=offs[-1]).astype(np.int64)
        w = rng.uniform(0.25, 2.0, size=offs[-1]).astype(np.float32)
        rag[f] = (vals, offs, w)
        g["deepfm_rag%d_values" % f], g["deepfm_rag%d_offsets" % f], g["deepfm_rag%d_weights" % f] = vals, offs, w
    g["deepfm_vocab"], g["deepfm_lin_extra_vocab"] = np.array(vocab), np.array(lin_extra)
    g["deepfm_ids"], g["deepfm_ids_extra"], g["deepfm_hidden"] = ids, ids_extra, np.array(hidden)

    weights = {}
    for f, v in enumerate(vocab):
        weights["dnn_fm_inputs/myself_input_layer/C%d_embedding/embedding_weights" % f] = (
            rng.standard_normal((v, K)) / np.sqrt(K)).astype(np.float32)
        weights["linear/linear_model/C%d/weights" % f] = (rng.standard_normal((v, 1)) * 0.05).astype(np.float32)
    for j, v in enumerate(lin_extra):
        weights["linear/linear_model/L%d/weights" % j] = (rng.standard_normal((v, 1)) * 0.05).astype(np.float32)
    weights["linear/linear_model/bias_weights"] = np.array([0.03125], np.float32)
    d = F * K
    for i, n in enumerate(hidden):
        pre = "dnn_fm/hiddenlayer_%d" % i
        weights[pre + "/kernel"] = _glorot(rng, d, n)
        weights[pre + "/bias"] = (rng.standard_normal(n) * 0.05).astype(np.float32)
        bn = pre + "/batchnorm_%d" % i
        weights[bn + "/gamma"] = rng.uniform(0.5, 1.5, n).astype(np.float32)
        weights[bn + "/beta"] = (rng.standard_normal(n) * 0.1).astype(np.float32)
        weights[bn + "/moving_mean"] = (rng.standard_normal(n) * 0.1).astype(np.float32)
        weights[bn + "/moving_variance"] = rng.uniform(0.5, 2.0, n).astype(np.float32)
        d = n
    weights["dnn_fm/logits/kernel"] = _glorot(rng, d, 1)
    weights["dnn_fm/logits/bias"] = np.array([-0.0625], np.float32)
    for k, v in weights.items():
        g["deepfm_var:" + k] = v

    for variant in ("onehot", "onehot_bn", "ragged"):
        combs = ["mean"] * F
        if variant == "ragged":
            combs[4] = "sum"
        dnn_cols = [S.EmbeddingColumn("C%d" % f, vocab[f], K, combs[f]) for f in range(F)]
        lin_cols = [S.CategoricalColumn("C%d" % f, vocab[f]) for f in range(F)] + [
            S.CategoricalColumn("L%d" % j, lin_extra[j]) for j in range(len(lin_extra))]
        feats = {"C%d" % f: ids[:, f:f + 1] for f in range(F)}
        feats.update({"L%d" % j: ids_extra[:, j:j + 1] for j in range(len(lin_extra))})
        if variant == "ragged":
            for f in (1, 4):
                feats["C%d" % f] = rag[f]
        for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
            S.reset(dt)
            S.VARS.update(weights)
            logits = mod._DeepFM_model_fn(
                features=feats, labels=None, mode=S._ModeKeys.PREDICT, head=_Head(), linear_feature_columns=lin_cols,
                dnn_feature_columns=dnn_cols, dnn_hidden_units=hidden, dnn_activation_fn=S._relu, dnn_dropout=0.5,
                fm_embedding_size=K, batch_norm=(variant == "onehot_bn"))
            g["deepfm_%s_logits_%s" % (variant, tag)] = np.asarray(logits)
            created = list(S.CREATED)
            # the pieces, through the same reference closures
            S.reset(dt)
            S.VARS.update(weights)
            with S._VarScope("dnn_fm_inputs"):
                inputs = mod.myself_input_layer(feats, list(set(dnn_cols)))                         # deepFM.py:176-177
            names = [c.name for c in dnn_cols]
            g["deepfm_%s_inputs_%s" % (variant, tag)] = np.concatenate([np.asarray(inputs[n]) for n in names], axis=1)
            with S._VarScope("dnn_fm"):
                both = mod._dnn_fm_logit_fn_builder(units=1, hidden_units=hidden, column_names=names, activation_fn=S._relu,
                                                    dropout=0.5, batch_norm=(variant == "onehot_bn"), fm_embedding_size=K)
                g["deepfm_%s_fm_%s" % (variant, tag)] = np.asarray(_closure(both, "fm_logit_fn")(inputs))
                g["deepfm_%s_dnn_%s" % (variant, tag)] = np.asarray(_closure(both, "dnn_logit_fn")(inputs, S._ModeKeys.PREDICT))
            # the dead duplicate _fm_logit_fn_builder (deepFM.py:343-360) must agree with the live closure
            dup = mod._fm_logit_fn_builder(names, K)(inputs)
            assert np.array_equal(np.asarray(dup), g["deepfm_%s_fm_%s" % (variant, tag)])
            with S._VarScope("linear"):
                lin = mod._linear_logit_fn_builder(units=1, feature_columns=lin_cols, sparse_combiner="sum")(feats)
            g["deepfm_%s_linear_%s" % (variant, tag)] = np.asarray(lin)
        g["deepfm_%s_created" % variant] = np.array(created)
    # fm_logit_fn alone at the BASELINE field shape (26 x 16)
    Fb, Kb = 26, 16
    emb = (rng.standard_normal((64, Fb * Kb)) * 0.25).astype(np.float32)
    g["fm26_emb"] = emb
    names = ["f%02d" % i for i in range(Fb)]
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        S.reset(dt)
        inp = {n: S._t(emb[:, i * Kb:(i + 1) * Kb].astype(dt)) for i, n in enumerate(names)}
        g["fm26_logit_%s" % tag] = np.asarray(mod._fm_logit_fn_builder(names, Kb)(inp))


class _Opt:   # _print_params_info wants a class (DeepCrossNetwork.py:253-257)
    pass


def dcn_cases(g, rng):
    mod = S.load_reference(os.path.join(REF, "models/DeepCrossNetwork/DeepCrossNetwork.py"), "ref_dcn")
    # ---- the bare closures -----------------------------------------------------------------------------------------------------
    for d in (51, 416, 429):
        B, L = 64, 3
        x0 = (rng.standard_normal((B, d)) * 0.25).astype(np.float32)
        x = (rng.standard_normal((B, d)) * 0.25).astype(np.float32)
        w = np.clip(rng.standard_normal((L, d)) * 0.1, -0.2, 0.2).astype(np.float32)
        b = np.clip(rng.standard_normal((L, d)) * 0.1, -0.2, 0.2).astype(np.float32)
        g["cross%d_x0" % d], g["cross%d_x" % d], g["cross%d_w" % d], g["cross%d_b" % d] = x0, x, w, b
        for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
            S.reset(dt)
            S.VARS.update({"cross_w": w, "cross_b": b})
            g["cross%d_op_%s" % (d, tag)] = np.asarray(mod._cross_op(S._t(x0.astype(dt)), S._t(x.astype(dt)),
                                                                     S._t(w[1].astype(dt)), S._t(b[1].astype(dt))))
            params = S.HParams(column_num=d, cross_layer_num=L)
            g["cross%d_arch_%s" % (d, tag)] = np.asarray(mod._cross_architecture(S._t(x0.astype(dt)), params))
    # ---- the whole model_fn: numeric + indicator + embedding columns, name-sorted input_layer ------------------------------------
    B = 40
    num_keys = ["age", "hours", "gain"]
    ind = [("workclass", 9), ("marital", 7)]
    embc = [("occupation", 50, 8, "mean"), ("native", 23, 4, "sqrtn")]
    hidden = [32, 16, 8]
    L = 3
    cols = ([S.NumericColumn(k) for k in num_keys] + [S.IndicatorColumn(k, n) for k, n in ind]
            + [S.EmbeddingColumn(k, v, dim, comb) for k, v, dim, comb in embc])
    feats = {k: rng.uniform(0, 1, size=(B, 1)).astype(np.float32) for k in num_keys}
    for k, n in ind:
        feats[k] = rng.integers(-1, n, size=(B, 1)).astype(np.int64)
    feats["occupation"] = rng.integers(-1, 50, size=(B, 1)).astype(np.int64)
    lens = rng.integers(0, 4, size=B)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    feats["native"] = (rng.integers(0, 23, size=offs[-1]).astype(np.int64), offs, None)
    for k in num_keys + [k for k, _ in ind] + ["occupation"]:
        g["dcn_feat:" + k] = feats[k]
    g["dcn_feat:native_values"], g["dcn_feat:native_offsets"] = feats["native"][0], offs
    d = len(num_keys) + sum(n for _, n in ind) + sum(dim for _, _, dim, _ in embc)
    pre = "dcn_model/input_from_feature_columns/"
    weights = {}
    for k, v, dim, _ in embc:
        weights[pre + "input_layer/%s_embedding/embedding_weights" % k] = (rng.standard_normal((v, dim)) / np.sqrt(dim)).astype(np.float32)
    weights[pre + "cross_w"] = np.clip(rng.standard_normal((L, d)) * 0.1, -0.2, 0.2).astype(np.float32)
    weights[pre + "cross_b"] = np.clip(rng.standard_normal((L, d)) * 0.1, -0.2, 0.2).astype(np.float32)
    fi = d
    for i, n in enumerate(hidden):
        hp = pre + "hidden_layer_%d" % i
        weights[hp + "/kernel"] = (rng.standard_normal((fi, n)) * np.sqrt(2.0 / (fi + n))).astype(np.float32)
        weights[hp + "/bias"] = (rng.standard_normal(n) * 0.05).astype(np.float32)
        if i < len(hidden) - 1:
            weights[hp + "/bn_%d/beta" % i] = (rng.standard_normal(n) * 0.1).astype(np.float32)
            weights[hp + "/bn_%d/moving_mean" % i] = (rng.standard_normal(n) * 0.1).astype(np.float32)
            weights[hp + "/bn_%d/moving_variance" % i] = rng.uniform(0.5, 2.0, n).astype(np.float32)
        fi = n
    weights["dcn_model/logits/dense/kernel"] = _glorot(rng, d + hidden[-1], 1)
    weights["dcn_model/logits/dense/bias"] = np.array([0.0625], np.float32)
    for k, v in weights.items():
        g["dcn_var:" + k] = v
    g["dcn_hidden"], g["dcn_d"] = np.array(hidden), np.array(d)
    labels = rng.integers(0, 2, size=B).astype(np.int64)
    g["dcn_labels"] = labels
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        S.reset(dt)
        S.VARS.update(weights)
        est = mod.DeepCrossNetwork(columns=cols, cross_layer_num=L, dnn_hidden_units=hidden, dnn_dropout=0.3,
                                   dnn_activation_fn=S._relu, optimizer=_Opt, optimizer_spec={}, batch_norm=True,
                                   learning_rate_spec={"learning_rate": 0.01})
        _stdout = sys.stdout
        sys.stdout = open(os.devnull, "w")      # _print_params_info
        try:
            spec = est.model_fn(feats, None, S._ModeKeys.PREDICT, est.params)
        finally:
            sys.stdout.close()
            sys.stdout = _stdout
        g["dcn_created"] = np.array(S.CREATED)
        for k in ("probabilities", "logistic", "class_ids"):
            g["dcn_%s_%s" % (k, tag)] = np.asarray(spec.predictions[k])
        # logits + the input layer through the same closures
        S.reset(dt)
        S.VARS.update(weights)
        params = S.HParams(feature_columns=cols, cross_layer_num=L, hidden_units=hidden, dnn_activation_fn=S._relu,
                           dnn_dropout=0.3, batch_norm=True, l2_reg=None, weight_column=None)
        with S._VarScope("dcn_model"):
            logits = mod._dcn_logit_fn_builder(params)(features=feats, mode=S._ModeKeys.PREDICT)
        g["dcn_logits_%s" % tag] = np.asarray(logits)
        with S._VarScope("dcn_model"), S._VarScope("input_from_feature_columns"):
            g["dcn_x0_%s" % tag] = np.asarray(S._input_layer(feats, cols))
        # _create_loss (MEAN reduction, DeepCrossNetwork.py:225-243)
        wl, ul, _, _ = mod._create_loss(features=feats, params=params, logits=logits,
                                        labels=S._t(labels.reshape(-1, 1)))
        g["dcn_loss_%s" % tag], g["dcn_unweighted_loss_%s" % tag] = np.asarray(wl), np.asarray(ul)


def esmm_cases(g, rng):
    """ESMM.py:62-175 -- _model_fn (PREDICT and EVAL), _base_model, _get_loss, with the reference text's own scopes:
    two towers under esmm/ctr_model and esmm/cvr_model, each with its OWN input_layer variables."""
    mod = S.load_reference(os.path.join(REF, "models/ESMM/ESMM.py"), "ref_esmm")
    B = 48
    embc = [("user", 40, 8, "mean"), ("item", 60, 8, "mean"), ("tags", 25, 4, "sqrtn")]
    num_keys = ["price", "age"]
    hidden = [24, 12]
    cols = [S.EmbeddingColumn(k, v, dim, comb) for k, v, dim, comb in embc] + [S.NumericColumn(k) for k in num_keys]
    feats = {k: rng.uniform(-1, 1, size=(B, 1)).astype(np.float32) for k in num_keys}
    feats["user"] = rng.integers(-1, 40, size=(B, 1)).astype(np.int64)
    feats["item"] = rng.integers(0, 60, size=(B, 1)).astype(np.int64)
    lens = rng.integers(0, 4, size=B)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    tag_w = rng.uniform(0.5, 2.0, size=offs[-1]).astype(np.float32)
    feats["tags"] = (rng.integers(0, 25, size=offs[-1]).astype(np.int64), offs, tag_w)
    feats["w_click"] = rng.uniform(0.5, 1.5, size=(B, 1)).astype(np.float32)
    for k in num_keys + ["user", "item", "w_click"]:
        g["esmm_feat:" + k] = feats[k]
    g["esmm_feat:tags_values"], g["esmm_feat:tags_offsets"], g["esmm_feat:tags_weights"] = feats["tags"][0], offs, tag_w
    d = sum(dim for _, _, dim, _ in embc) + len(num_keys)
    weights = {}
    for tower in ("ctr_model", "cvr_model"):
        pre = "esmm/%s/" % tower
        for k, v, dim, _ in embc:
            weights[pre + "input_layer/%s_embedding/embedding_weights" % k] = (rng.standard_normal((v, dim)) / np.sqrt(dim)).astype(np.float32)
        fi = d
        for i, n in enumerate(hidden):
            weights[pre + "hiddenlayer_%d/kernel" % i] = (rng.standard_normal((fi, n)) * np.sqrt(2.0 / (fi + n))).astype(np.float32)
            weights[pre + "hiddenlayer_%d/bias" % i] = (rng.standard_normal(n) * 0.05).astype(np.float32)
            fi = n
        weights[pre + "dense/kernel"] = _glorot(rng, fi, 1)
        weights[pre + "dense/bias"] = np.array([0.125 if tower == "ctr_model" else -0.25], np.float32)
    for k, v in weights.items():
        g["esmm_var:" + k] = v
    g["esmm_hidden"] = np.array(hidden)
    click = rng.integers(0, 2, size=B).astype(np.int64)
    convert = (click * rng.integers(0, 2, size=B)).astype(np.int64)
    g["esmm_click_label"], g["esmm_convert_label"] = click, convert
    real_get_loss = mod._get_loss
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        params = S.HParams(feature_columns=cols, ctr_weight_column="w_click", ctcvr_weight_column=None, hidden_units=hidden,
                           dnn_activation_fn=S._relu, dnn_dropout=None, optimizer=None)
        S.reset(dt)
        S.VARS.update(weights)
        spec = mod._model_fn(feats, None, S._ModeKeys.PREDICT, params)
        g["esmm_created"] = np.array(S.CREATED)
        for k in ("probabilities", "logistic", "class_ids"):
            g["esmm_%s_%s" % (k, tag)] = np.asarray(spec.predictions[k])
        # EVAL: the same text goes on to _get_loss; its arguments and results are recorded as the text hands them over
        seen = {}

        def spy(features, labels, logits, params_):
            out = real_get_loss(features, labels, logits, params_)
            seen["logits"], seen["out"] = logits, out
            return out
        mod._get_loss = spy
        try:
            S.reset(dt)
            S.VARS.update(weights)
            labels = {"click_label": S._t(click.copy()), "convert_label": S._t(convert.copy())}
            spec = mod._model_fn(feats, labels, S._ModeKeys.EVAL, params)
        finally:
            mod._get_loss = real_get_loss
        g["esmm_loss_%s" % tag] = np.asarray(spec.loss)
        g["esmm_ctr_logits_%s" % tag] = np.asarray(seen["logits"]["ctr_logits"])
        g["esmm_ctcvr_logits_%s" % tag] = np.asarray(seen["logits"]["ctcvr_logits"])
        g["esmm_weighted_loss_%s" % tag] = np.asarray(seen["out"][0])
        g["esmm_unweighted_loss_%s" % tag] = np.asarray(seen["out"][1])
        # one tower's input layer, through the reference's _base_model scopes
        S.reset(dt)
        S.VARS.update(weights)
        with S._VarScope("esmm"), S._VarScope("cvr_model"):
            g["esmm_cvr_inputs_%s" % tag] = np.asarray(S._input_layer(feats, cols))
            g["esmm_cvr_logits_%s" % tag] = np.asarray(mod._base_model(feats, S._ModeKeys.PREDICT, params))


def main():
    if not os.path.isdir(REF):
        print("make_ref_text: %s is absent (this generator runs in the build container only)" % REF)
        return 1
    rng = np.random.default_rng(20241003)
    g = {}
    deepfm_cases(g, rng)
    dcn_cases(g, rng)
    out = os.path.join(ROOT, "tests", "golden", "ref_text_v1.npz")
    np.savez_compressed(out, **g)
    print("wrote %s: %d arrays, %.1f KB" % (out, len(g), os.path.getsize(out) / 1024))
    g2 = {}
    esmm_cases(g2, np.random.default_rng(20241004))      # its own file and generator state: v1 stays byte-for-byte what it was
    out = os.path.join(ROOT, "tests", "golden", "ref_text_v2_esmm.npz")
    np.savez_compressed(out, **g2)
    print("wrote %s: %d arrays, %.1f KB" % (out, len(g2), os.path.getsize(out) / 1024))
    return 0


if __name__ == "__main__":
    sys.exit(main())
