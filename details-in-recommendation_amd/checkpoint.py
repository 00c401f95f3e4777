"""checkpoint.py -- parameters <-> the reference's TensorFlow checkpoint variable names and layouts.

SURVEY.md 8(f) rank 4.  The reference models are tf.estimator.Estimators that checkpoint into `model_dir`
(models/DeepFM/deepFM.py:56,138-140; models/DeepCrossNetwork/DeepCrossNetwork.py:36,115; models/ESMM/ESMM.py:60).
This module maps every parameter of the modules here to the variable name the reference graph would give it
(variable scopes: deepFM.py:173-196,206-209,286-317; DeepCrossNetwork.py:109,125,134,329-331,393-403;
ESMM.py:62-66,139-146) and to TensorFlow's layout (dense kernels are [in, out], i.e. the transpose of
nn.Linear.weight; linear_model weights are [vocab, 1]).  `export_tf_checkpoint` / `load_tf_checkpoint` write and read
TensorFlow's own checkpoint bundle format (tf_bundle.py: `model.ckpt-N.index` + `.data-00000-of-00001` + the `checkpoint`
state file of an Estimator's model_dir), so a checkpoint trained with the reference loads here by name without TensorFlow, and
one written here is what tf.train.load_checkpoint / warm_start_from (deepFM.py:71,140) read.  `export_serving` writes the
SavedModel-LIKE artefact of the reference's FinalExporter (DeepCrossNetwork/train.py:170-175): variables/ in bundle format +
the parsing / classification signature as JSON (the GraphDef itself cannot be produced without TensorFlow).  `export_npz` /
`load_npz` are the same mapping through a plain .npz.

[TF-upstream] names that cannot be verified in this container (TensorFlow is not installable): the
`<column>/embedding_weights`, `linear_model/<column>/weights`, `batch_normalization` (gamma/beta/moving_*) and
contrib `BatchNorm` (beta/moving_*) suffixes follow TF r1.10-r1.13 naming.
"""
import numpy as np
import torch

from ._input import categorical_of


def _lin(prefix, layer, out):
    out[prefix + "/kernel"] = (layer.weight, "T")       # TF dense kernel = weight^T
    out[prefix + "/bias"] = (layer.bias, None)


def tf_variable_map(model):
    """-> {tf_name: (parameter_or_buffer, layout)} with layout None | 'T' (transpose) | 'col' ([V] <-> [V,1])."""
    from .deepfm import DeepFM
    from .dcn import DeepCrossNetwork
    from .esmm import ESMM
    m = {}
    if isinstance(model, DeepFM):
        for c, p in zip(model.dnn_feature_columns, model.embedding_weights):
            m["dnn_fm_inputs/myself_input_layer/%s/embedding_weights" % c.name] = (p, None)    # deepFM.py:173-177,386
        for i, lin in enumerate(model.hidden):
            _lin("dnn_fm/hiddenlayer_%d" % i, lin, m)                                          # deepFM.py:293-300
            if len(model.bns):
                bn = model.bns[i]
                pre = "dnn_fm/hiddenlayer_%d/batchnorm_%d" % (i, i)                            # deepFM.py:304-308
                m[pre + "/gamma"], m[pre + "/beta"] = (bn.gamma, None), (bn.beta, None)
                m[pre + "/moving_mean"], m[pre + "/moving_variance"] = (bn.moving_mean, None), (bn.moving_variance, None)
        if len(model.hidden) or model.dnn_feature_columns:
            _lin("dnn_fm/logits", model.logits_layer, m)                                       # deepFM.py:311-317
        for c, p in zip(model.linear_feature_columns, model.linear_weights):
            m["linear/linear_model/%s/weights" % categorical_of(c).name] = (p, "col" if p.dim() == 1 else None)   # deepFM.py:206-213
        m["linear/linear_model/bias_weights"] = (model.linear_bias, None)
    elif isinstance(model, DeepCrossNetwork):
        il = model.input_layer
        for c, p in zip(il.emb_cols, il.embedding_weights):
            m["dcn_model/input_from_feature_columns/input_layer/%s/embedding_weights" % c.name] = (p, None)   # :109,125-126
        m["dcn_model/input_from_feature_columns/cross_w"] = (model.cross_w, None)              # :329-330
        m["dcn_model/input_from_feature_columns/cross_b"] = (model.cross_b, None)              # :331-332
        bi = 0
        n = len(model.hidden)
        for i, lin in enumerate(model.hidden):
            pre = "dcn_model/input_from_feature_columns/hidden_layer_%d" % i                   # :393-399
            _lin(pre, lin, m)
            if model.batch_norm and i < n - 1:
                bn = model.bns[bi]
                bi += 1
                m[pre + "/bn_%d/beta" % i] = (bn.beta, None)                                   # :403, 418-419
                m[pre + "/bn_%d/moving_mean" % i] = (bn.moving_mean, None)
                m[pre + "/bn_%d/moving_variance" % i] = (bn.moving_variance, None)
        _lin("dcn_model/logits/dense", model.logits_layer, m)                                  # :134-137
    elif isinstance(model, ESMM):
        for scope, tower in (("esmm/ctr_model", model.ctr_model), ("esmm/cvr_model", model.cvr_model)):   # ESMM.py:62-66
            il = tower.input_layer
            for c, p in zip(il.emb_cols, il.embedding_weights):
                m["%s/input_layer/%s/embedding_weights" % (scope, c.name)] = (p, None)          # ESMM.py:135
            for i, lin in enumerate(tower.hidden):
                _lin("%s/hiddenlayer_%d" % (scope, i), lin, m)                                  # ESMM.py:137-142
            _lin("%s/dense" % scope, tower.logits, m)                                           # ESMM.py:146
    else:
        raise TypeError("no TensorFlow name mapping for %s" % type(model).__name__)
    return m


def _to_tf(t, layout):
    a = t.detach().cpu().numpy()
    if layout == "T":
        a = a.T
    elif layout == "col":
        a = a.reshape(-1, 1)
    return np.ascontiguousarray(a)


def export_npz(model, path):
    """Write every mapped variable in TensorFlow's name and layout."""
    np.savez(path, **{k: _to_tf(t, lay) for k, (t, lay) in tf_variable_map(model).items()})


def load_npz(model, path, strict=True):
    """Load a .npz keyed by TensorFlow variable names (e.g. dumped from a reference checkpoint)."""
    data = np.load(path)
    vm = tf_variable_map(model)
    missing = [k for k in vm if k not in data.files]
    if strict and missing:
        raise KeyError("variables missing from %s: %s" % (path, missing[:5]))
    with torch.no_grad():
        for k, (t, lay) in vm.items():
            if k not in data.files:
                continue
            a = data[k]
            if lay == "T":
                a = a.T
            elif lay == "col":
                a = a.reshape(-1)
            if tuple(a.shape) != tuple(t.shape):
                raise ValueError("%s: checkpoint shape %s vs parameter %s" % (k, a.shape, tuple(t.shape)))
            t.copy_(torch.from_numpy(np.ascontiguousarray(a)).to(t.device, t.dtype))
    from . import ops
    ops.invalidate_caches()          # weight images cached per tensor version: a load is a new version whatever the counters say
    return missing


# ---- TensorFlow checkpoint bundles (tf_bundle.py) ---------------------------------------------------------------------------------------
def tf_tensors(model):
    """{tf variable name: ndarray in TensorFlow's layout} for every mapped parameter / buffer."""
    return {k: _to_tf(t, lay) for k, (t, lay) in tf_variable_map(model).items()}


def export_tf_checkpoint(model, model_dir, global_step=0, name="model.ckpt", num_ps_replicas=0):
    """Write `model_dir/<name>-<global_step>.{index,data-00000-of-00001}` + the `checkpoint` state file, keyed by the reference's
    variable names (+ the int64 `global_step` the Estimators keep).  num_ps_replicas > 0: the [vocab, K] embedding / linear weight
    variables are written as PARTITIONED variables, cut by the reference's partitioner (min_max_variable_partitioner(max_partitions =
    num_ps_replicas, min_slice_size = 64 << 20), models/DeepFM/deepFM.py:163-167: shard.partitions_for) -- what a run on parameter
    servers leaves in model_dir.  -> the checkpoint prefix."""
    import os
    from . import tf_bundle
    os.makedirs(model_dir, exist_ok=True)
    tensors = tf_tensors(model)
    partitions = {}
    if num_ps_replicas:
        from .shard import partitions_for
        for k, a in tensors.items():
            if k.endswith("embedding_weights") or k.endswith("/weights"):
                a2 = np.asarray(a)
                if a2.ndim == 2:
                    n = partitions_for(a2.shape[0], a2.shape[1], num_ps_replicas, bytes_per_element=a2.dtype.itemsize)
                    if n > 1:
                        partitions[k] = n
    tensors["global_step"] = np.array(int(global_step), dtype=np.int64)
    ckpt = "%s-%d" % (name, int(global_step))
    tf_bundle.write_bundle(os.path.join(model_dir, ckpt), tensors, partitions=partitions)
    tf_bundle.write_checkpoint_state(model_dir, ckpt)
    return os.path.join(model_dir, ckpt)


def load_tf_checkpoint(model, path, strict=True):
    """Load a TensorFlow checkpoint bundle (a prefix, or a model_dir whose `checkpoint` file names the latest one) into the
    module by variable name.  Optimizer slots (`.../Adagrad`, `.../Ftrl`, ...) and other extra variables are ignored.
    -> (missing names, global_step | None)."""
    import os
    from . import tf_bundle
    prefix = path
    if os.path.isdir(path):
        prefix = tf_bundle.latest_checkpoint(path)
        if prefix is None:
            raise FileNotFoundError("no `checkpoint` state file in %s" % path)
    vm = tf_variable_map(model)
    data = tf_bundle.read_bundle(prefix, names=set(vm) | {"global_step"})
    missing = [k for k in vm if k not in data]
    if strict and missing:
        raise KeyError("variables missing from %s: %s" % (prefix, missing[:5]))
    with torch.no_grad():
        for k, (t, lay) in vm.items():
            if k not in data:
                continue
            a = data[k]
            if lay == "T":
                a = a.T
            elif lay == "col":
                a = a.reshape(-1)
            if tuple(a.shape) != tuple(t.shape):
                raise ValueError("%s: checkpoint shape %s vs parameter %s" % (k, a.shape, tuple(t.shape)))
            t.copy_(torch.from_numpy(np.ascontiguousarray(a)).to(t.device, t.dtype))
    step = int(data["global_step"]) if "global_step" in data else None
    from . import ops
    ops.invalidate_caches()
    return missing, step


def export_serving(model, export_dir, global_step=0):
    """The FinalExporter artefact (DeepCrossNetwork/train.py:170-175, ESMM/train.py:194-199) as far as it exists without
    TensorFlow: `variables/variables.{index,data-...}` in bundle format (what a SavedModel holds) and `serving_signature.json`:
    the parsing feature spec of the columns ([TF-upstream] make_parse_example_spec) and the serving_default signature --
    ClassificationOutput(scores=probabilities) (DeepCrossNetwork.py:166-171)."""
    import json
    import os
    from . import tf_bundle
    from ._input import categorical_of
    from .feature_column import NumericColumn
    vdir = os.path.join(export_dir, "variables")
    os.makedirs(vdir, exist_ok=True)
    tensors = tf_tensors(model)
    tensors["global_step"] = np.array(int(global_step), dtype=np.int64)
    tf_bundle.write_bundle(os.path.join(vdir, "variables"), tensors)
    cols = list(getattr(model, "columns", None) or (list(getattr(model, "linear_feature_columns", [])) + list(getattr(model, "dnn_feature_columns", []))))
    spec = {}
    for c in cols:
        if isinstance(c, NumericColumn):
            spec[c.key] = {"kind": "FixedLenFeature", "shape": list(c.shape), "dtype": "float32"}
        else:
            cat = categorical_of(c)
            spec[cat.key] = {"kind": "VarLenFeature", "dtype": "int64"}
            if getattr(cat, "weight_key", None):
                spec[cat.weight_key] = {"kind": "VarLenFeature", "dtype": "float32"}
    sig = {"signature_def": {"serving_default": {"method_name": "tensorflow/serving/classify",
                                                 "inputs": {"inputs": "serialized tf.Example, parsed with feature_spec"},
                                                 "outputs": {"scores": "probabilities [B, n_classes]"}}},
           "feature_spec": spec, "variables": sorted(tensors), "model": type(model).__name__}
    with open(os.path.join(export_dir, "serving_signature.json"), "w") as f:
        json.dump(sig, f, indent=1, sort_keys=True)
    return export_dir
