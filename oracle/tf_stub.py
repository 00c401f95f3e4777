"""tf_stub.py -- TEST INFRASTRUCTURE (build container only; never imported by the product, never shipped as a path the
GPU box needs).  A NumPy stand-in for the few dozen `tensorflow` 1.x symbols that the reference's pure-glue closures
touch, installed into sys.modules so that the REFERENCE'S OWN SOURCE TEXT can be imported and executed from
/root/reference ("optional stronger oracle"):

    models/DeepFM/deepFM.py                       _DeepFM_model_fn, _dnn_fm_logit_fn_builder (fm_logit_fn :321-335,
                                                  dnn_logit_fn :284-319), _linear_logit_fn_builder :255-275,
                                                  myself_input_layer :363-400
    models/DeepCrossNetwork/DeepCrossNetwork.py   DeepCrossNetwork.__init__ -> _model_fn -> _dcn_logit_fn_builder :118-141,
                                                  _cross_op / _cross_architecture :336-367, _deep_architecture :370-410,
                                                  _create_estimator_spec :143-222, _create_loss :225-243

WHAT THIS PINS AND WHAT IT DOES NOT.  Executing the reference text pins the *op order, axes, constants, concat orders and
control flow* of those closures to the reference itself instead of to a re-typing of it.  Every primitive the text calls
(concat, reduce_sum, tensordot, dense, batch_norm, embedding lookup, linear_model, input_layer ...) is implemented HERE in
NumPy from TensorFlow 1.x's documented behaviour -- those are a stand-in for the real library, so the numerics of each
primitive are still ours ("[TF-upstream] stated, not verified").  Fixtures written from this harness are labelled
"stubbed tf" (tests/golden/make_ref_text.py) and do not lift parity to "pinned".

Tensors are a thin np.ndarray subclass (`T`) so Python operators keep working; variables come from a registry the caller
fills (`VARS[name] = array`) keyed by the variable-scope path the reference text itself builds.
"""
import contextlib
import importlib.abc
import importlib.machinery
import sys
import types

import numpy as np


class T(np.ndarray):
    """ndarray with the two Tensor methods the reference text calls."""

    def get_shape(self):
        shp = self.shape

        class _S(tuple):
            def as_list(s):
                return list(s)

            def num_elements(s):
                return int(np.prod(s))
        return _S(shp)

    @property
    def name(self):
        return "T"

    @property
    def dtype(self):                      # Tensor.dtype.base_dtype (ESMM.py:73); NumPy itself reads the C field, not this property
        return _DT(np.ndarray.dtype.__get__(self))


class _DT:
    """np.dtype with the one DType attribute the reference text reads (`base_dtype`); everything else is delegated."""

    def __init__(self, dt):
        self._dt = dt
        self.base_dtype = dt

    def __getattr__(self, item):
        return getattr(self._dt, item)

    def __eq__(self, other):
        return self._dt == (other._dt if isinstance(other, _DT) else other)

    def __hash__(self):
        return hash(self._dt)

    def __repr__(self):
        return repr(self._dt)


def _t(a):
    return np.asarray(a).view(T)


class _Auto(types.ModuleType):
    """A module that is also a callable / context manager / attribute factory: anything the reference text imports or
    touches that is not implemented explicitly (summaries, optimizers, metrics ...) resolves to one of these and does
    nothing."""

    def __init__(self, name):
        super().__init__(name)
        self.__path__ = []

    def __getattr__(self, item):
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        full = self.__name__ + "." + item
        child = sys.modules.get(full)
        if child is None:
            child = _Auto(full)
            sys.modules[full] = child
        object.__setattr__(self, item, child)
        return child

    def __call__(self, *a, **k):
        return _Auto(self.__name__ + "()")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def __mro_entries__(self, bases):
        return (object,)

    def __iter__(self):
        return iter(())

    def __bool__(self):
        return True


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname == "tensorflow" or fullname.startswith("tensorflow."):
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = sys.modules.get(spec.name)
        return m if m is not None else _Auto(spec.name)

    def exec_module(self, module):
        pass


def _resolve(path):
    parts = path.split(".")
    m = sys.modules.get(parts[0])
    if m is None:
        m = _Auto(parts[0])
        sys.modules[parts[0]] = m
    for p in parts[1:]:
        m = getattr(m, p)
    return m


def _put(path, value):
    mod, _, name = path.rpartition(".")
    object.__setattr__(_resolve(mod), name, value)


# ---- state the caller drives -------------------------------------------------------------------------------------------------
VARS = {}          # variable-scope path + "/" + name  ->  ndarray  (the caller's weights)
CREATED = []       # names requested through get_variable / dense / batch_norm, in creation order (the reference's layout)
COLLECTIONS = {}
_SCOPE = []
DTYPE = [np.float32]


def reset(dtype=np.float32):
    VARS.clear()
    CREATED.clear()
    COLLECTIONS.clear()
    del _SCOPE[:]
    DTYPE[0] = dtype


def _scope_name():
    return "/".join(s for s in _SCOPE if s)


class _VarScope:
    def __init__(self, name_or_scope=None, default_name=None, **kw):
        if isinstance(name_or_scope, _VarScope):
            self._abs, self.part = name_or_scope.name, None
        else:
            self._abs, self.part = None, name_or_scope if name_or_scope is not None else default_name
        self.name = None

    def __enter__(self):
        if self._abs is not None:      # re-entering a captured scope object: absolute path, as TF does
            self._saved = list(_SCOPE)
            _SCOPE[:] = self._abs.split("/")
        else:
            _SCOPE.append(self.part)
        self.name = _scope_name()
        return self

    def __exit__(self, *exc):
        if self._abs is not None:
            _SCOPE[:] = self._saved
        else:
            _SCOPE.pop()
        return False


def _get_variable(name, shape=None, **kw):
    full = (_scope_name() + "/" + name) if _scope_name() else name
    CREATED.append(full)
    if full not in VARS:
        raise KeyError("tf_stub: the reference text asked for variable %r (shape %s) that the caller did not provide" % (full, shape))
    v = np.asarray(VARS[full], dtype=DTYPE[0])
    if shape is not None and tuple(int(s) for s in shape) != v.shape:
        raise ValueError("tf_stub: variable %r has shape %s, the reference text asked for %s" % (full, v.shape, tuple(shape)))
    return _t(v)


# ---- primitives ([TF-upstream] behaviour restated in NumPy; the dtype follows the inputs: fp32 or fp64 runs) -------------------
def _concat(values, axis=0, name=None):
    return _t(np.concatenate([np.asarray(v) for v in values], axis=axis))


def _reshape(tensor, shape, name=None):
    return _t(np.reshape(np.asarray(tensor), tuple(int(s) for s in shape)))


def _reduce_sum(x, axis=None, keepdims=False, name=None):
    return _t(np.sum(np.asarray(x), axis=axis, keepdims=keepdims))


def _tensordot(a, b, axes):
    return _t(np.tensordot(np.asarray(a), np.asarray(b), axes=axes))


def _sigmoid(x, name=None):
    x = np.asarray(x)
    return _t(1.0 / (1.0 + np.exp(-x)))


def _softmax(x, name=None, axis=-1):
    x = np.asarray(x)
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return _t(e / e.sum(axis=axis, keepdims=True))


def _relu(x, name=None):
    return _t(np.maximum(np.asarray(x), 0))


_relu.__name__ = "relu"


def _dense(inputs, units, activation=None, name=None, **kw):
    """tf.layers.dense: kernel [in, units] + bias [units] under the scope `name` (a scope object re-enters its own path)."""
    with _VarScope(name if name is not None else "dense"):      # an unnamed tf.layers.dense is scoped "dense"
        kernel = _get_variable("kernel", shape=(np.asarray(inputs).shape[-1], units))
        bias = _get_variable("bias", shape=(units,))
    y = np.asarray(inputs) @ np.asarray(kernel) + np.asarray(bias)
    y = _t(y)
    return activation(y) if activation is not None else y


def _bn_infer(x, mean, var, beta, gamma, eps):
    inv = 1.0 / np.sqrt(np.asarray(var) + np.asarray(eps, dtype=np.asarray(x).dtype))
    if gamma is not None:
        inv = inv * np.asarray(gamma)
    return _t(np.asarray(x) * inv + (np.asarray(beta) - np.asarray(mean) * inv))


def _layers_batch_normalization(inputs, momentum=0.99, epsilon=1e-3, training=False, name=None, **kw):
    """tf.layers.batch_normalization (center + scale), inference form only."""
    if training:
        raise NotImplementedError("tf_stub: fixtures are generated in PREDICT / EVAL mode")
    n = np.asarray(inputs).shape[-1]
    with _VarScope(name):
        gamma, beta = _get_variable("gamma", (n,)), _get_variable("beta", (n,))
        mean, var = _get_variable("moving_mean", (n,)), _get_variable("moving_variance", (n,))
    return _bn_infer(inputs, mean, var, beta, gamma, epsilon)


def _contrib_batch_norm(inputs, decay=0.999, center=True, scale=False, epsilon=0.001, is_training=True, reuse=None,
                        scope=None, **kw):
    """tf.contrib.layers.batch_norm defaults: center, NO scale, epsilon 1e-3; inference form only."""
    if is_training:
        raise NotImplementedError("tf_stub: fixtures are generated in PREDICT / EVAL mode")
    n = np.asarray(inputs).shape[-1]
    with _VarScope(scope):
        beta = _get_variable("beta", (n,))
        gamma = _get_variable("gamma", (n,)) if scale else None
        mean, var = _get_variable("moving_mean", (n,)), _get_variable("moving_variance", (n,))
    return _bn_infer(inputs, mean, var, beta, gamma, epsilon)


def _weights_and_check(features, weight_column, logits, **kw):
    """[TF-upstream] head._get_weights_and_check_match_logits: 1.0 without a weight column, else the feature as float [B, 1]."""
    if weight_column is None:
        return 1.0
    w = np.asarray(features[weight_column], dtype=DTYPE[0])
    return _t(w.reshape(w.shape[0], -1))


def _cond(pred, true_fn, false_fn, **kw):
    return true_fn() if bool(pred) else false_fn()


def _sigmoid_xent(labels=None, logits=None, name=None, _sentinel=None):
    x, z = np.asarray(logits), np.asarray(labels)
    return _t(np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x))))


class _Reduction:
    NONE, SUM, MEAN, SUM_OVER_BATCH_SIZE, SUM_BY_NONZERO_WEIGHTS, SUM_OVER_NONZERO_WEIGHTS = (
        "none", "weighted_sum", "weighted_mean", "weighted_sum_over_batch_size", "weighted_sum_by_nonzero_weights",
        "weighted_sum_by_nonzero_weights")


def _compute_weighted_loss(losses, weights=1.0, reduction=_Reduction.SUM_BY_NONZERO_WEIGHTS, **kw):
    l = np.asarray(losses)
    w = np.broadcast_to(np.asarray(weights, dtype=l.dtype), l.shape)
    s = (l * w).sum()
    if reduction == _Reduction.MEAN:
        return _t(s / w.sum())
    if reduction == _Reduction.SUM:
        return _t(s)
    raise NotImplementedError(reduction)


class HParams:
    def __init__(self, **kw):
        self._names = []
        for k, v in kw.items():
            self.add_hparam(k, v)

    def add_hparam(self, name, value):
        if name in self._names:
            raise ValueError("Hyperparameter name is reserved: %s" % name)
        self._names.append(name)
        setattr(self, name, value)

    def values(self):
        return {k: getattr(self, k) for k in self._names}


class _Estimator:
    """tf.estimator.Estimator: keeps the model_fn the subclass hands it, so the harness can call it."""

    def __init__(self, model_fn=None, model_dir=None, config=None, params=None, warm_start_from=None):
        self.model_fn, self.params, self.config = model_fn, params, config


class _EstimatorSpec:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class _ModeKeys:
    TRAIN, EVAL, PREDICT = "train", "eval", "infer"


# ---- feature columns: light value objects; the lookups behind them are the [TF-upstream] semantics of SURVEY 8a A2/A6/A10 ------
class _DenseColumn:
    pass


class EmbeddingColumn(_DenseColumn):
    """embedding_column(categorical, dimension, combiner): one-hot or ragged ids looked up in VARS[scope/embedding_weights]."""

    def __init__(self, key, num_buckets, dimension, combiner="mean"):
        self.key, self.num_buckets, self.dimension, self.combiner = key, num_buckets, dimension, combiner
        self.name = key + "_embedding"
        self._var_scope_name = self.name

    @property
    def _variable_shape(self):
        return T.get_shape(np.empty((self.dimension,)))

    def _get_dense_tensor(self, inputs, weight_collections=None, trainable=None):
        table = np.asarray(_get_variable("embedding_weights", (self.num_buckets, self.dimension)))
        return _t(_bag_lookup(table, inputs.get(self.key), self.combiner))


class NumericColumn(_DenseColumn):
    def __init__(self, key, shape=(1,)):
        self.key, self.shape, self.name = key, tuple(shape), key
        self._var_scope_name = key

    @property
    def _variable_shape(self):
        return T.get_shape(np.empty(self.shape))

    def _get_dense_tensor(self, inputs, weight_collections=None, trainable=None):
        return _t(np.asarray(inputs.get(self.key), dtype=DTYPE[0]).reshape((-1,) + self.shape))


class IndicatorColumn(_DenseColumn):
    def __init__(self, key, num_buckets):
        self.key, self.num_buckets, self.name = key, num_buckets, key + "_indicator"
        self._var_scope_name = self.name

    @property
    def _variable_shape(self):
        return T.get_shape(np.empty((self.num_buckets,)))

    def _get_dense_tensor(self, inputs, weight_collections=None, trainable=None):
        ids = np.asarray(inputs.get(self.key)).reshape(-1)
        out = np.zeros((ids.shape[0], self.num_buckets), DTYPE[0])
        ok = ids >= 0
        out[np.nonzero(ok)[0], ids[ok]] = 1
        return _t(out)


class CategoricalColumn:
    """identity categorical column for linear_model (units = 1 weights [num_buckets, 1], bag-sum)."""

    def __init__(self, key, num_buckets):
        self.key, self.num_buckets, self.name = key, num_buckets, key


def _bag_lookup(table, feat, combiner):
    """[TF-upstream] safe_embedding_lookup_sparse: feat is ids [B] / [B,1] (one-hot) or (ids, offsets[, weights]) CSR per
    sample; id < 0 pruned, empty bag -> zeros, in-order sum, mean = /sum(w), sqrtn = /sqrt(sum(w^2))."""
    table = np.asarray(table)
    dt = table.dtype
    if isinstance(feat, tuple):
        ids, offs = np.asarray(feat[0]), np.asarray(feat[1])
        w = np.asarray(feat[2], dtype=dt) if len(feat) > 2 and feat[2] is not None else np.ones(ids.shape[0], dt)
    else:
        ids = np.asarray(feat).reshape(-1)
        offs = np.arange(ids.shape[0] + 1)
        w = np.ones(ids.shape[0], dt)
    B = offs.shape[0] - 1
    out = np.zeros((B, table.shape[1]), dt)
    for b in range(B):
        acc = np.zeros(table.shape[1], dt)
        sw = dt.type(0)
        sw2 = dt.type(0)
        for e in range(offs[b], offs[b + 1]):
            if ids[e] < 0:
                continue
            acc = acc + w[e] * table[ids[e]]
            sw = sw + w[e]
            sw2 = sw2 + w[e] * w[e]
        if combiner == "mean" and sw != 0:
            acc = acc / sw
        elif combiner == "sqrtn" and sw2 != 0:
            acc = acc / np.sqrt(sw2)
        out[b] = acc
    return out


class _LazyBuilder:
    def __init__(self, features):
        self._f = features

    def get(self, key):
        return self._f[key]


def _normalize_feature_columns(cols):
    return sorted(list(cols), key=lambda c: c.name)       # [TF-upstream] sorts by column name


def _input_layer(features, feature_columns, **kw):
    """tf.feature_column.input_layer: columns concatenated in NAME-SORTED order, each under <scope>/<column scope>."""
    outs = []
    b = _LazyBuilder(features)
    with _VarScope("input_layer"):
        for c in _normalize_feature_columns(feature_columns):
            with _VarScope(c._var_scope_name):
                t = np.asarray(c._get_dense_tensor(b))
            outs.append(t.reshape(t.shape[0], -1))
    return _t(np.concatenate(outs, axis=1))


def _linear_model(features, feature_columns, units=1, sparse_combiner="sum", cols_to_vars=None, **kw):
    """feature_column.linear_model: per column (name-sorted) a [num_buckets, units] weight looked up with the sparse
    combiner, summed over columns in that order, + bias_weights [units]."""
    total = None
    with _VarScope("linear_model"):
        for c in _normalize_feature_columns(feature_columns):
            with _VarScope(c.name):
                w = np.asarray(_get_variable("weights", (c.num_buckets, units)))
            if cols_to_vars is not None:
                cols_to_vars[c] = [_t(w)]
            part = _bag_lookup(w, features[c.key], sparse_combiner)
            total = part if total is None else total + part
        bias = np.asarray(_get_variable("bias_weights", (units,)))
    if cols_to_vars is not None:
        cols_to_vars["bias"] = [_t(bias.reshape(1, units))]       # the text reads bias[0][0]
    return _t(total + bias)


def _add_to_collection(name, value):
    COLLECTIONS.setdefault(name, []).append(value)


def _get_collection(name, scope=None):
    return list(COLLECTIONS.get(name, []))


def _add_n(inputs, name=None):
    acc = np.asarray(inputs[0])
    for x in inputs[1:]:
        acc = acc + np.asarray(x)
    return _t(acc)


def install():
    """Put the stub into sys.modules / sys.meta_path (idempotent)."""
    if any(isinstance(f, _Finder) for f in sys.meta_path):
        return
    if "tensorflow" in sys.modules and not isinstance(sys.modules["tensorflow"], _Auto):
        raise RuntimeError("a real tensorflow is already imported; the stub is for containers without it")
    sys.meta_path.insert(0, _Finder())
    import tensorflow as tf  # noqa: F401  (the _Auto root)

    both = {  # the public tf.* name and the private module the reference imports it from
        "concat": (_concat, ["tensorflow.python.ops.array_ops.concat"]),
        "reshape": (_reshape, ["tensorflow.python.ops.array_ops.reshape", "tensorflow.python.ops.gen_array_ops.reshape"]),
        "expand_dims": (lambda x, axis=-1, name=None: _t(np.expand_dims(np.asarray(x), axis)),
                        ["tensorflow.python.ops.array_ops.expand_dims"]),
        "shape": (lambda x, name=None: np.asarray(np.asarray(x).shape), ["tensorflow.python.ops.array_ops.shape"]),
        "square": (lambda x, name=None: _t(np.square(np.asarray(x))), ["tensorflow.python.ops.math_ops.square"]),
        "reduce_sum": (_reduce_sum, ["tensorflow.python.ops.math_ops.reduce_sum"]),
        "subtract": (lambda a, b, name=None: _t(np.asarray(a) - np.asarray(b)), ["tensorflow.python.ops.math_ops.subtract"]),
        "add_n": (_add_n, ["tensorflow.python.ops.math_ops.add_n"]),
        "tensordot": (_tensordot, []),
        "sigmoid": (_sigmoid, []),
        "zeros_like": (lambda x, **k: _t(np.zeros_like(np.asarray(x))), []),
        "ones_like": (lambda x, **k: _t(np.ones_like(np.asarray(x))), []),
        "argmax": (lambda x, axis=None, name=None, **k: _t(np.argmax(np.asarray(x), axis=axis).astype(np.int64)), []),
        "cast": (lambda x, dtype=None, name=None: bool(x) if dtype is bool else x, []),
        "to_float": (lambda x, name=None: _t(np.asarray(x, dtype=DTYPE[0])), []),
        "cond": (_cond, []),
        "multiply": (lambda a, b, name=None: _t(np.asarray(a) * np.asarray(b)), ["tensorflow.python.ops.math_ops.multiply"]),
        "add": (lambda a, b, name=None: _t(np.asarray(a) + np.asarray(b)), ["tensorflow.python.ops.math_ops.add"]),
        "log": (lambda x, name=None: _t(np.log(np.asarray(x))), ["tensorflow.python.ops.math_ops.log"]),
        "clip_by_value": (lambda t, lo, hi, name=None: _t(np.minimum(np.maximum(np.asarray(t), np.asarray(lo)), np.asarray(hi))),
                          ["tensorflow.python.ops.clip_ops.clip_by_value"]),
        "convert_to_tensor": (lambda v, dtype=None, name=None, **k: _t(np.asarray(v, dtype=dtype)),
                              ["tensorflow.python.framework.ops.convert_to_tensor"]),
        "get_variable": (_get_variable, []),
        "variable_scope": (_VarScope, ["tensorflow.python.ops.variable_scope.variable_scope"]),
        "name_scope": (lambda *a, **k: contextlib.nullcontext(), []),
    }
    for name, (fn, extra) in both.items():
        _put("tensorflow." + name, fn)
        for p in extra:
            _put(p, fn)
    _put("tensorflow.bool", bool)
    _put("tensorflow.float32", np.float32)
    _put("tensorflow.nn.relu", _relu)
    _put("tensorflow.python.ops.nn.relu", _relu)
    _put("tensorflow.nn.softmax", _softmax)
    _put("tensorflow.nn.sigmoid_cross_entropy_with_logits", _sigmoid_xent)
    _put("tensorflow.losses.compute_weighted_loss", _compute_weighted_loss)
    _put("tensorflow.losses.Reduction", _Reduction)
    _put("tensorflow.layers.dense", _dense)
    _put("tensorflow.python.layers.core.dense", _dense)
    _put("tensorflow.python.layers.normalization.batch_normalization", _layers_batch_normalization)
    _put("tensorflow.contrib.layers.batch_norm", _contrib_batch_norm)
    _put("tensorflow.contrib.training.HParams", HParams)
    _put("tensorflow.estimator.Estimator", _Estimator)
    _put("tensorflow.estimator.EstimatorSpec", _EstimatorSpec)
    _put("tensorflow.estimator.ModeKeys", _ModeKeys)
    _put("tensorflow.python.estimator.model_fn.ModeKeys", _ModeKeys)
    _put("tensorflow.feature_column.input_layer", _input_layer)
    _put("tensorflow.python.feature_column.feature_column_lib.linear_model", _linear_model)
    _put("tensorflow.python.feature_column.feature_column._normalize_feature_columns", _normalize_feature_columns)
    _put("tensorflow.python.feature_column.feature_column._LazyBuilder", _LazyBuilder)
    _put("tensorflow.python.feature_column.feature_column._DenseColumn", _DenseColumn)
    _put("tensorflow.python.estimator.canned.head._get_weights_and_check_match_logits", _weights_and_check)
    _put("tensorflow.python.framework.ops.add_to_collection", _add_to_collection)
    _put("tensorflow.python.framework.ops.get_collection", _get_collection)
    _put("tensorflow.python.ops.variable_scope.get_variable_scope",
         lambda: types.SimpleNamespace(name=_scope_name()))
    _put("tensorflow.python.training.sync_replicas_optimizer.SyncReplicasOptimizer", type("SyncReplicasOptimizer", (), {}))
    _put("tensorflow.python.ops.partitioned_variables.min_max_variable_partitioner",
         lambda max_partitions=1, axis=0, min_slice_size=256 << 10, bytes_per_string_element=16:
         types.SimpleNamespace(max_partitions=max_partitions, min_slice_size=min_slice_size))


def load_reference(path, modname):
    """Import one reference source file by path under the stub (build container only)."""
    import importlib.util
    install()
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
