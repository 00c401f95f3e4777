"""esmm.py -- ESMM (entire-space multi-task) forward on the HIP gather, keeping the reference constructor kwargs.

Mirrors models/ESMM/ESMM.py (reference):
  ESMM.__init__ kwargs                                     :23-32
  _model_fn: two towers under 'ctr_model' / 'cvr_model'    :62-78   (each tower creates its OWN embedding variables)
  _base_model: input_layer -> (dense(act, glorot_normal))* -> dense(1)       :130-147
  predictions: ctr/cvr sigmoid, ctcvr = product, ctcvr_logits = logit(clip(p, 1e-7, 1-1e-7))   :67-92
SURVEY.md 8(f) rank 3: the same gather kernel, used twice per step.

ESMM_W_D mirrors models/ESMM/ESMM_wide_deep.py (the wide & deep variant): every sub-model's logits are dnn_logits + linear_logits
  ESMM_W_D.__init__ kwargs                                  :75-99
  _base_model: dnn_logit_fn(units=1) + linear_logit_fn(units=1), summed        :194-280
  _linear_learning_rate = min(0.2, 1 / sqrt(#linear columns)), _DNN_LEARNING_RATE = 0.05    :30,33,72-73
"""
import torch
from torch import nn

from .dense import dense_act, mlp_head, mlp_head_supported, mlp_stack, mlp_stack_supported, tower_infer, units1
from .dcn import _glorot_normal_
from .deepfm import _dropout_train, _glorot_uniform_
from .input_layer import InputLayer
from ._input import checked_forward as _checked_forward
from . import ops

_EPSILON = 1e-7                                                                  # ESMM.py:19


class _BaseModel(nn.Module):
    def __init__(self, columns, hidden_units, activation, dropout=None):
        super().__init__()
        self.dropout = dropout
        self.input_layer = InputLayer(columns)                                   # ESMM.py:135
        self.activation = activation
        self.hidden = nn.ModuleList()
        d = self.input_layer.column_num
        for n in hidden_units:
            lin = nn.Linear(d, n)
            _glorot_normal_(lin.weight)                                          # ESMM.py:141
            nn.init.zeros_(lin.bias)
            self.hidden.append(lin)
            d = n
        self.logits = nn.Linear(d, 1)                                            # ESMM.py:146
        _glorot_uniform_(self.logits.weight)
        nn.init.zeros_(self.logits.bias)

    def forward(self, features, memo=None):
        src = self.input_layer.onehot_source(features, memo) if not self.dropout or not self.training else None
        if src is not None:                           # inference: the lookups, the tower and the logit layer in ONE launch
            fused = tower_infer(self.hidden, None, self.activation, head=self.logits, gather=(src[0], src[1], None, False))
            if fused is not None:
                return fused
        net = self.input_layer(features, memo=memo)
        # the input is a concatenation of embedding rows (ESMM.py:135 over embedding columns); in inference (the fused tower kernel splits its
        # input unscaled) the tables' magnitudes are checked as well
        emb_in = self.input_layer.embedding_only and (torch.is_grad_enabled() or self.input_layer.embedding_range_ok())
        fused = tower_infer(self.hidden, net, self.activation, head=self.logits, embedding_input=emb_in)      # inference: tower + logit layer in one launch
        if fused is not None:
            return fused
        if not self.dropout and mlp_stack_supported(self.hidden, net, self.activation):
            if mlp_head_supported(self.hidden, self.logits, net, self.activation):
                return mlp_head(self.hidden, self.logits, net, embedding_input=emb_in)   # training: tower + logit layer as one autograd node
            return units1(self.logits, mlp_stack(self.hidden, net, embedding_input=emb_in))      # training: the whole tower as one autograd node
        for lin in self.hidden:
            net = dense_act(lin, net, self.activation)                           # dir_dense_f32 when covered
            net = _dropout_train(self, net, self.dropout)                        # ESMM.py:143-144 (TRAIN only)
        return units1(self.logits, net)


class _LinearModel(nn.Module):
    """linear_model(units=1, sparse_combiner='sum') over categorical columns + bias (ESMM_wide_deep.py:247-262 ->
    [TF-upstream] _linear_logit_fn_builder): one weight per feature value (zeros-initialised, as TF's linear model) and a bias;
    one-hot columns run dir_linear_sparse_sum_f32 (ops.linear_logit) / its autograd node, multi-hot columns the bag kernel with K = 1."""

    def __init__(self, columns, sparse_combiner="sum"):
        super().__init__()
        from ._input import categorical_of
        self.columns = list(columns)
        self.sparse_combiner = sparse_combiner
        self.weights = nn.ParameterList([nn.Parameter(torch.zeros(categorical_of(c).num_buckets)) for c in self.columns])
        self.bias = nn.Parameter(torch.zeros(1))
        self._ts = None

    def _tableset(self):
        from . import ops
        key = tuple(p.data_ptr() for p in self.weights)
        if self._ts is None or self._ts[0] != key:
            ts = ops.TableSet([p.data for p in self.weights])
            ts.owners = list(self.weights)
            self._ts = (key, ts)
        return self._ts[1]

    def forward(self, features, memo=None):
        from . import ops
        from . import autograd as ag
        from ._input import collect_ids
        ts = self._tableset()
        got = collect_ids(self.columns, features, self.bias.device, memo)
        train = torch.is_grad_enabled()
        if got[0] == "onehot":
            if train:
                return ag.linear_logit(ts, got[1], self.bias, list(self.weights))
            return ops.linear_logit(ts, got[1], bias=self.bias.data)
        _, vals, offs, wts, B = got
        if not train:
            return ops.linear_logit(ts, vals, offs, wts, combiner=self.sparse_combiner, bias=self.bias.data, field_major=True)
        per = ag.embedding_bag(ts, vals, list(self.weights), combiner=self.sparse_combiner, offsets=offs, weights=wts, field_major=True)
        return per.view(B, len(self.columns), 1).sum(dim=1) + self.bias


class _BaseModelWD(nn.Module):
    """_base_model of ESMM_wide_deep.py:194-280: dnn tower logits + linear model logits (either part may be absent)."""

    def __init__(self, linear_columns, dnn_columns, hidden_units, activation, dropout=None):
        super().__init__()
        if not linear_columns and not dnn_columns:
            raise ValueError("Either linear_feature_columns or dnn_feature_columns must be defined.")          # :209-211
        if dnn_columns and not hidden_units:
            raise ValueError("dnn_hidden_units must be defined when dnn_feature_columns is specified.")          # :228-231
        self.dnn = _BaseModel(dnn_columns, hidden_units, activation, dropout) if dnn_columns else None
        self.linear = _LinearModel(linear_columns) if linear_columns else None

    def forward(self, features, memo=None):
        if not isinstance(features, dict):
            raise ValueError("features should be a dictionary of `Tensor`s. Given type: {}".format(type(features)))   # :206-208
        logits = None
        if self.dnn is not None:
            logits = self.dnn(features, memo)
        if self.linear is not None:
            lin = self.linear(features, memo)
            logits = lin if logits is None else logits + lin                     # :265-270
        return logits


class ESMM(nn.Module):
    def __init__(self, model_dir=None, columns=None, ctr_weight_column=None, ctcvr_weight_column=None,
                 dnn_hidden_units=None, dnn_dropout=None, config=None, dnn_activation_fn=torch.relu, optimizer=None):
        super().__init__()
        self.hparams = dict(model_dir=model_dir, ctr_weight_column=ctr_weight_column,
                            ctcvr_weight_column=ctcvr_weight_column, dnn_dropout=dnn_dropout, config=config,
                            optimizer=optimizer)
        hidden = list(dnn_hidden_units or [])
        self.ctr_model = _BaseModel(columns, hidden, dnn_activation_fn, dnn_dropout)          # ESMM.py:63-64
        self.cvr_model = _BaseModel(columns, hidden, dnn_activation_fn, dnn_dropout)          # ESMM.py:65-66

    @_checked_forward
    def forward(self, features):
        """-> {'ctr_logits', 'ctcvr_logits'} (the `logits` dict of ESMM.py:77)."""
        memo = {}                                                                # both towers read the same columns: one id matrix
        ctr_logits = self.ctr_model(features, memo)
        cvr_logits = self.cvr_model(features, memo)
        if (not torch.is_grad_enabled() and ctr_logits.is_cuda and ctr_logits.dtype == torch.float32 and ctr_logits.is_contiguous()
                and cvr_logits.is_contiguous() and ctr_logits.shape == cvr_logits.shape):
            ctcvr_logits = ops.esmm_head(ctr_logits, cvr_logits, _EPSILON)      # inference: :69-74 in one launch (seven library launches otherwise)
        else:
            ctcvr_logistic = torch.sigmoid(ctr_logits) * torch.sigmoid(cvr_logits)  # :69-71
            p = ctcvr_logistic.clamp(_EPSILON, 1 - _EPSILON)                         # :73-74
            ctcvr_logits = torch.log(p / (1 - p))
        out = {"ctr_logits": ctr_logits, "ctcvr_logits": ctcvr_logits, "cvr_logits": cvr_logits}
        from ._input import raise_pending
        raise_pending()                                                          # id-range verdicts of both towers' input layers
        return out

    def fused_sparse_adagrad(self, lr, initial_accumulator_value=0.1):
        """The fused sparse Adagrad on both towers' embedding tables (InputLayer.fused_sparse_adagrad); the two towers see the same
        ids, so their sorted updates share one sort per step (ops.share_sorted_entries).  -> the optimiser objects."""
        from . import ops
        a = self.ctr_model.input_layer.fused_sparse_adagrad(lr, initial_accumulator_value)
        b = self.cvr_model.input_layer.fused_sparse_adagrad(lr, initial_accumulator_value)
        for x, y in zip(a, b):
            ops.share_sorted_entries(x, y)
        return a + b

    def get_loss(self, features, labels, logits):
        """_get_loss (ESMM.py:150-175): labels {'click_label', 'convert_label'}; CTR and CTCVR sigmoid cross entropies, each
        MEAN-reduced with its own weight column, added -> (weighted_loss, unweighted_loss = ctr + ctcvr per example)."""
        from .train_spec import _weights_of, weighted_sigmoid_cross_entropy
        hp = self.hparams
        ctr_w = _weights_of(features, hp["ctr_weight_column"], logits["ctr_logits"])
        ctcvr_w = _weights_of(features, hp["ctcvr_weight_column"], logits["ctcvr_logits"])
        ctr_loss, ctr_un = weighted_sigmoid_cross_entropy(logits["ctr_logits"], labels["click_label"], ctr_w, "mean")
        ctcvr_loss, ctcvr_un = weighted_sigmoid_cross_entropy(logits["ctcvr_logits"], labels["convert_label"], ctcvr_w, "mean")
        return ctr_loss + ctcvr_loss, ctr_un + ctcvr_un

    @torch.no_grad()
    def predict(self, features):
        out = self.forward(features)
        ctr_logistic = torch.sigmoid(out["ctr_logits"])
        ctcvr_logistic = ctr_logistic * torch.sigmoid(out["cvr_logits"])
        two = torch.cat([torch.zeros_like(out["ctcvr_logits"]), out["ctcvr_logits"]], dim=-1)          # :83-86
        return {"probabilities": torch.softmax(two, dim=-1), "logistic": ctcvr_logistic,             # :88-91
                "class_ids": torch.argmax(two, dim=-1, keepdim=True), "ctr_logistic": ctr_logistic,
                "ctr_logits": out["ctr_logits"], "ctcvr_logits": out["ctcvr_logits"]}


class ESMM_W_D(ESMM):
    """The wide & deep ESMM (models/ESMM/ESMM_wide_deep.py:75-99): ctr_model and cvr_model are each dnn tower + linear model over their OWN
    variables; predictions, loss and metrics as ESMM (the reference's two files share them line for line).  linear_optimizer / dnn_optimizer
    are stored (the reference's defaults: Ftrl at min(0.2, 1 / sqrt(#linear columns)), Adagrad at 0.05 -- learning_rates())."""

    def __init__(self, model_dir=None, linear_feature_columns=None, dnn_feature_columns=None, linear_optimizer="Ftrl",
                 ctr_weight_column=None, ctcvr_weight_column=None, dnn_hidden_units=None, dnn_dropout=None, config=None,
                 dnn_optimizer="Adagrad", input_layer_partitioner=None, dnn_activation_fn=torch.relu):
        nn.Module.__init__(self)
        self.hparams = dict(model_dir=model_dir, ctr_weight_column=ctr_weight_column, ctcvr_weight_column=ctcvr_weight_column,
                            dnn_dropout=dnn_dropout, config=config, linear_optimizer=linear_optimizer, dnn_optimizer=dnn_optimizer,
                            input_layer_partitioner=input_layer_partitioner)
        lin, dnn, hidden = list(linear_feature_columns or []), list(dnn_feature_columns or []), list(dnn_hidden_units or [])
        self.n_linear_columns = len(lin)
        self.ctr_model = _BaseModelWD(lin, dnn, hidden, dnn_activation_fn, dnn_dropout)           # ESMM_wide_deep.py:103-114
        self.cvr_model = _BaseModelWD(lin, dnn, hidden, dnn_activation_fn, dnn_dropout)           # :115-126

    def learning_rates(self):
        """(linear, dnn) learning rates of the reference's default optimisers (ESMM_wide_deep.py:30,33,72-73)."""
        import math
        return (min(0.2, 1.0 / math.sqrt(self.n_linear_columns)) if self.n_linear_columns else 0.2), 0.05

    def fused_sparse_adagrad(self, lr, initial_accumulator_value=0.1):
        from . import ops
        a = self.ctr_model.dnn.input_layer.fused_sparse_adagrad(lr, initial_accumulator_value) if self.ctr_model.dnn is not None else []
        b = self.cvr_model.dnn.input_layer.fused_sparse_adagrad(lr, initial_accumulator_value) if self.cvr_model.dnn is not None else []
        for x, y in zip(a, b):
            ops.share_sorted_entries(x, y)
        return a + b
