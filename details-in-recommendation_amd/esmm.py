"""esmm.py -- ESMM (entire-space multi-task) forward on the HIP gather, keeping the reference constructor kwargs.

Mirrors models/ESMM/ESMM.py (reference):
  ESMM.__init__ kwargs                                     :23-32
  _model_fn: two towers under 'ctr_model' / 'cvr_model'    :62-78   (each tower creates its OWN embedding variables)
  _base_model: input_layer -> (dense(act, glorot_normal))* -> dense(1)       :130-147
  predictions: ctr/cvr sigmoid, ctcvr = product, ctcvr_logits = logit(clip(p, 1e-7, 1-1e-7))   :67-92
SURVEY.md 8(f) rank 3: the same gather kernel, used twice per step.
"""
import torch
from torch import nn

from .dense import dense_act, mlp_head, mlp_head_supported, mlp_stack, mlp_stack_supported, tower_infer, units1
from .dcn import _glorot_normal_
from .deepfm import _dropout_train, _glorot_uniform_
from .input_layer import InputLayer
from ._input import checked_forward as _checked_forward

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
        fused = tower_infer(self.hidden, net, self.activation, head=self.logits)          # inference: tower + logit layer in one launch
        if fused is not None:
            return fused
        if not self.dropout and mlp_stack_supported(self.hidden, net, self.activation):
            if mlp_head_supported(self.hidden, self.logits, net, self.activation):
                return mlp_head(self.hidden, self.logits, net)                           # training: tower + logit layer as one autograd node
            return units1(self.logits, mlp_stack(self.hidden, net))                      # training: the whole tower as one autograd node
        for lin in self.hidden:
            net = dense_act(lin, net, self.activation)                           # dir_dense_f32 when covered
            net = _dropout_train(self, net, self.dropout)                        # ESMM.py:143-144 (TRAIN only)
        return units1(self.logits, net)


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
        ctcvr_logistic = torch.sigmoid(ctr_logits) * torch.sigmoid(cvr_logits)  # :69-71
        p = ctcvr_logistic.clamp(_EPSILON, 1 - _EPSILON)                         # :73-74
        out = {"ctr_logits": ctr_logits, "ctcvr_logits": torch.log(p / (1 - p)), "cvr_logits": cvr_logits}
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
