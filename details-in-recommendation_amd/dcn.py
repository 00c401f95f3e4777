"""dcn.py -- Deep & Cross Network forward on the HIP path, keeping the reference constructor kwargs.

Mirrors models/DeepCrossNetwork/DeepCrossNetwork.py (reference):
  DeepCrossNetwork.__init__ kwargs                         :35-49
  _dcn_logit_fn_builder / dcn_logits_fn                    :118-141
  input_layer (columns concatenated SORTED BY NAME [TF-upstream])   :126   -> ops.embedding_bag into x0
  _cross_architecture / _cross_op                          :336-367      -> ops.cross_network (one launch)
  _deep_architecture (dense(act) -> BN on all but last)    :370-410      -> dense.dense_act (dir_dense_f32)
  concat + dense(1)                                        :134-139
  predictions                                              :153-165
"""
import math

import torch
from torch import nn

from .dense import dense_act, _RELUS, _MlpHeadFn, _Units1Fn, mlp_stack_supported, MLP_HEAD
from . import autograd as ag
from . import ops
from .deepfm import _BatchNormInfer, _dropout_train, _glorot_uniform_
from .input_layer import InputLayer
from ._input import checked_forward as _checked_forward
from ._input import raise_pending


def _glorot_normal_(w):  # glorot_normal_initializer, DeepCrossNetwork.py:397 ([TF-upstream] truncated normal)
    fan_out, fan_in = w.shape
    std = math.sqrt(2.0 / (fan_in + fan_out)) / 0.87962566103423978
    return nn.init.trunc_normal_(w, std=std, a=-2 * std, b=2 * std)


class DeepCrossNetwork(nn.Module):
    def __init__(self, model_dir=None, columns=None, cross_layer_num=2, dnn_hidden_units=None, dnn_dropout=None,
                 config=None, dnn_activation_fn=torch.relu, weight_column=None, optimizer=None, optimizer_spec=None,
                 batch_norm=True, l2_reg=None, learning_rate_spec=None):
        super().__init__()
        self.hparams = dict(model_dir=model_dir, dnn_dropout=dnn_dropout, config=config, weight_column=weight_column,
                            optimizer=optimizer, optimizer_spec=optimizer_spec, l2_reg=l2_reg,
                            learning_rate_spec=learning_rate_spec)
        self.input_layer = InputLayer(columns)                                   # DeepCrossNetwork.py:126
        self.columns = self.input_layer.columns                                  # name-sorted
        self.offsets = self.input_layer.offsets
        self.embedding_weights = self.input_layer.embedding_weights
        self.cross_layer_num = cross_layer_num
        self.activation = dnn_activation_fn
        d = self.input_layer.column_num
        self.column_num = d                                                      # DeepCrossNetwork.py:127-128
        # cross variables: truncated_normal(0, 0.1), DeepCrossNetwork.py:329-332
        self.cross_w = nn.Parameter(nn.init.trunc_normal_(torch.empty(cross_layer_num, d), std=0.1, a=-0.2, b=0.2))
        self.cross_b = nn.Parameter(nn.init.trunc_normal_(torch.empty(cross_layer_num, d), std=0.1, a=-0.2, b=0.2))
        # deep part
        dnn_hidden_units = list(dnn_hidden_units or [])
        self.hidden = nn.ModuleList()
        self.bns = nn.ModuleList()
        h = d
        for i, n in enumerate(dnn_hidden_units):
            lin = nn.Linear(h, n)
            _glorot_normal_(lin.weight)
            nn.init.zeros_(lin.bias)
            self.hidden.append(lin)
            if batch_norm and i < len(dnn_hidden_units) - 1:                     # :401
                self.bns.append(_BatchNormInfer(n, eps=1e-3, scale=False))       # contrib batch_norm: no gamma
            h = n
        self.batch_norm = batch_norm
        self.logits_layer = nn.Linear(d + (h if dnn_hidden_units else d), 1)    # :136-137
        _glorot_uniform_(self.logits_layer.weight)
        nn.init.zeros_(self.logits_layer.bias)

    def cross_architecture(self, x0):
        if torch.is_grad_enabled():
            return ag.cross_network(x0, self.cross_w, self.cross_b)
        return ops.cross_network(x0, self.cross_w.data, self.cross_b.data)       # :350-367

    def deep_architecture(self, net):
        n = len(self.hidden)
        bi = 0
        for i, lin in enumerate(self.hidden):                                    # :392-403
            bn = None
            if self.batch_norm and i < n - 1:
                bn = self.bns[bi]
                bi += 1
            net = dense_act(lin, net, self.activation, bn=bn)                    # dir_dense_f32 when covered; inference BN in its epilogue
            net = _dropout_train(self, net, self.hparams.get("dnn_dropout"))    # :405-408 (TRAIN only), after the BN
        return net

    def _deep_logit(self, net, wd):
        """Inference: _deep_architecture with the deep branch's share of the final dense(1), deep . w_d, folded into the last layer's
        epilogue (ops.dense_head: the [B, h] activation is never written) -> [B, 1], or None when that kernel does not cover the layer."""
        n = len(self.hidden)
        if not n or not (self.activation is None or self.activation in _RELUS) or torch.is_grad_enabled():
            return None
        bi = 0
        for i, lin in enumerate(self.hidden[:-1]):
            bn = None
            if self.batch_norm:
                bn = self.bns[bi]
                bi += 1
            net = dense_act(lin, net, self.activation, bn=bn)
        last = self.hidden[-1]
        if net.shape[1] != last.in_features:
            return None
        # the last layer reads the batch-normalised activation of the layer below (:400-403): bounded whatever the raw numeric columns
        # carried, so the fp16 x 2 form may run; without batch norm (or with one hidden layer: the input layer itself) it is bf16 x 3
        return ops.dense_head(net, last.weight, last.bias, wd, relu=self.activation is not None, bounded=bool(self.batch_norm) and n >= 2)

    def _train_logits(self, x0, cross):
        """Training: the final dense(1) over concat([cross, deep]) (:136-137) as cross . w_c + deep . w_d + bias WITHOUT the concat, the
        deep share as one autograd node with the last hidden layer (dense._MlpHeadFn: its backward takes dL/dlogit through the logit
        weights and the layer's ReLU in one pass over the activation, dir_units1_relu_backward_f32), the cross share with the
        elementwise backward (dense._Units1Fn).  As library ops the concat, the [B, d + h] x [d + h, 1] product and its backward were
        0.9 ms of the 8.8 ms step at 65 536 x (429 + 1024).  -> [B, 1], or None when the last layer is not covered (then the plain form)."""
        n = len(self.hidden)
        if (not MLP_HEAD or not n or self.cross_layer_num <= 0 or self.hparams.get("dnn_dropout") or not self.training
                or self.logits_layer.bias is None or self.hidden[-1].out_features > 4096):
            return None
        d = self.column_num
        net, bi = x0, 0
        for i, lin in enumerate(self.hidden[:-1]):                               # :392-403, as deep_architecture
            bn = None
            if self.batch_norm:
                bn = self.bns[bi]
                bi += 1
            net = dense_act(lin, net, self.activation, bn=bn)
        last = self.hidden[-1]
        if not mlp_stack_supported([last], net, self.activation):
            return None
        wl = self.logits_layer.weight                                            # [1, d + h]
        from . import dense as _dense
        _dense._FWD_BOUNDED[0] = bool(self.batch_norm) and n >= 2              # the last layer reads a batch-normalised activation: fp16 x 2 kernels
        try:
            deep_logit = _MlpHeadFn.apply(net, wl[:, d:], self.logits_layer.bias, last.weight, last.bias)
        finally:
            _dense._FWD_BOUNDED[0] = False
        return deep_logit + _Units1Fn.apply(cross, wl[:, :d], None)

    def _padded_cross_params(self):
        """cross_w / cross_b zero-padded to a multiple of 4 columns, cached until the parameters change."""
        key = (self.cross_w._version, self.cross_b._version, self.cross_w.data_ptr())
        if getattr(self, "_cross_pad_key", None) != key:
            pad = ops.pad4(self.column_num) - self.column_num
            self._cross_pad = (torch.nn.functional.pad(self.cross_w.data, (0, pad)), torch.nn.functional.pad(self.cross_b.data, (0, pad)))
            self._cross_pad_key = key
        return self._cross_pad

    def _forward_padded(self, features):
        """Inference with an input width that is not a multiple of 4 (429 = 26 x 16 + 13): the input layer writes x0 with row
        stride pad4(d) and zero pad columns ONCE; the cross kernel runs its 16-byte instantiation on it (zero-padded w, b keep
        the pad columns exactly 0), the first deep layer reads the same buffer with Kd = pad4(d) against a zero-padded weight,
        and the final dense(1) over concat([cross, deep]) (:136-137) is evaluated as cross . w_c + deep . w_d + bias -- the
        concat is never materialised, and cross . w_c comes out of the cross kernel itself (the cross output has no other reader).  Values: the same sums over the same real columns."""
        d, dp = self.column_num, ops.pad4(self.column_num)
        x0p = self.input_layer(features, pad_to=4)
        wp, bp = self._padded_cross_params()
        wl = self.logits_layer.weight                                            # [1, d + h]
        key = (wl._version, wl.data_ptr())
        if getattr(self, "_logit_split_key", None) != key:                       # the two halves of the final dense(1)'s weight, aligned copies, once per version
            self._logit_split = (torch.nn.functional.pad(wl.data[:, :d], (0, dp - d)), wl.data[:, d:].clone())
            self._logit_split_key = key
        wc, wd = self._logit_split
        cross_logit = ops.cross_network_head(x0p, wp, bp, wc)                    # [B, 1] = x_L . w_c in the cross kernel's epilogue: x_L is not written
        out = cross_logit.add_(self.logits_layer.bias)
        deep_logit = self._deep_logit(x0p, wd)                                   # the last deep layer with deep . w_d in its epilogue, when covered
        if deep_logit is not None:
            out.add_(deep_logit)
        else:
            deep = self.deep_architecture(x0p)                                   # dense_act pads the first weight (in_features d -> dp)
            if deep.is_cuda and deep.dtype == torch.float32 and deep.stride(1) == 1:
                out.add_(ops.units1(deep, wd))                                   # dir_units1_f32 (the library runs this one-column product as a GEMM: 27 us at 256 x 1024)
            else:
                out.addmm_(deep, wd.t())
        raise_pending()
        return out

    @_checked_forward
    def forward(self, features):
        if (not torch.is_grad_enabled() and not isinstance(features, torch.Tensor) and self.column_num % 4
                and len(self.hidden) and self.cross_layer_num > 0):
            return self._forward_padded(features)
        x0 = features if isinstance(features, torch.Tensor) else self.input_layer(features)
        cross = self.cross_architecture(x0)
        out = self._train_logits(x0, cross) if torch.is_grad_enabled() else None
        if out is None:
            deep = self.deep_architecture(x0)
            out = self.logits_layer(torch.cat([cross, deep], dim=-1))            # :136-137
        raise_pending()                                                          # id-range verdicts of the input layer's columns
        return out

    def create_loss(self, features, logits, labels):
        """_create_loss (DeepCrossNetwork.py:209-225): sigmoid cross entropy, weight_column weights, MEAN reduction
        -> (weighted_loss, unweighted_loss)."""
        from .train_spec import _weights_of, weighted_sigmoid_cross_entropy
        w = _weights_of(features, self.hparams["weight_column"], logits) if isinstance(features, dict) else None
        return weighted_sigmoid_cross_entropy(logits, labels, w, "mean")

    def train_step(self):
        """The reference's train_op for this model (_get_train_op_fn, DeepCrossNetwork.py:264-290) built from the
        constructor's optimizer / optimizer_spec / learning_rate_spec / l2_reg: see train_spec.TrainStep."""
        from .train_spec import TrainStep
        hp = self.hparams
        return TrainStep(self, optimizer=hp["optimizer"] or "Adam", optimizer_spec=hp["optimizer_spec"],
                         learning_rate_spec=hp["learning_rate_spec"], l2_reg=hp["l2_reg"],
                         l2_params=[lin.weight for lin in self.hidden])      # l2 on the deep kernels, :386-399

    @torch.no_grad()
    def predict(self, features):
        logits = self.forward(features)                                          # :153-165
        two = torch.cat([torch.zeros_like(logits), logits], dim=-1)
        return {"logits": logits, "logistic": torch.sigmoid(logits), "probabilities": torch.softmax(two, dim=-1),
                "class_ids": torch.argmax(two, dim=-1, keepdim=True)}
