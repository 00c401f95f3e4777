"""dcn.py -- Deep & Cross Network forward on the HIP path, keeping the reference constructor kwargs.

Mirrors models/DeepCrossNetwork/DeepCrossNetwork.py (reference):
  DeepCrossNetwork.__init__ kwargs                         :35-49
  _dcn_logit_fn_builder / dcn_logits_fn                    :118-141
  input_layer (columns concatenated SORTED BY NAME [TF-upstream])   :126   -> ops.embedding_bag into x0
  _cross_architecture / _cross_op                          :336-367      -> ops.cross_network (one launch)
  _deep_architecture (dense(act) -> BN on all but last)    :370-410      -> torch linear (rocBLAS)
  concat + dense(1)                                        :134-139
  predictions                                              :153-165
"""
import math

import torch
from torch import nn

from . import ops
from ._input import collect_ids, categorical_of
from .deepfm import _BatchNormInfer, _glorot_uniform_
from .feature_column import EmbeddingColumn, IndicatorColumn, NumericColumn, Ragged


def _glorot_normal_(w):  # glorot_normal_initializer, DeepCrossNetwork.py:397 ([TF-upstream] truncated normal)
    fan_out, fan_in = w.shape
    std = math.sqrt(2.0 / (fan_in + fan_out)) / 0.87962566103423978
    return nn.init.trunc_normal_(w, std=std, a=-2 * std, b=2 * std)


class DeepCrossNetwork(nn.Module):
    def __init__(self, model_dir=None, columns=None, cross_layer_num=2, dnn_hidden_units=None, dnn_dropout=None,
                 config=None, dnn_activation_fn=torch.relu, weight_column=None, optimizer=None, optimizer_spec=None,
                 batch_norm=True, l2_reg=None, learning_rate_spec=None):
        super().__init__()
        columns = list(columns or [])
        if not columns:
            raise ValueError("empty columns.")
        for c in columns:
            if not getattr(c, "is_dense", False):
                raise ValueError("Items of feature_columns must be a _DenseColumn. Given: {}".format(c))
        self.hparams = dict(model_dir=model_dir, dnn_dropout=dnn_dropout, config=config, weight_column=weight_column,
                            optimizer=optimizer, optimizer_spec=optimizer_spec, l2_reg=l2_reg,
                            learning_rate_spec=learning_rate_spec)
        # [TF-upstream] input_layer: sorted(feature_columns, key=lambda c: c.name)
        self.columns = sorted(columns, key=lambda c: c.name)
        self.cross_layer_num = cross_layer_num
        self.activation = dnn_activation_fn
        self.offsets = []
        d = 0
        for c in self.columns:
            self.offsets.append(d)
            d += c.dimension
        self.column_num = d                                                      # DeepCrossNetwork.py:127-128
        # embedding columns grouped by (dimension, combiner): one TableSet / one launch per group
        self.emb_cols = [c for c in self.columns if isinstance(c, EmbeddingColumn)]
        self.embedding_weights = nn.ParameterList()
        for c in self.emb_cols:
            w = torch.empty(c.num_buckets, c.dimension)
            s = 1.0 / math.sqrt(c.dimension)
            nn.init.trunc_normal_(w, std=s, a=-2 * s, b=2 * s)
            self.embedding_weights.append(nn.Parameter(w))
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
        self._ts_key = None

    def _tablesets(self):
        key = tuple(p.data_ptr() for p in self.embedding_weights)
        if self._ts_key != key:
            self._groups = []  # (TableSet, [column indices into self.emb_cols], combiner)
            seen = {}
            for i, c in enumerate(self.emb_cols):
                seen.setdefault((c.dimension, c.combiner), []).append(i)
            for (dim, comb), idxs in seen.items():
                # only runs of columns that are ADJACENT in the sorted concat can share one launch
                run = []
                for i in idxs:
                    if run and self._col_offset(self.emb_cols[i]) != self._col_offset(self.emb_cols[run[-1]]) + dim:
                        self._groups.append((ops.TableSet([self.embedding_weights[j].data for j in run]), run, comb))
                        run = []
                    run.append(i)
                self._groups.append((ops.TableSet([self.embedding_weights[j].data for j in run]), run, comb))
            self._ts_key = key
        return self._groups

    def _col_offset(self, col):
        return self.offsets[self.columns.index(col)]

    def input_layer(self, features):
        device = self.cross_w.device
        B = None
        x0 = None
        for c in self.columns:  # dense parts first (they tell B)
            if isinstance(c, NumericColumn):
                v = features[c.key].to(device=device, dtype=torch.float32).reshape(-1, c.dimension)
                if x0 is None:
                    B = v.shape[0]
                    x0 = torch.empty((B, self.column_num), dtype=torch.float32, device=device)
                x0[:, self._col_offset(c):self._col_offset(c) + c.dimension] = v
        for ts, idxs, comb in self._tablesets():
            cols = [self.emb_cols[i] for i in idxs]
            got = collect_ids(cols, features, device)
            nb = got[1].shape[0] if got[0] == "onehot" else got[4]
            if x0 is None:
                B = nb
                x0 = torch.empty((B, self.column_num), dtype=torch.float32, device=device)
            off = self._col_offset(cols[0])
            view = x0[:, off:off + len(cols) * cols[0].dimension]
            if got[0] == "onehot":
                ops.embedding_bag(ts, got[1], out=view)
            else:
                ops.embedding_bag(ts, got[1], got[2], got[3], combiner=comb, field_major=True, out=view)
        for c in self.columns:
            if isinstance(c, IndicatorColumn):  # multi-hot counts ([TF-upstream] indicator_column)
                ids = categorical_of(c).ids(features, device)
                if x0 is None:
                    B = ids.numel() if not isinstance(ids, tuple) else ids[1].numel() - 1
                    x0 = torch.empty((B, self.column_num), dtype=torch.float32, device=device)
                ind = torch.zeros((B, c.dimension), dtype=torch.float32, device=device)
                if isinstance(ids, tuple):
                    vals, offs, _ = ids
                    rows = torch.repeat_interleave(torch.arange(B, device=device), offs[1:] - offs[:-1])
                    ok = vals >= 0
                    ind.index_put_((rows[ok], vals[ok]), torch.ones(int(ok.sum()), device=device), accumulate=True)
                else:
                    ok = ids >= 0
                    ind[torch.arange(B, device=device)[ok], ids[ok]] = 1.0
                x0[:, self._col_offset(c):self._col_offset(c) + c.dimension] = ind
        return x0

    def cross_architecture(self, x0):
        return ops.cross_network(x0, self.cross_w.data, self.cross_b.data)       # :350-367

    def deep_architecture(self, net):
        n = len(self.hidden)
        bi = 0
        for i, lin in enumerate(self.hidden):                                    # :392-403
            net = self.activation(lin(net))
            if self.batch_norm and i < n - 1:
                net = self.bns[bi](net)
                bi += 1
        return net

    def forward(self, features):
        x0 = features if isinstance(features, torch.Tensor) else self.input_layer(features)
        cross = self.cross_architecture(x0)
        deep = self.deep_architecture(x0)
        return self.logits_layer(torch.cat([cross, deep], dim=-1))               # :136-137

    @torch.no_grad()
    def predict(self, features):
        logits = self.forward(features)                                          # :153-165
        two = torch.cat([torch.zeros_like(logits), logits], dim=-1)
        return {"logits": logits, "logistic": torch.sigmoid(logits), "probabilities": torch.softmax(two, dim=-1),
                "class_ids": torch.argmax(two, dim=-1, keepdim=True)}
