"""din.py -- Deep Interest Network: the local activation unit + pooling (DINAttentionPool) and the whole model (DIN).

NO REFERENCE CODE: /root/reference/README.md:27 lists DIN as a model and links arXiv:1706.06978; nothing under models/ implements it.
Everything here restates the paper (and, for the unit's 80-40-1 sigmoid MLP on [h, a, h - a, h * a], the paper's public code) and is
labelled paper-derived wherever it is tested (oracle/np_ref.py: din_attention_pool, din_model_logits).

  user behaviours  hist [B, T] item ids (+ lengths)  --+
  candidate ad     cand [B]    item id  ---------------+--> DINAttentionPool (csrc/din_wave.hip) --> interest vector [B, K]
  profile / context columns --> InputLayer (csrc/embedding_bag.hip) --------------------------------------+
  concat([columns, interest vector, candidate embedding]) --> 200-80 MLP (PReLU / Dice / ReLU) --> logit [B, 1]   (paper figure 2)

The constructor follows the other models' keyword style (model_dir, feature_columns, dnn_hidden_units, dnn_activation_fn, n_classes...);
forward(features) -> logits [B, 1]; predict(features) -> the binary head's dict as DeepFM / DeepCrossNetwork return it.
"""
import math

import torch
from torch import nn

from . import autograd as ag
from . import ops
from .dense import dense_act, units1

ACTIVATIONS = ("sigmoid", "prelu", "dice")
import os as _os
ROWS_TRAIN = _os.environ.get("DIR_DIN_ROWS_TRAIN", "1") != "0"      # development switch: 0 = PReLU / Dice train through the torch formulation (round 4)


class Dice(nn.Module):
    """Data-adaptive activation (arXiv:1706.06978 eq. 3):  f(s) = p(s) s + (1 - p(s)) alpha s,  p(s) = sigmoid((s - E[s]) / sqrt(Var[s] + eps)).
    TRAIN mode: E / Var of the mini-batch (over every leading dimension), moving statistics updated with `momentum`; otherwise the
    moving statistics.  alpha is learned per unit (initialised 0, as PReLU's 0.25 is PReLU's own default -- the paper gives none)."""

    def __init__(self, n, eps=1e-8, momentum=0.99):
        super().__init__()
        self.alpha = nn.Parameter(torch.zeros(n))
        self.register_buffer("moving_mean", torch.zeros(n))
        self.register_buffer("moving_variance", torch.ones(n))
        self.eps, self.momentum = eps, momentum
        self._folded = None

    def scale_shift(self):
        """(scale, shift) with p = sigmoid(scale * s + shift): the inference form the HIP unit takes.  Folded once per version
        of the moving statistics (four library kernels per call otherwise: 8 % of the cfg-4 model's forward)."""
        key = (self.moving_mean._version, self.moving_variance._version, self.moving_mean.data_ptr(),
               self.moving_variance.data_ptr(), self.eps)
        if self._folded is None or self._folded[0] != key:
            with torch.no_grad():
                scale = torch.rsqrt(self.moving_variance + self.eps)
                self._folded = (key, scale, -self.moving_mean * scale)
        return self._folded[1], self._folded[2]

    def forward(self, s, valid=None):
        """valid (optional, broadcastable to s without its last dimension): rows that take part in the batch statistics."""
        if self.training and torch.is_grad_enabled():
            flat = s.reshape(-1, s.shape[-1])
            if valid is not None:
                flat = flat[valid.reshape(-1)]
            if flat.shape[0] > 0:
                mean, var = flat.mean(dim=0), flat.var(dim=0, unbiased=False)
                with torch.no_grad():
                    self.moving_mean.mul_(self.momentum).add_(mean.detach(), alpha=1 - self.momentum)
                    self.moving_variance.mul_(self.momentum).add_(var.detach(), alpha=1 - self.momentum)
            else:
                mean, var = self.moving_mean, self.moving_variance
            p = torch.sigmoid((s - mean) * torch.rsqrt(var + self.eps))
        else:
            scale, shift = self.scale_shift()
            p = torch.sigmoid(s * scale + shift)
        return p * s + (1 - p) * self.alpha * s


class _PReLU(nn.Module):
    def __init__(self, n, init=0.25):
        super().__init__()
        self.alpha = nn.Parameter(torch.full((n,), float(init)))

    def forward(self, s, valid=None):
        return torch.where(s > 0, s, self.alpha * s)


def _make_act(kind, n):
    if kind == "prelu":
        return _PReLU(n)
    if kind == "dice":
        return Dice(n)
    return None


class DINAttentionPool(nn.Module):
    """history ids [B,T] + lengths [B] + candidate ids [B] -> pooled interest vector [B,K].  activation: the two hidden layers'
    activation -- "sigmoid" (default), "prelu" or "dice" (arXiv:1706.06978 section 5.3).  Inference runs the HIP unit for all three
    (dir_din_attention_pool[_act]_f32); training runs the fused HIP forward / backward for the sigmoid unit and, for PReLU / Dice, the unit
    over the compact row list on HIP kernels (_rows_train: Dice normalises with the mini-batch's statistics, so its layers are whole-batch
    passes); the differentiable torch formulation (_composite) remains for shapes those kernels do not take."""

    def __init__(self, vocab_size, embedding_dim=64, hidden_units=(80, 40), normalize=False, activation="sigmoid"):
        super().__init__()
        if activation not in ACTIVATIONS:
            raise ValueError("activation must be one of %s" % (ACTIVATIONS,))
        K, (H1, H2) = embedding_dim, hidden_units
        s = 1.0 / math.sqrt(K)
        self.table = nn.Parameter(nn.init.trunc_normal_(torch.empty(vocab_size, K), std=s, a=-2 * s, b=2 * s))
        self.W1 = nn.Parameter(nn.init.xavier_uniform_(torch.empty(4 * K, H1)))
        self.b1 = nn.Parameter(torch.zeros(H1))
        self.W2 = nn.Parameter(nn.init.xavier_uniform_(torch.empty(H1, H2)))
        self.b2 = nn.Parameter(torch.zeros(H2))
        self.W3 = nn.Parameter(nn.init.xavier_uniform_(torch.empty(H2, 1)).reshape(H2))
        self.b3 = nn.Parameter(torch.zeros(1))
        self.normalize = normalize
        self.activation = activation
        self.act1, self.act2 = _make_act(activation, H1), _make_act(activation, H2)
        self._act_cache = None

    def act_params(self):
        """[3 H1 + 3 H2] for the HIP unit (alpha, scale, shift per layer), from the activation modules' inference form."""
        if self.activation == "sigmoid":
            return None
        src = [self.act1.alpha, self.act2.alpha]
        if self.activation == "dice":
            src += [self.act1.moving_mean, self.act1.moving_variance, self.act2.moving_mean, self.act2.moving_variance]
        key = tuple((t._version, t.data_ptr()) for t in src)
        if self._act_cache is None or self._act_cache[0] != key:           # rebuilt when a parameter / statistic is written
            if self.activation == "prelu":
                vec = ops.din_act_params(self.b1.numel(), self.b2.numel(), self.act1.alpha, self.act2.alpha)
            else:
                s1, t1 = self.act1.scale_shift()
                s2, t2 = self.act2.scale_shift()
                vec = ops.din_act_params(self.b1.numel(), self.b2.numel(), self.act1.alpha, self.act2.alpha, s1, t1, s2, t2)
            self._act_cache = (key, vec)
        return self._act_cache[1]

    def _composite(self, hist, hist_len, cand):
        """The unit as differentiable torch ops (PReLU / Dice in TRAIN mode): same definition as include/dir_hip.h A13."""
        B, T = hist.shape
        K = self.table.shape[1]
        valid = hist >= 0
        if hist_len is not None:
            valid = valid & (torch.arange(T, device=hist.device).unsqueeze(0) < hist_len.clamp(0, T).unsqueeze(1))
        h = torch.nn.functional.embedding(hist.clamp_min(0), self.table, sparse=True)                      # [B, T, K]
        a = torch.nn.functional.embedding(cand.clamp_min(0), self.table, sparse=True) * (cand >= 0).unsqueeze(1)
        ae = a.unsqueeze(1).expand(B, T, K)
        u = torch.cat([h, ae, h - ae, h * ae], dim=-1)
        z1 = u @ self.W1 + self.b1
        z1 = torch.sigmoid(z1) if self.act1 is None else self.act1(z1, valid)
        z2 = z1 @ self.W2 + self.b2
        z2 = torch.sigmoid(z2) if self.act2 is None else self.act2(z2, valid)
        s = z2 @ self.W3 + self.b3
        if self.normalize:
            s = (s / math.sqrt(K)).masked_fill(~valid, float("-inf"))
            w = torch.softmax(s, dim=1)
            w = torch.where(valid.any(dim=1, keepdim=True), w, torch.zeros_like(w)).masked_fill(~valid, 0.0)
        else:
            w = s * valid
        return (w.unsqueeze(-1) * h).sum(dim=1)

    def _rows_train(self, hist, hist_len, cand):
        """PReLU / Dice in training on HIP kernels (round 5; csrc/din_rows_train.hip): the unit over the compact list of valid (sample, position)
        rows -- X' = [h | h * a | a] (dir_din_feat_rows_f32), pre1 = X' [Wh + Wd; Wp; Wa - Wd] + b1 and pre2 = y1 W2 + b2 on the dense
        kernels (forward, dL/dx and dL/dW: dense._DenseFn), the activations with the batch's statistics (autograd.ActRows: Dice =
        dir_bn_train_stats_f32 + dir_act_rows_train_f32, backward dir_act_rows_backward_f32 + dir_bn_train_backward_f32), the score layer
        (dir_units1_f32) and the per-sample weights + pooling (dir_din_pool_rows_f32).  Same definition as _composite (the torch
        formulation, kept for shapes these kernels do not take); the table gets a sparse gradient.  -> [B, K], or None: not covered."""
        from .dense import _DenseFn, _Units1Fn
        B, T = hist.shape
        K, H1, H2 = self.table.shape[1], self.b1.numel(), self.b2.numel()
        if (not self.table.is_cuda or self.table.dtype != torch.float32 or K % 4 or K > 256 or H1 % 4 or H2 % 4 or H1 < 16 or H2 < 16
                or H1 > 1024 or H2 > 1024 or B == 0):
            return None
        valid = hist >= 0
        if hist_len is not None:
            valid = valid & (torch.arange(T, device=hist.device).unsqueeze(0) < hist_len.clamp(0, T).unsqueeze(1))
        b_idx, j_idx = valid.nonzero(as_tuple=True)                  # rows in (b, j) order (the one host read of the step)
        if b_idx.numel() == 0:
            return None
        ids_h = hist[b_idx, j_idx]
        cnt = valid.sum(dim=1)
        row_off = (torch.cumsum(cnt, 0) - cnt).contiguous()
        X, Hc = ag.DinFeatRows.apply(self.table, ids_h, b_idx, row_off, cand)
        Wh, Wa, Wd, Wp = self.W1[:K], self.W1[K:2 * K], self.W1[2 * K:3 * K], self.W1[3 * K:]
        Wc = torch.cat([Wh + Wd, Wp, Wa - Wd], dim=0)               # [3K, H1]: autograd hands dL/dW1 back through this regrouping
        pre1 = _DenseFn.apply(X, Wc.t(), self.b1, False)
        y1 = ag.act_rows(pre1, self.act1) if ops.act_rows_supported(pre1) else self.act1(pre1)
        pre2 = _DenseFn.apply(y1, self.W2.t(), self.b2, False)
        y2 = ag.act_rows(pre2, self.act2) if ops.act_rows_supported(pre2) else self.act2(pre2)
        sc = _Units1Fn.apply(y2, self.W3.reshape(1, -1), self.b3)   # [N, 1]
        return ag.DinPoolRows.apply(sc.reshape(-1), Hc, row_off, B, self.normalize)

    def forward(self, hist, hist_len, cand, want_scores=False):
        train = torch.is_grad_enabled() and not want_scores and any(p.requires_grad for p in self.parameters())
        if train and self.activation == "sigmoid":
            return ag.din_attention_pool(self.table, hist, hist_len, cand, self.W1, self.b1, self.W2, self.b2, self.W3,
                                         self.b3, normalize=self.normalize)       # sparse table gradient
        if train:
            out = self._rows_train(hist, hist_len, cand) if ROWS_TRAIN else None
            return out if out is not None else self._composite(hist, hist_len, cand)
        return ops.din_attention_pool(self.table.data, hist, hist_len, cand, self.W1.data, self.b1.data, self.W2.data,
                                      self.b2.data, self.W3.data, self.b3.data, normalize=self.normalize,
                                      want_scores=want_scores, activation=self.activation, act_params=self.act_params(),
                                      range_of=(self.table, self.W1, self.W2, self.W3, self.b1, self.b2))


class DIN(nn.Module):
    """Deep Interest Network (arXiv:1706.06978, figure 2 right; /root/reference/README.md:27 -- no reference code).

    feature_columns : the user-profile / context columns (dense columns of feature_column.py: embedding / numeric / indicator),
                      looked up by InputLayer (name-sorted concat, as the reference's input_layer); may be empty / None.
    item_vocab_size, embedding_dim : the goods table shared by the behaviour sequence and the candidate ad (BASELINE config 4:
                      10 000 000 x 64).
    history_key, history_len_key, candidate_key : features[...] = hist [B, T] int64 (id < 0: padding), lengths [B] int32 (or absent:
                      every position counts), cand [B] int64.
    attention_hidden_units (80, 40), attention_activation, attention_normalize : the local activation unit (DINAttentionPool); the
                      paper keeps the un-normalised weights (attention_normalize=False).
    dnn_hidden_units (200, 80), dnn_activation_fn : "prelu" | "dice" | "relu" | "sigmoid" (the paper trains PReLU and Dice variants).
    n_classes : 2 (the binary head; logits [B, 1])."""

    def __init__(self, model_dir=None, feature_columns=None, item_vocab_size=None, embedding_dim=64, history_key="hist",
                 history_len_key="hist_len", candidate_key="cand", attention_hidden_units=(80, 40), attention_activation="sigmoid",
                 attention_normalize=False, dnn_hidden_units=(200, 80), dnn_activation_fn="dice", n_classes=2, weight_column=None,
                 optimizer="Adagrad", config=None):
        super().__init__()
        if item_vocab_size is None or item_vocab_size < 1:
            raise ValueError("item_vocab_size must be given.")
        if n_classes != 2:
            raise ValueError("DIN: the binary head only (n_classes=2).")
        if dnn_activation_fn not in ("prelu", "dice", "relu", "sigmoid"):
            raise ValueError("dnn_activation_fn must be one of prelu, dice, relu, sigmoid")
        if not dnn_hidden_units:
            raise ValueError("dnn_hidden_units must be given.")
        self.hparams = dict(model_dir=model_dir, weight_column=weight_column, optimizer=optimizer, config=config, n_classes=n_classes)
        self.keys = (history_key, history_len_key, candidate_key)
        self.input_layer = None
        width = 0
        if feature_columns:
            from .input_layer import InputLayer
            self.input_layer = InputLayer(feature_columns)
            width = self.input_layer.column_num
        self.attention = DINAttentionPool(item_vocab_size, embedding_dim, tuple(attention_hidden_units), normalize=attention_normalize,
                                          activation=attention_activation)
        self.K = embedding_dim
        self.dnn_activation_fn = dnn_activation_fn
        d = width + 2 * embedding_dim
        self.input_width = d
        self.hidden = nn.ModuleList()
        self.acts = nn.ModuleList()
        for n in dnn_hidden_units:
            lin = nn.Linear(d, n)
            nn.init.xavier_uniform_(lin.weight)
            nn.init.zeros_(lin.bias)
            self.hidden.append(lin)
            self.acts.append(_make_act(dnn_activation_fn, n) or nn.Identity())
            d = n
        self.logits_layer = nn.Linear(d, 1)
        nn.init.xavier_uniform_(self.logits_layer.weight)
        nn.init.zeros_(self.logits_layer.bias)
        self._cand_ts = None

    def _candidate_embedding(self, cand):
        """The candidate ad's own embedding row (a one-slot lookup of the goods table: csrc/embedding_bag.hip)."""
        table = self.attention.table
        if torch.is_grad_enabled() and table.requires_grad:
            return torch.nn.functional.embedding(cand.clamp_min(0), table, sparse=True) * (cand >= 0).unsqueeze(1)
        if self._cand_ts is None or self._cand_ts.tables[0].data_ptr() != table.data_ptr():
            self._cand_ts = ops.TableSet([table.data])
        return ops.embedding_bag(self._cand_ts, cand.reshape(-1, 1).contiguous())

    def forward(self, features):
        hk, lk, ck = self.keys
        hist, cand = features[hk], features[ck]
        hist_len = features.get(lk) if hasattr(features, "get") else None
        dev = self.attention.table.device
        hist, cand = hist.to(device=dev, dtype=torch.int64), cand.to(device=dev, dtype=torch.int64).reshape(-1)
        if hist_len is not None:
            hist_len = hist_len.to(device=dev, dtype=torch.int32).reshape(-1)
        pooled = self.attention(hist, hist_len, cand)
        pieces = []
        if self.input_layer is not None:
            pieces.append(self.input_layer(features)[:, :self.input_layer.column_num])
        pieces += [pooled, self._candidate_embedding(cand)]
        net = torch.cat(pieces, dim=1)
        for lin, act in zip(self.hidden, self.acts):
            if self.dnn_activation_fn == "relu":
                net = dense_act(lin, net, torch.relu)                      # dir_dense_f32 / dir_dense_bf16x3_f32 with the ReLU in the epilogue
            elif self.dnn_activation_fn == "sigmoid":
                net = torch.sigmoid(dense_act(lin, net, None))
            else:
                pre = dense_act(lin, net, None)
                if torch.is_grad_enabled() and ROWS_TRAIN and ops.act_rows_supported(pre):
                    net = ag.act_rows(pre, act)                            # TRAIN mode on HIP kernels (batch statistics: autograd.ActRows)
                elif torch.is_grad_enabled() or not pre.is_cuda or pre.shape[1] % 4 or pre.stride(0) % 4 or pre.data_ptr() % 16:
                    net = act(pre)                                         # uncovered shapes: torch ops
                elif self.dnn_activation_fn == "prelu":
                    net = ops.din_activation_rows_(pre, "prelu", act.alpha)              # HIP layer, then ONE in-place pass
                else:
                    sc, sh = act.scale_shift()
                    net = ops.din_activation_rows_(pre, "dice", act.alpha, sc, sh)
        return units1(self.logits_layer, net)

    def predict(self, features):
        """The binary head's predictions, with the keys DeepFM.predict / DeepCrossNetwork.predict return."""
        logits = self.forward(features)
        two = torch.cat([torch.zeros_like(logits), logits], dim=-1)
        return {"logits": logits, "logistic": torch.sigmoid(logits), "probabilities": torch.softmax(two, dim=-1),
                "class_ids": torch.argmax(two, dim=-1, keepdim=True)}
