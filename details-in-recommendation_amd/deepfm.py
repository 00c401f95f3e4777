"""deepfm.py -- DeepFM forward on the HIP path, keeping the reference constructor kwargs.

Mirrors models/DeepFM/deepFM.py (reference):
  DeepFM.__init__ kwargs                                  :55-73
  _DeepFM_model_fn: inputs -> dnn_fm logits + linear      :143-223
  myself_input_layer (one shared embedding set)           :363-400   -> ops.gather_fm / embedding_bag
  fm_logit_fn                                             :321-335   -> fused in ops.gather_fm
  dnn_logit_fn                                            :284-319   -> dense.dense_act (dir_dense_f32) + BN
  _linear_logit_fn_builder                                :255-275   -> ops.linear_logit
  head predictions (sigmoid / [1-p, p] / class_ids)       :107-117 + [TF-upstream] binary head
Training-only kwargs (optimizers, loss_reduction, warm_start_from, config, model_dir ...) are accepted
and stored; this module is the forward path.
"""
import math

import torch
from torch import nn

from .dense import dense_act, mlp_head, mlp_head_supported, mlp_stack, mlp_stack_supported, tower_infer, units1
from . import autograd as ag
from . import ops
from ._input import checked_forward as _checked_forward
from ._input import collect_ids, categorical_of, raise_pending


def _glorot_uniform_(w):  # [TF-upstream] glorot_uniform_initializer, deepFM.py:299
    fan_out, fan_in = w.shape
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return nn.init.uniform_(w, -lim, lim)


def _train_mode(module):
    """The reference's mode == TRAIN: the module is in train() state AND autograd is recording (inference calls run under
    torch.no_grad(), so an un-eval()'ed module still takes the inference forms)."""
    return module.training and torch.is_grad_enabled()


class _BatchNormInfer(nn.Module):
    """tf.layers.batch_normalization / contrib batch_norm (deepFM.py:303-308, DeepCrossNetwork.py:413-419): eps 1e-3, beta,
    optional gamma.  Inference: x * inv + (beta - mean * inv), inv = gamma * rsqrt(moving_variance + eps).  Training
    ([TF-upstream] training=True): the batch's mean and population variance normalise, and the moving statistics are
    updated as moving = moving * momentum + batch * (1 - momentum), momentum 0.999."""

    def __init__(self, n, eps=1e-3, scale=True, momentum=0.999):
        super().__init__()
        self.eps = eps
        self.momentum = momentum
        self.gamma = nn.Parameter(torch.ones(n)) if scale else None
        self.beta = nn.Parameter(torch.zeros(n))
        self.register_buffer("moving_mean", torch.zeros(n))
        self.register_buffer("moving_variance", torch.ones(n))

    def forward(self, x):
        if _train_mode(self):
            mean = x.mean(dim=0)
            var = x.var(dim=0, unbiased=False)
            with torch.no_grad():
                self.moving_mean.mul_(self.momentum).add_(mean.detach(), alpha=1.0 - self.momentum)
                self.moving_variance.mul_(self.momentum).add_(var.detach(), alpha=1.0 - self.momentum)
        else:
            mean, var = self.moving_mean, self.moving_variance
        inv = torch.rsqrt(var + self.eps)
        if self.gamma is not None:
            inv = inv * self.gamma
        return x * inv + (self.beta - mean * inv)


def _dropout_train(module, net, rate):
    """core_layers.dropout(net, rate, training=True) in TRAIN mode only (deepFM.py:301-302, DeepCrossNetwork.py:405-408)."""
    if rate and _train_mode(module):
        return torch.nn.functional.dropout(net, p=float(rate), training=True)
    return net


class DeepFM(nn.Module):
    def __init__(self, model_dir=None, linear_feature_columns=None, linear_optimizer="Ftrl",
                 linear_sparse_combiner="sum", dnn_feature_columns=None, dnn_optimizer="Adagrad",
                 dnn_hidden_units=None, dnn_activation_fn=torch.relu, dnn_dropout=None, fm_embedding_size=None,
                 n_classes=2, weight_column=None, label_vocabulary=None, input_layer_partitioner=None,
                 config=None, warm_start_from=None, loss_reduction="sum", batch_norm=False):
        super().__init__()
        linear_feature_columns = list(linear_feature_columns or [])
        dnn_feature_columns = list(dnn_feature_columns or [])
        if not (linear_feature_columns + dnn_feature_columns):
            raise ValueError("empty columns.")                                   # deepFM.py:104-105
        if n_classes is None or n_classes < 2:
            raise ValueError("n_classes must be >= 2")                            # [TF-upstream] head validation
        # deepFM.py:107-117: binary head (logits_dimension 1) or the multi-class softmax head (logits_dimension n_classes)
        self.n_classes = int(n_classes)
        self.units = 1 if n_classes == 2 else int(n_classes)
        for c in dnn_feature_columns:
            if not getattr(c, "is_dense", False):
                raise ValueError("Items of feature_columns must be a _DenseColumn. You can wrap a categorical "
                                 "column with an embedding_column or indicator_column. Given: {}".format(c))  # :371-373
            if fm_embedding_size is not None and c.dimension != fm_embedding_size:
                raise ValueError("every dnn column must have dimension fm_embedding_size (deepFM.py:329)")
        self.hparams = dict(model_dir=model_dir, linear_optimizer=linear_optimizer, dnn_optimizer=dnn_optimizer,
                            dnn_dropout=dnn_dropout, weight_column=weight_column, label_vocabulary=label_vocabulary,
                            input_layer_partitioner=input_layer_partitioner, config=config,
                            warm_start_from=warm_start_from, loss_reduction=loss_reduction)
        self.linear_feature_columns = linear_feature_columns
        self.dnn_feature_columns = dnn_feature_columns
        self.linear_sparse_combiner = linear_sparse_combiner
        self.activation = dnn_activation_fn
        self.K = fm_embedding_size if fm_embedding_size is not None else (
            dnn_feature_columns[0].dimension if dnn_feature_columns else 0)
        self.F = len(dnn_feature_columns)
        # embedding_weights per column ([TF-upstream] truncated normal, stddev 1/sqrt(K))
        self.embedding_weights = nn.ParameterList()
        for c in dnn_feature_columns:
            w = torch.empty(c.num_buckets, c.dimension)
            nn.init.trunc_normal_(w, std=1.0 / math.sqrt(c.dimension), a=-2.0 / math.sqrt(c.dimension),
                                  b=2.0 / math.sqrt(c.dimension))
            self.embedding_weights.append(nn.Parameter(w))
        # linear_model weights: zeros ([TF-upstream]; [vocab, units] -- kept as [vocab] for units = 1), bias zero
        self.linear_weights = nn.ParameterList(
            [nn.Parameter(torch.zeros(categorical_of(c).num_buckets) if self.units == 1 else
                          torch.zeros(categorical_of(c).num_buckets, self.units)) for c in linear_feature_columns])
        self.linear_bias = nn.Parameter(torch.zeros(self.units))
        # dnn (deepFM.py:292-317)
        self.hidden = nn.ModuleList()
        self.bns = nn.ModuleList()
        d = self.F * self.K
        if dnn_feature_columns:
            if dnn_hidden_units is None:
                raise ValueError("dnn_hidden_units must be given (deepFM.py:292 iterates it)")
            for n in dnn_hidden_units:
                lin = nn.Linear(d, n)
                _glorot_uniform_(lin.weight)
                nn.init.zeros_(lin.bias)
                self.hidden.append(lin)
                if batch_norm:
                    self.bns.append(_BatchNormInfer(n))
                d = n
            self.logits_layer = nn.Linear(d, self.units)                          # deepFM.py:311-317, units = head.logits_dimension
            _glorot_uniform_(self.logits_layer.weight)
            nn.init.zeros_(self.logits_layer.bias)
        self._emb_ts = None
        self._lin_ts = None

    # ---- TableSets follow the parameters' storage (rebuilt when .to()/.cuda() moved them) ------------
    def _tablesets(self):
        key = tuple([p.data_ptr() for p in ops.plain_list(self.embedding_weights)] + [p.data_ptr() for p in ops.plain_list(self.linear_weights)])
        if getattr(self, "_ts_key", None) != key:
            ops.refuse_rebuild_under_sink(self._emb_ts, self._lin_ts)
            self._emb_ts = ops.TableSet([p.data for p in self.embedding_weights]) if len(self.embedding_weights) else None
            self._lin_ts = ops.TableSet([p.data for p in self.linear_weights]) if len(self.linear_weights) else None
            if self._emb_ts is not None:
                self._emb_ts.owners = list(self.embedding_weights)      # HIP updates bump the parameters' version counters (ops.mark_written)
            if self._lin_ts is not None:
                self._lin_ts.owners = list(self.linear_weights)
            self._ts_key = key
        return self._emb_ts, self._lin_ts

    # ---- logit builders ---------------------------------------------------------------------------
    def _logits_of(self, net):
        return units1(self.logits_layer, net) if self.units == 1 else self.logits_layer(net)

    def dnn_logit_fn(self, net, adds=(), range_ok=None):
        """deepFM.py:284-319.  adds: [B, 1] logits to add to the result (fm_logit_fn's, deepFM.py:337-338) -- inside the fused tower
        kernel's epilogue when that runs, here otherwise.  range_ok: whether the rows `net` was gathered from sit inside the unscaled
        fp16 x 2 window (ops.f16_range_ok of THEIR tables' largest magnitude); None = net came from this model's own embedding_weights
        (the model's feature path).  A caller that feeds rows of other tables (ShardedDeepFMTrainer.predict: the sharded tables) passes
        its own verdict -- the guard used to measure the model's unused tables instead; False is always safe (bf16 x 3)."""
        if not _train_mode(self):
            if range_ok is None:
                range_ok = self._tablesets()[0].range_ok()
            fused = tower_infer(self.hidden, net, self.activation, bns=self.bns if len(self.bns) else None,
                                head=self.logits_layer if self.units == 1 else None, adds=adds if self.units == 1 else (),
                                embedding_input=bool(range_ok))       # net is the concat of the embedding columns (deepFM.py:288-291)
            if fused is not None:                                               # inference: the whole tower (+ logit layer) in one launch
                out = fused if self.units == 1 else self._logits_of(fused)
                for a in (adds if self.units != 1 else ()):
                    out = a + out
                return out
        out = self._dnn_logit_layers(net)
        for a in adds:
            out = a + out
        return out

    def _dnn_logit_layers(self, net):
        if not len(self.bns) and not self.hparams.get("dnn_dropout") and mlp_stack_supported(self.hidden, net, self.activation):
            if self.units == 1 and mlp_head_supported(self.hidden, self.logits_layer, net, self.activation):
                return mlp_head(self.hidden, self.logits_layer, net, embedding_input=True)      # training: tower + logit layer as one autograd node
            return self._logits_of(mlp_stack(self.hidden, net, embedding_input=True))           # training: the whole tower as one autograd node
        for i, lin in enumerate(self.hidden):                                   # deepFM.py:292-308
            if len(self.bns) and not _train_mode(self):                        # inference: no dropout, BN folded into the layer's epilogue
                net = dense_act(lin, net, self.activation, bn=self.bns[i])
                continue
            if len(self.bns) and not self.hparams.get("dnn_dropout"):          # training without dropout: layer + batch norm as one autograd node
                net = dense_act(lin, net, self.activation, bn=self.bns[i])
                continue
            net = dense_act(lin, net, self.activation)                          # dir_dense_f32 when covered
            net = _dropout_train(self, net, self.hparams.get("dnn_dropout"))    # :301-302 (TRAIN only), before the BN
            if len(self.bns):
                net = self.bns[i](net)
        return self._logits_of(net)                                              # :311-317

    def dnn_fm_logit_fn(self, features, device, got=None):
        emb_ts, _ = self._tablesets()
        if got is None:
            got = collect_ids(self.dnn_feature_columns, features, device)
        train = torch.is_grad_enabled()                                          # autograd path: HIP forward + HIP/sparse backward
        max_norm = self._max_norm()
        if got[0] == "onehot" and max_norm:                                      # clipping needs the bag kernel: one-entry bags
            ids = got[1]
            emb = (ag.embedding_bag(emb_ts, ids, ops.plain_list(self.embedding_weights), max_norm=max_norm) if train
                   else ops.embedding_bag(emb_ts, ids, max_norm=max_norm))
            fm = ag.fm_logit(emb, self.F, self.K) if train else ops.fm_logit(emb, self.F, self.K)
            return self.dnn_logit_fn(emb, adds=(fm,))
        if got[0] == "onehot":
            if train:
                emb, fm = ag.gather_fm(emb_ts, got[1], ops.plain_list(self.embedding_weights))
            else:
                emb, fm = ops.gather_fm(emb_ts, got[1])                          # inputs + fm_logit_fn, one pass
        else:
            _, vals, offs, wts, _ = got
            comb = [c.combiner for c in self.dnn_feature_columns]                # every embedding_column carries its own combiner
            if train:
                emb = ag.embedding_bag(emb_ts, vals, ops.plain_list(self.embedding_weights), offs, wts, combiner=comb, field_major=True,
                                       max_norm=max_norm)
                fm = ag.fm_logit(emb, self.F, self.K)
            else:
                emb = ops.embedding_bag(emb_ts, vals, offs, wts, combiner=comb, field_major=True, max_norm=max_norm)
                fm = ops.fm_logit(emb, self.F, self.K)
        return self.dnn_logit_fn(emb, adds=(fm,))                                # deepFM.py:337-338 ([B,1] + [B,units] broadcasts)

    def _same_categoricals(self):
        same = getattr(self, "_same_cats", None)
        if same is None:
            a = [categorical_of(c) for c in self.dnn_feature_columns]
            b = [categorical_of(c) for c in self.linear_feature_columns]
            same = self._same_cats = len(a) == len(b) and all(x is y for x, y in zip(a, b))
        return same

    def _max_norm(self):
        mn = [getattr(c, "max_norm", None) for c in self.dnn_feature_columns]
        if not mn or all(m == mn[0] for m in mn):
            return mn[0] if mn else None
        return mn                                            # one max_norm per column: the bag kernel takes a per-slot array

    def linear_logit_fn(self, features, device, got=None):
        """_linear_logit_fn_builder (deepFM.py:255-275): linear_model(units, sparse_combiner) + bias -> [B, units]."""
        _, lin_ts = self._tablesets()
        if got is None:
            got = collect_ids(self.linear_feature_columns, features, device)
        train = torch.is_grad_enabled()
        if self.units == 1 and got[0] == "onehot":
            if train:
                return ag.linear_logit(lin_ts, got[1], self.linear_bias, ops.plain_list(self.linear_weights))
            return ops.linear_logit(lin_ts, got[1], bias=self.linear_bias.data)
        if self.units == 1 and not train:
            _, vals, offs, wts, _ = got
            return ops.linear_logit(lin_ts, vals, offs, wts, combiner=self.linear_sparse_combiner,
                                    bias=self.linear_bias.data, field_major=True)
        # multi-hot under autograd, or units > 1 (multi-class head): the first-order term is a bag lookup with K = units and the
        # linear_model's sparse_combiner, summed over the columns; the bag op's backward gives the weights sparse gradients
        Fl = len(self.linear_feature_columns)
        comb = self.linear_sparse_combiner
        tabs = ops.plain_list(self.linear_weights)
        if got[0] == "onehot":
            args, kw = (got[1],), {}
            B = got[1].shape[0]
        else:
            _, vals, offs, wts, B = got
            args, kw = (vals,), dict(offsets=offs, weights=wts, field_major=True)
        if train:
            per = ag.embedding_bag(lin_ts, args[0], tabs, combiner=comb, **kw)
        else:
            per = ops.embedding_bag(lin_ts, args[0], combiner=comb, **kw)
        return per.view(B, Fl, self.units).sum(dim=1) + self.linear_bias

    @_checked_forward
    def forward(self, features):
        if not isinstance(features, dict):
            raise ValueError("features should be a dictionary of `Tensor`s. Given type: {}".format(type(features)))  # :159-161
        device = self.linear_bias.device
        logits = None
        got = None
        if self.dnn_feature_columns:
            got = collect_ids(self.dnn_feature_columns, features, device)
            pk = self._serving_pack() if got[0] == "onehot" else None
            if pk is not None:
                # inference, the DeepFM case (the linear columns ARE the dnn columns' categoricals, deepFM.py:89-95): concat, FM term and
                # first-order term from ONE packed 128-byte row per (sample, field), then the tower with both logits added in its epilogue
                logits = self._packed_logits(got[1], pk)
                raise_pending()
                return logits
            logits = self.dnn_fm_logit_fn(features, device, got)
        if self.linear_feature_columns:
            # the DeepFM case (deepFM.py:89-95): the linear columns ARE the dnn columns' categorical columns -- one id matrix (and one
            # range check) serves both terms, and their sorted sparse updates see the same tensor
            lin = self.linear_logit_fn(features, device, got if (got is not None and self._same_categoricals()) else None)
            logits = lin if logits is None else logits + lin                     # add_n, deepFM.py:223
        raise_pending()                                                          # the id-range verdicts (checked on the device, read here)
        return logits

    def create_loss(self, features, logits, labels):
        """The canned binary head's loss (deepFM.py:107-117: weight_column, loss_reduction -- default SUM, deepFM.py:72)
        -> (weighted_loss, unweighted_loss)."""
        from .train_spec import _weights_of, weighted_sigmoid_cross_entropy, weighted_softmax_cross_entropy
        w = _weights_of(features, self.hparams["weight_column"], logits) if isinstance(features, dict) else None
        red = str(self.hparams["loss_reduction"] or "sum").lower()
        if self.units > 1:                       # _multi_class_head_with_softmax_cross_entropy_loss (deepFM.py:112-117)
            return weighted_softmax_cross_entropy(logits, labels, w, red)
        return weighted_sigmoid_cross_entropy(logits, labels, w, red)

    def fused_sparse_adagrad(self, lr, initial_accumulator_value=0.1, packed=False):
        """Attach the fused HIP sparse-Adagrad update to the embedding tables (the reference trains them with
        dnn_optimizer='Adagrad', deepFM.py:61): backward() then updates them in place, with duplicate ids summed first, and
        the FM term's backward is folded into the update.  packed=True moves the tables into the packed training layout
        (ops.TableSet.train_rows: [embedding | accumulator] rows, one memory line per row and its optimiser state at K = 16);
        the embedding parameters become [vocab, K] views of it -- same values, same checkpoints, one-hot columns only."""
        emb_ts, _ = self._tablesets()
        if packed:
            if emb_ts.ld == emb_ts.K:
                emb_ts = ops.TableSet.train_rows([p.data for p in self.embedding_weights], initial_accumulator_value)
                for p, view in zip(self.embedding_weights, emb_ts.tables):
                    p.data = view
                emb_ts.owners = list(self.embedding_weights)
                self._emb_ts = emb_ts
                self._ts_key = tuple(p.data_ptr() for p in self.embedding_weights) + tuple(p.data_ptr() for p in self.linear_weights)
        self._sparse_adagrad = ops.SparseAdagrad(emb_ts, lr, initial_accumulator_value).attach()
        self._link_sparse_optimisers()
        return self._sparse_adagrad

    def fused_sparse_ftrl(self, lr=0.2, initial_accumulator_value=0.1, l1=0.0, l2=0.0, packed=False):
        """Attach the fused HIP sparse FTRL update to the linear weight columns (linear_optimizer='Ftrl', deepFM.py:58):
        backward() then updates them in place; the bias keeps a dense gradient for a dense optimiser.  packed=True (units = 1) moves
        the columns into the packed linear training layout (ops.TableSet.ftrl_rows: [w | n | z | -] rows, a touched id's weight and FTRL
        state in ONE 16-byte row); the weight parameters become [vocab, 1] views of it -- same values, same checkpoints, one-hot
        columns only."""
        _, lin_ts = self._tablesets()
        if lin_ts is None:
            raise ValueError("fused_sparse_ftrl: the model has no linear feature columns")
        if packed and lin_ts.ld == lin_ts.K:
            if self.units != 1:
                raise ValueError("fused_sparse_ftrl(packed=True): the packed linear rows hold one weight per id (units = 1)")
            lin_ts = ops.TableSet.ftrl_rows([p.data for p in self.linear_weights], initial_accumulator_value)
            for p, view in zip(self.linear_weights, lin_ts.tables):
                p.data = view.view(p.data.shape)                 # ([vocab] or [vocab, 1]: the parameter keeps its shape, row stride 4)
            lin_ts.owners = list(self.linear_weights)
            self._lin_ts = lin_ts
            self._ts_key = tuple(p.data_ptr() for p in self.embedding_weights) + tuple(p.data_ptr() for p in self.linear_weights)
        self._sparse_ftrl = ops.SparseFtrl(lin_ts, lr, initial_accumulator_value, l1, l2).attach()
        self._link_sparse_optimisers()
        return self._sparse_ftrl

    def _link_sparse_optimisers(self):
        """Adagrad on the embedding tables and FTRL on the linear columns of the SAME categorical columns see the same ids: one sort."""
        a, f = getattr(self, "_sparse_adagrad", None), getattr(self, "_sparse_ftrl", None)
        if a is not None and f is not None and self._same_categoricals():
            ops.share_sorted_entries(a, f)

    AUTO_PACK_MAX_BYTES = 32 << 30      # the packed copy of tables beyond this size is built on request only (pack_for_serving)

    def _pack_signature(self):
        return tuple((p._version, p.data_ptr()) for p in self.embedding_weights) + tuple((p._version, p.data_ptr()) for p in self.linear_weights)

    def _serving_pack(self):
        """The packed serving layout when it applies to this forward (inference, binary head, one-hot columns, the linear columns equal
        to the dnn columns' categoricals, no max_norm), built on first use and rebuilt when a parameter has changed since
        (tensor._version / storage): the default inference layout.  None: the reference layout's kernels run."""
        if torch.is_grad_enabled() or self.units != 1 or not self.hparams.get("serving_layout", "auto") or not self.linear_feature_columns:
            return None                                   # (hparams["serving_layout"] = None / DIR_SERVING_LAYOUT=reference: never pack)
        import os
        if os.environ.get("DIR_SERVING_LAYOUT") == "reference":
            return None
        if not self._same_categoricals() or any(getattr(c, "max_norm", None) for c in self.dnn_feature_columns):
            return None
        if not self.embedding_weights[0].is_cuda:
            return None
        sig = self._pack_signature()
        if getattr(self, "_packed", None) is not None and getattr(self, "_packed_sig", None) == sig and self._packed_extra is None:
            return self._packed
        ld = 32
        while ld < self.K + 1:
            ld *= 2
        if sum(int(p.shape[0]) for p in self.embedding_weights) * ld * 4 > self.AUTO_PACK_MAX_BYTES:
            return None
        self.pack_for_serving()
        return self._packed

    def pack_for_serving(self):
        """Inference-only: copy the embedding tables and the first-order weights of the same categorical columns into
        the packed 128-byte-row layout (ops.PackedTables), so that forward_ids() reads one memory line per
        (sample, field) for all three sparse terms.  Needs linear_feature_columns to start with the dnn columns'
        categorical columns in the same order (the DeepFM case: deepFM.py:89-95).  Re-pack after changing weights."""
        n = self.F
        lin_names = [categorical_of(c).name for c in self.linear_feature_columns[:n]]
        if len(lin_names) < n or lin_names != [categorical_of(c).name for c in self.dnn_feature_columns]:
            raise ValueError("pack_for_serving: the first %d linear columns must be the dnn columns' categorical columns" % n)
        self._packed = ops.PackedTables([p.data for p in self.embedding_weights], [p.data for p in self.linear_weights[:n]])
        self._packed_extra = ops.TableSet([p.data for p in self.linear_weights[n:]]) if len(self.linear_weights) > n else None
        self._packed_sig = self._pack_signature()
        return self._packed

    def _packed_logits(self, ids, pk):
        """Inference from the packed serving rows `pk`: DNN + FM + first-order logits of the columns the packed layout holds -- in one launch
        (the lookups inside the tower kernel, dir_deepfm_tower_bf16x3_f32) where that kernel covers the model, else the packed gather
        followed by dnn_logit_fn; the two are bit-identical."""
        if self.units == 1 and not _train_mode(self):
            logits = tower_infer(self.hidden, None, self.activation, bns=self.bns if len(self.bns) else None, head=self.logits_layer,
                                 gather=(pk, ids, self.linear_bias.data))
            if logits is not None:
                return logits
        emb, fm, lin = ops.gather_fm_linear(pk, ids, bias=self.linear_bias.data)
        return self.dnn_logit_fn(emb, adds=(fm, lin))

    def forward_ids(self, dnn_ids, linear_ids=None):
        """Fast path for pre-assembled one-hot id matrices [B, F] (any strides).  linear_ids: the ids of ALL linear
        columns [B, F_lin] (its first F columns equal dnn_ids in the DeepFM case)."""
        if linear_ids is not None and not torch.is_grad_enabled() and linear_ids.shape[1] == self.F:
            self._serving_pack()                                                 # the default inference layout (built / refreshed here)
        if (getattr(self, "_packed", None) is not None and linear_ids is not None and not torch.is_grad_enabled()
                and getattr(self, "_packed_sig", None) == self._pack_signature()):
            logits = self._packed_logits(dnn_ids, self._packed)
            if self._packed_extra is not None:
                logits = logits + ops.linear_logit(self._packed_extra, linear_ids[:, self.F:])
            return logits
        emb_ts, lin_ts = self._tablesets()
        train = torch.is_grad_enabled()
        if train:                                                                # differentiable: the tables get (sparse) gradients
            emb, fm = ag.gather_fm(emb_ts, dnn_ids, ops.plain_list(self.embedding_weights))
        else:
            emb, fm = ops.gather_fm(emb_ts, dnn_ids)
        logits = self.dnn_logit_fn(emb, adds=(fm,))
        if linear_ids is not None and lin_ts is not None:
            if self.units != 1:
                raise NotImplementedError("forward_ids: the multi-class head takes the features-dict path")
            if train:
                logits = logits + ag.linear_logit(lin_ts, linear_ids, self.linear_bias, ops.plain_list(self.linear_weights))
            else:
                logits = logits + ops.linear_logit(lin_ts, linear_ids, bias=self.linear_bias.data)
        return logits

    @torch.no_grad()
    def predict(self, features):
        """Head predictions: binary ([TF-upstream] _binary_logistic_head_with_sigmoid_cross_entropy_loss: logits, logistic,
        probabilities = softmax([0, logit]), class_ids) or, for n_classes > 2, the multi-class softmax head
        (_multi_class_head_with_softmax_cross_entropy_loss: logits, probabilities = softmax(logits), class_ids = argmax)."""
        logits = self.forward(features)
        if self.units > 1:
            return {"logits": logits, "probabilities": torch.softmax(logits, dim=-1),
                    "class_ids": torch.argmax(logits, dim=-1, keepdim=True)}
        logistic = torch.sigmoid(logits)
        two = torch.cat([torch.zeros_like(logits), logits], dim=-1)
        return {"logits": logits, "logistic": logistic, "probabilities": torch.softmax(two, dim=-1),
                "class_ids": torch.argmax(two, dim=-1, keepdim=True)}

    def tf_variable_names(self):
        """Reference checkpoint names of the parameters (deepFM.py:173-196,206-209 scopes)."""
        names = {}
        for c, _ in zip(self.dnn_feature_columns, self.embedding_weights):
            names["embedding_weights.%d" % len(names)] = "dnn_fm_inputs/myself_input_layer/%s/embedding_weights" % c.name
        for i in range(len(self.hidden)):
            names["hidden.%d.weight" % i] = "dnn_fm/hiddenlayer_%d/kernel" % i
            names["hidden.%d.bias" % i] = "dnn_fm/hiddenlayer_%d/bias" % i
        names["logits_layer.weight"] = "dnn_fm/logits/kernel"
        names["logits_layer.bias"] = "dnn_fm/logits/bias"
        for i, c in enumerate(self.linear_feature_columns):
            names["linear_weights.%d" % i] = "linear/linear_model/%s/weights" % categorical_of(c).name
        names["linear_bias"] = "linear/linear_model/bias_weights"
        return names
