"""xdeepfm.py -- xDeepFM forward (linear + CIN + DNN) on the HIP path.  No reference code exists
(/root/reference/README.md:28 links arXiv:1803.05170); the embedding / linear / DNN parts reuse the
DeepFM-style column handling of the reference (models/DeepFM/deepFM.py:169-177,255-319)."""
import math

import torch
from torch import nn

from .dense import dense_act, mlp_head, mlp_head_supported, mlp_stack, mlp_stack_supported, tower_infer, units1
from . import autograd as ag
from . import ops
from ._input import checked_forward as _checked_forward
from ._input import collect_ids, categorical_of
from .deepfm import _glorot_uniform_
import os as _os

CIN_STACK_NODE = _os.environ.get("DIR_CIN_STACK_NODE", "1") != "0"      # development switch: 0 keeps one autograd node per CIN layer


class XDeepFM(nn.Module):
    def __init__(self, linear_feature_columns=None, dnn_feature_columns=None, cin_layer_sizes=(128, 128, 128),
                 dnn_hidden_units=(400, 400), dnn_activation_fn=torch.relu):
        super().__init__()
        self.linear_feature_columns = list(linear_feature_columns or [])
        self.dnn_feature_columns = list(dnn_feature_columns or [])
        if not self.dnn_feature_columns:
            raise ValueError("empty columns.")
        self.m = len(self.dnn_feature_columns)
        self.D = self.dnn_feature_columns[0].dimension
        if any(c.dimension != self.D for c in self.dnn_feature_columns):
            raise ValueError("CIN needs one embedding dimension for every field")
        self.activation = dnn_activation_fn
        s = 1.0 / math.sqrt(self.D)
        self.embedding_weights = nn.ParameterList(
            [nn.Parameter(nn.init.trunc_normal_(torch.empty(c.num_buckets, self.D), std=s, a=-2 * s, b=2 * s))
             for c in self.dnn_feature_columns])
        self.linear_weights = nn.ParameterList(
            [nn.Parameter(torch.zeros(categorical_of(c).num_buckets)) for c in self.linear_feature_columns])
        self.linear_bias = nn.Parameter(torch.zeros(1))
        self.cin_layer_sizes = tuple(cin_layer_sizes)
        self.cin_W = nn.ParameterList()
        hp = self.m
        for h in self.cin_layer_sizes:
            self.cin_W.append(nn.Parameter(_glorot_uniform_(torch.empty(h, hp * self.m))))
            hp = h
        self.cin_out = nn.Linear(sum(self.cin_layer_sizes), 1)
        self.hidden = nn.ModuleList()
        d = self.m * self.D
        for n in dnn_hidden_units:
            lin = nn.Linear(d, n)
            _glorot_uniform_(lin.weight)
            nn.init.zeros_(lin.bias)
            self.hidden.append(lin)
            d = n
        self.dnn_out = nn.Linear(d, 1)
        self._ts_key = None

    def _tablesets(self):
        key = tuple([p.data_ptr() for p in ops.plain_list(self.embedding_weights)] + [p.data_ptr() for p in ops.plain_list(self.linear_weights)])
        if self._ts_key != key:
            ops.refuse_rebuild_under_sink(getattr(self, "_emb_ts", None), getattr(self, "_lin_ts", None))
            self._emb_ts = ops.TableSet([p.data for p in self.embedding_weights])
            self._emb_ts.owners = list(self.embedding_weights)          # HIP updates bump the parameters' version counters (ops.mark_written)
            self._lin_ts = ops.TableSet([p.data for p in self.linear_weights]) if len(self.linear_weights) else None
            if self._lin_ts is not None:
                self._lin_ts.owners = list(self.linear_weights)
            self._ts_key = key
        return self._emb_ts, self._lin_ts

    def fused_sparse_adagrad(self, lr, initial_accumulator_value=0.1):
        """Attach the fused HIP sparse Adagrad to the embedding tables (as DeepFM.fused_sparse_adagrad; the xDeepFM paper trains with
        the DeepFM recipe): backward() then updates them in place, duplicate ids summed first; the tables get no .grad."""
        emb_ts, _ = self._tablesets()
        self._sparse_adagrad = ops.SparseAdagrad(emb_ts, lr, initial_accumulator_value).attach()
        self._link_sparse_optimisers()
        return self._sparse_adagrad

    def fused_sparse_ftrl(self, lr=0.2, initial_accumulator_value=0.1, l1=0.0, l2=0.0):
        """Attach the fused HIP sparse FTRL to the linear weight columns (as DeepFM.fused_sparse_ftrl)."""
        _, lin_ts = self._tablesets()
        if lin_ts is None:
            raise ValueError("fused_sparse_ftrl: the model has no linear feature columns")
        self._sparse_ftrl = ops.SparseFtrl(lin_ts, lr, initial_accumulator_value, l1, l2).attach()
        self._link_sparse_optimisers()
        return self._sparse_ftrl

    def _same_categoricals(self):
        a = [categorical_of(c) for c in self.dnn_feature_columns]
        b = [categorical_of(c) for c in self.linear_feature_columns]
        return len(a) == len(b) and all(x is y for x, y in zip(a, b))

    def _link_sparse_optimisers(self):
        a, f = getattr(self, "_sparse_adagrad", None), getattr(self, "_sparse_ftrl", None)
        if a is not None and f is not None and self._same_categoricals():
            ops.share_sorted_entries(a, f)

    def cin(self, x0):
        """x0 [B, m, D] -> pooled features [B, sum(H_k)]."""
        B = x0.shape[0]
        if torch.is_grad_enabled() and (x0.requires_grad or any(W.requires_grad for W in self.cin_W)):
            if x0.is_cuda and x0.dtype == torch.float32 and CIN_STACK_NODE:
                return ag.cin_stack(x0.contiguous(), list(self.cin_W))      # the whole stack as one autograd node (autograd.CinStack)
            outs, xk = [], x0                  # per-layer nodes: autograd.CinLayer (dW on MFMA, dx via the forward)
            for W in self.cin_W:
                xk, p = ag.cin_layer(x0, xk, W)
                outs.append(p)
            return torch.cat(outs, dim=1)
        pooled = torch.empty((B, sum(self.cin_layer_sizes)), dtype=torch.float32, device=x0.device)
        xk, off = x0, 0
        last = len(self.cin_layer_sizes) - 1
        for k, (W, h) in enumerate(zip(self.cin_W, self.cin_layer_sizes)):
            # the last layer's [B,H,D] map feeds nothing: only its pooled sums are written
            xk, _ = ops.cin_layer(x0, xk, W.data, pooled=pooled[:, off:off + h], want_xout=k < last, w_owner=W)
            off += h
        return pooled

    def forward_embedded(self, emb, linear_logit=None, range_ok=None):
        """emb [B, m*D] (the embedding concatenation) -> logits.  range_ok: the fp16-range verdict of the tables the rows came from when those
        are not this module's own (the row-sharded tables: ops.f16_range_ok(ShardedTables.absmax()))."""
        B = emb.shape[0]
        logits = units1(self.cin_out, self.cin(emb.view(B, self.m, self.D)))      # (training: one pass for the layer's two gradients)
        return self._dnn_and_sum(emb, logits, linear_logit, self._tablesets()[0].range_ok() if range_ok is None else range_ok)

    @torch.no_grad()
    def forward_rows(self, rows, inv, linear_logit=None, absmax=None):
        """Inference logits [B, 1] from a row-sharded lookup WITHOUT its finish pass (round 6; shard.ShardedTables.lookup_rows): rows [n, D]
        as the exchange left them, inv [B, m] int64 = the position of (sample, field)'s row in `rows` (< 0: a zero row).  The CIN layers stage
        their x0 slices through `inv` (ops.cin_stack_gather) and the DNN tower looks its input rows up the same way inside its one launch
        (dense.tower_infer's gather form): the [B, m*D] concatenation is never written or read.  Bit for bit forward_embedded() on the
        materialised rows; where a kernel does not cover the shape the rows are materialised for it (the finish pass's work, here).
        absmax: the SHARDED tables' largest |value| (ShardedTables.absmax()) -- the tower's fp16-range verdict must not come from this
        module's own (unused) tables."""
        from .shard import rows_as_tables
        B = inv.shape[0]
        Hs = list(self.cin_layer_sizes)
        inv = inv if inv.is_contiguous() else inv.contiguous()
        emb = None
        if rows.is_cuda and B > 0 and ops.cin_gather_covers(self.m, self.D, Hs):
            pooled = torch.empty((B, sum(Hs)), dtype=torch.float32, device=rows.device)
            ops.cin_stack_gather(rows, inv, [W.data for W in self.cin_W], pooled, w_owners=list(self.cin_W))
        else:
            emb = ops.embedding_bag(rows_as_tables(rows, self.m), inv)
            pooled = self.cin(emb.view(B, self.m, self.D))
        logits = units1(self.cin_out, pooled)
        if self.dnn_out.out_features == 1 and emb is None:
            adds = (logits,) if linear_logit is None else (logits, linear_logit)
            fused = tower_infer(self.hidden, None, self.activation, head=self.dnn_out, adds=adds,
                                gather=(rows_as_tables(rows, self.m, absmax=absmax), inv, None, False))
            if fused is not None:
                return fused
        if emb is None:
            emb = ops.embedding_bag(rows_as_tables(rows, self.m), inv)
        return self._dnn_and_sum(emb, logits, linear_logit, True if absmax is None else ops.f16_range_ok(absmax))

    def _dnn_and_sum(self, emb, logits, linear_logit, range_ok):
        net = emb
        if self.dnn_out.out_features == 1:
            # inference: the DNN tower and its logit layer in one launch, the CIN's (and the linear term's) logits added in its epilogue
            adds = (logits,) if linear_logit is None else (logits, linear_logit)
            fused = tower_infer(self.hidden, net, self.activation, head=self.dnn_out, adds=adds, embedding_input=range_ok)   # net = the embedding concat
            if fused is not None:
                return fused
        if mlp_head_supported(self.hidden, self.dnn_out, net, self.activation):
            logits = logits + mlp_head(self.hidden, self.dnn_out, net, embedding_input=True)   # training: the tower + its logit layer as one autograd node
        else:
            if mlp_stack_supported(self.hidden, net, self.activation):
                net = mlp_stack(self.hidden, net, embedding_input=True)         # training: the whole tower as one autograd node
            else:
                for lin in self.hidden:
                    net = dense_act(lin, net, self.activation)                  # dir_dense_f32 when covered
            logits = logits + units1(self.dnn_out, net)
        if linear_logit is not None:
            logits = logits + linear_logit
        return logits

    @_checked_forward
    def forward(self, features):
        device = self.linear_bias.device
        emb_ts, lin_ts = self._tablesets()
        train = torch.is_grad_enabled()
        got = collect_ids(self.dnn_feature_columns, features, device)
        comb = self.dnn_feature_columns[0].combiner
        if train:      # sparse table gradients (autograd.EmbeddingBag / LinearLogit)
            tabs = ops.plain_list(self.embedding_weights)
            emb = ag.embedding_bag(emb_ts, got[1], tabs) if got[0] == "onehot" else \
                ag.embedding_bag(emb_ts, got[1], tabs, got[2], got[3], combiner=comb, field_major=True)
        elif got[0] == "onehot":
            emb = ops.embedding_bag(emb_ts, got[1])
        else:
            emb = ops.embedding_bag(emb_ts, got[1], got[2], got[3], combiner=comb, field_major=True)
        lin = None
        if lin_ts is not None:
            g2 = got if self._same_categoricals() else collect_ids(self.linear_feature_columns, features, device)
            if train and g2[0] == "onehot":
                lin = ag.linear_logit(lin_ts, g2[1], self.linear_bias, list(self.linear_weights))
            else:
                lin = ops.linear_logit(lin_ts, g2[1], bias=self.linear_bias.data) if g2[0] == "onehot" else \
                    ops.linear_logit(lin_ts, g2[1], g2[2], g2[3], bias=self.linear_bias.data, field_major=True)
        out = self.forward_embedded(emb, lin)
        from ._input import raise_pending
        raise_pending()
        return out
