"""input_layer.py -- the dense input layer of the reference's Estimators: [TF-upstream]
tf.feature_column.input_layer(features, feature_columns) as called at
models/DeepCrossNetwork/DeepCrossNetwork.py:126 and models/ESMM/ESMM.py:135.

Columns are concatenated SORTED BY NAME ([TF-upstream] input_layer sorts by column.name).  Embedding columns are
looked up by the HIP embedding-bag kernel straight into their slice of the output (row stride = total width, so
no concat pass); runs of name-adjacent embedding columns with equal (dimension, combiner) share one launch.
Numeric columns are copied, indicator columns are multi-hot counts (torch index ops: plumbing).
"""
import math

import torch
from torch import nn

from . import autograd as ag
from . import ops
from ._input import collect_ids, categorical_of, raise_pending
from .feature_column import EmbeddingColumn, IndicatorColumn, NumericColumn


class InputLayer(nn.Module):
    def __init__(self, columns):
        super().__init__()
        columns = list(columns or [])
        if not columns:
            raise ValueError("empty columns.")
        for c in columns:
            if not getattr(c, "is_dense", False):
                raise ValueError("Items of feature_columns must be a _DenseColumn. Given: {}".format(c))
        self.columns = sorted(columns, key=lambda c: c.name)
        self.offsets = []
        d = 0
        for c in self.columns:
            self.offsets.append(d)
            d += c.dimension
        self.column_num = d
        self.emb_cols = [c for c in self.columns if isinstance(c, EmbeddingColumn)]
        # every column an embedding column: the output is a concatenation of (combined) embedding rows -- the bounded magnitudes the
        # fp16 x 2 dense kernels want (dense.mlp_head(embedding_input=...)); numeric columns carry raw values (capital_gain: 99 999)
        self.embedding_only = len(self.emb_cols) == len(self.columns)
        self.embedding_weights = nn.ParameterList()
        for c in self.emb_cols:
            w = torch.empty(c.num_buckets, c.dimension)
            s = 1.0 / math.sqrt(c.dimension)          # [TF-upstream] embedding_column default initializer
            nn.init.trunc_normal_(w, std=s, a=-2 * s, b=2 * s)
            self.embedding_weights.append(nn.Parameter(w))
        self._ts_key = None

    def embedding_range_ok(self):
        """Whether the layer's output (embedding columns only) may feed the UNSCALED fp16 x 2 tower kernel: every table inside
        ops.TableSet.range_ok's window (weighted / mean-combined bags stay inside it; sum-combined bags of many rows are the caller's to bound)."""
        if not self.embedding_only:
            return False
        self._tablesets()
        return all(g[0].range_ok() for g in self._groups)

    def _col_offset(self, col):
        return self.offsets[self.columns.index(col)]

    def _tablesets(self):
        key = tuple([p.data_ptr() for p in ops.plain_list(self.embedding_weights)])
        if self._ts_key != key:
            ops.refuse_rebuild_under_sink(*[g[0] for g in getattr(self, "_groups", [])])
            self._groups = []  # (TableSet, [indices into self.emb_cols], [combiner per column], max_norm)
            seen = {}
            for i, c in enumerate(self.emb_cols):      # one launch takes one row width and one max_norm; combiners go per slot
                seen.setdefault((c.dimension, getattr(c, "max_norm", None)), []).append(i)

            def close(run, mn):
                ts = ops.TableSet([self.embedding_weights[j].data for j in run])
                ts.owners = [self.embedding_weights[j] for j in run]       # HIP updates bump the parameters' version counters
                self._groups.append((ts, run, [self.emb_cols[j].combiner for j in run], mn))

            for (dim, mn), idxs in seen.items():
                run = []
                for i in idxs:  # only runs ADJACENT in the sorted concat can share one launch
                    if run and self._col_offset(self.emb_cols[i]) != self._col_offset(self.emb_cols[run[-1]]) + dim:
                        close(run, mn)
                        run = []
                    run.append(i)
                close(run, mn)
            self._ts_key = key
        return self._groups

    def fused_sparse_adagrad(self, lr, initial_accumulator_value=0.1):
        """Attach the fused sparse Adagrad (include/dir_hip.h: dir_sparse_adagrad_sorted_f32; the reference's Adagrad on
        `embedding_weights`) to every group of embedding columns: backward() then updates the tables in place for one-hot
        inputs and they get no .grad.  Returns the optimiser objects (they own the accumulators)."""
        return [ops.SparseAdagrad(ts, lr, initial_accumulator_value=initial_accumulator_value).attach() for ts, _, _, _ in self._tablesets()]

    def fused_sparse_adam(self, beta1=0.9, beta2=0.999, eps=1e-8, clip_norm=0.0):
        """Attach the HIP tf.train.AdamOptimizer update (include/dir_hip.h: dir_sparse_adam_f32; the reference DCN's train_op on
        `embedding_weights`, DeepCrossNetwork.py:264-290) to every group of embedding columns it covers (no max_norm, K / 4 a power of
        two): backward() then steps ALL rows of those tables in place for one-hot inputs and they get no .grad.  -> (optimiser objects,
        the parameters they own); set .lr_t on each before every backward (train_spec.TrainStep does)."""
        opts, owned = [], []
        for ts, idxs, _, mn in self._tablesets():
            if mn or not ts.device.type == "cuda" or not ops.SparseAdam.covers(ts):
                continue
            opts.append(ops.SparseAdam(ts, beta1, beta2, eps, clip_norm).attach())
            owned += [self.embedding_weights[i] for i in idxs]
            opts[-1].owned = [self.embedding_weights[i] for i in idxs]          # parameter of table f of that optimiser's TableSet
        return opts, owned

    def _indicator(self, c, features, device, B):
        ids = categorical_of(c).ids(features, device)
        if B is None:
            B = ids.numel() if not isinstance(ids, tuple) else ids[1].numel() - 1
        ind = torch.zeros((B, c.dimension), dtype=torch.float32, device=device)
        if isinstance(ids, tuple):
            vals, offs, _ = ids
            rows = torch.repeat_interleave(torch.arange(B, device=device), offs[1:] - offs[:-1])
            ok = vals >= 0
            ind.index_put_((rows[ok], vals[ok]), torch.ones(int(ok.sum()), device=device), accumulate=True)
        else:
            ok = ids >= 0
            ind[torch.arange(B, device=device)[ok], ids[ok]] = 1.0
        return ind

    def _forward_train(self, features, device, memo=None):
        """Differentiable path: every column's block is its own tensor, concatenated in name order (autograd tracks
        the concat; the embedding blocks carry sparse table gradients, see autograd.EmbeddingBag)."""
        blocks, inside = {}, set()
        for ts, idxs, comb, mn in self._tablesets():
            cols = [self.emb_cols[i] for i in idxs]
            ew = ops.plain_list(self.embedding_weights)
            tabs = [ew[i] for i in idxs]
            got = collect_ids(cols, features, device, memo)
            if got[0] == "onehot":
                blk = ag.embedding_bag(ts, got[1], tabs, max_norm=mn)
            else:
                blk = ag.embedding_bag(ts, got[1], tabs, got[2], got[3], combiner=comb, field_major=True, max_norm=mn)
            # a group is a run of columns that are ADJACENT in the concat (see _tablesets): its block enters the concat whole, at
            # its first column's position (26 per-column slices would each cost a [B, 26*dim] zero fill + add in the backward)
            blocks[cols[0].name] = blk
            inside.update(c.name for c in cols[1:])
        pieces = []
        B = next(iter(blocks.values())).shape[0] if blocks else None
        for c in self.columns:
            if isinstance(c, NumericColumn):
                pieces.append(features[c.key].to(device=device, dtype=torch.float32).reshape(-1, c.dimension))
            elif isinstance(c, IndicatorColumn):
                pieces.append(self._indicator(c, features, device, B))
            elif c.name not in inside:
                pieces.append(blocks[c.name])
        return pieces[0] if len(pieces) == 1 else torch.cat(pieces, dim=1)

    def onehot_source(self, features, memo=None):
        """-> (TableSet, ids [B, F]) when this layer's output is exactly the concatenation of one-hot embedding lookups that ONE launch
        reads (every column an embedding column of one width, no max_norm, adjacent in the sorted concat) and the features carry one id
        per sample and column; else None.  What dense.tower_infer(..., gather=...) needs to do the lookups inside the tower kernel."""
        if torch.is_grad_enabled() or len(self.emb_cols) != len(self.columns) or not len(self.embedding_weights):
            return None
        groups = self._tablesets()
        if len(groups) != 1 or groups[0][3] or len(groups[0][1]) != len(self.emb_cols) or self._col_offset(self.emb_cols[groups[0][1][0]]) != 0:
            return None
        ts, idxs, _, _ = groups[0]
        got = collect_ids([self.emb_cols[i] for i in idxs], features, self.embedding_weights[0].device, memo)
        return (ts, got[1]) if got[0] == "onehot" else None

    def forward(self, features, pad_to=1, memo=None):
        """-> x0 [B, column_num].  pad_to = 4 (inference only): x0 is returned as [B, round_up(column_num, 4)] whose extra
        columns are zero -- the row stride the 16-byte paths of dir_dcn_cross_f32 / dir_dense_f32 want when column_num is odd
        (DCN's 26 x 16 + 13 = 429): written once here, no padded copy later."""
        device = self.embedding_weights[0].device if len(self.embedding_weights) else next(iter(
            v for v in features.values() if isinstance(v, torch.Tensor))).device
        if torch.is_grad_enabled():
            return self._forward_train(features, device, memo)
        B = None
        x0 = None

        ld = (self.column_num + pad_to - 1) // pad_to * pad_to

        def alloc(nb):
            buf = torch.empty((nb, ld), dtype=torch.float32, device=device)
            if ld != self.column_num:
                buf[:, self.column_num:].zero_()
            return buf

        run, run_off = [], 0          # name-adjacent numeric columns are written as ONE block (13 Criteo numerics: 2 kernels, not 13)

        def flush():
            nonlocal run, x0, B
            if run:
                blk = run[0] if len(run) == 1 else torch.cat(run, dim=1)
                if x0 is None:
                    B = blk.shape[0]
                    x0 = alloc(B)
                x0[:, run_off:run_off + blk.shape[1]] = blk
                run = []

        for c in self.columns:
            if isinstance(c, NumericColumn):
                v = features[c.key].to(device=device, dtype=torch.float32).reshape(-1, c.dimension)
                if not run:
                    run_off = self._col_offset(c)
                run.append(v)
            else:
                flush()
        flush()
        for ts, idxs, comb, mn in self._tablesets():
            cols = [self.emb_cols[i] for i in idxs]
            got = collect_ids(cols, features, device, memo)
            nb = got[1].shape[0] if got[0] == "onehot" else got[4]
            if x0 is None:
                B = nb
                x0 = alloc(B)
            off = self._col_offset(cols[0])
            view = x0[:, off:off + len(cols) * cols[0].dimension]
            if got[0] == "onehot":
                ops.embedding_bag(ts, got[1], out=view, max_norm=mn)
            else:
                ops.embedding_bag(ts, got[1], got[2], got[3], combiner=comb, field_major=True, out=view, max_norm=mn)
        for c in self.columns:
            if isinstance(c, IndicatorColumn):  # multi-hot counts ([TF-upstream] indicator_column)
                ind = self._indicator(c, features, device, B)
                if x0 is None:
                    B = ind.shape[0]
                    x0 = alloc(B)
                x0[:, self._col_offset(c):self._col_offset(c) + c.dimension] = ind
        return x0
