"""shard.py -- row-sharded embedding tables with an all-to-all lookup (RCCL over xGMI on the GPU box).

Semantic precedent in the reference: min_max_variable_partitioner(max_partitions=num_ps_replicas, ...)
row-partitions the embedding variables and lookups resolve with partition_strategy='div'
(models/DeepFM/deepFM.py:163-167; [TF-upstream] embedding_lookup).  Here every rank of a
torch.distributed group owns the contiguous 'div' row range of EVERY table, and one lookup is

    bucket   (HIP: route every id to its owner + counting sort by owner -> packed payload, inverse perm)
    exchange (all_to_all_single of the per-owner counts, then of the int64 payload)
    gather   (HIP on the owner: payload -> rows)
    exchange (all_to_all_single of the fp32 rows back)
    finish   (HIP: the fused gather+FM kernel with "table" = the received row buffer and ids = the inverse
              permutation: one pass un-permutes into [B_local, F*K] and produces the FM logit)

The collectives are torch.distributed.all_to_all_single (backend "nccl" = RCCL on ROCm; "gloo" in the CPU
tests); the split sizes cost one host read of 2*P integers per lookup.  world_size == 1 skips the collectives
(unless force_collective) but still runs the three HIP steps.  The HIP steps sit behind a small backend
object so that the CPU (gloo) tests can stand the oracle in for them; the default backend is the HIP one.
"""
import os

import torch
import torch.distributed as dist

from . import ops

_HOST_STAGED = os.environ.get("DIR_SHARD_HOST_STAGED") == "1"


def div_range(vocab, P, rank):
    """Rows [start, end) that `rank` owns under 'div' (first vocab % P shards hold one extra row)."""
    q, r = divmod(int(vocab), int(P))
    start = rank * (q + 1) if rank < r else r * (q + 1) + (rank - r) * q
    return start, start + (q + 1 if rank < r else q)


class HipBackend:
    """The product path: three launches of libdir_hip.so kernels."""

    def __init__(self, local_tables, vocab_dev, P):
        self.ts = ops.TableSet(local_tables)
        # row policy "auto": the owner-side gather streams (non-temporal) when this rank's shards exceed the
        # Infinity Cache, like the single-GPU gather
        self.vocab_dev = vocab_dev
        self.P = P
        self._back = None
        self._back_ts = None

    def bucket(self, flat_ids):
        return ops.shard_bucket(flat_ids, self.vocab_dev, self.P)

    def gather_packed(self, payload):
        return ops.gather_packed(self.ts, payload)

    def make_optimizer(self, lr, initial_accumulator_value):
        return ops.SparseAdagrad(self.ts, lr, initial_accumulator_value)

    def apply_adagrad(self, opt, payload, grad_rows):
        opt.step_payload(payload, grad_rows)                     # HIP: sorted (row, entry) pairs, per-tile segmented reduce

    def back_buffer(self, n, K, device):
        if self._back is None or self._back.shape[0] != n:
            self._back = torch.empty((n, K), dtype=torch.float32, device=device)
            self._back_ts = None
        return self._back

    def finish(self, back, inv, B, F, want_fm):
        if self._back_ts is None or self._back_ts.tables[0].data_ptr() != back.data_ptr():
            self._back_ts = ops.TableSet([back] * F)
            self._back_ts.row_policy = "reuse"   # just received: largely cache-resident
        if want_fm:
            return ops.gather_fm(self._back_ts, inv.view(B, F))
        return ops.embedding_bag(self._back_ts, inv.view(B, F)), None


class ShardedTables:
    """This rank's row shard of F tables [vocab_f, K]."""

    def __init__(self, local_tables, vocab, group=None, backend=None, force_collective=False):
        self.group = group
        self.force_collective = force_collective  # issue the all_to_all calls even when world_size == 1
        self.P = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.vocab = [int(v) for v in vocab]
        self.F = len(self.vocab)
        self.local_tables = list(local_tables)
        self.K = self.local_tables[0].shape[1]
        for f, t in enumerate(self.local_tables):
            s, e = div_range(self.vocab[f], self.P, self.rank)
            if t.shape[0] != e - s:
                raise ValueError("table %d: rank %d must hold rows [%d,%d) (%d rows), got %d" % (f, self.rank, s, e, e - s, t.shape[0]))
        self.device = self.local_tables[0].device
        self.vocab_dev = torch.tensor(self.vocab, dtype=torch.int64, device=self.device)
        self.backend = backend or HipBackend(self.local_tables, self.vocab_dev, self.P)

    @classmethod
    def from_full(cls, full_tables, group=None, **kw):
        """Slice replicated full tables down to this rank's shard (tests / small cases)."""
        P = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        vocab = [t.shape[0] for t in full_tables]
        loc = []
        for t in full_tables:
            s, e = div_range(t.shape[0], P, rank)
            loc.append(t[s:e].contiguous())
        return cls(loc, vocab, group=group, **kw)

    def _collective(self):
        return self.P > 1 or (self.force_collective and dist.is_initialized())

    def _a2a(self, out, inp, out_splits, in_splits):
        if self._collective():
            if _HOST_STAGED:     # development transport (several ranks on ONE GPU over gloo): never set on a multi-GPU node
                o = torch.empty(out.shape, dtype=out.dtype)
                dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=self.group)
                out.copy_(o)
            else:
                dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)
        else:
            out.copy_(inp)

    # ---- training: the gradient rows travel the forward's row exchange backwards and the OWNER updates its shard -------
    def enable_training(self, lr, initial_accumulator_value=0.1):
        """Attach the owner-side sparse Adagrad (the reference's dnn_optimizer='Adagrad', deepFM.py:61, applied by the
        parameter server that holds the partition: deepFM.py:163-167).  lookup_train() then returns an embedding matrix
        whose backward() updates every rank's shard in place; duplicate rows -- from one rank or several -- are summed
        first, i.e. one synchronous step over the global batch."""
        self.optimizer = self.backend.make_optimizer(lr, initial_accumulator_value)
        return self

    def lookup_train(self, ids):
        """Differentiable lookup: emb [B_local, F*K] with a grad_fn (the tables themselves get no .grad)."""
        if getattr(self, "optimizer", None) is None:
            raise RuntimeError("call enable_training(lr) first")
        anchor = torch.zeros((), dtype=torch.float32, device=ids.device, requires_grad=True)
        return _ShardedLookup.apply(self, ids, anchor)

    def _forward_saved(self, ids):
        B, F = ids.shape
        K, be = self.K, self.backend
        flat = ids.reshape(-1).contiguous()
        n = flat.numel()
        payload, inv, send_counts, _ = be.bucket(flat)
        recv_counts = torch.empty_like(send_counts)
        self._a2a(recv_counts, send_counts, None, None)
        both = torch.stack([send_counts, recv_counts]).tolist()
        sc, rc = [int(v) for v in both[0]], [int(v) for v in both[1]]
        recv = torch.empty(sum(rc), dtype=torch.int64, device=flat.device)
        self._a2a(recv, payload, rc, sc)
        rows = be.gather_packed(recv)
        back = torch.empty((n, K), dtype=torch.float32, device=flat.device)       # private: kept alive by autograd users
        self._a2a(back.view(-1), rows.reshape(-1), [c * K for c in sc], [c * K for c in rc])
        emb, _ = be.finish(back, inv, B, F, False)
        return emb, (inv, sc, rc, recv)

    def _backward_apply(self, saved, g_emb):
        inv, sc, rc, recv = saved
        K = self.K
        n = inv.numel()
        g = g_emb.contiguous().view(n, K)
        gsend = torch.empty_like(g)
        gsend[inv] = g                                   # entry e's row sits at position inv[e] of the exchange order
        grecv = torch.empty((recv.numel(), K), dtype=torch.float32, device=g.device)
        self._a2a(grecv.view(-1), gsend.view(-1), [c * K for c in rc], [c * K for c in sc])   # the forward exchange, reversed
        self.backend.apply_adagrad(self.optimizer, recv, grecv)

    def lookup(self, ids, want_fm=False):
        """ids [B_local, F] int64 (global row ids; < 0 pruned -> zeros) -> emb [B_local, F*K] fp32
        (and the FM second-order logit [B_local, 1] when want_fm)."""
        B, F = ids.shape
        if F != self.F:
            raise ValueError("ids must be [B, F=%d]" % self.F)
        K, be = self.K, self.backend
        flat = ids.reshape(-1).contiguous()
        n = flat.numel()
        payload, inv, send_counts, _ = be.bucket(flat)                      # HIP
        recv_counts = torch.empty_like(send_counts)
        self._a2a(recv_counts, send_counts, None, None)
        both = torch.stack([send_counts, recv_counts]).tolist()             # the one host read per lookup
        sc, rc = [int(v) for v in both[0]], [int(v) for v in both[1]]
        recv = torch.empty(sum(rc), dtype=torch.int64, device=flat.device)
        self._a2a(recv, payload, rc, sc)
        rows = be.gather_packed(recv)                                       # HIP (owner side)
        back = be.back_buffer(n, K, flat.device)
        self._a2a(back.view(-1), rows.reshape(-1), [c * K for c in sc], [c * K for c in rc])
        emb, fm = be.finish(back, inv, B, F, want_fm)                       # HIP: un-permute (+ FM)
        return (emb, fm) if want_fm else emb


class _ShardedLookup(torch.autograd.Function):
    """emb = ShardedTables.lookup(ids) with a backward that routes the row gradients to their owners and lets each owner
    apply the sparse Adagrad update to its shard (no gradient tensor is returned for the tables)."""

    @staticmethod
    def forward(ctx, st, ids, anchor):
        emb, saved = st._forward_saved(ids)
        ctx.st, ctx.saved = st, saved
        return emb

    @staticmethod
    def backward(ctx, g):
        ctx.st._backward_apply(ctx.saved, g)
        return None, None, None


# ---- the dense (replicated) side of multi-GPU training -----------------------------------------------------------------
def allreduce_grads(params, group=None, bucket_bytes=64 << 20, average=False):
    """Sum (or average) the .grad of replicated parameters over the ranks: the gradients are packed into flat buckets of about
    `bucket_bytes` (xGMI is point-to-point: few large all-reduces, not one per tensor), every bucket's all_reduce is issued
    before the first is waited for, then the results are copied back.  Parameters without a gradient are skipped on every rank
    alike (the caller guarantees the ranks agree on which parameters have gradients)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    host_staged = _HOST_STAGED or (dist.get_backend(group) == "gloo" and grads[0].is_cuda)
    buckets, cur, size = [], [], 0
    for g in grads:
        if g.is_sparse:
            raise ValueError("allreduce_grads: dense gradients only (sharded tables update at their owners)")
        cur.append(g)
        size += g.numel() * g.element_size()
        if size >= bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
    if cur:
        buckets.append(cur)
    flats, works = [], []
    for b in buckets:
        flat = torch.cat([g.reshape(-1) for g in b])
        if host_staged:
            flat = flat.cpu()
        flats.append(flat)
        works.append(dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True))
    scale = 1.0 / dist.get_world_size(group) if average else 1.0
    for b, flat, w in zip(buckets, flats, works):
        w.wait()
        off = 0
        for g in b:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g).to(g.device) * scale if average else flat[off:off + n].view_as(g))
            off += n


class ShardedDeepFMTrainer:
    """One synchronous multi-GPU training step of a DeepFM: the embedding tables are row-sharded over the ranks
    (ShardedTables: lookup = two all-to-alls, backward = the row exchange reversed + the owner's sparse Adagrad), the dense
    tower is replicated and its gradients are summed with bucketed all-reduces.  This is the reference's between-graph
    replicated training on parameter servers (partitioned embedding variables, deepFM.py:163-167; Adagrad on them, :61; the
    canned head's SUM loss reduction, :72 -- so gradients ADD over workers) restated for one process per GPU.
    `model` supplies the tower (dnn_logit_fn) and F, K; its own embedding_weights are not used."""

    def __init__(self, model, tables, lr_sparse, dense_optimizer, group=None, initial_accumulator_value=0.1):
        self.model, self.tables, self.group = model, tables, group
        self.dense_params = [p for n, p in model.named_parameters()
                             if not (n.startswith("embedding_weights") or n.startswith("linear_weights"))]
        self.dense_optimizer = dense_optimizer
        tables.enable_training(lr_sparse, initial_accumulator_value)

    def step(self, ids, labels):
        """ids [B_local, F] global row ids, labels [B_local, 1] -> this rank's summed loss (detached)."""
        from . import autograd as ag
        m = self.model
        self.dense_optimizer.zero_grad(set_to_none=True)
        emb = self.tables.lookup_train(ids)                                   # tables update inside backward()
        logits = ag.fm_logit(emb, m.F, m.K) + m.dnn_logit_fn(emb)             # fm_logit_fn + dnn_logit_fn, deepFM.py:337-338
        loss = torch.nn.functional.binary_cross_entropy_with_logits(logits, labels, reduction="sum")
        loss.backward()
        allreduce_grads(self.dense_params, self.group)
        self.dense_optimizer.step()
        return loss.detach()
