"""shard.py -- row-sharded embedding tables with an all-to-all lookup (RCCL over xGMI on the GPU box).

Semantic precedent in the reference: min_max_variable_partitioner(max_partitions=num_ps_replicas, ...)
row-partitions the embedding variables and lookups resolve with partition_strategy='div'
(models/DeepFM/deepFM.py:163-167; [TF-upstream] embedding_lookup).  Here every rank of a
torch.distributed group owns the contiguous 'div' row range of EVERY table, and one lookup is

    route (HIP)  ->  bucket by owner  ->  all_to_all(ids)  ->  local row gather (HIP)
                 ->  all_to_all(rows) ->  un-permute into [B_local, F*K]

The two collectives are torch.distributed.all_to_all_single (backend "nccl" = RCCL on ROCm; "gloo" in the
CPU tests).  world_size == 1 skips both collectives but still runs route / bucket / gather / un-permute,
so the single-GPU run exercises the same kernels.  The route and gather steps are injectable so that the
CPU (gloo) tests can stand in the oracle for the two HIP kernels; the defaults are the HIP ops.
"""
import torch
import torch.distributed as dist

from . import ops


def div_range(vocab, P, rank):
    """Rows [start, end) that `rank` owns under 'div' (first vocab % P shards hold one extra row)."""
    q, r = divmod(int(vocab), int(P))
    start = rank * (q + 1) if rank < r else r * (q + 1) + (rank - r) * q
    return start, start + (q + 1 if rank < r else q)


class ShardedTables:
    """This rank's row shard of F tables [vocab_f, K]."""

    def __init__(self, local_tables, vocab, group=None, route_fn=None, gather_fn=None, force_collective=False):
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
        self._route = route_fn or (lambda ids: ops.shard_route(ids, self.vocab_dev, self.P))
        if gather_fn is None:
            self._ts = ops.TableSet(self.local_tables)
            gather_fn = lambda slot, row: ops.gather_rows(self._ts, slot, row)  # noqa: E731
        self._gather = gather_fn

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

    def _a2a(self, out, inp, out_splits, in_splits):
        if self.P == 1 and not (self.force_collective and dist.is_initialized()):
            out.copy_(inp)
        else:
            dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)

    def lookup(self, ids):
        """ids [B_local, F] int64 (global row ids; < 0 pruned -> zeros) -> [B_local, F*K] fp32."""
        B, F = ids.shape
        if F != self.F:
            raise ValueError("ids must be [B, F=%d]" % self.F)
        P, K = self.P, self.K
        flat = ids.reshape(-1).contiguous()
        n = flat.numel()
        owner, local = self._route(flat)                                   # HIP: 'div' owner + local row
        owner = owner.to(torch.int64)
        order = torch.argsort(owner, stable=True)                          # bucket by owner (P buckets)
        send_counts = torch.bincount(owner, minlength=P)
        slot = (torch.arange(n, device=flat.device, dtype=torch.int64) % F)
        payload = (local * F + slot)[order]                                # local row and slot in one int64
        payload = torch.where(local[order] < 0, torch.full_like(payload, -1), payload)
        recv_counts = torch.empty_like(send_counts)
        self._a2a(recv_counts, send_counts, None, None)
        sc, rc = send_counts.tolist(), recv_counts.tolist()                # host sync: split sizes
        recv = torch.empty(sum(rc), dtype=torch.int64, device=flat.device)
        self._a2a(recv, payload, rc, sc)
        rrow = torch.where(recv < 0, recv, torch.div(recv, F, rounding_mode="floor"))
        rslot = torch.where(recv < 0, torch.zeros_like(recv), recv % F).to(torch.int32)
        rows = self._gather(rslot, rrow)                                   # HIP: owner-side row gather
        back = torch.empty((n, K), dtype=torch.float32, device=flat.device)
        self._a2a(back.view(-1), rows.reshape(-1), [c * K for c in sc], [c * K for c in rc])
        out = torch.empty((n, K), dtype=torch.float32, device=flat.device)
        out[order] = back                                                  # un-permute
        return out.view(B, F * K)
