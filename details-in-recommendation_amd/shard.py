"""shard.py -- row-sharded embedding tables with an all-to-all lookup (RCCL over xGMI on the GPU box).

Semantic precedent in the reference: min_max_variable_partitioner(max_partitions=num_ps_replicas, min_slice_size=64 << 20)
row-partitions the embedding variables and lookups resolve with partition_strategy='div'
(models/DeepFM/deepFM.py:163-167; [TF-upstream] embedding_lookup).  Here every rank of a torch.distributed group owns
contiguous 'div' row slices of the tables (by default one slice of EVERY table; `partitions=` applies the reference's
slice-count rule instead) and one lookup is, per micro-batch,

    bucket   (HIP: route every id to its owner, one fixed-capacity slab per owner -> payload slabs, inverse positions)
    exchange (all_to_all_single of the slabs, EQUAL splits: no split sizes ever reach the host)
    gather   (HIP on the owner: slab -> rows)
    exchange (all_to_all_single of the fp32 rows back, equal splits)
    finish   (HIP: the fused gather+FM kernel with "table" = the received row buffer and ids = the inverse positions:
              one pass un-permutes into [B_local, F*K] and produces the FM logit)

The batch is cut into `chunks` micro-batches on two streams, software-pipelined so that the id exchange of chunk c+1 is
issued before the row exchange of chunk c: owner gather / finish of neighbouring chunks run under the (link-bound) row
exchange.  A slab that is too small is detected on the device (overflow flag, read AFTER the whole pipeline has been
enqueued, so the stream never drains); the lookup is then repeated on the exact variable-size path (one host read of the
split sizes) and the capacity grows.  `dedup=True` sends each (slot, row) once per owner and chunk ([TF-upstream]
embedding_lookup_sparse's unique-before-gather): on skewed ids it cuts the link-bound bytes.

The collectives are torch.distributed.all_to_all_single (backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).
world_size == 1 skips the collectives (unless force_collective) but still runs the three HIP steps.  The HIP steps sit
behind a small backend object so that the CPU (gloo) tests can stand the oracle in for them; the default backend is the
HIP one.
"""
import contextlib
import math
import os

import torch
import torch.distributed as dist

from . import ops

_HOST_STAGED = os.environ.get("DIR_SHARD_HOST_STAGED") == "1"
MIN_SLICE_SIZE = 64 << 20     # deepFM.py:167


def div_range(vocab, P, rank):
    """Rows [start, end) that `rank` owns under 'div' (first vocab % P shards hold one extra row)."""
    q, r = divmod(int(vocab), int(P))
    start = rank * (q + 1) if rank < r else r * (q + 1) + (rank - r) * q
    return start, start + (q + 1 if rank < r else q)


def partitions_for(vocab, K, max_partitions, min_slice_size=MIN_SLICE_SIZE, bytes_per_element=4):
    """[TF-upstream] partitioned_variables.min_max_variable_partitioner(max_partitions, axis=0, min_slice_size) as the reference
    calls it (models/DeepFM/deepFM.py:163-167, ESMM_wide_deep.py:213-217): a [vocab, K] variable is cut along axis 0 into
        max(1, min(vocab, max_partitions, ceil(vocab * K * bytes_per_element / min_slice_size)))
    slices (so every slice holds at least min_slice_size bytes, and num_ps_replicas = 0 leaves the variable whole).  The
    slices are the 'div' row ranges (the first vocab % n slices hold one extra row: div_range)."""
    total = int(vocab) * int(K) * int(bytes_per_element)
    return max(1, min(int(vocab), int(max_partitions), int(math.ceil(total / float(min_slice_size)))))


def place_slices(parts, P):
    """Rank of slice 0 of every table: the slices are dealt round-robin over the ranks in creation order ([TF-upstream]
    replica_device_setter's default strategy places each slice -- a variable of its own -- on the next ps task)."""
    first, nxt = [], 0
    for p in parts:
        first.append(nxt % P)
        nxt += p
    return first


def local_slice(vocab, parts, first, P, rank):
    """Rows [start, end) of a table that `rank` holds: slice j = (rank - first) mod P when j < parts, else nothing."""
    j = (rank - first) % P
    return div_range(vocab, parts, j) if j < parts else (0, 0)


def _round_up(x, m):
    return (int(x) + m - 1) // m * m


class HipBackend:
    """The product path: launches of libdir_hip.so kernels."""

    def __init__(self, local_tables, vocab_dev, P, parts_dev=None, first_dev=None):
        self.ts = ops.TableSet(local_tables)
        # row policy "auto": the owner-side gather streams (non-temporal) when this rank's shards exceed the
        # Infinity Cache, like the single-GPU gather
        self.vocab_dev = vocab_dev
        self.P = P
        self.parts_dev, self.first_dev = parts_dev, first_dev
        self._back = None
        self._back_ts = None
        self._finish_ts = {}

    # ---- variable-size (exact) path -----------------------------------------------------------------------
    def bucket(self, flat_ids):
        return ops.shard_bucket(flat_ids, self.vocab_dev, self.P, parts=self.parts_dev, first=self.first_dev)

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

    def finish(self, back, inv, B, F, want_fm, out=None, fm=None):
        if self._back_ts is None or self._back_ts.tables[0].data_ptr() != back.data_ptr() or self._back_ts.vocab[0] != back.shape[0]:
            self._back_ts = ops.TableSet([back] * F)
            self._back_ts.row_policy = "reuse"   # just received: largely cache-resident
        if want_fm:
            return ops.gather_fm(self._back_ts, inv.view(B, F), out=out, fm=fm)
        return ops.embedding_bag(self._back_ts, inv.view(B, F), out=out), None

    # ---- fixed-capacity path ------------------------------------------------------------------------------
    def new_workspace(self, device):
        return torch.zeros(128, dtype=torch.int32, device=device)      # dir_shard_bucket_cap_workspace_bytes: slab counters + arrival counter

    def bucket_cap(self, ids2d, cap, payload, inv, counts, overflow, workspace, stat=None, dedup=False):
        """ids2d [Bc, F] -> slabs (packed headers) + inv; dedup: inv comes back field-major (inv2d() gives the [Bc, F] view)."""
        if dedup:
            ops.shard_bucket_cap_dedup(ids2d, self.vocab_dev, self.P, cap, payload, inv, counts, overflow, workspace,
                                       parts=self.parts_dev, first=self.first_dev, stat=stat)
        else:
            ids2d = ids2d if ids2d.is_contiguous() else ids2d.contiguous()
            ops.shard_bucket_cap(ids2d.reshape(-1), self.vocab_dev, self.P, cap, payload, inv, counts, overflow, workspace,
                                 parts=self.parts_dev, first=self.first_dev, stat=stat)

    @staticmethod
    def inv2d(inv, Bc, F, dedup):
        return inv.view(F, Bc).t() if dedup else inv.view(Bc, F)

    def slab_stat(self, recv_all, n_slabs, cap, stat):
        ops.shard_slab_stat(recv_all, n_slabs, cap, stat)

    def gather_slabs(self, recv, cap, out):
        ops.gather_slabs(self.ts, recv, self.P, cap, out)

    def finish_chunk(self, back, inv2d, want_fm, out, fm):
        key = (back.data_ptr(), back.shape[0], inv2d.shape[1])
        ts = self._finish_ts.get(key)
        if ts is None:
            if len(self._finish_ts) > 64:
                self._finish_ts.clear()
            ts = self._finish_ts[key] = ops.TableSet([back] * inv2d.shape[1])
            ts.row_policy = "reuse"
        if want_fm:
            ops.gather_fm(ts, inv2d, out=out, fm=fm)
        else:
            ops.embedding_bag(ts, inv2d, out=out)


_ROWS_TS = {}


def rows_as_tables(rows, F, absmax=None):
    """The received row buffer of a lookup_consume() micro-batch as a TableSet of F identical "tables" (one per slot), cached by
    address: what ops.gather_fm / ops.tower(gather=...) take as their tables, with the inverse positions as ids.
    absmax: the largest |value| of the SHARDED tables (ShardedTables.absmax()): the magnitude bound a kernel that picks its split
    arithmetic by the tables' range (ops.tower(split=None)) must see -- the buffer's own contents change with every lookup, behind the
    back of TableSet.absmax()'s version-keyed cache."""
    key = (rows.data_ptr(), rows.shape[0], rows.shape[1], F)
    ts = _ROWS_TS.get(key)
    if ts is None:
        if len(_ROWS_TS) > 64:
            _ROWS_TS.clear()
        ts = _ROWS_TS[key] = ops.TableSet([rows] * F)
        ts.row_policy = "reuse"      # just received: largely cache-resident
    if absmax is not None:
        ts.absmax = lambda every=1, _v=float(absmax): _v
    return ts


class _Plan:
    """Persistent buffers of the fixed-capacity pipeline for one (local batch size, buffer slot, slab capacity): stable addresses,
    so nothing is allocated per lookup and a lookup can be captured in a HIP graph.  Only the slab capacity has to agree across
    the ranks (equal-split exchanges); everything sized by the batch is local."""

    def __init__(self, st, B, cap):
        dev, P, F, K = st.device, st.P, st.F, st.K
        self.B, self.cap = B, cap
        C = self.C = st._C()
        per = -(-B // C) if B else 0
        self.bounds = [(min(B, c * per), min(B, (c + 1) * per)) for c in range(C)]       # a chunk may be empty (B < C)
        i64 = dict(dtype=torch.int64, device=dev)
        alias = not st._collective()                     # one rank, no collectives: receive buffers ARE the send buffers
        slab = P * (cap + 1)
        self.send_all = torch.empty(C * slab, **i64)
        self.recv_all = self.send_all if alias else torch.empty(C * slab, **i64)
        self.send = [self.send_all[c * slab:(c + 1) * slab] for c in range(C)]
        self.recv = [self.recv_all[c * slab:(c + 1) * slab] for c in range(C)]
        self.inv = [torch.empty(max(1, (e - s) * F), **i64)[:(e - s) * F] for s, e in self.bounds]
        self.counts = torch.zeros((C, P), **i64)
        self.flags = torch.zeros((C, 1), dtype=torch.int32, device=dev)
        self.ws = [st.backend.new_workspace(dev) for _ in range(C)]
        self.rows = [torch.empty((P * cap, K), dtype=torch.float32, device=dev) for _ in range(C)]
        self.back = self.rows if alias else [torch.empty((P * cap, K), dtype=torch.float32, device=dev) for _ in range(C)]
        self.cstat = torch.zeros((C, 2), **i64)           # per chunk [overflow, max demand] of THIS rank (diagnostic)
        self.stat = torch.zeros(2, **i64)                 # ... over all chunks and ranks, read off the received slab headers
        self.host = torch.empty(2, dtype=torch.int64, pin_memory=dev.type == "cuda")
        self.fin = None                                   # events behind the last enqueued lookup's final kernels (one per side stream)


class _Lookup:
    """A lookup whose pipeline has been enqueued (on the side streams on a GPU).  result() makes the caller's stream wait for it,
    applies the overflow policy and returns emb (or (emb, fm)).  Issue the NEXT lookup before calling result() -- or before the
    compute that consumes this one -- and the exchange runs under that compute."""

    def __init__(self, st, plan, ids, want_fm, out, fm, done, exact=None, consumer=None):
        self.st, self.plan, self.ids, self.want_fm, self.out, self.fm, self.done = st, plan, ids, want_fm, out, fm, done
        self.exact = exact
        self.consumer = consumer
        self.fin = plan.fin if plan is not None else None      # this lookup's completion events on the side streams
        self.joined = exact is not None
        self.checked = exact is not None or done is False

    def join(self):
        """The caller's stream waits for THIS lookup's last kernels (events recorded when it was enqueued), not for whatever else has
        been queued on the side streams since: a later lookup keeps running under the compute that follows."""
        if not self.joined:
            self.joined = True
            if self.fin:
                cur = torch.cuda.current_stream(self.st.device)
                for e in self.fin:
                    cur.wait_event(e)

    def result(self):
        st = self.st
        self.join()
        if self.exact is not None:
            return self.exact
        if not self.checked:
            if st.check == "eager":
                self.checked = True
                over, demand = st._read_flags(self.plan, self.done)
                st._learn(self.plan, over, demand)
                if over:                                      # rare: repeat on the exact path (results overwrite out / fm in stream order)
                    st.stats["fallbacks"] += 1
                    st._lookup_exact(self.ids, self.want_fm, out=self.out, fm=self.fm, consumer=self.consumer)
        return (self.out, self.fm) if self.want_fm else self.out


class _RowsLookup:
    """Handle of ShardedTables.lookup_rows_async."""

    def __init__(self, lk, chunks):
        self.lk, self.chunks = lk, chunks

    def result(self):
        self.lk.result()
        return list(self.chunks)


class ShardedTables:
    """This rank's row slices of F tables [vocab_f, K].

    partitions: None = every table cut into P slices, slice r on rank r (the north star's row sharding); "reference" = the
      reference partitioner's slice-count rule (partitions_for with max_partitions = P) with the slices dealt round-robin; or an
      explicit list of slice counts.  local_tables[f] holds local_slice(...) rows of table f (possibly zero rows).
    chunks: micro-batches per lookup when there is an exchange to hide (pipelined on two streams; every collective costs
      host time, so 2 by default; without collectives -- one rank -- a lookup is ONE chunk on the caller's stream);
    slack: slab capacity = ceil(slack * n / P) entries per owner and chunk (None: n / P + 8 sigma of the binomial count + 64:
      ~3 % padding at 16 384 x 26 ids over 8 ranks);  max_batch: the largest local batch any rank will look up (sizes the slabs;
      default: the first lookup's batch, MAX over the ranks);
    mode: "auto" (fixed capacity, exact fallback on overflow; switches to "exact" when the owners' demand is so uneven that
      padding would cost more than the host read), "fixed", "exact";
    check: "eager" (result() reads the overflow verdict: no wait when the next lookup was issued first; an overflow is repaired on
      the exact path), "lazy" (a lookup's verdict is read right after the NEXT lookup has been enqueued; an overflow raises),
      "never" (graph capture; call check_overflow() yourself);
    dedup: send each (slot, row) once per owner, chunk and 2048/4096-sample tile (csrc/ids.hip: bucket_cap_dedup_k);
    side_cus: confine the lookup's side streams to that many compute units each (for lookups prefetched under MFMA-bound compute).

    Ranks may look up DIFFERENT local batch sizes (an uneven last batch): the only quantity the equal-split exchanges need to agree
    on is the slab capacity, which is agreed once (first lookup: one host MAX) and afterwards changes only on statistics every rank
    reads identically off the slab headers.  Every rank must call lookup the same number of times (SPMD)."""

    def __init__(self, local_tables, vocab, group=None, backend=None, force_collective=False, partitions=None, chunks=2,
                 slack=None, mode="auto", check="eager", dedup=False, max_batch=None, side_cus=None):
        self.group = group
        self.force_collective = force_collective  # issue the all_to_all calls even when world_size == 1
        self.P = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.vocab = [int(v) for v in vocab]
        self.F = len(self.vocab)
        self.local_tables = list(local_tables)
        self.K = self.local_tables[0].shape[1]
        P = self.P
        if partitions is None:
            self.parts, self.first = [P] * self.F, [0] * self.F
        else:
            self.parts = ([partitions_for(v, self.K, P) for v in self.vocab] if partitions == "reference"
                          else [int(p) for p in partitions])
            if len(self.parts) != self.F or any(p < 1 or p > P for p in self.parts):
                raise ValueError("partitions: one slice count in [1, P=%d] per table" % P)
            self.first = place_slices(self.parts, P)
        for f, t in enumerate(self.local_tables):
            s, e = local_slice(self.vocab[f], self.parts[f], self.first[f], P, self.rank)
            if t.shape[0] != e - s:
                raise ValueError("table %d: rank %d must hold rows [%d,%d) (%d rows), got %d" % (f, self.rank, s, e, e - s, t.shape[0]))
        self.device = self.local_tables[0].device
        self.vocab_dev = torch.tensor(self.vocab, dtype=torch.int64, device=self.device)
        custom = partitions is not None
        self.parts_dev = torch.tensor(self.parts, dtype=torch.int32, device=self.device) if custom else None
        self.first_dev = torch.tensor(self.first, dtype=torch.int32, device=self.device) if custom else None
        self.backend = backend or HipBackend(self.local_tables, self.vocab_dev, P, self.parts_dev, self.first_dev)
        if mode not in ("auto", "fixed", "exact") or check not in ("eager", "lazy", "never"):
            raise ValueError("mode: auto | fixed | exact; check: eager | lazy | never")
        self.chunks, self.slack, self.mode, self.check, self.dedup = max(1, int(chunks)), slack, mode, check, bool(dedup)
        self.max_batch = max_batch
        self.side_cus = side_cus      # None: ordinary side streams; n: the lookup's two side streams are confined to n CUs each
        self._plans = {}
        self._cap = None              # the agreed slab capacity (collective mode); _cap0: its first value (the no-skew demand)
        self._cap_train = None        # the training pipeline's own capacity (never shrunk by de-duplicated inference verdicts)
        self._cap0 = None
        self._use_exact = mode == "exact"
        self._unchecked = []          # lookups whose overflow verdict has not been read yet (check = lazy / never)
        self._slot = 0
        self._inflight = {}
        self._streams = None
        self.stats = {"lookups": 0, "fallbacks": 0, "cap": None}
        self._updates = 0             # owner-side optimiser steps applied so far (the same on every rank: absmax()'s collective decision)
        self._absmax_all = None

    @classmethod
    def from_full(cls, full_tables, group=None, **kw):
        """Slice replicated full tables down to this rank's shard (tests / small cases)."""
        P = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        vocab = [t.shape[0] for t in full_tables]
        K = full_tables[0].shape[1]
        partitions = kw.get("partitions")
        if partitions is None:
            parts, first = [P] * len(vocab), [0] * len(vocab)
        else:
            parts = [partitions_for(v, K, P) for v in vocab] if partitions == "reference" else [int(p) for p in partitions]
            first = place_slices(parts, P)
        loc = []
        for f, t in enumerate(full_tables):
            s, e = local_slice(vocab[f], parts[f], first[f], P, rank)
            loc.append(t[s:e].contiguous())
        return cls(loc, vocab, group=group, **kw)

    def _collective(self):
        return self.P > 1 or (self.force_collective and dist.is_initialized())

    def _C(self):
        """Micro-batches per lookup: a constructor constant (never a function of the local batch: every rank issues the same
        number of collectives)."""
        return self.chunks if self._collective() else 1

    def _host_staged(self, t):
        return _HOST_STAGED or (t.is_cuda and dist.get_backend(self.group) == "gloo")

    def _a2a(self, out, inp, out_splits, in_splits):
        if self._collective():
            if self._host_staged(out):     # development transport (several ranks on ONE GPU over gloo): never on a multi-GPU node
                o = torch.empty(out.shape, dtype=out.dtype)
                dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=self.group)
                out.copy_(o)
            else:
                dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)
        else:
            out.copy_(inp)

    def _a2a_equal(self, out, inp):
        """Equal-split all-to-all, asynchronous where the transport allows: -> a work handle (or None when done / aliased)."""
        if not self._collective():
            return None                                   # out IS inp (see _Plan)
        if self._host_staged(out):
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu(), group=self.group)
            out.copy_(o)
            return None
        return dist.all_to_all_single(out, inp, group=self.group, async_op=True)

    # ---- training: the gradient rows travel the forward's row exchange backwards and the OWNER updates its shard -------
    def enable_training(self, lr, initial_accumulator_value=0.1):
        """Attach the owner-side sparse Adagrad (the reference's dnn_optimizer='Adagrad', deepFM.py:61, applied by the
        parameter server that holds the partition: deepFM.py:163-167).  lookup_train() then returns an embedding matrix
        whose backward() updates every rank's shard in place; duplicate rows -- from one rank or several -- are summed
        first, i.e. one synchronous step over the global batch."""
        self.optimizer = self.backend.make_optimizer(lr, initial_accumulator_value)
        return self

    def lookup_train(self, ids):
        """Differentiable lookup: emb [B_local, F*K] with a grad_fn (the tables themselves get no .grad).  Runs the fixed-capacity
        pipeline (no split sizes reach the host; the gradient rows go back through the same equal-split slabs); the exact
        variable-size path only when mode == "exact", after an overflow, or when "auto" has given up on slabs.  One training lookup
        may be outstanding (forward -> backward) per ShardedTables."""
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
        return emb, ("exact", inv, sc, rc, recv)

    def _forward_saved_fixed(self, ids):
        """Training forward on the fixed-capacity pipeline (never de-duplicated: the backward needs one slab position per entry).
        The overflow verdict is read here (the one host wait of a training lookup; it covers work that is long done when the
        dense part of the model has been enqueued in between)."""
        B, F = ids.shape
        out = torch.empty((B, F * self.K), dtype=torch.float32, device=ids.device)
        plan = self._plan(B, "train")
        done = self._enqueue(plan, ids, False, out, None, dedup=False)
        lk = _Lookup(self, plan, ids, False, out, None, done)
        lk.join()
        if done is not False:
            over, demand = self._read_flags(plan, done)
            self._learn(plan, over, demand)
            if over:
                self.stats["fallbacks"] += 1
                return self._forward_saved(ids)
        return out, ("fixed", plan)

    def _backward_apply(self, saved, g_emb):
        K = self.K
        self._updates += 1
        if saved[0] == "fixed":
            plan = saved[1]
            P, cap = self.P, plan.cap
            C = plan.C
            g2 = g_emb.contiguous()
            works, grecv_l = [], []
            for c, (s, e) in enumerate(plan.bounds):
                # entry i's gradient row goes to slab position inv[i] (row P*cap = a dump row for pruned entries); owners receive
                # the slabs through the forward row exchange reversed (equal splits again)
                gsend = torch.zeros((P * cap + 1, K), dtype=torch.float32, device=g2.device)
                if e > s:
                    inv = plan.inv[c]
                    idx = torch.where(inv < 0, torch.full_like(inv, P * cap), inv)
                    gsend.index_copy_(0, idx, g2[s:e].reshape(-1, K))
                grecv = torch.empty((P * cap, K), dtype=torch.float32, device=g2.device)
                if self._collective():
                    self._a2a(grecv.view(-1), gsend[:P * cap].reshape(-1), None, None)
                else:
                    grecv = gsend[:P * cap]
                grecv_l.append(grecv)
            # ONE update over all micro-batches (duplicates of a row -- from any chunk, any rank -- are summed before the accumulator
            # moves: a synchronous step over the global batch)
            slabs = plan.recv_all.view(C * P, cap + 1)
            hdr = slabs[:, 0] & 0xffffffff
            pos = torch.arange(cap, device=slabs.device)
            pay = torch.where(pos.unsqueeze(0) < hdr.unsqueeze(1), slabs[:, 1:], torch.full_like(slabs[:, 1:], -1)).reshape(-1)
            self.backend.apply_adagrad(self.optimizer, pay, grecv_l[0] if C == 1 else torch.cat(grecv_l, dim=0))
            return
        _, inv, sc, rc, recv = saved
        n = inv.numel()
        g = g_emb.contiguous().view(n, K)
        gsend = torch.empty_like(g)
        gsend[inv] = g                                   # entry e's row sits at position inv[e] of the exchange order
        grecv = torch.empty((recv.numel(), K), dtype=torch.float32, device=g.device)
        self._a2a(grecv.view(-1), gsend.view(-1), [c * K for c in rc], [c * K for c in sc])   # the forward exchange, reversed
        self.backend.apply_adagrad(self.optimizer, recv, grecv)

    # ---- the exact, variable-size lookup (one host read of the split sizes) --------------------------------
    def _lookup_exact(self, ids, want_fm, out=None, fm=None, consumer=None):
        B, F = ids.shape
        K, be = self.K, self.backend
        flat = ids.reshape(-1).contiguous()
        n = flat.numel()
        payload, inv, send_counts, _ = be.bucket(flat)                      # HIP
        recv_counts = torch.empty_like(send_counts)
        self._a2a(recv_counts, send_counts, None, None)
        both = torch.stack([send_counts, recv_counts]).tolist()             # the host read of this path
        sc, rc = [int(v) for v in both[0]], [int(v) for v in both[1]]
        recv = torch.empty(sum(rc), dtype=torch.int64, device=flat.device)
        self._a2a(recv, payload, rc, sc)
        rows = be.gather_packed(recv)                                       # HIP (owner side)
        back = be.back_buffer(n, K, flat.device)
        self._a2a(back.view(-1), rows.reshape(-1), [c * K for c in sc], [c * K for c in rc])
        if consumer is not None:                                            # (lookup_consume: the caller's kernel reads the rows where they are)
            consumer(0, B, back, inv.view(B, F))
            return None, None
        return be.finish(back, inv, B, F, want_fm, out=out, fm=fm)          # HIP: un-permute (+ FM)

    # ---- the fixed-capacity, pipelined lookup ---------------------------------------------------------------
    def _default_cap(self, n_chunk):
        P = self.P
        if self.slack is not None:
            return _round_up(max(1, math.ceil(float(self.slack) * n_chunk / P)), 16)
        p = 1.0 / P
        return min(_round_up(n_chunk * p + 8.0 * math.sqrt(n_chunk * p * (1 - p)) + 64, 16), _round_up(max(n_chunk, 16), 16))

    def _agree_cap(self, B):
        """The ONE agreement the equal-split exchanges need, at the first fixed-capacity lookup (every rank's first lookup: SPMD):
        slab capacity = MAX over the ranks of the default for their batch (or max_batch).  Later changes (_learn) are functions of
        numbers every rank reads identically off the slab headers, so no rank ever decides alone."""
        Bm = max(int(B), int(self.max_batch or 0))
        n_chunk = max(1, -(-Bm // self._C())) * self.F
        cap = self._default_cap(n_chunk)
        t = torch.tensor([cap, n_chunk], dtype=torch.int64)
        if dist.get_backend(self.group) == "gloo":
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        else:
            td = t.to(self.device)
            dist.all_reduce(td, op=dist.ReduceOp.MAX, group=self.group)
            t = td.cpu()
        self._cap = self._cap_train = self._cap0 = int(t[0])
        self._share0 = float(t[1]) / self.P               # an owner's share of a micro-batch without skew

    def _plan(self, B, slot=0):
        if self._collective():
            if self._cap is None:
                self._agree_cap(B)
            # two capacities: the training pipeline is never de-duplicated (one slab position per entry), so what de-duplicated
            # inference lookups learn (a SMALLER capacity) must not shrink its slabs -- sharing one number made every training lookup
            # after an inference lookup overflow and run twice
            cap = self._cap_train if slot == "train" else self._cap
        else:
            cap = _round_up(max(B * self.F, 16), 16)      # one rank: a slab holds the whole batch, nothing can overflow
        key = (B, slot, cap)
        plan = self._plans.get(key)
        if plan is None:
            def live(k):      # plans of other (batch, slot) pairs whose slabs still have the capacity in use for their kind
                if (k[0], k[1]) == (B, slot):
                    return False
                return not self._collective() or k[2] == (self._cap_train if k[1] == "train" else self._cap)
            self._plans = {k: v for k, v in self._plans.items() if live(k)}
            if len(self._plans) > 8:
                self._plans.clear()
            plan = self._plans[key] = _Plan(self, B, cap)
            plan.train = slot == "train"
        self.stats["cap"] = cap
        return plan

    def _ensure_streams(self):
        if self._streams is None and self.device.type == "cuda":
            if self.side_cus:
                self._streams = [_masked_stream(self.device, self.side_cus, k) for k in range(2)]
            else:
                self._streams = [torch.cuda.Stream(device=self.device), torch.cuda.Stream(device=self.device)]
            self._chk_stream = torch.cuda.Stream(device=self.device)
        return self._streams

    def _enqueue(self, plan, ids, want_fm, out, fm, dedup, consumer=None):
        """Enqueue one lookup's pipeline.  -> False (nothing to check: one rank, slabs hold the whole batch), None (statistic in
        plan.stat, no event: CPU backend) or the event after which plan.host holds [overflow, demand]."""
        B, F = ids.shape
        be, P = self.backend, self.P
        C = plan.C
        cap = plan.cap
        if not self._collective():
            # one rank, no exchange: the three kernels back to back on the caller's stream
            be.bucket_cap(ids, cap, plan.send[0], plan.inv[0], plan.counts[0], plan.flags[0], plan.ws[0], stat=plan.cstat[0], dedup=dedup)
            be.gather_slabs(plan.recv[0], cap, plan.rows[0])
            if B and consumer is not None:
                consumer(0, B, plan.back[0], be.inv2d(plan.inv[0], B, F, dedup))
            elif B:
                be.finish_chunk(plan.back[0], be.inv2d(plan.inv[0], B, F, dedup), want_fm, out, fm if want_fm else None)
            return False
        S = self._ensure_streams()
        cur = torch.cuda.current_stream(self.device) if S else None
        if S:
            for s in S:
                s.wait_stream(cur)

        def on(c):
            return torch.cuda.stream(S[c % 2]) if S else contextlib.nullcontext()

        def bucket(c):
            s, e = plan.bounds[c]
            be.bucket_cap(ids[s:e], cap, plan.send[c], plan.inv[c], plan.counts[c], plan.flags[c], plan.ws[c], stat=plan.cstat[c], dedup=dedup)
            return self._a2a_equal(plan.recv[c], plan.send[c])

        def wait(w):
            if w is not None:
                w.wait()

        def finish(c):
            s, e = plan.bounds[c]
            if e > s and consumer is not None:
                consumer(s, e, plan.back[c], be.inv2d(plan.inv[c], e - s, F, dedup))
            elif e > s:
                be.finish_chunk(plan.back[c], be.inv2d(plan.inv[c], e - s, F, dedup), want_fm, out[s:e], fm[s:e] if want_fm else None)

        wi, wr = [None] * C, [None] * C
        ev_ids = []
        with on(0):
            wi[0] = bucket(0)
        for c in range(C):
            if c + 1 < C:
                with on(c + 1):
                    wi[c + 1] = bucket(c + 1)
            with on(c):
                wait(wi[c])
                if S:
                    ev_ids.append(S[c % 2].record_event())
                be.gather_slabs(plan.recv[c], cap, plan.rows[c])
                wr[c] = self._a2a_equal(plan.back[c], plan.rows[c])
            if c >= 1:
                with on(c - 1):
                    wait(wr[c - 1])
                    finish(c - 1)
        # every micro-batch's slabs have arrived: their headers carry every sender's demand -> the verdict all ranks agree on
        done = None
        ctx = torch.cuda.stream(self._chk_stream) if S else contextlib.nullcontext()
        with ctx:
            if S:
                for e in ev_ids:
                    self._chk_stream.wait_event(e)
            be.slab_stat(plan.recv_all, C * P, cap, plan.stat)
            if S:
                plan.host.copy_(plan.stat, non_blocking=True)
                done = self._chk_stream.record_event()
        with on(C - 1):
            wait(wr[C - 1])
            finish(C - 1)
        plan.fin = [s.record_event() for s in S] if S else None
        return done

    def _read_flags(self, plan, done):
        """The overflow verdict and the largest demand of a lookup, identical on every rank.  On the GPU the header scan and its copy
        to pinned memory ran on their own stream right behind the LAST id exchange: the host waits for work that is long done while
        the row exchanges and finish kernels keep the GPU busy."""
        if done is not None:
            done.synchronize()
            host = plan.host
        else:
            host = plan.stat.cpu()
        return bool(host[0]), int(host[1])

    def _learn(self, plan, over, demand):
        """Capacity policy after a checked lookup (inputs identical on every rank): grow to the observed demand after an overflow;
        with dedup, shrink to what de-duplication left; give up on fixed slabs (mode auto) when one owner wants more than twice
        the no-skew share -- padding every slab to that size would cost more link bytes than the exact path's host read."""
        train = getattr(plan, "train", False)
        if plan.cap != (self._cap_train if train else self._cap):
            return                                         # a verdict about slabs that are no longer in use
        cap = plan.cap
        if over:
            cap = max(cap, _round_up(demand * 1.25 + 64, 16))
        elif self.dedup and not train:                     # only verdicts of de-duplicated lookups say what de-duplication left
            want = _round_up(demand * 1.25 + 64, 16)
            if want < 0.7 * cap:
                cap = want
        if train:
            self._cap_train = cap
        else:
            self._cap = cap
        if self.mode == "auto" and demand > 2.0 * self._share0 + 16:
            self._use_exact = True

    def _drain_unchecked(self, block):
        keep = []
        for lk in self._unchecked:
            if lk.checked:
                continue
            if not block and lk.done is not None and not lk.done.query():
                keep.append(lk)
                continue
            lk.checked = True
            over, demand = self._read_flags(lk.plan, lk.done)
            self._learn(lk.plan, over, demand)
            if over:
                self._unchecked = keep
                raise RuntimeError("ShardedTables: a slab of an earlier fixed-capacity lookup overflowed (demand %d > capacity %d): "
                                   "that result was incomplete" % (demand, lk.plan.cap))
        self._unchecked = keep

    def check_overflow(self):
        """check="lazy"/"never": read the verdicts of the fixed-capacity lookups not checked yet; raises if a slab overflowed (that
        result was incomplete: repeat it with mode="exact" or a larger slack)."""
        self._drain_unchecked(block=True)
        return False

    def lookup_async(self, ids, want_fm=False, out=None, fm=None, consumer=None):
        """Enqueue a lookup and return a handle; handle.result() -> emb [B_local, F*K] (or (emb, fm)).  Two lookups can be in
        flight (double-buffered plans): issue lookup i+1, then consume lookup i -- the exchange of i+1 runs under that compute.
        consumer: see lookup_consume (no emb / fm are produced then)."""
        B, F = ids.shape
        if F != self.F:
            raise ValueError("ids must be [B, F=%d]" % self.F)
        self.stats["lookups"] += 1
        if self._use_exact or (B == 0 and not self._collective()):
            emb, fmo = self._lookup_exact(ids, want_fm, out=out, fm=fm, consumer=consumer)
            return _Lookup(self, None, ids, want_fm, emb, fmo, None, exact=(emb, fmo) if want_fm else (emb if consumer is None else ()), consumer=consumer)
        if out is None and consumer is None:
            out = torch.empty((B, F * self.K), dtype=torch.float32, device=ids.device)
        if want_fm and fm is None:
            fm = torch.empty((B, 1), dtype=torch.float32, device=ids.device)
        slot = self._slot
        self._slot ^= 1
        prev = self._inflight.get(slot)
        if prev is not None:
            prev.join()                                   # its buffers are about to be reused
        plan = self._plan(B, slot)
        done = self._enqueue(plan, ids, want_fm, out, fm, dedup=self.dedup, consumer=consumer)
        lk = _Lookup(self, plan, ids, want_fm, out, fm, done, consumer=consumer)
        if done is False:
            lk.joined = True                              # ran on the caller's stream
        self._inflight[slot] = lk
        if not lk.checked and self.check != "eager":
            # lazy: the verdicts of the EARLIER lookups are read now, after this one has been enqueued (their header scans finished
            # long ago: the host does not wait and the streams never drain); never: they wait for check_overflow()
            if self.check == "lazy":
                self._drain_unchecked(block=True)
            else:
                self._unchecked = self._unchecked[-1:]    # "never": a plan slot's verdict is overwritten two lookups later -- keep the
            self._unchecked.append(lk)                   # two newest handles only (check_overflow() reads what is still there)
        return lk

    STAGES = ("bucket", "a2a_ids", "owner_gather", "a2a_rows", "finish")

    def stage_times(self, ids, want_fm=True, iters=5):
        """Diagnostic: the fixed-capacity pipeline of ONE lookup with its five stages run back to back on the caller's stream and a
        HIP event between them (the product path, _enqueue, overlaps the chunks on two side streams; here nothing overlaps, so the
        stages add up to MORE than a pipelined lookup costs).  -> ({stage: microseconds}, emb, fm): per stage the sum over the
        lookup's chunks, median over `iters` runs after one warm-up run, on THIS rank (a collective's time includes waiting for the
        slowest peer to arrive).  Every rank must call it (the exchanges are collectives).  The results of the last run are returned
        so that the caller can check them."""
        import time
        B, F = ids.shape
        if F != self.F:
            raise ValueError("ids must be [B, F=%d]" % self.F)
        be, cuda = self.backend, self.device.type == "cuda"
        plan = self._plan(B, "stages")
        cap = plan.cap
        out = torch.empty((B, F * self.K), dtype=torch.float32, device=ids.device)
        fm = torch.empty((B, 1), dtype=torch.float32, device=ids.device) if want_fm else None
        runs = []

        def mark():
            if cuda:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                return e
            return time.perf_counter()

        def wait(w):
            if w is not None:
                w.wait()

        for it in range(iters + 1):
            marks = []
            for c, (s, e) in enumerate(plan.bounds):
                m = [mark()]
                be.bucket_cap(ids[s:e], cap, plan.send[c], plan.inv[c], plan.counts[c], plan.flags[c], plan.ws[c], stat=plan.cstat[c], dedup=self.dedup)
                m.append(mark())
                wait(self._a2a_equal(plan.recv[c], plan.send[c]))
                m.append(mark())
                be.gather_slabs(plan.recv[c], cap, plan.rows[c])
                m.append(mark())
                wait(self._a2a_equal(plan.back[c], plan.rows[c]))
                m.append(mark())
                if e > s:
                    be.finish_chunk(plan.back[c], be.inv2d(plan.inv[c], e - s, F, self.dedup), want_fm, out[s:e], fm[s:e] if want_fm else None)
                m.append(mark())
                marks.append(m)
            if cuda:
                torch.cuda.synchronize(self.device)
            if it:
                runs.append([sum((m[k].elapsed_time(m[k + 1]) * 1e3 if cuda else (m[k + 1] - m[k]) * 1e6) for m in marks) for k in range(5)])
        runs.sort(key=lambda r: sum(r))
        med = runs[len(runs) // 2]
        return dict(zip(self.STAGES, med)), out, fm

    def absmax(self, every=32):
        """The largest |value| over ALL ranks' slices of the tables: the local TableSet.absmax(), then a MAX all-reduce.  WHETHER the
        collective runs is decided from rank-invariant state only -- no cached figure yet, `every` owner-side updates since the last one
        (_backward_apply counts them: every rank applies every step, SPMD), or an ops.invalidate_caches() -- never from a rank-local
        float (the ranks used to compare their re-measured local maxima with the cached ones; sparse updates that left one
        rank's largest row untouched sent that rank past the all-reduce while its peers entered it).  Tables written by any other route
        (a checkpoint loaded into the shards): call ops.invalidate_caches() on every rank."""
        hit = self._absmax_all
        gen = ops._CACHE_GEN[0]
        if hit is not None and hit[0] == gen and (self._updates == hit[1] or (every > 1 and 0 < self._updates - hit[1] < every)):
            return hit[2]
        ts = getattr(self.backend, "ts", None)
        val = float(ts.absmax()) if ts is not None else 0.0
        if self._collective():
            t = torch.tensor([val], dtype=torch.float32, device=self.device)
            if self._host_staged(t):
                t = t.cpu()
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            val = float(t[0])
        self._absmax_all = (gen, self._updates, val)
        return val

    def lookup_consume(self, ids, consumer):
        """The lookup WITHOUT its finish pass (round 5).  consumer(s, e, rows, inv) is called once per micro-batch (on the stream that
        micro-batch's pipeline runs on) with the received rows of local samples [s, e): rows [n, K] fp32 exactly as the exchange
        left them, inv [e - s, F] int64 = the position of (sample, slot)'s row in `rows`, < 0 where the id was pruned / out of range --
        what the finish pass would gather from.  A consumer that looks its input rows up itself -- ops.tower(gather=(rows_as_tables(rows,
        F), inv, ...)): the DeepFM tower kernel with the FM term, or ops.gather_fm -- reads them there, and the [B, F*K] concatenation is
        never written or read: the rank-local passes of a lookup are bucket + owner gather only.
        On a slab overflow (check="eager") the lookup is repeated on the exact path and the consumer called again for the whole batch:
        its outputs must be plain overwrites.  Returns after the caller's stream has been made to wait for the consumer's kernels."""
        if self.check != "eager":
            raise ValueError("lookup_consume needs check='eager' (the overflow repair calls the consumer again)")
        self.lookup_async(ids, want_fm=False, consumer=consumer).result()

    def lookup_rows_async(self, ids):
        """Enqueue a lookup WITHOUT its finish pass and hand the received rows to the caller (round 6: what lookup_consume gives a callback,
        as a handle -- so that the NEXT lookup can be issued before this one's rows are consumed, like lookup_async).  handle.result() makes
        the caller's stream wait for the exchange, applies the overflow policy and returns a list of (s, e, rows, inv) -- one entry per
        non-empty micro-batch, samples [s, e) of the local batch: rows [n, K] fp32 exactly as the exchange left them, inv [e - s, F] int64 =
        the position of (sample, slot)'s row in `rows`, < 0 where the id was pruned / out of range.  A kernel that takes (tables, ids) reads
        them as (rows_as_tables(rows, F), inv); ops.cin_stack_gather and ops.tower(gather=) take them directly: the [B, F*K] concatenation
        is never written.  The buffers belong to the lookup's plan slot: valid until the lookup after the next one is enqueued (two slots).
        Needs check='eager' (an overflow is repaired on the exact path before the rows are handed out: then ONE entry for the whole batch)."""
        if self.check != "eager":
            raise ValueError("lookup_rows needs check='eager' (an overflow is repaired before the rows are handed out)")
        chunks = []

        def collect(s, e, rows, inv):
            if s == 0:
                chunks.clear()                     # (the repair on the exact path calls again, for the whole batch)
            chunks.append((s, e, rows, inv))
        return _RowsLookup(self.lookup_async(ids, want_fm=False, consumer=collect), chunks)

    def lookup_rows(self, ids):
        return self.lookup_rows_async(ids).result()

    def lookup(self, ids, want_fm=False, out=None, fm=None):
        """ids [B_local, F] int64 (global row ids; < 0 or >= vocab_f -> zeros) -> emb [B_local, F*K] fp32
        (and the FM second-order logit [B_local, 1] when want_fm).  out / fm: preallocated results (stable addresses)."""
        return self.lookup_async(ids, want_fm=want_fm, out=out, fm=fm).result()


_MASKED = []      # (hipStream_t, ExternalStream) pairs kept alive for the life of the process


def _masked_stream(device, n_cus, which):
    """A HIP stream whose kernels may only run on n_cus of the 256 compute units (hipExtStreamCreateWithCUMask), wrapped for torch.
    The lookup's kernels and RCCL's run there; the interaction kernels on the caller's stream keep the rest of the chip, so a lookup
    prefetched under a CIN or a tower costs it those CUs for as long as the lookup runs instead of fragmenting every CU's residency.
    The CUs are spread evenly over the mask (one per 256 / n_cus bits: every XCD and shader engine contributes)."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    total = torch.cuda.get_device_properties(device).multi_processor_count
    n = max(1, min(int(n_cus), total))
    step = max(1, total // n)
    words = (total + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    for k in range(n):
        bit = (k * step + which * (step // 2)) % total      # the two side streams take interleaved CUs
        mask[bit // 32] |= 1 << (bit % 32)
    st = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
    if rc != 0 or not st.value:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed (%d)" % rc)
    ext = torch.cuda.ExternalStream(st.value, device=device)
    _MASKED.append((st, ext))
    return ext


class _ShardedLookup(torch.autograd.Function):
    """emb = ShardedTables.lookup(ids) with a backward that routes the row gradients to their owners and lets each owner
    apply the sparse Adagrad update to its shard (no gradient tensor is returned for the tables)."""

    @staticmethod
    def forward(ctx, st, ids, anchor):
        if st._use_exact or ids.shape[0] == 0 and not st._collective():
            emb, saved = st._forward_saved(ids)
        else:
            emb, saved = st._forward_saved_fixed(ids)
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


@torch.no_grad()
def xdeepfm_predict(model, tables, ids, linear_logit=None):
    """Inference logits [B_local, 1] of an xdeepfm.XDeepFM over ROW-SHARDED embedding tables (BASELINE config 5: the config the reference's
    README.md:28 model shards; partitioner precedent deepFM.py:163-167).  With check='eager' the lookup runs WITHOUT its finish pass
    (lookup_rows): per micro-batch the CIN layers and the tower read the received rows through the inverse positions
    (XDeepFM.forward_rows) -- the rank-local passes of the lookup are bucket + owner gather only; otherwise lookup() + forward_embedded().
    Bit for bit the same logits either way wherever a micro-batch and the whole batch take the same dense kernels (those are chosen by row
    count: ops.dense_small_covers / dense_mid_covers / TOWER_MIN_ROWS).  `model`'s own embedding tables are not used; linear_logit [B_local, 1]: the first-order term,
    computed by the caller (its 4-byte rows live wherever the caller keeps them)."""
    amax = tables.absmax()
    B = ids.shape[0]
    was_training = model.training
    model.eval()
    try:
        if tables.check == "eager":
            out = torch.empty((B, 1), dtype=torch.float32, device=ids.device)
            for s, e, rows, inv in tables.lookup_rows(ids):
                if e > s:
                    out[s:e] = model.forward_rows(rows, inv, None if linear_logit is None else linear_logit[s:e], absmax=amax)
            return out
        return model.forward_embedded(tables.lookup(ids), linear_logit, range_ok=ops.f16_range_ok(amax))
    finally:
        model.train(was_training)


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

    @torch.no_grad()
    def predict(self, ids):
        """Inference logits [B_local, 1] of the same FM + DNN model over the row-sharded tables.  Where the one-launch tower kernel covers
        the model (dense.tower_infer's gather form: K = 16, F <= 26, ReLU tower with a units = 1 head) the lookup runs WITHOUT its finish
        pass (lookup_consume): per micro-batch the tower kernel reads the received rows through the inverse positions and folds the FM
        term -- the [B, F*K] concatenation is never written; otherwise lookup(want_fm=True) + dnn_logit_fn.  Bit for bit the same logits."""
        from .dense import tower_infer
        m = self.model
        was_training = m.training
        m.eval()
        try:
            B = ids.shape[0]
            out = torch.empty((B, 1), dtype=torch.float32, device=ids.device)
            amax = self.tables.absmax()
            state = {"fused": True}

            def consumer(s, e, rows, inv):
                lg = tower_infer(m.hidden, None, m.activation, bns=m.bns if len(m.bns) else None, head=m.logits_layer,
                                 gather=(rows_as_tables(rows, m.F, absmax=amax), inv, None, True)) if state["fused"] else None
                if lg is None:                                      # not covered (or too few rows): the finish pass's job, here
                    state["fused"] = False
                    emb, fm = ops.gather_fm(rows_as_tables(rows, m.F), inv)
                    lg = m.dnn_logit_fn(emb, adds=(fm,), range_ok=ops.f16_range_ok(amax))
                out[s:e] = lg
            if m.units == 1 and self.tables.check == "eager":
                self.tables.lookup_consume(ids, consumer)
                return out
            emb, fm = self.tables.lookup(ids, want_fm=True)
            return m.dnn_logit_fn(emb, adds=(fm,), range_ok=ops.f16_range_ok(amax))      # the SHARDED tables' magnitude, not the model's own
        finally:
            m.train(was_training)
