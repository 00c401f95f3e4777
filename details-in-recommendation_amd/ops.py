"""ops.py -- functional ops over torch ROCm tensors; one per reference closure, each a thin call through
the C ABI (include/dir_hip.h) on torch's current HIP stream.

torch is plumbing here: device memory, streams, tensors as buffers.  Every op requires CUDA(ROCm)
tensors and the in-tree libdir_hip.so; there is no CPU or eager fallback (a missing library or a CPU
tensor raises).

Reference closures (relative to /root/reference):
  embedding_bag / TableSet   myself_input_layer            models/DeepFM/deepFM.py:363-400
  fm_logit                   fm_logit_fn                   models/DeepFM/deepFM.py:321-335
  gather_fm                  deepFM.py:169-177 + :321-335 fused (one pass over the rows)
  linear_logit               _linear_logit_fn_builder      models/DeepFM/deepFM.py:255-275
  cross_op / cross_network   _cross_op/_cross_architecture models/DeepCrossNetwork/DeepCrossNetwork.py:336-367
  din_attention_pool         (no reference code; README.md:27)
  cin_layer                  (no reference code; README.md:28)
"""
import ctypes
import os

import torch

from . import _lib

SUM, MEAN, SQRTN = 0, 1, 2
_COMBINERS = {"sum": SUM, "mean": MEAN, "sqrtn": SQRTN, SUM: SUM, MEAN: MEAN, SQRTN: SQRTN}
PRUNE_NONPOSITIVE_WEIGHTS = 1
STREAM_ROWS = 2  # DIR_GATHER_STREAM_ROWS: non-temporal row loads (uniform ids over tables >> Infinity Cache)
INFINITY_CACHE_BYTES = 256 << 20


def _stream():
    """The current HIP stream of the current device as the C ABI takes it.  (torch.cuda.current_stream() builds a Stream object per call:
    5.7 us, 30 times per DeepFM training step; the raw query is 0.3 us.)"""
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _dev(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("%s must be a CUDA (ROCm) tensor: the HIP path has no CPU fallback" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


class TableSet:
    """F embedding tables [vocab_f, K] plus the device array of their base pointers (built once; the
    kernels read table f's base with a scalar load).  Mirrors the per-column `embedding_weights`
    variables created under myself_input_layer (deepFM.py:386-390)."""

    def __init__(self, tables, ld=None):
        tables = list(tables)
        if not tables:
            raise ValueError("empty columns.")  # deepFM.py:104-105
        K = tables[0].shape[-1] if tables[0].dim() == 2 else 1
        for i, t in enumerate(tables):
            _dev(t, torch.float32, "table %d" % i)
            if ld is None and not t.is_contiguous():
                raise ValueError("table %d must be contiguous" % i)
            if ld is not None and (t.dim() != 2 or tuple(t.stride()) != (ld, 1)):
                raise ValueError("table %d must be a [vocab, K] view with row stride %d" % (i, ld))
            k = t.shape[-1] if t.dim() == 2 else 1
            if k != K:
                raise ValueError("all tables of one TableSet share K (got %d and %d)" % (K, k))
            if (K * 4) % 16 == 0 and t.data_ptr() % 16:
                raise ValueError("table %d is not 16-byte aligned" % i)
        self.tables = tables
        self.F = len(tables)
        self.K = K
        self.vocab = [int(t.shape[0]) for t in tables]
        self.device = tables[0].device
        self.ld = K if ld is None else int(ld)   # floats between rows (> K: packed training rows, see train_rows)
        self.accums = None                       # train_rows: the Adagrad accumulators living in the same rows
        self._ptrs = torch.tensor([t.data_ptr() for t in tables], dtype=torch.int64, device=self.device)
        self.vocab_dev = torch.tensor(self.vocab, dtype=torch.int64, device=self.device)
        # row-read policy of the one-hot gather: "stream" = DIR_GATHER_STREAM_ROWS, "reuse" = cacheable,
        # "auto" = stream iff the tables cannot live in the 256 MiB Infinity Cache anyway.  Callers who
        # know their ids are heavily skewed should set "reuse" (see include/dir_hip.h).
        self.row_policy = "auto"
        self.nbytes = sum(int(t.shape[0]) * self.ld * 4 for t in tables)
        self.owners = []        # tensors sharing these tables' storage under ANOTHER version counter (the nn.Parameters whose .data the
        #                         tables are): written() bumps them too, so caches keyed on parameter._version see HIP updates
        self.grad_sink = None   # callable(ids, d_rows) consuming the gather's row gradients (see SparseAdagrad.attach)
        self.fm_sink = None     # callable(ids, d_rows | None, d_fm, field_sums): the same with the FM backward folded in

    @property
    def ptrs(self):
        """Device array of the table base pointers, for the kernels that read [vocab, K] rows K floats apart."""
        if self.ld != self.K:
            raise ValueError("this TableSet holds packed training rows (row stride %d, K = %d): only gather_fm and the "
                             "SparseAdagrad update read that layout" % (self.ld, self.K))
        return self._ptrs

    @classmethod
    def train_rows(cls, tables, initial_accumulator_value=0.1):
        """Packed TRAINING layout (include/dir_hip.h: dir_gather_fm_rows_f32): every slot's table becomes [vocab, 2K] rows
        = [embedding | Adagrad accumulator] in one arena (K = 16: one 128-byte line per row and its optimiser state).
        .tables are [vocab, K] VIEWS of it (row stride 2K) initialised from `tables`, .accums the accumulator views."""
        tables = list(tables)
        K = int(tables[0].shape[1])
        ld = 2 * K
        vocab = [int(t.shape[0]) for t in tables]
        dev = tables[0].device
        arena = torch.empty(sum(vocab) * ld + 32, dtype=torch.float32, device=dev)
        off = (-(arena.data_ptr() // 4)) % 32              # rows 128-byte aligned
        rows = []
        for t, v in zip(tables, vocab):
            blk = arena[off:off + v * ld].view(v, ld)
            blk[:, :K] = t
            blk[:, K:] = initial_accumulator_value
            rows.append(blk)
            off += v * ld
        ts = cls([r[:, :K] for r in rows], ld=ld)
        ts.accums = [r[:, K:] for r in rows]
        ts.arena, ts.rows = arena, rows
        return ts

    @classmethod
    def ftrl_rows(cls, weights, initial_accumulator_value=0.1):
        """Packed linear TRAINING layout (include/dir_hip.h: dir_linear_onehot_rows_f32 / dir_sparse_ftrl_rows_sorted_f32): every
        column's first-order weights become [vocab, 4] rows = [w | n | z | unused] in one arena -- a touched id's FTRL update reads and
        writes one 16-byte row instead of one element of three arrays.  .tables are [vocab, 1] VIEWS (row stride 4) initialised from
        `weights` ([vocab] or [vocab, 1]), .accums / .linears the n / z views (n = initial_accumulator_value, z = 0)."""
        weights = [w.reshape(-1, 1) for w in weights]
        vocab = [int(w.shape[0]) for w in weights]
        dev = weights[0].device
        arena = torch.zeros(sum(vocab) * 4 + 4, dtype=torch.float32, device=dev)
        off = (-(arena.data_ptr() // 4)) % 4               # rows 16-byte aligned
        rows = []
        for w, v in zip(weights, vocab):
            blk = arena[off:off + v * 4].view(v, 4)
            blk[:, 0:1] = w
            blk[:, 1] = initial_accumulator_value
            rows.append(blk)
            off += v * 4
        ts = cls([r[:, 0:1] for r in rows], ld=4)
        ts.accums = [r[:, 1:2] for r in rows]
        ts.linears = [r[:, 2:3] for r in rows]
        ts.arena, ts.rows = arena, rows
        return ts

    def absmax(self, every=1):
        """max |embedding value| over the tables, measured once per version of the tables and their owners (one pass + one sync when an
        in-place update bumped a counter): tower(gather=..., split=None) keeps a table past F16_RANGE_GUARD off the fp16 x 2 kernel.
        every > 1: a measurement may be up to `every` OPTIMISER STEPS old -- version bumps that the fused updaters made through
        mark_written (the ledger hip_bumps counts them): what a training loop that evaluates between steps can afford, one pass and one
        sync per `every` steps.  Any other version change (load_state_dict, a copy_, a torch optimiser) re-measures at once, and so does
        invalidate_caches() (a checkpoint loaded after one eval forward used to keep the pre-load figure for 32 x F bumps)."""
        watched = list(self.tables) + list(self.owners)
        sig = tuple(t._version for t in watched)
        hit = getattr(self, "_absmax", None)
        stale = hit is None or hit[3] != _CACHE_GEN[0] or len(hit[0]) != len(sig)
        if not stale and hit[0] != sig:
            bumps = tuple(hip_bumps(t) for t in watched)
            moved = [v - v0 for v, v0 in zip(sig, hit[0])]
            stale = every <= 1 or any(d != b - b0 or d < 0 or d >= every for d, b, b0 in zip(moved, bumps, hit[2]))
        if stale:
            if self.device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                if hit is not None:
                    return hit[1]                  # (a sync cannot be captured: the last measurement stands)
                raise RuntimeError("TableSet.absmax: measure the tables (one eager call) before capturing a graph")
            m = float(torch.stack([t.abs().max() for t in self.tables if t.numel()] or [torch.zeros((), device=self.device)]).max())
            hit = self._absmax = (sig, m, tuple(hip_bumps(t) for t in watched), _CACHE_GEN[0])
        return hit[1]

    def range_ok(self, every=32):
        """Whether rows of these tables may feed an UNSCALED fp16 x 2 kernel (the fused inference tower, the DIN unit): the largest |value|
        inside [F16_SMALL_GUARD, F16_RANGE_GUARD) -- below the upper guard nothing overflows fp16 (a factor 2 of headroom: a table that
        drifts between two measurements `every` steps apart must more than double to reach 65 504), above the lower one a 1e-5 relative
        bar holds (an fp16 x 2 element below 2^-3 keeps an ABSOLUTE 2^-25).  Callers fall back to bf16 x 3 (fp32's exponent range)."""
        return f16_range_ok(self.absmax(every))

    def gather_flags(self):
        if self.row_policy == "stream" or (self.row_policy == "auto" and self.nbytes > 2 * INFINITY_CACHE_BYTES):
            return STREAM_ROWS
        return 0

    def written(self, *more):
        """A raw-pointer kernel has just updated the tables (and `more`: optimiser slots) in place: see mark_written."""
        arena = getattr(self, "arena", None)
        if arena is not None:
            mark_written(arena, *self.owners, *more)
        else:
            mark_written(*self.tables, *self.owners, *more)

    def refresh(self):
        """Rebuild the pointer array (after tables were re-allocated, e.g. .to())."""
        self._ptrs = torch.tensor([t.data_ptr() for t in self.tables], dtype=torch.int64, device=self.device)



def f16_range_ok(absmax):
    """The magnitude window of the UNSCALED fp16 x 2 kernels (see TableSet.range_ok); an all-zero operand is fine."""
    return absmax == 0.0 or (F16_SMALL_GUARD <= absmax < F16_RANGE_GUARD)


_CACHE_GEN = [0]                 # bumped by invalidate_caches(): every magnitude measurement taken before it is stale
_HIP_BUMPS = None                # version-counter owner (a view's base) -> how many of its version bumps mark_written made


def _vc_owner(t):
    return t._base if t._base is not None else t


def hip_bumps(t):
    """How many of tensor t's version bumps came from mark_written (a fused updater's step), not from torch ops: the magnitude caches
    (TableSet.absmax / weight_absmax with every > 1) tolerate only those."""
    return 0 if _HIP_BUMPS is None else _HIP_BUMPS.get(_vc_owner(t), 0)


def mark_written(*tensors):
    """A raw-pointer kernel has just written these tensors in place: bump their autograd version counters, as an in-place torch op would
    have.  Everything cached per weight version (DeepFM's packed serving rows, tower_image / dense_bf3_image, dense._packed_cached,
    dense._bn_affine, DCN's cross image) keys on tensor._version, so an eval -> train -> eval loop in one process (the reference's
    train_and_evaluate) rebuilds them after a HIP optimiser step.  Views share their base's counter."""
    global _HIP_BUMPS
    ts = [t for t in tensors if isinstance(t, torch.Tensor)]
    if ts:
        torch._C._autograd._unsafe_set_version_counter(ts, [t._version + 1 for t in ts])
        if _HIP_BUMPS is None:
            from torch.utils.weak import WeakIdKeyDictionary
            _HIP_BUMPS = WeakIdKeyDictionary()
        seen = set()
        for t in ts:
            o = _vc_owner(t)
            if id(o) not in seen:
                seen.add(id(o))
                _HIP_BUMPS[o] = _HIP_BUMPS.get(o, 0) + 1

_FROZEN_WEIGHTS = [False]


class frozen_weights:
    """with ops.frozen_weights(): ... capture ...  -- a capture taken inside takes the VALID per-version cache entries (weight images, packed
    weights, folded batch norms: built by an earlier eager call on the same versions) instead of re-packing inside the graph.  For SERVING
    graphs whose weights do not change between replays: a one-launch DeepFM forward at 1 024-4 096 rows carries three pack launches
    otherwise (15-20 us of ~70).  The price is the rule CapturedStep keeps: a replay after an in-place weight update still runs the images
    of capture time -- capture again after loading new weights.  serving.GraphedForward(..., frozen_weights=True) uses it."""

    def __enter__(self):
        self._old = _FROZEN_WEIGHTS[0]
        _FROZEN_WEIGHTS[0] = True
        return self

    def __exit__(self, *exc):
        _FROZEN_WEIGHTS[0] = self._old
        return False


def capture_bypasses_caches(t=None):
    """True while the current stream is capturing and the per-version caches must not be read (the default rule; see frozen_weights)."""
    return (t is None or t.is_cuda) and torch.cuda.is_current_stream_capturing() and not _FROZEN_WEIGHTS[0]


class CapturedStep:
    """A training (or inference) step captured in a HIP graph, replayed with the bookkeeping eager steps do on the host.

        step = ops.CapturedStep(fn, written=ops.written_of(model))      # warm-up calls, then ONE capture of fn()
        step.replay()                                                    # graph.replay() + mark_written(*written)

    A replay runs the captured optimiser / batch-norm kernels, which write parameters, tables and moving statistics through raw pointers:
    nothing on the host moves their version counters, so everything cached per version (serving rows, weight images, folded batch norms)
    would stay stale for an eager eval after the replays.  replay() bumps the counters of `written` after every replay, exactly as the
    eager updaters do (mark_written).  Inside the capture every per-version cache is bypassed (dense_bf3_image, tower_image, the modules'
    packed weights): the pack kernels are part of the graph and read the weights as they are at replay time.  What a captured step must
    not do: anything that syncs (a first-time range measurement: run one eager step before capturing -- the constructor does)."""

    def __init__(self, fn, written=(), warmup=3, pool=None):
        self.fn, self.written = fn, [t for t in written if isinstance(t, torch.Tensor)]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # the standard recipe: warm up on a side stream (workspaces, caches, pointer tables exist)
            for _ in range(max(1, warmup)):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        mark_written(*self.written)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, pool=pool):
            self.out = fn()
        mark_written(*self.written)                        # (the capture itself does not run the kernels; counters move like a step's anyway)

    def replay(self):
        self.graph.replay()
        mark_written(*self.written)
        return self.out


def written_of(*modules_or_tensors):
    """Every parameter and buffer of the given modules (and the given tensors): what a captured training step may write."""
    out = []
    for m in modules_or_tensors:
        if isinstance(m, torch.Tensor):
            out.append(m)
        else:
            out += list(m.parameters()) + list(m.buffers())
    return out


def refuse_rebuild_under_sink(*tablesets):
    """The modules rebuild their TableSets when the parameters' storage moved (.to() / .cuda() / a re-pointed .data).  A fused optimiser
    attached to the old set (grad_sink / fm_sink) holds that set's pointers and its own slots on the old storage: dropping it silently
    would leave those tables without any optimiser (they get no .grad either).  Refuse instead."""
    for ts in tablesets:
        if ts is not None and (ts.grad_sink is not None or ts.fm_sink is not None):
            raise RuntimeError("the embedding tables moved (model.to() / .cuda() / new storage) after a fused sparse optimiser was attached to "
                               "them: move the model first, then call fused_sparse_* / TrainStep")

def plain_list(plist):
    """The parameters of an nn.ParameterList as a plain Python list: iterating a ParameterList resolves every element through
    __getitem__ / _get_abs_string_index -- 52 tables cost a forward ~0.15 ms of host time where the per-step key over their data_ptr()s
    is built (DeepFM._tablesets); the module's own parameter dict holds them in order (always current: replaced elements keep their slot)."""
    params = getattr(plist, "_parameters", None)
    if params is not None and len(params) == len(plist):
        return list(params.values())
    return list(plist)


def _as_tableset(tables):
    return tables if isinstance(tables, TableSet) else TableSet(tables)


def _onehot_strides(ids, F):
    """ids [B,F] (any strides) or [F,B] given as transposed view -> (B, stride_b, stride_f)."""
    if ids.dim() != 2 or ids.shape[1] != F:
        raise ValueError("one-hot ids must be [B, F=%d], got %s" % (F, tuple(ids.shape)))
    return ids.shape[0], ids.stride(0), ids.stride(1)


def slot_combiners(ts, combiner):
    """`combiner` as the C ABI takes it: (slot_combiner device int32 [F] | None, combiner code).  A str / code applies to
    every slot; a sequence gives one combiner per slot (every embedding_column carries its own, DeepCrossNetwork/train.py:99)."""
    if isinstance(combiner, (str, int)):
        return None, _COMBINERS[combiner]
    codes = [_COMBINERS[c] for c in combiner]
    if len(codes) != ts.F:
        raise ValueError("one combiner per slot: got %d for F=%d" % (len(codes), ts.F))
    if all(c == codes[0] for c in codes):
        return None, codes[0]
    key = tuple(codes)
    cache = ts.__dict__.setdefault("_slot_combiner_cache", {})
    if key not in cache:
        cache[key] = torch.tensor(codes, dtype=torch.int32, device=ts.device)
    return cache[key], codes[0]


def slot_max_norms(ts, max_norm):
    """`max_norm` as the C ABI takes it: (slot_max_norm device fp32 [F] | None, max_norm).  None / a number applies to every slot; a
    sequence gives one value per slot (None / 0: that slot is not clipped)."""
    if max_norm is None or isinstance(max_norm, (int, float)):
        return None, float(max_norm or 0.0)
    vals = [float(m or 0.0) for m in max_norm]
    if len(vals) != ts.F:
        raise ValueError("one max_norm per slot: got %d for F=%d" % (len(vals), ts.F))
    if any(v < 0 for v in vals):
        raise ValueError("max_norm must be >= 0")
    if all(v == vals[0] for v in vals):
        return None, vals[0]
    key = tuple(vals)
    cache = ts.__dict__.setdefault("_slot_max_norm_cache", {})
    if key not in cache:
        cache[key] = torch.tensor(vals, dtype=torch.float32, device=ts.device)
    return cache[key], 0.0


def _leave_hint(t, name, *bits):
    """Attach the maxima a kernel left beside tensor t as t.<name> = (*bits, t._version): valid while t is unmodified.  Tensors created under
    torch.inference_mode() track no version counter (reading ._version raises), so nothing could tell a later in-place write: no hint is
    left on them and their consumers run their own max pass (DCN / ESMM / xDeepFM inference under inference_mode raised)."""
    if not t.is_inference():
        setattr(t, name, (*bits, t._version))


def _bits_ws(device):
    """The ticket / block-maxima workspace of the kernels that leave a tensor maximum (row_absmax_bits, the bits gathers): one per
    (device, stream) -- calls on one stream are ordered."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _ABSMAX_WS.get(key)
    if ws is None:
        ws = _ABSMAX_WS[key] = torch.zeros(int(_lib.load().dir_row_absmax_workspace_words()), dtype=torch.int32, device=device)
    return ws


def embedding_bag(tables, ids, offsets=None, weights=None, combiner="mean", field_major=False, flags=0,
                  out=None, max_norm=None, want_bits=False):
    """Multi-slot embedding bag -> [B, F*K] (slot order).

    one-hot : ids LongTensor [B, F] (arbitrary strides, e.g. torch.stack(per_field).t()).
    multi-hot: ids [nnz], offsets [B*F+1]; bag(b,f) = b*F+f, or f*B+b when field_major (the layout
               that concatenating per-column CSR inputs gives); weights optional [nnz].
    combiner: one name for all slots or a sequence of F names; max_norm: [TF-upstream] embedding_column(max_norm=), one value or a
    sequence of F values (None / 0: no clipping for that slot).
    ids outside [0, vocab_f) contribute nothing (id < 0: pruned as in the reference; id >= vocab_f: zeros as on TF GPU).
    """
    ts = _as_tableset(tables)
    _dev(ids, torch.int64, "ids")
    lib = _lib.load()
    F, K = ts.F, ts.K
    if offsets is None:
        B, sb, sf = _onehot_strides(ids, F)
    else:
        _dev(offsets, torch.int64, "offsets")
        if not (ids.is_contiguous() and offsets.is_contiguous()):
            raise ValueError("multi-hot ids/offsets must be contiguous")
        B = (offsets.numel() - 1) // F
        if offsets.numel() != B * F + 1:
            raise ValueError("offsets must have B*F+1 entries")
        sb, sf = (1, B) if field_major else (F, 1)
        if weights is not None:
            _dev(weights, torch.float32, "weights")
    if out is None:
        out = torch.empty((B, F * K), dtype=torch.float32, device=ts.device)
    _dev(out, torch.float32, "out")
    if offsets is None:
        flags |= ts.gather_flags()
    slot_comb, comb = slot_combiners(ts, combiner)
    slot_mn, mn = slot_max_norms(ts, max_norm)
    if (want_bits and offsets is None and slot_mn is None and not mn and K % 4 == 0 and B > 0 and out.stride(0) % 4 == 0
            and out.data_ptr() % 16 == 0 and out.stride(1) == 1):
        # one-hot lookups whose output feeds a row-scaled fp16 x 2 layer (an input layer made of embedding columns: the ESMM towers): the
        # rows kernel copies the same rows and leaves the output's row / tensor maxima on it (dir_gather_fm_rows_bits_f32) -- no
        # dir_row_absmax_bits_f32 pass over the [B, F*K] output in front of the first dense layer
        buf = torch.empty(B + 4, dtype=torch.int32, device=ts.device)
        rb, ab = buf[:B], buf[B:B + 1]
        _lib.check(lib.dir_gather_fm_rows_bits_f32(_ptr(ts._ptrs), _ptr(ts.vocab_dev), F, K, ts.ld, _ptr(ids), sb, sf, flags, B, _ptr(out),
                                                   out.stride(0), None, None, _ptr(rb), _ptr(ab), _ptr(_bits_ws(ts.device)), _stream()))
        _leave_hint(out, '_dir_bits', rb, ab)
        return out
    _lib.check(lib.dir_embedding_bag_ex2_f32(_ptr(ts.ptrs), _ptr(ts.vocab_dev), F, K, _ptr(ids), _ptr(offsets), _ptr(weights), sb, sf,
                                             _ptr(slot_comb), comb, _ptr(slot_mn), mn, flags, B, _ptr(out), out.stride(0), _stream()))
    return out


def check_ids(tables, ids, offsets=None, field_major=False):
    """Debug aid: raise DIR_E_RANGE if any id >= vocab ([TF-upstream] CPU lookups raise InvalidArgument)."""
    ts = _as_tableset(tables)
    lib = _lib.load()
    if offsets is None:
        B, sb, sf = _onehot_strides(ids, ts.F)
    else:
        B = (offsets.numel() - 1) // ts.F
        sb, sf = (1, B) if field_major else (ts.F, 1)
    bad = torch.zeros(1, dtype=torch.int32, device=ts.device)
    _lib.check(lib.dir_check_ids(_ptr(ts.vocab_dev), ts.F, _ptr(ids), _ptr(offsets), sb, sf, B, _ptr(bad), _stream()))
    n = int(bad.item())
    if n:
        raise _lib.DirError(_lib.DIR_E_RANGE, "%d ids are >= their table's vocabulary size" % n)


def fm_logit(emb, F, K, out=None):
    """fm_logit_fn (deepFM.py:321-335): emb [B, F*K] -> [B, 1]."""
    _dev(emb, torch.float32, "emb")
    if emb.dim() != 2 or emb.shape[1] != F * K or emb.stride(1) != 1:
        raise ValueError("emb must be [B, F*K] with unit inner stride")  # deepFM.py:329 reshape contract
    B = emb.shape[0]
    if out is None:
        out = torch.empty((B, 1), dtype=torch.float32, device=emb.device)
    _lib.check(_lib.load().dir_fm_second_order_f32(_ptr(emb), emb.stride(0), B, F, K, _ptr(out), _stream()))
    return out


def gather_fm(tables, ids, want_emb=True, out=None, fm=None, fsum=None, want_bits=False):
    """Fused one-hot gather + FM second-order: returns (emb [B, F*K] or None, fm_logit [B, 1]).  fsum [B, K] (optional
    output): the field sums S[b] = sum_f e[b,f] the FM backward needs (dir_gather_fm_rows_f32; also the kernel that reads
    packed training rows, TableSet.train_rows).  want_bits (with the rows kernel: packed training rows or fsum): the gather also leaves
    emb's row / tensor maxima on it (dir_gather_fm_rows_bits_f32), where the row-scaled fp16 x 2 layer that reads emb finds them
    instead of running dir_row_absmax_bits_f32's pass over emb (36 us of a 1.5 ms DeepFM training step)."""
    ts = _as_tableset(tables)
    _dev(ids, torch.int64, "ids")
    B, sb, sf = _onehot_strides(ids, ts.F)
    if want_emb and out is None:
        out = torch.empty((B, ts.F * ts.K), dtype=torch.float32, device=ts.device)
    if fm is None:
        fm = torch.empty((B, 1), dtype=torch.float32, device=ts.device)
    if ts.ld != ts.K or fsum is not None:
        if fsum is not None and (fsum.shape != (B, ts.K) or not fsum.is_contiguous()):
            raise ValueError("fsum must be a contiguous [B, K] tensor")
        if want_bits and want_emb and B > 0 and ts.K % 4 == 0:
            lib = _lib.load()
            ws = _bits_ws(ts.device)
            buf = torch.empty(B + 4, dtype=torch.int32, device=ts.device)
            rb, ab = buf[:B], buf[B:B + 1]
            _lib.check(lib.dir_gather_fm_rows_bits_f32(_ptr(ts._ptrs), _ptr(ts.vocab_dev), ts.F, ts.K, ts.ld, _ptr(ids), sb, sf,
                                                       ts.gather_flags(), B, _ptr(out), out.stride(0), _ptr(fm), _ptr(fsum), _ptr(rb), _ptr(ab),
                                                       _ptr(ws), _stream()))
            _leave_hint(out, '_dir_bits', rb, ab)
            return out, fm
        _lib.check(_lib.load().dir_gather_fm_rows_f32(_ptr(ts._ptrs), _ptr(ts.vocab_dev), ts.F, ts.K, ts.ld, _ptr(ids), sb, sf,
                                                      ts.gather_flags(), B, _ptr(out) if want_emb else None,
                                                      out.stride(0) if want_emb else 0, _ptr(fm), _ptr(fsum), _stream()))
        return (out if want_emb else None), fm
    _lib.check(_lib.load().dir_gather_fm_fused_f32(_ptr(ts.ptrs), _ptr(ts.vocab_dev), ts.F, ts.K, _ptr(ids), sb, sf, ts.gather_flags(), B,
                                                   _ptr(out) if want_emb else None,
                                                   out.stride(0) if want_emb else 0, _ptr(fm), _stream()))
    return (out if want_emb else None), fm


def linear_logit(weights, ids, offsets=None, entry_weights=None, combiner="sum", bias=None, field_major=False,
                 out=None, accumulate=False):
    """First-order term, units = 1 (deepFM.py:255-275): -> [B, 1]."""
    ts = _as_tableset(weights)
    if ts.K != 1:
        raise ValueError("linear_logit: weights are [vocab] or [vocab, 1] columns (units = 1)")
    _dev(ids, torch.int64, "ids")
    if offsets is None:
        B, sb, sf = _onehot_strides(ids, ts.F)
    else:
        _dev(offsets, torch.int64, "offsets")
        B = (offsets.numel() - 1) // ts.F
        sb, sf = (1, B) if field_major else (ts.F, 1)
    if out is None:
        if accumulate:
            raise ValueError("accumulate needs out")
        out = torch.empty((B, 1), dtype=torch.float32, device=ts.device)
    if bias is not None:
        _dev(bias, torch.float32, "bias")
    if ts.ld != ts.K:                           # packed linear training rows (TableSet.ftrl_rows): the weight is column 0 of a 16-byte row
        if offsets is not None:
            raise ValueError("linear_logit: packed linear training rows serve one-hot columns only")
        _lib.check(_lib.load().dir_linear_onehot_rows_f32(_ptr(ts._ptrs), ts.ld, _ptr(ts.vocab_dev), ts.F, _ptr(ids), sb, sf, _ptr(bias),
                                                          int(accumulate), B, _ptr(out), _stream()))
        return out
    _lib.check(_lib.load().dir_linear_sparse_sum_f32(_ptr(ts.ptrs), _ptr(ts.vocab_dev), ts.F, _ptr(ids), _ptr(offsets), _ptr(entry_weights),
                                                     sb, sf, _COMBINERS[combiner], _ptr(bias), int(accumulate), B,
                                                     _ptr(out), _stream()))
    return out


def cross_network(x0, w, b, out=None):
    """_cross_architecture (DeepCrossNetwork.py:350-367): x0 [B,d], w,b [L,d] -> x_L [B,d]."""
    _dev(x0, torch.float32, "x0")
    _dev(w, torch.float32, "w")
    _dev(b, torch.float32, "b")
    if x0.dim() != 2 or x0.stride(1) != 1:
        raise ValueError("x0 must be [B, d] with unit inner stride")
    B, d = x0.shape
    if w.dim() != 2 or w.shape[1] != d or tuple(b.shape) != tuple(w.shape):
        raise ValueError("w and b must be [L, d=%d]" % d)
    w = w.contiguous()
    b = b.contiguous()
    if out is None:
        out = torch.empty((B, d), dtype=torch.float32, device=x0.device)
    _lib.check(_lib.load().dir_dcn_cross_f32(_ptr(x0), x0.stride(0), _ptr(w), _ptr(b), w.shape[0], B, d, _ptr(out),
                                             out.stride(0), _stream()))
    return out


def cross_network_head(x0, w, b, head_w, want_x=False):
    """cross_network followed by the cross branch's share of the final dense(1) over concat([cross, deep]) (DeepCrossNetwork.py:136-137)
    in the same launch (include/dir_hip.h: dir_dcn_cross_head_f32): -> (x_L . head_w) [B, 1] (, x_L [B, d] when want_x; otherwise the
    cross output never reaches memory)."""
    _dev(x0, torch.float32, "x0")
    if x0.dim() != 2 or x0.stride(1) != 1:
        raise ValueError("x0 must be [B, d] with unit inner stride")
    B, d = x0.shape
    w = _dev(w, torch.float32, "w").contiguous()
    b = _dev(b, torch.float32, "b").contiguous()
    hw = _dev(head_w, torch.float32, "head_w").reshape(-1).contiguous()
    if w.dim() != 2 or w.shape[1] != d or tuple(b.shape) != tuple(w.shape) or hw.numel() != d:
        raise ValueError("cross_network_head: w, b [L, d=%d], head_w [d]" % d)
    out = torch.empty((B, d), dtype=torch.float32, device=x0.device) if want_x else None
    ho = torch.empty((B, 1), dtype=torch.float32, device=x0.device)
    _lib.check(_lib.load().dir_dcn_cross_head_f32(_ptr(x0), x0.stride(0), _ptr(w), _ptr(b), w.shape[0], B, d, _ptr(hw), _ptr(out),
                                                  out.stride(0) if want_x else 0, _ptr(ho), _stream()))
    return (ho, out) if want_x else ho


def pad4(d):
    return (d + 3) // 4 * 4


def cross_network_padded(x0p, w, b, out=None):
    """cross_network on a row-padded input: x0p [B, dp] holds the d = w.shape[1] real columns followed by dp - d ZERO columns
    (dp = pad4(d); what InputLayer(pad_to=4) produces for DCN's 429-wide input).  w, b [L, d] are zero-padded to [L, dp], so a
    pad column stays exactly 0 through every layer (0 * xw + 0 + 0) and the dot products see only zeros there: the real columns
    are computed exactly as cross_network does, in the same order (DeepCrossNetwork.py:345-346), on the 16-byte vector path.
    -> [B, dp] with the same zero tail."""
    L, d = w.shape
    dp = x0p.shape[1]
    if dp != pad4(d) or x0p.stride(1) != 1:
        raise ValueError("cross_network_padded: x0p must be [B, pad4(d)=%d] with unit inner stride" % pad4(d))
    if dp != d:
        w = torch.nn.functional.pad(w, (0, dp - d))
        b = torch.nn.functional.pad(b, (0, dp - d))
    return cross_network(x0p, w, b, out=out)


def cross_op(x0, x, w, b, out=None):
    """_cross_op (DeepCrossNetwork.py:336-347), one layer: y = x0 * (x . w)[:, None] + b + x."""
    _dev(x0, torch.float32, "x0")
    _dev(x, torch.float32, "x")
    if x0.shape != x.shape or x0.dim() != 2 or x0.stride(1) != 1 or x.stride() != x0.stride():
        raise ValueError("cross_op: x0 and x must be [B, d] with the same strides")
    B, d = x0.shape
    w = _dev(w, torch.float32, "w").reshape(-1).contiguous()
    b = _dev(b, torch.float32, "b").reshape(-1).contiguous()
    if w.numel() != d or b.numel() != d:
        raise ValueError("cross_op: w and b must be [d=%d]" % d)
    if out is None:
        out = torch.empty((B, d), dtype=torch.float32, device=x0.device)
    _lib.check(_lib.load().dir_dcn_cross_op_f32(_ptr(x0), _ptr(x), x0.stride(0), _ptr(w), _ptr(b), B, d, _ptr(out),
                                                out.stride(0), _stream()))
    return out


DIN_ACTIVATIONS = {"sigmoid": 0, "prelu": 1, "dice": 2}


def din_act_params(H1, H2, alpha1, alpha2, scale1=None, shift1=None, scale2=None, shift2=None):
    """The [3 H1 + 3 H2] parameter vector of the PReLU / Dice unit (include/dir_hip.h: dir_din_attention_pool_act_f32) from per-unit
    tensors: alpha1, scale1, shift1, alpha2, scale2, shift2 (missing scale -> 1, missing shift -> 0; PReLU reads the alphas only)."""
    dev = alpha1.device
    one = lambda n: torch.ones(n, dtype=torch.float32, device=dev)      # noqa: E731
    zero = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)    # noqa: E731
    parts = [alpha1, scale1 if scale1 is not None else one(H1), shift1 if shift1 is not None else zero(H1),
             alpha2, scale2 if scale2 is not None else one(H2), shift2 if shift2 is not None else zero(H2)]
    return torch.cat([t.detach().to(torch.float32).reshape(-1) for t in parts]).contiguous()


def din_activation_rows_(x, activation, alpha, scale=None, shift=None):
    """PReLU / Dice (inference form) over the rows of x [B, N], in place (include/dir_hip.h: dir_din_activation_rows_f32)."""
    _dev(x, torch.float32, "x")
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("din_activation_rows_: x must be [B, N] with unit column stride")
    B, N = x.shape
    vec = [None if t is None else _dev(t.detach().contiguous(), torch.float32, "activation parameter") for t in (alpha, scale, shift)]
    if hasattr(x, "_dir_bits"):
        del x._dir_bits                         # the rows' maxima a dense kernel left on x no longer describe it
    _lib.check(_lib.load().dir_din_activation_rows_f32(_ptr(x), x.stride(0) if B > 1 else max(N, x.stride(0)), B, N, DIN_ACTIVATIONS[activation],
                                                       _ptr(vec[0]), _ptr(vec[1]), _ptr(vec[2]), _stream()))
    return x


DIN_ARITHS = {"f32": 0, "bf16x3": 1, "f16x2": 2}
DIN_RANGE_RECHECK = 32          # training: the table / weights are re-measured every this many in-place updates (weight_absmax(every=...))


def din_arith(table, weights, arith=None):
    """The arithmetic code of the DIN unit's MFMA layers for this call (include/dir_hip.h: DIR_DIN_ARITH_*).  arith: "f32" | "bf16x3" |
    "f16x2" by name; None = "auto": the DIR_DIN_ARITH environment switch when set (A/B runs), otherwise fp16 x 2 (UNSCALED: the kernel splits
    table rows and their products as they are) while the table and every weight matrix measure inside f16_range_ok's window, bf16 x 3 (fp32's
    exponent range) outside it -- a table scaled by 2^20 gives the bf16 x 3 answer, not inf; one scaled by 2^-10 keeps a relative 1e-5."""
    if arith is not None:
        return DIN_ARITHS[arith]
    if os.environ.get("DIR_DIN_ARITH"):
        return -1
    if not f16_range_ok(weight_absmax(table, DIN_RANGE_RECHECK)):
        return DIN_ARITHS["bf16x3"]
    for w in weights:
        if not f16_range_ok(weight_absmax(w, DIN_RANGE_RECHECK)):
            return DIN_ARITHS["bf16x3"]
    return DIN_ARITHS["f16x2"]


# development switch: 0 = the fp16 x 2 forward stays on din_wave_k (one wave per sample) instead of the packed kernel of round 6
DIN_PACKED = os.environ.get("DIR_DIN_PACKED", "1") != "0"
_DIN_PACK_WS = {}


def din_pack_covers(K, T, H1, H2):
    """Shapes dir_din_attention_pool_packed_f32 takes (include/dir_hip.h)."""
    return K == 64 and 1 <= T <= 65535 and 0 < H1 <= 80 and 0 < H2 <= 48 and H1 % 4 == 0 and H2 % 4 == 0


_DIN_PACK_IMAGES = {}


def din_pack_image(owners, args, activation, ap):
    """The packed DIN kernel's weight image (include/dir_hip.h: dir_din_pack_weights_f32), built once per version of the weights: keyed on
    the LONG-LIVED tensors `owners` (W1, W2, W3 -- a module's nn.Parameters; their .data views are new objects on every call) plus b1, b2
    and the activation parameters, by identity, version and storage.  While a stream is capturing no cache is consulted or filled: the pack
    kernel becomes part of the graph and reads the weights as they are at replay time (the rule of every per-version cache here)."""
    import weakref
    W1, b1, W2, b2, W3, b3 = args
    lib = _lib.load()
    H1, H2 = W1.shape[1], W2.shape[1]
    watched = list(owners) + ([] if len(owners) >= 5 else [b1, b2]) + ([ap] if ap is not None else [])      # range_of may name b1 / b2's owners too
    capturing = torch.cuda.is_current_stream_capturing()
    key = tuple(t.data_ptr() for t in (W1, b1, W2, b2, W3)) + (activation, ap.data_ptr() if ap is not None else 0, W1.device.index)
    sig = tuple((id(t), t._version if not t.is_inference() else -1) for t in watched) + (_CACHE_GEN[0],)
    hit = _DIN_PACK_IMAGES.get(key)
    if hit is not None and not capture_bypasses_caches() and hit[0] == sig and all(r() is t for r, t in zip(hit[1], watched)):
        return hit[2]
    img = torch.empty(int(lib.dir_din_pack_image_bytes()), dtype=torch.uint8, device=W1.device)
    _lib.check(lib.dir_din_pack_weights_f32(_ptr(W1), _ptr(b1), H1, _ptr(W2), _ptr(b2), H2, _ptr(W3), DIN_ACTIVATIONS[activation], _ptr(ap),
                                            _ptr(img), _stream()))
    if not capturing and not any(t.is_inference() for t in watched):
        if len(_DIN_PACK_IMAGES) > 64:
            _DIN_PACK_IMAGES.clear()
        _DIN_PACK_IMAGES[key] = (sig, [weakref.ref(t) for t in watched], img)
    return img


def _din_pack_ws(device, nbytes):
    """The packed DIN kernel's workspace: one per (device, stream), grown on demand -- calls on one stream are ordered, so they may share it."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _DIN_PACK_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _DIN_PACK_WS[key] = torch.empty(max(nbytes, 4096), dtype=torch.uint8, device=device)
    return ws


def din_attention_pool(table, hist, hist_len, cand, W1, b1, W2, b2, W3, b3, normalize=False, want_scores=False, activation="sigmoid",
                       act_params=None, arith=None, range_of=None):
    """DIN local activation unit + pooling (include/dir_hip.h A13): -> out [B,K] (, scores [B,T]).  activation "prelu" / "dice": the
    paper's own hidden activations (dir_din_attention_pool_act_f32; act_params from din_act_params).  arith: see din_arith.
    range_of: (table, W1, W2, W3[, b1, b2]) as the LONG-LIVED tensors whose magnitudes din_arith measures and remembers per version (and whose
    versions key the packed kernel's weight image) -- a module passes
    its nn.Parameters here and `.data` views as operands (a `.data` view is a new tensor object with its own version counter on every
    call: measured afresh each time, a 2.56 GB pass for the cfg-4 table)."""
    _dev(table, torch.float32, "table")
    _dev(hist, torch.int64, "hist")
    _dev(cand, torch.int64, "cand")
    if hist_len is not None:
        _dev(hist_len, torch.int32, "hist_len")
    B, T = hist.shape
    K = table.shape[1]
    H1, H2 = W1.shape[1], W2.shape[1]
    if tuple(W1.shape) != (4 * K, H1) or tuple(W2.shape) != (H1, H2) or W3.numel() != H2:
        raise ValueError("DIN weights must be W1 [4K,H1], W2 [H1,H2], W3 [H2]")
    args = [t.contiguous() for t in (W1, b1, W2, b2, W3, b3)]
    for t in args:
        _dev(t, torch.float32, "DIN weight")
    # include/dir_hip.h declares hist [B,T], cand [B] and hist_len [B] as dense arrays (no stride arguments): a strided view
    # (ids[:, j], lens[:, 0]) must be packed before its data_ptr() is handed over
    hist, cand = hist.contiguous(), cand.contiguous()
    if hist_len is not None:
        hist_len = hist_len.contiguous()
    if cand.numel() != B or (hist_len is not None and hist_len.numel() != B):
        raise ValueError("DIN: cand and hist_len must have one entry per sample")
    out = torch.empty((B, K), dtype=torch.float32, device=table.device)
    scores = torch.empty((B, T), dtype=torch.float32, device=table.device) if want_scores else None
    ap = None
    if activation != "sigmoid":
        if activation not in DIN_ACTIVATIONS:
            raise ValueError("DIN activation must be one of %s" % sorted(DIN_ACTIVATIONS))
        if act_params is None or act_params.numel() != 3 * H1 + 3 * H2:
            raise ValueError("DIN %s unit: act_params must hold 3 H1 + 3 H2 floats (ops.din_act_params)" % activation)
        ap = _dev(act_params.contiguous(), torch.float32, "act_params")
    rsrc = range_of if range_of is not None else (table, W1, W2, W3)
    code = din_arith(rsrc[0], rsrc[1:4], arith) if B > 0 else -1
    if code == DIN_ARITHS["f16x2"] and DIN_PACKED and din_pack_covers(K, T, H1, H2):
        # round 6: the packed kernel (rows of consecutive samples end to end in the MFMA tiles, a static equal-weight partition of the samples)
        lib = _lib.load()
        ws = _din_pack_ws(table.device, int(lib.dir_din_pack_workspace_bytes(B, 1 if want_scores else 0)))
        img = din_pack_image(rsrc[1:], args, activation, ap)
        _lib.check(lib.dir_din_attention_pool_packed_f32(_ptr(table), K, _ptr(hist), _ptr(hist_len), _ptr(cand), T, _ptr(args[0]), _ptr(args[1]), H1,
                                                         _ptr(args[2]), _ptr(args[3]), H2, _ptr(args[4]), _ptr(args[5]), int(bool(normalize)),
                                                         DIN_ACTIVATIONS[activation], _ptr(ap), _ptr(img), B, _ptr(out), _ptr(scores), _ptr(ws),
                                                         ws.numel(), _stream()))
        return (out, scores) if want_scores else out
    _lib.check(_lib.load().dir_din_attention_pool_arith_f32(_ptr(table), K, _ptr(hist), _ptr(hist_len), _ptr(cand), T,
                                                            _ptr(args[0]), _ptr(args[1]), H1, _ptr(args[2]), _ptr(args[3]),
                                                            H2, _ptr(args[4]), _ptr(args[5]), int(bool(normalize)),
                                                            DIN_ACTIVATIONS[activation], _ptr(ap), code, B, _ptr(out), _ptr(scores), _stream()))
    return (out, scores) if want_scores else out


# ---- the DIN unit's PReLU / Dice TRAINING path over the compact row list (csrc/din_rows_train.hip; no reference code: arXiv:1706.06978) -----
def act_rows_supported(s):
    """Shapes dir_act_rows_train_f32 / _backward_f32 (and dir_bn_train_stats_f32) take: float32 CUDA [M > 0, N], N % 4 == 0, N <= 1024."""
    return (s.is_cuda and s.dtype == torch.float32 and s.dim() == 2 and s.shape[0] > 0 and s.shape[1] % 4 == 0 and s.shape[1] <= 1024
            and s.stride(1) == 1 and s.stride(0) % 4 == 0 and s.data_ptr() % 16 == 0)


def _vec16(t, n, what):
    t = _dev(t.detach().reshape(-1).contiguous(), torch.float32, what)
    if t.numel() != n:
        raise ValueError("%s must hold %d values" % (what, n))
    return t if t.data_ptr() % 16 == 0 else t.clone()


def din_feat_rows(table, ids_h, b_idx, cand):
    """X [N, 3K] = [h_n | h_n * a_b | a_b] and Hc [N, K] = h_n over the valid (sample, position) rows (dir_din_feat_rows_f32)."""
    _dev(table, torch.float32, "table")
    N, K = ids_h.numel(), table.shape[1]
    X = torch.empty((N, 3 * K), dtype=torch.float32, device=table.device)
    Hc = torch.empty((N, K), dtype=torch.float32, device=table.device)
    _lib.check(_lib.load().dir_din_feat_rows_f32(_ptr(table), K, _ptr(_dev(ids_h, torch.int64, "ids_h").contiguous()),
                                                 _ptr(_dev(b_idx, torch.int64, "b_idx").contiguous()), _ptr(_dev(cand, torch.int64, "cand").contiguous()),
                                                 N, _ptr(X), _ptr(Hc), _stream()))
    return X, Hc


def din_feat_rows_backward(table, ids_h, row_off, cand, dX, dH):
    """-> grows [N + B, K]: dL/dh_n for the N history rows, then dL/da_b for the B candidates (dir_din_feat_rows_backward_f32)."""
    N, K, B = ids_h.numel(), table.shape[1], cand.numel()
    grows = torch.empty((N + B, K), dtype=torch.float32, device=table.device)
    _lib.check(_lib.load().dir_din_feat_rows_backward_f32(_ptr(table), K, _ptr(ids_h.contiguous()), _ptr(row_off.contiguous()), _ptr(cand.contiguous()),
                                                          B, N, _ptr(dX.contiguous()), _ptr(dH.contiguous()), _ptr(grows), _stream()))
    return grows


def act_rows_train(s, activation, alpha, scale=None, shift=None):
    """y = PReLU / Dice (s) out of place (dir_act_rows_train_f32); Dice: (scale, shift) of the statistics that normalise (the batch's in TRAIN mode)."""
    M, N = s.shape
    y = torch.empty((M, N), dtype=torch.float32, device=s.device)
    vec = [_vec16(alpha, N, "alpha")] + [None if t is None else _vec16(t, N, "scale / shift") for t in (scale, shift)]
    _lib.check(_lib.load().dir_act_rows_train_f32(_ptr(s), s.stride(0), M, N, DIN_ACTIVATIONS[activation], _ptr(vec[0]), _ptr(vec[1]), _ptr(vec[2]),
                                                  _ptr(y), y.stride(0), _stream()))
    return y


def act_rows_backward(g, s, activation, alpha, scale=None, shift=None):
    """g = dL/dy -> (d1 [M, N], gx [M, N] | None, galpha [N]) (dir_act_rows_backward_f32): d1 = g df/ds with the normalised pre-activation held
    fixed (PReLU: all of dL/ds), gx = dL/d(normalised pre-activation) (Dice), galpha = dL/dalpha."""
    M, N = s.shape
    if g.stride(1) != 1 or g.stride(0) % 4 or g.data_ptr() % 16:
        g = g.contiguous()
    lib = _lib.load()
    P = int(lib.dir_act_rows_backward_partials(M, N))
    d1 = torch.empty((M, N), dtype=torch.float32, device=s.device)
    gx = torch.empty((M, N), dtype=torch.float32, device=s.device) if activation == "dice" else None
    galpha = torch.empty(N, dtype=torch.float32, device=s.device)
    part = torch.empty((max(P, 1), N), dtype=torch.float32, device=s.device)
    vec = [_vec16(alpha, N, "alpha")] + [None if t is None else _vec16(t, N, "scale / shift") for t in (scale, shift)]
    _lib.check(lib.dir_act_rows_backward_f32(_ptr(g), g.stride(0), _ptr(s), s.stride(0), M, N, DIN_ACTIVATIONS[activation], _ptr(vec[0]), _ptr(vec[1]),
                                             _ptr(vec[2]), _ptr(d1), d1.stride(0), _ptr(gx), gx.stride(0) if gx is not None else N, _ptr(galpha),
                                             _ptr(part), P, _stream()))
    return d1, gx, galpha


def dice_train_backward(g, s, alpha, scale, shift, mean, inv):
    """Dice in training mode, the whole backward over rows in two passes (include/dir_hip.h: dir_dice_train_backward_f32): g = dL/dy [M, N],
    s the pre-activations, (scale, shift, mean, inv) this batch's statistics from bn_train_stats -> (ds [M, N], galpha [N]); what
    act_rows_backward + bn_train_backward + an add computed in seven reads and four writes of [M, N]."""
    M, N = s.shape
    if g.stride(1) != 1 or g.stride(0) % 4 or g.data_ptr() % 16:
        g = g.contiguous()
    if not bn_train_supported(s) or not bn_train_supported(g):
        raise ValueError("dice_train_backward: g and s must be float32 [M > 0, N] with N and the row strides multiples of 4, 16-byte aligned")
    lib = _lib.load()
    P = int(lib.dir_bn_train_partials(M, N))
    ds = torch.empty((M, N), dtype=torch.float32, device=s.device)
    vec = torch.empty((4, N), dtype=torch.float32, device=s.device)           # galpha, the three coefficients
    part = torch.empty((P, 3, N), dtype=torch.float32, device=s.device)
    al, sc, sh = _vec16(alpha, N, "alpha"), _vec16(scale, N, "scale"), _vec16(shift, N, "shift")
    _lib.check(lib.dir_dice_train_backward_f32(_ptr(g), g.stride(0), _ptr(s), s.stride(0), M, N, _ptr(al), _ptr(sc), _ptr(sh), _ptr(mean.contiguous()),
                                               _ptr(inv.contiguous()), _ptr(ds), ds.stride(0), _ptr(vec[0]), _ptr(vec[1]), _ptr(part), P, _stream()))
    return ds, vec[0]


def din_pool_rows(scores, Hc, row_off, B, normalize):
    """-> (out [B, K], w [N]): the attention weights of every sample's rows and the weighted sum of its history rows (dir_din_pool_rows_f32)."""
    N, K = Hc.shape
    out = torch.empty((B, K), dtype=torch.float32, device=Hc.device)
    w = torch.empty(max(N, 1), dtype=torch.float32, device=Hc.device)[:N]
    _lib.check(_lib.load().dir_din_pool_rows_f32(_ptr(scores.contiguous()), _ptr(Hc), K, _ptr(row_off.contiguous()), B, N, int(bool(normalize)), _ptr(w),
                                                 _ptr(out), _stream()))
    return out, w


def din_pool_rows_backward(g, Hc, w, row_off, normalize):
    """g [B, K] -> (ds [N], dH [N, K]) (dir_din_pool_rows_backward_f32)."""
    N, K = Hc.shape
    B = g.shape[0]
    ds = torch.empty(max(N, 1), dtype=torch.float32, device=Hc.device)[:N]
    dH = torch.empty((N, K), dtype=torch.float32, device=Hc.device)
    _lib.check(_lib.load().dir_din_pool_rows_backward_f32(_ptr(g.contiguous()), _ptr(Hc), K, _ptr(w), _ptr(row_off.contiguous()), B, N,
                                                          int(bool(normalize)), _ptr(ds), _ptr(dH), _stream()))
    return ds, dH


def _masked_rows(x, mask, n):
    """x[mask] when the number of selected elements n is already known on the host: no second device-to-host read (boolean indexing
    sizes its result through one)."""
    if hasattr(torch, "nonzero_static") and x.is_cuda:
        try:
            return x.reshape(-1)[torch.nonzero_static(mask.reshape(-1), size=n).squeeze(1)]
        except (RuntimeError, NotImplementedError):
            pass
    return x[mask]


class DinTrainPlan:
    """Row bookkeeping one training call of the DIN unit needs, computed once (one host read) and shared by the forward that saves its
    activations and the backward: `valid` [B, T] (position inside the length and not pruned), row_off [B] (exclusive prefix of the valid
    counts: the compact gradient row list), tile_off [B] (exclusive prefix of ceil(len / 16): the records), N, n_tiles."""

    def __init__(self, hist, hist_len):
        B, T = hist.shape
        dev = hist.device
        valid = hist >= 0
        if hist_len is not None:
            valid &= torch.arange(T, device=dev).unsqueeze(0) < hist_len.clamp(0, T).unsqueeze(1)
        cnt = valid.sum(dim=1)
        incl = torch.cumsum(cnt, 0)
        lens = hist_len.clamp(0, T).to(torch.int64) if hist_len is not None else torch.full((B,), T, dtype=torch.int64, device=dev)
        tcnt = (lens + 15) // 16
        tincl = torch.cumsum(tcnt, 0)
        self.valid, self.row_off, self.tile_off = valid, (incl - cnt).contiguous(), (tincl - tcnt).contiguous()
        self.N, self.n_tiles = (int(v) for v in torch.stack([incl[-1], tincl[-1]]).tolist()) if B else (0, 0)     # the one host read


def din_attention_pool_save(table, hist, hist_len, cand, W1, b1, W2, b2, W3, b3, normalize=False, plan=None, range_of=None):
    """Training forward of the (K 64, H1 <= 80, H2 <= 48, T <= 64) unit (include/dir_hip.h: dir_din_attention_pool_save_f32): as
    din_attention_pool(..., want_scores=True), and every history row's hidden activations stay in a workspace for
    din_attention_pool_backward(..., saved=...), which then recomputes nothing.  -> out, scores, (plan, workspace)."""
    _dev(table, torch.float32, "table")
    _dev(hist, torch.int64, "hist")
    _dev(cand, torch.int64, "cand")
    if hist_len is not None:
        _dev(hist_len, torch.int32, "hist_len")
    B, T = hist.shape
    K = table.shape[1]
    H1, H2 = W1.shape[1], W2.shape[1]
    if tuple(W1.shape) != (4 * K, H1) or tuple(W2.shape) != (H1, H2) or W3.numel() != H2:
        raise ValueError("DIN weights must be W1 [4K,H1], W2 [H1,H2], W3 [H2]")
    if not din_backward_supported(K, T, H1, H2):
        raise _lib.DirError(-4, "din_attention_pool_save covers K = 64, H1 <= 80, H2 <= 48, T <= 64")
    args = [t.contiguous() for t in (W1, b1, W2, b2, W3, b3)]
    for t in args:
        _dev(t, torch.float32, "DIN weight")
    hist, cand = hist.contiguous(), cand.contiguous()
    if hist_len is not None:
        hist_len = hist_len.contiguous()
    if cand.numel() != B or (hist_len is not None and hist_len.numel() != B):
        raise ValueError("DIN: cand and hist_len must have one entry per sample")
    if plan is None:
        plan = DinTrainPlan(hist, hist_len)
    lib = _lib.load()
    need = int(lib.dir_din_backward_rows_workspace_bytes(K, H1, H2, plan.n_tiles))
    ws = torch.empty(need, dtype=torch.uint8, device=table.device)       # owned by this call's autograd node until its backward ran
    out = torch.empty((B, K), dtype=torch.float32, device=table.device)
    scores = torch.empty((B, T), dtype=torch.float32, device=table.device)
    rsrc = range_of if range_of is not None else (table, W1, W2, W3)
    code = din_arith(rsrc[0], rsrc[1:]) if B > 0 else -1
    _lib.check(lib.dir_din_attention_pool_save_arith_f32(_ptr(table), K, _ptr(hist), _ptr(hist_len), _ptr(cand), T, _ptr(args[0]), _ptr(args[1]), H1,
                                                         _ptr(args[2]), _ptr(args[3]), H2, _ptr(args[4]), _ptr(args[5]), int(bool(normalize)), code, B,
                                                         _ptr(out), _ptr(scores), _ptr(plan.tile_off), plan.n_tiles, _ptr(ws), need, _stream()))
    return out, scores, (plan, ws)


def dense_supported(x, weight):
    """Shapes dir_dense_f32 covers: fp32 CUDA tensors, in_features a multiple of 4, at least 16 output columns (narrower layers --
    the final units=1 logit layers -- are matrix-vector products: library code)."""
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 2 and x.shape[1] % 4 == 0
            and weight.shape[0] >= 16 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0)


# default arithmetic of dense / dense_gated: "auto" (bf16x3 where covered and not padding-bound, fp32 MFMA otherwise) | "f32" | "bf16x3"
DENSE_ARITH = os.environ.get("DIR_DENSE_ARITH", "auto")
# what arith="auto_bounded" puts in bf16x3's place: "f16x2_rows" (round 5: the same row-scaled fp16 x 2 kernel a general input gets -- the
# rows' exponents carried from the producing kernel's epilogue where the caller has them, one max pass otherwise -- so "bounded by
# construction" is no longer a promise the forward's correctness rests on), "f16x2" (round 4: the unscaled kernel, |x| < 65 504 and an
# absolute 2^-25 below 2^-3) or "bf16x3" (A/B switches)
DENSE_BOUNDED_SPLIT = os.environ.get("DIR_DENSE_BOUNDED_SPLIT", "f16x2_rows")
# the split of the dense BACKWARD kernels whose operand is a gradient: "f16x2" = dL/dx on dir_dense_f16x2_rows_f32 (rows of g scaled by powers
# of two) and, where the layer's input is bounded by construction, dL/dW on dir_dense_dw_f16x2_f32 (g scaled by one power of two);
# "bf16x3" = rounds 2-3's arithmetic
DENSE_BWD_SPLIT = os.environ.get("DIR_DENSE_BWD_SPLIT", "f16x2")
# what dense(arith="auto") runs on a GENERAL input (no "bounded by construction" promise) where it used to pick bf16x3: "f16x2_rows" = the
# row-scaled fp16 x 2 kernel behind one max pass over x (any magnitudes: a raw numeric column of 99 999 next to 0.1-scale embeddings; error
# relative to each row's largest element 2^-22 .. 2^-39, i.e. below fp32's own rounding of the row's dot products) | "bf16x3"
DENSE_GENERAL_SPLIT = os.environ.get("DIR_DENSE_GENERAL_SPLIT", "f16x2_rows")
# development switch: 0 = every fp16 x 2 backward kernel gets its scales from a max pass of its own (dir_row_absmax_bits_f32) instead of from
# the kernel that produced the gradient
DENSE_BWD_CARRY = os.environ.get("DIR_DENSE_BWD_CARRY", "1") != "0"
# development switch: 0 = a row-scaled forward layer does not leave its output's maxima on the tensor (every layer runs its own max pass)
DENSE_FWD_CARRY = os.environ.get("DIR_DENSE_FWD_CARRY", "1") != "0"
# Small batches run dir_dense_small_f32 where the fp32 kernel would run: up to 256 rows always (the reference's batch sizes, 100 / 256: 3.7-8 us
# per layer against the library's 7.5-9 and dir_dense_f32's 26-39; only 1024 x 1024 at 256 rows is behind the library, 16 against 11 us),
# up to DENSE_SMALL_ROWS rows while M N K stays under DENSE_SMALL_MNK (profiles/r05_dense_small_probe.txt; DIR_DENSE_SMALL_ROWS = 0: never)
DENSE_SMALL_ROWS = int(os.environ.get("DIR_DENSE_SMALL_ROWS", "512"))
DENSE_SMALL_MNK = 1.2e8


def dense_small_covers(M, Kd, N):
    """Whether dense() runs dir_dense_small_f32 for an [M, Kd] x [N, Kd] layer (given fp32 arithmetic and 16-byte aligned rows)."""
    return 0 < M <= DENSE_SMALL_ROWS and Kd % 4 == 0 and (M <= 256 or float(M) * N * Kd <= DENSE_SMALL_MNK)
# round 6: the rows in between (dir_dense_mid_f32: a workgroup per 32 / 64 x 64 tile, LDS-staged reduction) -- from where dense_small_covers stops
# up to DENSE_MID_ROWS rows (tools/dense_small_probe.py -> profiles/r06_dense_mid_probe.txt: within a few percent of the library's kernels at
# 400-wide layers and the 432 x 1024 layer from 512 to 4096 rows, ahead of them from 6144; 15-25 % behind on 1024 x 1024, the one shape
# where the library's fp32 GEMM is near the fp32-MFMA peak.  From 6144 rows dir_dense_f32's 128-row workgroups fill the chip.
# DIR_DENSE_MID_ROWS = 0: never)
DENSE_MID_ROWS = int(os.environ.get("DIR_DENSE_MID_ROWS", "6143"))


def dense_mid_covers(M, Kd, N):
    """Whether dense() runs dir_dense_mid_f32 for an [M, Kd] x [N, Kd] layer (given fp32 arithmetic and 16-byte aligned rows)."""
    return 0 < M <= DENSE_MID_ROWS and Kd % 4 == 0 and N >= 16 and not dense_small_covers(M, Kd, N)
DENSE_BF3_MIN_ROWS = 12288     # below this the 256-row tiles leave too much of the chip idle (tools/dense_bf3_probe.py: x1.14 at 16 384 rows, x0.58 at 4 096)
_DENSE_IMAGES = {}             # data_ptr -> (weakref to the weight tensor, version, shape, strides, image)


def invalidate_caches():
    """Drop every cached weight image (dense / tower bf16x3 images).  The caches follow tensor._version, which in-place torch ops bump;
    a write through `param.data`, a raw-pointer kernel or a checkpoint loader that copies into storage directly does not -- call this
    after such a write (checkpoint.load_* do)."""
    _DENSE_IMAGES.clear()
    _TOWER_IMAGES.clear()
    _WEIGHT_ABSMAX.clear()
    _DIN_PACK_IMAGES.clear()
    _CIN_POOLED_IMAGES.clear()
    _CACHE_GEN[0] += 1             # TableSet.absmax / ShardedTables.absmax measurements taken before this call are stale


def dense_bf16x3_covers(x, weight, out=None, gate=None):
    """Operands dir_dense_bf16x3_f32 accepts: Kd, N and every row stride a multiple of 4, 16-byte aligned bases."""
    M, Kd = x.shape
    N = weight.shape[0]
    ok = (Kd % 4 == 0 and N % 4 == 0 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0)
    if out is not None:
        ok = ok and out.stride(1) == 1 and out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0
    if gate is not None:
        ok = ok and gate.stride(1) == 1 and gate.stride(0) % 4 == 0 and gate.data_ptr() % 16 == 0
    return ok


def dense_runs_bf16x3(x, weight, arith=None):
    """Whether dense / dense_gated will run the bf16x3 kernel on these operands -- it packs its weight image from any strides, so the caller
    need not lay the weight (or its transpose) out for the fp32 kernel first."""
    return _dense_arith(arith, x, weight, None, None) == "bf16x3"


def dense_auto_arith(M, Kd, N):
    """What arith="auto" runs for a layer shape: bf16x3 (csrc/dense_bf3.hip) when the batch fills its 256-row workgroups and its
    column blocks (128 / 208 / 256 wide, whichever pads N least) and 32-wide k-steps pad the layer by at most a third."""
    if M < DENSE_BF3_MIN_ROWS or N < 64:
        return "f32"
    tiles = -(-N // 16)
    pad = min(-(-tiles // c) * c for c in (8, 13, 16)) * 16
    kpad = -(-Kd // 32) * 32
    return "bf16x3" if pad * kpad <= 1.34 * N * Kd else "f32"


_ABSMAX_WS = {}


# development switch: 0 = the training gather leaves no row maxima on its output (the first dense layer runs its own max pass)
GATHER_BITS = os.environ.get("DIR_GATHER_BITS", "1") != "0"


def row_absmax_bits(x, want_all=True):
    """(row_bits [M] int32, all_bits [1] int32 | None) of x [M, N] (N, the row stride multiples of 4, 16-byte aligned): the bit patterns of
    max_k |x[r, k]| and of max |x| (include/dir_hip.h: dir_row_absmax_bits_f32) -- the powers of two the row-scaled / tensor-scaled
    fp16 x 2 kernels multiply x by."""
    M, N = x.shape
    lib = _lib.load()
    ws = _bits_ws(x.device)                              # ticket word + block maxima, per (device, stream)
    buf = torch.empty(M + 4, dtype=torch.int32, device=x.device)
    rb, ab = buf[:M], buf[M:M + 1]
    _lib.check(lib.dir_row_absmax_bits_f32(_ptr(x), x.stride(0), M, N, _ptr(rb), _ptr(ab) if want_all else None, _ptr(ws), _stream()))
    return rb, (ab if want_all else None)


def _out_bits(bits_out, M, device):
    """A (row_bits [M], all_bits [1]) pair for a kernel epilogue to fill (the C entry zeroes what it must), appended to the caller's list;
    (None, None) if not wanted."""
    if bits_out is None or not DENSE_BWD_CARRY:
        return None, None
    buf = torch.empty(M + 4, dtype=torch.int32, device=device)
    pair = (buf[:M], buf[M:M + 1])
    bits_out.append(pair)
    return pair


def _rows_covered(x):
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= DENSE_BF3_MIN_ROWS and x.shape[1] % 4 == 0
            and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0)


def grad_bits(g, want_all=True):
    """row_absmax_bits of a gradient g [M, N] for the fp16 x 2 BACKWARD kernels, or None when DENSE_BWD_SPLIT is not "f16x2" or g is not a
    covered operand (then the callers keep bf16 x 3)."""
    if DENSE_BWD_SPLIT != "f16x2" or DENSE_ARITH != "auto" or not _rows_covered(g):
        return None
    return row_absmax_bits(g, want_all)


def dense_bf3_image(weight, split="bf16x3"):
    """The packed image (bf16 x 3, or fp16 x 2 pieces with split="f16x2") of a [N, Kd] fp32 weight of ANY strides
    (dir_dense_*_pack_strided_f32: a `.t()` view is packed straight from the storage of the tensor it transposes), cached per tensor and
    split until the tensor is modified in place (tensor._version) or goes away."""
    import weakref
    key = weight.data_ptr() if split == "bf16x3" else (weight.data_ptr(), split)
    sig = (weight._version, tuple(weight.shape), tuple(weight.stride()))
    capturing = weight.is_cuda and torch.cuda.is_current_stream_capturing()
    hit = _DENSE_IMAGES.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == sig and not capture_bypasses_caches(weight):
        return hit[2]
    N, Kd = weight.shape
    lib = _lib.load()
    nbytes = int(lib.dir_dense_bf16x3_image_bytes(Kd, N))
    img = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    pack = lib.dir_dense_f16x2_pack_strided_f32 if split == "f16x2" else lib.dir_dense_bf16x3_pack_strided_f32
    _lib.check(pack(_ptr(weight), weight.stride(0), weight.stride(1), Kd, N, _ptr(img), nbytes, _stream()))
    if capturing:
        # Under graph capture the caches are bypassed in BOTH directions: a hit would leave the pack kernel out of the graph --
        # every replay would then run against the image of the weights as they were at capture time, although the replayed optimiser
        # kernels keep changing them -- and an image stored now would live in the graph's private pool.  The pack is captured, its
        # image is graph-owned memory, and nothing is remembered.
        return img
    if len(_DENSE_IMAGES) > 256:
        _DENSE_IMAGES.clear()
    _DENSE_IMAGES[key] = (weakref.ref(weight), sig, img)
    return img


def _dense_arith(arith, x, weight, out, gate):
    arith = arith or DENSE_ARITH
    if arith not in ("auto", "f32", "bf16x3", "f16x2", "auto_bounded"):
        raise ValueError("dense: arith must be 'auto', 'auto_bounded', 'f32', 'bf16x3' or 'f16x2'")
    covered = dense_bf16x3_covers(x, weight, out, gate)
    if arith in ("auto", "auto_bounded"):
        split = "f16x2" if arith == "auto_bounded" and DENSE_BOUNDED_SPLIT == "f16x2" and gate is None else "bf16x3"
        return split if covered and dense_auto_arith(x.shape[0], x.shape[1], weight.shape[0]) == "bf16x3" else "f32"
    if arith in ("bf16x3", "f16x2") and not covered:
        raise ValueError("dense: arith=%r needs Kd, N and the row strides to be multiples of 4 and 16-byte aligned operands" % arith)
    if arith == "f16x2" and gate is not None:
        raise ValueError("dense: the gated (data-gradient) form runs bf16x3")
    return arith


def dense(x, weight, bias=None, relu=False, out=None, post_scale=None, post_shift=None, arith=None, row_bits=None, bits_out=None, xbits_out=None):
    """y = act(x @ weight.T + bias) (include/dir_hip.h: dir_dense_f32 / dir_dense_bf16x3_f32).  x [M, Kd], weight [N, Kd] (nn.Linear
    layout), bias [N].  post_scale / post_shift [N]: the inference batch-norm that follows the activation, as
    y * post_scale + post_shift in the same pass.  arith: "f32" (fp32 MFMA), "bf16x3" (three-way bf16 split of both operands on the
    bf16 pipe, fp32 accumulate: fp32-equivalent, not bitwise the same), "auto" (dense_auto_arith), None = DENSE_ARITH.
    "f16x2" (dir_dense_f16x2_f32: two fp16 pieces per operand, three products) / "auto_bounded" (what "auto" picks, with f16x2 in the
    place of bf16x3): for layers whose input is bounded by construction -- embedding concatenations, ReLU / batch-normalised
    activations, the CIN's pooled products (|x|, |W| < 65 504).
    row_bits (grad_bits(x)[0]): x is a gradient; where "auto" would run bf16x3 the row-scaled fp16 x 2 kernel runs
    (dir_dense_f16x2_rows_f32).  bits_out (a list): when that kernel ran, (row_bits, all_bits) of the OUTPUT is appended -- its epilogue
    leaves them, for the next layer of a backward chain.  xbits_out (a list): when the op ran its own max pass over x, x's all_bits ([1]
    int32: the bit pattern of max |x|) is appended -- what the layer's weight gradient scales x by (dense_dw(x_bits=...))."""
    _dev(x, torch.float32, "x")
    _dev(weight, torch.float32, "weight")
    M, Kd = x.shape
    N = weight.shape[0]
    if weight.shape[1] != Kd or x.stride(1) != 1:
        raise ValueError("dense: x [M, Kd] with unit inner stride, weight [N, Kd]")
    if bias is not None:
        bias = _dev(bias, torch.float32, "bias").contiguous()
        if bias.numel() != N:
            raise ValueError("dense: bias [N]")
    fresh = out is None
    if fresh:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    which = _dense_arith(arith, x, weight, out, None)
    asked = arith or DENSE_ARITH
    if which == "bf16x3" and (asked == "auto" or (asked == "auto_bounded" and DENSE_BOUNDED_SPLIT == "f16x2_rows")):
        if row_bits is None and DENSE_GENERAL_SPLIT == "f16x2_rows" and _rows_covered(x):
            hint = getattr(x, "_dir_bits", None)      # left by the row-scaled kernel that produced x (below), valid while x is unmodified
            if hint is not None and hint[2] == x._version and hint[0].numel() == M and hint[0].device == x.device:
                row_bits, xall = hint[0], hint[1]
            else:
                row_bits, xall = row_absmax_bits(x, want_all=xbits_out is not None)      # a general input: its rows' exponents first (one pass over x)
            if xbits_out is not None and xall is not None:
                xbits_out.append(xall)
        if row_bits is not None:
            which = "f16x2_rows"
    use_bf3 = which in ("bf16x3", "f16x2", "f16x2_rows")
    if not use_bf3 and (weight.stride(1) != 1 or weight.stride(0) % 4 or weight.data_ptr() % 16):
        weight = weight.contiguous()          # (the fp32 kernel reads rows with 16-byte loads; the bf16x3 image is packed from any strides)
    if post_scale is not None:
        post_scale, post_shift = _dev(post_scale, torch.float32, "post_scale").contiguous(), _dev(post_shift, torch.float32, "post_shift").contiguous()
        if post_scale.numel() != N or post_shift.numel() != N:
            raise ValueError("dense: post_scale / post_shift [N]")
    if which == "f32" and dense_small_covers(M, Kd, N) and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0:
        # small batches (the reference's 100 / 256): one wave per 16 x 16 tile, no LDS, no barrier (csrc/dense.hip: dense_small_k)
        _lib.check(_lib.load().dir_dense_small_f32(_ptr(x), x.stride(0), _ptr(weight), weight.stride(0), _ptr(bias), 1 if relu else 0,
                                                   _ptr(post_scale), _ptr(post_shift), M, Kd, N, _ptr(out), out.stride(0), _stream()))
        return out
    if which == "f32" and dense_mid_covers(M, Kd, N) and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0:
        # mid-size batches: a workgroup per 32 / 64 x 64 tile, the reduction through LDS (csrc/dense.hip: dense_mid_k)
        _lib.check(_lib.load().dir_dense_mid_f32(_ptr(x), x.stride(0), _ptr(weight), weight.stride(0), _ptr(bias), 1 if relu else 0,
                                                 _ptr(post_scale), _ptr(post_shift), M, Kd, N, _ptr(out), out.stride(0), _stream()))
        return out
    if which == "f16x2":
        _lib.check(_lib.load().dir_dense_f16x2_f32(_ptr(x), x.stride(0), _ptr(dense_bf3_image(weight, "f16x2")), _ptr(bias), 1 if relu else 0,
                                                   _ptr(post_scale), _ptr(post_shift), M, Kd, N, _ptr(out), out.stride(0), _stream()))
        return out
    if which == "f16x2_rows":
        # the epilogue leaves the row / tensor maxima of the output: for the caller's list, and (round 5) on a tensor this call created, where
        # the NEXT row-scaled layer finds them instead of running a max pass (inference chains: DCN's 1024-wide layers, the ESMM towers)
        carry = fresh and bits_out is None and DENSE_FWD_CARRY and N >= 16
        yb = _out_bits(bits_out if not carry else [], M, x.device)
        _lib.check(_lib.load().dir_dense_f16x2_rows_f32(_ptr(x), x.stride(0), _ptr(dense_bf3_image(weight, "f16x2")), _ptr(bias), 1 if relu else 0,
                                                        _ptr(post_scale), _ptr(post_shift), None, 0, M, Kd, N, _ptr(out), out.stride(0),
                                                        _ptr(row_bits), _ptr(yb[0]), _ptr(yb[1]), _stream()))
        if fresh and yb[0] is not None:
            _leave_hint(out, '_dir_bits', yb[0], yb[1])
        return out
    if use_bf3:
        _lib.check(_lib.load().dir_dense_bf16x3_f32(_ptr(x), x.stride(0), _ptr(dense_bf3_image(weight)), _ptr(bias), 1 if relu else 0,
                                                    _ptr(post_scale), _ptr(post_shift), None, 0, M, Kd, N, _ptr(out), out.stride(0), _stream()))
        return out
    if post_scale is not None:
        _lib.check(_lib.load().dir_dense_affine_f32(_ptr(x), x.stride(0), _ptr(weight), weight.stride(0), _ptr(bias), 1 if relu else 0,
                                                    _ptr(post_scale), _ptr(post_shift), M, Kd, N, _ptr(out), out.stride(0), _stream()))
        return out
    _lib.check(_lib.load().dir_dense_f32(_ptr(x), x.stride(0), _ptr(weight), weight.stride(0), _ptr(bias), 1 if relu else 0, M, Kd, N, _ptr(out),
                                         out.stride(0), _stream()))
    return out


def dense_head(x, weight, bias, head_w, relu=False, post_scale=None, post_shift=None, bounded=False):
    """act(x @ weight.T + bias) . head_w without writing the activation (include/dir_hip.h: dir_dense_bf16x3_head_f32): the last layer of a
    tower and its share of a following dense(1), e.g. DCN's deep branch (DeepCrossNetwork.py:136-137).  -> [M, 1], or None when the
    bf16x3 kernel does not cover the operands (the caller then runs the two steps).  bounded: x is bounded by construction (a
    batch-normalised activation): the fp16 x 2 form of the kernel (dir_dense_f16x2_head_f32) runs."""
    _dev(x, torch.float32, "x")
    _dev(weight, torch.float32, "weight")
    M, Kd = x.shape
    N = weight.shape[0]
    if weight.shape[1] != Kd or x.stride(1) != 1 or M < DENSE_BF3_MIN_ROWS or DENSE_ARITH not in ("auto", "bf16x3"):
        return None
    # (bounded here means a batch-normalised input: |x| <= sqrt(B) by the normalisation itself, not a promise about data; the weight image's
    # rows carry their own power-of-two scales)
    f16 = bool(bounded) and DENSE_BOUNDED_SPLIT in ("f16x2", "f16x2_rows") and DENSE_ARITH == "auto"
    if weight.stride(1) != 1 or weight.stride(0) % 4 or weight.data_ptr() % 16:
        weight = weight.contiguous()
    if not dense_bf16x3_covers(x, weight):
        return None
    bias = _dev(bias, torch.float32, "bias").contiguous() if bias is not None else None
    hw = _dev(head_w, torch.float32, "head_w").reshape(-1).contiguous()
    if hw.numel() != N:
        raise ValueError("dense_head: head_w [N]")
    if hw.data_ptr() % 16:                                # e.g. the tail of a wider weight row (DCN: columns d.. of the final dense(1))
        hw = hw.clone()
    if post_scale is not None:
        post_scale, post_shift = post_scale.contiguous(), post_shift.contiguous()
    lib = _lib.load()
    ncb = int(lib.dir_dense_bf16x3_head_blocks(N))
    part = torch.empty((ncb, M), dtype=torch.float32, device=x.device)
    fn = lib.dir_dense_f16x2_head_f32 if f16 else lib.dir_dense_bf16x3_head_f32
    _lib.check(fn(_ptr(x), x.stride(0), _ptr(dense_bf3_image(weight, "f16x2" if f16 else "bf16x3")), _ptr(bias), 1 if relu else 0, _ptr(post_scale),
                  _ptr(post_shift), M, Kd, N, _ptr(hw), None, 0, _ptr(part), _stream()))
    if ncb > 1 and M % 4 == 0:                            # the column blocks' partial dots, added in block order by ONE launch (ncb launches as torch ops)
        out = torch.empty(M, dtype=torch.float32, device=x.device)
        _lib.check(lib.dir_sum_partials_f32(_ptr(part), ncb, M, 0, _ptr(out), _stream()))
        return out.reshape(M, 1)
    out = part[0].clone() if ncb > 1 else part[0]
    for cb in range(1, ncb):                              # block order: a fixed order
        out += part[cb]
    return out.reshape(M, 1)


TOWER_MAX_WIDTH = 416
# Below this many rows the per-layer kernels run.  A one-launch forward costs one tile's time whatever the batch (63.5 us from a HIP graph for DeepFM
# at B <= 4096: profiles/r06_tower_min_rows.txt); the per-layer route takes 48 / 57 / 66 / 78 us at B = 256 / 512 / 1024 / 2048 -- break-even near
# 1000 rows (round 5's 4096 was set for tower_bf3_k's 128-row tiles and its serialised lookups)
TOWER_MIN_ROWS = int(os.environ.get("DIR_TOWER_MIN_ROWS", "1024"))
TOWER_GATHER = os.environ.get("DIR_TOWER_GATHER", "1")      # 0: DeepFM inference as two launches (packed gather, then the tower)
# "cs": tower_cs_k (round 6, default: 64-row workgroups, the layer input in LDS, a wave owns output columns and reads its weights straight from L2) /
# "rows": tower_bf3_k (a wave owns 16 rows and all columns) for the fp16 x 2 arithmetic; bf16 x 3 requests run tower_bf3_k under either setting.
# One box, A B A B (profiles/r06_tower_cs.txt, r06_tower_cs_gather.txt): on a given input 0.1965 against 0.2063 ms, DeepFM in one launch 0.2142
# against 0.2262, ESMM 0.3239 against 0.3372.  The two kernels agree to rounding, not bit for bit: ONE switch for the plain and the gather form keeps
# those two bitwise equal.
TOWER_KERNEL = os.environ.get("DIR_TOWER_KERNEL", "cs")
TOWER_MIN_WIDTH = 128          # dense.tower_infer: a stage always computes 13 column tiles, so a narrower layer (ESMM's 80-wide one) pads more
                               # than the fusion saves (ESMM forward 0.574 ms layer by layer, 0.580 fused)
TOWER = os.environ.get("DIR_TOWER", "auto")      # "0": never fuse (per-layer kernels)
_TOWER_IMAGES = {}


def tower_covers(x, weights, head=None):
    """Shapes dir_tower_bf16x3_f32 takes: 1..4 layers, every width and the input width multiples of 4 and <= 416."""
    if TOWER == "0" or not (1 <= len(weights) <= 4) or x.dim() != 2 or not x.is_cuda or x.dtype != torch.float32:
        return False
    dims = [x.shape[1]] + [int(w.shape[0]) for w in weights]
    if any(d % 4 or d > TOWER_MAX_WIDTH or d <= 0 for d in dims) or any(int(w.shape[1]) != dims[i] for i, w in enumerate(weights)):
        return False
    return x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 and (DENSE_ARITH in ("auto", "bf16x3"))


def tower_gather_covers(pt, weights):
    """Serving tables and layers dir_deepfm_tower_bf16x3_f32 accepts: K = 16 (one 16-column tile per slot), F <= 26, 1..4 layers of
    widths <= 416 that are multiples of 4, the first one reading the F*K-wide concatenation."""
    ws = list(weights)
    if TOWER == "0" or TOWER_GATHER == "0" or DENSE_ARITH not in ("auto", "bf16x3"):
        return False
    if pt is None or pt.K != 16 or pt.F > 26 or not (1 <= len(ws) <= 4):
        return False
    k = pt.F * pt.K
    for w in ws:
        if w.dim() != 2 or w.shape[1] != k or w.shape[0] % 4 or w.shape[0] > TOWER_MAX_WIDTH or w.dtype != torch.float32 or not w.is_cuda:
            return False
        k = int(w.shape[0])
    return True

# The split arithmetic of the fused tower (csrc/tower_bf3.hip): "f16x2" (two fp16 pieces per operand, three products: two thirds of the LDS
# traffic and half the matrix instructions; |activations|, |weights| < 65 504 -- the towers read embedding rows and ReLU activations) or
# "bf16x3" (three bf16 pieces, six products: fp32's exponent range).
TOWER_SPLIT = os.environ.get("DIR_TOWER_SPLIT", "f16x2")      # what split=None means in tower(): callers with unbounded inputs pass "bf16x3"
# split=None also checks what can be checked without a per-call sync: a weight or packed serving row at or above this magnitude (measured
# once per image build / per version of the tables) routes the launch to "bf16x3".  Activations stay the caller's contract (checking them
# would cost a pass and a sync per call): callers whose inputs are not embedding rows or ReLU activations of such pass split="bf16x3".
F16_RANGE_GUARD = 32768.0
# ... and a largest |value| BELOW this routes to bf16 x 3 as well: an unscaled fp16 x 2 element under 2^-3 carries an absolute error of 2^-25,
# i.e. 2^-25 / rms(x) relative on a dot product -- past the 1e-5 bar once a tensor's values sit around 2^-9 (round 5)
F16_SMALL_GUARD = 2.0 ** -6


def tower_image(weight, split=None):
    """The packed image (bf16 x 3 or fp16 x 2 pieces) of one layer's [N, K] weight in the tower kernel's k order, cached per tensor and
    split until the tensor is modified in place (tensor._version) or goes away."""
    import weakref
    split = split or TOWER_SPLIT
    key = (weight.data_ptr(), split)
    sig = (weight._version, tuple(weight.shape), tuple(weight.stride()))
    capturing = weight.is_cuda and torch.cuda.is_current_stream_capturing()
    hit = _TOWER_IMAGES.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == sig and not capture_bypasses_caches(weight):
        return hit[2]
    N, K = weight.shape
    lib = _lib.load()
    nbytes = int(lib.dir_tower_cs_image_bytes(K, N) if split == "f16x2cs" else lib.dir_tower_bf16x3_image_bytes(K, N))
    img = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    w = weight if weight.stride(1) == 1 else weight.contiguous()
    pack = {"f16x2": lib.dir_tower_f16x2_pack_f32, "f16x2cs": lib.dir_tower_cs_f16x2_pack_f32}.get(split, lib.dir_tower_bf16x3_pack_f32)
    _lib.check(pack(_ptr(w), w.stride(0), K, N, _ptr(img), nbytes, _stream()))
    if capturing:
        return img                                # (as dense_bf3_image: the pack is part of the graph, the image graph-owned, nothing cached)
    if len(_TOWER_IMAGES) > 256:
        _TOWER_IMAGES.clear()
    _TOWER_IMAGES[key] = (weakref.ref(weight), sig, img)
    return img


_WEIGHT_ABSMAX = {}


def weight_absmax(weight, every=1):
    """max |w| of a tensor, measured once per version (one reduction and one sync when the tensor changed: inference pays it once; nothing is
    packed for it -- ADVICE r4: the tower's guard used to build the fp16 x 2 image just to read this number).  every > 1: a measurement may
    be up to `every` in-place updates old (training loops: one pass and one sync per `every` steps, see TableSet.absmax)."""
    import weakref
    key = weight.data_ptr()
    sig = (weight._version, tuple(weight.shape))
    hit = _WEIGHT_ABSMAX.get(key)
    if hit is not None and hit[0]() is weight and hit[1][1] == sig[1]:
        moved = sig[0] - hit[1][0]
        # `every` covers only the bumps fused updaters made (mark_written's ledger); any other write re-measures
        if moved == 0 or (every > 1 and 0 < moved < every and hip_bumps(weight) - hit[3] == moved):
            return hit[2]
    if weight.is_cuda and torch.cuda.is_current_stream_capturing():
        if hit is not None and hit[0]() is weight:
            return hit[2]                          # the magnitude as last measured (a sync cannot be captured): one eager call measures it
        raise RuntimeError("weight_absmax: run one eager call before capturing a graph (the fp16 range guard reads the weights' magnitudes once)")
    if weight.numel():
        lo, hi = torch.aminmax(weight.detach())          # (one pass, no |w| temporary: the DIN item table is 2.56 GB)
        m = float(torch.maximum(hi, -lo))
    else:
        m = 0.0
    if len(_WEIGHT_ABSMAX) > 512:
        _WEIGHT_ABSMAX.clear()
    _WEIGHT_ABSMAX[key] = (weakref.ref(weight), sig, m, hip_bumps(weight))
    return m


def _tower_split_for(weights, pt=None):
    """What split=None resolves to: TOWER_SPLIT, or "bf16x3" when a weight or a packed row is outside F16_RANGE_GUARD."""
    if TOWER_SPLIT != "f16x2":
        return TOWER_SPLIT
    # tower_cs_k scales every stored row by its own power of two (round 6, DIR_TOWER_RS): the magnitude of the rows it looks up no longer matters
    row_scaled = TOWER_KERNEL == "cs" and os.environ.get("DIR_TOWER_RS", "1") != "0"
    if pt is not None and not row_scaled and not f16_range_ok(pt.absmax()):
        return "bf16x3"
    for w in weights:
        if not f16_range_ok(weight_absmax(w)):
            return "bf16x3"
    return "f16x2"


def tower(x, weights, biases=None, relu=True, post_scale=None, post_shift=None, head=None, adds=(), out=None, gather=None, split=None):
    """A DNN tower in one launch (include/dir_hip.h: dir_tower_bf16x3_f32; dnn_logit_fn, deepFM.py:284-319).  x [M, Kd]; weights: 1..4
    nn.Linear weights [N_l, K_l]; biases: list of [N_l] or None; relu: bool or per-layer list; post_scale / post_shift: per-layer lists
    of [N_l] vectors or None entries (the folded inference batch-norm).  head = (w [N_last] or [1, N_last], b [1]): -> logits [M, 1]
    (+ the [M] / [M, 1] tensors in adds, at most two); without a head -> the last activation [M, N_last].
    gather = (PackedTables, ids [M, F], linear bias [1] or None) with x = None: DeepFM inference in one launch
    (dir_deepfm_tower_bf16x3_f32) -- the input rows are looked up inside the kernel and the FM and first-order terms join the logit,
    bit for bit the result of gather_fm_linear + tower(..., adds=(fm, lin)).
    split: "f16x2" | "bf16x3" | None = TOWER_SPLIT (env DIR_TOWER_SPLIT): the kernel's split arithmetic (dir_tower_f16x2_f32 /
    dir_deepfm_tower_f16x2_f32 or the bf16x3 entries)."""
    if split not in (None, "f16x2", "bf16x3"):
        raise ValueError("tower: split must be 'f16x2' or 'bf16x3'")
    L = len(weights)
    if gather is not None:
        pt, ids, lin_bias = gather[:3]
        want_fm = bool(gather[3]) if len(gather) > 3 else True
        lin_col = getattr(pt, "lin_col", -1)
        _dev(ids, torch.int64, "ids")
        if x is not None or (head is None and (want_fm or lin_col >= 0)) or not tower_gather_covers(pt, weights):
            raise ValueError("tower(gather=...): x = None, a head for the FM / first-order terms, K = 16, F <= 26 and widths the tower covers")
        M, sb, sf = _onehot_strides(ids, pt.F)
        Kd = pt.F * pt.K
    else:
        _dev(x, torch.float32, "x")
        if not tower_covers(x, weights):
            raise ValueError("tower: 1..4 layers, input and layer widths multiples of 4 and <= %d, x 16-byte aligned with a row stride multiple of 4" % TOWER_MAX_WIDTH)
        M, Kd = x.shape
    if split is None:
        split = _tower_split_for([_dev(w, torch.float32, "weight") for w in weights], pt if gather is not None else None)
    lib = _lib.load()
    relu_l = list(relu) if isinstance(relu, (list, tuple)) else [bool(relu)] * L
    keep = []

    def vec(seq, l, n, what):
        t = seq[l] if seq is not None else None
        if t is None:
            return None
        t = _dev(t, torch.float32, what).reshape(-1).contiguous()
        if t.numel() != n:
            raise ValueError("tower: %s of layer %d must have %d entries" % (what, l, n))
        keep.append(t)
        return t.data_ptr()
    Ns = (ctypes.c_int * L)(*[int(w.shape[0]) for w in weights])
    acts = (ctypes.c_int * L)(*[1 if r else 0 for r in relu_l])
    # fp16 x 2: the column-split kernel (csrc/tower_cs.hip, round 6) unless DIR_TOWER_KERNEL=rows asks for tower_bf3_k's fp16 x 2 form
    cs = split == "f16x2" and TOWER_KERNEL == "cs"
    imgs_t = [tower_image(_dev(w, torch.float32, "weight"), "f16x2cs" if cs else split) for w in weights]
    f_tower = lib.dir_tower_cs_f16x2_f32 if cs else (lib.dir_tower_f16x2_f32 if split == "f16x2" else lib.dir_tower_bf16x3_f32)
    f_gather = lib.dir_deepfm_tower_cs_f16x2_f32 if cs else (lib.dir_deepfm_tower_f16x2_f32 if split == "f16x2" else lib.dir_deepfm_tower_bf16x3_f32)
    VP = ctypes.c_void_p * L
    imgs = VP(*[t.data_ptr() for t in imgs_t])
    b_arr = VP(*[vec(biases, l, Ns[l], "bias") for l in range(L)])
    s_arr = VP(*[vec(post_scale, l, Ns[l], "post_scale") for l in range(L)])
    h_arr = VP(*[vec(post_shift, l, Ns[l], "post_shift") for l in range(L)])
    if head is not None:
        hw = _dev(head[0], torch.float32, "head weight").reshape(-1).contiguous()
        hb = _dev(head[1], torch.float32, "head bias").reshape(-1).contiguous()
        if hw.numel() != Ns[L - 1] or hb.numel() != 1 or len(adds) > 2:
            raise ValueError("tower: head = (w [N_last], b [1]), at most two addends")
        add = []
        for a in adds:
            a = _dev(a, torch.float32, "addend").reshape(-1).contiguous()
            if a.numel() != M:
                raise ValueError("tower: an addend has one value per row")
            add.append(a)
        if out is None:
            out = torch.empty((M, 1), dtype=torch.float32, device=hw.device)
        if gather is not None:
            lb = _dev(lin_bias, torch.float32, "linear bias").reshape(-1) if lin_bias is not None else None
            _lib.check(f_gather(_ptr(pt.ptrs), _ptr(pt.vocab_dev), pt.F, pt.K, pt.ld, lin_col, _ptr(ids), sb, sf, int(want_fm), M, _ptr(lb),
                                                       L, Ns, imgs, b_arr, s_arr, h_arr, acts, _ptr(hw), _ptr(hb), _ptr(add[0]) if add else None,
                                                       _ptr(add[1]) if len(add) > 1 else None, _ptr(out), out.stride(0), _stream()))
            return out
        _lib.check(f_tower(_ptr(x), x.stride(0), M, Kd, L, Ns, imgs, b_arr, s_arr, h_arr, acts, _ptr(hw), _ptr(hb),
                                            _ptr(add[0]) if add else None, _ptr(add[1]) if len(add) > 1 else None, _ptr(out), out.stride(0), _stream()))
        return out
    if adds:
        raise ValueError("tower: addends need a head")
    if gather is not None:
        if out is None:
            out = torch.empty((M, Ns[L - 1]), dtype=torch.float32, device=ids.device)
        _lib.check(f_gather(_ptr(pt.ptrs), _ptr(pt.vocab_dev), pt.F, pt.K, pt.ld, lin_col, _ptr(ids), sb, sf, 0, M, None, L, Ns,
                                                   imgs, b_arr, s_arr, h_arr, acts, None, None, None, None, _ptr(out), out.stride(0), _stream()))
        return out
    if out is None:
        out = torch.empty((M, Ns[L - 1]), dtype=torch.float32, device=x.device)
    _lib.check(f_tower(_ptr(x), x.stride(0), M, Kd, L, Ns, imgs, b_arr, s_arr, h_arr, acts, None, None, None, None, _ptr(out),
                                        out.stride(0), _stream()))
    return out


def dense_gated(x, weight, gate, out=None, arith=None, row_bits=None, bits_out=None):
    """where(gate > 0, x @ weight.T, 0) (include/dir_hip.h: dir_dense_gated_f32 / dir_dense_bf16x3_f32 with a gate): x [M, Kd],
    weight [N, Kd], gate [M, N].  row_bits (grad_bits(x)[0]): where "auto" would run bf16x3 the row-scaled fp16 x 2 kernel runs."""
    _dev(x, torch.float32, "x")
    _dev(weight, torch.float32, "weight")
    _dev(gate, torch.float32, "gate")
    M, Kd = x.shape
    N = weight.shape[0]
    if weight.shape[1] != Kd or x.stride(1) != 1 or tuple(gate.shape) != (M, N) or gate.stride(1) != 1:
        raise ValueError("dense_gated: x [M, Kd], weight [N, Kd], gate [M, N], unit inner strides")
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    use_bf3 = _dense_arith(arith, x, weight, out, gate) == "bf16x3"
    if not use_bf3 and (weight.stride(1) != 1 or weight.stride(0) % 4 or weight.data_ptr() % 16):
        weight = weight.contiguous()
    if use_bf3 and row_bits is not None and (arith or DENSE_ARITH) == "auto":
        yb = _out_bits(bits_out, M, x.device)
        _lib.check(_lib.load().dir_dense_f16x2_rows_f32(_ptr(x), x.stride(0), _ptr(dense_bf3_image(weight, "f16x2")), None, 0, None, None, _ptr(gate),
                                                        gate.stride(0), M, Kd, N, _ptr(out), out.stride(0), _ptr(row_bits), _ptr(yb[0]), _ptr(yb[1]),
                                                        _stream()))
        return out
    if use_bf3:
        _lib.check(_lib.load().dir_dense_bf16x3_f32(_ptr(x), x.stride(0), _ptr(dense_bf3_image(weight)), None, 0, None, None, _ptr(gate),
                                                    gate.stride(0), M, Kd, N, _ptr(out), out.stride(0), _stream()))
        return out
    _lib.check(_lib.load().dir_dense_gated_f32(_ptr(x), x.stride(0), _ptr(weight), weight.stride(0), _ptr(gate), gate.stride(0), M, Kd, N,
                                               _ptr(out), out.stride(0), _stream()))
    return out


DENSE_DW_MIN_ROWS = 8192
DENSE_DW_ARITH = os.environ.get("DIR_DENSE_DW_ARITH", "auto")        # "f32": the weight gradients of the towers stay on the library


def dense_dw_auto_arith(M, N, K):
    """"bf16x3" where dir_dense_dw_bf16x3_f32 (weight and bias gradient in one pass; dir_dense_dw_f16x2_f32 when the caller has the scales)
    is the faster formulation.  tools/small_dw_probe.py at 65 536 rows (profiles/r04_small_dw_probe.txt): it beats the library GEMM + column
    sum at every output from 40 x 80 to 1024 x 128 (80 x 200: 47 vs 75 us, 200 x 80: 34 vs 76, 360 x 416: 190 vs 243, 320 x 320: 153 vs 165)
    and the fp32 FMA kernel "small" wherever that applies (80 x 64: 30 vs 40 us, 80 x 200: 47 vs 78) -- except very thin outputs (16 x 416:
    51 vs 43) and a row block that is at least half padding under a long reduction-free side (128 x 1024: 188 vs 147).  Below
    DENSE_DW_MIN_ROWS rows: "small" for outputs up to 128 x 256, otherwise "f32" (the library GEMM in row slices)."""
    if DENSE_ARITH == "f32" or DENSE_DW_ARITH == "f32" or N % 4 or K % 4:
        return "f32"
    fits_small = N <= 128 and K <= 256 and -(-N // 8) * -(-K // 8) <= 256 and M >= 2048 and N * K >= 512
    if M < DENSE_DW_MIN_ROWS or N < 32:
        return "small" if fits_small else "f32"
    if -(-N // 256) * 256 >= 2 * N and N * K > 60000:
        return "f32"
    return "bf16x3"


def dense_dw(g, x, arith=None, want_bias=False, g_bits=None, x_bits=None):
    """dW [N, K] = g^T x, the kernel gradient of a dense layer (include/dir_hip.h: dir_dense_dw_bf16x3_f32): g [M, N], x [M, K], unit
    inner strides.  arith None / "auto": dense_dw_auto_arith; "f32": the library GEMM in row slices; "bf16x3": the MFMA kernel;
    "small": the fp32 FMA kernel for N <= 128, K <= 256 (dir_dense_dw_small_f32).
    want_bias: -> (dW, db) with db [N] = g.sum(0), the bias gradient (in the kernel's pass over g on the bf16x3 path).
    g_bits (grad_bits(g)[1]) given by a caller who ALSO vouches that x is bounded by construction (an embedding concatenation, an
    activation of such a tower): where "auto" would run bf16x3, dir_dense_dw_f16x2_f32 runs; "f16x2" asks for it by name.
    x_bits (a [1] int32 device tensor: the bit pattern of an upper bound of max |x|, e.g. the all_bits the forward's row-scaled kernel or
    max pass left for this input): x is scaled by a power of two as well (dir_dense_dw_f16x2_scaled_f32) and nothing has to be vouched."""
    _dev(g, torch.float32, "g")
    _dev(x, torch.float32, "x")
    if g.dim() != 2 or x.dim() != 2 or g.shape[0] != x.shape[0] or g.stride(1) != 1 or x.stride(1) != 1:
        raise ValueError("dense_dw: g [M, N], x [M, K], unit inner strides")
    M, N = g.shape
    K = x.shape[1]
    arith = arith or DENSE_ARITH
    covered = N % 4 == 0 and K % 4 == 0 and g.stride(0) % 4 == 0 and x.stride(0) % 4 == 0 and g.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0
    if arith == "auto":
        arith = dense_dw_auto_arith(M, N, K) if covered else "f32"
        if arith == "bf16x3" and g_bits is not None and DENSE_BWD_SPLIT == "f16x2":
            arith = "f16x2"
    if arith == "f16x2" and g_bits is None:
        raise ValueError("dense_dw(arith='f16x2') needs g_bits (grad_bits(g)[1])")
    if arith in ("bf16x3", "small", "f16x2") and not covered:
        raise ValueError("dense_dw(arith='%s'): N, K and the row strides must be multiples of 4, g and x 16-byte aligned" % arith)
    if arith == "small" and (N > 128 or K > 256 or -(-N // 8) * -(-K // 8) > 256):
        raise ValueError("dense_dw(arith='small') covers N <= 128, K <= 256 (at most 256 register tiles of 8 x 8)")
    if arith == "f32":
        if M >= 8192 and M % 16 == 0 and g.is_contiguous() and x.is_contiguous():
            dW = torch.bmm(g.view(16, M // 16, N).transpose(1, 2), x.view(16, M // 16, K)).sum(dim=0)
        else:
            dW = g.t() @ x
        return (dW, g.sum(dim=0)) if want_bias else dW
    if arith not in ("bf16x3", "small", "f16x2"):
        raise ValueError("dense_dw: arith must be 'auto', 'f32', 'bf16x3', 'f16x2' or 'small'")
    lib = _lib.load()
    if arith == "f16x2":
        nbytes = int(lib.dir_dense_dw_bf16x3_workspace_bytes(M, N, K))
        ws = torch.empty(max(16, nbytes), dtype=torch.uint8, device=g.device)
        dW = torch.empty((N, K), dtype=torch.float32, device=g.device)
        db = torch.empty(N, dtype=torch.float32, device=g.device) if want_bias else None
        if x_bits is not None:
            _lib.check(lib.dir_dense_dw_f16x2_scaled_f32(_ptr(g), g.stride(0), _ptr(x), x.stride(0), M, N, K, _ptr(dW), dW.stride(0), _ptr(db), _ptr(ws),
                                                         nbytes, _ptr(g_bits), _ptr(x_bits), _stream()))
        else:
            _lib.check(lib.dir_dense_dw_f16x2_f32(_ptr(g), g.stride(0), _ptr(x), x.stride(0), M, N, K, _ptr(dW), dW.stride(0), _ptr(db), _ptr(ws), nbytes,
                                                  _ptr(g_bits), _stream()))
        return (dW, db) if want_bias else dW
    wsq, run = ((lib.dir_dense_dw_bf16x3_workspace_bytes, lib.dir_dense_dw_bf16x3_f32) if arith == "bf16x3" else
                (lib.dir_dense_dw_small_workspace_bytes, lib.dir_dense_dw_small_f32))
    nbytes = int(wsq(M, N, K))
    ws = torch.empty(max(16, nbytes), dtype=torch.uint8, device=g.device)
    dW = torch.empty((N, K), dtype=torch.float32, device=g.device)
    db = torch.empty(N, dtype=torch.float32, device=g.device) if want_bias else None
    _lib.check(run(_ptr(g), g.stride(0), _ptr(x), x.stride(0), M, N, K, _ptr(dW), dW.stride(0), _ptr(db), _ptr(ws), nbytes, _stream()))
    return (dW, db) if want_bias else dW


def units1_relu_backward_supported(y):
    """Shapes dir_units1_relu_backward_f32 takes: a float32 CUDA [B, N] activation with N and its row stride multiples of 4, N <= 4096."""
    return (y.is_cuda and y.dtype == torch.float32 and y.dim() == 2 and y.stride(1) == 1 and y.shape[1] % 4 == 0 and y.shape[1] <= 4096
            and y.stride(0) % 4 == 0 and y.data_ptr() % 16 == 0)


def units1_relu_backward(g, w, y, want_bits=False):
    """Backward of logit = y . w + bias (units = 1; deepFM.py:311-317, ESMM.py:146) through y = relu(pre) in one pass
    (include/dir_hip.h: dir_units1_relu_backward_f32).  g [B] or [B, 1] = dL/dlogit, w [N] or [1, N], y [B, N] ->
    (dpre [B, N] = where(y > 0, g * w, 0), dw [N] = sum_b g * y, dbias_y [N] = sum_b dpre).  want_bits: a fourth result, (row_bits [B],
    all_bits [1]) int32 -- upper bounds of dpre's row maxima and of max |dpre| as bit patterns (dir_units1_relu_backward_bits_f32), what
    grad_bits(dpre) would otherwise make a pass over dpre for; None when the fp16 x 2 backward is off."""
    _dev(g, torch.float32, "g")
    _dev(w, torch.float32, "w")
    _dev(y, torch.float32, "y")
    B, N = y.shape
    if g.numel() != B or w.numel() != N:
        raise ValueError("units1_relu_backward: g [B], w [N], y [B, N]")
    if not units1_relu_backward_supported(y):
        raise ValueError("units1_relu_backward: y must be float32 [B, N] with N and its row stride multiples of 4 (N <= 4096), 16-byte aligned")
    g = g.reshape(B).contiguous()
    w = w.reshape(N).contiguous()
    if w.data_ptr() % 16:                                   # (a slice of a wider weight row: DCN's logit weights behind the cross columns)
        w = w.clone()
    lib = _lib.load()
    gx = torch.empty((B, N), dtype=torch.float32, device=y.device)
    P = int(lib.dir_units1_relu_backward_partials(B, N))
    if P == 0:
        z = torch.zeros(N, dtype=torch.float32, device=y.device)
        return (gx, z, z.clone(), None) if want_bits else (gx, z, z.clone())
    part = torch.empty((P, 2, N), dtype=torch.float32, device=y.device)
    bits = None
    if want_bits and DENSE_BWD_CARRY and DENSE_BWD_SPLIT == "f16x2" and DENSE_ARITH == "auto" and B >= DENSE_BF3_MIN_ROWS:
        buf = torch.empty(B + 4, dtype=torch.int32, device=y.device)
        bits = (buf[:B], buf[B:B + 1])
        _lib.check(lib.dir_units1_relu_backward_bits_f32(_ptr(g), _ptr(w), _ptr(y), y.stride(0), B, N, _ptr(gx), gx.stride(0), _ptr(part), P,
                                                         _ptr(bits[0]), _ptr(bits[1]), _stream()))
    else:
        _lib.check(lib.dir_units1_relu_backward_f32(_ptr(g), _ptr(w), _ptr(y), y.stride(0), B, N, _ptr(gx), gx.stride(0), _ptr(part), P, _stream()))
    s = part[0] if P == 1 else part.sum(dim=0)
    return (gx, s[1], s[0], bits) if want_bits else (gx, s[1], s[0])


def units1(x, w, bias=None, out=None):
    """logit = x . w + bias, a dense layer with one unit (include/dir_hip.h: dir_units1_f32; deepFM.py:311-317, ESMM.py:146,
    DeepCrossNetwork.py:136-137): x [B, N] (unit column stride), w [N] or [1, N], bias [1] or None -> [B, 1]."""
    _dev(x, torch.float32, "x")
    _dev(w, torch.float32, "w")
    if x.dim() != 2 or w.numel() != x.shape[1] or (x.shape[0] > 0 and x.stride(1) != 1):
        raise ValueError("units1: x [B, N] with unit column stride, w [N]")
    B, N = x.shape
    w = w.reshape(N).contiguous()
    if bias is not None:
        bias = _dev(bias, torch.float32, "bias").reshape(-1)
        if bias.numel() != 1:
            raise ValueError("units1: bias holds one value")
    if out is None:
        out = torch.empty((B, 1), dtype=torch.float32, device=x.device)
    elif tuple(out.shape) != (B, 1) or out.dtype != torch.float32 or not out.is_cuda:
        raise ValueError("units1: out [B, 1] float32 on the GPU")
    _lib.check(_lib.load().dir_units1_f32(_ptr(x), x.stride(0) if B > 0 else N, B, N, _ptr(w), _ptr(bias), _ptr(out), out.stride(0) if B > 0 else 1,
                                          _stream()))
    return out


def units1_backward(g, w, x, want_gx=True):
    """Backward of logit = x . w (units = 1) for an activation of any width that is not a ReLU output (include/dir_hip.h:
    dir_units1_backward_f32; DCN's cross output under DeepCrossNetwork.py:136-137).  g [B] or [B, 1], w [N] or [1, N], x [B, N] (unit column
    stride) -> (gx [B, N] = g * w | None, dw [N] = sum_b g * x)."""
    _dev(g, torch.float32, "g")
    _dev(w, torch.float32, "w")
    _dev(x, torch.float32, "x")
    B, N = x.shape
    if g.numel() != B or w.numel() != N or (B > 0 and x.stride(1) != 1):
        raise ValueError("units1_backward: g [B], w [N], x [B, N] with unit column stride")
    g = g.reshape(B).contiguous()
    w = w.reshape(N).contiguous()
    lib = _lib.load()
    gx = torch.empty((B, N), dtype=torch.float32, device=x.device) if want_gx else None
    dw = torch.empty(N, dtype=torch.float32, device=x.device)
    P = int(lib.dir_units1_backward_partials(B, N))
    part = torch.empty((max(P, 1), N), dtype=torch.float32, device=x.device)
    _lib.check(lib.dir_units1_backward_f32(_ptr(g), _ptr(w), _ptr(x), x.stride(0) if B > 0 else N, B, N, _ptr(gx), N, _ptr(dw), _ptr(part), P,
                                           _stream()))
    return gx, dw


def adagrad_dense_(w, accum, grad, lr, eps=0.0):
    """Adagrad step of a dense variable in place (include/dir_hip.h: dir_adagrad_dense_f32; deepFM.py:61): accum += grad^2,
    w -= lr * grad / (sqrt(accum) + eps).  Contiguous float32 CUDA tensors of one shape."""
    for t, n in ((w, "w"), (accum, "accum"), (grad, "grad")):
        _dev(t, torch.float32, n)
        if not t.is_contiguous() or t.shape != w.shape:
            raise ValueError("adagrad_dense_: w / accum / grad must be contiguous tensors of one shape")
    _lib.check(_lib.load().dir_adagrad_dense_f32(_ptr(w), _ptr(accum), _ptr(grad), w.numel(), float(lr), float(eps), _stream()))
    mark_written(w, accum)
    return w


def adagrad_dense_multi_(ws, accums, grads, lr, eps=0.0):
    """adagrad_dense_ on a list of variables in one launch per 16 (include/dir_hip.h: dir_adagrad_dense_multi_f32): the pointers travel in
    the kernel arguments.  Contiguous float32 CUDA tensors; ws[i], accums[i], grads[i] of one shape."""
    n = len(ws)
    if n == 0:
        return
    for w, a, g in zip(ws, accums, grads):
        if not (w.is_cuda and w.dtype == torch.float32 and a.dtype == torch.float32 and g.dtype == torch.float32 and w.is_contiguous()
                and a.is_contiguous() and g.is_contiguous() and a.shape == w.shape and g.shape == w.shape):
            raise ValueError("adagrad_dense_multi_: contiguous float32 CUDA tensors, one shape per variable")
    P = ctypes.c_void_p * n
    L = ctypes.c_int64 * n
    _lib.check(_lib.load().dir_adagrad_dense_multi_f32(P(*[w.data_ptr() for w in ws]), P(*[a.data_ptr() for a in accums]),
                                                       P(*[g.data_ptr() for g in grads]), L(*[w.numel() for w in ws]), n, float(lr), float(eps),
                                                       _stream()))
    mark_written(*ws, *accums)


def ftrl_dense_(w, accum, linear, grad, lr, l1=0.0, l2=0.0):
    """FTRL-Proximal step of a dense variable in place (include/dir_hip.h: dir_ftrl_dense_f32; the linear bias under deepFM.py:58):
    w, accum (n), linear (z), grad: contiguous float32 CUDA tensors of one shape."""
    for t, n in ((w, "w"), (accum, "accum"), (linear, "linear"), (grad, "grad")):
        _dev(t, torch.float32, n)
        if not t.is_contiguous() or t.shape != w.shape:
            raise ValueError("ftrl_dense_: w / accum / linear / grad must be contiguous tensors of one shape")
    _lib.check(_lib.load().dir_ftrl_dense_f32(_ptr(w), _ptr(accum), _ptr(linear), _ptr(grad), w.numel(), float(lr), float(l1), float(l2), _stream()))
    mark_written(w, accum, linear)
    return w


def bn_train_supported(y):
    """Shapes the training-mode batch-norm kernels take (include/dir_hip.h: dir_bn_train_stats_f32): the same class as
    units1_relu_backward_supported, at least one row."""
    return units1_relu_backward_supported(y) and y.shape[0] > 0


def bn_train_stats(y, gamma, beta, moving_mean, moving_var, eps, momentum):
    """Batch statistics of y [B, N] in one read (include/dir_hip.h: dir_bn_train_stats_f32; deepFM.py:303-308, DeepCrossNetwork.py:400-403):
    -> (mean, inv = rsqrt(var + eps), scale = inv * gamma, shift = beta - mean * scale), each [N]; the moving statistics (tensors or None)
    are updated in place as moving * momentum + batch * (1 - momentum).  The normalised activation is y * scale + shift."""
    _dev(y, torch.float32, "y")
    if not bn_train_supported(y):
        raise ValueError("bn_train_stats: y must be float32 [B > 0, N] with N and its row stride multiples of 4 (N <= 4096), 16-byte aligned")
    B, N = y.shape
    lib = _lib.load()
    P = int(lib.dir_bn_train_partials(B, N))
    out = torch.empty((4, N), dtype=torch.float32, device=y.device)
    part = torch.empty((P, 2, N), dtype=torch.float32, device=y.device)
    vecs = [None if t is None else t.detach() for t in (gamma, beta, moving_mean, moving_var)]
    for t in vecs:
        if t is not None and (t.dtype != torch.float32 or t.numel() != N or not t.is_contiguous() or t.device != y.device):
            raise ValueError("bn_train_stats: gamma / beta / moving statistics must be contiguous float32 [N] tensors on y's device")
    _lib.check(lib.dir_bn_train_stats_f32(_ptr(y), y.stride(0), B, N, float(eps), float(momentum), _ptr(vecs[0]), _ptr(vecs[1]), _ptr(vecs[2]),
                                          _ptr(vecs[3]), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(out[3]), _ptr(part), P, _stream()))
    mark_written(moving_mean, moving_var)
    return out[0], out[1], out[2], out[3]


def bn_train_backward(g, y, mean, inv, gamma=None, relu_gate=False):
    """Backward of the training-mode batch norm (include/dir_hip.h: dir_bn_train_backward_f32): g = dL/d(y * scale + shift) [B, N] ->
    (gy [B, N], gbeta [N], ggamma [N]); relu_gate: gy is zeroed where y <= 0 (y = relu(pre) of the layer below: gy is dL/dpre)."""
    _dev(g, torch.float32, "g")
    _dev(y, torch.float32, "y")
    if g.shape != y.shape or not bn_train_supported(y) or not bn_train_supported(g):
        raise ValueError("bn_train_backward: g and y must be float32 [B > 0, N] with N and the row strides multiples of 4, 16-byte aligned")
    B, N = y.shape
    lib = _lib.load()
    P = int(lib.dir_bn_train_partials(B, N))
    gy = torch.empty((B, N), dtype=torch.float32, device=y.device)
    vec = torch.empty((5, N), dtype=torch.float32, device=y.device)           # gbeta, ggamma, the three coefficients
    part = torch.empty((P, 2, N), dtype=torch.float32, device=y.device)
    gam = None if gamma is None else gamma.detach().contiguous()
    _lib.check(lib.dir_bn_train_backward_f32(_ptr(g), g.stride(0), _ptr(y), y.stride(0), B, N, _ptr(mean.contiguous()), _ptr(inv.contiguous()),
                                             _ptr(gam), 1 if relu_gate else 0, _ptr(gy), gy.stride(0), _ptr(vec[0]), _ptr(vec[1]), _ptr(vec[2]),
                                             _ptr(part), P, _stream()))
    return gy, vec[0], vec[1]


def din_backward_supported(K, T, H1, H2):
    """Shapes the fused DIN backward covers (include/dir_hip.h: dir_din_attention_pool_backward_f32)."""
    return K == 64 and T <= 64 and H1 <= 80 and H2 <= 48 and H1 % 4 == 0 and H2 % 4 == 0


_DIN_BWD_WS = {}


def din_attention_pool_backward(table, hist, hist_len, cand, W1, b1, W2, b2, W3, b3, g, normalize=False, scores=None, saved=None):
    """Backward of din_attention_pool given g = dL/dout [B, K] (include/dir_hip.h: dir_din_attention_pool_backward_f32 + the
    per-sample term's two small GEMMs).  scores [B, T]: the forward's attention weights (din_attention_pool(..., want_scores=True));
    with them the two-kernel form runs (dir_din_attention_pool_backward_rows_f32: no softmax recompute, wave-per-sample row pass +
    streaming weight-gradient pass); DIR_DIN_BWD_ROWS=0 keeps the round-1 single kernel.  saved: the (plan, workspace) of
    din_attention_pool_save on the same inputs -- the row pass then reads the hidden activations instead of recomputing them.  -> dict:
      ids_h [N] int64, gh [N, K]: the valid history positions' table rows and their gradients, (b, j) order;
      ga [B, K]: the candidate rows' gradients;  gW1 [4K, H1], gb1, gW2, gb2, gW3 [H2], gb3 [1]."""
    _dev(table, torch.float32, "table")
    _dev(hist, torch.int64, "hist")
    _dev(cand, torch.int64, "cand")
    _dev(g, torch.float32, "g")
    if hist_len is not None:
        _dev(hist_len, torch.int32, "hist_len")
    B, T = hist.shape
    K = table.shape[1]
    H1, H2 = W1.shape[1], W2.shape[1]
    if tuple(W1.shape) != (4 * K, H1) or tuple(W2.shape) != (H1, H2) or W3.numel() != H2 or tuple(g.shape) != (B, K):
        raise ValueError("DIN backward: W1 [4K,H1], W2 [H1,H2], W3 [H2], g [B,K]")
    lib = _lib.load()
    need = int(lib.dir_din_backward_workspace_bytes(K, H1, H2))
    if need <= 0 or T > 64:
        raise _lib.DirError(-4, "din_attention_pool_backward: the fused backward covers K = 64, H1 <= 80, H2 <= 48, T <= 64")
    dev = table.device
    args = [t.contiguous() for t in (W1, b1, W2, b2, W3, b3)]
    for t in args:
        _dev(t, torch.float32, "DIN weight")
    hist, g, cand = hist.contiguous(), g.contiguous(), cand.contiguous()
    if hist_len is not None:
        hist_len = hist_len.contiguous()
    if B == 0:                                           # an empty batch: zero gradients, nothing to launch (empty tensors have no storage)
        f32 = dict(dtype=torch.float32, device=dev)
        z = torch.zeros((0, K), **f32)
        return {"ids_h": hist.reshape(-1)[:0], "gh": z, "ga": z, "grows": z, "gW1": torch.zeros((4 * K, H1), **f32), "gb1": torch.zeros(H1, **f32),
                "gW2": torch.zeros((H1, H2), **f32), "gb2": torch.zeros(H2, **f32), "gW3": torch.zeros(H2, **f32), "gb3": torch.zeros(1, **f32)}
    use_rows = scores is not None and os.environ.get("DIR_DIN_BWD_ROWS", "1") != "0"
    if saved is not None and not use_rows:
        raise ValueError("DIN backward: saved activations need the forward's scores")
    plan = saved[0] if saved is not None else DinTrainPlan(hist, hist_len)
    valid, row_off, tile_off, N, n_tiles = plan.valid, plan.row_off, plan.tile_off, plan.N, plan.n_tiles
    if use_rows:
        _dev(scores, torch.float32, "scores")
        if tuple(scores.shape) != (B, T) or not scores.is_contiguous():
            raise ValueError("DIN backward: scores must be a contiguous [B, T] tensor")
    ws = _DIN_BWD_WS.get(dev)
    if ws is None or ws.numel() < need:
        ws = _DIN_BWD_WS[dev] = torch.empty(need, dtype=torch.uint8, device=dev)
    f32 = dict(dtype=torch.float32, device=dev)
    grows = torch.empty((N + B, K), **f32)                 # the table gradient's rows in one buffer: history rows, then the candidates'
    gh, ga, S = grows[:N], grows[N:], torch.empty((B, H1), **f32)
    gAP, gW2, gb2 = torch.empty((2 * K, H1), **f32), torch.empty((H1, H2), **f32), torch.empty(H2, **f32)
    gW3, gb3 = torch.empty(H2, **f32), torch.empty(1, **f32)
    if use_rows:
        need2 = int(lib.dir_din_backward_rows_workspace_bytes(K, H1, H2, n_tiles))
        if saved is not None:
            ws2 = saved[1]
            if ws2.numel() < need2:
                raise ValueError("DIN backward: the saved workspace is smaller than this batch needs")
        else:
            ws2 = _DIN_BWD_WS.get((dev, "rows"))
            if ws2 is None or ws2.numel() < need2:
                ws2 = _DIN_BWD_WS[(dev, "rows")] = torch.empty(need2, dtype=torch.uint8, device=dev)
        entry = lib.dir_din_attention_pool_backward_saved_f32 if saved is not None else lib.dir_din_attention_pool_backward_rows_f32
        _lib.check(entry(
            _ptr(table), K, _ptr(hist), _ptr(hist_len), _ptr(cand), T, _ptr(args[0]), _ptr(args[1]), H1, _ptr(args[2]), _ptr(args[3]), H2,
            _ptr(args[4]), _ptr(args[5]), int(bool(normalize)), B, _ptr(g), _ptr(scores), _ptr(row_off), _ptr(tile_off), n_tiles,
            _ptr(gh), _ptr(ga), _ptr(S), _ptr(gAP), _ptr(gW2), _ptr(gb2), _ptr(gW3), _ptr(gb3), _ptr(ws2), need2, _stream()))
    else:
        _lib.check(lib.dir_din_attention_pool_backward_f32(
            _ptr(table), K, _ptr(hist), _ptr(hist_len), _ptr(cand), T, _ptr(args[0]), _ptr(args[1]), H1, _ptr(args[2]), _ptr(args[3]), H2,
            _ptr(args[4]), _ptr(args[5]), int(bool(normalize)), B, _ptr(g), _ptr(row_off), _ptr(gh), _ptr(ga), _ptr(S), _ptr(gAP),
            _ptr(gW2), _ptr(gb2), _ptr(gW3), _ptr(gb3), _ptr(ws), _stream()))
    # the per-sample term a.(Wa - Wd) + b1: ga += S C^T (dir_dense_f32, C is its [out, in] weight) and C's gradient S^T a
    W1 = args[0]
    C = W1[K:2 * K] - W1[2 * K:3 * K]
    a = table[cand.clamp(min=0)] * (cand >= 0).unsqueeze(1)
    if _DIN_GA_LIBRARY or not dense_supported(S, C):
        ga.addmm_(S, C.t())                           # (uncovered widths; DIR_DIN_GA_LIBRARY=1: development A/B switch)
    else:
        ga.add_(dense(S, C.contiguous(), None, relu=False))
    gCt, gb1 = dense_dw(S, a, want_bias=True)            # S^T a [H1, K] and S's column sums in one pass (dir_dense_dw_small_f32 at 80 x 64)
    gC = gCt.t()
    gA, gWp = gAP[:K], gAP[K:]
    return {"ids_h": _masked_rows(hist, valid, N), "gh": gh, "ga": ga, "grows": grows, "gW1": torch.cat([gA, gC, gA - gC, gWp], dim=0), "gb1": gb1,
            "gW2": gW2, "gb2": gb2, "gW3": gW3, "gb3": gb3}


_DIN_GA_LIBRARY = os.environ.get("DIR_DIN_GA_LIBRARY", "0") == "1"

# default arithmetic of cin_layer: "auto" (the split kernels on the shapes they cover, fp32 MFMA otherwise) | "f32" | "bf16x3" | "f16x2"
CIN_ARITH = os.environ.get("DIR_CIN_ARITH", "auto")
# which split arithmetic "auto" gives a FORWARD layer (csrc/cin_bf3.hip): "f16x2" (two fp16 pieces per operand, three products: half the
# matrix instructions; operands are embeddings / activations / weights) or "bf16x3" (three bf16 pieces, six products: fp32's exponent
# range).  Layers whose left operand is a gradient (the backward's data-gradient forms) always take bf16x3.
CIN_FWD_SPLIT = os.environ.get("DIR_CIN_FWD_SPLIT", "f16x2")


# the split of the CIN BACKWARD kernels whose left operand is a gradient (data-gradient form, forward-form contractions, weight gradient):
# "f16x2" = two fp16 pieces with the gradient scaled by a power of two (per row / per tensor; include/dir_hip.h:
# dir_cin_layer_dot_add_f16x2_f32, dir_cin_layer_grad_f16x2_f32, dir_cin_dw_f16x2_f32), "bf16x3" = rounds 2-3's arithmetic
CIN_BWD_SPLIT = os.environ.get("DIR_CIN_BWD_SPLIT", "f16x2")


def cin_bf16x3_covers(m, D):
    """Shapes dir_cin_layer_bf16x3_f32 accepts (csrc/cin_bf3.hip)."""
    return 1 <= m <= 40 and D in (4, 8, 16, 32)


def cin_auto_arith(m, D, Hp, H):
    """What arith="auto" runs: the bf16x3 kernel where it accepts the shape and its padding does not eat its advantage.  It computes
    columns in blocks of 128 plus one last block of 32 / 64 / 96 / 128 and i in blocks of 32 (Hp <= 32) or 64, at 1.7-2 x the
    fp32-MFMA kernel's rate per padded product (profiles/r02_sweep_shapes.md): it is chosen while padded work <= 1.8 x real work."""
    if not cin_bf16x3_covers(m, D):
        return "f32"
    hpad = -(-H // 32) * 32
    ipad = 32 if Hp <= 32 else -(-Hp // 64) * 64
    return "bf16x3" if hpad * ipad <= 1.8 * H * Hp else "f32"


# The forward layers leave their output's row maxima on the tensor (dir_cin_layer1_bits_f16x2_f32, xout_row_bits).  The next layer does NOT
# read them instead of scanning its rows (measured slower: 2.31-2.43 against 2.20-2.33 ms for the 128 x 128 layer at B = 65 536,
# profiles/r05_cin_rs_probe.txt: the scan doubles as a prefetch burst of the workgroup's xk slice); it uses them for the DEVICE-SIDE VERDICT
# of dir_cin_layer_auto_f16x2_f32: inside the magnitude window where the unscaled split is as accurate, the plain kernel runs (2.03-2.13 ms),
# outside it the row-scaled one.  DIR_CIN_ROW_BITS_CARRY=0: no maxima are left, every later layer runs the row-scaled kernel.
CIN_ROW_BITS_CARRY = os.environ.get("DIR_CIN_ROW_BITS_CARRY", "1") != "0"
CIN_L1_PAIRS = os.environ.get("DIR_CIN_L1_PAIRS", "1") != "0"      # development switch: 0 runs a stack's first layer on the general kernel
CIN_POOLED_LAST = os.environ.get("DIR_CIN_POOLED_LAST", "1") != "0"  # development switch: 0 runs a pooled-only layer on the layer kernel


# development switch: 0 = the pooled last layer of an inference stack stays on the two-pass form (cin_pool_z + the dense kernel)
CIN_POOLED_FUSED = os.environ.get("DIR_CIN_POOLED_FUSED", "1") != "0"
_CIN_POOLED_IMAGES = {}


def cin_pooled_fused_covers(m, Hp, H, D):
    """Shapes dir_cin_pooled_last_bf16x3_f32 takes (include/dir_hip.h)."""
    return D == 16 and 1 <= m <= 32 and Hp >= 1 and 4 <= H <= 128 and H % 4 == 0


def cin_pooled_image(W, m, Hp, D, owner=None):
    """The fused pooled layer's weight image (dir_cin_pooled_pack_f32), once per version of W (while a stream is capturing: packed inside the
    graph, nothing remembered -- the rule of every per-version cache here).  owner: the LONG-LIVED tensor whose identity and version stand for
    W's (a module passes its nn.Parameter and `.data` as W: a `.data` view is a new object with a fresh version counter on every call)."""
    import weakref
    lib = _lib.load()
    H = W.shape[0]
    own = W if owner is None else owner
    capturing = torch.cuda.is_current_stream_capturing()
    key = (W.data_ptr(), tuple(W.shape), tuple(W.stride()), m, Hp)
    sig = (own._version if not own.is_inference() else -1, _CACHE_GEN[0])
    hit = _CIN_POOLED_IMAGES.get(key)
    if hit is not None and not capturing and hit[0]() is own and hit[1] == sig:
        return hit[2]
    Wc = W if W.is_contiguous() else W.contiguous()
    nbytes = int(lib.dir_cin_pooled_image_bytes(m, Hp, H, D))
    img = torch.empty(nbytes, dtype=torch.uint8, device=W.device)
    _lib.check(lib.dir_cin_pooled_pack_f32(_ptr(Wc), m, Hp, H, D, _ptr(img), nbytes, _stream()))
    if not capturing and not own.is_inference():
        if len(_CIN_POOLED_IMAGES) > 64:
            _CIN_POOLED_IMAGES.clear()
        _CIN_POOLED_IMAGES[key] = (weakref.ref(own), sig, img)
    return img


def cin_pooled_covers(m, D, Hp):
    """Shapes of the pooled-only form of a layer (csrc/cin_pool.hip + the dense kernels): m <= 64, D in {4, 8, 16, 32}, Hp * m a multiple of 4."""
    return m <= 64 and D in (4, 8, 16, 32) and (Hp * m) % 4 == 0


def _row_bits_hint(t, n):
    """The row maxima the kernel that produced `t` left on it (t._dir_row_bits = (bits [n] int32, t._version at that time)), or None: the
    tensor was modified since (an in-place op bumps its version), is another tensor, or never had them."""
    hint = getattr(t, "_dir_row_bits", None)
    if hint is not None and hint[1] == t._version and hint[0].numel() == n and hint[0].device == t.device:
        return hint[0]
    return None


def cin_pool_z(x0, xk, want_bits=False):
    """Z [B, Hp*m], Z[b, i*m + j] = sum_d xk[b,i,d] x0[b,j,d] (include/dir_hip.h: dir_cin_pool_z_f32): what a layer whose map only feeds its
    pooled sums needs of its inputs -- pooled = Z @ W.T.  want_bits: -> (Z, row_bits [B] int32: the bit pattern of max |Z[b, :]|, the row
    scales of the dense product that follows -- dir_cin_pool_z_bits_f32)."""
    _dev(x0, torch.float32, "x0")
    _dev(xk, torch.float32, "xk")
    B, m, D = x0.shape
    Hp = xk.shape[1]
    if not (x0.is_contiguous() and xk.is_contiguous()) or xk.shape[0] != B or xk.shape[2] != D:
        raise ValueError("cin_pool_z: contiguous x0 [B,m,D], xk [B,Hp,D]")
    Z = torch.empty((B, Hp * m), dtype=torch.float32, device=x0.device)
    if want_bits:
        zb = torch.empty(max(B, 1), dtype=torch.int32, device=x0.device)[:B]
        _lib.check(_lib.load().dir_cin_pool_z_bits_f32(_ptr(x0), _ptr(xk), m, Hp, D, B, _ptr(Z), _ptr(zb), _stream()))
        return Z, zb
    _lib.check(_lib.load().dir_cin_pool_z_f32(_ptr(x0), _ptr(xk), m, Hp, D, B, _ptr(Z), _stream()))
    return Z


def cin_pool_dx(x0, xk, dZ, add_pooled=None, dx0=None):
    """The data gradients of a pooled-only layer from dZ [B, Hp*m] = g_pooled @ W (include/dir_hip.h: dir_cin_pool_dx_f32):
    dxk[b,i,d] = sum_j dZ[b,i,j] x0[b,j,d] (+ add_pooled[b,i]: the pooled gradient of the layer below), dx0[b,j,d] = sum_i dZ[b,i,j] xk[b,i,d]
    (accumulated into `dx0` when given).  -> (dxk [B,Hp,D], dx0 [B,m,D])."""
    _dev(x0, torch.float32, "x0")
    _dev(xk, torch.float32, "xk")
    _dev(dZ, torch.float32, "dZ")
    B, m, D = x0.shape
    Hp = xk.shape[1]
    if not (x0.is_contiguous() and xk.is_contiguous() and dZ.is_contiguous()) or tuple(dZ.shape) != (B, Hp * m):
        raise ValueError("cin_pool_dx: contiguous x0 [B,m,D], xk [B,Hp,D], dZ [B,Hp*m]")
    if add_pooled is not None:
        _dev(add_pooled, torch.float32, "add_pooled")
        if tuple(add_pooled.shape) != (B, Hp) or (B > 0 and add_pooled.stride(1) != 1):
            raise ValueError("cin_pool_dx: add_pooled must be [B, Hp] with unit column stride")
    acc = dx0 is not None
    if acc and (tuple(dx0.shape) != (B, m, D) or not dx0.is_contiguous() or dx0.dtype != torch.float32):
        raise ValueError("cin_pool_dx: dx0 must be a contiguous float32 [B, m, D] tensor")
    dxk = torch.empty((B, Hp, D), dtype=torch.float32, device=x0.device)
    out = dx0 if acc else torch.empty((B, m, D), dtype=torch.float32, device=x0.device)
    _lib.check(_lib.load().dir_cin_pool_dx_f32(_ptr(x0), _ptr(xk), _ptr(dZ), m, Hp, D, B, _ptr(add_pooled),
                                               add_pooled.stride(0) if add_pooled is not None and B > 0 else Hp, _ptr(dxk), _ptr(out),
                                               1 if acc else 0, _stream()))
    return dxk, out


def cin_layer(x0, xk, W, pooled=None, want_xout=True, arith=None, z_out=None, grad_operand=False, g_bits_out=None, w_owner=None):
    """One CIN layer (include/dir_hip.h A14): x0 [B,m,D], xk [B,Hp,D], W [H, Hp*m] ->
    (xout [B,H,D], pooled [B,H]); `pooled` may be a [B,H] view into a wider buffer (row stride kept).
    want_xout=False skips the [B,H,D] write (the last layer of a stack only feeds its pooled sums): xout is None.
    arith: "f32" = dir_cin_layer_f32 (fp32 MFMA, an exact fma chain); "bf16x3" = dir_cin_layer_bf16x3_f32 (three-way bf16 split of
    both operands, six products on the bf16 pipe, fp32 accumulate: fp32-equivalent, not bitwise the same; raises on a shape that
    kernel does not cover); "auto" = cin_auto_arith(m, D, Hp, H); None = CIN_ARITH (env DIR_CIN_ARITH, default "auto").
    When xk IS x0 (same storage: the first layer of a stack, 8 <= m <= 40) the bf16x3 arithmetic runs over the unordered field pairs
    (dir_cin_layer1_bf16x3_f32): the same sums in another order.
    want_xout=False with arith "auto" / "bf16x3": the pooled sums alone are sum_{i,j} W[h,i,j] Z[b,i,j], Z = sum_d xk x0 (cin_pool_z) --
    the sum over d first, then ONE dense product, 1/D of the layer's matrix work; z_out (a list): Z is appended to it when that form ran
    (the backward of a stack reuses it).
    arith "f16x2" = dir_cin_layer_f16x2_f32 / dir_cin_layer1_f16x2_f32: two fp16 pieces per operand, three products (|operands| < 65 504;
    scaled error 3-6e-7 on embedding-scale data); "auto" picks it for forward layers (CIN_FWD_SPLIT) unless grad_operand says that xk is
    a gradient (the backward's forward-form contractions): small magnitudes belong on bf16x3."""
    arith = arith or CIN_ARITH
    if arith not in ("auto", "f32", "bf16x3", "f16x2", "f16x2_grad"):
        raise ValueError("cin_layer: arith must be 'auto', 'f32', 'bf16x3', 'f16x2' or 'f16x2_grad'")
    auto = arith == "auto"
    if (not want_xout and arith != "f32" and CIN_POOLED_LAST and x0.dim() == 3 and xk.dim() == 3 and x0.shape[0] > 0
            and cin_pooled_covers(x0.shape[1], x0.shape[2], xk.shape[1]) and x0.is_cuda and x0.is_contiguous() and xk.is_contiguous()
            and W.dim() == 2 and W.shape[1] == xk.shape[1] * x0.shape[1]):
        B, H = x0.shape[0], W.shape[0]
        if pooled is None:
            pooled = torch.empty((B, H), dtype=torch.float32, device=x0.device)
        if (z_out is None and CIN_POOLED_FUSED and arith in ("auto", "bf16x3") and cin_pooled_fused_covers(x0.shape[1], xk.shape[1], H, x0.shape[2])
                and pooled.stride(1) == 1 and pooled.stride(0) % 4 == 0 and pooled.data_ptr() % 16 == 0):
            # round 6 (inference: nobody asks for Z): the two passes fused -- Z is formed in registers and fed to the matrix pipe, its 872 MB at
            # the BASELINE shape neither written nor read (include/dir_hip.h: dir_cin_pooled_last_bf16x3_f32)
            m, D, Hp = x0.shape[1], x0.shape[2], xk.shape[1]
            _lib.check(_lib.load().dir_cin_pooled_last_bf16x3_f32(_ptr(x0), _ptr(xk), _ptr(cin_pooled_image(W, m, Hp, D, owner=w_owner)), m, Hp, H, D, B, _ptr(pooled),
                                                                  pooled.stride(0), _stream()))
            return None, pooled
        # Z (sums over d of products) is a general input: dense "auto" runs its row-scaled fp16 x 2 kernel on it, the rows' maxima left by
        # the kernel that forms Z (no max pass over the [B, Hp*m] matrix)
        Z, zb = cin_pool_z(x0, xk, want_bits=True)
        if pooled.stride(1) == 1 and pooled.stride(0) % 4 == 0 and pooled.data_ptr() % 16 == 0:
            dense(Z, W, out=pooled, row_bits=zb)
        else:
            pooled.copy_(dense(Z, W, row_bits=zb))
        if z_out is not None:
            z_out.append(Z)
        return None, pooled
    if arith == "auto":
        arith = cin_auto_arith(x0.shape[1], x0.shape[2], xk.shape[1], W.shape[0])
        pairs = CIN_L1_PAIRS and xk.data_ptr() == x0.data_ptr() and xk.shape[1] == x0.shape[1] and 8 <= x0.shape[1] <= 40 and x0.shape[0] > 0
        if arith == "bf16x3" and CIN_FWD_SPLIT == "f16x2_unscaled" and not grad_operand:
            arith = "f16x2"                                      # A/B switch: round 4's routing (the general layers on the unscaled kernel)
        elif arith == "bf16x3" and CIN_FWD_SPLIT == "f16x2" and not grad_operand:
            # forward layers on SCALED fp16 x 2 (round 5): the first layer's pair form scales the rows of its x0 slice and the pair weights
            # inside the kernel ("f16x2" below), every other layer runs the row-scaled kernel the gradients use ("f16x2_grad": rows of xk
            # and the tensor W scaled by exact powers of two) -- no operand's magnitude is assumed any more
            arith = "f16x2" if pairs else "f16x2_grad"
        elif arith == "bf16x3" and CIN_BWD_SPLIT == "f16x2" and grad_operand and xk.data_ptr() != x0.data_ptr():
            arith = "f16x2_grad"                                 # xk is a gradient: fp16 x 2 with its rows scaled inside the kernel
    _dev(x0, torch.float32, "x0")
    _dev(xk, torch.float32, "xk")
    _dev(W, torch.float32, "W")
    if not (x0.is_contiguous() and xk.is_contiguous() and W.is_contiguous()):
        raise ValueError("cin_layer operands must be contiguous")
    B, m, D = x0.shape
    Hp = xk.shape[1]
    H = W.shape[0]
    if W.shape[1] != Hp * m or xk.shape[0] != B or xk.shape[2] != D:
        raise ValueError("cin_layer: W must be [H, Hp*m], xk [B,Hp,D]")
    xout = torch.empty((B, H, D), dtype=torch.float32, device=x0.device) if want_xout else None
    if pooled is None:
        pooled = torch.empty((B, H), dtype=torch.float32, device=x0.device)
    if arith in ("bf16x3", "f16x2", "f16x2_grad"):
        if not cin_bf16x3_covers(m, D):
            raise ValueError("cin_layer: arith=%r covers m <= 40 and D in {4,8,16,32} (got m=%d, D=%d)" % (arith, m, D))
        lib = _lib.load()
        f_l1 = lib.dir_cin_layer1_f16x2_f32 if arith == "f16x2" else lib.dir_cin_layer1_bf16x3_f32
        f_ly = {"f16x2": lib.dir_cin_layer_f16x2_f32, "f16x2_grad": lib.dir_cin_layer_grad_f16x2_f32}.get(arith, lib.dir_cin_layer_bf16x3_f32)
        # the rows' maxima of xout ride on the tensor for the next layer's row scales (one column block: H <= 128)
        ob = torch.empty(B * D, dtype=torch.int32, device=x0.device) if (CIN_ROW_BITS_CARRY and want_xout and H <= 128 and B > 0
                                                                         and arith in ("f16x2", "f16x2_grad")) else None
        if arith != "f16x2_grad" and CIN_L1_PAIRS and xk.data_ptr() == x0.data_ptr() and Hp == m and 8 <= m <= 40 and B > 0:
            # the first layer of a stack (xk IS x0): a quadratic form in x0 -- the kernel multiplies the m (m + 1) / 2 unordered pairs only
            # (dir_cin_layer1_bf16x3_f32)
            nbytes = int(lib.dir_cin_layer1_bf16x3_workspace_bytes(m, H))
            ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=x0.device)
            wp = ctypes.c_void_p(ws.data_ptr() + (-ws.data_ptr()) % 256)
            if arith == "f16x2" and ob is not None:
                _lib.check(lib.dir_cin_layer1_bits_f16x2_f32(_ptr(x0), _ptr(W), m, H, D, B, _ptr(xout), _ptr(pooled), pooled.stride(0), wp, nbytes,
                                                             _ptr(ob), _stream()))
                _leave_hint(xout, '_dir_row_bits', ob)
                return xout, pooled
            _lib.check(f_l1(_ptr(x0), _ptr(W), m, H, D, B, _ptr(xout) if want_xout else None, _ptr(pooled), pooled.stride(0), wp, nbytes, _stream()))
            return xout, pooled
        nbytes = int(lib.dir_cin_bf16x3_workspace_bytes(m, Hp, H))
        ws = torch.empty(max(16, nbytes), dtype=torch.uint8, device=x0.device)
        if arith == "f16x2_grad" and not grad_operand and g_bits_out is None:
            # a FORWARD layer on the row-scaled kernel: the producer's row maxima in, this layer's out
            ib = _row_bits_hint(xk, B * D) if CIN_ROW_BITS_CARRY else None
            if ib is not None and B > 0:        # the producer's row maxima: the plain / row-scaled verdict is taken on the device
                _lib.check(lib.dir_cin_layer_auto_f16x2_f32(_ptr(x0), _ptr(xk), _ptr(W), m, Hp, H, D, B, _ptr(xout) if want_xout else None, _ptr(pooled),
                                                            pooled.stride(0), _ptr(ws), nbytes, _ptr(ib), _ptr(ob), _stream()))
            else:
                _lib.check(lib.dir_cin_layer_rows_f16x2_f32(_ptr(x0), _ptr(xk), _ptr(W), m, Hp, H, D, B, _ptr(xout) if want_xout else None, _ptr(pooled),
                                                            pooled.stride(0), _ptr(ws), nbytes, None, _ptr(ob), _stream()))
            if ob is not None:
                _leave_hint(xout, '_dir_row_bits', ob)
            return xout, pooled
        if arith == "f16x2_grad":          # (g_bits_out, a list: the bit pattern of max |xk| -- a by-product of the row maxima -- is appended)
            gbits = torch.empty(1, dtype=torch.int32, device=x0.device) if (g_bits_out is not None and B > 0) else None
            _lib.check(f_ly(_ptr(x0), _ptr(xk), _ptr(W), m, Hp, H, D, B, _ptr(xout) if want_xout else None, _ptr(pooled), pooled.stride(0), _ptr(ws),
                            nbytes, _ptr(gbits), _stream()))
            if gbits is not None:
                g_bits_out.append(gbits)
            return xout, pooled
        _lib.check(f_ly(_ptr(x0), _ptr(xk), _ptr(W), m, Hp, H, D, B, _ptr(xout) if want_xout else None, _ptr(pooled), pooled.stride(0), _ptr(ws),
                        nbytes, _stream()))
        return xout, pooled
    mt = next((t for t in CIN_FIELD_TILES if t >= m), m)
    if mt != m and B > 0:
        # the kernel's fast (interleaved-staging) path exists for the instantiated field counts only; a zero field and zero
        # weight columns change nothing in the sum and cost one small copy (m = 39 -> 40: 85 -> ~135 TFLOP/s)
        x0 = torch.nn.functional.pad(x0, (0, 0, 0, mt - m))
        W = torch.nn.functional.pad(W.view(H, Hp, m), (0, mt - m)).reshape(H, Hp * mt)
        m = mt
    _lib.check(_lib.load().dir_cin_layer_f32(_ptr(x0), _ptr(xk), _ptr(W), m, Hp, H, D, B,
                                             _ptr(xout) if want_xout else None, _ptr(pooled),
                                             pooled.stride(0), _stream()))
    return xout, pooled


def esmm_head(ctr_logits, cvr_logits, eps=1e-7):
    """ESMM's ctcvr logit from the two towers' logits (include/dir_hip.h: dir_esmm_head_f32; ESMM.py:67-77): log(p / (1 - p)) with
    p = clip(sigmoid(ctr) * sigmoid(cvr), eps, 1 - eps).  Inference only (no autograd)."""
    _dev(ctr_logits, torch.float32, "ctr_logits")
    _dev(cvr_logits, torch.float32, "cvr_logits")
    if ctr_logits.shape != cvr_logits.shape or not ctr_logits.is_contiguous() or not cvr_logits.is_contiguous():
        raise ValueError("esmm_head: two contiguous tensors of one shape")
    out = torch.empty_like(ctr_logits)
    _lib.check(_lib.load().dir_esmm_head_f32(_ptr(ctr_logits), _ptr(cvr_logits), ctr_logits.numel(), float(eps), _ptr(out), _stream()))
    return out


def cin_gather_covers(m, D, Hs):
    """Shapes and routing switches under which cin_stack_gather runs EXACTLY the kernels cin_layer(arith=None) would run on the materialised
    x0 (the default forward routing: the first layer over field pairs on scaled fp16 x 2, later layers behind the device-side verdict, the
    last layer in its fused pooled form)."""
    Hs = list(Hs)
    if not Hs or not (CIN_ARITH == "auto" and CIN_FWD_SPLIT == "f16x2" and CIN_L1_PAIRS and CIN_ROW_BITS_CARRY and CIN_POOLED_LAST and CIN_POOLED_FUSED):
        return False
    if not (cin_bf16x3_covers(m, D) and 8 <= m <= 40) or any(h > 128 for h in Hs[:-1]):
        return False
    hp = m
    for h in Hs[:-1]:
        if cin_auto_arith(m, D, hp, h) != "bf16x3":
            return False
        hp = h
    return len(Hs) >= 2 and cin_pooled_covers(m, D, hp) and cin_pooled_fused_covers(m, hp, Hs[-1], D)


def cin_stack_gather(rows, inv, Ws, pooled, w_owners=None):
    """The CIN stack of an inference forward with x0 read THROUGH INVERSE POSITIONS (round 6: the row-sharded lookup without its finish pass,
    shard.ShardedTables.lookup_rows): rows [n, D] fp32 as the exchange left them, inv [B, m] int64 = the position of (sample, field)'s row
    in `rows` (< 0: a zero row).  Ws: the layers' weights [H_k, H_{k-1} * m]; pooled [B, sum H_k] (a view with a row stride is fine) receives
    every layer's pooled sums.  The [B, m * D] concatenation is never written or read; bit for bit what cin_layer gives on it
    (dir_cin_layer1_f16x2_gather_f32, dir_cin_layer_f16x2_gather_f32, dir_cin_pooled_last_bf16x3_gather_f32).  Raises where
    cin_gather_covers says no: the caller then materialises x0 (ops.gather_fm over shard.rows_as_tables) and runs cin_layer."""
    _dev(rows, torch.float32, "rows")
    _dev(inv, torch.int64, "inv")
    if rows.dim() != 2 or inv.dim() != 2 or not rows.is_contiguous() or not inv.is_contiguous():
        raise ValueError("cin_stack_gather: rows must be a contiguous [n, D], inv a contiguous [B, m]")
    B, m = inv.shape
    D = rows.shape[1]
    Hs = [int(W.shape[0]) for W in Ws]
    if not cin_gather_covers(m, D, Hs):
        raise ValueError("cin_stack_gather: shape / routing not covered (cin_gather_covers): m=%d D=%d layers=%r" % (m, D, Hs))
    if pooled.shape != (B, sum(Hs)) or pooled.stride(1) != 1:
        raise ValueError("cin_stack_gather: pooled must be [B, sum(H_k)] with unit column stride")
    if B == 0:
        return pooled
    lib = _lib.load()
    dev = rows.device
    xk, ib, hp, off = None, None, m, 0
    for k, (W, h) in enumerate(zip(Ws, Hs)):
        _dev(W, torch.float32, "W")
        if W.shape[1] != hp * m or not W.is_contiguous():
            raise ValueError("cin_stack_gather: layer %d's W must be a contiguous [H, Hp*m]" % k)
        pk = pooled[:, off:off + h]
        if k == len(Hs) - 1:
            if not (pk.stride(0) % 4 == 0 and pk.data_ptr() % 16 == 0):
                raise ValueError("cin_stack_gather: the last layer's pooled view must be 16-byte aligned with a row stride that is a multiple of 4")
            img = cin_pooled_image(W, m, hp, D, owner=None if w_owners is None else w_owners[k])
            _lib.check(lib.dir_cin_pooled_last_bf16x3_gather_f32(_ptr(rows), _ptr(inv), _ptr(xk), _ptr(img), m, hp, h, D, B, _ptr(pk), pk.stride(0), _stream()))
            break
        xout = torch.empty((B, h, D), dtype=torch.float32, device=dev)
        ob = torch.empty(B * D, dtype=torch.int32, device=dev)
        if k == 0:
            nbytes = int(lib.dir_cin_layer1_bf16x3_workspace_bytes(m, h))
            ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
            wp = ctypes.c_void_p(ws.data_ptr() + (-ws.data_ptr()) % 256)
            _lib.check(lib.dir_cin_layer1_f16x2_gather_f32(_ptr(rows), _ptr(inv), _ptr(W), m, h, D, B, _ptr(xout), _ptr(pk), pk.stride(0), wp, nbytes, _ptr(ob),
                                                           _stream()))
        else:
            nbytes = int(lib.dir_cin_bf16x3_workspace_bytes(m, hp, h))
            ws = torch.empty(max(16, nbytes), dtype=torch.uint8, device=dev)
            _lib.check(lib.dir_cin_layer_f16x2_gather_f32(_ptr(rows), _ptr(inv), _ptr(xk), _ptr(W), m, hp, h, D, B, _ptr(xout), _ptr(pk), pk.stride(0), _ptr(ws),
                                                          nbytes, _ptr(ib), _ptr(ob), _stream()))
        xk, ib, hp = xout, ob, h
        off += h
    return pooled


# ---- id paths (A3) ---------------------------------------------------------------------------------
def fingerprint64(s):
    if isinstance(s, str):
        s = s.encode("utf-8")
    return int(_lib.load().dir_fingerprint64(s, len(s)))


def hash_bucket_strings(strings, num_buckets):
    """categorical_column_with_hash_bucket on string keys (DeepCrossNetwork/train.py:85-86): host."""
    bs = [s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in strings]
    n = len(bs)
    arr = (ctypes.c_char_p * n)(*bs)
    lens = (ctypes.c_int64 * n)(*[len(b) for b in bs])
    out = (ctypes.c_int64 * n)()
    _lib.check(_lib.load().dir_hash_bucket_fast(arr, lens, n, num_buckets, out))
    return torch.tensor(list(out), dtype=torch.int64)


def hash_bucket_ints(keys, num_buckets):
    """Integer keys hashed on the device as their decimal text ([TF-upstream] as_string -> hash)."""
    _dev(keys, torch.int64, "keys")
    keys = keys.contiguous()
    out = torch.empty_like(keys)
    _lib.check(_lib.load().dir_hash_bucket_i64_device(_ptr(keys), keys.numel(), num_buckets, _ptr(out), _stream()))
    return out


def hash_bucket_ints_fields(keys, buckets_dev):
    """keys [B, F] (contiguous) with one bucket count per field (device int64 [F]) -> ids [B, F]."""
    _dev(keys, torch.int64, "keys")
    _dev(buckets_dev, torch.int64, "buckets_dev")
    keys = keys.contiguous()
    out = torch.empty_like(keys)
    _lib.check(_lib.load().dir_hash_bucket_i64_fields_device(_ptr(keys), keys.numel(), _ptr(buckets_dev), buckets_dev.numel(),
                                                             _ptr(out), _stream()))
    return out


def hash_bucket_bytes(bytes_dev, offsets, num_buckets):
    """Byte strings already on the device (uint8 buffer + int64 offsets [n+1]) -> ids [n]."""
    if bytes_dev.dtype != torch.uint8 or not bytes_dev.is_cuda:
        raise TypeError("bytes_dev must be a CUDA uint8 tensor")
    _dev(offsets, torch.int64, "offsets")
    n = offsets.numel() - 1
    out = torch.empty(n, dtype=torch.int64, device=offsets.device)
    _lib.check(_lib.load().dir_hash_bucket_bytes_device(_ptr(bytes_dev.contiguous()), _ptr(offsets.contiguous()), n,
                                                        num_buckets, _ptr(out), _stream()))
    return out


def bucketize(x, boundaries):
    """bucketized_column: number of boundaries <= x -> int64 ids."""
    _dev(x, torch.float32, "x")
    _dev(boundaries, torch.float32, "boundaries")
    x = x.contiguous()
    out = torch.empty(x.shape, dtype=torch.int64, device=x.device)
    _lib.check(_lib.load().dir_bucketize_f32(_ptr(x), x.numel(), _ptr(boundaries.contiguous()), boundaries.numel(),
                                             _ptr(out), _stream()))
    return out


def shard_route(ids, vocab_dev, P):
    """'div' owner / local row of every id of a flattened [.., F] id array; vocab_dev: device int64 [F].
    -> (owner int32, local int64), same shape as ids."""
    _dev(ids, torch.int64, "ids")
    _dev(vocab_dev, torch.int64, "vocab_dev")
    ids = ids.contiguous()
    owner = torch.empty(ids.shape, dtype=torch.int32, device=ids.device)
    local = torch.empty_like(ids)
    _lib.check(_lib.load().dir_shard_route(_ptr(ids), ids.numel(), _ptr(vocab_dev), vocab_dev.numel(), P, _ptr(owner),
                                           _ptr(local), _stream()))
    return owner, local


def gather_rows(tables, slot, row):
    """Owner-side flat lookup of the sharded path: out[i] = tables[slot[i]][row[i]] (row < 0 -> zeros)."""
    ts = _as_tableset(tables)
    _dev(row, torch.int64, "row")
    if slot is not None:
        _dev(slot, torch.int32, "slot")
    n = row.numel()
    out = torch.empty((n, ts.K), dtype=torch.float32, device=ts.device)
    _lib.check(_lib.load().dir_gather_rows_f32(_ptr(ts.ptrs), ts.K, _ptr(slot), _ptr(row.contiguous()), n, _ptr(out),
                                               _stream()))
    return out


def shard_bucket(ids, vocab_dev, P, payload=None, inv=None, parts=None, first=None):
    """Route + counting-sort by owner in one call (requester side of the sharded lookup, variable-size form).
    -> (payload [n] int64 grouped by owner, inv [n] int64, counts [P] int64, starts [P] int64), all on device.
    parts / first: device int32 [F] (slices per table and the rank of slice 0) or None = every table cut P ways."""
    _dev(ids, torch.int64, "ids")
    ids = ids.contiguous()
    n = ids.numel()
    lib = _lib.load()
    F = vocab_dev.numel()
    if payload is None:
        payload = torch.empty(n, dtype=torch.int64, device=ids.device)
    if inv is None:
        inv = torch.empty(n, dtype=torch.int64, device=ids.device)
    counts = torch.empty(P, dtype=torch.int64, device=ids.device)
    starts = torch.empty(P, dtype=torch.int64, device=ids.device)
    ws = torch.empty(max(1, int(lib.dir_shard_bucket_workspace_bytes(n, P))), dtype=torch.uint8, device=ids.device)
    _lib.check(lib.dir_shard_bucket(_ptr(ids), n, _ptr(vocab_dev), _ptr(parts), _ptr(first), F, P, _ptr(payload), _ptr(inv), _ptr(counts),
                                    _ptr(starts), _ptr(ws), _stream()))
    return payload, inv, counts, starts


def shard_bucket_cap(ids, vocab_dev, P, cap, payload, inv, counts, overflow, workspace, parts=None, first=None, stat=None):
    """Fixed-capacity requester side (include/dir_hip.h: dir_shard_bucket_cap) into caller-owned buffers: payload
    [P*(cap+1)] int64 slabs, inv [n] int64, counts [P] int64, overflow [1] int32, workspace (zeroed once) -- no host read."""
    _dev(ids, torch.int64, "ids")
    if not ids.is_contiguous():
        raise ValueError("shard_bucket_cap: ids must be contiguous")
    _lib.check(_lib.load().dir_shard_bucket_cap(_ptr(ids), ids.numel(), _ptr(vocab_dev), _ptr(parts), _ptr(first), vocab_dev.numel(), P,
                                                cap, _ptr(payload), _ptr(inv), _ptr(counts), _ptr(overflow), _ptr(stat), _ptr(workspace), _stream()))


def shard_bucket_cap_dedup(ids2d, vocab_dev, P, cap, payload, inv, counts, overflow, workspace, parts=None, first=None, stat=None):
    """The fixed-capacity requester side with duplicates removed inside 2048 / 4096-sample tiles of one slot (include/dir_hip.h:
    dir_shard_bucket_cap_dedup).  ids2d [B, F] int64 with any strides; inv [F*B] int64 comes back FIELD-MAJOR (inv[f*B + b]): hand
    inv.view(F, B).t() to the finish gather."""
    _dev(ids2d, torch.int64, "ids")
    B, F = ids2d.shape
    if F != vocab_dev.numel() or inv.numel() != B * F:
        raise ValueError("shard_bucket_cap_dedup: ids [B, F], inv [F*B]")
    _lib.check(_lib.load().dir_shard_bucket_cap_dedup(_ptr(ids2d), ids2d.stride(0), ids2d.stride(1), B, _ptr(vocab_dev), _ptr(parts), _ptr(first),
                                                      F, P, cap, _ptr(payload), _ptr(inv), _ptr(counts), _ptr(overflow), _ptr(stat), _ptr(workspace),
                                                      _stream()))


def shard_slab_stat(recv, n_slabs, cap, stat):
    """Headers of n_slabs received slabs (cap + 1 words apart, contiguous) -> stat int64 [2] = {some sender's demand > cap, the largest
    demand} (include/dir_hip.h: dir_shard_slab_stat): the overflow verdict every rank agrees on, read off the id exchange."""
    _dev(recv, torch.int64, "recv")
    if recv.numel() < n_slabs * (cap + 1) or not recv.is_contiguous():
        raise ValueError("shard_slab_stat: recv must hold n_slabs contiguous slabs of cap + 1 words")
    _lib.check(_lib.load().dir_shard_slab_stat(_ptr(recv), n_slabs, cap, _ptr(stat), _stream()))


SLAB_SANITIZE = 4


def gather_slabs(tables, recv, P, cap, out, sanitize=False):
    """Owner side of the fixed-capacity exchange: recv [P*(cap+1)] int64 slabs -> out [P*cap, K] (rows behind a slab's header
    are left untouched)."""
    ts = _as_tableset(tables)
    _dev(recv, torch.int64, "recv")
    _dev(out, torch.float32, "out")
    _lib.check(_lib.load().dir_gather_slabs_f32(_ptr(ts.ptrs), ts.F, ts.K, _ptr(recv), P, cap,
                                                ts.gather_flags() | (SLAB_SANITIZE if sanitize else 0), _ptr(out), _stream()))
    return out


def gather_packed(tables, payload, out=None):
    """Owner side: payload p = local_row*F + slot (p < 0 -> zeros) -> rows [n, K]."""
    ts = _as_tableset(tables)
    _dev(payload, torch.int64, "payload")
    n = payload.numel()
    if out is None:
        out = torch.empty((n, ts.K), dtype=torch.float32, device=ts.device)
    _lib.check(_lib.load().dir_gather_packed_f32(_ptr(ts.ptrs), ts.F, ts.K, _ptr(payload.contiguous()), n,
                                                 ts.gather_flags(), _ptr(out), _stream()))
    return out


# ---- backward of the interaction ops --------------------------------------------------
def fm_logit_backward(emb, g, F, K, add_in=None, out=None):
    """d fm_logit / d emb: demb[b,f,:] = g[b] * (sum_f' e[b,f',:] - e[b,f,:]) (+ add_in).  g: [B] or [B,1]."""
    _dev(emb, torch.float32, "emb")
    g = _dev(g, torch.float32, "g").reshape(-1).contiguous()
    B = emb.shape[0]
    if emb.shape[1] != F * K or emb.stride(1) != 1 or g.numel() != B:
        raise ValueError("fm_logit_backward: emb [B, F*K], g [B]")
    if add_in is not None:
        _dev(add_in, torch.float32, "add_in")
        if add_in.shape != emb.shape or add_in.stride(1) != 1:
            raise ValueError("add_in must match emb")
    if out is None:
        out = torch.empty((B, F * K), dtype=torch.float32, device=emb.device)
    _lib.check(_lib.load().dir_fm_second_order_backward_f32(_ptr(emb), emb.stride(0), _ptr(g), _ptr(add_in),
                                                            add_in.stride(0) if add_in is not None else 0, B, F, K,
                                                            _ptr(out), out.stride(0), _stream()))
    return out


def cross_network_backward(x0, w, b, gout):
    """Backward of cross_network: -> (gx0 [B,d], gw [L,d], gb [L,d])."""
    for t, n in ((x0, "x0"), (w, "w"), (b, "b"), (gout, "gout")):
        _dev(t, torch.float32, n)
    B, d = x0.shape
    L = w.shape[0]
    w, b = w.contiguous(), b.contiguous()
    if gout.shape != x0.shape or gout.stride(1) != 1 or x0.stride(1) != 1:
        raise ValueError("gout must be [B, d] like x0")
    lib = _lib.load()
    gx0 = torch.empty((B, d), dtype=torch.float32, device=x0.device)
    gw = torch.empty((L, d), dtype=torch.float32, device=x0.device)
    gb = torch.empty((L, d), dtype=torch.float32, device=x0.device)
    ws = torch.empty(max(1, int(lib.dir_dcn_cross_backward_workspace_bytes(L, d))), dtype=torch.uint8, device=x0.device)
    _lib.check(lib.dir_dcn_cross_backward_f32(_ptr(x0), x0.stride(0), _ptr(w), _ptr(b), L, _ptr(gout), gout.stride(0), B, d,
                                              _ptr(gx0), gx0.stride(0), _ptr(gw), _ptr(gb), _ptr(ws), _stream()))
    return gx0, gw, gb


CIN_DW_SYM = os.environ.get("DIR_CIN_DW_SYM", "1") != "0"      # development switch: 0 keeps the general kernels on the first layer


def cin_dw_auto_arith(m, D, Hp, H):
    """What cin_dw(arith="auto") runs: the bf16x3 kernel (csrc/cin_dw_bf3.hip) for D >= 8 and layers at least 96 channels wide on the
    xk side (a wave's 8 column tiles are i tiles: narrower layers leave them idle) with at most a third of the 128 x 128 output
    blocks padding; the fp32-MFMA kernel otherwise."""
    if D not in (8, 16, 32) or Hp < 96:
        return "f32"
    pad = (-(-H // 128) * 128) * (-(-Hp // 128) * 128) * (-(-m // 4) * 4)
    return "bf16x3" if pad <= 1.34 * H * Hp * m else "f32"


def cin_dw(x0, xk, G, dW=None, accumulate=False, arith=None, g_absmax_bits=None):
    """Weight gradient of one CIN layer (include/dir_hip.h, dir_cin_dw_f32): x0 [B,m,D], xk [B,Hp,D], G = dL/dxout
    [B,H,D] -> dW [H, Hp*m] (added into `dW` when accumulate).  arith: "f32" | "bf16x3" | "auto" (cin_dw_auto_arith) | None = CIN_ARITH;
    when xk IS x0 (the first layer of a stack: same storage) "auto" and "bf16x3" run the symmetric kernel dir_cin_dw_sym_bf16x3_f32
    ("bf16x3_sym" asks for it by name).  "f16x2" (what "auto" resolves to under CIN_BWD_SPLIT where it used to pick bf16x3):
    dir_cin_dw_f16x2_f32, G scaled by one power of two; g_absmax_bits: an int32 / uint32 device tensor holding the bit pattern of an upper
    bound of max |G| (None: the entry runs its own max pass)."""
    for t, n in ((x0, "x0"), (xk, "xk"), (G, "G")):
        _dev(t, torch.float32, n)
        if not t.is_contiguous():
            raise ValueError("cin_dw operands must be contiguous")
    B, m, D = x0.shape
    Hp, H = xk.shape[1], G.shape[1]
    if xk.shape[0] != B or xk.shape[2] != D or G.shape[0] != B or G.shape[2] != D:
        raise ValueError("cin_dw: xk must be [B,Hp,D] and G [B,H,D]")
    if dW is None:
        if accumulate:
            raise ValueError("cin_dw: accumulate needs dW")
        dW = torch.empty((H, Hp * m), dtype=torch.float32, device=x0.device)
    elif dW.shape != (H, Hp * m) or not dW.is_contiguous():
        raise ValueError("cin_dw: dW must be a contiguous [H, Hp*m] tensor")
    lib = _lib.load()
    arith = arith or CIN_ARITH
    first_layer = xk.data_ptr() == x0.data_ptr() and xk.shape == x0.shape and D in (8, 16, 32) and m <= 64
    if arith == "bf16x3_sym" and not first_layer:
        raise ValueError("cin_dw: arith='bf16x3_sym' is the first layer's kernel: xk must BE x0 (same storage), D in {8, 16, 32}, m <= 64")
    if arith == "bf16x3_sym" or (first_layer and arith in ("auto", "bf16x3") and CIN_DW_SYM):
        # the first layer of a stack (xk is x0): dW is symmetric in (i, j) and the unordered pairs are the GEMM's columns (dir_cin_dw_sym_bf16x3_f32)
        nbytes = int(lib.dir_cin_dw_sym_bf16x3_workspace_bytes(m, H, D, B))
        ws = torch.empty(max(16, nbytes), dtype=torch.uint8, device=x0.device)
        if arith == "auto" and CIN_BWD_SPLIT == "f16x2" and g_absmax_bits is not None and B > 0:
            # fp16 x 2 with G scaled by one power of two (dir_cin_dw_sym_f16x2_f32): the caller brings max |G| (no pass over G here)
            _lib.check(lib.dir_cin_dw_sym_f16x2_f32(_ptr(x0), _ptr(G), m, H, D, B, 1 if accumulate else 0, _ptr(dW), _ptr(ws), nbytes,
                                                    _ptr(g_absmax_bits), _stream()))
            return dW
        _lib.check(lib.dir_cin_dw_sym_bf16x3_f32(_ptr(x0), _ptr(G), m, H, D, B, 1 if accumulate else 0, _ptr(dW), _ptr(ws), nbytes, _stream()))
        return dW
    if arith == "auto":
        arith = cin_dw_auto_arith(m, D, Hp, H)
        if arith == "bf16x3" and CIN_BWD_SPLIT == "f16x2":
            arith = "f16x2"
    if arith == "f16x2":
        if D not in (8, 16, 32):
            raise ValueError("cin_dw: arith='f16x2' covers D in {8, 16, 32} (got %d)" % D)
        nbytes = int(lib.dir_cin_dw_f16x2_workspace_bytes(m, Hp, H, D, B))
        ws = torch.empty(max(16, nbytes), dtype=torch.uint8, device=x0.device)
        _lib.check(lib.dir_cin_dw_f16x2_f32(_ptr(x0), _ptr(xk), _ptr(G), m, Hp, H, D, B, 1 if accumulate else 0, _ptr(dW), _ptr(ws), nbytes,
                                            _ptr(g_absmax_bits), _stream()))
        return dW
    if arith == "bf16x3":
        if D not in (8, 16, 32):
            raise ValueError("cin_dw: arith='bf16x3' covers D in {8, 16, 32} (got %d)" % D)
        nbytes = int(lib.dir_cin_dw_bf16x3_workspace_bytes(m, Hp, H, D, B))
        ws = torch.empty(max(16, nbytes), dtype=torch.uint8, device=x0.device)
        _lib.check(lib.dir_cin_dw_bf16x3_f32(_ptr(x0), _ptr(xk), _ptr(G), m, Hp, H, D, B, 1 if accumulate else 0, _ptr(dW), _ptr(ws), nbytes,
                                             _stream()))
        return dW
    ws = torch.empty(max(16, int(lib.dir_cin_dw_workspace_bytes(m, Hp, H, D, B))), dtype=torch.uint8, device=x0.device)
    _lib.check(lib.dir_cin_dw_f32(_ptr(x0), _ptr(xk), _ptr(G), m, Hp, H, D, B, 1 if accumulate else 0, _ptr(dW), _ptr(ws),
                                  _stream()))
    return dW


CIN_MAX_FIELDS = 40   # largest register-resident operand of dir_cin_layer_f32
CIN_FIELD_TILES = (8, 16, 26, 40)   # field counts dir_cin_layer_f32 is instantiated for


def cin_dx(x0, xk, W, G):
    """Both data gradients of one CIN layer in one pass (include/dir_hip.h, dir_cin_dx_f32): -> (dxk [B,Hp,D],
    dx0 [B,m,D]).  Needs H <= 256, Hp <= 256, m <= 64.  W is permuted here to the kernel's LDS image
    Wp[nb][j][h][n][cc] = W[h, (nb*32*ct + 32*cc + n)*m + j] (a 1.7 MB tensor at the BASELINE shape)."""
    for t, n in ((x0, "x0"), (xk, "xk"), (W, "W"), (G, "G")):
        _dev(t, torch.float32, n)
        if not t.is_contiguous():
            raise ValueError("cin_dx operands must be contiguous")
    B, m, D = x0.shape
    Hp, H = xk.shape[1], W.shape[0]
    if W.shape[1] != Hp * m or G.shape != (B, H, D) or xk.shape[0] != B or xk.shape[2] != D:
        raise ValueError("cin_dx: W must be [H, Hp*m], xk [B,Hp,D], G [B,H,D]")
    ct = 1 if Hp <= 32 else 2 if Hp <= 64 else 4
    nb = -(-Hp // (32 * ct))                                 # column blocks (2 when 128 < Hp <= 256)
    W3 = W.view(H, Hp, m)
    if nb * 32 * ct != Hp:
        W3 = torch.nn.functional.pad(W3, (0, 0, 0, nb * 32 * ct - Hp))
    Wp = W3.reshape(H, nb, ct, 32, m).permute(1, 4, 0, 3, 2).contiguous()       # [nb][m][H][32 n][ct]
    dxk = torch.empty((B, Hp, D), dtype=torch.float32, device=x0.device)
    dx0 = torch.empty((B, m, D), dtype=torch.float32, device=x0.device)
    _lib.check(_lib.load().dir_cin_dx_f32(_ptr(x0), _ptr(xk), _ptr(Wp), _ptr(G), m, Hp, H, D, B, _ptr(dxk), _ptr(dx0), _stream()))
    return dxk, dx0


def cin_dx_bf16x3(x0, xk, W, G, add_pooled=None, dx0=None, split=None, g_bits_out=None):
    """Both data gradients of one CIN layer on the bf16x3 kernel (include/dir_hip.h, dir_cin_layer_dot_bf16x3_f32): the forward
    contraction on the permuted weight W1[i, h*m+j] = W[h, i*m+j] with G as its left operand gives dxk, and the same T_j tiles dotted
    with xk give dx0 (partial sums per column block and half of i, added in a fixed order by dir_sum_partials_f32).
    add_pooled [B, Hp] (unit column stride): added to dxk[b, i, :] in the kernel's epilogue (dir_cin_layer_dot_add_bf16x3_f32) -- in the
    backward of a stack that sum is the layer below's dL/dxout.  dx0: a [B, m, D] tensor the partial sums are ACCUMULATED into (the
    running dx0 of a stack); None: a new tensor.  split: "bf16x3" | "f16x2" (dir_cin_layer_dot_add_f16x2_f32: G's rows scaled by powers
    of two inside the kernel) | None = CIN_BWD_SPLIT.  g_bits_out (a list, fp16 x 2 only): a [1] int32 tensor with the bit pattern of
    max |G| -- a by-product of the kernel's row maxima -- is appended (cin_dw's g_absmax_bits).  -> (dxk [B,Hp,D], dx0 [B,m,D])."""
    split = split or CIN_BWD_SPLIT
    if split not in ("bf16x3", "f16x2"):
        raise ValueError("cin_dx_bf16x3: split must be 'bf16x3' or 'f16x2'")
    for t, n in ((x0, "x0"), (xk, "xk"), (W, "W"), (G, "G")):
        _dev(t, torch.float32, n)
        if not t.is_contiguous():
            raise ValueError("cin_dx_bf16x3 operands must be contiguous")
    B, m, D = x0.shape
    Hp, H = xk.shape[1], W.shape[0]
    if W.shape[1] != Hp * m or G.shape != (B, H, D) or xk.shape[0] != B or xk.shape[2] != D:
        raise ValueError("cin_dx_bf16x3: W must be [H, Hp*m], xk [B,Hp,D], G [B,H,D]")
    if not cin_bf16x3_covers(m, D):
        raise ValueError("cin_dx_bf16x3 covers m <= 40 and D in {4,8,16,32} (got m=%d, D=%d)" % (m, D))
    lib = _lib.load()
    W1 = W.view(H, Hp, m).permute(1, 0, 2).reshape(Hp, H * m).contiguous()
    # roles in the kernel: left operand G (reduction over its H channels), output columns = the Hp channels of xk
    nbytes = int(lib.dir_cin_bf16x3_workspace_bytes(m, H, Hp))
    P = int(lib.dir_cin_bf16x3_dot_partials(m, H, Hp))
    ws = torch.empty(max(16, nbytes), dtype=torch.uint8, device=x0.device)
    dxk = torch.empty((B, Hp, D), dtype=torch.float32, device=x0.device)
    parts = torch.empty((P, B, m, D), dtype=torch.float32, device=x0.device)
    if add_pooled is not None:
        _dev(add_pooled, torch.float32, "add_pooled")
        if tuple(add_pooled.shape) != (B, Hp) or (B > 0 and add_pooled.stride(1) != 1):
            raise ValueError("cin_dx_bf16x3: add_pooled must be [B, Hp] with unit column stride")
    ald = add_pooled.stride(0) if add_pooled is not None and B > 0 else Hp
    if split == "f16x2":
        gbits = torch.empty(1, dtype=torch.int32, device=x0.device) if (g_bits_out is not None and B > 0) else None
        _lib.check(lib.dir_cin_layer_dot_add_f16x2_f32(_ptr(x0), _ptr(G), _ptr(W1), _ptr(xk), m, H, Hp, D, B, _ptr(add_pooled), ald, _ptr(dxk),
                                                       _ptr(parts), _ptr(ws), nbytes, _ptr(gbits), _stream()))
        if gbits is not None:
            g_bits_out.append(gbits)
    else:
        _lib.check(lib.dir_cin_layer_dot_add_bf16x3_f32(_ptr(x0), _ptr(G), _ptr(W1), _ptr(xk), m, H, Hp, D, B, _ptr(add_pooled), ald, _ptr(dxk),
                                                        _ptr(parts), _ptr(ws), nbytes, _stream()))
    acc = dx0 is not None
    if acc and (tuple(dx0.shape) != (B, m, D) or not dx0.is_contiguous() or dx0.dtype != torch.float32):
        raise ValueError("cin_dx_bf16x3: dx0 must be a contiguous float32 [B, m, D] tensor")
    n = B * m * D
    if n % 4 or B == 0:
        s_ = parts[0] if P == 1 else parts.sum(dim=0)
        return dxk, (dx0.add_(s_) if acc else s_)
    out = dx0 if acc else torch.empty((B, m, D), dtype=torch.float32, device=x0.device)
    _lib.check(lib.dir_sum_partials_f32(_ptr(parts), P, n, 1 if acc else 0, _ptr(out), _stream()))
    return dxk, out


def cin_layer_backward(x0, xk, W, G, need_x0=True, need_xk=True, need_w=True, force_forward_form=False, arith=None):
    """Backward of cin_layer given G = dL/dxout [B,H,D] (pooled gradient already broadcast in):
    -> (dx0 [B,m,D] | None, dxk [B,Hp,D] | None, dW [H,Hp*m] | None).
    Data gradients: cin_dx_bf16x3 (bf16 pipe, one pass for both) where arith (None = CIN_ARITH; "auto": cin_auto_arith of the
    transposed problem) selects it; dir_cin_dx_f32 (fp32 MFMA, one pass for both) when H, Hp <= 256 and m <= 64; otherwise, or
    with force_forward_form, the forward contraction with permuted weights (include/dir_hip.h):
      dxk = cin_layer(x0, G, W1),  W1[i, h*m+j]  = W[h, i*m+j]
      dx0 = sum over channel groups g of cin_layer(xk[:, g], G, W2g),  W2g[j, h*mg+ig] = W[h, (g0+ig)*m+j]."""
    B, m, D = x0.shape
    Hp, H = xk.shape[1], W.shape[0]
    W3 = W.view(H, Hp, m)
    dxk = dx0 = dW = None
    arith = arith or CIN_ARITH
    dw_arith = arith                                 # cin_dw resolves "auto" by its own rule
    split = None if arith == "auto" else "bf16x3"    # an explicit "bf16x3" means that arithmetic; "auto" follows CIN_BWD_SPLIT
    if arith == "auto":
        arith = cin_auto_arith(m, D, H, Hp)          # the data gradients' GEMM: reduction over H, Hp output columns
    if (need_xk or need_x0) and arith == "bf16x3" and cin_bf16x3_covers(m, D) and not force_forward_form:
        dxk, dx0 = cin_dx_bf16x3(x0, xk, W, G, split=split)
        need_xk = need_x0 = False
    if (need_xk or need_x0) and H <= 256 and Hp <= 256 and m <= 64 and not force_forward_form:
        dxk, dx0 = cin_dx(x0, xk, W, G)             # one pass for both (G stationary in registers)
        need_xk = need_x0 = False
    if need_xk:
        W1 = W3.permute(1, 0, 2).reshape(Hp, H * m).contiguous()
        dxk, _ = cin_layer(x0, G, W1, grad_operand=True)
    if need_x0:
        for g0 in range(0, Hp, CIN_MAX_FIELDS):
            mg = min(CIN_MAX_FIELDS, Hp - g0)
            xg = xk if mg == Hp else xk[:, g0:g0 + mg, :].contiguous()
            W2 = W3[:, g0:g0 + mg, :].permute(2, 0, 1).reshape(m, H * mg).contiguous()
            part, _ = cin_layer(xg, G, W2, grad_operand=True)
            dx0 = part if dx0 is None else dx0.add_(part)
    if need_w:
        dW = cin_dw(x0, xk, G, arith=dw_arith)
    return dx0, dxk, dW


def cin_stack_backward(x0, xks, Ws, g_pooled, need_x0=True, arith=None, z_top=None):
    """Backward of a whole CIN stack (xDeepFM: pooled = concat_k sum_d X^k) given g_pooled [B, sum H_k] (unit column stride): xks[k] is
    layer k's input (xks[0] IS x0), Ws[k] [H_k, H_{k-1} * m].  -> (dx0 [B, m, D] | None, [dW_k]).  Layer by layer from the top as
    cin_layer_backward, but the sum dL/dxout_k = dL/dxk_{k+1} + g_pooled_k (broadcast over d) comes out of layer k+1's data-gradient
    kernel (its epilogue adds the pooled gradient), and every layer's dx0 share is accumulated into one tensor by the partial-sum pass:
    no [B, H, D] add and no [B, m, D] add per layer.  The TOP layer's map feeds only its pooled sums, so its backward is the pooled form
    (csrc/cin_pool.hip): dW = g^T Z and dZ = g W on the dense kernels, the data gradients by cin_pool_dx -- 1/D of the matrix work;
    z_top: that layer's Z from the forward (cin_layer(..., want_xout=False, z_out=...)), recomputed when None."""
    B, m, D = x0.shape
    L = len(Ws)
    Hs = [int(W.shape[0]) for W in Ws]
    if tuple(g_pooled.shape) != (B, sum(Hs)) or (B > 0 and g_pooled.stride(1) != 1):
        raise ValueError("cin_stack_backward: g_pooled must be [B, sum(H_k)] with unit column stride")
    offs = [0]
    for h in Hs:
        offs.append(offs[-1] + h)
    gps = [g_pooled[:, offs[k]:offs[k + 1]] for k in range(L)]
    arith = arith or CIN_ARITH
    dWs, dx0, G = [None] * L, None, None
    for k in range(L - 1, -1, -1):
        xk, W, H = xks[k], Ws[k], Hs[k]
        Hp = xk.shape[1]
        if (G is None and arith != "f32" and CIN_POOLED_LAST and B > 0 and cin_pooled_covers(m, D, Hp) and H % 4 == 0
                and gps[k].stride(0) % 4 == 0 and gps[k].data_ptr() % 16 == 0 and xk.is_contiguous()):
            # the top layer in its pooled form: everything on [B, Hp*m] rows
            gk = gps[k]
            Z = z_top if z_top is not None else cin_pool_z(x0, xk)
            # dW = g^T Z as (Z^T g)^T: Hp*m output rows fill dense_dw's 256-row blocks, H = 128 rows would leave half of each empty
            # (394 against 619 us at the BASELINE shape)
            dWs[k] = dense_dw(Z, gk).t().contiguous()
            if k == 0 and not need_x0:
                break
            gb = grad_bits(gk, want_all=False) if (arith == "auto" and CIN_BWD_SPLIT == "f16x2") else None
            dZ = dense(gk, W.t(), row_bits=gb[0] if gb is not None else None)      # dL/dZ = g W: the gradient's rows scaled inside the fp16 x 2 kernel
            G, dx0 = cin_pool_dx(x0, xk, dZ, add_pooled=gps[k - 1] if k > 0 else None, dx0=dx0)
            continue
        if G is None:                                        # the top layer's map feeds nothing but its pooled sums
            G = gps[k].reshape(B, H, 1).expand(B, H, D).contiguous()
        a = cin_auto_arith(m, D, H, Hp) if arith == "auto" else arith
        dot_first = (k > 0 and arith == "auto" and a == "bf16x3" and cin_bf16x3_covers(m, D) and CIN_BWD_SPLIT == "f16x2"
                     and cin_dw_auto_arith(m, D, Hp, H) == "bf16x3")
        if dot_first:
            # fp16 x 2 on both kernels: the data-gradient kernel runs FIRST and leaves max |G| (its row maxima's by-product) for the weight
            # gradient's tensor scale -- no max pass over the 537 MB of G (95 us at the BASELINE shape)
            gb = []
            dxk, dx0 = cin_dx_bf16x3(x0, xk, W, G, add_pooled=gps[k - 1], dx0=dx0, g_bits_out=gb)
            dWs[k] = cin_dw(x0, xk, G, arith=arith, g_absmax_bits=gb[0] if gb else None)
            G = dxk
            continue
        first = k == 0 and xk.data_ptr() == x0.data_ptr() and xk.shape == x0.shape
        if first and need_x0:
            # The first layer (xk IS x0): x0 receives dL/dxk and dL/dx0 of this layer, and their sum is ONE forward-form contraction of G
            # with the symmetrised weights W[h,i,j] + W[h,j,i] (0.96 ms against 1.34 ms for the two-output form at the BASELINE shape,
            # tools/cin_l1_dx_probe.py).  It runs FIRST: its row-scaled fp16 x 2 kernel leaves max |G| for the weight gradient's scale.
            W3 = W.view(H, m, m)
            Ws_ = (W3 + W3.transpose(1, 2)).permute(1, 0, 2).reshape(m, H * m).contiguous()          # [i, h*m + j]
            gb = []
            tot, _ = cin_layer(x0, G, Ws_, arith=arith, grad_operand=True, g_bits_out=gb)
            dWs[k] = cin_dw(x0, xk, G, arith=arith, g_absmax_bits=gb[0] if gb else None)
            return (tot if dx0 is None else dx0.add_(tot)), dWs
        dWs[k] = cin_dw(x0, xk, G, arith=arith)
        if k == 0 and not need_x0:
            break
        below = gps[k - 1] if k > 0 else None
        if a == "bf16x3" and cin_bf16x3_covers(m, D):
            dxk, dx0 = cin_dx_bf16x3(x0, xk, W, G, add_pooled=below, dx0=dx0, split=None if arith == "auto" else "bf16x3")
        else:
            d0, dxk, _ = cin_layer_backward(x0, xk, W, G, need_w=False, arith=arith)
            dx0 = d0 if dx0 is None else dx0.add_(d0)
            if below is not None:
                dxk = dxk.add_(below.unsqueeze(2))
        G = dxk
    if need_x0:
        dx0 = dx0.add_(G)                                    # layer 1's xk IS x0: its dL/dxk is the last share
    return (dx0 if need_x0 else None), dWs


class _SortedShare:
    """Two sorted sparse updates of one training step that see the same (row, entry) pairs -- DeepFM's Adagrad on the embedding tables
    and FTRL on the linear columns of the same categorical columns (deepFM.py:58,61) -- sort once: whichever runs first in backward()
    leaves a token (the id tensor's identity and version, its layout, the stream) and its workspace, the other takes the sorted pairs
    from it (include/dir_hip.h: dir_sparse_*_sorted_*_from_f32).  A token is used at most once and only by the other optimiser."""

    def __init__(self):
        self.token = None
        self.owner = None
        self.ws = None
        self.ws_tensor = None
        self.hits = 0                       # sorts saved so far

    @staticmethod
    def _token(ids, B, sb, sf):
        return (ids.data_ptr(), ids._version, B, sb, sf, torch.cuda.current_stream(ids.device).cuda_stream)

    def take(self, who, ids, B, sb, sf):
        """-> the other optimiser's workspace pointer if it sorted exactly these entries last, else None."""
        if self.token is not None and self.owner is not who and self.token == self._token(ids, B, sb, sf) \
                and self.owner._ws is self.ws_tensor:               # (the owner's workspace has not been reallocated since)
            ws = self.ws
            self.token = None
            self.hits += 1
            return ws
        return None

    def leave(self, who, ids, B, sb, sf, ws):
        self.token, self.owner, self.ws, self.ws_tensor = self._token(ids, B, sb, sf), who, ws, who._ws


def share_sorted_entries(a, b):
    """Let two sorted sparse optimisers over table sets with the same vocabularies share one sort per step (see _SortedShare).
    -> True if they were linked."""
    if list(a.ts.vocab) != list(b.ts.vocab) or a.ts.F != b.ts.F or getattr(a, "method", "sorted") != "sorted" \
            or os.environ.get("DIR_SHARE_SORT", "1") == "0":           # (development switch for A/B runs)
        return False
    a._share = b._share = _SortedShare()
    return True


class SparseAdagrad:
    """Fused sparse Adagrad over a TableSet.  Holds the accumulators ([TF-upstream] initial_accumulator_value = 0.1).
    method "sorted" (default; include/dir_hip.h: dir_sparse_adagrad_sorted_f32): radix sort of (row, entry) pairs +
    per-tile segmented reduce -- skew-proof and bitwise reproducible.  method "chains" (dir_sparse_adagrad_f32): per-row
    chains built with integer atomics -- a little faster on near-unique ids, serialises on hot rows.
    A TableSet.train_rows set brings its accumulators with it (packed [embedding | accumulator] rows: one read and one write
    per touched row; dir_sparse_adagrad_sorted_rows_f32); step_fm folds the FM backward into the update on either layout."""

    def __init__(self, tables, lr, initial_accumulator_value=0.1, method="sorted"):
        if method not in ("sorted", "chains"):
            raise ValueError("method must be 'sorted' or 'chains'")
        self.ts = _as_tableset(tables)
        self.lr = float(lr)
        self.method = method
        dev = self.ts.device
        if self.ts.accums is not None:           # packed training rows: the accumulators live beside the embeddings
            if method != "sorted":
                raise ValueError("packed training rows are updated by the sorted method")
            self.accums = self.ts.accums
        else:
            self.accums = [torch.full_like(t, initial_accumulator_value) for t in self.ts.tables]
        self.acc_ptrs = torch.tensor([a.data_ptr() for a in self.accums], dtype=torch.int64, device=dev)
        base = [0]
        for v in self.ts.vocab[:-1]:
            base.append(base[-1] + v)
        self.total_rows = sum(self.ts.vocab)
        self.head_base = torch.tensor(base, dtype=torch.int64, device=dev)
        self.head = torch.full((self.total_rows,), -1, dtype=torch.int32, device=dev) if method == "chains" else None
        self._next = None
        self._ws = None
        self._share = None

    def _rows_update(self, lib, ids, B, sb, sf, grad, fm_g, fm_sum):
        """dir_sparse_adagrad_sorted_rows_f32, or its _from form when the linked optimiser has just sorted the same entries."""
        ts = self.ts
        ws, need = self._sorted_ws(lib, B)
        src = self._share.take(self, ids, B, sb, sf) if self._share is not None else None
        args = (_ptr(ts._ptrs), _ptr(self.acc_ptrs), ts.ld, ts.F, ts.K, _ptr(ids), sb, sf, _ptr(grad), grad.stride(0) if grad is not None else 0,
                _ptr(fm_g), _ptr(fm_sum), self.lr, B, _ptr(self.head_base), self.total_rows, ws, need)
        self._written()
        if src is not None:
            _lib.check(lib.dir_sparse_adagrad_sorted_rows_from_f32(*args, src, _stream()))
            return
        _lib.check(lib.dir_sparse_adagrad_sorted_rows_f32(*args, _stream()))
        if self._share is not None:
            self._share.leave(self, ids, B, sb, sf, ws)

    def _written(self):
        """The update kernels write the tables and accumulators through raw pointers: version counters bumped here (ops.mark_written)."""
        if getattr(self.ts, "arena", None) is not None:
            self.ts.written()
        else:
            self.ts.written(*self.accums)

    def attach(self):
        """Consume the gather's row gradients directly in backward (autograd.GatherFm): loss.backward() then
        performs the embedding update itself; the tables get no .grad."""
        self.ts.grad_sink = self.step
        if self.method == "sorted":
            self.ts.fm_sink = self.step_fm
        return self

    def _sorted_ws(self, lib, B):
        ts = self.ts
        need = int(lib.dir_sparse_adagrad_sorted_workspace_bytes(B, ts.F, ts.K, self.total_rows))
        if need <= 0:
            raise _lib.DirError(-4, "sparse_adagrad_sorted: unsupported size (B*F < 2^31, total rows < 2^32-1)")
        if self._ws is None or self._ws.numel() < need + 256:
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=ts.device)
        off = (-self._ws.data_ptr()) % 256
        return ctypes.c_void_p(self._ws.data_ptr() + off), need

    def step_fm(self, ids, grad, fm_g, fm_sum):
        """The update with the FM backward folded in (include/dir_hip.h: dir_sparse_adagrad_sorted_rows_f32): entry (b, f)'s
        gradient is (fm_sum[b] - row) * fm_g[b] + grad[b, f].  grad [B, F*K] or None, fm_g [B] / [B, 1], fm_sum [B, K] from
        gather_fm(..., fsum=).  Tables bit-identical to fm_logit_backward(add_in=grad) followed by step()."""
        ts = self.ts
        _dev(ids, torch.int64, "ids")
        B, sb, sf = _onehot_strides(ids, ts.F)
        if grad is not None:
            _dev(grad, torch.float32, "grad")
            if grad.shape != (B, ts.F * ts.K) or grad.stride(1) != 1:
                raise ValueError("grad must be [B, F*K] with unit inner stride")
        _dev(fm_g, torch.float32, "fm_g")
        _dev(fm_sum, torch.float32, "fm_sum")
        if fm_g.numel() != B or not fm_g.is_contiguous() or fm_sum.shape != (B, ts.K) or not fm_sum.is_contiguous():
            raise ValueError("fm_g must be a contiguous [B] / [B, 1] tensor and fm_sum a contiguous [B, K] one")
        if B == 0:
            return
        self._rows_update(_lib.load(), ids, B, sb, sf, grad, fm_g, fm_sum)

    def step(self, ids, grad):
        """ids [B, F] int64 (any strides), grad [B, F*K] fp32: d loss / d gathered rows."""
        ts = self.ts
        _dev(ids, torch.int64, "ids")
        _dev(grad, torch.float32, "grad")
        B, sb, sf = _onehot_strides(ids, ts.F)
        if grad.shape != (B, ts.F * ts.K) or grad.stride(1) != 1:
            raise ValueError("grad must be [B, F*K] with unit inner stride")
        lib = _lib.load()
        if ts.ld != ts.K or (self._share is not None and self.method == "sorted"):
            if B == 0:
                return
            self._rows_update(lib, ids, B, sb, sf, grad, None, None)
            return
        if self.method == "chains":
            if self._next is None or self._next.numel() < B * ts.F:
                self._next = torch.empty(B * ts.F, dtype=torch.int32, device=ts.device)
            _lib.check(lib.dir_sparse_adagrad_f32(_ptr(ts.ptrs), _ptr(self.acc_ptrs), ts.F, ts.K, _ptr(ids), sb, sf,
                                                  _ptr(grad), grad.stride(0), self.lr, B, _ptr(self.head_base),
                                                  self.total_rows, _ptr(self.head), _ptr(self._next), _stream()))
            self._written()
            return
        if B == 0:
            return
        need = int(lib.dir_sparse_adagrad_sorted_workspace_bytes(B, ts.F, ts.K, self.total_rows))
        if need <= 0:
            raise _lib.DirError(-4, "sparse_adagrad_sorted: unsupported size (B*F < 2^31, total rows < 2^32-1)")
        if self._ws is None or self._ws.numel() < need + 256:
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=ts.device)
        off = (-self._ws.data_ptr()) % 256
        _lib.check(lib.dir_sparse_adagrad_sorted_f32(_ptr(ts.ptrs), _ptr(self.acc_ptrs), ts.F, ts.K, _ptr(ids), sb, sf,
                                                     _ptr(grad), grad.stride(0), self.lr, B, _ptr(self.head_base),
                                                     self.total_rows, ctypes.c_void_p(self._ws.data_ptr() + off), need, _stream()))
        self._written()

    def step_payload(self, payload, grad):
        """Owner side of a sharded backward (shard.ShardedTables): payload [n] int64 (local_row * F + slot, < 0 pruned) as
        received from all ranks, grad [n, K] in the same order (include/dir_hip.h: dir_sparse_adagrad_sorted_payload_f32)."""
        ts = self.ts
        _dev(payload, torch.int64, "payload")
        _dev(grad, torch.float32, "grad")
        n = payload.numel()
        if grad.shape != (n, ts.K) or not grad.is_contiguous() or not payload.is_contiguous():
            raise ValueError("grad must be a contiguous [n, K] tensor matching the payload")
        if n == 0:
            return
        lib = _lib.load()
        ws, need = _sorted_ws(self, lib, n, 1, ts.K, self.total_rows, ts.device)
        _lib.check(lib.dir_sparse_adagrad_sorted_payload_f32(_ptr(ts.ptrs), _ptr(self.acc_ptrs), ts.F, ts.K, _ptr(payload), n,
                                                             _ptr(grad), self.lr, _ptr(self.head_base), self.total_rows, ws, need,
                                                             _stream()))
        self._written()


class SparseAdam:
    """tf.train.AdamOptimizer on the embedding tables of a TableSet, in place, from the gather's row gradients (include/dir_hip.h:
    dir_sparse_adam_f32; the reference's train_op: DeepCrossNetwork.py:264-290, train.py:119-124).  [TF-upstream] semantics: every row of
    m, v and var moves every step (IndexedSlices gradients are zero outside the looked-up rows), each table's gradient is clipped on its
    own with tf.clip_by_norm(g, clip_norm) first.  Holds m and v (zeros, like the slots TF creates).  Set .lr_t before every step
    (lr * sqrt(1 - beta2^t) / (1 - beta1^t), t = global_step + 1); attach() makes loss.backward() perform the update."""

    def __init__(self, tables, beta1=0.9, beta2=0.999, eps=1e-8, clip_norm=0.0):
        self.ts = _as_tableset(tables)
        ts = self.ts
        if ts.ld != ts.K:
            raise ValueError("SparseAdam: plain [vocab, K] tables")
        self.beta1, self.beta2, self.eps, self.clip_norm = float(beta1), float(beta2), float(eps), float(clip_norm)
        self.lr_t = 0.0
        self.ms = [torch.zeros_like(t) for t in ts.tables]
        self.vs = [torch.zeros_like(t) for t in ts.tables]
        dev = ts.device
        self.m_ptrs = torch.tensor([t.data_ptr() for t in self.ms], dtype=torch.int64, device=dev)
        self.v_ptrs = torch.tensor([t.data_ptr() for t in self.vs], dtype=torch.int64, device=dev)
        base = [0]
        for v in ts.vocab[:-1]:
            base.append(base[-1] + v)
        self.total_rows = sum(ts.vocab)
        self.row_base = torch.tensor(base, dtype=torch.int64, device=dev)
        self._ws = None
        self._ws_B = -1
        self.steps = 0

    @staticmethod
    def covers(ts):
        return ts.ld == ts.K and ts.F <= 64 and ts.K % 4 == 0 and 64 % (ts.K // 4) == 0 and sum(ts.vocab) < 0xffffffff

    def attach(self):
        self.ts.grad_sink = self.step
        return self

    def step(self, ids, grad):
        """ids [B, F] int64 (any strides), grad [B, F*K] fp32 = d loss / d gathered rows: one Adam step of ALL rows of every table."""
        ts = self.ts
        _dev(ids, torch.int64, "ids")
        _dev(grad, torch.float32, "grad")
        B, sb, sf = _onehot_strides(ids, ts.F)
        if grad.shape != (B, ts.F * ts.K) or grad.stride(1) != 1:
            raise ValueError("grad must be [B, F*K] with unit inner stride")
        lib = _lib.load()
        need = int(lib.dir_sparse_adam_workspace_bytes(B, ts.F, ts.K, self.total_rows))
        if need <= 0:
            raise _lib.DirError(-4, "sparse_adam: unsupported size (F <= 64, K / 4 a power of two, B*F < 2^31, total rows < 2^32-1)")
        first = 0
        if self._ws is None or self._ws_B < B:
            # the marks (behind a 512-byte header) must survive from step to step: the buffer is allocated once per batch size (grown,
            # never shrunk), zeroed by the first call
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=ts.device)
            self._ws_B = B
            first = 1
        off = (-self._ws.data_ptr()) % 256
        _lib.check(lib.dir_sparse_adam_f32(_ptr(ts.ptrs), _ptr(self.m_ptrs), _ptr(self.v_ptrs), ts.F, ts.K, _ptr(ids), sb, sf, _ptr(grad),
                                           grad.stride(0), self.lr_t, self.beta1, self.beta2, self.eps, self.clip_norm, B, _ptr(self.row_base),
                                           self.total_rows, ctypes.c_void_p(self._ws.data_ptr() + off), self._ws.numel() - off, first, _stream()))
        ts.written(*self.ms, *self.vs)
        self.steps += 1


    def step_table_grad(self, f, grad):
        """Table f's step from a gradient that arrived as its .grad instead of through the sink -- multi-hot / weighted columns, whose
        backward (autograd.EmbeddingBag) returns sparse row gradients: the same tf.train.AdamOptimizer step of ALL rows (clip_by_norm on
        the table's gradient first) on this optimiser's m / v, with torch ops (not a hot path: the one-hot lookups take the HIP update)."""
        t, m, v = self.ts.tables[f], self.ms[f], self.vs[f]
        g = (grad.coalesce().to_dense() if grad.is_sparse else grad).to(torch.float32)
        if self.clip_norm > 0:
            g = g * (self.clip_norm / torch.clamp(torch.linalg.vector_norm(g), min=self.clip_norm))
        m.mul_(self.beta1).add_(g, alpha=1.0 - self.beta1)
        v.mul_(self.beta2).addcmul_(g, g, value=1.0 - self.beta2)
        t.addcdiv_(m, v.sqrt().add_(self.eps), value=-self.lr_t)
        mark_written(*self.ts.owners[f:f + 1])


def _sorted_ws(holder, lib, n_entries, F, K, total_rows, device):
    need = int(lib.dir_sparse_adagrad_sorted_workspace_bytes(n_entries, F, K, total_rows))
    if need <= 0:
        raise _lib.DirError(-4, "sorted sparse update: unsupported size (entries < 2^31, total rows < 2^32-1)")
    if holder._ws is None or holder._ws.numel() < need + 256:
        holder._ws = torch.empty(need + 256, dtype=torch.uint8, device=device)
    off = (-holder._ws.data_ptr()) % 256
    return ctypes.c_void_p(holder._ws.data_ptr() + off), need


class SparseFtrl:
    """Fused sparse FTRL-Proximal over a TableSet (include/dir_hip.h: dir_sparse_ftrl_sorted_f32; the reference's
    linear_optimizer='Ftrl', deepFM.py:58, with [TF-upstream] tf.train.FtrlOptimizer defaults: initial accumulator 0.1,
    learning_rate_power -0.5, l1 = l2 = 0).  step(ids, grad): grad is [B, F*K] (one gradient row per slot) or [B, K] (the same
    row for every slot: the linear term, whose d logit is shared by all columns)."""

    def __init__(self, tables, lr=0.2, initial_accumulator_value=0.1, l1=0.0, l2=0.0):
        self.ts = _as_tableset(tables)
        self.lr, self.l1, self.l2 = float(lr), float(l1), float(l2)
        dev = self.ts.device
        self.packed = getattr(self.ts, "linears", None) is not None      # TableSet.ftrl_rows: n and z live in the weights' rows
        if self.packed:
            self.accums, self.linears = self.ts.accums, self.ts.linears
        else:
            if self.ts.ld != self.ts.K:
                raise ValueError("SparseFtrl: tables with a row stride must be TableSet.ftrl_rows")
            self.accums = [torch.full_like(t, initial_accumulator_value) for t in self.ts.tables]
            self.linears = [torch.zeros_like(t) for t in self.ts.tables]
        self.acc_ptrs = torch.tensor([a.data_ptr() for a in self.accums], dtype=torch.int64, device=dev)
        self.lin_ptrs = torch.tensor([z.data_ptr() for z in self.linears], dtype=torch.int64, device=dev)
        base = [0]
        for v in self.ts.vocab[:-1]:
            base.append(base[-1] + v)
        self.total_rows = sum(self.ts.vocab)
        self.row_base = torch.tensor(base, dtype=torch.int64, device=dev)
        self._ws = None
        self._share = None

    def attach(self):
        """Consume the linear term's gradient directly in backward (autograd.LinearLogit)."""
        self.ts.grad_sink = self.step
        return self

    def step(self, ids, grad):
        ts = self.ts
        _dev(ids, torch.int64, "ids")
        _dev(grad, torch.float32, "grad")
        B, sb, sf = _onehot_strides(ids, ts.F)
        if grad.dim() != 2 or grad.shape[0] != B or grad.stride(1) != 1 or grad.shape[1] not in (ts.K, ts.F * ts.K):
            raise ValueError("grad must be [B, F*K] or [B, K] with unit inner stride")
        if B == 0:
            return
        slot_stride = ts.K if grad.shape[1] == ts.F * ts.K and ts.F > 1 else 0
        lib = _lib.load()
        need = int(lib.dir_sparse_adagrad_sorted_workspace_bytes(B, ts.F, ts.K, self.total_rows))
        if need <= 0:
            raise _lib.DirError(-4, "sparse_ftrl_sorted: unsupported size (B*F < 2^31, total rows < 2^32-1)")
        if self._ws is None or self._ws.numel() < need + 256:
            self._ws = torch.empty(need + 256, dtype=torch.uint8, device=ts.device)
        off = (-self._ws.data_ptr()) % 256
        ws = ctypes.c_void_p(self._ws.data_ptr() + off)
        src = self._share.take(self, ids, B, sb, sf) if self._share is not None else None
        ts.written(*self.accums, *self.linears)
        if self.packed:
            _lib.check(lib.dir_sparse_ftrl_rows_sorted_f32(_ptr(ts._ptrs), ts.F, _ptr(ids), sb, sf, _ptr(grad), grad.stride(0), slot_stride,
                                                           self.lr, self.l1, self.l2, B, _ptr(self.row_base), self.total_rows, ws, need, src,
                                                           _stream()))
            if src is None and self._share is not None:
                self._share.leave(self, ids, B, sb, sf, ws)
            return
        args = (_ptr(ts.ptrs), _ptr(self.acc_ptrs), _ptr(self.lin_ptrs), ts.F, ts.K, _ptr(ids), sb, sf, _ptr(grad), grad.stride(0), slot_stride,
                self.lr, self.l1, self.l2, B, _ptr(self.row_base), self.total_rows, ws, need)
        if src is not None:
            _lib.check(lib.dir_sparse_ftrl_sorted_from_f32(*args, src, _stream()))
            return
        _lib.check(lib.dir_sparse_ftrl_sorted_f32(*args, _stream()))
        if self._share is not None:
            self._share.leave(self, ids, B, sb, sf, ws)


class PackedTables:
    """Serving layout: one [vocab_f, ld] buffer per slot, embedding at columns [0, K), first-order weight at column K
    (include/dir_hip.h: dir_gather_fm_linear_packed_f32).  ld = 32 floats = one 128-byte line for K <= 31.  Built from
    the reference-layout parameters (embedding tables [V,K] and linear weight columns [V])."""

    def __init__(self, emb_tables, lin_weights=None):
        emb_tables = list(emb_tables)
        K = emb_tables[0].shape[1]
        ld = 32
        while ld < K + 1:
            ld *= 2
        self.F, self.K, self.ld = len(emb_tables), K, ld
        self.lin_col = K if lin_weights is not None else -1
        self.device = emb_tables[0].device
        self.vocab = [int(t.shape[0]) for t in emb_tables]
        # one arena, every slot's rows 128-byte aligned
        total = sum(self.vocab) * ld
        arena = torch.zeros(total + 32, dtype=torch.float32, device=self.device)
        off = (-(arena.data_ptr() // 4)) % 32
        self.arena = arena
        self.rows = []
        for f, t in enumerate(emb_tables):
            v = self.vocab[f]
            blk = arena[off:off + v * ld].view(v, ld)
            blk[:, :K] = t
            if lin_weights is not None:
                blk[:, K] = lin_weights[f].reshape(-1)
            self.rows.append(blk)
            off += v * ld
        self.ptrs = torch.tensor([r.data_ptr() for r in self.rows], dtype=torch.int64, device=self.device)
        self.vocab_dev = torch.tensor(self.vocab, dtype=torch.int64, device=self.device)
        self.nbytes = total * 4
        self._absmax = None

    def absmax(self):
        """max |value| over the packed rows, measured once (the rows are a serving snapshot; rebuild the pack after changing them)."""
        if self._absmax is None:
            self._absmax = float(self.arena.abs().max()) if self.arena.numel() else 0.0
        return self._absmax

    def flags(self):
        return STREAM_ROWS if self.nbytes > 2 * INFINITY_CACHE_BYTES else 0


def gather_fm_linear(pt, ids, bias=None, want_emb=True, out=None, fm=None, lin=None):
    """DeepFM's three sparse terms in one pass over packed rows: -> (emb [B,F*K] | None, fm [B,1], lin [B,1] | None)."""
    _dev(ids, torch.int64, "ids")
    B, sb, sf = _onehot_strides(ids, pt.F)
    if want_emb and out is None:
        out = torch.empty((B, pt.F * pt.K), dtype=torch.float32, device=pt.device)
    if fm is None:
        fm = torch.empty((B, 1), dtype=torch.float32, device=pt.device)
    if lin is None and pt.lin_col >= 0:
        lin = torch.empty((B, 1), dtype=torch.float32, device=pt.device)
    _lib.check(_lib.load().dir_gather_fm_linear_packed_f32(_ptr(pt.ptrs), _ptr(pt.vocab_dev), pt.F, pt.K, pt.ld, pt.lin_col, _ptr(ids), sb, sf,
                                                           pt.flags(), B, _ptr(out) if want_emb else None,
                                                           out.stride(0) if want_emb else 0, _ptr(fm), _ptr(bias),
                                                           _ptr(lin), _stream()))
    return (out if want_emb else None), fm, lin
