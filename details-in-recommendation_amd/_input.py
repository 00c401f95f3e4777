"""_input.py -- turn a reference-style `features` dict into the id layout the kernels take.

One-hot columns give a [B] id vector each; stacking them field-major ([F, B], one copy) and passing the
transposed VIEW keeps the C ABI's (stride_b, stride_f) = (1, B) without a second pass.  As soon as one
column is Ragged (multi-hot), every column is expressed as CSR and concatenated field-major
(bag(b, f) = f*B + b), which is exactly what per-column SparseTensors concatenate to.
"""
import os

import torch

# DIR_CHECK_IDS: "1" (default) = TensorFlow's behaviour: the forward that was fed an out-of-range id raises.  The check is one
# fused pass on the device (dir_check_ids, ~6 us); its 4-byte verdict is read at the END of the forward, after every other kernel
# of the forward has been enqueued, so the host waits for a kernel that ran long ago and the GPU never idles (+3 % on the ESMM
# forward).  "deferred" = read a verdict only once it has arrived (never waits; raises at the latest on a later forward or on
# raise_pending(block=True));  "0" = no check (the kernels treat such ids as pruned either way).
CHECK_MODE = {"0": "off", "deferred": "deferred"}.get(os.environ.get("DIR_CHECK_IDS", "1"), "sync")
CHECK_IDS = CHECK_MODE != "off"


def categorical_of(col):
    """The categorical column behind a dense wrapper (embedding / indicator); categorical columns -- including
    weighted_categorical_column, which also carries a `.categorical_column` -- are returned as they are."""
    if getattr(col, "is_dense", False) and hasattr(col, "categorical_column"):
        return col.categorical_column
    return col


def _range_checked(cats):
    """(indices, num_buckets) of the columns whose ids TensorFlow range-checks: categorical_column_with_identity without a
    default_value asserts 0 <= id < num_buckets on the sparse values ([TF-upstream] _IdentityCategoricalColumn; a dense -1
    is the 'missing' marker and is dropped before the check).  Hash / vocabulary / bucketized ids are in range by construction."""
    idx = [i for i, c in enumerate(cats) if getattr(c, "range_checked", False)]
    return idx, [cats[i].num_buckets for i in idx]


_PENDING = []        # checks whose verdict has not been read yet: (pinned flag, event | None, describe())
_SIDE = {}           # device -> (side stream, pinned ring of verdict slots, next slot)


_BOUNDS = {}


def checked_forward(fn):
    """Decorator of a model's forward: the id-range verdicts it posts belong to THAT forward.  If the forward raises midway, its unread
    verdicts are dropped with it instead of surfacing as a ValueError in a later, unrelated forward (of this or another model); on a
    normal return the forward itself has called raise_pending().  (A standalone InputLayer / collect_ids caller reads its verdicts with
    raise_pending(block=True).)"""
    import functools

    @functools.wraps(fn)
    def wrap(self, *args, **kwargs):
        mark = len(_PENDING)
        try:
            return fn(self, *args, **kwargs)
        except BaseException:
            del _PENDING[mark:]
            raise
    return wrap


def _post_verdict(any_bad, describe):
    """Queue a device-side verdict (a 1-element tensor, non-zero = violation) for raise_pending()."""
    if any_bad.is_cuda and torch.cuda.is_current_stream_capturing():
        return          # inside a HIP-graph capture nothing may be read back: the captured forward runs unchecked (like DIR_CHECK_IDS=0)
    if any_bad.is_cuda:
        dev = any_bad.device
        st = _SIDE.get(dev)
        if st is None:
            st = _SIDE[dev] = [torch.cuda.Stream(device=dev), torch.zeros(64, dtype=torch.int32, pin_memory=True), 0]
        ring = st[1]
        if len(_PENDING) >= 48:       # nobody is collecting the verdicts (or the host runs far ahead): free the ring's slots
            raise_pending(block=True)
        host = ring[st[2]:st[2] + 1]
        st[2] = (st[2] + 1) % 64
        # the 4-byte copy rides the CURRENT stream right behind the check kernel (no cross-stream dependency: a side stream's
        # wait/record pairs cost the ESMM forward 0.6 ms); only the host-side read is deferred
        host.copy_(any_bad.reshape(1), non_blocking=True)
        ev = torch.cuda.current_stream(dev).record_event()
        _PENDING.append((host, ev, describe))
    else:
        _PENDING.append((any_bad.reshape(1), None, describe))


def check_onehot_matrix(cats, ids_bf, device):
    """The range check of a stacked one-hot id matrix [B, F] (any strides) in ONE pass of dir_check_ids (13.6 MB read at the
    BASELINE batch: ~5 us; the torch formulation -- stack, two compares, or, any -- cost 0.14 ms of GPU time per call)."""
    if CHECK_MODE == "off":
        return
    idx, nb = _range_checked(cats)
    if not idx:
        return
    from . import _lib, ops
    key = (tuple(idx), tuple(nb), len(cats), str(device))
    bounds = _BOUNDS.get(key)
    if bounds is None:
        b = [torch.iinfo(torch.int64).max] * len(cats)
        for i, n in zip(idx, nb):
            b[i] = n
        if len(_BOUNDS) > 64:
            _BOUNDS.clear()
        bounds = _BOUNDS[key] = torch.tensor(b, dtype=torch.int64, device=device)
    bad = torch.empty(1, dtype=torch.int32, device=device)
    B, F = ids_bf.shape
    _lib.check(_lib.load().dir_check_ids(ops._ptr(bounds), F, ops._ptr(ids_bf), None, ids_bf.stride(0), ids_bf.stride(1), B, ops._ptr(bad),
                                         ops._stream()))

    def describe():
        t = ids_bf.t()
        viol = ((t >= bounds.unsqueeze(1)) | (t < -1)).nonzero()[0]
        c = cats[int(viol[0])]
        return ("categorical_column_with_identity %r: id %d is outside [0, num_buckets=%d) and no default_value is set"
                % (c.key, int(t[int(viol[0]), int(viol[1])]), c.num_buckets))

    _post_verdict(bad, describe)


def check_id_range(cats, got, device):
    """Raise like TF's InvalidArgumentError when an identity column (default_value=None) is fed an id outside [0, num_buckets)
    (-1 = missing is allowed).  One fused comparison for all columns.  On the GPU the verdict is NOT waited for here: it is copied
    to pinned memory on a side stream and read by raise_pending() at the end of the model's forward, after the rest of the forward
    has been enqueued -- the host then waits for a check kernel that ran long ago instead of draining the stream mid-forward
    (the kernels treat such ids as pruned, so running them first is harmless).  DIR_CHECK_IDS=0 skips the check."""
    if CHECK_MODE == "off":
        return
    idx, nb = _range_checked(cats)
    if not idx:
        return
    vals = [got[i][0] if isinstance(got[i], tuple) else got[i] for i in idx]
    sizes = [int(v.numel()) for v in vals]
    nbt = torch.tensor(nb, dtype=torch.int64, device=device)

    def compare():
        if len(set(sizes)) == 1:      # the common case (one-hot columns of one batch): one broadcast comparison, no index expansion
            st = torch.stack([v.reshape(-1) for v in vals])
            return ((st >= nbt.unsqueeze(1)) | (st < -1)).reshape(-1)
        return torch.cat([((v.reshape(-1) >= n) | (v.reshape(-1) < -1)) for v, n in zip(vals, nb)])

    def describe():
        bad = compare()
        pos = int(bad.nonzero()[0])
        k = 0
        while pos >= sizes[k]:
            pos -= sizes[k]
            k += 1
        c = cats[idx[k]]
        return ("categorical_column_with_identity %r: id %d is outside [0, num_buckets=%d) and no default_value is set"
                % (c.key, int(vals[k].reshape(-1)[pos]), c.num_buckets))

    _post_verdict(compare().any().to(torch.int32), describe)


def raise_pending(block=None):
    """Read the verdicts of the id-range checks issued so far; raises ValueError for the first violation.  block=None follows
    DIR_CHECK_IDS: "sync" waits for every verdict; the default reads only those that have already arrived (never stalls)."""
    if block is None:
        block = CHECK_MODE == "sync"
    while _PENDING:
        host, ev, describe = _PENDING[0]
        if ev is not None:
            if block:
                ev.synchronize()
            elif not ev.query():
                return
        _PENDING.pop(0)
        if bool(host[0]):
            _PENDING.clear()
            raise ValueError(describe())


_IDS_ALIAS = os.environ.get("DIR_IDS_ALIAS", "1") != "0"      # development switch: 0 = always stack the id columns (round 4's behaviour)


def _columns_of_one_matrix(got):
    """The one-hot id columns as ONE strided [B, F] view when they already are the columns (or rows) of one int64 matrix -- what an input
    pipeline that batches the categorical features into one tensor hands over as a dict of views --, else None.  The kernels take ids
    with any (stride_b, stride_f): no stacked copy (13.6 MB per step at B = 65 536, F = 26)."""
    t0 = got[0]
    base = t0._base                                        # (views of one tensor share ._base: separate tensors leave here at once)
    if base is None or base.dtype != torch.int64 or t0.dim() != 1:
        return None
    F, B = len(got), t0.numel()
    if B == 0:
        return None
    shape, stride = t0.shape, t0.stride()
    for g in got:
        if g._base is not base or g.shape != shape or (B > 1 and g.stride() != stride):
            return None
    sb = stride[0] if B > 1 else 1
    ptrs = [g.data_ptr() for g in got]
    step = (ptrs[1] - ptrs[0]) // 8 if F > 1 else 1
    if any(ptrs[f] - ptrs[0] != 8 * f * step for f in range(F)):
        return None
    if step <= 0 or sb <= 0:
        return None
    if sb != 1 and not torch.is_grad_enabled():
        # Sample-major ids ([B, F] rows): the training kernels (4-8 lanes per sample) read them as they are -- deepfm_train 1.489 -> 1.465 ms
        # without the stack --, but the fused inference kernels give a LANE a sample, and 64 lanes then read 64 different rows of the id
        # matrix per field: the stack (a transposing copy: field-major ids, coalesced across samples) is worth its 15 us there
        # (deepfm_full 0.2757 vs 0.2782 ms, esmm_full 0.410 vs 0.420 ms aliased; profiles/r05_ab_ids_alias.txt).
        return None
    return torch.as_strided(t0, (B, F), (sb, step))


def collect_ids(columns, features, device, memo=None):
    """columns: categorical columns (or embedding/indicator wrappers).
    -> ("onehot", ids_view [B,F])  or  ("ragged", values, offsets [F*B+1], weights|None, B)
    memo: a dict that lives for ONE forward over one features dict -- a second request for the same categorical columns (ESMM's two
    towers read the same 26 columns into their own tables) gets the same tensors back: one stack, one range check, and the two
    towers' sorted sparse updates see one id tensor."""
    cats = [categorical_of(c) for c in columns]
    if memo is not None:
        key = tuple(id(c) for c in cats)
        if key in memo:
            return memo[key]
        out = collect_ids(columns, features, device)
        memo[key] = out
        return out
    got = [c.ids(features, device) for c in cats]
    if all(not isinstance(g, tuple) for g in got):
        B = got[0].numel()
        for g in got:
            if g.numel() != B:
                raise ValueError("features disagree on the batch size")
        ids_bf = _columns_of_one_matrix(got) if _IDS_ALIAS else None
        if ids_bf is None:
            ids_bf = torch.stack(got, dim=0).t()
        if ids_bf.is_cuda:
            check_onehot_matrix(cats, ids_bf, device)     # one fused pass over the matrix the gather is about to read
        else:
            check_id_range(cats, got, device)
        return ("onehot", ids_bf)
    check_id_range(cats, got, device)
    B = None
    for g in got:
        b = (g[1].numel() - 1) if isinstance(g, tuple) else g.numel()
        if B is None:
            B = b
        elif b != B:
            raise ValueError("features disagree on the batch size")
    vals, offs, wts = [], [], []
    any_w = any(isinstance(g, tuple) and g[2] is not None for g in got)
    base = 0
    for g in got:
        if isinstance(g, tuple):
            v, o, w = g
        else:
            v, o, w = g, torch.arange(B + 1, dtype=torch.int64, device=device), None
        vals.append(v)
        offs.append(o[:-1] + base)
        base += int(v.numel())
        if any_w:
            wts.append(w if w is not None else torch.ones(v.numel(), dtype=torch.float32, device=device))
    offs.append(torch.tensor([base], dtype=torch.int64, device=device))
    return ("ragged", torch.cat(vals), torch.cat(offs), torch.cat(wts) if any_w else None, B)
