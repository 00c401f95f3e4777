"""_input.py -- turn a reference-style `features` dict into the id layout the kernels take.

One-hot columns give a [B] id vector each; stacking them field-major ([F, B], one copy) and passing the
transposed VIEW keeps the C ABI's (stride_b, stride_f) = (1, B) without a second pass.  As soon as one
column is Ragged (multi-hot), every column is expressed as CSR and concatenated field-major
(bag(b, f) = f*B + b), which is exactly what per-column SparseTensors concatenate to.
"""
import os

import torch

CHECK_IDS = os.environ.get("DIR_CHECK_IDS", "1") != "0"


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


def check_id_range(cats, got, device):
    """Raise like TF's InvalidArgumentError when an identity column (default_value=None) is fed an id outside [0, num_buckets)
    (-1 = missing is allowed).  One fused comparison and ONE host read for all columns; DIR_CHECK_IDS=0 skips it (the kernels
    then treat such ids as pruned: they never read or write outside a table)."""
    if not CHECK_IDS:
        return
    idx, nb = _range_checked(cats)
    if not idx:
        return
    vals = [got[i][0] if isinstance(got[i], tuple) else got[i] for i in idx]
    sizes = [int(v.numel()) for v in vals]
    nbt = torch.tensor(nb, dtype=torch.int64, device=device)
    if len(set(sizes)) == 1:          # the common case (one-hot columns of one batch): one broadcast comparison, no index expansion
        st = torch.stack([v.reshape(-1) for v in vals])
        bad = (st >= nbt.unsqueeze(1)) | (st < -1)
    else:
        st = None
        bad = torch.cat([((v.reshape(-1) >= n) | (v.reshape(-1) < -1)) for v, n in zip(vals, nb)])
    if bool(bad.any()):
        pos = int(bad.reshape(-1).nonzero()[0])
        k = 0
        while pos >= sizes[k]:
            pos -= sizes[k]
            k += 1
        c = cats[idx[k]]
        raise ValueError("categorical_column_with_identity %r: id %d is outside [0, num_buckets=%d) and no default_value is set"
                         % (c.key, int(vals[k].reshape(-1)[pos]), c.num_buckets))


def collect_ids(columns, features, device):
    """columns: categorical columns (or embedding/indicator wrappers).
    -> ("onehot", ids_view [B,F])  or  ("ragged", values, offsets [F*B+1], weights|None, B)"""
    cats = [categorical_of(c) for c in columns]
    got = [c.ids(features, device) for c in cats]
    check_id_range(cats, got, device)
    if all(not isinstance(g, tuple) for g in got):
        B = got[0].numel()
        for g in got:
            if g.numel() != B:
                raise ValueError("features disagree on the batch size")
        return ("onehot", torch.stack(got, dim=0).t())
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
