"""_input.py -- turn a reference-style `features` dict into the id layout the kernels take.

One-hot columns give a [B] id vector each; stacking them field-major ([F, B], one copy) and passing the
transposed VIEW keeps the C ABI's (stride_b, stride_f) = (1, B) without a second pass.  As soon as one
column is Ragged (multi-hot), every column is expressed as CSR and concatenated field-major
(bag(b, f) = f*B + b), which is exactly what per-column SparseTensors concatenate to.
"""
import torch


def categorical_of(col):
    """The categorical column behind a dense wrapper (embedding / indicator); categorical columns -- including
    weighted_categorical_column, which also carries a `.categorical_column` -- are returned as they are."""
    if getattr(col, "is_dense", False) and hasattr(col, "categorical_column"):
        return col.categorical_column
    return col


def collect_ids(columns, features, device):
    """columns: categorical columns (or embedding/indicator wrappers).
    -> ("onehot", ids_view [B,F])  or  ("ragged", values, offsets [F*B+1], weights|None, B)"""
    got = [categorical_of(c).ids(features, device) for c in columns]
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
