"""autograd.py -- differentiable wrappers of the HIP ops (backward of the interaction path).

The reference trains its Estimators through TensorFlow autodiff with Adagrad on the dnn/fm/embedding variables and
FTRL on the linear ones (models/DeepFM/deepFM.py:58,61,225-245).  Here every forward op keeps running in HIP; its
backward is either a HIP kernel (FM second-order, DCN cross: csrc/backward.hip) or, for the table gradients, the
SPARSE gradient the reference's embedding lookups produce (IndexedSlices): indices = the looked-up ids, values = the
incoming row gradients times the bag coefficient.  The sparse tensors are handed to torch.optim (Adagrad accepts
sparse gradients and, like TensorFlow's _apply_sparse_duplicate_indices, sums duplicate ids before the update).
"""
import os

import torch

from . import ops


def _sparse_rows(ids, rows, vocab):
    """Sparse [vocab, K] (or [vocab]) gradient: entry i adds rows[i] to row ids[i]; pruned ids (< 0) add nothing."""
    ok = ids >= 0
    idx = torch.where(ok, ids, torch.zeros_like(ids))
    if rows.dim() == 2:
        vals = rows * ok.unsqueeze(1).to(rows.dtype)
        size = (vocab, rows.shape[1])
    else:
        vals = rows * ok.to(rows.dtype)
        size = (vocab,)
    return torch.sparse_coo_tensor(idx.reshape(1, -1), vals.contiguous(), size=size)


class GatherFm(torch.autograd.Function):
    """emb, fm = gather_fm(tables, ids) with d/d(tables) as sparse gradients (one-hot slots)."""

    @staticmethod
    def forward(ctx, ts, ids, *tables):
        ctx.ts = ts
        ctx.fused = ts.fm_sink is not None
        if ctx.fused:
            # a fused optimiser that also takes the FM backward (ops.SparseAdagrad.step_fm): keep the [B, K] field sums, not emb
            fsum = torch.empty((ids.shape[0], ts.K), dtype=torch.float32, device=ts.device)
            emb, fm = ops.gather_fm(ts, ids, fsum=fsum, want_bits=ops.GATHER_BITS)      # (+ emb's row maxima, for the first dense layer)
            ctx.save_for_backward(ids, fsum)
        else:
            if ts.ld != ts.K:
                raise ValueError("packed training rows train through a fused optimiser (ops.SparseAdagrad(...).attach())")
            emb, fm = ops.gather_fm(ts, ids)
            ctx.save_for_backward(ids, emb)
        return emb, fm

    @staticmethod
    def backward(ctx, g_emb, g_fm):
        ids, emb = ctx.saved_tensors
        ts = ctx.ts
        F, K = ts.F, ts.K
        if ctx.fused:
            add_in = g_emb.contiguous() if g_emb is not None else None
            if g_fm is not None:
                ts.fm_sink(ids, add_in, g_fm.contiguous(), emb)      # emb slot holds the field sums here
            elif add_in is not None:
                ts.grad_sink(ids, add_in)
            return (None, None) + (None,) * F
        add_in = g_emb.contiguous() if g_emb is not None else None
        if g_fm is not None:
            demb = ops.fm_logit_backward(emb, g_fm.contiguous(), F, K, add_in=add_in)      # HIP: FM + DNN branch, one pass
        else:
            demb = add_in
        if ts.grad_sink is not None:
            # fused optimiser attached (ops.SparseAdagrad): the row gradients go straight to its HIP update kernel;
            # no sparse tensors are built and the tables receive no .grad
            ts.grad_sink(ids, demb)
            return (None, None) + (None,) * F
        grads = []
        for f in range(F):
            grads.append(_sparse_rows(_in_range(ids[:, f].contiguous(), ts.vocab[f]), demb[:, f * K:(f + 1) * K], ts.vocab[f]))
        return (None, None) + tuple(grads)


def _clip_backward(rows, g, max_norm):
    """Gradient through [TF-upstream] clip_by_norm: y = r * m / max(||r||, m).  ||r|| <= m: y = r (dy/dr = I);
    otherwise y = m r / ||r||  ->  dL/dr = m (g / ||r|| - r (r . g) / ||r||^3)."""
    n = rows.norm(dim=1, keepdim=True)
    big = n > max_norm
    nn_ = n.clamp_min(1e-30)
    gc = max_norm * (g / nn_ - rows * ((rows * g).sum(dim=1, keepdim=True) / (nn_ * nn_ * nn_)))
    return torch.where(big, gc, g)


class EmbeddingBag(torch.autograd.Function):
    """One-hot or multi-hot embedding bag (one combiner, or one per slot; optional max_norm) with sparse table gradients."""

    @staticmethod
    def forward(ctx, ts, ids, offsets, weights, combiner, field_major, out, max_norm, *tables):
        res = ops.embedding_bag(ts, ids, offsets, weights, combiner=combiner, field_major=field_major, out=out, max_norm=max_norm,
                                want_bits=ops.GATHER_BITS and offsets is None and out is None)
        ctx.ts, ctx.combiner, ctx.field_major, ctx.max_norm = ts, combiner, field_major, max_norm
        ctx.save_for_backward(ids, offsets if offsets is not None else torch.empty(0), weights if weights is not None else torch.empty(0))
        ctx.has_offsets, ctx.has_weights = offsets is not None, weights is not None
        return res

    @staticmethod
    def backward(ctx, g):
        ids, offsets, weights = ctx.saved_tensors
        ts = ctx.ts
        F, K = ts.F, ts.K
        g = g.contiguous() if g.stride(1) == 1 else g.clone()
        grads = []
        nfix = 8
        mns = list(ctx.max_norm) if isinstance(ctx.max_norm, (list, tuple)) else [ctx.max_norm] * F      # one max_norm per slot
        any_mn = any(m for m in mns)
        if not ctx.has_offsets and ts.grad_sink is not None and not any_mn:   # fused optimiser attached (ops.SparseAdagrad.attach): no table .grad
            ts.grad_sink(ids, g)
            return (None,) * (nfix + F)
        if not ctx.has_offsets:
            for f in range(F):
                idf, gf = ids[:, f].contiguous(), g[:, f * K:(f + 1) * K]
                if mns[f]:
                    gf = _clip_backward(ts.tables[f][idf.clamp(0, ts.vocab[f] - 1)], gf, float(mns[f]))
                grads.append(_sparse_rows(_in_range(idf, ts.vocab[f]), gf.reshape(-1) if ts.tables[f].dim() == 1 else gf, ts.vocab[f]))
        else:
            B = (offsets.numel() - 1) // F
            lens = offsets[1:] - offsets[:-1]
            bag_of = torch.repeat_interleave(torch.arange(B * F, device=ids.device), lens)
            if ctx.field_major:
                f_of, b_of = bag_of // B, bag_of % B
            else:
                b_of, f_of = bag_of // F, bag_of % F
            vocab_e = ts.vocab_dev[f_of]
            w = weights if ctx.has_weights else torch.ones(ids.numel(), dtype=torch.float32, device=ids.device)
            valid = ((ids >= 0) & (ids < vocab_e)).to(torch.float32)
            wv = w * valid
            slot, comb0 = ops.slot_combiners(ts, ctx.combiner)
            den_mean = torch.zeros(B * F, dtype=torch.float32, device=ids.device).index_add_(0, bag_of, wv if ctx.has_weights else valid)
            den_sqrt = torch.zeros(B * F, dtype=torch.float32, device=ids.device).index_add_(0, bag_of, (wv * wv) if ctx.has_weights else valid).sqrt()
            if slot is None:
                den = den_mean if comb0 == ops.MEAN else den_sqrt if comb0 == ops.SQRTN else torch.ones_like(den_mean)
            else:
                fb = (torch.arange(B * F, device=ids.device) // B) if ctx.field_major else (torch.arange(B * F, device=ids.device) % F)
                code = slot.to(torch.int64)[fb]
                den = torch.where(code == ops.MEAN, den_mean, torch.where(code == ops.SQRTN, den_sqrt, torch.ones_like(den_mean)))
            coef = wv / den.clamp_min(1e-30)[bag_of]
            gv = g.view(B, F, K)[b_of, f_of] * coef.unsqueeze(1)
            for f in range(F):
                sel = f_of == f
                idf, gf = ids[sel], gv[sel]
                if mns[f]:
                    gf = _clip_backward(ts.tables[f][idf.clamp(0, ts.vocab[f] - 1)], gf, float(mns[f]))
                grads.append(_sparse_rows(_in_range(idf, ts.vocab[f]), gf.reshape(-1) if ts.tables[f].dim() == 1 else gf, ts.vocab[f]))
        return (None,) * nfix + tuple(grads)


def _in_range(ids, vocab):
    """ids outside [0, vocab) become -1 (pruned): the forward kernels give them no row, so they get no gradient."""
    return torch.where(ids < vocab, ids, torch.full_like(ids, -1))


class LinearLogit(torch.autograd.Function):
    """First-order term (units = 1, one-hot ids, 'sum'): sparse gradients for the weight columns, dense for the bias."""

    @staticmethod
    def forward(ctx, ts, ids, bias, *weights):
        if ts.ld != ts.K and ts.grad_sink is None:
            raise ValueError("packed linear training rows train through a fused optimiser (ops.SparseFtrl(...).attach())")
        out = ops.linear_logit(ts, ids, bias=bias.detach())
        ctx.ts = ts
        ctx.save_for_backward(ids)
        return out

    @staticmethod
    def backward(ctx, g):
        (ids,) = ctx.saved_tensors
        ts = ctx.ts
        gv = g.reshape(-1)
        if ts.grad_sink is not None:     # fused optimiser attached (ops.SparseFtrl): d logit [B, 1] is every slot's gradient row
            ts.grad_sink(ids, gv.contiguous().reshape(-1, 1))
            return (None, None, gv.sum().reshape(1)) + (None,) * ts.F
        grads = [_sparse_rows(_in_range(ids[:, f].contiguous(), ts.vocab[f]), gv, ts.vocab[f]) for f in range(ts.F)]
        return (None, None, gv.sum().reshape(1)) + tuple(grads)


class FmLogit(torch.autograd.Function):
    @staticmethod
    def forward(ctx, emb, F, K):
        ctx.F, ctx.K = F, K
        ctx.save_for_backward(emb)
        return ops.fm_logit(emb, F, K)

    @staticmethod
    def backward(ctx, g):
        (emb,) = ctx.saved_tensors
        return ops.fm_logit_backward(emb, g.contiguous(), ctx.F, ctx.K), None, None


class CrossNetwork(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x0, w, b):
        ctx.save_for_backward(x0, w, b)
        return ops.cross_network(x0, w, b)

    @staticmethod
    def backward(ctx, g):
        x0, w, b = ctx.saved_tensors
        g = g.contiguous() if g.stride(1) == 1 else g.clone()
        gx0, gw, gb = ops.cross_network_backward(x0, w, b, g)
        return gx0, gw, gb


class CinLayer(torch.autograd.Function):
    """One CIN layer with both outputs (xout [B,H,D], pooled [B,H]); backward through ops.cin_layer_backward."""

    @staticmethod
    def forward(ctx, x0, xk, W):
        ctx.save_for_backward(x0, xk, W)
        ctx.set_materialize_grads(False)     # an unused output (the last layer's xout) arrives as None, not as zeros
        xout, pooled = ops.cin_layer(x0, xk, W)
        return xout, pooled

    @staticmethod
    def backward(ctx, g_xout, g_pooled):
        x0, xk, W = ctx.saved_tensors
        B, H = x0.shape[0], W.shape[0]
        D = x0.shape[2]
        if g_xout is None and g_pooled is None:
            return None, None, None
        if g_xout is None:
            G = g_pooled.reshape(B, H, 1).expand(B, H, D).contiguous()
        elif g_pooled is None:
            G = g_xout.contiguous()
        else:
            G = g_xout + g_pooled.reshape(B, H, 1)      # pooled = sum_d xout
        need = ctx.needs_input_grad
        dx0, dxk, dW = ops.cin_layer_backward(x0, xk, W, G, need_x0=need[0], need_xk=need[1], need_w=need[2])
        return dx0, dxk, dW


class CinStack(torch.autograd.Function):
    """A whole CIN stack as ONE autograd node: x0 [B, m, D], W_k [H_k, H_{k-1} * m] -> pooled [B, sum H_k] (xDeepFM's CIN output; the last
    layer's map is not written).  Backward: ops.cin_stack_backward -- dL/dxout of every layer is completed inside the data-gradient kernel
    of the layer above and dx0 is accumulated by the partial-sum pass (per-layer nodes cost one [B, H, D] add and one [B, m, D] add each)."""

    @staticmethod
    def forward(ctx, x0, *Ws):
        B = x0.shape[0]
        Hs = [int(W.shape[0]) for W in Ws]
        pooled = torch.empty((B, sum(Hs)), dtype=torch.float32, device=x0.device)
        xks, xk, off, zl = [], x0, 0, []
        for k, (W, h) in enumerate(zip(Ws, Hs)):
            xks.append(xk)
            xk, _ = ops.cin_layer(x0, xk, W, pooled=pooled[:, off:off + h], want_xout=k + 1 < len(Ws), z_out=zl)
            off += h
        ctx.L = len(Ws)
        ctx.has_z = len(zl) == 1                              # the top layer ran in its pooled form: its Z [B, Hp*m] serves the backward
        ctx.save_for_backward(x0, *xks[1:], *Ws, *zl)
        return pooled

    @staticmethod
    @torch.no_grad()
    def backward(ctx, g):
        saved = ctx.saved_tensors
        L = ctx.L
        x0, xks, Ws = saved[0], [saved[0]] + list(saved[1:L]), list(saved[L:2 * L])
        if g.dim() != 2 or (g.shape[0] > 0 and g.stride(1) != 1):
            g = g.contiguous()
        dx0, dWs = ops.cin_stack_backward(x0, xks, Ws, g, need_x0=ctx.needs_input_grad[0], z_top=saved[2 * L] if ctx.has_z else None)
        return (dx0,) + tuple(dWs)


def cin_stack(x0, Ws):
    """pooled [B, sum H_k] of a CIN stack through CinStack (x0 contiguous float32 [B, m, D] on the GPU)."""
    return CinStack.apply(x0, *Ws)


def cin_layer(x0, xk, W):
    return CinLayer.apply(x0, xk, W)


def _tn_matmul(A, Bm, splits=128):
    """A^T @ Bm for tall-skinny A [N, m], Bm [N, n] (N in the millions, m, n ~ 100): the reduction over N is split into
    `splits` batched GEMMs that are then added (rocBLAS has no good single kernel for this shape: 2.6 ms vs 0.3 ms)."""
    N = A.shape[0]
    per = N // splits
    if per < 64:
        return A.t() @ Bm
    nm = per * splits
    out = torch.bmm(A[:nm].view(splits, per, A.shape[1]).transpose(1, 2), Bm[:nm].view(splits, per, Bm.shape[1])).sum(dim=0)
    if nm < N:
        out = out + A[nm:].t() @ Bm[nm:]
    return out


_DIN_COMPOSITE_BACKWARD = os.environ.get("DIR_DIN_COMPOSITE_BACKWARD", "0") == "1"   # dev switch: A/B against the fused kernel
_DIN_SAVE_ACTIVATIONS = os.environ.get("DIR_DIN_SAVE", "1") != "0"                   # 0: the backward recomputes the two hidden layers (round 2's form)


class DinAttentionPool(torch.autograd.Function):
    """DIN local activation unit + pooling (include/dir_hip.h A13).  Forward: the fused HIP kernel.  Backward: the fused HIP
    backward (ops.din_attention_pool_backward) for the shape class it covers (K = 64, H1 <= 80, H2 <= 48, T <= 64); other
    shapes take the hand-derived GPU composite below: the unit recomputed over the VALID (sample, position) rows in the
    regrouped form the forward kernel uses ([h, a, h-a, h*a].W1 = h.(Wh+Wd) + (h*a).Wp + a.(Wa-Wd)), six GEMMs through rocBLAS,
    the elementwise steps and per-sample segment sums as torch kernels.  Either way the table gets a sparse gradient (history
    rows and candidate rows) and no autograd graph is built."""

    @staticmethod
    def forward(ctx, table, hist, hist_len, cand, W1, b1, W2, b2, W3, b3, normalize):
        ctx.normalize = bool(normalize)
        B, T = hist.shape
        fused = ops.din_backward_supported(table.shape[1], T, W1.shape[1], W2.shape[1]) and not _DIN_COMPOSITE_BACKWARD
        ctx.saved_state = None
        if fused and B > 0 and _DIN_SAVE_ACTIVATIONS:
            # the forward leaves z1 / z2 of every history row in a workspace this node owns: the backward recomputes nothing
            out, scores, ctx.saved_state = ops.din_attention_pool_save(table.detach(), hist, hist_len, cand, W1.detach(), b1.detach(),
                                                                       W2.detach(), b2.detach(), W3.detach(), b3.detach(), normalize=normalize,
                                                                       range_of=(table, W1, W2, W3))
        else:
            res = ops.din_attention_pool(table.detach(), hist, hist_len, cand, W1.detach(), b1.detach(), W2.detach(),
                                         b2.detach(), W3.detach(), b3.detach(), normalize=normalize, want_scores=fused,
                                         range_of=(table, W1, W2, W3))
            out, scores = res if fused else (res, None)   # the attention weights [B, T] feed the backward (no softmax recompute)
        ctx.has_scores = scores is not None
        ctx.save_for_backward(table, hist, hist_len, cand, W1, b1, W2, b2, W3, b3, *([scores] if scores is not None else []))
        return out

    @staticmethod
    @torch.no_grad()
    def backward(ctx, g):
        table, hist, hist_len, cand, W1, b1, W2, b2, W3, b3 = ctx.saved_tensors[:10]
        scores = ctx.saved_tensors[10] if ctx.has_scores else None
        B, T = hist.shape
        K, H1 = table.shape[1], W1.shape[1]
        dev = table.device
        g = g.contiguous()
        if ops.din_backward_supported(K, T, H1, W2.shape[1]) and not _DIN_COMPOSITE_BACKWARD:
            r = ops.din_attention_pool_backward(table, hist, hist_len, cand, W1, b1, W2, b2, W3, b3, g, normalize=ctx.normalize,
                                                scores=scores, saved=ctx.saved_state)
            ctx.saved_state = None                                  # the workspace (1.8 GB at config 4) goes back to the allocator
            gtab = None
            if ctx.needs_input_grad[0]:
                ok = cand >= 0                                      # a pruned candidate (zero vector in the forward) adds nothing
                idx = torch.cat([r["ids_h"], torch.where(ok, cand, torch.zeros_like(cand))]).unsqueeze(0)
                r["ga"].mul_(ok.unsqueeze(1))                       # (a view of r["grows"]: no [N + B, K] copy for the concatenation, 428 MB at cfg 4)
                gtab = torch.sparse_coo_tensor(idx, r["grows"], table.shape)
            return (gtab, None, None, None, r["gW1"], r["gb1"], r["gW2"], r["gb2"], r["gW3"].reshape(W3.shape),
                    r["gb3"].reshape(b3.shape), None)
        valid = hist >= 0
        if hist_len is not None:
            valid &= torch.arange(T, device=dev).unsqueeze(0) < hist_len.clamp(0, T).unsqueeze(1)
        b_idx, j_idx = valid.nonzero(as_tuple=True)                 # rows in (b, j) order
        ids_h = hist[b_idx, j_idx]
        w3 = W3.reshape(-1)
        Wh, Wa, Wd, Wp = W1[:K], W1[K:2 * K], W1[2 * K:3 * K], W1[3 * K:]
        A, C = Wh + Wd, Wa - Wd
        AP = torch.cat([A, Wp], dim=0)                              # [2K, H1]
        # ---- recompute the unit on the valid rows -------------------------------------------------------------------
        h = table[ids_h]                                            # [N, K]
        cok = cand >= 0                                             # a pruned candidate is the zero vector in the forward
        a = table[cand.clamp(min=0)] * cok.unsqueeze(1)             # [B, K]
        ab = a[b_idx]
        X = torch.cat([h, h * ab], dim=1)                           # [N, 2K]
        z1 = torch.sigmoid_(torch.addmm((a @ C + b1)[b_idx], X, AP))
        z2 = torch.sigmoid_(torch.addmm(b2, z1, W2))
        s = (z2 * w3.unsqueeze(0)).sum(dim=1) + b3
        gb = g[b_idx]
        if ctx.normalize:                                           # softmax over the valid positions of each sample
            x = s * (1.0 / K ** 0.5)
            mx = torch.full((B,), float("-inf"), device=dev).scatter_reduce(0, b_idx, x, "amax")
            e = torch.exp_(x - mx[b_idx])
            w = e / torch.zeros(B, device=dev).index_add_(0, b_idx, e)[b_idx]
        else:
            w = s
        # ---- backward -------------------------------------------------------------------------------------------------
        dw = (gb * h).sum(dim=1)                                    # d out / d w_j . g
        if ctx.normalize:
            t = torch.zeros(B, device=dev).index_add_(0, b_idx, w * dw)
            ds = w * (dw - t[b_idx]) * (1.0 / K ** 0.5)
        else:
            ds = dw
        dpre2 = (ds.unsqueeze(1) * w3.unsqueeze(0)) * z2 * (1.0 - z2)
        gW3 = (z2 * ds.unsqueeze(1)).sum(dim=0)                    # (a [H2 x N] gemv runs at 15 ms in rocBLAS)
        gW2 = _tn_matmul(z1, dpre2)
        dpre1 = (dpre2 @ W2.t()) * z1 * (1.0 - z1)
        gWx = _tn_matmul(X, dpre1)                                  # [2K, H1]: d(Wh+Wd), dWp
        S = torch.zeros((B, H1), device=dev).index_add_(0, b_idx, dpre1)
        gC = a.t() @ S                                              # d(Wa-Wd)
        gA, gWp = gWx[:K], gWx[K:]
        gW1 = torch.cat([gA, gC, gA - gC, gWp], dim=0)              # dWh, dWa, dWd = dWh - dWa, dWp
        dX = dpre1 @ AP.t()                                         # [N, 2K]
        gh = dX[:, :K] + dX[:, K:] * ab + w.unsqueeze(1) * gb
        ga = torch.zeros((B, K), device=dev).index_add_(0, b_idx, dX[:, K:] * h) + S @ C.t()
        gtab = None
        if ctx.needs_input_grad[0]:
            idx = torch.cat([ids_h, cand.clamp(min=0)]).unsqueeze(0)
            gtab = torch.sparse_coo_tensor(idx, torch.cat([gh, ga * cok.unsqueeze(1)]), table.shape)
        return (gtab, None, None, None, gW1, dpre1.sum(dim=0), gW2, dpre2.sum(dim=0), gW3.reshape(W3.shape),
                ds.sum().reshape(b3.shape), None)


def din_attention_pool(table, hist, hist_len, cand, W1, b1, W2, b2, W3, b3, normalize=False):
    return DinAttentionPool.apply(table, hist, hist_len, cand, W1, b1, W2, b2, W3, b3, normalize)


# ---- PReLU / Dice layers and the DIN unit's row-list training path (round 5; csrc/din_rows_train.hip) -----------------------------------------
_DICE_FUSED_BWD = os.environ.get("DIR_DICE_FUSED_BWD", "1") != "0"      # development switch: 0 keeps the three-kernel backward of training-mode Dice


class ActRows(torch.autograd.Function):
    """y = PReLU / Dice (s) over rows s [M, N] with HIP forward and backward (no reference code: arXiv:1706.06978 section 5.3, din.Dice).
    Dice in TRAIN mode: the batch statistics come from dir_bn_train_stats_f32 (which also advances the module's moving statistics with
    its momentum), the backward is the direct term + the batch-norm backward of the gradient with respect to the normalised
    pre-activation (dir_bn_train_backward_f32) + dL/dalpha; in eval mode (grad enabled, module.eval()) the moving statistics are
    constants.  PReLU: one pass each way."""

    @staticmethod
    def forward(ctx, s, alpha, kind, stats):
        # stats: None (PReLU) | ("batch", moving_mean, moving_variance, eps, momentum) | ("moving", scale, shift)
        ctx.kind = kind
        mean = inv = scale = shift = None
        if kind == "dice":
            if stats[0] == "batch":
                mean, inv, scale, shift = ops.bn_train_stats(s, None, None, stats[1], stats[2], stats[3], stats[4])
            else:
                scale, shift = stats[1], stats[2]
        y = ops.act_rows_train(s, kind, alpha, scale, shift)
        ctx.batch = kind == "dice" and stats[0] == "batch"
        ctx.save_for_backward(s, alpha, *([scale, shift] if kind == "dice" else []), *([mean, inv] if ctx.batch else []))
        return y

    @staticmethod
    @torch.no_grad()
    def backward(ctx, g):
        saved = ctx.saved_tensors
        s, alpha = saved[0], saved[1]
        scale, shift = (saved[2], saved[3]) if ctx.kind == "dice" else (None, None)
        if ctx.batch and _DICE_FUSED_BWD and ops.bn_train_supported(s):      # two passes over (g, s) instead of three kernels and an add
            d1, galpha = ops.dice_train_backward(g, s, alpha, scale, shift, saved[4], saved[5])
            return d1, galpha.reshape(alpha.shape) if ctx.needs_input_grad[1] else None, None, None
        d1, gx, galpha = ops.act_rows_backward(g, s, ctx.kind, alpha, scale, shift)
        if ctx.batch:
            mean, inv = saved[4], saved[5]
            dbn, _, _ = ops.bn_train_backward(gx, s, mean, inv, gamma=None)      # the statistics' share of dL/ds
            d1.add_(dbn)
        elif ctx.kind == "dice":
            # moving statistics (module.eval() with grad enabled): x_hat = scale * s + shift still depends on s, so its share
            # gx * scale joins dL/ds (only d1 used to be returned -- frozen-statistics fine-tuning got wrong gradients)
            d1.addcmul_(gx, scale.reshape(1, -1))
        return d1, galpha.reshape(alpha.shape) if ctx.needs_input_grad[1] else None, None, None


def act_rows(s, module):
    """module: din.Dice / din._PReLU.  The HIP training form of module(s) for a [M, N] activation it covers (ops.act_rows_supported)."""
    kind = "dice" if hasattr(module, "moving_mean") else "prelu"
    if kind == "prelu":
        return ActRows.apply(s, module.alpha, "prelu", None)
    if module.training:
        return ActRows.apply(s, module.alpha, "dice", ("batch", module.moving_mean, module.moving_variance, module.eps, module.momentum))
    sc, sh = module.scale_shift()
    return ActRows.apply(s, module.alpha, "dice", ("moving", sc, sh))


class DinFeatRows(torch.autograd.Function):
    """(X [N, 3K], Hc [N, K]) of the valid (sample, position) rows; backward: a sparse gradient for the item table (history rows, then candidates)."""

    @staticmethod
    def forward(ctx, table, ids_h, b_idx, row_off, cand):
        X, Hc = ops.din_feat_rows(table.detach(), ids_h, b_idx, cand)
        ctx.save_for_backward(table, ids_h, row_off, cand)
        return X, Hc

    @staticmethod
    @torch.no_grad()
    def backward(ctx, dX, dH):
        table, ids_h, row_off, cand = ctx.saved_tensors
        N, K = ids_h.numel(), table.shape[1]
        if dX is None:
            dX = torch.zeros((N, 3 * K), dtype=torch.float32, device=table.device)
        if dH is None:
            dH = torch.zeros((N, K), dtype=torch.float32, device=table.device)
        grows = ops.din_feat_rows_backward(table, ids_h, row_off, cand, dX, dH)
        idx = torch.cat([ids_h, cand.clamp(min=0)]).unsqueeze(0)
        return torch.sparse_coo_tensor(idx, grows, table.shape), None, None, None, None


class DinPoolRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scores, Hc, row_off, B, normalize):
        out, w = ops.din_pool_rows(scores.detach(), Hc.detach(), row_off, B, normalize)
        ctx.normalize = bool(normalize)
        ctx.save_for_backward(Hc, w, row_off)
        return out

    @staticmethod
    @torch.no_grad()
    def backward(ctx, g):
        Hc, w, row_off = ctx.saved_tensors
        ds, dH = ops.din_pool_rows_backward(g, Hc, w, row_off, ctx.normalize)
        return ds, dH, None, None, None


def gather_fm(ts, ids, tables):
    return GatherFm.apply(ts, ids, *tables)


def embedding_bag(ts, ids, tables, offsets=None, weights=None, combiner="mean", field_major=False, out=None, max_norm=None):
    return EmbeddingBag.apply(ts, ids, offsets, weights, combiner, field_major, out, max_norm, *tables)


def linear_logit(ts, ids, bias, weights):
    return LinearLogit.apply(ts, ids, bias, *weights)


def fm_logit(emb, F, K):
    return FmLogit.apply(emb, F, K)


def cross_network(x0, w, b):
    return CrossNetwork.apply(x0, w, b)


_DENSE_OPT_HIP = os.environ.get("DIR_DENSE_OPT_HIP", "1") != "0"      # development switch: 0 keeps the library's dense Adagrad / torch-op FTRL steps


class Adagrad(torch.optim.Adagrad):
    """torch.optim.Adagrad ([TF-upstream] tf.train.AdagradOptimizer's rule with eps = 0; the reference's dnn_optimizer='Adagrad',
    deepFM.py:61) whose step on dense float32 CUDA variables is ONE elementwise HIP pass per variable (dir_adagrad_dense_f32) instead of
    the library's five multi-tensor passes (0.16-0.27 ms of a 2-3 ms training step spent on a few hundred thousand weights).  Sparse
    gradients, lr_decay, weight_decay, maximize and non-CUDA variables take the library's step."""

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        rest = []
        for group in self.param_groups:
            plain = not group["lr_decay"] and not group["weight_decay"] and not group.get("maximize", False)
            ws, sums, gs = [], [], []
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if not (_DENSE_OPT_HIP and plain and not g.is_sparse and p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and p.is_contiguous()
                        and g.is_contiguous()):
                    rest.append(p)
                    continue
                st = self.state[p]
                st["step"] += 1
                ws.append(p)
                sums.append(st["sum"])
                gs.append(g)
            if ws:                                 # all of the group's dense variables in one launch per 16 (dir_adagrad_dense_multi_f32)
                ops.adagrad_dense_multi_(ws, sums, gs, group["lr"], group["eps"])
        if rest:                                   # the library's step for everything else: hide the gradients already applied
            held = [(p, p.grad) for group in self.param_groups for p in group["params"] if p.grad is not None and all(p is not r for r in rest)]
            for p, _ in held:
                p.grad = None
            try:
                super().step()
            finally:
                for p, g in held:
                    p.grad = g
        return loss


class Ftrl(torch.optim.Optimizer):
    """FTRL-Proximal with [TF-upstream] tf.train.FtrlOptimizer's defaults and update rule (the reference's
    linear_optimizer='Ftrl', deepFM.py:58): learning_rate_power = -0.5, initial_accumulator_value = 0.1,
    l1 = l2 = 0.  Dense and sparse gradients (sparse ones are coalesced first, as TensorFlow sums duplicate ids)."""

    def __init__(self, params, lr=0.2, initial_accumulator_value=0.1, l1=0.0, l2=0.0):
        super().__init__(params, dict(lr=lr, init=initial_accumulator_value, l1=l1, l2=l2))

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            lr, l1, l2 = group["lr"], group["l1"], group["l2"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["accum"] = torch.full_like(p, group["init"])
                    st["linear"] = torch.zeros_like(p)
                g = p.grad
                if (_DENSE_OPT_HIP and not g.is_sparse and p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and p.is_contiguous()
                        and g.is_contiguous()):
                    ops.ftrl_dense_(p, st["accum"], st["linear"], g, lr, l1, l2)          # one elementwise HIP pass (dir_ftrl_dense_f32)
                    continue
                if g.is_sparse:
                    g = g.coalesce()
                    idx = g.indices()[0]
                    gv = g.values()
                    n, z, w = st["accum"][idx], st["linear"][idx], p[idx]
                else:
                    idx, gv = None, g
                    n, z, w = st["accum"], st["linear"], p
                n_new = n + gv * gv
                sigma = (n_new.sqrt() - n.sqrt()) / lr
                z_new = z + gv - sigma * w
                quad = n_new.sqrt() / lr + 2 * l2
                w_new = torch.where(z_new.abs() > l1, (torch.sign(z_new) * l1 - z_new) / quad, torch.zeros_like(z_new))
                if idx is None:
                    st["accum"].copy_(n_new); st["linear"].copy_(z_new); p.copy_(w_new)
                else:
                    st["accum"][idx] = n_new; st["linear"][idx] = z_new; p[idx] = w_new
