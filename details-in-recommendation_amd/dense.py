"""dense.py -- the hidden layers of the DNN towers on the HIP dense kernel (include/dir_hip.h: dir_dense_f32).

Mirrors tf.layers.dense(units, activation) as the reference uses it (dnn_logit_fn, models/DeepFM/deepFM.py:295-300;
_deep_architecture, models/DeepCrossNetwork/DeepCrossNetwork.py:394-399; _base_model, models/ESMM/ESMM.py:139-142): matmul +
bias + activation, here in ONE pass (bias and ReLU on the MFMA accumulators).  `dense_act(lin, x, activation)` is what the
models call: the kernel when it covers the layer (CUDA fp32, >= 16 units, activation ReLU or none, at least MIN_ROWS rows; an
in_features that is not a multiple of 4 is zero-padded), `activation(lin(x))` otherwise (the units = 1 logit layers are
matrix-vector products: library forward, elementwise backward).

Weight rows are handed to the kernel with a stride that is a multiple of 64 floats: a 416-float stride (1 664 bytes, the first
DeepFM / DCN layer) costs the kernel -- and rocBLAS -- 20 % (tools/dense_sweep.py); the padded copy is 640 KB per layer.
"""
import torch
import torch.nn.functional as F

from . import ops

_RELUS = (torch.relu, F.relu, torch.nn.functional.relu)
# Below this many rows a layer stays on the library: a 128-row tile per workgroup leaves most CUs idle on small batches, where
# the library's small-M kernels win (DeepFM forward as a HIP-graph replay: 78 vs 126 us at batch 256, 127 vs 135 us at 4 096, then
# 221 vs 177 us at 8 192 -- `DIR_BENCH_SMALL_BATCH=n bench.py --workload small_batch`); DIR_DENSE_MIN_ROWS overrides.
# Round 5: batches ops.dense_small_covers() accepts (up to 256 rows always -- the reference's own 100 / 256 -- and up to 512 for the narrower
# layers) run dir_dense_small_f32 instead of the library.  Round 6: what lies between that and MIN_ROWS runs dir_dense_mid_f32
# (ops.dense_mid_covers): no batch size sends a covered layer to the library any more (DIR_DENSE_MID_ROWS=0 restores round 5's routing).
import os as _os
MIN_ROWS = int(_os.environ.get("DIR_DENSE_MIN_ROWS", "6144"))
_PACK_CACHE = {}
# How the hidden layers of the models were routed since the last reset: "hip" (dir_dense_* kernels) or "library" (rocBLAS / hipBLASLt
# through nn.Linear: batches under MIN_ROWS, uncovered activations, < 16 units), keyed by (in, out): bench.py puts it in the line.
ROUTING = {"hip": {}, "library": {}}


def _route(kind, lin):
    d = ROUTING[kind]
    key = "%dx%d" % (lin.in_features, lin.out_features)
    d[key] = d.get(key, 0) + 1


def reset_routing():
    ROUTING["hip"].clear()
    ROUTING["library"].clear()


def pack_weight(weight):
    """[N, Kd] -> the same values with row stride round_up(Kd, 64) (a view [:, :Kd] of a padded buffer); no copy when Kd % 64 == 0."""
    N, Kd = weight.shape
    ld = (Kd + 63) // 64 * 64
    w = weight.detach()
    if ld == Kd and w.is_contiguous():
        return w
    buf = torch.empty((N, ld), dtype=w.dtype, device=w.device)      # the padding columns are never read (the kernel stops at Kd)
    buf[:, :Kd].copy_(w)
    return buf[:, :Kd]


STRIDED_WEIGHTS = _os.environ.get("DIR_DENSE_STRIDED_W", "1") != "0"      # development switch: 0 always makes the padded / transposed copy


def _kernel_weight(x, weight):
    """The weight argument for ops.dense / ops.dense_gated in the training paths: the tensor (or `.t()` view) itself when the bf16x3 kernel
    will run -- its image is packed straight from any strides, a [400, 400] transpose copy per layer and step is 12 us -- otherwise the
    64-float-stride copy the fp32 kernel wants."""
    w = weight.detach()
    if STRIDED_WEIGHTS and x.is_cuda and x.dtype == torch.float32 and w.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and ops.dense_runs_bf16x3(x, w):
        return w
    return pack_weight(weight)


def _packed_cached(weight):
    """Inference: one padded copy per weight tensor, refreshed when the parameter is modified in place (tensor._version)."""
    key = id(weight)
    hit = _PACK_CACHE.get(key)
    capturing = weight.is_cuda and torch.cuda.is_current_stream_capturing()      # under capture: pack inside the graph, remember nothing (ops.CapturedStep)
    if hit is not None and hit[0] is weight and hit[1] == weight._version and hit[2] == weight.data_ptr() and not ops.capture_bypasses_caches(weight):
        return hit[3]
    packed = pack_weight(weight)
    if capturing:
        return packed
    if len(_PACK_CACHE) > 256:
        _PACK_CACHE.clear()
    _PACK_CACHE[key] = (weight, weight._version, weight.data_ptr(), packed)
    return packed


# dense_dw_auto_arith's answers that are HIP kernels (narrow layers -- an 80-wide last tower layer, ESMM.py:130-147 -- went to the library's
# batched GEMM + a sum before: 75 us against 47 / 36 on the bf16 x 3 / fp16 x 2 kernels, profiles/r04_small_dw_probe.txt)
_DW_KERNELS = ("bf16x3", "small")


def _tn_matmul(g, x, splits=16, g_bits=None, x_bits=None):
    """g^T x for tall operands ([M, N]^T [M, Kd], M = batch rows): the library's single TN GEMM runs at 0.33 of the fp32 MFMA peak
    at 65 536 x 400 x 416 (a 400 x 416 output leaves most CUs idle); sixteen batched row slices + one sum run at 0.53
    (tools/tn_gemm_probe.py: 425 -> 261 us)."""
    M = g.shape[0]
    if g.is_cuda and g.dtype == torch.float32 and x.dtype == torch.float32 and g.stride(1) == 1 and x.stride(1) == 1 \
            and ops.dense_dw_auto_arith(M, g.shape[1], x.shape[1]) in _DW_KERNELS:
        return ops.dense_dw(g, x, g_bits=g_bits, x_bits=x_bits)        # dir_dense_dw_{bf16x3,f16x2}_f32 where the operands are covered (tools/dense_dw_probe.py)
    if M >= 8192 and M % splits == 0 and g.is_contiguous() and x.is_contiguous():
        return torch.bmm(g.view(splits, M // splits, -1).transpose(1, 2), x.view(splits, M // splits, -1)).sum(dim=0)
    return g.t() @ x


def _packed_cached_padded(weight, pad):
    """Inference: the weight with `pad` zero input columns appended, packed; cached like _packed_cached."""
    key = (id(weight), pad)
    hit = _PACK_CACHE.get(key)
    capturing = weight.is_cuda and torch.cuda.is_current_stream_capturing()
    if hit is not None and hit[0] is weight and hit[1] == weight._version and hit[2] == weight.data_ptr() and not ops.capture_bypasses_caches(weight):
        return hit[3]
    packed = pack_weight(F.pad(weight.detach(), (0, pad)))
    if capturing:
        return packed
    if len(_PACK_CACHE) > 256:
        _PACK_CACHE.clear()
    _PACK_CACHE[key] = (weight, weight._version, weight.data_ptr(), packed)
    return packed


def _wb_grads(g, x, need_w, need_b, g_bits=None, x_bits=None):
    """(dL/dW, dL/db) of a dense layer from g = dL/d(pre-activation) and its input x: one pass of dir_dense_dw_bf16x3_f32 for both where
    it covers the shape, otherwise the library GEMM and a column sum."""
    if need_w and need_b and g.is_cuda and g.dtype == torch.float32 and x.dtype == torch.float32 and g.stride(1) == 1 and x.stride(1) == 1 \
            and ops.dense_dw_auto_arith(g.shape[0], g.shape[1], x.shape[1]) in _DW_KERNELS:
        gw, gb = ops.dense_dw(g, x, want_bias=True, g_bits=g_bits, x_bits=x_bits)
        if gb is not None:
            return gw, gb
    return (_tn_matmul(g, x, g_bits=g_bits, x_bits=x_bits) if need_w else None), (g.sum(dim=0) if need_b else None)


def _gbits(g, x_bounded, Kin, need_x=True, need_w=True, carried=None):
    """(row_bits, g_bits) for the fp16 x 2 backward kernels of one layer with Kin inputs: row_bits for dL/dx = g W (its other operand is the
    weight), g_bits for dL/dW = g^T x only when x_bounded -- since round 5: the forward left the bit pattern of max |x| for this layer's
    input (the weight gradient scales x by a power of two as well: dir_dense_dw_f16x2_scaled_f32), no longer a caller's promise -- and
    None otherwise (bf16 x 3).  (None, None) when neither product would run a split-arithmetic kernel at this shape (then the max pass
    over g is not run either)."""
    if not g.is_cuda or g.dim() != 2:
        return None, None
    M, N = g.shape
    rows = need_x and Kin % 4 == 0 and ops.dense_auto_arith(M, N, Kin) == "bf16x3"
    allb = need_w and x_bounded and ops.dense_dw_auto_arith(M, N, Kin) == "bf16x3"
    if not (rows or allb):
        return None, None
    gb = carried if carried is not None else ops.grad_bits(g, want_all=allb)      # carried: left by the kernel that produced g
    if gb is None:
        return None, None
    return (gb[0] if rows else None), (gb[1] if allb else None)


class _DenseFn(torch.autograd.Function):
    """y = act(x W^T + b) with the HIP kernel forward and for dL/dx (= g W, the same kernel on W^T); dL/dW = g^T x and
    dL/db = sum g go through the library (a reduction over the batch rows: a different shape class)."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu, bounded=False):
        xb = []
        y = ops.dense(x, _kernel_weight(x, weight), bias, relu=relu, xbits_out=xb)
        ctx.relu = relu
        ctx.xbits = xb[0] if xb else None           # max |x| (device): what lets dL/dW run the scaled fp16 x 2 kernel (`bounded` is a hint no more)
        ctx.save_for_backward(x, weight, y if relu else None)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @torch.no_grad()
    def backward(ctx, g):
        x, weight, y = ctx.saved_tensors
        if ctx.relu:
            g = g * (y > 0)
        g = g.contiguous()
        gx = gw = gb = None
        rb, ab = _gbits(g, ctx.xbits is not None, x.shape[1], ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        if ctx.needs_input_grad[0]:
            wt = _kernel_weight(g, weight.t())                     # [Kd, N]: the "weight" of the transposed product
            gx = ops.dense(g, wt, None, relu=False, row_bits=rb) if ops.dense_supported(g, wt) else g @ weight
        gw, gb = _wb_grads(g, x, ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2], g_bits=ab, x_bits=ctx.xbits if ab is not None else None)
        return gx, gw, gb, None, None


BN_TRAIN_FUSED = _os.environ.get("DIR_BN_TRAIN_FUSED", "1") != "0"      # development switch: 0 keeps the torch formulation of the training batch norm


class _DenseBnFn(torch.autograd.Function):
    """Hidden layer + training-mode batch norm as ONE autograd node: out = BN_batch(act(x W^T + b)) (deepFM.py:295-308 without dropout,
    DeepCrossNetwork.py:394-403).  Forward: the dense kernel, one read of its output for the batch statistics (dir_bn_train_stats_f32,
    which also advances the moving statistics) and one elementwise pass.  Backward: dir_bn_train_backward_f32 returns dL/d(pre-activation)
    straight through the batch norm AND the ReLU gate (two reads of (g, y), one write -- as torch ops: two column sums, five elementwise
    passes, a compare and a mask multiply), then the layer's weight / bias / data gradients as in _DenseFn."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, bn, relu, bounded=False):
        xb = []
        y = ops.dense(x, _kernel_weight(x, weight), bias, relu=relu, xbits_out=xb)
        mean, inv, scale, shift = ops.bn_train_stats(y, gamma, beta, bn.moving_mean, bn.moving_variance, bn.eps, bn.momentum)
        ctx.relu = relu
        ctx.xbits = xb[0] if xb else None
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, weight, y, mean, inv, gamma)
        return torch.addcmul(shift, y, scale)

    @staticmethod
    @torch.no_grad()
    def backward(ctx, g):
        x, weight, y, mean, inv, gamma = ctx.saved_tensors
        if g.stride(1) != 1 or g.stride(0) % 4 or g.data_ptr() % 16:
            g = g.contiguous()
        gpre, gbeta, ggamma = ops.bn_train_backward(g, y, mean, inv, gamma, relu_gate=ctx.relu)
        gx = None
        rb, ab = _gbits(gpre, ctx.xbits is not None, x.shape[1], ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        if ctx.needs_input_grad[0]:
            wt = _kernel_weight(gpre, weight.t())
            gx = ops.dense(gpre, wt, None, relu=False, row_bits=rb) if ops.dense_supported(gpre, wt) else gpre @ weight
        gw, gb = _wb_grads(gpre, x, ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2], g_bits=ab, x_bits=ctx.xbits if ab is not None else None)
        return (gx, gw, gb, ggamma if gamma is not None and ctx.needs_input_grad[3] else None, gbeta if ctx.needs_input_grad[4] else None,
                None, None, None)


def _dense_bn_train(bn, x, weight, bias, relu, bounded=False):
    """The training forms of a covered hidden layer: with a batch norm in TRAIN mode the fused node, otherwise _DenseFn + the module.
    bounded: the caller vouches that x is bounded by construction (the weight gradient may then take the fp16 x 2 kernel)."""
    if (bn is not None and BN_TRAIN_FUSED and bn.training and torch.is_grad_enabled() and weight.shape[0] % 4 == 0 and weight.shape[0] <= 4096
            and x.shape[0] > 0):
        return _DenseBnFn.apply(x, weight, bias, bn.gamma, bn.beta, bn, relu, bounded)
    return _apply_bn(bn, _DenseFn.apply(x, weight, bias, relu, bounded))


def _forward_layers(ys, h, params, L, bounded, last_bits):
    """The forward of a dense + ReLU stack on the row-scaled fp16 x 2 kernel where it is covered: layer 0 runs one max pass over its input,
    every layer's epilogue leaves the row / tensor maxima of its OUTPUT -- the next layer's row scales and, for the backward, the bound its
    weight gradient scales that input by.  Appends the activations to ys; -> [max |input of layer l| as a [1] int32 device tensor or None]."""
    xbits, rb, carried_all = [], None, None
    for l in range(L):
        left = [] if (l + 1 < L or last_bits) else None
        xb = []
        h = ops.dense(h, _kernel_weight(h, params[2 * l]), params[2 * l + 1], relu=True, arith="auto_bounded" if bounded else None,
                      row_bits=rb, bits_out=left, xbits_out=xb)
        xbits.append(xb[0] if xb else carried_all)
        rb, carried_all = left[0] if left else (None, None)
        ys.append(h)
    return xbits


class _MlpStackFn(torch.autograd.Function):
    """A stack of dense + ReLU layers as ONE autograd node.  Forward: dir_dense_f32 per layer.  Backward, per layer from the top:
    dL/dW = g^T x (batched library GEMM), dL/db = sum g, and the data gradient goes straight through the previous layer's ReLU in
    the kernel's epilogue (dir_dense_gated_f32) -- the separate `g * (y > 0)` pass over [B, units] survives only for the top
    layer."""

    @staticmethod
    def forward(ctx, x, *params):
        bounded = bool(_FWD_BOUNDED[0])         # (set by mlp_stack around the apply: a flag, not a tensor argument)
        L = len(params) // 2
        ys, h = [], x
        ctx.xbits = _forward_layers(ys, h, params, L, bounded, last_bits=False)
        h = ys[-1]
        ctx.L = L
        ctx.save_for_backward(x, *params[0::2], *ys)
        return h

    @staticmethod
    @torch.no_grad()
    def backward(ctx, g):
        L = ctx.L
        saved = ctx.saved_tensors
        x, ws, ys = saved[0], saved[1:1 + L], saved[1 + L:]
        g = (g * (ys[-1] > 0)).contiguous()
        grads = [None] * (2 * L)
        gx = None
        carried = None
        for l in range(L - 1, -1, -1):
            xin = ys[l - 1] if l > 0 else x
            xb = ctx.xbits[l]
            rb, ab = _gbits(g, xb is not None, xin.shape[1], l > 0 or ctx.needs_input_grad[0], ctx.needs_input_grad[1 + 2 * l], carried)
            grads[2 * l], grads[2 * l + 1] = _wb_grads(g, xin, ctx.needs_input_grad[1 + 2 * l], ctx.needs_input_grad[2 + 2 * l], g_bits=ab,
                                                       x_bits=xb if ab is not None else None)
            wt = _kernel_weight(g, ws[l].t())
            if l > 0:
                left = []                           # the row-scaled kernel leaves its output's row / tensor maxima: the next layer's scales
                g = ops.dense_gated(g, wt, xin, row_bits=rb, bits_out=left)
                carried = left[0] if left else None
            elif ctx.needs_input_grad[0]:
                gx = ops.dense(g, wt, None, relu=False, row_bits=rb)
        return (gx,) + tuple(grads)


MLP_HEAD = _os.environ.get("DIR_MLP_HEAD", "1") != "0"        # development switch: 0 keeps the two-node formulation (mlp_stack + units1)


class _MlpHeadFn(torch.autograd.Function):
    """A stack of dense + ReLU layers AND the units = 1 logit layer on top of it (deepFM.py:284-317, ESMM.py:130-147) as one autograd
    node.  Forward: dir_dense_f32 per layer, the head as a library GEMM with one output column.  Backward: the head's three gradients and the ReLU gate of the
    top hidden layer in ONE pass over its output (dir_units1_relu_backward_f32: dL/dpre_top, dL/dw_head and the top layer's bias
    gradient; as separate torch ops: an outer product, g * y, a compare, a mask multiply and two column sums), then _MlpStackFn's
    loop."""

    @staticmethod
    def forward(ctx, x, head_w, head_b, *params):
        bounded = bool(_FWD_BOUNDED[0])
        L = len(params) // 2
        ys, h = [], x
        ctx.xbits = _forward_layers(ys, h, params, L, bounded, last_bits=False)
        h = ys[-1]
        ctx.L = L
        ctx.save_for_backward(x, head_w, *params[0::2], *ys)
        return ops.units1(h, head_w, head_b)        # (dir_units1_f32; the library ran it as a one-column GEMM: 26 us at 65 536 x 400, its GEMV 62)

    @staticmethod
    @torch.no_grad()
    def backward(ctx, g):
        L = ctx.L
        saved = ctx.saved_tensors
        x, head_w, ws, ys = saved[0], saved[1], saved[2:2 + L], saved[2 + L:]
        g_head_b = g.reshape(-1).sum(dim=0, keepdim=True) if ctx.needs_input_grad[2] else None
        g, gw_head, gb_top, carried = ops.units1_relu_backward(g, head_w, ys[-1], want_bits=True)
        grads = [None] * (2 * L)
        gx = None
        for l in range(L - 1, -1, -1):
            xin = ys[l - 1] if l > 0 else x
            xb = ctx.xbits[l]
            rb, ab = _gbits(g, xb is not None, xin.shape[1], l > 0 or ctx.needs_input_grad[0], ctx.needs_input_grad[3 + 2 * l], carried)
            xb = xb if ab is not None else None
            if l == L - 1:                                           # the top layer's bias gradient came with the head's backward
                grads[2 * l] = _tn_matmul(g, xin, g_bits=ab, x_bits=xb) if ctx.needs_input_grad[3 + 2 * l] else None
                grads[2 * l + 1] = gb_top if ctx.needs_input_grad[4 + 2 * l] else None
            else:
                grads[2 * l], grads[2 * l + 1] = _wb_grads(g, xin, ctx.needs_input_grad[3 + 2 * l], ctx.needs_input_grad[4 + 2 * l], g_bits=ab, x_bits=xb)
            wt = _kernel_weight(g, ws[l].t())
            if l > 0:
                left = []
                g = ops.dense_gated(g, wt, xin, row_bits=rb, bits_out=left)
                carried = left[0] if left else None
            elif ctx.needs_input_grad[0]:
                gx = ops.dense(g, wt, None, relu=False, row_bits=rb)
        return (gx, gw_head.reshape(1, -1) if ctx.needs_input_grad[1] else None, g_head_b) + tuple(grads)


def mlp_head_supported(lins, head, x, activation):
    """mlp_stack_supported and a units = 1 head with a bias on top (the fused head backward reads the top layer's output once)."""
    return (MLP_HEAD and mlp_stack_supported(lins, x, activation) and head.out_features == 1 and head.bias is not None
            and head.in_features == lins[-1].out_features and head.in_features <= 4096)


_FWD_BOUNDED = [False]      # the forward of the node being applied may take the fp16 x 2 layers (its input is an embedding concatenation)


def mlp_head(lins, head, x, embedding_input=False):
    """head(relu(lin_L(... relu(lin_1(x))))) -> [B, 1] through _MlpHeadFn (check mlp_head_supported first).  embedding_input: the caller
    vouches that x is a concatenation of embedding rows (DeepFM's dnn input, ESMM's input layer over embedding columns): the forward layers
    and the weight gradients (g scaled by one power of two, x as it is) may then run the fp16 x 2 kernels; without it the first layer runs
    the row-scaled general-input kernel and its weight gradient bf16 x 3."""
    params = []
    for lin in lins:
        params += [lin.weight, lin.bias]
    _FWD_BOUNDED[0] = bool(embedding_input)
    try:
        return _MlpHeadFn.apply(x, head.weight, head.bias, *params)
    finally:
        _FWD_BOUNDED[0] = False


def mlp_stack_supported(lins, x, activation):
    """A run of nn.Linear layers + ReLU the stack node covers: every layer on the kernel, biases present, and the data gradient
    of the first layer expressible as a dense product too (its in_features a multiple of 4 and >= 16)."""
    if activation not in _RELUS or not len(lins) or not torch.is_grad_enabled() or x.shape[0] < MIN_ROWS:
        return False
    h_dim = x.shape[1]
    if not ops.dense_supported(x, lins[0].weight) or h_dim < 16:
        return False
    for lin in lins:
        if lin.bias is None or lin.in_features != h_dim or lin.in_features % 4 or lin.out_features % 4 or lin.out_features < 16:
            return False
        h_dim = lin.out_features
    return True


def mlp_stack(lins, x, embedding_input=False):
    """relu(lin_L(... relu(lin_1(x)))) through _MlpStackFn (check mlp_stack_supported first); embedding_input as in mlp_head."""
    params = []
    for lin in lins:
        params += [lin.weight, lin.bias]
    _FWD_BOUNDED[0] = bool(embedding_input)
    try:
        return _MlpStackFn.apply(x, *params)
    finally:
        _FWD_BOUNDED[0] = False


class _Units1Fn(torch.autograd.Function):
    """The units = 1 logit layers (deepFM.py:311-317, DeepCrossNetwork.py:137, ESMM.py:146) in training: the library's backward
    of a [B, in] x [in, 1] product is a TN GEMM with one output row (112-158 us at 65 536 x 400) and a rank-1 GEMM (40 us); as
    elementwise products they cost 72 and 22 us (tools/units1_probe.py)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        if x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and weight.dtype == torch.float32:
            return ops.units1(x, weight, bias)      # dir_units1_f32: one pass over x, any width
        y = x @ weight.t()
        return y + bias if bias is not None else y

    @staticmethod
    @torch.no_grad()
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        gb = g.sum(dim=0) if ctx.has_bias and ctx.needs_input_grad[2] else None
        if ctx.needs_input_grad[1] and x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32 and g.dtype == torch.float32:
            # one pass over x for both gradients (dir_units1_backward_f32): torch's column sum of g * x takes 0.68 ms at 65 536 x 429
            gx, gw = ops.units1_backward(g, weight, x, want_gx=ctx.needs_input_grad[0])
            return gx, gw.reshape(1, -1), gb
        gx = g * weight if ctx.needs_input_grad[0] else None                    # [B, 1] * [1, in]
        gw = (g * x).sum(dim=0, keepdim=True) if ctx.needs_input_grad[1] else None
        return gx, gw, gb


def units1(lin, x):
    """lin(x) for an nn.Linear with one output unit; the elementwise backward when training on the GPU."""
    if lin.out_features == 1 and x.is_cuda and x.dim() == 2 and torch.is_grad_enabled() and (x.requires_grad or lin.weight.requires_grad):
        return _Units1Fn.apply(x, lin.weight, lin.bias)
    if lin.out_features == 1 and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.stride(1) == 1 and lin.weight.dtype == torch.float32 \
            and not (torch.is_grad_enabled() and (x.requires_grad or lin.weight.requires_grad)):
        return ops.units1(x, lin.weight.detach(), lin.bias.detach() if lin.bias is not None else None)
    return lin(x)


_BN_CACHE = {}


def _bn_affine(bn):
    """Inference batch-norm as one per-column affine (scale, shift), cached until a statistic or parameter changes."""
    key = id(bn)
    ver = (bn.moving_mean._version, bn.moving_variance._version, bn.beta._version, bn.gamma._version if bn.gamma is not None else -1,
           bn.beta.data_ptr())
    hit = _BN_CACHE.get(key)
    capturing = bn.moving_mean.is_cuda and torch.cuda.is_current_stream_capturing()
    if hit is not None and hit[0] is bn and hit[1] == ver and not ops.capture_bypasses_caches(bn.moving_mean):
        return hit[2], hit[3]
    inv = torch.rsqrt(bn.moving_variance + bn.eps)
    if bn.gamma is not None:
        inv = inv * bn.gamma.data
    shift = bn.beta.data - bn.moving_mean * inv
    if capturing:
        return inv.contiguous(), shift.contiguous()
    if len(_BN_CACHE) > 256:
        _BN_CACHE.clear()
    _BN_CACHE[key] = (bn, ver, inv.contiguous(), shift.contiguous())
    return _BN_CACHE[key][2], _BN_CACHE[key][3]


def dense_act(lin, x, activation=None, bn=None, bounded_input=False):
    """activation(lin(x)) for an nn.Linear `lin`, on dir_dense_f32 when the layer is covered.  x may arrive row-padded with zero
    columns up to the next multiple of 4 (InputLayer(pad_to=4), inference): then only the weight gets its zero columns.
    bn: the batch-norm module that follows the activation (deepFM.py:303-308, DeepCrossNetwork.py:400-403) -- in inference it is
    folded into the kernel's epilogue (one pass over the layer's output instead of three); otherwise it is applied here.
    bounded_input: the caller vouches that x is bounded by construction (a batch-normalised activation, an embedding concatenation): in
    training the layer's weight gradient may then run the fp16 x 2 kernel."""
    y = _dense_act(lin, x, activation, bn, bounded_input)
    return y


def _apply_bn(bn, y):
    return bn(y) if bn is not None else y


def _dense_act(lin, x, activation, bn, bounded=False):
    relu = activation in _RELUS
    prepadded = x.dim() == 2 and x.shape[1] != lin.in_features and x.shape[1] == lin.in_features + (-lin.in_features) % 4
    if (activation is None or relu) and x.is_cuda and x.dim() == 2 and lin.out_features >= 16 and (
            x.shape[0] >= MIN_ROWS or ops.dense_small_covers(x.shape[0], lin.in_features + (-lin.in_features) % 4, lin.out_features)
            or ops.dense_mid_covers(x.shape[0], lin.in_features + (-lin.in_features) % 4, lin.out_features)):
        _route("hip", lin)
        train = torch.is_grad_enabled() and (x.requires_grad or lin.weight.requires_grad)
        fold = bn is not None and not (bn.training and torch.is_grad_enabled())
        ps, psh = _bn_affine(bn) if fold else (None, None)
        if prepadded and not train and x.dtype == torch.float32:
            y = ops.dense(x, _packed_cached_padded(lin.weight, x.shape[1] - lin.in_features), lin.bias, relu=relu, post_scale=ps, post_shift=psh)
            return y if fold else _apply_bn(bn, y)
        if prepadded:
            x = x[:, :lin.in_features]
            prepadded = False
        pad = (-x.shape[1]) % 4
        if pad and x.dtype == torch.float32:
            # in_features not a multiple of 4 (DCN's 429-wide first layer: 26 x 16 + 13 numeric columns): zero-pad x and the weight
            # to the next multiple of 4 -- one [B, in] copy (45 us at 65 536 x 429) against 200 us saved on the GEMM
            xp = F.pad(x, (0, pad))
            if train:
                return _dense_bn_train(bn, xp, F.pad(lin.weight, (0, pad)), lin.bias, relu, bounded)     # pad's backward slices the gradients back
            y = ops.dense(xp, _packed_cached_padded(lin.weight, pad), lin.bias, relu=relu, post_scale=ps, post_shift=psh)
            return y if fold else _apply_bn(bn, y)
        if ops.dense_supported(x, lin.weight):
            if train:
                return _dense_bn_train(bn, x, lin.weight, lin.bias, relu, bounded)
            y = ops.dense(x, _packed_cached(lin.weight), lin.bias, relu=relu, post_scale=ps, post_shift=psh)
            return y if fold else _apply_bn(bn, y)
    if prepadded:
        x = x[:, :lin.in_features]
    _route("library", lin)
    y = lin(x)
    return _apply_bn(bn, activation(y) if activation is not None else y)


def tower_infer(lins, x, activation, bns=None, head=None, adds=(), gather=None, embedding_input=False):
    """Inference of a whole tower -- activation(lin_l(...)) for every nn.Linear in `lins`, each followed by its batch-norm (bns[l], folded
    to an affine), and the units = 1 logit layer `head` on top when given -- in ONE launch of dir_tower_bf16x3_f32 (csrc/tower_bf3.hip:
    the activations of a 128-row tile stay in registers from layer to layer).  -> the logits [B, 1] (+ the [B, 1] tensors in adds, at
    most two: the other terms of add_n, deepFM.py:217-223) or, without a head, the last activation; None when the tower is not covered
    (autograd recording, fewer than ops.TOWER_MIN_ROWS rows, a width over 416, under ops.TOWER_MIN_WIDTH or not a multiple of 4, an activation other than ReLU /
    none, a head with more than one unit): the caller then runs it layer by layer.
    gather = (ops.PackedTables, ids [B, F], linear bias) with x = None: the input rows are looked up inside the same launch and the FM
    and first-order terms join the logit (dir_deepfm_tower_bf16x3_f32; needs a head)."""
    if gather is not None:
        if (x is not None or (head is None and (len(gather) < 4 or gather[3])) or torch.is_grad_enabled() or gather[1].shape[0] < ops.TOWER_MIN_ROWS
                or not ops.tower_gather_covers(gather[0], [l.weight for l in lins])):
            return None
    elif torch.is_grad_enabled() and (x.requires_grad or any(l.weight.requires_grad for l in lins)):
        return None
    if not len(lins) or (gather is None and (x.dim() != 2 or x.shape[0] < ops.TOWER_MIN_ROWS)) or not (activation is None or activation in _RELUS):
        return None
    if bns is not None and len(bns) and any(b.training and torch.is_grad_enabled() for b in bns):
        return None
    ws = [l.weight for l in lins]
    if (gather is None and not ops.tower_covers(x, ws)) or min(int(w.shape[0]) for w in ws) < (ops.TOWER_MIN_WIDTH if gather is None else 16):
        return None                                   # (with the lookups inside, the launch it saves outweighs the idle column tiles of a narrow layer)
    if head is not None and (head.out_features != 1 or head.bias is None or head.in_features != lins[-1].out_features):
        return None
    ps = psh = None
    if bns is not None and len(bns):
        aff = [_bn_affine(b) if b is not None else (None, None) for b in bns]
        aff += [(None, None)] * (len(lins) - len(aff))
        ps, psh = [a[0] for a in aff], [a[1] for a in aff]
    # embedding_input: the caller vouches that x is a concatenation of embedding rows (DeepFM's dnn input, deepFM.py:288-291) -- the
    # bounded magnitudes the fp16 x 2 arithmetic wants; a general input layer may carry raw numeric columns (the adult-census
    # capital_gain column of DeepCrossNetwork/train.py reaches 99 999 > fp16's 65 504) and stays on bf16 x 3.  The gather form reads
    # embedding rows by construction.
    return ops.tower(x, ws, [l.bias for l in lins], relu=activation is not None, post_scale=ps, post_shift=psh,
                     head=(head.weight, head.bias) if head is not None else None, adds=adds if head is not None else (), gather=gather,
                     split=None if (gather is not None or embedding_input) else "bf16x3")
