"""The fp16 x 2 kernels on operands OUTSIDE the magnitudes TensorFlow's initialisers produce:
the DEFAULT paths must neither overflow (|v| >= 65 504 -> inf) nor lose relative precision on small operands (below 2^-3 the unscaled
split keeps an absolute 2^-25 per element only).  Since round 5 the forward CIN and dense kernels scale BOTH operands by exact powers of
two inside the kernels (rows of the left operand, the weight as a tensor / per row), so the bars here are scale-invariant:

    |err| <= 1e-5 * (|ref| + rms(ref))          against float64

and scaling an operand by a power of two must scale the result by exactly that power, bit for bit (nothing reaches fp16's subnormals or
its overflow).  No reference code exists for the CIN (README.md:28 -> arXiv:1803.05170); the dense layers are deepFM.py:295-300."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bar(got, ref, tol=1e-5):
    ref = ref.double()
    err = (got.double() - ref).abs()
    rms = ref.pow(2).mean().sqrt()
    assert bool(torch.isfinite(got).all()), "non-finite values in the result"
    worst = float((err / (ref.abs() + rms)).max())
    assert worst <= tol, "scale-invariant error %.2e > %.0e" % (worst, tol)
    return worst


def _cin_ref(x0, xk, W):
    B, m, D = x0.shape
    Hp = xk.shape[1]
    W3 = W.double().view(-1, Hp, m)
    return torch.einsum("hij,bid,bjd->bhd", W3, xk.double(), x0.double())


@pytest.mark.parametrize("sx", [2.0 ** -10, 1.0, 2.0 ** 20])
@pytest.mark.parametrize("sw", [2.0 ** -12, 1.0, 2.0 ** 20])
def test_cin_default_path_is_scale_free(built_lib, sx, sw):
    """ops.cin_layer's default ("auto") on a table scaled by 2^-10 / 2^20 and a CIN weight scaled by 2^-12 / 2^20: first layer (pair form),
    a middle layer (xk = another tensor) and the pooled-only last layer, against float64."""
    from dir_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(11)
    B, m, D, H = 12288, 26, 16, 128
    x0 = torch.randn((B, m, D), generator=g, device=dev) * 0.25 * sx
    W1 = torch.randn((H, m * m), generator=g, device=dev) * (sw / m)
    x1, p1 = ops.cin_layer(x0, x0, W1)
    ref1 = _cin_ref(x0, x0, W1)
    _bar(x1, ref1)
    _bar(p1, ref1.sum(-1))
    # middle layer: xk is the (float32) output of layer 1
    W2 = torch.randn((H, H * m), generator=g, device=dev) * (sw / (H * m) ** 0.5)
    x2, p2 = ops.cin_layer(x0, x1, W2)
    ref2 = _cin_ref(x0, x1, W2)
    _bar(x2, ref2)
    _bar(p2, ref2.sum(-1))
    # last layer of a stack: pooled sums only
    none, p3 = ops.cin_layer(x0, x1, W2, want_xout=False)
    assert none is None
    _bar(p3, ref2.sum(-1))


def test_cin_power_of_two_scales_come_out_exactly(built_lib):
    """x0 * 2^a, W * 2^b -> the first layer's result * 2^(2a + b) and the middle layer's * 2^(a + b) (xk unchanged), bit for bit."""
    from dir_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(12)
    B, m, D, H = 12288, 26, 16, 128
    x0 = torch.randn((B, m, D), generator=g, device=dev) * 0.25
    xk = torch.randn((B, H, D), generator=g, device=dev)
    W1 = torch.randn((H, m * m), generator=g, device=dev) / m
    W2 = torch.randn((H, H * m), generator=g, device=dev) / (H * m) ** 0.5
    base1, bp1 = ops.cin_layer(x0, x0, W1)
    base2, bp2 = ops.cin_layer(x0, xk, W2)
    for a, b in ((-14, 9), (17, -6), (30, 30), (-30, -20)):
        xs = x0 * 2.0 ** a
        y1, q1 = ops.cin_layer(xs, xs, W1 * 2.0 ** b)
        assert torch.equal(y1, base1 * 2.0 ** (2 * a + b)) and torch.equal(q1, bp1 * 2.0 ** (2 * a + b)), (a, b)
        y2, q2 = ops.cin_layer(xs, xk * 2.0 ** a, W2 * 2.0 ** b)
        assert torch.equal(y2, base2 * 2.0 ** (2 * a + b)) and torch.equal(q2, bp2 * 2.0 ** (2 * a + b)), (a, b)


def test_xdeepfm_module_on_scaled_tables_and_weights(built_lib):
    """The XDeepFM module's default inference forward with its embedding tables scaled by 2^20 and its CIN weights by 2^20 (finite, equal to
    the float64 restatement) and with tables scaled by 2^-10 (relative bar)."""
    from dir_amd import feature_column as fc
    from dir_amd.xdeepfm import XDeepFM
    dev = torch.device("cuda:0")
    F, V, K, B = 26, 500, 16, 12288
    cats = [fc.categorical_column_with_identity("C%d" % i, V) for i in range(F)]
    torch.manual_seed(5)
    model = XDeepFM(linear_feature_columns=cats, dnn_feature_columns=[fc.embedding_column(c, K) for c in cats], dnn_hidden_units=[64, 64],
                    cin_layer_sizes=(64, 64)).to(dev).eval()
    ids = torch.randint(0, V, (B, F), device=dev)
    for st, sw in ((2.0 ** 20, 2.0 ** 20), (2.0 ** -10, 1.0)):
        with torch.no_grad():
            for t in model.embedding_weights:
                t.mul_(st)
            for w in model.cin_W:
                w.mul_(sw)
            x0f = torch.stack([model.embedding_weights[i][ids[:, i]] for i in range(F)], dim=1)
            pooled = model.cin(x0f.contiguous())
            x0 = x0f.double()
            xk, ref = x0, []
            for W in model.cin_W:
                xk = torch.einsum("hij,bid,bjd->bhd", W.double().view(W.shape[0], xk.shape[1], F), xk, x0)
                ref.append(xk.sum(-1))
            _bar(pooled, torch.cat(ref, dim=1))
            for t in model.embedding_weights:
                t.div_(st)
            for w in model.cin_W:
                w.div_(sw)


@pytest.mark.parametrize("sx", [2.0 ** -14, 1.0, 2.0 ** 18])
def test_bounded_dense_forward_and_weight_gradient_are_scale_free(built_lib, sx):
    """dense(arith="auto_bounded") -- what the training towers run on an embedding concatenation -- and the tower's weight gradient with the
    input scaled by 2^-14 / 2^18: the forward on the row-scaled kernel, dL/dW with BOTH operands scaled by tensor powers of two."""
    from dir_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(13)
    M, Kd, N = 16384, 416, 400
    x = torch.randn((M, Kd), generator=g, device=dev) * 0.25 * sx
    W = torch.randn((N, Kd), generator=g, device=dev) * 0.05
    b = torch.randn((N,), generator=g, device=dev) * 0.1 * sx
    xb, left = [], []
    y = ops.dense(x, W, b, relu=True, arith="auto_bounded", bits_out=left, xbits_out=xb)
    ref = torch.relu(x.double() @ W.double().t() + b.double())
    _bar(y, ref)
    assert xb and int(xb[0].item()) == int(x.abs().max().view(torch.int32).item())                   # the max pass's tensor maximum of x
    assert left and int(left[0][1].item()) == int(y.abs().max().view(torch.int32).item())           # the epilogue's of y
    gy = torch.randn((M, N), generator=g, device=dev) * 1e-4
    gb = ops.grad_bits(gy)
    dW = ops.dense_dw(gy, x, g_bits=gb[1], x_bits=xb[0])
    _bar(dW, gy.double().t() @ x.double(), tol=2e-5)
    # the scale of x comes out of dW exactly
    x2 = x * 2.0 ** 7
    xb2 = ops.row_absmax_bits(x2, want_all=True)[1]
    assert torch.equal(ops.dense_dw(gy, x2, g_bits=gb[1], x_bits=xb2), dW * 2.0 ** 7)


def test_mlp_head_trains_on_large_inputs(built_lib):
    """The fused tower node (dense.mlp_head, embedding_input=True: DeepFM's training tower) on an input of magnitude 2^17: finite gradients
    equal to float64 autograd (the unscaled fp16 x 2 kernels of round 4 returned inf here)."""
    from dir_amd import dense
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    M, Kd = 16384, 416
    lins = [torch.nn.Linear(Kd, 400).to(dev), torch.nn.Linear(400, 400).to(dev)]
    head = torch.nn.Linear(400, 1).to(dev)
    with torch.no_grad():
        lins[0].weight.mul_(2.0 ** -15)                       # keeps the activations O(1) behind the 2^17 input
        # every unit ACTIVE (pre-activations ~ 8 +- 0.3): a ReLU gate that flips between the fp32 forward and the float64 reference moves
        # dL/dW by g x -- with |x| ~ 2^15 one flipped gate among 6.5 M shows as a 1e-2 error (gating itself: tests/test_gpu_backward.py)
        lins[0].bias.fill_(8.0)
        lins[1].bias.fill_(8.0)
    x = (torch.randn((M, Kd), device=dev) * 0.25 * 2.0 ** 17).requires_grad_(True)
    assert dense.mlp_head_supported(lins, head, x, torch.nn.functional.relu)
    out = dense.mlp_head(lins, head, x, embedding_input=True)
    gout = torch.randn((M, 1), device=dev) * 1e-3
    out.backward(gout)
    params = [lins[0].weight, lins[0].bias, lins[1].weight, lins[1].bias, head.weight]
    got = [p.grad.clone() for p in params] + [x.grad.clone()]
    xd = x.detach().double().requires_grad_(True)
    pd = [p.detach().double().requires_grad_(True) for p in params]
    h = torch.relu(xd @ pd[0].t() + pd[1])
    h = torch.relu(h @ pd[2].t() + pd[3])
    o = h @ pd[4].t() + head.bias.detach().double()
    refs = torch.autograd.grad(o, pd + [xd], gout.double())
    _bar(out.detach(), o.detach())
    for gg, rr in zip(got, refs):
        _bar(gg, rr, tol=5e-5)


@pytest.mark.parametrize("log2s", [20, -10, 0])
def test_din_unit_checks_its_operands_range(built_lib, oracle, log2s, monkeypatch):
    """The DIN unit's fp16 x 2 kernel splits table rows unscaled, so ops.din_attention_pool MEASURES max |table| / max |W| (per tensor
    version) and takes bf16 x 3 outside [2^-6, 2^15): a table scaled by 2^20 returns the float64 answer (round 4's default returned inf),
    one scaled by 2^-10 keeps a relative 1e-5.  W1's blocks are scaled back so that the unit sees the same pre-activations."""
    from dir_amd import ops
    monkeypatch.delenv("DIR_DIN_ARITH", raising=False)
    rng = np.random.default_rng(7)
    B, T, K, H1, H2, V = 256, 50, 64, 80, 40, 2000
    s = 2.0 ** log2s
    table = (rng.standard_normal((V, K)) * 0.3 * s).astype(np.float32)
    hist = rng.integers(-1, V, size=(B, T)).astype(np.int64)
    hl = rng.integers(0, T + 1, size=B).astype(np.int32)
    cand = rng.integers(0, V, size=B).astype(np.int64)
    W1 = (rng.standard_normal((4 * K, H1)) * 0.1).astype(np.float32)
    W1[:3 * K] /= s                                        # [h, a, h - a] blocks
    W1[3 * K:] /= s * s                                    # h * a block
    b1 = (rng.standard_normal(H1) * 0.1).astype(np.float32)
    W2 = (rng.standard_normal((H1, H2)) * 0.2).astype(np.float32); b2 = (rng.standard_normal(H2) * 0.1).astype(np.float32)
    W3 = (rng.standard_normal(H2) * 0.3).astype(np.float32); b3 = np.array([0.05], np.float32)
    dev = lambda a: torch.from_numpy(a).cuda()             # noqa: E731
    t, w1, w2, w3 = dev(table), dev(W1), dev(W2), dev(W3)
    code = ops.din_arith(t, (w1, w2, w3))
    assert code == (ops.DIN_ARITHS["f16x2"] if log2s == 0 else ops.DIN_ARITHS["bf16x3"])
    for normalize in (False, True):
        ref_o, ref_s = oracle.din_attention_pool(table, hist, hl, cand, W1, b1, W2, b2, W3, b3, normalize=normalize, acc64=True)
        got_o, got_s = ops.din_attention_pool(t, dev(hist), dev(hl), dev(cand), w1, dev(b1), w2, dev(b2), w3, dev(b3), normalize=normalize,
                                              want_scores=True)
        _bar(got_s, torch.from_numpy(ref_s).cuda())
        _bar(got_o, torch.from_numpy(ref_o).cuda())
    # the measurement follows the tensor's version: an in-place rescale is seen (inference: every change; training: within DIN_RANGE_RECHECK updates)
    t2 = dev((rng.standard_normal((V, K)) * 0.3).astype(np.float32))
    assert ops.din_arith(t2, ()) == ops.DIN_ARITHS["f16x2"]
    t2.mul_(2.0 ** 18)
    keep = ops.DIN_RANGE_RECHECK
    ops.DIN_RANGE_RECHECK = 1
    try:
        assert ops.din_arith(t2, ()) == ops.DIN_ARITHS["bf16x3"]
    finally:
        ops.DIN_RANGE_RECHECK = keep


def test_cin_rows_of_very_different_magnitudes_and_zero_operands(built_lib):
    """Rows of xk spread over 50 binades in ONE batch (samples scaled by 2^-20 .. 2^30): the device-side verdict sees a tensor outside the plain
    kernel's window and names the row-scaled kernel, whose per-row scales keep EVERY row accurate relative to its own size (a per-row bar, not
    the batch's rms); all-zero operands give exact zeros, never NaN (a zero row's scale is clamped, a zero W's too)."""
    from dir_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(21)
    B, m, D, H = 12288, 26, 16, 128
    x0 = torch.randn((B, m, D), generator=g, device=dev) * 0.25
    W1 = torch.randn((H, m * m), generator=g, device=dev) / m
    W2 = torch.randn((H, H * m), generator=g, device=dev) / (H * m) ** 0.5
    sc = torch.pow(2.0, torch.randint(-20, 31, (B, 1, 1), generator=g, device=dev).float())
    x0s = x0 * sc                                          # every sample at its own scale
    x1, _ = ops.cin_layer(x0s, x0s, W1)
    x2, p2 = ops.cin_layer(x0s, x1, W2)                    # x1 carries the first layer's row maxima: the verdict path
    r1 = _cin_ref(x0s, x0s, W1)
    r2 = _cin_ref(x0s, x1, W2)
    for got, ref in ((x1, r1), (x2, r2)):
        assert bool(torch.isfinite(got).all())
        rms = ref.pow(2).mean(dim=(1, 2), keepdim=True).sqrt()                 # per SAMPLE
        worst = float(((got.double() - ref).abs() / (ref.abs() + rms)).max())
        assert worst <= 1e-5, worst
    prms = r2.sum(-1).pow(2).mean(dim=1, keepdim=True).sqrt()
    assert float(((p2.double() - r2.sum(-1)).abs() / (r2.sum(-1).abs() + prms)).max()) <= 1e-5
    z = torch.zeros_like(x0)
    for a, b, w in ((z, z, W1), (x0, x0, torch.zeros_like(W1))):
        y, p = ops.cin_layer(a, b, w)
        assert not bool(y.any()) and not bool(p.any())
        y2, p2_ = ops.cin_layer(x0, y, W2)                  # a zero xk with its (zero) row maxima: the verdict takes the row-scaled kernel
        assert not bool(y2.any()) and not bool(p2_.any())


def test_cin_verdict_sees_tiny_rows_inside_an_in_window_tensor(built_lib):
    """a batch whose LARGEST value sits inside the plain kernel's window [2^-4, 2^15) while some samples are 2^-14 .. 2^-9 of it.  The
    device-side verdict used to look at the tensor maximum only and ran the unscaled kernel, leaving those rows an absolute 2^-25 per
    element; it now also needs the smallest non-zero row maximum >= 2^-8 and otherwise names the row-scaled kernel -- held to the PER-SAMPLE
    bar of the test above."""
    from dir_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(33)
    B, m, D, H = 12288, 26, 16, 128
    x0 = torch.randn((B, m, D), generator=g, device=dev) * 0.5
    W1 = torch.randn((H, m * m), generator=g, device=dev) / m
    W2 = torch.randn((H, H * m), generator=g, device=dev) / (H * m) ** 0.5
    sc = torch.ones((B, 1, 1), device=dev)
    sc[::7] = torch.pow(2.0, torch.randint(-7, -4, (len(sc[::7]), 1, 1), generator=g, device=dev).float())     # x1 rows scale with sc^2: 2^-14 .. 2^-10
    x0s = x0 * sc
    x1, _ = ops.cin_layer(x0s, x0s, W1)
    amax = float(x1.abs().max())
    assert 2.0 ** -4 <= amax < 2.0 ** 15                   # the tensor is inside the window: only the row floor can name the row-scaled kernel
    x2, p2 = ops.cin_layer(x0s, x1, W2)
    r2 = _cin_ref(x0s, x1, W2)
    rms = r2.pow(2).mean(dim=(1, 2), keepdim=True).sqrt()
    worst = float(((x2.double() - r2).abs() / (r2.abs() + rms)).max())
    assert worst <= 1e-5, worst
    prms = r2.sum(-1).pow(2).mean(dim=1, keepdim=True).sqrt()
    assert float(((p2.double() - r2.sum(-1)).abs() / (r2.sum(-1).abs() + prms)).max()) <= 1e-5
