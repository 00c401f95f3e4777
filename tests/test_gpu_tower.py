"""GPU tests of the fused tower kernel (csrc/tower_bf3.hip, dir_tower_bf16x3_f32): dnn_logit_fn (models/DeepFM/deepFM.py:284-319) in one
launch against a float64 restatement, against the layer-by-layer kernels, rerun-bitwise, and inside the DeepFM module."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["rows", "cs"])
def _tower_kernel(request):
    """Every test of this file under both tower kernels: tower_bf3_k (the default) and the column-split tower_cs_k (csrc/tower_cs.hip,
    DIR_TOWER_KERNEL=cs; fp16 x 2 only -- bf16 x 3 requests run tower_bf3_k under either setting)."""
    from dir_amd import ops
    old = ops.TOWER_KERNEL
    ops.TOWER_KERNEL = request.param
    yield
    ops.TOWER_KERNEL = old


def _ref64(x, Ws, bs, relus, scales, shifts, head=None, adds=()):
    h = x.astype(np.float64)
    for W, b, r, sc, sh in zip(Ws, bs, relus, scales, shifts):
        h = h @ W.astype(np.float64).T
        if b is not None:
            h = h + b.astype(np.float64)
        if r:
            h = np.maximum(h, 0.0)
        if sc is not None:
            h = h * sc.astype(np.float64) + sh.astype(np.float64)
    if head is not None:
        h = h @ head[0].astype(np.float64).reshape(-1, 1) + float(head[1][0])
        for a in adds:
            h = h + a.astype(np.float64).reshape(-1, 1)
    return h


def _scaled_err(got, ref):
    return float((np.abs(got.astype(np.float64) - ref) / (1 + np.abs(ref))).max())


@pytest.mark.parametrize("M,Kd,Ns,head,nadd,bn,relu", [
    (300, 416, [400, 400, 400], True, 2, False, True),        # the DeepFM tower (deepFM.py:284-319) + fm and linear logits added
    (129, 64, [32], False, 0, False, True),                   # one narrow layer, activation out, a partial tile
    (5000, 416, [360, 200, 80], True, 0, False, True),        # ESMM's tower (ESMM.py:139-146)
    (1000, 52, [100, 36], True, 1, True, True),               # batch-norm affine behind the activation (deepFM.py:303-308)
    (257, 416, [416, 4, 416, 16], False, 0, True, False),     # four layers, no activation, widths from 4 to 416
    (1, 16, [8], True, 0, False, True),
])
def test_tower_matches_float64_and_layerwise(built_lib, M, Kd, Ns, head, nadd, bn, relu):
    from dir_amd import ops
    rng = np.random.default_rng(M + Kd + len(Ns))
    x = (rng.standard_normal((M, Kd)) * 0.5).astype(np.float32)
    dims = [Kd] + Ns
    Ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(len(Ns))]
    bs = [(rng.standard_normal(n) * 0.1).astype(np.float32) for n in Ns]
    scales = [(1 + 0.2 * rng.standard_normal(n)).astype(np.float32) if bn else None for n in Ns]
    shifts = [(0.1 * rng.standard_normal(n)).astype(np.float32) if bn else None for n in Ns]
    hw = (rng.standard_normal(Ns[-1]) / np.sqrt(Ns[-1])).astype(np.float32)
    hb = np.array([0.3], np.float32)
    adds = [rng.standard_normal(M).astype(np.float32) for _ in range(nadd)]
    d = lambda a: torch.from_numpy(a).cuda() if a is not None else None  # noqa: E731
    xd, Wd, bd = d(x), [d(w) for w in Ws], [d(b) for b in bs]
    kw = dict(relu=relu, post_scale=[d(s) for s in scales] if bn else None, post_shift=[d(s) for s in shifts] if bn else None)
    if head:
        got = ops.tower(xd, Wd, bd, head=(d(hw), d(hb)), adds=[d(a) for a in adds], **kw)
        assert tuple(got.shape) == (M, 1)
    else:
        got = ops.tower(xd, Wd, bd, **kw)
        assert tuple(got.shape) == (M, Ns[-1])
    ref = _ref64(x, Ws, bs, [relu] * len(Ns), scales, shifts, (hw, hb) if head else None, adds)
    assert _scaled_err(got.cpu().numpy(), ref) <= 1e-5
    # the same tower layer by layer on the fp32-MFMA kernel: both sit inside the 1e-5 bar, so they agree to 2e-5
    h = xd
    for l in range(len(Ns)):
        ok = ops.dense_supported(h, Wd[l])
        if ok:
            h = ops.dense(h, Wd[l], bd[l], relu=relu, post_scale=kw["post_scale"][l] if bn else None, post_shift=kw["post_shift"][l] if bn else None,
                          arith="f32")
        else:
            h = h @ Wd[l].t() + bd[l]
            h = torch.relu(h) if relu else h
            h = h * kw["post_scale"][l] + kw["post_shift"][l] if bn else h
    if head:
        h = h @ d(hw).reshape(-1, 1) + d(hb)
        for a in adds:
            h = h + d(a).reshape(-1, 1)
    assert _scaled_err(got.cpu().numpy(), h.double().cpu().numpy()) <= 2e-5
    again = ops.tower(xd, Wd, bd, head=(d(hw), d(hb)), adds=[d(a) for a in adds], **kw) if head else ops.tower(xd, Wd, bd, **kw)
    assert torch.equal(got, again)


def test_tower_full_size_rerun_and_image_refresh(built_lib):
    """BASELINE config 2's tower at its own size (65 536 x 416 -> 400 -> 400 -> 400 -> 1): against float64 on a row sample, 5 reruns
    bitwise equal, and the cached weight image follows an in-place weight update."""
    from dir_amd import ops
    gen = torch.Generator(device="cuda").manual_seed(3)
    M, dims = 65536, [416, 400, 400, 400]
    x = torch.randn((M, dims[0]), generator=gen, device="cuda") * 0.25
    Ws = [torch.randn((dims[i + 1], dims[i]), generator=gen, device="cuda") / dims[i] ** 0.5 for i in range(3)]
    bs = [torch.randn((dims[i + 1],), generator=gen, device="cuda") * 0.1 for i in range(3)]
    hw, hb = torch.randn((1, 400), generator=gen, device="cuda") / 20, torch.full((1,), 0.1, device="cuda")
    got = ops.tower(x, Ws, bs, head=(hw, hb))
    for _ in range(5):
        assert torch.equal(ops.tower(x, Ws, bs, head=(hw, hb)), got)
    rows = torch.arange(0, M, 257, device="cuda")
    h = x[rows].double()
    for W, b in zip(Ws, bs):
        h = torch.relu(h @ W.double().t() + b.double())
    ref = (h @ hw.double().t() + hb.double()).cpu().numpy()
    assert _scaled_err(got[rows].cpu().numpy(), ref) <= 1e-5
    Ws[1].mul_(0.5)                                        # in place: tensor._version moves, the image is re-packed
    h = x[rows].double()
    for W, b in zip(Ws, bs):
        h = torch.relu(h @ W.double().t() + b.double())
    ref2 = (h @ hw.double().t() + hb.double()).cpu().numpy()
    assert _scaled_err(ops.tower(x, Ws, bs, head=(hw, hb))[rows].cpu().numpy(), ref2) <= 1e-5


def test_tower_refuses_uncovered_shapes(built_lib):
    from dir_amd import ops
    x = torch.zeros((64, 420), device="cuda")
    assert not ops.tower_covers(x, [torch.zeros((400, 420), device="cuda")])                 # wider than 416
    assert not ops.tower_covers(torch.zeros((64, 16), device="cuda"), [torch.zeros((6, 16), device="cuda")])   # width not a multiple of 4
    with pytest.raises(ValueError):
        ops.tower(x, [torch.zeros((400, 420), device="cuda")])
    assert ops.tower(torch.zeros((0, 16), device="cuda"), [torch.zeros((8, 16), device="cuda")]).shape == (0, 8)


def test_deepfm_inference_takes_the_fused_tower_and_packed_rows(built_lib, oracle):
    """DeepFM.forward under no_grad at a batch the fused path covers: packed serving rows (default layout) + the tower kernel with the FM
    and first-order logits added in its epilogue == the reference-layout, layer-by-layer forward within 1e-5, and the float64 graph."""
    import os
    from dir_amd.deepfm import DeepFM
    from dir_amd import feature_column as fc, ops
    torch.manual_seed(11)
    B, F, K, V = 8192, 26, 16, 5000
    cats = [fc.categorical_column_with_identity("C%d" % i, V) for i in range(F)]
    model = DeepFM(linear_feature_columns=cats, dnn_feature_columns=[fc.embedding_column(c, K) for c in cats],
                   dnn_hidden_units=[400, 400, 400], fm_embedding_size=K).cuda()
    with torch.no_grad():
        for p in model.linear_weights:
            p.normal_(0, 0.05)
    ids = torch.randint(0, V, (B, F), device="cuda")
    feats = {"C%d" % i: ids[:, i].contiguous() for i in range(F)}
    with torch.no_grad():
        fused = model(feats)
        assert model._packed is not None                                         # the default inference layout was built
        os.environ["DIR_SERVING_LAYOUT"] = "reference"
        old = ops.TOWER
        ops.TOWER = "0"
        try:
            plain = model(feats)
        finally:
            ops.TOWER = old
            del os.environ["DIR_SERVING_LAYOUT"]
    assert _scaled_err(fused.cpu().numpy(), plain.double().cpu().numpy()) <= 2e-5
    # float64 graph (fm_logit_fn + dnn_logit_fn + linear, deepFM.py:217-223)
    E = torch.stack([model.embedding_weights[f].double()[ids[:, f]] for f in range(F)], 1)      # [B, F, K]
    fm = 0.5 * ((E.sum(1) ** 2) - (E ** 2).sum(1)).sum(1, keepdim=True)
    h = E.reshape(B, F * K)
    for lin in model.hidden:
        h = torch.relu(h @ lin.weight.double().t() + lin.bias.double())
    dnn = h @ model.logits_layer.weight.double().t() + model.logits_layer.bias.double()
    lin = sum(model.linear_weights[f].double().reshape(-1)[ids[:, f]] for f in range(F)).reshape(B, 1) + model.linear_bias.double()
    assert _scaled_err(fused.cpu().numpy(), (fm + dnn + lin).detach().cpu().numpy()) <= 1e-5
    # training changes a table: the packed copy is refreshed on the next inference forward
    with torch.no_grad():
        model.embedding_weights[3].add_(0.01)
        after = model(feats)
    assert not torch.equal(after, fused)


@pytest.mark.parametrize("B,F,hidden,bn", [(8192, 26, [400, 400, 400], False), (4101, 13, [256, 128], False), (5000, 26, [400, 400], True)])
def test_deepfm_one_launch_inference_is_bitwise_the_two_launch_path(built_lib, B, F, hidden, bn):
    """dir_deepfm_tower_bf16x3_f32 (the lookups, the FM term and the first-order term inside the tower kernel) against the packed gather
    followed by the tower with both logits as addends, through ops and through the module: bit for bit.  Ids outside the vocabulary on
    both sides (pruned: zero rows), a strided [F, B]-major id matrix, a partial last tile."""
    from dir_amd.deepfm import DeepFM
    from dir_amd import feature_column as fc, ops
    torch.manual_seed(5 + F)
    K, V = 16, 3000
    cats = [fc.categorical_column_with_identity("C%d" % i, V) for i in range(F)]
    model = DeepFM(linear_feature_columns=cats, dnn_feature_columns=[fc.embedding_column(c, K) for c in cats], dnn_hidden_units=hidden,
                   fm_embedding_size=K, batch_norm=bn).cuda().eval()
    with torch.no_grad():
        for p in model.linear_weights:
            p.normal_(0, 0.05)
        model.linear_bias.fill_(0.125)
    ids = torch.randint(0, V, (B, F), device="cuda")
    ids[torch.rand((B, F), device="cuda") < 0.02] = -1
    ids[torch.rand((B, F), device="cuda") < 0.02] = V + 7
    ids_fb = ids.t().contiguous().t()                                             # same values, strides (1, B)
    assert ids_fb.stride() == (1, B)
    with torch.no_grad():
        model.forward_ids(ids, ids)                                               # builds the packed serving rows
        pt = model._packed
        ws, bs = [l.weight for l in model.hidden], [l.bias for l in model.hidden]
        assert ops.tower_gather_covers(pt, ws)
        head = (model.logits_layer.weight, model.logits_layer.bias)
        emb, fm, lin = ops.gather_fm_linear(pt, ids, bias=model.linear_bias.data)
        if not bn:
            two = ops.tower(emb, ws, bs, head=head, adds=(fm, lin))
            one = ops.tower(None, ws, bs, head=head, gather=(pt, ids, model.linear_bias.data))
            assert torch.equal(one, two)
            assert torch.equal(ops.tower(None, ws, bs, head=head, gather=(pt, ids_fb, model.linear_bias.data)), two)
        # through the module: the default path is the one-launch kernel; DIR_TOWER_GATHER=0's path is the two-launch one
        fused = model.forward_ids(ids, ids)
        old = ops.TOWER_GATHER
        ops.TOWER_GATHER = "0"
        try:
            plain = model.forward_ids(ids, ids)
        finally:
            ops.TOWER_GATHER = old
        assert torch.equal(fused, plain)
        ok = ids.clamp(0, V - 1)                                                  # (the features-dict path validates ids, as the reference's columns do)
        feats = {"C%d" % i: ok[:, i].contiguous() for i in range(F)}
        assert torch.equal(model(feats), model.forward_ids(ok, ok))
        # reruns are bitwise equal
        assert torch.equal(model.forward_ids(ids, ids), fused)
    # the C entry refuses what the kernel does not cover
    from dir_amd._lib import DirError
    with pytest.raises(ValueError):
        ops.tower(None, ws, bs, gather=(pt, ids, None))                           # no head: the FM term has nothing to join


def test_esmm_inference_towers_do_their_own_lookups(built_lib):
    """ESMM (ESMM.py:62-78,130-147) under no_grad at a batch the kernel covers: each tower's lookups, hidden layers and logit layer in one
    launch (dir_deepfm_tower_bf16x3_f32 with want_fm = 0 on the [vocab, 16] tables) against the gather + layer-by-layer path: the same
    1e-5-class bar as the other bf16x3 / fp32 comparisons, bitwise equal reruns, and the no-head form of the entry against the plain tower."""
    from dir_amd.esmm import ESMM
    from dir_amd import feature_column as fc, ops
    torch.manual_seed(21)
    B, F, K, V = 6000, 26, 16, 4000
    cols = [fc.embedding_column(fc.categorical_column_with_identity("C%02d" % i, V), K) for i in range(F)]
    model = ESMM(columns=cols, dnn_hidden_units=[360, 200, 80]).cuda().eval()
    ids = torch.randint(0, V, (B, F), device="cuda")
    feats = {"C%02d" % i: ids[:, i].contiguous() for i in range(F)}
    with torch.no_grad():
        src = model.ctr_model.input_layer.onehot_source(feats)
        assert src is not None and src[0].F == F and torch.equal(src[1], ids)
        fused = model(feats)
        old = ops.TOWER_GATHER
        ops.TOWER_GATHER = "0"
        try:
            plain = model(feats)
        finally:
            ops.TOWER_GATHER = old
        for k in ("ctr_logits", "cvr_logits", "ctcvr_logits"):
            assert _scaled_err(fused[k].cpu().numpy(), plain[k].double().cpu().numpy()) <= 2e-5, k
            assert torch.equal(model(feats)[k], fused[k])
        # the entry without a head: the last activation, bit for bit the plain tower on the gathered rows
        ts = src[0]
        ws, bs = [l.weight for l in model.ctr_model.hidden], [l.bias for l in model.ctr_model.hidden]
        emb = ops.embedding_bag(ts, ids)
        assert torch.equal(ops.tower(None, ws, bs, gather=(ts, ids, None, False)), ops.tower(emb, ws, bs))


@pytest.mark.parametrize("M,Kd,N,relu,affine", [(1, 4, 4, False, False), (300, 416, 400, True, False), (20000, 416, 400, True, True), (13000, 3328, 128, False, False),
                                                (700, 128, 200, True, False), (257, 36, 208, False, True)])
def test_dense_f16x2_matches_float64(built_lib, M, Kd, N, relu, affine):
    """dir_dense_f16x2_f32 (two fp16 pieces per operand, three products; csrc/dense_bf3.hip) against float64 at the 1e-5 bar of the other
    arithmetics, on embedding-scale operands; reruns bitwise equal; "auto_bounded" is that kernel wherever "auto" picks bf16x3."""
    from dir_amd import ops
    g = torch.Generator().manual_seed(M + Kd)
    x = (torch.randn(M, Kd, generator=g) * 0.3).cuda()
    W = (torch.randn(N, Kd, generator=g) / Kd ** 0.5).cuda()
    b = (torch.randn(N, generator=g) * 0.1).cuda()
    ps = (torch.rand(N, generator=g) + 0.5).cuda() if affine else None
    psh = (torch.randn(N, generator=g) * 0.1).cuda() if affine else None
    got = ops.dense(x, W, b, relu=relu, post_scale=ps, post_shift=psh, arith="f16x2")
    ref = x.double() @ W.double().t() + b.double()
    if relu:
        ref = ref.clamp_min(0)
    if affine:
        ref = ref * ps.double() + psh.double()
    err = float(((got.double() - ref).abs() / (1 + ref.abs())).max())
    assert err <= 1e-5, err
    assert torch.equal(ops.dense(x, W, b, relu=relu, post_scale=ps, post_shift=psh, arith="f16x2"), got)
    if ops.dense_auto_arith(M, Kd, N) == "bf16x3" and ops.DENSE_BOUNDED_SPLIT == "f16x2":
        assert torch.equal(ops.dense(x, W, b, relu=relu, post_scale=ps, post_shift=psh, arith="auto_bounded"), got)
    with pytest.raises(ValueError):
        ops.dense(x, W, arith="f16x3")


def test_tower_splits_agree_and_general_inputs_stay_on_bf16x3(built_lib):
    """The two split arithmetics of the fused tower against float64 and against each other; dense.tower_infer routes a general input
    (no embedding_input hint, no gather) to bf16 x 3 -- raw numeric columns may exceed fp16's range (DeepCrossNetwork/train.py's
    capital_gain reaches 99 999) -- and an embedding input to fp16 x 2."""
    from dir_amd import ops
    from dir_amd import dense as D
    torch.manual_seed(1)
    M = 8192
    lins = [torch.nn.Linear(416, 400).cuda(), torch.nn.Linear(400, 400).cuda()]
    head = torch.nn.Linear(400, 1).cuda()
    x = (torch.randn(M, 416, device="cuda") * 0.3)
    ws, bs = [l.weight.data for l in lins], [l.bias.data for l in lins]
    ref = x.double()
    for l in lins:
        ref = (ref @ l.weight.double().t() + l.bias.double()).clamp_min(0)
    ref = ref @ head.weight.double().t() + head.bias.double()
    outs = {}
    for sp in ("f16x2", "bf16x3"):
        outs[sp] = ops.tower(x, ws, bs, head=(head.weight.data, head.bias.data), split=sp)
        assert float(((outs[sp].double() - ref).abs() / (1 + ref.abs())).max()) <= 1e-5, sp
    assert not torch.equal(outs["f16x2"], outs["bf16x3"])           # two arithmetics, two roundings
    with torch.no_grad():
        general = D.tower_infer(lins, x, torch.relu, head=head)
        emb = D.tower_infer(lins, x, torch.relu, head=head, embedding_input=True)
    assert torch.equal(general, outs["bf16x3"]) and torch.equal(emb, outs[ops.TOWER_SPLIT])
    big = x.clone()
    big[:, 7] = 99999.0                                              # a raw numeric column beyond fp16's 65 504
    with torch.no_grad():
        gb = D.tower_infer(lins, big, torch.relu, head=head)
    assert bool(torch.isfinite(gb).all())


@pytest.mark.parametrize("B,N,strided", [(65536, 400, False), (8193, 429, False), (5000, 80, False), (4097, 1024, True), (3, 7, False), (0, 16, False),
                                         (777, 4096, False), (1000, 384, True)])
def test_units1_forward(built_lib, B, N, strided):
    """dir_units1_f32 (the logit heads in training, DCN's dense(1) over d = 429, xDeepFM's CIN output layer): x . w + bias against float64,
    bitwise equal run to run, through a column view of a wider tensor (row stride != N), with and without a bias."""
    from dir_amd import ops
    gen = torch.Generator(device="cuda").manual_seed(B + N)
    wide = torch.randn((B, N + (12 if strided else 0)), generator=gen, device="cuda")
    x = wide[:, 4:4 + N] if strided else wide
    w = torch.randn((1, N), generator=gen, device="cuda") / N ** 0.5
    b = torch.randn(1, generator=gen, device="cuda")
    for bias in (b, None):
        y = ops.units1(x, w, bias)
        assert tuple(y.shape) == (B, 1)
        ref = x.double() @ w.double().t() + (bias.double() if bias is not None else 0.0)
        if B:
            assert float((y.double() - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))
        assert torch.equal(y, ops.units1(x, w, bias))
    with pytest.raises(ValueError):
        ops.units1(x, w[:, :-1], b)


def test_narrow_weight_gradients_run_the_hip_kernel(built_lib):
    """dense._tn_matmul / _wb_grads route narrow outputs (an 80-wide last tower layer, ESMM.py:130-147) to dir_dense_dw_bf16x3_f32 instead
    of the library's batched GEMM + sum: against float64 and bit for bit the kernel called by name."""
    from dir_amd import ops
    from dir_amd import dense as D
    gen = torch.Generator(device="cuda").manual_seed(3)
    M, N, K = 16384, 80, 200
    g = torch.randn((M, N), generator=gen, device="cuda") * 1e-3
    x = torch.randn((M, K), generator=gen, device="cuda")
    assert ops.dense_dw_auto_arith(M, N, K) == "bf16x3"
    gw, gb = D._wb_grads(g, x, True, True)
    kw, kb = ops.dense_dw(g, x, arith="bf16x3", want_bias=True)
    assert torch.equal(gw, kw) and torch.equal(gb, kb) and torch.equal(D._tn_matmul(g, x), kw)
    ref = g.double().t() @ x.double()
    assert float((gw.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert float((gb.double() - g.double().sum(0)).abs().max()) <= 1e-5 * float(g.double().sum(0).abs().max())


def test_tower_default_split_checks_the_fp16_range_of_weights_and_packed_rows(built_lib):
    """split=None measures what it can without a per-call sync: a weight (per image build) or a packed serving row (per PackedTables)
    at or above ops.F16_RANGE_GUARD routes the launch to bf16 x 3 -- bit for bit the explicit split="bf16x3" result, finite and within
    the bf16 x 3 bar of float64 -- and back to fp16 x 2 once the value is gone."""
    from dir_amd import ops
    if ops.TOWER_SPLIT != "f16x2":
        pytest.skip("DIR_TOWER_SPLIT overrides the default")
    torch.manual_seed(5)
    M = 4096
    W1 = torch.randn(400, 416, device="cuda") / 20
    W2 = torch.randn(400, 400, device="cuda") / 20
    hw, hb = torch.randn(400, device="cuda") / 20, torch.zeros(1, device="cuda")
    x = torch.randn(M, 416, device="cuda") * 0.3
    base = ops.tower(x, [W1, W2], head=(hw, hb))
    assert torch.equal(base, ops.tower(x, [W1, W2], head=(hw, hb), split="f16x2"))
    W1[3, 5] = 1.0e5
    x[:, 5] = 1e-3                                                    # keeps the activations themselves in range
    got = ops.tower(x, [W1, W2], head=(hw, hb))
    assert torch.equal(got, ops.tower(x, [W1, W2], head=(hw, hb), split="bf16x3")) and bool(torch.isfinite(got).all())
    ref = ((x.double() @ W1.double().t()).clamp_min(0) @ W2.double().t()).clamp_min(0) @ hw.double().view(-1, 1) + hb.double()
    assert float(((got.double() - ref).abs() / (1 + ref.abs())).max()) <= 1e-5
    W1[3, 5] = 0.5
    assert torch.equal(ops.tower(x, [W1, W2], head=(hw, hb)), ops.tower(x, [W1, W2], head=(hw, hb), split="f16x2"))
    # packed serving rows: one table value past the range
    F, K, V = 26, 16, 1000
    tabs = [torch.randn(V, K, device="cuda") * 0.1 for _ in range(F)]
    lins = [torch.randn(V, device="cuda") * 0.1 for _ in range(F)]
    ids = torch.randint(0, V, (M, F), device="cuda")
    pt = ops.PackedTables(tabs, lins)
    assert pt.absmax() < ops.F16_RANGE_GUARD
    a = ops.tower(None, [W1, W2], head=(hw, hb), gather=(pt, ids, None))
    assert torch.equal(a, ops.tower(None, [W1, W2], head=(hw, hb), gather=(pt, ids, None), split="f16x2"))
    tabs[2][17, 3] = 7.0e4
    ids[:8, 2] = 17
    pt2 = ops.PackedTables(tabs, lins)
    assert pt2.absmax() == 7.0e4
    b = ops.tower(None, [W1, W2], head=(hw, hb), gather=(pt2, ids, None))
    wide = ops.tower(None, [W1, W2], head=(hw, hb), gather=(pt2, ids, None), split="bf16x3")
    assert bool(torch.isfinite(b).all())
    if ops.TOWER_KERNEL == "cs":      # round 6: tower_cs_k scales every stored row, the looked-up rows' magnitude no longer routes -- and need not
        assert torch.equal(b, ops.tower(None, [W1, W2], head=(hw, hb), gather=(pt2, ids, None), split="f16x2"))
        assert float(((b.double() - wide.double()).abs() / (1 + wide.double().abs())).max()) <= 1e-5
    else:
        assert torch.equal(b, wide)


@pytest.mark.parametrize("M,N,K,gate", [(12800, 400, 416, True), (12801, 400, 400, True), (13000, 208, 128, False), (12544, 1024, 432, True),
                                        (12290, 64, 200, False)])
def test_dense_backward_on_scaled_fp16x2(built_lib, M, N, K, gate):
    """The fp16 x 2 backward kernels of a dense layer (round 4): dL/dx = g W on dir_dense_f16x2_rows_f32 (every row of g times a power of
    two from dir_row_absmax_bits_f32, exact) and dL/dW = g^T x on dir_dense_dw_f16x2_f32 (g times ONE power of two).  Against float64 at the
    bf16 x 3 kernels' bars; a gradient whose rows differ by powers of two over 60 binades gives the same dL/dx bits times those powers,
    a 2^-40 times smaller g the same dL/dW bits times 2^-40; the bias gradient is the bf16 x 3 kernel's bit for bit."""
    from dir_amd import ops
    gen = torch.Generator(device="cuda").manual_seed(M + N)
    g = torch.randn((M, N), generator=gen, device="cuda") * 1e-4 * (torch.rand((M, 1), generator=gen, device="cuda") * 3 + 0.01)
    x = torch.randn((M, K), generator=gen, device="cuda") * 0.3
    W = torch.randn((N, K), generator=gen, device="cuda") / K ** 0.5           # nn.Linear layout [out, in]
    gt = (torch.rand((M, K), generator=gen, device="cuda") > 0.4).float() if gate else None
    rb, ab = ops.grad_bits(g)
    assert torch.equal(rb, g.abs().amax(dim=1).view(torch.int32)) and int(ab) == int(g.abs().max().view(torch.int32))
    # dL/dx
    ref = g.double() @ W.double()
    if gate:
        ref = ref * gt.double()
        got = ops.dense_gated(g, W.t(), gt, row_bits=rb)
        old = ops.dense_gated(g, W.t(), gt)
    else:
        got = ops.dense(g, W.t(), row_bits=rb)
        old = ops.dense(g, W.t())
    scale = float(ref.abs().max())
    for y in (got, old):
        assert float(((y.double() - ref).abs() / (scale + ref.abs())).max()) <= 1e-5
    pw = torch.from_numpy(np.ldexp(1.0, np.random.default_rng(M).integers(-50, 10, size=(M, 1))).astype(np.float32)).cuda()
    g2 = g * pw
    rb2, _ = ops.grad_bits(g2)
    got2 = ops.dense_gated(g2, W.t(), gt, row_bits=rb2) if gate else ops.dense(g2, W.t(), row_bits=rb2)
    assert torch.equal(got2, got * pw)
    left = []                                   # the kernel's epilogue leaves its OUTPUT's row / tensor maxima (the next layer's scales): exact
    again = ops.dense_gated(g, W.t(), gt, row_bits=rb, bits_out=left) if gate else ops.dense(g, W.t(), row_bits=rb, bits_out=left)
    assert torch.equal(again, got)
    assert torch.equal(left[0][0], got.abs().amax(dim=1).view(torch.int32)) and int(left[0][1]) == int(got.abs().max().view(torch.int32))
    # the chain's first gradient: dir_units1_relu_backward_bits_f32 leaves upper bounds |g[r]| max|w| of its rows (tight when the largest
    # weight's unit is active), never below the row's true maximum
    yact = torch.relu(torch.randn((M, K), generator=gen, device="cuda"))
    g1 = torch.randn(M, generator=gen, device="cuda") * 1e-3
    w1 = torch.randn(K, generator=gen, device="cuda")
    dpre, dw1, db1, bits1 = ops.units1_relu_backward(g1, w1, yact, want_bits=True)
    plain = ops.units1_relu_backward(g1, w1, yact)
    assert torch.equal(dpre, plain[0]) and torch.equal(dw1, plain[1]) and torch.equal(db1, plain[2])
    true_rows = dpre.abs().amax(dim=1)
    bound = bits1[0].view(torch.float32)
    assert bool((bound >= true_rows).all()) and bool(torch.equal(bound, (g1 * w1.abs().max()).abs()))
    assert float(bits1[1].view(torch.float32)) == float(bound.max())
    # dL/dW, dL/db
    if ops.dense_dw_auto_arith(M, N, K) != "bf16x3":
        return
    refw = g.double().t() @ x.double()
    dW, db = ops.dense_dw(g, x, want_bias=True, g_bits=ab)                      # "auto" + g_bits: the fp16 x 2 kernel
    bW, bb = ops.dense_dw(g, x, want_bias=True, arith="bf16x3")
    mag = float(refw.abs().max())
    for w in (dW, bW):
        assert float(((w.double() - refw).abs() / (mag + refw.abs())).max()) <= 1e-5
    assert torch.equal(db, bb)
    assert torch.equal(ops.dense_dw(g, x, arith="f16x2", g_bits=ab), dW)
    s = 2.0 ** -40
    _, ab2 = ops.grad_bits(g * s)
    assert torch.equal(ops.dense_dw(g * s, x, arith="f16x2", g_bits=ab2), dW * s)
    assert not torch.equal(dW, bW)                                              # (two arithmetics: not the same bits)


@pytest.mark.parametrize("M,Kd,N", [(12800, 432, 1024), (13001, 360, 200), (12288, 64, 208)])
def test_dense_general_inputs_on_row_scaled_fp16x2(built_lib, monkeypatch, M, Kd, N):
    """dense(arith="auto") on a GENERAL input -- a raw numeric column of 99 999 (DeepCrossNetwork/train.py's capital_gain) beside embedding-scale
    columns and 1e-6-scale ones, all-zero rows, rows that differ by 60 binades -- runs the row-scaled fp16 x 2 kernel behind one max pass over x
    (ops.DENSE_GENERAL_SPLIT): finite, within 1e-5 of float64 scaled by each row's own magnitude (the bf16 x 3 kernel's bar), bitwise
    reproducible, and scaling a row by a power of two scales its outputs by exactly that power."""
    from dir_amd import ops
    gen = torch.Generator(device="cuda").manual_seed(M + Kd)
    x = torch.randn((M, Kd), generator=gen, device="cuda") * 0.3
    x[:, 5] = torch.rand(M, generator=gen, device="cuda") * 99999.0
    x[:, 7] *= 1e-6
    x[::97] = 0.0
    W = torch.randn((N, Kd), generator=gen, device="cuda") / Kd ** 0.5
    b = torch.randn(N, generator=gen, device="cuda") * 0.1
    assert ops.DENSE_GENERAL_SPLIT == "f16x2_rows" and ops.dense_auto_arith(M, Kd, N) == "bf16x3"
    got = ops.dense(x, W, b, relu=False)
    assert bool(torch.isfinite(got).all())
    ref = x.double() @ W.double().t() + b.double()
    mag = (x.double().abs() @ W.double().abs().t()) + 1.0                       # what fp32 rounding of the row's products is relative to
    assert float(((got.double() - ref).abs() / mag).max()) <= 1e-5
    b3 = ops.dense(x, W, b, relu=False, arith="bf16x3")
    assert float(((b3.double() - ref).abs() / mag).max()) <= 1e-5 and not torch.equal(b3, got)
    assert torch.equal(ops.dense(x, W, b, relu=False), got)
    pw = torch.from_numpy(np.ldexp(1.0, np.random.default_rng(M).integers(-40, 20, size=(M, 1))).astype(np.float32)).cuda()
    lin = ops.dense(x, W)
    assert torch.equal(ops.dense(x * pw, W), lin * pw)
    monkeypatch.setattr(ops, "DENSE_GENERAL_SPLIT", "bf16x3")
    assert torch.equal(ops.dense(x, W, b, relu=False), b3)


@pytest.mark.parametrize("M,Kd,N", [(1, 4, 1), (100, 416, 400), (256, 400, 400), (37, 432, 1024), (256, 1024, 1024), (65, 1024, 520), (17, 20, 33),
                                     (512, 360, 200), (300, 200, 80), (16, 16, 16), (255, 52, 7)])
def test_dense_small_batches(built_lib, M, Kd, N):
    """dir_dense_small_f32 (round 5: the layers at the reference's batch sizes, DeepCrossNetwork/train.py:16-17): one wave per 16 x 16 tile, the
    workgroup's four waves splitting the reduction -- row / column / k tails, the two-row-tile form of the wide layers, bias, ReLU and the
    folded inference batch norm, against float64; rerun bitwise equal; and ops.dense routes to it by itself."""
    from dir_amd import ops, _lib
    import ctypes
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(M * 7 + N)
    x = torch.randn((M, Kd), generator=g, device=dev) * 0.5
    W = torch.randn((N, Kd), generator=g, device=dev) * 0.1
    b = torch.randn((N,), generator=g, device=dev) * 0.2
    ps = torch.rand((N,), generator=g, device=dev) + 0.5
    sh = torch.randn((N,), generator=g, device=dev) * 0.1
    lib = _lib.load()
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None      # noqa: E731
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for relu, bias, post in ((True, b, None), (False, None, None), (True, b, (ps, sh))):
        y = torch.full((M, N + 3), 7.0, device=dev)                  # a wider output buffer: the row stride is honoured, the pad untouched
        _lib.check(lib.dir_dense_small_f32(p(x), x.stride(0), p(W), W.stride(0), p(bias), 1 if relu else 0, p(post[0]) if post else None,
                                           p(post[1]) if post else None, M, Kd, N, p(y), y.stride(0), st))
        ref = x.double() @ W.double().t()
        if bias is not None:
            ref = ref + bias.double()
        if relu:
            ref = torch.relu(ref)
        if post:
            ref = ref * post[0].double() + post[1].double()
        err = float(((y[:, :N].double() - ref).abs() / (1 + ref.abs())).max())
        assert err <= 1e-5, err
        assert bool((y[:, N:] == 7.0).all())
        y2 = torch.empty((M, N), device=dev)
        _lib.check(lib.dir_dense_small_f32(p(x), x.stride(0), p(W), W.stride(0), p(bias), 1 if relu else 0, p(post[0]) if post else None,
                                           p(post[1]) if post else None, M, Kd, N, p(y2), y2.stride(0), st))
        assert torch.equal(y2, y[:, :N])
    if ops.dense_small_covers(M, Kd, N) and N >= 16:
        assert torch.equal(ops.dense(x, W, b, relu=True), torch.relu(y2) if False else ops.dense(x, W, b, relu=True, arith="f32"))


@pytest.mark.parametrize("gather", [False, True])
def test_tower_interior_activations_of_any_magnitude(built_lib, gather):
    """VERDICT r5 item 3 (the range hole): weights and inputs inside fp16's range whose INTERIOR activations are not -- all-positive operands
    grow from O(1) to ~2e3 behind layer 1 and ~1e6 behind layer 2, past fp16's 65 504.  tower_cs_k scales every stored row by a power of two
    (DIR_TOWER_RS, default on): the result stays within the scale-invariant 1e-5 of float64; tower_bf3_k (and DIR_TOWER_RS=0) overflow there."""
    import os
    from dir_amd import ops
    if ops.TOWER_KERNEL != "cs":
        pytest.skip("tower_bf3_k splits interior activations unscaled (DESIGN.md section 7, item 9): this test is tower_cs_k's")
    rng = np.random.default_rng(3)
    M, F, K = 1500, 26, 16
    Kd, Ns = F * K, [400, 400, 64]
    tabs = [np.abs(rng.standard_normal((50, K))).astype(np.float32) * 4 for _ in range(F)]
    ids = rng.integers(0, 50, size=(M, F)).astype(np.int64)
    x = np.concatenate([tabs[f][ids[:, f]] for f in range(F)], axis=1)
    dims = [Kd] + Ns
    Ws = [np.abs(rng.standard_normal((dims[i + 1], dims[i]))).astype(np.float32) * 2 for i in range(3)]
    bs = [(rng.standard_normal(n) * 0.1).astype(np.float32) for n in Ns]
    hw = (rng.standard_normal(Ns[-1]) / np.sqrt(Ns[-1])).astype(np.float32)
    hb = np.array([0.5], np.float32)
    d = lambda a: torch.from_numpy(a).cuda()      # noqa: E731
    ref = _ref64(x, Ws, bs, [True] * 3, [None] * 3, [None] * 3, (hw, hb), [])
    assert np.abs(ref).max() > 1e6                                       # (the activations really are out of fp16's range)
    if gather:
        ts = ops.TableSet([d(t) for t in tabs])
        got = ops.tower(None, [d(w) for w in Ws], [d(b) for b in bs], head=(d(hw), d(hb)), gather=(ts, d(ids), None, False), split="f16x2")
    else:
        got = ops.tower(d(x), [d(w) for w in Ws], [d(b) for b in bs], head=(d(hw), d(hb)), split="f16x2")
    g = got.cpu().numpy().astype(np.float64)
    assert np.isfinite(g).all()
    assert np.abs(g - ref).max() <= 1e-5 * (np.abs(ref).max())
    os.environ["DIR_TOWER_RS"] = "0"
    try:
        raw = ops.tower(d(x), [d(w) for w in Ws], [d(b) for b in bs], head=(d(hw), d(hb)), split="f16x2").cpu().numpy()
    finally:
        del os.environ["DIR_TOWER_RS"]
    assert not np.isfinite(raw).all() or np.abs(raw - ref).max() > 1e-3 * np.abs(ref).max()      # what the scaling is for


@pytest.mark.parametrize("affine", [False, True])
def test_tower_rows_of_very_different_magnitudes_in_one_workgroup(built_lib, affine):
    """Row scaling is per batch row: neighbours whose inputs differ by 2^40 (some far below fp16's subnormals, some whose activations pass
    65 504) in the same 64-row tile, each held to 1e-5 of ITS OWN largest output -- a per-tensor scale could not do that.  With the affine
    (|scale| up to 500, shifts of either sign) the bound's constants (max |scale|, max |scale x bias| + max |shift|) are exercised too."""
    from dir_amd import ops
    if ops.TOWER_KERNEL != "cs":
        pytest.skip("tower_cs_k's row scaling")
    rng = np.random.default_rng(11)
    M, Kd, Ns = 700, 96, [200, 120, 48]
    x = rng.standard_normal((M, Kd)).astype(np.float32)
    expo = rng.integers(-30, 11, size=M)
    x = (x * np.exp2(expo)[:, None]).astype(np.float32)
    x[5] = 0.0                                                # an all-zero row
    dims = [Kd] + Ns
    Ws = [(rng.standard_normal((dims[i + 1], dims[i])) * 1.5).astype(np.float32) for i in range(3)]
    if affine:
        bs = [(rng.standard_normal(n) * 0.01).astype(np.float32) for n in Ns]
        scs = [(rng.standard_normal(n) * 200).astype(np.float32) for n in Ns]
        shs = [(rng.standard_normal(n) * 3).astype(np.float32) for n in Ns]
    else:
        bs, scs, shs = [None] * 3, [None] * 3, [None] * 3     # homogeneous: every row's outputs scale with its input
    d = lambda a: None if a is None else torch.from_numpy(a).cuda()      # noqa: E731
    ref = _ref64(x, Ws, bs, [True] * 3, scs, shs)
    got = ops.tower(d(x), [d(w) for w in Ws], None if bs[0] is None else [d(b) for b in bs], relu=True,
                    post_scale=None if not affine else [d(s) for s in scs], post_shift=None if not affine else [d(s) for s in shs],
                    split="f16x2").cpu().numpy().astype(np.float64)
    assert np.isfinite(got).all()
    rowmax = np.abs(ref).max(axis=1, keepdims=True)
    err = np.abs(got - ref) / np.maximum(rowmax, 1e-300)
    assert float(err[rowmax[:, 0] > 0].max()) <= 1e-5
    if not affine:
        assert not got[5].any()
        assert rowmax.max() / rowmax[rowmax > 0].min() > 1e9  # (the rows really span many orders of magnitude)


@pytest.mark.parametrize("mag", [1e-7, 3e4])
def test_one_launch_tower_takes_tables_of_any_magnitude_on_the_split_kernel(built_lib, mag):
    """split=None used to send packed serving rows outside [2^-6, 2^15) to the bf16 x 3 kernel (the unscaled fp16 pieces lose their second
    piece below, overflow above); tower_cs_k's row scaling makes the looked-up rows' magnitude irrelevant, so the guard now only reads the
    weights.  Rows of magnitude 1e-7 and 3e4 (x 26 slots: FM terms up to ~1e10) through the one-launch DeepFM form, against float64."""
    from dir_amd import ops
    if ops.TOWER_KERNEL != "cs":
        pytest.skip("tower_cs_k's row scaling")
    rng = np.random.default_rng(5)
    M, F, K = 900, 26, 16
    Kd, Ns = F * K, [200, 80]
    tabs = [(rng.standard_normal((40, K)) * mag).astype(np.float32) for _ in range(F)]
    ids = rng.integers(0, 40, size=(M, F)).astype(np.int64)
    x = np.concatenate([tabs[f][ids[:, f]] for f in range(F)], axis=1)
    dims = [Kd] + Ns
    Ws = [(rng.standard_normal((dims[i + 1], dims[i])) / np.sqrt(dims[i])).astype(np.float32) for i in range(2)]
    hw = (rng.standard_normal(Ns[-1]) / np.sqrt(Ns[-1])).astype(np.float32)
    hb = np.array([0.0], np.float32)
    d = lambda a: torch.from_numpy(a).cuda()      # noqa: E731
    ref = _ref64(x, Ws, [None, None], [True] * 2, [None] * 2, [None] * 2, (hw, hb), [])
    ts = ops.TableSet([d(t) for t in tabs])
    assert ops._tower_split_for([d(w) for w in Ws], ts) == "f16x2"
    got = ops.tower(None, [d(w) for w in Ws], None, head=(d(hw), d(hb)), gather=(ts, d(ids), None, False)).cpu().numpy().astype(np.float64)
    assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()


def test_row_scaling_changes_nothing_for_ordinary_magnitudes(built_lib):
    """Rows whose maxima (bounds) lie inside the no-scale windows keep s = 0 and take the epilogue without the multiplications: with inputs
    and weights of ordinary magnitude tower_cs_k's result with DIR_TOWER_RS on and off is the same, bit for bit (the DeepFM tower's shape,
    plain and with an affine; a batch with a partial last tile)."""
    import os
    from dir_amd import ops
    if ops.TOWER_KERNEL != "cs":
        pytest.skip("tower_cs_k's row scaling")
    gen = torch.Generator(device="cuda").manual_seed(9)
    M, dims = 5000, [416, 400, 400, 400]
    x = torch.randn((M, dims[0]), generator=gen, device="cuda") * 0.25
    Ws = [torch.randn((dims[i + 1], dims[i]), generator=gen, device="cuda") / dims[i] ** 0.5 for i in range(3)]
    bs = [torch.randn((dims[i + 1],), generator=gen, device="cuda") * 0.1 for i in range(3)]
    sc = [1 + 0.2 * torch.randn((dims[i + 1],), generator=gen, device="cuda") for i in range(3)]
    sh = [0.1 * torch.randn((dims[i + 1],), generator=gen, device="cuda") for i in range(3)]
    hw, hb = torch.randn((1, 400), generator=gen, device="cuda") / 20, torch.full((1,), 0.1, device="cuda")
    for kw in (dict(), dict(post_scale=sc, post_shift=sh)):
        on = ops.tower(x, Ws, bs, head=(hw, hb), split="f16x2", **kw)
        on_act = ops.tower(x, Ws, bs, split="f16x2", **kw)
        os.environ["DIR_TOWER_RS"] = "0"
        try:
            off = ops.tower(x, Ws, bs, head=(hw, hb), split="f16x2", **kw)
            off_act = ops.tower(x, Ws, bs, split="f16x2", **kw)
        finally:
            del os.environ["DIR_TOWER_RS"]
        assert torch.equal(on, off) and torch.equal(on_act, off_act)
