"""The DIN model (details-in-recommendation_amd/din.py; paper-derived: /root/reference/README.md:27 links arXiv:1706.06978, no reference
code) and the PReLU / Dice forms of its local activation unit (dir_din_attention_pool_act_f32) against the oracle restatements."""
import os

import numpy as np
import pytest
import torch

from oracle import np_ref as R

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def _close(got, ref, tol=1e-5):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    err = np.abs(got - ref) / (1 + np.abs(ref))
    assert float(err.max()) <= tol, "max scaled error %.3e" % float(err.max())


def _unit_case(rng, V, K, B, T, H1, H2):
    table = (rng.standard_normal((V, K)) * 0.3).astype(np.float32)
    hist = rng.integers(-1, V, size=(B, T)).astype(np.int64)
    hl = rng.integers(0, T + 1, size=B).astype(np.int32)
    cand = rng.integers(-1, V, size=B).astype(np.int64)
    W1 = (rng.standard_normal((4 * K, H1)) * 0.08).astype(np.float32)
    b1 = (rng.standard_normal(H1) * 0.1).astype(np.float32)
    W2 = (rng.standard_normal((H1, H2)) * 0.2).astype(np.float32)
    b2 = (rng.standard_normal(H2) * 0.1).astype(np.float32)
    W3 = (rng.standard_normal(H2) * 0.3).astype(np.float32)
    b3 = np.array([0.05], np.float32)
    ap = np.concatenate([rng.uniform(-0.2, 0.6, H1), rng.uniform(0.3, 3.0, H1), rng.standard_normal(H1) * 0.5,
                         rng.uniform(-0.2, 0.6, H2), rng.uniform(0.3, 3.0, H2), rng.standard_normal(H2) * 0.5]).astype(np.float32)
    return table, hist, hl, cand, (W1, b1, W2, b2, W3, b3), ap


@pytest.mark.parametrize("activation", ["prelu", "dice"])
@pytest.mark.parametrize("arith", ["f16x2", "bf16x3", "f32"])
@pytest.mark.parametrize("B,T,H1,H2,normalize", [(1, 1, 4, 4, False), (37, 50, 80, 40, False), (64, 64, 80, 48, True), (300, 17, 36, 8, True),
                                                 (129, 33, 64, 16, False), (2000, 50, 80, 40, True)])
def test_din_unit_prelu_dice_match_oracle(built_lib, oracle, activation, arith, B, T, H1, H2, normalize):
    """dir_din_attention_pool_act_f32 (din_wave_k<.., ACT>) vs the double-accumulating C oracle and the float64 NumPy restatement: masked
    positions, pruned ids, missing candidates, padded hidden widths; |err| <= 1e-5 (1 + |ref|) for the pooled vector AND the weights."""
    from dir_amd import ops
    rng = np.random.default_rng(B * 1000 + T * 10 + H1 + (7 if activation == "dice" else 0))
    table, hist, hl, cand, unit, ap = _unit_case(rng, 500, 64, B, T, H1, H2)
    os.environ["DIR_DIN_ARITH"] = arith
    try:
        cu = [torch.from_numpy(a).cuda() for a in (table, hist, hl, cand) + unit]
        out, scores = ops.din_attention_pool(*cu, normalize=normalize, want_scores=True, activation=activation, act_params=torch.from_numpy(ap).cuda())
        again = ops.din_attention_pool(*cu, normalize=normalize, activation=activation, act_params=torch.from_numpy(ap).cuda())
        torch.cuda.synchronize()
    finally:
        del os.environ["DIR_DIN_ARITH"]
    assert torch.equal(out, again)                                           # rerun: bitwise equal
    ref, ref_s = oracle.din_attention_pool(table, hist, hl, cand, *unit, normalize=normalize, acc64=True, activation=activation, act_params=ap)
    _close(_np(out), ref)
    _close(_np(scores), ref_s)
    if B <= 300:
        ref64, _ = R.din_attention_pool(table, hist, hl, cand, *unit, normalize=normalize, activation=activation, act_params=ap)
        _close(_np(out), ref64)


def test_din_unit_act_argument_errors(built_lib):
    from dir_amd import ops, _lib
    rng = np.random.default_rng(0)
    table, hist, hl, cand, unit, ap = _unit_case(rng, 50, 64, 4, 5, 8, 8)
    cu = [torch.from_numpy(a).cuda() for a in (table, hist, hl, cand) + unit]
    with pytest.raises(ValueError):
        ops.din_attention_pool(*cu, activation="prelu")                       # no parameters
    with pytest.raises(ValueError):
        ops.din_attention_pool(*cu, activation="gelu", act_params=torch.from_numpy(ap).cuda())
    t32, h32, l32, c32, u32, a32 = _unit_case(rng, 50, 32, 4, 5, 8, 8)         # K = 32: the PReLU / Dice unit covers K = 64 only
    with pytest.raises(_lib.DirError):
        ops.din_attention_pool(*[torch.from_numpy(a).cuda() for a in (t32, h32, l32, c32) + u32], activation="dice",
                               act_params=torch.from_numpy(a32).cuda())


def _randomize(model, rng):
    with torch.no_grad():
        for m in model.modules():
            if hasattr(m, "alpha"):
                m.alpha.copy_(torch.from_numpy(rng.uniform(-0.2, 0.6, m.alpha.numel()).astype(np.float32)))
            if hasattr(m, "moving_mean") and m.__class__.__name__ == "Dice":
                m.moving_mean.copy_(torch.from_numpy((rng.standard_normal(m.moving_mean.numel()) * 0.3).astype(np.float32)))
                m.moving_variance.copy_(torch.from_numpy(rng.uniform(0.2, 2.0, m.moving_variance.numel()).astype(np.float32)))
        a = model.attention
        a.b1.normal_(0, 0.05); a.b2.normal_(0, 0.05); a.b3.fill_(0.02)
        for lin in list(model.hidden) + [model.logits_layer]:
            lin.bias.normal_(0, 0.05)


def _act_params_np(mod):
    if mod.activation == "sigmoid":
        return None
    return _np(mod.act_params())


def _mlp_np(model):
    out = []
    for lin, act in zip(model.hidden, model.acts):
        kind = model.dnn_activation_fn
        params = None
        if kind == "prelu":
            params = (_np(act.alpha),)
        elif kind == "dice":
            sc, sh = act.scale_shift()
            params = (_np(act.alpha), _np(sc), _np(sh))
        out.append((_np(lin.weight), _np(lin.bias), kind, params))
    return out


@pytest.mark.parametrize("att_act,dnn_act,normalize,with_columns", [("sigmoid", "dice", False, True), ("prelu", "prelu", False, True),
                                                                     ("dice", "dice", True, False), ("dice", "relu", False, True),
                                                                     ("sigmoid", "sigmoid", True, False)])
def test_din_model_matches_oracle(built_lib, oracle, att_act, dnn_act, normalize, with_columns):
    """DIN(nn.Module): profile / context columns through InputLayer, behaviour sequence + candidate through the HIP unit, concat -> 200-80
    MLP -> logit; inference forward and predict() against oracle/np_ref.din_model_logits (float64, paper-derived)."""
    from dir_amd import feature_column as fc
    from dir_amd.din import DIN
    rng = np.random.default_rng(11)
    torch.manual_seed(5)
    V, K, B, T = 3000, 64, 257, 40
    cols = None
    if with_columns:
        cols = [fc.embedding_column(fc.categorical_column_with_identity("gender", 3), 4), fc.embedding_column(fc.categorical_column_with_identity("city", 200), 8),
                fc.numeric_column("age")]
    model = DIN(feature_columns=cols, item_vocab_size=V, embedding_dim=K, attention_activation=att_act, attention_normalize=normalize,
                dnn_hidden_units=(200, 80), dnn_activation_fn=dnn_act).cuda().eval()
    _randomize(model, rng)
    hist = rng.integers(-1, V, size=(B, T)).astype(np.int64)
    hl = rng.integers(0, T + 1, size=B).astype(np.int32)
    cand = rng.integers(0, V, size=B).astype(np.int64)
    feats = {"hist": torch.from_numpy(hist).cuda(), "hist_len": torch.from_numpy(hl).cuda(), "cand": torch.from_numpy(cand).cuda()}
    x_cols = None
    if with_columns:
        g, c, age = rng.integers(0, 3, B), rng.integers(0, 200, B), rng.random(B).astype(np.float32)
        feats.update({"gender": torch.from_numpy(g).cuda(), "city": torch.from_numpy(c).cuda(), "age": torch.from_numpy(age).cuda()})
        il = model.input_layer
        blocks = {}
        for col, w in zip(il.emb_cols, il.embedding_weights):
            blocks[col.name] = _np(w)[{"gender_embedding": g, "city_embedding": c}[col.name]]
        blocks["age"] = age[:, None]
        x_cols = np.concatenate([blocks[col.name] for col in il.columns], axis=1)       # name-sorted, as the reference's input_layer
    with torch.no_grad():
        got = model(feats)
        pred = model.predict(feats)
    a = model.attention
    ref = R.din_model_logits(x_cols, _np(a.table), hist, hl, cand, (_np(a.W1), _np(a.b1), _np(a.W2), _np(a.b2), _np(a.W3), _np(a.b3)),
                             _mlp_np(model), (_np(model.logits_layer.weight), _np(model.logits_layer.bias)), normalize=normalize,
                             activation=att_act, act_params=_act_params_np(a))
    assert tuple(got.shape) == (B, 1)
    _close(_np(got), ref, tol=2e-5)
    assert sorted(pred) == ["class_ids", "logistic", "logits", "probabilities"]
    _close(_np(pred["logistic"]), 1 / (1 + np.exp(-ref)), tol=2e-5)
    assert np.array_equal(_np(pred["class_ids"])[:, 0], (ref[:, 0] > 0).astype(np.int64)) or float(np.abs(ref).min()) < 1e-4


@pytest.mark.parametrize("att_act", ["sigmoid", "prelu", "dice"])
def test_din_model_trains(built_lib, att_act):
    """TRAIN mode: the HIP forward / backward for the sigmoid unit, the differentiable torch formulation for PReLU / Dice; the composite's
    forward equals the HIP inference unit once the Dice statistics are frozen; every parameter receives a finite gradient and a few SGD
    steps lower the loss."""
    from dir_amd.din import DIN
    torch.manual_seed(3)
    rng = np.random.default_rng(4)
    V, K, B, T = 500, 64, 128, 20
    model = DIN(item_vocab_size=V, embedding_dim=K, attention_activation=att_act, dnn_hidden_units=(64, 32), dnn_activation_fn="dice").cuda()
    hist = torch.from_numpy(rng.integers(-1, V, size=(B, T)).astype(np.int64)).cuda()
    hl = torch.from_numpy(rng.integers(1, T + 1, size=B).astype(np.int32)).cuda()
    cand = torch.from_numpy(rng.integers(0, V, size=B).astype(np.int64)).cuda()
    y = (torch.rand((B, 1), device="cuda") < 0.4).float()
    feats = {"hist": hist, "hist_len": hl, "cand": cand}
    if att_act != "sigmoid":
        model.eval()
        comp = model.attention._composite(hist, hl, cand).detach()
        with torch.no_grad():
            hip = model.attention(hist, hl, cand)
        _close(_np(comp), _np(hip), tol=2e-5)
    model.train()
    opt = torch.optim.SGD([p for n, p in model.named_parameters() if "table" not in n], lr=0.05)
    losses = []
    for _ in range(6):
        opt.zero_grad(set_to_none=True)
        model.attention.table.grad = None
        loss = torch.nn.functional.binary_cross_entropy_with_logits(model(feats), y)
        loss.backward()
        for n, p in model.named_parameters():
            assert p.grad is not None, n
            g = p.grad.coalesce().values() if p.grad.is_sparse else p.grad
            assert bool(torch.isfinite(g).all()), n
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < losses[0]


@pytest.mark.parametrize("activation", ["prelu", "dice"])
@pytest.mark.parametrize("normalize", [False, True])
def test_din_unit_trains_prelu_dice_on_hip_rows(built_lib, activation, normalize):
    """the PReLU / Dice unit in TRAIN mode on HIP kernels (din.DINAttentionPool._rows_train: Dice with the mini-batch's
    statistics) against float64 autograd of the torch formulation of the same definition (DINAttentionPool._composite restated in double):
    the interest vector, dL/dtable (sparse), dL/dW1..b3, dL/dalpha, and the moving statistics Dice advances -- at 5e-5."""
    from dir_amd.din import DINAttentionPool
    import dir_amd.din as din_mod
    dev = torch.device("cuda:0")
    torch.manual_seed(17)
    V, K, B, T = 300, 64, 96, 50
    unit = DINAttentionPool(V, K, (80, 40), normalize=normalize, activation=activation).to(dev).train()
    with torch.no_grad():
        unit.b1.normal_(0, 0.1); unit.b2.normal_(0, 0.1); unit.b3.fill_(0.05)
        unit.act1.alpha.uniform_(0.05, 0.4); unit.act2.alpha.uniform_(0.05, 0.4)
    ref = DINAttentionPool(V, K, (80, 40), normalize=normalize, activation=activation).to(dev).double().train()
    ref.load_state_dict({k: v.double() for k, v in unit.state_dict().items()})
    g = torch.Generator(device=dev).manual_seed(3)
    hist = torch.randint(-1, V, (B, T), generator=g, device=dev)
    hl = torch.randint(0, T + 1, (B,), generator=g, device=dev, dtype=torch.int32)
    hl[0] = 0
    cand = torch.randint(0, V, (B,), generator=g, device=dev)
    cand[3] = -1
    gout = torch.randn((B, K), generator=g, device=dev)
    assert din_mod.ROWS_TRAIN
    out = unit(hist, hl, cand)
    out.backward(gout)
    rout = ref._composite(hist, hl, cand)
    rout.backward(gout.double())

    def close(a, b, tol=5e-5, what=""):
        err = float(((a.double() - b.double()).abs() / (1 + b.double().abs())).max())
        assert err <= tol, "%s: %.2e" % (what, err)
    close(out, rout, what="interest vector")
    close(unit.table.grad.to_dense(), ref.table.grad.to_dense(), what="dL/dtable")
    for n in ("W1", "b1", "W2", "b2", "W3", "b3"):
        close(getattr(unit, n).grad, getattr(ref, n).grad, what="dL/d" + n)
    close(unit.act1.alpha.grad, ref.act1.alpha.grad, what="dL/dalpha1")
    close(unit.act2.alpha.grad, ref.act2.alpha.grad, what="dL/dalpha2")
    if activation == "dice":
        close(unit.act1.moving_mean, ref.act1.moving_mean, what="moving mean")
        close(unit.act2.moving_variance, ref.act2.moving_variance, what="moving variance")
    # a second identical step: bitwise the same gradients (fixed summation orders)
    g1 = [p.grad.clone() if not p.grad.is_sparse else p.grad.coalesce().values().clone() for p in unit.parameters()]
    unit2 = DINAttentionPool(V, K, (80, 40), normalize=normalize, activation=activation).to(dev).train()
    sd = {k: v.float() for k, v in ref.state_dict().items()}
    if activation == "dice":                                   # the moving statistics of BEFORE the step
        for k in list(sd):
            if "moving_mean" in k:
                sd[k] = torch.zeros_like(sd[k])
            if "moving_variance" in k:
                sd[k] = torch.ones_like(sd[k])
    unit2.load_state_dict(sd)
    unit2(hist, hl, cand).backward(gout)
    g2 = [p.grad.clone() if not p.grad.is_sparse else p.grad.coalesce().values().clone() for p in unit2.parameters()]
    assert all(torch.equal(a, b) for a, b in zip(g1, g2))


@pytest.mark.parametrize("activation", ["prelu", "dice"])
def test_act_rows_layers_train_on_hip(built_lib, activation):
    """autograd.ActRows on a [B, 200] layer output (the DIN model's PReLU / Dice MLP layers in TRAIN mode): forward, dL/ds, dL/dalpha and the
    moving statistics against float64 autograd of din.Dice / din._PReLU."""
    from dir_amd import autograd as ag
    from dir_amd.din import Dice, _PReLU
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    M, N = 4099, 200
    mod = (Dice(N) if activation == "dice" else _PReLU(N)).to(dev).train()
    with torch.no_grad():
        mod.alpha.uniform_(-0.2, 0.5)
    ref = (Dice(N) if activation == "dice" else _PReLU(N)).to(dev).double().train()
    ref.load_state_dict({k: v.double() for k, v in mod.state_dict().items()})
    s = (torch.randn((M, N), device=dev) * 1.5 + 0.3).requires_grad_(True)
    sd = s.detach().double().requires_grad_(True)
    gy = torch.randn((M, N), device=dev)
    y = ag.act_rows(s, mod)
    y.backward(gy)
    yr = ref(sd)
    yr.backward(gy.double())
    for a, b, what in ((y, yr, "y"), (s.grad, sd.grad, "dL/ds"), (mod.alpha.grad, ref.alpha.grad, "dL/dalpha")):
        err = float(((a.double() - b).abs() / (1 + b.abs())).max())
        assert err <= 5e-5, "%s: %.2e" % (what, err)
    if activation == "dice":
        assert float((mod.moving_mean.double() - ref.moving_mean).abs().max()) <= 1e-6
        assert float((mod.moving_variance.double() - ref.moving_variance).abs().max()) <= 1e-6


def test_dice_rows_eval_mode_gradient(built_lib):
    """Dice with MOVING statistics (module.eval(), grad enabled -- frozen-statistics fine-tuning): the normalised pre-activation
    scale * s + shift still depends on s, so dL/ds = d1 + gx * scale.  Against float64 autograd of din.Dice in eval mode."""
    from dir_amd import autograd as ag
    from dir_amd.din import Dice
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    M, N = 2053, 80
    mod = Dice(N).to(dev)
    with torch.no_grad():
        mod.alpha.uniform_(-0.2, 0.5)
        mod.moving_mean.normal_(0.2, 0.5)
        mod.moving_variance.uniform_(0.3, 2.5)
    mod.eval()
    ref = Dice(N).to(dev).double()
    ref.load_state_dict({k: v.double() for k, v in mod.state_dict().items()})
    ref.eval()
    s = (torch.randn((M, N), device=dev) * 1.5 + 0.3).requires_grad_(True)
    sd = s.detach().double().requires_grad_(True)
    gy = torch.randn((M, N), device=dev)
    mm, mv = mod.moving_mean.clone(), mod.moving_variance.clone()
    y = ag.act_rows(s, mod)
    y.backward(gy)
    yr = ref(sd)
    yr.backward(gy.double())
    for a, b, what in ((y, yr, "y"), (s.grad, sd.grad, "dL/ds"), (mod.alpha.grad, ref.alpha.grad, "dL/dalpha")):
        err = float(((a.double() - b).abs() / (1 + b.abs())).max())
        assert err <= 5e-5, "%s: %.2e" % (what, err)
    assert torch.equal(mm, mod.moving_mean) and torch.equal(mv, mod.moving_variance)      # eval mode: the statistics do not move
