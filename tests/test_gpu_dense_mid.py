"""dir_dense_mid_f32 (round 6, csrc/dense.hip: dense_mid_k) -- the hidden layers of the towers (tf.layers.dense: models/DeepFM/deepFM.py:295-300,
models/DeepCrossNetwork/DeepCrossNetwork.py:394-399) at MID-SIZE batches, where round 5 still sent them to the library: against float64, and the
models' PRODUCT routing (dense.MIN_ROWS = 6144, not the suite's override) at B = 1024 / 2048 / 4096 (the reference sets its batch size by flag:
models/DeepCrossNetwork/train.py:16-17)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M", [513, 1000, 2048, 4095, 6143])
@pytest.mark.parametrize("Kd,N", [(416, 400), (432, 1024), (1024, 1024), (360, 200), (200, 80), (100, 40), (36, 16), (4, 17)])
def test_dense_mid_matches_float64(built_lib, M, Kd, N):
    from dir_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + Kd)
    x = torch.randn((M, Kd), generator=g, device="cuda") * 0.3
    W = torch.randn((N, Kd), generator=g, device="cuda") * (1.0 / Kd ** 0.5)
    b = torch.randn((N,), generator=g, device="cuda") * 0.1
    assert ops.dense_mid_covers(M, Kd, N)
    ref = x.double() @ W.double().t() + b.double()
    for relu in (False, True):
        y = ops.dense(x, W, b, relu=relu, arith="f32")
        r = torch.relu(ref) if relu else ref
        assert float(((y.double() - r).abs() / (1 + r.abs())).max()) <= 1e-5
    # the folded inference batch norm in the epilogue, a strided input and a strided output
    ps, psh = torch.rand((N,), generator=g, device="cuda") + 0.5, torch.randn((N,), generator=g, device="cuda") * 0.1
    xs = torch.zeros((M, Kd + 8), device="cuda")
    xs[:, :Kd] = x
    out = torch.full((M, N + 4), 7.0, device="cuda")
    y = ops.dense(xs[:, :Kd], W, b, relu=True, post_scale=ps, post_shift=psh, out=out[:, :N], arith="f32")
    r = torch.relu(ref) * ps.double() + psh.double()
    assert float(((y.double() - r).abs() / (1 + r.abs())).max()) <= 1e-5
    assert bool((out[:, N:] == 7.0).all())                       # nothing written past the N columns
    assert torch.equal(ops.dense(x, W, b, relu=True, arith="f32"), ops.dense(x, W, b, relu=True, arith="f32"))      # rerun: bitwise


def test_dense_mid_argument_errors(built_lib):
    import ctypes
    p = ctypes.c_void_p(256)
    rc = built_lib.dir_dense_mid_f32(p, 416, p, 416, None, 0, None, None, 1024, 414, 400, p, 400, None)
    assert rc == -4 and b"multiples of 4" in built_lib.dir_last_error()
    rc = built_lib.dir_dense_mid_f32(p, 416, p, 416, None, 0, p, None, 1024, 416, 400, p, 400, None)
    assert rc == -1 and b"come together" in built_lib.dir_last_error()
    rc = built_lib.dir_dense_mid_f32(None, 416, p, 416, None, 0, None, None, 1024, 416, 400, p, 400, None)
    assert rc == -1 and b"null pointer" in built_lib.dir_last_error()


@pytest.mark.parametrize("B", [768, 1024, 2048, 4096])      # (768: under ops.TOWER_MIN_ROWS, where DeepFM's layers run one by one too)
@pytest.mark.parametrize("which", ["deepfm", "dcn"])
def test_models_at_mid_batches_stay_on_hip_kernels(built_lib, monkeypatch, B, which):
    """DeepFM (26 x 16, 400-400-400) and DCN (d = 429, cross 3, deep 1024-1024 with batch norm) inference with the PRODUCT's routing threshold:
    every hidden layer runs dir_dense_mid_f32 (no nn.Linear), and the logits equal the library-routed forward (DIR_DENSE_MID_ROWS = 0: round 5's
    routing, torch fp32) within 1e-5."""
    from dir_amd import dense as D, ops
    from dir_amd import feature_column as fc
    from dir_amd.deepfm import DeepFM
    from dir_amd.dcn import DeepCrossNetwork
    monkeypatch.setattr(D, "MIN_ROWS", 6144)
    F, K, V = 26, 16, 5000
    g = torch.Generator(device="cuda").manual_seed(B)
    cats = [fc.categorical_column_with_identity("C%02d" % i, V) for i in range(F)]
    if which == "deepfm":
        model = DeepFM(linear_feature_columns=cats, dnn_feature_columns=[fc.embedding_column(c, K) for c in cats], dnn_hidden_units=[400, 400, 400],
                       fm_embedding_size=K).cuda().eval()
    else:
        nums = [fc.numeric_column("I%02d" % i) for i in range(13)]
        model = DeepCrossNetwork(columns=[fc.embedding_column(c, K) for c in cats] + nums, cross_layer_num=3, dnn_hidden_units=[1024, 1024],
                                 batch_norm=True).cuda().eval()
    feats = {"C%02d" % i: torch.randint(0, V, (B,), generator=g, device="cuda") for i in range(F)}
    feats.update({"I%02d" % i: torch.rand((B,), generator=g, device="cuda") for i in range(13)})
    with torch.no_grad():
        D.reset_routing()
        got = model(feats)
        routed = {k: dict(v) for k, v in D.ROUTING.items()}
        monkeypatch.setattr(ops, "DENSE_MID_ROWS", 0)
        D.reset_routing()
        ref = model(feats)
        assert D.ROUTING["library"] or (which == "deepfm" and B >= ops.TOWER_MIN_ROWS), dict(D.ROUTING)      # (round 5's routing: these layers on nn.Linear)
    # (from ops.TOWER_MIN_ROWS rows DeepFM's whole tower is ONE fused launch and routes no layer at all)
    assert (routed["hip"] or (which == "deepfm" and B >= ops.TOWER_MIN_ROWS)) and all(int(k.split("x")[1]) < 16 for k in routed["library"]), routed
    assert float(((got.double() - ref.double()).abs() / (1 + ref.double().abs())).max()) <= 1e-5
