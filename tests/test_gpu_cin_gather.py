"""The CIN stack and the xDeepFM forward over a row-sharded lookup WITHOUT its finish pass (round 6): x0 read through the
inverse positions of the received rows.  The bar is BITWISE equality with the plain kernels on the materialised x0 (same kernels, only the
staging of x0 differs); the plain kernels are held to the oracle in tests/test_gpu_parity.py.  Reference: the CIN restates arXiv:1803.05170
(/root/reference/README.md:28); the sharding follows /root/reference/models/DeepFM/deepFM.py:163-167."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rows_inv(B, m, D, n_extra, seed, pruned=True):
    """A row list in shuffled order with spare rows nobody points at, and the inverse positions [B, m] (some < 0)."""
    g = torch.Generator().manual_seed(seed)
    n = B * m + n_extra
    rows = (torch.randn((n, D), generator=g) * 0.25).cuda()
    perm = torch.randperm(n, generator=g)[:B * m]
    inv = perm.view(B, m).clone()
    if pruned and B > 0:
        mask = torch.rand((B, m), generator=g) < 0.05
        inv[mask] = -1
        inv[0, :] = -1                                      # a sample with no live field at all
    return rows, inv.cuda()


def _materialise(rows, inv):
    B, m = inv.shape
    x0 = rows[inv.clamp_min(0).reshape(-1)].view(B, m, rows.shape[1]).clone()
    x0[inv < 0] = 0.0
    return x0


def _weights(m, Hs, seed):
    g = torch.Generator().manual_seed(seed)
    Ws, hp = [], m
    for h in Hs:
        Ws.append((torch.randn((h, hp * m), generator=g) * (1.0 / (hp * m) ** 0.5)).cuda())
        hp = h
    return Ws


def _plain_stack(x0, Ws, Hs):
    from dir_amd import ops
    B = x0.shape[0]
    pooled = torch.empty((B, sum(Hs)), dtype=torch.float32, device=x0.device)
    xk, off = x0, 0
    for k, (W, h) in enumerate(zip(Ws, Hs)):
        xk, _ = ops.cin_layer(x0, xk, W, pooled=pooled[:, off:off + h], want_xout=k + 1 < len(Hs))
        off += h
    return pooled


@pytest.mark.parametrize("B,m,Hs", [(300, 26, (128, 128, 128)), (257, 26, (96, 64, 32)), (1, 26, (64, 64)), (1000, 20, (64, 32)),
                                    (4099, 32, (128, 64)), (130, 40, (64, 128))])
def test_cin_stack_gather_is_the_plain_stack_bit_for_bit(built_lib, B, m, Hs):
    from dir_amd import ops
    D = 16
    if not ops.cin_gather_covers(m, D, Hs):
        pytest.skip("not covered: m=%d Hs=%r" % (m, Hs))
    rows, inv = _rows_inv(B, m, D, n_extra=77, seed=B + m)
    Ws = _weights(m, Hs, seed=3)
    ref = _plain_stack(_materialise(rows, inv), Ws, Hs)
    got = torch.full((B, sum(Hs)), float("nan"), device="cuda")
    ops.cin_stack_gather(rows, inv, Ws, got)
    assert torch.equal(got, ref)
    again = torch.empty_like(got)
    ops.cin_stack_gather(rows, inv, Ws, again)
    assert torch.equal(again, got)
    # the plain stack itself against float64 on a few samples (the gather form inherits the plain kernels' parity)
    x0 = _materialise(rows, inv)[:8].double().cpu().numpy()
    xk, cols = x0, []
    for W in Ws:
        Wn = W.double().cpu().numpy().reshape(W.shape[0], xk.shape[1], m)
        xk = np.einsum("hij,bid,bjd->bhd", Wn, xk, x0)
        cols.append(xk.sum(-1))
    want = np.concatenate(cols, 1)
    scale = np.abs(want).max() + 1e-30
    assert np.abs(got[:8].double().cpu().numpy() - want).max() / scale < 1e-5


def test_cin_gather_covers_follows_the_default_routing(built_lib):
    from dir_amd import ops
    assert ops.cin_gather_covers(26, 16, (128, 128, 128))             # BASELINE config 5
    assert not ops.cin_gather_covers(26, 16, (128,))                  # a one-layer stack has no pooled last layer behind a first layer
    assert not ops.cin_gather_covers(6, 16, (64, 32))                 # too few fields for the pair form
    assert not ops.cin_gather_covers(26, 8, (64, 32))                 # the fused pooled layer is D = 16
    rows, inv = _rows_inv(16, 6, 16, 0, 1)
    with pytest.raises(ValueError):
        ops.cin_stack_gather(rows, inv, _weights(6, (64, 32), 1), torch.empty((16, 96), device="cuda"))


def test_cin_gather_entries_check_their_arguments(built_lib):
    lib = built_lib
    rows, inv = _rows_inv(32, 26, 16, 0, 2)
    W = _weights(26, (64,), 4)[0]
    nbytes = int(lib.dir_cin_layer1_bf16x3_workspace_bytes(26, 64))
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device="cuda")
    wp = ctypes.c_void_p(ws.data_ptr() + (-ws.data_ptr()) % 256)
    xout = torch.empty((32, 64, 16), device="cuda")
    pooled = torch.empty((32, 64), device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
    rc = lib.dir_cin_layer1_f16x2_gather_f32(P(rows), None, P(W), 26, 64, 16, 32, P(xout), P(pooled), 64, wp, nbytes, None, None)
    assert rc != 0 and b"x0_inv" in lib.dir_last_error()
    rc = lib.dir_cin_layer_f16x2_gather_f32(P(rows), None, P(xout), P(W), 26, 64, 64, 16, 32, P(xout), P(pooled), 64, wp, nbytes, None, None, None)
    assert rc != 0 and b"x0_inv" in lib.dir_last_error()
    rc = lib.dir_cin_pooled_last_bf16x3_gather_f32(P(rows), None, P(xout), P(ws), 26, 64, 64, 16, 32, P(pooled), 64, None)
    assert rc != 0 and b"x0_inv" in lib.dir_last_error()
    # B = 0: nothing to do, null pointers allowed
    assert lib.dir_cin_layer1_f16x2_gather_f32(None, None, P(W), 26, 64, 16, 0, None, None, 64, wp, nbytes, None, None) == 0


def _xdeepfm(m, D, Hs, hidden, seed=0):
    from dir_amd.xdeepfm import XDeepFM
    from dir_amd import feature_column as fc
    torch.manual_seed(seed)
    cats = [fc.categorical_column_with_identity("C%d" % i, 50) for i in range(m)]
    return XDeepFM(linear_feature_columns=None, dnn_feature_columns=[fc.embedding_column(c, D) for c in cats], cin_layer_sizes=Hs,
                   dnn_hidden_units=hidden).cuda().eval()


@pytest.mark.parametrize("B", [700, 9000])
@pytest.mark.parametrize("dedup", [False, True])
def test_xdeepfm_predict_over_sharded_tables_has_no_finish_pass_and_the_same_logits(built_lib, B, dedup):
    """One rank (the exchange-free pipeline: bucket + owner gather + consumer): shard.xdeepfm_predict through lookup_rows against
    lookup() + forward_embedded(), bit for bit; 9000 rows take the tower's gather form too, 700 materialise the rows for the tower only."""
    from dir_amd import ops, shard
    m, D = 26, 16
    g = torch.Generator().manual_seed(5)
    vocab = [1000 + 37 * f for f in range(m)]
    full = [(torch.randn((v, D), generator=g) * 0.25).cuda() for v in vocab]
    st = shard.ShardedTables.from_full(full, check="eager", dedup=dedup)
    model = _xdeepfm(m, D, (64, 64, 32), (128, 128))     # (widths the one-launch tower takes in BOTH forms: the plain one wants >= ops.TOWER_MIN_WIDTH)
    ids = torch.stack([torch.randint(-1, v + 3, (B,), generator=g) for v in vocab], dim=1).cuda()      # pruned and out-of-range ids too
    lin = (torch.randn((B, 1), generator=g) * 0.1).cuda()
    with torch.no_grad():
        ref = model.forward_embedded(st.lookup(ids), lin, range_ok=ops.f16_range_ok(st.absmax()))
    launches = []
    real = ops.embedding_bag

    def spy(*a, **k):
        launches.append(1)
        return real(*a, **k)
    ops.embedding_bag = spy
    try:
        got = shard.xdeepfm_predict(model, st, ids, lin)
    finally:
        ops.embedding_bag = real
    assert torch.equal(got, ref)
    if B >= 2 * ops.TOWER_MIN_ROWS:                      # (two micro-batches, each large enough for the tower's one-launch form)
        assert not launches, "the rows were materialised although every kernel covers the gather form"
    # the handle form: the next lookup issued before this one's rows are consumed
    h0 = st.lookup_rows_async(ids)
    h1 = st.lookup_rows_async(ids.flip(0))
    out0 = torch.empty_like(ref)
    for s, e, rows, inv in h0.result():
        out0[s:e] = model.forward_rows(rows, inv, lin[s:e], absmax=st.absmax())
    out1 = torch.empty_like(ref)
    for s, e, rows, inv in h1.result():
        out1[s:e] = model.forward_rows(rows, inv, lin.flip(0)[s:e], absmax=st.absmax())
    assert torch.equal(out0, ref) and torch.equal(out1.flip(0), ref)


def test_lookup_rows_overflow_is_repaired_before_the_rows_are_handed_out(built_lib):
    from dir_amd import shard
    m, D, B = 26, 16, 600
    g = torch.Generator().manual_seed(8)
    full = [(torch.randn((400, D), generator=g) * 0.25).cuda() for _ in range(m)]
    st = shard.ShardedTables.from_full(full, check="eager", force_collective=False)
    ids = torch.randint(0, 400, (B, m), generator=g).cuda()
    chunks = st.lookup_rows(ids)
    emb = torch.empty((B, m * D), device="cuda")
    for s, e, rows, inv in chunks:
        x = rows[inv.clamp_min(0).reshape(-1)].view(e - s, m, D)
        x = torch.where((inv >= 0)[:, :, None], x, torch.zeros_like(x))
        emb[s:e] = x.reshape(e - s, m * D)
    assert torch.equal(emb, st.lookup(ids))
    with pytest.raises(ValueError):
        shard.ShardedTables.from_full(full, check="lazy").lookup_rows(ids)
