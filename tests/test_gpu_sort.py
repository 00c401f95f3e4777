"""The in-tree radix sort (csrc/radix_sort.hip) behind the sorted sparse updates: bit-exact against a stable CPU sort, and the property the
round needed it for -- a training step that sorts can be captured in a HIP graph and replayed after other eager sorts have run."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _sort(lib, keys, vals, bits):
    n = keys.numel()
    need = int(lib.dir_debug_radix_sort_workspace_bytes(n, bits))
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    ws.random_(0, 255)                                   # "any content"
    ko, vo = torch.empty_like(keys), torch.empty_like(vals)
    ki, vi = keys.clone(), vals.clone()
    p = lambda t: ctypes.c_void_p(t.data_ptr())           # noqa: E731
    rc = lib.dir_debug_radix_sort_pairs_u32(p(ki), p(vi), n, bits, p(ko), p(vo), p(ws), need, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, lib.dir_last_error()
    return ko, vo


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 511, 8191, 8192, 8193, 100003, 65536 * 26, 3000017])
@pytest.mark.parametrize("bits", [1, 7, 8, 9, 16, 17, 25, 27, 32])
def test_radix_sort_matches_stable_cpu_sort(built_lib, n, bits):
    if n > 200000 and bits not in (9, 25, 32):
        pytest.skip("large sizes: the pass structures 1 x 9, 3 x 9 and 4 x 8 only")
    rng = np.random.default_rng(n * 131 + bits)
    hi = (1 << bits) - 1
    kinds = ["uniform", "few", "equal", "zipf"] if n > 64 else ["uniform", "few"]
    for kind in kinds:
        if kind == "uniform":
            k = rng.integers(0, hi + 1, size=n, dtype=np.uint64)
        elif kind == "few":
            k = rng.choice(rng.integers(0, hi + 1, size=5, dtype=np.uint64), size=n)
        elif kind == "equal":
            k = np.full(n, hi // 3, dtype=np.uint64)
        else:
            k = np.minimum(rng.zipf(1.05, size=n).astype(np.uint64), np.uint64(hi))
        k = k.astype(np.uint32)
        v = np.arange(n, dtype=np.uint32)                 # equal keys must keep their input order: the values come out ascending
        order = np.argsort(k, kind="stable")
        ko, vo = _sort(built_lib, torch.from_numpy(k.view(np.int32)).cuda(), torch.from_numpy(v.view(np.int32)).cuda(), bits)
        torch.cuda.synchronize()
        assert np.array_equal(ko.cpu().numpy().view(np.uint32), k[order]), (kind, n, bits)
        assert np.array_equal(vo.cpu().numpy().view(np.uint32), v[order]), (kind, n, bits)


def test_radix_sort_ignores_key_bits_above_bits(built_lib):
    """Only the low `bits` bits order the pairs (the callers' keys never carry more, but the contract says so)."""
    rng = np.random.default_rng(5)
    n, bits = 50000, 12
    k = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    v = np.arange(n, dtype=np.uint32)
    order = np.argsort(k & np.uint32((1 << bits) - 1), kind="stable")
    ko, vo = _sort(built_lib, torch.from_numpy(k.view(np.int32)).cuda(), torch.from_numpy(v.view(np.int32)).cuda(), bits)
    assert np.array_equal(ko.cpu().numpy().view(np.uint32), k[order]) and np.array_equal(vo.cpu().numpy().view(np.uint32), v[order])


def _slot_sort(lib, ids, vocab):
    B, F = ids.shape
    row_base = torch.tensor(np.concatenate([[0], np.cumsum(vocab)[:-1]]), dtype=torch.int64, device="cuda")
    total = int(sum(vocab))
    need = int(lib.dir_debug_slot_sort_workspace_bytes(B, F, total))
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    ws.random_(0, 255)                                   # "any content"
    ko = torch.empty(B * F, dtype=torch.int32, device="cuda")
    vo = torch.empty(B * F, dtype=torch.int32, device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr())           # noqa: E731
    rc = lib.dir_debug_slot_sort_entries(p(ids), ids.stride(0), ids.stride(1), F, B, p(row_base), total, p(ko), p(vo), p(ws), need,
                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, lib.dir_last_error()
    torch.cuda.synchronize()
    return ko.cpu().numpy().view(np.uint32), vo.cpu().numpy().view(np.uint32), row_base.cpu().numpy(), total


@pytest.mark.parametrize("B,vocab", [
    (4096, [1000] * 5),                                        # one 10-bit digit per slot
    (8192, [100000] * 26),                                     # two digits
    (8193, [7, 1024, 1025, 1 << 20, (1 << 20) - 1, 3000000, 1]),      # 1, 2 (vocab 1024: the pruned key needs an 11th bit), 2, 3, 2, 3 digits, 1
    (65536, [1000000] * 26),                                   # BASELINE's shape: 3 launches, every segment sits out the first
    (20000, [40000000, 5, 300]),                               # bits(total_rows) = 26: three launches, one segment uses all of them
    (5000, [(1 << 31) + 12345, 900]),                          # 32-bit total: four launches
    (4100, [50 + 37 * i for i in range(40)]),                  # more than 32 slots: two transposition chunks; 1 and 2 digits mixed
])
@pytest.mark.parametrize("kind", ["uniform", "few", "zipf", "strided"])
def test_slot_sort_matches_stable_cpu_sort(built_lib, B, vocab, kind):
    """dir_debug_slot_sort_entries (the sort the sorted sparse updates run on ids [B, F], B >= 4096) against numpy: keys = global rows
    (total_rows for pruned ids: negative, >= the slot's vocabulary), ordered by (slot, local row with pruned last, batch position);
    slots of very different vocabularies (1 to 4 digit launches) in one call; ids through a strided view."""
    F = len(vocab)
    rng = np.random.default_rng(B + F + len(kind))
    cols = []
    for v in vocab:
        if kind == "few":
            c = rng.choice(rng.integers(0, v, size=4), size=B)
        elif kind == "zipf":
            c = np.minimum(rng.zipf(1.1, size=B) - 1, v - 1)
        else:
            c = rng.integers(0, v, size=B)
        c = c.astype(np.int64)
        c[rng.random(B) < 0.03] = -1                      # pruned
        c[rng.random(B) < 0.02] = v + int(rng.integers(0, 5))      # out of the vocabulary: pruned too
        cols.append(c)
    ids_np = np.stack(cols, axis=1)
    if kind == "strided":
        wide = torch.full((B, 2 * F + 3), -7, dtype=torch.int64, device="cuda")
        ids = wide[:, 1:1 + 2 * F:2]
        ids.copy_(torch.from_numpy(ids_np))
    else:
        ids = torch.from_numpy(ids_np).cuda()
    ko, vo, row_base, total = _slot_sort(built_lib, ids, vocab)
    ek, ev = [], []
    for f, v in enumerate(vocab):
        c = ids_np[:, f]
        ok = (c >= 0) & (c < v)
        local = np.where(ok, c, v)
        order = np.argsort(local, kind="stable")
        ek.append(np.where(ok[order], row_base[f] + c[order], total).astype(np.uint32))
        ev.append((order * F + f).astype(np.uint32))
    assert np.array_equal(ko, np.concatenate(ek)) and np.array_equal(vo, np.concatenate(ev))


def test_slot_sort_covers_what_it_says(built_lib):
    assert built_lib.dir_debug_slot_sort_workspace_bytes(4095, 26, 26000) == 0          # small batches sort global keys
    assert built_lib.dir_debug_slot_sort_workspace_bytes(4096, 26, 26000) > 0


def _sparse_step_factory(seed, V=50000, B=8192, F=26, K=16):
    from dir_amd import ops
    gen = torch.Generator(device="cuda").manual_seed(seed)
    tables = [torch.randn((V, K), generator=gen, device="cuda") * 0.25 for _ in range(F)]
    ts = ops.TableSet.train_rows(tables)
    opt = ops.SparseAdagrad(ts, lr=0.01)
    lin = ops.TableSet([torch.zeros((V,), device="cuda") for _ in range(F)])
    ftrl = ops.SparseFtrl(lin, lr=0.2)
    ops.share_sorted_entries(opt, ftrl)
    ids = torch.randint(0, V, (B, F), generator=gen, device="cuda")
    out = torch.empty((B, F * K), device="cuda")
    fm = torch.empty((B, 1), device="cuda")
    fsum = torch.empty((B, K), device="cuda")
    gfm = torch.randn((B, 1), generator=gen, device="cuda") * 0.01
    gd = torch.randn((B, F * K), generator=gen, device="cuda") * 0.01
    glin = torch.randn((B, 1), generator=gen, device="cuda") * 0.01

    def step():
        ops.gather_fm(ts, ids, out=out, fm=fm, fsum=fsum)
        opt.step_fm(ids, gd, gfm, fsum)
        ftrl.step(ids, glin)
    return step, ts, lin, ids


def test_sparse_training_step_replays_from_a_hip_graph(built_lib):
    """The sparse side of a DeepFM training step (gather + FM forward, the fused sorted Adagrad with the FM backward folded in, the
    sorted FTRL sharing its sort) captured in a torch.cuda.CUDAGraph (= hipGraph): replays are bitwise equal to the same steps run
    eagerly on an identical twin, and STAY valid after 80 eager steps of another optimiser in between -- with rocPRIM's sort (7
    hipMemsetAsync per call: memset nodes) that replay faulted (profiles/NOTES.md R4.3)."""
    a, ts_a, lin_a, ids_a = _sparse_step_factory(1)
    b, ts_b, lin_b, ids_b = _sparse_step_factory(1)
    other, _, _, _ = _sparse_step_factory(2)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            b()
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(3):
        a()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        b()
    gen = torch.Generator(device="cuda").manual_seed(9)
    for r in range(3):
        new_ids = torch.randint(0, 50000, tuple(ids_a.shape), generator=gen, device="cuda")
        ids_a.copy_(new_ids)
        ids_b.copy_(new_ids)
        g.replay()
        a()
        if r == 1:
            for _ in range(80):
                other()
        torch.cuda.synchronize()
        # (the arenas' alignment padding is uninitialised: compare the row blocks -- embeddings and accumulators)
        assert all(torch.equal(x, y) for x, y in zip(ts_a.rows, ts_b.rows)), "replay %d differs from the eager twin" % r
        assert all(torch.equal(x, y) for x, y in zip(lin_a.tables, lin_b.tables))
