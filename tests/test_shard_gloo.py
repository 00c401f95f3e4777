"""Multi-process CPU test of the sharded lookup (dir_amd.shard.ShardedTables) over the gloo backend.

The exchange logic under test is exactly what runs on the GPU box under RCCL: 'div' routing, bucketing by
owner, all_to_all of ids, owner-side row gather, all_to_all of rows, un-permute.  The two HIP kernels
cannot run without a GPU, so a NumPy/oracle backend stands in for them here through the `backend` injection
point (test infrastructure; the product default is shard.HipBackend)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, vocab, K, B, seed, out_q, train=False):
    try:
        _worker_body(rank, world, port, vocab, K, B, seed, out_q, train)
    except Exception:                      # surface the reason instead of leaving the parent to time out
        import traceback
        out_q.put((rank, traceback.format_exc(), 0, 0))


def _worker_body(rank, world, port, vocab, K, B, seed, out_q, train=False):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        from dir_amd.shard import ShardedTables, div_range
        from oracle import np_ref as R
        F = len(vocab)
        rng = np.random.default_rng(seed)           # same full tables on every rank
        full = [rng.standard_normal((v, K)).astype(np.float32) for v in vocab]
        rng_b = np.random.default_rng(seed + 100 + rank)  # each rank draws its own batch
        ids = np.stack([rng_b.integers(-1, v, size=B) for v in vocab], axis=1).astype(np.int64)
        local = []
        for f, v in enumerate(vocab):
            s, e = div_range(v, world, rank)
            local.append(torch.from_numpy(full[f][s:e].copy()))

        class OracleBackend:
            """NumPy stand-ins for the three HIP steps (bucket / gather_packed / finish)."""

            def bucket(self, flat):
                a = flat.numpy()
                n = a.size
                own = np.empty(n, np.int64)
                loc = np.empty(n, np.int64)
                for f in range(F):
                    sel = np.arange(f, n, F)
                    o, l = R.shard_div_owner(np.maximum(a[sel], 0), vocab[f], world)
                    neg = a[sel] < 0
                    own[sel] = np.where(neg, sel % world, o)
                    loc[sel] = np.where(neg, -1, l)
                order = np.argsort(own, kind="stable")
                inv = np.empty(n, np.int64)
                inv[order] = np.arange(n)
                packed = np.where(loc < 0, -1, loc * F + (np.arange(n) % F))[order]
                counts = np.bincount(own, minlength=world).astype(np.int64)
                starts = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)
                return (torch.from_numpy(packed), torch.from_numpy(inv), torch.from_numpy(counts), torch.from_numpy(starts))

            def gather_packed(self, payload):
                p = payload.numpy()
                out = np.zeros((p.size, K), np.float32)
                for i, v in enumerate(p):
                    if v >= 0:
                        out[i] = local[v % F].numpy()[v // F]
                return torch.from_numpy(out)

            def back_buffer(self, n, K_, device):
                return torch.empty((n, K_), dtype=torch.float32)

            def finish(self, back, inv, B_, F_, want_fm):
                emb = back.numpy()[inv.numpy()].reshape(B_, F_ * K)
                fm = None
                if want_fm:
                    from oracle import oracle as O
                    fm = torch.from_numpy(O.fm_second_order(emb, F_, K).reshape(B_, 1))
                return torch.from_numpy(emb), fm

            def make_optimizer(self, lr, init):
                return {"lr": lr, "acc": [np.full(t.shape, init, np.float64) for t in local]}

            def apply_adagrad(self, opt, payload, grad_rows):
                p, g = payload.numpy(), grad_rows.numpy().astype(np.float64)
                for f in range(F):
                    sel = (p >= 0) & (p % F == f)
                    rows = p[sel] // F
                    gsum = np.zeros(local[f].shape)
                    np.add.at(gsum, rows, g[sel])
                    t = np.zeros(local[f].shape[0], bool)
                    t[rows] = True
                    opt["acc"][f][t] += gsum[t] ** 2
                    w = local[f].numpy().astype(np.float64)
                    w[t] -= opt["lr"] * gsum[t] / np.sqrt(opt["acc"][f][t])
                    local[f].copy_(torch.from_numpy(w.astype(np.float32)))

        if train:
            st = ShardedTables(local, vocab, backend=OracleBackend()).enable_training(lr=0.05, initial_accumulator_value=0.1)
            gout = rng_b.standard_normal((B, F * K)).astype(np.float32)
            emb = st.lookup_train(torch.from_numpy(ids))
            fwd_ok = bool(np.array_equal(emb.detach().numpy(), R.embedding_bag_onehot(full, ids)))
            (emb * torch.from_numpy(gout)).sum().backward()
            # reference: one synchronous Adagrad step on the FULL tables over the batches of all ranks
            allb = [None] * world
            dist.all_gather_object(allb, (ids, gout))
            ok = fwd_ok
            for f, v in enumerate(vocab):
                gsum = np.zeros((v, K))
                for ids_r, g_r in allb:
                    sel = ids_r[:, f] >= 0
                    np.add.at(gsum, ids_r[sel, f], g_r[sel, f * K:(f + 1) * K].astype(np.float64))
                t = np.abs(gsum).sum(1) > 0
                acc = np.full((v, K), 0.1)
                acc[t] += gsum[t] ** 2
                ref = full[f].astype(np.float64)
                ref[t] -= 0.05 * gsum[t] / np.sqrt(acc[t])
                s_, e_ = div_range(v, world, rank)
                ok = ok and bool(np.allclose(local[f].numpy(), ref[s_:e_], rtol=1e-6, atol=1e-7))
            out_q.put((rank, ok, B, F * K))
            return
        st = ShardedTables(local, vocab, backend=OracleBackend())
        got, fm = st.lookup(torch.from_numpy(ids), want_fm=True)
        got = got.numpy()
        ref = R.embedding_bag_onehot(full, ids)
        from oracle import oracle as O
        ok = bool(np.array_equal(got, ref)) and bool(np.array_equal(fm.numpy()[:, 0], O.fm_second_order(ref, F, K)))
        out_q.put((rank, ok, int(got.shape[0]), int(got.shape[1])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,vocab", [(2, [10, 7, 33]), (3, [100, 5, 64, 9]), (2, [1000] * 6)])
def test_sharded_lookup_matches_full_tables(world, vocab):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    K, B = 8, 37
    procs = [ctx.Process(target=_worker, args=(r, world, port, vocab, K, B, 4321, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    for rank, ok, b, w in res:
        assert not isinstance(ok, str), "rank %d raised:\n%s" % (rank, ok)
        assert ok, "rank %d: sharded lookup differs from the full-table gather" % rank
        assert (b, w) == (B, len(vocab) * K)


@pytest.mark.parametrize("world,vocab", [(2, [10, 7, 33]), (3, [40, 5, 64, 9])])
def test_sharded_training_step_matches_full_table_adagrad(world, vocab):
    """lookup_train + backward over gloo: every rank's row gradients reach the owners (the forward exchange reversed) and
    the owners' shards end up equal to one synchronous Adagrad step on the full tables over all ranks' batches."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    K, B = 8, 29
    procs = [ctx.Process(target=_worker, args=(r, world, port, vocab, K, B, 777, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, b, w in res:
        assert not isinstance(ok, str), "rank %d raised:\n%s" % (rank, ok)
        assert ok, "rank %d: shard differs from the full-table Adagrad step" % rank


def test_div_range_covers_vocab():
    import sys
    sys.path.insert(0, ROOT)
    from dir_amd.shard import div_range
    from oracle import np_ref as R
    for V, P in [(10, 4), (7, 8), (1000000, 8), (13, 2)]:
        edges = [div_range(V, P, r) for r in range(P)]
        assert edges[0][0] == 0 and edges[-1][1] == V
        for (s0, e0), (s1, e1) in zip(edges, edges[1:]):
            assert e0 == s1
        ids = np.arange(V) if V <= 1000 else np.random.default_rng(0).integers(0, V, 1000)
        own, loc = R.shard_div_owner(ids, V, P)
        for i, o, l in zip(ids, own, loc):
            s, e = edges[o]
            assert s <= i < e and l == i - s


def _allreduce_worker(rank, world, port, out_q):
    try:
        import sys
        sys.path.insert(0, ROOT)
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        import datetime
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
        try:
            from dir_amd.shard import allreduce_grads
            g = torch.Generator().manual_seed(11)
            shapes = [(7, 5), (3,), (64, 33), (1,), (9, 2, 2)]
            params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes] + [torch.nn.Parameter(torch.zeros(4))]   # the last has no grad
            base = [torch.randn(s, generator=g) for s in shapes]
            for p, b in zip(params, base):
                p.grad = b * (rank + 1)
            allreduce_grads(params, bucket_bytes=1024)                      # several buckets
            tot = sum(r + 1 for r in range(world))
            ok = all(torch.allclose(p.grad, b * tot, rtol=1e-6, atol=1e-6) for p, b in zip(params, base)) and params[-1].grad is None
            for p, b in zip(params, base):
                p.grad = b * (rank + 1)
            allreduce_grads(params, average=True)
            ok = ok and all(torch.allclose(p.grad, b * tot / world, rtol=1e-6, atol=1e-6) for p, b in zip(params, base))
            out_q.put((rank, ok))
        finally:
            dist.destroy_process_group()
    except Exception:
        import traceback
        out_q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 3])
def test_allreduce_grads_sums_replicated_gradients(world):
    """shard.allreduce_grads (the dense side of ShardedDeepFMTrainer): bucketed flat all-reduce = per-tensor sum / mean."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_allreduce_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok in res:
        assert not isinstance(ok, str), "rank %d raised:\n%s" % (rank, ok)
        assert ok, "rank %d: all-reduced gradients differ" % rank
