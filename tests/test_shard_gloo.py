"""Multi-process CPU test of the sharded lookup (dir_amd.shard.ShardedTables) over the gloo backend.

The exchange logic under test is exactly what runs on the GPU box under RCCL: 'div' routing, bucketing by
owner, all_to_all of ids, owner-side row gather, all_to_all of rows, un-permute.  The two HIP kernels
cannot run without a GPU, so a NumPy/oracle backend stands in for them here through the `backend` injection
point (test infrastructure; the product default is shard.HipBackend)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """A rendezvous token for one process group: the path of a FileStore file (no TCP port to clash on -- a port probed free here can be
    taken again before rank 0 binds it on a shared host)."""
    import tempfile
    return os.path.join(tempfile.mkdtemp(prefix="dir_pg_"), "store")


def _worker(rank, world, port, vocab, K, B, seed, out_q, train=False, opts=None):
    try:
        _worker_body(rank, world, port, vocab, K, B, seed, out_q, train, opts or {})
    except Exception:                      # surface the reason instead of leaving the parent to time out
        import traceback
        out_q.put((rank, traceback.format_exc(), 0, 0))


def _worker_body(rank, world, port, vocab, K, B, seed, out_q, train, opts):
    import sys
    sys.path.insert(0, ROOT)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    import datetime
    dist.init_process_group("gloo", init_method="file://" + port, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        from dir_amd.shard import ShardedTables, div_range, local_slice, place_slices
        from oracle import np_ref as R
        F = len(vocab)
        parts = opts.get("partitions") or [world] * F
        first = place_slices(parts, world) if opts.get("partitions") else [0] * F
        rng = np.random.default_rng(seed)           # same full tables on every rank
        full = [rng.standard_normal((v, K)).astype(np.float32) for v in vocab]
        rng_b = np.random.default_rng(seed + 100 + rank)  # each rank draws its own batch
        hi = opts.get("id_hi")                       # small id range: many duplicates (dedup) / one hot owner (overflow)
        ids = np.stack([rng_b.integers(-1, min(v, hi) if hi else v, size=B) for v in vocab], axis=1).astype(np.int64)
        if opts.get("out_of_range"):
            ids[::5, 0] = vocab[0] + 3               # an id nobody owns: zeros, like a pruned id
        local = []
        for f, v in enumerate(vocab):
            s, e = local_slice(v, parts[f], first[f], world, rank)
            local.append(torch.from_numpy(full[f][s:e].copy()))

        def route(a):
            """owner rank / local row of every entry of a flat [.., F] id array (-1: pruned or out of range)."""
            n = a.size
            own = np.full(n, -1, np.int64)
            loc = np.full(n, -1, np.int64)
            for f in range(F):
                sel = np.arange(f, n, F)
                ok = (a[sel] >= 0) & (a[sel] < vocab[f])
                o, l = R.shard_div_owner(np.where(ok, a[sel], 0), vocab[f], parts[f])
                own[sel] = np.where(ok, (np.asarray(o) + first[f]) % world, -1)
                loc[sel] = np.where(ok, l, -1)
            return own, loc

        class OracleBackend:
            """NumPy stand-ins for the HIP steps of both lookup paths."""

            # ---- exact path ----
            def bucket(self, flat):
                a = flat.numpy()
                n = a.size
                own, loc = route(a)
                own = np.where(own < 0, np.arange(n) % world, own)          # pruned entries travel as -1 payloads
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

            def finish(self, back, inv, B_, F_, want_fm, out=None, fm=None):
                iv = inv.numpy()
                emb = np.where((iv >= 0)[:, None], back.numpy()[np.maximum(iv, 0)], 0).astype(np.float32).reshape(B_, F_ * K)
                fmv = None
                if want_fm:
                    from oracle import oracle as O
                    fmv = torch.from_numpy(O.fm_second_order(emb, F_, K).reshape(B_, 1))
                    if fm is not None:
                        fm.copy_(fmv)
                        fmv = fm
                emb = torch.from_numpy(emb)
                if out is not None:
                    out.copy_(emb)
                    emb = out
                return emb, fmv

            # ---- fixed-capacity path ----
            def new_workspace(self, device):
                return torch.zeros(64, dtype=torch.int32)

            def bucket_cap(self, ids2d, cap, payload, inv, counts, overflow, workspace, stat=None, dedup=False):
                a = ids2d.numpy().reshape(-1)
                own, loc = route(a)
                pay = payload.numpy().reshape(world, cap + 1)
                iv = inv.numpy()
                iv[:] = -1
                fill = np.zeros(world, np.int64)
                seen = {}
                for i in range(a.size):
                    o = own[i]
                    if o < 0:
                        continue
                    p = loc[i] * F + (i % F)
                    if dedup and (o, p) in seen:                 # an EXACT unique per owner (the HIP kernel's is per tile: same result)
                        iv[i] = seen[(o, p)]
                        continue
                    if fill[o] < cap:
                        pay[o, 1 + fill[o]] = p
                        iv[i] = o * cap + fill[o]
                    if dedup:
                        seen[(o, p)] = iv[i]
                    fill[o] += 1
                pay[:, 0] = np.minimum(fill, cap) | (int(fill.max()) << 32)   # header: valid slots | this sender's largest demand
                counts.copy_(torch.from_numpy(fill))
                overflow.fill_(int((fill > cap).any()))
                if stat is not None:
                    stat[0] = int((fill > cap).any())
                    stat[1] = int(fill.max())

            @staticmethod
            def inv2d(inv, Bc, F_, dedup):
                return inv.view(Bc, F_)

            def slab_stat(self, recv_all, n_slabs, cap, stat):
                h = recv_all.numpy().reshape(n_slabs, cap + 1)[:, 0] >> 32
                stat[0] = int(h.max() > cap)
                stat[1] = int(h.max())

            def gather_slabs(self, recv, cap, out):
                r = recv.numpy().reshape(world, cap + 1)
                o = out.numpy()
                for sl in range(world):
                    for j in range(int(r[sl, 0] & 0xffffffff)):
                        v = r[sl, 1 + j]
                        o[sl * cap + j] = local[v % F].numpy()[v // F]

            def finish_chunk(self, back, inv2d, want_fm, out, fm):
                b_, f_ = inv2d.shape
                self.finish(back, inv2d.reshape(-1), b_, f_, want_fm, out=out, fm=fm)

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

        kw = {k: opts[k] for k in ("partitions", "chunks", "slack", "mode", "check", "dedup") if k in opts}
        if opts.get("absmax"):
            # ShardedTables.absmax must decide its MAX all-reduce from rank-invariant state.  Sparse updates move the
            # largest |value| of ONE rank's shard only; the ranks must still enter (or skip) the collective together, and the lookup's
            # all_to_all that follows must pair up.
            class _TS:
                def absmax(self, every=1):
                    return max([float(t.abs().max()) for t in local if t.numel()] or [0.0])
            be = OracleBackend()
            be.ts = _TS()
            st = ShardedTables(local, vocab, backend=be, **kw).enable_training(lr=0.05, initial_accumulator_value=0.1)
            calls = [0]
            real = dist.all_reduce

            def absmax(**k):                                   # st.absmax with its all-reduces counted
                def counted(*a, **kk):
                    calls[0] += 1
                    return real(*a, **kk)
                dist.all_reduce = counted
                try:
                    return st.absmax(**k)
                finally:
                    dist.all_reduce = real
            tid = torch.from_numpy(ids)

            def gmax():
                m = [None] * world
                dist.all_gather_object(m, be.ts.absmax())
                return float(np.float32(max(m)))
            ok = absmax() == gmax() and calls[0] == 1
            ok = ok and absmax() == gmax() and calls[0] == 1          # nothing moved: cached, no collective
            want_calls, held = 1, gmax()
            for step in range(1, 41):
                if rank == 1:                                  # only rank 1's largest value moves (what a sparse owner-side update does)
                    local[0][0, 0] = 100.0 + step
                emb = st.lookup_train(tid)
                emb.sum().backward()
                if step == 32:
                    want_calls, held = want_calls + 1, gmax()
                got = absmax(every=32)
                ok = ok and calls[0] == want_calls and got == held      # the same decision and the same figure on EVERY rank
                st.lookup(tid)                                  # the exchange that follows must pair up on every rank
            ok = ok and held > 100.0
            from dir_amd import ops as _ops
            _ops.invalidate_caches()                           # a checkpoint load: every rank re-measures together
            ok = ok and absmax(every=32) == gmax() and calls[0] == want_calls + 1
            out_q.put((rank, bool(ok), calls[0], 0))
            return
        if train:
            st = ShardedTables(local, vocab, backend=OracleBackend(), **kw).enable_training(lr=0.05, initial_accumulator_value=0.1)
            gout = rng_b.standard_normal((B, F * K)).astype(np.float32)
            fb0 = None
            if opts.get("infer_first"):
                # de-duplicated inference lookups shrink THEIR slabs; the training pipeline (never de-duplicated) keeps its own
                # capacity.  Warm both (the skewed ids make the first training lookup grow its slabs once), then: inference, training,
                # inference -- no lookup may overflow and run twice any more
                for _ in range(int(opts["infer_first"])):
                    st.lookup(torch.from_numpy(ids))
                st.lookup_train(torch.from_numpy(ids))
                st.lookup(torch.from_numpy(ids))
                fb0 = st.stats["fallbacks"]
            emb = st.lookup_train(torch.from_numpy(ids))
            fwd_ok = bool(np.array_equal(emb.detach().numpy(), R.embedding_bag_onehot(full, ids)))
            if fb0 is not None:
                fwd_ok = fwd_ok and st.stats["fallbacks"] == fb0 and st._cap < st._cap0 <= st._cap_train
                st.lookup(torch.from_numpy(ids))                         # ... and the next inference lookup still uses its shrunk slabs
                fwd_ok = fwd_ok and st.stats["fallbacks"] == fb0 and st.stats["cap"] == st._cap < st._cap0
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
                s_, e_ = local_slice(v, parts[f], first[f], world, rank)
                ok = ok and bool(np.allclose(local[f].numpy(), ref[s_:e_], rtol=1e-6, atol=1e-7))
            out_q.put((rank, ok, B, F * K))
            return
        st = ShardedTables(local, vocab, backend=OracleBackend(), **kw)
        ids_ref = ids.copy()
        for f in range(F):
            ids_ref[ids_ref[:, f] >= vocab[f], f] = -1
        ref = R.embedding_bag_onehot(full, ids_ref)
        from oracle import oracle as O
        ok = True
        for rep in range(int(opts.get("repeats", 1))):          # repeated lookups: the capacity policy adapts between them
            got, fm = st.lookup(torch.from_numpy(ids), want_fm=True)
            ok = ok and bool(np.array_equal(got.numpy(), ref)) and bool(np.array_equal(fm.numpy()[:, 0], O.fm_second_order(ref, F, K)))
        got2 = st.lookup(torch.from_numpy(ids))                 # without the FM logit
        ok = ok and bool(np.array_equal(got2.numpy(), ref))
        want = opts.get("expect")
        if want == "fallback":
            ok = ok and st.stats["fallbacks"] >= 1
        elif want == "no_fallback":
            ok = ok and st.stats["fallbacks"] == 0
        elif want == "exact_after":
            ok = ok and st._use_exact
        if opts.get("dedup") and want == "shrinks":
            ok = ok and st.stats["cap"] < st._cap0           # the slabs shrank to what de-duplication left
        # ranks with DIFFERENT local batch sizes (an uneven last batch; one rank may have nothing), two "epochs": only the slab
        # capacity has to agree, and it never depends on a rank's own batch size after the first lookup
        for sizes in opts.get("uneven", []):
            Bu = sizes[rank]
            idu = np.stack([rng_b.integers(-1, v, size=Bu) for v in vocab], axis=1).astype(np.int64).reshape(Bu, F)
            refu = R.embedding_bag_onehot(full, idu) if Bu else np.zeros((0, F * K), np.float32)
            gotu = st.lookup(torch.from_numpy(idu))
            ok = ok and tuple(gotu.shape) == (Bu, F * K) and bool(np.array_equal(gotu.numpy(), refu))
        # the diagnostic form of a lookup (stages back to back, a time per stage): same results
        if opts.get("stages"):
            us, es, fs = st.stage_times(torch.from_numpy(ids), want_fm=True, iters=2)
            ok = ok and set(us) == set(ShardedTables.STAGES) and all(v >= 0 for v in us.values())
            ok = ok and bool(np.array_equal(es.numpy(), ref)) and bool(np.array_equal(fs.numpy()[:, 0], O.fm_second_order(ref, F, K)))
        # the lookup without its finish pass: the consumer gathers from the received rows through the inverse positions itself
        if opts.get("consume"):
            got_c = torch.full((B, F * K), float("nan"))
            calls = []

            def consumer(s_, e_, rows, inv):
                calls.append((s_, e_))
                r3 = rows[inv.clamp(min=0)] * (inv >= 0).unsqueeze(-1).to(rows.dtype)        # [e - s, F, K]; pruned -> zeros
                got_c[s_:e_] = r3.reshape(e_ - s_, F * K)
            fb = st.stats["fallbacks"]
            st.lookup_consume(torch.from_numpy(ids), consumer)
            ok = ok and bool(np.array_equal(got_c.numpy(), ref)) and len(calls) >= 1
            if want == "fallback":          # (slabs still too small on this call: the repair calls the consumer again, for the whole batch)
                ok = ok and (st.stats["fallbacks"] == fb or calls[-1] == (0, B))
            # the handle form (round 6: lookup_rows_async): two lookups in flight, the rows of the first consumed after the second is issued
            ids_b = np.ascontiguousarray(ids[::-1])
            h_a = st.lookup_rows_async(torch.from_numpy(ids))
            h_b = st.lookup_rows_async(torch.from_numpy(ids_b))
            for h_, ref_ in ((h_a, ref), (h_b, ref[::-1])):
                got_r = torch.full((B, F * K), float("nan"))
                cover = 0
                for s_, e_, rows, inv in h_.result():
                    r3 = rows[inv.clamp(min=0)] * (inv >= 0).unsqueeze(-1).to(rows.dtype)
                    got_r[s_:e_] = r3.reshape(e_ - s_, F * K)
                    cover += e_ - s_
                ok = ok and cover == B and bool(np.array_equal(got_r.numpy(), ref_))
        # two lookups in flight (double-buffered plans), consumed in order, then a third reusing the first one's buffers
        if opts.get("async"):
            batches = [np.stack([rng_b.integers(-1, v, size=B) for v in vocab], axis=1).astype(np.int64) for _ in range(3)]
            h0 = st.lookup_async(torch.from_numpy(batches[0]), want_fm=True)
            h1 = st.lookup_async(torch.from_numpy(batches[1]))
            e0, f0 = h0.result()
            h2 = st.lookup_async(torch.from_numpy(batches[2]), want_fm=True)
            e1 = h1.result()
            e2, f2 = h2.result()
            for e_, b_ in ((e0, batches[0]), (e1, batches[1]), (e2, batches[2])):
                ok = ok and bool(np.array_equal(e_.numpy(), R.embedding_bag_onehot(full, b_)))
            ok = ok and bool(np.array_equal(f0.numpy()[:, 0], O.fm_second_order(R.embedding_bag_onehot(full, batches[0]), F, K)))
        out_q.put((rank, ok, int(got.shape[0]), int(got.shape[1])))
    finally:
        dist.destroy_process_group()


def _run(world, vocab, K, B, seed, train=False, opts=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, vocab, K, B, seed, q, train, opts)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == list(range(world))
    for rank, ok, b, w in res:
        assert not isinstance(ok, str), "rank %d raised:\n%s" % (rank, ok)
        assert ok, "rank %d: sharded result differs from the full-table computation" % rank
    return res


@pytest.mark.parametrize("world,vocab,opts", [
    (2, [10, 7, 33], {"expect": "no_fallback"}),                                    # fixed-capacity pipeline, default slack
    (2, [10, 7, 33], {"expect": "no_fallback", "consume": True}),                  # ... and the lookup without its finish pass
    (3, [100, 5, 64, 9], {"chunks": 3, "out_of_range": True, "consume": True}),
    (2, [50, 50, 50], {"slack": 0.5, "mode": "fixed", "expect": "fallback", "consume": True}),   # the overflow repair re-runs the consumer
    (3, [100, 5, 64, 9], {"chunks": 3, "out_of_range": True}),
    (2, [1000] * 6, {"chunks": 1}),
    (2, [50, 50, 50], {"slack": 0.5, "mode": "fixed", "expect": "fallback", "repeats": 2}),   # slabs too small: exact fallback, then grown
    (3, [3000, 3000], {"id_hi": 40, "chunks": 1, "expect": "exact_after", "repeats": 2}),         # every id owned by rank 0: auto gives up on slabs
    (2, [200, 300, 100], {"dedup": True, "id_hi": 12, "chunks": 2, "repeats": 4, "B": 600, "expect": "shrinks"}),   # duplicates sent once
    (3, [64, 64, 9, 200], {"dedup": True, "slack": 0.6, "mode": "fixed", "repeats": 3}),
    (3, [40, 90, 64, 9, 17], {"partitions": [1, 3, 2, 1, 2]}),                       # the partitioner rule's slice counts, round-robin placement
    (2, [10, 7, 33], {"mode": "exact"}),
    (2, [30, 30], {"check": "lazy", "repeats": 2}),
    (2, [40, 25, 60], {"uneven": [[37, 20], [11, 37], [0, 9], [64, 3]]}),          # unequal local batches, incl. an empty one
    (3, [40, 25, 60], {"uneven": [[5, 37, 20], [50, 1, 0]], "dedup": True, "id_hi": 15}),
    (2, [50, 20, 33], {"async": True}),
    (3, [50, 20, 33, 7], {"stages": True, "chunks": 2}),
    (2, [50, 20, 33, 7], {"stages": True, "dedup": True, "id_hi": 9}),
    (3, [50, 20, 33], {"async": True, "dedup": True, "chunks": 3, "check": "lazy"}),
])
def test_sharded_lookup_matches_full_tables(world, vocab, opts):
    K, B = 8, opts.get("B", 37)
    res = _run(world, vocab, K, B, 4321, opts=opts)
    for rank, ok, b, w in res:
        assert (b, w) == (B, len(vocab) * K)


@pytest.mark.parametrize("world,vocab,opts", [(2, [10, 7, 33], None), (3, [40, 5, 64, 9], None), (3, [40, 5, 64, 9], {"partitions": [2, 1, 3, 1]}),
                                              (2, [200, 300, 100], {"dedup": True, "id_hi": 12, "chunks": 2, "B": 600, "infer_first": 3})])
def test_sharded_training_step_matches_full_table_adagrad(world, vocab, opts):
    """lookup_train + backward over gloo: every rank's row gradients reach the owners (the forward exchange reversed) and
    the owners' shards end up equal to one synchronous Adagrad step on the full tables over all ranks' batches."""
    _run(world, vocab, 8, (opts or {}).get("B", 29), 777, train=True, opts=opts)


def test_sharded_absmax_collective_is_rank_invariant():
    """Two ranks: predict-style absmax(), 40 owner-side updates that move only rank 1's largest value, absmax(every=32) after each: both
    ranks issue the SAME number of MAX all-reduces (one at start, one at update 32, one after invalidate_caches) and read the same value."""
    res = _run(2, [10, 7, 33], 8, 16, 99, opts={"absmax": True})
    assert len({r[2] for r in res}) == 1 and res[0][2] == 3


def test_partitioner_slice_count_rule():
    """min_max_variable_partitioner(max_partitions=num_ps_replicas, min_slice_size=64 << 20) as called at deepFM.py:163-167:
    slices = max(1, min(rows, max_partitions, ceil(bytes / min_slice_size))); KATs from the BASELINE configurations."""
    import sys
    sys.path.insert(0, ROOT)
    from dir_amd.shard import partitions_for, place_slices, local_slice
    assert partitions_for(1_000_000, 16, 8) == 1            # cfg 2: 64 000 000 B < 64 MiB -> ceil(0.954) = 1: NOT split
    assert partitions_for(1_048_576, 16, 8) == 1            # exactly 64 MiB: still one slice
    assert partitions_for(1_048_577, 16, 8) == 2            # one row more: ceil(1.0000009) = 2
    assert partitions_for(10_000_000, 64, 8) == 8           # cfg 4: 2.56 GB -> 39 wanted, capped by the 8 servers
    assert partitions_for(100_000_000, 16, 8) == 8          # cfg 5
    assert partitions_for(100_000_000, 16, 128) == 96       # 6.4e9 / 64 MiB = 95.4 -> 96
    assert partitions_for(5, 1 << 26, 8) == 5               # never more slices than rows
    assert partitions_for(1_000_000, 16, 0) == 1            # num_ps_replicas = 0 (deepFM.py:162): unpartitioned
    assert partitions_for(3_000_000, 16, 8, min_slice_size=256 << 10, bytes_per_element=4) == 8
    parts = [1, 8, 3, 1]
    first = place_slices(parts, 8)
    assert first == [0, 1, 1, 4]                            # slices dealt round-robin over the ranks in creation order
    # every row of every table is held by exactly one rank
    for v, p, f0 in zip([10, 1000, 7, 3], parts, first):
        cover = np.zeros(v, int)
        for r in range(8):
            s, e = local_slice(v, p, f0, 8, r)
            cover[s:e] += 1
        assert (cover == 1).all()


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
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        import datetime
        dist.init_process_group("gloo", init_method="file://" + port, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
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
