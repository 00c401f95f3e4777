#!/usr/bin/env python3
"""bench.py -- throughput of the embedding-lookup + feature-interaction hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM.

Default workload (BASELINE.json configs[1], the configuration the metric is quoted on):
  deepfm_gather_fm: DeepFM batch 65 536, 26 sparse fields x 1M vocab x dim 16 -- the fused multi-slot
  gather + FM second-order kernel, writing the [B, 416] concat the DNN consumes and the [B] FM logit.
Other workloads (--workload) time the other configs' kernels for DESIGN.md / profiles; they are not
the driver's bench line.

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling.  Every rank owns the 'div'
row shard of every table, draws its own batch of 65 536 samples over the GLOBAL vocabulary, and a step
is ShardedTables.lookup (route -> all_to_all ids -> owner gather -> all_to_all rows -> un-permute) followed
by the FM kernel.  value = samples all ranks processed / max-over-ranks time.

The default line also carries two secondary legs, timed after the primary steps: `secondary_zipf` (N = 1: the same kernel on
Zipf(1.05) ids) and `secondary_cfg5_xdeepfm_cin` (every N: BASELINE configs[4], xDeepFM CIN 3 x 128 on a 1e8-row table,
row-sharded when N > 1; if this leg hangs a watchdog prints the primary line and exits with code 3).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0      # measured float4 copy ceiling (same guide)
XGMI_LINK_GBS = 153.6      # one xGMI link, one direction (7 per GPU, point to point)
MFMA_F32_PEAK_TF = 157.3   # fp32-input MFMA peak
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak (the 2:1-sparsity figure is not used)
# What one fp32 multiply-add costs on the matrix pipe, in bf16-MFMA flops: the bf16x3 kernels issue six bf16 piece products per fp32
# product; an fp32-input MFMA runs at 1/16 of the bf16 rate (MI355X_MICROARCH.md, Matrix cores).  Every MFMA-bound line prices the
# ALGORITHMIC flops of each kernel with the factor of the pipe mode that kernel executes on and divides by the dense bf16 peak, so
# `frac` is the share of the step the matrix pipe would need at peak: never above 1.
PIPE_COST = {"bf16x3": 6.0, "f32": 16.0, "f16x2": 3.0}      # f16x2: two fp16 pieces per operand, three products (csrc/cin_bf3.hip, round 4)


def _cin_fwd_label(ops):
    f16 = getattr(ops, "CIN_FWD_SPLIT", "") == "f16x2" and ops.CIN_ARITH == "auto"
    return "fp16x2 split (layers 1-2) / pooled last layer" if f16 else "bf16x3 split"


def cin_flops(ops, B, m, D, Hs, arith=None, forward=True, backward=False):
    """CIN stack -> (fp32-equivalent algorithmic flops, the same in bf16-pipe flops, {kernel: arithmetic}).  Forward: one
    [B*D, Hp*m] x [Hp*m, H] contraction per layer.  Backward: two of that size per layer -- T = G x W (both data gradients, one
    pass) and the weight gradient -- each on the arithmetic ops.cin_layer_backward's "auto" rule picks."""
    arith = arith or ops.CIN_ARITH
    alg = pipe = 0.0
    modes = {}
    hp = m
    for k, h in enumerate(Hs):
        f = 2.0 * B * D * hp * m * h
        if (k == len(Hs) - 1 and arith in ("auto", "bf16x3") and getattr(ops, "CIN_POOLED_LAST", False) and ops.cin_pooled_covers(m, D, hp)
                and h % 4 == 0):
            # The last layer's map only feeds its pooled sums: the sum over d is taken first (csrc/cin_pool.hip, on the vector ALUs) and the
            # contraction with W is ONE dense product on [B, hp*m] rows -- 1/D of the definition's flops reach the matrix pipe; the backward
            # is two more such products (dW on dense_dw's arithmetic, dZ on the dense kernel's).
            if forward and not backward and getattr(ops, "CIN_POOLED_FUSED", False) and ops.cin_pooled_fused_covers(m, hp, h, D):
                # round 6, inference: the two passes fused (csrc/cin_pooled.hip): bf16 x 3, one 32-wide k-step per input channel (m fields padded to 32)
                alg, pipe = alg + f, pipe + f / D * (32.0 / m) * PIPE_COST["bf16x3"]
                modes["fwd%d" % (k + 1)] = "pooled, fused (Z in registers; bf16x3, 1/D of the flops on the pipe, fields padded to 32)"
            elif forward:
                alg, pipe = alg + f, pipe + f / D * PIPE_COST["bf16x3" if ops.dense_auto_arith(B, hp * m, h) == "bf16x3" else "f32"]
                modes["fwd%d" % (k + 1)] = "pooled (sum over d first; 1/D of the flops on the pipe)"
            if backward:
                dwa = ops.dense_dw_auto_arith(B, hp * m, h)
                dza = ops.dense_auto_arith(B, h, hp * m)
                alg = alg + 2 * f
                pipe = pipe + f / D * (PIPE_COST["bf16x3" if dwa == "bf16x3" else "f32"] + PIPE_COST["bf16x3" if dza == "bf16x3" else "f32"])
                modes["dx%d" % (k + 1)] = modes["dw%d" % (k + 1)] = "pooled"
            hp = h
            continue
        if forward:
            a = ops.cin_auto_arith(m, D, hp, h) if arith == "auto" else arith
            if arith == "auto" and a == "bf16x3" and getattr(ops, "CIN_FWD_SPLIT", "bf16x3") == "f16x2":
                a = "f16x2"                       # what "auto" gives a forward layer (ops.cin_layer)
            if k == 0 and a in ("bf16x3", "f16x2") and getattr(ops, "CIN_L1_PAIRS", False) and 8 <= m <= 40:
                # the first layer (xk is x0): dir_cin_layer1_bf16x3_f32 multiplies the m (m + 1) / 2 unordered pairs only (padded to 64-pair halves)
                # -- priced on the reduction slots it executes
                slots = -(-(m * (m + 1) // 2) // 64) * 64
                alg, pipe = alg + f, pipe + f * slots / float(m * m) * PIPE_COST[a]
                modes["fwd1"] = a + "_pairs"
            else:
                alg, pipe = alg + f, pipe + f * PIPE_COST[a]
                modes["fwd%d" % (k + 1)] = a
        if backward:
            bwd16 = arith == "auto" and getattr(ops, "CIN_BWD_SPLIT", "bf16x3") == "f16x2"     # "auto": gradient operands on scaled fp16 x 2
            a = (ops.cin_auto_arith(m, D, h, hp) if arith == "auto" else arith) if ops.cin_bf16x3_covers(m, D) else "f32"
            if bwd16 and a == "bf16x3":
                a = "f16x2"
            alg, pipe = alg + f, pipe + f * PIPE_COST[a]
            modes["dx%d" % (k + 1)] = a
            a = ops.cin_dw_auto_arith(m, D, hp, h) if arith == "auto" else arith
            if bwd16 and a == "bf16x3":
                a = "f16x2"
            if k == 0 and arith in ("auto", "bf16x3") and getattr(ops, "CIN_DW_SYM", False) and D in (8, 16, 32) and m <= 64:
                # the first layer (xk is x0): dir_cin_dw_sym_bf16x3_f32 multiplies the m (m + 1) / 2 unordered pairs only -- priced on what it executes
                sym = "f16x2" if bwd16 else "bf16x3"          # (the stack's backward hands the fp16 x 2 kernel the tensor scale of its contraction)
                alg, pipe = alg + f, pipe + f * (m + 1) / (2.0 * m) * PIPE_COST[sym]
                modes["dw1"] = sym + "_sym"
            else:
                alg, pipe = alg + f, pipe + f * PIPE_COST[a]
                modes["dw%d" % (k + 1)] = a
        hp = h
    return alg, pipe, modes


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="deepfm_gather_fm",
                    choices=["deepfm_gather_fm", "gather_only", "fm_only", "linear", "dcn_cross", "dcn_cross_backward", "din", "din_train", "cin", "cin_backward", "deepfm_full", "multihot_bag", "deepfm_train", "dcn_train", "xdeepfm_full", "xdeepfm_train",
                             "sharded_1gpu", "sharded_deepfm_1gpu", "transform", "dcn_full", "train_sparse", "small_batch", "deepfm_sparse_packed", "mlp_dense", "esmm_full", "esmm_train", "deepfm_full_packed",
                             "din_full"])
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--fields", type=int, default=26)
    ap.add_argument("--vocab", type=int, default=1000000)
    ap.add_argument("--dim", type=int, default=16)
    ap.add_argument("--id-dist", default="uniform", choices=["uniform", "zipf"])
    ap.add_argument("--id-layout", default="bf", choices=["bf", "fb"], help="ids stored [B,F] or [F,B]")
    ap.add_argument("--rotate", type=int, default=4, help="distinct id batches rotated through")
    ap.add_argument("--adagrad-method", default="sorted", choices=["sorted", "chains"], help="train_sparse: SparseAdagrad method")
    ap.add_argument("--train-layout", default="packed", choices=["packed", "split", "split_unfused"],
                    help="train_sparse: packed = [embedding | accumulator] rows + FM backward folded into the update; split = separate "
                         "[V,K] tables and accumulators (the reference's variable layout) with the fold; split_unfused = round 1's path")
    ap.add_argument("--cin-arith", default=None, choices=["auto", "f32", "bf16x3"],
                    help="cin: arithmetic of ops.cin_layer (default: the library default, 'auto' = bf16x3 where covered)")
    ap.add_argument("--cross-d", type=int, default=416, help="dcn_cross / dcn_cross_backward: row width (416 = 26 x 16; 429 with the 13 dense)")
    ap.add_argument("--graph", action="store_true",
                    help="capture the step in HIP graphs (torch.cuda.CUDAGraph, one per rotated id batch) after the warmup and time the replays: "
                         "what a launch-bound step costs without the host's launch gaps (training workloads; N = 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--torch-profile", default=None, help="development: write torch.profiler's device time per (op, input shapes) of 4 untimed "
                                                          "steps after the warmup to this file (which torch ops a step still runs beside the HIP kernels)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-child", action="store_true", help="internal: the cpu_baseline worker process (see cpu_child_main)")
    return ap.parse_args()


def make_ids(torch, args, gen, device, vocab):
    B, F = args.batch, args.fields
    out = []
    for _ in range(args.rotate):
        if args.id_dist == "uniform":
            ids = torch.randint(0, vocab, (B, F), generator=gen, device=device)
        else:  # Zipf(1.05) by inverse-CDF on a power law, clipped to the vocabulary
            u = torch.rand((B, F), generator=gen, device=device, dtype=torch.float64)
            a = 1.05
            ids = ((vocab ** (1 - a) - 1) * u + 1).pow(1 / (1 - a)).floor().long().clamp_(1, vocab) - 1
        if args.id_layout == "fb":
            ids = ids.t().contiguous().t()
        out.append(ids)
    return out


_SYNTH_C1 = 0x9E3779B97F4A7C15 - (1 << 64)      # splitmix64's constants as signed 64-bit integers (torch int64 arithmetic wraps)
_SYNTH_C2 = 0xBF58476D1CE4E5B9 - (1 << 64)
_SYNTH_C3 = 0x94D049BB133111EB - (1 << 64)
_SYNTH_STD = 37837.22703        # sqrt(4 * (65536^2 - 1) / 12): the standard deviation of a sum of four uniform 16-bit integers


def _wrap64(v):
    """A Python integer reduced to the signed 64-bit value torch's wrapping int64 arithmetic would hold."""
    v &= (1 << 64) - 1
    return v - (1 << 64) if v >> 63 else v


def synth_rows(torch, f, rows, K, sigma):
    """Rows `rows` (int64 [n], GLOBAL row ids) of synthetic table f as a closed-form function of (f, row, column): an integer hash
    (splitmix64's finaliser on f, row * K + column), its four 16-bit fields summed (Irwin-Hall: close to normal), centred and scaled to
    standard deviation sigma.  Integer arithmetic, one exact int -> float conversion and one fp32 multiply: the same bits on every
    device and rank, so at N > 1 every rank can regenerate the rows a lookup must return without holding the other ranks' shards."""
    x = rows.reshape(-1, 1) * K + torch.arange(K, device=rows.device, dtype=torch.int64) + _wrap64((f + 1) * _SYNTH_C1)
    x = (x ^ ((x >> 30) & ((1 << 34) - 1))) * _SYNTH_C2          # (logical shifts: torch's >> on int64 is arithmetic)
    x = (x ^ ((x >> 27) & ((1 << 37) - 1))) * _SYNTH_C3
    x = x ^ ((x >> 31) & ((1 << 33) - 1))
    s = (x & 0xffff) + ((x >> 16) & 0xffff) + ((x >> 32) & 0xffff) + ((x >> 48) & 0xffff)
    return (s - 131070).to(torch.float32) * (sigma / _SYNTH_STD)


def synth_shard(torch, f, start, end, K, sigma, device, block=1 << 20):
    """Rows [start, end) of synthetic table f on `device` (generated in blocks: the int64 temporaries are 4 x the block's bytes)."""
    t = torch.empty((end - start, K), dtype=torch.float32, device=device)
    for s in range(start, end, block):
        e = min(end, s + block)
        t[s - start:e - start] = synth_rows(torch, f, torch.arange(s, e, device=device, dtype=torch.int64), K, sigma)
    return t


def sharded_parity_check(torch, dist, ops, st, ids, K, sigma, world, backend, device, rows=4096):
    """Before any timing at N > 1: every rank looks up the first `rows` samples of its batch through the sharded pipeline and compares
    the result BIT FOR BIT with the rows regenerated from the closed form (synth_rows) -- which needs no other rank's memory -- and the
    fused FM logit with the FM kernel run on those regenerated rows.  Also counts the ranks a collective actually reaches.
    -> the `parity_check` object of the line (identical on every rank: MIN / SUM over the ranks)."""
    n = min(rows, ids.shape[0])
    chk = ids[:n].contiguous()
    F = chk.shape[1]
    emb, fm = st.lookup(chk, want_fm=True)
    want = torch.cat([synth_rows(torch, f, chk[:, f], K, sigma) for f in range(F)], dim=1)
    want_fm = ops.fm_logit(want, F, K)
    torch.cuda.synchronize()
    ok_rows = bool(torch.equal(emb, want))
    ok_fm = bool(torch.equal(fm, want_fm))
    nz = float((want != 0).float().mean())
    cdev = device if backend == "nccl" else "cpu"
    t = torch.tensor([1.0, float(ok_rows), float(ok_fm)], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    t = t.tolist()
    return {"ok": int(t[1]) == world and int(t[2]) == world and nz > 0.99, "ranks_seen": int(t[0]), "ranks_rows_bit_exact": int(t[1]),
            "ranks_fm_bit_exact": int(t[2]), "samples_per_rank": n, "rows_per_rank": n * F,
            "against": "closed-form synthetic rows (bench.synth_rows), regenerated on the checking rank; FM logit against dir_fm_second_order_f32 on them",
            "fallbacks": st.stats["fallbacks"]}


def sharded_stage_times(torch, dist, st, ids, world, backend, device, iters=5):
    """Per-stage microseconds of one sharded lookup (ShardedTables.stage_times: the stages back to back on one stream, HIP events
    between them), MAX over the ranks."""
    us, _, _ = st.stage_times(ids, want_fm=True, iters=iters)
    names = list(st.STAGES)
    t = torch.tensor([us[k] for k in names], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    out = {k: round(v, 2) for k, v in zip(names, t.tolist())}
    out["sum"] = round(sum(out.values()), 2)
    out["note"] = ("stages of one lookup run back to back on one stream (max over ranks, median of %d); the timed steps pipeline two "
                   "chunks on two side streams and neighbouring lookups, so a step costs less than this sum" % iters)
    return out


def torch_profile(torch, step, path, n=4):
    """--torch-profile: device time per step of every (op, input shapes) and kernel, largest first."""
    from torch.profiler import profile, ProfilerActivity
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for i in range(n):
            step(i)
        torch.cuda.synchronize()
    rows = []
    for e in prof.key_averages(group_by_input_shape=True):
        t = getattr(e, "self_device_time_total", None)
        if t is None:
            t = e.self_cuda_time_total
        if t > 0:
            rows.append((t / n, e.count / n, e.key, str(e.input_shapes)[:120]))
    rows.sort(reverse=True)
    with open(path, "w") as f:
        for t, c, k, sh in rows[:120]:
            f.write("%9.1f us  x%-6.1f %-60s %s\n" % (t, c, k[:60], sh))


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_topology():
    """(threads to run, physical cores available, logical cpus available, cgroup cpu quota or None) for THIS process: the cores of
    the affinity mask (SMT siblings counted once), capped by the cgroup's CPU quota -- more runnable threads than the quota allows
    are throttled by the scheduler, which is what made round 2's 256-thread passes spread 0.67 .. 90 ms."""
    cpus = sorted(os.sched_getaffinity(0))
    cores = set()
    for c in cpus:
        try:
            sib = open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip()
        except OSError:
            sib = str(c)
        cores.add(sib)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        pass
    phys = len(cores)
    n = phys if quota is None else max(1, min(phys, int(quota)))
    return n, phys, len(cpus), quota


def cpu_child_main():
    """`bench.py --cpu-child`: the cpu_baseline leg's worker.  A fresh process started by bench.py BEFORE anything touches the GPU (so it
    is an ordinary child, never a re-exec of a GPU process) that never imports torch: it pins OpenMP to physical cores
    (OMP_PLACES=cores, OMP_PROC_BIND=close, one thread per core, set before libgomp is loaded with the oracle), then serves one JSON
    request per stdin line: generate the workload's synthetic inputs (NumPy, same shapes and distributions as the GPU leg), run the
    oracle op (oracle/dir_oracle.c: the C/OpenMP port of the reference op sequence) for about `seconds`, report the median pass."""
    n, phys, logical, quota = _cpu_topology()
    if os.environ.get("DIR_BENCH_CPU_THREADS"):
        n = int(os.environ["DIR_BENCH_CPU_THREADS"])
    os.environ["OMP_NUM_THREADS"] = str(n)
    os.environ["OMP_PLACES"] = "cores"
    os.environ["OMP_PROC_BIND"] = "close"
    import numpy as np
    from oracle import oracle as O
    O.build()
    O.lib()
    topo = {"cores": n, "physical_cores_available": phys, "logical_cpus_available": logical, "cgroup_cpu_quota": quota,
            "omp": "OMP_NUM_THREADS=%d OMP_PLACES=cores OMP_PROC_BIND=close" % n, "cpu_model": _cpu_model()}
    for line in sys.stdin:
        req = json.loads(line)
        rng = np.random.default_rng(1234)
        wl, B, secs = req["workload"], int(req["samples"]), float(req["seconds"])
        if wl == "gather_fm":
            F, K, V = req["fields"], req["dim"], req["vocab"]
            tables = [rng.standard_normal((V, K), dtype=np.float32) * np.float32(1.0 / K ** 0.5) for _ in range(F)]
            ids = rng.integers(0, V, size=(B, F), dtype=np.int64)
            emb = np.zeros((B, F * K), np.float32)     # allocated once and touched: a fresh 109 MB buffer per pass times page faults

            def run():
                O.embedding_bag(tables, ids, out=emb)
                O.fm_second_order(emb, F, K)
            what = "gather+FM (per-field lookup -> concat -> FM four-op form) over %d samples x %d fields (dim %d, vocab %d)" % (B, F, K, V)
        elif wl == "dcn_cross":
            d, L = req["d"], req["layers"]
            x0 = rng.standard_normal((B, d), dtype=np.float32) * np.float32(0.25)
            w = np.clip(rng.standard_normal((L, d), dtype=np.float32) * np.float32(0.1), -0.2, 0.2)
            bb = np.clip(rng.standard_normal((L, d), dtype=np.float32) * np.float32(0.1), -0.2, 0.2)
            run = lambda: O.dcn_cross(x0, w, bb)  # noqa: E731
            what = "%d cross layers (DeepCrossNetwork.py:345-346 op order) over %d rows x %d" % (L, B, d)
        elif wl == "din":
            T, K, V, H1, H2 = req["T"], req["dim"], req["vocab"], req["H1"], req["H2"]
            table = rng.standard_normal((V, K), dtype=np.float32) * np.float32(0.125)
            hist = rng.integers(0, V, size=(B, T), dtype=np.int64)
            hl = rng.integers(1, T + 1, size=B).astype(np.int32)
            cand = rng.integers(0, V, size=B, dtype=np.int64)
            W1 = rng.standard_normal((4 * K, H1), dtype=np.float32) * np.float32(0.05)
            W2 = rng.standard_normal((H1, H2), dtype=np.float32) * np.float32(0.1)
            W3 = rng.standard_normal((H2,), dtype=np.float32) * np.float32(0.1)
            z = lambda k: np.zeros(k, np.float32)  # noqa: E731
            run = lambda: O.din_attention_pool(table, hist, hl, cand, W1, z(H1), W2, z(H2), W3, z(1), normalize=True)  # noqa: E731
            what = "DIN attention pool (T %d, dim %d, table %d rows, MLP %d-%d-%d-1, softmax) over %d samples" % (T, K, V, 4 * K, H1, H2, B)
        elif wl == "cin":
            m, D, Hs = req["m"], req["D"], req["layers"]
            x0 = rng.standard_normal((B, m, D), dtype=np.float32) * np.float32(0.25)
            Ws, hp = [], m
            for h in Hs:
                Ws.append(rng.standard_normal((h, hp * m), dtype=np.float32) * np.float32(1.0 / (hp * m) ** 0.5))
                hp = h

            def run():
                xk = x0
                for W in Ws:
                    xk, _ = O.cin_layer(x0, xk, W)
            what = "CIN %s (m %d, D %d) over %d samples" % ("x".join(str(h) for h in Hs), m, D, B)
        else:
            print(json.dumps({"error": "unknown workload %r" % wl}), flush=True)
            continue
        times = []
        t_all = time.perf_counter()
        warm = int(req.get("warm", 2))
        for p in range(warm + 100000):
            t0 = time.perf_counter()
            run()
            dt = time.perf_counter() - t0
            if p >= warm:
                times.append(dt)
            if len(times) >= int(req.get("min_passes", 5)) and time.perf_counter() - t_all >= secs:
                break
        times.sort()
        med = times[len(times) // 2]
        res = dict(topo)
        res.update({"value": B / med, "unit": "samples/s", "kind": "port", "median_pass_ms": med * 1e3,
                    "p10_pass_ms": times[len(times) // 10] * 1e3, "p90_pass_ms": times[(len(times) * 9) // 10] * 1e3,
                    "p90_over_p10": times[(len(times) * 9) // 10] / times[len(times) // 10],
                    "sample": "median of %d passes (after %d untimed) of %s; oracle/dir_oracle.c (C/OpenMP port of the reference op sequence) on %d "
                              "threads pinned one per physical core, %.1f s in all" % (len(times), warm, what, n, time.perf_counter() - t_all)})
        print(json.dumps(res), flush=True)


class CpuBaseline:
    """Parent side: the child is started first thing in main() (before torch / HIP are touched) and sits blocked on its stdin until
    the GPU timing is over; ask() then sends one request and waits for the answer."""

    def __init__(self):
        import subprocess
        self.p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-child"], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                  text=True, cwd=ROOT)

    def ask(self, req, timeout=600.0):
        import select
        try:
            self.p.stdin.write(json.dumps(req) + "\n")
            self.p.stdin.flush()
            r, _, _ = select.select([self.p.stdout], [], [], timeout)
            if not r:
                return {"error": "cpu_baseline child timed out"}
            line = self.p.stdout.readline()
            return json.loads(line) if line.strip() else {"error": "cpu_baseline child exited (rc %s)" % self.p.poll()}
        except (OSError, ValueError) as exc:
            return {"error": repr(exc)[:200]}

    def close(self):
        try:
            self.p.stdin.close()
            self.p.wait(timeout=10)
        except Exception:
            self.p.kill()


def measured_ceilings(torch, ops_lib, slab, stream_ptr, sizes=(("at_gather_size", 109_051_904), ("asymptotic", 536_870_912)), iters=12,
                      scratch_bytes=1 << 30):
    """This box's streaming ceilings, measured in this run with dir_debug_stream_{read,copy}_f32 (csrc/diag.hip: 16 B per lane, four loads in
    flight per lane, non-temporal loads): a linear read of n bytes and a copy of n bytes (read n + write n), for n = the gather's own row
    bytes (109 MB: a ~20 us kernel, ramp and tail included -- what a kernel of the gather's size can reach) and n = 512 MB (the asymptotic
    rate).  `slab` is ONE allocation holding all tables (1.66 GB at config 2): the source window ROTATES through it, and the destination
    window rotates through a separate scratch of `scratch_bytes` (1 GiB), so no window is touched again before >= 0.9 GB of other traffic
    went by -- neither side can sit in the 256 MiB Infinity Cache.  -> {label: {"bytes": n, "read_GBps": .., "copy_GBps": .., "copy_us": ..}}.
    (round 3's version fell into a 64 MB-window branch with a fixed 64 MB destination that did fit the Infinity Cache.)"""
    import ctypes
    flat = slab.reshape(-1)
    total = flat.numel()
    scratch = torch.empty(min(scratch_bytes, max(total * 4, 4096)) // 4, dtype=torch.float32, device=slab.device)
    sink = torch.zeros(4, device=slab.device)
    out = {}
    for label, nbytes in sizes:
        n = min(nbytes // 4, total, scratch.numel()) // 4 * 4
        if n <= 0:
            continue
        src_w = [o for o in range(0, total - n + 1, n)] or [0]
        dst_w = [o for o in range(0, scratch.numel() - n + 1, n)] or [0]
        res = {"bytes": n * 4, "src_windows": len(src_w), "dst_windows": len(dst_w)}
        for name in ("read", "copy"):
            def run(i):
                sp = ctypes.c_void_p(flat.data_ptr() + src_w[i % len(src_w)] * 4)
                if name == "read":
                    rc = ops_lib.dir_debug_stream_read_f32(sp, n, ctypes.c_void_p(sink.data_ptr()), stream_ptr)
                else:
                    rc = ops_lib.dir_debug_stream_copy_f32(sp, ctypes.c_void_p(scratch.data_ptr() + dst_w[i % len(dst_w)] * 4), n, stream_ptr)
                assert rc == 0
            for i in range(3):
                run(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(iters):
                run(3 + i)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / iters
            res[name + "_GBps"] = (n * 4 * (1 if name == "read" else 2)) / (us * 1e-6) / 1e9
            res[name + "_us"] = us
        out[label] = res
    del scratch
    return out


def primary_line(args, wl, cfg, roof, units, world, el, dev_ms, traffic, extra):
    """The contract line without the optional legs (used by the watchdog path)."""
    launch_us = dev_ms * 1e3 / args.steps
    res = {"metric": "CTR samples/sec (embedding gather + FM 2nd-order, 26-field batch 65536)", "value": units * world * args.steps / el,
           "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": el * 1e3 / args.steps,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": cfg, "world_size": world}
    ach = roof["alg_bytes"] / (launch_us * 1e-6) / 1e9
    res["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                       "kernel": roof["kernel"], "alg_bytes_per_launch": roof["alg_bytes"], "avg_launch_us": launch_us}
    return res


def cfg5_leg(torch, dist, ops, ShardedTables, div_range, args, world, rank, device, gen, backend, steps=5, warmup=None):
    """BASELINE config 5: xDeepFM CIN (3 x 128, m = 26, D = 16) on a 10^8-row embedding table (26 slots x 3 846 153 rows), the
    table row-sharded 'div' over the ranks with the lookup's two all-to-alls when N > 1.  Per-GPU batch fixed (weak scaling)."""
    B, F, K = args.batch, 26, 16
    # warm-up of this leg: 40 steps (120 ms) by default -- the matrix-pipe clocks need tens of milliseconds of load to come up after the
    # host-side set-up before this leg, and with 2 warm-up steps the 5 timed ones (15 ms) lay inside that ramp: 3.26 vs 2.97 ms on one box
    # (profiles/r06_warm.txt); the fp32-MFMA comparison leg warms up for 10 of its 14 ms steps
    if warmup is None:
        warmup = int(os.environ.get("DIR_BENCH_CFG5_WARMUP", "40"))
    Vf = int(os.environ.get("DIR_BENCH_CFG5_ROWS", "100000000")) // F       # (the env switch: a smaller table for the launch tests)
    Hs = (128, 128, 128)
    sigma = 1.0 / (K ** 0.5)
    sharded = world > 1 or os.environ.get("DIR_BENCH_CFG5_SHARDED") == "1"
    consume = False
    if not sharded:
        ts = ops.TableSet([torch.randn((Vf, K), generator=gen, device=device) * sigma for _ in range(F)])
        st = None
    else:
        # world == 1 with DIR_BENCH_CFG5_SHARDED=1: the sharded code path with its collectives issued (RCCL, one rank) -- what the
        # exchange costs next to the CIN when it is hidden under it
        if world == 1 and not dist.is_initialized():
            import tempfile
            dist.init_process_group("nccl", init_method="file://" + os.path.join(tempfile.mkdtemp(prefix="dir_pg_"), "store"), rank=0,
                                    world_size=1, device_id=device)            # (a FileStore: no TCP port to clash on)
        loc = []
        for f in range(F):
            s0, e0 = div_range(Vf, world, rank)
            loc.append(synth_shard(torch, f, s0, e0, K, sigma, device))       # closed-form rows: the parity check below regenerates them
        # (side streams confined to a few CUs -- ShardedTables(side_cus=n) -- made this leg SLOWER on one GPU: 9.3 vs 7.8 ms at 8-32 CUs,
        # profiles/r03_cfg5_overlap.md; the default leaves them unmasked)
        side = os.environ.get("DIR_BENCH_CFG5_SIDE_CUS", "0")
        # round 6: the lookup WITHOUT its finish pass -- the CIN layers stage x0 through the inverse positions of the received rows
        # (ShardedTables.lookup_rows_async + ops.cin_stack_gather); DIR_BENCH_CFG5_CONSUME=0: lookup_async(out=) + the plain layers
        consume = os.environ.get("DIR_BENCH_CFG5_CONSUME", "1") == "1" and ops.cin_gather_covers(F, K, (128, 128, 128))
        st = ShardedTables(loc, [Vf] * F, force_collective=True, check="eager" if consume else "lazy", max_batch=B, side_cus=int(side) or None)
    idsl = [torch.randint(0, Vf, (B, F), generator=gen, device=device) for _ in range(2)]
    parity = None
    if st is not None:
        parity = sharded_parity_check(torch, dist, ops, st, idsl[0], K, sigma, world, backend, device)
        if not parity["ok"]:
            return {"error": "the sharded lookup failed its parity check before timing", "parity_check": parity}
    Ws, hp = [], F
    for h in Hs:
        Ws.append(torch.randn((h, hp * F), generator=gen, device=device) * (1.0 / (hp * F) ** 0.5))
        hp = h
    pooled = torch.empty((B, sum(Hs)), dtype=torch.float32, device=device)
    embs = [torch.empty((B, F * K), dtype=torch.float32, device=device) for _ in range(2)]
    inflight = {}

    def cin(x0, arith):
        xk, off = x0, 0
        for k, (W, h) in enumerate(zip(Ws, Hs)):
            xk, _ = ops.cin_layer(x0, xk, W, pooled=pooled[:, off:off + h], want_xout=k + 1 < len(Hs), arith=arith)
            off += h

    def step(i, arith):
        if st is None:
            # (round 6: the gather of batch i + 1 on a side stream under the CIN of batch i measured SLOWER -- 3.23 vs 3.10 ms: the persistent CIN
            # workgroups lose the CUs and the bandwidth the gather takes; the local lookup stays in front of its CIN)
            return cin(ops.embedding_bag(ts, idsl[i % 2], out=embs[i % 2]).view(B, F, K), arith)
        # software pipeline: the lookup of batch i+1 is enqueued (side streams) BEFORE the CIN of batch i, so its two exchanges and
        # three kernels run under 7 ms of matrix work: the step costs max(CIN, exchange), not their sum
        if consume and arith is None:
            cur = inflight.pop(i, None) or st.lookup_rows_async(idsl[i % 2])
            inflight[i + 1] = st.lookup_rows_async(idsl[(i + 1) % 2])
            for s_, e_, rows, inv in cur.result():
                ops.cin_stack_gather(rows, inv, Ws, pooled[s_:e_])
            return
        cur = inflight.pop(i, None) or st.lookup_async(idsl[i % 2], out=embs[i % 2])
        inflight[i + 1] = st.lookup_async(idsl[(i + 1) % 2], out=embs[(i + 1) % 2])
        cin(cur.result().view(B, F, K), arith)

    def run(arith, fn=None, warmup=warmup):     # the contract's timing (barrier + synchronize on both sides, max over ranks) for one arithmetic
        fn = fn or step
        inflight.clear()
        for i in range(warmup):
            fn(i, arith)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(warmup, warmup + steps):
            fn(i, arith)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el

    flops, hp = 0, F
    for h in Hs:
        flops += 2 * B * K * hp * F * h
        hp = h
    # Two arithmetics of the same layer: the default is the bf16x3 kernel (every fp32 operand split into three bf16 pieces,
    # six piece products on the bf16 matrix pipe, fp32 accumulate: same 1e-5 parity bar); the fp32-MFMA kernel is timed beside it.
    el = run(None)
    if consume:
        st.lookup(idsl[0], out=embs[0])                                   # (the rows form never wrote the concatenation: the CIN-only leg needs one)
    x_fixed = embs[0].view(B, F, K)
    el_cin = run(None, fn=lambda i, arith: cin(x_fixed, arith))          # the CIN alone on a resident x0: what the lookup adds on top
    el32 = run("f32", warmup=min(10, warmup))
    tf, tf32 = flops * steps / el / 1e12, flops * steps / el32 / 1e12
    default_is_bf3 = ops.CIN_ARITH in ("auto", "bf16x3")
    _, pipe_flops, _ = cin_flops(ops, B, F, K, Hs)          # bf16-pipe flops the default arithmetic executes (the first layer over field pairs)
    if st is not None:
        st.check_overflow()
    return {"metric": "samples/sec (xDeepFM CIN 3x128 + embedding lookup, table 1e8 x 16%s)" % (" row-sharded" if world > 1 else ""),
            "value": B * world * steps / el, "unit": "samples/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": el * 1e3 / steps, "cin_only_ms_per_step": el_cin * 1e3 / steps, "lookup_exposed_frac": (el - el_cin) / el_cin,
            "lookup": ("ShardedTables.lookup_async of batch i+1 issued before the CIN of batch i (2 all_to_all per chunk, 2 chunks, check %s, "
                       "side streams on %s CUs each)%s" % (st.check, st.side_cus or "all", "; NO finish pass: the CIN layers read x0 through the inverse positions "
                                                          "of the received rows (lookup_rows_async + cin_stack_gather)" if consume else "")
                       if st is not None else "local gather (one GPU holds the table)"), "scaling": "weak",
            "dtype": ("f32 via fp16x2 split (layers 1-2) / bf16x3 split (pooled last layer), f32 accumulate"
                      if getattr(ops, "CIN_FWD_SPLIT", "") == "f16x2" and ops.CIN_ARITH == "auto" else "f32 via bf16x3 split, f32 accumulate")
            if default_is_bf3 else "f32",
            "per_gpu_fp32_equiv_TFLOPs_lookup_included": tf,
            "per_gpu_bf16_pipe_TFLOPs_executed": pipe_flops * steps / el / 1e12 if default_is_bf3 else None,
            "per_gpu_frac_of_bf16_mfma_peak": pipe_flops * steps / el / 1e12 / MFMA_BF16_PEAK_TF if default_is_bf3 else None,
            "fp32_mfma_kernel": {"dtype": "f32", "ms_per_step": el32 * 1e3 / steps, "value": B * world * steps / el32,
                                 "per_gpu_TFLOPs_lookup_included": tf32, "per_gpu_frac_of_fp32_mfma_peak": tf32 / MFMA_F32_PEAK_TF},
            "parity_check": parity,
            "config": {"workload": "xdeepfm_cin_sharded", "batch_per_gpu": B, "m": F, "D": K, "layers": list(Hs), "table_rows": Vf * F}}


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): start the N ranks HERE, one fresh child process
    per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment, the same command line), forward rank 0's JSON line and the
    first non-zero exit code.  This parent never imports torch and never touches the GPU: the children are ordinary child processes,
    not re-execs of a process that initialised HIP.  A rank that cannot find its device exits non-zero (main()), which ends the job:
    the line is never printed with fewer ranks than --gpus asked for."""
    import signal
    import socket
    import subprocess
    n = args.gpus
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    base = dict(os.environ)
    import tempfile
    base.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                 "DIR_BENCH_LAUNCHED_BY": "bench.py", "DIR_BENCH_STORE_FILE": os.path.join(tempfile.mkdtemp(prefix="dir_bench_pg_"), "store")})
    base.setdefault("GLOO_SOCKET_IFNAME", "lo")
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: what RCCL needs between processes on these hosts
    base.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    for r in range(n):
        env = dict(base)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "GROUP_RANK": "0"})
        # rank 0's stdout carries the line (forwarded below); the other ranks print nothing on stdout by contract, and whatever a
        # library writes there goes to this process's stderr so it cannot be mistaken for the line
        procs.append(subprocess.Popen(cmd, env=env, cwd=os.getcwd(), stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                      text=(r == 0) or None, start_new_session=True))
    limit = float(os.environ.get("DIR_BENCH_LAUNCH_TIMEOUT", "1500"))
    t0 = time.time()
    rc = 0
    out0 = []
    import threading
    rd = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()), daemon=True)
    rd.start()
    live = set(range(n))
    while live:
        for r in sorted(live):
            c = procs[r].poll()
            if c is not None:
                live.discard(r)
                if c != 0 and rc == 0:
                    rc = c if c > 0 else 128 - c
                    sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n" % (r, c))
        if rc != 0 or time.time() - t0 > limit:
            if rc == 0:
                rc = 124
                sys.stderr.write("bench.py: the %d-rank job did not finish within %.0f s\n" % (n, limit))
            for r in live:                      # exactly the process groups started above
                try:
                    os.killpg(procs[r].pid, signal.SIGTERM)
                except OSError:
                    pass
            t1 = time.time()
            while any(procs[r].poll() is None for r in live) and time.time() - t1 < 10:
                time.sleep(0.1)
            for r in live:
                if procs[r].poll() is None:
                    try:
                        os.killpg(procs[r].pid, signal.SIGKILL)
                    except OSError:
                        pass
            break
        time.sleep(0.05)
    rd.join(timeout=10)
    lines = [l for l in out0 if l.strip()]
    js = [l for l in lines if l.lstrip().startswith("{")]
    for l in lines:
        if l not in js:
            sys.stderr.write(l)
    if js:
        try:
            got = json.loads(js[-1])
            if rc == 0 and got.get("n_gpus") != n:
                sys.stderr.write("bench.py: rank 0 reported n_gpus=%r, --gpus asked for %d\n" % (got.get("n_gpus"), n))
                rc = 5
        except ValueError:
            rc = rc or 5
        if rc in (0, 3):          # 3: the primary line is complete, the config-5 secondary leg hung (see the watchdog in main())
            sys.stdout.write(js[-1] if js[-1].endswith("\n") else js[-1] + "\n")
            sys.stdout.flush()
    elif rc == 0:
        sys.stderr.write("bench.py: rank 0 printed no result line\n")
        rc = 5
    return rc


CPU_BASELINE_WORKLOADS = ("deepfm_gather_fm", "gather_only", "dcn_cross", "din", "cin")


def main():
    args = parse()
    if args.cpu_child:
        return cpu_child_main()
    if args.gpus < 1:
        sys.stderr.write("bench.py: --gpus must be >= 1\n")
        return 2
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)          # the parent of the N ranks: no torch, no HIP in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        # a launcher (torch.distributed.run) started a different number of ranks than --gpus names: refuse rather than print a line whose
        # n_gpus is not what was asked for
        if rank == 0:
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d, or run `python bench.py --gpus %d` "
                             "and let it start the ranks itself)\n" % (args.gpus, world, args.gpus, args.gpus))
        return 2
    cpu = None
    if world == 1 and not args.no_cpu_baseline and args.workload in CPU_BASELINE_WORKLOADS:
        cpu = CpuBaseline()         # started before torch / HIP are touched; idle (blocked on stdin) until the GPU timing is over
    import torch
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    # development switches (a 1-GPU box): DIR_BENCH_BACKEND=gloo + DIR_BENCH_SAME_DEVICE=1 + DIR_SHARD_HOST_STAGED=1 run the
    # multi-rank code path as several processes on cuda:0 with the exchange staged through host memory; the driver's runs
    # use the defaults (one rank per GPU, RCCL)
    backend = os.environ.get("DIR_BENCH_BACKEND", "nccl")
    if os.environ.get("DIR_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    same_device = os.environ.get("DIR_BENCH_SAME_DEVICE") == "1"
    ndev = torch.cuda.device_count()          # counts devices without initialising HIP on this image
    if ndev < (1 if same_device else world):
        sys.stderr.write("bench.py: rank %d: --gpus %d needs %d visible GPU(s), this process sees %d\n" % (rank, world, 1 if same_device else world, ndev))
        if cpu is not None:
            cpu.close()
        return 4
    if world > 1:
        torch.cuda.set_device(local_rank)
        # ranks started by launch_ranks() meet over a FileStore (no TCP port to clash on); under a launcher of the caller's the
        # MASTER_ADDR / MASTER_PORT it exported
        store = os.environ.get("DIR_BENCH_STORE_FILE")
        kw = {"init_method": "file://" + store, "rank": rank, "world_size": world} if store else {}
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), **kw)
        else:
            dist.init_process_group(backend, **kw)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    device = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(device)
    import dir_amd
    dir_amd.load_library()
    from dir_amd import ops
    from dir_amd import autograd as ag_opt      # Adagrad / Ftrl on dense variables (the reference's dnn / linear optimisers)
    from dir_amd.shard import ShardedTables, div_range

    B, F, V, K = args.batch, args.fields, args.vocab, args.dim
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    wl = args.workload
    roof = None
    step = None
    units = B
    cfg = {"workload": wl, "batch": B}
    multi = {}            # N > 1: parity_check / stage_us of the sharded lookup, measured before the timed region

    if wl in ("deepfm_gather_fm", "gather_only", "fm_only", "linear", "deepfm_full", "deepfm_full_packed"):
        sigma = 1.0 / (K ** 0.5)  # [TF-upstream] embedding_column default initializer stddev
        cfg.update({"fields": F, "vocab_per_field": V, "dim": K, "ids": args.id_dist, "id_layout": args.id_layout})
        if world == 1:
            # the F tables are the reference's F separate [V, K] variables; here they are views of ONE allocation so that the ceiling
            # probe below can stream windows larger than one table through the same memory
            slab = torch.randn((F * V, K), generator=gen, device=device).mul_(sigma)
            tables = [slab[f * V:(f + 1) * V] for f in range(F)]
            ts = ops.TableSet(tables)
            if args.id_dist == "zipf":
                ts.row_policy = "reuse"
            cfg["row_policy"] = ts.row_policy
            idsl = make_ids(torch, args, gen, device, V)
            out = torch.empty((B, F * K), dtype=torch.float32, device=device)
            fm = torch.empty((B, 1), dtype=torch.float32, device=device)
            if wl == "deepfm_gather_fm":
                step = lambda i: ops.gather_fm(ts, idsl[i % len(idsl)], out=out, fm=fm)  # noqa: E731
                alg = B * (F * (8 + 2 * 4 * K) + 4)      # ids + rows read + concat written + logit
                kname = "gather_onehot_k<fm,out>"
            elif wl == "gather_only":
                step = lambda i: ops.embedding_bag(ts, idsl[i % len(idsl)], out=out)  # noqa: E731
                alg = B * F * (8 + 2 * 4 * K)
                kname = "gather_onehot_k<out>"
            elif wl == "fm_only":
                ops.embedding_bag(ts, idsl[0], out=out)
                step = lambda i: ops.fm_logit(out, F, K, out=fm)  # noqa: E731
                alg = B * (4 * F * K + 4)
                kname = "fm_k"
            elif wl == "linear":
                wts = ops.TableSet([torch.randn((V,), generator=gen, device=device) * 0.01 for _ in range(F)])
                bias = torch.zeros(1, device=device)
                step = lambda i: ops.linear_logit(wts, idsl[i % len(idsl)], bias=bias, out=fm)  # noqa: E731
                alg = B * (F * (8 + 4) + 4)
                kname = "linear_onehot_k"
            else:  # deepfm_full[_packed]: the whole DeepFM forward (gather+FM, linear term, 400-400-400 tower on dir_dense_f32)
                from dir_amd.deepfm import DeepFM
                from dir_amd import feature_column as fc
                cats = [fc.categorical_column_with_identity("C%d" % i, V) for i in range(F)]
                model = DeepFM(linear_feature_columns=cats, dnn_feature_columns=[fc.embedding_column(c, K) for c in cats],
                               dnn_hidden_units=[400, 400, 400], fm_embedding_size=K).to(device)
                if wl == "deepfm_full_packed":      # serving layout: one 128-byte row per feature value (embedding + first-order weight)
                    model.pack_for_serving()
                def step(i):
                    with torch.no_grad():
                        model.forward_ids(idsl[i % len(idsl)], idsl[i % len(idsl)])
                alg = B * (F * (8 + 2 * 4 * K) + 4)
                kname = "deepfm forward (gather+FM kernel bytes only)"
            roof = {"bound": "hbm", "alg_bytes": alg, "kernel": kname}
            cfg["parallelism"] = "single GPU, tables resident (1.66 GB)"
        else:
            # tables at N > 1: a closed form of (slot, row, column) instead of a generator stream, so that any rank can regenerate the
            # rows its lookups must return (the parity check below) without holding the other ranks' shards
            loc = []
            for f in range(F):
                s, e = div_range(V, world, rank)
                loc.append(synth_shard(torch, f, s, e, K, sigma, device))
            st = ShardedTables(loc, [V] * F, check="lazy", max_batch=B)
            idsl = make_ids(torch, args, gen, device, V)
            multi["parity_check"] = sharded_parity_check(torch, dist, ops, st, idsl[0], K, sigma, world, backend, device)
            if not multi["parity_check"]["ok"]:
                # a number measured on a lookup that returns wrong rows is worth nothing: no line, exit code 6 on every rank
                if rank == 0:
                    sys.stderr.write("bench.py: the sharded lookup FAILED its parity check before timing: %s\n" % json.dumps(multi["parity_check"]))
                dist.destroy_process_group()
                return 6
            multi["stage_us"] = sharded_stage_times(torch, dist, st, idsl[0], world, backend, device)
            cfg["tables"] = "closed-form synthetic rows (bench.synth_rows), sigma 1/sqrt(K)"
            outs = [torch.empty((B, F * K), dtype=torch.float32, device=device) for _ in range(2)]
            fms = [torch.empty((B, 1), dtype=torch.float32, device=device) for _ in range(2)]
            inflight = {}

            def step(i):
                # software pipeline over steps: batch i+1's lookup is enqueued before batch i's result is taken (double-buffered plans),
                # so a step never waits for its own overflow verdict and neighbouring lookups overlap on the side streams
                cur = inflight.pop(i, None) or st.lookup_async(idsl[i % len(idsl)], want_fm=True, out=outs[i % 2], fm=fms[i % 2])
                inflight[i + 1] = st.lookup_async(idsl[(i + 1) % len(idsl)], want_fm=True, out=outs[(i + 1) % 2], fm=fms[(i + 1) % 2])
                cur.result()
            alg = B * (F * (8 + 2 * 4 * K) + 4)
            # what bounds the lookup-only line at N > 1: with uniform ids (P-1)/P of every batch's rows (4K bytes) and ids (8 bytes) cross
            # xGMI, 1/P of them to each peer over that pair's own link (the node is fully connected: 7 links x ~153.6 GB/s per GPU)
            roof = {"bound": "xgmi", "alg_bytes": alg, "link_bytes": B * F * (4 * K + 8) / world,
                    "kernel": "sharded lookup: bucket_cap_k -> all_to_all ids -> gather_slabs_k -> all_to_all rows -> gather_onehot_k<fm,out>"}
            cfg["parallelism"] = "tables row-sharded 'div' over %d GPUs, RCCL all_to_all x2 per micro-batch (2 per lookup), lookups pipelined, check lazy" % world
    elif wl == "sharded_1gpu":
        # the sharded code path on one GPU without collectives: what the exchange costs besides the network
        sigma = 1.0 / (K ** 0.5)
        loc = [torch.randn((V, K), generator=gen, device=device) * sigma for _ in range(F)]
        st = ShardedTables(loc, [V] * F)
        idsl = make_ids(torch, args, gen, device, V)
        step = lambda i: st.lookup(idsl[i % len(idsl)], want_fm=True)  # noqa: E731
        roof = {"bound": "hbm", "alg_bytes": B * (F * (8 + 2 * 4 * K) + 4), "kernel": "bucket + gather_packed + gather_onehot_k"}
        cfg.update({"fields": F, "vocab_per_field": V, "dim": K})
    elif wl == "sharded_deepfm_1gpu":
        # Sharded lookup -> DeepFM tower (FM term + 400-400-400 + head) on one GPU without collectives, two ways:
        #   DIR_BENCH_SHARD_CONSUME=1 (default): ShardedTables.lookup_consume -- the tower kernel reads the received rows through the inverse
        #     positions; rank-local passes: bucket + owner gather, then the tower's own lookups (no [B, F*K] concatenation exists);
        #   DIR_BENCH_SHARD_CONSUME=0: lookup(want_fm=True) (bucket + owner gather + finish pass) + the plain tower with the FM addend.
        from dir_amd.shard import rows_as_tables
        sigma = 1.0 / (K ** 0.5)
        loc = [torch.randn((V, K), generator=gen, device=device) * sigma for _ in range(F)]
        st = ShardedTables(loc, [V] * F)
        idsl = make_ids(torch, args, gen, device, V)
        Ws = [torch.randn((400, F * K), generator=gen, device=device) * 0.05, torch.randn((400, 400), generator=gen, device=device) * 0.05,
              torch.randn((400, 400), generator=gen, device=device) * 0.05]
        bs = [torch.zeros(400, device=device) for _ in Ws]
        hw, hb = torch.randn((400,), generator=gen, device=device) * 0.05, torch.zeros(1, device=device)
        logit = torch.empty((B, 1), dtype=torch.float32, device=device)
        consume = os.environ.get("DIR_BENCH_SHARD_CONSUME", "1") != "0"

        def consumer(s, e, rows, inv):
            ops.tower(None, Ws, bs, head=(hw, hb), gather=(rows_as_tables(rows, F), inv, None, True), out=logit[s:e], split="f16x2")

        if consume:
            step = lambda i: st.lookup_consume(idsl[i % len(idsl)], consumer)  # noqa: E731
        else:
            def step(i):
                emb, fm = st.lookup(idsl[i % len(idsl)], want_fm=True)
                ops.tower(emb, Ws, bs, head=(hw, hb), adds=(fm,), out=logit, split="f16x2")
        roof = {"bound": "hbm", "alg_bytes": B * (F * (8 + 2 * 4 * K) + 4), "kernel": "bucket + gather_slabs + " + ("tower_bf3_k<GATHER> over the received rows" if consume else "gather_onehot_k (finish) + tower_bf3_k")}
        cfg.update({"fields": F, "vocab_per_field": V, "dim": K, "finish_pass": not consume, "tower": "416-400-400-400-1 + FM"})
    elif wl == "deepfm_sparse_packed":
        # DeepFM's three sparse terms (concat, FM, linear) from packed 128-byte rows in ONE pass, vs gather_fm + linear
        sigma = 1.0 / (K ** 0.5)
        tables = [torch.randn((V, K), generator=gen, device=device) * sigma for _ in range(F)]
        lws = [torch.randn((V,), generator=gen, device=device) * 0.01 for _ in range(F)]
        pt = ops.PackedTables(tables, lws)
        del tables
        bias = torch.zeros(1, device=device)
        idsl = make_ids(torch, args, gen, device, V)
        out = torch.empty((B, F * K), dtype=torch.float32, device=device)
        fm = torch.empty((B, 1), dtype=torch.float32, device=device)
        lin = torch.empty((B, 1), dtype=torch.float32, device=device)
        step = lambda i: ops.gather_fm_linear(pt, idsl[i % len(idsl)], bias=bias, out=out, fm=fm, lin=lin)  # noqa: E731
        roof = {"bound": "hbm", "alg_bytes": B * (F * (8 + 2 * 4 * K + 4) + 8), "kernel": "gather_packed_rows_k (concat + FM + linear)"}
        cfg.update({"fields": F, "vocab_per_field": V, "dim": K, "row_bytes": pt.ld * 4})
    elif wl == "train_sparse":
        # the sparse side of one DeepFM training step on the config-2 shape:
        # gather + FM forward, FM backward (+ the DNN branch's gradient), fused sparse Adagrad on the 26 tables
        sigma = 1.0 / (K ** 0.5)
        tables = [torch.randn((V, K), generator=gen, device=device) * sigma for _ in range(F)]
        layout = args.train_layout if args.adagrad_method == "sorted" else "split_unfused"
        ts = ops.TableSet.train_rows(tables) if layout == "packed" else ops.TableSet(tables)
        del tables
        opt = ops.SparseAdagrad(ts, lr=0.01, method=args.adagrad_method)
        idsl = make_ids(torch, args, gen, device, V)
        out = torch.empty((B, F * K), dtype=torch.float32, device=device)
        fm = torch.empty((B, 1), dtype=torch.float32, device=device)
        fsum = torch.empty((B, K), dtype=torch.float32, device=device)
        gfm = torch.randn((B, 1), generator=gen, device=device) * 0.01
        gdnn = torch.randn((B, F * K), generator=gen, device=device) * 0.01
        demb = torch.empty_like(out)

        if layout == "split_unfused":
            def step(i):
                ids = idsl[i % len(idsl)]
                ops.gather_fm(ts, ids, out=out, fm=fm)
                ops.fm_logit_backward(out, gfm, F, K, add_in=gdnn, out=demb)
                opt.step(ids, demb)
        else:
            def step(i):
                ids = idsl[i % len(idsl)]
                ops.gather_fm(ts, ids, out=out, fm=fm, fsum=fsum)
                opt.step_fm(ids, gdnn, gfm, fsum)
        # forward 3 540 B + FM backward (emb, dnn grad read, demb written) + adagrad (ids, demb, w and accum read+write)
        # folded: forward + field sums written, then per entry ids, sort key/value, DNN gradient row, w and accum read + write
        alg = (B * ((F * (8 + 8 * K) + 4) + 3 * 4 * F * K + F * (8 + 4 + 4 * K + 4 * 4 * K)) if layout == "split_unfused" else
               B * ((F * (8 + 8 * K) + 4) + 4 * K + F * (8 + 4 + 4 * K + 4 * 4 * K)))
        roof = {"bound": "hbm", "alg_bytes": alg,
                "kernel": ("gather_onehot_k + fm_bwd_k + " if layout == "split_unfused" else "gather_packed_rows_k (+ field sums) + ") +
                          ("rss_keys_k + rss_hist_k + rs_pass_k<10, slot> (in-tree slot-major radix sort) + adagrad_tile_k" + ("" if layout == "split_unfused" else "<FM folded in>") +
                           " + adagrad_fix_k" if args.adagrad_method == "sorted" else "adagrad_link_k + adagrad_apply_k")}
        cfg.update({"fields": F, "vocab_per_field": V, "dim": K, "adagrad": args.adagrad_method, "ids": args.id_dist, "layout": layout})
    elif wl == "multihot_bag":
        # weighted variable-length bags (dataset/SequenceTensorFlowDataset/test4.py:50-59 style input): per
        # (sample, field) a bag of 0..8 ids with fp32 weights, combiner "mean", CSR sample-major
        sigma = 1.0 / (K ** 0.5)
        tables = [torch.randn((V, K), generator=gen, device=device) * sigma for _ in range(F)]
        ts = ops.TableSet(tables)
        lens = torch.randint(0, 9, (B * F,), generator=gen, device=device)
        offs = torch.cat([torch.zeros(1, dtype=torch.int64, device=device), lens.cumsum(0)])
        nnz = int(offs[-1].item())
        vals = [torch.randint(0, V, (nnz,), generator=gen, device=device) for _ in range(args.rotate)]
        wts = torch.rand((nnz,), generator=gen, device=device) + 0.25
        out = torch.empty((B, F * K), dtype=torch.float32, device=device)
        step = lambda i: ops.embedding_bag(ts, vals[i % len(vals)], offs, wts, combiner="mean", out=out)  # noqa: E731
        roof = {"bound": "hbm", "alg_bytes": nnz * (8 + 4 + 4 * K) + B * F * (8 + 4 * K), "kernel": "bag_csr_k (weighted mean)"}
        cfg.update({"fields": F, "vocab_per_field": V, "dim": K, "nnz": nnz, "bag_len": "U{0..8}", "combiner": "mean", "weights": True})
    elif wl == "deepfm_train":
        # one whole DeepFM training step with the reference's optimisers (deepFM.py:58,61): forward (gather+FM kernel,
        # linear term, 400-400-400 MLP on the dense kernels), BCE loss, backward (HIP FM backward, head + gated data gradients, sparse row gradients), fused sorted
        # sparse Adagrad on the 26 embedding tables and fused sparse FTRL on the 26 linear columns inside backward(), torch Adagrad on the MLP
        from dir_amd.deepfm import DeepFM
        from dir_amd import feature_column as fc
        from dir_amd.autograd import Ftrl
        cats = [fc.categorical_column_with_identity("C%d" % i, V) for i in range(F)]
        model = DeepFM(linear_feature_columns=cats, dnn_feature_columns=[fc.embedding_column(c, K) for c in cats],
                       dnn_hidden_units=[400, 400, 400], fm_embedding_size=K).to(device)
        model.fused_sparse_adagrad(lr=0.01, packed=(args.train_layout == "packed"))
        ftrl_rows = args.train_layout == "packed" and os.environ.get("DIR_BENCH_FTRL_ROWS", "1") != "0"
        model.fused_sparse_ftrl(lr=0.2, packed=ftrl_rows)       # packed linear training rows [w | n | z | -] beside the packed embedding rows
        lin = [model.linear_bias]                          # the weight columns are updated by the fused kernel inside backward()
        skip = {id(p) for p in model.linear_weights} | {id(p) for p in lin} | {id(p) for p in model.embedding_weights}
        opt_dense = ag_opt.Adagrad([p for p in model.parameters() if id(p) not in skip], lr=0.01, initial_accumulator_value=0.1)
        opt_lin = Ftrl(lin, lr=0.2)
        idsl = make_ids(torch, args, gen, device, V)
        featl = [{"C%d" % f: ids[:, f] for f in range(F)} for ids in idsl]
        labels = (torch.rand((B, 1), generator=gen, device=device) < 0.25).float()

        def step(i):
            opt_dense.zero_grad(set_to_none=True)
            opt_lin.zero_grad(set_to_none=True)
            loss = torch.nn.functional.binary_cross_entropy_with_logits(model(featl[i % len(featl)]), labels)
            loss.backward()
            opt_dense.step()
            opt_lin.step()
        roof = {"bound": "hbm", "alg_bytes": B * ((F * (8 + 8 * K) + 4) + 3 * 4 * F * K + F * (8 + 4 + 4 * K + 4 * 4 * K)),
                "kernel": "whole training step; bytes = the sparse side only (gather+FM, FM backward, sparse Adagrad)"}
        cfg.update({"fields": F, "vocab_per_field": V, "dim": K, "ids": args.id_dist, "mlp": [400, 400, 400], "layout": args.train_layout,
                    "linear_rows": "packed [w|n|z|-]" if ftrl_rows else "three arrays",
                    "optimizers": "sparse Adagrad + sparse FTRL (HIP, sorted, inside backward) + torch Adagrad (MLP)"})
    elif wl == "small_batch":
        # the reference's own batch size (256; 100 for evaluation: DeepCrossNetwork/train.py:16-17): nothing is bound but launch latency and the
        # longest dependent chain.  DeepFM (26 x 16, 400-400-400) and DCN (d = 429, 3 cross layers, deep 1024-1024) inference, one forward
        # eager and as a HIP-graph replay; the timed step is the DeepFM replay.  Since round 5 every hidden layer at this size runs
        # dir_dense_small_f32 (dense_routing.library_layers_per_step == {}).
        from dir_amd.deepfm import DeepFM
        from dir_amd.dcn import DeepCrossNetwork
        from dir_amd import feature_column as fc
        from dir_amd import dense as _dense_mod
        from dir_amd.serving import GraphedForward
        Bs = int(os.environ.get("DIR_BENCH_SMALL_BATCH", "256"))
        cats = [fc.categorical_column_with_identity("C%d" % i, V) for i in range(F)]
        model = DeepFM(linear_feature_columns=cats, dnn_feature_columns=[fc.embedding_column(c, K) for c in cats],
                       dnn_hidden_units=[400, 400, 400], fm_embedding_size=K).to(device).eval()
        nums = [fc.numeric_column("I%d" % i) for i in range(13)]
        dcn = DeepCrossNetwork(columns=[fc.embedding_column(c, K) for c in cats] + nums, cross_layer_num=3, dnn_hidden_units=[1024, 1024]).to(device).eval()
        ids = torch.randint(0, V, (Bs, F), generator=gen, device=device)
        dfeats = {"C%d" % i: ids[:, i].contiguous() for i in range(F)}
        dfeats.update({"I%d" % i: torch.rand((Bs, 1), generator=gen, device=device) for i in range(13)})
        fwd = lambda x: model.forward_ids(x, x)  # noqa: E731
        fwd_dcn = lambda x: dcn(dfeats)          # noqa: E731

        def latency(fn, n=200):
            with torch.no_grad():
                for _ in range(20):
                    fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e6
        lat = {}
        for name, f in (("deepfm", fwd), ("dcn", fwd_dcn)):
            _dense_mod.reset_routing()
            g_ = GraphedForward(f, ids)
            with torch.no_grad():
                ref = f(ids)
            assert torch.equal(g_(ids), ref)
            lat[name] = {"eager_us": round(latency(lambda: f(ids)), 2), "graph_replay_us": round(latency(lambda: g_.graph.replay()), 2),
                         "library_layers": dict(_dense_mod.ROUTING["library"]), "hip_layers": sorted(_dense_mod.ROUTING["hip"])}
            gf_ = GraphedForward(f, ids, frozen_weights=True)      # the serving form: weight images of the warm-up calls, no pack launch per replay
            assert torch.equal(gf_(ids), ref)
            lat[name]["graph_replay_frozen_weights_us"] = round(latency(lambda: gf_.graph.replay()), 2)
            if name == "deepfm":
                graphed = g_
        if Bs > ops.DENSE_SMALL_ROWS:
            # mid-size batches (round 6): the same forwards with round 5's routing -- these layers on the library's GEMMs -- beside ours
            mid_rows, ops.DENSE_MID_ROWS = ops.DENSE_MID_ROWS, 0
            for name, f in (("deepfm", fwd), ("dcn", fwd_dcn)):
                _dense_mod.reset_routing()
                gl_ = GraphedForward(f, ids)
                lat[name]["library_routing"] = {"eager_us": round(latency(lambda: f(ids)), 2), "graph_replay_us": round(latency(lambda: gl_.graph.replay()), 2),
                                                "library_layers": dict(_dense_mod.ROUTING["library"])}
            ops.DENSE_MID_ROWS = mid_rows
        _dense_mod.reset_routing()
        step = lambda i: graphed(ids)  # noqa: E731
        units = Bs
        roof = {"bound": "hbm", "alg_bytes": Bs * (F * (8 + 2 * 4 * K) + 4), "kernel": "DeepFM forward, batch %d, hipGraph replay" % Bs}
        cfg.update({"batch": Bs, "fields": F, "latency_per_forward": lat, "eager_us_per_forward": lat["deepfm"]["eager_us"]})
    elif wl == "transform":
        # raw Criteo-style features -> gather-ready ids on the device
        # (26 integer categorical keys hashed per field, 13 dense values bucketised into 10 buckets)
        keys = torch.randint(-2**40, 2**40, (B, F), generator=gen, device=device)
        nbf = torch.full((F,), V, dtype=torch.int64, device=device)
        dense = torch.rand((B, 13), generator=gen, device=device)
        bd = torch.linspace(0.1, 0.9, 9, device=device)

        def step(i):
            ops.hash_bucket_ints_fields(keys, nbf)
            ops.bucketize(dense, bd)
        roof = {"bound": "hbm", "alg_bytes": B * F * 16 + B * 13 * 12, "kernel": "hash_bucket_i64_k + bucketize_k"}
        cfg.update({"fields": F, "dense": 13})
    elif wl == "dcn_full":
        # BASELINE configs[2]: DCN, 3 cross layers on the 26-field schema (+13 dense -> d = 429), deep 1024-1024
        from dir_amd.dcn import DeepCrossNetwork
        from dir_amd import feature_column as fc
        cols = [fc.embedding_column(fc.categorical_column_with_identity("C%02d" % i, V), K) for i in range(F)]
        cols += [fc.numeric_column("I%02d" % i) for i in range(13)]
        model = DeepCrossNetwork(columns=cols, cross_layer_num=3, dnn_hidden_units=[1024, 1024], batch_norm=True).to(device)
        idsl = make_ids(torch, args, gen, device, V)
        dense = torch.rand((B, 13), generator=gen, device=device)
        feats = []
        for ids in idsl:
            f = {"C%02d" % i: ids[:, i].contiguous() for i in range(F)}
            f.update({"I%02d" % i: dense[:, i].contiguous() for i in range(13)})
            feats.append(f)
        with torch.no_grad():
            model(feats[0])

        def step(i):
            with torch.no_grad():
                model(feats[i % len(feats)])
        roof = {"bound": "hbm", "alg_bytes": B * (F * (8 + 2 * 4 * K)), "kernel": "DCN forward (embedding-bag bytes only)"}
        cfg.update({"fields": F, "dense": 13, "d": model.column_num, "cross_layers": 3, "deep": [1024, 1024]})
    elif wl == "dcn_train":
        # whole DCN training step with the reference's train_op (Adam eps 1e-4, cosine decay, per-tensor clip_by_norm 100:
        # DeepCrossNetwork.py:264-290, DeepCrossNetwork/train.py:111-125) on the configs[2] model
        from dir_amd.dcn import DeepCrossNetwork
        from dir_amd import feature_column as fc
        cols = [fc.embedding_column(fc.categorical_column_with_identity("C%02d" % i, V), K) for i in range(F)]
        cols += [fc.numeric_column("I%02d" % i) for i in range(13)]
        model = DeepCrossNetwork(columns=cols, cross_layer_num=3, dnn_hidden_units=[1024, 1024], batch_norm=True, optimizer="Adam",
                                 optimizer_spec={"epsilon": 1e-4},
                                 learning_rate_spec={"learning_rate": 0.001, "decay_method": "cosine_decay", "decay_steps": 3000,
                                                     "alpha": 0.5}).to(device)
        train_op = model.train_step()
        idsl = make_ids(torch, args, gen, device, V)
        dense = torch.rand((B, 13), generator=gen, device=device)
        feats = []
        for ids in idsl:
            f = {"C%02d" % i: ids[:, i].contiguous() for i in range(F)}
            f.update({"I%02d" % i: dense[:, i].contiguous() for i in range(13)})
            feats.append(f)
        labels = (torch.rand((B, 1), generator=gen, device=device) < 0.25).float()

        def step(i):
            train_op(torch.nn.functional.binary_cross_entropy_with_logits(model(feats[i % len(feats)]), labels))
        roof = {"bound": "hbm", "alg_bytes": B * (F * (8 + 2 * 4 * K)), "kernel": "whole DCN training step (embedding-bag bytes only)"}
        cfg.update({"fields": F, "dense": 13, "d": model.column_num, "cross_layers": 3, "deep": [1024, 1024],
                    "train_op": "Adam(eps 1e-4) dense on all variables incl. tables, cosine decay, clip_by_norm 100 per tensor"})
    elif wl in ("esmm_full", "esmm_train"):
        # ESMM (models/ESMM/ESMM.py:62-92): two towers, each with its OWN embedding variables over the same 26 columns, 360-200-80
        # hidden units; forward, or a training step (CTR + CTCVR losses, ESMM.py:150-175) with torch Adagrad (sparse on the tables)
        from dir_amd.esmm import ESMM
        from dir_amd import feature_column as fc
        cols = [fc.embedding_column(fc.categorical_column_with_identity("C%02d" % i, V), K) for i in range(F)]
        model = ESMM(columns=cols, dnn_hidden_units=[360, 200, 80]).to(device)
        idsl = make_ids(torch, args, gen, device, V)
        feats = [{"C%02d" % i: ids[:, i].contiguous() for i in range(F)} for ids in idsl]
        if wl == "esmm_full":
            def step(i):
                with torch.no_grad():
                    model(feats[i % len(feats)])
        else:
            labels = {"click_label": (torch.rand((B, 1), generator=gen, device=device) < 0.25).float(),
                      "convert_label": (torch.rand((B, 1), generator=gen, device=device) < 0.05).float()}
            dense_p = [p for n, p in model.named_parameters() if "embedding_weights" not in n]
            opt_d = ag_opt.Adagrad(dense_p, lr=0.05, initial_accumulator_value=0.1, eps=0.0)
            sparse_opts = model.fused_sparse_adagrad(0.05)        # both towers' tables; the two sorted updates share one sort per step
            cfg["sparse_optimizers"] = len(sparse_opts)

            def step(i):
                f = feats[i % len(feats)]
                opt_d.zero_grad(set_to_none=True)
                loss, _ = model.get_loss(f, labels, model(f))
                loss.backward()                                   # the fused sparse Adagrad updates the 52 tables inside backward
                opt_d.step()
        roof = {"bound": "hbm", "alg_bytes": 2 * B * (F * (8 + 2 * 4 * K)), "kernel": "ESMM %s (two embedding-bag passes' bytes only)" % wl}
        cfg.update({"fields": F, "towers": 2, "hidden": [360, 200, 80]})
    elif wl in ("xdeepfm_full", "xdeepfm_train"):
        # BASELINE configs[4] on one GPU: xDeepFM (CIN 128-128-128 + DNN 400-400 + linear) forward, or a whole training step
        # (CIN backward on MFMA, fused sparse Adagrad / FTRL on the tables and linear columns inside backward, torch Adagrad on everything dense)
        from dir_amd.xdeepfm import XDeepFM
        from dir_amd import feature_column as fc
        cats = [fc.categorical_column_with_identity("C%d" % i, V) for i in range(F)]
        model = XDeepFM(linear_feature_columns=cats, dnn_feature_columns=[fc.embedding_column(c, K) for c in cats],
                        cin_layer_sizes=(128, 128, 128), dnn_hidden_units=(400, 400)).to(device)
        idsl = make_ids(torch, args, gen, device, V)
        featl = [{"C%d" % f: ids[:, f] for f in range(F)} for ids in idsl]
        if wl == "xdeepfm_full":
            def step(i):
                with torch.no_grad():
                    model(featl[i % len(featl)])
            alg, pipe, modes = cin_flops(ops, B, F, K, (128, 128, 128))
            roof = {"bound": "mfma", "alg_flops": alg, "pipe_flops": pipe, "kernel": "xDeepFM forward (CIN flops only): cin_bf3_k x3",
                    "modes": modes, "dtype": "f32 (CIN: f32 via %s, f32 accumulate)" % _cin_fwd_label(ops)}
        else:
            # the DeepFM recipe (deepFM.py:58,61): fused sorted sparse Adagrad on the embedding tables and FTRL on the linear columns
            # inside backward() (one sort for both), torch Adagrad on everything dense
            model.fused_sparse_adagrad(lr=0.01)
            model.fused_sparse_ftrl(lr=0.2)
            sparse_ids = {id(p) for p in model.embedding_weights} | {id(p) for p in model.linear_weights}
            opt_d = ag_opt.Adagrad([p for p in model.parameters() if id(p) not in sparse_ids], lr=0.01, initial_accumulator_value=0.1)
            labels = (torch.rand((B, 1), generator=gen, device=device) < 0.25).float()

            def step(i):
                opt_d.zero_grad(set_to_none=True)
                torch.nn.functional.binary_cross_entropy_with_logits(model(featl[i % len(featl)]), labels).backward()
                opt_d.step()
            alg, pipe, modes = cin_flops(ops, B, F, K, (128, 128, 128), forward=True, backward=True)
            roof = {"bound": "mfma", "alg_flops": alg, "pipe_flops": pipe, "modes": modes,
                    "kernel": "xDeepFM training step (CIN forward + backward flops only): cin_bf3_k, cin_bf3_k<DOT>, cin_dw_bf3_k (layer 1: cin_dw_k)",
                    "dtype": "f32 (CIN forward: f32 via %s; backward: %s; f32 accumulate)" % (
                        _cin_fwd_label(ops), "scaled fp16x2 / bf16x3 split" if "f16x2" in [v for k_, v in modes.items() if k_.startswith("d")] else "bf16x3 split")}
        cfg.update({"fields": F, "vocab_per_field": V, "dim": K, "cin": [128, 128, 128], "dnn": [400, 400]})
    elif wl == "dcn_cross":
        d, L = args.cross_d, 3
        dp = ops.pad4(d)
        # the DCN input layer's layout: row stride pad4(d), zero pad columns (d = 429 -> 432); algorithmic bytes count the d real columns
        x0 = torch.zeros((B, dp), device=device)
        x0[:, :d] = torch.randn((B, d), generator=gen, device=device) * 0.25
        w = torch.zeros((L, dp), device=device)
        w[:, :d] = (torch.randn((L, d), generator=gen, device=device) * 0.1).clamp_(-0.2, 0.2)
        bb = torch.zeros((L, dp), device=device)
        bb[:, :d] = (torch.randn((L, d), generator=gen, device=device) * 0.1).clamp_(-0.2, 0.2)
        out = torch.empty_like(x0)
        step = lambda i: ops.cross_network(x0, w, bb, out=out)  # noqa: E731
        roof = {"bound": "hbm", "alg_bytes": B * 2 * 4 * d + 2 * L * d * 4, "kernel": "cross_k"}
        cfg.update({"d": d, "row_stride": dp, "layers": L})
    elif wl == "dcn_cross_backward":
        d, L = args.cross_d, 3
        x0 = torch.randn((B, d), generator=gen, device=device) * 0.25
        w = (torch.randn((L, d), generator=gen, device=device) * 0.1).clamp_(-0.2, 0.2)
        bb = (torch.randn((L, d), generator=gen, device=device) * 0.1).clamp_(-0.2, 0.2)
        gout = torch.randn((B, d), generator=gen, device=device) * 0.01
        step = lambda i: ops.cross_network_backward(x0, w, bb, gout)  # noqa: E731
        # x0 and gout read, gx0 written (the forward is recomputed per row on chip)
        roof = {"bound": "hbm", "alg_bytes": B * 3 * 4 * d, "kernel": "cross_bwd_k + reduce_partials_k"}
        cfg.update({"d": d, "layers": L})
    elif wl == "din":
        T, Kd, Vd, H1, H2 = 50, 64, 10000000, 80, 40
        table = torch.randn((Vd, Kd), generator=gen, device=device) * 0.125
        hist = torch.randint(0, Vd, (B, T), generator=gen, device=device)
        hl = torch.randint(1, T + 1, (B,), generator=gen, device=device, dtype=torch.int32)
        cand = torch.randint(0, Vd, (B,), generator=gen, device=device)
        W1 = torch.randn((4 * Kd, H1), generator=gen, device=device) * 0.05
        b1 = torch.zeros(H1, device=device)
        W2 = torch.randn((H1, H2), generator=gen, device=device) * 0.1
        b2 = torch.zeros(H2, device=device)
        W3 = torch.randn((H2,), generator=gen, device=device) * 0.1
        b3 = torch.zeros(1, device=device)
        step = lambda i: ops.din_attention_pool(table, hist, hl, cand, W1, b1, W2, b2, W3, b3, normalize=True)  # noqa: E731
        survey = B * T * 2 * (4 * Kd * H1 + H1 * H2 + H2)   # every one of the T positions, 4K-wide layer 1 (not what is priced)
        # what the kernel executes, and what is priced: valid history rows only, layer 1 regrouped to a 2K reduction -> 220
        # v_mfma_f32_16x16x4_f32 equivalents (1024 multiply-adds each) per 16 rows.  Round 6: the packed kernel (din_pack_k) lays the rows of
        # consecutive samples end to end, so the rows are counted as they are (rows / 16 tiles; the wave-per-sample kernel -- DIR_DIN_PACKED=0,
        # or an arithmetic other than fp16 x 2 -- pads every sample to whole tiles and is priced on those)
        din_arith = os.environ.get("DIR_DIN_ARITH") if os.environ.get("DIR_DIN_ARITH") in ("f32", "bf16x3") else "f16x2"
        packed = ops.DIN_PACKED and din_arith == "f16x2" and not os.environ.get("DIR_DIN_ARITH")
        rows = hl.clamp(max=T).double().sum().item()
        rt = rows / 16.0 if packed else ((hl.clamp(max=T) + 15) // 16).double().sum().item()
        executed = 2.0 * 1024 * rt * (4 * 32 + 4 * 8 + 3 * 20)
        kname = "din_pack_k" if packed else "din_wave_k"
        # SURVEY 8d's bytes: (T + 1)(4K + 8) + 4 + 4K per sample at full T; length-aware: the rows and ids the kernel reads
        bytes_full = B * ((T + 1) * (4 * Kd + 8) + 4 + 4 * Kd)
        bytes_len = rows * (4 * Kd + 8) + B * (4 * Kd + 8 + 4 + 4 * Kd)
        roof = {"bound": "mfma", "alg_flops": executed, "pipe_flops": executed * PIPE_COST[din_arith],
                "kernel": "%s (%s)" % (kname, "fp32 MFMA 16x16x4" if din_arith == "f32" else din_arith + " on MFMA 16x16x32"),
                "modes": {kname: din_arith}, "survey_8d_flops": survey, "hbm_bytes_full_T": bytes_full, "hbm_bytes_length_aware": bytes_len,
                "dtype": "f32" if din_arith == "f32" else "f32 via %s split, f32 accumulate" % din_arith,
                "note": "flops = the MFMAs of the valid history rows (masked positions are skipped, layer 1 is regrouped to a 2K reduction); "
                        "SURVEY 8d's all-T, 4K-wide count is reported as survey_8d_flops and not priced; hbm_frac = SURVEY 8d's 13 724 B per "
                        "sample (full T) / ms_per_step / 8 TB/s, hbm_frac_length_aware on the rows actually read"}
        cfg.update({"T": T, "dim": Kd, "vocab": Vd, "mlp": [4 * Kd, H1, H2, 1]})
    elif wl == "din_full":
        # BASELINE configs[3] as a whole model (dir_amd.din.DIN; paper-derived, README.md:27): behaviour sequence T 50 + candidate through the
        # local activation unit on the 10 M x 64 goods table, concat([interest vector, candidate embedding]) -> 200-80 MLP -> logit
        from dir_amd.din import DIN
        T, Kd, Vd = 50, 64, 10000000
        att_act = os.environ.get("DIR_BENCH_DIN_ACT", "sigmoid")
        dnn_act = os.environ.get("DIR_BENCH_DIN_DNN_ACT", "dice")
        model = DIN(item_vocab_size=Vd, embedding_dim=Kd, attention_hidden_units=(80, 40), attention_activation=att_act, attention_normalize=True,
                    dnn_hidden_units=(200, 80), dnn_activation_fn=dnn_act).to(device).eval()
        hists = [torch.randint(0, Vd, (B, T), generator=gen, device=device) for _ in range(2)]
        hl = torch.randint(1, T + 1, (B,), generator=gen, device=device, dtype=torch.int32)
        cand = torch.randint(0, Vd, (B,), generator=gen, device=device)
        featl = [{"hist": h, "hist_len": hl, "cand": cand} for h in hists]

        def step(i):
            with torch.no_grad():
                model(featl[i % 2])
        rt = ((hl.clamp(max=T) + 15) // 16).double().sum().item()
        unit = 2.0 * 1024 * rt * (4 * 32 + 4 * 8 + 3 * 20)                      # as the `din` workload: the MFMAs the unit issues over valid rows
        mlp = 2.0 * B * (2 * Kd * 200 + 200 * 80)
        din_arith = os.environ.get("DIR_DIN_ARITH") if os.environ.get("DIR_DIN_ARITH") in ("f32", "bf16x3") else "f16x2"
        mlp_modes = {"mlp_%d" % (i + 1): ("bf16x3" if ops.dense_auto_arith(B, a, b_) == "bf16x3" else "f32") for i, (a, b_) in enumerate(((2 * Kd, 200), (200, 80)))}
        pipe = unit * PIPE_COST[din_arith] + 2.0 * B * (2 * Kd * 200 * PIPE_COST[mlp_modes["mlp_1"]] + 200 * 80 * PIPE_COST[mlp_modes["mlp_2"]])
        roof = {"bound": "mfma", "alg_flops": unit + mlp, "pipe_flops": pipe, "modes": dict(mlp_modes, din_wave_k=din_arith),
                "kernel": "DIN forward: din_wave_k (%s unit, %s) + candidate lookup + dense 128-200-80 (%s) + units-1 head" % (att_act, din_arith, dnn_act),
                "dtype": "f32" if din_arith == "f32" else "f32 via %s split, f32 accumulate" % din_arith,
                "note": "flops = the unit's MFMAs over valid history rows + the two hidden layers; the logit layer and the elementwise activations are not priced"}
        cfg.update({"T": T, "dim": Kd, "vocab": Vd, "attention": [4 * Kd, 80, 40, 1], "attention_activation": att_act, "dnn": [2 * Kd, 200, 80, 1],
                    "dnn_activation": dnn_act})
    elif wl == "mlp_dense":
        # the three hidden layers of DeepFM's DNN tower (416 -> 400 -> 400 -> 400, ReLU): dir_dense_f32, or torch (rocBLAS GEMM +
        # ReLU pass) with DIR_BENCH_DENSE=torch
        dims = [F * K, 400, 400, 400]
        x = torch.randn((B, dims[0]), generator=gen, device=device) * 0.25
        Wl = [torch.randn((dims[i + 1], dims[i]), generator=gen, device=device) / dims[i] ** 0.5 for i in range(3)]
        bl = [torch.randn((dims[i + 1],), generator=gen, device=device) * 0.1 for i in range(3)]
        ys = [torch.empty((B, dims[i + 1]), dtype=torch.float32, device=device) for i in range(3)]
        use_torch = os.environ.get("DIR_BENCH_DENSE") == "torch"
        use_tower = os.environ.get("DIR_BENCH_DENSE", "tower") == "tower" and ops.tower_covers(x, Wl)   # "layers": one dense_bf3_k per layer
        if not use_torch and not use_tower:
            from dir_amd.dense import pack_weight
            Wl = [pack_weight(w) for w in Wl]      # row stride a multiple of 64 floats (dense.py)

        def step(i):
            if use_tower:
                return ops.tower(x, Wl, bl, relu=True, out=ys[2])
            h = x
            for l in range(3):
                h = torch.relu(torch.addmm(bl[l], h, Wl[l].t())) if use_torch else ops.dense(h, Wl[l], bl[l], relu=True, out=ys[l])
        alg = pipe = 0.0
        modes = {}
        for i in range(3):
            f = 2.0 * B * dims[i] * dims[i + 1]
            a = (getattr(ops, "TOWER_SPLIT", "bf16x3") if use_tower else "f32" if use_torch or ops.DENSE_ARITH == "f32" else ops.DENSE_ARITH if ops.DENSE_ARITH != "auto"
                 else ops.dense_auto_arith(B, dims[i], dims[i + 1]))
            alg, pipe = alg + f, pipe + f * PIPE_COST[a]
            modes["layer%d" % (i + 1)] = a
        roof = {"bound": "mfma", "alg_flops": alg, "pipe_flops": pipe, "modes": modes,
                "kernel": "rocBLAS GEMM + relu x3" if use_torch else "tower_bf3_k (three layers, one launch)" if use_tower else
                          "dense_bf3_k<13> x3" if "bf16x3" in modes.values() else "dense_k<80, relu> x3",
                "dtype": ("f32 via fp16x2 split, f32 accumulate" if "f16x2" in modes.values() else
                          "f32 via bf16x3 split, f32 accumulate" if "bf16x3" in modes.values() else "f32")}
        cfg.update({"layers": dims})
    elif wl == "din_train":
        # forward (fused kernel) + backward (autograd.DinAttentionPool: fused HIP backward, sparse table gradient) of the DIN unit
        from dir_amd import autograd as ag
        T, Kd, Vd, H1, H2 = 50, 64, 10000000, 80, 40
        table = (torch.randn((Vd, Kd), generator=gen, device=device) * 0.125).requires_grad_(True)
        hist = torch.randint(0, Vd, (B, T), generator=gen, device=device)
        hl = torch.randint(1, T + 1, (B,), generator=gen, device=device, dtype=torch.int32)
        cand = torch.randint(0, Vd, (B,), generator=gen, device=device)
        ws = [(torch.randn((4 * Kd, H1), generator=gen, device=device) * 0.05).requires_grad_(True), torch.zeros(H1, device=device, requires_grad=True),
              (torch.randn((H1, H2), generator=gen, device=device) * 0.1).requires_grad_(True), torch.zeros(H2, device=device, requires_grad=True),
              (torch.randn((H2,), generator=gen, device=device) * 0.1).requires_grad_(True), torch.zeros(1, device=device, requires_grad=True)]
        gout = torch.randn((B, Kd), generator=gen, device=device) * 0.01

        def step(i):
            table.grad = None
            for w in ws:
                w.grad = None
            ag.din_attention_pool(table, hist, hl, cand, *ws, normalize=True).backward(gout)
        att_act = os.environ.get("DIR_BENCH_DIN_ACT", "sigmoid")
        if att_act in ("prelu", "dice"):
            # the paper's own unit activations in TRAIN mode (Dice: mini-batch statistics): the row-list path on HIP kernels (din.DINAttentionPool._rows_train,
            # csrc/din_rows_train.hip; DIR_DIN_ROWS_TRAIN=0: round 4's torch formulation)
            from dir_amd.din import DINAttentionPool
            unit = DINAttentionPool(Vd, Kd, (H1, H2), normalize=True, activation=att_act).to(device).train()
            del table

            def step(i):           # noqa: F811
                for p_ in unit.parameters():
                    p_.grad = None
                unit(hist, hl, cand).backward(gout)
            cfg["activation"] = att_act
        survey = 3 * B * T * 2 * (4 * Kd * H1 + H1 * H2 + H2)
        rt = ((hl.clamp(max=T) + 15) // 16).double().sum().item()
        executed = 2.0 * 1024 * rt * (220 + 660)     # 16x16x4-MFMA equivalents per 16-row tile: forward 220; backward 220 recompute + 440
        din_arith = os.environ.get("DIR_DIN_ARITH") if os.environ.get("DIR_DIN_ARITH") in ("f32", "bf16x3") else "f16x2"      # the forward's arithmetic; the backward kernels are fp32 MFMA
        pipe = 2.0 * 1024 * rt * (220 * PIPE_COST[din_arith] + 660 * PIPE_COST["f32"])
        roof = {"bound": "mfma", "alg_flops": executed, "pipe_flops": pipe, "modes": {"din_wave_k": din_arith, "din_rows_k": "f32", "din_wgrad_k": "f32"},
                "kernel": "din_wave_k (%s) + din_rows_k + din_wgrad_k (fp32 MFMA 16x16x4)" % din_arith, "survey_8d_flops": survey,
                "note": "flops = the MFMAs the three kernels issue over valid history rows; 3x SURVEY 8d's forward count is survey_8d_flops, not priced"}
        cfg.update({"T": T, "dim": Kd, "vocab": Vd, "mlp": [4 * Kd, H1, H2, 1]})
    elif wl == "cin":
        m, D, Hs = F, K, (128, 128, 128)
        x0 = torch.randn((B, m, D), generator=gen, device=device) * 0.25
        Ws, hp = [], m
        for h in Hs:
            Ws.append(torch.randn((h, hp * m), generator=gen, device=device) * (1.0 / (hp * m) ** 0.5))
            hp = h
        pooled = torch.empty((B, sum(Hs)), dtype=torch.float32, device=device)

        def step(i):
            xk, off = x0, 0
            for k, (W, h) in enumerate(zip(Ws, Hs)):   # as XDeepFM.cin: the last layer's map feeds nothing
                xk, _ = ops.cin_layer(x0, xk, W, pooled=pooled[:, off:off + h], want_xout=k + 1 < len(Hs), arith=args.cin_arith)
                off += h
        arith = args.cin_arith or ops.CIN_ARITH
        alg, pipe, modes = cin_flops(ops, B, m, D, Hs, arith=arith)
        bf3 = any(str(v).startswith(("bf16x3", "f16x2", "pooled")) for v in modes.values())
        f16 = any(str(v).startswith("f16x2") for v in modes.values())
        # priced on the pipe it runs on: six bf16 (three fp16) piece products per fp32 product (csrc/cin_bf3.hip) against the dense bf16 peak;
        # the fp32-equivalent rate (the algorithm's flops / time) is reported beside it
        roof = {"bound": "mfma", "alg_flops": alg, "pipe_flops": pipe, "modes": modes, "kernel": ("cin_bf3_k<PAIRS> (layer 1) + cin_bf3_k (layer 2) + the last layer in its pooled form (cin_pooled_k: fused; training: cin_pool_z_k + dense_bf3_k)"
                           if any(str(v).startswith("pooled") for v in modes.values()) else "cin_bf3_k x3") if bf3 else "cin_k x3",
                "dtype": ("f32 via fp16x2 split (layers 1-2) / bf16x3 split (pooled last layer), f32 accumulate" if f16 else
                          "f32 via bf16x3 split, f32 accumulate") if bf3 else "f32"}
        arith = ("f16x2" if f16 else "bf16x3") if bf3 else "f32"
        cfg.update({"m": m, "D": D, "layers": list(Hs), "outputs": "pooled [B,384]; xout of layers 1-2", "arith": arith})

    elif wl == "cin_backward":
        # backward of the 3-layer CIN stack given dL/dpooled (the forward's saved activations are inputs):
        # per layer dW (cin_dw_k) + dxk and dx0 in one pass (cin_dx_k), both fp32 MFMA
        m, D, Hs = F, K, (128, 128, 128)
        x0 = torch.randn((B, m, D), generator=gen, device=device) * 0.25
        Ws, hp = [], m
        for h in Hs:
            Ws.append(torch.randn((h, hp * m), generator=gen, device=device) * (1.0 / (hp * m) ** 0.5))
            hp = h
        xks, xk = [x0], x0
        for k, W in enumerate(Ws[:-1]):
            xk, _ = ops.cin_layer(x0, xk, W)
            xks.append(xk)
        gp = torch.randn((B, sum(Hs)), generator=gen, device=device) * 0.1
        zl = []
        ops.cin_layer(x0, xks[-1], Ws[-1], want_xout=False, z_out=zl)       # the top layer's Z = sum_d xk x0: one more saved activation
        z_top = zl[0] if zl else None

        def step(i):
            if os.environ.get("DIR_CIN_STACK_NODE", "1") != "0":
                return ops.cin_stack_backward(x0, xks, Ws, gp, z_top=z_top)       # what autograd.CinStack.backward runs
            gx = None            # per-layer form: gradient flowing into xout of the layer being processed
            off = sum(Hs)
            for k in range(len(Hs) - 1, -1, -1):
                h = Hs[k]
                off -= h
                g_p = gp[:, off:off + h].reshape(B, h, 1)
                G = g_p.expand(B, h, D).contiguous() if gx is None else gx.add_(g_p)
                dx0, gx, dW = ops.cin_layer_backward(x0, xks[k], Ws[k], G)
        # two GEMMs of the forward's size per layer: dW (reduction over rows) and T = G x W (both data gradients)
        alg, pipe, modes = cin_flops(ops, B, m, D, Hs, forward=False, backward=True)
        roof = {"bound": "mfma", "alg_flops": alg, "pipe_flops": pipe, "modes": modes,
                "kernel": "cin_bf3_k<DOT> (data gradients) + cin_dw_bf3_k (weight gradient; layer 1: cin_dw_sym_bf3_k over the unordered field pairs; the top layer in its pooled form: dense kernels + cin_pool_dx_k)",
                "dtype": ("f32 via fp16x2 split with the gradient operand scaled by powers of two (layer-2 kernels, layer-1 contraction) / bf16x3 split "
                          "(layer-1 dW, pooled top layer), f32 accumulate") if "f16x2" in modes.values() else "f32 via bf16x3 split, f32 accumulate"}
        cfg.update({"m": m, "D": D, "layers": list(Hs)})

    # ---- warmup, then EXACTLY --steps timed steps bracketed by barrier + synchronize ------------------
    if args.graph:
        if world > 1:
            sys.stderr.write("bench.py: --graph is a single-GPU option\n")
            return 2
        # the standard capture recipe: warm up on a side stream (every cache, workspace and pointer table exists before the capture), then
        # one graph per rotated id batch (the ids are part of the captured launches), replayed round robin
        eager_step = step
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(max(3, args.rotate)):
                eager_step(i)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graphs = []
        for i in range(max(1, args.rotate)):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                eager_step(i)
            graphs.append(g)
        step = lambda i: graphs[i % len(graphs)].replay()  # noqa: E731
        cfg["launch"] = "hipGraph replay (%d graphs, one per rotated id batch)" % len(graphs)
    for i in range(args.warmup):
        step(i)
    if args.torch_profile:
        torch_profile(torch, step, args.torch_profile)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                      # HIP events on the stream the kernels are launched on
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    ev1.record()
    issue = time.perf_counter() - t0           # host time to ISSUE the steps: close to the step time = the step is launch-bound on this host
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    cfg["host_issue_ms_per_step"] = round(issue / max(1, args.steps) * 1e3, 4)
    dev_ms = ev0.elapsed_time(ev1)
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    # ---- secondary leg of the default run: BASELINE config 5 (xDeepFM CIN 3 x 128 over a 10^8-row table, row-sharded when N > 1).
    # Same contract (barrier-bracketed, max over ranks, whole-job samples/s); never allowed to cost the primary line: a watchdog on
    # every rank gives up after DIR_BENCH_SECONDARY_TIMEOUT seconds and rank 0 then prints the primary result alone.
    secondary = None
    if wl == "deepfm_gather_fm" and args.id_dist == "uniform" and os.environ.get("DIR_BENCH_NO_SECONDARY") != "1":
        import threading
        done = threading.Event()
        primary = {"el": el, "dev_ms": dev_ms}

        def bail():
            if done.is_set():
                return
            if rank == 0:
                sys.stderr.write("bench.py: the config-5 secondary leg timed out; printing the primary line only\n")
                line = primary_line(args, wl, cfg, roof, units, world, primary["el"], primary["dev_ms"], None, None)
                line["secondary_cfg5_xdeepfm_cin"] = {"error": "timeout"}      # the hang is recorded in the line itself
                print(json.dumps(line), flush=True)
            # the primary measurement above is complete and is printed, but a process that touched the GPU and hung in a kernel or a
            # collective must not report success: exit code 3 on every rank (DIR_BENCH_SECONDARY_LENIENT=1: 0, for debugging only)
            os._exit(0 if os.environ.get("DIR_BENCH_SECONDARY_LENIENT") == "1" else 3)
        timer = threading.Timer(float(os.environ.get("DIR_BENCH_SECONDARY_TIMEOUT", "240")), bail)
        timer.daemon = True
        timer.start()
        try:
            secondary = cfg5_leg(torch, dist, ops, ShardedTables, div_range, args, world, rank, device, gen, backend)
        except Exception as exc:        # symmetric failures only (same code on every rank); the watchdog covers the rest
            secondary = {"error": repr(exc)[:300]}
        done.set()
        timer.cancel()

    if rank == 0:
        ms_per_step = el * 1e3 / args.steps
        value = units * world * args.steps / el
        res = {"metric": "CTR samples/sec (embedding gather + FM 2nd-order, 26-field batch 65536)" if wl == "deepfm_gather_fm"
               else "samples/sec (%s)" % wl,
               "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic", "config": cfg}
        # what actually ran: the process group's size as torch.distributed reports it (not the --gpus argument), and how the ranks were started
        res["world_size"] = dist.get_world_size() if dist.is_initialized() else 1
        res["launcher"] = os.environ.get("DIR_BENCH_LAUNCHED_BY") or ("torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else
                                                                      "external" if "WORLD_SIZE" in os.environ else "single process")
        try:        # whole-model workloads: which hidden layers ran on the HIP dense kernels and which fell to the library (dense.MIN_ROWS)
            from dir_amd import dense as _dense
            if _dense.ROUTING["hip"] or _dense.ROUTING["library"]:
                calls = max(1, args.warmup + args.steps)
                res["dense_routing"] = {"min_rows_for_hip_dense": _dense.MIN_ROWS,
                                        "hip_layers_per_step": {k: round(v / calls, 2) for k, v in _dense.ROUTING["hip"].items()},
                                        "library_layers_per_step": {k: round(v / calls, 2) for k, v in _dense.ROUTING["library"].items()},
                                        "note": "layers inside fused nodes (tower_bf3_k, mlp_stack / mlp_head autograd nodes) are HIP and not counted here"}
        except Exception:
            pass
        if world > 1:
            res["backend"] = backend + (" (exchange staged through host memory)" if os.environ.get("DIR_SHARD_HOST_STAGED") == "1" else "")
            if "link_bytes" in roof:
                res["per_peer_bytes_per_step"] = roof["link_bytes"]
            # the sharded path checks itself before it is timed: the lookup's rows bit-exact against regenerated table rows on every rank,
            # the number of ranks a collective reached, the lookup's stages in microseconds (all measured before the timed region)
            res.update(multi)
            if "parity_check" in multi:
                res["rccl_ranks_seen" if backend == "nccl" else "ranks_seen"] = multi["parity_check"]["ranks_seen"]
        # ONE clock: roofline.achieved / frac come from the same timed region as value and ms_per_step (host clock around
        # the barrier-bracketed steps, max over ranks); the HIP-event average over the same steps (events on the launch stream) is
        # carried beside it as hip_event_avg_launch_us / frac_hip_events, and the per-launch median further down as frac_at_median
        launch_us = ms_per_step * 1e3
        hip_us = dev_ms * 1e3 / args.steps
        if roof["bound"] == "xgmi":
            link = roof["link_bytes"] / (launch_us * 1e-6) / 1e9
            hbm = roof["alg_bytes"] / (launch_us * 1e-6) / 1e9
            res["roofline"] = {"bound": "xgmi", "achieved": link, "peak": XGMI_LINK_GBS, "unit": "GB/s", "frac": link / XGMI_LINK_GBS,
                               "traffic": None, "kernel": roof["kernel"], "link_bytes_per_step": roof["link_bytes"],
                               "link_bound_us_per_step": roof["link_bytes"] / XGMI_LINK_GBS / 1e3, "avg_step_us": launch_us,
                               "accounting": "bytes one GPU sends to ONE peer per step (ids 8 B + rows 4K B of 1/P of the batch's entries) "
                                             "/ step time, against one xGMI link (153.6 GB/s per direction); every pair has its own link",
                               "hbm": {"alg_bytes_per_launch": roof["alg_bytes"], "achieved_GBps": hbm, "frac_of_8TBps": hbm / HBM_PEAK_GBS}}
        elif roof["bound"] == "hbm":
            ach = roof["alg_bytes"] / (launch_us * 1e-6) / 1e9
            traffic, tmeta = None, {}
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath) and world == 1:
                try:
                    tmeta = json.load(open(tpath))
                    traffic = tmeta.get(wl)
                except Exception:
                    traffic = None
            res["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": ach / HBM_PEAK_GBS, "frac_of_guide_copy_ceiling": ach / HBM_COPY_GBS,
                               "traffic": traffic, "traffic_source": ("profiles/traffic.json (rocprofv3 --pmc passes of this command at %s; not re-measured in this run)"
                                                                      % tmeta.get("_measured_at", "an earlier HEAD")) if traffic else None,
                               "kernel": roof["kernel"], "alg_bytes_per_launch": roof["alg_bytes"],
                               "avg_launch_us": launch_us}
            if world == 1 and wl in ("deepfm_gather_fm", "gather_only"):
                # (a) what the launch costs on the DRAM side: beyond L2 a 64-byte row miss occupies a 128-byte slot
                # (profiles/r01_memprobe_rows.txt: random 64-B rows 2.97 TB/s vs 128-B rows 4.98 TB/s at the same request rate),
                # so the row reads count twice; ids and the concat / logit writes stream.
                row_b = 4 * K
                dram = B * F * (8 + (2 * row_b if row_b < 128 else row_b) + row_b) + (4 * B if wl == "deepfm_gather_fm" else 0)
                res["roofline"]["dram_side_bytes"] = dram
                res["roofline"]["dram_side_GBps"] = dram / (launch_us * 1e-6) / 1e9
                # (b) this box's streaming ceilings, measured now (rotating windows of the tables: nothing cache-resident)
                # (b) this box's streaming ceilings, measured now: at the gather's own size (a ~20 us kernel: ramp and tail included) and
                # asymptotically (512 MB windows); nothing cache-resident (see measured_ceilings)
                import ctypes
                ceil = measured_ceilings(torch, dir_amd.load_library(), slab, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
                res["roofline"]["measured_ceilings"] = ceil
                big = ceil.get("asymptotic") or next(iter(ceil.values()))
                res["roofline"]["measured_read_ceiling_GBps"] = big["read_GBps"]
                res["roofline"]["measured_copy_ceiling_GBps"] = big["copy_GBps"]
                res["roofline"]["guide_copy_ceiling_GBps"] = HBM_COPY_GBS
                res["roofline"]["frac_of_measured_copy_ceiling"] = ach / big["copy_GBps"]
                res["roofline"]["dram_side_frac_of_measured_copy_ceiling"] = res["roofline"]["dram_side_GBps"] / big["copy_GBps"]
                res["roofline"]["dram_side_frac_of_guide_copy_ceiling"] = res["roofline"]["dram_side_GBps"] / HBM_COPY_GBS
                # (c) the same kernel at 1x / 2x / 4x the batch: separates the launch's ramp and tail from its steady state (the rate a
                # 4x longer launch reaches is what the ~60 us one could reach without them)
                if os.environ.get("DIR_BENCH_NO_SWEEP") != "1":
                    sweep = []
                    for mult in (1, 2, 4):
                        Bs = B * mult
                        sids = [torch.randint(0, V, (Bs, F), generator=gen, device=device) for _ in range(2)]
                        so = torch.empty((Bs, F * K), dtype=torch.float32, device=device)
                        sfm = torch.empty((Bs, 1), dtype=torch.float32, device=device)
                        run_b = (lambda i: ops.gather_fm(ts, sids[i % 2], out=so, fm=sfm)) if wl == "deepfm_gather_fm" else \
                                (lambda i: ops.embedding_bag(ts, sids[i % 2], out=so))
                        for i in range(5):
                            run_b(i)
                        torch.cuda.synchronize()
                        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        s0.record()
                        for i in range(40):
                            run_b(i)
                        s1.record()
                        torch.cuda.synchronize()
                        us = s0.elapsed_time(s1) * 1e3 / 40
                        gb = roof["alg_bytes"] * mult / (us * 1e-6) / 1e9
                        sweep.append({"batch": Bs, "avg_launch_us": us, "achieved_GBps": gb, "frac": gb / HBM_PEAK_GBS})
                        del sids, so, sfm
                    res["roofline"]["batch_sweep"] = sweep
                    # (d) why the 4 x B point is 10-12 % slower per row: the [B, 416] output.  One 109 MB buffer overwritten
                    # by every launch lives in the 256 MiB Infinity Cache (behind L2: invisible to the L2 <-> fabric counters), so most of its
                    # write-back never reaches HBM; rotate the output through 4 buffers (436 MB written before a line is reused, what the
                    # 4 x B launch does by itself) and the same kernel at the same B pays for it (tools/gather_out_window_probe.py).
                    ow = []
                    for nb_ in (1, 4):
                        outs_ = [torch.empty((B, F * K), dtype=torch.float32, device=device) for _ in range(nb_)]
                        run_o = (lambda i: ops.gather_fm(ts, idsl[i % len(idsl)], out=outs_[i % nb_], fm=fm)) if wl == "deepfm_gather_fm" else \
                                (lambda i: ops.embedding_bag(ts, idsl[i % len(idsl)], out=outs_[i % nb_]))
                        for i in range(8):
                            run_o(i)
                        torch.cuda.synchronize()
                        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        s0.record()
                        for i in range(40):
                            run_o(i)
                        s1.record()
                        torch.cuda.synchronize()
                        us = s0.elapsed_time(s1) * 1e3 / 40
                        ow.append({"out_buffers": nb_, "MB_written_before_reuse": round(nb_ * B * F * K * 4 / 1e6), "avg_launch_us": us,
                                   "frac": roof["alg_bytes"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS})
                        del outs_
                    res["roofline"]["output_window"] = ow
                    # first-class: what the headline costs when its output really goes to HBM -- the same kernel, same B, the
                    # output rotating through 4 buffers (436 MB written before a line is reused: nothing of it survives in the Infinity Cache)
                    res["roofline"]["frac_output_to_hbm"] = ow[1]["frac"]
                    res["roofline"]["ms_per_step_output_to_hbm"] = ow[1]["avg_launch_us"] * 1e-3
                    res["roofline"]["output_to_hbm_note"] = ("frac / ms_per_step: ONE output buffer overwritten by every launch (109 MB: most of its write-back "
                                                             "stays in the 256 MiB Infinity Cache); *_output_to_hbm: the output rotating through 4 buffers")
        else:
            # achieved = the algorithmic flops of each kernel priced on the pipe mode it executes on (PIPE_COST), in bf16-MFMA flops per
            # second, against the dense bf16 peak: the share of the step the matrix pipe needs at peak
            pipe_fl = roof.get("pipe_flops", roof["alg_flops"] * PIPE_COST["f32"])
            ach = pipe_fl / (launch_us * 1e-6) / 1e12
            res["roofline"] = {"bound": "mfma", "achieved": ach, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                               "frac": ach / MFMA_BF16_PEAK_TF, "traffic": None, "kernel": roof["kernel"],
                               "accounting": "bf16-MFMA flops: algorithmic fp32 flops x 6 where the kernel runs the bf16x3 split, x 16 where it "
                                             "runs fp32-input MFMA (1/16 of the bf16 rate); peak = dense bf16 MFMA",
                               "pipe_flops_per_step": pipe_fl, "alg_flops_per_step": roof["alg_flops"], "avg_step_us": launch_us,
                               "fp32_equiv_TFLOPs": roof["alg_flops"] / (launch_us * 1e-6) / 1e12}
            for k in ("modes", "note", "survey_8d_flops"):
                if k in roof:
                    res["roofline"][k] = roof[k]
            if "hbm_bytes_full_T" in roof:        # the DIN unit: SURVEY 8d's bytes beside the matrix-pipe share
                res["roofline"]["hbm_frac"] = roof["hbm_bytes_full_T"] / (launch_us * 1e-6) / 1e9 / HBM_PEAK_GBS
                res["roofline"]["hbm_frac_length_aware"] = roof["hbm_bytes_length_aware"] / (launch_us * 1e-6) / 1e9 / HBM_PEAK_GBS
                res["roofline"]["hbm_bytes_full_T"] = roof["hbm_bytes_full_T"]
                res["roofline"]["hbm_bytes_length_aware"] = roof["hbm_bytes_length_aware"]
            if "dtype" in roof:
                res["dtype"] = roof["dtype"]
        res["roofline"]["clock"] = "host clock over the timed region (the one ms_per_step and value use)"
        res["roofline"]["hip_event_avg_launch_us"] = hip_us
        res["roofline"]["frac_hip_events"] = res["roofline"]["frac"] * launch_us / hip_us if hip_us > 0 else None
        if world == 1 and roof["bound"] in ("hbm", "mfma"):
            # per-launch distribution (median + p10 / p90 of >= 100 launches), at any --steps, outside the timed region: one HIP
            # event per step on the launch stream (events inside the timed region would put a timestamp packet between the launches)
            nl = 200 if launch_us < 2000 else 30
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(nl + 1)]
            evs[0].record()
            for i in range(nl):
                step(i)
                evs[i + 1].record()
            torch.cuda.synchronize()
            per = sorted(evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(nl))
            res["roofline"]["launch_us_median"] = per[nl // 2]
            res["roofline"]["launch_us_p10"] = per[nl // 10]
            res["roofline"]["launch_us_p90"] = per[(nl * 9) // 10]
            res["roofline"]["launch_us_samples"] = nl
            if roof["bound"] == "hbm":
                res["roofline"]["frac_at_median"] = roof["alg_bytes"] / (per[nl // 2] * 1e-6) / 1e9 / HBM_PEAK_GBS
            else:
                res["roofline"]["frac_at_median"] = res["roofline"]["pipe_flops_per_step"] / (per[nl // 2] * 1e-6) / 1e12 / MFMA_BF16_PEAK_TF
        if world == 1 and wl == "deepfm_gather_fm" and args.id_dist == "uniform":
            # secondary, cache-assisted case: Zipf(1.05) ids, rows read with the cacheable policy
            import copy
            zargs = copy.copy(args)
            zargs.id_dist = "zipf"
            zids = make_ids(torch, zargs, gen, device, V)
            ts.row_policy = "reuse"
            for i in range(200):                 # (10 ms: keeps the clocks where the primary leg left them)
                ops.gather_fm(ts, zids[i % len(zids)], out=out, fm=fm)
            torch.cuda.synchronize()
            z0, z1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            z0.record()
            for i in range(100):
                ops.gather_fm(ts, zids[i % len(zids)], out=out, fm=fm)
            z1.record()
            torch.cuda.synchronize()
            zus = z0.elapsed_time(z1) * 10.0
            res["secondary_zipf"] = {"ids": "zipf(1.05)", "row_policy": "reuse", "avg_launch_us": zus,
                                     "samples_per_s": B / (zus * 1e-6), "achieved_GBps": roof["alg_bytes"] / (zus * 1e-6) / 1e9,
                                     "frac": roof["alg_bytes"] / (zus * 1e-6) / 1e9 / HBM_PEAK_GBS}
            ts.row_policy = "auto"
        if cpu is not None:
            # the reference's CPU path beside every samples/s figure (north_star): the oracle's port of the same op, in the child started
            # before the GPU was touched, on a BOUNDED sample of the same workload (full batch where a pass is cheap)
            secs = args.cpu_seconds
            if wl in ("deepfm_gather_fm", "gather_only"):
                req = {"workload": "gather_fm", "samples": B, "fields": F, "dim": K, "vocab": V, "seconds": secs, "warm": 3, "min_passes": 10}
            elif wl == "dcn_cross":
                req = {"workload": "dcn_cross", "samples": B, "d": args.cross_d, "layers": 3, "seconds": secs, "warm": 3, "min_passes": 10}
            elif wl == "din":
                req = {"workload": "din", "samples": min(B, 8192), "T": 50, "dim": 64, "vocab": 10000000, "H1": 80, "H2": 40, "seconds": secs}
            else:
                req = {"workload": "cin", "samples": min(B, 1024), "m": F, "D": K, "layers": [128, 128, 128], "seconds": secs}
            res["cpu_baseline"] = cpu.ask(req)
            if secondary is not None and "error" not in secondary:
                secondary["cpu_baseline"] = cpu.ask({"workload": "cin", "samples": min(B, 1024), "m": 26, "D": 16, "layers": [128, 128, 128],
                                                     "seconds": min(secs, 8.0)})
            cpu.close()
        if secondary is not None:
            res["secondary_cfg5_xdeepfm_cin"] = secondary
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
