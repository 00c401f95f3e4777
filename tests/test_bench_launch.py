"""bench.py's launch contract: `python bench.py --gpus N` starts the N ranks itself (fresh child processes, the
parent never touches the GPU) and never prints a line whose n_gpus differs from what was asked for."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(argv, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + argv, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


def _visible_gpus():
    import torch
    return torch.cuda.device_count()


def test_more_ranks_than_devices_fails_without_a_line():
    """N != visible devices: non-zero exit, the reason on stderr, NO result line (here: N = devices + 2 so the case exists on every box)."""
    n = _visible_gpus() + 2
    r = _run(["--gpus", str(n), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], timeout=300)
    assert r.returncode != 0
    assert "visible GPU" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.lstrip().startswith("{")]


def test_world_size_mismatch_is_refused():
    """A launcher that started a different number of ranks than --gpus names is refused (exit 2) before torch is imported."""
    r = _run(["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, timeout=60)
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr and not r.stdout.strip()
    r = _run(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"}, timeout=60)
    assert r.returncode == 2 and not r.stdout.strip()


def test_launcher_parent_does_not_import_torch():
    """The parent of the ranks must stay clear of the GPU: launch_ranks() runs before `import torch` and imports nothing that does."""
    src = open(BENCH).read()
    main = src[src.index("def main():"):]
    assert main.index("return launch_ranks(args)") < main.index("import torch")
    body = src[src.index("def launch_ranks(args):"):src.index("CPU_BASELINE_WORKLOADS = (")]
    assert "import torch" not in body and "dir_amd" not in body


@pytest.mark.gpu
def test_gpus_2_launches_two_ranks(built_lib):
    """`python bench.py --gpus 2` on the one-GPU test box: two ranks on cuda:0, gloo with the exchange staged through host memory (RCCL
    refuses two ranks on one device; on a node the same command runs one rank per GPU over RCCL)."""
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8192", "--no-cpu-baseline"],
             {"DIR_BENCH_BACKEND": "gloo", "DIR_BENCH_SAME_DEVICE": "1", "DIR_SHARD_HOST_STAGED": "1", "DIR_BENCH_NO_SECONDARY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.lstrip().startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["world_size"] == 2 and res["launcher"] == "bench.py"
    assert res["steps"] == 3 and res["scaling"] == "weak"
    assert res["roofline"]["bound"] == "xgmi" and res["per_peer_bytes_per_step"] == 8192 * 26 * (64 + 8) / 2
    assert res["value"] > 0
    # the sharded path checked itself before it was timed: rows bit-exact against regenerated table rows on every rank,
    # the ranks a collective reached, the lookup's stages in microseconds
    pc = res["parity_check"]
    assert pc["ok"] is True and pc["ranks_seen"] == 2 and pc["ranks_rows_bit_exact"] == 2 and pc["ranks_fm_bit_exact"] == 2
    assert pc["samples_per_rank"] == 4096 and res["ranks_seen"] == 2
    assert set(res["stage_us"]) >= {"bucket", "a2a_ids", "owner_gather", "a2a_rows", "finish", "sum"}
    assert all(res["stage_us"][k] > 0 for k in ("bucket", "a2a_ids", "owner_gather", "a2a_rows", "finish"))
    assert res["roofline"]["hip_event_avg_launch_us"] > 0 and "clock" in res["roofline"]


@pytest.mark.gpu
def test_gpus_2_cfg5_leg_checks_itself(built_lib):
    """The config-5 secondary leg of the same command (xDeepFM CIN on the row-sharded 1e8-row table; here a small batch): its own parity check."""
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4096", "--no-cpu-baseline"],
             {"DIR_BENCH_BACKEND": "gloo", "DIR_BENCH_SAME_DEVICE": "1", "DIR_SHARD_HOST_STAGED": "1", "DIR_BENCH_CFG5_ROWS": "2600000", "DIR_BENCH_CFG5_WARMUP": "2"})
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([l for l in r.stdout.splitlines() if l.lstrip().startswith("{")][-1])
    sec = res["secondary_cfg5_xdeepfm_cin"]
    assert "error" not in sec, sec
    assert sec["n_gpus"] == 2 and sec["parity_check"]["ok"] is True and sec["parity_check"]["ranks_seen"] == 2


def test_synthetic_rows_are_a_closed_form():
    """bench.synth_rows: any rank can regenerate any row (what the N > 1 parity check relies on); shards are slices of one function."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    full = bench.synth_rows(torch, 5, torch.arange(0, 20000), 16, 0.25)
    assert torch.equal(bench.synth_shard(torch, 5, 777, 9000, 16, 0.25, "cpu", block=1000), full[777:9000])
    pick = torch.tensor([19999, 0, 31, 31])
    assert torch.equal(bench.synth_rows(torch, 5, pick, 16, 0.25), full[pick])
    assert abs(float(full.mean())) < 5e-3 and abs(float(full.std()) - 0.25) < 5e-3
    assert not torch.equal(bench.synth_rows(torch, 6, pick, 16, 0.25), full[pick])
