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
