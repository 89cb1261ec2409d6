"""CPU: `python bench.py --gpus N` with no launcher around it starts its N ranks itself (the driver's stand-alone call), and
refuses a WORLD_SIZE that contradicts --gpus instead of asserting half way in."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(kw)
    return e


def test_self_launch_builds_one_rank_per_gpu():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "3", "--warmup", "1", "--dry-run-launch"],
                       capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 0, r.stderr
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(BENCH) + 1:]
    assert tail == ["--gpus", "4", "--steps", "3", "--warmup", "1"]          # the ranks get the caller's flags verbatim


def test_self_launch_really_spawns_ranks(tmp_path):
    """End to end without a GPU: the children start under torch.distributed.run, see WORLD_SIZE=2 and get as far as the
    device selection, where a CPU-only container has to stop — with the launcher propagating the non-zero exit code."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=_env(MOLLY_DIST_BACKEND="gloo"), timeout=600)
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert r.returncode == 0, r.stderr[-2000:]
        line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == 2 and line["comm"]["world_size"] == 2
    else:
        assert r.returncode != 0
        assert "torch.distributed" in r.stderr or "ChildFailedError" in r.stderr or "cuda" in r.stderr.lower()


def test_world_size_mismatch_is_a_clear_error():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--no-cpu-baseline"], capture_output=True, text=True,
                       env=_env(WORLD_SIZE="4", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
