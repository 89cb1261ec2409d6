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


def test_micro_batch_spec_and_flop_counts():
    """`--micro` (the GA windows of BASELINE configs 3 / 4: scripts/train/examples/run_train_4B_z2_b1.sh:29,47,
    run_train_8B_z0_b1.sh:29,47) and the executed-attention count it feeds; the secondary block's commands name those configs."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    B = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(B)
    assert B.parse_micro("protein:512") == [[("protein", 512)]]
    assert B.parse_micro("protein:1024;dna:1000") == [[("protein", 1024)], [("dna", 1000)]]
    c3 = B.parse_micro(B.SECONDARY["c3"][B.SECONDARY["c3"].index("--micro") + 1])
    assert len(c3) == 2 and c3[0] == [("dna", 512), ("rna", 512), ("protein", 512)]
    assert B.SECONDARY["c3"][:6] == ["--model", "4b", "--batch", "1", "--seq", "3072"]
    assert B.SECONDARY["c4"][:6] == ["--model", "8b", "--batch", "1", "--seq", "4096"]
    from molly_amd import config as C
    cfg = C.molly("1.7b")
    enc = {"dna_rna": cfg.dna_rna_config, "protein": cfg.protein_config}
    one = B.attention_flops_per_step(cfg.text_config, enc, 8, 2048, [[("protein", 512)]])
    t = cfg.text_config
    want = 3 * 4 * t.num_hidden_layers * t.num_attention_heads * t.head_dim * 1024 * 2048 * 8 + \
        4 * cfg.protein_config.num_hidden_layers * cfg.protein_config.hidden_size * 512 * 512 * 8
    assert one == want
    two = B.attention_flops_per_step(cfg.text_config, enc, 8, 2048, [[("protein", 512)], [("dna", 512)]])
    assert two > 2 * one - 4 * 33 * 1280 * 512 * 512 * 8 - 1 and two < 2 * one       # NT-500M has 24 layers, ESM2-650M 33
