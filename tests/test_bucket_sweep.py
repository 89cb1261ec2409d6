"""CPU: the bookkeeping of bench.py's bucket-size x reduce-scatter-algorithm sweep (molly_amd.trainer.zero2.sweep_exchange; reference
role: the fixed `reduce_bucket_size` / `allgather_bucket_size` of src/configs/ds_z2_config.json:18-27) and the no-exchange
communicator bench.py uses to measure what the overlap leaves exposed — no GPU, no process group."""
import os
import subprocess
import sys

import pytest
import torch

from molly_amd.trainer.zero2 import _NullComm, sweep_exchange

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sweep_picks_the_fastest_and_reports_every_candidate():
    times = {(64, "rccl"): 310.0, (128, "rccl"): 305.5, (256, "rccl"): 301.25, (64, "a2a"): 299.0, (128, "a2a"): 299.0, (256, "a2a"): 307.0}
    calls = []

    def measure(mib, algo):
        calls.append((mib, algo))
        return times[(mib, algo)]
    cands = [(m, a) for a in ("rccl", "a2a") for m in (64, 128, 256)]
    r = sweep_exchange(cands, measure, 2)
    assert calls == cands                                         # every candidate measured once, in the order given
    assert r["ms_per_step"] == {"64/rccl": 310.0, "128/rccl": 305.5, "256/rccl": 301.25, "64/a2a": 299.0, "128/a2a": 299.0, "256/a2a": 307.0}
    assert r["chosen"] == {"bucket_mib": 64, "rs_algo": "a2a", "key": "64/a2a"}      # a tie goes to the earlier (smaller) bucket
    assert r["steps_each"] == 2 and "errors" not in r


def test_sweep_survives_candidates_that_fail():
    def measure(mib, algo):
        if mib == 1024:
            raise RuntimeError("HIP out of memory")
        if algo == "a2a":
            raise NotImplementedError("all_to_all_single on this backend")
        return 100.0 + mib
    r = sweep_exchange([(m, a) for a in ("rccl", "a2a") for m in (512, 1024)], measure, 1)
    assert r["chosen"]["key"] == "512/rccl" and r["ms_per_step"]["1024/rccl"] is None and r["ms_per_step"]["512/a2a"] is None
    assert set(r["errors"]) == {"1024/rccl", "512/a2a", "1024/a2a"} and "out of memory" in r["errors"]["1024/rccl"]
    with pytest.raises(RuntimeError, match="every candidate failed"):
        sweep_exchange([(64, "a2a")], measure, 1)


def test_a_build_failure_on_one_rank_is_skipped_by_every_rank_together():
    """`build` (the optimizer with the candidate's layout) can fail on ONE rank (out of memory on the fullest one).  The verdict
    goes through `agree` — here a stand-in for the all-reduce MIN: 'rank 1' fails the 1 GiB layouts — and the candidate is recorded with the
    failing rank's message and NOT measured by anybody; a failure inside `measure` (collectives in flight) is not caught at all."""
    built, measured = [], []

    def build(mib, algo):
        built.append((mib, algo))

    def agree(ok, err=None):
        mib = built[-1][0] if built else 0
        if ok and mib == 1024:
            return False, "rank 1: OutOfMemoryError: HIP out of memory"      # another rank failed
        return ok, err

    def measure(mib, algo):
        measured.append((mib, algo))
        return 300.0 - mib / 100.0
    r = sweep_exchange([(256, "rccl"), (1024, "rccl"), (512, "rccl")], measure, 2, build=build, agree=agree)
    assert built == [(256, "rccl"), (1024, "rccl"), (512, "rccl")] and measured == [(256, "rccl"), (512, "rccl")]
    assert r["ms_per_step"]["1024/rccl"] is None and "rank 1" in r["errors"]["1024/rccl"] and r["chosen"]["key"] == "512/rccl"

    def bad_measure(mib, algo):
        raise RuntimeError("NCCL error in the settling step")
    with pytest.raises(RuntimeError, match="settling step"):           # aborts the job: never 'skip and go on' with collectives enqueued
        sweep_exchange([(256, "rccl"), (512, "rccl")], bad_measure, 2, build=build, agree=lambda ok, err=None: (ok, err))


def test_the_sweep_keeps_the_best_so_far_when_its_wall_budget_runs_out():
    """--tune-budget-s.  A fake clock: every candidate 'takes' 50 s, the budget is 120 s — the fourth candidate is not
    started (the check runs before each candidate after the first, through `agree`: one rank over budget stops all), the result says so."""
    now = [0.0]

    def measure(mib, algo):
        now[0] += 50.0
        return 400.0 - mib
    asked = []

    def agree(ok, err=None):
        asked.append(ok)
        return ok, err
    r = sweep_exchange([(64, "rccl"), (128, "rccl"), (256, "rccl"), (512, "rccl"), (1024, "rccl")], measure, 2, agree=agree, budget_s=120.0,
                       clock=lambda: now[0])
    assert list(r["ms_per_step"]) == ["64/rccl", "128/rccl", "256/rccl"] and r["chosen"]["key"] == "256/rccl"
    assert r["truncated"] is True and r["not_run"] == ["512/rccl", "1024/rccl"] and r["budget_s"] == 120.0
    assert asked == [True, True, False]
    r2 = sweep_exchange([(64, "rccl"), (128, "rccl")], measure, 2, budget_s=1e9, clock=lambda: now[0])
    assert "truncated" not in r2
    # `chosen` is what rank 0 broadcasts
    r3 = sweep_exchange([(64, "rccl")], lambda m, a: 1.0, 1, broadcast=lambda obj: dict(obj, via="rank0"))
    assert r3["chosen"]["via"] == "rank0"


def test_p2p_refuses_itself_without_peer_access_or_under_expandable_segments():
    """the direct peer exchange decides BEFORE mapping anything (trainer/p2p.py::p2p_refusal, then an agreement over
    all ranks; Zero2Optimizer falls back to rs_algo='a2a' and records why)."""
    from molly_amd.trainer.p2p import p2p_refusal
    full = lambda a, b: True
    assert p2p_refusal([0, 1, 2, 3], 2, can_access=full, alloc_conf="") is None
    assert p2p_refusal([0, 0, 0, 0], 1, can_access=lambda a, b: False, alloc_conf="") is None      # ranks sharing one device (the test harness)
    why = p2p_refusal([0, 1, 2, 3], 0, can_access=lambda a, b: (a, b) != (0, 3), alloc_conf="")
    assert why and "device 0" in why and "device 3" in why and "hipDeviceCanAccessPeer" in why
    assert "expandable segments" in p2p_refusal([0, 1], 0, can_access=full, alloc_conf="max_split_size_mb:64, expandable_segments:True")


def test_fractional_bucket_sizes_keep_distinct_keys():
    r = sweep_exchange([(3.3, "rccl"), (64.0, "rccl")], lambda m, a: m, 1)
    assert set(r["ms_per_step"]) == {"3.3/rccl", "64/rccl"} and r["chosen"]["bucket_mib"] == 3.3


def test_null_comm_exchanges_nothing():
    c = _NullComm()
    region = torch.arange(8.0)
    c.reduce_scatter(region[2:4], region)
    c.all_gather(region, region[2:4])
    c.all_reduce(region)
    c.all_reduce_region(region)
    assert torch.equal(region, torch.arange(8.0)) and c.rs_algo == "none"


def test_bench_dry_run_launch_passes_the_sweep_flags_through():
    """`--dry-run-launch` prints the rank launch command instead of running it: the sweep's flags reach the ranks unchanged."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run-launch", "--bucket-ab-steps", "3",
                        "--bucket-ab-mib", "64,256"], capture_output=True, text=True, timeout=120,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    cmd = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])["launch"]
    assert "--nproc-per-node=8" in cmd and "127.0.0.1" in cmd
    i = cmd.index("--bucket-ab-steps")
    assert cmd[i + 1] == "3" and cmd[cmd.index("--bucket-ab-mib") + 1] == "64,256" and "--dry-run-launch" not in cmd
