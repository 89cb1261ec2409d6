"""GPU: the multi-rank training path end to end with TWO processes on the ONE GPU of the test box (gloo moves the bytes; RCCL
refuses two ranks per device, and the 8-GPU run belongs to the driver): real HIP kernels, overlapped reduce-scatter /
all-gather on the communication stream with the engine's per-layer hooks.  After two steps on different per-rank batches the
replicas are bitwise identical, and equal a single process that averages the two micro-batch gradients."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _model(meta):
    from test_gpu_lora import _build
    return _build(meta)


def _batches(meta):
    from molly_amd.synth import synth_batch
    sp = {k: tuple(v) for k, v in meta["config"]["special_ids"].items()}
    return [synth_batch(2, 256, [("protein", 64)], seed=50 + r, text_vocab=1000, special_ids=sp, pad_id=1000) for r in range(2)]


def _args(b):
    return [b[k] for k in ("input_ids", "attention_mask", "omic_ids", "omic_info_list", "labels")]


def _worker(rank, world, port, meta, ret, stage=2, rs_algo=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from molly_amd.trainer import Zero2Optimizer
    m = _model(meta)
    opt = Zero2Optimizer(m._rt.P.flat, m._rt.G.flat, m.n_decay, lr=1e-3, weight_decay=1e-2, max_grad_norm=1.0,
                         chunk_elems=1 << 18, stage=stage, rs_algo=rs_algo)   # several buckets even on the tiny model
    assert opt.overlap and opt.world == 2 and len(opt.buckets) > 2
    m.attach_optimizer(opt)
    b = _batches(meta)[rank]
    norms, after = [], []
    for _ in range(2):
        m.forward_backward(*_args(b))
        norms.append(float(opt.step(lr=1e-3).item()))
        opt.wait_all_params()
        torch.cuda.synchronize()
        after.append(m._rt.P.flat.cpu().clone())
    ret[rank] = (after[1], norms, after[0])
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_overlapped_zero2(tiny_meta):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, 29571, tiny_meta, ret), nprocs=2, join=True)
    (P0, n0, _), (P1, n1, _) = ret[0], ret[1]
    assert torch.equal(P0, P1) and n0 == n1                       # replicas bit-identical after the all-gather
    # single process: same two micro-batches, gradients summed (GA semantics) then averaged by the 1/world the optimizer
    # folds into its clip scale -> emulate with grad_scale through two accumulate steps and a halved gradient
    from molly_amd.trainer import Zero2Optimizer
    m = _model(tiny_meta)
    opt = Zero2Optimizer(m._rt.P.flat, m._rt.G.flat, m.n_decay, lr=1e-3, weight_decay=1e-2, max_grad_norm=1.0)
    b = _batches(tiny_meta)
    norms = []
    for _ in range(2):
        m.forward_backward(*_args(b[0]))
        m.forward_backward(*_args(b[1]), accumulate=True)
        m._rt.G.flat.mul_(0.5)                                    # exact in bf16
        norms.append(float(opt.step(lr=1e-3).item()))
    torch.cuda.synchronize()
    ref = m._rt.P.flat.cpu()
    # the two-rank run sums bf16 gradients across ranks in the collective, the single process accumulates inside the
    # kernels' fp32 epilogues: same mathematics, different rounding -> agree to bf16 resolution of a 1e-3 update
    assert abs(norms[0] - n0[0]) <= 2e-2 * norms[0]
    d = (P0.float() - ref.float()).abs()
    assert (d <= 2 ** -7 * ref.float().abs() + 2.5e-3).all(), d.max().item()      # <= one bf16 step of the parameter
    assert (P0 != ref).float().mean().item() < 0.35


def test_two_ranks_zero0_fallback_equals_zero2(tiny_meta):
    """SURVEY.md 8(e) ZeRO-0 fallback on the real HIP path: overlapped per-bucket all-reduce, whole-buffer AdamW on the side
    stream under the next forward, no all-gather.  Against the ZeRO-2 run the summed gradients are bit-identical (one bf16
    rounding per two-term sum); the squared gradient norm is one pass over the buffer instead of per-shard partial sums, so
    the clip coefficient moves in its last fp32 bits and a few parameters land on the other side of a bf16 rounding."""
    mgr = mp.Manager()
    res = {}
    for stage, port in ((2, 29575), (0, 29577)):
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, port, tiny_meta, ret, stage), nprocs=2, join=True)
        assert torch.equal(ret[0][0], ret[1][0]) and ret[0][1] == ret[1][1]
        res[stage] = ret[0]
    # Step 1 starts from identical parameters and gradients; the squared norm is summed in two orders -> fp32 resolution, and
    # the clip coefficient's last bits put a handful of parameters on the other side of a bf16 rounding (measured: 3-4 of 1.9 M).
    assert abs(res[0][1][0] - res[2][1][0]) <= 4e-7 * res[2][1][0], (res[0][1], res[2][1])
    a, b = res[0][2].float(), res[2][2].float()
    assert ((a - b).abs() <= 2 ** -6 * b.abs() + 1e-30).all()
    assert (a != b).float().mean().item() < 1e-4
    # Step 2 is the same computation from parameters that differ in those few roundings.  How far its gradient norm moves depends
    # on WHICH weights they hit (one bf16 step on one 3.8e-3 weight moved it by 1.4e-4; flips on 1e-7-sized weights by 1e-7), and
    # the clip coefficient, hence every update, moves with it: bounded, not pinned.
    assert abs(res[0][1][1] - res[2][1][1]) <= 1e-3 * res[2][1][1], (res[0][1], res[2][1])
    a, b = res[0][0].float(), res[2][0].float()
    assert (a - b).abs().max().item() <= 2.5e-3                          # <= the two updates of lr = 1e-3 themselves


def test_two_ranks_all_to_all_reduce_scatter(tiny_meta):
    """rs_algo="a2a" (SURVEY.md 5 option 2) on the real kernels: all_to_all_single + molly_reduce_rows_bf16 (fp32 sum of the
    received copies in rank order), overlapped like the library reduce-scatter.  Two ranks: one rounding per sum either way, so
    the parameters equal the library path bit for bit."""
    mgr = mp.Manager()
    res = {}
    for algo, port in (("rccl", 29581), ("a2a", 29583)):
        ret = mgr.dict()
        mp.spawn(_worker, args=(2, port, tiny_meta, ret, 2, algo), nprocs=2, join=True)
        assert torch.equal(ret[0][0], ret[1][0]) and ret[0][1] == ret[1][1]
        res[algo] = ret[0]
    assert torch.equal(res["a2a"][0], res["rccl"][0]) and res["a2a"][1] == res["rccl"][1]


def _worker_any_world(rank, world, port, meta, ret, rs_algo):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from molly_amd.synth import synth_batch
    from molly_amd.trainer import Zero2Optimizer
    m = _model(meta)
    opt = Zero2Optimizer(m._rt.P.flat, m._rt.G.flat, m.n_decay, lr=1e-3, weight_decay=1e-2, max_grad_norm=1.0,
                         chunk_elems=1 << 17, stage=2, rs_algo=rs_algo)
    assert opt.overlap and opt.world == world and len(opt.buckets) > 2 and opt.rs_algo == rs_algo
    m.attach_optimizer(opt)
    sp = {k: tuple(v) for k, v in meta["config"]["special_ids"].items()}
    b = synth_batch(2, 256, [("protein", 64)], seed=70 + rank, text_vocab=1000, special_ids=sp, pad_id=1000)
    norms = []
    for _ in range(3):
        m.forward_backward(*_args(b))
        norms.append(float(opt.step(lr=1e-3).item()))
        opt.wait_all_params()
        torch.cuda.synchronize()
    if rs_algo == "p2p":
        opt.comm.check()                                     # no spin gave up
    ret[rank] = (m._rt.P.flat.cpu().clone(), norms)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_direct_peer_exchange_equals_all_to_all_bit_for_bit(tiny_meta, world):
    """rs_algo="p2p" (SURVEY.md 5 option 3, trainer/p2p.py + csrc/p2p.hip): every rank maps the peers' gradient and parameter buffers
    (CUDA IPC), the owner of a chunk READS the other copies in place and adds them in fp32 in rank order, then WRITES its updated
    parameter chunk into the peers' buffers; flags that only grow order producers and consumers.  `world` processes on the one GPU
    of the box, three overlapped steps on different per-rank batches: the replicas are bit-identical, and bit-identical to the
    all-to-all reduce-scatter (the same arithmetic on copies that were moved instead of read in place).  Never run over links: no
    speed is claimed."""
    mgr = mp.Manager()
    res = {}
    for algo in ("a2a", "p2p"):
        ret = mgr.dict()
        mp.spawn(_worker_any_world, args=(world, _free_port(), tiny_meta, ret, algo), nprocs=world, join=True)
        for r in range(1, world):
            assert torch.equal(ret[0][0], ret[r][0]) and ret[0][1] == ret[r][1], (algo, r)
        res[algo] = ret[0]
    assert torch.equal(res["p2p"][0], res["a2a"][0]) and res["p2p"][1] == res["a2a"][1]


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_launches_its_own_ranks_end_to_end():
    """`python bench.py --gpus 2` with NO launcher around it (what the driver runs): bench.py starts the two ranks itself, they
    run the pre-flight, the timed steps and the exposed-communication measurement, and rank 0 prints the one JSON line.  Both
    ranks share the one GPU of the test box and gloo moves the bytes (RCCL refuses two ranks per device)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MOLLY_BENCH_DEVICE="0", MOLLY_DIST_BACKEND="gloo", MOLLY_BENCH_WATCHDOG_S="600")     # (a hang = the ranks' stacks on stderr, not a silent timeout)
    # (--tune-budget-s 0.01: the pre-warm-up tuning runs out of its wall budget behind the first candidate of each sweep — the ranks stop
    # TOGETHER, keep the best so far and say so: the truncated sweep rehearsed on real kernels)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--seq", "1024", "--k-protein", "256", "--exposed-comm-steps", "2", "--bucket-ab-steps", "2", "--tune-budget-s", "0.01"], capture_output=True,
                       text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                    # exactly ONE JSON line on stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0 and "cpu_baseline" not in d
    c = d["comm"]
    assert c["world_size"] == 2 and c["backend"] == "gloo" and c["preflight"]["all_reduce"] is True
    assert c["comm_bytes_per_step_per_gpu"] > 0 and c["buckets"] >= 1 and "step_ms_p50_no_overlap" in c
    assert d["config"]["global_batch"] == 4
    # round 4: the launch-shape A/B before the warm-up, the per-bucket timings and the backend's own count of the world
    ab = c["gemm_mode_ab"]
    assert set(ab["ms_per_step"]) == {"-3"} and ab["truncated"] is True and ab["chosen"] == "-3" and ab["ms_per_step"]["-3"] > 0
    assert str(c["gemm_blocks_mode"]) == ab["chosen"]
    assert c["world_size_by_all_reduce"] == 2
    t = c["timings_us"]
    assert t["reduce_scatter"]["n"] == 2 * c["buckets"] and t["all_gather"]["n"] == 2 * c["buckets"]          # 2 timed steps
    assert len(t["reduce_scatter"]["us_per_bucket"]) == c["buckets"] and t["all_gather"]["us_p50"] > 0
    # round 5: the bucket size x reduce-scatter algorithm sweep before the warm-up (every candidate rebuilt the optimizer and ran on
    # both ranks), the layout the timed region ran with, and what the overlap left exposed (same steps with no exchange at all)
    ab = c["bucket_ab"]
    every = [f"{m}/{a}" for a in ("rccl", "a2a") for m in ("64", "128", "256", "512", "1024")]
    assert list(ab["ms_per_step"]) == every[:1] and ab["truncated"] is True and ab["not_run"] == every[1:] and ab["budget_s"] == 0.01
    ok = {k: v for k, v in ab["ms_per_step"].items() if v is not None}
    assert len(ok) == 1 and all(v > 0 for v in ok.values()), ab
    ch = ab["chosen"]
    assert ch["key"] in ok and ok[ch["key"]] == min(ok.values()) and c["rs_algo"] == ch["rs_algo"]
    assert abs(c["bucket_mib"] - ch["bucket_mib"]) < 1.0, (c["bucket_mib"], ch)
    assert c["step_ms_p50_no_exchange"] > 0 and "exposed_comm_ms" in c
    assert d["batch8_reference"] is None                          # (N > 1: no batch-8 continuity figure)


def test_four_ranks_contend_for_one_chip():
    """`bench.py --gpus 4` on the ONE GPU of the box (gloo as the transport): four ranks' GEMMs, communication
    streams and optimizer kernels share the chip — each launch shape of the 256x256 GEMM (-3 | dyn | 0) runs for real beside the
    other ranks' kernels during the launch-shape A/B, then the timed steps run in the chosen one.  Nothing above world 2 had ever
    executed the overlapped path; a hang or a wrong ticket shows here as a timeout or a diverged loss.  Reference role: the 8 ranks
    of scripts/train/examples/run_train_4B_z2_b1.sh:60-66."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MOLLY_GEMM_PERSISTENT_MULTI")}
    env.update(MOLLY_BENCH_DEVICE="0", MOLLY_DIST_BACKEND="gloo", MOLLY_BENCH_WATCHDOG_S="600")
    # (Qwen3-0.6B as the decoder: gloo moves every byte of the exchange through host memory — 1.5 GB per step instead of the 1.7B model's 3.4 GB; the
    # kernels, streams, hooks and launch shapes under test are the same)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--model", "0.6b", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--seq", "1024", "--k-protein", "256", "--exposed-comm-steps", "0", "--gemm-mode-ab-steps", "1",
                        "--bucket-ab-steps", "0"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    c = d["comm"]
    assert d["n_gpus"] == 4 and c["world_size"] == 4 and c["world_size_by_all_reduce"] == 4 and d["config"]["global_batch"] == 8
    # (round 6: the bucket sweep is NOT rehearsed here any more.  Four gloo ranks on one GPU hung in the first candidate's reduce-scatter on most
    # boxes of the pool — at the commit that had passed twice the same day, with either transport, on loopback, with a drain + barrier between
    # layouts: profiles/r06_logs/four_ranks_hang.log.  gloo blocks the host in every collective and moves the bytes through host memory; the
    # sweep's logic runs at world 2 on real kernels (the test above), at world 2 and 4 on CPU (test_zero2_gloo.py))
    assert c.get("bucket_ab") is None
    assert c["buckets"] >= 2 and c["overlap"] is True
    assert set(c["gemm_mode_ab"]["ms_per_step"]) == {"-3", "dyn", "0"}
    assert 0.5 < d["loss"] < 20.0                                           # the step still trains (random-init CE ~ ln V = 11.9)


def test_failed_bring_up_prints_one_json_line_and_exits_non_zero():
    """a communication bring-up that fails ends the run from a fresh state — ONE JSON line carrying the error on
    stdout, exit code != 0, nothing retried (never a re-exec of a GPU-initialised process).  Provoked here with a backend name
    torch.distributed does not know."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MOLLY_BENCH_DEVICE="0", MOLLY_DIST_BACKEND="no_such_backend")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 2 and "init_process_group" in d["error"], d


def _rccl_world1_worker(rank, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    import datetime
    dist.init_process_group("nccl", rank=0, world_size=1, timeout=datetime.timedelta(minutes=5), device_id=dev)
    from molly_amd.trainer.zero2 import _DistComm, preflight_collectives
    rep = preflight_collectives(dev)
    # the in-place forms the ZeRO step uses, on a bucket-sized buffer, through RCCL's own kernels (world 1: a copy, but the library
    # is loaded, the communicator exists, the aliased send / receive buffers are accepted and the stream semantics are RCCL's)
    comm = _DistComm()
    g = torch.arange(1 << 22, device=dev).remainder(251).to(torch.bfloat16)
    want = g.clone()
    side = torch.cuda.Stream(device=dev, priority=-1)
    ev = torch.cuda.Event(); ev.record(); side.wait_event(ev)
    with torch.cuda.stream(side):
        comm.reduce_scatter(g, g)
        comm.all_gather(g, g)
        comm.all_reduce_region(g)
    torch.cuda.current_stream().wait_stream(side)
    ok = bool(torch.equal(g, want))
    try:
        ver = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception as e:      # noqa: BLE001
        ver = f"unknown ({e})"
    ret[0] = (rep, ok, ver)
    dist.barrier(device_ids=[0])
    dist.destroy_process_group()


def test_rccl_first_contact_world_one():
    """RCCL itself on the test box (one rank is all one GPU allows): process-group construction exactly as bench.py does it
    (backend nccl, device_id, timeout — reference src/train.py:606-610), the collective pre-flight, and the three in-place collectives
    of the ZeRO step on a high-priority side stream.  It cannot measure xGMI; it does catch a library that does not load, a refused
    aliased buffer or a device_id / IPC-mode problem before the driver's first 8-GPU run does."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_rccl_world1_worker, args=(_free_port(), ret), nprocs=1, join=True)
    rep, ok, ver = ret[0]
    assert ok and rep["world"] == 1 and rep["all_reduce"] is True and rep["inplace_reduce_scatter"] and rep["inplace_all_gather"], (rep, ok, ver)


def test_world2_smoke_entry_point():
    """__graft_entry__.smoke() under a 2-rank launcher: pre-flight, two overlapped ZeRO-2 steps, replicas bit-identical."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MOLLY_BENCH_DEVICE="0", MOLLY_DIST_BACKEND="gloo")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(root, "__graft_entry__.py"), "smoke"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "smoke_dist ok: world 2" in r.stdout
