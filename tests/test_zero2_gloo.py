"""CPU, world_size 2, gloo: the ZeRO-2 step's partitioning + collectives (reduce-scatter / all-reduce / all-gather) give
the same parameters on every rank as a single-process AdamW over the averaged gradients.  The shard arithmetic is
injected as a torch stand-in HERE (test-only): the product's arithmetic is the HIP kernels (tests/test_gpu_kernels.py)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import molly_ref as R


class TorchKernels:
    def sqnorm(self, g, out, accumulate):
        s = g.float().pow(2).sum()
        out[0] = out[0] + s if accumulate else s

    def clip_coef(self, norm_sq, max_norm, pre_scale, norm_out, coef_out):
        n = norm_sq[0].sqrt() * pre_scale
        norm_out[0] = n
        coef_out[0] = torch.clamp(max_norm / (n + 1e-6), max=1.0) * pre_scale

    def adamw(self, master, m, v, grad, param_out, lr, b1, b2, eps, wd, step, gscale):
        R.adamw_step(master, grad.float() * gscale[0], m, v, step, lr, wd, b1, b2, eps)
        param_out.copy_(master.to(param_out.dtype))

    def reduce_rows(self, x2d, out):
        out.copy_(x2d.float().sum(0).to(out.dtype))


def _worker(rank, world, port, n, n_decay, chunk, ret, stage=2, staged=False, rs_algo=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from molly_amd.trainer.zero2 import Zero2Optimizer
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(n, generator=g)
    P = p0.clone().bfloat16()
    grads = [torch.randn(n, generator=torch.Generator().manual_seed(10 + r)).bfloat16() for r in range(world)]
    G = grads[rank].clone()
    from molly_amd.trainer.zero2 import _DistComm
    opt = Zero2Optimizer(P, G, n_decay, lr=1e-2, max_grad_norm=1.0, chunk_elems=chunk, kernels=TorchKernels(), stage=stage,
                         comm=_DistComm(staged=True) if staged else None, rs_algo=rs_algo)
    for _ in range(2):
        G.copy_(grads[rank])
        norm = opt.step()
    ret[rank] = (P.clone(), float(norm))
    dist.destroy_process_group()


@pytest.mark.parametrize("chunk", [64, 1 << 20])
def test_zero2_two_ranks_equals_single_process(chunk):
    n, n_decay, world = 1024, 768, 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, 29541 + (chunk % 7), n, n_decay, chunk, ret), nprocs=world, join=True)
    P0, n0 = ret[0]
    P1, n1 = ret[1]
    assert torch.equal(P0, P1) and n0 == n1                      # replicas agree bit for bit after the all-gather
    # single-process reference: AdamW on bf16(sum of rank grads)/world, clip 1.0
    p = torch.randn(n, generator=torch.Generator().manual_seed(0)).bfloat16().float()
    grads = [torch.randn(n, generator=torch.Generator().manual_seed(10 + r)).bfloat16() for r in range(world)]
    gsum = (grads[0].float() + grads[1].float()).bfloat16().float()      # gloo reduces in bf16
    m = torch.zeros(n)
    v = torch.zeros(n)
    for step in (1, 2):
        gavg = gsum / world
        total, coef = R.clip_coef([gavg], 1.0)
        gg = gavg * coef
        R.adamw_step(p[:n_decay], gg[:n_decay], m[:n_decay], v[:n_decay], step, 1e-2, 1e-2)
        R.adamw_step(p[n_decay:], gg[n_decay:], m[n_decay:], v[n_decay:], step, 1e-2, 0.0)
    assert abs(total.item() - n0) < 1e-3 * n0
    # bf16 parameters: equal up to one bf16 ulp (the fp32 masters differ by reduction-order noise only)
    ref = p.bfloat16().float()
    assert bool(((P0.float() - ref).abs() <= 2 ** -7 * ref.abs() + 1e-6).all())
    assert (P0.float() != ref).float().mean() < 0.05


def test_zero0_fallback_equals_zero2():
    """The reference's ds_z0 fallback (examples/run_train_1B_z2_b1.sh:63): all-reduced gradients, replicated
    AdamW.  With two ranks the bf16 sum has one rounding whatever the collective, so the parameters match stage 2 bit for bit."""
    n, n_decay, world = 1024, 768, 2
    mgr = mp.Manager()
    out = {}
    for stage, port in ((2, 29561), (0, 29563)):
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, n, n_decay, 64, ret, stage), nprocs=world, join=True)
        assert torch.equal(ret[0][0], ret[1][0]) and ret[0][1] == ret[1][1]
        out[stage] = ret[0]
    assert torch.equal(out[0][0], out[2][0])
    assert abs(out[0][1] - out[2][1]) <= 1e-6 * out[2][1]      # norm: one pass over the buffer vs per-shard partial sums


def test_zero_stage_of_deepspeed_config(tmp_path):
    from molly_amd.train import zero_stage_of
    assert zero_stage_of(None) == 2
    assert zero_stage_of("src/configs/ds_z0_config.json") == 0          # file absent: the z<N> of the name
    assert zero_stage_of("src/configs/ds_z1_config.json") == 2
    f = tmp_path / "ds.json"
    f.write_text('{"zero_optimization": {"stage": 0}}')
    assert zero_stage_of(str(f)) == 0
    with pytest.raises(ValueError):
        zero_stage_of("src/configs/ds_z3_config.json")


def test_bucket_partition_covers_buffer_once():
    from molly_amd.trainer.zero2 import Zero2Optimizer
    P = torch.zeros(4096, dtype=torch.bfloat16)
    opt = Zero2Optimizer(P, P.clone(), 1000, chunk_elems=1000, kernels=TorchKernels())
    assert sum(per * opt.world for _, per in opt.buckets) == 4096
    assert all(per % 8 == 0 for _, per in opt.buckets)


def _preflight_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from molly_amd.trainer.zero2 import preflight_collectives
    ret[rank] = preflight_collectives(torch.device("cpu"), n_per_rank=64)
    dist.destroy_process_group()


def test_preflight_collectives_two_ranks():
    """bench.py's first contact with the communication backend: the in-place reduce-scatter / all-gather / all-reduce forms
    the ZeRO step uses return their known answers (gloo here; RCCL on the GPU box runs the same function)."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_preflight_worker, args=(2, 29571, ret), nprocs=2, join=True)
    for r in (0, 1):
        rep = ret[r]
        assert rep["world"] == 2 and rep["all_reduce"] is True
        assert rep["inplace_reduce_scatter"] and rep["inplace_all_gather"] and rep["staged"] is False


def test_staged_collectives_equal_inplace():
    """The fallback `preflight_collectives` would select (scratch chunk instead of aliasing) steps to the same parameters,
    bit for bit, as the in-place collectives."""
    mgr = mp.Manager()
    a, b = mgr.dict(), mgr.dict()
    mp.spawn(_worker, args=(2, 29573, 1024, 768, 64, a, 2, True), nprocs=2, join=True)
    mp.spawn(_worker, args=(2, 29575, 1024, 768, 64, b), nprocs=2, join=True)
    assert torch.equal(a[0][0], a[1][0]) and torch.equal(a[0][0], b[0][0]) and a[0][1] == b[0][1]


def test_all_to_all_reduce_scatter_equals_library_reduce_scatter():
    """rs_algo="a2a" (SURVEY.md §5 option 2: all_to_all_single + local fp32 reduction in rank order): with two ranks a bf16 sum
    has one rounding whatever the order, so parameters and norm match the library reduce-scatter bit for bit."""
    mgr = mp.Manager()
    a, b = mgr.dict(), mgr.dict()
    mp.spawn(_worker, args=(2, 29577, 1024, 768, 64, a, 2, False, "a2a"), nprocs=2, join=True)
    mp.spawn(_worker, args=(2, 29579, 1024, 768, 64, b), nprocs=2, join=True)
    assert torch.equal(a[0][0], a[1][0]) and torch.equal(a[0][0], b[0][0]) and a[0][1] == b[0][1]


def test_p2p_request_on_buffers_it_cannot_map_falls_back_to_all_to_all():
    """rs_algo="p2p" where the transport cannot run (CPU buffers here; on a GPU node: a device pair without peer access, expandable
    segments): P2PComm refuses itself on every rank together BEFORE mapping anything, Zero2Optimizer records the reason and exchanges with
    all_to_all + the local fp32 sum instead — the same parameters, bit for bit, as asking for "a2a" (VERDICT r05 item 8b)."""
    mgr = mp.Manager()
    a, b = mgr.dict(), mgr.dict()
    mp.spawn(_worker, args=(2, 29581, 1024, 768, 64, a, 2, False, "p2p"), nprocs=2, join=True)
    mp.spawn(_worker, args=(2, 29583, 1024, 768, 64, b, 2, False, "a2a"), nprocs=2, join=True)
    assert torch.equal(a[0][0], a[1][0]) and torch.equal(a[0][0], b[0][0]) and a[0][1] == b[0][1]


def _fallback_reason_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from molly_amd.trainer.zero2 import Zero2Optimizer
    P = torch.zeros(512).bfloat16()
    opt = Zero2Optimizer(P, P.clone(), 256, chunk_elems=64, kernels=TorchKernels(), rs_algo="p2p")
    ret[rank] = (opt.rs_algo, opt.rs_algo_fallback)
    dist.destroy_process_group()


def test_p2p_fallback_reason_is_recorded_on_every_rank():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_fallback_reason_worker, args=(2, 29585, ret), nprocs=2, join=True)
    for r in (0, 1):
        algo, why = ret[r]
        assert algo == "a2a" and "p2p refused" in why and "not bf16 CUDA tensors" in why


def _sweep_worker(rank, world, port, ret, budget):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from molly_amd.trainer.zero2 import Zero2Optimizer, dist_agree, sweep_exchange
    n = 4096
    P = torch.randn(n, generator=torch.Generator().manual_seed(0)).bfloat16()
    G = torch.randn(n, generator=torch.Generator().manual_seed(10 + rank)).bfloat16()
    state = {"opt": None, "built": [], "measured": []}
    fake_now = [0.0]

    def build(mib, algo):
        state["built"].append(mib)
        if rank == 1 and mib == 3:
            raise MemoryError("out of memory building the 3-unit layout (rank 1 only)")
        state["opt"] = Zero2Optimizer(P, G.clone(), 3072, chunk_elems=int(mib) * 64, kernels=TorchKernels(), rs_algo=algo)

    def measure(mib, algo):
        state["measured"].append(mib)
        state["opt"].step()                                          # real collectives of the candidate's layout
        fake_now[0] += 50.0 if rank == 0 else 10.0                   # rank 0 is the slow one: ITS clock must stop both
        t = torch.tensor([100.0 + mib + rank])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def bcast(obj):
        box = [obj]
        dist.broadcast_object_list(box, src=0)
        return box[0]
    r = sweep_exchange([(m, "rccl") for m in (1, 3, 2, 4, 5)], measure, 1, build=build, agree=dist_agree(torch.device("cpu")),
                       budget_s=budget, broadcast=bcast, clock=lambda: fake_now[0])
    ret[rank] = (r, state["built"], state["measured"])
    dist.destroy_process_group()


def test_sweep_at_world_four_runs_every_candidate():
    """The same sweep at world 4 with no budget and no failing build: every rank builds and measures every candidate in the same order
    (four ranks' in-place reduce-scatters / all-gathers of freshly built layouts back to back) and chooses the same one."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sweep_worker, args=(4, 29589, ret, None), nprocs=4, join=True)
    r0 = ret[0][0]
    for r in range(4):
        assert ret[r][0]["chosen"] == r0["chosen"] and ret[r][2] == [1, 2, 4, 5] and ret[r][1] == [1, 3, 2, 4, 5]
    assert "truncated" not in r0 and r0["ms_per_step"]["3/rccl"] is None


def test_sweep_over_two_ranks_skips_and_stops_together():
    """The pre-warm-up sweep of bench.py --gpus N rehearsed at world 2 on gloo with real optimizers and collectives: a layout that fails to
    BUILD on rank 1 only is skipped by both ranks (neither measures it: no mismatched collectives), and when the slow rank's clock passes the
    budget both stop before the same candidate and keep the best so far; `chosen` is rank 0's."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sweep_worker, args=(2, 29587, ret, 120.0), nprocs=2, join=True)
    (r0, built0, meas0), (r1, built1, meas1) = ret[0], ret[1]
    assert r0 == r1
    assert built0 == built1 == [1, 3, 2, 4] and meas0 == meas1 == [1, 2, 4]     # 3: built (and failed on rank 1), measured by nobody; 5: never started
    assert r0["ms_per_step"]["3/rccl"] is None and "rank 1" in r0["errors"]["3/rccl"] and "out of memory" in r0["errors"]["3/rccl"]
    assert r0["truncated"] is True and r0["not_run"] == ["5/rccl"] and r0["chosen"]["key"] == "1/rccl"
