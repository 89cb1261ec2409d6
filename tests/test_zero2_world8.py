"""CPU, world_size 8, gloo: the ZeRO-2 step at the world size the reference trains at (8 ranks of DeepSpeed ZeRO-2:
reference scripts/train/examples/run_train_4B_z2_b1.sh:60-66, src/configs/ds_z2_config.json:18-27).  World 2 cannot expose a
dependence on the reduction order (a two-term bf16 sum has one rounding); eight terms can.  Covered here: replicas bit-identical,
ZeRO-2 == a single-process AdamW to a stated tolerance, the all-to-all reduce-scatter == a rank-ordered fp32 sum bit for bit,
stage 0 against stage 2, a buffer that is NOT a multiple of world * chunk (tail bucket) with >= 7 buckets.
The shard arithmetic is a torch stand-in HERE (test-only); the product's arithmetic is the HIP kernels."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import molly_ref as R
from test_zero2_gloo import TorchKernels

WORLD = 8
N, N_DECAY, CHUNK = 8 * 64 * 7 + 8 * 24, 2900, 64          # 7 full buckets of 8 x 64 + a tail bucket of 8 x 24 elements


class RankOrderKernels(TorchKernels):
    def reduce_rows(self, x2d, out):                         # what molly_reduce_rows_bf16 does: fp32, rank order, one rounding
        acc = x2d[0].float().clone()
        for r in range(1, x2d.shape[0]):
            acc += x2d[r].float()
        out.copy_(acc.to(out.dtype))


def _grads(world, n):
    return [torch.randn(n, generator=torch.Generator().manual_seed(10 + r)).bfloat16() for r in range(world)]


def _worker(rank, world, port, ret, stage, rs_algo, steps):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from molly_amd.trainer.zero2 import Zero2Optimizer
    P = torch.randn(N, generator=torch.Generator().manual_seed(0)).bfloat16()
    grads = _grads(world, N)
    G = grads[rank].clone()
    opt = Zero2Optimizer(P, G, N_DECAY, lr=1e-2, max_grad_norm=1.0, chunk_elems=CHUNK, kernels=RankOrderKernels(), stage=stage,
                         rs_algo=rs_algo)
    assert len(opt.buckets) == 8 and opt.buckets[-1][1] == 24 and all(per == 64 for _, per in opt.buckets[:-1])
    owned = None
    if steps == 0:                                           # only the gradient exchange: what does this rank own afterwards?
        opt.reduce_scatter_grads()
        owned = torch.cat([G[s + rank * per:s + (rank + 1) * per] for s, per in opt.buckets]).clone()
    norm = 0.0
    for _ in range(steps):
        G.copy_(grads[rank])
        norm = float(opt.step())
    ret[rank] = (P.clone(), norm, owned)
    dist.destroy_process_group()


def _run(port, stage=2, rs_algo=None, steps=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(WORLD, port, ret, stage, rs_algo, steps), nprocs=WORLD, join=True)
    return [ret[r] for r in range(WORLD)]


def _single_process(gsum_bf16):
    """AdamW on (sum of rank grads) / world with clip 1.0, oracle arithmetic (oracle/molly_ref.py: clip_coef, adamw_step)."""
    p = torch.randn(N, generator=torch.Generator().manual_seed(0)).bfloat16().float()
    m, v = torch.zeros(N), torch.zeros(N)
    for step in (1, 2):
        gavg = gsum_bf16.float() / WORLD
        total, coef = R.clip_coef([gavg], 1.0)
        gg = gavg * coef
        R.adamw_step(p[:N_DECAY], gg[:N_DECAY], m[:N_DECAY], v[:N_DECAY], step, 1e-2, 1e-2)
        R.adamw_step(p[N_DECAY:], gg[N_DECAY:], m[N_DECAY:], v[N_DECAY:], step, 1e-2, 0.0)
    return p.bfloat16().float(), float(total)


def _check_against_single_process(res, gsum_bf16, lr=1e-2, steps=2):
    ref, total = _single_process(gsum_bf16)
    P0 = res[0].float()
    assert abs(res[1] - total) <= 5e-3 * total
    d = (P0 - ref).abs()
    assert bool((d <= 2 * lr * steps + 2 ** -7 * ref.abs()).all())
    assert (d > 2 ** -7 * ref.abs() + 1e-6).float().mean() < 0.005
    assert (P0 != ref).float().mean() < 0.10


def _fp32_rank_order_sum():
    grads = _grads(WORLD, N)
    acc = grads[0].float().clone()
    for r in range(1, WORLD):
        acc += grads[r].float()
    return acc


def test_zero2_eight_ranks_replicas_identical_and_equal_single_process():
    out = _run(29601)
    for r in range(1, WORLD):
        assert torch.equal(out[0][0], out[r][0]) and out[0][1] == out[r][1]
    # tolerance: the collective adds bf16 partial sums hop by hop (up to 7 roundings of 2^-9 relative each against one for the fp32
    # sum).  AdamW's update is lr * g / (|g| + eps) in its first steps: a gradient whose eight terms cancel to ~0 can come out with
    # the other SIGN, and that element then moves by lr per step the other way — the bound for every element is 2 * lr * steps;
    # all but a fraction of a percent agree to one bf16 ulp of the parameter
    _check_against_single_process(out[0], _fp32_rank_order_sum().bfloat16())


def test_all_to_all_reduce_scatter_is_a_rank_ordered_fp32_sum_bit_for_bit():
    """SURVEY.md §5 option 2 at 8 ranks: every owned gradient element == bf16(fp32 sum over ranks in rank order) exactly —
    an order no collective algorithm can change — tail bucket included; then the whole step against the single process."""
    from molly_amd.trainer.zero2 import Zero2Optimizer
    want = _fp32_rank_order_sum().bfloat16()
    out = _run(29603, rs_algo="a2a", steps=0)
    P = torch.zeros(N, dtype=torch.bfloat16)
    lay = Zero2Optimizer(P, P.clone(), N_DECAY, chunk_elems=CHUNK, kernels=TorchKernels())     # world 1: only for the bucket cut
    buckets = []
    off = 0
    while off < N:
        per = min(CHUNK, (N - off) // WORLD)
        buckets.append((off, per))
        off += per * WORLD
    assert len(buckets) == 8 and lay.n == N
    for r in range(WORLD):
        exp = torch.cat([want[s + r * per:s + (r + 1) * per] for s, per in buckets])
        assert torch.equal(out[r][2], exp), r
    full = _run(29605, rs_algo="a2a", steps=2)
    for r in range(1, WORLD):
        assert torch.equal(full[0][0], full[r][0])
    ref, total = _single_process(want)
    P0 = full[0][0].float()
    # the gradient sum is now exactly the reference's; what is left is the fp32 order of the squared norm (per-shard partial sums)
    assert abs(full[0][1] - total) <= 1e-5 * total
    assert (P0 != ref).float().mean() < 0.01
    assert bool(((P0 - ref).abs() <= 2 ** -7 * ref.abs() + 1e-6).all())


def test_zero0_against_zero2_eight_ranks():
    """Stage 0 all-reduces what stage 2 reduce-scatters.  With eight bf16 terms the two collectives may round in different orders:
    replicas stay bit-identical within each stage, the stages agree to one bf16 ulp of the parameters."""
    z2, z0 = _run(29607, stage=2), _run(29609, stage=0)
    for out in (z2, z0):
        for r in range(1, WORLD):
            assert torch.equal(out[0][0], out[r][0])
    a, b = z2[0][0].float(), z0[0][0].float()
    d = (a - b).abs()
    assert bool((d <= 2 * 1e-2 * 2 + 2 ** -7 * b.abs()).all())              # a sign flip of a cancelling gradient: 2 * lr * steps
    assert (d > 2 ** -7 * b.abs() + 1e-6).float().mean() < 0.005
    assert (a != b).float().mean() < 0.10
    assert abs(z2[0][1] - z0[0][1]) <= 5e-3 * z0[0][1]
