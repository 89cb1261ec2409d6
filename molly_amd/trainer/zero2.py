"""ZeRO-2-style data-parallel optimizer step on a flat bf16 parameter buffer, RCCL over xGMI.

Replaces DeepSpeed ZeRO stage 2 as the reference configures it (reference: src/configs/ds_z2_config.json:18-27 —
bf16 module, fp32 master + Adam moments sharded 1/DP, bucketed gradient reduce-scatter, global-norm clipping, sharded
AdamW, bucketed all-gather of the updated bf16 parameters; driver ACC:utils/deepspeed.py:264-281) and the HF defaults the
reference inherits (AdamW betas .9/.999 eps 1e-8, weight_decay 1e-2 on matrices only, max_grad_norm 1.0, linear warmup:
src/trainer/omics_trainer.py:53-60).

MI355X layout: the flat buffer is cut into BUCKETS of `world * chunk` contiguous elements; inside every bucket rank r
owns chunk r.  Each bucket is therefore ONE in-place `reduce_scatter_tensor` / `all_gather_into_tensor` on a contiguous
region (no staging copies), and on the fully connected 8-GPU xGMI mesh each of those moves `chunk` elements over each
of the 7 links concurrently.  Default chunk = 16 Mi elements (32 MiB bf16 per link per bucket).
World size 1 skips the collectives and keeps the same code path.

`stage=0` is the reference's ZeRO-0 fallback (what examples/run_train_1B_z2_b1.sh:63 selects for the 1.7B model: plain
data parallelism): the same buckets are ALL-REDUCED in place instead of reduce-scattered, every rank keeps the whole fp32
master / moment set and runs AdamW over the whole buffer, and nothing is all-gathered.  Same hooks, same overlap; with
two ranks the summed gradients are bit-identical to stage 2 (a two-term bf16 sum does not depend on the reduction order);
the gradient norm is one pass over the buffer instead of per-shard partial sums (last fp32 bits).

Overlap (default on for world > 1): collectives run on a dedicated communication stream.
  * reduce-scatter: the backward reports "gradients in [lo, hi) are final" layer by layer (engine hook); every bucket that
    is fully covered is reduce-scattered immediately, beside the backward of the earlier layers.  The reference's
    DeepSpeed config has overlap_comm:false; overlapping changes no value (each bucket's reduction is independent).
  * all-gather: launched bucket by bucket right after the shard AdamW (no-decay/gain region first, then in flat order =
    the order the next forward consumes parameters); the forward waits per bucket (`wait_params`) just before first use.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from ..tracing import roctx


def linear_warmup_lr(step: int, base_lr: float, warmup: int, total: int) -> float:
    """HF get_linear_schedule_with_warmup; `step` = scheduler steps already taken (0 for the first optimizer step)."""
    if step < warmup:
        return base_lr * step / max(1, warmup)
    return base_lr * max(0.0, (total - step) / max(1, total - warmup))


class _DistComm:
    """torch.distributed collectives (RCCL on ROCm); in-place on contiguous bucket regions (NCCL/RCCL define both in-place
    forms: reduce-scatter with recvbuff == sendbuff + rank*recvcount, all-gather with sendbuff == recvbuff + rank*sendcount).
    `staged=True` goes through a scratch chunk instead (one extra copy per bucket) — selected by `preflight_collectives`
    if the in-place forms ever returned a wrong answer on the installed backend."""

    def __init__(self, group=None, staged: bool = False, rs_algo: str = "rccl", reduce_rows=None):
        """rs_algo: "rccl" = the library's reduce_scatter (ring or tree, its choice) | "a2a" = SURVEY.md §5 option 2:
        `all_to_all_single` (rank r's chunk j travels straight to rank j: on the fully connected xGMI mesh all 7 links of every
        GPU carry one chunk at once, whatever algorithm the library would have picked for a reduce-scatter) + a LOCAL fp32
        reduction of the `world` received copies in rank order (reduce_rows: the HIP kernel, or a torch stand-in in the CPU
        tests) — one rounding per element, an order no collective algorithm can change (SURVEY.md §7.3 'ZeRO-2 result parity')."""
        assert rs_algo in ("rccl", "a2a")
        self.group = group
        self.staged = staged
        self.rs_algo = rs_algo
        self.reduce_rows = reduce_rows
        self._scratch = None
        self._recv = None

    def _tmp(self, like):
        if self._scratch is None or self._scratch.numel() < like.numel() or self._scratch.dtype != like.dtype:
            self._scratch = torch.empty(like.numel(), dtype=like.dtype, device=like.device)
        return self._scratch[:like.numel()]

    def reduce_scatter(self, out_chunk, region):
        if self.rs_algo == "a2a":
            if self._recv is None or self._recv.numel() < region.numel() or self._recv.dtype != region.dtype:
                self._recv = torch.empty(region.numel(), dtype=region.dtype, device=region.device)
            recv = self._recv[:region.numel()]
            dist.all_to_all_single(recv, region, group=self.group)           # equal splits: chunk j -> rank j
            world = region.numel() // out_chunk.numel()
            self.reduce_rows(recv.view(world, out_chunk.numel()), out_chunk)
            return
        if self.staged:
            tmp = self._tmp(out_chunk)
            dist.reduce_scatter_tensor(tmp, region, group=self.group)
            out_chunk.copy_(tmp)
            return
        dist.reduce_scatter_tensor(out_chunk, region, group=self.group)

    def all_gather(self, region, chunk):
        if self.staged:
            tmp = self._tmp(chunk)
            tmp.copy_(chunk)
            dist.all_gather_into_tensor(region, tmp, group=self.group)
            return
        dist.all_gather_into_tensor(region, chunk, group=self.group)

    def all_reduce(self, t):
        dist.all_reduce(t, group=self.group)

    def all_reduce_region(self, region):
        dist.all_reduce(region, group=self.group)


class _NullComm:
    """MEASUREMENT ONLY (bench.py at N > 1, after the timed region): the same step with no bytes exchanged — every collective a
    no-op, so each rank keeps its own gradients and the replicas diverge from here on.  `step time with the exchange overlapped` minus
    `step time without any exchange` is the communication the overlap failed to hide (`comm.exposed_comm_ms`)."""
    rs_algo = "none"
    staged = False

    def reduce_scatter(self, out_chunk, region):
        pass

    def all_gather(self, region, chunk):
        pass

    def all_reduce(self, t):
        pass

    def all_reduce_region(self, region):
        pass


def sweep_exchange(candidates, measure, steps_each: int, build=None, agree=None, budget_s: Optional[float] = None, broadcast=None,
                   clock=None):
    """Which (bucket size, reduce-scatter algorithm) runs this job's step fastest on this job's own ranks?  The reference fixes
    `reduce_bucket_size` / `allgather_bucket_size` at 5e8 elements in a config file (src/configs/ds_z2_config.json:18-27); here the
    first multi-GPU run measures its own.

    Per candidate: `build(bucket_mib, rs_algo)` (optional) constructs the optimizer with that layout — the step that can fail on ONE
    rank (out of memory at 1 GiB buckets on the fullest rank, an IPC open the driver refuses) — then `agree(ok, err) -> (ok_all, err)`
    makes the verdict COLLECTIVE (bench.py: an all-reduce MIN + the first failing rank's message): if any rank failed, every rank
    records the error and skips the candidate together, so no rank ever runs a layout the others do not (a rank that moved
    on alone met the others inside the previous candidate's collectives — a hang until the 30-minute timeout, or wrong-sized reductions).
    `measure(bucket_mib, rs_algo)` then runs one settling step + `steps_each` timed ones and returns the step time in ms ALREADY
    maximised over the ranks; an exception out of it is NOT caught: collectives are in flight by then, and the only safe thing is to
    abort the job (bench.py prints its one failure line and exits non-zero).  Without `build`, `measure` does both and its exceptions are
    recorded and skipped as before (single process, the CPU tests).

    `budget_s`: wall budget of the whole sweep; before each candidate every rank asks `agree(not over_budget)` — one rank over its budget
    stops all of them — and the best so far is kept: `truncated` / `not_run` in the result say so.  `broadcast(obj) -> obj` (bench.py:
    rank 0's) makes `chosen` the same object everywhere even if the table's floats differed in the last bit.

    Returns {"ms_per_step": {"<MiB>/<algo>": ms | None}, "errors": {...}, "chosen": {"bucket_mib": ..., "rs_algo": ...}, "steps_each": n
    [, "truncated": True, "not_run": [...], "budget_s": ...]}; ties go to the earlier candidate (smaller bucket first)."""
    import time
    clock = clock or time.monotonic
    agree = agree or (lambda ok, err=None: (ok, err))
    t0 = clock()
    table, errors, not_run = {}, {}, []
    best_key, best_ms, best = None, None, None
    candidates = list(candidates)
    for ci, (mib, algo) in enumerate(candidates):
        key = f"{mib:g}/{algo}"
        if budget_s is not None and ci > 0:
            go, _ = agree(clock() - t0 <= budget_s, None)
            if not go:
                not_run = [f"{m:g}/{a}" for m, a in candidates[ci:]]
                break
        if build is not None:
            err = None
            try:
                build(mib, algo)
            except Exception as e:                                # noqa: BLE001 — made collective below
                err = f"{type(e).__name__}: {str(e)[:200]}"
            ok, err_all = agree(err is None, err)
            if not ok:
                table[key] = None
                errors[key] = err_all or err or "a rank failed to build this layout"
                continue
            ms = float(measure(mib, algo))                        # collectives in flight: a failure here aborts (see above)
        else:
            try:
                ms = float(measure(mib, algo))
            except Exception as e:                                # noqa: BLE001 — recorded, the sweep goes on
                table[key] = None
                errors[key] = f"{type(e).__name__}: {str(e)[:200]}"
                continue
        table[key] = round(ms, 2)
        if best_ms is None or ms < best_ms:
            best_key, best_ms, best = key, ms, (mib, algo)
    if best is None:
        raise RuntimeError(f"bucket sweep: every candidate failed: {errors}")
    chosen = {"bucket_mib": best[0], "rs_algo": best[1], "key": best_key}
    if broadcast is not None:
        chosen = broadcast(chosen)
    out = {"ms_per_step": table, "chosen": chosen, "steps_each": steps_each}
    if errors:
        out["errors"] = errors
    if not_run:
        out["truncated"] = True
        out["not_run"] = not_run
        out["budget_s"] = budget_s
    return out


def dist_agree(device, group=None):
    """-> agree(ok, err) for `sweep_exchange` over torch.distributed: MIN over the ranks of `ok`, and the message of the lowest failing rank."""
    def agree(ok, err=None):
        flag = torch.tensor([1.0 if ok else 0.0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if flag.item() > 0:
            return True, None
        errs = [None] * dist.get_world_size(group)
        dist.all_gather_object(errs, err, group=group)
        first = next(((r, e) for r, e in enumerate(errs) if e), None)
        return False, (f"rank {first[0]}: {first[1]}" if first else None)
    return agree


_STAGED_DEFAULT = False        # set by preflight_collectives when the in-place forms misbehave on this backend


def preflight_collectives(device, group=None, n_per_rank: int = 4096) -> dict:
    """First contact with the communication backend, before anything expensive is built: the three collectives of the step
    (in-place reduce-scatter, in-place all-gather, all-reduce) on small integer-valued bf16 buffers with known answers.
    A wrong in-place result switches every later `_DistComm` to its staged form (and says so); a wrong staged result raises.
    Returns a small report for the bench line."""
    global _STAGED_DEFAULT
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    rep = {"world": world, "inplace_reduce_scatter": None, "inplace_all_gather": None, "all_reduce": None}

    def fresh():
        # element j of rank r's buffer = (j % 8) + r: sums over ranks are small integers, exact in bf16
        base = (torch.arange(n_per_rank * world, device=device) % 8).to(torch.bfloat16)
        return base + rank
    want_sum = (torch.arange(n_per_rank * world, device=device) % 8).float() * world + world * (world - 1) / 2
    for staged in (False, True):
        comm = _DistComm(group, staged=staged)
        buf = fresh()
        mine = buf[rank * n_per_rank:(rank + 1) * n_per_rank]
        try:                                    # a backend may also REFUSE the aliased form outright: same verdict as a wrong sum
            comm.reduce_scatter(mine, buf)
            ok_rs = bool(torch.equal(mine.float(), want_sum[rank * n_per_rank:(rank + 1) * n_per_rank]))
        except (RuntimeError, ValueError) as e:
            if staged:
                raise
            rep["inplace_error"] = str(e)[:200]
            ok_rs = False
        buf2 = torch.zeros(n_per_rank * world, dtype=torch.bfloat16, device=device)
        buf2[rank * n_per_rank:(rank + 1) * n_per_rank] = rank + 1
        want_ag = (torch.arange(n_per_rank * world, device=device) // n_per_rank + 1).float()
        try:
            comm.all_gather(buf2, buf2[rank * n_per_rank:(rank + 1) * n_per_rank])
            ok_ag = bool(torch.equal(buf2.float(), want_ag))
        except (RuntimeError, ValueError) as e:
            if staged:
                raise
            rep["inplace_error"] = str(e)[:200]
            ok_ag = False
        flags = torch.tensor([float(ok_rs), float(ok_ag)], device=device)
        dist.all_reduce(flags, op=dist.ReduceOp.MIN, group=group)              # every rank takes the same decision
        ok_rs, ok_ag = bool(flags[0] > 0), bool(flags[1] > 0)
        if not staged:
            rep["inplace_reduce_scatter"], rep["inplace_all_gather"] = ok_rs, ok_ag
        if ok_rs and ok_ag:
            _STAGED_DEFAULT = staged
            break
        if staged:
            raise RuntimeError(f"collectives return wrong results on this backend (staged form too): rs={ok_rs} ag={ok_ag}")
    t = fresh()
    dist.all_reduce(t, group=group)
    rep["all_reduce"] = bool(torch.equal(t.float(), want_sum))
    if not rep["all_reduce"]:
        raise RuntimeError("all_reduce returned a wrong sum on this backend")
    rep["staged"] = _STAGED_DEFAULT
    return rep


class _HipKernels:
    """The shard arithmetic: HIP kernels through the C ABI (no CPU fallback in the product)."""

    def __init__(self, device):
        from .. import ops
        self.ops = ops
        self.ws = torch.empty(ops.lib().query("molly_sqnorm_blocks"), dtype=torch.float32, device=device)

    def sqnorm(self, g, out, accumulate):
        self.ops.sqnorm(g, out, self.ws, accumulate=accumulate)

    def clip_coef(self, norm_sq, max_norm, pre_scale, norm_out, coef_out):
        # coef_out is scal[2:3]; the slot behind it (scal[3]) counts skipped (non-finite-norm) steps
        self.skipped = coef_out.as_strided((1,), (1,), coef_out.storage_offset() + 1)
        self.ops.clip_coef(norm_sq, max_norm, pre_scale, norm_out, coef_out, self.skipped)

    def adamw(self, master, m, v, grad, param_out, lr, b1, b2, eps, wd, step, gscale):
        self.ops.adamw_step(master, m, v, grad, param_out, lr, b1, b2, eps, wd, step, gscale, getattr(self, "skipped", None))

    def reduce_rows(self, x2d, out):
        self.ops.reduce_rows(x2d, out)


class Zero2Optimizer:
    def __init__(self, flat_params: torch.Tensor, flat_grads: torch.Tensor, n_decay: int, lr: float = 3e-5,
                 betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2, max_grad_norm: float = 1.0,
                 group=None, chunk_elems: int = 16 * 1024 * 1024, kernels=None, overlap: Optional[bool] = None,
                 comm=None, stage: int = 2, rs_algo: Optional[str] = None):
        assert stage in (0, 2), "stage: 2 = sharded optimizer state (ZeRO-2), 0 = replicated (plain DP all-reduce)"
        self.stage = stage
        self.P, self.G = flat_params, flat_grads
        self.n = flat_params.numel()
        self.n_decay = n_decay
        self.lr, self.betas, self.eps, self.wd, self.max_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.group = group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        assert self.n % (8 * self.world) == 0, "flat buffer must be padded to a multiple of 8*world elements"
        self.chunk = min(chunk_elems, self.n // self.world)
        self.chunk -= self.chunk % 8
        self.bucket = self.chunk * self.world
        # buckets: [(start, elems_per_rank)]; the tail bucket is smaller but still divisible by 8*world
        self.buckets: List[Tuple[int, int]] = []
        off = 0
        while off < self.n:
            per = min(self.chunk, (self.n - off) // self.world)
            self.buckets.append((off, per))
            off += per * self.world
        assert off == self.n
        self.owned = sum(per for _, per in self.buckets) if stage == 2 else self.n
        dev = flat_params.device
        # fp32 master / moments for the owned chunks, packed (stage 0: the whole buffer, in flat order)
        self.master = torch.empty(self.owned, dtype=torch.float32, device=dev)
        if stage == 0:
            self.master.copy_(flat_params.float())
        pos = 0
        for start, per in (self.buckets if stage == 2 else ()):
            lo = start + self.rank * per
            self.master[pos:pos + per].copy_(flat_params[lo:lo + per].float())
            pos += per
        self.m = torch.zeros_like(self.master)
        self.v = torch.zeros_like(self.master)
        self.scal = torch.zeros(4, dtype=torch.float32, device=dev)       # [0]=norm^2 [1]=norm [2]=coef [3]=skipped steps
        self.k = kernels if kernels is not None else _HipKernels(dev)
        self.t = 0
        # ---- overlap machinery -------------------------------------------------------------------------------
        rs_algo = rs_algo or os.environ.get("MOLLY_RS_ALGO", "rccl")
        self.rs_algo_fallback = None            # why a requested transport was not used (recorded in bench.py's comm line)
        if comm is None and rs_algo == "p2p" and self.world > 1:
            # direct peer exchange over mapped peer memory (trainer/p2p.py; SURVEY.md 5 option 3): no collective for the buckets at all.
            # The transport refuses itself — on every rank together, before anything is mapped — where the devices cannot reach each
            # other's memory or the allocator's IPC handles would not cover the buffers: all-to-all then (the same arithmetic, bit for bit)
            assert stage == 2, "rs_algo='p2p' is the ZeRO-2 exchange"
            from .p2p import P2PComm, P2PUnavailable
            try:
                comm = P2PComm(flat_grads, flat_params, len(self.buckets), group)
            except P2PUnavailable as e:
                self.rs_algo_fallback = f"p2p refused ({e}); using a2a"
                rs_algo = "a2a"
        self.comm = comm if comm is not None else _DistComm(group, staged=_STAGED_DEFAULT, rs_algo=rs_algo,
                                                            reduce_rows=getattr(self.k, "reduce_rows", None))
        self.rs_algo = getattr(self.comm, "rs_algo", "rccl")
        self.overlap = (self.world > 1) if overlap is None else overlap
        self.overlap = self.overlap and flat_params.is_cuda
        self.P_out = flat_params                # AdamW writes here; the all-gather publishes into P (same buffer in production)
        self.hooked = False                     # set by OmicsOne.attach_optimizer: somebody will call wait_params()
        # per-bucket timing of the exchange (bench.py at N > 1): [(kind, bucket, start event, end event)] on the stream the
        # collective ran on; None = off (two events per collective are not free)
        self.comm_events = None
        # one rank, hooked: nothing to exchange, but the AdamW pass itself (HBM-bound, 28 B per parameter) can run on a side
        # stream under the NEXT step's first layers (MFMA-bound), bucket by bucket in the order the forward consumes
        # parameters; the forward's wait_params() calls then wait for AdamW events exactly as they wait for all-gather events
        self.async_update = (self.world == 1 or stage == 0) and comm is None and flat_params.is_cuda and kernels is None
        self._async_armed = False
        self.ustream = None
        if self.overlap:
            if self.world > 1 and flat_params.is_cuda:
                # RCCL's collective kernels hold CUs for milliseconds.  The persistent GEMM launches exactly one block per
                # CU, each owning 1/256 of the tiles: with a few CUs taken, the blocks that cannot start wait for a whole
                # share to finish and the launch takes twice as long (-40 % with 16 CUs held; measured on one GPU with a
                # stand-in that holds CUs: tools/diag/gemm_beside_hog.py).  Launch shapes for N > 1,
                # MOLLY_GEMM_PERSISTENT_MULTI = -t static blocks of at most t tiles placed by the hardware dispatcher (default -3,
                # round 2's shape: plain launches, nothing a block waits for) | dyn 256 resident blocks that DRAW their tiles
                # (gemm.hip 'DYN': best beside the stand-in, -13 % / -24 % with 16 / 64 CUs held, but one exposed ticket per
                # launch and — like every shape here — never yet run beside real RCCL kernels, so it is opt-in: ADVICE r03) |
                # 0 one block per tile | 256 (or 1) static, one block per CU.  bench.py at N > 1 measures the shapes against
                # each other on the job's own ranks before its warm-up and keeps the fastest (`comm.gemm_mode_ab`), so the
                # first multi-GPU run records the measurement this default is waiting for.
                # The optimizer only RECORDS the wish: `OmicsOne.attach_optimizer` applies it to the GEMM context of the model
                # this optimizer steps — no other model, evaluator or later test in the process inherits it.
                mode = os.environ.get("MOLLY_GEMM_PERSISTENT_MULTI", "-3")
                self.gemm_blocks_mode = "dyn" if mode == "dyn" else 256 if mode == "1" else int(mode)
            self.cstream = torch.cuda.Stream(device=dev, priority=-1)     # collectives first whenever CUs free up
            self._rs_done = [False] * len(self.buckets)
            self._ag_events = [None] * len(self.buckets)
            self._ag_waited = [True] * len(self.buckets)

    def set_overlap(self, on: bool):
        """Switch between the overlapped exchange (collectives on the communication stream) and the synchronous one (the
        reference's overlap_comm:false) between two steps — bench.py measures the exposed communication time this way."""
        if self.world == 1 or not self.P.is_cuda:
            return
        self.wait_all_params()
        torch.cuda.current_stream().wait_stream(self.cstream) if hasattr(self, "cstream") else None
        torch.cuda.synchronize()
        if on and not hasattr(self, "cstream"):
            self.cstream = torch.cuda.Stream(device=self.P.device, priority=-1)
        self._rs_done = [False] * len(self.buckets)
        self._ag_events = [None] * len(self.buckets)
        self._ag_waited = [True] * len(self.buckets)
        self.overlap = bool(on)

    # ---- hooks the engines call -------------------------------------------------------------------------------
    def on_grads_final(self, lo: int, hi: int):
        """Gradients with flat offsets in [lo, hi) will not change any more in this optimizer step."""
        if not self.overlap:
            return
        ev = None
        for b, (start, per) in enumerate(self.buckets):
            if self._rs_done[b] or start < lo or start + per * self.world > hi:
                continue
            if ev is None:
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                self.cstream.wait_event(ev)
            with torch.cuda.stream(self.cstream):
                self._reduce_bucket(start, per)
            self._rs_done[b] = True

    def wait_params(self, lo: int, hi: int):
        """Make the current stream wait until parameters with flat offsets in [lo, hi) have been all-gathered (or, on one
        rank, updated by the side-stream AdamW)."""
        if not (self.overlap or self._async_armed):
            return
        # The events were recorded on ONE stream in the order [last bucket, 0, 1, ...] (all_gather_params / _update_on_side_stream): the one
        # recorded latest among those wanted implies all the earlier ones, so ONE cross-stream wait per call instead of one per bucket
        # (a decoder layer spans ~3 buckets; each wait is a barrier packet the launch stream drains for: 17 us per layer in the forward)
        last = len(self.buckets) - 1
        best, best_pos = -1, -1
        for b, (start, per) in enumerate(self.buckets):
            if self._ag_waited[b] or start >= hi or start + per * self.world <= lo:
                continue
            pos = 0 if b == last else b + 1
            if pos > best_pos:
                best, best_pos = b, pos
        if best < 0:
            return
        torch.cuda.current_stream().wait_event(self._ag_events[best])
        if best_pos >= 0:
            self._ag_waited[last] = True
        for b in range(min(best_pos, last)):               # positions 1 .. best_pos are buckets 0 .. best_pos - 1
            self._ag_waited[b] = True

    def wait_all_params(self):
        self.wait_params(0, self.n)

    def _timed(self, kind: str, start: int):
        """context manager: HIP events on the current stream around one bucket's collective when comm_events is on."""
        opt = self

        class _T:
            def __enter__(self_t):
                if opt.comm_events is not None and opt.P.is_cuda:
                    self_t.e0 = torch.cuda.Event(enable_timing=True)
                    self_t.e0.record(torch.cuda.current_stream())

            def __exit__(self_t, *exc):
                if opt.comm_events is not None and opt.P.is_cuda:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record(torch.cuda.current_stream())
                    opt.comm_events.append((kind, start, self_t.e0, e1))
                return False
        return _T()

    def comm_timings(self) -> dict:
        """{kind: {"n": collectives timed, "us_p50": ..., "us_max": ..., "us_per_bucket": [first step's buckets in launch order]}}
        from the events gathered since `comm_events = []` (synchronises).  With the exchange overlapped a collective's time
        includes what it waited for CUs beside the backward — that is the number the bucket size is tuned against."""
        if not self.comm_events:
            return {}
        torch.cuda.synchronize()
        import statistics
        out = {}
        for kind in sorted({k for k, *_ in self.comm_events}):
            ev = [(b, e0.elapsed_time(e1) * 1e3) for k, b, e0, e1 in self.comm_events if k == kind]
            us = [t for _, t in ev]
            nb = len({b for b, _ in ev})
            out[kind] = {"n": len(us), "us_p50": round(statistics.median(us), 1), "us_max": round(max(us), 1),
                         "us_per_bucket": [round(t, 1) for _, t in ev[:nb]]}
        return out

    def set_gemm_blocks_mode(self, mode, model=None):
        """Change the GEMM launch shape this optimizer asks for (see __init__) and, given the model it steps, apply it."""
        self.gemm_blocks_mode = "dyn" if mode == "dyn" else int(mode)
        if model is not None:
            model.attach_optimizer(self)

    def _reduce_bucket(self, start: int, per: int):
        region = self.G[start:start + per * self.world]
        with self._timed("all_reduce" if self.stage == 0 else "reduce_scatter", start):
            if self.stage == 0:
                self.comm.all_reduce_region(region)
            else:
                self.comm.reduce_scatter(region[self.rank * per:(self.rank + 1) * per], region)

    # ---- pieces (also used by the multi-process CPU tests) ---------------------------------------------------
    def reduce_scatter_grads(self):
        if self.overlap:
            self.on_grads_final(0, self.n)                     # whatever the backward did not cover yet
            torch.cuda.current_stream().wait_stream(self.cstream)
            self._rs_done = [False] * len(self.buckets)
            return
        if self.world == 1:
            return
        for start, per in self.buckets:
            self._reduce_bucket(start, per)

    def all_gather_params(self):
        if self.stage == 0:
            return                                             # every rank updated every parameter itself
        if self.overlap:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.cstream.wait_event(ev)
            # gains/biases (tail of the flat buffer) are needed by the very first kernel of the next forward: publish first
            order = [len(self.buckets) - 1] + list(range(len(self.buckets) - 1))
            with torch.cuda.stream(self.cstream):
                for b in order:
                    start, per = self.buckets[b]
                    with self._timed("all_gather", start):
                        self.comm.all_gather(self.P[start:start + per * self.world],
                                             self.P_out[start + self.rank * per:start + (self.rank + 1) * per])
                    e = torch.cuda.Event()
                    e.record(self.cstream)
                    self._ag_events[b] = e
                    self._ag_waited[b] = False
            if not self.hooked:                                # nobody will wait per bucket: stay correct, lose the overlap
                self.wait_all_params()
            return
        if self.world == 1:
            return
        for start, per in self.buckets:
            region = self.P[start:start + per * self.world]
            with self._timed("all_gather", start):
                self.comm.all_gather(region, region[self.rank * per:(self.rank + 1) * per])

    def grad_norm_and_clip(self):
        first = True
        sharded = self.world > 1 and self.stage == 2
        for start, per in (self.buckets if sharded else [(0, self.n)]):     # whole buffer here: one pass
            lo = start + self.rank * per if sharded else 0
            self.k.sqnorm(self.G[lo:lo + per], self.scal[0:1], accumulate=not first)
            first = False
        if sharded:
            self.comm.all_reduce(self.scal[0:1])
        # gradients were SUMMED over ranks; DeepSpeed averages them: fold 1/world into the scale
        self.k.clip_coef(self.scal[0:1], self.max_norm, 1.0 / self.world, self.scal[1:2], self.scal[2:3])

    def step(self, lr: Optional[float] = None):
        """reduce-scatter -> global-norm clip -> AdamW on the owned chunks -> all-gather.  Returns the (device) grad norm."""
        self.t += 1
        lr = self.lr if lr is None else lr
        if hasattr(self.comm, "check"):
            self.comm.check()                   # direct peer exchange: a wait of an earlier step gave up (plain read of a pinned word, no sync)
        if self.ustream is not None:
            # the previous step's side-stream AdamW reads the clip coefficient this step is about to overwrite; in the training
            # loop it finished long ago (the backward waited for every parameter), this only orders back-to-back step() calls
            torch.cuda.current_stream().wait_stream(self.ustream)
        with roctx("molly: gradient reduce-scatter (tail)"):
            self.reduce_scatter_grads()
        with roctx("molly: grad norm + clip"):
            self.grad_norm_and_clip()
        with roctx("molly: AdamW + all-gather"):
            return self._update_and_publish(lr)

    def _update_and_publish(self, lr: float):
        pos = 0
        whole = self.world == 1 or self.stage == 0               # this rank updates the whole buffer
        if whole and self.async_update and self.hooked:
            self._update_on_side_stream(lr)
            return self.scal[1]
        if whole:
            # one rank owns everything in flat order: one launch per decay class instead of one per bucket (elementwise, so
            # bit-identical to the bucketed launches; the bucket structure only exists to pipeline the exchange)
            for a, b, wd in ((0, self.n_decay, self.wd), (self.n_decay, self.n, 0.0)):
                if b > a:
                    self.k.adamw(self.master[a:b], self.m[a:b], self.v[a:b], self.G[a:b], self.P_out[a:b], lr,
                                 self.betas[0], self.betas[1], self.eps, wd, self.t, self.scal[2:3])
        for start, per in (self.buckets if not whole else ()):
            lo, hi = start + self.rank * per, start + (self.rank + 1) * per
            # split at the decay / no-decay boundary of the flat layout
            for a, b, wd in ((lo, min(hi, self.n_decay), self.wd), (max(lo, self.n_decay), hi, 0.0)):
                if b > a:
                    s = pos + (a - lo)
                    self.k.adamw(self.master[s:s + b - a], self.m[s:s + b - a], self.v[s:s + b - a], self.G[a:b],
                                 self.P_out[a:b], lr, self.betas[0], self.betas[1], self.eps, wd, self.t, self.scal[2:3])
            pos += per
        self.all_gather_params()
        return self.scal[1]

    def _update_on_side_stream(self, lr: float):
        if self.ustream is None:
            self.ustream = torch.cuda.Stream(device=self.P.device)
            self._ag_events = [None] * len(self.buckets)
            self._ag_waited = [True] * len(self.buckets)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())                       # clip coefficient and final gradients are ready
        self.ustream.wait_event(ev)
        # gains/biases (tail of the flat buffer) are read by the very first kernel of the next forward: update them first
        order = [len(self.buckets) - 1] + list(range(len(self.buckets) - 1))
        with torch.cuda.stream(self.ustream):
            for b in order:
                lo, per = self.buckets[b]
                hi = lo + per * self.world
                for a, e, wd in ((lo, min(hi, self.n_decay), self.wd), (max(lo, self.n_decay), hi, 0.0)):
                    if e > a:
                        self.k.adamw(self.master[a:e], self.m[a:e], self.v[a:e], self.G[a:e], self.P_out[a:e], lr,
                                     self.betas[0], self.betas[1], self.eps, wd, self.t, self.scal[2:3])
                done = torch.cuda.Event()
                done.record(self.ustream)
                self._ag_events[b] = done
                self._ag_waited[b] = False
        self._async_armed = True

    def skipped_steps(self) -> int:
        """Optimizer steps skipped because the gradient norm was not finite (host sync: call when logging).  With the direct peer exchange a
        skipped step may be a peer that never arrived: that raises here (and at the transport's next call) instead of being counted."""
        n = int(self.scal[3].item())
        if hasattr(self.comm, "check"):
            self.comm.check()
        return n

    @torch.no_grad()
    def refresh_master(self):
        """Re-read the fp32 masters from the bf16 parameters (after a checkpoint was loaded into them); moments are kept."""
        if self.stage == 0:
            self.master.copy_(self.P.float())
            return
        pos = 0
        for start, per in self.buckets:
            lo = start + self.rank * per
            self.master[pos:pos + per].copy_(self.P[lo:lo + per].float())
            pos += per

    def comm_bytes_per_step(self) -> int:
        """bytes each rank sends (= receives) per optimizer step: RS + AG of bf16, (world-1)/world of the buffer each."""
        # (stage 0: one all-reduce = the same two passes over the ring)
        return 0 if self.world == 1 else 2 * 2 * self.n * (self.world - 1) // self.world
