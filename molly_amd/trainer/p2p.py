"""Direct peer exchange for the ZeRO-2 step: the third reduce-scatter / all-gather transport of `Zero2Optimizer` (SURVEY.md §5 option 3;
reference role: DeepSpeed ZeRO-2's collectives, src/configs/ds_z2_config.json:18-27 behind deepspeed.initialize, src/train.py:606-614).

`rs_algo="rccl"` asks the library for a reduce-scatter (ring or tree, its choice), `"a2a"` moves every chunk over its own xGMI link with
`all_to_all_single` into a receive buffer and sums locally.  Here nothing is sent: every rank maps the peers' flat gradient and
parameter buffers into its own address space (the runtime's IPC, through torch's CUDA-IPC handles), the OWNER of a chunk reads the
seven other copies in place over the seven links at once and adds them in fp32 in rank order (`molly_p2p_reduce_bf16` — the same
arithmetic, bit for bit, as the all-to-all variant), and after the AdamW shard step writes its parameter chunk straight into the
peers' parameter buffers (`molly_p2p_push_bf16`).  No receive buffer, no second pass, no collective kernel holding CUs.

Hand-shake (csrc/p2p.hip): per bucket two flags per rank that only ever grow — "my gradients of bucket b are final for exchange #s",
"my parameter chunk of bucket b has arrived everywhere for exchange #s" — set by a one-thread kernel behind the producing kernels,
awaited by a bounded spin kernel in front of the consuming ones.  Why the buffers may be reused without a third flag: a rank runs
AdamW only after the all-reduced gradient norm, i.e. after EVERY rank has finished every reduce (nobody still reads gradients when
the next backward overwrites them), and parameters are pushed only after AdamW, i.e. after every rank has finished its backward
(nobody still reads the old parameters).

A wait that gives up is FATAL for the step (round 6): the error word lives in pinned host memory the kernels write at system scope;
`molly_p2p_reduce_bf16` reads it and writes NaN instead of sums, so the all-reduced gradient norm is non-finite on EVERY rank and the
optimizer step is skipped everywhere (no replica ever applies a stale term), and the host reads the word — a plain memory read, no
synchronisation — at every later call and raises `P2PTimeout`.  The bound is wall-clock (`timeout_s`, default 600 s: a checkpoint save
or a data-loader stall on one rank must not trip it).

The transport REFUSES ITSELF before any buffer is mapped (`P2PUnavailable`, every rank takes the same decision) when some pair of
devices cannot access each other (`hipDeviceCanAccessPeer`), when the allocator uses expandable segments (their IPC handles do not
cover a whole tensor), or when a buffer is not a CUDA tensor; `Zero2Optimizer` then falls back to `rs_algo="a2a"` and records why.

State: validated on one GPU with two and four processes, bit-identical to rs_algo="a2a" (tests/test_gpu_two_ranks.py).  It has never
run over links; nothing is claimed about its speed, and `bench.py` does not select it."""
from __future__ import annotations

import ctypes
from typing import List

import torch
import torch.distributed as dist

from .. import ops


def _share(t: torch.Tensor):
    """A picklable description of a CUDA tensor another process can open (torch's own CUDA-IPC: the allocation's IPC handle + offset)."""
    from torch.multiprocessing.reductions import reduce_tensor
    return reduce_tensor(t)


def _open(desc) -> torch.Tensor:
    fn, args = desc
    return fn(*args)


class P2PUnavailable(RuntimeError):
    """Direct peer exchange cannot run in this job (the reason is the message); nothing has been mapped — fall back to a collective."""


class P2PTimeout(RuntimeError):
    """A peer did not raise its flag within the deadline: the step that saw it was skipped on every rank (NaN-poisoned sums)."""


def p2p_refusal(device_of_rank, my_rank: int, can_access=None, alloc_conf: str = None):
    """Why this rank cannot take part in a direct peer exchange, or None.  device_of_rank: every rank's device index (ranks of one node);
    can_access(a, b): hipDeviceCanAccessPeer (default torch.cuda.can_device_access_peer).  Pure: the CPU tests call it with stand-ins."""
    import os
    conf = alloc_conf if alloc_conf is not None else (os.environ.get("PYTORCH_HIP_ALLOC_CONF", "") + "," + os.environ.get("PYTORCH_CUDA_ALLOC_CONF", ""))
    if "expandable_segments:true" in conf.replace(" ", "").lower():
        return "the caching allocator uses expandable segments (an IPC handle does not cover a whole tensor there)"
    if can_access is None:
        can_access = torch.cuda.can_device_access_peer
    mine = device_of_rank[my_rank]
    for r, d in enumerate(device_of_rank):
        if r != my_rank and d != mine and not can_access(mine, d):
            return f"device {mine} (rank {my_rank}) cannot access device {d} (rank {r}): hipDeviceCanAccessPeer is false"
    return None


class P2PComm:
    """The `_DistComm` interface (reduce_scatter / all_gather / all_reduce on bucket regions of the flat buffers) over mapped peer memory.
    `grads` / `params`: this rank's flat bf16 buffers (what Zero2Optimizer was built on); `n_buckets`: how many distinct regions will be
    exchanged per step (flags are indexed by the bucket's position in call order within a step)."""
    rs_algo = "p2p"
    staged = False

    def __init__(self, grads: torch.Tensor, params: torch.Tensor, n_buckets: int, group=None, timeout_s: float = 600.0):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # ---- may this job exchange through mapped peer memory at all?  Decided by every rank together BEFORE anything is mapped
        why = None
        if not (grads.is_cuda and params.is_cuda and grads.dtype == torch.bfloat16 and params.dtype == torch.bfloat16):
            why = "the flat buffers are not bf16 CUDA tensors"
        elif self.world > 16:
            why = f"world {self.world} > 16"
        devs: List = [None] * self.world
        dist.all_gather_object(devs, grads.device.index if grads.is_cuda else -1, group=group)
        if why is None:
            why = p2p_refusal(devs, self.rank)
        whys: List = [None] * self.world
        dist.all_gather_object(whys, why, group=group)
        bad = [(r, w) for r, w in enumerate(whys) if w]
        if bad:
            raise P2PUnavailable(f"rank {bad[0][0]}: {bad[0][1]}")
        self.G, self.P = grads, params
        self.nb = n_buckets
        self.timeout_us = max(1, int(timeout_s * 1e6))
        dev = grads.device
        # flags: [2 * n_buckets] ints per rank (gradient-final, parameter-arrived)
        self.flags = torch.zeros(2 * n_buckets, dtype=torch.int32, device=dev)
        # the error word: pinned host memory (device-visible at the same address) — the wait kernels write it at system scope, the reduce kernel
        # reads it, the host reads it WITHOUT a synchronisation at every call (a timeout of an earlier step surfaces at the next one)
        self.err = torch.zeros(1, dtype=torch.int32).pin_memory()
        # sharing and opening can fail on ONE rank (a buffer that is a view into an allocation the runtime will not export, an open the
        # driver refuses): every step that can fail is followed by an agreement, so the ranks give up TOGETHER and no rank ever waits on a flag
        # nobody will raise
        def agree(err):
            errs: List = [None] * self.world
            dist.all_gather_object(errs, err, group=group)
            bad = [(r, e) for r, e in enumerate(errs) if e]
            if bad:
                self._peers = []
                raise P2PUnavailable(f"rank {bad[0][0]}: {bad[0][1]}")
        mine, err = None, None
        try:
            mine = (_share(self.G), _share(self.P), _share(self.flags))
        except Exception as e:                            # noqa: BLE001 — reported to every rank, the transport is refused
            err = f"cannot export the buffers for IPC: {type(e).__name__}: {str(e)[:160]}"
        agree(err)
        every: List = [None] * self.world
        dist.all_gather_object(every, mine, group=group)
        self._peers = []                                  # keep the opened tensors alive: they own the IPC mappings
        g_ptr, p_ptr, f_ptr = [], [], []
        try:
            for r, d in enumerate(every):
                if r == self.rank:
                    g, p, f = self.G, self.P, self.flags
                else:
                    g, p, f = _open(d[0]), _open(d[1]), _open(d[2])
                    if not (g.numel() == self.G.numel() and p.numel() == self.P.numel() and f.numel() == self.flags.numel()):
                        raise ValueError(f"rank {r}'s buffers have other sizes than this rank's")
                self._peers.append((g, p, f))
                g_ptr.append(g.data_ptr()); p_ptr.append(p.data_ptr()); f_ptr.append(f.data_ptr())
        except Exception as e:                            # noqa: BLE001
            err = f"cannot open a peer's buffers: {type(e).__name__}: {str(e)[:160]}"
        agree(err)                                        # (also the barrier: every rank has every mapping before anyone raises a flag)
        self._g_base, self._p_base = g_ptr, p_ptr
        self._f_arr = (ctypes.c_void_p * self.world)(*f_ptr)
        self.seq = 0                                      # exchange number: one per optimizer step
        self._rs_calls = self._ag_calls = 0

    # ---- helpers
    def _ptrs(self, bases, byte_off):
        return (ctypes.c_void_p * self.world)(*[b + byte_off for b in bases])

    def check(self, sync: bool = False):
        """Raise if a wait gave up.  A plain read of the pinned error word (no synchronisation): every entry point calls it, so a timeout of
        step s — whose sums were poisoned, i.e. whose optimizer step was skipped on every rank — stops the job at step s + 1 at the latest.
        sync=True drains the device first (tests, end of a run)."""
        if sync:
            torch.cuda.synchronize()
        e = int(self.err[0])
        if e:
            raise P2PTimeout(f"p2p exchange: peer {e - 1} did not raise its flag within {self.timeout_us / 1e6:g} s (rank {self.rank}, exchange "
                             f"{self.seq}); the step that waited for it was skipped on every rank (non-finite gradient norm)")

    # ---- the _DistComm interface
    def reduce_scatter(self, out_chunk: torch.Tensor, region: torch.Tensor):
        """region = this rank's gradients of one bucket (world chunks); out_chunk = region[rank * per : (rank + 1) * per] receives the
        fp32 rank-order sum of that chunk over all ranks."""
        self.check()
        b = self._rs_calls % self.nb                      # the k-th reduce of a step is the same bucket on every rank (same call order)
        if b == 0:
            self.seq += 1                                 # a new exchange: every flag's next value
        self._rs_calls += 1
        st = torch.cuda.current_stream().cuda_stream
        lib = ops.lib()
        lib.call("molly_p2p_flag_set", st, self.flags.data_ptr() + 4 * b, self.seq)            # my gradients of this bucket are final
        lib.call("molly_p2p_flag_wait", st, self._f_arr, self.world, b, self.seq, self.timeout_us, self.err)
        off = (out_chunk.data_ptr() - self.G.data_ptr())                                         # my chunk's byte offset in every G
        assert 0 <= off < self.G.numel() * 2
        # (given the error word: a wait that gave up — this bucket's or an earlier one's — turns the sums into NaN: see the module docstring)
        lib.call("molly_p2p_reduce_bf16", st, self._ptrs(self._g_base, off), self.world, out_chunk.numel(), out_chunk, self.err)

    def all_gather(self, region: torch.Tensor, chunk: torch.Tensor):
        """chunk = this rank's updated parameters of one bucket (inside region, inside P): written into every peer's P at the same offset;
        returns (in stream order) when every rank's chunk of this bucket has arrived here."""
        self.check()
        b = self._ag_calls % self.nb
        self._ag_calls += 1
        st = torch.cuda.current_stream().cuda_stream
        lib = ops.lib()
        off = chunk.data_ptr() - self.P.data_ptr()
        assert 0 <= off < self.P.numel() * 2
        lib.call("molly_p2p_push_bf16", st, chunk, self._ptrs(self._p_base, off), self.world, self.rank, chunk.numel())
        lib.call("molly_p2p_flag_set", st, self.flags.data_ptr() + 4 * (self.nb + b), self.seq)
        lib.call("molly_p2p_flag_wait", st, self._f_arr, self.world, self.nb + b, self.seq, self.timeout_us, self.err)

    def all_reduce(self, t: torch.Tensor):
        dist.all_reduce(t, group=self.group)              # one scalar (the squared gradient norm): the library's collective

    def all_reduce_region(self, region: torch.Tensor):
        raise NotImplementedError("p2p exchange: ZeRO stage 0 (plain all-reduce) keeps the library's collective")
