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


class P2PComm:
    """The `_DistComm` interface (reduce_scatter / all_gather / all_reduce on bucket regions of the flat buffers) over mapped peer memory.
    `grads` / `params`: this rank's flat bf16 buffers (what Zero2Optimizer was built on); `n_buckets`: how many distinct regions will be
    exchanged per step (flags are indexed by the bucket's position in call order within a step)."""
    rs_algo = "p2p"
    staged = False

    def __init__(self, grads: torch.Tensor, params: torch.Tensor, n_buckets: int, group=None, max_spins: int = 1 << 24):
        assert grads.is_cuda and params.is_cuda and grads.dtype == torch.bfloat16 and params.dtype == torch.bfloat16
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        assert self.world <= 16
        self.G, self.P = grads, params
        self.nb = n_buckets
        self.max_spins = max_spins
        dev = grads.device
        # flags: [2 * n_buckets] ints per rank (gradient-final, parameter-arrived); err: one int
        self.flags = torch.zeros(2 * n_buckets, dtype=torch.int32, device=dev)
        self.err = torch.zeros(1, dtype=torch.int32, device=dev)
        mine = (_share(self.G), _share(self.P), _share(self.flags))
        every: List = [None] * self.world
        dist.all_gather_object(every, mine, group=group)
        self._peers = []                                  # keep the opened tensors alive: they own the IPC mappings
        g_ptr, p_ptr, f_ptr = [], [], []
        for r, d in enumerate(every):
            if r == self.rank:
                g, p, f = self.G, self.P, self.flags
            else:
                g, p, f = _open(d[0]), _open(d[1]), _open(d[2])
                assert g.numel() == self.G.numel() and p.numel() == self.P.numel() and f.numel() == self.flags.numel()
            self._peers.append((g, p, f))
            g_ptr.append(g.data_ptr()); p_ptr.append(p.data_ptr()); f_ptr.append(f.data_ptr())
        self._g_base, self._p_base = g_ptr, p_ptr
        self._f_arr = (ctypes.c_void_p * self.world)(*f_ptr)
        self.seq = 0                                      # exchange number: one per optimizer step
        self._rs_calls = self._ag_calls = 0
        dist.barrier(group=group)                         # every rank has every mapping before anyone raises a flag

    # ---- helpers
    def _ptrs(self, bases, byte_off):
        return (ctypes.c_void_p * self.world)(*[b + byte_off for b in bases])

    def check(self):
        """Raise if a spin gave up (synchronises)."""
        e = int(self.err.item())
        if e:
            raise RuntimeError(f"p2p exchange: peer {e - 1} never raised its flag (rank {self.rank}, exchange {self.seq})")

    # ---- the _DistComm interface
    def reduce_scatter(self, out_chunk: torch.Tensor, region: torch.Tensor):
        """region = this rank's gradients of one bucket (world chunks); out_chunk = region[rank * per : (rank + 1) * per] receives the
        fp32 rank-order sum of that chunk over all ranks."""
        b = self._rs_calls % self.nb                      # the k-th reduce of a step is the same bucket on every rank (same call order)
        if b == 0:
            self.seq += 1                                 # a new exchange: every flag's next value
        self._rs_calls += 1
        st = torch.cuda.current_stream().cuda_stream
        lib = ops.lib()
        lib.call("molly_p2p_flag_set", st, self.flags.data_ptr() + 4 * b, self.seq)            # my gradients of this bucket are final
        lib.call("molly_p2p_flag_wait", st, self._f_arr, self.world, b, self.seq, self.max_spins, self.err)
        off = (out_chunk.data_ptr() - self.G.data_ptr())                                         # my chunk's byte offset in every G
        assert 0 <= off < self.G.numel() * 2
        lib.call("molly_p2p_reduce_bf16", st, self._ptrs(self._g_base, off), self.world, out_chunk.numel(), out_chunk)

    def all_gather(self, region: torch.Tensor, chunk: torch.Tensor):
        """chunk = this rank's updated parameters of one bucket (inside region, inside P): written into every peer's P at the same offset;
        returns (in stream order) when every rank's chunk of this bucket has arrived here."""
        b = self._ag_calls % self.nb
        self._ag_calls += 1
        st = torch.cuda.current_stream().cuda_stream
        lib = ops.lib()
        off = chunk.data_ptr() - self.P.data_ptr()
        assert 0 <= off < self.P.numel() * 2
        lib.call("molly_p2p_push_bf16", st, chunk, self._ptrs(self._p_base, off), self.world, self.rank, chunk.numel())
        lib.call("molly_p2p_flag_set", st, self.flags.data_ptr() + 4 * (self.nb + b), self.seq)
        lib.call("molly_p2p_flag_wait", st, self._f_arr, self.world, self.nb + b, self.seq, self.max_spins, self.err)

    def all_reduce(self, t: torch.Tensor):
        dist.all_reduce(t, group=self.group)              # one scalar (the squared gradient norm): the library's collective

    def all_reduce_region(self, region: torch.Tensor):
        raise NotImplementedError("p2p exchange: ZeRO stage 0 (plain all-reduce) keeps the library's collective")
