#!/usr/bin/env python3
"""The fused LoRA kernels on the rows of the B = 16 step (M = 32768, rank 64 padded): us and the HBM rate of their algorithmic bytes.
    python tools/r05/bench_lora_kernels.py            (MOLLY_LIB_PATH=tools/variants/libmolly_<v>.so for a variant)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molly_amd import ops  # noqa: E402

M = 32768
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()


def timed(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


for K in (2048, 6144):
    x, A = rnd(M, K), rnd(64, K)
    xd = torch.empty_like(x); t = torch.empty(M, 64, dtype=torch.bfloat16, device=dev)
    dx = rnd(M, K)
    us = timed(lambda: ops.lora_down_drop(x, A, 0.05, 1234, 2.0, xd=xd, out=t))
    print(f"K={K}: down + dropout + xd   {us:7.1f} us  {2 * M * K * 2 / us / 1e6:5.2f} TB/s")
    us = timed(lambda: ops.lora_down_drop(x, A, 0.0, 0, 2.0, out=t))
    print(f"K={K}: down, plain (dt form) {us:7.1f} us  {M * K * 2 / us / 1e6:5.2f} TB/s")
    us = timed(lambda: ops.lora_up_drop_acc(t, A, dx, 0.05, 1234))
    print(f"K={K}: up + mask + add to dx {us:7.1f} us  {2 * M * K * 2 / us / 1e6:5.2f} TB/s")
