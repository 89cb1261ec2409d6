#!/usr/bin/env python3
"""Does reading part of a weight matrix shortly before the decode-row GEMM that streams it shorten that GEMM?  (Is the 256 MB
memory-side Infinity Cache a place to park the next projection's first megabytes during the previous launch's tail?)
Six copies of the matrix are rotated so nothing survives from the previous use; the probe reads the first `pre` MB of the copy
about to be used (a plain reduction kernel), then the GEMM alone is timed.   python tools/diag/mall_prefetch_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molly_amd import ops  # noqa: E402


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
    c = ops.GemmContext()
    c.ensure_workspace(256 << 20)
    M = 32
    for name, n, k in [("8b gate|up", 24576, 4096), ("8b down", 4096, 12288), ("8b qkv", 6144, 4096)]:
        x = rnd(M, k)
        ws = [rnd(n, k) for _ in range(6)]
        out = torch.empty(M, n, dtype=torch.bfloat16, device=dev)
        nbytes = n * k * 2
        for pre_mb in (0, 8, 25, 50, 100, 200):
            pre = min(pre_mb << 20, nbytes)
            ts = []
            with ops.use_gemm_context(c):
                for it in range(24):
                    w = ws[it % 6]
                    if pre:
                        w.view(-1)[: pre // 2].view(torch.int32).sum()          # reads `pre` bytes of this copy
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    ops.gemm_nt(x, w, out=out)
                    e1.record()
                    e1.synchronize()
                    if it >= 6:
                        ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            print(f"{name:12s} {nbytes >> 20:4d} MB  prefetched {pre >> 20:4d} MB: GEMM median {ts[len(ts) // 2]:6.1f} us  min {ts[0]:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
