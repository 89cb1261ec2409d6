#!/usr/bin/env python3
"""Sustained TFLOP/s (2 s of back-to-back launches: the clock the power manager settles at included) of the step's K = 2,048 forward shapes by the
tile walk's GROUP_M (M-tiles an XCD walks side by side: it sets how many operand panels its 32 concurrent tiles share).
    python tools/r05/sweep_group_m.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molly_amd import ops  # noqa: E402
from molly_amd._lib import lib  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
M = 32768
for name, N, K in (("qkv fwd", 4096, 2048), ("gate|up fwd", 12288, 2048), ("down fwd", 2048, 6144)):
    a, b = rnd(M, K), rnd(N, K)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    row = []
    for gm in (4, 1, 2, 8, 16, 4):
        lib().call("molly_gemm_set_group_m", gm)
        for _ in range(5):
            ops.gemm_nt(a, b, out=c)
        torch.cuda.synchronize()
        n = int(2.0 * 1.3e15 / (2.0 * M * N * K))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ops.gemm_nt(a, b, out=c)
        e1.record(); torch.cuda.synchronize()
        row.append(f"group_m {gm:2d}: {2.0 * M * N * K * n / (e0.elapsed_time(e1) * 1e-3) / 1e12:7.1f}")
    lib().call("molly_gemm_set_group_m", 4)
    print(f"{name:12s} " + "   ".join(row), flush=True)
