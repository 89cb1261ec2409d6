#!/usr/bin/env python3
"""Does a decode-row GEMM run faster when its weight matrix was read just before (Infinity-Cache resident) than from HBM?  Qwen3-8B's projections at
32 rows: cold (512 MB written elsewhere first), touched (cold, then the matrix read once by a reduction kernel), hot (the same launch again).
    python tools/r05/probe_mall_weights.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molly_amd import ops  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
ops.ensure_gemm_workspace(256 << 20)
B = 32
for name, N, K in (("q|k|v", 6144, 4096), ("o", 4096, 4096), ("down", 4096, 12288), ("gate|up", 24576, 4096)):
    w, x = rnd(N, K), rnd(B, K)
    out = torch.empty(B, N, dtype=torch.bfloat16, device=dev)
    f = lambda: ops.gemm_nt(x, w, out=out)
    f(); torch.cuda.synchronize()
    res = {}
    for mode in ("cold", "touched", "hot"):
        ts = []
        for _ in range(7):
            if mode != "hot":
                flush.fill_(1)
            if mode == "touched":
                w.view(torch.int32).sum()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        res[mode] = sorted(ts)[len(ts) // 2]
    mb = N * K * 2 / 1e6
    print(f"{name:8s} {mb:6.1f} MB   " + "   ".join(f"{m} {t:6.1f} us = {mb / t / 1e0 * 1e-6 * 1e6 / 1e3:5.2f} TB/s" for m, t in res.items()), flush=True)
