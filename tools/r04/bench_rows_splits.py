#!/usr/bin/env python3
"""K-sliced decode-row GEMM (M = 32) + combine launch at a forced number of K slices (MOLLY_ROWS_FORCE_SPLITS, read once per process):
    for s in 2 3 4 5 6 8; do MOLLY_ROWS_FORCE_SPLITS=$s python tools/r04/bench_rows_splits.py; done"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molly_amd import ops  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
c = ops.GemmContext(); c.ensure_workspace(256 << 20)
out = []
for name, n, k in (("8b qkv", 6144, 4096), ("8b o", 4096, 4096), ("8b down", 4096, 12288), ("4b qkv", 6144, 2560), ("4b o", 2560, 4096), ("4b down", 2560, 9728)):
    x = rnd(32, k)
    ws = [rnd(n, k) for _ in range(8)]
    y = torch.empty(32, n, dtype=torch.bfloat16, device=dev)
    with ops.use_gemm_context(c):
        for w in ws:
            ops.gemm_nt(x, w, out=y)
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                for w in ws:
                    ops.gemm_nt(x, w, out=y)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 32)
        out.append(f"{name} {best * 1e3:5.1f} ({c.get('last_config') // 1000})")
print(f"splits {os.environ.get('MOLLY_ROWS_FORCE_SPLITS', 'auto'):>4}: " + " | ".join(out), flush=True)
