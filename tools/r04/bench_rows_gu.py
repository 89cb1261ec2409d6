#!/usr/bin/env python3
"""gate | up with SwiGLU at decode rows (M = 32): K slices + combine launch (rows_gu 0) against the one-slice kernel that forms the activation
from its accumulators (64 | 128 W rows per tile).  Six rotated weight copies: every launch streams from HBM."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molly_amd import ops  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
for name, ff, k in (("8b", 12288, 4096), ("256t", 16384, 4096), ("384t", 24576, 4096), ("4b", 9728, 2560), ("1.7b", 6144, 2048)):
    for m in (32,):
        x = rnd(m, k)
        ws = [rnd(2 * ff, k) * k ** -0.5 for _ in range(6)]
        act = torch.empty(m, ff, dtype=torch.bfloat16, device=dev)
        out = []
        for mode in (0, 32, 64, 128):
            c = ops.GemmContext(); c.ensure_workspace(256 << 20); c.set("rows_gu", mode)
            with ops.use_gemm_context(c):
                if not ops.gemm_rows_tail_supported(m, 2 * ff, k, "swiglu"):
                    out.append("n/a"); continue
                for w in ws:
                    ops.gemm_rows_swiglu(x, w, None, act)
                best = 1e9
                for _ in range(3):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(4):
                        for w in ws:
                            ops.gemm_rows_swiglu(x, w, None, act)
                    e1.record(); torch.cuda.synchronize()
                    best = min(best, e0.elapsed_time(e1) / 24)
                out.append(f"{mode}: {best * 1e3:6.1f} us {2 * ff * k * 2 / best / 1e9:5.2f} TB/s (cfg {c.get('last_config')})")
        print(f"{name:5s} M={m} ff={ff} K={k}  " + " | ".join(out), flush=True)
