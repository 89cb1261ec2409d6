#!/usr/bin/env python3
"""norm + rotary forward at a decoder layer's shape (q | k heads of 128, M tokens): us per launch and TB/s (q | k read once, written once).
    MOLLY_ROPE_FWD_FAST=0 | MOLLY_ROPE_FWD_ROWS_MIN=1000000 | (default) python tools/r06/bench_rope_fwd.py [M] [nh] [nkv] [hd] [T]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molly_amd import ops  # noqa: E402

M, nh, nkv, hd, T = (int(x) for x in (sys.argv[1:] + ["32768", "16", "8", "128", "2048"][len(sys.argv) - 1:]))
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.rand(M, (nh + 2 * nkv) * hd, device="cuda", generator=g) * 2 - 1).bfloat16()
qk = torch.empty(M, (nh + nkv) * hd, dtype=torch.bfloat16, device="cuda")
qn = (torch.rand(hd, device="cuda", generator=g) + 0.5).bfloat16() if hd == 128 else None
kn = (torch.rand(hd, device="cuda", generator=g) + 0.5).bfloat16() if hd == 128 else None
cos, sin = torch.rand(T, hd // 2, device="cuda", generator=g), torch.rand(T, hd // 2, device="cuda", generator=g)
big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")          # 1 GiB written between rounds: cold caches, as in the step
best = 1e9
for _ in range(7):
    big.fill_(1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4):
        ops.norm_rope_fwd(qkv, qk, nh, nkv, hd, T, qn, kn, cos, sin, q_scale=1.0 if hd == 128 else hd ** -0.5)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 4)
by = 2.0 * M * (nh + nkv) * hd * 2
print(f"{os.environ.get('MOLLY_ROPE_FWD_FAST', '-')}/{os.environ.get('MOLLY_ROPE_FWD_ROWS_MIN', '-')} M {M} heads {nh}+{nkv} x {hd}: {best * 1e3:8.1f} us  {by / best / 1e9:6.2f} TB/s")
