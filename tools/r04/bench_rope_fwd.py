#!/usr/bin/env python3
"""norm + rope forward in isolation at the decoder's shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molly_amd import ops
from molly_amd.qwen3 import rope_tables
M, T, nq, nk, hd = int(os.environ.get("ROPE_M", 32768)), 2048, 16, 8, 128
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
sets = [(rnd(M, (nq + 2 * nk) * hd), torch.empty(M, (nq + nk) * hd, dtype=torch.bfloat16, device=dev)) for _ in range(3)]
qw, kw = rnd(hd), rnd(hd)
cos, sin = rope_tables(T, hd, 1e6, dev, torch.bfloat16)
def f(i):
    src, d = sets[i % 3]
    ops.norm_rope_fwd(src, d, nq, nk, hd, T, qw, kw, cos, sin, eps=1e-6)
for i in range(3): f(i)
best = 1e9
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(9): f(i)
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 9)
byt = M * (nq + nk) * hd * 2 * 2
print(f"norm_rope_fwd M={M}: {best*1e3:.1f} us  {byt/best/1e9:.2f} TB/s")
