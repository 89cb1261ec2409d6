#!/usr/bin/env python3
"""RMSNorm forward with the transposed second store against RMSNorm + transpose launch (M = 32,768 x 2,048; 16,384 x 2,048)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molly_amd import ops


def timeit(fn, n=20):
    fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best * 1e3


for M, H in ((32768, 2048), (16384, 2048), (32768, 1024)):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(M, H, device="cuda", generator=g).bfloat16()
    w = (1 + 0.1 * torch.randn(H, device="cuda", generator=g)).bfloat16()
    y, yt = torch.empty_like(x), torch.empty(H, M, dtype=torch.bfloat16, device="cuda")
    y2, yt2 = torch.empty_like(x), torch.empty(H, M, dtype=torch.bfloat16, device="cuda")
    ops.rmsnorm_fwd(x, w, 1e-6, out=y)
    ops.transpose(y, yt)
    ops.rmsnorm_fwd(x, w, 1e-6, out=y2, out_t=yt2)
    torch.cuda.synchronize()
    print(f"M {M} H {H}: y equal {torch.equal(y, y2)}, yT equal {torch.equal(yt, yt2)}")
    a = timeit(lambda: ops.rmsnorm_fwd(x, w, 1e-6, out=y))
    b = timeit(lambda: ops.transpose(y, yt))
    c = timeit(lambda: ops.rmsnorm_fwd(x, w, 1e-6, out=y2, out_t=yt2))
    print(f"   rmsnorm {a:.1f} us + transpose {b:.1f} us = {a + b:.1f} us;  fused {c:.1f} us  ({3 * M * H * 2 / c / 1e6:.2f} TB/s of x + y + yT)")
