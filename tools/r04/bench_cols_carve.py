#!/usr/bin/env python3
"""A forward GEMM a few tiles past whole rounds (Qwen3-4B q|k|v at one sample per GPU: 3,072 x 6,144 x 2,560 = 288 tiles of 256 x 256): the launcher's
choice (stream-K) against a column carve — the first 21 tile columns as one plain round (252 tiles), the last 3 as a K-sliced launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molly_amd import ops  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
c = ops.GemmContext(); c.ensure_workspace(512 << 20)


def timeit(fn, n=20):
    for _ in range(3):
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


for M, N, K in ((3072, 6144, 2560), (4096, 6144, 4096), (3072, 9728, 2560)):
    xs = [rnd(M, K) for _ in range(4)]
    ws = [rnd(N, K) for _ in range(4)]
    y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    with ops.use_gemm_context(c):
        i = [0]
        def whole():
            i[0] += 1
            ops.gemm_nt(xs[i[0] % 4], ws[i[0] % 4], out=y)
        t0 = timeit(whole); cfg0 = c.get("last_config")
        tm, tn = -(-M // 256), -(-N // 256)
        rem = (tm * tn) % 256
        cols = -(-rem // tm)                               # tile columns to carve so that the rest is <= whole rounds
        n0 = (tn - cols) * 256
        def carved():
            i[0] += 1
            x, w = xs[i[0] % 4], ws[i[0] % 4]
            ops.gemm_nt(x, w[:n0], out=y[:, :n0])
            ops.gemm_nt(x, w[n0:], out=y[:, n0:])
        t1 = timeit(carved); cfg1 = c.get("last_config")
        ref = y.clone(); whole(); torch.cuda.synchronize()
        print(f"M={M} N={N} K={K}: {tm * tn} tiles; whole {t0:6.1f} us (cfg {cfg0}) | first {tn - cols} tile columns + last {cols}: {t1:6.1f} us (cfg of the strip {cfg1})", flush=True)
