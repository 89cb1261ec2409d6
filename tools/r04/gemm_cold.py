#!/usr/bin/env python3
"""The decoder layer's GEMM shapes with operands that are NOT resident in the 256 MB Infinity Cache: every launch of the timed loop
takes the next of R operand sets (activations, weights and outputs of their own; R x footprint > 1 GB), as consecutive layers of a
step do.  tools/bench_gemm.py re-runs one launch on one set of buffers — anything under 256 MB then streams from the cache, which
flatters kernels that are sensitive to memory latency.  Columns: hot (one set), cold (rotating sets) for molly's kernel under a few
launch knobs, and torch.matmul (hipBLASLt / rocBLAS) hot and cold.      python tools/r04/gemm_cold.py [M ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molly_amd import ops  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
SHAPES = [("qkv fwd", "nt", 4096, 2048), ("o fwd", "nt", 2048, 2048), ("gate|up fwd", "nt", 12288, 2048), ("down fwd", "nt", 2048, 6144),
          ("qkv dgrad", "nn", 2048, 4096), ("down dgrad", "nn", 6144, 2048), ("gate|up dgrad", "nn", 2048, 12288)]
KNOBS = [("default", {}), ("group_m=2", {"group_m": 2}), ("group_m=8", {"group_m": 8}), ("group_m=16", {"group_m": 16}),
         ("blocks=-3", {"persistent_blocks": -3}), ("blocks=0", {"persistent_blocks": 0})]


def timeit(fns, reps):
    for f in fns:
        f()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            fns[i % len(fns)]()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def main():
    Ms = [int(x) for x in sys.argv[1:]] or [16384, 32768]
    ctxs = {}
    for name, kv in KNOBS:
        c = ops.GemmContext()
        c.ensure_workspace(1 << 28)
        for k, v in kv.items():
            c.set(k, v)
        ctxs[name] = c
    for M in Ms:
        print(f"M = {M}   (TFLOP/s; hot = one operand set, cold = rotating sets larger than the Infinity Cache)")
        print(f"{'shape':14s} {'sets':>4s} " + " ".join(f"{n + ' hot':>14s} {n + ' cold':>15s}" for n, _ in KNOBS[:1]) + " " +
              " ".join(f"{n + ' cold':>15s}" for n, _ in KNOBS[1:]) + f" {'torch hot':>10s} {'torch cold':>10s}")
        for name, form, n, k in SHAPES:
            foot = (M * k + n * k + M * n) * 2
            R = max(2, int((1200 << 20) // foot) + 1)
            sets = [(rnd(M, k), rnd(n, k) if form == "nt" else rnd(k, n), torch.empty(M, n, dtype=torch.bfloat16, device=dev)) for _ in range(R)]
            fl = 2.0 * M * n * k
            reps = max(R * 2, 8)

            def ours(s, ctx):
                a, b, c = s
                def f():
                    with ops.use_gemm_context(ctx):
                        if form == "nt":
                            ops.gemm_nt(a, b, out=c)
                        else:
                            ops.gemm(a, b, out=c, b_kmajor=True)
                return f

            def vend(s):
                a, b, c = s
                return (lambda: torch.matmul(a, b.t(), out=c)) if form == "nt" else (lambda: torch.matmul(a, b, out=c))
            cols = []
            c0 = ctxs["default"]
            cols.append(fl / timeit([ours(sets[0], c0)], reps) / 1e9)
            for nm, _ in KNOBS:
                cols.append(fl / timeit([ours(s, ctxs[nm]) for s in sets], reps) / 1e9)
            th = fl / timeit([vend(sets[0])], reps) / 1e9
            tc = fl / timeit([vend(s) for s in sets], reps) / 1e9
            print(f"{name:14s} {R:4d} {cols[0]:14.0f} {cols[1]:15.0f} " + " ".join(f"{x:15.0f}" for x in cols[2:]) + f" {th:10.0f} {tc:10.0f}")
            del sets


if __name__ == "__main__":
    main()
