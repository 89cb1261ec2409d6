#!/usr/bin/env python3
"""Same-process A/B of the streaming epilogue (gemm256_kernel<SE>, context key stream_epi) on the step's GEMM shapes: interleaved rounds, HIP
events, random operands; columns = stream_epi 0 | 1 (| 3: residual too) and torch.matmul for orientation.  Also checks bit-identity of the outputs.
    python tools/r06/ab_stream_epi.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molly_amd import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
SHAPES = [("qkv fwd", "nt", M, 4096, 2048, False), ("o fwd +res", "nt", M, 2048, 2048, True), ("gate|up (plain)", "nt", M, 12288, 2048, False),
          ("down fwd +res", "nt", M, 2048, 6144, True), ("lm_head fwd 8k", "nt", 8192, 151936, 2048, False),
          ("qkv dgrad", "nn", M, 2048, 4096, False), ("gate|up dgrad", "nn", M, 2048, 12288, False), ("o dgrad", "nn", M, 2048, 2048, False),
          ("down dgrad (plain)", "nn", M, 6144, 2048, False)]
MODES = [0, 1, 3]


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
    ctxs = {}
    for m in MODES:
        ctxs[m] = ops.GemmContext()
        ctxs[m].ensure_workspace(256 << 20)
        ctxs[m].set("stream_epi", m)
    print(f"{'shape':20s} {'M':>6s} {'N':>7s} {'K':>6s} " + " ".join(f"{'se=' + str(m) + ' us/TF/cfg':>22s}" for m in MODES) + "   torch TF   bit-identical")
    for name, form, m, n, k, res in SHAPES:
        a = rnd(m, k)
        b = rnd(n, k) if form == "nt" else rnd(k, n)
        r = rnd(m, n) if res else None
        outs = {md: torch.empty(m, n, dtype=torch.bfloat16, device=dev) for md in MODES}
        kw = dict(res=r) if res else {}
        if form == "nn":
            kw["b_kmajor"] = True
        best = {md: 1e9 for md in MODES}
        cfg = {}
        for rr in range(7):
            for md in MODES:
                with ops.use_gemm_context(ctxs[md]):
                    ops.gemm(a, b, out=outs[md], **kw)
                    cfg[md] = ctxs[md].get("last_config")
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        ops.gemm(a, b, out=outs[md], **kw)
                    e1.record()
                    torch.cuda.synchronize()
                    best[md] = min(best[md], e0.elapsed_time(e1) / 5)
        tb = 1e9
        bt = b.t() if form == "nt" else b
        for rr in range(7):
            c = torch.matmul(a, bt)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                c = torch.matmul(a, bt)
            e1.record()
            torch.cuda.synchronize()
            tb = min(tb, e0.elapsed_time(e1) / 5)
        same = all(torch.equal(outs[0], outs[md]) for md in MODES)
        fl = 2.0 * m * n * k
        print(f"{name:20s} {m:6d} {n:7d} {k:6d} " + " ".join(f"{best[md] * 1e3:9.1f} {fl / best[md] / 1e9:6.0f} {cfg[md]:5d}" for md in MODES)
              + f"   {fl / tb / 1e9:7.0f}   {same}", flush=True)
        del a, b, outs


if __name__ == "__main__":
    main()
