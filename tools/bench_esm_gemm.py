#!/usr/bin/env python3
"""The ESM-2 650M forward GEMMs of the headline step (4096 tokens) with their real epilogues, per tile configuration
(0 = heuristic, 128 = 128x128 kernel, 512 = 256x256 kernel).  python tools/bench_esm_gemm.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molly_amd import ops  # noqa: E402
from molly_amd._lib import lib  # noqa: E402

M = int(os.environ.get("M", 4096))
CASES = [("qkv   +bias", 3840, 1280, dict(bias=True)), ("o     +bias+res", 1280, 1280, dict(bias=True, res=True)),
         ("ffn1  +bias+gelu", 5120, 1280, dict(bias=True, gelu=True)), ("ffn1  plain", 5120, 1280, dict()),
         ("ffn2  +bias+res", 1280, 5120, dict(bias=True, res=True)), ("proj  +bias", 2048, 1280, dict(bias=True))]


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
    tiles = [0, 128, 512]
    print(f"{'gemm':18s} {'N':>6s} {'K':>6s} " + " ".join(f"{'tile ' + str(t):>16s}" for t in tiles))
    for name, n, k, ep in CASES:
        a, b = rnd(M, k), rnd(n, k)
        out = torch.empty(M, n, dtype=torch.bfloat16, device=dev)
        bias = rnd(n) if ep.get("bias") else None
        res = rnd(M, n) if ep.get("res") else None
        best = {t: 1e9 for t in tiles}
        cfg = {}
        for r in range(5):
            for t in tiles:
                lib().call("molly_gemm_force_tile", t)
                ops.gemm_nt(a, b, out=out, bias=bias, res=res, gelu=bool(ep.get("gelu")))
                cfg[t] = lib().query("molly_gemm_last_config")
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    ops.gemm_nt(a, b, out=out, bias=bias, res=res, gelu=bool(ep.get("gelu")))
                e1.record()
                torch.cuda.synchronize()
                best[t] = min(best[t], e0.elapsed_time(e1) / 5)
        lib().call("molly_gemm_force_tile", 0)
        print(f"{name:18s} {n:6d} {k:6d} " + " ".join(f"{best[t]*1e3:6.1f}us {2.0*M*n*k/best[t]/1e9:5.0f} c{cfg[t]:<5d}"[:16].rjust(16) for t in tiles))


if __name__ == "__main__":
    main()
