#!/usr/bin/env python3
"""The encoders' forward GEMMs (ESM-2 650M / NT-500M widths) with their real epilogues at the row counts the BASELINE configs
give them: M = 4096 (C2: 8 samples x 512 residues), 1024 and 512 (C3 / C4 / C5 at B = 1).  Columns: the launcher's default, the
tiled decode-row kernel taking the small grids as 64-row tiles (MOLLY_GEMM_KEY_ROWS_MAX_M), and the 128x128 kernel forced (MOLLY_3STAGE=1: its
3-stage ring, a knob measured equal).  python tools/bench_esm_gemm.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molly_amd import ops  # noqa: E402

CASES = [("qkv   +bias", 3840, 1280, dict(bias=True)), ("o     +bias+res", 1280, 1280, dict(bias=True, res=True)),
         ("ffn1  +bias+gelu", 5120, 1280, dict(bias=True, gelu=True)), ("ffn2  +bias+res", 1280, 5120, dict(bias=True, res=True)),
         ("proj  +bias", 2048, 1280, dict(bias=True)), ("proj 8b +bias", 4096, 1280, dict(bias=True))]


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
    ctxs = {"default": ops.GemmContext(), "64-row tiles": ops.GemmContext(), "128x128": ops.GemmContext()}
    for c in ctxs.values():
        c.ensure_workspace(256 << 20)
    ctxs["default"].set("rows_max_m", 64)            # (first column: without the 64-row tiles, which are the default up to 1,024 rows)
    ctxs["64-row tiles"].set("rows_max_m", 8192)
    ctxs["128x128"].set("force_tile", 128); ctxs["128x128"].set("small3", int(os.environ.get("MOLLY_3STAGE", "0")))
    if os.environ.get("MORE"):                       # (MORE=1: + the 256x256 kernel forced, and stream-K forced wherever it applies)
        ctxs["256x256"] = ops.GemmContext(); ctxs["256x256"].ensure_workspace(256 << 20); ctxs["256x256"].set("force_tile", 512)
        ctxs["stream-K"] = ops.GemmContext(); ctxs["stream-K"].ensure_workspace(256 << 20); ctxs["stream-K"].set("streamk", 2)
    for M in (int(x) for x in os.environ.get("M", "4096,1024,512").split(",")):
        print(f"M = {M}")
        print(f"{'gemm':18s} {'N':>6s} {'K':>6s} " + " ".join(f"{t + ' us/TF/cfg':>24s}" for t in ctxs))
        for name, n, k, ep in CASES:
            a, b = rnd(M, k), rnd(n, k)
            out = torch.empty(M, n, dtype=torch.bfloat16, device=dev)
            bias = rnd(n) if ep.get("bias") else None
            res = rnd(M, n) if ep.get("res") else None
            best = {t: 1e9 for t in ctxs}
            cfg = {}
            for r in range(5):
                for t, c in ctxs.items():
                    with ops.use_gemm_context(c):
                        ops.gemm_nt(a, b, out=out, bias=bias, res=res, gelu=bool(ep.get("gelu")))
                        cfg[t] = c.get("last_config")
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(10):
                            ops.gemm_nt(a, b, out=out, bias=bias, res=res, gelu=bool(ep.get("gelu")))
                        e1.record()
                        torch.cuda.synchronize()
                        best[t] = min(best[t], e0.elapsed_time(e1) / 10)
            print(f"{name:18s} {n:6d} {k:6d} " + " ".join(f"{best[t] * 1e3:8.1f} {2.0 * M * n * k / best[t] / 1e9:6.0f} {cfg[t]:8d}" for t in ctxs))


if __name__ == "__main__":
    main()
