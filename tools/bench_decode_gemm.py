#!/usr/bin/env python3
"""Decode-step GEMMs (M = batch rows against whole weight matrices, six copies rotated so they stream from HBM): effective weight
bandwidth of the launcher's default (streaming decode-row kernel, cfg 1016, or the tiled one, cfg 32 + 1000 x slices), the tiled
kernel alone (streaming kernel off), and round 2's path (K split over the chip through the 256x256 kernel + fp32 slabs + reduce
launch).  python tools/bench_decode_gemm.py [--batch 32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molly_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, nargs="+", default=[32])
    args = ap.parse_args()
    dev = "cuda"
    new, tiled, old = ops.GemmContext(), ops.GemmContext(), ops.GemmContext()
    for c in (new, tiled, old):
        c.ensure_workspace(256 << 20)
    tiled.set("skinny", 0)
    old.set("skinny", 0); old.set("rows_tiled", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
    for m in args.batch:
        for name, n, k, f32 in [("8b qkv", 6144, 4096, 0), ("8b o", 4096, 4096, 0), ("8b gate|up", 24576, 4096, 0), ("8b down", 4096, 12288, 0),
                                ("8b lm_head", 151936, 4096, 1), ("4b gate|up", 19456, 2560, 0), ("4b down", 2560, 9728, 0),
                                ("1.7b qkv", 4096, 2048, 0), ("1.7b o", 2048, 2048, 0), ("1.7b gate|up", 12288, 2048, 0), ("1.7b down", 2048, 6144, 0)]:
            a = rnd(m, k)
            ws = [rnd(n, k) for _ in range(2 if n > 100000 else 6)]
            out = torch.empty(m, n, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
            res = {}
            for tag, c in (("default", new), ("tiled", tiled), ("r02 split-K", old)):
                with ops.use_gemm_context(c):
                    for w in ws:
                        ops.gemm_nt(a, w, out=out)
                    best = 1e9
                    for _ in range(3):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(4):
                            for w in ws:
                                ops.gemm_nt(a, w, out=out)
                        e1.record()
                        torch.cuda.synchronize()
                        best = min(best, e0.elapsed_time(e1) / (4 * len(ws)))
                res[tag] = (best * 1e3, n * k * 2 / (best * 1e-3) / 1e12, c.get("last_config"))
            print(f"{name:14s} M={m:3d} N={n:6d} K={k:6d}  " + " | ".join(f"{t}: {v[0]:7.1f} us {v[1]:5.2f} TB/s (cfg {v[2]})" for t, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
