#!/usr/bin/env python3
"""Decode-step GEMMs (M = batch rows against whole weight matrices, weights rotated so they stream from HBM): effective weight
bandwidth of the heuristic's choice vs the forced 128x128 kernel.  python tools/bench_decode_gemm.py [--batch 32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molly_amd import ops  # noqa: E402
from molly_amd._lib import lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    args = ap.parse_args()
    dev = "cuda"
    ops.ensure_gemm_workspace(256 << 20)
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
    m = args.batch
    for name, n, k in [("8b qkv", 6144, 4096), ("8b o", 4096, 4096), ("8b gate|up", 24576, 4096), ("8b down", 4096, 12288),
                       ("1.7b qkv", 4096, 2048), ("1.7b o", 2048, 2048), ("1.7b gate|up", 12288, 2048), ("1.7b down", 2048, 6144)]:
        a = rnd(m, k)
        ws = [rnd(n, k) for _ in range(6)]
        out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        res = {}
        for tile in (0, 128):
            lib().call("molly_gemm_force_tile", tile)
            for w in ws:
                ops.gemm_nt(a, w, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                for w in ws:
                    ops.gemm_nt(a, w, out=out)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 24
            res[tile] = (t * 1e3, n * k * 2 / (t * 1e-3) / 1e12, lib().query("molly_gemm_last_config"))
        print(f"{name:14s} M={m} N={n:6d} K={k:6d}  " + " | ".join(
            f"{'heuristic' if t == 0 else '128x128'}: {v[0]:6.1f} us {v[1]:5.2f} TB/s (cfg {v[2]})" for t, v in res.items()), flush=True)
    lib().call("molly_gemm_force_tile", 0)


if __name__ == "__main__":
    main()
