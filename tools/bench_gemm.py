#!/usr/bin/env python3
"""GEMM micro-benchmark on the shapes of Molly-1.7B's train step (random data, HIP events, interleaved rounds in one
process — guide §5.4 rules 24/25).  python tools/bench_gemm.py [--tiles 0 128 512] [--torch]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molly_amd import ops  # noqa: E402
from molly_amd._lib import lib  # noqa: E402

M = 16384
SHAPES = [  # (name, form, M, N, K)
    ("qkv fwd", "nt", M, 4096, 2048), ("o fwd", "nt", M, 2048, 2048), ("gate|up fwd", "nt", M, 12288, 2048),
    ("down fwd", "nt", M, 2048, 6144), ("lm_head fwd", "nt", M, 151936, 2048),
    ("qkv dgrad", "nn", M, 2048, 4096), ("gate|up dgrad", "nn", M, 2048, 12288), ("down dgrad", "nn", M, 6144, 2048),
    ("lm_head dgrad", "nn", M, 2048, 151936),
    ("qkv wgrad", "tn", 4096, 2048, M), ("gate|up wgrad", "tn", 12288, 2048, M), ("down wgrad", "tn", 2048, 6144, M),
    ("lm_head wgrad", "tn", 151936, 2048, M),
    ("head fwd 4k", "nt", 4096, 151936, 2048), ("head dgrad 4k", "nn", 4096, 2048, 151936),
    ("head wgrad 4k", "tn", 151936, 2048, 4096),
    ("esm qkv", "nt", 4096, 3840, 1280), ("esm ffn1", "nt", 4096, 5120, 1280), ("esm ffn2", "nt", 4096, 1280, 5120),
    ("esm o", "nt", 4096, 1280, 1280), ("esm proj", "nt", 4096, 2048, 1280),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", type=int, nargs="+", default=[0, 128])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--torch", action="store_true", help="add a torch.matmul (hipBLASLt/rocBLAS) column for orientation")
    ap.add_argument("--sched", type=int, nargs="+", default=None,
                    help="compare barrier schedules of the 256x256 kernel: 0 four-phase, 1 two-phase, -1 per-form default")
    ap.add_argument("--lib-b", default=None,
                    help="a second build of the library (e.g. molly_amd/libmolly_hip_b.so = -DMOLLY_GEMM_ASM_DMA=0): every shape "
                         "is timed through both, interleaved in one process; columns 'A' (in-tree) and 'B'")
    ap.add_argument("--persist", type=int, nargs="+", default=None,
                    help="compare resident-block counts of the persistent 256x256 kernel (0 = one block per tile)")
    args = ap.parse_args()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
    if args.sched is not None:
        args.tiles = [800000 + 10 + x for x in args.sched]     # column key = 512 kernel with that schedule
    if args.persist is not None:
        args.tiles = [512000 + pv for pv in args.persist]      # column key = 512 kernel with that resident-block count
    libs = {"A": lib()}
    if args.lib_b:
        from molly_amd._lib import MollyLib
        libs["B"] = MollyLib(args.lib_b)
        wsb = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        libs["B"].call("molly_gemm_set_workspace", wsb, wsb.numel() * 4)
        ops.ensure_gemm_workspace(1 << 30)
        args.tiles = ["A", "B"]

    def run(t, a, b, out, kw):
        if isinstance(t, str):
            ak, bk = int(kw.get("a_kmajor", False)), int(kw.get("b_kmajor", False))
            K_, M_ = (a.shape if ak else a.shape[::-1])
            N_ = b.shape[1] if bk else b.shape[0]
            libs[t].call("molly_gemm_bf16", torch.cuda.current_stream().cuda_stream, a, b, out, None, None, M_, N_, K_,
                         a.stride(0), b.stride(0), out.stride(0), 0, 0, ak, bk)
        else:
            ops.gemm(a, b, out=out, **kw)
    print(f"{'shape':16s} {'form':4s} {'M':>7s} {'N':>7s} {'K':>7s} " + " ".join(f"{'BM' + str(t) + ' TF/s':>12s}" for t in args.tiles))
    for name, form, m, n, k in SHAPES:
        if args.only and args.only not in name:
            continue
        if form == "nt":
            a, b = rnd(m, k), rnd(n, k)
            kw = {}
        elif form == "nn":
            a, b = rnd(m, k), rnd(k, n)
            kw = dict(b_kmajor=True)
        else:
            a, b = rnd(k, m), rnd(k, n)
            kw = dict(a_kmajor=True, b_kmajor=True)
        out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        best = {t: 1e9 for t in args.tiles}
        for r in range(args.rounds):
            for t in args.tiles:
                if isinstance(t, str):
                    pass
                elif t >= 800000:
                    lib().call("molly_gemm_force_tile", 0)
                    lib().call("molly_gemm_set_schedule", t - 800010)
                elif t >= 511000:
                    lib().call("molly_gemm_force_tile", 512)
                    lib().call("molly_gemm_set_persistent_blocks", t - 512000)
                else:
                    lib().call("molly_gemm_force_tile", t)
                run(t, a, b, out, kw)              # warm
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    run(t, a, b, out, kw)
                e1.record()
                torch.cuda.synchronize()
                best[t] = min(best[t], e0.elapsed_time(e1) / 3)
        lib().call("molly_gemm_force_tile", 0)
        lib().call("molly_gemm_set_persistent_blocks", 256)
        lib().call("molly_gemm_set_schedule", -1)
        fl = 2.0 * m * n * k
        tcol = ""
        if args.torch:
            ta = a.t() if form == "tn" else a
            tb = b.t() if form == "nt" else b
            tbest = 1e9
            for r in range(args.rounds):
                torch.matmul(ta, tb, out=out)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    torch.matmul(ta, tb, out=out)
                e1.record()
                torch.cuda.synchronize()
                tbest = min(tbest, e0.elapsed_time(e1) / 3)
            tcol = f"   torch {fl / (tbest * 1e-3) / 1e12:8.1f}"
        print(f"{name:16s} {form:4s} {m:7d} {n:7d} {k:7d} " + " ".join(f"{fl / (best[t] * 1e-3) / 1e12:12.1f}" for t in args.tiles) + tcol)
        del a, b, out


if __name__ == "__main__":
    main()
