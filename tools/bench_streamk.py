#!/usr/bin/env python3
"""Stream-K against what it replaces, on the grids that do not fill whole rounds of the 256 CUs: the ESM-2 650M projections of the
headline step (M = 4096), the decoder GEMMs of BASELINE configs 3 / 4 at their defining B = 1 (M = 3072 / 4096) and the 1.7B model
at B = 1 / 2.  Columns: in-tree library with stream-K forced, its default (the launcher's cost model picks), stream-K off (128x128 kernel / split-K
slabs + reduce launch / partly filled rounds), and — when tools/variants/libmolly_r02.so exists — round 2's library.  Random
data, HIP events, interleaved rounds in one process.   python tools/bench_streamk.py [--only esm]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from molly_amd import ops  # noqa: E402
from molly_amd._lib import MollyLib, lib  # noqa: E402

SHAPES = [  # (name, form, M, N, K, epilogue)
    ("esm qkv +bias", "nt", 4096, 3840, 1280, "b"), ("esm o +bias+res", "nt", 4096, 1280, 1280, "br"),
    ("esm ffn1 +bias+gelu", "nt", 4096, 5120, 1280, "bg"), ("esm ffn2 +bias+res", "nt", 4096, 1280, 5120, "br"),
    ("esm proj +bias", "nt", 4096, 2048, 1280, "b"),
    ("esm o M=1024", "nt", 1024, 1280, 1280, "br"), ("esm ffn1 M=1024", "nt", 1024, 5120, 1280, "bg"),
    ("4b qkv fwd", "nt", 3072, 6144, 2560, ""), ("4b o fwd +res", "nt", 3072, 2560, 4096, "r"),
    ("4b down fwd +res", "nt", 3072, 2560, 9728, "r"), ("4b qkv dgrad", "nn", 3072, 2560, 6144, ""),
    ("4b o dgrad", "nn", 3072, 4096, 2560, ""), ("4b gate|up dgrad", "nn", 3072, 2560, 19456, ""),
    ("4b lm_head fwd 768", "nt", 768, 151936, 2560, ""),
    ("8b qkv fwd", "nt", 4096, 6144, 4096, ""), ("1.7b qkv B=1", "nt", 2048, 4096, 2048, ""),
    ("1.7b o B=1 +res", "nt", 2048, 2048, 2048, "r"), ("1.7b down B=2 +res", "nt", 4096, 2048, 6144, "r"),
    ("1.7b qkv wgrad", "tn", 4096, 2048, 16384, ""), ("1.7b down wgrad", "tn", 2048, 6144, 16384, ""),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
    on, auto, off = ops.GemmContext(), ops.GemmContext(), ops.GemmContext()
    for c in (on, auto, off):
        c.ensure_workspace(1 << 30)
    on.set("streamk", 2)          # wherever stream-K can run
    off.set("streamk", 0)         # never; `auto` keeps the default: the launcher's cost model decides
    old = None
    pth = os.path.join(ROOT, "tools", "variants", "libmolly_r02.so")
    if os.path.exists(pth):
        old = MollyLib(pth, strict=False)
        wso = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        old.call("molly_gemm_set_workspace", wso, wso.numel() * 4)
    cols = ["stream-K", "default", "off"] + (["r02 lib"] if old else [])
    print(f"{'shape':22s} {'M':>6s} {'N':>7s} {'K':>6s}  " + "  ".join(f"{c + ' us / TF/s / cfg':>26s}" for c in cols))
    st = torch.cuda.current_stream().cuda_stream
    for name, form, M, N, K, ep in SHAPES:
        if a.only and a.only not in name:
            continue
        x = rnd(K, M) if form == "tn" else rnd(M, K)
        w = rnd(N, K) if form == "nt" else rnd(K, N)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        bias = rnd(N) if "b" in ep else None
        res = rnd(M, N) if "r" in ep else None
        flags = (1 if bias is not None else 0) | (2 if "g" in ep else 0) | (4 if res is not None else 0)
        kw = dict(a_kmajor=form == "tn", b_kmajor=form != "nt")

        def run(which):
            if which == 3:
                old.call("molly_gemm_bf16", st, x, w, out, bias, res, M, N, K, x.stride(0), w.stride(0), N, N if res is not None else 0,
                         flags, int(form == "tn"), int(form != "nt"))
                return old.fn["molly_gemm_last_config"]()
            c = (on, auto, off)[which]
            with ops.use_gemm_context(c):
                ops.gemm(x, w, out=out, bias=bias, res=res, gelu="g" in ep, **kw)
            return c.get("last_config")
        best = [1e9] * len(cols)
        cfg = [0] * len(cols)
        for r in range(a.rounds):
            for i in range(len(cols)):
                cfg[i] = run(i)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run(i)
                e1.record()
                torch.cuda.synchronize()
                best[i] = min(best[i], e0.elapsed_time(e1) / 10)
        fl = 2.0 * M * N * K
        print(f"{name:22s} {M:6d} {N:7d} {K:6d}  " + "  ".join(f"{best[i] * 1e3:9.1f} {fl / best[i] / 1e9:7.0f} {cfg[i]:8d}" for i in range(len(cols))))
    print("stream-K wait timeouts:", on.streamk_timeouts())


if __name__ == "__main__":
    main()
