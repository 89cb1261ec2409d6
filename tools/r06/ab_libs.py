#!/usr/bin/env python3
"""Several builds of libmolly_hip.so on the step's GEMM shapes in ONE process (interleaved rounds, HIP events, random operands; guide 5.4 rule 24).
    python tools/r06/ab_libs.py [--m 32768] name=path[,env-free knob stream_epi=V] ...
e.g. python tools/r06/ab_libs.py base=molly_amd/libmolly_hip.so:0 se=molly_amd/libmolly_hip.so:1 f0=tools/variants/libmolly_sef0.so:1
(the number after the colon is the context's stream_epi).  Prints us / TFLOP/s per column and whether every column's output equals the first's."""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molly_amd._lib import MollyLib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=32768)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--torch", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--ldc-pad", type=int, default=0, help="extra elements per output row (is the store burst sensitive to the row stride?)")
    ap.add_argument("cols", nargs="+")
    a = ap.parse_args()
    M = a.m
    shapes = [("qkv fwd", "nt", M, 4096, 2048, 0), ("o fwd +res", "nt", M, 2048, 2048, 4), ("gate|up (plain)", "nt", M, 12288, 2048, 0),
              ("down fwd +res", "nt", M, 2048, 6144, 4), ("qkv dgrad", "nn", M, 2048, 4096, 0), ("gate|up dgrad", "nn", M, 2048, 12288, 0),
              ("o dgrad", "nn", M, 2048, 2048, 0), ("down dgrad (plain)", "nn", M, 6144, 2048, 0),
              # the fused epilogues: SwiGLU backward in the down-projection's dgrad (res = [gate | up] [M][2 ff], C = d[gate | up]), SwiGLU forward in
              # the gate | up projection (C = [gate | up] [M][2 ff], res = the activation [M][ff])
              ("down dgrad +swiglu'", "nn", M, 6144, 2048, 128), ("gate|up +swiglu", "nt", M, 12288, 2048, 64)]
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
    cols, libs = [], {}
    for c in a.cols:
        name, rest = c.split("=", 1)
        path, se = rest.rsplit(":", 1)
        path = os.path.abspath(path)
        if path not in libs:
            libs[path] = MollyLib(path, strict=False)
        L = libs[path]
        h = ctypes.c_void_p()
        assert L.fn["molly_gemm_ctx_create"](ctypes.byref(h)) == 0
        h = h.value
        ws = torch.empty(64 << 20, dtype=torch.float32, device=dev)
        L.call("molly_gemm_ctx_set_workspace", h, ws, ws.numel() * 4)
        L.call("molly_gemm_ctx_set", h, 17, int(se))
        cols.append((name, L, h, ws))
    st = torch.cuda.current_stream().cuda_stream
    print(f"{'shape':20s} {'M':>6s} {'N':>7s} {'K':>6s} " + " ".join(f"{n + ' us/TF':>16s}" for n, *_ in cols) + ("   torch TF" if a.torch else "") + "   same")
    for name, form, m, n, k, flags in shapes:
        if a.only and a.only not in name:
            continue
        A = rnd(m, k)
        B = rnd(n, k) if form == "nt" else rnd(k, n)
        R = rnd(m, n) if flags & 4 else rnd(m, 2 * n) if flags & 128 else torch.empty(m, n // 2, dtype=torch.bfloat16, device=dev) if flags & 64 else None
        ldc = (2 * n if flags & 128 else n) + a.ldc_pad
        outs = [torch.empty(m, ldc, dtype=torch.bfloat16, device=dev) for _ in cols]
        acts = [torch.empty(m, n // 2, dtype=torch.bfloat16, device=dev) for _ in cols] if flags & 64 else None

        def run(i):
            _, L, h, _ = cols[i]
            res = acts[i] if flags & 64 else R
            L.call("molly_gemm_bf16_ctx", h, st, A, B, outs[i], None, res, m, n, k, A.stride(0), B.stride(0), ldc, res.stride(0) if res is not None else 0,
                   flags, 0, int(form == "nn"))
        best = [1e9] * len(cols)
        for rr in range(a.rounds):
            for i in range(len(cols)):
                run(i)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    run(i)
                e1.record()
                torch.cuda.synchronize()
                best[i] = min(best[i], e0.elapsed_time(e1) / 5)
        tcol = ""
        if a.torch:
            tb = 1e9
            bt = B.t() if form == "nt" else B
            for rr in range(a.rounds):
                c = torch.matmul(A, bt)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    c = torch.matmul(A, bt)
                e1.record()
                torch.cuda.synchronize()
                tb = min(tb, e0.elapsed_time(e1) / 5)
            tcol = f"   {2.0 * m * n * k / tb / 1e9:8.0f}"
        fl = 2.0 * m * n * k
        same = [bool(torch.equal(outs[0], o)) and (acts is None or bool(torch.equal(acts[0], acts[i]))) for i, o in enumerate(outs)]
        print(f"{name:20s} {m:6d} {n:7d} {k:6d} " + " ".join(f"{b * 1e3:9.1f} {fl / b / 1e9:6.0f}" for b in best) + tcol + "   " + "".join("y" if x else "N" for x in same), flush=True)


if __name__ == "__main__":
    main()
