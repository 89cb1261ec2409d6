#!/usr/bin/env python3
"""Forward attention: attn_fwd_kernel against attn_fwd_pipe_kernel (MOLLY_ATTN_FWD_PIPE read per call), causal and not, one process.
    python tools/r05/bench_attn_pipe.py [--shapes 8,2048,16,8,128;16,2048,16,8,128] [--pipes 0,8] [--reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molly_amd import ops  # noqa: E402


def timeit(fn, n):
    fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="8,2048,16,8,128;16,2048,16,8,128")
    ap.add_argument("--pipes", default="0,8")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--causal", default="1,0")
    a = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(0)
    for shape in a.shapes.split(";"):
        B, T, nh, nkv, hd = (int(x) for x in shape.split(","))
        M = B * T
        qkv = torch.randn(M, (nh + 2 * nkv) * hd, device="cuda", generator=g).bfloat16()
        q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
        o = torch.empty(M, nh * hd, dtype=torch.bfloat16, device="cuda")
        lse = torch.empty(B, nh, T, dtype=torch.float32, device="cuda")
        ref = {}
        for causal in (int(c) for c in a.causal.split(",")):
            fl = 4.0 * B * nh * T * T * hd / (2 if causal else 1)
            for pipe in a.pipes.split(","):
                os.environ["MOLLY_ATTN_FWD_PIPE"] = pipe
                t = timeit(lambda: ops.attn_fwd(q, k, v, B, T, nh, nkv, hd, hd ** -0.5, bool(causal), out=o, lse=lse), a.reps)
                key = causal
                if key not in ref:
                    ref[key] = (o.clone(), lse.clone())
                    same = ""
                else:
                    same = f"  max|dO| vs pipe {a.pipes.split(',')[0]}: {(o.float() - ref[key][0].float()).abs().max().item():.2e}"
                print(f"shape {shape:22s} causal {causal} pipe {pipe}: {t * 1e3:8.1f} us  {fl / t / 1e9:7.1f} TF/s{same}", flush=True)


if __name__ == "__main__":
    main()
