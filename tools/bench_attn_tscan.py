#!/usr/bin/env python3
"""Attention time against the sequence length at a constant token count (16,384 tokens, 16 q / 8 kv heads x 128, causal):
the number of query blocks is constant, the key steps per block grow with T, so time(T) = fixed-per-block + per-key-step * steps.
    python tools/bench_attn_tscan.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molly_amd import ops  # noqa: E402
from bench_attn import timeit  # noqa: E402


def main():
    nh, nkv, hd = 16, 8, 128
    M = 16384
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(M, (nh + 2 * nkv) * hd, device="cuda", generator=g).bfloat16()
    q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
    o = torch.empty(M, nh * hd, dtype=torch.bfloat16, device="cuda")
    do = torch.randn(M, nh * hd, device="cuda", generator=g).bfloat16()
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = dqkv[:, :nh * hd], dqkv[:, nh * hd:(nh + nkv) * hd], dqkv[:, (nh + nkv) * hd:]
    sc = hd ** -0.5
    res = {}
    for T in (512, 1024, 2048, 4096, 8192):
        B = M // T
        lse = torch.empty(B, nh, T, dtype=torch.float32, device="cuda")
        delta = torch.empty(B, nh, T, dtype=torch.float32, device="cuda")
        tf = timeit(lambda: ops.attn_fwd(q, k, v, B, T, nh, nkv, hd, sc, True, out=o, lse=lse)) * 1e3
        tb = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, B, T, nh, nkv, hd, sc, True, dq, dk, dv, delta_ws=delta)) * 1e3
        steps = (T / 128 + 1) / 2                      # average 128-key steps per 128-row query block
        fl = 4.0 * B * nh * T * T * hd / 2
        res[T] = (tf, tb, steps)
        print(f"T={T:5d} B={B:3d}: fwd {tf:8.1f} us ({fl / tf / 1e6:6.0f} TF/s)   bwd {tb:8.1f} us ({2.5 * fl / tb / 1e6:6.0f} TF/s)   "
              f"avg key steps per query block {steps:.1f}")
    nblk = M * nh / 128 / 256.0                        # query blocks per CU (fwd: one block = 128 rows of one head)
    for name, idx in (("fwd", 0), ("bwd", 1)):
        (t1, s1), (t2, s2) = (res[2048][idx], res[2048][2]), (res[8192][idx], res[8192][2])
        per_step = (t2 - t1) / (s2 - s1)
        fixed = t1 - per_step * s1
        print(f"{name}: per key step {per_step / nblk:.2f} us per block-on-a-CU, fixed {fixed / nblk:.2f} us per query block "
              f"(= {fixed / t1 * 100:.0f} % of the T=2048 time)")


if __name__ == "__main__":
    main()
