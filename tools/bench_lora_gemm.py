#!/usr/bin/env python3
"""The six rank-r GEMM shapes of one LoRA target (M = 16384 tokens, r = 64) as the engine issues them: time, config, and the
bytes of the big operand per second.  python tools/bench_lora_gemm.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molly_amd import ops  # noqa: E402
from molly_amd._lib import lib  # noqa: E402


def timeit(fn, n=10):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev, M, r = "cuda", 16384, 64
    ops.ensure_gemm_workspace(1 << 30)
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
    tT = torch.empty(2048 * M, dtype=torch.bfloat16, device=dev)
    for fin, fout in ((2048, 2048), (2048, 6144), (6144, 2048)):
        x, dy, A, B = rnd(M, fin), rnd(M, fout), rnd(r, fin), rnd(fout, r)
        t, dt = rnd(M, r), rnd(M, r)
        y, dx = rnd(M, fout), rnd(M, fin)
        dA, dB = torch.empty(r, fin, dtype=torch.bfloat16, device=dev), torch.empty(fout, r, dtype=torch.bfloat16, device=dev)
        tt = tT[:r * M].view(r, M)
        cases = [
            ("t = x A^T", lambda: ops.gemm_nt(x, A, out=t), x.numel() * 2),
            ("y += t B^T", lambda: ops.gemm_nt(t, B, out=y, accumulate=True), 2 * y.numel() * 2),
            ("dB = dy^T t (transpose t + TO gemm)", lambda: (ops.transpose(t, tt), ops.gemm(tt, dy, out=dB, b_kmajor=True, trans_out=True)), dy.numel() * 2),
            ("dt = dy B", lambda: ops.gemm(dy, B, out=dt, b_kmajor=True), dy.numel() * 2),
            ("dA = dt^T x (transpose dt + gemm)", lambda: (ops.transpose(dt, tt), ops.gemm(tt, x, out=dA, b_kmajor=True)), x.numel() * 2),
            ("dx += dt A", lambda: ops.gemm(dt, A, out=dx, accumulate=True, b_kmajor=True), 2 * dx.numel() * 2),
        ]
        print(f"--- in={fin} out={fout}")
        for name, fn, nbytes in cases:
            us = timeit(fn)
            print(f"{name:40s} {us:7.1f} us   {nbytes / us / 1e6:5.2f} TB/s   cfg {lib().query('molly_gemm_last_config')}")


if __name__ == "__main__":
    main()
