#!/usr/bin/env python3
"""The grouped weight-gradient launch of one decoder layer at config 3's sizes (Qwen3-4B, 3072 rows) and at the headline's (1.7B, 32768 rows),
with and without MOLLY_GEMM_ACCUMULATE (the second micro-batch of a GA = 2 step accumulates): us and TFLOP/s.
    python tools/r05/bench_grouped_c3.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molly_amd import ops  # noqa: E402
from molly_amd.qwen3 import _carve_remainder  # noqa: E402

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()


def layer(M, h, nqkv, nq, ff):
    # (operand^T [narrow][M], other [M][wide], out, transposed-out) as qwen3._wgrad_layer queues them
    probs = []
    for n_out, k_in in ((nqkv, h), (h, nq), (2 * ff, h), (h, ff)):
        if k_in <= n_out:
            probs.append((rnd(k_in, M), rnd(M, n_out), torch.zeros(n_out, k_in, dtype=torch.bfloat16, device=dev), True))
        else:
            probs.append((rnd(n_out, M), rnd(M, k_in), torch.zeros(n_out, k_in, dtype=torch.bfloat16, device=dev), False))
    return probs


def timed(f, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


ops.ensure_gemm_workspace(1 << 30)
for name, M, h, nqkv, nq, ff in (("Qwen3-4B, 3072 rows", 3072, 2560, 6144, 4096, 9728), ("Qwen3-8B, 4096 rows", 4096, 4096, 6144, 4096, 12288),
                                 ("Qwen3-1.7B, 32768 rows", 32768, 2048, 4096, 2048, 6144)):
    probs = layer(M, h, nqkv, nq, ff)
    fl = sum(2.0 * a.shape[0] * b.shape[1] * M for a, b, _, _ in probs)
    grouped, carved = _carve_remainder(probs)
    for acc in (False, True):
        us = timed(lambda: ops.gemm_grouped(grouped, accumulate=acc))
        flg = sum(2.0 * a.shape[0] * b.shape[1] * M for a, b, _, _ in grouped)
        print(f"{name}: grouped launch of {len(grouped)} problems{' (+ a carved one)' if carved else ''}, accumulate={acc}: {us:8.1f} us  {flg / us / 1e6:7.0f} TFLOP/s")
