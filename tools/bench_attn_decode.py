#!/usr/bin/env python3
"""The decode-step attention (one query token per sample over the KV cache; decode.hip) at BASELINE config 5's sizes: effective
KV-cache bandwidth.  MOLLY_DECODE_BLOCKS / MOLLY_DECODE_MIN_KEYS move the split-KV policy.  python tools/bench_attn_decode.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molly_amd import ops  # noqa: E402


def main():
    dev = "cuda"
    for B, nh, nkv, hd, L in [(32, 32, 8, 128, 3100), (32, 32, 8, 128, 1024), (8, 32, 8, 128, 3100), (1, 32, 8, 128, 3100), (32, 16, 8, 128, 2100)]:
        Tmax = L + 64
        g = torch.Generator(device=dev).manual_seed(0)
        # several caches rotated so the reads come from HBM, not the 256 MB Infinity Cache
        nc = max(2, min(6, int(3e9 // (2 * B * Tmax * nkv * hd * 2))))
        kcs = [(torch.rand(B, Tmax, nkv * hd, device=dev, generator=g) - 0.5).bfloat16() for _ in range(nc)]
        vcs = [(torch.rand(B, Tmax, nkv * hd, device=dev, generator=g) - 0.5).bfloat16() for _ in range(nc)]
        q = (torch.rand(B, nh * hd, device=dev, generator=g) - 0.5).bfloat16()
        out = torch.empty(B, nh * hd, dtype=torch.bfloat16, device=dev)
        lo = torch.zeros(B, dtype=torch.int32, device=dev)
        hi = torch.full((B,), L, dtype=torch.int32, device=dev)
        ws = ops.attn_decode_workspace(B, nh, hd, dev)
        run = lambda i: ops.attn_decode(q, kcs[i % nc], vcs[i % nc], out, lo, hi, B, Tmax, nh, nkv, hd, hd ** -0.5, kv_len_hint=L, workspace=ws)
        for i in range(nc):
            run(i)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(4 * nc):
                run(i)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / (4 * nc))
        nbytes = 2 * B * L * nkv * hd * 2
        print(f"B={B:3d} heads {nh}/{nkv} len {L:5d}: {best * 1e3:7.1f} us  {nbytes / best / 1e9:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
