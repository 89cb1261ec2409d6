#!/usr/bin/env python3
"""Randomised attention problems (batch, length incl. lengths that are no multiple of any tile, GQA ratios, head dim 64 / 128, causal
or bidirectional, per-sample key ranges as left or right padding) through molly_attn_fwd / molly_attn_bwd against an fp32 torch
softmax-attention with autograd.      python tools/fuzz_attn.py [--cases 60] [--seed 0]"""
import argparse
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molly_amd import ops  # noqa: E402


def ref_attn(q, k, v, B, T, nh, nkv, hd, scale, causal, lo, hi):
    q4 = q.float().reshape(B, T, nh, hd).transpose(1, 2)
    k4 = k.float().reshape(B, T, nkv, hd).transpose(1, 2).repeat_interleave(nh // nkv, 1)
    v4 = v.float().reshape(B, T, nkv, hd).transpose(1, 2).repeat_interleave(nh // nkv, 1)
    s = (q4 @ k4.transpose(-1, -2)) * scale
    idx = torch.arange(T, device=q.device)
    ok = (idx[None, :] >= lo[:, None]) & (idx[None, :] < hi[:, None])             # [B, Tk]
    mask = ok[:, None, None, :].expand(B, nh, T, T)
    if causal:
        mask = mask & (idx[None, None, None, :] <= idx[None, None, :, None])
    s = s.masked_fill(~mask, float("-inf"))
    p = torch.softmax(s, -1)
    p = torch.nan_to_num(p, nan=0.0)                                             # rows without a live key -> 0 (the kernel's convention)
    return (p @ v4).transpose(1, 2).reshape(B * T, nh * hd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = random.Random(a.seed)
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(a.seed)
    bad = 0
    for case in range(a.cases):
        hd = rng.choice([64, 128])
        nkv = rng.choice([1, 2, 4])
        nh = nkv * rng.choice([1, 2, 4])
        B = rng.choice([1, 2, 3])
        T = rng.choice([17, 64, 100, 128, 200, 257, 384, 520, 1000])
        causal = rng.random() < 0.6
        pad = rng.choice(["none", "right", "left"])
        lo = torch.zeros(B, dtype=torch.int32, device=dev)
        hi = torch.full((B,), T, dtype=torch.int32, device=dev)
        if pad == "right":
            hi = torch.tensor([rng.randint(max(1, T // 3), T) for _ in range(B)], dtype=torch.int32, device=dev)
        elif pad == "left":
            lo = torch.tensor([rng.randint(0, T - max(1, T // 3)) for _ in range(B)], dtype=torch.int32, device=dev)
        M = B * T
        qkv = (torch.randn(M, (nh + 2 * nkv) * hd, device=dev, generator=g) * 0.7).bfloat16()
        q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
        scale = hd ** -0.5
        o, lse = ops.attn_fwd(q, k, v, B, T, nh, nkv, hd, scale, causal, lo if pad != "none" else None, hi if pad != "none" else None)
        do = (torch.randn(M, nh * hd, device=dev, generator=g) * 0.5).bfloat16()
        dqkv = torch.full_like(qkv, 7.0)
        dq, dk, dv = dqkv[:, :nh * hd], dqkv[:, nh * hd:(nh + nkv) * hd], dqkv[:, (nh + nkv) * hd:]
        ops.attn_bwd(q, k, v, o, do, lse, B, T, nh, nkv, hd, scale, causal, dq, dk, dv, lo if pad != "none" else None, hi if pad != "none" else None)
        torch.cuda.synchronize()
        qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
        ref = ref_attn(qr, kr, vr, B, T, nh, nkv, hd, scale, causal, lo.long(), hi.long())
        # rows that are padding themselves (query outside the key range) are don't-care for the caller: compare live rows
        idx = torch.arange(T, device=dev)
        live = ((idx[None, :] >= lo[:, None]) & (idx[None, :] < hi[:, None])).reshape(-1)
        ref.backward(do.float() * live[:, None].float())
        # the kernel gets dO on every row; zero it on dead rows the same way for a like-for-like comparison
        do2 = (do.float() * live[:, None].float()).bfloat16()
        ops.attn_bwd(q, k, v, o, do2, lse, B, T, nh, nkv, hd, scale, causal, dq, dk, dv, lo if pad != "none" else None, hi if pad != "none" else None)
        torch.cuda.synchronize()
        errs = {"o": ((o.float() - ref.detach())[live].abs().max().item(), ref.detach()[live].abs().max().item()),
                "dq": ((dq.float() - qr.grad)[live].abs().max().item(), qr.grad.abs().max().item()),
                "dk": ((dk.float() - kr.grad).abs().max().item(), kr.grad.abs().max().item()),
                "dv": ((dv.float() - vr.grad).abs().max().item(), vr.grad.abs().max().item())}
        fail = [n for n, (e, m) in errs.items() if not (e <= 3e-2 * m + 2e-3)]
        if fail or not bool(torch.isfinite(dqkv.float()).all()):
            bad += 1
            print(f"case {case}: B={B} T={T} nh={nh} nkv={nkv} hd={hd} causal={causal} pad={pad}: " +
                  " ".join(f"{n} {e:.3g}/{m:.3g}" for n, (e, m) in errs.items()), "FAIL", fail)
    print(f"{a.cases} cases, {bad} failures")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
