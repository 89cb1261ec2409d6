#!/usr/bin/env python3
"""Randomised shapes through every operand form of molly_gemm_bf16 (and the grouped launch) against an fp32 torch product:
edge tiles in M and N, K from one K-tile up, ragged contraction lengths on the both-k-major form, split-K, every epilogue flag.
    python tools/gemm_diag/fuzz_gemm.py [--cases 300] [--seed 0]"""
import argparse
import os
import random
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molly_amd import ops  # noqa: E402
from molly_amd._lib import lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = random.Random(a.seed)
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(a.seed)
    rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
    ops.ensure_gemm_workspace(1 << 28)
    bad = 0
    try:
        bad = _cases(a, rng, rnd, dev)
    finally:
        lib().call("molly_gemm_set_persistent_blocks", 256)
        lib().call("molly_gemm_force_tile", 0)
        lib().call("molly_gemm_ctx_set", None, ops.GEMM_KEYS["dynamic"], 0)
    print(f"{a.cases} cases, {bad} failures")
    sys.exit(1 if bad else 0)


def _cases(a, rng, rnd, dev):
    bad = 0
    for case in range(a.cases):
        form = rng.choice(["nt", "nn", "tn", "tn", "grp"])
        mode = rng.choice([256, 256, 0, -3])
        tile = rng.choice([0, 0, 512])
        lib().call("molly_gemm_set_persistent_blocks", mode)
        lib().call("molly_gemm_force_tile", tile)
        lib().call("molly_gemm_ctx_set", None, ops.GEMM_KEYS["dynamic"], rng.choice([0, 1]) if mode == 256 else 0)
        # (M <= 64 in the nt form: the two decode-row kernels; 8192 x 3000+: more than one round, the dynamic tile fetch)
        M = rng.choice([1, 8, 17, 33, 48, 64, 136, 256, 300, 512, 1000, 1026, 2048, 4096, 8192]) if form != "grp" else rng.choice([256, 512, 768, 1024])
        N = rng.choice([64, 128, 200, 256, 520, 1024, 2048, 3000, 4096, 6144])
        if form in ("nt", "nn"):
            K = 64 * rng.choice([1, 2, 3, 5, 8, 17, 32, 64])
        else:
            K = rng.choice([64, 72, 200, 512, 1000, 1032, 4096, 16384])
        if form == "nt":
            x, w = rnd(M, K), rnd(N, K)
            ref = x.float() @ w.float().t()
            flags = rng.choice(["", "res", "bias", "bias_gelu", "acc"])
            out = rnd(M, N) if flags == "acc" else torch.full((M, N), 7.0, dtype=torch.bfloat16, device=dev)
            bias, res = rnd(N), rnd(M, N)
            if flags == "acc":
                ref = ref + out.float()
            if "bias" in flags:
                ref = ref + bias.float()
            if flags == "bias_gelu":
                ref = torch.nn.functional.gelu(ref)
            if flags == "res":
                ref = ref + res.float()
            ops.gemm_nt(x, w, out=out, bias=bias if "bias" in flags else None, res=res if flags == "res" else None,
                        gelu=flags == "bias_gelu", accumulate=flags == "acc")
        elif form == "nn":
            x, w = rnd(M, K), rnd(K, N)
            ref = x.float() @ w.float()
            out = torch.full((M, N), 7.0, dtype=torch.bfloat16, device=dev)
            ops.gemm(x, w, out=out, b_kmajor=True)
        elif form == "tn":
            M8 = max(8, M // 8 * 8)
            x, w = rnd(K, M8), rnd(K, N // 8 * 8 or 8)
            ref = x.float().t() @ w.float()
            out = torch.full(tuple(ref.shape), 7.0, dtype=torch.bfloat16, device=dev)
            ops.gemm(x, w, out=out, a_kmajor=True, b_kmajor=True)
        else:
            K = 64 * rng.choice([2, 8, 64, 256])
            probs, refs = [], []
            for _ in range(rng.choice([1, 2, 4])):
                m = rng.choice([256, 512, 1024, 264]); n = rng.choice([256, 1024, 2048, 520])
                to = rng.random() < 0.5
                xa, wb = rnd(m, K), rnd(K, n)
                o = torch.full((n, m) if to else (m, n), 7.0, dtype=torch.bfloat16, device=dev)
                probs.append((xa, wb, o, to))
                r = xa.float() @ wb.float()
                refs.append(r.t() if to else r)
            ops.gemm_grouped(probs)
            out = torch.cat([p[2].reshape(-1) for p in probs])
            ref = torch.cat([r.reshape(-1) for r in refs])
        torch.cuda.synchronize()
        err = (out.float() - ref).abs().max().item()
        tol = 2e-2 * ref.abs().max().item() + 1e-2
        if not (err <= tol) or not bool(torch.isfinite(out.float()).all()):
            bad += 1
            print(f"case {case}: {form} M={M} N={N} K={K} mode={mode} tile={tile} cfg={lib().fn['molly_gemm_last_config']()}: max err {err:.4g} > {tol:.4g}")
    return bad


if __name__ == "__main__":
    main()
