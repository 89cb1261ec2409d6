#!/usr/bin/env python3
"""Attention micro-benchmark at Molly-1.7B's shape (B=8, T=2048, 16 q heads / 8 kv heads x 128), random data."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molly_amd import ops  # noqa: E402


def timeit(fn, n=5):
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
    import os
    B, T, nh, nkv, hd = (int(x) for x in os.environ.get("ATTN_SHAPE", "8,2048,16,8,128").split(","))   # e.g. 16,2048,64,8,128 (the guide's)
    M = B * T
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(M, (nh + 2 * nkv) * hd, device="cuda", generator=g).bfloat16()
    q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
    o = torch.empty(M, nh * hd, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B, nh, T, dtype=torch.float32, device="cuda")
    do = torch.randn(M, nh * hd, device="cuda", generator=g).bfloat16()
    dqkv = torch.empty_like(qkv)
    dq, dk, dv = dqkv[:, :nh * hd], dqkv[:, nh * hd:(nh + nkv) * hd], dqkv[:, (nh + nkv) * hd:]
    delta = torch.empty(B, nh, T, dtype=torch.float32, device="cuda")
    sc = hd ** -0.5
    fwd_fl = 4.0 * B * nh * T * T * hd / 2
    t = timeit(lambda: ops.attn_fwd(q, k, v, B, T, nh, nkv, hd, sc, True, out=o, lse=lse))
    print(f"attn fwd causal  : {t * 1e3:8.1f} us  {fwd_fl / t / 1e9:7.1f} TF/s (algorithmic 4*B*nh*T^2*hd/2)")
    t = timeit(lambda: ops.attn_bwd(q, k, v, o, do, lse, B, T, nh, nkv, hd, sc, True, dq, dk, dv, delta_ws=delta))
    print(f"attn bwd causal  : {t * 1e3:8.1f} us  {2.5 * fwd_fl / t / 1e9:7.1f} TF/s (algorithmic 2.5x fwd)")
    # ESM shape: 8 x 512, 20 heads x 64, bidirectional
    B2, T2, nh2, hd2 = 8, 512, 20, 64
    qkv2 = torch.randn(B2 * T2, 3 * nh2 * hd2, device="cuda", generator=g).bfloat16()
    o2 = torch.empty(B2 * T2, nh2 * hd2, dtype=torch.bfloat16, device="cuda")
    t = timeit(lambda: ops.attn_fwd(qkv2[:, :1280], qkv2[:, 1280:2560], qkv2[:, 2560:], B2, T2, nh2, nh2, hd2, 1.0, False,
                                    out=o2, lse=False))
    print(f"esm attn fwd     : {t * 1e3:8.1f} us  {4.0 * B2 * nh2 * T2 * T2 * hd2 / t / 1e9:7.1f} TF/s")


    if "--torch" in sys.argv:
        # orientation only: torch's fused SDPA (whatever backend this build ships) on the same problem
        import torch.nn.functional as F
        q4 = q.reshape(B, T, nh, hd).transpose(1, 2).contiguous().requires_grad_(True)
        k4 = k.reshape(B, T, nkv, hd).transpose(1, 2).contiguous().requires_grad_(True)
        v4 = v.reshape(B, T, nkv, hd).transpose(1, 2).contiguous().requires_grad_(True)
        try:
            f = lambda: F.scaled_dot_product_attention(q4, k4, v4, is_causal=True, enable_gqa=True)
            with torch.no_grad():
                t = timeit(f)
            print(f"torch sdpa fwd   : {t * 1e3:8.1f} us  {fwd_fl / t / 1e9:7.1f} TF/s")
            out = f()
            g4 = torch.randn_like(out)
            t = timeit(lambda: torch.autograd.grad(f(), (q4, k4, v4), g4))
            print(f"torch sdpa fwd+bwd: {t * 1e3:8.1f} us  {3.5 * fwd_fl / t / 1e9:7.1f} TF/s")
        except Exception as e:      # noqa: BLE001
            print("torch sdpa unavailable:", type(e).__name__, str(e)[:200])


if __name__ == "__main__":
    main()
