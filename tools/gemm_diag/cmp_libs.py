#!/usr/bin/env python3
"""Bitwise comparison of two builds of the library over GEMM shapes / flags (in-tree vs tools/variants/libmolly_<v>.so)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from molly_amd._lib import MollyLib, lib
A, Bv = lib(), MollyLib(os.path.join(ROOT, "tools", "variants", f"libmolly_{sys.argv[1]}.so"), strict=False)
# second argument 0: stream-K off in the in-tree library (then every case must be bit-identical to a build that has no stream-K);
# default 1: the grids that take the stream-K launch are allowed the fp32 reordering of their K-range sums and are reported apart
SK = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if "molly_gemm_set_streamk" in A.fn:
    A.call("molly_gemm_set_streamk", SK)
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
st = torch.cuda.current_stream().cuda_stream
ws = torch.empty(1 << 28, dtype=torch.float32, device=dev)
for L in (A, Bv):
    L.call("molly_gemm_set_workspace", ws, ws.numel() * 4)
bad = 0
cases = []
for M in (4096, 1026, 300, 16384, 2048):
    for (N, K) in ((6144, 4096), (4096, 4096), (4096, 12288), (2048, 2048), (1280, 1280), (5120, 1280), (1280, 5120), (24576, 4096), (1000, 512)):
        for form in ("nt", "nn"):
            for flags in (0, 4, 1, 3, 5, 8):
                cases.append((M, N, K, form, flags))
for (M, N, K, form, flags) in cases:
    if M * N > 16384 * 8192:
        continue
    a = rnd(M, K)
    b = rnd(N, K) if form == "nt" else rnd(K, N)
    bias = rnd(N)
    res = rnd(M, N)
    outs = []
    for L in (A, Bv):
        out = rnd(M, N) if flags & 8 else torch.full((M, N), 7.0, dtype=torch.bfloat16, device=dev)
        if flags & 8:
            out.copy_(res)
        L.call("molly_gemm_bf16", st, a, b, out, bias, res, M, N, K, K, K if form == "nt" else N, N, N, flags, 0, 0 if form == "nt" else 1)
        outs.append(out)
    torch.cuda.synchronize()
    if not torch.equal(outs[0], outs[1]):
        d = (outs[0].float() - outs[1].float()).abs()
        if A.fn['molly_gemm_last_config']() // 1000 >= 50 and d.max().item() <= 2 ** -7 * outs[1].float().abs().max().item():
            nsk = globals().get("nsk", 0) + 1
            continue
        rows = (d.max(dim=1).values > 0).nonzero().flatten()
        cols = (d.max(dim=0).values > 0).nonzero().flatten()
        print(f"MISMATCH M={M} N={N} K={K} {form} flags={flags}: max {d.max().item():.3g}, rows {rows[:4].tolist()}..{rows[-1].item()} ({len(rows)}), cols {cols[:4].tolist()}..{cols[-1].item()} ({len(cols)}), cfg {A.fn['molly_gemm_last_config']()}")
        bad += 1
print("cases", len(cases), "mismatches", bad, "stream-K cases within one bf16 step of the one-pass result:", globals().get("nsk", 0))
