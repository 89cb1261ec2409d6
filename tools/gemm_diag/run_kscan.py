#!/usr/bin/env python3
"""Fixed cost per output tile of the 256x256 GEMM: time(K) = a + b*K at constant M, N; a = prologue + epilogue, b = the K-loop.
    python tools/gemm_diag/run_kscan.py [variant ...]       variants = tools/variants/libmolly_<variant>.so (tools/build_variant.py),
                                                              timed interleaved with the in-tree library in one process"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from molly_amd._lib import MollyLib, lib
libs = {"product": lib()}
for v in sys.argv[1:]:
    libs[v] = MollyLib(os.path.join(ROOT, "tools", "variants", f"libmolly_{v}.so"))
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
st = torch.cuda.current_stream().cuda_stream
Ks = (1024, 2048, 4096, 8192)
for form, M, N in (("nt", 16384, 4096), ("nt", 16384, 12288), ("nn", 16384, 2048), ("nn", 16384, 6144)):
    data = {}
    for K in Ks:
        data[K] = (rnd(M, K), rnd(N, K) if form == "nt" else rnd(K, N), torch.empty(M, N, dtype=torch.bfloat16, device=dev))
    ts = {(v, K): 1e9 for v in libs for K in Ks}
    for r in range(5):
        for K in Ks:
            a, b, out = data[K]
            for v, L in libs.items():
                f = lambda: L.call("molly_gemm_bf16", st, a, b, out, None, None, M, N, K, K, K if form == "nt" else N, N, 0, 0, 0,
                                   0 if form == "nt" else 1)
                f()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    f()
                e1.record(); torch.cuda.synchronize()
                ts[(v, K)] = min(ts[(v, K)], e0.elapsed_time(e1) / 3 * 1e3)
    tiles = (M // 256) * (N // 256) / 256.0          # tiles per CU
    for v in libs:
        slope = (ts[(v, 8192)] - ts[(v, 2048)]) / (8192 - 2048)
        icpt = ts[(v, 2048)] - slope * 2048
        print(f"{form} M={M} N={N} {v:12s}: " + "  ".join(f"K={K}: {ts[(v, K)]:7.1f} us {2.0*M*N*K/ts[(v, K)]/1e6:5.0f} TF/s" for K in Ks) +
              f"   | per tile: fixed {icpt / tiles:5.2f} us, per 64-deep K-tile {slope * 64 / tiles:5.3f} us, asymptote {2.0*M*N/slope/1e6:5.0f} TF/s")
