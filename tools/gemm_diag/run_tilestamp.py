#!/usr/bin/env python3
"""Where one output tile of the 256x256 GEMM spends its time (tools/build_variant.py tilestamp --patch tilestamp):
    python tools/gemm_diag/run_tilestamp.py"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from molly_amd._lib import MollyLib
VARIANT = sys.argv[1] if len(sys.argv) > 1 else "tilestamp"
L = MollyLib(os.path.join(ROOT, "tools", "variants", "libmolly_" + VARIANT + ".so"))
rd = L.cdll.molly_exp_read_tstamps
rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
st = torch.cuda.current_stream().cuda_stream
names = ["wait K0", "K-tile 0", "K-tile 1", "K-tile 2", "K-tile 3", "middle (each)", "last two (each)", "closing barrier", "prefetch issue",
         "epilogue"]
CASES = (("nt", 16384, 4096, 2048), ("nt", 16384, 4096, 8192), ("nn", 16384, 6144, 2048), ("nt", 16384, 12288, 2048))
if os.environ.get("TILESTAMP_EPILOGUES"):
    # the epilogues of the B = 16 step: plain, + residual (o / down forward), SwiGLU forward (gate|up), SwiGLU backward (down dgrad)
    CASES = (("nt", 32768, 2048, 2048), ("nt+res", 32768, 2048, 2048), ("nt+res", 32768, 2048, 6144),
             ("nn", 32768, 6144, 2048), ("nn+swiglu_bwd", 32768, 6144, 2048))
for form, M, N, K in CASES:
    form, _, epi = form.partition("+")
    a, b = rnd(M, K), (rnd(N, K) if form == "nt" else rnd(K, N))
    flags, res, ldres, ldc = 0, None, 0, N
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    if epi == "res":
        flags, res, ldres = 4, rnd(M, N), N
    elif epi == "swiglu":
        flags, res, ldres = 64, torch.empty(M, N // 2, dtype=torch.bfloat16, device=dev), N // 2
    elif epi == "swiglu_bwd":
        flags, res, ldres, ldc = 128, rnd(M, 2 * N), 2 * N, 2 * N
        out = torch.empty(M, 2 * N, dtype=torch.bfloat16, device=dev)
    f = lambda: L.call("molly_gemm_bf16", st, a, b, out, None, res, M, N, K, K, K if form == "nt" else N, ldc, ldres, flags, 0, 0 if form == "nt" else 1)
    form = form + ("+" + epi if epi else "")
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    rd(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 5
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    buf = np.zeros(2048 * 8 * 16, dtype=np.uint64)
    rd(buf.ctypes.data, 0)
    q = buf.reshape(2048, 8, 16)[:256].astype(np.float64)          # persistent grid: 256 blocks
    nt = q[:, :, 11].mean() / reps                                   # tiles per wave per launch
    nk = K // 64
    tot = q[:, :, :10].sum(axis=2).mean() / reps                     # ticks per wave per launch
    tick_us = us / tot                                               # the waves cover the kernel end to end
    print(f"{form} M={M} N={N} K={K}: {us:.1f} us per launch, {nt:.1f} tiles per CU, {us / nt:.2f} us per tile")
    for gi, gname in ((0, "wave group 0"), (1, "wave group 1")):
        w = q[:, gi * 4:(gi + 1) * 4, :].mean(axis=(0, 1)) / reps / nt * tick_us
        mid = w[5] / max(nk - 6, 1)
        parts = [w[0], w[1], w[2], w[3], w[4], mid, w[6] / 2, w[7], w[8], w[9]]
        print(f"   {gname}: " + "  ".join(f"{n} {v:.2f}" for n, v in zip(names, parts)) + "   (us)")
