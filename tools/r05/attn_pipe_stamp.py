#!/usr/bin/env python3
"""Where a wave of attn_fwd_pipe_kernel spends the cycles of a FAST tile (python tools/build_variant.py attnstamp -DMOLLY_ATTN_STAMP=1 first):
s_memtime laps of the two steps' phases, the wait for the LDS-DMA, the barrier and the DMA issue."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from molly_amd._lib import MollyLib
L = MollyLib(os.path.join(ROOT, "tools", "variants", "libmolly_attnstamp.so"))
rd = L.cdll.molly_exp_attn_stamps
rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
NW = int(os.environ.get("MOLLY_ATTN_FWD_PIPE", "8"))
causal = int(os.environ.get("ATTN_CAUSAL", "1"))
B, T, nh, nkv, hd = (int(x) for x in os.environ.get("ATTN_SHAPE", "8,2048,16,8,128").split(","))
M = B * T
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(M, (nh + 2 * nkv) * hd, device="cuda", generator=g).bfloat16()
q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
o = torch.empty(M, nh * hd, dtype=torch.bfloat16, device="cuda")
lse = torch.empty(B, nh, T, dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
ld = qkv.stride(0)
f = lambda: L.call("molly_attn_fwd", st, q, k, v, o, lse, None, None, B, T, nh, nkv, hd, ld, ld, ld, nh * hd, hd ** -0.5, causal)
for _ in range(3):
    f()
torch.cuda.synchronize()
rd(None, 1)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 5
e0.record()
for _ in range(reps):
    f()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
buf = np.zeros(32768 * 4 * 8, dtype=np.uint64)
rd(buf.ctypes.data, 0)
nblk = nh * B * ((T + 255) // 256)          # 256 query rows per workgroup in both forms (8 waves x 32, 4 waves x 64)
a = buf[:nblk * NW * 8].reshape(nblk, NW, 8).astype(np.float64) / reps
tiles = a[:, :, 7].sum()
tot = a[:, :, :7].sum()
names = ["step 0 phase A (S^T MFMAs | exp, sum, pack)", "step 0 phase B (P.V MFMAs | row max)", "step 1 phase A", "wait: own LDS-DMA landed + own reads returned",
         "workgroup barrier", "LDS-DMA issue (K(t+3), V(t+2))", "step 1 phase B"]
print(f"NW {NW} causal {causal} shape {B},{T},{nh},{nkv},{hd}: {us:.1f} us per launch (stamped build), {tiles:.0f} fast tiles x waves, {tot / tiles:.0f} ticks per fast tile per wave"
      f" (32 MFMAs = 1024 matrix-pipe cycles per wave, 2048 per SIMD)")
for i, n in enumerate(names):
    print(f"   {n:58s} {a[:, :, i].sum() / tiles:8.1f} ticks per tile  ({100 * a[:, :, i].sum() / tot:5.1f} %)")
for w in range(NW):
    tw = a[:, w, 7].sum()
    print(f"   wave {w}: " + " ".join(f"{a[:, w, i].sum() / max(tw, 1):7.1f}" for i in range(7)))
