#!/usr/bin/env python3
"""Where a wave of the attention forward spends its cycles (tools/build_variant.py attnstamp -DMOLLY_ATTN_STAMP=1):
s_memtime laps per segment of the half-step, at two workgroups per CU and at one (MOLLY_ATTN_LDS_PAD=40960)."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from molly_amd._lib import MollyLib
L = MollyLib(os.path.join(ROOT, "tools", "variants", "libmolly_attnstamp.so"))
rd = L.cdll.molly_exp_attn_stamps
rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
B, T, nh, nkv, hd = 8, 2048, 16, 8, 128
M = B * T
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(M, (nh + 2 * nkv) * hd, device="cuda", generator=g).bfloat16()
q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
o = torch.empty(M, nh * hd, dtype=torch.bfloat16, device="cuda")
lse = torch.empty(B, nh, T, dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
ld = qkv.stride(0)
f = lambda: L.call("molly_attn_fwd", st, q, k, v, o, lse, None, None, B, T, nh, nkv, hd, ld, ld, ld, nh * hd, hd ** -0.5, 1)
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
nblk = nh * B * (T // 128)
a = buf.reshape(32768, 4, 8)[:nblk].astype(np.float64) / reps
halves = a[:, :, 7].sum()
tot = a[:, :, :6].sum()
names = ["loop overhead + next tile's LDS-DMA issue", "K fragment reads + 8 S^T MFMAs (to completion)", "softmax: max, rescale test, 16 exp2, row sum, bf16 pack",
         "8 P.V MFMAs (issue) + V fragment waits", "wait for next tile + workgroup barrier", "epilogue (O rows, LSE)"]
print(f"{os.environ.get('MOLLY_ATTN_LDS_PAD', '0')} B of LDS pad: {us:.1f} us per launch (stamped build), {halves:.0f} half-steps in all, "
      f"{tot / halves:.0f} ticks per half-step per wave")
for i, n in enumerate(names):
    print(f"   {n:58s} {a[:, :, i].sum() / halves:8.1f} ticks per half-step  ({100 * a[:, :, i].sum() / tot:5.1f} %)")
# ticks -> time: the waves of a CU cover the launch end to end when both workgroup slots stay full
print(f"   (one tick = one s_memtime count; 512 matrix-pipe cycles per half-step and wave would be {512:.0f} core cycles)")
