#!/usr/bin/env python3
"""Probe: do HBM-bound elementwise kernels overlap with the persistent 256x256 GEMM when launched on another stream?
(premise of running a layer's weight-gradient GEMMs on a side stream under the next layer's backward chain)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molly_amd import ops
from molly_amd._lib import lib

M, h, ff = 16384, 2048, 6144
g = torch.Generator(device="cuda").manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).bfloat16()
xT, dy, dw = rnd(h, M), rnd(M, 2 * ff), torch.empty(2 * ff, h, dtype=torch.bfloat16, device="cuda")      # gate|up wgrad (trans_out)
gu, dact, dgu = rnd(M, 2 * ff), rnd(M, ff), torch.empty(M, 2 * ff, dtype=torch.bfloat16, device="cuda")
x, w, gg, dx = rnd(M, h), rnd(h), rnd(M, h), torch.empty(M, h, dtype=torch.bfloat16, device="cuda")
dwn = torch.zeros(h, dtype=torch.bfloat16, device="cuda")
ops.ensure_gemm_workspace(1 << 30)
side = torch.cuda.Stream(priority=0)
low = torch.cuda.Stream(priority=0)

def gemm():
    ops.gemm(xT, dy, out=dw, b_kmajor=True, trans_out=True)

def ew():
    for _ in range(3):
        ops.swiglu_bwd(gu, dact, dgu)
        ops.rmsnorm_bwd(x, w, gg, dwn, 1e-6, dres=gg, dx=dx)

def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best

def both():
    ev = torch.cuda.Event(); ev.record()
    side.wait_event(ev)
    with torch.cuda.stream(side):
        gemm()
    ew()
    torch.cuda.current_stream().wait_stream(side)

for pb in (256, 0):
    lib().call("molly_gemm_set_persistent_blocks", pb)
    tg, te, tb = timed(gemm), timed(ew), timed(both)
    print(f"persistent_blocks={pb}: gemm {tg*1e3:.0f} us, elementwise {te*1e3:.0f} us, serial {1e3*(tg+te):.0f} us, concurrent {tb*1e3:.0f} us "
          f"(hidden {1e3*(tg+te-tb):.0f} us)")
lib().call("molly_gemm_set_persistent_blocks", 256)
