#!/usr/bin/env python3
"""Build and run the register-staged one-wave-per-SIMD GEMM experiment (gemm4w_reg.hip) beside the library's kernel and torch.matmul:
    python tools/r06/gemm4w/run.py [--build-only] [M]        (ABL=NOREAD,NOSTAGE,NOBAR: timing-only ablations, comma separated)"""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
SRC = os.environ.get("SRC", "gemm4w_reg.hip")


def so_for(var):
    return os.path.join(HERE, "lib" + SRC[:-4] + ("_" + var.replace(",", "_") if var else "") + ".so")


def build(var):
    so = so_for(var)
    if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(os.path.join(HERE, SRC)):
        return so
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast", "-shared",
           *["-DABL_" + v for v in var.split(",") if v], "-I" + os.path.join(ROOT, "molly_amd/csrc"), "-I" + os.path.join(ROOT, "include"), "-x", "hip",
           os.path.join(HERE, SRC), "-o", so]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-3000:])
    return so


def main():
    variants = [v for v in os.environ.get("ABLS", "").split(";")]          # e.g. ABLS=";NOREAD;NOSTAGE;NOBAR;NOREAD,NOSTAGE"
    sos = {v: build(v) for v in variants}
    if "--build-only" in sys.argv:
        return
    import torch
    sys.path.insert(0, ROOT)
    from molly_amd import ops
    libs = {}
    for v, so in sos.items():
        L = ctypes.CDLL(so)
        L.gemm4w_reg_nt.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 6
        libs[v or "4w"] = L
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
    st = torch.cuda.current_stream().cuda_stream
    M = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 32768
    print(f"{'shape':12s} {'N':>6s} {'K':>6s}   {'gemm256 us/TF':>16s} {'torch us/TF':>16s} " + " ".join(f"{k + ' us/TF':>18s}" for k in libs) + "   bit-identical")
    for name, n, k in (("qkv fwd", 4096, 2048), ("o fwd", 2048, 2048), ("gate|up fwd", 12288, 2048), ("down fwd", 2048, 6144)):
        a, b = rnd(M, k), rnd(n, k)
        ref = ops.gemm_nt(a, b)
        out = torch.zeros_like(ref)
        fns = {"gemm256": lambda: ops.gemm_nt(a, b, out=ref), "torch": lambda: torch.matmul(a, b.t(), out=ref)}
        for key, L in libs.items():
            fns[key] = (lambda L=L: L.gemm4w_reg_nt(st, a.data_ptr(), b.data_ptr(), out.data_ptr(), M, n, k, k, k, n))
        same = None
        if "4w" in libs:
            assert fns["4w"]() == 0
            torch.cuda.synchronize()
            ref2 = ops.gemm_nt(a, b)
            same = bool(torch.equal(out, ref2))
        best = {key: 1e9 for key in fns}
        for _ in range(6):
            for key, fn in fns.items():
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                best[key] = min(best[key], e0.elapsed_time(e1) / 4)
        tf = lambda ms: 2.0 * M * n * k / ms / 1e9
        print(f"{name:12s} {n:6d} {k:6d}   " + " ".join(f"{best[key] * 1e3:9.1f} {tf(best[key]):6.0f}" for key in fns) + f"   {same}", flush=True)


if __name__ == "__main__":
    main()
