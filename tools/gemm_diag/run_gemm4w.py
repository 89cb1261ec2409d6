#!/usr/bin/env python3
"""Build and run the one-wave-per-SIMD GEMM experiment (tools/gemm_diag/gemm4w.hip) beside the library's 8-wave kernel:
python tools/gemm_diag/run_gemm4w.py [--build-only]"""
import ctypes
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
VAR = os.environ.get("ABL", "")                       # timing-only ablations: NODMA, NOREAD, NOBAR (comma separated)
SO = os.path.join(HERE, "libgemm4w" + ("_" + VAR.replace(",", "_") if VAR else "") + ".so")


def build():
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast", "-shared",
           *["-DABL_" + v for v in VAR.split(",") if v], "-I" + os.path.join(ROOT, "molly_amd/csrc"), "-I" + os.path.join(ROOT, "include"), "-x", "hip",
           os.path.join(HERE, "gemm4w.hip"), "-o", SO]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-3000:])


def main():
    if not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(HERE, "gemm4w.hip")):
        build()
    if "--build-only" in sys.argv:
        return
    import torch
    sys.path.insert(0, ROOT)
    from molly_amd import ops
    L = ctypes.CDLL(SO)
    L.gemm4w_nt.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 6
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
    st = torch.cuda.current_stream().cuda_stream
    M = 16384
    for name, n, k in (("qkv fwd", 4096, 2048), ("o fwd", 2048, 2048), ("gate|up fwd", 12288, 2048), ("down fwd", 2048, 6144)):
        a, b = rnd(M, k), rnd(n, k)
        ref = ops.gemm_nt(a, b)
        out = torch.empty_like(ref)
        run4 = lambda: L.gemm4w_nt(st, a.data_ptr(), b.data_ptr(), out.data_ptr(), M, n, k, k, k, n)
        assert run4() == 0
        torch.cuda.synchronize()
        same = torch.equal(out, ref)
        err = (out.float() - ref.float()).abs().max().item()
        best = {"8w": 1e9, "4w": 1e9}
        for _ in range(5):
            for key, fn in (("8w", lambda: ops.gemm_nt(a, b, out=ref)), ("4w", run4)):
                fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                best[key] = min(best[key], e0.elapsed_time(e1) / 3)
        tf = lambda ms: 2.0 * M * n * k / ms / 1e9
        print(f"{name:12s} N={n:6d} K={k:5d}  8-wave {best['8w']*1e3:7.1f} us {tf(best['8w']):5.0f} TF/s   4-wave {best['4w']*1e3:7.1f} us "
              f"{tf(best['4w']):5.0f} TF/s   bit-identical {same}  max|d| {err:.3g}")


if __name__ == "__main__":
    main()
