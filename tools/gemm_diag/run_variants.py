#!/usr/bin/env python3
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
names = sys.argv[1:] or ["base", "nodma", "noread", "nomma", "nodma_noread", "nobar"]
libs = {}
for n in names:
    L = ctypes.CDLL(os.path.join(ROOT, f"tools/gemm_diag/libgemm_{n}.so"))
    L.molly_gemm_bf16.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 10
    L.molly_gemm_bf16.restype = ctypes.c_int
    if os.environ.get("SCHED"):
        L.molly_gemm_set_schedule(int(os.environ["SCHED"]))
    libs[n] = L
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
M = 16384
SHAPES = [("gate|up fwd", "nt", M, 12288, 2048), ("down fwd", "nt", M, 2048, 6144), ("qkv fwd", "nt", M, 4096, 2048),
          ("gate|up dgrad", "nn", M, 2048, 12288), ("down dgrad", "nn", M, 6144, 2048)]
st = torch.cuda.current_stream().cuda_stream
print(f"{'shape':16s} {'form':4s} " + " ".join(f"{n:>13s}" for n in names))
for name, form, m, n, k in SHAPES:
    a = rnd(m, k)
    b = rnd(n, k) if form == "nt" else rnd(k, n)
    out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    def run(L):
        rc = L.molly_gemm_bf16(st, a.data_ptr(), b.data_ptr(), out.data_ptr(), None, None, m, n, k, k, k if form == "nt" else n, n, 0, 0,
                               0, 0 if form == "nt" else 1)
        assert rc == 0
    best = {x: 1e9 for x in names}
    for r in range(5):
        for x in names:
            run(libs[x])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                run(libs[x])
            e1.record(); torch.cuda.synchronize()
            best[x] = min(best[x], e0.elapsed_time(e1) / 3)
    print(f"{name:16s} {form:4s} " + " ".join(f"{best[x]*1e3:7.0f}us{2.0*m*n*k/best[x]/1e9:5.0f}" for x in names))
