#!/usr/bin/env python3
import ctypes, os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = ctypes.CDLL(os.path.join(ROOT, f"tools/gemm_diag/libgemm_{sys.argv[1] if len(sys.argv) > 1 else 'seg'}.so"))
L.molly_gemm_bf16.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 10
L.molly_exp_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
M = 16384
SHAPES = [("gate|up fwd", "nt", M, 12288, 2048), ("down fwd", "nt", M, 2048, 6144),
          ("gate|up dgrad", "nn", M, 2048, 12288), ("down dgrad", "nn", M, 6144, 2048)]
st = torch.cuda.current_stream().cuda_stream
for sched in ([int(x) for x in os.environ.get("SCHEDS", "-1").split(",")]):
  L.molly_gemm_set_schedule(sched)
  for name, form, m, n, k in SHAPES:
    a = rnd(m, k)
    b = rnd(n, k) if form == "nt" else rnd(k, n)
    out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    def run():
        assert L.molly_gemm_bf16(st, a.data_ptr(), b.data_ptr(), out.data_ptr(), None, None, m, n, k, k, k if form == "nt" else n, n, 0, 0,
                               0, 0 if form == "nt" else 1) == 0
    for _ in range(3): run()
    torch.cuda.synchronize()
    L.molly_exp_read_stamps(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    buf = np.zeros(256 * 8 * 20, dtype=np.uint64)
    L.molly_exp_read_stamps(buf.ctypes.data, 1)
    s = buf.reshape(256, 8, 20).astype(np.float64)
    print(f"sched {sched:2d} {name:14s} {form} {e0.elapsed_time(e1)*1e3:6.0f}us")
    for grp, sl in (("g0", slice(0, 4)), ("g1", slice(4, 8))):
        x = s[:, sl, :]
        nk = x[:, :, 16].sum()
        busy = x[:, :, 0:8].sum(axis=(0, 1)) / nk
        bw = x[:, :, 8:16].sum(axis=(0, 1)) / nk
        print(f"   {grp} busy " + " ".join(f"{v:6.0f}" for v in busy) + f" | sum {busy.sum():6.0f}")
        print(f"   {grp} wait " + " ".join(f"{v:6.0f}" for v in bw) + f" | sum {bw.sum():6.0f}")
