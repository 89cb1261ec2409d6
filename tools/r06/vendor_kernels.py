#!/usr/bin/env python3
"""Which hipBLASLt kernels does torch.matmul pick for the step's forward / dgrad shapes?  Run under
`rocprofv3 --kernel-trace --stats` and read the kernel names (Tensile encodes the macro tile, the matrix instruction, the
prefetch depths and the workgroup mapping in them).  Orientation only: nothing of the product path calls torch.matmul."""
import sys
import torch

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
SHAPES = [("qkv fwd", 4096, 2048), ("o fwd", 2048, 2048), ("gate|up fwd", 12288, 2048), ("down fwd", 2048, 6144)]
g = torch.Generator(device="cuda").manual_seed(0)
for name, n, k in SHAPES:
    a = (torch.rand(M, k, device="cuda", generator=g) * 2 - 1).bfloat16()
    w = (torch.rand(n, k, device="cuda", generator=g) * 2 - 1).bfloat16()
    for _ in range(3):
        c = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        c = a @ w.t()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print(f"{name:12s} M {M} N {n} K {k}: {t * 1e3:8.1f} us  {2.0 * M * n * k / t / 1e9:7.0f} TF/s", flush=True)
