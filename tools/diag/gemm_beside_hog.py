#!/usr/bin/env python3
"""The 256x256 GEMM's launch shapes beside a kernel that holds some CUs (what a collective's kernel does at N > 1): one block per
CU walking a static list (persistent, 256), one block per tile (0), blocks of at most t tiles in whole rounds (-t), and 256 blocks
that DRAW their tiles (dyn: MOLLY_GEMM_KEY_DYNAMIC).  The hog (molly_probe_hog) owns
`--cus` CUs for the whole measurement on a second stream; the GEMM is timed on the main stream.
    python tools/diag/gemm_beside_hog.py [--cus 0 16 32 64]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molly_amd import ops  # noqa: E402
from molly_amd._lib import lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cus", type=int, nargs="+", default=[0, 16, 32, 64])
    ap.add_argument("--modes", nargs="+", default=["256", "0", "-3", "dyn"])
    ap.add_argument("--shapes", default="b8", help="b8: Molly-1.7B at 8 x 2048 tokens (the headline); b1: Molly-4B / 8B at one sample per GPU "
                                                   "(BASELINE configs 3 / 4: what the 8-GPU runs launch)")
    a = ap.parse_args()
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) * 2 - 1).bfloat16()
    shapes = [("qkv fwd", "nt", 16384, 4096, 2048), ("gate|up fwd", "nt", 16384, 12288, 2048), ("down dgrad", "nn", 16384, 6144, 2048),
              ("gate|up dgrad", "nn", 16384, 2048, 12288), ("8b qkv B=1 (stream-K)", "nt", 4096, 6144, 4096)]
    if a.shapes == "b1":
        shapes = [("4b qkv fwd", "nt", 3072, 6144, 2560), ("4b gate|up fwd", "nt", 3072, 19456, 2560), ("4b down fwd", "nt", 3072, 2560, 9728),
                  ("4b down dgrad", "nn", 3072, 9728, 2560), ("8b gate|up fwd", "nt", 4096, 24576, 4096), ("8b o fwd", "nt", 4096, 4096, 4096),
                  ("8b down dgrad", "nn", 4096, 12288, 4096)]
    ctx = ops.GemmContext()
    ctx.ensure_workspace(1 << 28)
    side = torch.cuda.Stream()
    sink = torch.zeros(4, dtype=torch.int32, device=dev)
    print(f"{'shape':22s} {'CUs held':>8s} " + " ".join(f"{('mode ' + str(m)):>10s}" for m in a.modes) + "   (TF/s)")
    with ops.use_gemm_context(ctx):
      for name, form, M, N, K in shapes:
          x = rnd(M, K)
          w = rnd(N, K) if form == "nt" else rnd(K, N)
          out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
          run = (lambda: ops.gemm_nt(x, w, out=out)) if form == "nt" else (lambda: ops.gemm(x, w, out=out, b_kmajor=True))
          for cus in a.cus:
              row = []
              for mode in a.modes:
                  ctx.set("persistent_blocks", 256 if mode == "dyn" else int(mode))
                  ctx.set("dynamic", 1 if mode == "dyn" else 0)
                  best = 1e9
                  for _ in range(3):
                      run()
                      torch.cuda.synchronize()
                      if cus:
                          with torch.cuda.stream(side):
                              lib().call("molly_probe_hog", side.cuda_stream, cus, 20000, sink)      # 20 ms: outlives the timed launches
                          torch.cuda._sleep(200000)                                                # let the hog's blocks land first
                      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                      e0.record()
                      for _ in range(5):
                          run()
                      e1.record()
                      e1.synchronize()
                      best = min(best, e0.elapsed_time(e1) / 5)
                      torch.cuda.synchronize()
                  row.append(2.0 * M * N * K / best / 1e9)
              print(f"{name:22s} {cus:8d} " + " ".join(f"{v:10.0f}" for v in row))


if __name__ == "__main__":
    main()
