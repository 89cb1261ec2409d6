#!/usr/bin/env python3
"""The sampling launch of a decode step (reference inference settings: temperature 0.8, top-k 20, top-p 0.95, repetition penalty 1.1) and the greedy
argmax on [32, 151,936] fp32 logits."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from molly_amd import ops  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
lg = [torch.randn(32, 151936, device="cuda", generator=g) * 3 for _ in range(4)]
gen = torch.randint(0, 151936, (32, 200), device="cuda", generator=g)


def timeit(fn, n=20):
    for _ in range(3):
        fn(0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print(f"sample_logits (penalty over 200 generated tokens, temperature, top-k 20, top-p 0.95): {timeit(lambda i: ops.sample_logits(lg[i % 4], gen, 1.1, 0.8, 20, 0.95, 1234, i)):.1f} us")
print(f"argmax: {timeit(lambda i: ops.argmax(lg[i % 4])):.1f} us")
