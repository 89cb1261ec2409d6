#!/usr/bin/env python3
"""Step time of the Enc-Head baselines at the encoders' real sizes, random weights:
python tools/bench_baseline.py [--type ESM|NT|NT+ESM|ESM+ESM|NT+NT] [--batch 8] [--len 512] [--freeze]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--type", default="ESM")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--len", type=int, default=512)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--freeze", action="store_true")
    a = ap.parse_args()
    from molly_amd import config as C
    from molly_amd.baselines import BackboneWithClsHead
    from molly_amd.trainer import Zero2Optimizer
    m = BackboneWithClsHead(a.type, nt_model=C.nt_500m_human_ref(), esm_model=C.esm2_650m(), num_labels=2)
    if a.freeze:
        m.freeze_backbone()
    m.prepare("cuda", random_init_seed=0)
    opt = Zero2Optimizer(m._rt.P.flat, m._rt.G.flat, m.n_decay, lr=1e-5)
    g = torch.Generator().manual_seed(0)
    n_in = len(m.part_names)
    xs = []
    for eng in m._rt.eng:
        x = torch.randint(4, min(eng.cfg.vocab_size, 24), (a.batch, a.len), generator=g)
        x[:, 0] = 0
        xs.append(x.cuda())
    labels = torch.randint(0, 2, (a.batch,), generator=g).cuda()
    args = (xs[0], xs[1] if n_in > 1 else None, None, None)
    for i in range(a.steps + 2):
        if i == 2:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        loss = m.forward_backward(*args, labels=labels)
        opt.step(lr=1e-5)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    toks = a.batch * a.len * n_in
    print(f"{a.type}{' frozen' if a.freeze else ''}: B={a.batch} L={a.len}  {dt * 1e3:.1f} ms/step  {toks / dt:,.0f} encoder tokens/s  "
          f"loss {loss.item():.4f}  trainable {m._rt.P.numel / 1e6:.0f} M")


if __name__ == "__main__":
    main()
