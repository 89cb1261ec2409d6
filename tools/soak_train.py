#!/usr/bin/env python3
"""Soak: N optimizer steps of the headline configuration on ONE repeated synthetic batch (the loss must fall monotonically-ish,
stay finite, and the step time must stay flat).  python tools/soak_train.py [--steps 40]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--lr", type=float, default=1e-4)
    a = ap.parse_args()
    import molly_amd
    from molly_amd import config as C
    from molly_amd.synth import synth_batch
    from molly_amd.trainer import Zero2Optimizer
    cfg = C.molly("1.7b", k_tokens=512)
    m = molly_amd.OmicsOne(cfg)
    m.model = molly_amd.Qwen3ForCausalLM(cfg.text_config)
    m.dna_rna_model = molly_amd.EsmForMaskedLM(cfg.dna_rna_config)
    m.protein_model = molly_amd.EsmForMaskedLM(cfg.protein_config)
    m.prepare("cuda", random_init_seed=1234)
    opt = Zero2Optimizer(m._rt.P.flat, m._rt.G.flat, m.n_decay, lr=a.lr)
    m.attach_optimizer(opt)
    batches = [synth_batch(int(os.environ.get("SOAK_BATCH", "16")), 2048, [("protein", 512)], seed=42 + i) for i in range(2)]
    losses, times = [], []
    for s in range(a.steps):
        b = batches[s % 2]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = m.forward_backward(b["input_ids"], b["attention_mask"], b["omic_ids"], b["omic_info_list"], b["labels"])
        gn = opt.step(lr=a.lr)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        losses.append(loss.item())
        if s % 5 == 0 or s == a.steps - 1:
            print(f"step {s:3d} loss {losses[-1]:8.4f} grad-norm {gn.item():8.3f} {times[-1]*1e3:6.1f} ms", flush=True)
    assert all(map(lambda v: v == v and abs(v) < 1e4, losses)), "non-finite loss"
    assert losses[-1] < losses[0] - 1.0, (losses[0], losses[-1])
    print(f"loss {losses[0]:.3f} -> {losses[-1]:.3f}; step time median {sorted(times)[len(times)//2]*1e3:.1f} ms, last {times[-1]*1e3:.1f} ms")


if __name__ == "__main__":
    main()
