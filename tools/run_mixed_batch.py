#!/usr/bin/env python3
"""One training step of Molly-<size> on a mixed DNA + RNA + protein batch (BASELINE configs 3/4 shapes: all three encoder
paths active) — a functional check at real sizes; prints loss, step time and finite-ness of the gradients."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="4b")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--seq", type=int, default=3072)
    ap.add_argument("--k", type=int, default=512)
    args = ap.parse_args()
    import molly_amd
    from molly_amd import config as C
    from molly_amd.synth import synth_batch
    from molly_amd.trainer import Zero2Optimizer
    cfg = C.molly(args.model, k_tokens=args.k)
    m = molly_amd.OmicsOne(cfg)
    m.model = molly_amd.Qwen3ForCausalLM(cfg.text_config)
    m.dna_rna_model = molly_amd.EsmForMaskedLM(cfg.dna_rna_config)
    m.protein_model = molly_amd.EsmForMaskedLM(cfg.protein_config)
    m.prepare("cuda", random_init_seed=1234)
    opt = Zero2Optimizer(m._rt.P.flat, m._rt.G.flat, m.n_decay, lr=3e-5)
    m.attach_optimizer(opt)
    b = synth_batch(args.batch, args.seq, [("dna", args.k), ("rna", args.k), ("protein", args.k)], seed=3)
    a = [b[k] for k in ("input_ids", "attention_mask", "omic_ids", "omic_info_list", "labels")]
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.time()
        loss = m.forward_backward(*a)
        gn = opt.step(lr=3e-5)
        torch.cuda.synchronize()
        print(f"step {it}: loss {loss.item():.4f} grad-norm {gn.item():.4f} {1e3 * (time.time() - t0):.1f} ms "
              f"finite={bool(torch.isfinite(m._rt.G.flat.float()).all())}", flush=True)


if __name__ == "__main__":
    main()
