#!/usr/bin/env python3
"""In-process A/B of the whole Molly-1.7B training step under two settings of one library knob (boxes differ by +-2 %, so
variants are only ever compared inside one process, alternating).
    python tools/ab_step.py --knob schedule --a -1 --b 0
    python tools/ab_step.py --knob persistent_blocks --a 256 --b -3        (keys of ops.GEMM_KEYS: the model's own GEMM context)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--knob", required=True)
    ap.add_argument("--a", type=int, required=True)
    ap.add_argument("--b", type=int, required=True)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--set", nargs="*", default=[], help="other keys held fixed for both variants: key=value ...")
    args = ap.parse_args()
    import molly_amd
    from molly_amd import config as C
    from molly_amd._lib import lib
    from molly_amd.synth import synth_batch
    from molly_amd.trainer import Zero2Optimizer
    cfg = C.molly("1.7b", k_tokens=512)
    m = molly_amd.OmicsOne(cfg)
    m.model = molly_amd.Qwen3ForCausalLM(cfg.text_config)
    m.dna_rna_model = molly_amd.EsmForMaskedLM(cfg.dna_rna_config)
    m.protein_model = molly_amd.EsmForMaskedLM(cfg.protein_config)
    m.prepare("cuda", random_init_seed=1234)
    opt = Zero2Optimizer(m._rt.P.flat, m._rt.G.flat, m.n_decay, lr=3e-5)
    m.attach_optimizer(opt)
    b = synth_batch(args.batch, 2048, [("protein", 512)], seed=42)
    a = [b[k] for k in ("input_ids", "attention_mask", "omic_ids", "omic_info_list", "labels")]

    def step():
        m.forward_backward(*a)
        opt.step(lr=3e-5)
    for kv in args.set:
        k, v = kv.split("=")
        m._rt.gemm_ctx.set(k, int(v))
    for _ in range(2):
        step()
    res = {args.a: [], args.b: []}
    for _ in range(args.rounds):
        for v in (args.a, args.b):
            m._rt.gemm_ctx.set(args.knob.replace('molly_gemm_set_', ''), v)
            step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                step()
            e1.record()
            torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / 3)
    for v, t in res.items():
        print(f"{args.knob}({v}): min {min(t):.2f} ms  median {sorted(t)[len(t) // 2]:.2f} ms")


if __name__ == "__main__":
    main()
