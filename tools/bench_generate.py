#!/usr/bin/env python3
"""Prefill + greedy-decode throughput of `OmicsOne.generate`'s engine (BASELINE config 5 shape: left-padded prompts with one
protein span, KV-cache decode).  Random-init weights, synthetic prompts; times the GenerationSession directly with HIP events.
    python tools/bench_generate.py --model 8b --batch 32 --prompt 3072 --new 64"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="1.7b")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--prompt", type=int, default=3072)
    ap.add_argument("--k-protein", type=int, default=1024)
    ap.add_argument("--new", type=int, default=64)
    ap.add_argument("--api", action="store_true", help="also time OmicsOne.generate(do_sample=True, ...) end to end")
    args = ap.parse_args()
    import molly_amd
    from molly_amd import config as C, ops
    from molly_amd.generate import GenerationSession
    from molly_amd.synth import synth_batch
    dev = torch.device("cuda", 0)
    cfg = C.molly(args.model, k_tokens=args.k_protein)
    m = molly_amd.OmicsOne(cfg)
    m.model = molly_amd.Qwen3ForCausalLM(cfg.text_config)
    m.dna_rna_model = molly_amd.EsmForMaskedLM(cfg.dna_rna_config)
    m.protein_model = molly_amd.EsmForMaskedLM(cfg.protein_config)
    m.prepare(dev, train_llm=False, train_mlp=False, random_init_seed=1234)
    B, T = args.batch, args.prompt
    b = synth_batch(B, T, [("protein", args.k_protein)], seed=1)
    res = {}
    for rep in range(2):                                   # first pass warms allocations
        sess = GenerationSession(m, args.new)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        logits = sess.prefill(b["input_ids"], b["attention_mask"], b["omic_ids"], b["omic_info_list"])
        e1.record()
        for _ in range(args.new):
            logits = sess.step(ops.argmax(logits))
        e2.record()
        torch.cuda.synchronize()
        res = {"model": args.model, "batch": B, "prompt_len": T, "k_protein": args.k_protein, "new_tokens": args.new,
               "prefill_ms": round(e0.elapsed_time(e1), 2), "prefill_tokens_per_s": round(B * T / e0.elapsed_time(e1) * 1e3, 1),
               "decode_ms_per_step": round(e1.elapsed_time(e2) / args.new, 3),
               "decode_tokens_per_s": round(B * args.new / e1.elapsed_time(e2) * 1e3, 1)}
        del sess
    if args.api:
        # the reference-facing call (OmicsOne.generate, reference src/model/omics_one.py:187-232) at the reference's inference settings
        # (src/inference_lora.py:293-298), end to end: prefill + `new` sampled steps + everything the loop does on the host
        import time
        gen = torch.Generator(device="cpu").manual_seed(0)
        kw = dict(input_ids=b["input_ids"], attention_mask=b["attention_mask"], omic_ids=b["omic_ids"], omic_info_list=b["omic_info_list"],
                  max_new_tokens=args.new, do_sample=True, temperature=0.8, top_p=0.95, top_k=20, repetition_penalty=1.1, generator=gen)
        m.generate(**kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = m.generate(**kw)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        res["api_sampled_total_ms"] = round((t1 - t0) * 1e3, 1)
        res["api_sampled_ms_per_step_after_prefill"] = round(((t1 - t0) * 1e3 - res["prefill_ms"]) / out.shape[1], 3)
    n_par = sum(v.numel() for v in m._rt.base.views.values())
    res["weight_stream_GBps_decode"] = round(2 * n_par / (res["decode_ms_per_step"] * 1e-3) / 1e9, 1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
