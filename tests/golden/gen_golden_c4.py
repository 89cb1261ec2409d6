#!/usr/bin/env python3
"""BASELINE configs[3] ("Molly-8B ... seq_len 4k text + 1024-residue protein + 6 kbp DNA, long-context encoder stress") at its
DEFINING sizes through the REFERENCE's OmicsOne (fp32, CPU, installed HF modules), everything trainable:
  * Qwen3-8B decoder widths (h 4096, 32 q / 8 kv heads x 128, ff 12288, untied head), TWO decoder layers, T = 4096;
  * the protein encoder at FULL ESM2-650M shape (33 layers, 1280 / 20 x 64 / 5120, rotary, token-dropout), K = 1024;
  * the DNA encoder at FULL NT-500M-human-ref shape (24 layers, learned absolute positions, max_position_embeddings 1002),
    K = 1000 — NOT a multiple of 64, position ids 2..1001 (HF:models/esm/modeling_esm.py:1050-1063);
  * one sample carrying both spans (omic rows of different K: a list of rows, what process_omic_sequences iterates over).
Only the vocabulary is reduced (2048) to keep the head cheap on CPU.  Weights come from molly_amd.synth by name on both
sides; stored: sub-sampled forward tensors, loss, norm + first 256 entries of every gradient, and the reference's own bf16
forward as the tolerance yardstick.  Runs ONLY in the build container (~10 min, ~40 GB).

    python tests/golden/gen_golden_c4.py      # writes tests/golden/c4_fp32.npz, c4_meta.json
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import OUT, build_reference_model, grads_summary  # noqa: E402

C4 = dict(
    text=dict(vocab_size=2048, hidden_size=4096, intermediate_size=12288, num_hidden_layers=2,
              num_attention_heads=32, num_key_value_heads=8, head_dim=128, max_position_embeddings=40960,
              rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=False, attention_bias=False,
              pad_token_id=1000, eos_token_id=1001),
    dna_rna=dict(vocab_size=4105, hidden_size=1280, intermediate_size=5120, num_hidden_layers=24,
                 num_attention_heads=20, max_position_embeddings=1002, position_embedding_type="absolute",
                 token_dropout=False, pad_token_id=1, mask_token_id=2, layer_norm_eps=1e-12,
                 emb_layer_norm_before=False, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0),
    protein=dict(vocab_size=33, hidden_size=1280, intermediate_size=5120, num_hidden_layers=33,
                 num_attention_heads=20, max_position_embeddings=1026, position_embedding_type="rotary",
                 token_dropout=True, pad_token_id=1, mask_token_id=32, layer_norm_eps=1e-5,
                 emb_layer_norm_before=False, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0),
    K=1024, K_protein=1024, K_dna=1000, B=1, T=4096, seed_w=4401, seed_b=44,
    special_ids={"dna": [1010, 1011, 1012], "rna": [1013, 1014, 1015], "protein": [1016, 1017, 1018]},
)
SUB_T, SUB_H = 16, 16


def sub(t):
    return t.detach().float().numpy()[:, ::SUB_T, ::SUB_H].copy()


def make_batch(c):
    from molly_amd.synth import synth_batch
    sp = {k: tuple(v) for k, v in c["special_ids"].items()}
    return synth_batch(c["B"], c["T"], [("protein", c["K_protein"]), ("dna", c["K_dna"])], c["seed_b"], text_vocab=1000,
                       special_ids=sp, pad_id=1000, mixed_k=True)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    t0 = time.time()
    model, shapes = build_reference_model(C4, attn="sdpa")        # sdpa: eager would hold 2 GB score tensors per layer
    model.dna_rna_project_token_num, model.protein_project_token_num = C4["K_dna"], C4["K_protein"]
    model.train()
    batch = make_batch(C4)
    info = batch["omic_info_list"]
    out = {("in/" + k): batch[k].numpy() for k in ("input_ids", "labels", "attention_mask")}
    out["in/omic_protein"] = batch["omic_ids"][0][0].numpy()
    out["in/omic_dna"] = batch["omic_ids"][0][1].numpy()
    caught, enc_out = {}, {}
    hooks = [
        model.model.model.register_forward_pre_hook(
            lambda mod, a, kw: caught.__setitem__("inputs_embeds", kw["inputs_embeds"].detach().clone()), with_kwargs=True),
        model.model.model.norm.register_forward_hook(lambda mod, a, o: caught.__setitem__("final_hidden", o.detach().clone())),
        model.protein_model.esm.register_forward_hook(
            lambda mod, a, o: enc_out.__setitem__("protein", o.last_hidden_state.detach().clone())),
        model.dna_rna_model.esm.register_forward_hook(
            lambda mod, a, o: enc_out.__setitem__("dna_rna", o.last_hidden_state.detach().clone())),
    ]
    res = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], omic_ids=batch["omic_ids"],
                omic_info_list=info, labels=batch["labels"])
    for h in hooks:
        h.remove()
    print(f"forward done {time.time() - t0:.0f} s, loss {res.loss.item():.5f}", flush=True)
    out["fwd/loss"] = np.float64(res.loss.item())
    out["fwd/logits"] = sub(res.logits)
    for k in ("inputs_embeds", "final_hidden"):
        out["fwd/" + k] = sub(caught[k])
    out["fwd/enc_protein"] = sub(enc_out["protein"])
    out["fwd/enc_dna_rna"] = sub(enc_out["dna_rna"])
    res.loss.backward()
    g = {}
    grads_summary(model, g)
    out.update({k: v for k, v in g.items() if not k.startswith("gfull/")})
    print(f"backward done {time.time() - t0:.0f} s, grad tensors {sum(k.startswith('gnorm/') for k in out)}", flush=True)
    del res, g
    model_bf = model.to(torch.bfloat16).eval()
    with torch.no_grad():
        rb = model_bf(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], omic_ids=batch["omic_ids"],
                      omic_info_list=info, labels=batch["labels"])
    out["bf16/loss"] = np.float64(rb.loss.item())
    out["bf16/logits"] = sub(rb.logits)
    print("bf16 reference: loss", rb.loss.item(), "max|dlogit|", float(np.abs(out["bf16/logits"] - out["fwd/logits"]).max()),
          "max|logit|", float(np.abs(out["fwd/logits"]).max()), flush=True)
    np.savez_compressed(os.path.join(OUT, "c4_fp32.npz"), **out)
    with open(os.path.join(OUT, "c4_meta.json"), "w") as f:
        json.dump({"config": C4, "omic_info_list": info, "sub": [SUB_T, SUB_H],
                   "state_dict_shapes": {k: list(v) for k, v in shapes.items()},
                   "transformers": __import__("transformers").__version__, "torch": torch.__version__}, f, indent=1)
    print(f"done {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
