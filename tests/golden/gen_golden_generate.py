#!/usr/bin/env python3
"""G7 golden vector: greedy decode through the REFERENCE's `OmicsOne.generate`
(reference: src/model/omics_one.py:187-233 -> HF `generate(inputs_embeds=...)`) on the tiny model, left-padded prompts of
different lengths with one omic span each (the shape `qwen_omics_collate_fn_inference` produces, reference
src/dataset/omics_dataset.py:387-391: pad on the left, span starts shifted).  Runs ONLY in the build container.

Stored: the prompts, the 8 new tokens per sample the reference generated (do_sample=False) and, per step, the margin between
the best and the second-best logit of the reference's fp32 forward — the GPU test teacher-forces the reference's tokens and
requires the HIP path's argmax to agree wherever that margin is not a near-tie.

    python tests/golden/gen_golden_generate.py      # writes tests/golden/generate_g7.json
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import OUT, TINY, build_reference_model  # noqa: E402

N_NEW = 8


def left_padded_batch(cfgd):
    from molly_amd.synth import synth_batch
    sp = {k: tuple(v) for k, v in cfgd["special_ids"].items()}
    T, K = 192, cfgd["K"]
    rows = []
    for i, (typ, valid) in enumerate((("protein", 192), ("rna", 150), ("dna", 101))):
        b = synth_batch(1, valid, [(typ, K)], seed=100 + i, text_vocab=1000, special_ids=sp, pad_id=1000)
        pad = T - valid
        ids = torch.cat([torch.full((pad,), 1000, dtype=torch.int64), b["input_ids"][0]])
        mask = torch.cat([torch.zeros(pad, dtype=torch.int64), torch.ones(valid, dtype=torch.int64)])
        info = [{"type": typ, "start": b["omic_info_list"][0][0]["start"] + pad}]
        rows.append((ids, mask, b["omic_ids"][0], info))
    return {"input_ids": torch.stack([r[0] for r in rows]), "attention_mask": torch.stack([r[1] for r in rows]),
            "omic_ids": torch.stack([r[2] for r in rows]), "omic_info_list": [r[3] for r in rows]}


def main():
    torch.manual_seed(0)
    m, _ = build_reference_model(TINY, torch.float32)
    m.eval()
    b = left_padded_batch(TINY)
    # the reference hard-codes max_new_tokens=3072 and the config's eos/pad ids; bound the length through the kwargs it
    # forwards (`**generate_kwargs`, omics_one.py:201, 231) — HF lets the later keyword win is NOT guaranteed, so patch it
    orig = m.model.generate

    def bounded(*a, **kw):
        kw["max_new_tokens"] = N_NEW
        return orig(*a, **kw)
    m.model.generate = bounded
    with torch.no_grad():
        new = m.generate(input_ids=b["input_ids"], attention_mask=b["attention_mask"], omic_ids=b["omic_ids"],
                         omic_info_list=b["omic_info_list"], do_sample=False)
    assert new.shape == (3, N_NEW), new.shape
    # per-step margins from a full re-forward of prompt + generated tokens (fp32, same module)
    margins = []
    with torch.no_grad():
        hs = m.model.get_input_embeddings()(b["input_ids"])
        hs = m.process_omic_sequences(hs, b["omic_ids"], b["omic_info_list"], hs.device)
        emb_new = m.model.get_input_embeddings()(new)
        full = torch.cat([hs, emb_new], 1)
        mask = torch.cat([b["attention_mask"], torch.ones_like(new)], 1)
        pos = (mask.cumsum(1) - 1).clamp(min=0)
        logits = m.model(inputs_embeds=full, attention_mask=mask, position_ids=pos).logits
        T = b["input_ids"].shape[1]
        for t in range(N_NEW):
            lg = logits[:, T - 1 + t].float()
            top = lg.topk(2, dim=-1)
            assert torch.equal(top.indices[:, 0], new[:, t]), (t, top.indices[:, 0], new[:, t])
            margins.append((top.values[:, 0] - top.values[:, 1]).tolist())
    out = {"n_new": N_NEW, "input_ids": b["input_ids"].tolist(), "attention_mask": b["attention_mask"].tolist(),
           "omic_ids": b["omic_ids"].tolist(), "omic_info_list": b["omic_info_list"], "new_tokens": new.tolist(),
           "margins": margins}
    with open(os.path.join(OUT, "generate_g7.json"), "w") as f:
        json.dump(out, f)
    print("new tokens", new.tolist())
    print("min margin per step", [round(min(x), 4) for x in margins])


if __name__ == "__main__":
    main()
