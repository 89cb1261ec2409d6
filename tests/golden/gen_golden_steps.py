#!/usr/bin/env python3
"""G5 of SURVEY.md 8(c): a short TRAJECTORY of the step loop on the reference's OmicsOne (tiny config, fp32, CPU) with the
installed pieces the reference's trainer uses — torch.optim.AdamW (lr 3e-4 here, wd 1e-2 on HF's decay set, betas .9/.999,
eps 1e-8), torch clip_grad_norm_(1.0), transformers.get_linear_schedule_with_warmup(warmup = ceil(0.1 * total)) — driven in
the order of the reference's loop (src/trainer/domain_loss.py:594-724): gradient accumulation over GA = 2 micro-batches
WITHOUT dividing the loss (the **kwargs quirk, :1011-1013), clip -> optimizer.step -> scheduler.step -> zero_grad.
Four optimizer steps over eight different micro-batches.  Runs ONLY in the build container.

    python tests/golden/gen_golden_steps.py      # writes tests/golden/tiny_steps.npz
"""
import copy
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import OUT, TINY, build_reference_model, make_batch  # noqa: E402

GA, STEPS, TOTAL, LR = 2, 4, 10, 3e-4


def main():
    from transformers import get_linear_schedule_with_warmup
    torch.manual_seed(0)
    model, _ = build_reference_model(TINY)
    model.train()
    for sub in (model.dna_rna_model, model.protein_model):               # --train-llm --train-mlp: encoders frozen
        for p in sub.parameters():
            p.requires_grad_(False)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    no_decay = [n for n, _ in named if n.endswith("bias") or "norm" in n.lower()]
    opt = torch.optim.AdamW([{"params": [p for n, p in named if n not in no_decay], "weight_decay": 1e-2},
                             {"params": [p for n, p in named if n in no_decay], "weight_decay": 0.0}],
                            lr=LR, betas=(0.9, 0.999), eps=1e-8)
    sched = get_linear_schedule_with_warmup(opt, math.ceil(0.1 * TOTAL), TOTAL)
    out = {}
    losses, norms, lrs = [], [], []
    for s in range(STEPS):
        for k in range(GA):
            cfg = copy.deepcopy(TINY)
            cfg["seed_b"] = 500 + 10 * (s * GA + k)
            b = make_batch(cfg)
            for key in ("input_ids", "labels", "attention_mask", "omic_ids"):
                out[f"in/{s}/{k}/{key}"] = b[key].numpy()
            if s == 0 and k == 0:
                info = b["omic_info_list"]
            assert b["omic_info_list"][0][0]["type"] == info[0][0]["type"]
            out[f"in/{s}/{k}/starts"] = np.array([[d["start"] for d in row] for row in b["omic_info_list"]])
            res = model(input_ids=b["input_ids"], attention_mask=b["attention_mask"], omic_ids=b["omic_ids"],
                        omic_info_list=b["omic_info_list"], labels=b["labels"])
            res.loss.backward()                                          # NOT divided by GA: gradients sum over the window
            losses.append(res.loss.item())
        lrs.append(sched.get_last_lr()[0])
        norms.append(torch.nn.utils.clip_grad_norm_([p for _, p in named], 1.0).item())
        opt.step()
        sched.step()
        opt.zero_grad()
    out["loss"] = np.array(losses).reshape(STEPS, GA)
    out["grad_norm"] = np.array(norms)
    out["lr"] = np.array(lrs)
    for n, p in named:
        out["pnorm/" + n] = np.float64(p.detach().double().norm().item())
        out["phead/" + n] = p.detach().flatten()[:256].numpy().copy()
    out["meta"] = np.array([GA, STEPS, TOTAL])
    out["base_lr"] = np.float64(LR)
    np.savez_compressed(os.path.join(OUT, "tiny_steps.npz"), **out)
    print("losses", np.round(out["loss"], 4).tolist(), "\ngrad norms", np.round(norms, 4).tolist(), "\nlr", lrs)


if __name__ == "__main__":
    main()
