#!/usr/bin/env python3
"""Golden gradients of the ENCODER parameters (reference `--train-bio`, src/utils/tools.py:326-330: the encoders stay
ordinary trainable sub-modules, so the reference's autograd runs through `EsmForMaskedLM` inside `OmicsOne.forward`).
Same tiny model and batch as gen_golden.py; stores, per encoder tensor the reference's forward reads, the gradient norm and
its first 256 entries.  Runs ONLY in the build container.

    python tests/golden/gen_golden_trainbio.py      # writes tests/golden/tiny_trainbio.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import OUT, TINY, build_reference_model, make_batch  # noqa: E402


def main():
    torch.manual_seed(0)
    m, _ = build_reference_model(TINY, torch.float32)
    m.train()                                   # dropout probabilities are 0 in the fixture configs
    b = make_batch(TINY)
    out = m(input_ids=b["input_ids"], attention_mask=b["attention_mask"], omic_ids=b["omic_ids"],
            omic_info_list=b["omic_info_list"], labels=b["labels"])
    out.loss.backward()
    dump = {"loss": np.float32(out.loss.item())}
    n = 0
    for name, p in m.named_parameters():
        if not name.startswith(("dna_rna_model.", "protein_model.")) or p.grad is None:
            continue
        if "lm_head" in name or "contact_head" in name:
            continue                            # heads the reference computes and discards: zero gradient, not on the path
        g = p.grad.detach().float()
        dump["gnorm/" + name] = np.float64(g.double().norm().item())
        dump["ghead/" + name] = g.flatten()[:256].numpy()
        n += 1
    np.savez_compressed(os.path.join(OUT, "tiny_trainbio.npz"), **dump)
    print(f"loss {out.loss.item():.6f}; {n} encoder tensors")


if __name__ == "__main__":
    main()
