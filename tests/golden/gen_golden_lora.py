#!/usr/bin/env python3
"""LoRA fixture.  PEFT (un-pinned in the reference's requirements.txt:24, not installed here, no network) cannot be imported, so
the adapter math is anchored one level down instead: the REFERENCE's OmicsOne over the installed HF Qwen3ForCausalLM, with every
target nn.Linear the reference's discovery loop selects (src/utils/tools.py:352-361: q,k,v,o,gate,up,down; lm_head excluded)
replaced IN THE HF MODULE TREE by a wrapper that follows PEFT's published `lora.Linear.forward` (peft/tuners/lora/layer.py):
        result = base_layer(x);  result = result + lora_B(lora_A(lora_dropout(x))) * scaling,   scaling = lora_alpha / r
(nn.Dropout(p) in train mode, nn.Identity for p = 0), base and encoders frozen, projectors trainable (pre_train_lora,
src/utils/tools.py:345-396).  HF's attention / MLP / loss code, torch autograd and the reference's injection path are the real
ones; only the 6-line wrapper is written here.  Dumps loss, logits and the gradients of every lora_A / lora_B / projector.
r = 8, alpha = 16 (scaling 2: a dropped or squared scale is caught), dropout 0, B non-zero.  Build container only.

    python tests/golden/gen_golden_lora.py      # writes tests/golden/tiny_lora.npz
"""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import OUT, TINY, build_reference_model, make_batch  # noqa: E402

R_, ALPHA, SEED = 8, 16.0, 77


class LoraLinear(nn.Module):
    def __init__(self, base: nn.Linear, r: int, alpha: float, p: float):
        super().__init__()
        self.base_layer = base
        self.lora_A = nn.Linear(base.in_features, r, bias=False)
        self.lora_B = nn.Linear(r, base.out_features, bias=False)
        self.lora_dropout = nn.Dropout(p) if p > 0 else nn.Identity()
        self.scaling = alpha / r

    def forward(self, x):
        result = self.base_layer(x)
        return result + self.lora_B(self.lora_A(self.lora_dropout(x))) * self.scaling


def main():
    torch.manual_seed(0)
    from molly_amd.synth import synth_tensor
    model, _ = build_reference_model(TINY)
    for p in model.parameters():
        p.requires_grad_(False)
    names = []
    for li, layer in enumerate(model.model.model.layers):
        for parent, leaf in ((layer.self_attn, "q_proj"), (layer.self_attn, "k_proj"), (layer.self_attn, "v_proj"),
                             (layer.self_attn, "o_proj"), (layer.mlp, "gate_proj"), (layer.mlp, "up_proj"), (layer.mlp, "down_proj")):
            w = LoraLinear(getattr(parent, leaf), R_, ALPHA, 0.0)
            sub = "self_attn." if parent is layer.self_attn else "mlp."
            base = f"model.model.layers.{li}.{sub}{leaf}"
            with torch.no_grad():
                w.lora_A.weight.copy_(synth_tensor(base + ".lora_A.weight", tuple(w.lora_A.weight.shape), SEED) * (50.0 / R_))
                w.lora_B.weight.copy_(synth_tensor(base + ".lora_B.weight", tuple(w.lora_B.weight.shape), SEED) * 2.5)
            setattr(parent, leaf, w)
            names.append((base, w))
    for proj in (model.dna_rna_projector, model.protein_projector):
        for p in proj.parameters():
            p.requires_grad_(True)
    model.train()
    batch = make_batch(TINY)
    res = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], omic_ids=batch["omic_ids"],
                omic_info_list=batch["omic_info_list"], labels=batch["labels"])
    res.loss.backward()
    out = {"loss": np.float64(res.loss.item()), "logits": res.logits.detach().numpy()[:, ::2].copy(),
           "r": np.int64(R_), "alpha": np.float64(ALPHA), "seed": np.int64(SEED)}
    for base, w in names:
        out["g/" + base + ".lora_A.weight"] = w.lora_A.weight.grad.numpy().copy()
        out["g/" + base + ".lora_B.weight"] = w.lora_B.weight.grad.numpy().copy()
    for n in ("dna_rna_projector", "protein_projector"):
        out[f"g/{n}.weight"] = getattr(model, n).weight.grad.numpy().copy()
        out[f"g/{n}.bias"] = getattr(model, n).bias.grad.numpy().copy()
    assert not any(p.grad is not None for n, p in model.named_parameters() if "lora_" not in n and "projector" not in n)
    np.savez_compressed(os.path.join(OUT, "tiny_lora.npz"), **out)
    print("loss", res.loss.item(), "tensors", len(out))


if __name__ == "__main__":
    main()
