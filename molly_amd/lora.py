"""LoRA adapters in PEFT's on-disk layout — inference side (BASELINE config 5; reference: src/inference_lora.py:208-234 loads
`PeftModel.from_pretrained(self.model.model, ckpt)` + `dna_rna_projector.bin` / `protein_projector.bin`).

PEFT layout (adapter_config.json + adapter_model.safetensors|.bin): for every target Linear of the LLM
    base_model.model.model.layers.N.<path>.lora_A.weight  [r, in]      base_model.model.model.layers.N.<path>.lora_B.weight  [out, r]
and y = W x + (lora_alpha / r) * B (A x).  For inference the adapter is MERGED at load:  W += (alpha/r) * B A  — one
rank-r GEMM per target (the library's bf16 MFMA kernel, fp32 accumulate into the fp32 view of W), after which the hot
path is exactly the base path (no per-token adapter cost, same kernels).  Targets = the reference's discovery rule
(src/utils/tools.py:352-372): every nn.Linear leaf name of the LLM except lm_head -> q,k,v,o,gate,up,down.
LoRA *training* (dropout 0.05, adapter gradients) is SURVEY.md §8f-1 and not built yet.
"""
from __future__ import annotations

import json
import os
from typing import Dict

import torch

from . import ops

TARGETS = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")


def load_adapter_tensors(path: str) -> Dict[str, torch.Tensor]:
    st = os.path.join(path, "adapter_model.safetensors")
    if os.path.exists(st):
        from safetensors.torch import load_file
        return load_file(st)
    b = os.path.join(path, "adapter_model.bin")
    if os.path.exists(b):
        return torch.load(b, map_location="cpu")
    raise FileNotFoundError(f"no adapter_model.safetensors / adapter_model.bin under {path}")


def merge_lora_adapter(model, path: str) -> int:
    """Merge a PEFT LoRA checkpoint into the (prepared) model's LLM weights; load the projector .bin files if present.
    Returns the number of merged target matrices."""
    rt = model._runtime()
    with open(os.path.join(path, "adapter_config.json")) as f:
        cfg = json.load(f)
    r, alpha = int(cfg["r"]), float(cfg["lora_alpha"])
    if cfg.get("use_rslora"):
        raise NotImplementedError("rsLoRA scaling is not what the reference configures")
    scale = alpha / r
    tens = load_adapter_tensors(path)
    merged = 0
    dev = rt.dev
    for key, A in tens.items():
        if not key.endswith("lora_A.weight"):
            continue
        kb = key.replace("lora_A.weight", "lora_B.weight")
        name = key[len("base_model.model."):].replace(".lora_A.weight", ".weight")     # -> model.layers.N....weight
        full = "model." + name                                                          # OmicsOne prefix
        if full not in rt.P.views:
            raise KeyError(f"adapter target {full} is not a parameter of this model")
        W = rt.P.views[full]
        Bm = tens[kb].to(dev, torch.bfloat16).contiguous()          # [out, r]
        Am = tens[key].to(dev, torch.bfloat16).contiguous()         # [r, in]
        # delta[out, in] = B[out, r] @ A[r, in]  : A operand = B (k-contiguous, K = r padded to 64), B operand = A (k-major)
        rp = (r + 63) // 64 * 64
        Bp = torch.zeros(Bm.shape[0], rp, dtype=torch.bfloat16, device=dev)
        Bp[:, :r] = Bm
        Ap = torch.zeros(rp, Am.shape[1], dtype=torch.bfloat16, device=dev)
        Ap[:r] = Am
        acc = W.float() / scale                                     # fp32 accumulate, then one rounding back to bf16
        ops.gemm(Bp, Ap, out=acc, accumulate=True, b_kmajor=True)
        W.copy_((acc * scale).to(torch.bfloat16))
        merged += 1
    for proj in ("dna_rna_projector", "protein_projector"):
        f = os.path.join(path, proj + ".bin")
        if os.path.exists(f):
            sd = torch.load(f, map_location="cpu")
            rt.P.views[proj + ".weight"].copy_(sd["weight"].to(dev))
            rt.P.views[proj + ".bias"].copy_(sd["bias"].to(dev))
    return merged
