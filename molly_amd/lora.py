"""LoRA adapters in PEFT's on-disk layout — inference side (BASELINE config 5; reference: src/inference_lora.py:208-234 loads
`PeftModel.from_pretrained(self.model.model, ckpt)` + `dna_rna_projector.bin` / `protein_projector.bin`).

PEFT layout (adapter_config.json + adapter_model.safetensors|.bin): for every target Linear of the LLM
    base_model.model.model.layers.N.<path>.lora_A.weight  [r, in]      base_model.model.model.layers.N.<path>.lora_B.weight  [out, r]
and y = W x + (lora_alpha / r) * B (A x).  For inference the adapter is MERGED at load:  W += (alpha/r) * B A  — one
rank-r GEMM per target (the library's bf16 MFMA kernel, fp32 accumulate into the fp32 view of W), after which the hot
path is exactly the base path (no per-token adapter cost, same kernels).  Targets = the reference's discovery rule
(src/utils/tools.py:352-372): every nn.Linear leaf name of the LLM except lm_head -> q,k,v,o,gate,up,down.

Training (reference `--use-lora`: src/train.py:654-657 -> pre_train_lora, src/utils/tools.py:345-396): the base LLM and
the encoders are frozen, every target gets A [r, in] ~ N(0, 1/r) ("gaussian") and B [out, r] = 0, the projectors stay
trainable, and  y = W x + (alpha/r) * B (A dropout_0.05(x)).  `LoraConfig` / `lora_specs` / `LoraRuntime` below describe
that state for `OmicsOne.prepare(lora=...)`; the branch itself (forward + backward on the MFMA GEMM) is in qwen3.py.
The rank is stored padded to a multiple of 64 (the GEMM's K granule); the pad rows/columns are zero, receive exactly zero
gradient, and are cut off when the adapter is written (`save_adapter`, PEFT layout).
PEFT is not importable offline, so this restates its published algorithm (peft/tuners/lora/layer.py Linear.forward,
LoraLayer.reset_lora_parameters) — parity for this branch is pinned against the torch-fp32 oracle only.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import Dict, List, Tuple

import torch

from . import ops

TARGETS = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj")
_PATH = {"q_proj": "self_attn.", "k_proj": "self_attn.", "v_proj": "self_attn.", "o_proj": "self_attn.",
         "gate_proj": "mlp.", "up_proj": "mlp.", "down_proj": "mlp."}


@dataclass
class LoraConfig:
    """reference: LoraConfig(r=args.lora_r, lora_alpha=64, lora_dropout=0.05, init_lora_weights="gaussian", bias="none")
    at src/utils/tools.py:379-387."""
    r: int = 64
    lora_alpha: float = 64.0
    lora_dropout: float = 0.05
    seed: int = 0                          # dropout-mask stream; the adapter init uses prepare()'s generator

    @property
    def r_pad(self) -> int:
        return (self.r + 63) // 64 * 64

    @property
    def scaling(self) -> float:
        return self.lora_alpha / self.r


def target_dims(cfg) -> Dict[str, Tuple[int, int]]:
    """module -> (in_features, out_features) of the LLM's nn.Linear leaves except lm_head (src/utils/tools.py:352-361)."""
    h, hd, nh, nkv, ff = cfg.hidden_size, cfg.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.intermediate_size
    return {"q_proj": (h, nh * hd), "k_proj": (h, nkv * hd), "v_proj": (h, nkv * hd), "o_proj": (nh * hd, h),
            "gate_proj": (h, ff), "up_proj": (h, ff), "down_proj": (ff, h)}


def lora_name(i: int, mod: str, which: str, prefix: str = "model.") -> str:
    return f"{prefix}model.layers.{i}.{_PATH[mod]}{mod}.lora_{which}.weight"


def lora_specs(cfg, lc: LoraConfig, prefix: str = "model.") -> List[Tuple[str, Tuple[int, ...]]]:
    dims, rp = target_dims(cfg), lc.r_pad
    specs = []
    for i in range(cfg.num_hidden_layers):
        for mod in TARGETS:
            fin, fout = dims[mod]
            specs.append((lora_name(i, mod, "A", prefix), (rp, fin)))
            specs.append((lora_name(i, mod, "B", prefix), (fout, rp)))
    return specs


class LoraRuntime:
    """Views of the adapter matrices (and their gradients) per layer and target, plus the dropout-seed schedule."""

    def __init__(self, cfg, lc: LoraConfig, params, grads, prefix: str = "model."):
        self.cfg, self.r, self.rp, self.scale, self.p, self.seed = lc, lc.r, lc.r_pad, lc.scaling, lc.lora_dropout, lc.seed
        L = cfg.num_hidden_layers
        self.A = [{m: params.views[lora_name(i, m, "A", prefix)] for m in TARGETS} for i in range(L)]
        self.B = [{m: params.views[lora_name(i, m, "B", prefix)] for m in TARGETS} for i in range(L)]
        if grads is not None:
            self.dA = [{m: grads.views[lora_name(i, m, "A", prefix)] for m in TARGETS} for i in range(L)]
            self.dB = [{m: grads.views[lora_name(i, m, "B", prefix)] for m in TARGETS} for i in range(L)]
        self.step = 0                      # one dropout stream per forward pass (micro-step)

    def init_gaussian(self, generator):
        """PEFT init_lora_weights="gaussian": A ~ N(0, 1/r), B = 0 (pad rows stay zero)."""
        for la, lb in zip(self.A, self.B):
            for m in TARGETS:
                la[m].zero_()
                la[m][:self.r].normal_(0.0, 1.0 / self.r, generator=generator)
                lb[m].zero_()

    def mask_seed(self, layer: int, mod: str) -> int:
        return ((self.seed & 0xffffffff) << 32) | ((self.step & 0xfffff) << 12) | (layer << 4) | TARGETS.index(mod)


def load_live_adapter(model, path: str) -> int:
    """Copy a PEFT LoRA checkpoint into the LIVE adapter of a model prepared with `lora=LoraConfig(...)` (continue training,
    or un-merged inference); loads the projector .bin files if present.  Returns the number of target matrices."""
    rt = model._runtime()
    lo = rt.llm.lora
    if lo is None:
        raise RuntimeError("load_live_adapter: the model was prepared without lora=...")
    with open(os.path.join(path, "adapter_config.json")) as f:
        cfg = json.load(f)
    if int(cfg["r"]) != lo.r or float(cfg["lora_alpha"]) / int(cfg["r"]) != lo.scale:
        raise ValueError(f"adapter r={cfg['r']} alpha={cfg['lora_alpha']} does not match the prepared LoraConfig")
    tens = load_adapter_tensors(path)
    n = 0
    for i in range(len(lo.A)):
        for m in TARGETS:
            key = "base_model.model." + lora_name(i, m, "A", "")
            if key not in tens:
                continue
            lo.A[i][m].zero_(); lo.B[i][m].zero_()
            lo.A[i][m][:lo.r].copy_(tens[key].to(rt.dev, torch.bfloat16))
            lo.B[i][m][:, :lo.r].copy_(tens[key.replace("lora_A", "lora_B")].to(rt.dev, torch.bfloat16))
            n += 1
    for proj in ("dna_rna_projector", "protein_projector"):
        f = os.path.join(path, proj + ".bin")
        if os.path.exists(f):
            sd = torch.load(f, map_location="cpu")
            rt.W[proj + ".weight"].copy_(sd["weight"].to(rt.dev))
            rt.W[proj + ".bias"].copy_(sd["bias"].to(rt.dev))
    return n


def adapter_state_dict(model) -> Dict[str, torch.Tensor]:
    """PEFT file layout: base_model.model.<llm path>.lora_A.weight [r, in] / lora_B.weight [out, r] (pad cut off)."""
    rt = model._runtime()
    lo = rt.llm.lora
    out = {}
    for i in range(len(lo.A)):
        for m in TARGETS:
            key = "base_model.model." + lora_name(i, m, "A", "")
            out[key] = lo.A[i][m][:lo.r].detach().to("cpu").clone()
            out[key.replace("lora_A", "lora_B")] = lo.B[i][m][:, :lo.r].detach().to("cpu").contiguous().clone()
    return out


def save_adapter(model, output_dir: str):
    """What `PeftModel.save_pretrained` + the projector dumps leave behind (reference: src/trainer/omics_trainer.py:89-103):
    adapter_config.json, adapter_model.safetensors (adapter_model.bin without safetensors), dna_rna_projector.bin,
    protein_projector.bin."""
    os.makedirs(output_dir, exist_ok=True)
    rt = model._runtime()
    lc = rt.llm.lora.cfg
    with open(os.path.join(output_dir, "adapter_config.json"), "w") as f:
        json.dump({"peft_type": "LORA", "task_type": "CAUSAL_LM", "r": lc.r, "lora_alpha": lc.lora_alpha,
                   "lora_dropout": lc.lora_dropout, "target_modules": list(TARGETS), "bias": "none",
                   "init_lora_weights": "gaussian", "use_rslora": False, "fan_in_fan_out": False}, f, indent=2)
    sd = adapter_state_dict(model)
    try:
        from safetensors.torch import save_file
        save_file(sd, os.path.join(output_dir, "adapter_model.safetensors"))
    except ImportError:
        torch.save(sd, os.path.join(output_dir, "adapter_model.bin"))
    for proj in ("dna_rna_projector", "protein_projector"):
        torch.save({"weight": rt.W[proj + ".weight"].detach().to("cpu").clone(),
                    "bias": rt.W[proj + ".bias"].detach().to("cpu").clone()}, os.path.join(output_dir, proj + ".bin"))


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
        if full not in rt.W:
            raise KeyError(f"adapter target {full} is not a parameter of this model")
        W = rt.W[full]
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
            rt.W[proj + ".weight"].copy_(sd["weight"].to(dev))
            rt.W[proj + ".bias"].copy_(sd["bias"].to(dev))
    return merged
