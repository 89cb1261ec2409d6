"""Configuration objects of the drop-in path.

`OmicsModalConfig` / `get_omics_one_config` keep the reference's names and fields (reference:
src/model/config.py:8-46, 49-86) but read plain JSON `config.json` files instead of going through
`transformers.AutoConfig` — the hot path has no dependency on transformers at run time.
`LlmConfig` / `EncConfig` carry exactly the HF config fields the arithmetic reads
(HF:models/qwen3/configuration_qwen3.py, HF:models/esm/configuration_esm.py).
"""
from __future__ import annotations

import json
import os
from dataclasses import asdict, dataclass, field
from typing import Optional


@dataclass
class LlmConfig:
    vocab_size: int = 151936
    hidden_size: int = 2048
    intermediate_size: int = 6144
    num_hidden_layers: int = 28
    num_attention_heads: int = 16
    num_key_value_heads: int = 8
    head_dim: int = 128
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6
    tie_word_embeddings: bool = True
    max_position_embeddings: int = 40960
    pad_token_id: Optional[int] = None
    eos_token_id: Optional[int] = 151645
    use_cache: bool = False
    gradient_checkpointing: bool = False
    use_return_dict: bool = True
    model_type: str = "qwen3"

    @classmethod
    def from_dict(cls, d: dict) -> "LlmConfig":
        d = dict(d)
        if "rope_parameters" in d and isinstance(d["rope_parameters"], dict):      # transformers 5.x layout
            d.setdefault("rope_theta", d["rope_parameters"].get("rope_theta", 1e6))
        if "head_dim" not in d or d["head_dim"] is None:
            d["head_dim"] = d["hidden_size"] // d["num_attention_heads"]
        if d.get("attention_bias"):
            raise NotImplementedError("Qwen3 attention_bias=True is not on Molly's path (all Qwen3 sizes use False)")
        return cls(**{k: v for k, v in d.items() if k in cls.__dataclass_fields__})

    def to_dict(self):
        return asdict(self)


@dataclass
class EncConfig:
    vocab_size: int = 33
    hidden_size: int = 1280
    intermediate_size: int = 5120
    num_hidden_layers: int = 33
    num_attention_heads: int = 20
    max_position_embeddings: int = 1026
    position_embedding_type: str = "rotary"          # "rotary" (ESM-2) | "absolute" (nucleotide-transformer v1)
    token_dropout: bool = True
    pad_token_id: int = 1
    mask_token_id: int = 32
    layer_norm_eps: float = 1e-5
    rope_theta: float = 10000.0
    emb_layer_norm_before: bool = False
    hidden_dropout_prob: float = 0.0
    attention_probs_dropout_prob: float = 0.0
    use_cache: bool = False
    gradient_checkpointing: bool = False
    model_type: str = "esm"

    @classmethod
    def from_dict(cls, d: dict) -> "EncConfig":
        d = dict(d)
        if d.get("emb_layer_norm_before"):
            raise NotImplementedError("emb_layer_norm_before=True (ESM-1b) is not on Molly's path")
        if d.get("is_folding_model"):
            raise NotImplementedError("ESMFold configs are not on Molly's path")
        return cls(**{k: v for k, v in d.items() if k in cls.__dataclass_fields__})

    def to_dict(self):
        return asdict(self)


def _load_json_config(path: str) -> dict:
    p = os.path.join(path, "config.json") if os.path.isdir(path) else path
    with open(p) as f:
        return json.load(f)


@dataclass
class OmicsModalConfig:
    """reference: src/model/config.py:8-46 (same field names and defaults)."""
    text_config: Optional[LlmConfig] = None
    dna_rna_config: Optional[EncConfig] = None
    protein_config: Optional[EncConfig] = None
    text_max_length: int = 2048
    dna_rna_project_token_num: int = 64
    dna_rna_max_length: int = 512
    protein_project_token_num: int = 64
    protein_max_length: int = 512
    gradient_checkpointing: bool = False
    use_cache: bool = False

    def __post_init__(self):
        if isinstance(self.text_config, dict):
            self.text_config = LlmConfig.from_dict(self.text_config)
        if isinstance(self.dna_rna_config, dict):
            self.dna_rna_config = EncConfig.from_dict(self.dna_rna_config)
        if isinstance(self.protein_config, dict):
            self.protein_config = EncConfig.from_dict(self.protein_config)


def get_omics_one_config(text_model_path, dna_rna_model_path, protein_model_path) -> OmicsModalConfig:
    """reference: src/model/config.py:49-86 — three model directories (or config.json paths) -> one config."""
    cfg = OmicsModalConfig(
        text_config=LlmConfig.from_dict(_load_json_config(text_model_path)),
        dna_rna_config=EncConfig.from_dict(_load_json_config(dna_rna_model_path)),
        protein_config=EncConfig.from_dict(_load_json_config(protein_model_path)),
    )
    for c in (cfg.text_config, cfg.dna_rna_config, cfg.protein_config):
        c.use_cache = cfg.use_cache
        c.gradient_checkpointing = cfg.gradient_checkpointing
    return cfg


# ---- public model-card shapes (no config.json files exist offline) -------------------
def qwen3(size: str) -> LlmConfig:
    table = {
        "0.6b": dict(hidden_size=1024, intermediate_size=3072, num_hidden_layers=28, num_attention_heads=16,
                     num_key_value_heads=8, tie_word_embeddings=True),
        "1.7b": dict(hidden_size=2048, intermediate_size=6144, num_hidden_layers=28, num_attention_heads=16,
                     num_key_value_heads=8, tie_word_embeddings=True),
        "4b": dict(hidden_size=2560, intermediate_size=9728, num_hidden_layers=36, num_attention_heads=32,
                   num_key_value_heads=8, tie_word_embeddings=True),
        "8b": dict(hidden_size=4096, intermediate_size=12288, num_hidden_layers=36, num_attention_heads=32,
                   num_key_value_heads=8, tie_word_embeddings=False),
    }
    return LlmConfig(**table[size.lower()])


def esm2_650m() -> EncConfig:
    return EncConfig(vocab_size=33, hidden_size=1280, intermediate_size=5120, num_hidden_layers=33,
                     num_attention_heads=20, max_position_embeddings=1026, position_embedding_type="rotary",
                     token_dropout=True, pad_token_id=1, mask_token_id=32)


def esm2_t6_8m() -> EncConfig:
    """ESM2-t6-8M: the protein encoder of BASELINE configs[0] (the reference's CPU-runnable mini run); head_dim 16."""
    return EncConfig(vocab_size=33, hidden_size=320, intermediate_size=1280, num_hidden_layers=6,
                     num_attention_heads=20, max_position_embeddings=1026, position_embedding_type="rotary",
                     token_dropout=True, pad_token_id=1, mask_token_id=32)


def nt_500m_human_ref() -> EncConfig:
    return EncConfig(vocab_size=4105, hidden_size=1280, intermediate_size=5120, num_hidden_layers=24,
                     num_attention_heads=20, max_position_embeddings=1002, position_embedding_type="absolute",
                     token_dropout=False, pad_token_id=1, mask_token_id=2)


def molly(size: str = "1.7b", k_tokens: int = 512) -> OmicsModalConfig:
    c = OmicsModalConfig(text_config=qwen3(size), dna_rna_config=nt_500m_human_ref(), protein_config=esm2_650m())
    c.dna_rna_project_token_num = c.protein_project_token_num = k_tokens
    return c
