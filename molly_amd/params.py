"""Flat parameter storage laid out for the hot path.

All bf16 tensors of one group live in ONE contiguous HBM allocation (`FlatBuffer`): the ZeRO-2 step
reduce-scatters / all-gathers / Adam-updates it as a single array, and adjacent placement lets the
kernels treat q|k|v and gate|up as one fused [N, K] GEMM operand without copying.  Every named view keeps
the reference's state-dict key (reference checkpoints: src/trainer/omics_trainer.py:85-105).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Tuple

import torch

from .config import EncConfig, LlmConfig

BF16 = torch.bfloat16


def is_no_decay(name: str) -> bool:
    """HF Trainer decay split (HF:trainer.py get_decay_parameter_names): biases and every norm gain are
    excluded from weight decay."""
    return name.endswith("bias") or "norm" in name.lower()


class FlatBuffer:
    def __init__(self, specs: Iterable[Tuple[str, Tuple[int, ...]]], device, dtype=BF16, align: int = 8,
                 pad_to: int = 8):
        self.names: List[str] = []
        self.offsets: Dict[str, int] = {}
        self.shapes: Dict[str, Tuple[int, ...]] = {}
        off = 0
        for name, shape in specs:
            n = 1
            for s in shape:
                n *= int(s)
            off = (off + align - 1) // align * align
            self.names.append(name)
            self.offsets[name] = off
            self.shapes[name] = tuple(int(s) for s in shape)
            off += n
        self.used = off
        self.numel = (off + pad_to - 1) // pad_to * pad_to
        self.flat = torch.zeros(self.numel, dtype=dtype, device=device)
        self.views: Dict[str, torch.Tensor] = {}
        for name in self.names:
            o, shp = self.offsets[name], self.shapes[name]
            n = 1
            for s in shp:
                n *= s
            self.views[name] = self.flat[o:o + n].view(shp)

    def span(self, first: str, last: str, cols: int) -> torch.Tensor:
        """One [rows, cols] view covering the adjacent tensors first..last (must be contiguous in the buffer)."""
        o0 = self.offsets[first]
        o1 = self.offsets[last] + self.views[last].numel()
        assert (o1 - o0) % cols == 0
        rows = (o1 - o0) // cols
        # adjacency check: no alignment gap may sit between the members
        names = self.names[self.names.index(first):self.names.index(last) + 1]
        assert sum(self.views[n].numel() for n in names) == o1 - o0, "span members are not densely adjacent"
        return self.flat[o0:o1].view(rows, cols)

    def like(self, dtype=None) -> "FlatBuffer":
        fb = object.__new__(FlatBuffer)
        fb.names, fb.offsets, fb.shapes, fb.used, fb.numel = self.names, self.offsets, self.shapes, self.used, self.numel
        fb.flat = torch.zeros(self.numel, dtype=dtype or self.flat.dtype, device=self.flat.device)
        fb.views = {}
        for name in self.names:
            o, shp = self.offsets[name], self.shapes[name]
            fb.views[name] = fb.flat[o:o + self.views[name].numel()].view(shp)
        return fb


# ---- name/shape tables (the reference's state-dict layout) -------------------------------------------------
def llm_param_specs(cfg: LlmConfig, prefix: str = "model.") -> List[Tuple[str, Tuple[int, ...]]]:
    h, hd, nh, nkv, ff = cfg.hidden_size, cfg.head_dim, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.intermediate_size
    specs = [(prefix + "model.embed_tokens.weight", (cfg.vocab_size, h))]
    for i in range(cfg.num_hidden_layers):
        lp = f"{prefix}model.layers.{i}."
        specs += [
            (lp + "self_attn.q_proj.weight", (nh * hd, h)),      # q|k|v adjacent -> fused QKV operand
            (lp + "self_attn.k_proj.weight", (nkv * hd, h)),
            (lp + "self_attn.v_proj.weight", (nkv * hd, h)),
            (lp + "self_attn.o_proj.weight", (h, nh * hd)),
            (lp + "mlp.gate_proj.weight", (ff, h)),               # gate|up adjacent -> fused operand
            (lp + "mlp.up_proj.weight", (ff, h)),
            (lp + "mlp.down_proj.weight", (h, ff)),
        ]
    if not cfg.tie_word_embeddings:
        specs.append((prefix + "lm_head.weight", (cfg.vocab_size, h)))
    return specs


def llm_norm_specs(cfg: LlmConfig, prefix: str = "model.") -> List[Tuple[str, Tuple[int, ...]]]:
    h, hd = cfg.hidden_size, cfg.head_dim
    specs = []
    for i in range(cfg.num_hidden_layers):
        lp = f"{prefix}model.layers.{i}."
        specs += [(lp + "input_layernorm.weight", (h,)), (lp + "post_attention_layernorm.weight", (h,)),
                  (lp + "self_attn.q_norm.weight", (hd,)), (lp + "self_attn.k_norm.weight", (hd,))]
    specs.append((prefix + "model.norm.weight", (h,)))
    return specs


def projector_specs(llm: LlmConfig, dna: EncConfig, prot: EncConfig):
    w = [("dna_rna_projector.weight", (llm.hidden_size, dna.hidden_size)),
         ("protein_projector.weight", (llm.hidden_size, prot.hidden_size))]
    b = [("dna_rna_projector.bias", (llm.hidden_size,)), ("protein_projector.bias", (llm.hidden_size,))]
    return w, b


def enc_param_specs(cfg: EncConfig, prefix: str) -> List[Tuple[str, Tuple[int, ...]]]:
    """`prefix` e.g. "protein_model."  Only what Molly's forward reads (the LM/contact heads are dead weight:
    SURVEY.md §0.4-8) — loaders simply skip the other checkpoint keys."""
    he, ffe = cfg.hidden_size, cfg.intermediate_size
    e = prefix + "esm."
    specs = [(e + "embeddings.word_embeddings.weight", (cfg.vocab_size, he))]
    if cfg.position_embedding_type == "absolute":
        specs.append((e + "embeddings.position_embeddings.weight", (cfg.max_position_embeddings, he)))
    for i in range(cfg.num_hidden_layers):
        lp = f"{e}encoder.layer.{i}."
        specs += [
            (lp + "attention.self.query.weight", (he, he)), (lp + "attention.self.key.weight", (he, he)),
            (lp + "attention.self.value.weight", (he, he)),
            (lp + "attention.self.query.bias", (he,)), (lp + "attention.self.key.bias", (he,)),
            (lp + "attention.self.value.bias", (he,)),
            (lp + "attention.output.dense.weight", (he, he)), (lp + "attention.output.dense.bias", (he,)),
            (lp + "attention.LayerNorm.weight", (he,)), (lp + "attention.LayerNorm.bias", (he,)),
            (lp + "intermediate.dense.weight", (ffe, he)), (lp + "intermediate.dense.bias", (ffe,)),
            (lp + "output.dense.weight", (he, ffe)), (lp + "output.dense.bias", (he,)),
            (lp + "LayerNorm.weight", (he,)), (lp + "LayerNorm.bias", (he,)),
        ]
    specs += [(e + "encoder.emb_layer_norm_after.weight", (he,)), (e + "encoder.emb_layer_norm_after.bias", (he,))]
    return specs


def trainable_specs(llm: LlmConfig, dna: EncConfig, prot: EncConfig):
    """Flat order of the trainable group: [decayed matrices ...][no-decay gains/biases ...] so the ZeRO shard
    step needs at most two AdamW launches per rank."""
    pw, pb = projector_specs(llm, dna, prot)
    decay = llm_param_specs(llm) + pw
    no_decay = llm_norm_specs(llm) + pb
    assert all(not is_no_decay(n) for n, _ in decay) and all(is_no_decay(n) for n, _ in no_decay)
    return decay, no_decay
