"""ESM-architecture encoder forward (ESM-2 rotary / nucleotide-transformer v1 absolute positions) on the HIP kernels.

Restates what the reference runs through HF `EsmForMaskedLM(..., output_hidden_states=True)["hidden_states"][-1]`
(reference: src/model/omics_one.py:75-91): HF:models/esm/modeling_esm.py:224-271 (embeddings), :350-396 (self
attention: q*hd^-0.5 before rotary, SDPA scale 1), :412-439 / :442-463 / :517-521 (pre-LN blocks, erf-GELU FFN),
:552-553 (emb_layer_norm_after — `hidden_states[-1]` is this tensor).  The MaskedLM head the reference computes and
throws away is never run.  Encoders are frozen on Molly's default path (reference:
src/utils/tools.py:315-317), so only the forward exists here; `--train-bio` is SURVEY.md §8f-4.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .config import EncConfig
from .params import FlatBuffer
from .qwen3 import rope_tables

BF16 = torch.bfloat16


class EsmEngine:
    def __init__(self, cfg: EncConfig, params: FlatBuffer, device, prefix: str, rope_table_dtype=BF16):
        if cfg.hidden_dropout_prob or cfg.attention_probs_dropout_prob:
            raise NotImplementedError("encoder dropout > 0 is not implemented (ESM-2 / NT checkpoints ship 0.0)")
        self.cfg, self.P, self.dev, self.pre = cfg, params, device, prefix + "esm."
        self.he, self.nh = cfg.hidden_size, cfg.num_attention_heads
        self.hd = self.he // self.nh
        if self.hd not in (64, 128):
            raise NotImplementedError(f"encoder head_dim={self.hd}: the attention kernel is built for 64 and 128")
        self.ffe = cfg.intermediate_size
        self.rope_table_dtype = rope_table_dtype
        v = params.views
        e = self.pre
        self.wemb = v[e + "embeddings.word_embeddings.weight"]
        self.pemb = v.get(e + "embeddings.position_embeddings.weight") if cfg.position_embedding_type == "absolute" else None
        self.layers = []
        for i in range(cfg.num_hidden_layers):
            lp = f"{e}encoder.layer.{i}."
            self.layers.append(dict(
                qkv_w=params.span(lp + "attention.self.query.weight", lp + "attention.self.value.weight", self.he),
                qkv_b=params.span(lp + "attention.self.query.bias", lp + "attention.self.value.bias", 3 * self.he)[0],
                ao_w=v[lp + "attention.output.dense.weight"], ao_b=v[lp + "attention.output.dense.bias"],
                ln1_w=v[lp + "attention.LayerNorm.weight"], ln1_b=v[lp + "attention.LayerNorm.bias"],
                i_w=v[lp + "intermediate.dense.weight"], i_b=v[lp + "intermediate.dense.bias"],
                o_w=v[lp + "output.dense.weight"], o_b=v[lp + "output.dense.bias"],
                ln2_w=v[lp + "LayerNorm.weight"], ln2_b=v[lp + "LayerNorm.bias"]))
        self.lnf_w = v[e + "encoder.emb_layer_norm_after.weight"]
        self.lnf_b = v[e + "encoder.emb_layer_norm_after.bias"]
        self.cap = (0, 0)

    def reserve(self, n_seq: int, K: int):
        if self.cap == (n_seq, K):
            return
        N, he = n_seq * K, self.he
        e = lambda *s, dt=BF16: torch.empty(*s, dtype=dt, device=self.dev)
        self.x, self.x2, self.ln = e(N, he), e(N, he), e(N, he)
        self.qkv, self.qk, self.att = e(N, 3 * he), e(N, 2 * he), e(N, he)
        self.mid = e(N, self.ffe)
        self.out = e(N, he)
        self.pos = e(n_seq, K, dt=torch.int32)
        self.klen = e(n_seq, dt=torch.int32)
        self.klo = torch.zeros(n_seq, dtype=torch.int32, device=self.dev)
        if self.cfg.position_embedding_type == "rotary":
            self.cos, self.sin = rope_tables(K, self.hd, self.cfg.rope_theta, self.dev, self.rope_table_dtype)
        else:
            self.cos = self.sin = None
        self.cap = (n_seq, K)

    def forward(self, ids: torch.Tensor) -> torch.Tensor:
        """ids int64 [n_seq, K] on the GPU, pad id 1 = masked key (reference: src/model/omics_one.py:70).
        Returns the final-LayerNormed hidden states [n_seq*K, he] (bf16)."""
        cfg = self.cfg
        n_seq, K = ids.shape
        self.reserve(n_seq, K)
        N = n_seq * K
        ops.esm_embed(ids, self.wemb, self.pemb, self.x, self.pos, self.klen, cfg.pad_token_id, cfg.mask_token_id,
                      cfg.token_dropout)
        x, x2 = self.x, self.x2
        for w in self.layers:
            ops.layernorm_fwd(x, w["ln1_w"], w["ln1_b"], cfg.layer_norm_eps, out=self.ln)
            ops.gemm_nt(self.ln, w["qkv_w"], out=self.qkv, bias=w["qkv_b"])
            # q *= hd^-0.5, then rotary on q,k (positions = arange(K): HF:esm:733-737); absolute models: scale only
            ops.norm_rope_fwd(self.qkv, self.qk, self.nh, self.nh, self.hd, K, None, None, self.cos, self.sin,
                              q_scale=self.hd ** -0.5)
            ops.attn_fwd(self.qk[:, :self.he], self.qk[:, self.he:], self.qkv[:, 2 * self.he:], n_seq, K, self.nh, self.nh,
                         self.hd, 1.0, False, self.klo, self.klen, out=self.att, lse=False)
            ops.gemm_nt(self.att, w["ao_w"], out=x2, bias=w["ao_b"], res=x)
            ops.layernorm_fwd(x2, w["ln2_w"], w["ln2_b"], cfg.layer_norm_eps, out=self.ln)
            ops.gemm_nt(self.ln, w["i_w"], out=self.mid, bias=w["i_b"], gelu=True)
            ops.gemm_nt(self.mid, w["o_w"], out=x, bias=w["o_b"], res=x2)
        ops.layernorm_fwd(x, self.lnf_w, self.lnf_b, cfg.layer_norm_eps, out=self.out)
        return self.out
