"""Deterministic synthetic weights and batches (no checkpoints / datasets exist offline).

Weights are drawn from numpy's PCG64 stream (stable across numpy versions by policy), keyed
by parameter NAME, so the reference-side fixture generator (tests/golden/gen_golden.py), the
CPU oracle and the HIP model all materialise bit-identical fp32 weights from (name, shape, seed)
without shipping a checkpoint.

Batch layout follows the reference collate contract (reference:
src/dataset/omics_dataset.py:451-503): input_ids/labels/attention_mask int64 [B,T],
omic_ids int64 [B,n_max,K], omic_info_list list[B] of list[n_max] of {"type","start"}.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, List, Tuple

import numpy as np
import torch

_NORM_TAGS = ("layernorm", "layer_norm", "rmsnorm", ".norm.", "_norm.", "LayerNorm")


def _is_norm_weight(name: str) -> bool:
    return any(t in name for t in _NORM_TAGS) and name.endswith("weight")


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int) -> torch.Tensor:
    """fp32 tensor for parameter `name`; independent stream per name so order never matters."""
    rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
    if name.endswith("inv_freq"):
        raise ValueError("inv_freq is derived, not synthesised")
    x = rng.standard_normal(size=tuple(shape), dtype=np.float32)
    if _is_norm_weight(name):
        x = 1.0 + 0.1 * x            # norm gains near 1 but not 1, so a dropped gain is caught
    elif name.endswith("bias"):
        x = 0.02 * x
    else:
        x = 0.02 * x                 # HF initializer_range
    return torch.from_numpy(np.ascontiguousarray(x))


# weight ties of the reference's sub-models: Qwen3 (tie_word_embeddings for 0.6B/1.7B/4B) and HF
# EsmForMaskedLM (lm_head.decoder tied to word_embeddings; the LM head is unused by Molly).
DEFAULT_TIES = (
    ("model.lm_head.weight", "model.model.embed_tokens.weight"),
    ("dna_rna_model.lm_head.decoder.weight", "dna_rna_model.esm.embeddings.word_embeddings.weight"),
    ("protein_model.lm_head.decoder.weight", "protein_model.esm.embeddings.word_embeddings.weight"),
)


def synth_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int,
                     tied: Iterable[Tuple[str, str]] = DEFAULT_TIES) -> Dict[str, torch.Tensor]:
    """`tied` = (alias, source) pairs, e.g. ("model.lm_head.weight", "model.model.embed_tokens.weight")."""
    out = {}
    alias = dict(tied)
    for name, shape in shapes.items():
        if name.endswith("inv_freq"):
            continue
        src = alias.get(name, name)
        out[name] = synth_tensor(src, tuple(shape), seed)
    return out


# ----------------------------------------------------------------------------------------------
# synthetic mixed omics/text batches (SURVEY.md §8d "Synthetic inputs")
# ----------------------------------------------------------------------------------------------
# 9 omic special tokens appended after Qwen's 151643 regular + 26 built-in specials
# (reference: src/train.py:73-85 adds them in this order).
SPECIAL_IDS = {
    "dna": (151669, 151670, 151671),       # start, end, pad
    "rna": (151672, 151673, 151674),
    "protein": (151675, 151676, 151677),
}


def synth_batch(B: int, T: int, spans: List[Tuple[str, int]], seed: int, text_vocab: int = 151643,
                enc_vocab: Dict[str, int] | None = None, ragged: bool = False,
                special_ids: Dict[str, Tuple[int, int, int]] | None = None, pad_id: int = 151643, mixed_k: bool = False):
    """One collated batch. `spans` = [(type, K), ...] per sample (all samples get the same span set,
    like the reference which pads every omic row to K = *_k_tokens, omics_dataset.py:430-444).
    Spans normally share one K (the reference's dataset stacks a sample's rows: omics_dataset.py:411).  `mixed_k=True` lifts
    that for BASELINE config 4 (protein K=1024 beside DNA K=1000): `omic_ids` is then a list (samples) of lists (rows) of
    1-D tensors — what `OmicsOne.process_omic_sequences` itself iterates over (reference src/model/omics_one.py:104-118)."""
    g = torch.Generator().manual_seed(seed)
    sp = special_ids or SPECIAL_IDS
    enc_vocab = enc_vocab or {"dna": 4100, "rna": 4100, "protein": 24}
    Ks = {k for _, k in spans}
    assert len(Ks) <= 1 or mixed_k, "collate stacks omic rows: one K per batch"
    K = max(Ks) if Ks else 0
    n_max = max(len(spans), 1)
    input_ids = torch.randint(0, text_vocab, (B, T), generator=g, dtype=torch.int64)
    labels = torch.full((B, T), -100, dtype=torch.int64)
    attention_mask = torch.ones((B, T), dtype=torch.int64)
    omic_ids = torch.ones((B, n_max, max(K, 1)), dtype=torch.int64)
    info: List[List[dict]] = []
    T_prompt = (3 * T) // 4
    for b in range(B):
        valid = T
        if ragged:
            valid = int(torch.randint(T // 2, T + 1, (1,), generator=g))
        tp = min(T_prompt, (3 * valid) // 4)
        row_info = []
        # place spans left to right inside the prompt
        cursor = 1
        budget = tp - sum(k + 2 for _, k in spans) - 1
        assert budget >= 0, "prompt too short for spans"
        for j, (typ, k) in enumerate(spans):
            slack = budget // max(len(spans) - j, 1)
            off = int(torch.randint(0, slack + 1, (1,), generator=g)) if slack > 0 else 0
            budget -= off
            start = cursor + off
            s_id, e_id, p_id = sp[typ]
            input_ids[b, start] = s_id
            input_ids[b, start + 1:start + 1 + k] = p_id
            input_ids[b, start + 1 + k] = e_id
            cursor = start + k + 2
            if typ == "protein":   # <cls>=0 ... <eos>=2, residues 4..23 (ESM vocab)
                body = torch.randint(4, 4 + enc_vocab["protein"] - 4, (k - 2,), generator=g)
                row = torch.cat([torch.tensor([0]), body, torch.tensor([2])])
            else:                  # NT: <cls>=3 then 6-mers 4..; optional tail pad with 1
                body = torch.randint(4, enc_vocab[typ], (k - 1,), generator=g)
                row = torch.cat([torch.tensor([3]), body])
            if ragged and k > 8:
                cut = int(torch.randint(k // 2, k + 1, (1,), generator=g))
                row[cut:] = 1
            omic_ids[b, j, :k] = row
            row_info.append({"type": typ, "start": start})
        while len(row_info) < n_max:
            row_info.append({"type": "pad", "start": -1})
        info.append(row_info)
        labels[b, tp:valid] = input_ids[b, tp:valid]
        if valid < T:
            input_ids[b, valid:] = pad_id
            attention_mask[b, valid:] = 0
    if mixed_k:
        omic_ids = [[omic_ids[b, j, :k].clone() for j, (_, k) in enumerate(spans)] for b in range(B)]
    return {"input_ids": input_ids, "labels": labels, "attention_mask": attention_mask,
            "omic_ids": omic_ids, "omic_info_list": info}
