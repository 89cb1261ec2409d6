"""Checkpoint and tokenizer loading for the launchers (reference: src/train.py:66-91 `setup_tokenizers`, :117-152
`from_pretrained` of the three sub-models; src/inference_lora.py:125-159, 181-205).

HF model directories ship `model.safetensors` (single file or shards listed in `model.safetensors.index.json`); older ones
`pytorch_model.bin` (or `pytorch_model.bin.index.json` shards).  All four layouts are read; the tensors go through the shell's
`load_state_dict`, which accepts-and-keeps the heads Molly never runs (MaskedLM / contact head) and reports anything the
forward NEEDS but the checkpoint lacks — `prepare()` then refuses to run on the hole (no silent random init)."""
from __future__ import annotations

import json
import os
from typing import Dict

import torch

OMIC_SPECIAL_TOKENS = ["<|dna_start|>", "<|dna_pad|>", "<|dna_end|>", "<|rna_start|>", "<|rna_pad|>", "<|rna_end|>",
                       "<|protein_start|>", "<|protein_pad|>", "<|protein_end|>"]        # order of src/train.py:73-83


def _read_file(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path, device="cpu")
    return torch.load(path, map_location="cpu")


def read_checkpoint_dir(path: str) -> Dict[str, torch.Tensor]:
    """All tensors of an HF model directory (or of a single weights file)."""
    if os.path.isfile(path):
        return _read_file(path)
    for single in ("model.safetensors", "pytorch_model.bin"):
        f = os.path.join(path, single)
        if os.path.exists(f):
            return _read_file(f)
    for index in ("model.safetensors.index.json", "pytorch_model.bin.index.json"):
        f = os.path.join(path, index)
        if os.path.exists(f):
            with open(f) as fh:
                shards = sorted(set(json.load(fh)["weight_map"].values()))
            sd = {}
            for s in shards:
                sd.update(_read_file(os.path.join(path, s)))
            return sd
    raise FileNotFoundError(f"{path}: no model.safetensors[.index.json] / pytorch_model.bin[.index.json]; "
                            "pass --no-load-pretrained for random init")


def load_pretrained(shell, path: str, what: str = "model", log=print):
    """`AutoModel*.from_pretrained(path)` for a parameter shell: load, then name every tensor the forward reads that the
    checkpoint did not provide (a tied lm_head is satisfied by the embedding)."""
    sd = read_checkpoint_dir(path)
    res = shell.load_state_dict(sd, strict=False)
    hollow = [n for n, t in torch.nn.Module.state_dict(shell).items() if t.is_meta]
    if hollow:
        raise RuntimeError(f"{what}: checkpoint {path} lacks {len(hollow)} tensors the forward reads (first: {hollow[:4]})")
    extra = [k for k in res.unexpected_keys]
    log(f"[molly_amd] {what}: {len(sd)} tensors from {path}" + (f", {len(extra)} unknown keys ignored (first: {extra[:3]})" if extra else ""))
    return res


def has_tokenizer_files(path) -> bool:
    return isinstance(path, str) and os.path.isdir(path) and any(
        os.path.exists(os.path.join(path, f)) for f in ("tokenizer.json", "tokenizer_config.json", "vocab.txt", "vocab.json"))


def setup_tokenizers(text_path, dna_rna_path, protein_path, log=print):
    """reference: src/train.py:66-91.  Model directories with tokenizer files -> the real tokenizers through
    `transformers.AutoTokenizer` (+ the 9 omic special tokens, same order); shape presets / bare config dirs (offline smoke
    runs with --no-load-pretrained) -> the deterministic stand-ins of molly_amd.data, said out loud.  Real weights with
    stand-in tokenizers would be garbage in, garbage out — that combination is refused by the launchers."""
    from .data import ToyOmicTokenizer, ToyTextTokenizer
    real = [has_tokenizer_files(p) for p in (text_path, dna_rna_path, protein_path)]
    if all(real):
        from transformers import AutoTokenizer
        tok = AutoTokenizer.from_pretrained(text_path, trust_remote_code=True)
        tok.add_special_tokens({"additional_special_tokens": OMIC_SPECIAL_TOKENS})
        return (tok, AutoTokenizer.from_pretrained(dna_rna_path, trust_remote_code=True),
                AutoTokenizer.from_pretrained(protein_path, trust_remote_code=True), True)
    log("[molly_amd] no tokenizer files under the model paths: using the deterministic stand-in tokenizers (smoke runs only)")
    return ToyTextTokenizer(), ToyOmicTokenizer("dna"), ToyOmicTokenizer("protein"), False
