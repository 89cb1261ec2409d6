"""Dataset / collate restatement — index exactness for the hot path's inputs.

Reproduces, bit for bit, the integer outputs of the reference's `OmicsDataset` and collate functions (reference:
src/dataset/omics_dataset.py:35-557): `input_ids`, `labels`, `attention_mask`, `omic_ids`, `omic_info_list[*].start`.
The loader stays host-side Python; tokenizers are injected (any object with the HuggingFace
call surface used below), so nothing here depends on `transformers`.

Reference quirks kept on purpose:
  * omic ids are collected in KIND order (all dna, then rna, then protein: :256-269) while `omic_info_list` is in TEXT
    order (:277) — the two lists pair by index downstream, so interleaved kinds pair "wrongly" exactly like upstream;
  * omic sequences are tokenised to `*_k_tokens` (:430-444), not to the `*_max_length` config fields;
  * truncation cuts the text at max_len but never re-checks span starts (:370-373);
  * `think` ids are tokenised and never used (:298-299);
  * Test mode left-pads and shifts every span start by the pad length (:384-391).
"""
from __future__ import annotations

import re
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch

SYSTEM_PROMPT = ("<|im_start|>system\nYou are a helpful knowledgeable and precise biomedical assistant.<|im_end|>\n"
                 "<|im_start|>user\n")
ASSISTANT_START = "<|im_end|>\n<|im_start|>assistant\n"

_TASK_IDS = [  # reference: convert_source_to_id, omics_dataset.py:172-214 (first match wins, in this order)
    "antibody_antigen", "cpd-prom_core", "CRISPROnTarget", "emp-H", "enhancer_activity", "Fluorescence-Fluorescence",
    "FunctionEC-FunctionEC", "Isoform-Isoform", "MeanRibosomeLoading-MeanRibosomeLoading", "Modification-Modification",
    "NoncodingRNAFamily-NoncodingRNAFamily", "pd-prom_300", "ProgrammableRNASwitches-ProgrammableRNASwitches",
    "promoter_enhancer_interaction", "rna_protein_interaction", "Solubility-Solubility", "Stability-Stability",
    "Thermostability-Thermostability", "tf-h", "tf-m"]


def convert_source_to_id(source: str) -> int:
    for i, key in enumerate(_TASK_IDS):
        if key in source:
            return i
    return 100


@dataclass
class DatasetConfig:
    """reference: src/dataset/omics_dataset.py:19-30 (same fields and defaults)."""
    max_len: int = 1024
    max_src_len: int = 1024
    mode: str = "sft"
    cal_metric_pos: int = -1
    padding: bool = True
    input_field: str = "input"
    output_field: str = "output"
    dna_rna_k_tokens: int = 128
    protein_k_tokens: int = 128
    type: str = ""


class OmicsDataset(torch.utils.data.Dataset):
    """`rows`: a parquet path, a pandas DataFrame or a list of dicts with the reference schema
    (task, input, think, output, label, kind, task_num — data_tools/write2parquet.py:111-114)."""

    _REGEX = {
        "dna": re.compile(r"<dna>\s*([ACGTNacgtn]+)\s*</dna>"),
        "rna": re.compile(r"<rna>\s*([ACGTNacgtn]+)\s*</rna>"),
        "protein": re.compile(r"<protein>\s*([ACDEFGHIKLMNPQRSTVWYBXZOU]+)\s*</protein>"),
    }

    def __init__(self, rows, tokenizer, dataset_config: DatasetConfig, dna_rna_tokenizer=None, protein_tokenizer=None,
                 read_nums=None, shuffle=False, seed=42, type=None, **kwargs):
        self.tokenizer, self.dna_rna_tokenizer, self.protein_tokenizer = tokenizer, dna_rna_tokenizer, protein_tokenizer
        c = self.dataset_config = dataset_config
        self.max_len, self.mode, self.cal_metric_pos, self.padding = c.max_len, c.mode, c.cal_metric_pos, c.padding
        self.dna_rna_project_token_num, self.protein_project_token_num = c.dna_rna_k_tokens, c.protein_k_tokens
        self.dataset_type = type
        t = tokenizer
        self.ids = {k: {p: t.convert_tokens_to_ids(f"<|{k}_{p}|>") for p in ("start", "pad", "end")}
                    for k in ("dna", "rna", "protein")}
        self.eos_id, self.pad_id = t.eos_token_id, t.pad_token_id
        self.system_prompt_ids = t.encode(SYSTEM_PROMPT, add_special_tokens=False)
        self.assistant_start_ids = t.encode(ASSISTANT_START, add_special_tokens=False)
        if isinstance(rows, str):
            if rows.endswith((".jsonl", ".json")):    # same schema, one JSON object per line (no parquet engine needed)
                import json as _json
                with open(rows) as f:
                    rows = [_json.loads(l) for l in f if l.strip()]
                if shuffle:
                    import pandas as pd
                    rows = pd.DataFrame(rows)
            else:
                import pandas as pd
                rows = pd.read_parquet(rows)
        if hasattr(rows, "to_dict"):          # DataFrame: same head()/shuffle semantics as the reference (:96-105)
            df = rows
            if read_nums:
                df = df.head(read_nums)
            if shuffle:
                df = df.sample(frac=1, random_state=np.random.default_rng(seed)).reset_index(drop=True)
            rows = df.to_dict("records")
        else:
            rows = list(rows)[:read_nums] if read_nums else list(rows)
            if shuffle:
                raise ValueError("shuffle of plain row lists is not defined by the reference; pass a DataFrame")
        self.rows = rows

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, idx: int):
        if idx < 0 or idx >= len(self.rows):
            raise IndexError(f"Index {idx} out of bounds for dataset with {len(self.rows)} items")
        out = self.process_sample(self.format_raw(self.rows[idx]))
        assert len(out["omic_ids"]) == len(out["omic_info_list"]), \
            f"Mismatch in Omic IDs and Omic info for sample {idx}: {len(out['omic_ids'])} vs {len(out['omic_info_list'])}"
        return out

    # reference: format_raw, :216-332
    def format_raw(self, sample: Dict[str, Any]) -> dict:
        enc = lambda s: self.tokenizer.encode(s, add_special_tokens=False)
        input_text = (sample.get("input", "") or "").strip()
        output_text = (sample.get("output", "") or "").strip()
        reasoning = (sample.get("think", "") or "").strip()
        seq_info, raw_seqs = [], []
        for kind in ("dna", "rna", "protein"):                     # KIND order (quirk)
            for m in self._REGEX[kind].finditer(input_text):
                seq_info.append({"type": kind, "start": m.start(), "end": m.end()})
                raw_seqs.append(m.group(1).upper())
        input_ids = list(self.system_prompt_ids)
        omic_info_list = []
        start = 0
        for info in sorted(seq_info, key=lambda x: x["start"]):    # TEXT order
            t = info["type"]
            input_ids.extend(enc(input_text[start:info["start"]]))
            omic_info_list.append({"type": t, "start": len(input_ids)})
            k = self.dna_rna_project_token_num if t in ("dna", "rna") else self.protein_project_token_num
            input_ids.append(self.ids[t]["start"])
            input_ids.extend([self.ids[t]["pad"]] * k)
            input_ids.append(self.ids[t]["end"])
            start = info["end"]
        if start < len(input_text):
            input_ids.extend(enc(input_text[start:]))
        out = {
            "input_ids": input_ids,
            "output_ids": enc(output_text) if output_text else [],
            "reasoning_token_ids": enc(reasoning) if reasoning else [],
            "omic_ids_list": [self._encode_sequence(s, seq_info[i]["type"]) for i, s in enumerate(raw_seqs)],
            "omic_info_list": omic_info_list,
            "task": sample.get("task", ""), "label": sample.get("label", ""),
        }
        if self.dataset_type == "Test":
            out.update(raw_input=input_text, raw_output=output_text)
        else:
            out.update(task_label=convert_source_to_id(sample.get("task")), task_num=sample.get("task_num"))
        return out

    # reference: process_sample, :335-418
    def process_sample(self, sample: Dict[str, Any]):
        input_ids = sample["input_ids"]
        cal_metric_pos, labels = None, None
        input_ids.extend(self.assistant_start_ids)
        output_ids = sample["output_ids"] if self.mode == "sft" else []
        if self.dataset_type != "Test":
            if self.mode == "pretrain":
                input_ids.append(self.eos_id)
            else:
                output_ids.append(self.eos_id)
            input_len = len(input_ids)
            input_ids.extend(output_ids)
            labels = ([-100] * input_len + output_ids) if self.mode == "sft" else input_ids.copy()
            if len(input_ids) > self.max_len:
                input_ids = input_ids[:self.max_len - 1] + [self.eos_id]
                labels = labels[:self.max_len - 1] + [self.eos_id]
            if self.cal_metric_pos is not None:
                cal_metric_pos = input_len + 1 + self.cal_metric_pos
            elif len(output_ids) > 0:
                cal_metric_pos = input_len + 1
        attention_mask = [1] * len(input_ids)
        if self.dataset_type == "Test":
            info = sample["omic_info_list"]
            pad_len = self.max_len - len(input_ids)
            if self.padding and pad_len > 0:
                input_ids[:0] = [self.pad_id] * pad_len                # LEFT pad
                attention_mask[:0] = [0] * pad_len
                for d in info:
                    d["start"] += pad_len
            return {"input_ids": torch.LongTensor(input_ids), "omic_ids": torch.stack(sample["omic_ids_list"]),
                    "omic_info_list": info, "attention_mask": torch.LongTensor(attention_mask), "task": sample["task"],
                    "raw_label": sample["label"], "raw_input": sample["raw_input"], "raw_output": sample["raw_output"]}
        pad_len = self.max_len - len(input_ids)
        if self.padding and pad_len > 0:
            input_ids.extend([self.pad_id] * pad_len)
            labels.extend([-100] * pad_len)
            attention_mask.extend([0] * pad_len)
        return {"input_ids": torch.LongTensor(input_ids), "omic_ids": torch.stack(sample["omic_ids_list"]),
                "omic_info_list": sample["omic_info_list"], "labels": torch.LongTensor(labels),
                "attention_mask": torch.LongTensor(attention_mask), "cal_metric_pos": cal_metric_pos,
                "task_label": torch.tensor(sample.get("task_label")), "task_num": torch.tensor(sample.get("task_num"))}

    # reference: _encode_sequence, :420-447
    def _encode_sequence(self, seq: str, seq_type: str) -> torch.Tensor:
        if not self.dna_rna_tokenizer:
            raise ValueError("DNA/RNA tokenizer is required but not provided")
        if not self.protein_tokenizer:
            raise ValueError("Protein tokenizer is required but not provided")
        if seq_type.lower() in ("dna", "rna"):
            tok, k = self.dna_rna_tokenizer, self.dna_rna_project_token_num
        elif seq_type.lower() == "protein":
            tok, k = self.protein_tokenizer, self.protein_project_token_num
        else:
            raise ValueError(f"Unsupported sequence type: {seq_type}")
        return tok(seq, padding="max_length", max_length=k, truncation=True, return_tensors="pt")["input_ids"].squeeze(0)


def _pad_info(omic_info_lists, n):
    for lst in omic_info_lists:
        if len(lst) < n:
            lst.extend([{"type": "pad", "start": -1}] * (n - len(lst)))


def qwen_omics_collate_fn(batch):
    """reference: src/dataset/omics_dataset.py:451-503."""
    pad = torch.nn.utils.rnn.pad_sequence
    omic_ids = [s.get("omic_ids", None) for s in batch]
    omic_ids = pad(omic_ids, batch_first=True, padding_value=1) if omic_ids else None
    info = [s.get("omic_info_list", []) for s in batch]
    _pad_info(info, omic_ids.shape[1])
    return {
        "input_ids": pad([s["input_ids"] for s in batch], batch_first=True, padding_value=0),
        "labels": pad([s["labels"] for s in batch], batch_first=True, padding_value=-100),
        "attention_mask": pad([s["attention_mask"] for s in batch], batch_first=True, padding_value=0),
        "omic_ids": omic_ids, "omic_info_list": info,
        "cal_metric_pos": [s.get("cal_metric_pos") for s in batch],
        "task_label": torch.stack([s.get("task_label") for s in batch]),
        "task_num": torch.stack([s.get("task_num") for s in batch]),
    }


def qwen_omics_collate_fn_inference(batch):
    """reference: src/dataset/omics_dataset.py:506-557."""
    pad = torch.nn.utils.rnn.pad_sequence
    omic_ids = [s.get("omic_ids", None) for s in batch]
    omic_ids = pad(omic_ids, batch_first=True, padding_value=1) if omic_ids else None
    info = [s.get("omic_info_list", []) for s in batch]
    _pad_info(info, omic_ids.shape[1])
    return {
        "input_ids": pad([s["input_ids"] for s in batch], batch_first=True, padding_value=0),
        "attention_mask": pad([s["attention_mask"] for s in batch], batch_first=True, padding_value=0),
        "omic_ids": omic_ids, "omic_info_list": info,
        "cal_metric_pos": [s.get("cal_metric_pos") for s in batch],
        "input": [s.get("raw_input") for s in batch], "raw_output": [s.get("raw_output") for s in batch],
        "raw_label": [s.get("raw_label") for s in batch], "raw_task": [s.get("task") for s in batch],
        "raw_kind": [s.get("kind") for s in batch],
    }


# ---- stand-in tokenizers (no vocab files exist offline; used by tests, fixtures and the mini launcher) ---------------
class ToyTextTokenizer:
    """Deterministic byte-level stand-in with the call surface the dataset uses.  ids: bytes 0..255, then specials."""
    SPECIALS = ["<|endoftext|>", "<|im_start|>", "<|im_end|>", "<|dna_start|>", "<|dna_end|>", "<|dna_pad|>",
                "<|rna_start|>", "<|rna_end|>", "<|rna_pad|>", "<|protein_start|>", "<|protein_end|>", "<|protein_pad|>"]

    def __init__(self):
        self.special = {t: 256 + i for i, t in enumerate(self.SPECIALS)}
        self._re = re.compile("(" + "|".join(re.escape(t) for t in self.SPECIALS) + ")")
        self.eos_token_id = self.special["<|im_end|>"]
        self.pad_token_id = self.special["<|endoftext|>"]
        self.vocab_size = 256 + len(self.SPECIALS)

    def convert_tokens_to_ids(self, tok):
        return self.special[tok]

    def encode(self, text, add_special_tokens=False):
        out = []
        for part in self._re.split(text):
            if part in self.special:
                out.append(self.special[part])
            else:
                out.extend(part.encode("utf-8"))
        return out

    def decode(self, ids, skip_special_tokens=True):
        inv = {v: k for k, v in self.special.items()}
        out, run = [], bytearray()
        for i in (int(x) for x in ids):
            if i < 256:
                run.append(i)
                continue
            if run:
                out.append(run.decode("utf-8", errors="replace"))
                run = bytearray()
            if not skip_special_tokens and i in inv:
                out.append(inv[i])
        if run:
            out.append(run.decode("utf-8", errors="replace"))
        return "".join(out)

    def batch_decode(self, batch, skip_special_tokens=True):
        return [self.decode(row, skip_special_tokens) for row in (batch.tolist() if hasattr(batch, "tolist") else batch)]


class ToyOmicTokenizer:
    """ESM-style stand-in: <cls> body <eos>? truncated/padded to max_length with pad id 1.
    protein: one id per residue (ESM-2 vocab order); dna: non-overlapping 6-mers, NT-style (<cls>=3)."""
    ESM = "LAGVSERTIDPKQNFYMHWCXBUZO"

    def __init__(self, kind: str):
        self.kind = kind

    def __call__(self, seq, padding="max_length", max_length=64, truncation=True, return_tensors="pt"):
        if self.kind == "protein":
            ids = [0] + [4 + self.ESM.index(c) if c in self.ESM else 3 for c in seq] + [2]
        else:
            body = []
            for i in range(0, len(seq) - len(seq) % 6, 6):
                v = 0
                for c in seq[i:i + 6]:
                    v = v * 4 + "ACGT".find(c) if c in "ACGT" else v * 4
                body.append(4 + v)
            body += [4096 + "ACGTN".index(c) if c in "ACGTN" else 4100 for c in seq[len(seq) - len(seq) % 6:]]
            ids = [3] + body
        ids = ids[:max_length]
        ids = ids + [1] * (max_length - len(ids))
        return {"input_ids": torch.tensor([ids], dtype=torch.int64)}
