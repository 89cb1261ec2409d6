"""CPU: molly_amd.data (dataset + collate restatement) against integer outputs of the REFERENCE's OmicsDataset on the same
rows and tokenizers (tests/golden/gen_golden_dataset.py).  Everything here is int64 / int: bit-exact."""
import copy
import json
import os

import pytest
import torch

from molly_amd.data import (DatasetConfig, OmicsDataset, ToyOmicTokenizer, ToyTextTokenizer, convert_source_to_id,
                            qwen_omics_collate_fn, qwen_omics_collate_fn_inference)

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset_g1.json")


@pytest.fixture(scope="module")
def g1():
    with open(GOLD) as f:
        return json.load(f)


def _ds(g1, case):
    cfg = DatasetConfig(max_len=case["max_len"], mode=case["mode"], cal_metric_pos=None, dna_rna_k_tokens=case["kd"],
                        protein_k_tokens=case["kp"])
    return OmicsDataset(copy.deepcopy(g1["rows"]), ToyTextTokenizer(), cfg, dna_rna_tokenizer=ToyOmicTokenizer("dna"),
                        protein_tokenizer=ToyOmicTokenizer("protein"), type=case["type"])


def _plain(v):
    return v.tolist() if torch.is_tensor(v) else v


def test_items_bit_exact(g1):
    n = 0
    for case in g1["cases"]:
        ds = _ds(g1, case)
        for i, exp in zip(case["idxs"], case["items"]):
            got = ds[i]
            assert set(got.keys()) == set(exp.keys()), (case["mode"], case["type"], i)
            for k, v in exp.items():
                assert _plain(got[k]) == v, (case["mode"], case["type"], i, k)
                n += 1
    assert n > 100


def test_collate_bit_exact(g1):
    for case in g1["cases"]:
        ds = _ds(g1, case)
        for b in case["batches"]:
            sel = [ds[case["idxs"][j]] for j in b["sel"]]
            got = (qwen_omics_collate_fn_inference if case["type"] == "Test" else qwen_omics_collate_fn)(sel)
            for k, v in b.items():
                if k == "sel":
                    continue
                assert _plain(got[k]) == v, (case["mode"], case["type"], k)


def test_quirks_are_reproduced(g1):
    case = g1["cases"][0]
    ds = _ds(g1, case)
    it = ds[2]          # protein appears BEFORE the rna spans in the text, but omic ids are collected dna, rna, protein
    # ("ACGUACGT" has a U: the reference's RNA regex only accepts ACGTN, so that span is plain text)
    assert [d["type"] for d in it["omic_info_list"]] == ["protein", "rna"]
    assert it["omic_ids"][0, 0].item() == 3 and it["omic_ids"][1, 0].item() == 0      # row 0 = rna (<cls>=3), row 1 = protein
    tr = ds[4]          # truncated at max_len with eos; the span start is NOT re-checked
    assert len(tr["input_ids"]) == case["max_len"] and tr["input_ids"][-1].item() == ds.eos_id
    with pytest.raises(RuntimeError):
        ds[1]           # a row without omics cannot be stacked — the reference fails identically
    assert convert_source_to_id("xx-tf-m-yy") == 19 and convert_source_to_id("nope") == 100


def test_left_pad_shifts_starts(g1):
    case = [c for c in g1["cases"] if c["type"] == "Test"][0]
    ds = _ds(g1, case)
    it = ds[0]
    pad = int((it["attention_mask"] == 0).sum())
    assert pad > 0 and bool((it["input_ids"][:pad] == ds.pad_id).all())
    s = it["omic_info_list"][0]["start"]
    assert it["input_ids"][s].item() == ds.ids["protein"]["start"]
    assert it["input_ids"][s + 1 + case["kp"]].item() == ds.ids["protein"]["end"]
