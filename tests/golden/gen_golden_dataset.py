#!/usr/bin/env python3
"""G1 fixtures: run the REFERENCE's OmicsDataset + collate (reference: src/dataset/omics_dataset.py) over synthetic rows with
the toy tokenizers of molly_amd/data.py and dump the integer outputs.  Build-container only (imports /root/reference)."""
import json
import os
import sys
import tempfile

import pandas as pd
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from gen_golden import import_reference  # noqa: E402  (installs the dependency stubs)
from molly_amd.data import ToyOmicTokenizer, ToyTextTokenizer  # noqa: E402

ROWS = [
    dict(task="Solubility-Solubility", input="Is <protein>MKTAYIAKQRQISFVKSHFSRQ</protein> soluble?", think="", output="Yes.",
         label="1", kind="protein", task_num=3),
    dict(task="tf-h", input="No omics here at all, just text.", think="hmm", output="Nothing to see", label="0", kind="",
         task_num=1),
    dict(task="rna_protein_interaction", input="Does <protein>ACDEFGHIKL</protein> bind <rna>ACGUACGT</rna> or <rna>ACGTTTGACA</rna>?",
         think="", output="It binds the first.", label="1", kind="rna-protein", task_num=2),
    dict(task="promoter_enhancer_interaction", input="<dna>ACGTACGTACGTNNACGTAC</dna> vs <dna>TTTTGGGGCCCCAAAATTTTGGGG</dna>: interact?",
         think="", output="No", label="0", kind="dna-dna", task_num=7),
    dict(task="unknown-task", input="x " * 60 + "<protein>MKV</protein> tail " + "y " * 80, think="", output="z " * 40, label="",
         kind="protein", task_num=0),     # forces truncation at max_len
    dict(task="cpd-prom_core", input="  leading/trailing spaces <dna>acgtacgtac</dna>  ", think="", output="", label="",
         kind="dna", task_num=5),         # empty output
]


def main():
    import_reference()
    from dataset.omics_dataset import DatasetConfig, OmicsDataset, qwen_omics_collate_fn, qwen_omics_collate_fn_inference
    with tempfile.TemporaryDirectory() as td:
        pq = os.path.join(td, "rows.parquet")
        pd.DataFrame(ROWS).to_parquet(pq)
        out = {"rows": ROWS, "cases": []}
        for mode, typ, max_len, kd, kp in (("sft", None, 192, 8, 8), ("pretrain", None, 160, 10, 10), ("sft", "Test", 176, 12, 12)):
            cfg = DatasetConfig(max_len=max_len, mode=mode, cal_metric_pos=None, dna_rna_k_tokens=kd, protein_k_tokens=kp)
            ds = OmicsDataset(pq, ToyTextTokenizer(), cfg, dna_rna_tokenizer=ToyOmicTokenizer("dna"),
                              protein_tokenizer=ToyOmicTokenizer("protein"), type=typ)
            # rows without omics cannot be stacked (torch.stack of an empty list raises in the reference too): skip row 1
            idxs = [0, 2, 3, 4, 5]
            items = [ds[i] for i in idxs]
            case = {"mode": mode, "type": typ, "max_len": max_len, "kd": kd, "kp": kp, "idxs": idxs, "items": []}
            for it in items:
                case["items"].append({k: (v.tolist() if torch.is_tensor(v) else v) for k, v in it.items()})
            if kd == kp or typ == "Test":
                pass
            # collate only batches whose omic rows share K (the reference's pad_sequence needs equal trailing dims)
            groups = [[0, 1], [2]] if kd != kp else [[0, 1, 2]]
            case["batches"] = []
            for gsel in groups:
                sel = [ds[idxs[g]] for g in gsel]
                same_k = len({tuple(s["omic_ids"].shape[1:]) for s in sel}) == 1
                if not same_k:
                    continue
                b = (qwen_omics_collate_fn_inference if typ == "Test" else qwen_omics_collate_fn)(sel)
                case["batches"].append({"sel": gsel, **{k: (v.tolist() if torch.is_tensor(v) else v) for k, v in b.items()}})
            out["cases"].append(case)
    with open(os.path.join(ROOT, "tests", "golden", "dataset_g1.json"), "w") as f:
        json.dump(out, f)
    print("wrote dataset_g1.json:", [(c["mode"], c["type"], len(c["items"]), len(c["batches"])) for c in out["cases"]])


if __name__ == "__main__":
    main()
