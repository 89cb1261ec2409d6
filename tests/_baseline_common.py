"""Shared by the Enc-Head baseline tests: cases and synthetic weights of tests/golden/baseline_cls.npz."""
import json
import os

import numpy as np
import torch

from conftest import GOLD

PARTS = {"NT": (("backbone", "dna_rna"),), "ESM": (("backbone", "protein"),), "NT+ESM": (("nt", "dna_rna"), ("esm", "protein")),
         "NT+NT": (("nt1", "dna_rna"), ("nt2", "dna_rna")), "ESM+ESM": (("esm1", "protein"), ("esm2", "protein"))}


def load_gold():
    g = dict(np.load(os.path.join(GOLD, "baseline_cls.npz"), allow_pickle=False))
    cases = [tuple(c) for c in json.loads(str(g["meta/cases"]))]
    return g, cases


def baseline_state_dict(meta, mtype, num_labels, seed):
    """fp32 weights of one case: every key the reference's module holds on the path, values by name (synth_tensor)."""
    from molly_amd.config import EncConfig
    from molly_amd.params import enc_param_specs
    from molly_amd.synth import synth_state_dict
    shapes, cfgs, dim = {}, [], 0
    for attr, kind in PARTS[mtype]:
        cfg = EncConfig.from_dict(meta["config"][kind])
        cfgs.append(cfg)
        dim += cfg.hidden_size
        shapes.update({n: tuple(s) for n, s in enc_param_specs(cfg, attr + ".")})
    shapes["head.weight"] = (num_labels, dim)
    shapes["head.bias"] = (num_labels,)
    return synth_state_dict(shapes, seed), cfgs


def case_inputs(g, name, mtype):
    xs = [torch.from_numpy(g[f"{name}/x{i + 1}"]) for i in range(len(PARTS[mtype]))]
    return xs, torch.from_numpy(g[f"{name}/labels"])
