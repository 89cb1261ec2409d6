#!/usr/bin/env python3
"""Golden vectors of the reference's Enc-Head baselines (reference: baselines/model.py:33-215, `BackboneWithClsHead`;
SURVEY.md 8f-4).  The REFERENCE class is imported from /root/reference/baselines/model.py and run on CPU in fp32 (its own
dtype, baselines/model.py:83,93); its checkpoint loader `AutoModelForMaskedLM.from_pretrained(path, device_map="cuda:0")`
is replaced by a function returning a tiny random-init HF model (there are no checkpoints and no GPU in this container) —
the loader, not the arithmetic.  Every weight is then overwritten with `molly_amd.synth.synth_state_dict`.
Runs ONLY in the build container.

    python tests/golden/gen_golden_baseline.py      # writes tests/golden/baseline_cls.npz
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import OUT, TINY  # noqa: E402

CASES = [  # (case name, model_type, multi_answer, num_labels)
    ("esm", "ESM", False, 3), ("nt", "NT", False, 2), ("nt_esm", "NT+ESM", False, 5), ("esm_esm", "ESM+ESM", False, 2),
    ("nt_nt", "NT+NT", False, 2), ("esm_multi", "ESM", True, 6),
]
B, K = 4, 48
SEED_W, SEED_B = 77, 78


def import_reference_baseline():
    spec = importlib.util.spec_from_file_location("ref_baselines_model", "/root/reference/baselines/model.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from transformers import AutoModelForMaskedLM, EsmConfig

    class Loader:
        @staticmethod
        def from_pretrained(name, **kw):
            c = EsmConfig(**TINY["dna_rna" if name == "nt" else "protein"])
            c._attn_implementation = "eager"
            return AutoModelForMaskedLM.from_config(c).to(kw.get("torch_dtype", torch.float32))

    mod.AutoModelForMaskedLM = Loader
    return mod.BackboneWithClsHead


def make_ids(kind, seed):
    """[B, K] ids: <cls> first, random residues / 6-mers, <eos>, right padding (pad id 1) of different lengths."""
    g = torch.Generator().manual_seed(seed)
    cfg = TINY["dna_rna" if kind == "nt" else "protein"]
    cls_id, eos_id = (3, 2) if kind == "nt" else (0, 2)
    lo, hi = (6, cfg["vocab_size"]) if kind == "nt" else (4, 24)
    ids = torch.full((B, K), cfg["pad_token_id"], dtype=torch.int64)
    for b in range(B):
        n = K if b == 0 else int(torch.randint(K // 3, K, (1,), generator=g))
        ids[b, 0] = cls_id
        ids[b, 1:n - 1] = torch.randint(lo, hi, (n - 2,), generator=g)
        ids[b, n - 1] = eos_id if kind != "nt" else ids[b, n - 2]          # NT has no <eos>
    if kind != "nt":
        ids[1, 3] = cfg["mask_token_id"]                                 # exercises the token-dropout rescale
    return ids


def main():
    from molly_amd.synth import synth_state_dict
    Ref = import_reference_baseline()
    dump = {}
    for name, mtype, multi, nl in CASES:
        torch.manual_seed(0)
        m = Ref(mtype, nt_model="nt", esm_model="esm", num_labels=nl, multi_answer=multi)
        sd = m.state_dict()
        shapes = {k: tuple(v.shape) for k, v in sd.items() if not k.endswith("inv_freq")}
        # HF ties every backbone's lm_head.decoder to its word embeddings: give both names the embedding's values
        ties = [(k, k.replace("lm_head.decoder.weight", "esm.embeddings.word_embeddings.weight")) for k in shapes
                if k.endswith("lm_head.decoder.weight")]
        new = synth_state_dict(shapes, SEED_W, tied=ties)
        missing, unexpected = m.load_state_dict(new, strict=False)
        assert not unexpected and all(k.endswith("inv_freq") for k in missing)
        m = m.float().train()
        kinds = [{"NT": "nt", "ESM": "esm"}[t] for t in mtype.split("+")]
        xs = [make_ids(k, SEED_B + i) for i, k in enumerate(kinds)]
        masks = [(x != 1).long() for x in xs]
        g = torch.Generator().manual_seed(SEED_B + 9)
        labels = (torch.rand(B, nl, generator=g) < 0.4).float() if multi else torch.randint(0, nl, (B,), generator=g)
        out = m(xs[0], xs[1] if len(xs) > 1 else None, masks[0], masks[1] if len(masks) > 1 else None, labels)
        out.loss.backward()
        dump[f"{name}/loss"] = np.float32(out.loss.item())
        dump[f"{name}/logits"] = out.logits.detach().numpy()
        dump[f"{name}/labels"] = labels.numpy()
        for i, x in enumerate(xs):
            dump[f"{name}/x{i + 1}"] = x.numpy()
        n = 0
        for pn, p in m.named_parameters():
            if "lm_head" in pn or "contact_head" in pn or p.grad is None:
                continue                     # heads of the MaskedLM wrapper: not on the path (zero / no gradient)
            gr = p.grad.detach().float()
            dump[f"{name}/gnorm/{pn}"] = np.float64(gr.double().norm().item())
            dump[f"{name}/ghead/{pn}"] = gr.flatten()[:128].numpy()
            n += 1
        print(f"{name}: {mtype} loss {out.loss.item():.6f}, {n} gradient tensors")
    dump["meta/seed_w"] = np.int64(SEED_W)
    import json
    dump["meta/cases"] = np.array(json.dumps([list(c) for c in CASES]))
    np.savez_compressed(os.path.join(OUT, "baseline_cls.npz"), **dump)


if __name__ == "__main__":
    main()
