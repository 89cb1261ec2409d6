#!/usr/bin/env python3
"""Trainable-set fixture.  Runs ONLY in the build container: imports the REFERENCE's `utils/tools.py` (stubbed optional deps,
SURVEY.md Appendix E) and runs its own `set_up_trainable_param` (reference: src/utils/tools.py:313-338, `freeze_subtree`
:277-311) and the LoRA target discovery loop of `pre_train_lora` (:352-361, restated here only as far as reading
`named_modules()` — peft itself is not importable) on the reference's `OmicsOne` over tiny HF sub-models, for every flag
combination its scripts use.  Records, per combination: which state-dict names are still Parameters that require grad, which
became buffers, and that `state_dict()` keeps every key.  -> tests/golden/trainable_sets.json

    python tests/golden/gen_golden_trainable.py
"""
import json
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as G  # noqa: E402


def main():
    torch.manual_seed(0)
    out = {"config": {k: G.TINY[k] for k in ("text", "dna_rna", "protein", "K")}, "cases": []}
    for train_llm, train_mlp, train_bio in ((True, True, False), (False, True, False), (True, True, True), (False, False, False),
                                           (True, False, False)):
        m, shapes = G.build_reference_model(G.TINY)
        import utils.tools as T                      # the reference's module (sys.path set by build_reference_model)
        keys_before = [k for k in m.state_dict().keys()]
        T.set_up_trainable_param(m, types.SimpleNamespace(train_llm=train_llm, train_mlp=train_mlp, train_bio=train_bio))
        params = {n: bool(p.requires_grad) for n, p in m.named_parameters()}
        buffers = [n for n, _ in m.named_buffers() if n in shapes]
        assert sorted(m.state_dict().keys()) == sorted(keys_before)
        out["cases"].append({"train_llm": train_llm, "train_mlp": train_mlp, "train_bio": train_bio,
                             "trainable": sorted(n for n, rg in params.items() if rg),
                             "frozen_parameters": sorted(n for n, rg in params.items() if not rg),
                             "buffers": sorted(buffers), "n_state_dict_keys": len(keys_before)})
    # LoRA target discovery (src/utils/tools.py:352-361): leaf names of nn.Linear modules of the LLM except lm_head
    m, _ = G.build_reference_model(G.TINY)
    targets, seen = [], set()
    for name, module in m.model.named_modules():
        if isinstance(module, torch.nn.Linear):
            t = name.split(".")[-1]
            if t != "lm_head" and t not in seen:
                targets.append(t)
                seen.add(t)
    out["lora_targets_discovered"] = targets
    with open(os.path.join(G.OUT, "trainable_sets.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote trainable_sets.json:", [(c["train_llm"], c["train_mlp"], c["train_bio"], len(c["trainable"]), len(c["buffers"]))
                                         for c in out["cases"]], targets)


if __name__ == "__main__":
    main()
