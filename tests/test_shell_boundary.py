"""CPU: the HF-named parameter shells survive the calls the REFERENCE makes on its sub-models:
  * src/inference_lora.py:243-250 — plain `load_state_dict(torch.load(...))` (strict, no assign) of a checkpoint that carries
    the encoders' unused MaskedLM/contact heads, then `.to(torch.bfloat16).to(device).eval()`;
  * src/utils/tools.py:277-338 — `freeze_subtree` (Parameters re-registered as buffers) / `set_up_trainable_param`;
  * src/utils/tools.py:352-361 — LoRA target discovery over `named_modules()`.
The expected trainable sets come from the reference's own functions run in the build container
(tests/golden/trainable_sets.json, gen_golden_trainable.py)."""
import json
import os

import pytest
import torch
import torch.nn as nn

import molly_amd
from molly_amd.config import EncConfig, LlmConfig, OmicsModalConfig

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def fx():
    with open(os.path.join(GOLD, "trainable_sets.json")) as f:
        return json.load(f)


def build(fx):
    c = fx["config"]
    cfg = OmicsModalConfig(text_config=LlmConfig.from_dict(c["text"]), dna_rna_config=EncConfig.from_dict(c["dna_rna"]),
                           protein_config=EncConfig.from_dict(c["protein"]))
    cfg.dna_rna_project_token_num = cfg.protein_project_token_num = c["K"]
    m = molly_amd.OmicsOne(cfg)
    m.model = molly_amd.Qwen3ForCausalLM(cfg.text_config)
    m.dna_rna_model = molly_amd.EsmForMaskedLM(cfg.dna_rna_config)
    m.protein_model = molly_amd.EsmForMaskedLM(cfg.protein_config)
    return m


def reference_like_checkpoint(m, seed=3):
    """What the reference's OmicsTrainer.save_model writes: every key of our shells PLUS the HF encoders' dead heads."""
    g = torch.Generator().manual_seed(seed)
    sd = {k: torch.randn(v.shape, generator=g) * 0.02 for k, v in m.state_dict().items()}
    for pre, V, he in (("dna_rna_model.", m.dna_rna_config.vocab_size, m.dna_rna_config.hidden_size),
                       ("protein_model.", m.protein_config.vocab_size, m.protein_config.hidden_size)):
        sd[pre + "lm_head.bias"] = torch.zeros(V)
        sd[pre + "lm_head.dense.weight"] = torch.randn(he, he, generator=g)
        sd[pre + "lm_head.dense.bias"] = torch.zeros(he)
        sd[pre + "lm_head.layer_norm.weight"] = torch.ones(he)
        sd[pre + "lm_head.layer_norm.bias"] = torch.zeros(he)
        sd[pre + "lm_head.decoder.weight"] = torch.randn(V, he, generator=g)
        sd[pre + "esm.contact_head.regression.weight"] = torch.randn(1, 4, generator=g)
        sd[pre + "esm.contact_head.regression.bias"] = torch.zeros(1)
    sd["protein_model.esm.rotary_embeddings.inv_freq"] = torch.ones(8)
    return sd


def test_plain_strict_load_then_to_bf16_to_device_eval(fx):
    m = build(fx)
    assert all(p.is_meta for p in m.model.parameters())
    sd = reference_like_checkpoint(m)
    res = m.load_state_dict(sd)                                    # the reference's exact call: strict, no assign
    assert not res.missing_keys and not res.unexpected_keys
    assert not any(t.is_meta for t in m.state_dict().values())
    for k in ("model.model.layers.1.mlp.down_proj.weight", "protein_model.esm.encoder.layer.0.attention.self.key.bias",
              "dna_rna_projector.weight"):
        assert torch.equal(m.state_dict()[k], sd[k]), k            # the values arrived (no silent no-op)
    m2 = m.to(torch.bfloat16).to("cpu").eval()                     # src/inference_lora.py:249-250 (device = cpu here)
    assert m2 is m and not m.training
    assert m.model.model.layers[0].self_attn.q_proj.weight.dtype == torch.bfloat16
    # the checkpoint round-trips with the reference's exact key set (dead encoder heads included)
    assert sorted(m.state_dict().keys()) == sorted(sd.keys())


def test_to_device_leaves_unloaded_shells_alone_and_prepare_refuses_them(fx):
    m = build(fx)
    m.to(torch.bfloat16).to("cpu")                                 # must not raise on meta tensors
    with pytest.raises(RuntimeError, match="no value"):
        m.prepare("cuda")                                          # ... but nothing may run on them either
    sd = reference_like_checkpoint(m)
    partial = {k: v for k, v in sd.items() if "layers.1.mlp" not in k}
    m.load_state_dict(partial, strict=False)
    with pytest.raises(RuntimeError, match="layers.1.mlp"):
        m.prepare("cuda")


def _rebuffer(root: nn.Module, sub: str):
    """What the reference's freeze_subtree does to `root.<sub>`: every Parameter below it becomes a buffer of the same name."""
    top = getattr(root, sub)
    for name, p in list(top.named_parameters()):          # de-duplicated: of two tied names only the first is re-registered
        mod = top
        *path, leaf = name.split(".")
        for k in path:
            mod = mod[int(k)] if k.isdigit() else getattr(mod, k)
        delattr(mod, leaf)
        mod.register_buffer(leaf, p.detach())


def _apply_reference_flags(m, train_llm, train_mlp, train_bio):
    if not train_bio:
        _rebuffer(m, "dna_rna_model"); _rebuffer(m, "protein_model")
    if not train_mlp:
        _rebuffer(m, "dna_rna_projector"); _rebuffer(m, "protein_projector")
    if not train_llm:
        _rebuffer(m, "model")
    for sub, flag in (("dna_rna_model", train_bio), ("protein_model", train_bio), ("dna_rna_projector", train_mlp),
                      ("protein_projector", train_mlp), ("model", train_llm)):
        for p in getattr(m, sub).parameters():
            p.requires_grad = flag


def test_trainable_sets_match_the_references_set_up_trainable_param(fx):
    for case in fx["cases"]:
        m = build(fx)
        m.load_state_dict(reference_like_checkpoint(m))
        keys = set(nn.Module.state_dict(m).keys())
        _apply_reference_flags(m, case["train_llm"], case["train_mlp"], case["train_bio"])
        got = sorted(n for n, p in m.named_parameters() if p.requires_grad)
        want = sorted(n for n in case["trainable"] if n in keys)
        assert got == want, (case, set(got) ^ set(want))
        got_buf = sorted(n for n, _ in m.named_buffers())
        assert got_buf == sorted(n for n in case["buffers"] if n in keys)
        assert sorted(n for n, p in m.named_parameters() if not p.requires_grad) == \
            sorted(n for n in case["frozen_parameters"] if n in keys)      # e.g. the tied lm_head.weight of a frozen LLM
        assert set(nn.Module.state_dict(m).keys()) == keys          # buffers stay in the checkpoint, like the reference's
        assert m.infer_trainable() == (case["train_llm"], case["train_mlp"], case["train_bio"])


def test_lora_target_discovery_sees_the_same_linear_leaves(fx):
    m = build(fx)
    targets, seen = [], set()
    for name, module in m.model.named_modules():
        if isinstance(module, nn.Linear):
            t = name.split(".")[-1]
            if t != "lm_head" and t not in seen:
                targets.append(t); seen.add(t)
    assert targets == fx["lora_targets_discovered"]
    from molly_amd.lora import TARGETS
    assert sorted(TARGETS) == sorted(targets)
