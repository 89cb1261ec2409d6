#!/usr/bin/env python3
"""Golden-vector generator.  Runs ONLY in the build container (it imports /root/reference and the
installed HuggingFace `transformers`); its outputs (tests/golden/*.npz, *.json) are committed and are
the only thing that travels to the GPU box.

What it does: imports the REFERENCE's `OmicsOne` (reference: src/model/omics_one.py:10-233) with stubbed
optional deps (recipe: SURVEY.md Appendix E), attaches random-init HF sub-models exactly like the
reference's `--no-load-pretrained` mode (reference: src/train.py:107-116), overwrites every weight with
`molly_amd.synth.synth_state_dict` (so no checkpoint has to be shipped), runs forward / backward / one
AdamW step on CPU and dumps inputs + expected outputs.

    python tests/golden/gen_golden.py            # writes tests/golden/tiny_*.npz
"""
import importlib.machinery
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    sys.modules[name] = m
    return m


def import_reference():
    ds = _stub("deepspeed")
    _stub("deepspeed.ops")
    ds.ops = sys.modules["deepspeed.ops"]
    ds.comm = _stub("deepspeed.comm", is_initialized=lambda: False, get_rank=lambda: 0)
    _stub("swanlab")
    _stub("texttable")
    _stub("colorama")

    class _P:  # placeholder types for `from peft import ...`
        pass

    _stub("peft", LoraConfig=_P, get_peft_model=None, prepare_model_for_kbit_training=None, PeftModel=_P)
    from transformers.modeling_outputs import CausalLMOutputWithPast
    _stub("trainer", CausalLMOutputWithPast=CausalLMOutputWithPast)
    sys.path.insert(0, "/root/reference/src")
    from model.omics_one import OmicsOne  # noqa
    from model.config import OmicsModalConfig  # noqa
    return OmicsOne, OmicsModalConfig


# ----------------------------------------------------------------------------------------------
TINY = dict(
    text=dict(vocab_size=1024, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
              num_attention_heads=4, num_key_value_heads=2, head_dim=128, max_position_embeddings=4096,
              rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=True, attention_bias=False,
              pad_token_id=1000, eos_token_id=1001),
    dna_rna=dict(vocab_size=4105, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                 num_attention_heads=2, max_position_embeddings=80, position_embedding_type="absolute",
                 token_dropout=False, pad_token_id=1, mask_token_id=2, layer_norm_eps=1e-5,
                 emb_layer_norm_before=False, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0),
    protein=dict(vocab_size=33, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                 num_attention_heads=2, max_position_embeddings=1026, position_embedding_type="rotary",
                 token_dropout=True, pad_token_id=1, mask_token_id=32, layer_norm_eps=1e-5,
                 emb_layer_norm_before=False, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0),
    K=64, B=2, T=256, seed_w=1234, seed_b=42,
    special_ids={"dna": [1010, 1011, 1012], "rna": [1013, 1014, 1015], "protein": [1016, 1017, 1018]},
)


def build_reference_model(cfgd, dtype=torch.float32, attn="eager"):
    from transformers import AutoModelForCausalLM, AutoModelForMaskedLM, EsmConfig, Qwen3Config
    OmicsOne, OmicsModalConfig = import_reference()
    tc = Qwen3Config(**cfgd["text"])
    dc = EsmConfig(**cfgd["dna_rna"])
    pc = EsmConfig(**cfgd["protein"])
    for c in (tc, dc, pc):
        c._attn_implementation = attn
    cfg = OmicsModalConfig(text_config=tc, dna_rna_config=dc, protein_config=pc)
    cfg.dna_rna_project_token_num = cfg.protein_project_token_num = cfgd["K"]
    m = OmicsOne(cfg)
    m.model = AutoModelForCausalLM.from_config(tc)
    m.dna_rna_model = AutoModelForMaskedLM.from_config(dc)
    m.protein_model = AutoModelForMaskedLM.from_config(pc)
    from molly_amd.synth import synth_state_dict
    sd = m.state_dict()
    shapes = {k: tuple(v.shape) for k, v in sd.items() if not k.endswith("inv_freq")}
    new = synth_state_dict(shapes, cfgd["seed_w"])
    missing, unexpected = m.load_state_dict(new, strict=False)
    assert not unexpected and all(k.endswith("inv_freq") for k in missing), (missing, unexpected)
    m = m.to(dtype)
    return m, shapes


def make_batch(cfgd):
    from molly_amd.synth import synth_batch
    sp = {k: tuple(v) for k, v in cfgd["special_ids"].items()}
    # sample 0: protein + dna ; sample 1: rna + (pad row).  Built from two calls then merged so that
    # the pad-row / ragged / right-padded-text cases of the reference collate are all present.
    b0 = synth_batch(1, cfgd["T"], [("protein", cfgd["K"]), ("dna", cfgd["K"])], cfgd["seed_b"],
                     text_vocab=1000, special_ids=sp, pad_id=1000)
    b1 = synth_batch(1, cfgd["T"], [("rna", cfgd["K"])], cfgd["seed_b"] + 1, text_vocab=1000,
                     special_ids=sp, pad_id=1000, ragged=True)
    omic1 = torch.ones((1, 2, cfgd["K"]), dtype=torch.int64)
    omic1[:, :1] = b1["omic_ids"]
    batch = {
        "input_ids": torch.cat([b0["input_ids"], b1["input_ids"]]),
        "labels": torch.cat([b0["labels"], b1["labels"]]),
        "attention_mask": torch.cat([b0["attention_mask"], b1["attention_mask"]]),
        "omic_ids": torch.cat([b0["omic_ids"], omic1]),
        "omic_info_list": [b0["omic_info_list"][0], b1["omic_info_list"][0] + [{"type": "pad", "start": -1}]],
    }
    return batch


def grads_summary(model, out):
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach().float()
        out["gnorm/" + n] = np.float64(g.double().norm().item())
        out["ghead/" + n] = g.flatten()[:256].numpy().copy()
        if g.numel() <= 1 << 16:
            out["gfull/" + n] = g.numpy().copy()


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    cfgd = TINY
    model, shapes = build_reference_model(cfgd)
    model.train()
    # reference trainable set for `--train-llm --train-mlp` (reference: src/utils/tools.py:313-338):
    for p in model.dna_rna_model.parameters():
        p.requires_grad_(False)
    for p in model.protein_model.parameters():
        p.requires_grad_(False)
    batch = make_batch(cfgd)
    out = {}
    for k in ("input_ids", "labels", "attention_mask", "omic_ids"):
        out["in/" + k] = batch[k].numpy()
    info = batch["omic_info_list"]

    # ---- forward (fp32, eager attention = textbook path) -------------------------------------
    caught = {}
    h1 = model.model.model.register_forward_pre_hook(
        lambda mod, a, kw: caught.__setitem__("inputs_embeds", kw["inputs_embeds"].detach().clone()),
        with_kwargs=True)
    h2 = model.model.model.norm.register_forward_hook(
        lambda mod, a, o: caught.__setitem__("final_hidden", o.detach().clone()))
    layer_out = []
    hs = [l.register_forward_hook(lambda mod, a, o: layer_out.append(
        (o[0] if isinstance(o, tuple) else o).detach().clone())) for l in model.model.model.layers]
    enc_out = {}
    h3 = model.protein_model.esm.register_forward_hook(
        lambda mod, a, o: enc_out.__setitem__("protein", o.last_hidden_state.detach().clone()))
    h4 = model.dna_rna_model.esm.register_forward_hook(
        lambda mod, a, o: enc_out.__setitem__("dna_rna", o.last_hidden_state.detach().clone()))
    res = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"],
                omic_ids=batch["omic_ids"], omic_info_list=info, labels=batch["labels"])
    for h in [h1, h2, h3, h4] + hs:
        h.remove()
    out["fwd/loss"] = np.float64(res.loss.item())
    out["fwd/logits"] = res.logits.detach().numpy()[:, ::2].copy()       # every 2nd position
    out["fwd/inputs_embeds"] = caught["inputs_embeds"].numpy()
    out["fwd/final_hidden"] = caught["final_hidden"].numpy()
    for i, lo in enumerate(layer_out):
        out[f"fwd/layer{i}"] = lo.numpy()
    out["fwd/enc_protein"] = enc_out["protein"].numpy()
    out["fwd/enc_dna_rna"] = enc_out["dna_rna"].numpy()
    print("loss", res.loss.item(), "logits", tuple(res.logits.shape))

    # ---- backward ---------------------------------------------------------------------------
    res.loss.backward()
    grads_summary(model, out)

    # ---- one optimizer step, HF/DeepSpeed semantics -----------------------------
    # AdamW lr 3e-5 wd 1e-2 betas(.9,.999) eps 1e-8, decay only on non-bias / non-norm params
    # (HF: trainer.py get_decay_parameter_names), clip 1.0, linear warmup (step 1 of 10, ratio .1 ->
    # warmup_steps=1 -> lr at the first step = 0 with HF's scheduler; we record lr explicitly).
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    no_decay = [n for n, _ in named if n.endswith("bias") or "norm" in n.lower()]
    groups = [
        {"params": [p for n, p in named if n not in no_decay], "weight_decay": 1e-2},
        {"params": [p for n, p in named if n in no_decay], "weight_decay": 0.0},
    ]
    lr = 3e-5
    opt = torch.optim.AdamW(groups, lr=lr, betas=(0.9, 0.999), eps=1e-8)
    gn = torch.nn.utils.clip_grad_norm_([p for _, p in named], 1.0)
    out["opt/grad_norm"] = np.float64(gn.item())
    opt.step()
    for n, p in named:
        out["opt/pnorm/" + n] = np.float64(p.detach().double().norm().item())
        out["opt/phead/" + n] = p.detach().flatten()[:256].numpy().copy()
    out["opt/no_decay"] = np.array(sorted(no_decay))
    out["opt/lr"] = np.float64(lr)
    opt.zero_grad()
    # second forward after the step: pins the whole update through the loss
    with torch.no_grad():
        res2 = model(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"],
                     omic_ids=batch["omic_ids"], omic_info_list=info, labels=batch["labels"])
    out["opt/loss_after"] = np.float64(res2.loss.item())
    print("grad_norm", gn.item(), "loss_after", res2.loss.item())

    np.savez_compressed(os.path.join(OUT, "tiny_fp32.npz"), **out)
    with open(os.path.join(OUT, "tiny_meta.json"), "w") as f:
        json.dump({"config": cfgd, "omic_info_list": info,
                   "state_dict_shapes": {k: list(v) for k, v in shapes.items()},
                   "transformers": __import__("transformers").__version__,
                   "torch": torch.__version__}, f, indent=1)

    # ---- bf16 reference CPU path (tolerance fixture) ----------------------------
    model_bf, _ = build_reference_model(cfgd, dtype=torch.bfloat16)
    model_bf.eval()
    with torch.no_grad():
        rb = model_bf(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"],
                      omic_ids=batch["omic_ids"], omic_info_list=info, labels=batch["labels"])
    np.savez_compressed(os.path.join(OUT, "tiny_bf16.npz"),
                        loss=np.float64(rb.loss.item()),
                        logits=rb.logits.float().numpy()[:, ::2].copy())
    d = (rb.logits.float() - res.logits.detach()).abs()
    print("bf16 vs fp32 reference: loss", rb.loss.item(), "max|dlogit|", d.max().item(),
          "max|logit|", res.logits.abs().max().item())


if __name__ == "__main__":
    main()
