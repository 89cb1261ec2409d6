"""GPU: BASELINE configs[4] — Molly-8B LoRA inference (reference src/inference_lora.py), batch = 32 greedy decode on one
MI355X, fused encoder -> projector -> Qwen3 prefill — at its DEFINING size (36 layers, 4096 wide, real vocabulary, the
33-layer ESM2-650M-shaped protein encoder at K = 512, r = 64 adapters on all seven targets), through properties that do not
need a CPU run of an 8 B-parameter model:

  (1) live adapter == merged adapter: the same PEFT-layout checkpoint attached un-merged (PEFT's default, reference
      src/inference_lora.py:214-216) and merged into the base weights (this repo's default at load) give the same prefill logits;
  (2) KV-cache decode == re-forward: the logits of decode step t equal the last-position logits of a fresh prefill over the
      grown sequence (teacher forcing, so a bf16 argmax tie cannot derail the comparison);
  (3) left padding is invisible: row i of the 32-row left-padded batch (reference Test collate, src/dataset/omics_dataset.py:
      384-391) gives the logits of the same sample run alone without padding;
  (4) greedy generation is deterministic, stays inside the vocabulary, and the hipGraph replay of the decode step equals the
      eager step bit for bit.
Tolerances are stated where they are used; the measured values are printed (run with -s)."""
import json

import pytest
import torch

pytestmark = pytest.mark.gpu

B, T, K, NEW = 32, 1024, 512, 4


def _model(lora=None):
    import molly_amd
    from molly_amd import config as C
    cfg = C.molly("8b")
    cfg.dna_rna_project_token_num, cfg.protein_project_token_num = K, K
    m = molly_amd.OmicsOne(cfg)
    m.model = molly_amd.Qwen3ForCausalLM(cfg.text_config)
    m.dna_rna_model = molly_amd.EsmForMaskedLM(cfg.dna_rna_config)
    m.protein_model = molly_amd.EsmForMaskedLM(cfg.protein_config)
    return m.prepare("cuda", train_llm=False, train_mlp=False, random_init_seed=1234, lora=lora)


def _adapter(cfg, path, r=64, alpha=128):
    """A synthetic adapter in PEFT's file layout with NON-zero B (a trained adapter; PEFT's gaussian init leaves B = 0)."""
    from safetensors.torch import save_file
    from molly_amd.lora import TARGETS, target_dims
    dims = target_dims(cfg)
    g = torch.Generator().manual_seed(5)
    tens = {}
    for i in range(cfg.num_hidden_layers):
        for t in TARGETS:
            sub = "self_attn" if t in ("q_proj", "k_proj", "v_proj", "o_proj") else "mlp"
            fin, fout = dims[t]
            tens[f"base_model.model.model.layers.{i}.{sub}.{t}.lora_A.weight"] = (torch.randn(r, fin, generator=g) * 0.02).bfloat16()
            tens[f"base_model.model.model.layers.{i}.{sub}.{t}.lora_B.weight"] = (torch.randn(fout, r, generator=g) * 0.02).bfloat16()
    save_file(tens, str(path / "adapter_model.safetensors"))
    with open(path / "adapter_config.json", "w") as f:
        json.dump({"r": r, "lora_alpha": alpha, "target_modules": list(TARGETS), "peft_type": "LORA"}, f)


def _batch():
    """32 prompts of different lengths, each with one 512-residue protein span, LEFT-padded to T (the span start moves with the
    padding, like the reference's Test-mode collate)."""
    from molly_amd.synth import synth_batch
    b = synth_batch(B, T, [("protein", K)], seed=21)
    ids, mask = b["input_ids"].clone(), torch.ones(B, T, dtype=torch.long)
    info = [[dict(d) for d in row] for row in b["omic_info_list"]]
    pads = [(i * 37) % 301 for i in range(B)]                       # 0 .. 300, row 0 unpadded
    for i, p in enumerate(pads):
        if p:
            ids[i] = torch.cat([torch.full((p,), 151643), b["input_ids"][i, :-p]])
            mask[i, :p] = 0
            info[i][0]["start"] += p
            assert info[i][0]["start"] + K + 2 <= T
    return ids, mask, b["omic_ids"], info, pads


def _rel(a, b):
    return (a.float() - b.float()).abs().max().item() / b.float().abs().max().item()


def _rms(a, b):
    return ((a.float() - b.float()).square().mean().sqrt() / b.float().square().mean().sqrt()).item()


def test_config5_molly8b_batch32_lora_greedy_decode(tmp_path):
    from molly_amd.generate import GenerationSession
    from molly_amd.lora import LoraConfig, load_live_adapter, merge_lora_adapter, save_adapter
    live = _model(lora=LoraConfig(r=64, lora_alpha=128.0, lora_dropout=0.0))     # scaling 2: a dropped or squared alpha/r shows
    _adapter(live.text_config, tmp_path)
    assert load_live_adapter(live, str(tmp_path)) == 36 * 7
    ids, mask, omic, info, pads = _batch()
    V = live.text_config.vocab_size

    s_live = GenerationSession(live, max_new_tokens=NEW)
    lg_live = s_live.prefill(ids, mask, omic, info).clone()
    assert lg_live.shape == (B, V) and bool(torch.isfinite(lg_live).all())
    # the un-merged adapter is visible: without it the logits are different numbers
    lo = live._rt.llm.lora
    live._rt.llm.lora = None
    lg_base = GenerationSession(live, max_new_tokens=NEW).prefill(ids, mask, omic, info).clone()
    live._rt.llm.lora = lo
    d_adapter = _rel(lg_base, lg_live)
    assert d_adapter > 0.3, d_adapter                       # the adapter moves the logits by a large fraction of their range
    # live decode steps under teacher forcing (kept for the comparison with the merged model)
    forced = [lg_live.argmax(-1)]
    dec_live = []
    for t in range(NEW - 1):
        dec_live.append(s_live.step(forced[-1]).clone())
        forced.append(dec_live[-1].argmax(-1))
    w_probe = live._rt.W["model.model.layers.0.self_attn.q_proj.weight"][:64].clone()
    # what OmicsTrainer.save_model leaves under --use-lora (src/trainer/omics_trainer.py:89-103): the adapter as PEFT writes it
    # plus the two projector .bin files (the projectors are part of the LoRA run's trainable set, so they travel with it)
    ckpt = tmp_path / "ckpt"
    save_adapter(live, str(ckpt))
    del s_live, live, lo
    torch.cuda.empty_cache()

    merged = _model()
    assert torch.equal(merged._rt.W["model.model.layers.0.self_attn.q_proj.weight"][:64], w_probe)   # same base weights
    assert merge_lora_adapter(merged, str(ckpt)) == 36 * 7
    assert not torch.equal(merged._rt.W["model.model.layers.0.self_attn.q_proj.weight"][:64], w_probe)
    s = GenerationSession(merged, max_new_tokens=NEW)
    lg = s.prefill(ids, mask, omic, info).clone()
    # (1) live == merged.  Merging rounds W + s*B*A to bf16 once (2^-9 relative per weight), the live path rounds the adapter
    # branch's own activations instead: two bf16 computations of the same function, 36 decoder + 33 encoder layers deep, on a
    # random-init model.  Yardsticks: IDENTICAL weights run two ways (properties 2 and 3 below) differ by 3.0-3.4 % of max|logit|
    # at this depth (measured); the reference's own bf16-vs-fp32 distance 57 layers deep is 7.6 % (tests/test_gpu_config4.py).
    # Measured here: 5.2-5.5 %.  Bound: 8 %, and at most a quarter of what the adapter itself changes.
    e1, r1 = _rel(lg, lg_live), _rms(lg, lg_live)
    agree = (lg.argmax(-1) == lg_live.argmax(-1)).float().mean().item()
    # (2) decode == re-forward, step by step; and the merged decode follows the live decode
    e2, e1d = 0.0, 0.0
    new = torch.empty(B, 0, dtype=torch.long)
    cur = lg
    for t in range(NEW - 1):
        nxt = forced[t]
        new = torch.cat([new, nxt.cpu()[:, None]], 1)
        cur = s.step(nxt).clone()
        e1d = max(e1d, _rel(cur, dec_live[t]))
        grown_ids = torch.cat([ids, new], 1)
        grown_mask = torch.cat([mask, torch.ones(B, new.shape[1], dtype=torch.long)], 1)
        ref = GenerationSession(merged, max_new_tokens=1).prefill(grown_ids, grown_mask, omic, info)
        e2 = max(e2, _rel(cur, ref))
    # (3) left padding: rows 0 (no pad), 8 (296 pads) and 31 run alone, unpadded
    e3 = 0.0
    for i in (0, 8, 31):
        p = pads[i]
        inf = [[dict(info[i][0], start=info[i][0]["start"] - p)]]
        alone = GenerationSession(merged, max_new_tokens=1).prefill(ids[i:i + 1, p:], mask[i:i + 1, p:], omic[i:i + 1], inf)
        e3 = max(e3, _rel(lg[i:i + 1], alone))
    print(f"config 5: adapter moves the logits by {d_adapter:.3f}; live vs merged prefill {e1:.4f} (rms {r1:.4f}, argmax agreement {agree:.2f}) / decode {e1d:.4f}; decode vs re-forward {e2:.4f}; padded vs alone {e3:.4f} "
          f"(fractions of max|logit|)")
    assert e1 <= 8e-2 and e1d <= 8e-2 and e1 <= 0.25 * d_adapter, (e1, e1d, d_adapter)
    assert e2 <= 5e-2, e2                                   # measured 3.4 % (the 2-layer tiny model: <= 3 %)
    assert e3 <= 5e-2, e3                                   # measured 3.0 %
    # (4) greedy generation: deterministic, in range; graph replay == eager
    out1 = merged.generate(ids, mask, omic, info, do_sample=False, max_new_tokens=6)
    out2 = merged.generate(ids, mask, omic, info, do_sample=False, max_new_tokens=6)
    assert out1.shape == (B, 6) and out1.dtype == torch.int64 and torch.equal(out1, out2)
    assert int(out1.min()) >= 0 and int(out1.max()) < V
    assert torch.equal(out1[:, 0].cpu(), lg.argmax(-1).cpu())
    sg = GenerationSession(merged, max_new_tokens=NEW, use_graph=True)
    se = GenerationSession(merged, max_new_tokens=NEW, use_graph=False)
    a, b = sg.prefill(ids, mask, omic, info), se.prefill(ids, mask, omic, info)
    assert torch.equal(a, b)
    for t in range(NEW - 1):
        a, b = sg.step(forced[t]).clone(), se.step(forced[t]).clone()
        assert torch.equal(a, b), t
