"""GPU: BASELINE.json configs[0] — the reference's CPU-runnable "mini" run (Qwen3-0.6B backbone + ESM2-t6-8M protein encoder,
fp32 on CPU, 2 short protein-text samples) as a parity case at its REAL shapes: the HIP path in bf16 against the oracle in
fp32 (the CPU path the reference would run).  ESM2-t6-8M has 320 hidden / 20 heads = head_dim 16: the small-head attention
kernel; Qwen3-0.6B has hidden 1024 < n_heads*head_dim = 2048."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_mini_config_real_shapes_vs_cpu_oracle():
    import molly_amd
    from molly_amd import config as C
    from molly_amd.synth import synth_batch, synth_state_dict
    from oracle import molly_ref as R
    llm = C.qwen3("0.6b")
    esm8m = dict(vocab_size=33, hidden_size=320, intermediate_size=1280, num_hidden_layers=6, num_attention_heads=20,
                 max_position_embeddings=1026, position_embedding_type="rotary", token_dropout=True, pad_token_id=1,
                 mask_token_id=32, layer_norm_eps=1e-5, emb_layer_norm_before=False, hidden_dropout_prob=0.0,
                 attention_probs_dropout_prob=0.0)
    prot = C.EncConfig.from_dict(esm8m)
    dna = C.EncConfig.from_dict({**esm8m, "vocab_size": 4105, "position_embedding_type": "absolute", "token_dropout": False,
                                 "max_position_embeddings": 130, "mask_token_id": 2})
    cfg = C.OmicsModalConfig(text_config=llm, dna_rna_config=dna, protein_config=prot)
    cfg.dna_rna_project_token_num = cfg.protein_project_token_num = 64
    m = molly_amd.OmicsOne(cfg)
    m.model = molly_amd.Qwen3ForCausalLM(cfg.text_config)
    m.dna_rna_model = molly_amd.EsmForMaskedLM(cfg.dna_rna_config)
    m.protein_model = molly_amd.EsmForMaskedLM(cfg.protein_config)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = synth_state_dict(shapes, seed=7)
    m.load_state_dict(sd, strict=False, assign=True)
    m.prepare("cuda")
    b = synth_batch(2, 256, [("protein", 64)], seed=5, ragged=True)            # 2 short protein-text samples, real vocab
    args = [b[k] for k in ("input_ids", "attention_mask", "omic_ids", "omic_info_list", "labels")]
    with torch.no_grad():
        out = m(*args)
    loss_train = m.forward_backward(*args).item()
    torch.cuda.synchronize()

    o_llm = R.LlmCfg(**{k: getattr(llm, k) for k in R.LlmCfg.__dataclass_fields__})
    o_prot = R.EncCfg(**{k: getattr(prot, k) for k in R.EncCfg.__dataclass_fields__})
    o_dna = R.EncCfg(**{k: getattr(dna, k) for k in R.EncCfg.__dataclass_fields__})
    names = ["protein_projector.weight", "model.model.layers.27.mlp.down_proj.weight", "model.model.layers.0.self_attn.q_norm.weight"]
    leaves = {n: sd[n].clone().requires_grad_(True) for n in names}
    osd = dict(sd)
    osd.update(leaves)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    ref_loss, ref_logits = R.omics_forward(osd, o_llm, o_dna, o_prot, b, {"dna_rna": 64, "protein": 64})
    ref_loss.backward()
    valid = b["attention_mask"].bool()
    err = (out.logits.float().cpu() - ref_logits.detach())[valid].abs().max().item()
    scale = ref_logits.detach().abs().max().item()
    print(f"mini config: max|dlogit| {err:.4f} of max|logit| {scale:.3f}; loss {out.loss.item():.4f} vs {ref_loss.item():.4f}")
    assert err <= 3e-2 * scale                                                  # stated bf16 tolerance
    assert abs(out.loss.item() - ref_loss.item()) <= 5e-3 and abs(loss_train - ref_loss.item()) <= 5e-3
    for n, leaf in leaves.items():
        got = m._rt.G.views[n].float().cpu()
        rel = (got - leaf.grad).abs().max().item() / leaf.grad.abs().max().item()
        assert rel < 8e-2, (n, rel)
