"""GPU: the whole hot path (encoders -> projector -> injection -> Qwen3 fwd/bwd -> CE) through molly_amd.OmicsOne
against (a) the golden vectors produced by the reference's OmicsOne and (b) the CPU oracle on the same inputs."""
import numpy as np
import pytest
import torch

from conftest import tiny_batch, tiny_state_dict

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def build_tiny(meta, table_dtype=BF):
    import molly_amd
    from molly_amd.config import EncConfig, LlmConfig, OmicsModalConfig
    c = meta["config"]
    cfg = OmicsModalConfig(text_config=LlmConfig.from_dict(c["text"]), dna_rna_config=EncConfig.from_dict(c["dna_rna"]),
                           protein_config=EncConfig.from_dict(c["protein"]))
    cfg.dna_rna_project_token_num = cfg.protein_project_token_num = c["K"]
    m = molly_amd.OmicsOne(cfg)
    m.model = molly_amd.Qwen3ForCausalLM.from_config(cfg.text_config)
    m.dna_rna_model = molly_amd.EsmForMaskedLM.from_config(cfg.dna_rna_config)
    m.protein_model = molly_amd.EsmForMaskedLM.from_config(cfg.protein_config)
    sd = tiny_state_dict(meta)
    # state-dict key layout must be the reference's (minus the heads Molly never reads)
    mine = set(m.state_dict().keys())
    ref = set(sd.keys())
    assert mine <= ref, sorted(mine - ref)[:5]
    dead = [k for k in ref - mine if not any(t in k for t in ("lm_head", "contact_head"))]
    assert not dead, dead[:5]
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not missing, missing
    m.prepare("cuda", rope_table_dtype=table_dtype)
    return m


def test_forward_logits_and_loss_vs_reference_golden(tiny_meta, tiny_gold, tiny_gold_bf16):
    m = build_tiny(tiny_meta)
    batch = tiny_batch(tiny_gold, tiny_meta)
    with torch.no_grad():
        out = m(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], omic_ids=batch["omic_ids"],
                omic_info_list=batch["omic_info_list"], labels=batch["labels"])
    torch.cuda.synchronize()
    logits = out.logits.float().cpu().numpy()[:, ::2]
    ref = tiny_gold["fwd/logits"]
    refb = tiny_gold_bf16["logits"]
    # stated bf16 tolerance: max|dlogit| <= 3e-2 * max|logit|; the reference's own bf16 CPU path
    # sits at ~1.1e-2 * max|logit| from its fp32 path on this fixture.
    tol = 3e-2 * np.abs(ref).max()
    valid = batch["attention_mask"].numpy()[:, ::2].astype(bool)
    err = np.abs(logits - ref)[valid].max()
    errb = np.abs(refb - ref)[valid].max()
    print(f"max|dlogit| ours {err:.4f}  reference-bf16 {errb:.4f}  tol {tol:.4f}")
    assert err <= tol
    assert abs(out.loss.item() - float(tiny_gold["fwd/loss"])) <= 2e-3


def test_backward_grads_vs_reference_golden(tiny_meta, tiny_gold):
    m = build_tiny(tiny_meta)
    batch = tiny_batch(tiny_gold, tiny_meta)
    loss = m.forward_backward(batch["input_ids"], batch["attention_mask"], batch["omic_ids"], batch["omic_info_list"],
                              batch["labels"])
    torch.cuda.synchronize()
    assert abs(loss.item() - float(tiny_gold["fwd/loss"])) <= 2e-3
    G = m._rt.G.views
    names = [k[len("gnorm/"):] for k in tiny_gold if k.startswith("gnorm/")]
    worst = 0.0
    for n in names:
        if n == "model.lm_head.weight":
            continue
        g = G[n].float().cpu()
        ref_norm = float(tiny_gold["gnorm/" + n])
        # direction + magnitude on the stored head of each tensor; bf16 grads: 2% of the tensor's scale
        head = torch.from_numpy(tiny_gold["ghead/" + n])
        got = g.flatten()[:256]
        scale = max(head.abs().max().item(), 1e-8)
        rel = (got - head).abs().max().item() / scale
        worst = max(worst, rel)
        assert rel < 6e-2, (n, rel)
        assert abs(g.double().norm().item() - ref_norm) <= 3e-2 * ref_norm + 1e-6, (n, g.norm().item(), ref_norm)
    print("worst relative grad-head error", worst)


def test_matches_cpu_oracle_on_fresh_seeded_batch(tiny_meta):
    """Same seeded inputs through the HIP path and the oracle (not the stored fixture): ragged text + ragged omics."""
    from molly_amd.synth import synth_batch
    from oracle import molly_ref as R
    c = tiny_meta["config"]
    m = build_tiny(tiny_meta)
    sp = {k: tuple(v) for k, v in c["special_ids"].items()}
    batch = synth_batch(3, 384, [("protein", 64), ("rna", 64)], seed=7, text_vocab=1000, special_ids=sp, pad_id=1000,
                        ragged=True)
    with torch.no_grad():
        out = m(**{k: batch[k] for k in ("input_ids", "attention_mask", "omic_ids", "omic_info_list", "labels")})
    llm, dna, prot = R.cfgs_from_meta(c)
    sd = tiny_state_dict(tiny_meta)
    with torch.no_grad():
        loss, logits = R.omics_forward(sd, llm, dna, prot, batch, {"dna_rna": 64, "protein": 64})
    valid = batch["attention_mask"].bool()
    err = (out.logits.float().cpu() - logits)[valid].abs().max().item()
    assert err <= 3e-2 * logits.abs().max().item(), err
    assert abs(out.loss.item() - loss.item()) <= 2e-3


def test_error_behaviour_matches_reference(tiny_meta, tiny_gold):
    m = build_tiny(tiny_meta)
    batch = tiny_batch(tiny_gold, tiny_meta)
    bad = [[dict(d) for d in row] for row in batch["omic_info_list"]]
    bad[0][0]["type"] = "lipid"
    with pytest.raises(ValueError, match="Unsupported omic type"):
        m(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], omic_ids=batch["omic_ids"],
          omic_info_list=bad)
    with pytest.raises(AssertionError, match="Mismatch"):
        m(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], omic_ids=batch["omic_ids"],
          omic_info_list=[row[:1] for row in batch["omic_info_list"]])
    ids = batch["omic_ids"].clone()
    ids[0, 0, 3] = 40                       # protein vocab is 33
    with pytest.raises(AssertionError, match="out-of-range token"):
        m(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], omic_ids=ids,
          omic_info_list=batch["omic_info_list"])


def _shells(meta):
    import molly_amd
    from molly_amd.config import EncConfig, LlmConfig, OmicsModalConfig
    c = meta["config"]
    cfg = OmicsModalConfig(text_config=LlmConfig.from_dict(c["text"]), dna_rna_config=EncConfig.from_dict(c["dna_rna"]),
                           protein_config=EncConfig.from_dict(c["protein"]))
    cfg.dna_rna_project_token_num = cfg.protein_project_token_num = c["K"]
    m = molly_amd.OmicsOne(cfg)
    m.model = molly_amd.Qwen3ForCausalLM(cfg.text_config)                 # un-materialised shells, as a caller builds them
    m.dna_rna_model = molly_amd.EsmForMaskedLM(cfg.dna_rna_config)
    m.protein_model = molly_amd.EsmForMaskedLM(cfg.protein_config)
    return m


def test_reference_inference_call_sequence_on_the_shells(tiny_meta, tiny_gold):
    """src/inference_lora.py:243-250 verbatim: plain strict `load_state_dict(torch.load(...))` of a checkpoint with the
    reference's key set (encoder MaskedLM / contact heads included), `.to(torch.bfloat16).to(device)`, `.eval()`, then a
    forward — no prepare() call by the caller.  Must give the logits of the explicitly prepared model."""
    batch = tiny_batch(tiny_gold, tiny_meta)
    args = dict(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], omic_ids=batch["omic_ids"],
                omic_info_list=batch["omic_info_list"], labels=batch["labels"])
    m = _shells(tiny_meta)
    m.load_state_dict(tiny_state_dict(tiny_meta))
    m = m.to(torch.bfloat16).to(torch.device("cuda", 0))
    m.eval()
    assert m.model.model.layers[0].mlp.down_proj.weight.is_cuda
    with torch.no_grad():
        out = m(**args)
    assert m._rt.G is None                                               # .eval(): no trainable group was allocated
    ref = build_tiny(tiny_meta)
    with torch.no_grad():
        want = ref(**args)
    # the caller's .to(bf16) rounds the fp32 checkpoint exactly like prepare()'s copy into the bf16 flat buffer
    assert torch.equal(out.logits, want.logits) and torch.equal(out.loss, want.loss)
    # parameters now live in the flat buffers: a later load_state_dict copies INTO them (and is seen by the kernels)
    sd2 = {k: v * 0.5 for k, v in tiny_state_dict(tiny_meta).items()}
    m.load_state_dict(sd2)
    with torch.no_grad():
        out2 = m(**args)
    assert not torch.equal(out2.logits, out.logits)
    # a model that never received weights refuses to run
    hollow = _shells(tiny_meta).to(torch.bfloat16).to("cuda").eval()
    with pytest.raises(RuntimeError, match="no value"):
        hollow(**args)


def test_reference_freeze_then_train_call_sequence(tiny_meta, tiny_gold):
    """src/train.py:655-660 -> set_up_trainable_param(model, args) with --train-mlp only: the LLM and encoder sub-trees are
    re-registered as buffers (freeze_subtree).  `prepare_from_module_state` reads the trainable set off the module tree and
    must step exactly like `prepare(train_llm=False, train_mlp=True)`."""
    from test_shell_boundary import _apply_reference_flags
    batch = tiny_batch(tiny_gold, tiny_meta)
    a = [batch[k] for k in ("input_ids", "attention_mask", "omic_ids", "omic_info_list", "labels")]
    m = _shells(tiny_meta)
    m.load_state_dict(tiny_state_dict(tiny_meta))
    _apply_reference_flags(m, train_llm=False, train_mlp=True, train_bio=False)
    assert m.infer_trainable() == (False, True, False)
    m.prepare_from_module_state("cuda")
    loss = m.forward_backward(*a)
    ref = _shells(tiny_meta)
    ref.load_state_dict(tiny_state_dict(tiny_meta))
    ref.prepare("cuda", train_llm=False, train_mlp=True)
    loss_ref = ref.forward_backward(*a)
    torch.cuda.synchronize()
    assert torch.equal(loss, loss_ref) and torch.equal(m._rt.G.flat, ref._rt.G.flat)
    assert sorted(m._rt.G.views) == ["dna_rna_projector.bias", "dna_rna_projector.weight", "protein_projector.bias",
                                     "protein_projector.weight"]
    # the frozen sub-trees are still in the checkpoint, as buffers, and now point into HBM
    sd = m.state_dict()
    assert sd["model.model.layers.0.mlp.down_proj.weight"].is_cuda and len(sd) == len(ref.state_dict())
