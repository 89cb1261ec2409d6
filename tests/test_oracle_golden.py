"""CPU: the oracle (oracle/molly_ref.py) against golden vectors produced by the REFERENCE's OmicsOne
over HF modules (tests/golden/gen_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from conftest import tiny_batch, tiny_state_dict
from oracle import molly_ref as R


def _setup(meta, gold, grad=False):
    llm, dna, prot = R.cfgs_from_meta(meta["config"])
    sd = tiny_state_dict(meta)
    if grad:
        for k, v in sd.items():
            if k.startswith("model.") or "projector" in k:
                v.requires_grad_(True)
        sd["model.lm_head.weight"] = sd["model.model.embed_tokens.weight"]
    K = meta["config"]["K"]
    return sd, llm, dna, prot, tiny_batch(gold, meta), {"dna_rna": K, "protein": K}


def test_forward_matches_reference(tiny_meta, tiny_gold):
    sd, llm, dna, prot, batch, kt = _setup(tiny_meta, tiny_gold)
    col = {}
    with torch.no_grad():
        loss, logits = R.omics_forward(sd, llm, dna, prot, batch, kt, collect=col)
    g = tiny_gold
    np.testing.assert_allclose(col["enc"]["protein"].numpy(), g["fwd/enc_protein"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(col["enc"]["dna_rna"].numpy(), g["fwd/enc_dna_rna"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(col["inputs_embeds"].numpy(), g["fwd/inputs_embeds"], rtol=0, atol=1e-5)
    for i, h in enumerate(col["layers"]):
        np.testing.assert_allclose(h.numpy(), g[f"fwd/layer{i}"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(col["final_hidden"].numpy(), g["fwd/final_hidden"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(logits.numpy()[:, ::2], g["fwd/logits"], rtol=0, atol=5e-5)
    assert abs(loss.item() - float(g["fwd/loss"])) < 1e-5


def test_backward_and_adamw_match_reference(tiny_meta, tiny_gold):
    sd, llm, dna, prot, batch, kt = _setup(tiny_meta, tiny_gold, grad=True)
    loss, _ = R.omics_forward(sd, llm, dna, prot, batch, kt)
    loss.backward()
    g = tiny_gold
    names = [k[len("gnorm/"):] for k in g if k.startswith("gnorm/")]
    assert names, "fixture has no grads"
    params = {}
    for n in names:
        p = sd[n]
        assert p.grad is not None, n
        params[n] = p
        ref = float(g["gnorm/" + n])
        got = p.grad.double().norm().item()
        assert abs(got - ref) <= 1e-4 * max(ref, 1e-6) + 1e-7, (n, got, ref)
        np.testing.assert_allclose(p.grad.flatten()[:256].numpy(), g["ghead/" + n], rtol=0,
                                   atol=2e-6 + 1e-4 * np.abs(g["ghead/" + n]).max())
    # one clipped AdamW step with HF's decay/no-decay split
    total, coef = R.clip_coef([p.grad for p in params.values()], 1.0)
    assert abs(total.item() - float(g["opt/grad_norm"])) < 1e-4
    nd = set(g["opt/no_decay"].tolist())
    with torch.no_grad():
        for n, p in params.items():
            assert R.is_no_decay(n) == (n in nd), n
            gr = p.grad * coef
            m = torch.zeros_like(p)
            v = torch.zeros_like(p)
            R.adamw_step(p, gr, m, v, 1, float(g["opt/lr"]), 0.0 if R.is_no_decay(n) else 1e-2)
            np.testing.assert_allclose(p.flatten()[:256].numpy(), g["opt/phead/" + n], rtol=0, atol=1e-7)
            assert abs(p.double().norm().item() - float(g["opt/pnorm/" + n])) < 1e-5
        loss2, _ = R.omics_forward(sd, llm, dna, prot, batch, kt)
    assert abs(loss2.item() - float(g["opt/loss_after"])) < 1e-5


def test_index_outputs_bit_exact():
    ids = torch.tensor([[3, 5, 6, 1, 1], [1, 4, 1, 7, 8]])
    assert R.esm_position_ids(ids, 1).tolist() == [[2, 3, 4, 1, 1], [1, 2, 1, 3, 4]]


def test_group_omics_errors():
    with pytest.raises(ValueError):
        R.group_omics([[torch.ones(4, dtype=torch.long)]], [[{"type": "lipid", "start": 3}]])


def test_oracle_greedy_decode_reproduces_the_reference_generate(tiny_meta):
    """G7: tokens produced by the REFERENCE's `OmicsOne.generate(do_sample=False)` on left-padded prompts
    (tests/golden/generate_g7.json <- tests/golden/gen_golden_generate.py).  The oracle's re-forward with HF generate's
    position ids must pick the same token at every step, and see the same top-1/top-2 margin."""
    import json
    import os
    import torch
    from conftest import tiny_state_dict
    from oracle import molly_ref as R
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "generate_g7.json")))
    c = tiny_meta["config"]
    llm, dna, prot = R.cfgs_from_meta(c)
    sd = tiny_state_dict(tiny_meta)
    batch = {"input_ids": torch.tensor(g["input_ids"]), "attention_mask": torch.tensor(g["attention_mask"]),
             "omic_ids": torch.tensor(g["omic_ids"]), "omic_info_list": g["omic_info_list"]}
    ref = torch.tensor(g["new_tokens"])
    with torch.no_grad():
        for t in range(g["n_new"]):
            lg = R.generate_last_logits(sd, llm, dna, prot, batch, ref[:, :t], {"dna_rna": c["K"], "protein": c["K"]})
            top = lg.topk(2, dim=-1)
            assert torch.equal(top.indices[:, 0], ref[:, t]), t
            margin = top.values[:, 0] - top.values[:, 1]
            assert torch.allclose(margin, torch.tensor(g["margins"][t]), atol=2e-3), (t, margin)


def test_oracle_encoder_gradients_match_reference_trainbio_golden(tiny_meta, tiny_gold):
    """Gradients of every encoder tensor from the REFERENCE's autograd (tests/golden/tiny_trainbio.npz <-
    gen_golden_trainbio.py, `--train-bio` semantics) against the oracle's autograd on the same batch."""
    import os
    import numpy as np
    import torch
    from conftest import GOLD, tiny_batch, tiny_state_dict
    from oracle import molly_ref as R
    g = dict(np.load(os.path.join(GOLD, "tiny_trainbio.npz"), allow_pickle=False))
    sd = tiny_state_dict(tiny_meta)
    names = [k[len("gnorm/"):] for k in g if k.startswith("gnorm/")]
    assert len(names) >= 60 and all(n in sd for n in names)
    leaves = {n: sd[n].clone().requires_grad_(True) for n in names}
    sd.update(leaves)
    c = tiny_meta["config"]
    llm, dna, prot = R.cfgs_from_meta(c)
    loss, _ = R.omics_forward(sd, llm, dna, prot, tiny_batch(tiny_gold, tiny_meta), {"dna_rna": c["K"], "protein": c["K"]})
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) <= 2e-5
    for n in names:
        got = leaves[n].grad
        ref_norm = float(g["gnorm/" + n])
        assert abs(got.double().norm().item() - ref_norm) <= 2e-4 * ref_norm + 1e-9, n
        head = torch.from_numpy(g["ghead/" + n])
        assert (got.flatten()[:256] - head).abs().max().item() <= 2e-4 * max(head.abs().max().item(), 1e-9) + 1e-9, n


def test_oracle_lora_branch_equals_merged_weights(tiny_meta, tiny_gold):
    """oracle/molly_ref.py::lora_linear (PEFT's y = W x + (alpha/r) B A x): an adapter on every target gives the same
    logits as the base model with W + (alpha/r) B A merged in — the identity `merge_lora_adapter` relies on."""
    import torch
    from conftest import tiny_batch, tiny_state_dict
    from molly_amd.lora import TARGETS, lora_name, target_dims
    from molly_amd.config import LlmConfig
    from oracle import molly_ref as R
    c = tiny_meta["config"]
    llm, dna, prot = R.cfgs_from_meta(c)
    sd = tiny_state_dict(tiny_meta)
    dims = target_dims(LlmConfig.from_dict(c["text"]))
    g = torch.Generator().manual_seed(0)
    r, scaling = 8, 2.0
    lora, merged = dict(sd), dict(sd)
    lora["lora.scaling"] = scaling
    for i in range(llm.num_hidden_layers):
        for mod in TARGETS:
            fin, fout = dims[mod]
            A = torch.randn(r, fin, generator=g) * 0.05
            B = torch.randn(fout, r, generator=g) * 0.05
            lora[lora_name(i, mod, "A")] = A
            lora[lora_name(i, mod, "B")] = B
            wname = lora_name(i, mod, "A").replace(".lora_A.weight", ".weight")
            merged[wname] = sd[wname] + scaling * (B @ A)
    b = tiny_batch(tiny_gold, tiny_meta)
    k = {"dna_rna": c["K"], "protein": c["K"]}
    with torch.no_grad():
        l0, g0 = R.omics_forward(sd, llm, dna, prot, b, k)
        l1, g1 = R.omics_forward(lora, llm, dna, prot, b, k)
        l2, g2 = R.omics_forward(merged, llm, dna, prot, b, k)
    assert (g1 - g2).abs().max().item() <= 2e-4 * g2.abs().max().item() and abs(l1.item() - l2.item()) <= 1e-5
    assert (g1 - g0).abs().max().item() > 1e-2                       # and the adapter really changes the function


def test_enc_head_baselines_match_reference(tiny_meta):
    """Enc-Head baselines: the oracle's `cls_head_forward` against the reference's own
    `BackboneWithClsHead` (tests/golden/gen_golden_baseline.py): logits, loss and every parameter gradient, all five
    backbone combinations and the BCE (multi_answer) head."""
    from _baseline_common import baseline_state_dict, case_inputs, load_gold
    g, cases = load_gold()
    assert len(cases) >= 6
    for name, mtype, multi, nl in cases:
        sd, _ = baseline_state_dict(tiny_meta, mtype, nl, int(g["meta/seed_w"]))
        cfgs = [c for c in R.cfgs_from_meta(tiny_meta["config"])[1:]]                 # (dna_rna, protein) oracle configs
        by_kind = {"NT": cfgs[0], "ESM": cfgs[1]}
        ocfgs = [by_kind[t] for t in mtype.split("+")]
        for v in sd.values():
            v.requires_grad_(True)
        xs, labels = case_inputs(g, name, mtype)
        loss, logits = R.cls_head_forward(sd, mtype, ocfgs, xs, labels, multi_answer=bool(multi))
        np.testing.assert_allclose(logits.detach().numpy(), g[f"{name}/logits"], rtol=0, atol=2e-5)
        assert abs(loss.item() - float(g[f"{name}/loss"])) < 1e-5
        loss.backward()
        names = [k[len(name) + 7:] for k in g if k.startswith(name + "/gnorm/")]
        assert len(names) >= 37
        for n in names:
            ref = float(g[f"{name}/gnorm/{n}"])
            grad = sd[n].grad
            got = 0.0 if grad is None else grad.double().norm().item()
            assert abs(got - ref) <= 1e-4 * max(ref, 1e-6) + 1e-7, (name, n, got, ref)
            if grad is not None:
                np.testing.assert_allclose(grad.flatten()[:128].numpy(), g[f"{name}/ghead/{n}"], rtol=0,
                                           atol=2e-6 + 1e-4 * np.abs(g[f"{name}/ghead/{n}"]).max())


@pytest.mark.parametrize("tag", ["wide", "wide4b", "wide8b"])
def test_real_width_layer_matches_reference(tag):
    """G4: one layer of every stack at real widths — Molly-1.7B (2048 / 16q-8kv x 128 / 6144, tied head),
    4B (2560 / 32q-8kv / 9728) and 8B (4096 / 32q-8kv / 12288), untied heads; encoders 1280 / 20 x 64 / 5120 — reference
    OmicsOne in fp32: forward tensors, loss, all 56 / 57 gradients (encoders included)."""
    from conftest import wide_fixture
    meta, g = wide_fixture(tag)
    llm, dna, prot = R.cfgs_from_meta(meta["config"])
    sd = tiny_state_dict(meta)
    tied = meta["config"]["text"]["tie_word_embeddings"]
    for k, v in sd.items():
        if ("lm_head" not in k and "contact_head" not in k) or (k == "model.lm_head.weight" and not tied):
            v.requires_grad_(True)
    if tied:
        sd["model.lm_head.weight"] = sd["model.model.embed_tokens.weight"]
    K = meta["config"]["K"]
    st, sh = meta["sub"]
    col = {}
    loss, logits = R.omics_forward(sd, llm, dna, prot, tiny_batch(g, meta), {"dna_rna": K, "protein": K}, collect=col)
    sub = lambda t: t.detach().numpy()[:, ::st, ::sh]
    np.testing.assert_allclose(sub(col["enc"]["protein"]), g["fwd/enc_protein"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(sub(col["enc"]["dna_rna"]), g["fwd/enc_dna_rna"], rtol=0, atol=5e-5)
    np.testing.assert_allclose(sub(col["inputs_embeds"]), g["fwd/inputs_embeds"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(sub(col["layers"][0]), g["fwd/layer0"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(sub(col["final_hidden"]), g["fwd/final_hidden"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(sub(logits), g["fwd/logits"], rtol=0, atol=1e-4)
    assert abs(loss.item() - float(g["fwd/loss"])) < 1e-5
    loss.backward()
    names = [k[len("gnorm/"):] for k in g if k.startswith("gnorm/")]
    assert len(names) == (56 if tied else 57)
    for n in names:
        grad = sd[n].grad
        assert grad is not None, n
        ref = float(g["gnorm/" + n])
        assert abs(grad.double().norm().item() - ref) <= 1e-4 * max(ref, 1e-6) + 1e-7, (n, grad.norm().item(), ref)
        np.testing.assert_allclose(grad.flatten()[:256].numpy(), g["ghead/" + n], rtol=0,
                                   atol=2e-6 + 1e-4 * np.abs(g["ghead/" + n]).max())


def test_step_loop_trajectory_matches_reference_pieces(tiny_meta):
    """G5: four optimizer steps with GA = 2 on the reference's OmicsOne driven by torch AdamW, torch
    clip_grad_norm_ and HF's linear-warmup scheduler in the reference loop's order (gradients SUMMED over the window, lr 0 on
    the first step).  The oracle's own loop pieces (clip_coef, adamw_step, linear_warmup_lr) must walk the same trajectory."""
    import math
    from conftest import steps_fixture
    g, batch = steps_fixture()
    GA, STEPS, TOTAL = (int(x) for x in g["meta"])
    base = float(g["base_lr"])
    llm, dna, prot = R.cfgs_from_meta(tiny_meta["config"])
    sd = tiny_state_dict(tiny_meta)
    names = [k[len("pnorm/"):] for k in g if k.startswith("pnorm/")]
    for n in names:
        sd[n].requires_grad_(True)
    sd["model.lm_head.weight"] = sd["model.model.embed_tokens.weight"]
    K = tiny_meta["config"]["K"]
    state = {n: (torch.zeros_like(sd[n]), torch.zeros_like(sd[n])) for n in names}
    warm = math.ceil(0.1 * TOTAL)
    for s in range(STEPS):
        for k in range(GA):
            loss, _ = R.omics_forward(sd, llm, dna, prot, batch(s, k), {"dna_rna": K, "protein": K})
            loss.backward()                                             # autograd adds into .grad: the GA sum
            assert abs(loss.item() - g["loss"][s, k]) < 2e-5, (s, k, loss.item(), g["loss"][s, k])
        total, coef = R.clip_coef([sd[n].grad for n in names], 1.0)
        assert abs(total.item() - g["grad_norm"][s]) < 1e-4 * g["grad_norm"][s]
        lr = R.linear_warmup_lr(s, base, warm, TOTAL)
        assert abs(lr - g["lr"][s]) < 1e-12
        with torch.no_grad():
            for n in names:
                m, v = state[n]
                R.adamw_step(sd[n], sd[n].grad * coef, m, v, s + 1, lr, 0.0 if R.is_no_decay(n) else 1e-2)
                sd[n].grad = None
    for n in names:
        ref = float(g["pnorm/" + n])
        assert abs(sd[n].detach().double().norm().item() - ref) <= 1e-5 * ref + 1e-7, n
        np.testing.assert_allclose(sd[n].detach().flatten()[:256].numpy(), g["phead/" + n], rtol=0, atol=2e-6)


def test_config4_dna_encoder_full_depth_k1000_matches_reference():
    """BASELINE configs[3] at its defining encoder size (tests/golden/gen_golden_c4.py): the 24-layer NT-500M-shaped stack on a
    1000-token DNA row — absolute position ids 2..1001 of a 1002-row table (HF:models/esm/modeling_esm.py:1050-1063), a
    length that is no multiple of any tile — oracle vs the reference's fp32 output.  (The whole c4 fixture is pinned on the
    GPU side, tests/test_gpu_config4.py; the 8B-wide decoder at T = 4096 takes minutes on CPU.)"""
    import json
    import os
    from conftest import GOLD
    with open(os.path.join(GOLD, "c4_meta.json")) as f:
        meta = json.load(f)
    g = np.load(os.path.join(GOLD, "c4_fp32.npz"))
    _, dna, _ = R.cfgs_from_meta(meta["config"])
    from molly_amd.synth import synth_state_dict
    shapes = {k: tuple(v) for k, v in meta["state_dict_shapes"].items() if k.startswith("dna_rna_model.esm.")}
    sd = synth_state_dict(shapes, meta["config"]["seed_w"])
    ids = torch.from_numpy(g["in/omic_dna"])[None]
    assert ids.shape == (1, 1000) and int(R.esm_position_ids(ids, 1).max()) == 1001
    st, sh = meta["sub"]
    with torch.no_grad():
        out = R.esm_encoder(sd, "dna_rna_model.esm.", dna, ids)
    np.testing.assert_allclose(out.numpy()[:, ::st, ::sh], g["fwd/enc_dna_rna"], rtol=0, atol=2e-4)


def test_oracle_lora_branch_matches_hf_qwen3_with_peft_style_wrappers(tiny_meta, tiny_gold):
    """tests/golden/tiny_lora.npz (gen_golden_lora.py): the reference's OmicsOne over HF's Qwen3 with PEFT-style LoRA wrappers
    injected into the HF module tree on the seven targets (r 8, alpha 16 -> scaling 2, B non-zero), base frozen, projectors
    trainable.  The oracle's lora_linear branch + autograd must reproduce loss, logits and all 2*7*L + 4 gradients.
    (PEFT itself is not installed: the 6-line wrapper in the generator restates its published forward; everything around it
    — HF attention/MLP/loss, the injection path, autograd — is the real code.)"""
    import os
    from conftest import GOLD
    from molly_amd.synth import synth_tensor
    g = np.load(os.path.join(GOLD, "tiny_lora.npz"))
    r, alpha, seed = int(g["r"]), float(g["alpha"]), int(g["seed"])
    llm, dna, prot = R.cfgs_from_meta(tiny_meta["config"])
    sd = tiny_state_dict(tiny_meta)
    leaves = {}
    for k in g.files:
        if not k.startswith("g/"):
            continue
        n = k[2:]
        if ".lora_A." in n:
            leaves[n] = (synth_tensor(n, g[k].shape, seed) * (50.0 / r)).requires_grad_(True)
        elif ".lora_B." in n:
            leaves[n] = (synth_tensor(n, g[k].shape, seed) * 2.5).requires_grad_(True)
        else:
            leaves[n] = sd[n].clone().requires_grad_(True)
    sd.update(leaves)
    sd["lora.scaling"] = alpha / r
    K = tiny_meta["config"]["K"]
    loss, logits = R.omics_forward(sd, llm, dna, prot, tiny_batch(tiny_gold, tiny_meta), {"dna_rna": K, "protein": K})
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    np.testing.assert_allclose(logits.detach().numpy()[:, ::2], g["logits"], rtol=0, atol=5e-5)
    loss.backward()
    assert len(leaves) == 2 * 7 * llm.num_hidden_layers + 4
    for n, leaf in leaves.items():
        ref = g["g/" + n]
        np.testing.assert_allclose(leaf.grad.numpy(), ref, rtol=0, atol=2e-6 + 2e-4 * np.abs(ref).max(), err_msg=n)
