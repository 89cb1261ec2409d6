"""GPU: the `--use-lora` and `--train-mlp`-only training modes (reference: src/utils/tools.py:313-338, 345-396) —
adapter forward/backward on the HIP path against the torch-fp32 oracle's autograd, the dropout kernel, the frozen-base
backward, the PEFT-layout adapter writer and the merge-at-load inference path.  PEFT is not importable offline: the LoRA
branch is pinned against the oracle's restatement only (oracle/molly_ref.py::lora_linear)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import tiny_state_dict

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _build(meta, **prep):
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
    m.load_state_dict(tiny_state_dict(meta), strict=False)
    m.prepare("cuda", **prep)
    return m


def _batch(meta, seed=7, B=3, T=384):
    from molly_amd.synth import synth_batch
    sp = {k: tuple(v) for k, v in meta["config"]["special_ids"].items()}
    return synth_batch(B, T, [("protein", 64), ("rna", 64)], seed=seed, text_vocab=1000, special_ids=sp, pad_id=1000, ragged=True)


def _args(b):
    return [b[k] for k in ("input_ids", "attention_mask", "omic_ids", "omic_info_list", "labels")]


def _randomize_B(m, seed=3, std=0.05):
    """B = 0 at init makes the branch (and dA) vanish: give B values so every gradient path is exercised."""
    lo = m._rt.llm.lora
    g = torch.Generator(device="cuda").manual_seed(seed)
    for lb in lo.B:
        for t in lb.values():
            t[:, :lo.r].normal_(0.0, std, generator=g)


def _oracle_sd_with_lora(meta, m, masks=None):
    """fp32 oracle state dict + the adapter the GPU model holds (its bf16 values, pad cut off) as autograd leaves."""
    from molly_amd.lora import TARGETS, lora_name
    sd = tiny_state_dict(meta)
    lo = m._rt.llm.lora
    leaves = {}
    for i in range(len(lo.A)):
        for mod in TARGETS:
            for which, t in (("A", lo.A[i][mod][:lo.r]), ("B", lo.B[i][mod][:, :lo.r])):
                n = lora_name(i, mod, which)
                leaves[n] = t.detach().float().cpu().clone().requires_grad_(True)
    for n in ("dna_rna_projector.weight", "dna_rna_projector.bias", "protein_projector.weight", "protein_projector.bias"):
        leaves[n] = sd[n].clone().requires_grad_(True)
    sd.update(leaves)
    sd["lora.scaling"] = lo.scale
    if masks is not None:
        sd["lora.masks"] = masks
    return sd, leaves


def _check_grads(m, leaves, tol=6e-2):
    lo = m._rt.llm.lora
    G = m._rt.G.views
    worst = 0.0
    for n, leaf in leaves.items():
        ref = leaf.grad
        got = G[n].float().cpu()
        if ".lora_A." in n:
            assert torch.count_nonzero(got[lo.r:]) == 0, n            # pad rows: exactly zero gradient
            got = got[:lo.r]
        elif ".lora_B." in n:
            assert torch.count_nonzero(got[:, lo.r:]) == 0, n
            got = got[:, :lo.r]
        scale = ref.abs().max().item()
        assert scale > 0, n
        rel = (got - ref).abs().max().item() / scale
        worst = max(worst, rel)
        assert rel < tol, (n, rel)
        assert abs(got.norm().item() - ref.norm().item()) <= 3e-2 * ref.norm().item(), n
    return worst


def test_dropout_kernel():
    from molly_amd import ops
    n, p = 1 << 22, 0.05
    x = torch.randn(n, device="cuda").bfloat16()
    a = ops.dropout(x, p, seed=123)
    b = ops.dropout(x, p, seed=123)
    c = ops.dropout(x, p, seed=124)
    assert torch.equal(a, b) and not torch.equal(a, c)
    keep = a != 0
    frac = keep.float().mean().item() / (x != 0).float().mean().item()
    assert abs(frac - (1 - p)) < 1e-3, frac                                # sigma ~ 1e-4 at this n
    want = (x.float() / (1 - p)).bfloat16()
    assert torch.equal(a[keep], want[keep])
    # the two masks of different seeds are independent: joint keep rate = product
    both = ((a != 0) & (c != 0)).float().mean().item()
    assert abs(both - (1 - p) ** 2) < 2e-3
    # accumulate form: out += dropped(x), same mask
    base = torch.randn(n, device="cuda").bfloat16()
    acc = base.clone()
    ops.dropout(x, p, seed=123, out=acc, accumulate=True)
    assert torch.equal(acc, (base.float() + a.float()).bfloat16())
    y = x.clone()
    ops.scale_(y, 0.25)
    assert torch.equal(y, (x.float() * 0.25).bfloat16())


@pytest.mark.parametrize("M,K,scale", [(300, 256, 2.0), (4096, 2048, 1.0), (1000, 6144, 0.5)])
def test_fused_lora_down_projection_equals_dropout_then_gemm(M, K, scale):
    """molly_lora_down_drop_bf16 (round 5): t = scale * dropout(x) A^T and xd = dropout(x) in one launch against the three launches it
    replaces — the mask is the same function of (seed, element index), so xd is bit-identical; t differs by the fp32 summation
    order and by ONE rounding instead of two (GEMM output, then scale)."""
    from molly_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + K)
    x = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    A = (torch.randn(64, K, device="cuda", generator=g) / 8).bfloat16()
    p, seed = 0.05, (7 << 32) | 12345
    xd_ref = ops.dropout(x, p, seed)
    t_ref = ops.gemm_nt(xd_ref, A)
    if scale != 1.0:
        ops.scale_(t_ref, scale)
    xd = torch.full_like(x, 3.0)
    t = ops.lora_down_drop(x, A, p, seed, scale, xd=xd)
    torch.cuda.synchronize()
    assert torch.equal(xd, xd_ref)
    exact = (xd_ref.float() @ A.float().t()) * scale
    tol = 2 ** -7 * exact.abs() + 2 ** -7 * exact.abs().max() * 0.05
    assert bool(((t.float() - exact).abs() <= tol).all()), (t.float() - exact).abs().max().item()
    assert bool(((t.float() - t_ref.float()).abs() <= 2 * tol).all())
    t2 = ops.lora_down_drop(x, A, p, seed, scale)                # without the side output: same t
    assert torch.equal(t, t2)
    tT = torch.empty(64, M, dtype=torch.bfloat16, device="cuda")  # ... and t^T beside it (the adapter weight gradients' operand)
    t4 = ops.lora_down_drop(x, A, p, seed, scale, out_t=tT)
    assert torch.equal(t4, t) and torch.equal(tT, t.T.contiguous())
    # p = 0: the plain skinny product, also on a column slice of a wider buffer (the backward's dt = s * dy B reads slices of d_qkv / d_gu)
    wide = torch.randn(M, K + 128, device="cuda", generator=g).bfloat16()
    xs = wide[:, 64:64 + K]
    t3 = ops.lora_down_drop(xs, A, 0.0, 0, scale)
    exact3 = (xs.float() @ A.float().t()) * scale
    assert bool(((t3.float() - exact3).abs() <= 2 ** -7 * exact3.abs() + 2 ** -7 * exact3.abs().max() * 0.05).all())


@pytest.mark.parametrize("M,K", [(300, 256), (2048 + 17, 2048), (512, 6144)])
def test_fused_lora_input_gradient_equals_gemm_then_masked_accumulate(M, K):
    """molly_lora_up_drop_acc_bf16 (round 5): dx += mask * bf16(dt A) in one launch against GEMM + dropout-accumulate: the same mask,
    the product rounded to bf16 before the mask in both, so the results agree except where the fp32 summation order moved a
    rounding (a bf16 ulp of the product on a few elements)."""
    from molly_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M * 3 + K)
    dt = (torch.randn(M, 64, device="cuda", generator=g) / 4).bfloat16()
    A = (torch.randn(64, K, device="cuda", generator=g) / 4).bfloat16()
    base = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    p, seed = 0.05, (3 << 32) | 777
    tmp = ops.gemm(dt, A, b_kmajor=True)
    ref = base.clone()
    ops.dropout(tmp, p, seed, out=ref, accumulate=True)
    got = base.clone()
    ops.lora_up_drop_acc(dt, A, got, p, seed)
    torch.cuda.synchronize()
    keep = ops.dropout(torch.ones_like(tmp), p, seed) != 0
    assert torch.equal(got[~keep], base[~keep])                  # dropped elements: dx untouched
    d = (got.float() - ref.float()).abs()
    assert (d > 0).float().mean().item() < 2e-2, (d > 0).float().mean().item()
    assert bool((d <= 2 ** -6 * (tmp.float().abs() + base.float().abs()) + 1e-6).all()), d.max().item()
    exact = base.float() + keep.float() * (dt.float() @ A.float()) / (1 - p)
    assert bool(((got.float() - exact).abs() <= 2 ** -6 * exact.abs() + 2 ** -6 * (dt.float() @ A.float()).abs() + 1e-3).all())


@pytest.mark.parametrize("n", [2, 3])
def test_merged_lora_input_gradients_equal_the_launches_one_after_the_other(n):
    """molly_lora_up_drop_acc_multi_bf16: the adapters that share an input (q | k | v: 3, gate | up: 2) add their mask * (dt A) terms to dx in ONE pass —
    bit for bit the result of their launches one after the other (same masks, same order of the bf16 roundings); dt's are column slices of one buffer,
    as the model passes them."""
    from molly_amd import ops
    M, K = 1024 + 40, 2048
    g = torch.Generator(device="cuda").manual_seed(50 + n)
    dts = (torch.randn(M, 64 * n, device="cuda", generator=g) / 4).bfloat16()
    As = [(torch.randn(64, K, device="cuda", generator=g) / 4).bfloat16() for _ in range(n)]
    base = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    p, seeds = 0.05, [(u << 40) | 12345 + u for u in range(n)]
    ref = base.clone()
    for u in range(n):
        ops.lora_up_drop_acc(dts[:, 64 * u:64 * (u + 1)], As[u], ref, p, seeds[u])
    got = base.clone()
    ops.lora_up_drop_acc_multi([dts[:, 64 * u:64 * (u + 1)] for u in range(n)], As, got, p, seeds)
    torch.cuda.synchronize()
    assert torch.equal(got, ref), (got.float() - ref.float()).abs().max().item()
    assert not torch.equal(got, base)


def test_lora_forward_backward_vs_oracle(tiny_meta):
    """r = 8 (padded to 64), alpha = 16 (scaling 2), no dropout: loss and every adapter / projector gradient against
    the oracle's autograd; GA semantics (accumulate=True adds)."""
    from molly_amd.lora import LoraConfig
    from oracle import molly_ref as R
    m = _build(tiny_meta, train_llm=False, lora=LoraConfig(r=8, lora_alpha=16, lora_dropout=0.0))
    _randomize_B(m)
    b = _batch(tiny_meta)
    loss = m.forward_backward(*_args(b))
    torch.cuda.synchronize()
    sd, leaves = _oracle_sd_with_lora(tiny_meta, m)
    llm, dna, prot = R.cfgs_from_meta(tiny_meta["config"])
    ref_loss, _ = R.omics_forward(sd, llm, dna, prot, b, {"dna_rna": 64, "protein": 64})
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) <= 3e-3, (loss.item(), ref_loss.item())
    print("worst relative adapter-grad error", _check_grads(m, leaves))
    # the adapter changes the function: same batch through a model without it gives another loss
    g1 = m._rt.G.flat.clone()
    m.forward_backward(*_args(b), accumulate=True)
    torch.cuda.synchronize()
    assert torch.allclose(m._rt.G.flat.float(), 2 * g1.float(), rtol=2e-2, atol=1e-6)


def test_lora_dropout_backward_uses_the_forward_masks(tiny_meta):
    """p = 0.25: rebuild the masks the engine drew (same seeds through the dropout kernel) and hand them to the oracle."""
    from molly_amd import ops
    from molly_amd.lora import TARGETS, LoraConfig, target_dims
    from oracle import molly_ref as R
    p = 0.25
    m = _build(tiny_meta, train_llm=False, lora=LoraConfig(r=64, lora_alpha=64, lora_dropout=p, seed=5))
    _randomize_B(m)
    b = _batch(tiny_meta, seed=11)
    loss = m.forward_backward(*_args(b))
    torch.cuda.synchronize()
    lo = m._rt.llm.lora
    B_, T_ = b["input_ids"].shape
    dims = target_dims(m.text_config)
    masks = {}
    for i in range(len(lo.A)):
        for mod in TARGETS:
            ones = torch.ones(B_ * T_, dims[mod][0], dtype=BF, device="cuda")
            keep = ops.dropout(ones, p, lo.mask_seed(i, mod)) != 0
            masks[f"model.model.layers.{i}.{'self_attn' if mod[0] in 'qkvo' else 'mlp'}.{mod}"] = \
                (keep.float() / (1 - p)).cpu().view(B_, T_, -1)
    sd, leaves = _oracle_sd_with_lora(tiny_meta, m, masks)
    llm, dna, prot = R.cfgs_from_meta(tiny_meta["config"])
    ref_loss, _ = R.omics_forward(sd, llm, dna, prot, b, {"dna_rna": 64, "protein": 64})
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) <= 3e-3, (loss.item(), ref_loss.item())
    print("worst relative adapter-grad error (dropout)", _check_grads(m, leaves))
    # a second forward draws fresh masks (new stream per training forward) -> a different loss; eval draws none
    l1 = loss.item()                       # `loss` is a view of the engine's device scalar: read it before the next step
    l2 = m.forward_backward(*_args(b)).item()
    assert l2 != l1
    with torch.no_grad():
        e1 = m(*_args(b)).loss.item()
        e2 = m(*_args(b)).loss.item()
    assert e1 == e2


def test_projector_only_mode_matches_full_mode_projector_grads(tiny_meta):
    """`--train-mlp` without `--train-llm`: the frozen-base backward must hand the projectors the gradients the full
    backward does (same kernels, no base wgrad) — and nothing else is in the optimizer's group."""
    b = _batch(tiny_meta, seed=13)
    full = _build(tiny_meta)
    lf = full.forward_backward(*_args(b))
    proj = _build(tiny_meta, train_llm=False, train_mlp=True)
    lp = proj.forward_backward(*_args(b))
    torch.cuda.synchronize()
    assert lf.item() == lp.item()
    names = ["dna_rna_projector.weight", "protein_projector.weight", "dna_rna_projector.bias", "protein_projector.bias"]
    assert list(proj._rt.G.views.keys()) == names
    for n in names:
        assert torch.equal(full._rt.G.views[n], proj._rt.G.views[n]), n
    assert proj.n_decay == proj._rt.P.offsets["dna_rna_projector.bias"]
    assert [n for n, p_ in proj.named_parameters() if p_.requires_grad] == \
        ["dna_rna_projector.weight", "dna_rna_projector.bias", "protein_projector.weight", "protein_projector.bias"]


def test_train_llm_without_train_mlp_keeps_the_projectors_frozen(tiny_meta):
    """`--train-llm` alone (reference: src/utils/tools.py:313-338 accepts any flag combination; the set it leaves behind is the
    (True, False, False) case of tests/golden/trainable_sets.json): the flat ZeRO group is the LLM only, every LLM gradient
    equals the full mode's bit for bit, and an optimizer step leaves the projectors untouched."""
    import json
    from molly_amd.trainer import Zero2Optimizer
    with open(os.path.join(os.path.dirname(__file__), "golden", "trainable_sets.json")) as f:
        case = next(c for c in json.load(f)["cases"] if (c["train_llm"], c["train_mlp"], c["train_bio"]) == (True, False, False))
    b = _batch(tiny_meta, seed=21)
    full = _build(tiny_meta)
    lf = full.forward_backward(*_args(b))
    llm = _build(tiny_meta, train_llm=True, train_mlp=False)
    ll = llm.forward_backward(*_args(b))
    torch.cuda.synchronize()
    ll_v = ll.item()                      # (the returned loss is a view of the engine's device scalar: the next step overwrites it)
    assert lf.item() == ll_v
    assert not any("projector" in n for n in llm._rt.G.views)
    assert set(llm._rt.G.views) == {n for n in full._rt.G.views if "projector" not in n}
    for n, g in llm._rt.G.views.items():
        assert torch.equal(g, full._rt.G.views[n]), n
    keys = set(llm.state_dict().keys())
    assert sorted(n for n, p_ in llm.named_parameters() if p_.requires_grad) == sorted(n for n in case["trainable"] if n in keys)
    before = {n: llm._rt.W[n].clone() for n in llm._rt.W if "projector" in n}
    emb0 = llm._rt.W["model.model.embed_tokens.weight"].clone()
    opt = Zero2Optimizer(llm._rt.P.flat, llm._rt.G.flat, llm.n_decay, lr=1e-2)
    llm.attach_optimizer(opt)
    opt.step(lr=1e-2)
    opt.wait_all_params()
    torch.cuda.synchronize()
    assert len(before) == 4 and all(torch.equal(v, llm._rt.W[n]) for n, v in before.items())
    assert not torch.equal(emb0, llm._rt.W["model.model.embed_tokens.weight"])
    # a second micro-step still runs (the injected rows read the frozen projectors) and the loss moved
    l2 = llm.forward_backward(*_args(b))
    assert torch.isfinite(l2).item() and l2.item() != ll_v


def test_lora_training_saves_a_peft_adapter_that_merges_back(tiny_meta, tmp_path):
    """A few optimizer steps on the adapter group, then: the loss went down, the base did not move, the pad stayed zero,
    `save_adapter` wrote the PEFT layout, and a fresh model with the adapter MERGED reproduces the live-adapter logits."""
    from molly_amd.lora import LoraConfig, merge_lora_adapter, save_adapter
    from molly_amd.trainer import Zero2Optimizer
    m = _build(tiny_meta, train_llm=False, lora=LoraConfig(r=16, lora_alpha=64, lora_dropout=0.05, seed=1))
    base0 = m._rt.base.flat.clone()
    opt = Zero2Optimizer(m._rt.P.flat, m._rt.G.flat, m.n_decay, lr=2e-3, weight_decay=1e-2, max_grad_norm=1.0)
    m.attach_optimizer(opt)
    b = _batch(tiny_meta, seed=17)
    losses = []
    for _ in range(12):
        losses.append(m.forward_backward(*_args(b)).clone())
        opt.step(lr=2e-3)
    losses = torch.stack(losses).cpu()
    assert losses[-1] < losses[0] - 0.05, losses
    assert torch.equal(m._rt.base.flat, base0)
    lo = m._rt.llm.lora
    for la, lb in zip(lo.A, lo.B):
        for mod in la:
            assert torch.count_nonzero(la[mod][lo.r:]) == 0 and torch.count_nonzero(lb[mod][:, lo.r:]) == 0
            assert torch.count_nonzero(lb[mod][:, :lo.r]) > 0                     # B left zero
    d = str(tmp_path / "adapter")
    save_adapter(m, d)
    cfg = json.load(open(os.path.join(d, "adapter_config.json")))
    assert cfg["r"] == 16 and cfg["lora_alpha"] == 64 and cfg["peft_type"] == "LORA"
    from molly_amd.lora import load_adapter_tensors
    tens = load_adapter_tensors(d)
    k = "base_model.model.model.layers.1.mlp.down_proj.lora_B.weight"
    assert tuple(tens[k].shape) == (256, 16) and tuple(tens[k.replace("lora_B", "lora_A")].shape) == (16, 512)
    assert len(tens) == 2 * 7 * 2
    with torch.no_grad():
        live = m(*_args(b))
    fresh = _build(tiny_meta, train_llm=False, train_mlp=False)
    assert merge_lora_adapter(fresh, d) == 14
    with torch.no_grad():
        merged = fresh(*_args(b))
    valid = b["attention_mask"].bool()
    err = (live.logits.float() - merged.logits.float()).cpu()[valid].abs().max().item()
    assert err <= 3e-2 * live.logits.float().abs().max().item(), err
    assert abs(live.loss.item() - merged.loss.item()) <= 5e-3


def test_cli_lora_train_then_batch_inference_jsonl(tmp_path):
    """`--use-lora` through the launchers with the reference's flag names: train.py writes the PEFT adapter + projector
    files (src/trainer/omics_trainer.py:89-103), inference.py reads them back (merged and live) and writes the reference's
    JSONL records (src/inference_lora.py:316-323)."""
    from molly_amd import inference, train
    rows = [dict(task="Solubility-Solubility", input=f"Is <protein>{'MKTAYIAKQR' * (1 + i % 3)}</protein> soluble?", think="",
                 output="Yes." if i % 2 else "No.", label=str(i % 2), kind="protein", task_num=i) for i in range(8)]
    rows += [dict(task="tf-h", input=f"Does <dna>{'ACGTTGCA' * (2 + i % 4)}</dna> bind?", think="", output="It does.",
                  label="1", kind="dna", task_num=i) for i in range(8)]
    data = tmp_path / "mini.jsonl"
    data.write_text("\n".join(json.dumps(r) for r in rows))
    out = tmp_path / "ckpt"
    common = ["--text-model-path", "tiny", "--dna-rna-model-path", "tiny", "--protein-model-path", "tiny", "--no-load-pretrained",
              "--dna-rna-k-tokens", "64", "--protein-k-tokens", "64"]
    train.main(common + ["--output_dir", str(out), "--train-dataset-path", str(data), "--max-len", "256", "--use-lora",
                         "--lora_r", "16", "--per_device_train_batch_size", "4", "--train-iters", "3", "--learning_rate", "1e-3",
                         "--logging_steps", "1", "--bf16"])
    for f in ("adapter_config.json", "dna_rna_projector.bin", "protein_projector.bin"):
        assert (out / f).exists(), f
    assert (out / "adapter_model.safetensors").exists() or (out / "adapter_model.bin").exists()
    assert not (out / "pytorch_model.bin").exists()
    recs = {}
    for tag, extra in (("merged", []), ("live", ["--lora-live"])):
        jf = tmp_path / f"pred_{tag}.jsonl"
        inference.main(common + ["--trained-model-path", str(out), "--use-lora", *extra, "--dataset-path", str(data),
                                 "--max-length", "256", "--batch-size", "4", "--max-samples", "6", "--greedy",
                                 "--max-new-tokens", "5", "--json-file", str(jf)])
        recs[tag] = [json.loads(l) for l in jf.read_text().splitlines()]
        assert len(recs[tag]) == 6
        for r, src in zip(recs[tag], rows):
            assert set(r) == {"decoded_output", "input", "gt_output", "gt_label", "task", "kind"}
            assert r["task"] == src["task"] and r["gt_label"] == src["label"]
            # reference quirk kept: the Test-mode sample dict carries no "kind" (src/dataset/omics_dataset.py:393-402), so the
            # collate's sample.get("kind") (:528) — and the JSONL field — is always None
            assert r["kind"] is None
            assert isinstance(r["decoded_output"], str)
