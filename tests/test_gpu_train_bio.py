"""GPU: encoder training (`--train-bio`, reference src/utils/tools.py:326-330): the LayerNorm / GELU backward kernels
against torch autograd, and every encoder parameter gradient of the full step (projector -> ESM stack -> embeddings,
rotary ESM-2-style protein encoder and absolute-position NT-style DNA/RNA encoder) against the oracle's autograd."""
import pytest
import torch

from conftest import tiny_state_dict

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def test_layernorm_bwd_and_gelu_match_autograd():
    from molly_amd import ops
    g = torch.Generator(device="cuda").manual_seed(0)
    for rows, H in ((515, 1280), (64, 128), (1030, 2560)):
        x = torch.randn(rows, H, device="cuda", generator=g).bfloat16()
        w = (1 + 0.1 * torch.randn(H, device="cuda", generator=g)).bfloat16()
        b = (0.1 * torch.randn(H, device="cuda", generator=g)).bfloat16()
        gy = torch.randn(rows, H, device="cuda", generator=g).bfloat16()
        res = torch.randn(rows, H, device="cuda", generator=g).bfloat16()
        xr, wr, br = (t.float().requires_grad_(True) for t in (x, w, b))
        y = torch.nn.functional.layer_norm(xr, (H,), wr, br, 1e-5)
        y.backward(gy.float())
        dw = torch.empty(H, dtype=BF, device="cuda"); db = torch.empty(H, dtype=BF, device="cuda")
        dx = ops.layernorm_bwd(x, w, gy, dw, db, 1e-5, dres=res)
        assert (dx.float() - (xr.grad + res.float())).abs().max().item() <= 3e-2 * xr.grad.abs().max().item() + 3e-2
        for got, ref in ((dw, wr.grad), (db, br.grad)):
            assert (got.float() - ref).abs().max().item() <= 1e-2 * ref.abs().max().item() + 1e-2
        # accumulate form adds
        dw2, db2 = dw.clone(), db.clone()
        ops.layernorm_bwd(x, w, gy, dw2, db2, 1e-5, dw_accumulate=True)
        assert (dw2.float() - 2 * wr.grad).abs().max().item() <= 2e-2 * wr.grad.abs().max().item() + 2e-2
    z = (2 * torch.randn(1 << 16, device="cuda", generator=g)).bfloat16()
    dy = torch.randn(1 << 16, device="cuda", generator=g).bfloat16()
    zr = z.float().requires_grad_(True)
    yr = torch.nn.functional.gelu(zr)
    yr.backward(dy.float())
    assert (ops.gelu_fwd(z).float() - yr).abs().max().item() <= 2e-2
    assert (ops.gelu_bwd(z, dy).float() - zr.grad).abs().max().item() <= 3e-2


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


def _batch(meta, seed=7):
    from molly_amd.synth import synth_batch
    sp = {k: tuple(v) for k, v in meta["config"]["special_ids"].items()}
    return synth_batch(3, 384, [("protein", 64), ("rna", 64)], seed=seed, text_vocab=1000, special_ids=sp, pad_id=1000, ragged=True)


def _args(b):
    return [b[k] for k in ("input_ids", "attention_mask", "omic_ids", "omic_info_list", "labels")]


@pytest.mark.parametrize("mode", ["full", "bio_only"])
def test_encoder_gradients_vs_oracle_autograd(tiny_meta, mode):
    from oracle import molly_ref as R
    prep = dict(train_bio=True) if mode == "full" else dict(train_llm=False, train_mlp=False, train_bio=True)
    m = _build(tiny_meta, **prep)
    b = _batch(tiny_meta)
    loss = m.forward_backward(*_args(b))
    torch.cuda.synchronize()
    sd = tiny_state_dict(tiny_meta)
    names = [n for n in m._rt.G.views if n.startswith(("dna_rna_model.", "protein_model."))]
    assert names and all(n in sd for n in names)
    if mode == "bio_only":
        assert set(m._rt.G.views) == set(names)                        # nothing else is in the optimizer's group
    leaves = {n: sd[n].clone().requires_grad_(True) for n in names}
    sd.update(leaves)
    llm, dna, prot = R.cfgs_from_meta(tiny_meta["config"])
    ref_loss, _ = R.omics_forward(sd, llm, dna, prot, b, {"dna_rna": 64, "protein": 64})
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) <= 3e-3
    G = m._rt.G.views
    worst, checked = 0.0, 0
    for n, leaf in leaves.items():
        ref = leaf.grad
        got = G[n].float().cpu()
        if ref is None or ref.abs().max().item() == 0.0:               # e.g. unused embedding rows only -> whole tensor zero
            assert torch.count_nonzero(got) == 0, n
            continue
        scale = ref.abs().max().item()
        if n.endswith("key.bias"):
            # without rotary this gradient is exactly zero in exact arithmetic (the rows of dS sum to zero over the keys);
            # with rotary it is a small residue of the per-position rotations.  Judge it on the query-bias scale: what the
            # HIP path adds is the bf16 rounding of dK summed over the rows.
            scale = max(scale, leaves[n.replace("key.bias", "query.bias")].grad.abs().max().item())
            assert (got - ref).abs().max().item() <= 0.25 * scale, (n, (got - ref).abs().max().item(), scale)
            continue
        rel = (got - ref).abs().max().item() / scale
        worst = max(worst, rel)
        checked += 1
        assert rel < 8e-2, (n, rel)
        assert abs(got.norm().item() - ref.norm().item()) <= 4e-2 * ref.norm().item() + 1e-6, n
    print(f"{checked} encoder tensors, worst relative error {worst:.4f}")
    assert checked >= 60


def test_train_bio_steps_reduce_loss_and_move_encoders(tiny_meta):
    from molly_amd.trainer import Zero2Optimizer
    m = _build(tiny_meta, train_bio=True)
    rt = m._rt
    opt = Zero2Optimizer(rt.P.flat, rt.G.flat, m.n_decay, lr=1e-3, weight_decay=1e-2, max_grad_norm=1.0)
    m.attach_optimizer(opt)
    w0 = rt.W["protein_model.esm.encoder.layer.0.attention.self.query.weight"].clone()
    e0 = rt.W["dna_rna_model.esm.embeddings.word_embeddings.weight"].clone()
    b = _batch(tiny_meta, seed=3)
    losses = []
    for _ in range(8):
        losses.append(m.forward_backward(*_args(b)).clone())
        opt.step(lr=1e-3)
    losses = torch.stack(losses).cpu()
    assert losses[-1] < losses[0] - 0.3, losses
    assert not torch.equal(w0, rt.W["protein_model.esm.encoder.layer.0.attention.self.query.weight"])
    assert not torch.equal(e0, rt.W["dna_rna_model.esm.embeddings.word_embeddings.weight"])
    # decay split: every encoder bias / LayerNorm tensor sits in the no-decay tail
    for n, off in rt.P.offsets.items():
        if n.startswith(("dna_rna_model.", "protein_model.")):
            assert (off >= m.n_decay) == (n.endswith("bias") or "norm" in n.lower()), n


def test_encoder_gradients_vs_reference_golden(tiny_meta, tiny_gold):
    """The same gradients from the REFERENCE's autograd (tests/golden/tiny_trainbio.npz) on the golden batch."""
    import os
    import numpy as np
    from conftest import GOLD, tiny_batch
    g = dict(np.load(os.path.join(GOLD, "tiny_trainbio.npz"), allow_pickle=False))
    m = _build(tiny_meta, train_bio=True)
    b = tiny_batch(tiny_gold, tiny_meta)
    loss = m.forward_backward(*_args(b))
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["loss"])) <= 3e-3
    G = m._rt.G.views
    names = [k[len("gnorm/"):] for k in g if k.startswith("gnorm/")]
    worst = 0.0
    for n in names:
        got = G[n].float().cpu()
        head = torch.from_numpy(g["ghead/" + n])
        ref_norm = float(g["gnorm/" + n])
        scale = head.abs().max().item()
        if n.endswith("key.bias"):               # ~zero by construction (see the oracle test above): query-bias scale
            scale = max(scale, float(np.abs(g["ghead/" + n.replace("key.bias", "query.bias")]).max()))
            assert (got.flatten()[:256] - head).abs().max().item() <= 0.25 * scale, n
            continue
        if scale == 0.0:
            continue
        # the stored head is 256 entries: judge it on the tensor's own scale (its rms from the stored norm) as well
        scale = max(scale, 4.0 * ref_norm / got.numel() ** 0.5)
        rel = (got.flatten()[:256] - head).abs().max().item() / scale
        worst = max(worst, rel)
        assert rel < 8e-2, (n, rel)
        assert abs(got.double().norm().item() - ref_norm) <= 4e-2 * ref_norm + 1e-6, n
    print("worst relative error vs the reference's gradients", worst)


def test_lora_plus_train_bio_share_one_trainable_group(tiny_meta):
    """`--use-lora --train-bio`: adapters, projectors and both encoders in ONE flat ZeRO group, base LLM frozen."""
    from molly_amd.lora import LoraConfig
    from molly_amd.trainer import Zero2Optimizer
    m = _build(tiny_meta, train_llm=False, lora=LoraConfig(r=8, lora_alpha=16, lora_dropout=0.05, seed=2), train_bio=True)
    rt = m._rt
    names = list(rt.G.views)
    assert any(".lora_A." in n for n in names) and any(n.startswith("protein_model.") for n in names)
    assert "protein_projector.weight" in names and not any(n.endswith("mlp.down_proj.weight") for n in names)
    base0 = rt.base.flat.clone()
    opt = Zero2Optimizer(rt.P.flat, rt.G.flat, m.n_decay, lr=1e-3, max_grad_norm=1.0)
    m.attach_optimizer(opt)
    b = _batch(tiny_meta, seed=4)
    losses = []
    for _ in range(6):
        losses.append(m.forward_backward(*_args(b)).clone())
        assert torch.isfinite(rt.G.flat.float()).all()
        opt.step(lr=1e-3)
    assert torch.stack(losses)[-1] < losses[0] - 0.1
    assert torch.equal(base0, rt.base.flat)
    g = rt.G.views
    assert torch.count_nonzero(g["protein_model.esm.encoder.layer.0.attention.self.query.weight"]) > 0
    assert torch.count_nonzero(g["model.model.layers.0.self_attn.q_proj.lora_B.weight"]) > 0
