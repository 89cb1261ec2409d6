"""GPU: the Enc-Head baselines (reference baselines/model.py:33-215) on the encoder kernels against golden
vectors of the reference's own `BackboneWithClsHead` (fp32, tests/golden/gen_golden_baseline.py): logits and loss of
every backbone combination and both heads, every parameter gradient, the frozen-backbone mode, and a few optimizer steps.
The product computes in bf16 (fp32 accumulation): tolerances are bf16 resolution of the fp32 reference, stated per check."""
import pytest
import torch

from _baseline_common import PARTS, baseline_state_dict, case_inputs, load_gold

pytestmark = pytest.mark.gpu


def _build(meta, mtype, multi, nl, seed, freeze=False):
    from molly_amd.baselines import BackboneWithClsHead
    from molly_amd.config import EncConfig
    c = meta["config"]
    m = BackboneWithClsHead(mtype, nt_model=EncConfig.from_dict(c["dna_rna"]), esm_model=EncConfig.from_dict(c["protein"]),
                            num_labels=nl, multi_answer=bool(multi))
    sd, _ = baseline_state_dict(meta, mtype, nl, seed)
    missing, unexpected = m.load_state_dict(sd, strict=False, assign=True)
    assert not unexpected and not missing, (missing, unexpected)
    if freeze:
        m.freeze_backbone()
    return m.prepare("cuda"), sd


def test_state_dict_keys_are_the_reference_layout(tiny_meta):
    m, sd = _build(tiny_meta, "NT+ESM", False, 5, 1)
    keys = set(m.state_dict())
    assert keys == set(sd) and {"head.weight", "head.bias"} <= keys
    assert any(k.startswith("nt.esm.encoder.layer.0.") for k in keys) and any(k.startswith("esm.esm.") for k in keys)
    assert m.state_dict()["head.weight"].shape == (5, 256)


def test_forward_and_gradients_match_reference(tiny_meta):
    g, cases = load_gold()
    for name, mtype, multi, nl in cases:
        m, _ = _build(tiny_meta, mtype, multi, nl, int(g["meta/seed_w"]))
        xs, labels = case_inputs(g, name, mtype)
        masks = [(x != 1).long() for x in xs]
        args = (xs[0], xs[1] if len(xs) > 1 else None, masks[0], masks[1] if len(masks) > 1 else None)
        out = m(*args, labels=labels)
        ref_logits = torch.from_numpy(g[f"{name}/logits"])
        # bf16 encoders against the fp32 reference: logits are O(0.3), one bf16 step of the features is 2^-8 relative
        assert (out.logits.cpu() - ref_logits).abs().max().item() <= 2e-2, name
        assert abs(out.loss.item() - float(g[f"{name}/loss"])) <= 1e-2, name
        loss = m.forward_backward(*args, labels=labels)
        torch.cuda.synchronize()
        # the training forward rounds the FFN pre-activation to bf16 before GELU (what HF's Linear returns); the inference
        # forward applies GELU on the fp32 accumulator inside the GEMM epilogue
        assert abs(loss.item() - out.loss.item()) <= 3e-3
        assert abs(loss.item() - float(g[f"{name}/loss"])) <= 1e-2, name
        G = m._rt.G.views
        names = [k[len(name) + 7:] for k in g if k.startswith(name + "/gnorm/")]
        checked = 0
        for n in names:
            ref_norm = float(g[f"{name}/gnorm/{n}"])
            ref_head = torch.from_numpy(g[f"{name}/ghead/{n}"])
            got = G[n].float().cpu()
            if n.startswith("head."):
                got = got[:nl]
                assert torch.count_nonzero(G[n][nl:]) == 0                     # padded label rows: exactly zero gradient
            if ref_norm == 0.0:
                assert torch.count_nonzero(got) == 0, (name, n)
                continue
            if n.endswith("key.bias"):
                continue                                                       # ~0 by construction (see test_gpu_train_bio.py)
            scale = ref_head.abs().max().item()
            if scale > 0:
                err = (got.flatten()[:128] - ref_head).abs().max().item()
                # the fixture keeps the first 128 entries only: judge them on that slice's largest entry plus the tensor's
                # rms (bf16 rounding noise of the gradient does not shrink with the slice's own magnitude)
                rms = ref_norm / got.numel() ** 0.5
                assert err <= 0.1 * scale + 0.5 * rms + 1e-6, (name, n, err, scale, rms)
            assert abs(got.norm().item() - ref_norm) <= 6e-2 * ref_norm + 1e-7, (name, n, got.norm().item(), ref_norm)
            checked += 1
        assert checked >= 30, (name, checked)


def test_frozen_backbone_trains_only_the_head(tiny_meta):
    g, _ = load_gold()
    m, _ = _build(tiny_meta, "ESM", False, 3, int(g["meta/seed_w"]), freeze=True)
    assert set(m._rt.G.views) == {"head.weight", "head.bias"}
    xs, labels = case_inputs(g, "esm", "ESM")
    loss = m.forward_backward(xs[0], None, (xs[0] != 1).long(), None, labels=labels)
    assert abs(loss.item() - float(g["esm/loss"])) <= 1e-2
    got = m._rt.G.views["head.weight"][:3].float().cpu()
    ref = torch.from_numpy(g["esm/ghead/head.weight"])
    assert (got.flatten()[:128] - ref).abs().max().item() <= 0.1 * ref.abs().max().item()


@pytest.mark.parametrize("case", ["nt_esm", "esm_multi"])
def test_a_few_steps_reduce_the_loss(tiny_meta, case):
    from molly_amd.trainer import Zero2Optimizer
    g, cases = load_gold()
    name, mtype, multi, nl = next(c for c in cases if c[0] == case)
    m, _ = _build(tiny_meta, mtype, multi, nl, int(g["meta/seed_w"]))
    opt = Zero2Optimizer(m._rt.P.flat, m._rt.G.flat, m.n_decay, lr=2e-3, weight_decay=1e-2, max_grad_norm=1.0)
    xs, labels = case_inputs(g, name, mtype)
    masks = [(x != 1).long() for x in xs]
    args = (xs[0], xs[1] if len(xs) > 1 else None, masks[0], masks[1] if len(masks) > 1 else None)
    w0 = m._rt.P.views[PARTS[mtype][0][0] + ".esm.encoder.layer.0.attention.self.query.weight"].clone()
    losses = []
    for _ in range(8):
        losses.append(m.forward_backward(*args, labels=labels).item())
        opt.step(lr=2e-3)
    assert losses[-1] < 0.7 * losses[0], losses
    assert not torch.equal(w0, m._rt.P.views[PARTS[mtype][0][0] + ".esm.encoder.layer.0.attention.self.query.weight"])
    assert torch.count_nonzero(m._rt.P.views["head.weight"][nl:]) == 0           # padded rows never move
