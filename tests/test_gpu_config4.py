"""GPU: BASELINE configs[3] (Molly-8B, seq_len 4k text + 1024-residue protein + 6 kbp DNA) and configs[2] (Molly-4B, all three
modalities) at their DEFINING sizes.

(a) `c4_fp32.npz` (tests/golden/gen_golden_c4.py): the REFERENCE's OmicsOne in fp32 at Qwen3-8B decoder widths (two layers),
    T = 4096, the FULL 33-layer ESM2-650M-shaped protein encoder at K = 1024 and the FULL 24-layer NT-500M-shaped DNA encoder
    at K = 1000 (not a multiple of 64; learned absolute positions, position ids 2..1001 of a 1002-row table), one sample
    carrying both spans -> HIP path: encoder outputs, logits, loss, and all 948 gradients (encoders trainable).
    Tolerance: 57 bf16 layers deep the reference's OWN bf16 CPU path sits 0.38 (7.6 % of max|logit|) from its fp32 path; the
    bound is that yardstick, stated below, not the 3 % of the one-layer fixtures.
(b) full depth (36 layers, real vocabulary) through size-independent properties: bitwise determinism, accumulation over a
    protein micro-batch and a DNA micro-batch, train-path loss == eval-path loss, softmax rows sum to one at T = 4096.
(c) configs[2]: Molly-4B (tied head), T = 3072, DNA + RNA + protein spans of K = 512 in every sample, full depth, the same
    properties (what tools/run_mixed_batch.py only printed)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLD, tiny_state_dict

pytestmark = pytest.mark.gpu
KEYS = ("input_ids", "attention_mask", "omic_ids", "omic_info_list", "labels")


def _c4():
    with open(os.path.join(GOLD, "c4_meta.json")) as f:
        meta = json.load(f)
    return meta, dict(np.load(os.path.join(GOLD, "c4_fp32.npz"), allow_pickle=False))


def _c4_model(meta, **prep):
    import molly_amd
    from molly_amd.config import EncConfig, LlmConfig, OmicsModalConfig
    c = meta["config"]
    cfg = OmicsModalConfig(text_config=LlmConfig.from_dict(c["text"]), dna_rna_config=EncConfig.from_dict(c["dna_rna"]),
                           protein_config=EncConfig.from_dict(c["protein"]))
    cfg.dna_rna_project_token_num, cfg.protein_project_token_num = c["K_dna"], c["K_protein"]
    m = molly_amd.OmicsOne(cfg)
    m.model = molly_amd.Qwen3ForCausalLM(cfg.text_config)
    m.dna_rna_model = molly_amd.EsmForMaskedLM(cfg.dna_rna_config)
    m.protein_model = molly_amd.EsmForMaskedLM(cfg.protein_config)
    res = m.load_state_dict(tiny_state_dict(meta))            # strict: the HF encoders' dead heads are accepted and kept
    assert not res.missing_keys and not res.unexpected_keys
    return m.prepare("cuda", **prep)


def _c4_batch(meta, g):
    return {"input_ids": torch.from_numpy(g["in/input_ids"]), "labels": torch.from_numpy(g["in/labels"]),
            "attention_mask": torch.from_numpy(g["in/attention_mask"]),
            "omic_ids": [[torch.from_numpy(g["in/omic_protein"]), torch.from_numpy(g["in/omic_dna"])]],
            "omic_info_list": meta["omic_info_list"]}


def test_config4_sizes_forward_vs_reference():
    meta, g = _c4()
    m = _c4_model(meta)
    b = _c4_batch(meta, g)
    st, sh = meta["sub"]
    assert b["omic_ids"][0][1].numel() == 1000 and b["omic_ids"][0][0].numel() == 1024 and b["input_ids"].shape == (1, 4096)
    rt = m._rt
    # the encoders alone first (K = 1000 on the absolute-position stack: partial 64-key tiles, position ids up to 1001)
    for name, eng, ids in (("enc_dna_rna", rt.dna, b["omic_ids"][0][1]), ("enc_protein", rt.prot, b["omic_ids"][0][0])):
        out = eng.forward(ids[None].cuda()).float().cpu().numpy().reshape(1, ids.numel(), -1)[:, ::st, ::sh]
        ref = g["fwd/" + name]
        err = np.abs(out - ref).max()
        print(f"{name}: max|d| {err:.4f} of max|x| {np.abs(ref).max():.3f}")
        assert err <= 4e-2 * np.abs(ref).max(), (name, err)       # 24 / 33 bf16 layers: measured ~1 %
    with torch.no_grad():
        out = m(*[b[k] for k in KEYS])
    torch.cuda.synchronize()
    logits = out.logits.float().cpu().numpy()[:, ::st, ::sh]
    ref, refb = g["fwd/logits"], g["bf16/logits"]
    err, errb = np.abs(logits - ref).max(), np.abs(refb - ref).max()
    print(f"max|dlogit| ours {err:.4f}  reference-bf16 {errb:.4f}  max|logit| {np.abs(ref).max():.3f}")
    assert err <= max(3e-2 * np.abs(ref).max(), 1.0 * errb)      # no worse than the reference's own bf16 path
    assert abs(out.loss.item() - float(g["fwd/loss"])) <= 5e-3


def test_config4_sizes_all_gradients_vs_reference():
    meta, g = _c4()
    m = _c4_model(meta, train_bio=True)
    b = _c4_batch(meta, g)
    loss = m.forward_backward(*[b[k] for k in KEYS])
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["fwd/loss"])) <= 5e-3
    G = m._rt.G.views
    names = [k[len("gnorm/"):] for k in g if k.startswith("gnorm/")]
    names = [n for n in names if n in G]
    assert len(names) >= 24 * 16 + 33 * 16 + 2 * 11 + 8          # every encoder layer tensor, both decoder layers, the rest
    worst_n, worst_h, checked = 0.0, 0.0, 0
    for n in names:
        got = G[n].float().cpu()
        ref_norm = float(g["gnorm/" + n])
        if n.endswith("key.bias") or ref_norm < 1e-9:
            continue                                   # ~0 by construction (softmax shift invariance)
        head = torch.from_numpy(g["ghead/" + n])
        rms = ref_norm / got.numel() ** 0.5
        # deep stacks: the gradient reaching encoder layer 0 has crossed 2 decoder + 33 encoder layers in bf16
        err = (got.flatten()[:256] - head).abs().max().item()
        # measured: one 3e-7-sized query bias of encoder layer 20 at 1.05x of (6e-2, 0.5 rms); everything else below 0.6x
        # + an absolute floor: the query-bias gradients of the deep encoder layers are sums of ~1e3 signed bf16 terms that
        # cancel to ~1e-6 (layers 20, 25 measured at 1.05-1.25x of the relative bound alone)
        bound = 8e-2 * head.abs().max().item() + 1.0 * rms + 3e-6
        worst_h = max(worst_h, err / bound)
        assert err <= bound, (n, err, bound)
        dn = abs(got.double().norm().item() - ref_norm) / (ref_norm + 5e-4)
        worst_n = max(worst_n, dn)
        assert dn <= 4e-2, (n, got.norm().item(), ref_norm)
        checked += 1
    print(f"{checked} tensors: worst head error / bound {worst_h:.3f}, worst relative norm error {worst_n:.4f}")
    assert checked >= 880                                        # 948 tensors minus the 57 key biases


def _full(size, k_dna, k_prot):
    import molly_amd
    from molly_amd import config as C
    cfg = C.molly(size)
    cfg.dna_rna_project_token_num, cfg.protein_project_token_num = k_dna, k_prot
    m = molly_amd.OmicsOne(cfg)
    m.model = molly_amd.Qwen3ForCausalLM(cfg.text_config)
    m.dna_rna_model = molly_amd.EsmForMaskedLM(cfg.dna_rna_config)
    m.protein_model = molly_amd.EsmForMaskedLM(cfg.protein_config)
    return m.prepare("cuda", random_init_seed=1234)


def _properties(m, micro_batches, lnv_lo=10.0, lnv_hi=13.5):
    """determinism; accumulation over the micro-batches == sum of their separate gradients; train loss == eval loss.
    (Gradient copies stay bf16 and the comparison runs in slices: at 8.2 G parameters three fp32 copies would not fit.)"""
    rt = m._rt
    args = [[b[k] for k in KEYS] for b in micro_batches]
    singles, losses = [], []
    for a in args:
        l1 = m.forward_backward(*a).clone()
        g1 = rt.G.flat.clone()
        l2 = m.forward_backward(*a).clone()
        assert torch.equal(l1, l2) and torch.equal(g1, rt.G.flat)               # no atomics, fixed reduction orders
        assert lnv_lo < l1.item() < lnv_hi, l1.item()                           # ~ln(V) at random init
        with torch.no_grad():
            ev = m(*a).loss
        assert abs(ev.item() - l1.item()) <= 2e-3, (ev.item(), l1.item())       # inference path == training path
        singles.append(g1)
        losses.append(l1.item())
    if len(args) > 1:
        m.forward_backward(*args[0], final_micro=False)
        for i, a in enumerate(args[1:]):
            m.forward_backward(*a, accumulate=True, final_micro=i == len(args) - 2)
        torch.cuda.synchronize()
        n, step = rt.G.flat.numel(), 1 << 28
        gmax = max(float(s.abs().max()) for s in singles)
        worst = 0.0
        for o in range(0, n, step):
            parts = [s[o:o + step].float() for s in singles]
            assert all(bool(torch.isfinite(p_).all()) for p_ in parts)
            want = sum(parts)
            d = (rt.G.flat[o:o + step].float() - want).abs()
            bound = 2 ** -6 * want.abs() + 2 ** -7 * sum(p_.abs() for p_ in parts) + 2 ** -8 * gmax
            worst = max(worst, float((d / bound).max()))
            del parts, want, d, bound
        assert worst <= 1.0, worst
    return losses


def test_config4_full_depth_molly8b_properties():
    """Molly-8B, T = 4096, B = 1 per micro-batch (the reference's dataset cannot stack rows of different K, so a 1024-residue
    protein sample and a 1000-token DNA sample are two micro-batches of one GA window — examples/run_train_8B_z0_b1.sh:29,47)."""
    from molly_amd import ops
    from molly_amd.synth import synth_batch
    m = _full("8b", 1000, 1024)
    mb = [synth_batch(1, 4096, [("protein", 1024)], seed=7), synth_batch(1, 4096, [("dna", 1000)], seed=8)]
    assert mb[1]["omic_ids"].shape == (1, 1, 1000)
    _properties(m, mb)
    # softmax rows sum to one at T = 4096, 32 q / 8 kv heads: V = ones -> O = 1 for every query
    B, T, nh, nkv, hd = 1, 4096, 32, 8, 128
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B * T, (nh + 2 * nkv) * hd, device="cuda", generator=g).bfloat16()
    q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
    v.fill_(1.0)
    o, _ = ops.attn_fwd(q, k, v, B, T, nh, nkv, hd, hd ** -0.5, True)
    assert (o.float() - 1.0).abs().max().item() <= 2 ** -7
    # bidirectional, 20 heads x 64, 1000 keys (not a multiple of the 64-key tile), key range [0, 1000)
    K = 1000
    x = torch.randn(K, 3 * 1280, device="cuda", generator=g).bfloat16()
    x[:, 2560:] = 1.0
    hi = torch.tensor([K], dtype=torch.int32, device="cuda")
    lo = torch.zeros(1, dtype=torch.int32, device="cuda")
    o, _ = ops.attn_fwd(x[:, :1280], x[:, 1280:2560], x[:, 2560:], 1, K, 20, 20, 64, 1.0, False, lo, hi)
    assert (o.float() - 1.0).abs().max().item() <= 2 ** -7


def test_config3_full_depth_molly4b_three_modalities():
    """BASELINE configs[2]: Molly-4B (tied head, h 2560 != n_heads x head_dim 4096), T = 3072, one DNA, one RNA and one protein
    span of 512 tokens in every sample (both encoders active, the DNA/RNA encoder batching two rows per sample), B = 1, GA = 2
    (scripts/train/examples/run_train_4B_z2_b1.sh:29,47)."""
    from molly_amd.synth import synth_batch
    m = _full("4b", 512, 512)
    spans = [("dna", 512), ("rna", 512), ("protein", 512)]
    mb = [synth_batch(1, 3072, spans, seed=3), synth_batch(1, 3072, spans, seed=4)]
    mb[1]["attention_mask"][0, 2900:] = 0                       # a right-padded sample (ragged=True could not fit 3 x 514 tokens)
    mb[1]["labels"][0, 2900:] = -100
    mb[1]["input_ids"][0, 2900:] = 151643
    _properties(m, mb)
    # the three spans landed where the batch says: the injected rows are the projector's outputs, every other row the lookup
    rt = m._rt
    b = mb[0]
    ids = b["input_ids"].reshape(-1).cuda()
    st = m._stage(b["input_ids"], None, None, b["omic_ids"], b["omic_info_list"], want_sort=False)
    hs, _ = m._embed_and_inject(st, False)
    torch.cuda.synchronize()
    emb = rt.llm.embed[ids]
    ow = st.overwritten.bool()
    assert int(ow.sum()) == 3 * 512
    assert torch.equal(hs[~ow], emb[~ow]) and not torch.equal(hs[ow], emb[ow])
    for info in b["omic_info_list"][0]:
        s = info["start"]
        assert bool(ow[s + 1:s + 513].all()) and not bool(ow[s]) and not bool(ow[s + 513])
