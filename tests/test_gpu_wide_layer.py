"""GPU: G4 of SURVEY.md 8(c) — one layer of every stack at the REAL widths of Molly-1.7B (Qwen3 2048 / 16q-8kv x 128 / 6144),
4B (2560 / 32q-8kv / 9728) and 8B (4096 / 32q-8kv / 12288) with the encoders at 1280 / 20 x 64 / 5120 (absolute-position and
rotary) through the HIP path against golden vectors of the reference's
OmicsOne (fp32; tests/golden/gen_golden_wide.py).  Unlike the tiny fixture this one runs the production tile paths: the
256x256 GEMM, head_dim-128 GQA attention, 20-head encoders, real-width norms.  Tolerance: the stated bf16 bound
(max|dlogit| <= 3e-2 max|logit|); the reference's own bf16 CPU path is stored beside as the yardstick."""
import numpy as np
import pytest
import torch

from conftest import WIDE_TAGS, tiny_batch, tiny_state_dict, wide_fixture

pytestmark = pytest.mark.gpu


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
    missing, _ = m.load_state_dict(tiny_state_dict(meta), strict=False)
    assert not missing, missing
    return m.prepare("cuda", **prep)


@pytest.mark.parametrize("tag", WIDE_TAGS)
def test_forward_at_real_widths_vs_reference(tag):
    wide_meta, g = wide_fixture(tag)
    m = _build(wide_meta)
    batch = tiny_batch(g, wide_meta)
    st, sh = wide_meta["sub"]
    with torch.no_grad():
        out = m(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], omic_ids=batch["omic_ids"],
                omic_info_list=batch["omic_info_list"], labels=batch["labels"])
    torch.cuda.synchronize()
    logits = out.logits.float().cpu().numpy()[:, ::st, ::sh]
    ref, refb = g["fwd/logits"], g["bf16/logits"]
    valid = batch["attention_mask"].numpy()[:, ::st].astype(bool)
    err = np.abs(logits - ref)[valid].max()
    errb = np.abs(refb - ref)[valid].max()
    tol = 3e-2 * np.abs(ref).max()
    print(f"max|dlogit| ours {err:.4f}  reference-bf16 {errb:.4f}  tol {tol:.4f}")
    assert err <= tol
    assert abs(out.loss.item() - float(g["fwd/loss"])) <= 5e-3
    assert abs(out.loss.item() - float(g["fwd/loss"])) <= 3 * abs(float(g["bf16/loss"]) - float(g["fwd/loss"])) + 1e-3


@pytest.mark.parametrize("tag", WIDE_TAGS)
def test_all_gradients_at_real_widths_vs_reference(tag):
    """LLM + projectors + both encoders trainable: the 56 (tied head) / 57 gradient tensors of the fixture."""
    wide_meta, g = wide_fixture(tag)
    m = _build(wide_meta, train_bio=True)
    batch = tiny_batch(g, wide_meta)
    loss = m.forward_backward(batch["input_ids"], batch["attention_mask"], batch["omic_ids"], batch["omic_info_list"],
                              batch["labels"])
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["fwd/loss"])) <= 5e-3
    G = m._rt.G.views
    tied = wide_meta["config"]["text"]["tie_word_embeddings"]
    names = [k[len("gnorm/"):] for k in g if k.startswith("gnorm/") and not (tied and k == "gnorm/model.lm_head.weight")]
    assert len(names) >= 56
    worst = 0.0
    for n in names:
        got = G[n].float().cpu()
        ref_norm = float(g["gnorm/" + n])
        head = torch.from_numpy(g["ghead/" + n])
        if n.endswith("key.bias"):
            continue                                   # ~0 by construction; judged in test_gpu_train_bio.py
        # the stored head (first 256 entries) on its own largest entry plus the tensor's rms (rounding noise floor)
        rms = ref_norm / got.numel() ** 0.5
        err = (got.flatten()[:256] - head).abs().max().item()
        bound = 3e-2 * head.abs().max().item() + 0.2 * rms + 1e-9
        worst = max(worst, err / bound)
        assert err <= bound, (n, err, bound)
        assert abs(got.double().norm().item() - ref_norm) <= 2e-2 * ref_norm + 1e-7, (n, got.norm().item(), ref_norm)
    print(f"{len(names)} tensors, worst error / bound {worst:.3f}")
    # gradient accumulation through the same launches (grouped weight gradients at the 1.7B widths): a second micro-step on
    # the same batch with accumulate=True doubles every gradient (bf16 read-modify-write of the first pass's values)
    g1 = {n: G[n].float().clone() for n in names}
    m.forward_backward(batch["input_ids"], batch["attention_mask"], batch["omic_ids"], batch["omic_info_list"],
                       batch["labels"], accumulate=True)
    torch.cuda.synchronize()
    for n in names:
        ref2 = 2 * g1[n]
        assert (G[n].float() - ref2).abs().max().item() <= 2 ** -7 * ref2.abs().max().item() + 1e-12, n


@pytest.mark.parametrize("tag", ["wide", "wide4b"])
def test_ragged_batch_at_real_widths_vs_oracle(tag):
    """Fresh seeded batch whose token count is NOT a multiple of any tile (3 x 416 tokens, ragged right-padded text, ragged
    omic spans) through the production kernels at real widths, against the oracle on the same inputs: partial 256- and
    128-row tiles, partial attention blocks, the scored-row gather of lm_head + CE; loss and every gradient."""
    from molly_amd.synth import synth_batch
    from oracle import molly_ref as R
    meta, _ = wide_fixture(tag)
    c = meta["config"]
    m = _build(meta, train_bio=True)
    sp = {k: tuple(v) for k, v in c["special_ids"].items()}
    batch = synth_batch(3, 416, [("protein", 64), ("rna", 64)], seed=11, text_vocab=1000, special_ids=sp, pad_id=1000, ragged=True)
    loss = m.forward_backward(*[batch[k] for k in ("input_ids", "attention_mask", "omic_ids", "omic_info_list", "labels")])
    torch.cuda.synchronize()
    llm, dna, prot = R.cfgs_from_meta(c)
    sd = tiny_state_dict(meta)
    tied = c["text"]["tie_word_embeddings"]
    names = [n for n in m._rt.G.views if not (tied and n == "model.lm_head.weight")]
    for n in names:
        sd[n].requires_grad_(True)
    if tied:
        sd["model.lm_head.weight"] = sd["model.model.embed_tokens.weight"]
    ref_loss, _ = R.omics_forward(sd, llm, dna, prot, batch, {"dna_rna": 64, "protein": 64})
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) <= 5e-3, (loss.item(), ref_loss.item())
    G = m._rt.G.views
    checked = 0
    for n in names:
        ref = sd[n].grad
        got = G[n].float().cpu()
        if ref is None or ref.abs().max().item() == 0.0:
            assert torch.count_nonzero(got) == 0, n
            continue
        if n.endswith("key.bias"):
            continue
        scale = ref.abs().max().item()
        rel = (got - ref).abs().max().item() / scale
        assert rel < 6e-2, (n, rel)
        assert abs(got.double().norm().item() - ref.double().norm().item()) <= 2e-2 * ref.norm().item() + 1e-7, n
        checked += 1
    assert checked >= 50


def test_lora_rank64_at_real_widths_vs_oracle():
    """`--use-lora` at Molly-1.7B's widths: rank-64 adapters on all seven targets of the layer (the production rank: no
    padding), base frozen, projectors trainable; the fourteen adapter gradients go through ONE grouped launch.  Loss and every
    adapter / projector gradient against the oracle's autograd on a ragged 3 x 448-token batch (a multiple of 64, not of 256)."""
    from molly_amd.lora import LoraConfig
    from molly_amd.synth import synth_batch
    from oracle import molly_ref as R
    from test_gpu_lora import _check_grads, _oracle_sd_with_lora, _randomize_B
    meta, _ = wide_fixture("wide")
    c = meta["config"]
    m = _build(meta, train_llm=False, lora=LoraConfig(r=64, lora_alpha=64, lora_dropout=0.0))
    _randomize_B(m, std=0.02)
    sp = {k: tuple(v) for k, v in c["special_ids"].items()}
    b = synth_batch(3, 448, [("protein", 64), ("rna", 64)], seed=13, text_vocab=1000, special_ids=sp, pad_id=1000, ragged=True)
    loss = m.forward_backward(*[b[k] for k in ("input_ids", "attention_mask", "omic_ids", "omic_info_list", "labels")])
    torch.cuda.synchronize()
    assert m._rt.llm.lora_tT is not None                                 # the grouped adapter-gradient path is the one that ran
    sd, leaves = _oracle_sd_with_lora(meta, m)
    llm, dna, prot = R.cfgs_from_meta(c)
    ref_loss, _ = R.omics_forward(sd, llm, dna, prot, b, {"dna_rna": 64, "protein": 64})
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) <= 5e-3, (loss.item(), ref_loss.item())
    worst = _check_grads(m, leaves)
    print("worst relative adapter-grad error at real widths", worst)
    assert len(leaves) == 14 + 4


def test_transposed_second_stores_give_the_transpose_launches_gradients(monkeypatch):
    """Round 5: with the grouped weight-gradient launch on, the two RMSNorm forwards and the attention forward store their results a second time,
    transposed (qwen3.py::reserve: xnT / xn2T / attnT), instead of a transpose launch per operand in the backward.  Same numbers either way:
    every gradient tensor of a 1.7B-width layer at 2 x 1,024 tokens must be bit-identical with MOLLY_NORM_TRANSPOSED_STORE=0."""
    from molly_amd.synth import synth_batch
    wide_meta, g = wide_fixture("wide")
    sp = {k: tuple(v) for k, v in wide_meta["config"]["special_ids"].items()}
    vocab = wide_meta["config"]["text"]["vocab_size"]
    b = synth_batch(2, 1024, [("protein", wide_meta["config"]["K"])], seed=11, text_vocab=min(vocab, 1000), special_ids=sp, pad_id=min(vocab, 1000))
    args = [b[k] for k in ("input_ids", "attention_mask", "omic_ids", "omic_info_list", "labels")]
    monkeypatch.setenv("MOLLY_GROUPED_WGRAD", "2")                  # (the grouped launch is otherwise taken at the step's row counts only)
    grads = {}
    for arm in ("1", "0"):
        monkeypatch.setenv("MOLLY_NORM_TRANSPOSED_STORE", arm)
        m = _build(wide_meta, train_llm=True, train_mlp=True)
        loss = m.forward_backward(*args)
        torch.cuda.synchronize()
        a0 = m._rt.llm.A[0]
        assert ("xnT" in a0 and "attnT" in a0) == (arm == "1"), sorted(a0)          # the arm really ran the path it names
        grads[arm] = ({n: v.clone() for n, v in m._rt.G.views.items()}, float(loss))
    assert grads["1"][1] == grads["0"][1]
    for n, v in grads["1"][0].items():
        assert torch.equal(v, grads["0"][0][n]), n


def test_lora_up_projection_as_trailing_k_tiles_equals_the_accumulating_launches(monkeypatch):
    """Round 5: at the step's row counts the adapters' y += t B^T rides in the base projection (qwen3.py: `lora_kx`, molly_gemm_kx_bf16_ctx) — the
    fused q|k|v and gate|up projections through block-diagonal stacks of their targets' B, gate|up with its SwiGLU epilogue.  4 x 2,048 tokens of a
    1.7B-width layer, rank 64, dropout 0.05 (the masks are functions of (seed, layer, target, step): the same in both arms): the loss and all fourteen
    adapter gradients + the projectors' against the per-target accumulating launches (MOLLY_LORA_KX=0) within bf16 rounding; then, without
    dropout, against the oracle's autograd."""
    from molly_amd import qwen3
    from molly_amd.lora import LoraConfig
    from molly_amd.synth import synth_batch
    from oracle import molly_ref as R
    from test_gpu_lora import _check_grads, _oracle_sd_with_lora, _randomize_B
    meta, _ = wide_fixture("wide")
    c = meta["config"]
    sp = {k: tuple(v) for k, v in c["special_ids"].items()}
    b = synth_batch(4, 2048, [("protein", 64)], seed=17, text_vocab=1000, special_ids=sp, pad_id=1000)
    args = [b[k] for k in ("input_ids", "attention_mask", "omic_ids", "omic_info_list", "labels")]
    runs = {}
    for arm in (True, False):
        monkeypatch.setattr(qwen3, "_LORA_KX", arm)
        m = _build(meta, train_llm=False, lora=LoraConfig(r=64, lora_alpha=64, lora_dropout=0.05))
        _randomize_B(m, std=0.02)
        loss = m.forward_backward(*args)
        torch.cuda.synchronize()
        assert m._rt.llm.lora_kx == arm                                    # the arm ran the path it names
        with torch.no_grad():                                              # the inference forward (no mask, live adapters) takes the same launches
            logits = m(input_ids=b["input_ids"], attention_mask=b["attention_mask"], omic_ids=b["omic_ids"],
                       omic_info_list=b["omic_info_list"]).logits.float().cpu()
        runs[arm] = (float(loss), {n: v.float().clone() for n, v in m._rt.G.views.items()}, logits)
    assert abs(runs[True][0] - runs[False][0]) <= 2e-3, (runs[True][0], runs[False][0])
    assert (runs[True][2] - runs[False][2]).abs().max().item() <= 3e-2 * runs[False][2].abs().max().item()
    for n, g1 in runs[True][1].items():
        g0 = runs[False][1][n]
        assert (g1 - g0).norm().item() <= 0.03 * g0.norm().item() + 1e-6, (n, (g1 - g0).norm().item(), g0.norm().item())
    # without dropout: the K-extended path against the oracle
    monkeypatch.setattr(qwen3, "_LORA_KX", True)
    m = _build(meta, train_llm=False, lora=LoraConfig(r=64, lora_alpha=64, lora_dropout=0.0))
    _randomize_B(m, std=0.02)
    b2 = synth_batch(4, 2048, [("protein", 64), ("rna", 64)], seed=18, text_vocab=1000, special_ids=sp, pad_id=1000)   # (both projectors get a gradient)
    loss = m.forward_backward(*[b2[k] for k in ("input_ids", "attention_mask", "omic_ids", "omic_info_list", "labels")])
    torch.cuda.synchronize()
    assert m._rt.llm.lora_kx
    sd, leaves = _oracle_sd_with_lora(meta, m)
    llm, dna, prot = R.cfgs_from_meta(c)
    ref_loss, _ = R.omics_forward(sd, llm, dna, prot, b2, {"dna_rna": 64, "protein": 64})
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) <= 5e-3, (loss.item(), ref_loss.item())
    _check_grads(m, leaves)
