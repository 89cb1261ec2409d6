"""GPU: edge cases of the hot path against the oracle — batches without any omic span, samples with and without spans
mixed (collate's {"type": "pad", "start": -1} fillers, reference src/dataset/omics_dataset.py:480-492), spans at the very
start/end of the prompt, a single sample, GA accumulation over different-length micro-batches, and the all-ignored-labels
batch (reference: CrossEntropy mean over zero tokens = NaN, HF:loss/loss_utils.py:32-47)."""
import math

import pytest
import torch

from conftest import tiny_state_dict

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
    m.load_state_dict(tiny_state_dict(meta), strict=False)
    m.prepare("cuda", **prep)
    return m


def _synth(meta, B, T, spans, seed, ragged=False):
    from molly_amd.synth import synth_batch
    sp = {k: tuple(v) for k, v in meta["config"]["special_ids"].items()}
    return synth_batch(B, T, spans, seed=seed, text_vocab=1000, special_ids=sp, pad_id=1000, ragged=ragged)


def _args(b):
    return [b[k] for k in ("input_ids", "attention_mask", "omic_ids", "omic_info_list", "labels")]


def _oracle(meta, b, grads_for=()):
    from oracle import molly_ref as R
    sd = tiny_state_dict(meta)
    leaves = {n: sd[n].clone().requires_grad_(True) for n in grads_for}
    sd.update(leaves)
    llm, dna, prot = R.cfgs_from_meta(meta["config"])
    with torch.set_grad_enabled(bool(grads_for)):
        loss, logits = R.omics_forward(sd, llm, dna, prot, b, {"dna_rna": 64, "protein": 64})
        if grads_for:
            loss.backward()
    return loss, logits, leaves


def _check_forward(m, meta, b):
    with torch.no_grad():
        out = m(*_args(b))
    loss, logits, _ = _oracle(meta, b)
    valid = b["attention_mask"].bool()
    err = (out.logits.float().cpu() - logits)[valid].abs().max().item()
    assert err <= 3e-2 * logits.abs().max().item(), err
    assert abs(out.loss.item() - loss.item()) <= 3e-3
    return out


def test_text_only_batch_and_absent_modalities(tiny_meta):
    """No span anywhere: omic rows are the collate's all-pad filler; also omic_ids=None (plain text batches)."""
    m = _build(tiny_meta)
    b = _synth(tiny_meta, 2, 128, [], seed=1)
    assert all(i["type"] == "pad" for row in b["omic_info_list"] for i in row)
    _check_forward(m, tiny_meta, b)
    with torch.no_grad():
        o1 = m(b["input_ids"], b["attention_mask"], None, None, b["labels"])
        o2 = m(*_args(b))
    assert torch.equal(o1.logits, o2.logits)
    loss = m.forward_backward(*_args(b)).item()
    ref, _, leaves = _oracle(tiny_meta, b, ["model.model.layers.0.mlp.down_proj.weight", "model.model.norm.weight"])
    assert abs(loss - ref.item()) <= 3e-3
    G = m._rt.G.views
    for n, leaf in leaves.items():
        rel = (G[n].float().cpu() - leaf.grad).abs().max().item() / leaf.grad.abs().max().item()
        assert rel < 6e-2, (n, rel)
    for n in ("dna_rna_projector.weight", "protein_projector.weight", "dna_rna_projector.bias", "protein_projector.bias"):
        assert torch.count_nonzero(G[n]) == 0, n                       # absent modality: exactly zero gradient


def test_mixed_samples_with_and_without_spans(tiny_meta):
    """Sample 0 keeps its protein + rna spans, sample 1 loses both (filler rows), sample 2 keeps only the rna span."""
    m = _build(tiny_meta)
    b = _synth(tiny_meta, 3, 384, [("protein", 64), ("rna", 64)], seed=5, ragged=True)
    sp = tiny_meta["config"]["special_ids"]
    specials = {i for v in sp.values() for i in v}

    def drop(bi, j):
        info = b["omic_info_list"][bi][j]
        s = info["start"]
        seg = b["input_ids"][bi, s:s + 66]
        assert int(seg[0]) in specials
        b["input_ids"][bi, s:s + 66] = torch.randint(0, 1000, (66,), generator=torch.Generator().manual_seed(bi * 7 + j))
        b["omic_ids"][bi, j] = 1
        b["omic_info_list"][bi][j] = {"type": "pad", "start": -1}
    drop(1, 0); drop(1, 1); drop(2, 0)
    lab = b["labels"]
    lab[lab != -100] = b["input_ids"][lab != -100]
    _check_forward(m, tiny_meta, b)
    names = ["protein_projector.weight", "dna_rna_projector.weight", "dna_rna_projector.bias",
             "model.model.embed_tokens.weight"]
    loss = m.forward_backward(*_args(b)).item()
    ref, _, leaves = _oracle(tiny_meta, b, names)
    assert abs(loss - ref.item()) <= 3e-3
    for n, leaf in leaves.items():
        got = m._rt.G.views[n].float().cpu()
        rel = (got - leaf.grad).abs().max().item() / leaf.grad.abs().max().item()
        assert rel < 6e-2, (n, rel)


def test_single_sample_span_at_prompt_edges(tiny_meta):
    """B = 1; the span starts at position 0 (start index 0 -> rows 1..K) and a second one ends right before the answer."""
    m = _build(tiny_meta)
    b = _synth(tiny_meta, 1, 256, [("protein", 64), ("dna", 64)], seed=9)
    sp = tiny_meta["config"]["special_ids"]
    T_prompt = 192
    ids = b["input_ids"]
    ids[0, :T_prompt] = torch.randint(0, 1000, (T_prompt,), generator=torch.Generator().manual_seed(0))
    for j, (typ, start) in enumerate((("protein", 0), ("dna", T_prompt - 66))):
        s_id, e_id, p_id = sp[typ]
        ids[0, start] = s_id; ids[0, start + 1:start + 65] = p_id; ids[0, start + 65] = e_id
        b["omic_info_list"][0][j] = {"type": typ, "start": start}
    _check_forward(m, tiny_meta, b)


def test_all_labels_ignored_gives_nan_loss_like_the_reference(tiny_meta):
    m = _build(tiny_meta)
    b = _synth(tiny_meta, 2, 128, [("protein", 64)], seed=11)
    b["labels"][:] = -100
    ref, _, _ = _oracle(tiny_meta, b)
    assert math.isnan(ref.item())
    with torch.no_grad():
        out = m(*_args(b))
    assert math.isnan(out.loss.item())
    loss = m.forward_backward(*_args(b)).item()
    assert math.isnan(loss)


def test_gradient_accumulation_sums_micro_batches_of_different_lengths(tiny_meta):
    """GA: gradients of the micro-steps ADD (each micro loss is its own token mean; the reference does not divide by the
    number of micro-steps — SURVEY.md §0.4-4); micro-batches may differ in B and T (buffers are re-reserved)."""
    m = _build(tiny_meta)
    b1 = _synth(tiny_meta, 2, 256, [("protein", 64)], seed=21)
    b2 = _synth(tiny_meta, 3, 384, [("rna", 64)], seed=22, ragged=True)
    names = ["model.model.layers.1.self_attn.q_proj.weight", "protein_projector.weight", "dna_rna_projector.weight",
             "model.model.layers.0.input_layernorm.weight"]
    m.forward_backward(*_args(b1))
    m.forward_backward(*_args(b2), accumulate=True)
    torch.cuda.synchronize()
    _, _, l1 = _oracle(tiny_meta, b1, names)
    _, _, l2 = _oracle(tiny_meta, b2, names)
    for n in names:
        z = lambda t: t.grad if t.grad is not None else torch.zeros_like(t)        # modality absent from that micro-batch
        ref = z(l1[n]) + z(l2[n])
        got = m._rt.G.views[n].float().cpu()
        rel = (got - ref).abs().max().item() / ref.abs().max().item()
        assert rel < 6e-2, (n, rel)


def test_frozen_encoder_forward_replayed_from_a_graph_equals_the_eager_launches(tiny_meta, monkeypatch):
    """The frozen encoders' forward is captured into a hipGraph on the second call with one shape (esm.py::_forward_replayed): the replays
    must be the eager launches' values bit for bit on FRESH ids (the captured input is a copy target, not the first batch), a shape change must
    drop the graph, and the engine must come back to replaying on the new shape."""
    from molly_amd import esm
    monkeypatch.setattr(esm, "_ENC_GRAPH", True)
    m = _build(tiny_meta, train_llm=True, train_mlp=True)
    eng = m._rt.prot
    g = torch.Generator().manual_seed(3)
    vocab = eng.cfg.vocab_size

    def ids(n, K, ragged):
        t = torch.randint(4, vocab, (n, K), generator=g)
        if ragged:
            t[0, K - 5:] = eng.cfg.pad_token_id                  # right-padded row: masked keys
        return t.cuda()

    def eager(x):
        return eng._forward_frozen(x, *x.shape).clone()

    from molly_amd import ops
    ctx = m._rt.gemm_ctx
    ctx.ensure_workspace(0)
    with ops.use_gemm_context(ctx):
        _replay_checks(eng, ids, eager)


def test_a_failed_encoder_graph_capture_leaves_the_engine_eager(tiny_meta, monkeypatch):
    """a capture that raises (an op illegal inside a capture, no memory for the graph pool) must not be retried on every later
    call — the engine warns once, stays eager and keeps producing the eager values."""
    from molly_amd import esm, ops
    monkeypatch.setattr(esm, "_ENC_GRAPH", True)
    m = _build(tiny_meta, train_llm=True, train_mlp=True)
    eng = m._rt.prot
    eng._g_recaptures = 0
    x = torch.randint(4, eng.cfg.vocab_size, (3, 64), generator=torch.Generator().manual_seed(5)).cuda()
    ctx = m._rt.gemm_ctx
    ctx.ensure_workspace(0)
    with ops.use_gemm_context(ctx):
        eng.forward(x)                                           # first call with the shape: eager (and the buffers exist)
        ref = eng._forward_frozen(x, 3, 64).clone()

        class Boom:
            def __init__(self, *a, **k):
                pass

            def __enter__(self):
                raise RuntimeError("hipErrorStreamCaptureUnsupported (stand-in)")

            def __exit__(self, *a):
                return False
        monkeypatch.setattr(torch.cuda, "graph", Boom)
        with pytest.warns(UserWarning, match="running eagerly"):
            got = eng.forward(x).clone()                         # second call: the capture is attempted and fails
        assert torch.equal(got, ref) and eng._g is None and eng._g_recaptures >= 4
        got = eng.forward(x).clone()                             # later calls do not try again (Boom would raise: no warning, no error)
        assert torch.equal(got, ref)


def _replay_checks(eng, ids, eager):
    eng._g_recaptures = 0
    for call in range(6):
        x = ids(3, 64, ragged=call % 2 == 1)
        got = eng.forward(x).clone()
        assert (eng._g is not None) == (call >= 1), call         # eager, capture + replay, replay ...
        assert torch.equal(got, eager(x)), call
    assert eng._g_recaptures == 1
    x = ids(2, 32, ragged=False)                                 # another shape: buffers are reallocated, the graph must go
    got = eng.forward(x).clone()
    assert eng._g is None and torch.equal(got, eager(x))
    for call in range(2):
        x = ids(2, 32, ragged=True)
        got = eng.forward(x).clone()
        assert torch.equal(got, eager(x))
    assert eng._g is not None and eng._g_recaptures == 2
