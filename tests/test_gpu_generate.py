"""GPU: prefill + KV-cache decode (OmicsOne.generate) against the oracle run as a full re-forward on the grown sequence
(teacher forcing, so bf16 argmax ties cannot derail the comparison), left-padded batch like the reference's Test collate."""
import pytest
import torch

from conftest import tiny_state_dict
from test_gpu_model import build_tiny

pytestmark = pytest.mark.gpu


def _left_padded_batch(meta):
    from molly_amd.synth import synth_batch
    c = meta["config"]
    sp = {k: tuple(v) for k, v in c["special_ids"].items()}
    b = synth_batch(2, 192, [("protein", 64)], seed=11, text_vocab=1000, special_ids=sp, pad_id=1000)
    # make sample 1 shorter and LEFT-pad it (reference Test mode: omics_dataset.py:384-391 shifts the span start too)
    pad = 37
    ids, mask = b["input_ids"].clone(), b["attention_mask"].clone()
    ids[1] = torch.cat([torch.full((pad,), 1000), b["input_ids"][1, :-pad]])
    mask[1] = torch.cat([torch.zeros(pad, dtype=torch.long), torch.ones(192 - pad, dtype=torch.long)])
    info = [[dict(d) for d in row] for row in b["omic_info_list"]]
    info[1][0]["start"] += pad
    return ids, mask, b["omic_ids"], info


def test_decode_logits_match_oracle_reforward(tiny_meta):
    from molly_amd.generate import GenerationSession
    from oracle import molly_ref as R
    m = build_tiny(tiny_meta)
    ids, mask, omic, info = _left_padded_batch(tiny_meta)
    sess = GenerationSession(m, max_new_tokens=6)
    logits = sess.prefill(ids, mask, omic, info)
    c = tiny_meta["config"]
    llm, dna, prot = R.cfgs_from_meta(c)
    sd = tiny_state_dict(tiny_meta)

    batch = {"input_ids": ids, "attention_mask": mask, "omic_ids": omic, "omic_info_list": info}
    oracle_last_logits = lambda new_tokens: R.generate_last_logits(sd, llm, dna, prot, batch, new_tokens,
                                                                   {"dna_rna": c["K"], "protein": c["K"]})

    new = torch.empty(2, 0, dtype=torch.long)
    for step in range(5):
        ref = oracle_last_logits(new)
        err = (logits.float().cpu() - ref).abs().max().item()
        assert err <= 3e-2 * ref.abs().max().item(), (step, err)
        nxt = ref.argmax(-1)                     # teacher forcing with the oracle's choice
        new = torch.cat([new, nxt[:, None]], 1)
        logits = sess.step(nxt)


def test_generate_api_returns_new_tokens_only_and_stops_at_eos(tiny_meta):
    m = build_tiny(tiny_meta)
    ids, mask, omic, info = _left_padded_batch(tiny_meta)
    out = m.generate(ids, mask, omic, info, do_sample=False, max_new_tokens=7)
    assert out.shape == (2, 7) and out.dtype == torch.int64
    # greedy is deterministic
    assert torch.equal(out, m.generate(ids, mask, omic, info, do_sample=False, max_new_tokens=7))
    # eos handling: declare the first generated token of sample 0 to be EOS -> that row is padded afterwards
    m.text_config.eos_token_id = int(out[0, 0])
    m.text_config.pad_token_id = 1000
    out2 = m.generate(ids, mask, omic, info, do_sample=False, max_new_tokens=7)
    assert out2[0, 0] == out[0, 0] and bool((out2[0, 1:] == 1000).all())
    g = torch.Generator(device="cuda").manual_seed(0)
    s1 = m.generate(ids, mask, omic, info, do_sample=True, temperature=0.8, top_p=0.95, top_k=20, repetition_penalty=1.1,
                    max_new_tokens=5, generator=g)
    assert s1.shape[0] == 2 and s1.shape[1] <= 5


def test_lora_adapter_merge_peft_layout(tiny_meta, tiny_gold, tmp_path):
    """A synthetic adapter in PEFT's file layout, merged at load; logits must equal the oracle run on W + (alpha/r) B A."""
    import json
    from safetensors.torch import save_file
    from conftest import tiny_batch
    from molly_amd.lora import TARGETS, merge_lora_adapter
    from oracle import molly_ref as R
    m = build_tiny(tiny_meta)
    sd = tiny_state_dict(tiny_meta)
    r, alpha = 8, 64
    g = torch.Generator().manual_seed(5)
    tens = {}
    for i in range(tiny_meta["config"]["text"]["num_hidden_layers"]):
        for t in TARGETS:
            sub = "self_attn" if t in ("q_proj", "k_proj", "v_proj", "o_proj") else "mlp"
            w = sd[f"model.model.layers.{i}.{sub}.{t}.weight"]
            A = torch.randn(r, w.shape[1], generator=g) * 0.02
            Bm = torch.randn(w.shape[0], r, generator=g) * 0.02
            tens[f"base_model.model.model.layers.{i}.{sub}.{t}.lora_A.weight"] = A
            tens[f"base_model.model.model.layers.{i}.{sub}.{t}.lora_B.weight"] = Bm
            sd[f"model.model.layers.{i}.{sub}.{t}.weight"] = w + (alpha / r) * (Bm @ A)
    save_file(tens, str(tmp_path / "adapter_model.safetensors"))
    json.dump({"r": r, "lora_alpha": alpha, "target_modules": list(TARGETS), "peft_type": "LORA"},
              open(tmp_path / "adapter_config.json", "w"))
    torch.save({"weight": sd["protein_projector.weight"] * 1.5, "bias": sd["protein_projector.bias"]},
               tmp_path / "protein_projector.bin")
    sd["protein_projector.weight"] = sd["protein_projector.weight"] * 1.5
    assert merge_lora_adapter(m, str(tmp_path)) == 14
    batch = tiny_batch(tiny_gold, tiny_meta)
    with torch.no_grad():
        out = m(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], omic_ids=batch["omic_ids"],
                omic_info_list=batch["omic_info_list"])
        llm, dna, prot = R.cfgs_from_meta(tiny_meta["config"])
        _, ref = R.omics_forward(sd, llm, dna, prot, {k: v for k, v in batch.items() if k != "labels"}, {"dna_rna": 64, "protein": 64})
    valid = batch["attention_mask"].bool()
    err = (out.logits.float().cpu() - ref)[valid].abs().max().item()
    assert err <= 3e-2 * ref.abs().max().item(), err


def test_greedy_decode_follows_the_reference_generate_golden(tiny_meta):
    """G7: the 8 tokens the REFERENCE's `OmicsOne.generate(do_sample=False)` produced for three left-padded prompts of
    different lengths (tests/golden/generate_g7.json, made by tests/golden/gen_golden_generate.py).  Teacher-forced with the
    reference's tokens: the HIP path's argmax must be the reference's token wherever the reference's own top-1/top-2 margin
    is not a near-tie under bf16 (0.05), and in its top-2 otherwise."""
    import json
    import os
    from molly_amd.generate import GenerationSession
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "generate_g7.json")))
    m = build_tiny(tiny_meta)
    ids, mask = torch.tensor(g["input_ids"]), torch.tensor(g["attention_mask"])
    omic, info = torch.tensor(g["omic_ids"]), g["omic_info_list"]
    ref, margins = torch.tensor(g["new_tokens"]), torch.tensor(g["margins"])           # [B, n], [n, B]
    sess = GenerationSession(m, g["n_new"])
    logits = sess.prefill(ids, mask, omic, info)
    strict = 0
    for t in range(g["n_new"]):
        top2 = logits.float().topk(2, dim=-1).indices.cpu()
        for b in range(ids.shape[0]):
            if margins[t, b] > 0.05:
                assert top2[b, 0] == ref[b, t], (t, b, top2[b].tolist(), int(ref[b, t]))
                strict += 1
            else:
                assert int(ref[b, t]) in top2[b].tolist(), (t, b)
        if t + 1 < g["n_new"]:
            logits = sess.step(ref[:, t].cuda())
    assert strict >= 12
    # free-running API call: same tokens as long as no near-tie was crossed
    with torch.no_grad():
        out = m.generate(input_ids=ids, attention_mask=mask, omic_ids=omic, omic_info_list=info, do_sample=False,
                         max_new_tokens=g["n_new"]).cpu()
    assert out.shape == ref.shape
    for b in range(ids.shape[0]):
        for t in range(g["n_new"]):
            if margins[t, b] <= 0.05:
                break
            assert out[b, t] == ref[b, t], (b, t)


def test_sampling_kernel_matches_hf_logits_processors():
    """csrc/sampling.hip against HuggingFace's own processor classes (the third-party home of this arithmetic,
    HF:generation/logits_process.py) at the reference's inference settings (src/inference_lora.py:293-298: repetition penalty 1.1,
    temperature 0.8, top_k 20, top_p 0.95) on the real vocabulary: same surviving tokens, same probabilities, and the Philox
    draws follow that distribution."""
    from molly_amd import ops
    from transformers.generation.logits_process import (RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper,
                                                        TopKLogitsWarper, TopPLogitsWarper)
    rows, V = 6, 151936
    g = torch.Generator().manual_seed(3)
    logits = torch.randn(rows, V, generator=g) * 3.0
    logits[1] *= 4.0                                              # a peaked row: top-p cuts deep into the top-k set
    logits[2, :] = torch.randn(V, generator=g) * 0.01             # a flat row: all 20 survive
    gen = torch.randint(0, V, (rows, 37), generator=g)
    gen[:, 5] = gen[:, 4]                                         # duplicates are penalised once
    top = logits.topk(3, dim=1).indices
    gen[:, 0] = top[:, 0]                                         # the current best token has been generated before
    gen[3, 1] = top[3, 1]
    scores = logits.clone()
    for proc in (RepetitionPenaltyLogitsProcessor(1.1), TemperatureLogitsWarper(0.8), TopKLogitsWarper(20), TopPLogitsWarper(0.95)):
        scores = proc(gen, scores)
    want = scores.softmax(-1)
    dev_logits = logits.cuda()
    nxt, (probs, ids, n_kept) = ops.sample_logits(dev_logits.clone(), gen.cuda(), 1.1, 0.8, 20, 0.95, seed=1234, step=0, debug_cap=32)
    torch.cuda.synchronize()
    for r in range(rows):
        keep = torch.nonzero(want[r] > 0).reshape(-1)
        n = int(n_kept[r])
        assert n == len(keep), (r, n, len(keep))
        got_ids = ids[r, :n].cpu()
        assert sorted(got_ids.tolist()) == sorted(keep.tolist())
        assert torch.allclose(probs[r, :n].cpu(), want[r, got_ids], rtol=1e-4, atol=1e-6), r
        assert bool((probs[r, :n - 1] >= probs[r, 1:n]).all())   # descending
        assert int(nxt[r]) in keep.tolist()
    assert int(n_kept[1]) < int(n_kept[2]) and int(n_kept[2]) >= 19     # flat row: at most the smallest of the 20 is cut
    # the draws: 3000 steps on the same scores -> empirical frequencies within 4 sigma of the probabilities
    counts = torch.zeros(rows, V, dtype=torch.int32)
    outs = []
    for step in range(3000):
        outs.append(ops.sample_logits(dev_logits.clone(), gen.cuda(), 1.1, 0.8, 20, 0.95, seed=99, step=step))
    torch.cuda.synchronize()
    outs = torch.stack(outs, 1).cpu()
    for r in range(rows):
        c = torch.bincount(outs[r], minlength=V).float()
        assert bool((c[want[r] == 0] == 0).all())                # never a removed token
        p = want[r]
        sigma = (3000 * p * (1 - p)).sqrt()
        sel = p > 0
        assert bool(((c[sel] - 3000 * p[sel]).abs() <= 4 * sigma[sel] + 2).all()), r
    # same (seed, step) -> same token; another seed -> another stream
    a = ops.sample_logits(dev_logits.clone(), gen.cuda(), 1.1, 0.8, 20, 0.95, seed=7, step=5)
    b = ops.sample_logits(dev_logits.clone(), gen.cuda(), 1.1, 0.8, 20, 0.95, seed=7, step=5)
    assert torch.equal(a, b)


def test_sampling_long_history_and_tie_overflow_are_deterministic():
    """(a) more than 4096 generated tokens enter the repetition penalty (the kernel once took at most 4 per thread), each
    DISTINCT token once; (b) more than 1024 scores tying with the k-th largest: the survivors are the strictly larger scores
    plus the ties of lowest token id, the same set on every launch."""
    from molly_amd import ops
    V = 151936
    g = torch.Generator().manual_seed(11)
    logits = torch.randn(2, V, generator=g) * 2.0
    gen = torch.randint(0, V, (2, 9000), generator=g)
    gen[:, 4500:] = gen[:, :4500]                                 # every token listed twice
    top = logits.topk(20, dim=1).indices
    gen[:, 8000:8020] = top                                       # the whole top-20 was generated before (positions > 4096)
    want = logits.clone()
    for r in range(2):
        ids = gen[r].unique()
        v = want[r, ids]
        want[r, ids] = torch.where(v < 0, v * 1.3, v / 1.3)
    wk = (want / 0.8).topk(20, dim=1)
    _, (probs, ids, n_kept) = ops.sample_logits(logits.cuda(), gen.cuda(), 1.3, 0.8, 20, 1.0, seed=1, step=0, debug_cap=32)
    torch.cuda.synchronize()
    for r in range(2):
        assert int(n_kept[r]) == 20
        assert sorted(ids[r, :20].cpu().tolist()) == sorted(wk.indices[r].tolist())
        assert torch.allclose(probs[r, :20].cpu(), wk.values[r].softmax(-1), rtol=1e-4, atol=1e-6)
    # (b) 3000 scores tie at the top, top_k = 1024, five scores above them
    tie = torch.full((1, V), -5.0)
    pos = torch.randperm(V, generator=g)[:3000].sort().values
    tie[0, pos] = 1.0
    above = torch.tensor([17, 40000, 90001, 120000, 151935])
    above = above[~torch.isin(above, pos)]
    tie[0, above] = 2.0
    expect = sorted(above.tolist() + pos[:1024 - len(above)].tolist())
    got = None
    for rep in range(3):
        _, (p_, ids_, n_) = ops.sample_logits(tie.cuda(), None, 1.0, 1.0, 1024, 1.0, seed=2, step=rep, debug_cap=1024)
        torch.cuda.synchronize()
        assert int(n_[0]) == 1024
        cur = sorted(ids_[0].cpu().tolist())
        assert cur == expect
        got = cur if got is None else got
        assert cur == got


def test_sampled_generate_calls_with_one_generator_are_independent(tiny_meta):
    """Advisor finding (round 2): the Philox seed was generator.initial_seed(), which never advances — every generate() call of
    a dataset drew the same uniforms.  The seed is now drawn from the generator: same manual_seed -> same tokens, two calls in
    a row -> different streams (as torch.multinomial behaves in the reference path, src/inference_lora.py:293-298)."""
    m = build_tiny(tiny_meta)
    ids, mask, omic, info = _left_padded_batch(tiny_meta)
    kw = dict(do_sample=True, temperature=1.5, top_p=1.0, top_k=200, repetition_penalty=1.0, max_new_tokens=12)
    g = torch.Generator(device="cuda").manual_seed(0)
    a = m.generate(ids, mask, omic, info, generator=g, **kw)
    b = m.generate(ids, mask, omic, info, generator=g, **kw)
    assert not torch.equal(a, b)                                   # 24 draws from >= 200 candidates each: equal only by a bug
    g2 = torch.Generator(device="cuda").manual_seed(0)
    assert torch.equal(a, m.generate(ids, mask, omic, info, generator=g2, **kw))
    torch.manual_seed(123)                                         # no generator: torch's global stream, consumed per call
    c = m.generate(ids, mask, omic, info, **kw)
    d = m.generate(ids, mask, omic, info, **kw)
    assert not torch.equal(c, d)
    torch.manual_seed(123)
    assert torch.equal(c, m.generate(ids, mask, omic, info, **kw))


def test_captured_decode_graph_survives_a_replaced_gemm_scratch(tiny_meta):
    """a session's captured decode graph carries the model context's scratch pointers.  When something else on the same
    model grows — i.e. replaces — that scratch, the session must keep the captured tensor alive and capture again, not replay into
    freed memory.  Reference role: HF generate's decode loop, `src/model/omics_one.py:220-232`."""
    from molly_amd.generate import GenerationSession
    m = build_tiny(tiny_meta)
    ids, mask, omic, info = _left_padded_batch(tiny_meta)

    def run(disturb):
        sess = GenerationSession(m, max_new_tokens=8, use_graph=True)
        logits = sess.prefill(ids, mask, omic, info)
        outs = []
        for step in range(6):
            nxt = logits.argmax(-1)
            logits = sess.step(nxt)
            outs.append(logits.clone())
            if disturb and step == 2:
                assert sess._graph is not None
                old = sess._graph_ws_ptr
                ctx = sess.rt.gemm_ctx
                ctx.ensure_workspace(ctx.ws.numel() * 4 + (64 << 20), sess.rt.dev)      # somebody else's larger request
                assert ctx.ws.data_ptr() != old and sess._graph_ws.data_ptr() == old     # the captured tensor is still alive
                torch.empty(ctx.ws.numel(), dtype=torch.float32, device="cuda").fill_(float("nan"))   # whatever reuses memory
        return outs

    a, b = run(False), run(True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_gemm_scratch_refuses_to_grow_inside_a_capture():
    from molly_amd import ops
    ctx = ops.GemmContext()
    ctx.ensure_workspace(1 << 20)
    g = torch.cuda.CUDAGraph()
    with pytest.raises(RuntimeError, match="during a hipGraph capture"):
        with torch.cuda.graph(g):
            ctx.ensure_workspace(ctx.ws.numel() * 4 + (8 << 20))


def test_generate_with_no_repeat_ngram(tiny_meta):
    """`no_repeat_ngram_size` through the reference's generate signature (src/model/omics_one.py:199-200, 227): greedy decoding of the
    random-init tiny model repeats itself at once without it; with n = 2 no bigram of a row's new tokens occurs twice.  (The ban itself
    is checked against HuggingFace's processor in tests/test_no_repeat_ngram.py.)"""
    m = build_tiny(tiny_meta)
    ids, mask, omic, info = _left_padded_batch(tiny_meta)
    out = m.generate(ids, mask, omic, info, do_sample=False, max_new_tokens=24, no_repeat_ngram_size=2)
    assert out.shape == (2, 24)
    for row in out.tolist():
        grams = list(zip(row[:-1], row[1:]))
        assert len(grams) == len(set(grams)), row


def test_session_reorder_gathers_the_cache_rows(tiny_meta):
    """Beam search moves beams between rows: after `reorder(perm)` row i must continue exactly as row perm[i] would have (KV cache,
    position ids, visible-key bounds).  Twin sessions on one batch, one of them reordered after the first step."""
    from molly_amd.generate import GenerationSession
    m = build_tiny(tiny_meta)
    ids, mask, omic, info = _left_padded_batch(tiny_meta)          # two rows with different valid lengths
    a, b = GenerationSession(m, 6), GenerationSession(m, 6)
    la, lb = a.prefill(ids, mask, omic, info), b.prefill(ids, mask, omic, info)
    assert torch.equal(la, lb)
    t1 = la.argmax(-1)
    la, lb = a.step(t1).clone(), b.step(t1).clone()
    perm = torch.tensor([1, 0], device=la.device)
    a.reorder(perm)
    t2 = lb.argmax(-1)
    for _ in range(3):                                             # through the eager step, the capture and a replay
        la, lb = a.step(t2[perm]).clone(), b.step(t2).clone()
        assert torch.allclose(la, lb[perm], rtol=0, atol=2e-2 * float(lb.abs().max())), float((la - lb[perm]).abs().max())
        t2 = lb.argmax(-1)


def test_generate_beam_search_through_the_reference_signature(tiny_meta):
    """The reference's generate accepts `num_beams` and drops it (src/model/omics_one.py:199, 220-232): so does this one — `num_beams=3`
    returns the greedy continuation, with a warning.  Beam search is this build's extension behind `molly_num_beams`; the procedure is
    pinned to HuggingFace's beam search token for token on CPU (tests/test_beam_search.py), and here it runs on the HIP decode session:
    deterministic, the best hypothesis scores at least as high under the model as the greedy continuation, one beam equals greedy,
    beam sampling runs and is reproducible under a seeded generator."""
    import warnings
    m = build_tiny(tiny_meta)
    ids, mask, omic, info = _left_padded_batch(tiny_meta)
    greedy = m.generate(ids, mask, omic, info, do_sample=False, max_new_tokens=6)
    type(m)._warned_num_beams = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ignored = m.generate(ids, mask, omic, info, do_sample=False, max_new_tokens=6, num_beams=3)
    assert torch.equal(greedy, ignored) and any("num_beams" in str(x.message) for x in w)
    one = m.generate(ids, mask, omic, info, do_sample=False, max_new_tokens=6, molly_num_beams=1)
    assert torch.equal(greedy, one)
    beams = m.generate(ids, mask, omic, info, do_sample=False, max_new_tokens=6, molly_num_beams=3)
    assert beams.shape == (2, 6) and beams.dtype == torch.int64
    assert torch.equal(beams, m.generate(ids, mask, omic, info, do_sample=False, max_new_tokens=6, molly_num_beams=3))

    def seq_logprob(tokens):                                       # sum of log-probabilities of `tokens` under teacher forcing
        from molly_amd.generate import GenerationSession
        s = GenerationSession(m, 6)
        lg = s.prefill(ids, mask, omic, info)
        tot = torch.zeros(2, device=lg.device)
        for t in range(tokens.shape[1]):
            tot += torch.log_softmax(lg.float(), -1).gather(1, tokens[:, t:t + 1].to(lg.device)).squeeze(1)
            if t + 1 < tokens.shape[1]:
                lg = s.step(tokens[:, t].to(lg.device))
        return tot
    assert bool((seq_logprob(beams) >= seq_logprob(greedy) - 5e-2).all())
    # beam sampling: the same procedure with drawn continuations; reproducible under the caller's generator
    g = torch.Generator(device="cuda").manual_seed(5)
    s1 = m.generate(ids, mask, omic, info, molly_num_beams=3, do_sample=True, temperature=0.8, top_k=20, top_p=0.95, max_new_tokens=6, generator=g)
    g = torch.Generator(device="cuda").manual_seed(5)
    s2 = m.generate(ids, mask, omic, info, molly_num_beams=3, do_sample=True, temperature=0.8, top_k=20, top_p=0.95, max_new_tokens=6, generator=g)
    assert s1.shape[0] == 2 and s1.shape[1] <= 6 and torch.equal(s1, s2)


def test_generate_kwargs_are_honoured_or_refused_never_dropped(tiny_meta):
    """reference src/model/omics_one.py:200-204, 220-232: every extra keyword goes to HuggingFace's generate, and under world > 1 the
    reference sets use_cache=False.  Here: use_cache (either value: the KV cache is this rank's own memory, same tokens), min_new_tokens
    (HF's MinNewTokensLengthLogitsProcessor) and the sampling extras are honoured; anything this build does not implement raises instead of
    silently changing the output."""
    m = build_tiny(tiny_meta)
    ids, mask, omic, info = _left_padded_batch(tiny_meta)
    base = m.generate(ids, mask, omic, info, do_sample=False, max_new_tokens=6)
    assert torch.equal(base, m.generate(ids, mask, omic, info, do_sample=False, max_new_tokens=6, use_cache=False))
    assert torch.equal(base, m.generate(ids, mask, omic, info, do_sample=False, max_new_tokens=6, use_cache=True))
    # min_new_tokens: make the first greedy token of sample 0 the EOS — without it the row stops at once, with it EOS cannot be chosen before
    # three new tokens exist
    m.text_config.eos_token_id = int(base[0, 0])
    m.text_config.pad_token_id = 1000
    stopped = m.generate(ids, mask, omic, info, do_sample=False, max_new_tokens=6)
    assert bool((stopped[0, 1:] == 1000).all())
    kept = m.generate(ids, mask, omic, info, do_sample=False, max_new_tokens=6, min_new_tokens=3)
    assert not bool((kept[0, :3] == int(base[0, 0])).any()) and kept.shape[1] >= 3
    for bad in (dict(stopping_criteria=[lambda *a: True]), dict(bad_words_ids=[[5]]), dict(num_return_sequences=2)):
        with pytest.raises(NotImplementedError, match=next(iter(bad))):
            m.generate(ids, mask, omic, info, do_sample=False, max_new_tokens=2, **bad)
    with pytest.raises(TypeError, match="eos_token_id"):
        m.generate(ids, mask, omic, info, do_sample=False, max_new_tokens=2, eos_token_id=3)
