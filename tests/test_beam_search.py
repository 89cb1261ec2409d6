"""CPU: molly_amd.beam.beam_search against HuggingFace's own beam search.  Beam search is an extension of this build
(`OmicsOne.generate(molly_num_beams=N)`): the reference's generate has a `num_beams` parameter but does not forward it
(src/model/omics_one.py:199, 220-232).  What is pinned here is the procedure HF runs for `generate(inputs_embeds=..., num_beams=k)` with
no `input_ids`: a small random HF Qwen3 model provides the logits for BOTH sides — HF's own `generate` and molly's restatement driven
through two callbacks (next logits for these rows / gather the rows) — so every token must agree: the scorer, the early-stop
heuristic, the EOS bookkeeping and the padding of shorter hypotheses are pinned to HF's behaviour independently of any kernel.  (The
installed transformers is 5.x; the reference pins 4.53, whose BeamSearchScorer the 4.50 refactor reproduced.)"""
import pytest
import torch

from molly_amd.beam import beam_search


def _hf_model():
    from transformers import Qwen3Config, Qwen3ForCausalLM
    torch.manual_seed(0)
    cfg = Qwen3Config(vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, head_dim=8, max_position_embeddings=128, tie_word_embeddings=False)
    m = Qwen3ForCausalLM(cfg).eval()
    with torch.no_grad():
        for p in m.parameters():            # sharper logits: hypotheses differ in score, EOS competes
            p.mul_(3.0)
    return m


class _Provider:
    """next-position logits by a full re-forward of prompt embeddings + the tokens generated so far (no cache: the rows are the state)."""

    def __init__(self, m, emb, mask, nb):
        self.m, self.emb, self.mask = m, emb.repeat_interleave(nb, 0), mask.repeat_interleave(nb, 0)
        self.toks = torch.empty(self.emb.shape[0], 0, dtype=torch.long)

    def logits(self):
        e = torch.cat([self.emb, self.m.get_input_embeddings()(self.toks)], 1)
        msk = torch.cat([self.mask, torch.ones_like(self.toks)], 1)
        pos = (msk.cumsum(1) - 1).clamp(min=0)
        with torch.no_grad():
            return self.m(inputs_embeds=e, attention_mask=msk, position_ids=pos).logits[:, -1].float()

    def reorder(self, rows):
        self.toks = self.toks[rows]

    def step(self, tokens):
        self.toks = torch.cat([self.toks, tokens[:, None]], 1)
        return self.logits()


@pytest.mark.parametrize("nb,n_new,eos,lp,early", [(3, 10, None, 1.0, False), (4, 12, [5, 9], 1.0, False), (3, 12, [7], 2.0, False),
                                                   (2, 8, [3, 5, 7, 11, 13], 0.0, True), (4, 14, [5, 9, 20, 33], 1.0, "never"),
                                                   (3, 9, [5], 1.0, True)])
def test_tokens_equal_huggingface_generate(nb, n_new, eos, lp, early):
    m = _hf_model()
    B, T = 3, 6
    g = torch.Generator().manual_seed(1)
    emb = torch.randn(B, T, 32, generator=g) * 0.5
    mask = torch.ones(B, T, dtype=torch.long)
    mask[1, :2] = 0                                                   # a left-padded row
    with torch.no_grad():
        want = m.generate(inputs_embeds=emb, attention_mask=mask, num_beams=nb, do_sample=False, max_new_tokens=n_new, eos_token_id=eos,
                          pad_token_id=1, length_penalty=lp, early_stopping=early, use_cache=False)
    pr = _Provider(m, emb, mask, nb)
    got = beam_search(pr.logits(), pr.step, pr.reorder, B, nb, n_new, eos, 1, lp, early)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert torch.equal(got, want), (got.tolist(), want.tolist())


def test_logits_processors_follow_huggingface_in_beam_mode():
    """repetition penalty and the n-gram ban act on the LOG-PROBABILITIES of the running rows (HF's beam mode); the rows change beams
    every step, so the n-gram table is rebuilt from the row's tokens."""
    from molly_amd.generate import NoRepeatNGram, _process_logits
    m = _hf_model()
    B, T, nb, n_new = 2, 5, 3, 10
    g = torch.Generator().manual_seed(2)
    emb = torch.randn(B, T, 32, generator=g) * 0.5
    mask = torch.ones(B, T, dtype=torch.long)
    with torch.no_grad():
        want = m.generate(inputs_embeds=emb, attention_mask=mask, num_beams=nb, do_sample=False, max_new_tokens=n_new, eos_token_id=[9],
                          pad_token_id=1, repetition_penalty=1.3, no_repeat_ngram_size=2, use_cache=False)

    def proc(generated, lp):
        if generated.shape[1] > 0:
            lp = _process_logits(lp, generated, None, None, None, 1.3)
        if generated.shape[1] + 1 >= 2:
            ng = NoRepeatNGram(2, generated.shape[0])
            for t in range(generated.shape[1]):
                ng.push(generated[:, t].tolist())
            lp = ng.apply(lp)
        return lp
    pr = _Provider(m, emb, mask, nb)
    got = beam_search(pr.logits(), pr.step, pr.reorder, B, nb, n_new, [9], 1, 1.0, False, proc)
    assert torch.equal(got, want), (got.tolist(), want.tolist())


def test_beam_sampling_draws_equal_huggingface_under_one_seed():
    """do_sample=True with beams: HF draws the K continuations with torch.multinomial from softmax(accumulated scores) after the warpers
    acted on the log-probabilities.  On CPU, with the global generator seeded alike before each side, the same procedure consumes the
    same random numbers: token-for-token equality again."""
    from molly_amd.generate import _process_logits
    m = _hf_model()
    B, T, nb, n_new = 2, 5, 3, 9
    g = torch.Generator().manual_seed(3)
    emb = torch.randn(B, T, 32, generator=g) * 0.5
    mask = torch.ones(B, T, dtype=torch.long)
    torch.manual_seed(11)
    with torch.no_grad():
        want = m.generate(inputs_embeds=emb, attention_mask=mask, num_beams=nb, do_sample=True, temperature=0.8, top_k=20, top_p=0.95,
                          max_new_tokens=n_new, eos_token_id=[9], pad_token_id=1, use_cache=False)
    pr = _Provider(m, emb, mask, nb)
    first = pr.logits()
    torch.manual_seed(11)
    got = beam_search(first, pr.step, pr.reorder, B, nb, n_new, [9], 1, 1.0, False,
                      lambda gen, lp: _process_logits(lp, gen, 0.8, 20, 0.95, None), do_sample=True)
    assert torch.equal(got, want), (got.tolist(), want.tolist())
