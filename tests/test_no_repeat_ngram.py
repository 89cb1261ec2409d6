"""CPU: `no_repeat_ngram_size` of OmicsOne.generate (reference src/model/omics_one.py:199-200, 227 passes it to HF generate) — the
incremental n-gram table of molly_amd.generate.NoRepeatNGram bans exactly what HuggingFace's own NoRepeatNGramLogitsProcessor bans,
step by step, on random token streams with many repeats."""
import pytest
import torch

from molly_amd.generate import NoRepeatNGram


@pytest.mark.parametrize("n", [1, 2, 3, 4])
def test_bans_equal_huggingface_processor(n):
    from transformers.generation.logits_process import NoRepeatNGramLogitsProcessor
    B, V, steps = 3, 7, 40                       # a 7-token vocabulary repeats n-grams constantly
    g = torch.Generator().manual_seed(n)
    hf = NoRepeatNGramLogitsProcessor(n)
    mine = NoRepeatNGram(n, B)
    seq = torch.empty(B, 0, dtype=torch.long)
    for _ in range(steps):
        scores = torch.randn(B, V, generator=g)
        want = hf(seq, scores.clone())           # HF sees the generated tokens only (generate() with inputs_embeds)
        got = mine.apply(scores.clone())
        assert torch.equal(torch.isinf(want), torch.isinf(got)), (n, seq.tolist())
        assert torch.equal(torch.nan_to_num(want, neginf=-1e9), torch.nan_to_num(got, neginf=-1e9))
        nxt = torch.randint(0, V, (B,), generator=g)
        seq = torch.cat([seq, nxt[:, None]], 1)
        mine.push(nxt.tolist())
