"""Beam search (and beam sampling): an EXTENSION of this build, reached through `OmicsOne.generate(molly_num_beams=N)`.

The reference's `OmicsOne.generate` has a `num_beams` parameter but never forwards it (src/model/omics_one.py:199, 220-232: the call of
`self.model.generate` passes do_sample / temperature / top_p / top_k / no_repeat_ngram_size / pad / eos / **generate_kwargs), so beam
search is not reachable through the reference's own API; molly_amd's `num_beams` is ignored in the same way.  What this module
restates is HuggingFace's `GenerationMixin._beam_search` (HF:generation/utils.py) as it runs when `generate` is given
`inputs_embeds` only — an EMPTY `input_ids`, so the decoder prompt length is 0 and every length in the scorer counts generated tokens
only — over two callbacks, "logits of the next position for these B * num_beams rows" and "reorder the rows' KV cache", so that the
same code runs on the HIP decode session (molly_amd/generate.py) and, in the CPU test, on HuggingFace's own model, where its output is
compared token for token with `hf_model.generate(num_beams=...)` (tests/test_beam_search.py).

Procedure (one step, per batch row):
  1. log-softmax of the logits (fp32), logits processors applied to the LOG-PROBABILITIES (HF's order in beam mode), plus the
     running score of the beam the row belongs to; the first step starts with beam 0 at 0 and the others at -1e9, so that the
     identical prompts do not produce num_beams copies of one continuation;
  2. the K = max(2, 1 + #eos) * num_beams best (beam, token) continuations over the row's num_beams * V candidates;
  3. a continuation "stops" when its token is an EOS or the length limit is reached; the num_beams best NON-stopping ones continue;
  4. stopping continuations that rank among the first num_beams enter the row's finished list with score / length ** length_penalty
     (unless the row's list is closed: full with early_stopping=True, or the heuristic of step 6 said nothing can improve it); the
     list keeps its num_beams best;
  5. the KV cache rows are gathered to the continuing beams;
  6. heuristic: the best running score / (current | maximal) length ** length_penalty must still beat the worst finished score,
     else the row is closed; the loop ends when every row is closed, or every row is full (early_stopping=True), or nothing continues.
The best finished hypothesis of every row is returned, padded to the longest."""
from __future__ import annotations

from typing import Callable, Optional, Sequence

import torch

NEG = -1.0e9


def _gather(t: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """rows `idx` [B, k] of the beam dimension (1) of t [B, n, ...]."""
    while idx.dim() < t.dim():
        idx = idx.unsqueeze(-1)
    return torch.take_along_dim(t, idx, dim=1)


@torch.no_grad()
def beam_search(first_logits: torch.Tensor, step: Callable[[torch.Tensor], torch.Tensor], reorder: Callable[[torch.Tensor], None],
                batch: int, num_beams: int, max_new_tokens: int, eos_token_id: Optional[Sequence[int]] = None,
                pad_token_id: Optional[int] = None, length_penalty: float = 1.0, early_stopping=False,
                process_log_probs: Optional[Callable[[torch.Tensor, torch.Tensor], torch.Tensor]] = None, do_sample: bool = False,
                generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """first_logits: [batch * num_beams, V] logits of the first new position (rows b * num_beams + k all hold row b's prompt).
    step(tokens [batch * num_beams] int64) -> logits of the next position; reorder(row_idx [batch * num_beams] int64): row i of the
    cache becomes what row row_idx[i] was.  process_log_probs(generated [batch * num_beams, t], log_probs) -> log_probs: the logits
    processors (repetition penalty, n-gram ban; with do_sample the warpers: temperature, top-k, top-p), applied as HF applies them in
    beam mode.  do_sample: beam SAMPLING — the K continuations of step 2 are drawn without replacement from softmax(accumulated
    scores) instead of being the K best (HF `_get_top_k_continuations`); everything else is unchanged.  Returns [batch, n_new] int64."""
    dev = first_logits.device
    B, nb, L = batch, num_beams, max_new_tokens
    V = first_logits.shape[-1]
    eos = torch.as_tensor(list(eos_token_id), device=dev, dtype=torch.int64) if eos_token_id is not None and len(eos_token_id) else None
    n_eos = 0 if eos is None else int(eos.numel())
    K = max(2, 1 + n_eos) * nb
    fill = pad_token_id if pad_token_id is not None else (int(eos[0]) if eos is not None else -1)      # HF falls back to eos[0] only when pad is None (0 is a valid pad id)
    in_top = torch.zeros(K, dtype=torch.bool, device=dev)
    in_top[:nb] = True

    run_seq = torch.full((B, nb, L), fill, dtype=torch.int64, device=dev)
    fin_seq = run_seq.clone()
    run_score = torch.zeros(B, nb, dtype=torch.float32, device=dev)
    run_score[:, 1:] = NEG
    fin_score = torch.full((B, nb), NEG, dtype=torch.float32, device=dev)
    fin_len = torch.zeros(B, nb, dtype=torch.int64, device=dev)
    fin_flag = torch.zeros(B, nb, dtype=torch.bool, device=dev)
    open_row = torch.ones(B, 1, dtype=torch.bool, device=dev)                  # the heuristic has not closed the row
    offs = (torch.arange(B, device=dev) * nb)[:, None]

    logits = first_logits
    cur = 0
    while True:
        lp = torch.log_softmax(logits.float(), dim=-1)
        if process_log_probs is not None:
            lp = process_log_probs(run_seq.reshape(B * nb, L)[:, :cur], lp)
        acc = (lp.view(B, nb, V) + run_score[:, :, None]).reshape(B, nb * V)
        if do_sample:
            top_idx = torch.multinomial(torch.softmax(acc, dim=-1), K, generator=generator)
            top_score = torch.gather(acc, 1, top_idx)
        else:
            top_score, top_idx = torch.topk(acc, K)
        src_beam, tok = top_idx // V, top_idx % V
        cand_seq = _gather(run_seq, src_beam)
        cand_seq[:, :, cur] = tok
        stops = torch.full_like(tok, cur + 1 >= L, dtype=torch.bool)
        if eos is not None:
            stops = stops | torch.isin(tok, eos)
        # the beams that go on
        go_score = top_score + stops.float() * NEG
        go_idx = torch.topk(go_score, nb)[1]
        run_seq, run_score = _gather(cand_seq, go_idx), _gather(go_score, go_idx)
        go_beam = _gather(src_beam, go_idx)
        # the finished list
        just = stops & in_top[None, :]
        s = top_score / float((cur + 1) ** length_penalty)
        s = s + (fin_flag.all(-1, keepdim=True) & (early_stopping is True)).float() * NEG
        s = s + (~open_row).float() * NEG
        s = s + (~just).float() * NEG
        m_seq = torch.cat((fin_seq, cand_seq), 1)
        m_score = torch.cat((fin_score, s), 1)
        m_len = torch.cat((fin_len, torch.full_like(tok, cur + 1)), 1)
        m_flag = torch.cat((fin_flag, just), 1)
        keep = torch.topk(m_score, nb)[1]
        fin_seq, fin_score, fin_len, fin_flag = _gather(m_seq, keep), _gather(m_score, keep), _gather(m_len, keep), _gather(m_flag, keep)

        cur += 1
        hyp_len = L if (early_stopping == "never" and length_penalty > 0.0) else cur
        best_running = run_score[:, :1] / float(hyp_len ** length_penalty)
        worst_fin = torch.where(fin_flag, fin_score.min(1, keepdim=True)[0], torch.full_like(fin_score, NEG))
        open_row = open_row & (best_running > worst_fin).any(-1, keepdim=True)
        more = bool(open_row.any()) and not (bool(fin_flag.all()) and early_stopping is True) and not bool(stops.all())
        if not more:
            break
        rows = (go_beam + offs).reshape(-1)
        reorder(rows)
        logits = step(run_seq[:, :, cur - 1].reshape(-1))
    n_out = int(fin_len[:, 0].max())
    return fin_seq[:, 0, :n_out]
