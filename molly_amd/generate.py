"""Prefill + KV-cache decode for `OmicsOne.generate` (reference: src/model/omics_one.py:187-233 -> HF `generate` with
`inputs_embeds`, DynamicCache, sampling processors).  Prefill reuses the training forward kernels (flash attention over the
left-padded prompt, positions = cumsum(mask)-1 as HF does: HF:generation/utils.py:751-773) and writes every layer's
post-RoPE K and V into a token-major cache; each decode step is M = batch GEMMs + one `molly_attn_decode` per layer.
Token selection (temperature / top-k / top-p / repetition penalty, HF logits processors) runs on the [B, V] logits with
torch ops — a few KB of work per step, not a kernel target.
"""
from __future__ import annotations

from typing import Optional

import os

import torch

from . import ops

BF16 = torch.bfloat16


class GenerationSession:
    def __init__(self, model, max_new_tokens: int, use_graph: bool = None):
        import os
        self.use_graph = (os.environ.get("MOLLY_DECODE_GRAPH", "1") != "0") if use_graph is None else use_graph
        self.m = model
        self.rt = model._runtime()
        self.eng = self.rt.llm
        self.max_new = max_new_tokens

    # ---- prefill -------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def prefill(self, input_ids, attention_mask, omic_ids, omic_info_list) -> torch.Tensor:
        """Returns the logits of the last prompt position [B, V] (fp32)."""
        with ops.use_gemm_context(self.rt.gemm_ctx):
            return self._prefill(input_ids, attention_mask, omic_ids, omic_info_list)

    def _prefill(self, input_ids, attention_mask, omic_ids, omic_info_list) -> torch.Tensor:
        m, rt, e = self.m, self.rt, self.eng
        B, T = input_ids.shape
        dev = rt.dev
        self.B, self.T = B, T
        self.Tmax = T + self.max_new
        lo, hi = m._kv_range(attention_mask, B, T, dev)
        mask = attention_mask.cpu().long() if attention_mask is not None else torch.ones(B, T, dtype=torch.long)
        pos = (mask.cumsum(1) - 1).clamp(min=0)                              # HF: position_ids = cumsum(mask)-1, pads -> 0/1
        self.lo = lo if lo is not None else torch.zeros(B, dtype=torch.int32, device=dev)
        self.n_valid = mask.sum(1).to(torch.int32).to(dev)                    # next position id per sample
        # the session's GEMM scratch is sized BEFORE its first GEMM (the launcher's split-K choices look at how much there is: a
        # workspace that grew between the prefill and a later one of the same shape could cut the second differently)
        ops.ensure_gemm_workspace(32 * 4 * max(B, 8) * max(2 * e.ff, e.nqkv), dev)
        hs, _ = m._embed_and_inject(m._stage(input_ids, None, None, omic_ids, omic_info_list, want_sort=False), False)
        e.reserve(B * T, B, T, training=False)
        # rope tables must cover prompt + generated positions
        from .qwen3 import rope_tables
        self.cos, self.sin = rope_tables(self.Tmax, e.hd, e.cfg.rope_theta, dev, e.rope_table_dtype)
        positions = pos.reshape(-1).to(torch.int32).to(dev)
        L, nkvd = e.L, e.nkv * e.hd
        self.kc = torch.zeros(L, B, self.Tmax, nkvd, dtype=BF16, device=dev)
        self.vc = torch.zeros(L, B, self.Tmax, nkvd, dtype=BF16, device=dev)
        rows = (torch.arange(B)[:, None] * self.Tmax + torch.arange(T)[None, :]).reshape(-1).to(torch.int32).to(dev)
        a = e.A[0]
        x = hs
        nq = e.nh * e.hd
        nk_ = e.nkv * e.hd
        if e.lora is not None:
            self.lt = torch.empty(B * T, e.lora.rp, dtype=BF16, device=dev)
        for i in range(L):
            w = e.W[i]
            ops.rmsnorm_fwd(x, w["ln1"], e.cfg.rms_norm_eps, out=a["xn"])
            ops.gemm_nt(a["xn"], w["qkv"], out=a["qkv"])
            self._lora(i, "q_proj", a["xn"], a["qkv"][:, :nq])
            self._lora(i, "k_proj", a["xn"], a["qkv"][:, nq:nq + nk_])
            self._lora(i, "v_proj", a["xn"], a["qkv"][:, nq + nk_:])
            # q/k-norm + rotary + the cache write of the whole prompt (row b * Tmax + t) in one launch
            ops.norm_rope_fwd(a["qkv"], a["qk"], e.nh, e.nkv, e.hd, T, w["qn"], w["kn"], self.cos, self.sin,
                              positions=positions, eps=e.cfg.rms_norm_eps,
                              kcache=self.kc[i].view(B * self.Tmax, nkvd), vcache=self.vc[i].view(B * self.Tmax, nkvd), slot=rows)
            ops.attn_fwd(a["qk"][:, :nq], a["qk"][:, nq:], a["qkv"][:, e.nqk:], B, T, e.nh, e.nkv, e.hd, e.hd ** -0.5, True,
                         lo, hi, out=a["attn"], lse=False)
            ops.gemm_nt(a["attn"], w["o"], out=a["x2"], res=x)
            self._lora(i, "o_proj", a["attn"], a["x2"])
            ops.rmsnorm_fwd(a["x2"], w["ln2"], e.cfg.rms_norm_eps, out=a["xn2"])
            if e.fused_swiglu:                                              # (no live adapters) SwiGLU in the gate|up GEMM's epilogue, as in training
                ops.gemm_gate_up_swiglu(a["xn2"], w["gu"], a["gu"], a["act"])
            else:
                ops.gemm_nt(a["xn2"], w["gu"], out=a["gu"])
                self._lora(i, "gate_proj", a["xn2"], a["gu"][:, :e.ff])
                self._lora(i, "up_proj", a["xn2"], a["gu"][:, e.ff:])
                ops.swiglu_fwd(a["gu"], out=a["act"])
            ops.gemm_nt(a["act"], w["down"], out=e.x_out, res=a["x2"])
            self._lora(i, "down_proj", a["act"], e.x_out)
            x = e.x_out
        last = x.view(B, T, e.h)[:, T - 1, :].contiguous()                    # left-padded prompts end at T-1
        self.cur_len = T
        self._alloc_step(B)
        return self._head(last)

    def _lora(self, i, mod, x, y):
        """live (un-merged) adapter: y += s * (x A^T) B^T, no dropout at inference (PEFT eval mode)."""
        lo = self.eng.lora
        if lo is None:
            return
        t = self.lt[:x.shape[0]]
        ops.gemm_nt(x, lo.A[i][mod], out=t)
        if lo.scale != 1.0:
            ops.scale_(t, lo.scale)
        ops.gemm_nt(t, lo.B[i][mod], out=y, accumulate=True)

    def _alloc_step(self, B):
        e, dev = self.eng, self.rt.dev
        z = lambda *s: torch.empty(*s, dtype=BF16, device=dev)
        self.s = dict(x=z(B, e.h), xn=z(B, e.h), qkv=z(B, e.nqkv), qk=z(B, e.nqk), attn=z(B, e.nh * e.hd), x2=z(B, e.h),
                      xn2=z(B, e.h), gu=z(B, 2 * e.ff), act=z(B, e.ff), hn=z(B, e.h))
        # static per-step state (device): token ids, position ids, cache slot and visible-key bound of the step's token
        self.tok = torch.zeros(B, dtype=torch.int64, device=dev)
        # (three rows of ONE tensor: the step advances them with one launch)
        self.state = torch.empty(3, B, dtype=torch.int32, device=dev)
        self.pos, self.slot, self.hi = self.state[0], self.state[1], self.state[2]
        self.pos.copy_(self.n_valid)                                           # position id = number of valid tokens so far
        self.slot.copy_(torch.arange(B, device=dev, dtype=torch.int32) * self.Tmax + self.cur_len)
        self.hi.fill_(self.cur_len + 1)
        self.logits = torch.empty(B, e.V, dtype=torch.float32, device=dev)
        self._graph, self._graph_ws, self._graph_ws_ptr, self._steps_done = None, None, 0, 0
        self.dec_ws = ops.attn_decode_workspace(B, e.nh, e.hd, dev)
        # decode GEMMs are M = B rows against whole weight matrices: the library splits K over the chip and needs slab space
        ops.ensure_gemm_workspace(32 * 4 * max(B, 8) * max(2 * e.ff, e.nqkv), dev)

    def _head(self, x):
        e = self.eng
        hn = ops.rmsnorm_fwd(x, e.norm_w, e.cfg.rms_norm_eps)
        return ops.gemm_nt(hn, e.head, out_dtype=torch.float32)

    # ---- one decode step ------------------------------------------------------------------------------------------
    def _step_body(self) -> torch.Tensor:
        """All device work of one step, reading/writing only the session's static tensors (tok, pos, slot, hi, caches), so
        the launch sequence is identical every step and can be replayed from a hipGraph."""
        e, s, B = self.eng, self.s, self.B
        nq, nkvd = e.nh * e.hd, e.nkv * e.hd
        ops.copy_rows(e.embed, s["x"], B, src_idx64=self.tok)
        x = s["x"]
        eps = e.cfg.rms_norm_eps
        # merged or no adapters: the launches behind a projection fold into it where the decode-row GEMM takes the shape (batch 17..64 on
        # the larger matrices) — residual + next RMSNorm behind o / down, SwiGLU behind gate|up — and the cache appends into norm + rope
        fuse = e.lora is None and os.environ.get("MOLLY_DECODE_FUSE", "1") != "0"
        H = x.shape[1]
        f_qkv = fuse and ops.gemm_rows_tail_supported(B, nq + 2 * nkvd, H, "qkv")
        # (the fused attention is built for groups of 1 / 2 / 4 / 8 query heads per KV head: other ratios take the plain path)
        f_att = (f_qkv and e.hd in (64, 128) and e.nh % e.nkv == 0 and e.nh // e.nkv in (1, 2, 4, 8)
                 and os.environ.get("MOLLY_DECODE_FUSE_ATTN", "1") != "0")
        f_o = fuse and ops.gemm_rows_tail_supported(B, H, nq, "norm")
        f_gu = fuse and ops.gemm_rows_tail_supported(B, 2 * e.ff, H, "swiglu")
        f_dn = fuse and ops.gemm_rows_tail_supported(B, H, e.ff, "norm")
        normed = False                                  # s["xn"] already holds ln1(x) of the layer about to run
        for i in range(e.L):
            w = e.W[i]
            if not normed:
                ops.rmsnorm_fwd(x, w["ln1"], eps, out=s["xn"])
            kc_i, vc_i = self.kc[i].view(B * self.Tmax, nkvd), self.vc[i].view(B * self.Tmax, nkvd)
            if f_att:                                   # projection as K-slice slabs; everything up to the attention output in ONE launch
                slabs, n_slabs = ops.gemm_rows_slabs(s["xn"], w["qkv"])
                ops.attn_decode_qkv(slabs, n_slabs, w["qn"], w["kn"], self.cos, self.sin, self.pos, eps, self.kc[i], self.vc[i], self.slot,
                                    s["attn"], self.lo, self.hi, B, self.Tmax, e.nh, e.nkv, e.hd, e.hd ** -0.5, kv_len_hint=self.Tmax,
                                    workspace=self.dec_ws)
            elif f_qkv:                                 # projection + q/k-norm + rotary + cache append: GEMM and one tail launch
                ops.gemm_rows_qkv(s["xn"], w["qkv"], s["qk"], e.nh, e.nkv, e.hd, w["qn"], w["kn"], self.cos, self.sin, self.pos, eps,
                                  kc_i, vc_i, self.slot)
            else:
                ops.gemm_nt(s["xn"], w["qkv"], out=s["qkv"])
                self._lora(i, "q_proj", s["xn"], s["qkv"][:, :nq])
                self._lora(i, "k_proj", s["xn"], s["qkv"][:, nq:nq + nkvd])
                self._lora(i, "v_proj", s["xn"], s["qkv"][:, nq + nkvd:])
                if fuse:
                    ops.norm_rope_fwd(s["qkv"], s["qk"], e.nh, e.nkv, e.hd, 1, w["qn"], w["kn"], self.cos, self.sin, positions=self.pos,
                                      eps=eps, kcache=kc_i, vcache=vc_i, slot=self.slot)
                else:
                    ops.norm_rope_fwd(s["qkv"], s["qk"], e.nh, e.nkv, e.hd, 1, w["qn"], w["kn"], self.cos, self.sin,
                                      positions=self.pos, eps=eps)
                    ops.copy_rows(s["qk"][:, nq:], kc_i, B, dst_idx32=self.slot)
                    ops.copy_rows(s["qkv"][:, e.nqk:], vc_i, B, dst_idx32=self.slot)
            if not f_att:
                ops.attn_decode(s["qk"], self.kc[i], self.vc[i], s["attn"], self.lo, self.hi, B, self.Tmax, e.nh, e.nkv, e.hd,
                                e.hd ** -0.5, kv_len_hint=self.Tmax, workspace=self.dec_ws)
            if f_o:
                ops.gemm_rows_norm(s["attn"], w["o"], s["x2"], w["ln2"], eps, s["xn2"], res=x)
            else:
                ops.gemm_nt(s["attn"], w["o"], out=s["x2"], res=x)
                self._lora(i, "o_proj", s["attn"], s["x2"])
                ops.rmsnorm_fwd(s["x2"], w["ln2"], eps, out=s["xn2"])
            if f_gu:
                ops.gemm_rows_swiglu(s["xn2"], w["gu"], None, s["act"])          # (gate | up itself is not needed again)
            else:
                ops.gemm_nt(s["xn2"], w["gu"], out=s["gu"])
                self._lora(i, "gate_proj", s["xn2"], s["gu"][:, :e.ff])
                self._lora(i, "up_proj", s["xn2"], s["gu"][:, e.ff:])
                ops.swiglu_fwd(s["gu"], out=s["act"])
            if f_dn:                                    # ... + the NEXT block's input norm (the final norm behind the last block)
                last = i + 1 == e.L
                ops.gemm_rows_norm(s["act"], w["down"], s["x"], e.norm_w if last else e.W[i + 1]["ln1"], eps,
                                   s["hn"] if last else s["xn"], res=s["x2"])
                normed = True
            else:
                ops.gemm_nt(s["act"], w["down"], out=s["x"], res=s["x2"])
                self._lora(i, "down_proj", s["act"], s["x"])
            x = s["x"]
        if not normed:
            ops.rmsnorm_fwd(x, e.norm_w, eps, out=s["hn"])
        ops.gemm_nt(s["hn"], e.head, out=self.logits)
        # advance the per-sample position / cache slot / visible-key bound for the next step (device-side, graph-replayable)
        self.state += 1
        return self.logits

    @torch.no_grad()
    def reorder(self, rows: torch.Tensor):
        """Row i of the session becomes what row rows[i] was (beam search: the cache follows the beams that continue — HF's
        `DynamicCache.reorder_cache`, HF:cache_utils.py).  In place, so the captured decode graph's pointers stay valid; only the
        filled part of the caches moves."""
        rows = rows.to(self.rt.dev).long()
        if bool((rows == torch.arange(rows.numel(), device=rows.device)).all()):
            return                                      # every beam stays where it is: nothing moves
        n = self.cur_len
        # layer by layer: the temporary is one layer's filled cache (B x n x kv width), not all L of them — at Qwen3-8B with a 3 k-token
        # prompt and B * num_beams rows the whole-cache gather was a multi-GB allocation beside a cache sized for 3,072 new tokens
        for i in range(self.kc.shape[0]):
            self.kc[i, :, :n] = self.kc[i, :, :n].index_select(0, rows)
            self.vc[i, :, :n] = self.vc[i, :, :n].index_select(0, rows)
        for t in (self.pos, self.hi, self.lo, self.n_valid):
            t.copy_(t.index_select(0, rows))
        # (slot = row * Tmax + cur_len belongs to the ROW, not to the beam that moved into it)

    @torch.no_grad()
    def step(self, token_ids: torch.Tensor) -> torch.Tensor:
        """token_ids int64 [B] (the tokens chosen from the previous logits).  Returns next-token logits [B, V] fp32 (a
        buffer that the next step overwrites).  The first step runs eagerly (it also sizes workspaces and sets kernel
        attributes), the second is captured into a hipGraph, later steps replay it: the step's launches (7 per layer with merged adapters: round 4) become one."""
        assert self.cur_len < self.Tmax, "generation budget exhausted"
        with ops.use_gemm_context(self.rt.gemm_ctx):
            return self._step(token_ids)

    def _step(self, token_ids: torch.Tensor) -> torch.Tensor:
        self.tok.copy_(token_ids.to(self.rt.dev), non_blocking=True)
        self.cur_len += 1
        self._steps_done += 1
        if not self.use_graph or self._steps_done == 1:
            return self._step_body()
        # The captured launches carry the context's scratch pointers (split-K slabs, the decode-row tails' slabs, stream-K and
        # dynamic-fetch counters).  Another session or a training reserve() on the same model may have grown — i.e. replaced — that
        # scratch since the capture: the graph then keeps its own reference to the tensor it was captured with (`_graph_ws`, so a
        # replay in flight never reads freed memory) and is dropped and captured again before the next replay.
        ws = self.rt.gemm_ctx.ws
        if self._graph is not None and (ws is None or ws.data_ptr() != self._graph_ws_ptr):
            self._graph = None
        if self._graph is None:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._step_body()
            assert self.rt.gemm_ctx.ws is ws, "the GEMM scratch was replaced during graph capture"
            self._graph, self._graph_ws, self._graph_ws_ptr = g, ws, (ws.data_ptr() if ws is not None else 0)
            # capture does not execute: fall through to the replay
        self._graph.replay()
        return self.logits


def _process_logits(logits, generated, temperature, top_k, top_p, repetition_penalty):
    """HF logits processors in HF's order: repetition penalty, temperature, top-k, top-p."""
    if repetition_penalty and repetition_penalty != 1.0 and generated.shape[1] > 0:
        score = torch.gather(logits, 1, generated)
        score = torch.where(score < 0, score * repetition_penalty, score / repetition_penalty)
        logits = logits.scatter(1, generated, score)
    if temperature and temperature != 1.0:
        logits = logits / temperature
    if top_k:
        kth = torch.topk(logits, min(top_k, logits.shape[-1]))[0][..., -1, None]
        logits = logits.masked_fill(logits < kth, float("-inf"))
    if top_p is not None and top_p < 1.0:
        sl, si = torch.sort(logits, descending=False)
        cp = sl.softmax(-1).cumsum(-1)
        remove = cp <= (1 - top_p)
        remove[..., -1:] = False
        logits = logits.masked_fill(remove.scatter(1, si, remove), float("-inf"))
    return logits


class NoRepeatNGram:
    """HF `NoRepeatNGramLogitsProcessor` (HF:generation/logits_process.py, `_calc_banned_ngram_tokens`): a token is banned when it
    would complete an n-gram that already occurs in the row's generated sequence.  HF rebuilds the n-gram table of every row at every
    step; here it is kept per row and extended by the one n-gram each new token completes.  The reference passes
    `no_repeat_ngram_size` through to HF generate (src/model/omics_one.py:199-200, 227); with `inputs_embeds` and no `input_ids` HF's
    `input_ids` hold the GENERATED tokens only, so the prompt never contributes n-grams."""

    def __init__(self, n: int, batch: int):
        assert n >= 1
        self.n = n
        self.table = [dict() for _ in range(batch)]            # row -> {(n-1)-gram prefix: set of tokens that followed it}
        self.hist = [[] for _ in range(batch)]

    def push(self, tokens):
        """tokens: the B tokens just appended (python ints)."""
        n = self.n
        for b, t in enumerate(tokens):
            h = self.hist[b]
            h.append(int(t))
            if len(h) >= n:
                self.table[b].setdefault(tuple(h[len(h) - n:len(h) - 1]), set()).add(h[-1])

    def banned(self):
        """per row: the tokens that must not come next (HF: none while fewer than n - 1... tokens exist: cur_len + 1 < n)."""
        n, out = self.n, []
        for b, h in enumerate(self.hist):
            if len(h) + 1 < n:
                out.append(())
            else:
                out.append(tuple(self.table[b].get(tuple(h[len(h) + 1 - n:]), ())))
        return out

    def apply(self, logits: torch.Tensor) -> torch.Tensor:
        rows, cols = [], []
        for b, toks in enumerate(self.banned()):
            rows += [b] * len(toks)
            cols += list(toks)
        if rows:
            logits = logits.clone()
            logits[torch.tensor(rows, device=logits.device), torch.tensor(cols, device=logits.device)] = float("-inf")
        return logits


@torch.no_grad()
def generate(model, input_ids, attention_mask=None, omic_ids=None, omic_info_list=None, max_new_tokens=3072, do_sample=True,
             temperature=0.8, top_p=0.95, top_k=None, repetition_penalty=None, pad_token_id=None, eos_token_id=None,
             generator: Optional[torch.Generator] = None, no_repeat_ngram_size: Optional[int] = None, num_beams: int = 1,
             length_penalty: float = 1.0, early_stopping=False, min_new_tokens: int = 0) -> torch.Tensor:
    """Returns the NEW tokens only, int64 [B, n_new] (what HF returns when called with inputs_embeds and no input_ids).
    min_new_tokens: HF's MinNewTokensLengthLogitsProcessor — the eos scores are -inf while fewer new tokens than that exist (it runs among
    the processors, in front of the warpers)."""
    if num_beams and num_beams > 1:
        if min_new_tokens:
            raise NotImplementedError("generate: min_new_tokens together with molly_num_beams > 1 is not built")
        return _generate_beams(model, input_ids, attention_mask, omic_ids, omic_info_list, max_new_tokens, do_sample, repetition_penalty,
                               pad_token_id, eos_token_id, no_repeat_ngram_size, int(num_beams), length_penalty, early_stopping,
                               temperature, top_k, top_p, generator)
    sess = GenerationSession(model, max_new_tokens)
    logits = sess.prefill(input_ids, attention_mask, omic_ids, omic_info_list)
    B = input_ids.shape[0]
    dev = logits.device
    eos = None if eos_token_id is None else torch.as_tensor(eos_token_id, device=dev).reshape(-1)
    pad = pad_token_id if pad_token_id is not None else (int(eos[0]) if eos is not None else 0)
    out = torch.empty(B, 0, dtype=torch.int64, device=dev)
    unfinished = torch.ones(B, dtype=torch.bool, device=dev)
    # sampling with a top-k (the reference's inference settings: top_k 20) runs in ONE kernel per step: penalty, temperature,
    # top-k, top-p, softmax and the draw (csrc/sampling.hip).  Without a top-k the torch restatement below is used.
    fused = bool(do_sample and top_k and 1 <= top_k <= 1024 and logits.dtype == torch.float32)
    # The Philox stream is keyed by (seed, step, row).  The seed is DRAWN from the generator (or torch's global one), so it is
    # reproducible under manual_seed and — like torch.multinomial in the reference path — advances the generator's state:
    # two calls with one generator (inference.py passes one for the whole dataset) get independent samples.
    seed = 0
    if fused:
        gdev = generator.device if generator is not None else "cpu"
        seed = int(torch.randint(0, 1 << 62, (1,), generator=generator, device=gdev, dtype=torch.int64).item())
    ngram = NoRepeatNGram(int(no_repeat_ngram_size), B) if no_repeat_ngram_size else None
    for it in range(max_new_tokens):
        if ngram is not None:
            # HF's order: repetition penalty, then the n-gram ban, then the warpers; a banned logit is -inf and the penalty leaves -inf
            # where it is, so banning first gives the same scores
            logits = ngram.apply(logits)
        if min_new_tokens and it < min_new_tokens and eos is not None:
            logits = logits.clone()
            logits[:, eos] = float("-inf")
        if fused:
            nxt = ops.sample_logits(logits, out if out.shape[1] else None, repetition_penalty, temperature, top_k, top_p,
                                    seed, it)
        else:
            lg = _process_logits(logits, out, temperature if do_sample else None, top_k if do_sample else None,
                                 top_p if do_sample else None, repetition_penalty)
            if do_sample:
                nxt = torch.multinomial(lg.softmax(-1), 1, generator=generator).squeeze(1)
            else:
                nxt = ops.argmax(lg) if lg.dtype == torch.float32 else lg.argmax(-1)
        nxt = torch.where(unfinished, nxt, torch.full_like(nxt, pad))
        out = torch.cat([out, nxt[:, None]], 1)
        if ngram is not None:
            ngram.push(nxt.tolist())
        if eos is not None:
            unfinished = unfinished & ~torch.isin(nxt, eos)
            if not bool(unfinished.any()):
                break
        if out.shape[1] == max_new_tokens:
            break
        logits = sess.step(nxt)
    return out


def _generate_beams(model, input_ids, attention_mask, omic_ids, omic_info_list, max_new_tokens, do_sample, repetition_penalty, pad_token_id,
                    eos_token_id, no_repeat_ngram_size, num_beams, length_penalty, early_stopping, temperature=None, top_k=None, top_p=None,
                    generator=None):
    """Beam search, this build's extension (`OmicsOne.generate(molly_num_beams=N)`; the reference's own `num_beams` parameter is dropped
    by its generate, src/model/omics_one.py:220-232, and is ignored here too).  Procedure = HF `_beam_search` (molly_amd/beam.py): the prompt rows
    repeated num_beams times through one prefill, then molly_amd.beam.beam_search over the decode session (logits from
    `GenerationSession.step`, the KV cache gathered by `GenerationSession.reorder`).  With do_sample=True this is HF's beam sampling:
    the warpers (temperature, top-k, top-p) act on the log-probabilities and the continuations are drawn by torch.multinomial — the
    same procedure; the random stream is the caller's generator on the GPU, so draws differ from a CPU run of HF."""
    from .beam import beam_search
    B, nb = input_ids.shape[0], num_beams
    rep = lambda t: None if t is None else t.repeat_interleave(nb, 0)
    rep_list = lambda l: None if l is None else [x for x in l for _ in range(nb)]
    omic_rep = rep(omic_ids) if torch.is_tensor(omic_ids) else rep_list(omic_ids)
    sess = GenerationSession(model, max_new_tokens)
    logits = sess.prefill(rep(input_ids), rep(attention_mask), omic_rep, rep_list(omic_info_list))
    eos = None if eos_token_id is None else [int(x) for x in torch.as_tensor(eos_token_id).reshape(-1).tolist()]

    def processors(generated, lp):
        # HF applies the processors to the log-probabilities in beam mode: repetition penalty, then the n-gram ban
        if repetition_penalty and repetition_penalty != 1.0 and generated.shape[1] > 0:
            lp = _process_logits(lp, generated, None, None, None, repetition_penalty)
        if no_repeat_ngram_size and generated.shape[1] + 1 >= no_repeat_ngram_size:
            ng = NoRepeatNGram(int(no_repeat_ngram_size), generated.shape[0])       # rows change beams every step: rebuilt, as HF does
            for t in range(generated.shape[1]):
                ng.push(generated[:, t].tolist())
            lp = ng.apply(lp)
        if do_sample:
            lp = _process_logits(lp, generated, temperature, top_k, top_p, None)
        return lp
    use_proc = (repetition_penalty and repetition_penalty != 1.0) or no_repeat_ngram_size or do_sample
    gen = generator
    if do_sample and generator is not None and generator.device.type != logits.device.type:
        gen = torch.Generator(device=logits.device).manual_seed(int(torch.randint(0, 1 << 62, (1,), generator=generator).item()))
    return beam_search(logits, sess.step, sess.reorder, B, nb, max_new_tokens, eos, pad_token_id, length_penalty, early_stopping,
                       processors if use_proc else None, do_sample=bool(do_sample), generator=gen)
