"""Qwen3 decoder forward/backward on the HIP kernels.

Restates (without importing) what the reference executes through HuggingFace + Liger + flash-attn when
`OmicsOne.forward` calls `self.model(inputs_embeds=..., labels=...)` (reference: src/model/omics_one.py:175-184):
HF:models/qwen3/modeling_qwen3.py:294-323 (layer), :240-281 (attention), :76-83 (MLP), :367-425 (model),
:448-507 + HF:loss/loss_utils.py:49-71 (lm_head + shifted CE).  The backward is hand-scheduled: there is no
autograd graph and no tracing; every saved activation lives in a pre-allocated HBM slab.

GEMM forms: forward  y = x W^T          -> A=x k-contiguous, B=W k-contiguous
                        dgrad    dx = dy W           -> A=dy k-contiguous, B=W k-major        (no W^T copy)
                        wgrad    dW = dy^T x         -> the narrower operand transposed once (tr-read kernel), then
                                                        k-contiguous A x k-major B with the result stored transposed;
                                                        token counts that are not a multiple of 64: both operands k-major
"""
from __future__ import annotations

import os

import math
from typing import Dict, Optional

import torch

from . import ops
from .config import LlmConfig
from .lora import TARGETS as LORA_TARGETS
from .params import FlatBuffer

BF16 = torch.bfloat16
# LoRA training: dropout fused into the two rank-r contractions next to it (csrc/lora.hip, round 5); MOLLY_LORA_FUSED=0 restores the
# stand-alone dropout launches + GEMMs (same masks: A/B runs and the comparison in tests/test_gpu_lora.py)
_LORA_FUSED = os.environ.get("MOLLY_LORA_FUSED", "1") != "0"
# ... and the adapters' up-projection y += t B^T as trailing K-tiles of the base projection (molly_gemm_kx_bf16_ctx; MOLLY_LORA_KX=0: the
# accumulating launch per target again)
_LORA_KX = os.environ.get("MOLLY_LORA_KX", "1") != "0"


def _ceil(a, b):
    return (a + b - 1) // b * b


def rope_tables(n_pos: int, head_dim: int, theta: float, device, table_dtype=BF16):
    """cos/sin [n_pos, head_dim/2] fp32.  HF computes them in fp32 and casts to the model dtype
    (HF:models/qwen3/modeling_qwen3.py:126-137) — `table_dtype` reproduces that rounding."""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    fr = torch.arange(n_pos, dtype=torch.float32)[:, None] * inv[None, :]
    return (fr.cos().to(table_dtype).float().to(device).contiguous(),
            fr.sin().to(table_dtype).float().to(device).contiguous())


class Qwen3Engine:
    def __init__(self, cfg: LlmConfig, params: FlatBuffer, grads: Optional[FlatBuffer], device, prefix: str = "model.",
                 ce_chunk_rows: int = 16384, rope_table_dtype=BF16, lora=None, backward: Optional[bool] = None):
        """grads: flat gradient buffer of the base weights (None = base frozen: the backward then only propagates to the
        inputs and to the adapters).  lora: a `LoraRuntime` (molly_amd.lora) or None.  backward: build the backward at all
        (default: when anything upstream or inside is trainable, i.e. grads or lora given)."""
        self.cfg, self.P, self.G, self.dev, self.pre = cfg, params, grads, device, prefix
        self.lora = lora
        self.train_base = grads is not None
        self.has_backward = (grads is not None or lora is not None) if backward is None else backward
        self.h, self.hd = cfg.hidden_size, cfg.head_dim
        self.nh, self.nkv, self.ff, self.V, self.L = (cfg.num_attention_heads, cfg.num_key_value_heads,
                                                      cfg.intermediate_size, cfg.vocab_size, cfg.num_hidden_layers)
        self.nqkv = (self.nh + 2 * self.nkv) * self.hd
        self.nqk = (self.nh + self.nkv) * self.hd
        self.ce_chunk_rows = ce_chunk_rows
        self.rope_table_dtype = rope_table_dtype
        # SwiGLU in the gate|up GEMM's epilogue (MOLLY_GEMM_SWIGLU): not with LoRA (the adapters add to gate / up after the
        # base GEMM), needs ff % 128 == 0 (every Qwen3 size); MOLLY_FUSED_SWIGLU=0 restores the two-kernel path
        self.fused_swiglu = (lora is None and self.ff % 128 == 0 and os.environ.get("MOLLY_FUSED_SWIGLU", "1") != "0")
        # its backward in the down-projection's dgrad epilogue (MOLLY_GEMM_SWIGLU_BWD): d(act) never reaches HBM.  Not with LoRA
        # (the down_proj adapter adds to d(act)); MOLLY_FUSED_SWIGLU_BWD=0 restores dgrad + swiglu_bwd
        self.fused_swiglu_bwd = lora is None and os.environ.get("MOLLY_FUSED_SWIGLU_BWD", "1") != "0"
        self.cap = 0
        self.tT = None
        self.tTg, self._pend = None, []   # grouped per-layer weight gradients (reserve)
        self.grads_final_hook = None      # callable(lo, hi): flat-offset range of gradients that became final (ZeRO overlap)
        self.wait_params_hook = None      # callable(lo, hi): block the stream until those parameters are all-gathered
        self._views()
        lp0, lpl = f"{prefix}model.layers.0.", f"{prefix}model.layers.{self.L - 1}."
        self.layer_lo = [params.offsets[f"{prefix}model.layers.{i}.self_attn.q_proj.weight"] for i in range(self.L)]
        self.layers_hi = params.offsets[lpl + "mlp.down_proj.weight"] + params.views[lpl + "mlp.down_proj.weight"].numel()

    # ---- weight views ---------------------------------------------------------------------------------------
    def _layer_views(self, buf: FlatBuffer, i: int):
        lp = f"{self.pre}model.layers.{i}."
        return dict(
            qkv=buf.span(lp + "self_attn.q_proj.weight", lp + "self_attn.v_proj.weight", self.h),
            o=buf.views[lp + "self_attn.o_proj.weight"],
            gu=buf.span(lp + "mlp.gate_proj.weight", lp + "mlp.up_proj.weight", self.h),
            down=buf.views[lp + "mlp.down_proj.weight"],
            ln1=buf.views[lp + "input_layernorm.weight"], ln2=buf.views[lp + "post_attention_layernorm.weight"],
            qn=buf.views[lp + "self_attn.q_norm.weight"], kn=buf.views[lp + "self_attn.k_norm.weight"],
        )

    def _views(self):
        self.W = [self._layer_views(self.P, i) for i in range(self.L)]
        self.embed = self.P.views[self.pre + "model.embed_tokens.weight"]
        self.head = self.embed if self.cfg.tie_word_embeddings else self.P.views[self.pre + "lm_head.weight"]
        self.norm_w = self.P.views[self.pre + "model.norm.weight"]
        if self.G is not None:
            self.dW = [self._layer_views(self.G, i) for i in range(self.L)]
            self.d_embed = self.G.views[self.pre + "model.embed_tokens.weight"]
            self.d_head = self.d_embed if self.cfg.tie_word_embeddings else self.G.views[self.pre + "lm_head.weight"]
            self.d_norm_w = self.G.views[self.pre + "model.norm.weight"]

    # ---- activation slabs -----------------------------------------------------------------------------------
    def reserve(self, M: int, B: int, T: int, training: bool):
        """Allocate (once) every buffer a forward/backward over M = B*T tokens needs."""
        if self.cap >= M and getattr(self, "_train_alloc", False) >= training and self.BT == (B, T):
            return
        dev, h, ff = self.dev, self.h, self.ff
        e = lambda *s, dt=BF16: torch.empty(*s, dtype=dt, device=dev)
        self.cap, self.BT, self._train_alloc = M, (B, T), training
        nl = self.L if training else 1                    # inference keeps one layer's worth of scratch
        self.A = []
        for _ in range(nl):
            self.A.append(dict(x=e(M, h), xn=e(M, h), qkv=e(M, self.nqkv), qk=e(M, self.nqk), attn=e(M, self.nh * self.hd),
                               lse=e(B, self.nh, T, dt=torch.float32), x2=e(M, h), xn2=e(M, h), gu=e(M, 2 * ff), act=e(M, ff)))
            if self.lora is not None:
                # s * (dropout(x) A^T) of every target, kept for the adapter gradients
                self.A[-1]["lt"] = {m: e(M, self.lora.rp) for m in LORA_TARGETS}
                if training and self.lora.p > 0.0:
                    # dropout(x) of every target as well (0.6 GB per layer at 16 Ki tokens of Molly-1.7B — HBM is 288 GB):
                    # the backward then reads it instead of regenerating the mask in another pass over x
                    dims = {"q_proj": h, "k_proj": h, "v_proj": h, "o_proj": self.nh * self.hd, "gate_proj": h, "up_proj": h,
                            "down_proj": ff}
                    self.A[-1]["lxd"] = {m: e(M, d) for m, d in dims.items()}
        if self.lora is not None:
            self.lora_dt = e(M, self.lora.rp)
            self.lora_dt3 = e(M, 3 * self.lora.rp) if (training and _LORA_FUSED and os.environ.get("MOLLY_LORA_UP_MERGE", "1") != "0") else None
            self.lora_bT = e(self.lora.rp * max(self.nqkv, 2 * ff, h))
            # adapter gradients (dB, dA of the seven targets) of one layer as ONE grouped launch: each keeps its transposed
            # rank-r operand (t^T, dt^T: rp x M) until the layer's last dgrad; dropout(x) then has to survive the dA GEMM, so
            # the masked dx term goes through a buffer of its own
            self.lora_tT = None
            if M % 64 == 0 and self.lora.rp % 64 == 0 and os.environ.get("MOLLY_GROUPED_WGRAD", "1") != "0":
                self.lora_tT = {m: (e(self.lora.rp * M), e(self.lora.rp * M)) for m in LORA_TARGETS}
                self.lora_tmp = e(M * max(h, ff, self.nh * self.hd))
                if training and _LORA_FUSED and self.lora.rp == 64:
                    # t^T of every target and layer, written by the kernel that produces t (no transpose launch in the backward: 4 MB each)
                    for a in self.A:
                        a["ltT"] = {m: e(self.lora.rp, M) for m in LORA_TARGETS}
            if training and _LORA_FUSED and self.lora.rp == 64 and getattr(self, "lora_bTs", None) is None:
                # B^T of every target (the skinny dt = s * dy B reads it as its `A`), refreshed by ONE launch per backward
                odim = {"q_proj": self.nh * self.hd, "k_proj": self.nkv * self.hd, "v_proj": self.nkv * self.hd, "o_proj": h, "gate_proj": ff,
                        "up_proj": ff, "down_proj": h}
                self.lora_bTs = [{m: e(self.lora.rp, odim[m]) for m in LORA_TARGETS} for _ in range(self.L)]
                self.lora_pack_t = ops.lora_pack_items([(self.lora.B[i][m], self.lora_bTs[i][m]) for i in range(self.L) for m in LORA_TARGETS])
        self.lora_kx = False
        if self.lora is not None and _LORA_KX and _LORA_FUSED and self.lora.rp == 64:
            # The up-projection of every adapter rides in its base projection as K-tiles 'behind' the weight's own (one accumulation, one rounding, no
            # read-modify-write pass over the projection's output: 2.7 GB per layer at 32 Ki tokens): the t of a fused projection's targets sit
            # side by side in ONE [M, 64 n] buffer, their B matrices as the diagonal blocks of ONE [N, 64 n] stack (zeros elsewhere, refreshed by one
            # launch per forward) — and the gate|up projection keeps its SwiGLU epilogue, which the separate accumulation had cost.
            rp, nqd, nkd = 64, self.nh * self.hd, self.nkv * self.hd
            shapes = ((self.nqkv, h, 3 * rp, False, False), (h, nqd, rp, False, True), (2 * ff, h, 2 * rp, True, False), (h, ff, rp, False, True))
            if all(ops.gemm_kx_supported(M, n, k, k2, swiglu=sw, res=rs) for n, k, k2, sw, rs in shapes):
                self.lora_kx = True
                for a in self.A:
                    a["ltq"], a["ltg"] = e(M, 3 * rp), e(M, 2 * rp)
                    a["lt"] = {"q_proj": a["ltq"][:, :rp], "k_proj": a["ltq"][:, rp:2 * rp], "v_proj": a["ltq"][:, 2 * rp:],
                               "gate_proj": a["ltg"][:, :rp], "up_proj": a["ltg"][:, rp:], "o_proj": a["lt"]["o_proj"], "down_proj": a["lt"]["down_proj"]}
                if getattr(self, "lora_b2", None) is None:
                    z = lambda *sh: torch.zeros(*sh, dtype=BF16, device=dev)
                    self.lora_b2, pairs = [], []
                    for i in range(self.L):
                        bq, bg = z(self.nqkv, 3 * rp), z(2 * ff, 2 * rp)
                        self.lora_b2.append({"qkv": bq, "gu": bg})
                        Bi = self.lora.B[i]
                        pairs += [(Bi["q_proj"], bq[:nqd, :rp]), (Bi["k_proj"], bq[nqd:nqd + nkd, rp:2 * rp]), (Bi["v_proj"], bq[nqd + nkd:, 2 * rp:]),
                                  (Bi["gate_proj"], bg[:ff, :rp]), (Bi["up_proj"], bg[ff:, rp:])]
                    self.lora_pack = ops.lora_pack_items(pairs)
        self.x_out = e(M, h)
        self.hn = e(M, h)
        self.C = min(self.ce_chunk_rows, M)
        self.logits = e(self.C, self.V)
        self.row_loss = e(M, dt=torch.float32)
        self.scal = torch.zeros(8, dtype=torch.float32, device=dev)   # [0]=1/n_valid [1]=n_valid [2]=loss
        self.cos, self.sin = rope_tables(T, self.hd, self.cfg.rope_theta, dev, self.rope_table_dtype)
        if training:
            self.d_a = e(M, h); self.d_b = e(M, h); self.d_c = e(M, h)      # residual-stream gradients (ping-pong)
            self.d_act = e(M, ff); self.d_gu = e(M, 2 * ff)
            self.d_attn = e(M, self.nh * self.hd); self.d_qkv = e(M, self.nqkv); self.d_qk = e(M, self.nqk)
            self.delta = e(B, self.nh, T, dt=torch.float32)
            # one sample per GPU: the dK / dV passes of the attention backward split by query head (attention.hip SPLIT)
            n_ws = ops.attn_bwd_workspace(B, T, self.nh, self.nkv, self.hd) if os.environ.get("MOLLY_ATTN_DKV_SPLIT", "1") != "0" else 0
            self.attn_ws = e(n_ws, dt=torch.float32) if n_ws else None
            self.hn_s = e(M, h); self.dh_s = e(M, h)                       # compacted scored rows (head GEMMs)
            self.tT = e(max(h, self.nh * self.hd) * M)                     # transposed narrow operand of the wgrad GEMMs
            # the four weight-gradient GEMMs of a layer as ONE grouped launch (no split-K, no reduce launches) when their
            # 256x256 tiles fill whole rounds of the 256 CUs: each keeps its own transposed narrow operand until the launch
            self.tTg, self._pend = None, []
            if self.train_base and M % 64 == 0 and os.environ.get("MOLLY_GROUPED_WGRAD", "1") != "0":
                dims = [(self.nqkv, h), (h, self.nh * self.hd), (2 * ff, h), (h, ff)]        # (out, in) of qkv, o, gate|up, down
                if all(min(n, k) % 64 == 0 for n, k in dims):
                    per = [-(-n // 256) * -(-k // 256) for n, k in dims]
                    eff = lambda t: t / (-(-t // 256) * 256)
                    # worth it when the single launches would go through split-K (their own grids do not fill the chip:
                    # Qwen3-0.6B/1.7B widths) and the combined grid does; where every weight gradient fills the chip by
                    # itself (4B: two of four, 8B: all) deferring the GEMMs only costs cache locality (measured -2...-4 %)
                    split = sum(1 for t in per if not (t >= 200 and eff(t) >= 0.8))
                    # (at one sample per GPU the contraction is only M / 64 = 48-64 K-tiles long: the launches' fixed costs and split-K
                    # reduces weigh more, and two of four is enough — Molly-4B at M = 3072: 211.2 -> 209.0 ms/step; Molly-8B, one of
                    # four, stays ungrouped: 396.6 against 398.7 grouped.  MOLLY_GROUPED_WGRAD=2 forces it for an A/B)
                    if ((split >= 3 or (split >= 2 and M <= 4096)) and eff(sum(per)) >= 0.85) or os.environ.get("MOLLY_GROUPED_WGRAD") == "2":
                        self.tTg = [e(min(n, k) * M) for n, k in dims]
                        # round 5: the two RMSNorm outputs of a layer are stored a second time, TRANSPOSED, by the norm kernel itself
                        # (molly_rmsnorm_fwd_t): the q|k|v and gate|up weight gradients read that instead of a transpose launch's
                        # output (2 x 2 bytes x M x h per layer of extra saved activations: 7.5 GB at 16 x 2,048 tokens of Molly-1.7B)
                        if (h <= self.nqkv and h <= 2 * ff and ops.rmsnorm_fwd_t_supported(M, h)
                                and os.environ.get("MOLLY_NORM_TRANSPOSED_STORE", "1") != "0"):
                            for a in self.A:
                                a["xnT"], a["xn2T"] = e(h, M), e(h, M)
                            # ... and the attention forward its output (the o-projection's weight gradient reads attn^T)
                            if self.nh * self.hd <= h and T % 128 == 0 and self.hd in (64, 128):
                                for a in self.A:
                                    a["attnT"] = e(self.nh * self.hd, M)
            if not self.train_base:
                self.junk = torch.zeros(max(h, 2 * self.hd), dtype=BF16, device=dev)   # gain gradients nobody reads
            # split-K scratch for the wgrad GEMMs: 8 slabs of the largest per-layer weight
            ops.ensure_gemm_workspace(8 * 4 * max(2 * ff * h, self.nqkv * h), dev)
            nb1 = ops.lib().query("molly_rmsnorm_bwd_blocks", M)
            nb2 = ops.lib().query("molly_norm_rope_bwd_blocks")
            self.ws = torch.empty(max(nb1 * h, nb2 * 2 * self.hd), dtype=torch.float32, device=dev)
            # gain gradients deferred to ONE batched column reduction at the end of the backward: every norm backward keeps
            # its per-block partials in a workspace of its own (2L + 1 RMSNorm gains: nb1 x h floats each — 8 MB at h 2048;
            # 2L q/k-norm gains: nb2 x 2 hd) instead of launching a 64-block reduce behind itself
            self.ws_defer = None
            defer = self.train_base and os.environ.get("MOLLY_DEFER_COLSUM", "1") != "0"
            # round 6: the q/k-norm + rotary backward inside the attention backward's dQ / dK row epilogues (ops.attn_bwd_rope) instead of a kernel
            # of its own behind it — where the attention backward is not split by query head (one sample per GPU) and the gain gradients
            # either go through the deferred column reduction or are not wanted at all (frozen base)
            self.rope_fused = (self.hd == 128 and ops.attn_bwd_workspace(B, T, self.nh, self.nkv, self.hd) == 0 and (defer or not self.train_base)
                               and os.environ.get("MOLLY_ATTN_ROPE_FUSE", "1") != "0")
            nbq, nbk = ops.attn_bwd_rope_blocks(B, T, self.nh, self.nkv) if self.rope_fused else (0, 0)
            if self.rope_fused and not self.train_base:
                self.ws_ropeq = [torch.empty(nbq * self.hd, dtype=torch.float32, device=dev)] * self.L      # gain gradients nobody reads
                self.ws_ropek = [torch.empty(nbk * self.hd, dtype=torch.float32, device=dev)] * self.L
            if defer:
                L = self.L
                self.ws_rms = torch.empty(2 * L + 1, nb1 * h, dtype=torch.float32, device=dev)
                if self.rope_fused:
                    self.ws_ropeq = torch.empty(L, nbq * self.hd, dtype=torch.float32, device=dev)
                    self.ws_ropek = torch.empty(L, nbk * self.hd, dtype=torch.float32, device=dev)
                else:
                    self.ws_qk = torch.empty(L, nb2 * 2 * self.hd, dtype=torch.float32, device=dev)
                ent = [(self.ws_rms[2 * L], self.d_norm_w, nb1, h, h)]
                for i in range(L):
                    g = self.dW[i]
                    ent += [(self.ws_rms[2 * i], g["ln2"], nb1, h, h), (self.ws_rms[2 * i + 1], g["ln1"], nb1, h, h)]
                    if self.rope_fused:
                        ent += [(self.ws_ropeq[i], g["qn"], nbq, self.hd, self.hd), (self.ws_ropek[i], g["kn"], nbk, self.hd, self.hd)]
                    else:
                        ent += [(self.ws_qk[i], g["qn"], nb2, self.hd, 2 * self.hd),
                                (self.ws_qk[i][self.hd:], g["kn"], nb2, self.hd, 2 * self.hd)]
                self.ws_defer = ops.colsum_items(ent, dev)

    # ---- forward ---------------------------------------------------------------------------------------------
    def forward(self, inputs_embeds: torch.Tensor, B: int, T: int, kv_lo=None, kv_hi=None,
                labels_shifted: Optional[torch.Tensor] = None, training: bool = True, return_logits: bool = False,
                scored_rows: Optional[torch.Tensor] = None):
        """inputs_embeds [B*T, h] bf16.  labels_shifted int64 [B*T]: row r is scored against labels_shifted[r]
        (HF shift: pad one ignore column then drop the first, HF:loss/loss_utils.py:60-63).
        scored_rows (training only): int32 device tensor of the rows whose label is not ignore_index, in ascending order —
        the lm_head GEMMs and the CE then run on those rows only (rows with ignore_index contribute exactly zero loss and
        zero gradient, so the result is identical; SFT batches mask the whole prompt).
        Returns (loss device-scalar view or None, logits [B*T, V] or None)."""
        cfg, M = self.cfg, B * T
        self.reserve(M, B, T, training)
        self.B_, self.T_, self.kv = B, T, (kv_lo, kv_hi)
        if self.lora is not None and training:
            self.lora.step += 1                           # a fresh dropout stream per training forward
        kx = self.lora is not None and self.lora_kx
        if kx:
            ops.lora_pack_b(self.lora_pack)               # this step's B matrices into the stacked operands (all layers, one launch)
        x = inputs_embeds
        for i in range(self.L):
            a = self.A[i if training else 0]
            w = self.W[i]
            if self.wait_params_hook is not None:
                self.wait_params_hook(self.layer_lo[i], self.layer_lo[i + 1] if i + 1 < self.L else self.layers_hi)
            if training:
                a["x"].copy_(x) if x.data_ptr() != a["x"].data_ptr() else None
                xin = a["x"]
            else:
                xin = x
            ops.rmsnorm_fwd(xin, w["ln1"], cfg.rms_norm_eps, out=a["xn"], out_t=a.get("xnT") if training else None)
            if kx:
                for mod in ("q_proj", "k_proj", "v_proj"):
                    self._lora_t(i, a, mod, a["xn"], training)
                ops.gemm_nt_kx(a["xn"], w["qkv"], a["ltq"], self.lora_b2[i]["qkv"], a["qkv"])
            else:
                ops.gemm_nt(a["xn"], w["qkv"], out=a["qkv"])
            if self.lora is not None and not kx:
                nq_, nk_ = self.nh * self.hd, self.nkv * self.hd
                self._lora_fwd(i, a, "q_proj", a["xn"], a["qkv"][:, :nq_], training)
                self._lora_fwd(i, a, "k_proj", a["xn"], a["qkv"][:, nq_:nq_ + nk_], training)
                self._lora_fwd(i, a, "v_proj", a["xn"], a["qkv"][:, nq_ + nk_:], training)
            ops.norm_rope_fwd(a["qkv"], a["qk"], self.nh, self.nkv, self.hd, T, w["qn"], w["kn"], self.cos, self.sin,
                              eps=cfg.rms_norm_eps)
            ops.attn_fwd(a["qk"][:, :self.nh * self.hd], a["qk"][:, self.nh * self.hd:], a["qkv"][:, self.nqk:], B, T,
                         self.nh, self.nkv, self.hd, self.hd ** -0.5, True, kv_lo, kv_hi, out=a["attn"], lse=a["lse"],
                         out_t=a.get("attnT") if training else None)
            if kx:
                self._lora_t(i, a, "o_proj", a["attn"], training)
                ops.gemm_nt_kx(a["attn"], w["o"], a["lt"]["o_proj"], self.lora.B[i]["o_proj"], a["x2"], res=xin)
            else:
                ops.gemm_nt(a["attn"], w["o"], out=a["x2"], res=xin)
                if self.lora is not None:
                    self._lora_fwd(i, a, "o_proj", a["attn"], a["x2"], training)
            ops.rmsnorm_fwd(a["x2"], w["ln2"], cfg.rms_norm_eps, out=a["xn2"], out_t=a.get("xn2T") if training else None)
            if kx:
                self._lora_t(i, a, "gate_proj", a["xn2"], training)
                self._lora_t(i, a, "up_proj", a["xn2"], training)
                ops.gemm_nt_kx(a["xn2"], w["gu"], a["ltg"], self.lora_b2[i]["gu"], a["gu"], act=a["act"])
            elif self.fused_swiglu:
                # gate|up projection with the activation in its epilogue: one launch, no second pass over gu
                ops.gemm_gate_up_swiglu(a["xn2"], w["gu"], a["gu"], a["act"])
            else:
                ops.gemm_nt(a["xn2"], w["gu"], out=a["gu"])
                if self.lora is not None:
                    self._lora_fwd(i, a, "gate_proj", a["xn2"], a["gu"][:, :self.ff], training)
                    self._lora_fwd(i, a, "up_proj", a["xn2"], a["gu"][:, self.ff:], training)
                ops.swiglu_fwd(a["gu"], out=a["act"])
            nxt = self.A[i + 1]["x"] if (training and i + 1 < self.L) else self.x_out
            if kx:
                self._lora_t(i, a, "down_proj", a["act"], training)
                ops.gemm_nt_kx(a["act"], w["down"], a["lt"]["down_proj"], self.lora.B[i]["down_proj"], nxt, res=a["x2"])
            else:
                ops.gemm_nt(a["act"], w["down"], out=nxt, res=a["x2"])
                if self.lora is not None:
                    self._lora_fwd(i, a, "down_proj", a["act"], nxt, training)
            x = nxt
        ops.rmsnorm_fwd(self.x_out, self.norm_w, cfg.rms_norm_eps, out=self.hn)
        loss = None
        logits_all = None
        if return_logits:
            logits_all = torch.empty(M, self.V, dtype=BF16, device=self.dev)
            for c0 in range(0, M, self.C):
                c1 = min(M, c0 + self.C)
                ops.gemm_nt(self.hn[c0:c1], self.head, out=logits_all[c0:c1])
        if labels_shifted is not None:
            self.labels = labels_shifted
            self.scored_rows = scored_rows
            ops.count_valid(labels_shifted, self.scal[0:1], self.scal[1:2])
            if not (training and self.has_backward):
                # loss only (eval): logits chunk by chunk, no gradient written
                for c0 in range(0, M, self.C):
                    c1 = min(M, c0 + self.C)
                    lg = logits_all[c0:c1] if logits_all is not None else ops.gemm_nt(self.hn[c0:c1], self.head,
                                                                                       out=self.logits[:c1 - c0])
                    ops.ce_fwd_bwd(lg, labels_shifted[c0:c1], self.row_loss[c0:c1], self.scal[0:1], write_grad=False)
                ops.sum_f32(self.row_loss[:M], self.scal[2:3], scale=self.scal[0:1])
                loss = self.scal[2]
        return loss, logits_all

    # ---- LoRA branch (PEFT lora.Linear.forward: result + lora_B(lora_A(dropout(x))) * scaling) --------------------
    def _lora_xd(self, i: int, a: dict, mod: str, x: torch.Tensor, training: bool) -> torch.Tensor:
        lo = self.lora
        if not (training and lo.p > 0.0):
            return x
        return ops.dropout(x, lo.p, lo.mask_seed(i, mod), out=a["lxd"][mod])

    def _lora_t(self, i: int, a: dict, mod: str, x: torch.Tensor, training: bool):
        """t = s * dropout(x) A^T of one target into its slice of the fused projection's t buffer (the K-extended GEMM adds t B^T); dropout(x) is kept
        for the backward when there is a mask."""
        lo = self.lora
        p = lo.p if training else 0.0
        ops.lora_down_drop(x, lo.A[i][mod], p, lo.mask_seed(i, mod) if p > 0.0 else 0, lo.scale,
                           xd=a["lxd"][mod] if p > 0.0 else None, out=a["lt"][mod], out_t=a["ltT"][mod] if (training and "ltT" in a) else None)

    def _lora_fwd(self, i: int, a: dict, mod: str, x: torch.Tensor, y: torch.Tensor, training: bool):
        """y += s * (dropout(x) A^T) B^T; keeps t = s * dropout(x) A^T for the backward."""
        lo = self.lora
        t = a["lt"][mod]
        if training and lo.p > 0.0 and _LORA_FUSED and lo.rp == 64 and x.shape[1] % 64 == 0:
            # dropout, the rank-r down-projection and the alpha / r scaling in one launch; dropout(x) is written on the way (dA reads it)
            ops.lora_down_drop(x, lo.A[i][mod], lo.p, lo.mask_seed(i, mod), lo.scale, xd=a["lxd"][mod], out=t,
                               out_t=a["ltT"][mod] if "ltT" in a else None)
        else:
            ops.gemm_nt(self._lora_xd(i, a, mod, x, training), lo.A[i][mod], out=t)
            if lo.scale != 1.0:
                ops.scale_(t, lo.scale)
            if training and "ltT" in a:
                ops.transpose(t, a["ltT"][mod])
        ops.gemm_nt(t, lo.B[i][mod], out=y, accumulate=True)

    def _lora_bwd(self, i: int, a: dict, mod: str, x: torch.Tensor, dy: torch.Tensor, dx: torch.Tensor, accumulate: bool, group=None):
        """dB (+)= dy^T t ; dt = s * dy B ; dA (+)= dt^T dropout(x) ; dx += mask * (dt A).
        group (a list, or None): targets that share their input (q | k | v; gate | up) leave their (dt, A, seed) in it instead of launching the last
        term — the caller adds all of them to dx in ONE pass (`_lora_bwd_flush`)."""
        lo = self.lora
        grouped = getattr(self, "lora_tT", None) is not None
        M = dy.shape[0]
        if grouped:
            if "ltT" in a:
                tt = a["ltT"][mod]                                     # written by the forward beside t
            else:
                tt = self.lora_tT[mod][0].view(lo.rp, M)
                ops.transpose(a["lt"][mod], tt)
            self._pend.append((tt, dy, lo.dB[i][mod], True))           # dB^T[rp, out] = t^T dy, stored transposed
        else:
            self._wgrad(dy, a["lt"][mod], lo.dB[i][mod], accumulate)
        up_fused = lo.p > 0.0 and _LORA_FUSED and lo.rp == 64 and x.shape[1] % 128 == 0 and dx.is_contiguous()
        if not (up_fused and getattr(self, "lora_dt3", None) is not None):
            group = None
        # (a group's dt's live side by side until its one pass over dx)
        dt = self.lora_dt if group is None else self.lora_dt3[:, lo.rp * len(group):lo.rp * (len(group) + 1)]
        if _LORA_FUSED and lo.rp == 64 and dy.shape[1] % 64 == 0 and dy.stride(1) == 1 and dy.stride(0) % 8 == 0:
            # dt = s * dy B as an operand stream (dy read once at the HBM rate) instead of a 64-column grid on the 128 x 128 GEMM kernel:
            # B [out, 64] transposed into a scratch (a 5 us launch), then the skinny product with the scale in its epilogue
            dtt = self.lora_tT[mod][1].view(lo.rp, M) if grouped else None
            if getattr(self, "lora_bTs", None) is not None:
                bt = self.lora_bTs[i][mod]                             # (refreshed once per backward: loss_and_backward)
            else:
                bt = self.lora_bT[:lo.rp * dy.shape[1]].view(lo.rp, dy.shape[1])
                ops.transpose(lo.B[i][mod], bt)
            ops.lora_down_drop(dy, bt, 0.0, 0, lo.scale, out=dt, out_t=dtt)
            dtt_done = dtt is not None
        else:
            dtt_done = False
            if group is not None:                                      # (the plain GEMM + scale want a contiguous dt)
                group, dt = None, self.lora_dt
            self._dgrad(dy, lo.B[i][mod], dt)
            if lo.scale != 1.0:
                ops.scale_(dt, lo.scale)
        xd = a["lxd"][mod] if lo.p > 0.0 else x                    # the forward's dropout(x), kept
        if grouped:
            dtt = self.lora_tT[mod][1].view(lo.rp, M)
            if not dtt_done:
                ops.transpose(dt, dtt)
            self._pend.append((dtt, xd, lo.dA[i][mod], False))         # dA[rp, in] = dt^T dropout(x)
        else:
            self._wgrad(dt, xd, lo.dA[i][mod], accumulate)
        if up_fused and group is not None:
            group.append((dt, lo.A[i][mod], lo.mask_seed(i, mod)))
        elif up_fused:
            ops.lora_up_drop_acc(dt, lo.A[i][mod], dx, lo.p, lo.mask_seed(i, mod))       # dx += mask * (dt A): one launch
        elif lo.p > 0.0:
            tmp = self.lora_tmp[:M * x.shape[1]].view(M, x.shape[1]) if grouped else xd   # (ungrouped: dropout(x) is dead by now)
            ops.gemm(dt, lo.A[i][mod], out=tmp, b_kmajor=True)
            ops.dropout(tmp, lo.p, lo.mask_seed(i, mod), out=dx, accumulate=True)
        else:
            ops.gemm(dt, lo.A[i][mod], out=dx, accumulate=True, b_kmajor=True)

    def _lora_bwd_flush(self, group, dx: torch.Tensor):
        """dx += sum of the group's mask * (dt A) terms: one read and one write of dx, the roundings of the per-target launches in their order."""
        if group:
            ops.lora_up_drop_acc_multi([g[0] for g in group], [g[1] for g in group], dx, self.lora.p, [g[2] for g in group])

    # ---- helpers ---------------------------------------------------------------------------------------------
    def _wgrad_layer(self, slot: int, dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, accumulate: bool, xt=None):
        """One of the four per-layer weight gradients (slot: 0 qkv, 1 o, 2 gate|up, 3 down).  With the grouped path the
        narrow operand is transposed NOW (while it is hot) into the slot's own buffer and the GEMM is deferred to
        `_wgrad_flush`; dy / x must stay unchanged until then (they do: see the call sites).  `xt`: x^T as its producer already
        stored it (the norm kernels' transposed second store) — no transpose launch then."""
        if self.tTg is None:
            return self._wgrad(dy, x, dw, accumulate)
        M, N = dy.shape
        K = x.shape[1]
        if K <= N:                                             # dw^T[K,N] = (x^T)[K,M] dy[M,N], stored transposed into dw
            if xt is None:
                xt = self.tTg[slot][:K * M].view(K, M)
                ops.transpose(x, xt)
            self._pend.append((xt, dy, dw, True))
        else:                                                  # dw[N,K] = (dy^T)[N,M] x[M,K]
            dyt = self.tTg[slot][:N * M].view(N, M)
            ops.transpose(dy, dyt)
            self._pend.append((dyt, x, dw, False))

    def _wgrad_flush(self, accumulate: bool):
        if not self._pend:
            return
        probs, self._pend = self._pend, []
        probs, carved = _carve_remainder(probs)
        ops.gemm_grouped(probs, accumulate=accumulate)
        if carved is not None:
            a, b, out, to = carved
            ops.gemm(a, b, out=out, accumulate=accumulate, b_kmajor=True, trans_out=to)

    def _wgrad(self, dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, accumulate: bool):
        """dw[N,K] (+)= dy[M,N]^T x[M,K], contraction over the M token rows.
        Fast form (tokens % 64 == 0): transpose the NARROWER operand once (an HBM-bound pass over the smaller matrix) and
        run the GEMM with a k-contiguous A and a k-major B — the layout pair the 256x256 kernel is fastest on (half the
        transposed LDS reads of the both-k-major form):
            x narrower : dw^T[K,N] = (x^T)[K,M] dy[M,N], stored transposed straight into dw
            dy narrower: dw[N,K]   = (dy^T)[N,M] x[M,K]
        Any other token count: both operands k-major (no transposes, any contraction length)."""
        M, N = dy.shape
        K = x.shape[1]
        if M % 64 == 0 and min(N, K) % 64 == 0 and self.tT is not None and min(N, K) * M <= self.tT.numel():
            if K <= N:
                xt = self.tT[:K * M].view(K, M)
                ops.transpose(x, xt)
                ops.gemm(xt, dy, out=dw, accumulate=accumulate, b_kmajor=True, trans_out=True)
            else:
                dyt = self.tT[:N * M].view(N, M)
                ops.transpose(dy, dyt)
                ops.gemm(dyt, x, out=dw, accumulate=accumulate, b_kmajor=True)
            return
        ops.gemm(dy, x, out=dw, accumulate=accumulate, a_kmajor=True, b_kmajor=True)

    @staticmethod
    def _dgrad(dy: torch.Tensor, w: torch.Tensor, out: torch.Tensor):
        """out[M,K] = dy[M,N] w[N,K]: w read in place as the k-major B operand."""
        ops.gemm(dy, w, out=out, b_kmajor=True)

    # ---- fused lm-head + CE forward/backward, then the decoder backward ---------------------------------------
    def loss_and_backward(self, accumulate: bool = False, final_micro: bool = True) -> torch.Tensor:
        """Runs lm_head + shifted CE (forward AND backward, chunked so [M,V] logits never exist at once: the role
        Liger's fused-linear-CE plays in the reference, src/train.py:130-132) and the whole decoder backward.
        Weight gradients land in the flat grad buffer (`accumulate` = add to what is there: GA micro-steps > 0).
        Returns d(inputs_embeds) [M, h]; the loss is in self.scal[2]."""
        cfg, B, T = self.cfg, self.B_, self.T_
        M = B * T
        kv_lo, kv_hi = self.kv
        if self.lora is not None and getattr(self, "lora_bTs", None) is not None:
            ops.lora_pack_bt(self.lora_pack_t)            # this step's B^T of every target and layer (one launch)
        dh = self.d_a
        rows = self.scored_rows
        if rows is not None:
            # compact to the scored rows: gather hn / labels, run the head on n_s rows, scatter d(hn) back (zeros elsewhere)
            n_s = rows.numel()
            hn_s = self.hn_s[:n_s]
            ops.copy_rows(self.hn, hn_s, n_s, src_idx32=rows)
            lab_s = self.labels.index_select(0, rows.long())       # index plumbing (int64 gather of the labels)
            dh.zero_()
            src_hn, src_lab, n_rows, dh_s = hn_s, lab_s, n_s, self.dh_s[:n_s]
        else:
            src_hn, src_lab, n_rows, dh_s = self.hn, self.labels, M, dh
        first = True
        tb = self.train_base
        for c0 in range(0, n_rows, self.C):
            c1 = min(n_rows, c0 + self.C)
            n = c1 - c0
            lg = self.logits[:n]
            ops.gemm_nt(src_hn[c0:c1], self.head, out=lg)
            ops.ce_fwd_bwd(lg, src_lab[c0:c1], self.row_loss[c0:c1], self.scal[0:1], write_grad=True)
            self._dgrad(lg, self.head, dh_s[c0:c1])                                   # d(hn) = dlogits · E
            if tb:
                # dE (+)= dlogits^T · hn   (tied embeddings: the gather-gradient is added later by the caller)
                self._wgrad(lg, src_hn[c0:c1], self.d_head, accumulate or not first)
            first = False
        if rows is not None:
            ops.copy_rows(dh_s, dh, n_rows, dst_idx32=rows)
            if n_rows == 0:
                self.row_loss[:1].zero_()
                if tb and not accumulate:
                    self.d_head.zero_()
        M_loss = n_rows
        ops.sum_f32(self.row_loss[:max(M_loss, 1)], self.scal[2:3], scale=self.scal[0:1])
        # frozen base: the gain gradients' per-block partials stay in the kernels' scratch and are never reduced (no column-sum launches)
        junk = None if tb else self.junk
        gw = (lambda g, k: g[k]) if tb else (lambda g, k: junk[:self.hd] if k == "qn" else
                                             (junk[self.hd:2 * self.hd] if k == "kn" else junk[:self.h]))
        lora = self.lora
        # final norm backward
        dx = self.d_b
        defer = tb and getattr(self, "ws_defer", None) is not None
        ops.rmsnorm_bwd(self.x_out, self.norm_w, dh, None if (defer or not tb) else self.d_norm_w,
                        cfg.rms_norm_eps, dx=dx, dw_accumulate=accumulate and tb,
                        workspace=self.ws_rms[2 * self.L] if defer else self.ws)
        spare = [self.d_a, self.d_c]
        acc_n = accumulate and tb
        nq, nk_ = self.nh * self.hd, self.nkv * self.hd
        for i in reversed(range(self.L)):
            a, w = self.A[i], self.W[i]
            g = self.dW[i] if tb else None
            # ---- MLP: x3 = x2 + down(silu(g)*u)
            if self.fused_swiglu_bwd:
                ops.gemm_down_dgrad_swiglu_bwd(dx, w["down"], a["gu"], self.d_gu)
            else:
                self._dgrad(dx, w["down"], self.d_act)
                if lora is not None:
                    self._lora_bwd(i, a, "down_proj", a["act"], dx, self.d_act, accumulate)
            if tb:
                self._wgrad_layer(3, dx, a["act"], g["down"], accumulate)
            if not self.fused_swiglu_bwd:
                ops.swiglu_bwd(a["gu"], self.d_act, self.d_gu)
            dxn2 = spare[0]
            self._dgrad(self.d_gu, w["gu"], dxn2)
            if lora is not None:
                grp = []
                self._lora_bwd(i, a, "gate_proj", a["xn2"], self.d_gu[:, :self.ff], dxn2, accumulate, group=grp)
                self._lora_bwd(i, a, "up_proj", a["xn2"], self.d_gu[:, self.ff:], dxn2, accumulate, group=grp)
                self._lora_bwd_flush(grp, dxn2)
            if tb:
                self._wgrad_layer(2, self.d_gu, a["xn2"], g["gu"], accumulate, xt=a.get("xn2T"))
            dx2 = spare[1]
            ops.rmsnorm_bwd(a["x2"], w["ln2"], dxn2, None if (defer or not tb) else gw(g, "ln2"), cfg.rms_norm_eps, dres=dx, dx=dx2,
                            dw_accumulate=acc_n, workspace=self.ws_rms[2 * i] if defer else self.ws)
            # ---- attention: x2 = x + o_proj(attn)
            self._dgrad(dx2, w["o"], self.d_attn)
            if lora is not None:
                self._lora_bwd(i, a, "o_proj", a["attn"], dx2, self.d_attn, accumulate)
            if tb:
                self._wgrad_layer(1, dx2, a["attn"], g["o"], accumulate, xt=a.get("attnT"))
            if getattr(self, "rope_fused", False):
                # d(q), d(k) leave the attention backward already through the rotary and the q/k-norm: straight into d(q | k | v)
                ops.attn_bwd_rope(a["qk"][:, :nq], a["qk"][:, nq:], a["qkv"][:, self.nqk:], a["attn"], self.d_attn, a["lse"], B, T,
                                  self.nh, self.nkv, self.hd, self.hd ** -0.5, True, self.d_qkv[:, self.nqk:], a["qkv"], w["qn"], w["kn"],
                                  self.cos, self.sin, cfg.rms_norm_eps, self.d_qkv, self.ws_ropeq[i], self.ws_ropek[i], kv_lo, kv_hi,
                                  delta_ws=self.delta)
            else:
                ops.attn_bwd(a["qk"][:, :nq], a["qk"][:, nq:], a["qkv"][:, self.nqk:], a["attn"], self.d_attn, a["lse"], B, T,
                             self.nh, self.nkv, self.hd, self.hd ** -0.5, True, self.d_qk[:, :nq], self.d_qk[:, nq:],
                             self.d_qkv[:, self.nqk:], kv_lo, kv_hi, delta_ws=self.delta, ws=self.attn_ws)
                ops.norm_rope_bwd(a["qkv"], self.d_qk, self.d_qkv, self.nh, self.nkv, self.hd, T, w["qn"], w["kn"], self.cos,
                                  self.sin, None if (defer or not tb) else gw(g, "qn"), None if (defer or not tb) else gw(g, "kn"), eps=cfg.rms_norm_eps,
                                  dw_accumulate=acc_n, workspace=self.ws_qk[i] if defer else self.ws)
            dxn = spare[0]
            self._dgrad(self.d_qkv, w["qkv"], dxn)
            if lora is not None:
                grp = []
                self._lora_bwd(i, a, "q_proj", a["xn"], self.d_qkv[:, :nq], dxn, accumulate, group=grp)
                self._lora_bwd(i, a, "k_proj", a["xn"], self.d_qkv[:, nq:nq + nk_], dxn, accumulate, group=grp)
                self._lora_bwd(i, a, "v_proj", a["xn"], self.d_qkv[:, nq + nk_:], dxn, accumulate, group=grp)
                self._lora_bwd_flush(grp, dxn)
                self._wgrad_flush(accumulate)                  # the layer's fourteen adapter gradients, one launch
            if tb:
                self._wgrad_layer(0, self.d_qkv, a["xn"], g["qkv"], accumulate, xt=a.get("xnT"))
                # dx, d_gu, dx2 and d_qkv are all still intact here (the norm backward below overwrites dx)
                self._wgrad_flush(accumulate)
            ops.rmsnorm_bwd(a["x"], w["ln1"], dxn, None if (defer or not tb) else gw(g, "ln1"), cfg.rms_norm_eps, dres=dx2, dx=dx,
                            dw_accumulate=acc_n, workspace=self.ws_rms[2 * i + 1] if defer else self.ws)
            if tb and final_micro and self.grads_final_hook is not None:
                self.grads_final_hook(self.layer_lo[i], self.layers_hi)       # matrices of layers i..L-1 are final
        if defer:
            ops.colsum_batched(self.ws_defer, accumulate=acc_n)                # all 4L + 1 gain gradients, one launch
        return dx


def _carve_remainder(probs):
    """The grouped weight-gradient launch walks its 256 x 256 tiles in rounds of 256 (one per CU).  A tile count a FEW past whole rounds —
    Qwen3-4B at one sample per GPU: 240 + 160 + 760 + 380 = 1,540 = 6 rounds + 4 tiles — costs a seventh tile-time on four CUs (14 % of the
    launch, 27 % of that step's GEMM time: profiles/r04_logs/c3_gemm_table.txt).  Then the last tile-row of one problem gives up its last R
    tiles: the problem becomes (at most) two rectangles of the grouped launch, now whole rounds, and the R tiles run as a GEMM of their own behind
    it, K split over the chip (~30 us against the ~85 us tile-time saved).  Every output element is still written by exactly one launch.
    -> (problems for the grouped launch, carved problem | None).  MOLLY_WGRAD_CARVE=0: off."""
    if os.environ.get("MOLLY_WGRAD_CARVE", "1") == "0" or len(probs) > 14:
        return probs, None
    tiles = [(-(-a.shape[0] // 256), -(-b.shape[1] // 256)) for a, b, _, _ in probs]
    total = sum(gm * gn for gm, gn in tiles)
    rem = total % 256
    K = probs[0][0].shape[1]
    if not (total > 256 and 0 < rem <= 16 and K >= 2048):
        return probs, None
    for idx in sorted(range(len(probs)), key=lambda i: -tiles[i][1]):
        gm, gn = tiles[idx]
        a, b, out, to = probs[idx]
        if gn < rem or a.shape[0] % 256 or b.shape[1] % 256 or (gm == 1 and gn == rem):
            continue
        r0, c0 = (gm - 1) * 256, (gn - rem) * 256
        new = []
        if gm > 1:
            new.append((a[:r0], b, out[:, :r0] if to else out[:r0], to))
        if gn > rem:
            new.append((a[r0:], b[:, :c0], out[:c0, r0:] if to else out[r0:, :c0], to))
        carved = (a[r0:], b[:, c0:], out[c0:, r0:] if to else out[r0:, c0:], to)
        return probs[:idx] + new + probs[idx + 1:], carved
    return probs, None
