"""ESM-architecture encoder forward (ESM-2 rotary / nucleotide-transformer v1 absolute positions) on the HIP kernels.

Restates what the reference runs through HF `EsmForMaskedLM(..., output_hidden_states=True)["hidden_states"][-1]`
(reference: src/model/omics_one.py:75-91): HF:models/esm/modeling_esm.py:224-271 (embeddings), :350-396 (self
attention: q*hd^-0.5 before rotary, SDPA scale 1), :412-439 / :442-463 / :517-521 (pre-LN blocks, erf-GELU FFN),
:552-553 (emb_layer_norm_after — `hidden_states[-1]` is this tensor).  The MaskedLM head the reference computes and
throws away is never run.  Encoders are frozen on Molly's default path (reference:
src/utils/tools.py:315-317): `forward` keeps one layer's worth of scratch.  With `grads` (reference `--train-bio`,
src/utils/tools.py:326-330) `forward(training=True)` keeps every layer's activations and `backward`
runs the hand-scheduled encoder backward: LayerNorm / erf-GELU / rotary / bidirectional flash-attention backward kernels,
dgrad GEMMs reading W in place, wgrad GEMMs over the token axis, bias gradients by column sums, and the embedding
gradient through a sorted index (token-dropout rescale folded in as a per-row factor).
"""
from __future__ import annotations

from typing import Optional

import os

import numpy as np
import torch

from . import ops
from .config import EncConfig
from .params import FlatBuffer
from .qwen3 import rope_tables

# the frozen forward replayed from a hipGraph (EsmEngine._forward_replayed): MOLLY_ENC_GRAPH=1 always, 0 never; default: in single-process
# jobs only — a capture beside RCCL's watchdog thread and in-flight collectives has never run on hardware here
_ENC_GRAPH = os.environ.get("MOLLY_ENC_GRAPH", "auto")
_ENC_GRAPH_MAX_ROWS = int(os.environ.get("MOLLY_ENC_GRAPH_MAX_ROWS", "4096"))


def _enc_graph_on() -> bool:
    if _ENC_GRAPH in ("0", "1", True, False):
        return _ENC_GRAPH in ("1", True)
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)

BF16 = torch.bfloat16


class EsmEngine:
    def __init__(self, cfg: EncConfig, params: FlatBuffer, device, prefix: str, rope_table_dtype=BF16,
                 grads: Optional[FlatBuffer] = None):
        if cfg.hidden_dropout_prob or cfg.attention_probs_dropout_prob:
            raise NotImplementedError("encoder dropout > 0 is not implemented (ESM-2 / NT checkpoints ship 0.0)")
        self.cfg, self.P, self.dev, self.pre = cfg, params, device, prefix + "esm."
        self.he, self.nh = cfg.hidden_size, cfg.num_attention_heads
        self.hd = self.he // self.nh
        if self.hd not in (8, 16, 24, 32, 40, 48, 64, 128):
            raise NotImplementedError(f"encoder head_dim={self.hd}: attention is built for 64/128 (MFMA) and 8..48 (small)")
        if grads is not None and self.hd not in (64, 128):
            raise NotImplementedError(f"--train-bio with encoder head_dim={self.hd}: the attention backward is built for "
                                      f"64 and 128 (mini encoders are forward-only)")
        self.ffe = cfg.intermediate_size
        self.rope_table_dtype = rope_table_dtype
        self.G = grads
        self.tT = None
        self.wemb, self.pemb, self.layers, self.lnf_w, self.lnf_b = self._bind(params)
        if grads is not None:
            self.d_wemb, self.d_pemb, self.dlayers, self.d_lnf_w, self.d_lnf_b = self._bind(grads)
        self.cap = (0, 0, False)
        self._g, self._g_key, self._g_seen, self._g_recaptures, self._g_prof = None, None, 0, 0, []

    def _bind(self, buf: FlatBuffer):
        cfg, v, e = self.cfg, buf.views, self.pre
        wemb = v[e + "embeddings.word_embeddings.weight"]
        pemb = v.get(e + "embeddings.position_embeddings.weight") if cfg.position_embedding_type == "absolute" else None
        layers = []
        for i in range(cfg.num_hidden_layers):
            lp = f"{e}encoder.layer.{i}."
            layers.append(dict(
                qkv_w=buf.span(lp + "attention.self.query.weight", lp + "attention.self.value.weight", self.he),
                qkv_b=buf.span(lp + "attention.self.query.bias", lp + "attention.self.value.bias", 3 * self.he)[0],
                ao_w=v[lp + "attention.output.dense.weight"], ao_b=v[lp + "attention.output.dense.bias"],
                ln1_w=v[lp + "attention.LayerNorm.weight"], ln1_b=v[lp + "attention.LayerNorm.bias"],
                i_w=v[lp + "intermediate.dense.weight"], i_b=v[lp + "intermediate.dense.bias"],
                o_w=v[lp + "output.dense.weight"], o_b=v[lp + "output.dense.bias"],
                ln2_w=v[lp + "LayerNorm.weight"], ln2_b=v[lp + "LayerNorm.bias"]))
        return wemb, pemb, layers, v[e + "encoder.emb_layer_norm_after.weight"], v[e + "encoder.emb_layer_norm_after.bias"]

    def reserve(self, n_seq: int, K: int, training: bool = False):
        if self.cap[:2] == (n_seq, K) and self.cap[2] >= training:
            return
        N, he = n_seq * K, self.he
        e = lambda *s, dt=BF16: torch.empty(*s, dtype=dt, device=self.dev)
        self.x, self.x2, self.ln = e(N, he), e(N, he), e(N, he)
        self.qkv, self.qk, self.att = e(N, 3 * he), e(N, 2 * he), e(N, he)
        self.mid = e(N, self.ffe)
        self.out = e(N, he)
        self.pos = e(n_seq, K, dt=torch.int32)
        self.klen = e(n_seq, dt=torch.int32)
        self.klo = torch.zeros(n_seq, dtype=torch.int32, device=self.dev)
        if self.cfg.position_embedding_type == "rotary":
            self.cos, self.sin = rope_tables(K, self.hd, self.cfg.rope_theta, self.dev, self.rope_table_dtype)
        else:
            self.cos = self.sin = None
        if training:
            # every layer's activations (ESM2-650M at 8 x 512 tokens: 33 x 190 MB)
            self.A = [dict(x=e(N, he), ln1=e(N, he), qkv=e(N, 3 * he), qk=e(N, 2 * he), att=e(N, he),
                           lse=e(n_seq, self.nh, K, dt=torch.float32), x2=e(N, he), ln2=e(N, he), z=e(N, self.ffe),
                           mid=e(N, self.ffe)) for _ in range(self.cfg.num_hidden_layers)]
            self.x_last = e(N, he)
            self.d_x, self.d_x2, self.d_ln = e(N, he), e(N, he), e(N, he)
            self.d_mid, self.d_z = e(N, self.ffe), e(N, self.ffe)
            self.d_att, self.d_qkv, self.d_qk = e(N, he), e(N, 3 * he), e(N, 2 * he)
            self.delta = e(n_seq, self.nh, K, dt=torch.float32)
            self.tT = e(he * N)                               # transposed narrow operand of the wgrad GEMMs
            ops.ensure_gemm_workspace(8 * 4 * self.ffe * he, self.dev)
            nb = ops.lib().query("molly_layernorm_bwd_blocks", N)
            self.ws = torch.empty(max(2 * nb * he, ops.lib().query("molly_norm_rope_bwd_blocks") * 2 * self.hd,
                                      ops.lib().query("molly_colsum_parts", N) * max(self.ffe, 3 * he)),
                                  dtype=torch.float32, device=self.dev)
        self.cap = (n_seq, K, training)
        self._g, self._g_seen = None, 0                       # a captured forward holds the old buffers' addresses

    def forward(self, ids: torch.Tensor, training: bool = False) -> torch.Tensor:
        """ids int64 [n_seq, K] on the GPU, pad id 1 = masked key (reference: src/model/omics_one.py:70).
        Returns the final-LayerNormed hidden states [n_seq*K, he] (bf16).  training=True keeps what `backward` needs."""
        cfg = self.cfg
        n_seq, K = ids.shape
        if training and self.G is None:
            raise RuntimeError("EsmEngine: training forward without a gradient buffer (prepare(train_bio=True))")
        self.reserve(n_seq, K, training)
        if training:
            return self._forward_train(ids, n_seq, K)
        # (only where the stack is launch-bound: up to 4,096 rows a launch is 5-30 us; at the headline's 8,224 rows nothing is gained and the
        # capture — a device synchronize — would land in a timed step)
        if (ids.is_cuda and n_seq * K <= _ENC_GRAPH_MAX_ROWS and self._g_recaptures < 4 and _enc_graph_on()
                and not torch.cuda.is_current_stream_capturing()):
            return self._forward_replayed(ids, n_seq, K)
        return self._forward_frozen(ids, n_seq, K)

    def _forward_replayed(self, ids, n_seq, K):
        """The frozen forward through a hipGraph.  At one sample per GPU (BASELINE configs 3 / 4: 512-1,024 rows) an encoder layer is eight launches of
        5-20 us each, issued from Python at ~10 us apiece: the stack was bound by the host (3.5 ms of idle GPU per step at config 3:
        profiles/r05_logs/c3_gaps.log).  Same launches, same buffers, same values: the second consecutive call with one shape (the first sizes scratch and
        sets kernel attributes) is captured, later calls copy the ids into the captured input and replay.
        A changed shape (reserve), another GEMM context or a replaced scratch tensor drops the graph; after four captures the engine stays eager
        (batches whose span length keeps changing would pay a capture — a device synchronize — each time).  MOLLY_ENC_GRAPH=0: always eager.
        Measured (same box, config 3): 198.0 -> 197.0 ms per step; neutral at 16 samples per GPU, where the launches are long."""
        # (the captured launches bake in the weight addresses too: a re-prepared engine — another parameter buffer — is another graph)
        key = (n_seq, K, ops._ctx(), self.wemb.data_ptr())
        ws = ops.current_gemm_scratch()
        if self._g is not None and (self._g_key != key or ws is None or ws.data_ptr() != self._g_ws_ptr):
            self._g, self._g_seen = None, 0
        if self._g is None:
            if self._g_key != key:
                self._g_key, self._g_seen = key, 0
            self._g_seen += 1
            if self._g_seen < 2 or ws is None:
                return self._forward_frozen(ids, n_seq, K)
            self._g_ids = torch.empty_like(ids)
            self._g_ids.copy_(ids)
            g = torch.cuda.CUDAGraph()
            # (bench.py's GEMM accounting lists every launch: the list entries made while capturing — no events are recorded inside a capture —
            # are kept and appended again at every replay, so executed FLOPs stay exact)
            prof, ops.GEMM_PROFILE = ops.GEMM_PROFILE, ([] if ops.GEMM_PROFILE is not None else None)
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self._forward_frozen(self._g_ids, n_seq, K)
                self._g_prof = ops.GEMM_PROFILE or []
            except Exception as e:                              # noqa: BLE001 — an op that is illegal inside a capture, no memory for the graph pool
                # stay eager for good: without this every later call retried the capture and failed again, and the encoder was unusable until
                # MOLLY_ENC_GRAPH=0 was set
                import warnings
                warnings.warn(f"EsmEngine: capturing the frozen forward failed ({type(e).__name__}: {str(e)[:120]}); running eagerly from now on")
                self._g, self._g_seen, self._g_recaptures = None, 0, 4
                ops.GEMM_PROFILE = prof
                return self._forward_frozen(ids, n_seq, K)
            finally:
                ops.GEMM_PROFILE = prof
            assert ops.current_gemm_scratch() is ws, "the GEMM scratch was replaced during graph capture"
            self._g, self._g_ws, self._g_ws_ptr = g, ws, ws.data_ptr()
            self._g_recaptures += 1
        else:
            self._g_ids.copy_(ids)
        self._g.replay()                                        # (a capture does not execute)
        if ops.GEMM_PROFILE is not None:
            ops.GEMM_PROFILE.extend(self._g_prof)
        return self.out

    def _forward_frozen(self, ids, n_seq, K):
        cfg = self.cfg
        ops.esm_embed(ids, self.wemb, self.pemb, self.x, self.pos, self.klen, cfg.pad_token_id, cfg.mask_token_id,
                      cfg.token_dropout)
        x, x2 = self.x, self.x2
        for w in self.layers:
            ops.layernorm_fwd(x, w["ln1_w"], w["ln1_b"], cfg.layer_norm_eps, out=self.ln)
            ops.gemm_nt(self.ln, w["qkv_w"], out=self.qkv, bias=w["qkv_b"])
            # q *= hd^-0.5, then rotary on q,k (positions = arange(K): HF:esm:733-737); absolute models: scale only
            ops.norm_rope_fwd(self.qkv, self.qk, self.nh, self.nh, self.hd, K, None, None, self.cos, self.sin,
                              q_scale=self.hd ** -0.5)
            ops.attn_fwd(self.qk[:, :self.he], self.qk[:, self.he:], self.qkv[:, 2 * self.he:], n_seq, K, self.nh, self.nh,
                         self.hd, 1.0, False, self.klo, self.klen, out=self.att, lse=False)
            ops.gemm_nt(self.att, w["ao_w"], out=x2, bias=w["ao_b"], res=x)
            ops.layernorm_fwd(x2, w["ln2_w"], w["ln2_b"], cfg.layer_norm_eps, out=self.ln)
            ops.gemm_nt(self.ln, w["i_w"], out=self.mid, bias=w["i_b"], gelu=True)
            ops.gemm_nt(self.mid, w["o_w"], out=x, bias=w["o_b"], res=x2)
        ops.layernorm_fwd(x, self.lnf_w, self.lnf_b, cfg.layer_norm_eps, out=self.out)
        return self.out

    # ---- training: same arithmetic, every layer's activations kept; the FFN pre-activation z is stored (bf16, as HF's
    # Linear returns it) and GELU runs as its own pass so the backward can differentiate it ---------------------------
    def _forward_train(self, ids, n_seq, K):
        cfg, he = self.cfg, self.he
        self._ids = ids
        ops.esm_embed(ids, self.wemb, self.pemb, self.A[0]["x"], self.pos, self.klen, cfg.pad_token_id, cfg.mask_token_id,
                      cfg.token_dropout)
        L = len(self.layers)
        for i, w in enumerate(self.layers):
            a = self.A[i]
            ops.layernorm_fwd(a["x"], w["ln1_w"], w["ln1_b"], cfg.layer_norm_eps, out=a["ln1"])
            ops.gemm_nt(a["ln1"], w["qkv_w"], out=a["qkv"], bias=w["qkv_b"])
            ops.norm_rope_fwd(a["qkv"], a["qk"], self.nh, self.nh, self.hd, K, None, None, self.cos, self.sin,
                              q_scale=self.hd ** -0.5)
            ops.attn_fwd(a["qk"][:, :he], a["qk"][:, he:], a["qkv"][:, 2 * he:], n_seq, K, self.nh, self.nh, self.hd, 1.0,
                         False, self.klo, self.klen, out=a["att"], lse=a["lse"])
            ops.gemm_nt(a["att"], w["ao_w"], out=a["x2"], bias=w["ao_b"], res=a["x"])
            ops.layernorm_fwd(a["x2"], w["ln2_w"], w["ln2_b"], cfg.layer_norm_eps, out=a["ln2"])
            ops.gemm_nt(a["ln2"], w["i_w"], out=a["z"], bias=w["i_b"])
            ops.gelu_fwd(a["z"], out=a["mid"])
            nxt = self.A[i + 1]["x"] if i + 1 < L else self.x_last
            ops.gemm_nt(a["mid"], w["o_w"], out=nxt, bias=w["o_b"], res=a["x2"])
        ops.layernorm_fwd(self.x_last, self.lnf_w, self.lnf_b, cfg.layer_norm_eps, out=self.out)
        return self.out

    def _wgrad(self, dy, x, dw, accumulate):
        from .qwen3 import Qwen3Engine
        Qwen3Engine._wgrad(self, dy, x, dw, accumulate)         # same layout choice; uses self.tT

    def backward(self, d_out: torch.Tensor, accumulate: bool = False):
        """d_out [n_seq*K, he] = gradient of the returned hidden states.  Encoder weight gradients land in the flat grad
        buffer (`accumulate`: add, GA micro-steps > 0)."""
        cfg, he = self.cfg, self.he
        n_seq, K, _ = self.cap
        eps = cfg.layer_norm_eps
        dg = lambda dy, w, out: ops.gemm(dy, w, out=out, b_kmajor=True)
        dx = ops.layernorm_bwd(self.x_last, self.lnf_w, d_out.contiguous(), self.d_lnf_w, self.d_lnf_b, eps, dx=self.d_x,
                               dw_accumulate=accumulate, workspace=self.ws)
        dx2b = self.d_x2                                        # dx lives in d_x, the mid-block gradient in d_x2
        for i in reversed(range(len(self.layers))):
            a, w, g = self.A[i], self.layers[i], self.dlayers[i]
            # x_next = x2 + mid W_o^T + b_o
            dg(dx, w["o_w"], self.d_mid)
            self._wgrad(dx, a["mid"], g["o_w"], accumulate)
            ops.colsum(dx, g["o_b"], accumulate=accumulate, workspace=self.ws)
            ops.gelu_bwd(a["z"], self.d_mid, self.d_z)
            dg(self.d_z, w["i_w"], self.d_ln)
            self._wgrad(self.d_z, a["ln2"], g["i_w"], accumulate)
            ops.colsum(self.d_z, g["i_b"], accumulate=accumulate, workspace=self.ws)
            dx2 = ops.layernorm_bwd(a["x2"], w["ln2_w"], self.d_ln, g["ln2_w"], g["ln2_b"], eps, dres=dx, dx=dx2b,
                                    dw_accumulate=accumulate, workspace=self.ws)
            # x2 = x + att W_ao^T + b_ao
            dg(dx2, w["ao_w"], self.d_att)
            self._wgrad(dx2, a["att"], g["ao_w"], accumulate)
            ops.colsum(dx2, g["ao_b"], accumulate=accumulate, workspace=self.ws)
            ops.attn_bwd(a["qk"][:, :he], a["qk"][:, he:], a["qkv"][:, 2 * he:], a["att"], self.d_att, a["lse"], n_seq, K,
                         self.nh, self.nh, self.hd, 1.0, False, self.d_qk[:, :he], self.d_qk[:, he:], self.d_qkv[:, 2 * he:],
                         self.klo, self.klen, delta_ws=self.delta)
            ops.norm_rope_bwd(a["qkv"], self.d_qk, self.d_qkv, self.nh, self.nh, self.hd, K, None, None, self.cos, self.sin,
                              None, None, workspace=self.ws, q_scale=self.hd ** -0.5)
            dg(self.d_qkv, w["qkv_w"], self.d_ln)
            self._wgrad(self.d_qkv, a["ln1"], g["qkv_w"], accumulate)
            ops.colsum(self.d_qkv, g["qkv_b"], accumulate=accumulate, workspace=self.ws)
            dx = ops.layernorm_bwd(a["x"], w["ln1_w"], self.d_ln, g["ln1_w"], g["ln1_b"], eps, dres=dx2, dx=self.d_x,
                                   dw_accumulate=accumulate, workspace=self.ws)
        self._embed_backward(dx, accumulate)

    def _embed_backward(self, dx: torch.Tensor, accumulate: bool):
        """HF:models/esm/modeling_esm.py:224-271 backward: word rows get (token-dropout rescale) x dx of their non-pad,
        non-<mask> positions; absolute position rows get dx of their non-pad positions."""
        cfg = self.cfg
        ids = self._ids.cpu().numpy()
        n_seq, K = ids.shape
        valid = ids != cfg.pad_token_id
        scale = np.ones((n_seq, K), np.float32)
        keep = valid.copy()
        if cfg.token_dropout:
            masked = ids == cfg.mask_token_id
            nv = np.maximum(valid.sum(1), 1).astype(np.float32)
            ratio = masked.sum(1).astype(np.float32) / nv
            scale *= ((1.0 - 0.15 * 0.8) / (1.0 - ratio))[:, None]
            keep &= ~masked
        if not accumulate:
            self.d_wemb.zero_()
            if self.d_pemb is not None:
                self.d_pemb.zero_()

        def scatter(table_ids, sel, dE, row_scale):
            rows = np.nonzero(sel.reshape(-1))[0].astype(np.int32)
            if len(rows) == 0:
                return
            key = table_ids.reshape(-1)[rows]
            order = rows[np.argsort(key, kind="stable")]
            sk = table_ids.reshape(-1)[order]
            bounds = np.nonzero(np.diff(sk))[0] + 1
            seg = np.concatenate([[0], bounds, [len(order)]]).astype(np.int32)
            uid = sk[seg[:-1]].astype(np.int64)
            t = lambda x: torch.from_numpy(x).to(self.dev)
            ops.embed_bwd(dx, t(order), t(seg), t(uid), len(uid), dE, row_scale=row_scale)

        scatter(ids, keep, self.d_wemb, torch.from_numpy(scale.reshape(-1)).to(self.dev) if cfg.token_dropout else None)
        if self.d_pemb is not None:
            pos = np.where(valid, np.cumsum(valid, 1) + cfg.pad_token_id, cfg.pad_token_id)
            scatter(pos, valid, self.d_pemb, None)
