"""Tensor-level wrappers over the C ABI (include/molly_hip.h).  PyTorch is plumbing here: it owns device
memory and the stream; every FLOP runs in libmolly_hip.so.  No wrapper has a torch fallback."""
from __future__ import annotations

from typing import Optional

import os

import torch

from ._lib import lib
from .tracing import roctx  # noqa: F401  (ops.roctx)

BF16 = torch.bfloat16
GEMM_BIAS, GEMM_GELU, GEMM_RESIDUAL, GEMM_ACCUMULATE, GEMM_OUT_F32, GEMM_TRANS_OUT, GEMM_SWIGLU, GEMM_SWIGLU_BWD = 1, 2, 4, 8, 16, 32, 64, 128


# optional timing of the dominant kernel (bench.py roofline): list of (start_event | None, end_event | None, flops, cfg, layout).
# Every launch is listed; HIP events bracket every GEMM_PROFILE_STRIDE-th one only (an event pair costs a few microseconds of
# dispatch per launch: around all ~390 GEMM launches of a step that is 2-3 % of the step being measured).
GEMM_PROFILE = None
GEMM_PROFILE_STRIDE = 1
_prof_n = 0


def _prof_begin():
    """-> start event when this launch is a sampled one, else None (and counts the launch)."""
    global _prof_n
    _prof_n += 1
    if _prof_n % GEMM_PROFILE_STRIDE or torch.cuda.is_current_stream_capturing():     # (no timing events inside a hipGraph capture)
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return e0


def _prof_end(prof, e0, flops, layout):
    e1 = None
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
    prof.append((e0, e1, flops, lib().query("molly_gemm_ctx_get", _ctx(), GEMM_KEYS["last_config"]), layout))


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, dtype=None, name="tensor"):
    if not t.is_cuda:
        raise RuntimeError(f"molly_amd: {name} must live on the GPU (no CPU path exists)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"molly_amd: {name} must be {dtype}, got {t.dtype}")
    if t.stride(-1) != 1:
        raise ValueError(f"molly_amd: {name} must be contiguous in its last dimension")


# ---- GEMM launch state (include/molly_hip.h: molly_gemm_ctx_*).  A GemmContext owns the launch knobs and the scratch memory of
# everything launched while it is current; `with use_gemm_context(ctx):` makes it current.  Outside any context the calling
# thread's default context of the library is used (tools, kernel tests).  A model owns one (OmicsOne.prepare), so an optimizer
# that wants another launch shape at N > 1 changes ITS model's context and nothing else in the process.
GEMM_KEYS = {"persistent_blocks": 1, "schedule": 2, "force_tile": 3, "group_m": 4, "small_grid_tile": 5, "min_ktiles": 6,
             "streamk": 7, "skinny": 8, "small3": 9, "dynamic": 10, "small_split": 11, "rows_tiled": 12, "dynamic_min_work": 13, "rows_max_m": 14, "rows_gu": 15, "rows_bn": 16, "stream_epi": 17, "last_config": 100}
STREAMK_SCRATCH = (64 + 8192 * 64 + 8 * 64) + 256 * 2 * 262144   # header (a counter line per tile) + two 256 KiB accumulator images per block of a 256-block launch


class GemmContext:
    def __init__(self):
        import ctypes
        h = ctypes.c_void_p()
        lib().call("molly_gemm_ctx_create", ctypes.addressof(h))
        self.handle = h.value
        self.ws = None
        import os
        if os.environ.get("MOLLY_GEMM_STREAMK") is not None:       # A/B and bisection knob: 0 off, 1 cost model (default), 2 wherever able
            self.set("streamk", int(os.environ["MOLLY_GEMM_STREAMK"]))
        # test mode: every context launches the 256x256 GEMM the way a rank of a multi-GPU job does (dyn: resident blocks that draw
        # their tiles, the N > 1 default; 0 / -t: round 2's shapes) — `MOLLY_TEST_GEMM_BLOCKS=dyn pytest -m gpu` runs the suite that way
        mode = os.environ.get("MOLLY_TEST_GEMM_BLOCKS")
        if mode:
            self.set("persistent_blocks", 256 if mode == "dyn" else int(mode))
            self.set("dynamic", 1 if mode == "dyn" else 0)
        # A/B knob for whole-step measurements: MOLLY_GEMM_SET="small3=1,group_m=8" applies to every context created afterwards
        for kv in filter(None, os.environ.get("MOLLY_GEMM_SET", "").split(",")):
            k, v = kv.split("=")
            self.set(k.strip(), int(v))

    def set(self, key: str, value: int):
        lib().call("molly_gemm_ctx_set", self.handle, GEMM_KEYS[key], int(value))

    def get(self, key: str) -> int:
        return lib().query("molly_gemm_ctx_get", self.handle, GEMM_KEYS[key])

    def ensure_workspace(self, nbytes: int, device="cuda"):
        nbytes = max(int(nbytes), 0) + STREAMK_SCRATCH
        if self.ws is None or self.ws.numel() * 4 < nbytes:
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                # molly_gemm_ctx_set_workspace clears the header with hipMemset + a device synchronize, and a captured launch
                # would keep the OLD pointer: scratch is sized before a capture starts (GenerationSession._alloc_step)
                raise RuntimeError(f"GEMM scratch would have to grow to {nbytes} bytes during a hipGraph capture; "
                                   "call ensure_workspace() with the largest size before capturing")
            self.ws = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
            lib().call("molly_gemm_ctx_set_workspace", self.handle, self.ws, self.ws.numel() * 4)
        return self.ws

    def streamk_timeouts(self) -> int:
        return lib().query("molly_gemm_ctx_streamk_timeouts", self.handle)

    def __del__(self):
        try:
            if self.handle:
                lib().call("molly_gemm_ctx_destroy", self.handle)
        except Exception:       # noqa: BLE001  (interpreter shutdown)
            pass
        self.handle = None


_CTX_STACK = []


def _ctx():
    """handle of the current GEMM context (None = the calling thread's default context)."""
    return _CTX_STACK[-1].handle if _CTX_STACK else None


class use_gemm_context:
    def __init__(self, ctx: Optional[GemmContext]):
        self.ctx = ctx

    def __enter__(self):
        if self.ctx is not None:
            _CTX_STACK.append(self.ctx)
        return self.ctx

    def __exit__(self, *exc):
        if self.ctx is not None:
            _CTX_STACK.pop()
        return False


_GEMM_WS = None


def ensure_gemm_workspace(nbytes: int = 512 << 20, device="cuda"):
    """Scratch for split-K partials and the stream-K slabs (idempotent; grows only) of the CURRENT context — the model's, inside
    `use_gemm_context`; else the thread's default context."""
    global _GEMM_WS
    if _CTX_STACK:
        return _CTX_STACK[-1].ensure_workspace(nbytes, device)
    nbytes = int(nbytes) + STREAMK_SCRATCH
    if _GEMM_WS is None or _GEMM_WS.numel() * 4 < nbytes:
        _GEMM_WS = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
        lib().call("molly_gemm_set_workspace", _GEMM_WS, _GEMM_WS.numel() * 4)
    return _GEMM_WS


def current_gemm_scratch():
    """The scratch tensor launches of the current context carry pointers into (None: not allocated yet) — what a captured hipGraph must keep alive
    and compare against before every replay."""
    return _CTX_STACK[-1].ws if _CTX_STACK else _GEMM_WS


def gemm_rows_tail_supported(M: int, N: int, K: int, tail: str) -> bool:
    """Would the current context run the decode-row GEMM M x N x K with `tail` ('norm' | 'swiglu') folded into its slab combine?"""
    return bool(lib().query("molly_gemm_rows_tail_supported", _ctx(), M, N, K, {"norm": 1, "swiglu": 2, "qkv": 3}[tail]))


def gemm_rows_qkv(x, w, dst, nq, nk, hd, qw, kw, cos, sin, positions, eps, kcache, vcache, slot, bias=None):
    """Decode rows: q | k | v = x w^T, then q/k-norm + rotary -> dst [M, (nq + nk) hd] and the KV-cache append, in the launch that
    combines the K slices."""
    M, K = x.shape
    N = w.shape[0]
    lib().call("molly_gemm_rows_qkv_bf16_ctx", _ctx(), _stream(), x, w, bias, M, N, K, x.stride(0), w.stride(0), qw, kw, cos, sin,
               positions, float(eps), nq, nk, hd, dst, dst.stride(0), kcache, vcache, slot, kcache.stride(0))
    return dst


def gemm_rows_norm(x, w, out, norm_w, eps, norm_out, res=None, bias=None):
    """out[M,N] = x w^T (+bias) (+res) and norm_out = RMSNorm(out) * norm_w in the launch that combines the K slices (decode rows)."""
    M, K = x.shape
    N = w.shape[0]
    flags = (GEMM_BIAS if bias is not None else 0) | (GEMM_RESIDUAL if res is not None else 0)
    lib().call("molly_gemm_rows_tail_bf16_ctx", _ctx(), _stream(), x, w, out, bias, res, M, N, K, x.stride(0), w.stride(0), out.stride(0),
               res.stride(0) if res is not None else 0, flags, 1, norm_w, float(eps), norm_out, norm_out.stride(0))
    return out, norm_out


def gemm_rows_swiglu(x, w, gu, act, bias=None):
    """gu[M, 2ff] = x w^T (+bias) (gu None: not kept) and act[M, ff] = silu(gate) * up — from the accumulators of a one-slice launch (M <= 32,
    context key rows_gu) or in the launch that combines the K slices (decode rows)."""
    M, K = x.shape
    N = w.shape[0]
    lib().call("molly_gemm_rows_tail_bf16_ctx", _ctx(), _stream(), x, w, gu, bias, None, M, N, K, x.stride(0), w.stride(0),
               gu.stride(0) if gu is not None else 0, 0, GEMM_BIAS if bias is not None else 0, 2, None, 0.0, act, act.stride(0))
    return gu, act


def gemm_nt(a, b, out=None, bias=None, res=None, gelu=False, accumulate=False, out_dtype=BF16):
    """out[M,N] = a[M,K] @ b[N,K]^T (+bias) (gelu) (+res) (+= out).  2-D views with arbitrary row stride."""
    return gemm(a, b, out, bias, res, gelu, accumulate, out_dtype, False, False)


def gemm_gate_up_swiglu(x, w_gu, gu, act):
    """gu[M, 2ff] = x @ w_gu^T (w_gu = [gate_proj | up_proj] rows) and act[M, ff] = silu(gate) * up from ONE launch
    (MOLLY_GEMM_SWIGLU): the activation is computed in the GEMM's epilogue from the accumulators of the tile."""
    _chk(x, BF16, "x"); _chk(w_gu, BF16, "w_gu"); _chk(gu, BF16, "gu"); _chk(act, BF16, "act")
    M, K = x.shape
    N = w_gu.shape[0]
    assert w_gu.shape[1] == K and tuple(gu.shape) == (M, N) and tuple(act.shape) == (M, N // 2) and N % 256 == 0
    prof = GEMM_PROFILE
    e0 = _prof_begin() if prof is not None else None
    lib().call("molly_gemm_bf16_ctx", _ctx(), _stream(), x, w_gu, gu, None, act, M, N, K, x.stride(0), w_gu.stride(0), gu.stride(0),
               act.stride(0), GEMM_SWIGLU, 0, 0)
    if prof is not None:
        _prof_end(prof, e0, 2.0 * M * N * K, (False, False))
    return act


def lora_pack_items(pairs):
    """Device table for lora_pack_b from [(src [rows, 64] contiguous bf16, dst view [rows, 64] with a row stride)]; keep the tensors alive."""
    import struct
    raw = b"".join(struct.pack("<QQii", s.data_ptr(), d.data_ptr(), s.shape[0], d.stride(0)) for s, d in pairs)
    for s, d in pairs:
        assert s.is_contiguous() and s.shape[1] == 64 and d.stride(1) == 1 and d.stride(0) % 8 == 0
        assert tuple(d.shape) in (tuple(s.shape), (64, s.shape[0]))          # (the transposing form: dst [64, rows])
    t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(pairs[0][0].device)
    return t, len(pairs), max(s.shape[0] for s, _ in pairs)


def lora_pack_bt(table):
    t, n, max_rows = table
    lib().call("molly_lora_pack_bt", _stream(), t, n, max_rows)


def lora_pack_b(table):
    t, n, max_rows = table
    lib().call("molly_lora_pack_b", _stream(), t, n, max_rows)


def gemm_kx_supported(M: int, N: int, K: int, K2: int, swiglu: bool = False, res: bool = False, bias: bool = False) -> bool:
    flags = (GEMM_SWIGLU if swiglu else 0) | (GEMM_RESIDUAL if res else 0) | (GEMM_BIAS if bias else 0)
    return bool(lib().query("molly_gemm_kx_supported", _ctx(), M, N, K, K2, flags))


def gemm_nt_kx(a, b, a2, b2, out, bias=None, res=None, act=None):
    """out[M, N] = a[M, K] @ b[N, K]^T + a2[M, K2] @ b2[N, K2]^T (+ bias) (+ res) in one accumulation (molly_gemm_kx_bf16_ctx): a LoRA branch's
    t B_lora^T riding in the base projection as K2 / 64 more K-tiles.  `act` ([M, N / 2]): the gate|up form — out = [gate | up] and
    act = silu(gate) * up from the epilogue (MOLLY_GEMM_SWIGLU).  Ask gemm_kx_supported first."""
    _chk(a, BF16, "a"); _chk(b, BF16, "b"); _chk(a2, BF16, "a2"); _chk(b2, BF16, "b2"); _chk(out, BF16, "out")
    M, K = a.shape
    N, K2 = b2.shape
    assert tuple(b.shape) == (N, K) and tuple(a2.shape) == (M, K2) and tuple(out.shape) == (M, N), (a.shape, b.shape, a2.shape, b2.shape, out.shape)
    flags, r = 0, res
    if act is not None:
        assert bias is None and res is None and tuple(act.shape) == (M, N // 2)
        _chk(act, BF16, "act")
        flags, r = GEMM_SWIGLU, act
    else:
        if bias is not None:
            _chk(bias, BF16, "bias"); flags |= GEMM_BIAS
        if res is not None:
            _chk(res, BF16, "res"); flags |= GEMM_RESIDUAL
    prof = GEMM_PROFILE
    e0 = _prof_begin() if prof is not None else None
    lib().call("molly_gemm_kx_bf16_ctx", _ctx(), _stream(), a, b, out, bias, r, M, N, K, a.stride(0), b.stride(0), out.stride(0),
               r.stride(0) if r is not None else 0, flags, a2, b2, K2, a2.stride(0), b2.stride(0))
    if prof is not None:
        _prof_end(prof, e0, 2.0 * M * N * (K + K2), (False, False))
    return out


def gemm_down_dgrad_swiglu_bwd(dy, w_down, gu, dgu):
    """d[gate | up][M, 2ff] from ONE launch (MOLLY_GEMM_SWIGLU_BWD): d(act) = dy[M, h] @ w_down[h, ff] stays in the accumulators,
    the SwiGLU backward runs in the epilogue against gu = [gate | up] — bit-identical to gemm(dy, w_down, b_kmajor=True) followed by
    swiglu_bwd, without d(act)'s HBM round trip."""
    _chk(dy, BF16, "dy"); _chk(w_down, BF16, "w_down"); _chk(gu, BF16, "gu"); _chk(dgu, BF16, "dgu")
    M, K = dy.shape
    ff = w_down.shape[1]
    assert w_down.shape[0] == K and tuple(gu.shape) == (M, 2 * ff) and tuple(dgu.shape) == (M, 2 * ff)
    prof = GEMM_PROFILE
    e0 = _prof_begin() if prof is not None else None
    lib().call("molly_gemm_bf16_ctx", _ctx(), _stream(), dy, w_down, dgu, None, gu, M, ff, K, dy.stride(0), w_down.stride(0),
               dgu.stride(0), gu.stride(0), GEMM_SWIGLU_BWD, 0, 1)
    if prof is not None:
        _prof_end(prof, e0, 2.0 * M * ff * K, (False, True))
    return dgu


def gemm(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
         res: Optional[torch.Tensor] = None, gelu: bool = False, accumulate: bool = False, out_dtype=BF16,
         a_kmajor: bool = False, b_kmajor: bool = False, trans_out: bool = False) -> torch.Tensor:
    """out[M,N] = op(a) @ op(b): a is [M,K] (or [K,M] when a_kmajor), b is [N,K] (or [K,N] when b_kmajor).
    trans_out: the result is stored transposed, `out` is [N, M]."""
    _chk(a, BF16, "a"); _chk(b, BF16, "b")
    K, M = (a.shape if a_kmajor else a.shape[::-1])
    K2, N = (b.shape if b_kmajor else b.shape[::-1])
    assert K == K2, (a.shape, b.shape, a_kmajor, b_kmajor)
    if out is None:
        assert not accumulate
        out = torch.empty((N, M) if trans_out else (M, N), dtype=out_dtype, device=a.device)
    _chk(out, None, "out")
    flags = GEMM_TRANS_OUT if trans_out else 0
    if bias is not None:
        _chk(bias, BF16, "bias"); flags |= GEMM_BIAS
    if gelu:
        flags |= GEMM_GELU
    if res is not None:
        _chk(res, BF16, "res"); flags |= GEMM_RESIDUAL
    if accumulate:
        flags |= GEMM_ACCUMULATE
    if out.dtype == torch.float32:
        flags |= GEMM_OUT_F32
    elif out.dtype != BF16:
        raise TypeError("gemm_nt: out must be bf16 or fp32")
    prof = GEMM_PROFILE
    e0 = _prof_begin() if prof is not None else None
    lib().call("molly_gemm_bf16_ctx", _ctx(), _stream(), a, b, out, bias, res, M, N, K, a.stride(0), b.stride(0), out.stride(0),
               res.stride(0) if res is not None else 0, flags, int(a_kmajor), int(b_kmajor))
    if prof is not None:
        _prof_end(prof, e0, 2.0 * M * N * K, (a_kmajor, b_kmajor))
    return out


def gemm_grouped(problems, accumulate: bool = False):
    """Up to 16 GEMMs out_i (+)= a_i @ b_i in ONE persistent launch (molly_gemm_grouped_bf16): a_i [M_i, K] k-contiguous, b_i [K, N_i]
    k-major, all sharing K; `problems` = [(a, b, out, trans_out)], trans_out: out_i is [N_i, M_i].  No split-K, no reduce."""
    assert 1 <= len(problems) <= 16
    K = problems[0][0].shape[1]
    desc = torch.empty(len(problems), 6, dtype=torch.int64)
    flags = GEMM_ACCUMULATE if accumulate else 0
    f32 = problems[0][2].dtype == torch.float32
    for i, (a, b, out, to) in enumerate(problems):
        _chk(a, BF16, "a"); _chk(b, BF16, "b"); _chk(out, None, "out")
        M, N = a.shape[0], b.shape[1]
        assert a.shape[1] == K and b.shape[0] == K and tuple(out.shape) == ((N, M) if to else (M, N)), (a.shape, b.shape, out.shape, to)
        assert (out.dtype == torch.float32) == f32
        desc[i, 0], desc[i, 1], desc[i, 2] = a.data_ptr(), b.data_ptr(), out.data_ptr()
        desc[i, 3] = M | (N << 32)
        desc[i, 4] = a.stride(0) | (b.stride(0) << 32)
        desc[i, 5] = out.stride(0) | (int(bool(to)) << 32)
    if f32:
        flags |= GEMM_OUT_F32
    prof = GEMM_PROFILE
    e0 = _prof_begin() if prof is not None else None
    lib().call("molly_gemm_grouped_bf16_ctx", _ctx(), _stream(), desc.data_ptr(), len(problems), K, flags)
    if prof is not None:
        _prof_end(prof, e0, sum(2.0 * a.shape[0] * b.shape[1] * K for a, b, _, _ in problems), (False, True))


def transpose(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(x, BF16, "x")
    R, C = x.shape
    if out is None:
        out = torch.empty((C, R), dtype=BF16, device=x.device)
    lib().call("molly_transpose_bf16", _stream(), x, out, R, C, x.stride(0), out.stride(0))
    return out


def rmsnorm_fwd_t_supported(rows: int, H: int) -> bool:
    return rows % 64 == 0 and H % 512 == 0 and H <= 2048


def rmsnorm_fwd(x, w, eps, out=None, out_t=None):
    """out_t (optional, [H, rows] contiguous): the result a second time, transposed (molly_rmsnorm_fwd_t) — what the weight gradient of
    the projection behind the norm reads as its k-contiguous operand."""
    _chk(x, BF16, "x")
    rows, H = x.shape
    assert x.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    if out_t is not None:
        assert out_t.dtype == BF16 and out_t.shape == (H, rows) and out_t.stride(1) == 1 and rmsnorm_fwd_t_supported(rows, H)
        lib().call("molly_rmsnorm_fwd_t", _stream(), x, w, out, out_t, rows, H, out_t.stride(0), float(eps))
        return out
    lib().call("molly_rmsnorm_fwd", _stream(), x, w, out, None, rows, H, float(eps))
    return out


def rmsnorm_bwd(x, w, g, dw, eps, dres=None, dx=None, dw_accumulate=True, workspace=None):
    rows, H = x.shape
    assert x.is_contiguous() and g.is_contiguous()
    if dx is None:
        dx = torch.empty_like(x)
    nb = lib().query("molly_rmsnorm_bwd_blocks", rows)
    if workspace is None:
        workspace = torch.empty(nb * H, dtype=torch.float32, device=x.device)
    assert workspace.numel() >= nb * H
    lib().call("molly_rmsnorm_bwd", _stream(), x, w, g, dres, dx, dw, int(dw is not None and dw.dtype == torch.float32),
               int(dw_accumulate), workspace, rows, H, float(eps))
    return dx


def norm_rope_fwd(src, dst, nq, nk, hd, T, qw=None, kw=None, cos=None, sin=None, positions=None, eps=1e-6, q_scale=1.0,
                  kcache=None, vcache=None, slot=None):
    """kcache / vcache ([rows, nk * hd]) + slot (int32 [M]): the decode step's cache append in the same launch — the k heads after
    norm + rotary and the v heads of src row m go to cache row slot[m]."""
    M = src.shape[0]
    if kcache is not None:
        lib().call("molly_norm_rope_cache_fwd", _stream(), src, dst, qw, kw, cos, sin, positions, M, T, nq, nk, hd,
                   src.stride(0), dst.stride(0), float(eps), float(q_scale), kcache, vcache, slot, kcache.stride(0))
        return dst
    lib().call("molly_norm_rope_fwd", _stream(), src, dst, qw, kw, cos, sin, positions, M, T, nq, nk, hd,
               src.stride(0), dst.stride(0), float(eps), float(q_scale))
    return dst


def norm_rope_bwd(src, g, dsrc, nq, nk, hd, T, qw, kw, cos, sin, dqw, dkw, positions=None, eps=1e-6,
                  dw_accumulate=True, workspace=None, q_scale=1.0):
    M = src.shape[0]
    nb = lib().query("molly_norm_rope_bwd_blocks")
    if workspace is None:
        workspace = torch.empty(nb * 2 * hd, dtype=torch.float32, device=src.device)
    lib().call("molly_norm_rope_bwd", _stream(), src, g, dsrc, qw, kw, cos, sin, positions, dqw, dkw,
               int(dqw.dtype == torch.float32) if dqw is not None else 0, int(dw_accumulate), workspace, M, T, nq, nk, hd,
               src.stride(0), g.stride(0), dsrc.stride(0), float(eps), float(q_scale))
    return dsrc


def colsum_items(entries, device):
    """Device table for colsum_batched: entries = [(partials fp32 tensor, out tensor, nb, H, row_stride)]."""
    import struct
    raw = b"".join(struct.pack("<QQiiii", part.data_ptr(), out.data_ptr(), nb, H, rs, 0) for part, out, nb, H, rs in entries)
    t = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
    return t, len(entries), max(e[3] for e in entries)


def colsum_batched(table, out_f32=False, accumulate=False):
    t, n, max_h = table
    lib().call("molly_colsum_batched", _stream(), t, n, max_h, int(out_f32), int(accumulate))


def swiglu_fwd(gu, out=None):
    rows, ff2 = gu.shape
    assert gu.is_contiguous()
    if out is None:
        out = torch.empty((rows, ff2 // 2), dtype=BF16, device=gu.device)
    lib().call("molly_swiglu_fwd", _stream(), gu, out, rows, ff2 // 2)
    return out


def swiglu_bwd(gu, dout, dgu=None):
    rows, ff2 = gu.shape
    if dgu is None:
        dgu = torch.empty_like(gu)
    lib().call("molly_swiglu_bwd", _stream(), gu, dout, dgu, rows, ff2 // 2)
    return dgu


def copy_rows(src, dst, n, src_idx64=None, src_idx32=None, dst_idx32=None, accumulate=False):
    H = src.shape[-1]
    lib().call("molly_copy_rows", _stream(), src, src_idx64, src_idx32, dst, dst_idx32, n, H, src.stride(-2),
               dst.stride(-2), int(accumulate))
    return dst


def layernorm_fwd(x, w, b, eps, out=None):
    rows, H = x.shape
    assert x.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    lib().call("molly_layernorm_fwd", _stream(), x, w, b, out, rows, H, float(eps))
    return out


def layernorm_bwd(x, w, g, dw, db, eps, dres=None, dx=None, dw_accumulate=False, workspace=None):
    rows, H = x.shape
    assert x.is_contiguous() and g.is_contiguous()
    if dx is None:
        dx = torch.empty_like(x)
    nb = lib().query("molly_layernorm_bwd_blocks", rows)
    if workspace is None:
        workspace = torch.empty(2 * nb * H, dtype=torch.float32, device=x.device)
    assert workspace.numel() >= 2 * nb * H
    lib().call("molly_layernorm_bwd", _stream(), x, w, g, dres, dx, dw, db, int(dw.dtype == torch.float32), int(dw_accumulate),
               workspace, rows, H, float(eps))
    return dx


def gelu_fwd(z, out=None):
    _chk(z, BF16, "z")
    assert z.is_contiguous()
    if out is None:
        out = torch.empty_like(z)
    lib().call("molly_gelu_fwd", _stream(), z, out, z.numel())
    return out


def gelu_bwd(z, dout, dz=None):
    assert z.is_contiguous() and dout.is_contiguous()
    if dz is None:
        dz = torch.empty_like(z)
    lib().call("molly_gelu_bwd", _stream(), z, dout, dz, z.numel())
    return dz


def attn_fwd(q, k, v, B, T, nh, nkv, hd, scale, causal, kv_lo=None, kv_hi=None, out=None, lse=None, out_t=None):
    """q/k/v: 2-D views [B*T, *] whose row holds the heads of one token (head h at column h*hd).
    out_t (optional, [nh * hd, B * T] contiguous, T % 128 == 0): the output a second time, transposed (molly_attn_fwd_ot)."""
    if out is None:
        out = torch.empty((B * T, nh * hd), dtype=BF16, device=q.device)
    if lse is None:
        lse = torch.empty((B, nh, T), dtype=torch.float32, device=q.device)
    elif lse is False:          # caller does not need the log-sum-exp (forward-only encoders)
        lse = None
    if out_t is not None:
        assert out_t.dtype == BF16 and out_t.shape == (nh * hd, B * T) and out_t.stride(1) == 1
        lib().call("molly_attn_fwd_ot", _stream(), q, k, v, out, out_t, lse, kv_lo, kv_hi, B, T, nh, nkv, hd, q.stride(0), k.stride(0),
                   v.stride(0), out.stride(0), out_t.stride(0), float(scale), int(causal))
        return out, lse
    lib().call("molly_attn_fwd", _stream(), q, k, v, out, lse, kv_lo, kv_hi, B, T, nh, nkv, hd, q.stride(0), k.stride(0),
               v.stride(0), out.stride(0), float(scale), int(causal))
    return out, lse


def attn_bwd_workspace(B, T, nh, nkv, hd) -> int:
    """fp32 elements of scratch with which `attn_bwd(ws=...)` splits the dK / dV passes by query head (0: not at these sizes)."""
    return lib().query("molly_attn_bwd_workspace", B, T, nh, nkv, hd)


def attn_bwd(q, k, v, o, do, lse, B, T, nh, nkv, hd, scale, causal, dq, dk, dv, kv_lo=None, kv_hi=None, delta_ws=None, ws=None):
    if delta_ws is None:
        delta_ws = torch.empty((B, nh, T), dtype=torch.float32, device=q.device)
    lib().call("molly_attn_bwd_ws", _stream(), q, k, v, o, do, lse, delta_ws, dq, dk, dv, kv_lo, kv_hi, B, T, nh, nkv, hd,
               q.stride(0), k.stride(0), v.stride(0), o.stride(0), do.stride(0), dq.stride(0), dk.stride(0), dv.stride(0),
               float(scale), int(causal), ws, ws.numel() if ws is not None else 0)
    return dq, dk, dv


def attn_bwd_rope_blocks(B, T, nh, nkv):
    """(rows of the q gain-gradient partials, rows of the k ones) `attn_bwd_rope` writes: one per workgroup of the dQ / dK kernels."""
    return lib().query("molly_attn_bwd_rope_blocks", B, T, nh, nkv, 0), lib().query("molly_attn_bwd_rope_blocks", B, T, nh, nkv, 1)


def attn_bwd_rope(q, k, v, o, do, lse, B, T, nh, nkv, hd, scale, causal, dv, x_saved, qw, kw, cos, sin, eps, dx, dwq_part, dwk_part,
                  kv_lo=None, kv_hi=None, delta_ws=None):
    """attn_bwd + norm_rope_bwd in one pass over dq / dk (round 6): q, k = the normed + rotated rows the forward saved, x_saved = the PRE-norm
    q | k | v projection rows, dx = the gradient of that projection (q and k parts written here; dv as before).  dwq_part [nbq, hd] / dwk_part
    [nbk, hd] fp32 (attn_bwd_rope_blocks) receive one row of gain-gradient partials per workgroup.  Head dim 128, attn_bwd_workspace(..) == 0."""
    if delta_ws is None:
        delta_ws = torch.empty((B, nh, T), dtype=torch.float32, device=q.device)
    lib().call("molly_attn_bwd_rope", _stream(), q, k, v, o, do, lse, delta_ws, dv, kv_lo, kv_hi, B, T, nh, nkv, hd, q.stride(0), k.stride(0),
               v.stride(0), o.stride(0), do.stride(0), dv.stride(0), float(scale), int(causal), x_saved, x_saved.stride(0), qw, kw, cos, sin,
               float(eps), dx, dx.stride(0), dwq_part, dwk_part)
    return dx


def ce_fwd_bwd(logits, labels, row_loss, scale, ignore_index=-100, write_grad=True):
    rows, V = logits.shape
    lib().call("molly_ce_fwd_bwd", _stream(), logits, labels, row_loss, scale, rows, V, logits.stride(0), ignore_index,
               int(write_grad))


def cls_loss_fwd_bwd(logits, V, row_loss, scale, labels=None, targets=None, ignore_index=-100, write_grad=True):
    """Enc-Head baseline losses on logits [rows, ld >= V] (bf16), in place -> d(logits): labels (int64) = cross entropy,
    targets (fp32 [rows, V]) = BCE with logits."""
    rows, ld = logits.shape
    assert logits.stride(1) == 1 and (labels is None) != (targets is None)
    lib().call("molly_cls_loss_fwd_bwd", _stream(), logits, labels, targets, row_loss, scale, rows, V, logits.stride(0),
               0 if labels is not None else 1, ignore_index, int(write_grad))


_ARGMAX_WS = {}


def argmax(logits: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Row-wise argmax of fp32 logits [rows, V] (first maximal index, like torch.argmax) -> int64 [rows]."""
    assert logits.dtype == torch.float32 and logits.stride(1) == 1
    rows, V = logits.shape
    if out is None:
        out = torch.empty(rows, dtype=torch.int64, device=logits.device)
    key = (logits.device, rows, _stream())           # (per stream: two streams' launches must not share the partials)
    ws = _ARGMAX_WS.get(key)
    if ws is None:
        ws = _ARGMAX_WS[key] = torch.empty(lib().query("molly_argmax_workspace", rows), dtype=torch.uint8, device=logits.device)
    lib().call("molly_argmax_f32_ws", _stream(), logits, out, rows, V, logits.stride(0), ws, ws.numel())
    return out


def sample_logits(logits, generated, repetition_penalty, temperature, top_k, top_p, seed, step, out=None, debug_cap=0):
    """HF's sampling processors + one draw per row on the device (molly_sample_logits).  logits fp32 [rows, V] (modified in
    place by the repetition penalty); generated int64 [rows, n] or None.  -> next tokens int64 [rows]
    (+ (probs, ids, n_kept) of the surviving tokens when debug_cap > 0)."""
    assert logits.dtype == torch.float32 and logits.stride(1) == 1
    rows, V = logits.shape
    if out is None:
        out = torch.empty(rows, dtype=torch.int64, device=logits.device)
    n_gen = 0 if generated is None else generated.shape[1]
    if n_gen:
        assert generated.dtype == torch.int64 and generated.stride(1) == 1
    dbg = None
    if debug_cap:
        dbg = (torch.zeros(rows, debug_cap, dtype=torch.float32, device=logits.device),
               torch.full((rows, debug_cap), -1, dtype=torch.int64, device=logits.device),
               torch.zeros(rows, dtype=torch.int32, device=logits.device))
    lib().call("molly_sample_logits", _stream(), logits, rows, V, logits.stride(0), generated if n_gen else None, n_gen,
               generated.stride(0) if n_gen else 0, float(repetition_penalty or 1.0), float(temperature or 1.0), int(top_k),
               float(top_p if top_p is not None else 1.0), int(seed) & ((1 << 64) - 1), int(step), out,
               dbg[0] if dbg else None, dbg[1] if dbg else None, dbg[2] if dbg else None, debug_cap)
    return (out, dbg) if debug_cap else out


def count_valid(labels, scale_out, count_out, ignore_index=-100):
    lib().call("molly_count_valid", _stream(), labels, labels.numel(), ignore_index, scale_out, count_out)


def sum_f32(x, out, scale=None, accumulate=False):
    lib().call("molly_sum_f32", _stream(), x, x.numel(), scale, out, int(accumulate))


def esm_embed(ids, word_emb, pos_emb, out, pos_ids, kv_len, pad_id, mask_id, token_dropout):
    n_seq, K = ids.shape
    H = word_emb.shape[1]
    lib().call("molly_esm_embed", _stream(), ids, word_emb, pos_emb, out, pos_ids, kv_len, n_seq, K, H, pad_id, mask_id,
               int(token_dropout))
    return out


def embed_bwd(g, order, seg_start, uid, n_unique, dE, row_scale=None, n_unique_dev=None):
    """dE[uid[u]] += sum over k in [seg_start[u], seg_start[u+1]) of row_scale[order[k]] * g[order[k]].
    n_unique_dev: device int holding the true segment count (n_unique is then the launch bound)."""
    lib().call("molly_embed_bwd", _stream(), g, order, seg_start, uid, n_unique, dE, dE.shape[1], g.stride(0), row_scale,
               n_unique_dev)


def reduce_rows(x2d, out):
    """out[i] = bf16(sum_r x2d[r, i]) in fp32, fixed row order (the local reduction of the all-to-all reduce-scatter)."""
    _chk(x2d, BF16, "x2d"); _chk(out, BF16, "out")
    assert x2d.is_contiguous() and out.is_contiguous() and x2d.shape[1] == out.numel()
    lib().call("molly_reduce_rows_bf16", _stream(), x2d, x2d.shape[0], x2d.shape[1], out)
    return out


def sqnorm(g, out, workspace, accumulate=False):
    lib().call("molly_sqnorm_bf16", _stream(), g, g.numel(), workspace, out, int(accumulate))


def clip_coef(norm_sq, max_norm, pre_scale, norm_out, coef_out, skipped=None):
    lib().call("molly_clip_coef", _stream(), norm_sq, float(max_norm), float(pre_scale), norm_out, coef_out, skipped)


def adamw_step(master, m, v, grad, param_out, lr, beta1, beta2, eps, wd, step, grad_scale=None, skipped=None):
    lib().call("molly_adamw_step", _stream(), master, m, v, grad, param_out, master.numel(), float(lr), float(beta1),
               float(beta2), float(eps), float(wd), int(step), grad_scale, skipped)


def cast_f32_to_bf16(x, out):
    lib().call("molly_cast_f32_to_bf16", _stream(), x, out, x.numel())
    return out


def colsum(x, out, accumulate=False, workspace=None):
    rows, H = x.shape
    npart = lib().query("molly_colsum_parts", rows)
    if workspace is None:
        workspace = torch.empty(npart * H, dtype=torch.float32, device=x.device)
    lib().call("molly_colsum_bf16", _stream(), x, rows, H, x.stride(0), out, int(out.dtype == torch.float32),
               int(accumulate), workspace)
    return out


def attn_decode_workspace(B, nh, hd, device):
    return torch.empty(lib().query("molly_attn_decode_workspace", B, nh, hd), dtype=torch.float32, device=device)


def attn_decode(q, kcache, vcache, out, kv_lo, kv_hi, B, Tmax, nh, nkv, hd, scale, kv_len_hint=0, workspace=None):
    lib().call("molly_attn_decode", _stream(), q, kcache, vcache, out, kv_lo, kv_hi, B, Tmax, nh, nkv, hd, q.stride(0),
               float(scale), int(kv_len_hint), workspace, workspace.numel() if workspace is not None else 0)
    return out


def gemm_rows_slabs(x, w):
    """Decode rows: x w^T left as K-slice slabs in the context's scratch -> (device address of float [n][M][N], n).  Valid until the
    context's next launch; consumed by attn_decode_qkv."""
    import ctypes
    M, K = x.shape
    N = w.shape[0]
    out = (ctypes.c_long * 2)()
    lib().call("molly_gemm_rows_slabs_bf16_ctx", _ctx(), _stream(), x, w, M, N, K, x.stride(0), w.stride(0), ctypes.addressof(out))
    return int(out[0]), int(out[1])


def attn_decode_qkv(slabs, n_slabs, qw, kw, cos, sin, positions, eps, kcache, vcache, slot, out, kv_lo, kv_hi, B, Tmax, nh, nkv, hd, scale,
                    kv_len_hint=0, workspace=None):
    """The decode step's attention from the q | k | v projection's K-slice slabs (a device address from gemm_rows_slabs, or a float32
    tensor [n][B][(nh + 2 nkv) hd]): slab combine + q/k-norm + rotary + KV-cache append + attention (+ the new key) in one launch."""
    lib().call("molly_attn_decode_qkv", _stream(), slabs, int(n_slabs), qw, kw, cos, sin, positions, float(eps), kcache, vcache, slot, out,
               kv_lo, kv_hi, B, Tmax, nh, nkv, hd, float(scale), int(kv_len_hint), workspace,
               workspace.numel() if workspace is not None else 0)
    return out


def dropout(x, p: float, seed: int, out=None, accumulate=False):
    """out (+)= x * keep / (1-p), keep a pure function of (seed, element index)."""
    _chk(x, BF16, "x")
    assert x.is_contiguous()
    if out is None:
        assert not accumulate
        out = torch.empty_like(x)
    assert out.is_contiguous() and out.numel() == x.numel()
    lib().call("molly_dropout_bf16", _stream(), x, out, x.numel(), float(p), int(seed) & ((1 << 64) - 1), int(accumulate))
    return out


def lora_down_drop(x, A, p: float, seed: int, scale: float = 1.0, xd=None, out=None, out_t=None):
    """t[M, R] = scale * (dropout(x) A^T) in one launch (molly_lora_down_drop_bf16); xd (optional, [M, K]) receives dropout(x).
    The mask is molly_dropout_bf16's function of (seed, flat element index)."""
    _chk(x, BF16, "x"); _chk(A, BF16, "A")
    assert x.stride(1) == 1 and A.is_contiguous() and x.shape[1] == A.shape[1]
    M, K = x.shape
    R = A.shape[0]
    if out is None:
        out = torch.empty(M, R, dtype=BF16, device=x.device)
    assert out.shape == (M, R) and out.stride(1) == 1
    if xd is not None:
        assert xd.is_contiguous() and xd.shape == x.shape and xd.dtype == BF16
    if out_t is not None:                                   # t^T [R, M] as well (the adapter weight gradients' operand)
        assert out_t.dtype == BF16 and tuple(out_t.shape) == (R, M) and out_t.stride(1) == 1
    lib().call("molly_lora_down_drop_t_bf16", _stream(), x, A, xd, out, M, K, R, x.stride(0), out.stride(0), float(p),
               int(seed) & ((1 << 64) - 1), float(scale), out_t, out_t.stride(0) if out_t is not None else 0)
    return out


def lora_up_drop_acc_multi(dts, As, dx, p: float, seeds):
    """dx[M, K] += sum over n <= 3 targets of mask_u * bf16(dt_u A_u), one read and one write of dx, the roundings of n lora_up_drop_acc launches one
    after the other (molly_lora_up_drop_acc_multi_bf16): q | k | v (one input), gate | up."""
    import ctypes
    n = len(dts)
    assert 1 <= n <= 3 and len(As) == n and len(seeds) == n
    _chk(dx, BF16, "dx")
    assert dx.is_contiguous()
    M, K = dx.shape
    for dt, A in zip(dts, As):
        _chk(dt, BF16, "dt"); _chk(A, BF16, "A")
        assert A.is_contiguous() and dt.stride(1) == 1 and tuple(dt.shape) == (M, A.shape[0]) and A.shape[1] == K and A.shape[0] == As[0].shape[0]
    vp = (ctypes.c_void_p * n)(*[t.data_ptr() for t in dts])
    ap = (ctypes.c_void_p * n)(*[t.data_ptr() for t in As])
    ld = (ctypes.c_int * n)(*[t.stride(0) for t in dts])
    sd = (ctypes.c_uint64 * n)(*[int(x) & ((1 << 64) - 1) for x in seeds])
    lib().call("molly_lora_up_drop_acc_multi_bf16", _stream(), n, vp, ap, dx, M, K, As[0].shape[0], ld, float(p), sd)
    return dx


def lora_up_drop_acc(dt, A, dx, p: float, seed: int):
    """dx[M, K] += mask * bf16(dt[M, R] A[R, K]) in one launch (molly_lora_up_drop_acc_bf16): the LoRA branch's input gradient."""
    _chk(dt, BF16, "dt"); _chk(A, BF16, "A"); _chk(dx, BF16, "dx")
    assert A.is_contiguous() and dx.is_contiguous() and dt.stride(1) == 1
    M, R = dt.shape
    K = A.shape[1]
    assert A.shape[0] == R and dx.shape == (M, K)
    lib().call("molly_lora_up_drop_acc_bf16", _stream(), dt, A, dx, M, K, R, dt.stride(0), float(p), int(seed) & ((1 << 64) - 1))
    return dx


def scale_(x, s: float):
    _chk(x, BF16, "x")
    assert x.is_contiguous()
    lib().call("molly_scale_bf16", _stream(), x, x.numel(), float(s))
    return x
