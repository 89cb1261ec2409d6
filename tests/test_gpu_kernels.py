"""GPU: each HIP kernel (through the C ABI) against a plain fp32 PyTorch statement of the same op on the same
seeded inputs.  Tolerances are bf16-output tolerances and are written next to each check."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from molly_amd import ops
    from molly_amd._lib import lib

DEV = "cuda"
BF = torch.bfloat16


def _rand(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def _close(got, ref, atol, rtol, what=""):
    got, ref = got.float(), ref.float()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off; max err {err.max().item():.4g} " \
                          f"(ref absmax {ref.abs().max().item():.4g})"


# ------------------------------------------------------------------------------------------------
def test_probe_mfma_layout_exact_integers():
    """A = asymmetric small ints, B = other small ints: D must equal A @ B^T exactly (pins the operand/C maps)."""
    A = torch.randint(-3, 4, (16, 32), generator=torch.Generator().manual_seed(1)).float()
    B = torch.randint(-3, 4, (16, 32), generator=torch.Generator().manual_seed(2)).float()
    D = torch.empty(16, 16, dtype=torch.float32, device=DEV)
    lib().call("molly_probe_mfma16", 0, A.to(DEV, BF), B.to(DEV, BF), D)
    torch.cuda.synchronize()
    assert torch.equal(D.cpu(), A @ B.T)


@pytest.mark.parametrize("stride", [16, 64, 128])
def test_probe_tr16_semantics(stride):
    """ds_read_b64_tr_b16: lane i of 16-lane group g must receive column i of rows 4g..4g+3 (element q = row q)."""
    tile = torch.arange(16 * stride, dtype=torch.int32).reshape(16, stride)
    out = torch.empty(64, 4, dtype=torch.int16, device=DEV)
    lib().call("molly_probe_tr16", 0, tile.to(torch.int16).to(DEV), out, stride)
    torch.cuda.synchronize()
    out = out.cpu().to(torch.int32)
    for lane in range(64):
        g, i = lane // 16, lane % 16
        for q in range(4):
            assert out[lane, q].item() == tile[4 * g + q, i].item(), (lane, q, out[lane].tolist())


# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 512), (300, 200, 128), (2048, 1024, 2048), (64, 1024, 256),
                                   (1, 128, 64)])
def test_gemm_nt_plain(M, N, K):
    a = _rand(M, K, seed=1).to(BF)
    b = _rand(N, K, seed=2).to(BF)
    out = ops.gemm_nt(a, b)
    ref = a.float() @ b.float().T
    # bf16 output: half-ulp 2^-9 relative + fp32 accumulation-order noise
    _close(out, ref, atol=2e-2 * math.sqrt(K / 64), rtol=8e-3, what=f"gemm {M}x{N}x{K}")


def test_gemm_nt_exact_small_integers():
    """Integer data: every product/sum is exact in fp32 and bf16-representable -> bit-exact result required."""
    M, N, K = 256, 128, 128
    g = torch.Generator().manual_seed(3)
    a = torch.randint(-2, 3, (M, K), generator=g).float()
    b = torch.randint(-2, 3, (N, K), generator=g).float()
    out = ops.gemm_nt(a.to(DEV, BF), b.to(DEV, BF), out_dtype=torch.float32)
    assert torch.equal(out.cpu(), a @ b.T)


def test_gemm_nt_epilogues():
    M, N, K = 200, 256, 128
    a, b = _rand(M, K, seed=4).to(BF), _rand(N, K, seed=5).to(BF)
    bias, res = _rand(N, seed=6).to(BF), _rand(M, N, seed=7).to(BF)
    base = a.float() @ b.float().T
    _close(ops.gemm_nt(a, b, bias=bias), base + bias.float(), 3e-2, 8e-3, "bias")
    _close(ops.gemm_nt(a, b, bias=bias, gelu=True), torch.nn.functional.gelu(base + bias.float()), 3e-2, 8e-3, "gelu")
    _close(ops.gemm_nt(a, b, res=res), base + res.float(), 3e-2, 8e-3, "residual")
    acc = _rand(M, N, seed=8)
    out = acc.clone()
    ops.gemm_nt(a, b, out=out, accumulate=True)
    _close(out, base + acc, 1e-3, 1e-5, "fp32 accumulate")
    # strided views (fused-QKV style): write into a column slice of a wider buffer
    wide = torch.zeros(M, 3 * N, dtype=BF, device=DEV)
    ops.gemm_nt(a, b, out=wide[:, N:2 * N])
    _close(wide[:, N:2 * N], base, 3e-2, 8e-3, "strided out")
    assert wide[:, :N].abs().max() == 0 and wide[:, 2 * N:].abs().max() == 0


def test_gemm_rejects_bad_k():
    a, b = _rand(64, 48).to(BF), _rand(64, 48).to(BF)
    with pytest.raises(RuntimeError, match="multiple of 64"):
        ops.gemm_nt(a, b)


def test_transpose():
    x = _rand(300, 200, seed=9).to(BF)
    assert torch.equal(ops.transpose(x), x.T.contiguous())


# ------------------------------------------------------------------------------------------------
def _rmsnorm_ref(x, w, eps):
    xf = x.float()
    return (w.float() * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(BF).float())


@pytest.mark.parametrize("rows,H", [(37, 256), (512, 2048), (64, 2560), (16, 4096), (8, 1024)])
def test_rmsnorm_fwd_bwd(rows, H):
    x = _rand(rows, H, seed=10).to(BF)
    w = (1 + 0.1 * _rand(H, seed=11)).to(BF)
    y = ops.rmsnorm_fwd(x, w, 1e-6)
    _close(y, _rmsnorm_ref(x, w, 1e-6), 1e-2, 8e-3, "rmsnorm fwd")
    g = _rand(rows, H, seed=12).to(BF)
    dres = _rand(rows, H, seed=13).to(BF)
    xr = x.float().requires_grad_(True)
    wr = w.float().requires_grad_(True)
    yr = wr * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6))
    yr.backward(g.float())
    dw = torch.zeros(H, dtype=torch.float32, device=DEV)
    dx = ops.rmsnorm_bwd(x, w, g, dw, 1e-6, dres=dres)
    _close(dx, xr.grad + dres.float(), 2e-2, 1e-2, "rmsnorm dx")
    _close(dw, wr.grad, 2e-3 * math.sqrt(rows), 1e-3, "rmsnorm dw")


@pytest.mark.parametrize("rows,H", [(64, 512), (320, 1024), (2048, 2048), (192, 1536)])
def test_rmsnorm_with_the_transposed_second_store(rows, H):
    """molly_rmsnorm_fwd_t (round 5): the norm kernel stores its result a second time transposed — the k-contiguous operand of the
    weight gradient of the projection behind it — instead of a transpose launch reading it back.  y is bit-identical to the plain
    kernel's, the transposed copy is its exact transpose."""
    x = _rand(rows, H, seed=20).to(BF)
    w = (1 + 0.1 * _rand(H, seed=21)).to(BF)
    y0 = ops.rmsnorm_fwd(x, w, 1e-6)
    yt = torch.full((H, rows), 9.0, dtype=BF, device=DEV)
    y1 = ops.rmsnorm_fwd(x, w, 1e-6, out_t=yt)
    assert torch.equal(y0, y1) and torch.equal(yt, y0.t().contiguous())


def test_swiglu_fwd_bwd():
    rows, ff = 130, 512
    gu = _rand(rows, 2 * ff, seed=14).to(BF)
    out = ops.swiglu_fwd(gu)
    gr = gu.float().requires_grad_(True)
    ref = torch.nn.functional.silu(gr[:, :ff]) * gr[:, ff:]
    _close(out, ref, 1e-2, 1e-2, "swiglu fwd")
    d = _rand(rows, ff, seed=15).to(BF)
    ref.backward(d.float())
    _close(ops.swiglu_bwd(gu, d), gr.grad, 1e-2, 1e-2, "swiglu bwd")


def _rope_tables(T, hd, theta, dtype=torch.float32):
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    fr = torch.arange(T).float()[:, None] * inv
    return fr.cos().to(dtype).float().to(DEV).contiguous(), fr.sin().to(dtype).float().to(DEV).contiguous()


def _norm_rope_ref(x, nq, nk, hd, T, qw, kw, cos, sin, eps, q_scale):
    M = x.shape[0]
    xh = x.view(M, nq + nk, hd)
    if qw is not None:
        w = torch.cat([qw.view(1, 1, hd).expand(1, nq, hd), kw.view(1, 1, hd).expand(1, nk, hd)], 1)
        xh = w * (xh * torch.rsqrt(xh.pow(2).mean(-1, keepdim=True) + eps))
    if q_scale != 1.0:
        xh = torch.cat([xh[:, :nq] * q_scale, xh[:, nq:]], 1)
    if cos is not None:
        pos = torch.arange(M, device=x.device) % T
        c = torch.cat([cos, cos], -1)[pos][:, None, :]
        s = torch.cat([sin, sin], -1)[pos][:, None, :]
        rot = torch.cat([-xh[..., hd // 2:], xh[..., :hd // 2]], -1)
        xh = xh * c + rot * s
    return xh.reshape(M, -1)


@pytest.mark.parametrize("hd,nq,nk,norm", [(128, 4, 2, True), (64, 2, 2, False)])
def test_norm_rope_fwd_bwd(hd, nq, nk, norm):
    B, T = 2, 96
    M = B * T
    extra = 2 * hd                      # v columns after q|k in the fused buffer
    src = _rand(M, (nq + nk) * hd + extra, seed=16).to(BF)
    cos, sin = _rope_tables(T, hd, 1e6 if norm else 1e4)
    qw = (1 + 0.1 * _rand(hd, seed=17)).to(BF) if norm else None
    kw = (1 + 0.1 * _rand(hd, seed=18)).to(BF) if norm else None
    q_scale = 1.0 if norm else hd ** -0.5
    dst = torch.empty(M, (nq + nk) * hd, dtype=BF, device=DEV)
    ops.norm_rope_fwd(src, dst, nq, nk, hd, T, qw, kw, cos, sin, eps=1e-6, q_scale=q_scale)
    xr = src[:, :(nq + nk) * hd].float().requires_grad_(True)
    qwr = qw.float().requires_grad_(True) if norm else None
    kwr = kw.float().requires_grad_(True) if norm else None
    ref = _norm_rope_ref(xr, nq, nk, hd, T, qwr, kwr, cos, sin, 1e-6, q_scale)
    _close(dst, ref, 1.5e-2, 1e-2, "norm_rope fwd")
    if not norm:
        return
    g = _rand(M, (nq + nk) * hd, seed=19).to(BF)
    ref.backward(g.float())
    dsrc = torch.zeros(M, (nq + nk) * hd + extra, dtype=BF, device=DEV)
    dqw = torch.zeros(hd, dtype=torch.float32, device=DEV)
    dkw = torch.zeros(hd, dtype=torch.float32, device=DEV)
    ops.norm_rope_bwd(src, g, dsrc, nq, nk, hd, T, qw, kw, cos, sin, dqw, dkw, eps=1e-6)
    _close(dsrc[:, :(nq + nk) * hd], xr.grad, 2e-2, 1e-2, "norm_rope dsrc")
    assert dsrc[:, (nq + nk) * hd:].abs().max() == 0
    _close(dqw, qwr.grad, 5e-2, 5e-3, "dq_norm_w")
    _close(dkw, kwr.grad, 5e-2, 5e-3, "dk_norm_w")


def test_layernorm_fwd():
    x = _rand(70, 1280, seed=20).to(BF)
    w, b = (1 + 0.1 * _rand(1280, seed=21)).to(BF), (0.1 * _rand(1280, seed=22)).to(BF)
    ref = torch.nn.functional.layer_norm(x.float(), (1280,), w.float(), b.float(), 1e-5)
    _close(ops.layernorm_fwd(x, w, b, 1e-5), ref, 1.5e-2, 8e-3, "layernorm")


def test_copy_rows_gather_scatter():
    E = _rand(50, 64, seed=23).to(BF)
    ids = torch.tensor([3, 49, 0, 3, 7], device=DEV, dtype=torch.int64)
    out = torch.zeros(5, 64, dtype=BF, device=DEV)
    ops.copy_rows(E, out, 5, src_idx64=ids)
    assert torch.equal(out, E[ids])
    dst = torch.zeros(10, 64, dtype=BF, device=DEV)
    rows = torch.tensor([9, -1, 2, 4, 0], device=DEV, dtype=torch.int32)
    ops.copy_rows(out, dst, 5, dst_idx32=rows)
    assert torch.equal(dst[9], out[0]) and torch.equal(dst[2], out[2]) and dst[1].abs().max() == 0


# ------------------------------------------------------------------------------------------------
def _attn_ref(q, k, v, B, T, nh, nkv, hd, scale, causal, lo=None, hi=None):
    qh = q.float().view(B, T, nh, hd).transpose(1, 2)
    kh = k.float().view(B, T, nkv, hd).transpose(1, 2).repeat_interleave(nh // nkv, 1)
    vh = v.float().view(B, T, nkv, hd).transpose(1, 2).repeat_interleave(nh // nkv, 1)
    s = qh @ kh.transpose(2, 3) * scale
    idx = torch.arange(T, device=q.device)
    ok = torch.ones(B, 1, T, T, dtype=torch.bool, device=q.device)
    if causal:
        ok = ok & (idx[None, :] <= idx[:, None])[None, None]
    if lo is not None:
        ok = ok & ((idx[None, :] >= lo[:, None]) & (idx[None, :] < hi[:, None]))[:, None, None, :]
    s = s.masked_fill(~ok, float("-inf"))
    p = torch.softmax(s, -1)
    p = torch.nan_to_num(p, nan=0.0)
    o = (p @ vh).transpose(1, 2).reshape(B * T, nh * hd)
    lse2 = torch.logsumexp(s, -1) / math.log(2.0)
    return o, lse2


@pytest.mark.parametrize("hd,nh,nkv,T,causal,ragged", [
    (128, 4, 2, 256, True, False), (128, 4, 2, 200, True, True), (64, 2, 2, 64, False, True),
    (64, 4, 4, 320, False, False), (128, 2, 1, 512, True, True),
    (128, 16, 8, 2048, True, False), (128, 8, 2, 4096, True, True), (64, 20, 20, 1024, False, True),   # the step's real lengths
    (16, 20, 20, 200, False, True), (32, 4, 2, 130, True, True), (48, 2, 2, 64, False, False)])    # small-head kernel
def test_attn_fwd(hd, nh, nkv, T, causal, ragged):
    B = 2
    M = B * T
    # fused token-major buffer: q heads | k heads | v heads
    qkv = _rand(M, (nh + 2 * nkv) * hd, seed=24).to(BF)
    q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
    lo = hi = None
    if ragged:
        lo = torch.tensor([0, 3], device=DEV, dtype=torch.int32)
        hi = torch.tensor([T, T - 37], device=DEV, dtype=torch.int32)
    scale = hd ** -0.5
    o, lse = ops.attn_fwd(q, k, v, B, T, nh, nkv, hd, scale, causal, lo, hi)
    ro, rl = _attn_ref(q, k, v, B, T, nh, nkv, hd, scale, causal, lo, hi)
    # rows with no visible key: ours = 0 / -inf by definition
    live = torch.isfinite(rl)
    rof = ro.view(B, T, nh, hd).transpose(1, 2)
    of = o.float().view(B, T, nh, hd).transpose(1, 2)
    _close(of[live], rof[live], 2e-2, 1e-2, "attn O")     # bf16 P and bf16 output
    _close(lse[live], rl[live], 2e-3, 1e-4, "attn lse2")


@pytest.mark.parametrize("hd,nh,nkv,T,causal,ragged", [(128, 4, 2, 256, True, False), (128, 16, 8, 2048, True, True), (64, 4, 4, 384, False, True)])
def test_attn_fwd_transposed_second_store(hd, nh, nkv, T, causal, ragged):
    """molly_attn_fwd_ot (round 5): the forward stores O a second time transposed (the o-projection's weight-gradient operand) instead of a
    transpose launch reading it back: O and LSE are bit-identical to the plain call's, OT is O's exact transpose."""
    B = 2
    M = B * T
    qkv = _rand(M, (nh + 2 * nkv) * hd, seed=44).to(BF)
    q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
    lo = hi = None
    if ragged:
        lo = torch.tensor([0, 3], device=DEV, dtype=torch.int32)
        hi = torch.tensor([T, T - 37], device=DEV, dtype=torch.int32)
    o0, l0 = ops.attn_fwd(q, k, v, B, T, nh, nkv, hd, hd ** -0.5, causal, lo, hi)
    ot = torch.full((nh * hd, M), 9.0, dtype=BF, device=DEV)
    o1, l1 = ops.attn_fwd(q, k, v, B, T, nh, nkv, hd, hd ** -0.5, causal, lo, hi, out_t=ot)
    assert torch.equal(o0, o1) and torch.equal(l0, l1) and torch.equal(ot, o0.t().contiguous())


def test_attn_fwd_survives_a_running_maximum_that_explodes():
    """Scores that grow by far more than 2^64 along the key axis: the online rescale (guide T13) must keep O and the LSE finite and equal to the
    fp32 reference (softmax concentrates on the late keys)."""
    hd, nh, nkv, B, T = 128, 2, 1, 1, 512
    g = torch.Generator(device=DEV).manual_seed(3)
    q = torch.randn(B * T, nh * hd, device=DEV, generator=g) * 0.3
    k = torch.randn(B * T, nkv * hd, device=DEV, generator=g) * 0.3
    v = torch.randn(B * T, nkv * hd, device=DEV, generator=g)
    q[:, 0::hd] = 24.0                                           # every query: a large component 0 ...
    k[:, 0] = torch.linspace(0.0, 64.0, T, device=DEV)           # ... that later keys meet with a growing one: score * log2(e) up to 24 * 64 * 0.1275 = 196: 2^196 has no fp32
    q, k, v = q.to(BF), k.to(BF), v.to(BF)
    scale = hd ** -0.5
    o, lse = ops.attn_fwd(q, k, v, B, T, nh, nkv, hd, scale, True)
    ro, rl = _attn_ref(q, k, v, B, T, nh, nkv, hd, scale, True)
    assert bool(torch.isfinite(o.float()).all()) and bool(torch.isfinite(lse).all())
    _close(o.float().view(B, T, nh, hd).transpose(1, 2), ro.view(B, T, nh, hd).transpose(1, 2), 2e-2, 1e-2, "attn O (exploding maximum)")
    _close(lse, rl, 2e-3, 1e-3, "attn lse2 (exploding maximum)")


def _attn_ref_grads(q, k, v, do, B, T, nh, nkv, hd, scale, causal, lo=None, hi=None, chunk=1024):
    """fp32 dQ / dK / dV of the same attention by the closed form (P = softmax(S), dV = P^T dO, dS = P * (dP - rowsum(dO * O)),
    dQ = scale dS K, dK = scale dS^T Q), query rows in chunks so that T = 4096 fits.  A query row with no visible key (left padding
    under the causal mask) has P = 0 by definition: it contributes NOTHING to dK / dV whatever its dO holds, and its dQ is 0 — so
    dK and dV are comparable on every parametrisation, dead rows or not."""
    M, g = B * T, nh // nkv
    qh = q.float().view(B, T, nh, hd).transpose(1, 2)
    kh = k.float().view(B, T, nkv, hd).transpose(1, 2).repeat_interleave(g, 1)
    vh = v.float().view(B, T, nkv, hd).transpose(1, 2).repeat_interleave(g, 1)
    doh = do.float().view(B, T, nh, hd).transpose(1, 2)
    idx = torch.arange(T, device=q.device)
    dq = torch.zeros(B, nh, T, hd, device=q.device)
    dk = torch.zeros(B, nh, T, hd, device=q.device)
    dv = torch.zeros(B, nh, T, hd, device=q.device)
    live = torch.zeros(B, nh, T, dtype=torch.bool, device=q.device)
    for r0 in range(0, T, chunk):
        r1 = min(T, r0 + chunk)
        s = qh[:, :, r0:r1] @ kh.transpose(2, 3) * scale
        ok = torch.ones(B, 1, r1 - r0, T, dtype=torch.bool, device=q.device)
        if causal:
            ok = ok & (idx[None, :] <= idx[r0:r1, None])[None, None]
        if lo is not None:
            ok = ok & ((idx[None, :] >= lo[:, None]) & (idx[None, :] < hi[:, None]))[:, None, None, :]
        alive = ok.any(-1, keepdim=True)
        s = s.masked_fill(~ok, float("-inf")).masked_fill(~alive, 0.0)
        p = torch.softmax(s, -1) * alive
        o = p @ vh
        dp = doh[:, :, r0:r1] @ vh.transpose(2, 3)
        ds = p * (dp - (doh[:, :, r0:r1] * o).sum(-1, keepdim=True))
        dq[:, :, r0:r1] = ds @ kh * scale
        dk += ds.transpose(2, 3) @ qh[:, :, r0:r1] * scale
        dv += p.transpose(2, 3) @ doh[:, :, r0:r1]
        live[:, :, r0:r1] = alive[..., 0].expand(B, nh, r1 - r0)
    fold = lambda t: t.view(B, nkv, g, T, hd).sum(2).transpose(1, 2).reshape(M, nkv * hd)
    return dq.transpose(1, 2).reshape(M, nh * hd), fold(dk), fold(dv), live.transpose(1, 2).reshape(M, nh)


def _ragged_ranges(kind, B, T, left=5, cut=41):
    """kind: False = full rows | True = sample 1 left- AND right-padded (its first `left` queries see no key under the causal mask) |
    "right" = sample 1 right-padded only (lo = 0, hi < T: what the training collate emits)."""
    if not kind:
        return None, None
    lo = torch.zeros(B, device=DEV, dtype=torch.int32)
    hi = torch.full((B,), T, device=DEV, dtype=torch.int32)
    if kind is True:
        lo[-1] = left
    hi[-1] = T - cut
    return lo, hi


ATTN_BWD_CASES = [
    (128, 4, 2, 256, True, False), (128, 4, 2, 200, True, True), (64, 2, 2, 128, False, True),
    (128, 2, 1, 320, True, True),
    (128, 16, 8, 2048, True, False), (128, 8, 2, 4096, True, True), (64, 20, 20, 1024, False, False),   # real lengths
    (128, 16, 8, 2048, True, "right")]                                                                   # right-padded, causal, the step's length


def _attn_bwd_run(hd, nh, nkv, T, causal, ragged, cut=41):
    B = 2
    M = B * T
    qkv = _rand(M, (nh + 2 * nkv) * hd, seed=30, scale=0.7).to(BF)
    q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
    lo, hi = _ragged_ranges(ragged, B, T, cut=cut)
    scale = hd ** -0.5
    o, lse = ops.attn_fwd(q, k, v, B, T, nh, nkv, hd, scale, causal, lo, hi)
    do = _rand(M, nh * hd, seed=31).to(BF)                     # NOT zeroed on dead rows: the kernels must ignore them by themselves
    dqkv = torch.zeros_like(qkv)
    dq, dk, dv = dqkv[:, :nh * hd], dqkv[:, nh * hd:(nh + nkv) * hd], dqkv[:, (nh + nkv) * hd:]
    ops.attn_bwd(q, k, v, o, do, lse, B, T, nh, nkv, hd, scale, causal, dq, dk, dv, lo, hi)
    rq, rk, rv, live = _attn_ref_grads(q, k, v, do, B, T, nh, nkv, hd, scale, causal, lo, hi)
    return (dq, dk, dv), (rq, rk, rv), live


def _attn_bwd_check(got, ref, live, nh, hd):
    dq, dk, dv = got
    rq, rk, rv = ref
    lm = live[..., None].expand(-1, nh, hd).reshape(live.shape[0], nh * hd)
    assert float((dq.float() * ~lm).abs().max()) == 0.0, "dQ of a query row that sees no key must be exactly 0"
    _close(dq, rq, 3e-2, 2e-2, "dQ")
    _close(dk, rk, 4e-2, 2e-2, "dK")          # always: dead query rows contribute nothing in the reference either
    _close(dv, rv, 4e-2, 2e-2, "dV")


@pytest.mark.parametrize("hd,nh,nkv,T,causal,ragged", ATTN_BWD_CASES)
def test_attn_bwd(hd, nh, nkv, T, causal, ragged):
    """dQ, dK and dV against the fp32 closed form on EVERY parametrisation (round 6: the dK / dV comparison used to be skipped whenever
    a sample had dead query rows, i.e. on every causal + left-padded case)."""
    got, ref, live = _attn_bwd_run(hd, nh, nkv, T, causal, ragged)
    if ragged is True and causal:
        assert not bool(live.all())                             # the case has the dead rows it is there for
    _attn_bwd_check(got, ref, live, nh, hd)


@pytest.mark.parametrize("hd,nh,nkv,T,causal,ragged", [c for c in ATTN_BWD_CASES if c[0] == 128] + [(128, 6, 2, 333, False, True), (128, 4, 4, 97, True, False)])
def test_attn_bwd_one_pass_dk_dv_is_bit_identical_to_the_two_passes(hd, nh, nkv, T, causal, ragged, monkeypatch):
    """attn_bwd_dkv_fused_kernel (dK and dV in one pass: V fragments from LDS, 32-row stages; the default at head dim 128) against the two
    single-output passes it replaces (MOLLY_ATTN_DKV_FUSED=0): the same MFMAs on the same operands in the same order, so dK and dV agree bit
    for bit — on every head-dim-128 case of the closed-form test plus a bidirectional ragged one and a length below one key block."""
    outs = []
    for fused in ("1", "0"):
        monkeypatch.setenv("MOLLY_ATTN_DKV_FUSED", fused)
        got, _, _ = _attn_bwd_run(hd, nh, nkv, T, causal, ragged)
        torch.cuda.synchronize()
        outs.append([g.clone() for g in got])
    for a, b, name in zip(outs[0], outs[1], ("dQ", "dK", "dV")):
        assert torch.equal(a, b), name


@pytest.mark.parametrize("hd,nh,nkv,T,causal,ragged", [(128, 4, 2, 200, True, True), (128, 16, 8, 2048, True, "right")])
@pytest.mark.parametrize("what", ["dK", "dV"])
def test_attn_bwd_check_notices_one_zeroed_key_tile(hd, nh, nkv, T, causal, ragged, what):
    """The comparison above is only worth something if it FAILS on a broken kernel: one 64-key tile of one kv head of dK (or dV) zeroed by
    hand must trip it."""
    got, ref, live = _attn_bwd_run(hd, nh, nkv, T, causal, ragged)
    broken = [t.clone() for t in got]
    t = broken[1 if what == "dK" else 2]
    t[T + 64:T + 128, hd:2 * hd] = 0                           # sample 1, keys 64..127, kv head 1
    with pytest.raises(AssertionError, match=what):
        _attn_bwd_check(broken, ref, live, nh, hd)
    _attn_bwd_check(got, ref, live, nh, hd)


@pytest.mark.parametrize("nh,nkv,T,B,ragged", [(4, 2, 256, 2, False), (4, 2, 200, 2, True), (16, 8, 2048, 8, "right"), (8, 2, 1100, 3, True)])
def test_attn_bwd_with_the_rotary_and_qk_norm_backward_in_its_row_epilogues(nh, nkv, T, B, ragged):
    """molly_attn_bwd_rope (round 6): the q/k-norm + rotary backward (HF:models/qwen3/modeling_qwen3.py:225-236) runs inside the dQ / dK kernels' row
    epilogues instead of norm_rope_bwd_kernel behind them.  Same arithmetic on the same bf16-rounded dq / dk rows; what differs is the order of the
    two 128-element row sums (8 elements per lane here, 4 + 4 there) and of the gain-gradient sums over rows: d(q | k | v) within bf16 rounding
    of the two-kernel path (and dV bit-identical), gain gradients to fp32 summation noise."""
    hd = 128
    M = B * T
    nq, nk = nh * hd, nkv * hd
    x = _rand(M, nq + 2 * nk, seed=50, scale=0.8).to(BF)                    # the pre-norm q | k | v projection rows
    qw, kw = (1.0 + 0.2 * _rand(hd, seed=51)).to(BF), (1.0 + 0.2 * _rand(hd, seed=52)).to(BF)
    from molly_amd.qwen3 import rope_tables
    cos, sin = rope_tables(T, hd, 1e6, DEV)
    qk = torch.empty(M, nq + nk, dtype=BF, device=DEV)
    ops.norm_rope_fwd(x, qk, nh, nkv, hd, T, qw, kw, cos, sin, eps=1e-6)
    q, k, v = qk[:, :nq], qk[:, nq:], x[:, nq + nk:]
    lo, hi = _ragged_ranges(ragged, B, T)
    scale = hd ** -0.5
    o, lse = ops.attn_fwd(q, k, v, B, T, nh, nkv, hd, scale, True, lo, hi)
    do = _rand(M, nq, seed=53).to(BF)
    # two kernels
    d_qk = torch.zeros(M, nq + nk, dtype=BF, device=DEV)
    dx_ref = torch.zeros(M, nq + 2 * nk, dtype=BF, device=DEV)
    ops.attn_bwd(q, k, v, o, do, lse, B, T, nh, nkv, hd, scale, True, d_qk[:, :nq], d_qk[:, nq:], dx_ref[:, nq + nk:], lo, hi)
    dqw_ref, dkw_ref = torch.zeros(hd, dtype=torch.float32, device=DEV), torch.zeros(hd, dtype=torch.float32, device=DEV)
    ops.norm_rope_bwd(x, d_qk, dx_ref, nh, nkv, hd, T, qw, kw, cos, sin, dqw_ref, dkw_ref, eps=1e-6, dw_accumulate=False)
    # one pass
    nbq, nbk = ops.attn_bwd_rope_blocks(B, T, nh, nkv)
    assert (nbq, nbk) == (nh * B * ((T + 127) // 128), nkv * B * ((T + 127) // 128))
    dx = torch.full((M, nq + 2 * nk), 7.0, dtype=BF, device=DEV)
    pq = torch.full((nbq, hd), float("nan"), dtype=torch.float32, device=DEV)
    pk_ = torch.full((nbk, hd), float("nan"), dtype=torch.float32, device=DEV)
    ops.attn_bwd_rope(q, k, v, o, do, lse, B, T, nh, nkv, hd, scale, True, dx[:, nq + nk:], x, qw, kw, cos, sin, 1e-6, dx, pq, pk_, lo, hi)
    torch.cuda.synchronize()
    assert torch.equal(dx[:, nq + nk:], dx_ref[:, nq + nk:])                      # dV: the same kernel, the same output
    assert bool(torch.isfinite(pq).all()) and bool(torch.isfinite(pk_).all())    # every workgroup wrote its row
    ref = dx_ref[:, :nq + nk].float()
    err = (dx[:, :nq + nk].float() - ref).abs()
    assert float(err.max()) <= 2e-2 * float(ref.abs().max()) + 1e-3, (float(err.max()), float(ref.abs().max()))
    assert float((err > 1e-2 * ref.abs() + 1e-3).float().mean()) < 1e-3         # almost everywhere to the last bf16 bit
    for got, want, name in ((pq.sum(0), dqw_ref, "d q_norm.weight"), (pk_.sum(0), dkw_ref, "d k_norm.weight")):
        _close(got, want, 2e-3 * float(want.abs().max()) + 1e-4, 2e-3, name)


@pytest.mark.parametrize("nh,nkv,T,ragged", [(32, 8, 1024, False), (16, 8, 2048, True), (8, 2, 1000, True), (32, 8, 3072, False)])
def test_attn_bwd_split_by_query_head(nh, nkv, T, ragged, monkeypatch):
    """One sample per GPU (BASELINE configs 3 / 4, scripts/train/examples/run_train_4B_z2_b1.sh:29): the dK / dV passes run one block
    per QUERY head and a third launch adds the group's fp32 images in head order.  Against the unsplit passes (same products, another
    order of the fp32 sums over heads: equal to bf16 rounding of near-ties), against the fp32 reference, and twice (bitwise equal)."""
    B, hd = 1, 128
    M = B * T
    n_ws = ops.attn_bwd_workspace(B, T, nh, nkv, hd)
    assert n_ws == 2 * nh * B * ((T + 127) // 128) * 16384
    assert ops.attn_bwd_workspace(8, 2048, 16, 8, hd) == 0          # the headline's grid fills the chip: no split
    assert ops.attn_bwd_workspace(1, 512, 20, 20, 64) == 0          # one query head per kv head / hd 64: nothing to split
    qkv = _rand(M, (nh + 2 * nkv) * hd, seed=40, scale=0.7).to(BF)
    q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
    lo = hi = None
    if ragged:
        lo = torch.tensor([3], device=DEV, dtype=torch.int32)
        hi = torch.tensor([T - 41], device=DEV, dtype=torch.int32)
    scale = hd ** -0.5
    o, lse = ops.attn_fwd(q, k, v, B, T, nh, nkv, hd, scale, True, lo, hi)
    do = _rand(M, nh * hd, seed=41).to(BF)
    outs = []
    for ws in (None, torch.empty(n_ws, dtype=torch.float32, device=DEV), torch.full((n_ws,), float("nan"), dtype=torch.float32, device=DEV)):
        dqkv = torch.zeros_like(qkv)
        dq, dk, dv = dqkv[:, :nh * hd], dqkv[:, nh * hd:(nh + nkv) * hd], dqkv[:, (nh + nkv) * hd:]
        ops.attn_bwd(q, k, v, o, do, lse, B, T, nh, nkv, hd, scale, True, dq, dk, dv, lo, hi, ws=ws)
        outs.append(dqkv)
    torch.cuda.synchronize()
    assert torch.equal(outs[1], outs[2])                            # reproducible, and every image it reads it wrote (NaN-filled scratch)
    assert torch.equal(outs[0][:, :nh * hd], outs[1][:, :nh * hd])  # dQ is not touched by the split
    d = (outs[0].float() - outs[1].float()).abs().max().item()
    ref = outs[0].float().abs().max().item()
    assert d <= 2e-2 * ref, (d, ref)
    # against fp32 at every length, ragged or not (round 6: was T <= 1024 and unpadded only)
    rq, rk, rv, live = _attn_ref_grads(q, k, v, do, B, T, nh, nkv, hd, scale, True, lo, hi)
    for name, o_ in (("unsplit", outs[0]), ("split", outs[1])):
        _close(o_[:, :nh * hd], rq, 3e-2, 2e-2, f"dQ ({name})")
        _close(o_[:, nh * hd:(nh + nkv) * hd], rk, 4e-2, 2e-2, f"dK ({name})")
        _close(o_[:, (nh + nkv) * hd:], rv, 4e-2, 2e-2, f"dV ({name})")
    # the split form's images come from ONE pass (attn_bwd_dkv_fused_kernel<128, true>) by default: the same bits as the two single-output passes
    monkeypatch.setenv("MOLLY_ATTN_DKV_FUSED", "0")
    dqkv = torch.zeros_like(qkv)
    ops.attn_bwd(q, k, v, o, do, lse, B, T, nh, nkv, hd, scale, True, dqkv[:, :nh * hd], dqkv[:, nh * hd:(nh + nkv) * hd], dqkv[:, (nh + nkv) * hd:],
                 lo, hi, ws=torch.empty(n_ws, dtype=torch.float32, device=DEV))
    torch.cuda.synchronize()
    assert torch.equal(dqkv, outs[1])


def test_ce_fwd_bwd_matches_torch():
    rows, V = 96, 1024
    logits = _rand(rows, V, seed=32, scale=2.0).to(BF)
    labels = torch.randint(0, V, (rows,), generator=torch.Generator().manual_seed(33)).to(DEV)
    labels[::5] = -100
    scale = torch.empty(1, dtype=torch.float32, device=DEV)
    cnt = torch.empty(1, dtype=torch.float32, device=DEV)
    ops.count_valid(labels, scale, cnt)
    assert cnt.item() == (labels != -100).sum().item()
    lr = logits.float().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lr, labels, ignore_index=-100, reduction="mean")
    ref.backward()
    row_loss = torch.empty(rows, dtype=torch.float32, device=DEV)
    work = logits.clone()
    ops.ce_fwd_bwd(work, labels, row_loss, scale)
    loss = torch.empty(1, dtype=torch.float32, device=DEV)
    ops.sum_f32(row_loss, loss, scale=scale)
    assert abs(loss.item() - ref.item()) < 2e-4 * max(1.0, abs(ref.item()))
    # the per-row losses are fp32 values computed from the same bf16 logits: held to fp32 accuracy, not to a bf16 tolerance
    per_row = torch.nn.functional.cross_entropy(logits.double(), labels, ignore_index=-100, reduction="none")
    keep = labels != -100
    assert (row_loss[keep].double() - per_row[keep]).abs().max().item() <= 4e-6 * per_row[keep].max().item()
    assert row_loss[~keep].abs().max().item() == 0.0
    _close(work, lr.grad, 2e-5, 1e-2, "dlogits")     # bf16 output of values <= 1/n_valid
    assert work[::5].abs().max() == 0


def test_embed_bwd_sorted_index():
    n, H, V = 300, 128, 50
    g = _rand(n, H, seed=34).to(BF)
    ids = torch.randint(0, V, (n,), generator=torch.Generator().manual_seed(35))
    ids[::7] = -1                                   # rows overwritten by omic embeddings: no gradient
    order = torch.argsort(ids, stable=True).to(torch.int32)
    uid, counts = torch.unique_consecutive(ids[order.long()], return_counts=True)
    seg = torch.zeros(len(uid) + 1, dtype=torch.int32)
    seg[1:] = torch.cumsum(counts, 0)
    dE = _rand(V, H, seed=36).to(BF)
    ref = dE.float().clone()
    keep = ids >= 0
    ref.index_add_(0, ids[keep].to(DEV), g.float()[keep.to(DEV)])
    ops.embed_bwd(g, order.to(DEV), seg.to(DEV), uid.to(DEV), len(uid), dE)
    _close(dE, ref, 3e-2, 1e-2, "embedding grad")


def test_adamw_and_clip_match_torch():
    n = 4096 + 8
    p0 = _rand(n, seed=37)
    gr = _rand(n, seed=38).to(BF)
    ws = torch.empty(1024, dtype=torch.float32, device=DEV)
    nsq = torch.empty(1, dtype=torch.float32, device=DEV)
    ops.sqnorm(gr, nsq, ws)
    assert abs(nsq.item() - gr.double().pow(2).sum().item()) < 2e-6 * nsq.item()      # fp32 sum of exact products: fp32-accurate
    norm = torch.empty(1, dtype=torch.float32, device=DEV)
    coef = torch.empty(1, dtype=torch.float32, device=DEV)
    ops.clip_coef(nsq, 1.0, 1.0, norm, coef)
    pt = torch.nn.Parameter(p0.clone())
    pt.grad = gr.float().clone()
    tn = torch.nn.utils.clip_grad_norm_([pt], 1.0)
    assert abs(norm.item() - tn.item()) < 4e-6 * tn.item()
    opt = torch.optim.AdamW([pt], lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    master, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    pout = torch.empty(n, dtype=BF, device=DEV)
    for step in (1, 2, 3):
        opt.step()
        ops.adamw_step(master, m, v, gr, pout, 3e-3, 0.9, 0.999, 1e-8, 1e-2, step, coef)
        _close(master, pt.detach(), 1e-6, 1e-5, f"adamw master step {step}")
    assert torch.equal(pout, master.to(BF))


@pytest.mark.parametrize("absolute,token_dropout", [(False, True), (True, False)])
def test_esm_embed(absolute, token_dropout):
    from oracle import molly_ref as R
    n, K, H, V = 3, 40, 128, 33
    g = torch.Generator().manual_seed(39)
    ids = torch.randint(4, 24, (n, K), generator=g)
    ids[0, 30:] = 1
    ids[1, 5] = 32 if token_dropout else 2       # a masked token
    ids[2, :] = torch.randint(4, 24, (K,), generator=g)
    cfg = R.EncCfg(vocab_size=V, hidden_size=H, intermediate_size=256, num_hidden_layers=1, num_attention_heads=2,
                   max_position_embeddings=64, position_embedding_type="absolute" if absolute else "rotary",
                   token_dropout=token_dropout, pad_token_id=1, mask_token_id=32 if token_dropout else 2)
    wemb = _rand(V, H, seed=40).to(BF)
    pemb = _rand(64, H, seed=41).to(BF) if absolute else None
    sd = {"e.embeddings.word_embeddings.weight": wemb.float().cpu()}
    if absolute:
        sd["e.embeddings.position_embeddings.weight"] = pemb.float().cpu()
    ref = R.esm_embeddings(sd, "e.", cfg, ids, (ids != 1).long())
    out = torch.empty(n * K, H, dtype=BF, device=DEV)
    pos = torch.empty(n, K, dtype=torch.int32, device=DEV)
    klen = torch.empty(n, dtype=torch.int32, device=DEV)
    ops.esm_embed(ids.to(DEV), wemb, pemb, out, pos, klen, 1, cfg.mask_token_id, token_dropout)
    _close(out.view(n, K, H).cpu(), ref, 2e-2, 1e-2, "esm embeddings")
    assert torch.equal(pos.cpu().long(), R.esm_position_ids(ids, 1))        # index outputs: bit-exact
    assert klen.cpu().tolist() == [30, K, K]


@pytest.mark.parametrize("M,N,K", [(256, 384, 512), (300, 200, 128), (2048, 1024, 2048)])
def test_gemm_nn_dgrad_form(M, N, K):
    """out[M,N] = a[M,K] @ w[K,N] with w read in place as the k-major B operand."""
    a = _rand(M, K, seed=50).to(BF)
    w = _rand(K, N, seed=51).to(BF)
    out = ops.gemm(a, w, b_kmajor=True)
    _close(out, a.float() @ w.float(), atol=2e-2 * math.sqrt(K / 64), rtol=8e-3, what=f"gemm NN {M}x{N}x{K}")


@pytest.mark.parametrize("Mtok,N,K", [(256, 384, 512), (200, 264, 136), (4096, 1024, 512), (77 * 8, 128, 256)])
def test_gemm_tn_wgrad_form(Mtok, N, K):
    """dW[N,K] = dy[Mtok,N]^T @ x[Mtok,K]: both operands k-major, contraction length = tokens (any value)."""
    dy = _rand(Mtok, N, seed=52).to(BF)
    x = _rand(Mtok, K, seed=53).to(BF)
    ref = dy.float().T @ x.float()
    out = ops.gemm(dy, x, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)
    _close(out, ref, atol=1e-3 * math.sqrt(Mtok), rtol=1e-4, what=f"gemm TN {Mtok}x{N}x{K}")
    acc = _rand(N, K, seed=54).to(BF)
    base = acc.clone()
    ops.gemm(dy, x, out=acc, accumulate=True, a_kmajor=True, b_kmajor=True)
    _close(acc, ref + base.float(), atol=3e-2 * math.sqrt(Mtok / 64), rtol=8e-3, what="gemm TN accumulate")


def test_gemm_tn_exact_small_integers():
    g = torch.Generator().manual_seed(55)
    dy = torch.randint(-2, 3, (192, 136), generator=g).float()
    x = torch.randint(-2, 3, (192, 72), generator=g).float()
    out = ops.gemm(dy.to(DEV, BF), x.to(DEV, BF), a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)
    assert torch.equal(out.cpu(), dy.T @ x)


@pytest.mark.parametrize("tile", [128, 512])
@pytest.mark.parametrize("form", ["nt", "nn", "tn"])
def test_gemm_both_tile_configs_all_forms(tile, form):
    """Both kernels (128x128 2-stage, 256x256 persistent ping-pong) forced explicitly, ragged M/N."""
    M, N, K = 520, 392, 640
    try:
        lib().call("molly_gemm_force_tile", tile)
        if form == "nt":
            a, b = _rand(M, K, seed=60).to(BF), _rand(N, K, seed=61).to(BF)
            out, ref = ops.gemm(a, b), a.float() @ b.float().T
        elif form == "nn":
            a, b = _rand(M, K, seed=62).to(BF), _rand(K, N, seed=63).to(BF)
            out, ref = ops.gemm(a, b, b_kmajor=True), a.float() @ b.float()
        else:
            Kt = 600                                       # token-count contraction, not a multiple of 64
            a, b = _rand(Kt, M, seed=64).to(BF), _rand(Kt, N, seed=65).to(BF)
            out, ref = ops.gemm(a, b, a_kmajor=True, b_kmajor=True), a.float().T @ b.float()
        _close(out, ref, atol=8e-2, rtol=8e-3, what=f"gemm {form} tile {tile}")
    finally:
        lib().call("molly_gemm_force_tile", 0)


@pytest.mark.parametrize("tile", [512])
def test_gemm_256_tile_many_k_steps_race_screen(tile):
    """Long K (many ring revolutions) repeated: the pipelined kernels must give identical results run to run."""
    M, N, K = 1024, 768, 8192
    a, b = _rand(M, K, seed=66).to(BF), _rand(N, K, seed=67).to(BF)
    try:
        lib().call("molly_gemm_force_tile", tile)
        first = ops.gemm(a, b, out_dtype=torch.float32).clone()
        for _ in range(20):
            assert torch.equal(ops.gemm(a, b, out_dtype=torch.float32), first)
        lib().call("molly_gemm_force_tile", 128)
        _close(first, ops.gemm(a, b, out_dtype=torch.float32), 1e-2, 1e-4, "256 vs 128 tile")
    finally:
        lib().call("molly_gemm_force_tile", 0)


def test_gemm_splitk_wgrad_path():
    """wgrad-shaped problem whose 256x256 grid is small: the heuristic splits K into fp32 slabs + reduce."""
    ops.ensure_gemm_workspace(256 << 20)
    Mtok, N, K = 8192, 1024, 768
    dy, x = _rand(Mtok, N, seed=70).to(BF), _rand(Mtok, K, seed=71).to(BF)
    ref = dy.float().T @ x.float()
    out = ops.gemm(dy, x, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32)
    _close(out, ref, atol=1e-3 * math.sqrt(Mtok), rtol=1e-4, what="split-K wgrad fp32")
    acc = _rand(N, K, seed=72).to(BF)
    base = acc.clone()
    ops.gemm(dy, x, out=acc, accumulate=True, a_kmajor=True, b_kmajor=True)
    _close(acc, ref + base.float(), atol=0.5, rtol=8e-3, what="split-K wgrad bf16 accumulate")
    first = ops.gemm(dy, x, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32).clone()
    for _ in range(5):
        assert torch.equal(ops.gemm(dy, x, a_kmajor=True, b_kmajor=True, out_dtype=torch.float32), first)


def test_transpose_fast_path_and_trans_out_gemm():
    x = _rand(512, 1024, seed=80).to(BF)
    assert torch.equal(ops.transpose(x), x.T.contiguous())             # 64-multiple shapes -> tr-read kernel
    # wgrad form with the narrow operand transposed and the result stored transposed
    Mtok, N, K = 2048, 768, 512
    dy, xx = _rand(Mtok, N, seed=81).to(BF), _rand(Mtok, K, seed=82).to(BF)
    ref = dy.float().T @ xx.float()                                      # dW [N, K]
    xt = ops.transpose(xx)                                               # [K, Mtok]
    dw = ops.gemm(xt, dy, b_kmajor=True, trans_out=True, out_dtype=torch.float32)
    assert dw.shape == (N, K)
    _close(dw, ref, atol=1e-3 * math.sqrt(Mtok), rtol=1e-4, what="trans_out wgrad")
    acc = _rand(N, K, seed=83).to(BF)
    base = acc.clone()
    ops.gemm(xt, dy, out=acc, accumulate=True, b_kmajor=True, trans_out=True)
    _close(acc, ref + base.float(), atol=0.3, rtol=8e-3, what="trans_out accumulate bf16")
    # split-K + transposed slabs (long contraction, small output)
    ops.ensure_gemm_workspace(256 << 20)
    Mtok = 8192
    dy, xx = _rand(Mtok, 1024, seed=84).to(BF), _rand(Mtok, 512, seed=85).to(BF)
    dw = ops.gemm(ops.transpose(xx), dy, b_kmajor=True, trans_out=True, out_dtype=torch.float32)
    _close(dw, dy.float().T @ xx.float(), atol=1e-3 * math.sqrt(Mtok), rtol=1e-4, what="trans_out split-K")


@pytest.mark.parametrize("hd,nh,nkv,B,Tmax,use_ws", [(128, 16, 8, 4, 1500, True), (128, 32, 8, 3, 3000, True), (128, 8, 8, 2, 700, True),
                                                     (64, 8, 1, 5, 900, True), (128, 16, 8, 4, 1500, False), (64, 4, 2, 33, 300, True)])
def test_attn_decode_flash_decoding(hd, nh, nkv, B, Tmax, use_ws):
    """Decode-step attention (GQA group per block, split-KV + merge) against torch fp32 softmax over the valid key window
    [lo, hi) of each sample (left-padded prompts: lo > 0; one sample with a single key; rows outside the window are poison)."""
    from molly_amd import ops
    g = torch.Generator(device="cuda").manual_seed(hd + nh + Tmax)
    q = torch.randn(B, (nh + 2 * nkv) * hd, device="cuda", generator=g).bfloat16()         # q heads first, as in the qkv row
    kc = torch.randn(B, Tmax, nkv * hd, device="cuda", generator=g).bfloat16()
    vc = torch.randn(B, Tmax, nkv * hd, device="cuda", generator=g).bfloat16()
    lo = torch.randint(0, Tmax // 3, (B,), generator=torch.Generator().manual_seed(1)).int()
    hi = (lo + torch.randint(1, Tmax - Tmax // 3, (B,), generator=torch.Generator().manual_seed(2)).int()).clamp(max=Tmax)
    hi[0] = lo[0] + 1
    for b in range(B):                                                                       # poison outside the window
        kc[b, :lo[b]] = float("nan"); kc[b, hi[b]:] = float("nan")
        vc[b, :lo[b]] = float("nan"); vc[b, hi[b]:] = float("nan")
    out = torch.empty(B, nh * hd, dtype=torch.bfloat16, device="cuda")
    ws = ops.attn_decode_workspace(B, nh, hd, "cuda") if use_ws else None
    ops.attn_decode(q, kc, vc, out, lo.cuda(), hi.cuda(), B, Tmax, nh, nkv, hd, hd ** -0.5, kv_len_hint=int(hi.max()), workspace=ws)
    torch.cuda.synchronize()
    for b in range(B):
        k = kc[b, lo[b]:hi[b]].float().view(-1, nkv, hd).repeat_interleave(nh // nkv, dim=1)   # [n, nh, hd]
        v = vc[b, lo[b]:hi[b]].float().view(-1, nkv, hd).repeat_interleave(nh // nkv, dim=1)
        qq = q[b, :nh * hd].float().view(nh, hd)
        p = torch.softmax(torch.einsum("hd,nhd->hn", qq, k) * hd ** -0.5, dim=-1)
        ref = torch.einsum("hn,nhd->hd", p, v).reshape(-1)
        err = (out[b].float() - ref).abs().max().item()
        assert err <= 2e-2 * max(ref.abs().max().item(), 1.0), (b, err)
    assert torch.isfinite(out.float()).all()


@pytest.mark.parametrize("M,N,K", [(32, 4096, 2048), (8, 12288, 2048), (32, 2048, 6144), (1, 2048, 2048), (48, 64, 4096)])
def test_gemm_decode_rows_splitk_with_epilogue(M, N, K):
    """M = batch rows (decode): the heuristic streams the weight through the 256x256 kernel with K split over the chip;
    bias / GELU / residual are then applied by the slab-reduce kernel — same results as the one-pass epilogue."""
    from molly_amd import ops
    from molly_amd._lib import lib
    ops.ensure_gemm_workspace(64 << 20)
    lib().call("molly_gemm_ctx_set", None, ops.GEMM_KEYS["skinny"], 0)      # round 2's path (the decode-row kernels have their own tests)
    lib().call("molly_gemm_ctx_set", None, ops.GEMM_KEYS["rows_tiled"], 0)
    g = torch.Generator(device="cuda").manual_seed(M + N)
    a = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    w = (torch.randn(N, K, device="cuda", generator=g) / K ** 0.5).bfloat16()
    bias = torch.randn(N, device="cuda", generator=g).bfloat16()
    res = torch.randn(M, N, device="cuda", generator=g).bfloat16()
    for kw in (dict(), dict(res=res), dict(bias=bias, gelu=True), dict(bias=bias, res=res)):
        out = ops.gemm_nt(a, w, **kw)
        cfg = lib().query("molly_gemm_last_config")
        lib().call("molly_gemm_force_tile", 128)
        ref = ops.gemm_nt(a, w, **kw)
        lib().call("molly_gemm_force_tile", 0)
        if N >= 2048:
            assert cfg // 1000 > 1 and cfg % 1000 == 512, cfg                 # went through the split-K slab path
        assert (out.float() - ref.float()).abs().max().item() <= 2e-2 * max(ref.float().abs().max().item(), 1.0), kw.keys()
    r32 = a.float() @ w.float().t()
    assert (ops.gemm_nt(a, w).float() - r32).abs().max().item() <= 2e-2 * r32.abs().max().item()
    lib().call("molly_gemm_ctx_set", None, ops.GEMM_KEYS["skinny"], 1)
    lib().call("molly_gemm_ctx_set", None, ops.GEMM_KEYS["rows_tiled"], 1)


@pytest.mark.parametrize("M", [1, 17, 32, 33, 64])
@pytest.mark.parametrize("N,K", [(24576, 4096), (4096, 12288), (1000, 1280), (128, 256), (40000, 512), (6144, 2560)])
def test_gemm_decode_rows_tiled_kernel(M, N, K):
    """The tiled decode-row kernel (gemm.hip gemm_rows_kernel: the M = 17..64 projections of BASELINE config 5's batch-32 decode,
    HF:models/qwen3/modeling_qwen3.py:76-83, 225-236 with one token per sample): exact on small integers with every epilogue, through
    the K-sliced path (fp32 slabs + reduce) and the direct one (many column tiles), ragged N, K-tile counts that do not divide."""
    g = torch.Generator(device="cuda").manual_seed(M * 131 + N + K)
    ints = lambda *s: torch.randint(-3, 4, s, device="cuda", generator=g).to(BF)
    x, w, bias, res = ints(M, K), ints(N, K), ints(N), ints(M, N)
    ref = x.float() @ w.float().t()
    c = ops.GemmContext()
    c.ensure_workspace(64 << 20)
    c.set("skinny", 0)
    with ops.use_gemm_context(c):
        out = ops.gemm_nt(x, w, out_dtype=torch.float32)
        cfg = c.get("last_config")
        assert cfg % 1000 == 32, cfg
        if M <= 32:      # 64-row tiles: as many K slices as keep tiles x slices <= 256 (one workgroup per CU), slices of >= 4 K-tiles
            assert cfg // 1000 == max(1, min(256 // -(-N // 64), (K // 64) // 4)), cfg
        else:            # 128-row tiles, priced rounds: one slice only where the column tiles alone fill the chip
            assert (cfg // 1000 == 1) == (N == 40000 or K == 256), cfg
        out_again = ops.gemm_nt(x, w, out_dtype=torch.float32)               # the tile counters were left at zero
        assert torch.equal(out_again, ref)
        assert torch.equal(out, ref)
        assert torch.equal(ops.gemm_nt(x, w, bias=bias, res=res, out_dtype=torch.float32), ref + bias.float() + res.float())
        acc = ints(M, N).float()
        want = acc + ref
        ops.gemm_nt(x, w, out=acc, accumulate=True)
        assert torch.equal(acc, want)
        got = ops.gemm_nt(x, w, bias=bias, gelu=True)
        wg = torch.nn.functional.gelu(ref + bias.float())
        assert (got.float() - wg).abs().max().item() <= 2 ** -7 * wg.abs().max().item()
        # a strided x (rows of a wider buffer) and a strided output
        wide = ints(M, K + 64)
        outw = torch.zeros(M, N + 8, dtype=torch.float32, device="cuda")
        ops.gemm_nt(wide[:, 64:], w, out=outw[:, :N])
        assert torch.equal(outw[:, :N], wide[:, 64:].float() @ w.float().t()) and outw[:, N:].abs().max().item() == 0


@pytest.mark.parametrize("M", [100, 512, 1000])
@pytest.mark.parametrize("N,K", [(1280, 1280), (1280, 5120), (3840, 1280), (5120, 1280)])
def test_gemm_small_grids_as_64_row_tiles(M, N, K):
    """Encoder projections at one sample per GPU (HF:models/esm/modeling_esm.py:350-463 at 512 / 1,000 rows: BASELINE configs 3 / 4 / 5):
    grids of at most 192 128x128 blocks run as 64-row tiles of the tiled decode-row kernel (MOLLY_GEMM_KEY_ROWS_MAX_M) — exact on
    small integers with the encoders' epilogues, ragged last row tile, K slices or not."""
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    ints = lambda *s: torch.randint(-3, 4, s, device="cuda", generator=g).to(BF)
    x, w, bias, res = ints(M, K), ints(N, K), ints(N), ints(M, N)
    ref = x.float() @ w.float().t()
    c = ops.GemmContext()
    c.ensure_workspace(64 << 20)
    small = -(-M // 128) * -(-N // 128) <= 192
    with ops.use_gemm_context(c):
        out = ops.gemm_nt(x, w, out_dtype=torch.float32)
        assert (c.get("last_config") % 1000 == 32) == small, (c.get("last_config"), small)
        assert torch.equal(out, ref)
        assert torch.equal(ops.gemm_nt(x, w, bias=bias, res=res, out_dtype=torch.float32), ref + bias.float() + res.float())
        acc = ints(M, N).float()
        want = acc + ref
        ops.gemm_nt(x, w, out=acc, accumulate=True)
        assert torch.equal(acc, want)
        got = ops.gemm_nt(x, w, bias=bias, gelu=True)
        wg = torch.nn.functional.gelu(ref + bias.float())
        assert (got.float() - wg).abs().max().item() <= 2 ** -7 * wg.abs().max().item()
    off = ops.GemmContext()
    off.ensure_workspace(64 << 20)
    off.set("rows_max_m", 64)
    with ops.use_gemm_context(off):
        assert torch.equal(ops.gemm_nt(x, w, out_dtype=torch.float32), ref) and off.get("last_config") % 1000 != 32


@pytest.mark.parametrize("H,FF", [(2560, 9728), (2048, 6144), (3584, 18944)])
def test_gemm_decode_rows_tails_at_other_model_widths(H, FF):
    """The same tails at the Qwen3-4B / 1.7B widths and a 3584 x 18944 pair (row lengths that are no multiple of 4,096; shapes the streaming kernel
    keeps are reported unsupported and skipped)."""
    M = 32
    g = torch.Generator(device="cuda").manual_seed(H)
    c = ops.GemmContext()
    c.ensure_workspace(64 << 20)
    rnd = lambda *s, sc=1.0: ((torch.rand(*s, device="cuda", generator=g) * 2 - 1) * sc).to(BF)
    checked = 0
    with ops.use_gemm_context(c):
        if ops.gemm_rows_tail_supported(M, H, FF, "norm"):
            x, w, res, gain = rnd(M, FF), rnd(H, FF, sc=FF ** -0.5), rnd(M, H), rnd(H)
            y0 = ops.gemm_nt(x, w, res=res)
            n0 = ops.rmsnorm_fwd(y0, gain, 1e-6)
            y1, n1 = torch.empty_like(y0), torch.empty_like(y0)
            ops.gemm_rows_norm(x, w, y1, gain, 1e-6, n1, res=res)
            assert torch.equal(y0, y1)
            d = (n0.float() - n1.float()).abs()
            assert d.max().item() <= 2 ** -7 * n0.float().abs().max().item() and (d > 0).float().mean().item() < 0.02
            checked += 1
        if ops.gemm_rows_tail_supported(M, 2 * FF, H, "swiglu"):
            x2, w2 = rnd(M, H), rnd(2 * FF, H, sc=H ** -0.5)
            gu0 = ops.gemm_nt(x2, w2)
            cfg0 = c.get("last_config")
            a0 = ops.swiglu_fwd(gu0)
            gu1, a1 = torch.empty_like(gu0), torch.empty_like(a0)
            c.set("rows_gu", 0)                              # K slices + the combine launch
            ops.gemm_rows_swiglu(x2, w2, gu1, a1)
            if c.get("last_config") == cfg0:                 # the same K slices as the plain GEMM took: the same sums
                assert torch.equal(gu0, gu1) and torch.equal(a0, a1)
            else:                                            # (a tail needs >= 2 slices; the plain GEMM of a wide matrix may take one)
                d = (gu1.float() - gu0.float()).abs()
                assert d.max().item() <= 2 ** -7 * gu0.float().abs().max().item() and (d > 0).float().mean().item() < 0.03
                assert torch.equal(a1, ops.swiglu_fwd(gu1))
            for mode in (1, 32, 64, 128):                    # one slice, the activation from the accumulators (1: the launcher picks the tile)
                c.set("rows_gu", mode)
                gu2, a2 = torch.full_like(gu0, float("nan")), torch.full_like(a0, float("nan"))
                ops.gemm_rows_swiglu(x2, w2, gu2, a2)
                assert c.get("last_config") in (1132, 1232, 1332)
                d = (gu2.float() - gu0.float()).abs()
                assert d.max().item() <= 2 ** -7 * gu0.float().abs().max().item() and (d > 0).float().mean().item() < 0.03
                assert torch.equal(a2, ops.swiglu_fwd(gu2))
            checked += 1
    assert checked >= 1


@pytest.mark.parametrize("M", [17, 32, 64])
def test_gemm_decode_rows_tails_equal_the_separate_kernels(M):
    """The decode step's launches folded into the decode-row GEMM's slab combine (molly_gemm_rows_tail_bf16_ctx): residual + RMSNorm
    behind o_proj / down_proj, SwiGLU behind gate|up (HF:models/qwen3/modeling_qwen3.py:50-63, 76-83, 262-276 with one token per
    sample) — bit-identical to the GEMM followed by the separate kernel, except where the fp32 order of the row's sum of squares
    moves rstd by an ulp (allowed: one bf16 ulp on a few elements)."""
    g = torch.Generator(device="cuda").manual_seed(M)
    H, FF = 4096, 12288
    c = ops.GemmContext()
    c.ensure_workspace(64 << 20)
    rnd = lambda *s, sc=1.0: ((torch.rand(*s, device="cuda", generator=g) * 2 - 1) * sc).to(BF)
    with ops.use_gemm_context(c):
        # residual + norm behind a [H x FF] projection
        x, w, res, gain = rnd(M, FF), rnd(H, FF, sc=FF ** -0.5), rnd(M, H), rnd(H)
        assert ops.gemm_rows_tail_supported(M, H, FF, "norm")
        y0 = ops.gemm_nt(x, w, res=res)
        n0 = ops.rmsnorm_fwd(y0, gain, 1e-6)
        y1, n1 = torch.empty_like(y0), torch.empty_like(y0)
        ops.gemm_rows_norm(x, w, y1, gain, 1e-6, n1, res=res)
        assert torch.equal(y0, y1)
        d = (n0.float() - n1.float()).abs()
        assert d.max().item() <= 2 ** -7 * n0.float().abs().max().item() and (d > 0).float().mean().item() < 0.02
        # SwiGLU behind a [2 FF' x H] projection
        x2, w2 = rnd(M, H), rnd(2 * 6144, H, sc=H ** -0.5)
        assert ops.gemm_rows_tail_supported(M, 2 * 6144, H, "swiglu")
        gu0 = ops.gemm_nt(x2, w2)
        cfg0 = c.get("last_config")
        a0 = ops.swiglu_fwd(gu0)
        gu1, a1 = torch.empty_like(gu0), torch.empty_like(a0)
        c.set("rows_gu", 0)                                  # K slices + the combine launch
        ops.gemm_rows_swiglu(x2, w2, gu1, a1)
        if c.get("last_config") == cfg0:                     # the same K slices as the plain GEMM took: the same sums
            assert torch.equal(gu0, gu1) and torch.equal(a0, a1)
        else:                                                # (a tail needs >= 2 slices; the plain GEMM of a wide matrix takes one)
            d = (gu1.float() - gu0.float()).abs()
            assert d.max().item() <= 2 ** -7 * gu0.float().abs().max().item() and (d > 0).float().mean().item() < 0.03
            assert torch.equal(a1, ops.swiglu_fwd(gu1))
        for bn in (32, 64, 128):
            # M <= 32: ONE K slice, the activation formed from the accumulators (no slabs): gate | up differ from the sliced sums by the
            # fp32 order only, and the activation is exactly SwiGLU of the gate | up the launch stored; without a gate | up output: the same
            c.set("rows_gu", bn)
            gu2, a2, a3 = torch.full_like(gu0, float("nan")), torch.full_like(a0, float("nan")), torch.full_like(a0, float("nan"))
            ops.gemm_rows_swiglu(x2, w2, gu2, a2)
            ops.gemm_rows_swiglu(x2, w2, None, a3)
            assert c.get("last_config") == 1032 + {32: 300, 64: 100, 128: 200}[bn] or M > 32
            d = (gu2.float() - gu0.float()).abs()
            assert d.max().item() <= 2 ** -7 * gu0.float().abs().max().item() and (d > 0).float().mean().item() < 0.03
            assert torch.equal(a2, ops.swiglu_fwd(gu2)) and torch.equal(a2, a3)
            if M > 32:
                assert torch.equal(gu2, gu0)
        # with a bias in front of either tail
        bias, bias2 = rnd(H), rnd(2 * 6144)
        yb0 = ops.gemm_nt(x, w, bias=bias, res=res)
        nb0 = ops.rmsnorm_fwd(yb0, gain, 1e-6)
        yb1, nb1 = torch.empty_like(yb0), torch.empty_like(yb0)
        ops.gemm_rows_norm(x, w, yb1, gain, 1e-6, nb1, res=res, bias=bias)
        assert torch.equal(yb0, yb1) and (nb0.float() - nb1.float()).abs().max().item() <= 2 ** -7 * nb0.float().abs().max().item()
        gub0 = ops.gemm_nt(x2, w2, bias=bias2)
        cfgb = c.get("last_config")
        ab0 = ops.swiglu_fwd(gub0)
        gub1, ab1 = torch.empty_like(gub0), torch.empty_like(ab0)
        c.set("rows_gu", 0)
        ops.gemm_rows_swiglu(x2, w2, gub1, ab1, bias=bias2)
        if c.get("last_config") == cfgb:
            assert torch.equal(gub0, gub1) and torch.equal(ab0, ab1)
        else:
            assert (gub1.float() - gub0.float()).abs().max().item() <= 2 ** -7 * gub0.float().abs().max().item()
            assert torch.equal(ab1, ops.swiglu_fwd(gub1))
        c.set("rows_gu", 64)
        ops.gemm_rows_swiglu(x2, w2, gub1, ab1, bias=bias2)
        assert (gub1.float() - gub0.float()).abs().max().item() <= 2 ** -7 * gub0.float().abs().max().item()
        assert torch.equal(ab1, ops.swiglu_fwd(gub1))
        assert not ops.gemm_rows_tail_supported(8, 4096, 4096, "norm")        # a streaming-kernel shape keeps its one launch
        assert not ops.gemm_rows_tail_supported(M, 151936, 4096, "norm")      # the lm_head is not a decode-row-kernel shape


def test_norm_rope_with_cache_append_equals_the_three_launches():
    """Decode step: q/k norm + rotary and the KV-cache append (HF DynamicCache.update, reference src/model/omics_one.py:220-232) in
    one launch — same q | k rows and the same cache rows as norm_rope_fwd + two copy_rows."""
    g = torch.Generator(device="cuda").manual_seed(3)
    B, nh, nkv, hd, Tmax = 5, 8, 2, 128, 40
    qkv = ((torch.rand(B, (nh + 2 * nkv) * hd, device="cuda", generator=g) * 2 - 1)).to(BF)
    qn, kn = (torch.rand(hd, device="cuda", generator=g) + 0.5).to(BF), (torch.rand(hd, device="cuda", generator=g) + 0.5).to(BF)
    cos = torch.rand(64, hd // 2, device="cuda", generator=g)
    sin = torch.rand(64, hd // 2, device="cuda", generator=g)
    pos = torch.tensor([3, 9, 0, 33, 17], dtype=torch.int32, device="cuda")
    slot = torch.tensor([b * Tmax + int(p) for b, p in enumerate(pos.tolist())], dtype=torch.int32, device="cuda")
    nq, nkvd = nh * hd, nkv * hd
    qk0 = torch.empty(B, nq + nkvd, dtype=BF, device="cuda")
    kc0 = torch.zeros(B * Tmax, nkvd, dtype=BF, device="cuda"); vc0 = torch.zeros_like(kc0)
    ops.norm_rope_fwd(qkv, qk0, nh, nkv, hd, 1, qn, kn, cos, sin, positions=pos)
    ops.copy_rows(qk0[:, nq:], kc0, B, dst_idx32=slot)
    ops.copy_rows(qkv[:, nq + nkvd:], vc0, B, dst_idx32=slot)
    qk1 = torch.empty_like(qk0)
    kc1 = torch.zeros_like(kc0); vc1 = torch.zeros_like(kc0)
    ops.norm_rope_fwd(qkv, qk1, nh, nkv, hd, 1, qn, kn, cos, sin, positions=pos, kcache=kc1, vcache=vc1, slot=slot)
    torch.cuda.synchronize()
    assert torch.equal(qk0, qk1) and torch.equal(kc0, kc1) and torch.equal(vc0, vc1)
    assert kc1.abs().sum().item() > 0 and vc1.abs().sum().item() > 0


@pytest.mark.parametrize("M,T,nh,nkv,hd,norm,with_pos", [(1000, 96, 5, 2, 128, True, False), (777, 64, 4, 4, 64, False, False),
                                                        (333, 1, 16, 8, 128, True, True), (37, 50, 3, 1, 128, True, False),
                                                        # >= 1,024 workgroups of token rows: norm_rope_fwd_rows_kernel (one lane group per token)
                                                        (16400, 96, 3, 2, 128, True, False), (32790, 64, 2, 1, 64, False, False),
                                                        (16390, 1, 4, 2, 128, True, True)])
def test_norm_rope_fast_kernel_is_bit_identical_to_the_general_one(M, T, nh, nkv, hd, norm, with_pos):
    """molly_norm_rope_fwd's compile-time-head_dim kernels (two heads per thread with 32-bit index arithmetic; one lane group per token row when
    the rows fill the chip: the training / prefill launch) against
    norm_rope_fwd_kernel, reached through the cache-append entry point: the same bits in every q | k row, on item counts that leave the last
    workgroup partly dead, with a position table, with T not a power of two, with and without the norm (ESM: q scaled, no norm)."""
    g = torch.Generator(device="cuda").manual_seed(11)
    nq, nkvd = nh * hd, nkv * hd
    qkv = ((torch.rand(M, nq + 2 * nkvd, device="cuda", generator=g) * 2 - 1)).to(BF)
    qn = (torch.rand(hd, device="cuda", generator=g) + 0.5).to(BF) if norm else None
    kn = (torch.rand(hd, device="cuda", generator=g) + 0.5).to(BF) if norm else None
    npos = 128
    cos = torch.rand(npos, hd // 2, device="cuda", generator=g)
    sin = torch.rand(npos, hd // 2, device="cuda", generator=g)
    pos = torch.randint(0, npos, (M,), device="cuda", generator=g).int() if with_pos else None
    q_scale = 1.0 if norm else hd ** -0.5
    guard = 7.0
    qk0 = torch.full((M + 1, nq + nkvd), guard, dtype=BF, device="cuda")
    ops.norm_rope_fwd(qkv, qk0[:M], nh, nkv, hd, T, qn, kn, cos, sin, positions=pos, q_scale=q_scale)
    qk1 = torch.empty(M, nq + nkvd, dtype=BF, device="cuda")
    kc = torch.zeros(M, nkvd, dtype=BF, device="cuda"); vc = torch.zeros_like(kc)
    slot = torch.arange(M, dtype=torch.int32, device="cuda")
    ops.norm_rope_fwd(qkv, qk1, nh, nkv, hd, T, qn, kn, cos, sin, positions=pos, q_scale=q_scale, kcache=kc, vcache=vc, slot=slot)
    torch.cuda.synchronize()
    assert torch.equal(qk0[:M], qk1)
    assert bool((qk0[M] == guard).all())                      # nothing written past the last item
    assert torch.equal(kc, qk1[:, nq:])


def test_gemm_decode_rows_qkv_tail_equals_the_separate_launches():
    """Decode step: fused q | k | v projection -> q/k-norm -> rotary -> KV-cache append as GEMM + ONE tail launch
    (molly_gemm_rows_qkv_bf16_ctx; HF:models/qwen3/modeling_qwen3.py:225-236) — bit-identical q | k rows and cache rows to
    gemm_nt + norm_rope_fwd + the two copy_rows."""
    g = torch.Generator(device="cuda").manual_seed(9)
    B, nh, nkv, hd, H, Tmax = 32, 32, 8, 128, 4096, 24
    nq, nkvd = nh * hd, nkv * hd
    rnd = lambda *s, sc=1.0: ((torch.rand(*s, device="cuda", generator=g) * 2 - 1) * sc).to(BF)
    x, w = rnd(B, H), rnd(nq + 2 * nkvd, H, sc=H ** -0.5 * 4)
    qn, kn = (rnd(hd) * 0.5 + 1.0).to(BF), (rnd(hd) * 0.5 + 1.0).to(BF)
    cos = torch.rand(64, hd // 2, device="cuda", generator=g)
    sin = torch.rand(64, hd // 2, device="cuda", generator=g)
    pos = torch.randint(0, Tmax, (B,), generator=torch.Generator().manual_seed(1)).int().cuda()
    slot = (torch.arange(B, device="cuda", dtype=torch.int32) * Tmax + pos).int()
    c = ops.GemmContext()
    c.ensure_workspace(64 << 20)
    with ops.use_gemm_context(c):
        assert ops.gemm_rows_tail_supported(B, nq + 2 * nkvd, H, "qkv")
        qkv = ops.gemm_nt(x, w)
        qk0 = torch.empty(B, nq + nkvd, dtype=BF, device="cuda")
        kc0 = torch.zeros(B * Tmax, nkvd, dtype=BF, device="cuda"); vc0 = torch.zeros_like(kc0)
        ops.norm_rope_fwd(qkv, qk0, nh, nkv, hd, 1, qn, kn, cos, sin, positions=pos)
        ops.copy_rows(qk0[:, nq:], kc0, B, dst_idx32=slot)
        ops.copy_rows(qkv[:, nq + nkvd:], vc0, B, dst_idx32=slot)
        qk1 = torch.empty_like(qk0)
        kc1 = torch.zeros_like(kc0); vc1 = torch.zeros_like(kc0)
        ops.gemm_rows_qkv(x, w, qk1, nh, nkv, hd, qn, kn, cos, sin, pos, 1e-6, kc1, vc1, slot)
    torch.cuda.synchronize()
    assert torch.equal(qk0, qk1) and torch.equal(kc0, kc1) and torch.equal(vc0, vc1)
    assert kc1.abs().sum().item() > 0 and vc1.abs().sum().item() > 0


def _decode_qkv_case(B, nh, nkv, hd, H, Tmax, n_keys, seed, ragged=True):
    g = torch.Generator(device="cuda").manual_seed(seed)
    nq, nkvd = nh * hd, nkv * hd
    rnd = lambda *s, sc=1.0: ((torch.rand(*s, device="cuda", generator=g) * 2 - 1) * sc).to(BF)
    x, w = rnd(B, H), rnd(nq + 2 * nkvd, H, sc=H ** -0.5 * 4)
    qn, kn = (rnd(hd) * 0.5 + 1.0).to(BF), (rnd(hd) * 0.5 + 1.0).to(BF)
    cos = torch.rand(Tmax, hd // 2, device="cuda", generator=g)
    sin = torch.rand(Tmax, hd // 2, device="cuda", generator=g)
    kc = rnd(B, Tmax, nkvd); vc = rnd(B, Tmax, nkvd)
    cg = torch.Generator().manual_seed(seed)
    # the new token sits at cache position n_keys (hi = n_keys + 1); left padding: the first lo keys of a row are not attended
    lo = (torch.randint(0, max(1, n_keys // 2), (B,), generator=cg) if ragged else torch.zeros(B, dtype=torch.long)).int().cuda()
    lo[0] = 0
    if ragged and B > 1:
        lo[1] = n_keys                                   # a row whose ONLY key is the new one
    hi = torch.full((B,), n_keys + 1, dtype=torch.int32, device="cuda")
    pos = (hi - 1 - lo).int()
    slot = (torch.arange(B, device="cuda", dtype=torch.int32) * Tmax + n_keys).int()
    return x, w, qn, kn, cos, sin, kc, vc, lo, hi, pos, slot


@pytest.mark.parametrize("B,nh,nkv,hd,H,n_keys,nw", [(32, 32, 8, 128, 4096, 1500, 0), (32, 32, 8, 128, 4096, 1500, 4), (8, 16, 8, 128, 2048, 700, 0),
                                                    (24, 16, 8, 64, 1024, 300, 0), (32, 32, 8, 128, 4096, 37, 16), (4, 8, 8, 128, 1024, 1100, 16), (2, 16, 2, 128, 1024, 600, 0),
                                                    (3, 8, 1, 64, 512, 90, 4)])
def test_attn_decode_from_the_projection_slabs_equals_the_three_launches(B, nh, nkv, hd, H, n_keys, nw, monkeypatch):
    """molly_attn_decode_qkv (slab combine + q/k-norm + rotary + cache append + attention over the old keys and the new one, ONE
    launch; blocks of 16 waves without key splits where B * n_kv_heads fills the chip) against molly_gemm_rows_qkv_bf16_ctx +
    molly_attn_decode: the appended cache rows bit for bit, the attention output within the fp32 reassociation of the softmax sums
    (the new key enters last instead of in cache order), and both against an fp32 torch attention on the same bf16 q | k | v."""
    import subprocess, sys, os, json
    if nw:                                               # the block shape is read once per process: a forced shape runs in a child
        code = ("import os, sys, torch; sys.path.insert(0, 'tests'); import test_gpu_kernels as t; "
                f"t._attn_decode_qkv_check({B}, {nh}, {nkv}, {hd}, {H}, {n_keys}); print('ok')")
        env = dict(os.environ, MOLLY_DECODE_NW=str(nw))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]
    else:
        _attn_decode_qkv_check(B, nh, nkv, hd, H, n_keys)


def _attn_decode_qkv_check(B, nh, nkv, hd, H, n_keys):
    Tmax = n_keys + 8
    x, w, qn, kn, cos, sin, kc, vc, lo, hi, pos, slot = _decode_qkv_case(B, nh, nkv, hd, H, Tmax, n_keys, seed=21)
    nq, nkvd = nh * hd, nkv * hd
    c = ops.GemmContext()
    c.ensure_workspace(64 << 20)
    ws = ops.attn_decode_workspace(B, nh, hd, "cuda")
    with ops.use_gemm_context(c):
        kc0, vc0 = kc.clone(), vc.clone()
        kc1, vc1 = kc.clone(), vc.clone()
        qk0 = torch.empty(B, nq + nkvd, dtype=BF, device="cuda")
        out0 = torch.empty(B, nq, dtype=BF, device="cuda")
        out1 = torch.full((B, nq), float("nan"), dtype=BF, device="cuda")
        if ops.gemm_rows_tail_supported(B, nq + 2 * nkvd, H, "qkv"):
            # the decode step's own pair of launches on either side: the GEMM's K slices through its tail | left as slabs
            ops.gemm_rows_qkv(x, w, qk0, nh, nkv, hd, qn, kn, cos, sin, pos, 1e-6, kc0.view(B * Tmax, nkvd), vc0.view(B * Tmax, nkvd), slot)
            slabs, n_slabs = ops.gemm_rows_slabs(x, w)
            assert n_slabs >= 2
        else:
            # (a shape the streaming decode-row kernel takes: slabs made here — three K slices in fp32 — combined in the kernel's order)
            ks = [0, H // 4, H // 2 + 64, H]
            slabs = torch.stack([x[:, a:b].float() @ w[:, a:b].float().t() for a, b in zip(ks[:-1], ks[1:])]).contiguous()
            n_slabs = 3
            acc = slabs[0].clone()
            acc += slabs[1]
            acc += slabs[2]
            ops.norm_rope_fwd(acc.to(BF), qk0, nh, nkv, hd, 1, qn, kn, cos, sin, positions=pos, eps=1e-6, kcache=kc0.view(B * Tmax, nkvd),
                              vcache=vc0.view(B * Tmax, nkvd), slot=slot)
        ops.attn_decode(qk0, kc0, vc0, out0, lo, hi, B, Tmax, nh, nkv, hd, hd ** -0.5, kv_len_hint=Tmax, workspace=ws)
        ops.attn_decode_qkv(slabs, n_slabs, qn, kn, cos, sin, pos, 1e-6, kc1, vc1, slot, out1, lo, hi, B, Tmax, nh, nkv, hd, hd ** -0.5,
                            kv_len_hint=Tmax, workspace=ws)
    torch.cuda.synchronize()
    assert torch.equal(kc0, kc1) and torch.equal(vc0, vc1)
    assert not torch.equal(kc1, kc)                      # the append happened
    # fp32 reference on the bf16 q and caches the separate launches produced
    q = qk0[:, :nq].float().view(B, nkv, nh // nkv, hd)
    K = kc0.float().view(B, Tmax, nkv, hd).permute(0, 2, 1, 3)
    V = vc0.float().view(B, Tmax, nkv, hd).permute(0, 2, 1, 3)
    sc = torch.einsum("bkgd,bktd->bkgt", q, K) * hd ** -0.5
    t = torch.arange(Tmax, device="cuda")[None, :]
    vis = (t >= lo[:, None]) & (t < hi[:, None])
    sc = sc.masked_fill(~vis[:, None, None, :], float("-inf"))
    ref = torch.einsum("bkgt,bktd->bkgd", torch.softmax(sc, -1), V).reshape(B, nq)
    scale = ref.abs().max().item()
    for out in (out0, out1):
        assert torch.isfinite(out.float()).all()
        assert (out.float() - ref).abs().max().item() <= 1.2e-2 * scale
    assert (out1.float() - out0.float()).abs().max().item() <= 8e-3 * scale
    assert (out1 != out0).float().mean().item() < 0.2


def test_gelu_epilogue_erf_accuracy_over_the_whole_range():
    """The GELU epilogue / kernels use a 1.5e-7-accurate erf (A&S 7.1.26): against torch's erf-GELU on a dense sweep of
    pre-activations (exact zeros, tiny, moderate, saturated tails, both signs) the bf16 results differ by at most one ulp."""
    from molly_amd import ops
    z = torch.cat([torch.linspace(-9, 9, 1 << 16), torch.tensor([0.0, -0.0, 1e-6, -1e-6, 30.0, -30.0, 0.5, -0.5]),
                   torch.randn(1 << 14) * 3]).cuda().bfloat16()
    z = z[: z.numel() // 8 * 8]
    got = ops.gelu_fwd(z).float()
    ref = torch.nn.functional.gelu(z.float()).bfloat16().float()
    ulp = 2.0 ** (torch.floor(torch.log2(ref.abs().clamp(min=1e-30))) - 7)
    assert ((got - ref).abs() <= ulp + 1e-30).all()
    assert (got != ref).float().mean().item() < 1e-2          # the flips sit in the negative tail, where 1 + erf cancels
    dz = ops.gelu_bwd(z, torch.ones_like(z)).float()
    zr = z.float().requires_grad_(True)
    torch.nn.functional.gelu(zr).sum().backward()
    assert (dz - zr.grad).abs().max().item() <= 8e-3


@pytest.mark.parametrize("persist", [256, 0, 8])
@pytest.mark.parametrize("accumulate", [False, True])
def test_gemm_grouped_equals_separate_launches(accumulate, persist):
    """Grouped launch (the four weight-gradient GEMMs of a layer in one persistent launch, mixed transposed / plain outputs,
    ragged sizes): bit-identical to the unsplit single-problem kernel on every problem, whatever was in the outputs before."""
    K = 1024 + 64
    shapes = [(512, 768, True), (256, 264, False), (1000, 512, True), (260, 1032, False)]       # (M, N, trans_out)
    probs, refs = [], []
    for i, (M, N, to) in enumerate(shapes):
        a = _rand(M, K, seed=70 + i).to(BF)
        b = _rand(K, N, seed=80 + i).to(BF)
        init = _rand(*((N, M) if to else (M, N)), seed=90 + i).to(BF)
        out = init.clone()
        ref = init.clone()
        lib().call("molly_gemm_force_tile", 512)                       # the unsplit 256x256 kernel
        ops.gemm(a, b, out=ref, accumulate=accumulate, b_kmajor=True, trans_out=to)
        lib().call("molly_gemm_force_tile", 0)
        probs.append((a, b, out, to))
        refs.append(ref)
    lib().call("molly_gemm_set_persistent_blocks", persist)          # one block per CU / per tile (multi-rank setting) / few
    try:
        ops.gemm_grouped(probs, accumulate=accumulate)
    finally:
        lib().call("molly_gemm_set_persistent_blocks", 256)
    torch.cuda.synchronize()
    for (a, b, out, to), ref in zip(probs, refs):
        assert torch.equal(out, ref), (out.float() - ref.float()).abs().max().item()


@pytest.mark.parametrize("form", ["nt", "nn", "to"])
def test_gemm_bf16_accumulate_is_one_rounding_of_the_fp32_sum(form):
    """MOLLY_GEMM_ACCUMULATE into a bf16 output (the weight gradients of the second micro-batch under gradient accumulation): the interior tiles'
    16-byte read-add-write epilogues and the ragged edge's generic one both give bf16(fp32 accumulators + old value) — compared bit for bit with the
    same launch's fp32 output added to the old values outside the kernel."""
    lib().call("molly_gemm_force_tile", 512)
    try:
        for M, N, K in ((512, 768, 512), (520, 776, 576)):                  # whole tiles; a ragged last tile row and column
            a = _rand(M, K, seed=130).to(BF)
            if form == "nt":
                b = _rand(N, K, seed=131).to(BF)
                f32 = ops.gemm_nt(a, b, out_dtype=torch.float32)
                base = _rand(M, N, seed=132).to(BF)
                out = base.clone()
                ops.gemm_nt(a, b, out=out, accumulate=True)
            else:
                b = _rand(K, N, seed=131).to(BF)
                to = form == "to"
                f32 = ops.gemm(a, b, b_kmajor=True, trans_out=to, out_dtype=torch.float32)
                base = _rand(*((N, M) if to else (M, N)), seed=132).to(BF)
                out = base.clone()
                ops.gemm(a, b, out=out, accumulate=True, b_kmajor=True, trans_out=to)
            ref = (f32 + base.float()).to(BF)
            assert torch.equal(out, ref), (form, M, N, (out.float() - ref.float()).abs().max().item())
    finally:
        lib().call("molly_gemm_force_tile", 0)


@pytest.mark.parametrize("epilogue", ["plain", "res", "swiglu"])
def test_gemm_k_extension_equals_the_sum_of_two_products(epilogue):
    """molly_gemm_kx_bf16_ctx: C = A B^T + A2 B2^T in ONE accumulation (the LoRA branch riding in the base projection as trailing K-tiles).
    Against fp32 on the full output; and against the two-launch form it replaces (base GEMM, then an accumulating K2-deep GEMM) within the one
    extra bf16 rounding that form has.  Shapes: whole tiles and a ragged last tile row / column; K2 = 64 and 192; row-strided A2 / B2 (column
    slices of wider buffers, as the block-diagonal stack of a fused projection's adapters is)."""
    for M, N, K, K2 in ((5120 + 256, 2048 + 512, 512, 64), (8192 + 40, 2048 + 256, 256, 192)):       # (>= 200 tiles: the launch's own rule)
        if epilogue == "swiglu":
            N = (N // 256) * 256
        if not ops.gemm_kx_supported(M, N, K, K2, swiglu=epilogue == "swiglu", res=epilogue == "res"):
            pytest.fail(f"the K-extended launch refuses {M} x {N} x {K} + {K2}")
        a, b = _rand(M, K, seed=140).to(BF), (_rand(N, K, seed=141) * 0.5).to(BF)
        a2w, b2w = _rand(M, K2 + 64, seed=142).to(BF), (_rand(N, K2 + 128, seed=143) * 0.5).to(BF)
        a2, b2 = a2w[:, 64:], b2w[:, 64:64 + K2]
        exact = a.float() @ b.float().T + a2.float() @ b2.float().T
        out = torch.empty(M, N, dtype=BF, device=DEV)
        if epilogue == "swiglu":
            act = torch.empty(M, N // 2, dtype=BF, device=DEV)
            ops.gemm_nt_kx(a, b, a2, b2, out, act=act)
            _close(out, exact, atol=2e-2 * math.sqrt((K + K2) / 64), rtol=8e-3, what="kx gate|up")
            g, u = out[:, :N // 2].float(), out[:, N // 2:].float()
            ref_act = (torch.nn.functional.silu(g).to(BF).float() * u)
            _close(act, ref_act, atol=2e-2, rtol=1e-2, what="kx activation")
            continue
        res = _rand(M, N, seed=144).to(BF) if epilogue == "res" else None
        ops.gemm_nt_kx(a, b, a2, b2, out, res=res)
        ref = exact + (res.float() if res is not None else 0.0)
        _close(out, ref, atol=2e-2 * math.sqrt((K + K2) / 64), rtol=8e-3, what="kx vs fp32")
        two = ops.gemm_nt(a, b, res=res)
        ops.gemm_nt(a2, b2, out=two, accumulate=True)
        # (the two-launch form rounds the base product first: its error is an ulp of THAT magnitude, wherever the branch cancels it)
        mag = torch.maximum(ref.abs(), (a.float() @ b.float().T + (res.float() if res is not None else 0.0)).abs())
        ulp = 2.0 ** (torch.floor(torch.log2(mag.clamp(min=1e-30))) - 7)
        assert bool(((out.float() - two.float()).abs() <= 2 * ulp + 1e-30).all())
        # one accumulation is at least as close to fp32 as two roundings
        assert (out.float() - ref).abs().mean().item() <= (two.float() - ref).abs().mean().item() * 1.02


def test_argmax_matches_torch_first_maximum():
    g = torch.Generator(device="cpu").manual_seed(5)
    for rows, V in ((32, 151936), (3, 1000), (5, 97)):
        x = torch.randn(rows, V, generator=g).to(DEV)
        x[0, 17] = x[0].max() + 1.0
        x[0, 900 % V] = x[0, 17]                      # tie: the FIRST maximal index wins
        x[1] = 0.25                                   # a constant row -> index 0
        if rows > 2:
            x[2, V - 1] = float("inf")
        got = ops.argmax(x)
        assert torch.equal(got, x.argmax(-1)), (got.tolist()[:5], x.argmax(-1).tolist()[:5])
    y = torch.randn(4, 2048, generator=g).to(DEV)
    y[3, 77] = float("nan")
    assert ops.argmax(y)[3].item() == 77              # NaN counts as the maximum (torch semantics)
    wide = torch.randn(6, 4096, generator=g).to(DEV)
    view = wide[:, :1000]                              # row pitch != V
    assert torch.equal(ops.argmax(view), view.argmax(-1))


def test_reduce_rows_is_the_fp32_sum_in_row_order():
    """molly_reduce_rows_bf16 (local half of the all-to-all reduce-scatter): bit-exact against an fp32 accumulation in row order."""
    g = torch.Generator(device=DEV).manual_seed(5)
    for rows, n in ((8, 1 << 20), (2, 4096), (5, 8)):
        x = (torch.randn(rows, n, device=DEV, generator=g) * 3).to(BF)
        out = torch.empty(n, dtype=BF, device=DEV)
        ops.reduce_rows(x, out)
        acc = torch.zeros(n, dtype=torch.float32, device=DEV)
        for r in range(rows):
            acc += x[r].float()
        assert torch.equal(out, acc.to(BF))


@pytest.mark.parametrize("M,ff,K", [(300, 256, 128), (1024, 3072, 1024), (16384, 6144, 2048)])
def test_gate_up_gemm_with_fused_swiglu_is_bit_identical_to_two_kernels(M, ff, K):
    """MOLLY_GEMM_SWIGLU: gu = x W_gu^T and act = silu(gate) * up from one launch (the B tile interleaves gate and up rows so a
    wave owns both values of every element) == molly_gemm + molly_swiglu_fwd, bit for bit, ragged M included."""
    x = _rand(M, K, seed=61).to(BF)
    w = _rand(2 * ff, K, seed=62, scale=0.05).to(BF)
    gu_ref = ops.gemm_nt(x, w)
    act_ref = ops.swiglu_fwd(gu_ref)
    gu = torch.full((M, 2 * ff), 7.0, dtype=BF, device=DEV)
    act = torch.full((M, ff), 7.0, dtype=BF, device=DEV)
    ops.gemm_gate_up_swiglu(x, w, gu, act)
    torch.cuda.synchronize()
    assert torch.equal(gu, gu_ref)
    assert torch.equal(act, act_ref)


@pytest.mark.parametrize("M,ff,K", [(300, 200, 128), (1024, 3072, 1024), (16384, 6144, 2048)])
def test_down_dgrad_with_fused_swiglu_bwd_is_bit_identical_to_two_kernels(M, ff, K):
    """MOLLY_GEMM_SWIGLU_BWD: d[gate | up] from the down-projection's dgrad launch == molly_gemm (dgrad form) + molly_swiglu_bwd,
    bit for bit (d(act) rounded to bf16 in the epilogue exactly as the GEMM would have stored it), ragged M and ff included;
    against torch autograd within the bf16 tolerance of the unfused test."""
    dy = _rand(M, K, seed=71).to(BF)
    w = _rand(K, ff, seed=72, scale=0.05).to(BF)                 # down_proj.weight [h, ff]: the k-major B operand
    gu = _rand(M, 2 * ff, seed=73).to(BF)
    dact_ref = ops.gemm(dy, w, b_kmajor=True)
    dgu_ref = ops.swiglu_bwd(gu, dact_ref)
    dgu = torch.full((M, 2 * ff), 7.0, dtype=BF, device=DEV)
    ops.gemm_down_dgrad_swiglu_bwd(dy, w, gu, dgu)
    torch.cuda.synchronize()
    assert torch.equal(dgu, dgu_ref)
    if M <= 1024:
        gr = gu.float().requires_grad_(True)
        (torch.nn.functional.silu(gr[:, :ff]) * gr[:, ff:]).backward(dact_ref.float())
        _close(dgu, gr.grad, 1e-2, 1e-2, "fused swiglu bwd vs autograd")


@pytest.mark.parametrize("form,M,N,K,res", [("nt", 2048, 4096, 2048, False), ("nt", 4096, 2048, 2048, True),
                                            ("nn", 4096, 6144, 2048, False), ("nn", 1024 + 40, 2048, 4096, False)])
def test_gemm_launch_shapes_are_bit_identical(form, M, N, K, res):
    """The 256x256 kernel launched as one block per CU (persistent, rolling prefetch across its tiles), as one block per tile, and
    as blocks of at most t tiles (what Zero2Optimizer selects at N > 1: molly_gemm_set_persistent_blocks(-3)) computes every tile
    the same way whoever runs it: bit-identical outputs, edge tiles included."""
    from molly_amd._lib import lib
    a = _rand(M, K, seed=81).to(BF)
    b = (_rand(N, K, seed=82, scale=0.05) if form == "nt" else _rand(K, N, seed=82, scale=0.05)).to(BF)
    r = _rand(M, N, seed=83).to(BF) if res else None
    outs = []
    try:
        lib().call("molly_gemm_force_tile", 512)
        for mode in (256, 0, -2, -3):
            lib().call("molly_gemm_set_persistent_blocks", mode)
            out = torch.full((M, N), 7.0, dtype=BF, device=DEV)
            if form == "nt":
                ops.gemm_nt(a, b, out=out, res=r)
            else:
                ops.gemm(a, b, out=out, res=r, b_kmajor=True)
            torch.cuda.synchronize()
            outs.append(out)
    finally:
        lib().call("molly_gemm_set_persistent_blocks", 256)
        lib().call("molly_gemm_force_tile", 0)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    ref = a.float() @ (b.float().t() if form == "nt" else b.float())
    if res:
        ref = ref + r.float()
    _close(outs[0], ref, 2e-2, 2e-2, "gemm vs fp32")


@pytest.mark.parametrize("M,N,K", [(512, 512, 192), (2048, 1024, 256), (8192, 4096, 2048), (16384, 2048, 1024), (4096, 12288, 2048)])
@pytest.mark.parametrize("blocks", [256, 0, -3])
def test_gemm_streaming_epilogue_is_bit_identical(M, N, K, blocks):
    """gemm256_kernel<SE> (round 6; context key stream_epi, on by default): plain NT launches of whole interior tiles run with un-swapped MFMA
    operands, B rows interleaved at staging, and every row half stored as whole 128-byte lines from inside the K loop, which runs on into the next
    tile.  Same products in the same order with the same roundings: the output must equal the plain instantiation's bit for bit — at three
    K-tiles (first / middle / last bodies once each), with several tiles per block (the cross-tile store of the lower row half), and in every
    launch shape — and a canary around the output must survive (a whole-line store one row off would hit it)."""
    a = _rand(M, K, seed=91).to(BF)
    b = _rand(N, K, seed=92, scale=0.05).to(BF)
    outs, cfgs = [], []
    for se in (0, 1):
        ctx = ops.GemmContext()
        ctx.ensure_workspace(64 << 20)
        ctx.set("stream_epi", se)
        ctx.set("persistent_blocks", blocks)
        ctx.set("force_tile", 512)
        buf = torch.full((M + 2, N), 7.0, dtype=BF, device=DEV)
        with ops.use_gemm_context(ctx):
            ops.gemm_nt(a, b, out=buf[1:M + 1])
            cfgs.append(ctx.get("last_config"))
        torch.cuda.synchronize()
        assert bool((buf[0] == 7.0).all()) and bool((buf[M + 1] == 7.0).all()), "the rows around the output were written"
        outs.append(buf[1:M + 1].clone())
    assert cfgs == [1512, 1514], cfgs                             # 512 + 1000 x one K slice; + 2: the streaming epilogue
    assert torch.equal(outs[0], outs[1])
    _close(outs[1], a.float() @ b.float().t(), 2e-2, 2e-2, "gemm (streaming epilogue) vs fp32")


def test_gemm_streaming_epilogue_applies_only_where_it_is_built():
    """Ragged tiles, an epilogue flag, a k-major operand or a K slice keep the plain instantiation (last_config says which ran)."""
    ctx = ops.GemmContext()
    ctx.ensure_workspace(64 << 20)
    ctx.set("force_tile", 512)
    with ops.use_gemm_context(ctx):
        a, b = _rand(512, 256, seed=1).to(BF), _rand(512, 256, seed=2).to(BF)
        ops.gemm_nt(a, b)
        assert ctx.get("last_config") == 1514
        ops.gemm_nt(a, b, res=_rand(512, 512, seed=3).to(BF))
        assert ctx.get("last_config") == 1512                      # residual add
        ops.gemm_nt(a[:500], b)
        assert ctx.get("last_config") == 1512                      # ragged M
        ops.gemm(a, _rand(256, 512, seed=4).to(BF), b_kmajor=True)
        assert ctx.get("last_config") == 1512                      # the dgrad form
        ops.gemm_nt(a[:, :128], b[:, :128])
        assert ctx.get("last_config") == 1512                      # two K-tiles
    torch.cuda.synchronize()


@pytest.mark.parametrize("M", [1, 7, 16, 32, 33, 64])
@pytest.mark.parametrize("N,K", [(4096, 4096), (1000, 1280), (256, 256), (6144, 2560), (12288, 2048)])
def test_gemm_decode_row_kernel(M, N, K):
    """M <= 64 forward GEMMs (the decode step's projections, HF:models/qwen3/modeling_qwen3.py:76-83 with one token per sample) on the
    weight-streaming kernel: exact on small integers (every epilogue), fp32 accumulation error only on random data, and the
    same numbers as round 2's split-K path up to the order of the K-quarter sums."""
    g = torch.Generator(device="cuda").manual_seed(M * 131 + N + K)
    ints = lambda *s: torch.randint(-3, 4, s, device="cuda", generator=g).to(BF)
    x, w, bias, res = ints(M, K), ints(N, K), ints(N), ints(M, N)
    ref = x.float() @ w.float().t()
    c = ops.GemmContext()
    c.ensure_workspace(64 << 20)
    c.set("rows_tiled", 0)                                      # (the tiled decode-row kernel takes the larger matrices: its own test)
    with ops.use_gemm_context(c):
        out = ops.gemm_nt(x, w, out_dtype=torch.float32)
        assert c.get("last_config") == 1016                      # the decode-row kernel
        assert torch.equal(out, ref)
        assert torch.equal(ops.gemm_nt(x, w, bias=bias, res=res, out_dtype=torch.float32), ref + bias.float() + res.float())
        acc = ints(M, N).float()
        want = acc + ref
        ops.gemm_nt(x, w, out=acc, accumulate=True)
        assert torch.equal(acc, want)
        got = ops.gemm_nt(x, w, bias=bias, gelu=True)
        wg = torch.nn.functional.gelu(ref + bias.float())
        assert (got.float() - wg).abs().max().item() <= 2 ** -7 * wg.abs().max().item()
        xr, wr = _rand(M, K, seed=5).to(BF), _rand(N, K, seed=6).to(BF)
        o1 = ops.gemm_nt(xr, wr, out_dtype=torch.float32)
    _close(o1, xr.float() @ wr.float().t(), atol=2e-3 * math.sqrt(K), rtol=1e-4, what="decode-row kernel fp32")
    off = ops.GemmContext()
    off.ensure_workspace(64 << 20)
    off.set("skinny", 0)
    with ops.use_gemm_context(off):
        o0 = ops.gemm_nt(xr, wr, out_dtype=torch.float32)
        assert off.get("last_config") != 1016
    _close(o1, o0, atol=2e-3 * math.sqrt(K), rtol=1e-4, what="decode-row kernel vs split-K path")


@pytest.mark.parametrize("rows,V,ld", [(32, 151936, 151936), (3, 1001, 1004), (5, 4096, 4100), (1, 7, 7), (4, 50000, 50000)])
def test_argmax_rows_first_maximal_index(rows, V, ld):
    """Greedy token choice (HF generate, do_sample=False): molly_argmax_f32 against numpy's argmax on the same fp32 rows — ties go to the
    FIRST maximal index, a NaN counts as the maximum; 16-byte loads where the row stride allows, scalar ones elsewhere."""
    g = torch.Generator(device="cuda").manual_seed(rows * 7 + V)
    buf = torch.randn(rows, ld, device="cuda", generator=g)
    x = buf[:, :V]
    x[0, V // 3] = 50.0
    x[0, V - 1] = 50.0                                       # a tie: the earlier index wins
    if rows > 1:
        x[1, :] = -3.0                                       # a constant row: index 0
    if rows > 2:
        x[2, V // 2] = float("nan")                          # NaN counts as the maximum
        x[2, 1] = 99.0
    got = ops.argmax(x).cpu().numpy()
    ref = x.cpu().numpy().argmax(-1)
    assert (got == ref).all(), (got, ref)
    assert got[0] == V // 3


@pytest.mark.parametrize("M,N,K,kmajor_b", [(3072, 6144, 2560, False), (3072, 6144, 2560, True), (2816, 6400, 512, False)])
def test_gemm_carves_the_tile_columns_past_whole_rounds(M, N, K, kmajor_b):
    """A grid a few 256 x 256 tiles past whole rounds of the chip (Qwen3-4B q | k | v at one sample per GPU: 288 tiles) runs as two launches — whole
    rounds + the last tile columns K-sliced (gemm.hip launch_gemm): exact on small integers with every epilogue, both B layouts, fp32 and bf16 outputs."""
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    ints = lambda *s: torch.randint(-3, 4, s, device="cuda", generator=g).to(BF)
    x, w, bias, res = ints(M, K), ints(N, K), ints(N), ints(M, N)
    ref = x.float() @ w.float().t()
    b = w.t().contiguous() if kmajor_b else w
    c = ops.GemmContext()
    c.ensure_workspace(256 << 20)
    with ops.use_gemm_context(c):
        out = ops.gemm(x, b, out_dtype=torch.float32, b_kmajor=kmajor_b)
        assert torch.equal(out, ref)
        if K >= 2048:
            assert c.get("last_config") // 1000 >= 2                   # the strip: K slices
        out2 = ops.gemm(x, b, bias=bias, res=res, out_dtype=torch.float32, b_kmajor=kmajor_b)
        assert torch.equal(out2, ref + bias.float() + res.float())
        acc = torch.ones(M, N, dtype=torch.float32, device="cuda")
        ops.gemm(x, b, out=acc, accumulate=True, b_kmajor=kmajor_b)
        assert torch.equal(acc, ref + 1.0)
        small = (x.float() * 0.25).to(BF)                              # bf16 output: values that stay exact
        outb = ops.gemm(small, b, b_kmajor=kmajor_b)
        assert torch.equal(outb.float(), (small.float() @ w.float().t()).to(BF).float())
