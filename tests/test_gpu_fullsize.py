"""GPU: the kernels at BASELINE configs[1] sizes (Molly-1.7B, T = 2048) through size-independent properties — sampled rows of every
GEMM shape of the step against fp32 matmul, run-to-run bitwise determinism of the whole step (no atomics anywhere), gradient
accumulation = sum, and the algorithmic invariants of attention and the norms — at the batch the headline TIMES (bench.py's default,
imported, not restated: VERDICT r04 'the benchmarked shape is not the tested shape') and at rounds 1-3's 8 samples."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import DEFAULT_BATCH  # noqa: E402

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
M = 16384
BATCHES = sorted({8, DEFAULT_BATCH})
SCORED = lambda b: b * 2048 // 4                                 # rows the lm_head sees: 25 % of the tokens carry a label


def _gemm_cases():
    out = []
    for b in BATCHES:
        m, ms = b * 2048, SCORED(b)
        out += [(f"B{b} qkv fwd", "nt", m, 4096, 2048), (f"B{b} o fwd", "nt", m, 2048, 2048), (f"B{b} gate|up fwd", "nt", m, 12288, 2048),
                (f"B{b} down fwd", "nt", m, 2048, 6144), (f"B{b} lm_head fwd (scored rows)", "nt", ms, 151936, 2048),
                (f"B{b} qkv dgrad", "nn", m, 2048, 4096), (f"B{b} o dgrad", "nn", m, 2048, 2048), (f"B{b} gate|up dgrad", "nn", m, 2048, 12288),
                (f"B{b} down dgrad", "nn", m, 6144, 2048), (f"B{b} lm_head dgrad", "nn", ms, 2048, 151936),
                (f"B{b} gate|up wgrad", "tn", 12288, 2048, m), (f"B{b} lm_head wgrad", "tn", 151936, 2048, ms),
                (f"B{b} qkv wgrad (transposed narrow operand, transposed output)", "to", 4096, 2048, m)]
    return out


@pytest.mark.parametrize("name,form,m,n,k", _gemm_cases())
def test_step_gemm_shapes_sampled_rows_vs_fp32(name, form, m, n, k):
    from molly_amd import ops
    ops.ensure_gemm_workspace(1 << 30)
    g = torch.Generator(device="cuda").manual_seed(len(name))
    rnd = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).bfloat16()
    rows = torch.randint(0, m, (48,), generator=torch.Generator().manual_seed(1)).cuda()
    if form == "nt":
        a, b = rnd(m, k), rnd(n, k)
        out = ops.gemm_nt(a, b)
        ref = a[rows].float() @ b.float().t()
    elif form == "nn":
        a, b = rnd(m, k), rnd(k, n)
        out = ops.gemm(a, b, b_kmajor=True)
        ref = a[rows].float() @ b.float()
    elif form == "tn":
        a, b = rnd(k, m), rnd(k, n)
        out = ops.gemm(a, b, a_kmajor=True, b_kmajor=True)
        ref = a[:, rows].float().t() @ b.float()
    else:                                               # dW^T[n? ...]: out[N, M] = (a[M,K] b[K,N])^T
        a, b = rnd(m, k), rnd(k, n)
        out = ops.gemm(a, b, b_kmajor=True, trans_out=True).t()
        ref = a[rows].float() @ b.float()
    got = out[rows].float()
    err = (got - ref).abs().max().item()
    assert err <= 1.5e-2 * ref.abs().max().item(), (name, err, ref.abs().max().item())


@pytest.mark.parametrize("batch", BATCHES)
def test_grouped_weight_gradient_launch_sampled_rows_vs_fp32(batch):
    """The four weight gradients of a decoder layer as the step launches them: ONE grouped launch over (transposed narrow operand,
    wide operand, output, orientation) with the remainder tiles carved out by qwen3._carve_remainder when the tile count is a few
    past whole rounds of the 256 CUs — at the headline's token count.  Sampled output rows of every problem against fp32."""
    from molly_amd import ops
    from molly_amd.qwen3 import _carve_remainder
    ops.ensure_gemm_workspace(1 << 30)
    Mt = batch * 2048
    g = torch.Generator(device="cuda").manual_seed(batch)
    rnd = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).bfloat16()
    # (dy width N, x width K) of qkv, o, gate|up, down at Qwen3-1.7B: the narrower operand is the one transposed
    probs, refs = [], []
    for n, k in ((4096, 2048), (2048, 2048), (12288, 2048), (2048, 6144)):
        dy, x = rnd(Mt, n), rnd(Mt, k)
        dw = torch.full((n, k), 7.0, dtype=BF, device="cuda")
        if k <= n:
            probs.append((x.t().contiguous(), dy, dw, True))         # dw^T[K, N] = x^T dy, stored transposed into dw
        else:
            probs.append((dy.t().contiguous(), x, dw, False))        # dw[N, K] = dy^T x
        refs.append((dy, x, dw))
    new, carved = _carve_remainder(list(probs))
    ops.gemm_grouped(new, accumulate=False)
    if carved is not None:
        a, b, out, to = carved
        ops.gemm(a, b, out=out, accumulate=False, b_kmajor=True, trans_out=to)
    torch.cuda.synchronize()
    for dy, x, dw in refs:
        rows = torch.randint(0, dw.shape[0], (32,), generator=torch.Generator().manual_seed(dw.shape[0])).cuda()
        rows[-1] = dw.shape[0] - 1                                   # the last tile row (where a carve would sit)
        ref = dy[:, rows].float().t() @ x.float()
        err = (dw[rows].float() - ref).abs().max().item()
        assert err <= 1.5e-2 * ref.abs().max().item(), (tuple(dw.shape), err, ref.abs().max().item())


def _molly_17b():
    import molly_amd
    from molly_amd import config as C
    cfg = C.molly("1.7b", k_tokens=512)
    m = molly_amd.OmicsOne(cfg)
    m.model = molly_amd.Qwen3ForCausalLM(cfg.text_config)
    m.dna_rna_model = molly_amd.EsmForMaskedLM(cfg.dna_rna_config)
    m.protein_model = molly_amd.EsmForMaskedLM(cfg.protein_config)
    m.prepare("cuda", random_init_seed=1234)
    return m


@pytest.mark.parametrize("batch", BATCHES)
def test_full_size_step_is_bitwise_deterministic_and_accumulation_adds(batch):
    from molly_amd.synth import synth_batch
    m = _molly_17b()
    b = synth_batch(batch, 2048, [("protein", 512)], seed=42)
    args = [b[k] for k in ("input_ids", "attention_mask", "omic_ids", "omic_info_list", "labels")]
    l1 = m.forward_backward(*args).clone()
    g1 = m._rt.G.flat.clone()
    for _ in range(5):                                                     # sporadic reorderings show up within a few runs
        l2 = m.forward_backward(*args).clone()
        assert torch.equal(l1, l2) and torch.equal(g1, m._rt.G.flat)      # no atomics, fixed reduction orders
    assert torch.isfinite(g1.float()).all() and 10.0 < l1.item() < 13.0    # ~ln(V) at random init
    m.forward_backward(*args, accumulate=True)
    torch.cuda.synchronize()
    # accumulate adds into bf16 (a tensor fed by two kernels — the tied embedding: head wgrad + gather gradient — rounds
    # twice): a few bf16 roundings of the doubled value, never more
    d = (m._rt.G.flat.float() - 2 * g1.float()).abs()
    bound = 2 ** -6 * (2 * g1.float()).abs() + 2 ** -8 * g1.float().abs().max()
    assert (d <= bound).all(), (d / bound).max().item()
    # rows whose label is ignored never reach the head: the embedding rows of tokens that only occur in the prompt get
    # their gradient from the input side only -> the tied matrix has no NaN/Inf and is not all zero
    ge = m._rt.G.views["model.model.embed_tokens.weight"]
    assert torch.count_nonzero(ge) > 0


def test_full_size_attention_invariants():
    """Softmax rows sum to one: with V = ones the output is exactly one for every live query; causal: moving a later key
    does not change earlier queries."""
    from molly_amd import ops
    B, T, nh, nkv, hd = 8, 2048, 16, 8, 128
    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B * T, (nh + 2 * nkv) * hd, device="cuda", generator=g).bfloat16()
    q, k, v = qkv[:, :nh * hd], qkv[:, nh * hd:(nh + nkv) * hd], qkv[:, (nh + nkv) * hd:]
    v.fill_(1.0)
    o, lse = ops.attn_fwd(q, k, v, B, T, nh, nkv, hd, hd ** -0.5, True)
    assert (o.float() - 1.0).abs().max().item() <= 2 ** -7
    v.copy_(torch.randn(B * T, nkv * hd, device="cuda", generator=g).bfloat16())
    o1, _ = ops.attn_fwd(q, k, v, B, T, nh, nkv, hd, hd ** -0.5, True)
    k2 = k.clone()
    k2.view(B, T, -1)[:, 1500:] += 1.0
    o2, _ = ops.attn_fwd(q, k2, v, B, T, nh, nkv, hd, hd ** -0.5, True)
    assert torch.equal(o1.view(B, T, -1)[:, :1500], o2.view(B, T, -1)[:, :1500])
    assert not torch.equal(o1.view(B, T, -1)[:, 1500:], o2.view(B, T, -1)[:, 1500:])


def test_full_size_norm_and_optimizer_invariants():
    from molly_amd import ops
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(M, 2048, device="cuda", generator=g).bfloat16()
    w = torch.ones(2048, dtype=BF, device="cuda")
    y = ops.rmsnorm_fwd(x, w, 1e-6)
    assert (y.float().pow(2).mean(-1) - 1.0).abs().max().item() < 2e-2      # unit rms rows
    y2 = ops.rmsnorm_fwd((x.float() * 4).bfloat16(), w, 1e-6)               # scale invariance up to eps = 1e-6: a handful
    neq = y != y2                                                           # of bf16 roundings may flip by one ulp
    assert neq.float().mean().item() < 1e-3
    assert ((y.float() - y2.float()).abs() <= 2 ** -7 * y.float().abs() + 1e-6).all()
    # AdamW with zero gradient and zero weight decay is the identity on the master weights; with decay it is p *= 1 - lr*wd
    n = 1 << 24
    p = torch.randn(n, device="cuda", generator=g)
    mst, m1, v1 = p.clone(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    grad = torch.zeros(n, dtype=BF, device="cuda")
    outp = torch.empty(n, dtype=BF, device="cuda")
    ops.adamw_step(mst, m1, v1, grad, outp, 3e-5, 0.9, 0.999, 1e-8, 0.0, 1)
    assert torch.equal(mst, p) and torch.equal(outp, p.bfloat16())
    ops.adamw_step(mst, m1, v1, grad, outp, 3e-5, 0.9, 0.999, 1e-8, 1e-2, 2)
    assert torch.allclose(mst, p * (1 - 3e-5 * 1e-2), rtol=1e-6, atol=0)
