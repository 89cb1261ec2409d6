"""GPU: the stream-K form of the 256x256 GEMM (molly_amd/csrc/gemm.hip, SKM) — the launch that replaces the 128x128 kernel and the
split-K slabs + reduce launch for every grid with M, N >= 256 that does not fill whole rounds of the 256 CUs (the encoders'
projections, HF:models/esm/modeling_esm.py:350-463; the decoder GEMMs of a B = 1 micro-batch,
scripts/train/examples/run_train_4B_z2_b1.sh:29).  Checked: exact on small integers (every form, every epilogue, ragged edges),
equal to the one-pass kernel up to the fp32 order of the K-range sums on random data, bit-identical from run to run and across
launch shapes that cut the work identically, flags left clean (a second launch right behind the first), no wait ever gave up."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _ctx(**knobs):
    from molly_amd import ops
    c = ops.GemmContext()
    c.ensure_workspace(0)
    knobs.setdefault("streamk", 2)            # 2 = stream-K wherever it is able to run (the default, 1, asks the cost model)
    knobs.setdefault("rows_max_m", 64)        # (small grids at <= 1,024 rows otherwise run as 64-row tiles of the decode-row kernel)
    for k, v in knobs.items():
        c.set(k, v)
    return c


def _ints(shape, g, lo=-3, hi=4):
    return torch.randint(lo, hi, shape, device="cuda", generator=g).to(BF)


FORMS = {"nt": dict(), "nn": dict(b_kmajor=True), "tn": dict(a_kmajor=True, b_kmajor=True)}


def _operands(form, M, N, K, g, ints):
    mk = (lambda *s: _ints(s, g)) if ints else (lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).to(BF))
    a = mk(K, M) if form == "tn" else mk(M, K)
    b = mk(N, K) if form == "nt" else mk(K, N)
    ref = (a.float().t() if form == "tn" else a.float()) @ (b.float().t() if form == "nt" else b.float())
    return a, b, ref


@pytest.mark.parametrize("form", ["nt", "nn", "tn"])
@pytest.mark.parametrize("M,N,K", [(4096, 1280, 1280), (4096, 5120, 1280), (4096, 1280, 5120), (3072, 6144, 2560), (1000, 1032, 832),
                                   (1024, 768, 512), (2304, 4352, 192 * 3), (32800, 4096, 512), (304, 3000, 16384)])
def test_streamk_exact_on_small_integers(form, M, N, K):
    """Integer operands: every product and every partial sum is exact in fp32, so stream-K must reproduce the fp32 reference
    bit for bit whatever the cut (tiles split 2-6 ways, ragged edges, K-tile counts that do not divide; 32800 x 4096 = 2064 tiles:
    more tiles than blocks could ever be — the counters are per TILE; found by the config-5 test, whose re-forward of a 1025-token
    batch puts exactly this o_proj shape through stream-K)."""
    if form != "nt" and M > 8192:
        pytest.skip("the many-tile case runs once")
    # (304 x 3000 x 16384: 24 tiles of 256 K-tiles on 256 CUs — found by tools/gemm_diag/fuzz_gemm.py: a tile may be cut into at
    # most 8 pieces, the launcher sizes the grid so)
    from molly_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a, b, ref = _operands(form, M, N, K, g, ints=True)
    c = _ctx()
    with ops.use_gemm_context(c):
        out = ops.gemm(a, b, out_dtype=torch.float32, **FORMS[form])
        assert c.get("last_config") // 1000 >= 50, c.get("last_config")       # it did take the stream-K launch
        out2 = ops.gemm(a, b, out_dtype=torch.float32, **FORMS[form])          # the flags were left clean
    torch.cuda.synchronize()
    assert torch.equal(out, ref) and torch.equal(out2, ref)
    assert c.streamk_timeouts() == 0


def test_streamk_is_chosen_where_it_pays_and_only_there():
    """The launcher's cost model (gemm.hip launch_cfg): the hand-off costs ~30 us per launch on this chip, so the default
    context takes stream-K for long contractions on grids just past a whole round (Qwen3-8B / 4B qkv at B = 1) and leaves the
    encoders' 20-K-tile projections to the 128x128 kernel / split-K."""
    from molly_amd import ops
    g = torch.Generator(device="cuda").manual_seed(2)
    c = ops.GemmContext()
    c.ensure_workspace(1 << 30)
    # (3,072 x 6,144 x 2,560 — 288 tiles, 32 past a round — was stream-K's until round 4 carved such remainders into a launch of their own)
    want = {(4096, 6144, 4096): True, (3072, 6144, 2560): False, (4096, 1280, 1280): False, (4096, 5120, 1280): False,
            (16384, 4096, 2048): False}
    for (M, N, K), sk in want.items():
        a, b, _ = _operands("nt", M, N, K, g, ints=False)
        with ops.use_gemm_context(c):
            ops.gemm(a, b)
        assert (c.get("last_config") // 1000 >= 50) == sk, (M, N, K, c.get("last_config"))


def test_streamk_epilogues_and_transposed_output():
    """bias + erf-GELU, bias + residual, accumulate (bf16 and fp32) and the transposed-output weight-gradient form through the
    reducer's generic store: against the one-pass kernel (stream-K off) on the same operands; integers -> bit-exact."""
    from molly_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    M, N, K = 4096, 1280, 1280
    a, w = _ints((M, K), g), _ints((N, K), g)
    bias, res = _ints((N,), g), _ints((M, N), g)
    on, off = _ctx(), _ctx(streamk=0)
    for kw in (dict(bias=bias, gelu=True), dict(bias=bias, res=res), dict(res=res), dict()):
        outs = []
        for c in (on, off):
            with ops.use_gemm_context(c):
                outs.append(ops.gemm_nt(a, w, **kw))
        assert on.get("last_config") // 1000 >= 50 and off.get("last_config") // 1000 < 50
        assert torch.equal(outs[0], outs[1]), kw.keys()
    for dt in (BF, torch.float32):
        base = _ints((M, N), g).to(dt)
        outs = []
        for c in (on, off):
            o = base.clone()
            with ops.use_gemm_context(c):
                ops.gemm_nt(a, w, out=o, accumulate=True)
            outs.append(o)
        assert torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], base)
    # dW^T = x^T dy stored transposed (MOLLY_GEMM_TRANS_OUT): A = x [tok, K'] k-contiguous?  the wgrad call of qwen3.py: A [M=K', tok]
    tok, n_out, k_in = 2048, 1280, 2304
    xt, dy = _ints((k_in, tok), g), _ints((tok, n_out), g)                     # A = x^T [k_in][tok] (k-contiguous), B = dy [tok][n_out] (k-major)
    outs = []
    for c in (on, off):
        with ops.use_gemm_context(c):
            outs.append(ops.gemm(xt, dy, b_kmajor=True, trans_out=True))
    assert on.get("last_config") // 1000 >= 50
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], (xt.float() @ dy.float()).t().to(BF))
    assert on.streamk_timeouts() == 0


@pytest.mark.parametrize("form,M,N,K", [("nt", 4096, 1280, 1280), ("nt", 4096, 5120, 1280), ("nt", 4096, 1280, 5120),
                                       ("nn", 3072, 2560, 6144), ("nt", 3072, 6144, 2560), ("tn", 2560, 1280, 4096)])
def test_streamk_random_data_close_to_one_pass_and_reproducible(form, M, N, K):
    """Random data: the K ranges of a tile are summed piece by piece instead of in one chain, so the result may differ from the
    one-pass kernel in the last fp32 bits (a bf16 output: at most one rounding step on a few elements).  It must not differ from
    ITSELF: same launch twice, and a launch with twice the blocks cut at the same places is a different sum order and is allowed
    to differ, but each is reproducible."""
    from molly_amd import ops
    g = torch.Generator(device="cuda").manual_seed(11)
    a, b, ref = _operands(form, M, N, K, g, ints=False)
    on, off = _ctx(), _ctx(streamk=0)
    with ops.use_gemm_context(on):
        o1 = ops.gemm(a, b, **FORMS[form])
        o2 = ops.gemm(a, b, **FORMS[form])
        f1 = ops.gemm(a, b, out_dtype=torch.float32, **FORMS[form])
    with ops.use_gemm_context(off):
        o0 = ops.gemm(a, b, **FORMS[form])
    assert on.get("last_config") // 1000 >= 50
    assert torch.equal(o1, o2)
    scale = ref.abs().max().item()
    assert (f1 - ref).abs().max().item() <= 2e-5 * scale * (K ** 0.5)       # fp32 accumulation error only
    d = (o1.float() - o0.float()).abs()
    assert d.max().item() <= 2 ** -7 * scale and (d > 0).float().mean().item() < 0.02
    for mode in (0, -3, 512):
        c = _ctx(persistent_blocks=mode)
        with ops.use_gemm_context(c):
            p1 = ops.gemm(a, b, **FORMS[form])
            p2 = ops.gemm(a, b, **FORMS[form])
        assert torch.equal(p1, p2), mode
        assert (p1.float() - o0.float()).abs().max().item() <= 2 ** -7 * scale
        assert c.streamk_timeouts() == 0
    assert on.streamk_timeouts() == 0


def test_streamk_under_uneven_load_with_warm_caches():
    """The hand-off under the conditions that expose a wrong protocol (guide 6 G16 pitfall 3): many launches back to back, slabs
    and counters warm in every cache from the launch before, CUs held by another stream's kernel so that blocks start at
    different times — every launch must equal the first, on every element."""
    from molly_amd import ops
    from molly_amd._lib import lib
    g = torch.Generator(device="cuda").manual_seed(3)
    a, b, _ = _operands("nt", 4096, 1280, 1280, g, ints=False)
    a2, b2, _ = _operands("nt", 3072, 6144, 2560, g, ints=False)
    c = _ctx()
    with ops.use_gemm_context(c):
        want, want2 = ops.gemm(a, b), ops.gemm(a2, b2)
        side = torch.cuda.Stream()
        sink = torch.zeros(16, dtype=torch.int32, device="cuda")
        for it in range(40):
            if it % 4 == 1:                                       # hold 24 CUs for ~0.3 ms on another stream
                with torch.cuda.stream(side):
                    lib().call("molly_probe_hog", side.cuda_stream, 24, 300, sink)
            got, got2 = ops.gemm(a, b), ops.gemm(a2, b2)
            assert torch.equal(got, want) and torch.equal(got2, want2), it
    torch.cuda.synchronize()
    assert c.streamk_timeouts() == 0
