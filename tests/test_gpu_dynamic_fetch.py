"""GPU: the dynamic tile fetch of the 256x256 GEMM (molly_amd/csrc/gemm.hip, DYN; MOLLY_GEMM_KEY_DYNAMIC) — 256 resident blocks that
draw their tiles from one ticket counter per XCD label instead of walking a static list; what a rank of a multi-GPU job uses so
that a collective's kernels holding CUs cost their share of the chip and not a whole extra round (the decoder GEMMs of
HF:models/qwen3/modeling_qwen3.py:76-83, 225-236 under DeepSpeed ZeRO-2's overlap_comm, reference configs/ds_z2_config.json).
Checked: bit-identical to the static walk on every form / epilogue / split-K (a tile is computed the same way whoever computes
it), the counters are back at zero after every launch (hundreds of launches back to back, mixed shapes), tile counts that are no
multiple of 8, and a launch beside a kernel that holds CUs (late-starting blocks)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
DYN_CNT_OFF = 64 + 8192 * 64            # include/molly_hip.h: the workspace header


def _ctx(dynamic, **knobs):
    from molly_amd import ops
    c = ops.GemmContext()
    c.ensure_workspace(1 << 28)
    c.set("dynamic", dynamic)
    c.set("streamk", 0)
    c.set("force_tile", 512)                 # every shape here on the 256x256 kernel (a 0.78-round grid would go to the 128x128 one)
    for k, v in knobs.items():
        c.set(k, v)
    return c


def _counters(c):
    torch.cuda.synchronize()
    return c.ws.view(torch.int32)[DYN_CNT_OFF // 4: DYN_CNT_OFF // 4 + 8 * 16: 16].cpu().tolist()


FORMS = {"nt": dict(), "nn": dict(b_kmajor=True), "tn": dict(a_kmajor=True, b_kmajor=True)}


def _operands(form, M, N, K, g):
    mk = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).to(BF)
    return (mk(K, M) if form == "tn" else mk(M, K)), (mk(N, K) if form == "nt" else mk(K, N))


@pytest.mark.parametrize("form", ["nt", "nn", "tn"])
@pytest.mark.parametrize("M,N,K", [(16384, 4096, 2048), (4096, 6144 + 256, 1024), (5000, 5000, 576), (8192, 2304, 512), (2048 + 256, 8192, 4096)])
def test_dynamic_fetch_equals_static_walk(form, M, N, K):
    from molly_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N + K)
    a, b = _operands(form, M, N, K, g)
    on, off = _ctx(1), _ctx(0)
    with ops.use_gemm_context(off):
        ref = ops.gemm(a, b, **FORMS[form])
        assert off.get("last_config") % 1000 in (512, 514)      # (514: the streaming-epilogue instantiation of the plain NT launch)
    with ops.use_gemm_context(on):
        outs = [ops.gemm(a, b, **FORMS[form]) for _ in range(3)]
        assert on.get("last_config") % 1000 == 513, on.get("last_config")
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, ref)
    assert _counters(on) == [0] * 8


def test_dynamic_fetch_epilogues_splitk_and_transposed_output():
    from molly_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    M, N, K = 8192, 4096, 1024
    a, b = _operands("nt", M, N, K, g)
    bias = (torch.rand(N, device="cuda", generator=g) - 0.5).to(BF)
    res = (torch.rand(M, N, device="cuda", generator=g) - 0.5).to(BF)
    a3 = (torch.rand(4096, M, device="cuda", generator=g) - 0.5).to(BF)
    x2 = (torch.rand(M, 4608, device="cuda", generator=g) - 0.5).to(BF)            # 16 x 18 = 288 tiles
    on, off = _ctx(1), _ctx(0)
    got, want = [], []
    for c, dst in ((off, want), (on, got)):
        with ops.use_gemm_context(c):
            dst.append(ops.gemm_nt(a, b, bias=bias, gelu=True, res=res))
            dst.append(ops.gemm_nt(a, b, res=res))
            acc = res.float().clone()
            dst.append(ops.gemm_nt(a, b, out=acc, accumulate=True, out_dtype=torch.float32))
            # weight-gradient form with the transposed output (k-contiguous A, k-major B), C^T stored
            dst.append(ops.gemm(a3, x2, b_kmajor=True, trans_out=True))
            dst.append(c.get("last_config"))
    torch.cuda.synchronize()
    assert got[-1] % 1000 == 513 and want[-1] % 1000 == 512
    for w, o in zip(want[:-1], got[:-1]):
        assert torch.equal(w, o)
    # split-K under the dynamic fetch: 72 tiles x 512 K-tiles, cut into 16 slices by the launcher = 1,152 drawn work items
    a2, b2 = _operands("nt", 2304, 2048, 32768, g)
    on.set("force_tile", 0); off.set("force_tile", 0)
    with ops.use_gemm_context(off):
        w2 = ops.gemm_nt(a2, b2)
        cfg_off = off.get("last_config")
    with ops.use_gemm_context(on):
        o2 = ops.gemm_nt(a2, b2)
        cfg_on = on.get("last_config")
    torch.cuda.synchronize()
    assert torch.equal(w2, o2), (cfg_off, cfg_on)
    assert cfg_off == 512 + 16000 and cfg_on == 513 + 16000, (cfg_off, cfg_on)
    assert _counters(on) == [0] * 8


def test_dynamic_fetch_counters_survive_many_mixed_launches():
    """400 launches back to back over shapes with 264 ... 1024 tiles (tile counts = 0..7 mod 8): every launch must find the counters
    at zero (a launch that did not would skip or repeat tiles) — outputs compared with the static walk at the end."""
    from molly_amd import ops
    g = torch.Generator(device="cuda").manual_seed(9)
    shapes = [(256 * tm, 256 * tn, 512 + 64 * (i % 5)) for i, (tm, tn) in enumerate([(33, 8), (17, 16), (29, 9), (64, 16), (19, 14), (53, 5), (37, 7), (23, 12)])]
    ops_ = [(_operands("nt", M, N, K, g)) for M, N, K in shapes]
    on, off = _ctx(1), _ctx(0)
    with ops.use_gemm_context(off):
        want = [ops.gemm_nt(a, b) for a, b in ops_]
    outs = [torch.empty_like(w) for w in want]
    with ops.use_gemm_context(on):
        for it in range(50):
            for (a, b), o in zip(ops_, outs):
                ops.gemm_nt(a, b, out=o)
                assert on.get("last_config") % 1000 == 513
    torch.cuda.synchronize()
    for w, o in zip(want, outs):
        assert torch.equal(w, o)
    assert _counters(on) == [0] * 8


def test_dynamic_fetch_beside_a_kernel_that_holds_cus():
    """64 CUs held by another stream's kernel for the whole launch: the blocks that find no CU start late and draw whatever is left;
    the result is the static walk's, the counters end at zero."""
    from molly_amd import ops
    from molly_amd._lib import lib
    g = torch.Generator(device="cuda").manual_seed(11)
    a, b = _operands("nt", 16384, 4096, 2048, g)
    on, off = _ctx(1), _ctx(0)
    with ops.use_gemm_context(off):
        want = ops.gemm_nt(a, b)
    side = torch.cuda.Stream()
    sink = torch.zeros(4, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        lib().call("molly_probe_hog", side.cuda_stream, 64, 20000, sink)
    torch.cuda._sleep(200000)
    with ops.use_gemm_context(on):
        outs = [ops.gemm_nt(a, b) for _ in range(4)]
    torch.cuda.synchronize()
    for o in outs:
        assert torch.equal(o, want)
    assert _counters(on) == [0] * 8
