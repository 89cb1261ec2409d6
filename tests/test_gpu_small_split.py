"""GPU: small grids with long contractions go through split-K by price (gemm.hip launch_cfg, MOLLY_GEMM_KEY_SMALL_SPLIT) — the
encoders' ffn2 at one sample per GPU (HF:models/esm/modeling_esm.py:440-463 at 512 / 1024 rows: BASELINE configs 3 / 4 / 5).
Exact on integer operands with every epilogue the encoders use, and the rule leaves the headline's shapes where they were."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _ctx(small_split):
    from molly_amd import ops
    c = ops.GemmContext()
    c.ensure_workspace(256 << 20)
    c.set("small_split", small_split)
    c.set("streamk", 0)
    c.set("rows_max_m", 64)                  # (up to 1,024 rows the 64-row tiles of the decode-row kernel take these grids by default: test_gpu_kernels.py)
    return c


@pytest.mark.parametrize("M,N,K", [(512, 1280, 5120), (1024, 1280, 5120), (1000, 1280, 5120), (512, 1024, 4096), (512, 2560, 9728)])
def test_small_long_k_grid_takes_split_k_and_is_exact(M, N, K):
    from molly_amd import ops
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randint(-3, 4, (M, K), device="cuda", generator=g).to(BF)
    b = torch.randint(-3, 4, (N, K), device="cuda", generator=g).to(BF)
    bias = torch.randint(-3, 4, (N,), device="cuda", generator=g).to(BF)
    res = torch.randint(-3, 4, (M, N), device="cuda", generator=g).to(BF)
    ref = a.float() @ b.float().t() + bias.float() + res.float()
    on, off = _ctx(1), _ctx(0)
    with ops.use_gemm_context(on):
        out = ops.gemm_nt(a, b, bias=bias, res=res, out_dtype=torch.float32)
        cfg_on = on.get("last_config")
    with ops.use_gemm_context(off):
        out0 = ops.gemm_nt(a, b, bias=bias, res=res, out_dtype=torch.float32)
        cfg_off = off.get("last_config")
    torch.cuda.synchronize()
    assert cfg_on % 1000 == 512 and cfg_on // 1000 >= 4, cfg_on          # split-K, slices of a few K-tiles
    assert cfg_off == 1128, cfg_off                                      # round 2: the 128x128 kernel
    assert torch.equal(out, ref) and torch.equal(out0, ref)


@pytest.mark.parametrize("M,N,K", [(4096, 1280, 1280), (4096, 5120, 1280), (4096, 1280, 5120), (512, 1280, 1280), (512, 5120, 1280),
                                   (16384, 4096, 2048), (3072, 2560, 4096)])
def test_rule_leaves_other_shapes_alone(M, N, K):
    from molly_amd import ops
    g = torch.Generator(device="cuda").manual_seed(1)
    a = (torch.rand(M, K, device="cuda", generator=g) - 0.5).to(BF)
    b = (torch.rand(N, K, device="cuda", generator=g) - 0.5).to(BF)
    on, off = _ctx(1), _ctx(0)
    with ops.use_gemm_context(on):
        o1 = ops.gemm_nt(a, b)
    with ops.use_gemm_context(off):
        o0 = ops.gemm_nt(a, b)
    assert on.get("last_config") == off.get("last_config")
    assert torch.equal(o0, o1)
