import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import torch
import test_gpu_kernels as t
from molly_amd import ops
BF = torch.bfloat16
for n_keys in (37, 300, 1500):
    B, nh, nkv, hd, H = 32, 32, 8, 128, 4096
    Tmax = n_keys + 8
    x, w, qn, kn, cos, sin, kc, vc, lo, hi, pos, slot = t._decode_qkv_case(B, nh, nkv, hd, H, Tmax, n_keys, seed=21)
    nq, nkvd = nh * hd, nkv * hd
    c = ops.GemmContext(); c.ensure_workspace(64 << 20)
    ws = ops.attn_decode_workspace(B, nh, hd, "cuda")
    with ops.use_gemm_context(c):
        kc0, vc0, kc1, vc1 = kc.clone(), vc.clone(), kc.clone(), vc.clone()
        qk0 = torch.empty(B, nq + nkvd, dtype=BF, device="cuda")
        out1 = torch.empty(B, nq, dtype=BF, device="cuda")
        ops.gemm_rows_qkv(x, w, qk0, nh, nkv, hd, qn, kn, cos, sin, pos, 1e-6, kc0.view(B * Tmax, nkvd), vc0.view(B * Tmax, nkvd), slot)
        slabs, n_slabs = ops.gemm_rows_slabs(x, w)
        ops.attn_decode_qkv(slabs, n_slabs, qn, kn, cos, sin, pos, 1e-6, kc1, vc1, slot, out1, lo, hi, B, Tmax, nh, nkv, hd, hd ** -0.5, kv_len_hint=Tmax, workspace=ws)
    torch.cuda.synchronize()
    for name, a, b_ in (("k", kc0, kc1), ("v", vc0, vc1)):
        d = (a != b_)
        print(n_keys, name, "differing elements", int(d.sum()), "rows(b)", d.any(-1).any(-1).nonzero().flatten().tolist()[:40],
              "positions", d.any(-1).any(0).nonzero().flatten().tolist()[:10], "lo", lo.tolist()[:6], "n_slabs", n_slabs)
        if d.any():
            bb = int(d.any(-1).any(-1).nonzero()[0]); tt = int(d[bb].any(-1).nonzero()[0])
            cols = d[bb, tt].nonzero().flatten().tolist()
            print("   first: b", bb, "t", tt, "cols", cols[:8], "...", len(cols), a[bb, tt, cols[:4]].tolist(), b_[bb, tt, cols[:4]].tolist())
