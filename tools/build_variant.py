#!/usr/bin/env python3
"""Build a variant of libmolly_hip.so with extra -D flags, for same-box A/B runs (MOLLY_LIB_PATH=<variant> python bench.py ...).
    python tools/build_variant.py nt0 -DMOLLY_NT_LOAD=0 -DMOLLY_NT_STORE=0      ->  tools/variants/libmolly_nt0.so
    python tools/build_variant.py nostore --patch nostore                       ->  a timing-only text patch of gemm.hip (PATCHES below)
The variant libraries are git-ignored (*.so) and travel to the GPU box with the snapshot."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from molly_amd import build as B  # noqa: E402


def _patch_nostore(t):
    """timing-only: the plain epilogue of gemm256_kernel converts its accumulators but never stores them."""
    t = t.replace("exact_stores = em0 + 256 <= eM", "exact_stores = false && em0 + 256 <= eM")
    k = t.index("// ---- epilogue: lane owns C[m")
    st = "                *reinterpret_cast<u32x2*>(c) = u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};"
    k2 = t.index(st, k)
    return t[:k2] + "                if (v[0] == 1.2345e30f) " + st.strip() + t[k2 + len(st):]


def _patch_lateprefetch(t):
    """valid: the next tile's K-tiles 0 and 1 are issued AFTER the epilogue's stores (so no store sits in front of a load in the
    in-order vmcnt queue) instead of before them."""
    a = """    if (more) {
        decode(vnext);
        issue(0, 0); issue(0, 1); issue(0, 2); issue(0, 3);
        if (nk > 1) { issue(1, 0); issue(1, 1); issue(1, 2); issue(1, 3); }
    }
"""
    assert t.count(a) == 1
    t = t.replace(a, "")
    b = "    if (!more) break;\n    vcur = vnext;"
    assert t.count(b) == 1
    t = t.replace(b, "    if (!more) break;\n" + a.replace("    if (more) {", "    {") + "    vcur = vnext;")
    t = t.replace("exact_stores = em0 + 256 <= eM", "exact_stores = false && em0 + 256 <= eM")
    return t.replace("exact_stores48 = !AT", "exact_stores48 = false && !AT")


def _patch_tilestamp(t):
    """s_memtime stamps around the parts of one tile of gemm256_kernel's four-phase schedule (read the SHARES): the wait for K-tile 0,
    K-tiles 0..3 one by one, the middle K-tiles, the last two, the closing barrier, the next tile's prefetch issue, the epilogue.
    Read with molly_exp_read_tstamps (tools/gemm_diag/run_tilestamp.py)."""
    def once(a, b):
        nonlocal t
        assert t.count(a) == 1, (t.count(a), a)
        t = t.replace(a, b)
    once("template <bool AT, bool BT, bool TO = false, bool P2 = false, bool GRP = false, bool SKM = false, bool DYN = false>", """__device__ unsigned long long g_tstamp[2048 * 8 * 16];
#define TSTAMP(x) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(x) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define TLAP(i) do { TSTAMP(tq1_); ts##i##_ += tq1_ - tq0_; tq0_ = tq1_; } while (0)
template <bool AT, bool BT, bool TO = false, bool P2 = false, bool GRP = false, bool SKM = false, bool DYN = false>""")
    once("    for (;;) {\n", """    unsigned long long tq0_, tq1_, ts0_ = 0, ts1_ = 0, ts2_ = 0, ts3_ = 0, ts4_ = 0, ts5_ = 0, ts6_ = 0, ts7_ = 0, ts8_ = 0, ts9_ = 0, ts10_ = 0, ntile_ = 0;
    for (;;) {
    TSTAMP(tq0_);
""")
    once("    if (wr == 1) SEG_BARRIER();               // stagger: group 1 runs one segment behind group 0\n",
         "    if (wr == 1) SEG_BARRIER();               // stagger: group 1 runs one segment behind group 0\n    TLAP(0);\n")
    a = "        abuf = abuf == 2 ? 0 : abuf + 1;\n        bbuf ^= 1;\n    }\n    };\n    if constexpr (GRP) {               // grouped launch"
    once(a, """        if (T == 0) TLAP(1); else if (T == 1) TLAP(2); else if (T == 2) TLAP(3); else if (T == 3) TLAP(4);
        else if (T >= nk - 2) TLAP(6); else TLAP(5);
""" + a)
    once("    if (wr == 0) SEG_BARRIER();               // balance the stagger barrier (every LDS read of this tile has returned)\n",
         "    if (wr == 0) SEG_BARRIER();               // balance the stagger barrier (every LDS read of this tile has returned)\n    TLAP(7);\n")
    once("    landed0 = roll && more;\n", "    landed0 = roll && more;\n    TLAP(8);\n")
    once("    }   // epilogue variants\n", "    }   // epilogue variants\n    TLAP(9); ++ntile_;\n")
    once("    }   // persistent tile loop\n", """    }   // persistent tile loop
    if (lane == 0 && blockIdx.x < 2048) { unsigned long long* q = g_tstamp + (blockIdx.x * 8 + wave) * 16;
        q[0] += ts0_; q[1] += ts1_; q[2] += ts2_; q[3] += ts3_; q[4] += ts4_; q[5] += ts5_; q[6] += ts6_; q[7] += ts7_; q[8] += ts8_;
        q[9] += ts9_; q[10] += ts10_; q[11] += ntile_; }
""")
    return t + """
extern "C" int molly_exp_read_tstamps(void* dst, int clear) {
    if (dst) (void)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_tstamp), sizeof(unsigned long long) * 2048 * 8 * 16);
    if (clear) { static unsigned long long z[2048 * 8 * 16]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tstamp), z, sizeof(z)); }
    return 0;
}
"""


def _patch_githead(t):
    """the committed gemm.hip (git show HEAD:...) instead of the working tree's: A/B of an uncommitted change."""
    return subprocess.run(["git", "-C", B.ROOT, "show", "HEAD:molly_amd/csrc/gemm.hip"], capture_output=True, text=True, check=True).stdout


def _patch_stamp_nostore(t):
    """tilestamp + the plain fast-path epilogue regroups but never stores (timing-only)."""
    t = _patch_tilestamp(t)
    a = "                    *reinterpret_cast<u32x4*>(c + sh * 32) = u32x4{d[2 * sh][0], d[2 * sh][1], d[2 * sh + 1][0], d[2 * sh + 1][1]};"
    assert t.count(a) == 1
    return t.replace(a, "                    if (d[2 * sh][0] == 0x12345678u) " + a.strip())


def _patch_stamp_reshot(t):
    """tilestamp + timing-only: the residual-add and SwiGLU-backward epilogues read their second operand from ONE 256 x 256 patch (every
    tile the same addresses: L2 hits) — is the epilogue's cost the operand's trip from HBM?"""
    t = _patch_tilestamp(t)
    a = "const bf16_t* r0 = p.res + (size_t)(em0 + wr * 128 + fr) * p.ldres + en0 + wc * 64 + fq * 8;"
    assert t.count(a) == 1
    t = t.replace(a, "const bf16_t* r0 = p.res + (size_t)(wr * 128 + fr) * p.ldres + wc * 64 + fq * 8;")
    a = "const bf16_t* g0 = p.res + (size_t)(em0 + wr * 128 + fr) * p.ldres + col;"
    assert t.count(a) == 1
    return t.replace(a, "const bf16_t* g0 = p.res + (size_t)(wr * 128 + fr) * p.ldres + wc * 64 + fq * 8;")


def _patch_swiglu_il(t):
    """probe (round 6): the SwiGLU-forward epilogue writes the SAVED gate / up pair-interleaved ([M][ff][2] = the tile's natural column order) through the wave's
    LDS slab as whole 128-byte lines; the activation as before.  The [gate | up] consumers do not know this layout: timing of the gate|up launch only."""
    a = """            bf16_t* gp = g0 + (size_t)i * 16 * eldc;
            *reinterpret_cast<u32x4*>(gp) = u32x4{gq[0][0], gq[0][1], gq[1][0], gq[1][1]};
            *reinterpret_cast<u32x4*>(gp + ff) = u32x4{uq[0][0], uq[0][1], uq[1][0], uq[1][1]};
            *reinterpret_cast<u32x4*>(a0 + (size_t)i * 16 * p.ldres) = u32x4{aq[0][0], aq[0][1], aq[1][0], aq[1][1]};
        }"""
    assert t.count(a) == 1
    b = """            (void)g0; (void)ff;
            *reinterpret_cast<u32x4*>(a0 + (size_t)i * 16 * p.ldres) = u32x4{aq[0][0], aq[0][1], aq[1][0], aq[1][1]};
            if (i & 1) {
                bf16_t* cil = reinterpret_cast<bf16_t*>(eC) + (size_t)(em0 + wr * 128 + epi_row) * eldc + en0 + wc * 64 + epi_ch * 8;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const u32x4 d = *reinterpret_cast<const u32x4*>(epi_slab + (k * 8 + epi_row) * 128 + ((epi_ch ^ epi_row) << 4));
                    *reinterpret_cast<u32x4*>(cil + (size_t)((i >> 1) * 32 + k * 8) * eldc) = d;
                }
            }
        }"""
    t = t.replace(a, b)
    # the interleaved (g, u) pairs into the slab BEFORE the lane swaps of the regrouped form: the lane's 4 gate + 4 up columns of block jj = one 16-byte chunk
    a2 = """#pragma unroll
            for (int w = 0; w < 2; ++w) {
                regroup_rows(gq[0][w], gq[1][w]);
                regroup_rows(uq[0][w], uq[1][w]);
                regroup_rows(aq[0][w], aq[1][w]);
            }"""
    assert t.count(a2) == 1
    b2 = """#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const u32x4 il = u32x4{__builtin_amdgcn_perm(uq[jj][0], gq[jj][0], 0x05040100u), __builtin_amdgcn_perm(uq[jj][0], gq[jj][0], 0x07060302u),
                                       __builtin_amdgcn_perm(uq[jj][1], gq[jj][1], 0x05040100u), __builtin_amdgcn_perm(uq[jj][1], gq[jj][1], 0x07060302u)};
                *reinterpret_cast<u32x4*>(epi_slab + ((i & 1) * 16 + fr) * 128 + (((4 * jj + fq) ^ (fr & 7)) << 4)) = il;
            }
#pragma unroll
            for (int w = 0; w < 2; ++w) regroup_rows(aq[0][w], aq[1][w]);"""
    return t.replace(a2, b2)


PATCHES = {"swiglu_il": _patch_swiglu_il, "stamp_reshot": _patch_stamp_reshot, "stamp_nostore": _patch_stamp_nostore, "githead": _patch_githead, "nostore": _patch_nostore, "lateprefetch": _patch_lateprefetch, "tilestamp": _patch_tilestamp}


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    patch = None
    if "--patch" in extra:
        i = extra.index("--patch")
        patch = PATCHES[extra[i + 1]]
        extra = extra[:i] + extra[i + 2:]
    out = os.path.join(B.ROOT, "tools", "variants")
    objd = os.path.join(out, name)
    os.makedirs(objd, exist_ok=True)

    def comp(src):
        obj = os.path.join(objd, src.rsplit(".", 1)[0] + ".o")
        path = os.path.join(B.CSRC, src)
        if patch is not None and src == "gemm.hip":
            path = os.path.join(objd, "gemm.hip")
            with open(path, "w") as f:
                f.write(patch(open(os.path.join(B.CSRC, src)).read()))
        cmd = [B.HIPCC] + B.FLAGS + extra + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", path, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr)
        return obj
    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(comp, B._sources()))
    lib = os.path.join(out, f"libmolly_{name}.so")
    subprocess.run([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    print(lib)


if __name__ == "__main__":
    main()
