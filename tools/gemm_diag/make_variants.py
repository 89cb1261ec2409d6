#!/usr/bin/env python3
"""Diagnostic builds of the 256x256 GEMM, generated from the product source (molly_amd/csrc/gemm.hip) by text patches of
gemm256_kernel's MAIN LOOP and compiled to tools/gemm_diag/libgemm_<variant>.so (never shipped, never loaded by the package):

  timing-only ablations (outputs wrong, only the time matters; guide §7 'Ablate'):
    nodma / nodmaA / nodmaB   no LDS-DMA staging in the loop (all / A half-tiles / B half-tiles)
    noread                    no ds_read of the operand fragments      nomma   no MFMAs (fragments kept alive)
    nobar                     no barriers                              noprio  no s_setprio flips
    samet                     every block stages tile (0,0): all staging traffic hits in L2
    bal                       two LDS-DMA pieces per wave in every load segment     rb   B(n0) of the next K-tile read in P3
    mfma32                    the same flops through v_mfma_f32_32x32x16_bf16
    nostore                   the plain epilogue converts its accumulators but never stores them
  valid builds:  base, swapab (B slots first in LDS), novm (timing-only: no counted vmcnt)
  stamped builds (s_memtime; read the SHARES, not the run time):
    seg      every barrier stamped on arrival and exit: per wave group and per K-tile, cycles of each load / compute segment
             and the wait at each barrier (run_seg.py);  suffix _seg combines with the variants above (e.g. bal_seg)
    stamp    LDS-DMA issue time and the counted-vmcnt wait
Usage: python tools/gemm_diag/make_variants.py base nodma seg ... ; python tools/gemm_diag/run_variants.py base nodma ... ;
SCHEDS=0,1 python tools/gemm_diag/run_seg.py seg"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(ROOT, "molly_amd/csrc/gemm.hip")).read()
if "    if constexpr (P2) {\n        // TWO phases per K-tile" not in src:
    sys.exit("make_variants.py patches the round-1 text of gemm256_kernel's main loop (its findings: DESIGN.md 7, 'Where the GEMM's "
             "cycles go').  The round-2 kernel (rolling prefetch, staging stream) no longer matches these patches: use "
             "tools/build_variant.py (--patch nostore | tilestamp | githead, or -D flags) with tools/gemm_diag/run_kscan.py, "
             "run_tilestamp.py and cmp_libs.py instead; `git checkout 259a1c5 -- molly_amd/csrc/gemm.hip` reproduces the round-1 builds.")
a = src.index("    if constexpr (P2) {\n        // TWO phases per K-tile")
b = src.index("    if (wr == 0) SEG_BARRIER();               // balance the stagger barrier")
loop = src[a:b]

def variant(name):
    l = loop
    if "nodmaA" in name:
        l = l.replace("issue(T + 2, 0);", "").replace("issue(T + 2, 1);", "")
    elif "nodmaB" in name:
        l = l.replace("issue(T + 2, 2);", "").replace("issue(T + 2, 3);", "")
    elif "nodma" in name:
        l = l.replace("issue(T + 2, 0);", "").replace("issue(T + 2, 1);", "").replace("issue(T + 2, 2);", "").replace("issue(T + 2, 3);", "")
    if "noread" in name:
        for s in ("readB(bbuf, 0, b0);", "readB(bbuf, 1, b1);", "readA(abuf, 0, af);", "readA(abuf, 1, af);"):
            l = l.replace(s, "")
    if "nomma" in name:
        import re
        l = re.sub(r"MMA_QUAD\((\d), (\d), (\w+), (\w+)\);", r"KEEP_FRAGS(\3, \4);", l)
    if "bal" in name:
        # timing-only: 2 LDS-DMA pieces per wave in EVERY load segment of the 4-phase schedule (B1 moves from P3 to P0; WAR unsafe)
        l = l.replace("""        readB(bbuf, 0, b0);
        readA(abuf, 0, af);
        SEG_BARRIER();""", """        readB(bbuf, 0, b0);
        readA(abuf, 0, af);
        if (T + 2 < nk) issue(T + 2, 3);
        SEG_BARRIER();""")
        l = l.replace("issue(T + 2, 2); issue(T + 2, 3);", "issue(T + 2, 2);")
    if "rb" in name.split("_"):
        # timing-only: B(n0) of the NEXT K-tile is read in P3 (no reads there today) instead of P0 (12 reads)
        l = l.replace("""        readB(bbuf, 0, b0);
        readA(abuf, 0, af);
        SEG_BARRIER();""", """        readA(abuf, 0, af);
        SEG_BARRIER();""")
        l = l.replace("""            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        SEG_BARRIER();
        MMA_QUAD(1, 0, af, b0);""", """            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        readB(bbuf ^ 1, 0, b0);
        SEG_BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        MMA_QUAD(1, 0, af, b1);""")
        l = l.replace("MMA_QUAD(1, 1, af, b1);", "MMA_QUAD(1, 1, af, b0);")
    if "novm" in name:
        l = l.replace('asm volatile("s_waitcnt vmcnt(8)" ::: "memory");', "")
    if "nobar" in name:
        l = l.replace("SEG_BARRIER();", "__builtin_amdgcn_sched_barrier(0);")
    pre = ""
    global_pre = ""
    if "stamp" in name:
        global_pre = """
__device__ unsigned long long g_stamp[256 * 8 * 4];
#define STAMP(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
"""
        l = l.replace("""            issue(T + 2, 2); issue(T + 2, 3);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");""", """            STAMP(s0); issue(T + 2, 2); issue(T + 2, 3); STAMP(s1);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); STAMP(s2);
            acc_issue += s1 - s0; acc_wait += s2 - s1;""")
        l = l.replace("""                issue(T + 2, 0); issue(T + 2, 1); issue(T + 2, 2); issue(T + 2, 3);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");""", """                STAMP(s0); issue(T + 2, 0); issue(T + 2, 1); issue(T + 2, 2); issue(T + 2, 3); STAMP(s1);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); STAMP(s2);
                acc_issue += s1 - s0; acc_wait += s2 - s1;""")
        l = l.replace("if (T + 2 < nk) issue(T + 2, 0);", "if (T + 2 < nk) { STAMP(s0); issue(T + 2, 0); STAMP(s1); acc_issue += s1 - s0; }")
        l = l.replace("if (T + 2 < nk) issue(T + 2, 1);", "if (T + 2 < nk) { STAMP(s0); issue(T + 2, 1); STAMP(s1); acc_issue += s1 - s0; }")
        pre = "    unsigned long long s0, s1, s2, acc_issue = 0, acc_wait = 0, t_begin, t_end; STAMP(t_begin);\n"
    if "noread" in name:
        pre = "    readB(0, 0, b0); readB(0, 1, b1); readA(0, 0, af);\n"
    keep = """
#define KEEP_FRAGS(AF, BF) do { \\
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) { \\
        _Pragma("unroll") for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(AF[kk][i])); \\
        _Pragma("unroll") for (int j = 0; j < 2; ++j) asm volatile("" :: "v"(BF[kk][j])); } } while (0)
"""
    post = ""
    if "seg" in name:
        global_pre = """
__device__ unsigned long long g_stamp[256 * 8 * 20];
#define SEGB(i) do { unsigned long long ta_, tb_; __builtin_amdgcn_sched_barrier(0); \\
    asm volatile("s_memtime %0\\n\\ts_barrier\\n\\ts_memtime %1\\n\\ts_waitcnt lgkmcnt(0)" : "=&s"(ta_), "=&s"(tb_) :: "memory"); \\
    __builtin_amdgcn_sched_barrier(0); busy_[i] += ta_ - tprev_; bw_[i] += tb_ - ta_; tprev_ = tb_; } while (0)
"""
        # number the barriers inside each schedule's loop body
        c = l.index("    } else {\n    for (int T = 0; T < nk; ++T) {")
        parts = [l[:c], l[c:]]
        for pi in range(2):
            i = 0
            while "SEG_BARRIER();" in parts[pi]:
                parts[pi] = parts[pi].replace("SEG_BARRIER();", f"SEGB({i});", 1)
                i += 1
        l = parts[0] + parts[1]
        pre = """    unsigned long long busy_[8] = {0,0,0,0,0,0,0,0}, bw_[8] = {0,0,0,0,0,0,0,0}, tprev_;
    asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(tprev_) :: "memory");
"""
        post = """    if (lane == 0) { unsigned long long* q = g_stamp + (blockIdx.x * 8 + wave) * 20;
        for (int i = 0; i < 8; ++i) { q[i] += busy_[i]; q[8 + i] += bw_[i]; } q[16] += (unsigned long long)nk; }
"""
    if "stamp" in name:
        post = """    STAMP(t_end);
    if (lane == 0) { unsigned long long* q = g_stamp + (blockIdx.x * 8 + wave) * 4; q[0] += t_end - t_begin; q[1] += acc_issue; q[2] += acc_wait; q[3] += (unsigned long long)nk; }
"""
    tail = src[b:]
    if "seg" in name:
        tail = tail.replace("exact_stores = em0 + 256 <= p.M", "exact_stores = false && em0 + 256 <= p.M")
        tail += """
extern "C" int molly_exp_read_stamps(void* dst, int clear) {
    if (dst) (void)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 256 * 8 * 20);
    if (clear) { static unsigned long long z[256 * 8 * 20]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z)); }
    return 0;
}
"""
    if "stamp" in name:
        tail = tail.replace("exact_stores = em0 + 256 <= p.M", "exact_stores = false && em0 + 256 <= p.M")
        tail += """
extern "C" int molly_exp_read_stamps(void* dst, int clear) {
    if (dst) (void)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * 256 * 8 * 4);
    if (clear) { static unsigned long long z[256 * 8 * 4]; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z)); }
    return 0;
}
"""
    if "nostore" in name or "lateprefetch" in name:
        # timing-only: the plain epilogue converts but never stores (nostore) — what the output stores cost a tile, including
        # their place in the in-order vmcnt queue in front of the next tile's K-tile 2;  lateprefetch (valid): the next tile's
        # K-tiles 0 and 1 are issued AFTER the epilogue's stores instead of before them
        k = tail.index("// ---- epilogue: lane owns C[m")
        if "nostore" in name:
            tail = tail.replace("exact_stores = em0 + 256 <= eM", "exact_stores = false && em0 + 256 <= eM")
            st = "                *reinterpret_cast<u32x2*>(c) = u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};"
            k2 = tail.index(st, k)
            tail = tail[:k2] + "                if (v[0] == 1.2345e30f) " + st.strip() + tail[k2 + len(st):]
    head = src[:a]
    if "samet" in name:
        # timing-only: every block stages tile (0, 0): all LDS-DMA traffic hits in L2
        head = head.replace("p.lda, m0 + which * 128, p.M", "p.lda, which * 128, p.M").replace("p.ldb, n0 + (which - 2) * 128, p.N", "p.ldb, (which - 2) * 128, p.N")
    if "mfma32" in name:
        # timing-only: same flops through v_mfma_f32_32x32x16_bf16 (8 per quadrant instead of 16 16x16x32): does the longer
        # MFMA free issue slots for the partner wave's load segment?
        head = head.replace("f32x4 acc[8][4];", "typedef float f32x16_ __attribute__((ext_vector_type(16))); f32x16_ acc32[8];")
        kk_ = head.index("void gemm256_kernel")
        head = head[:kk_] + head[kk_:].replace("for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};",
                            "for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;")
        k0 = head.index("#define MMA_QUAD(MH, NH, AF, BF)")
        k1 = head.index("#define SEG_BARRIER()")
        head = head[:k0] + """#define MMA_QUAD(MH, NH, AF, BF)                                                                            \\
    do {                                                                                                    \\
        __builtin_amdgcn_s_setprio(1);                                                                      \\
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                    \\
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                       \\
            acc32[((MH) * 2 + (NH)) * 2 + (i & 1)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(BF[kk][i >> 1], AF[kk][i], acc32[((MH) * 2 + (NH)) * 2 + (i & 1)], 0, 0, 0); \\
        __builtin_amdgcn_s_setprio(0);                                                                      \\
    } while (0)
""" + head[k1:]
    if "swapab" in name:
        # valid build: B slots first (0..64 KiB), A slots behind (64..160 KiB)
        head = head.replace("which < 2 ? smem + ((ktl % 3) * 2 + which) * HT : smem + (6 + (ktl & 1) * 2 + (which - 2)) * HT",
                            "which < 2 ? smem + (4 + (ktl % 3) * 2 + which) * HT : smem + ((ktl & 1) * 2 + (which - 2)) * HT")
        head = head.replace("const bf16_t* t = smem + (abuf * 2 + wr) * HT;", "const bf16_t* t = smem + (4 + abuf * 2 + wr) * HT;")
        head = head.replace("const bf16_t* t = smem + (6 + bbuf * 2 + (wc >> 1)) * HT;", "const bf16_t* t = smem + (bbuf * 2 + (wc >> 1)) * HT;")
    if "noprio" in name:
        head = head.replace("__builtin_amdgcn_s_setprio(1);", "").replace("__builtin_amdgcn_s_setprio(0);", "")
    if global_pre:
        k = head.index("template <bool AT, bool BT, bool TO = false, bool P2 = false>")
        head = head[:k] + global_pre + head[k:]
    if "mfma32" in name:
        tail = tail.replace("{acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]}", "{acc32[i][j*4+0], acc32[i][j*4+1], acc32[i][j*4+2], acc32[i][j*4+3]}")
        tail = tail.replace("= acc[i][j];", "= f32x4{acc32[i][j*4+0], acc32[i][j*4+1], acc32[i][j*4+2], acc32[i][j*4+3]};")
    s = head + keep + pre + l + post + tail
    # the frags must be declared before `pre`
    return s

names = sys.argv[1:] or ["base", "nodma", "noread", "nomma", "nodma_noread", "nobar"]
os.makedirs(os.path.join(ROOT, "tools/gemm_diag/build"), exist_ok=True)
for n in names:
    p = os.path.join(ROOT, f"tools/gemm_diag/build/gemm_{n}.hip")
    open(p, "w").write(variant(n))
    out = os.path.join(ROOT, f"tools/gemm_diag/libgemm_{n}.so")
    inc = ["-I" + os.path.join(ROOT, "molly_amd/csrc"), "-I" + os.path.join(ROOT, "include")]
    obj = p[:-4] + ".o"
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=fast"] + inc +
                       ["-x", "hip", "-c", p, "-o", obj], capture_output=True, text=True)
    if r.returncode == 0:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, obj,
                            os.path.join(ROOT, "molly_amd/csrc/build/capi.o")], capture_output=True, text=True)
    print(n, "rc", r.returncode, r.stderr[-2000:] if r.returncode else "")
