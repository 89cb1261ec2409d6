// EXPERIMENT (round 6, not part of the library): NT bf16 GEMM in the shape of the vendor's kernel for these problems
// (Custom_Cijk_..._MT256x256x64_MI16x16x1, 256 threads, 256 + 256 registers, 130 KB LDS: profiles/r06_logs/vendor_kernel_names.txt) —
// ONE wave per SIMD, 4 waves (2 x 2) x 128x128 outputs (256 accumulator registers in the AGPR file), BK = 64, TWO 64 KB LDS stages, operands
// staged THROUGH REGISTERS (global_load_dwordx4 -> ds_write_b128; rounds 1-2 built this form with LDS-DMA, whose issue cost a lone wave
// cannot hide: LOG.md 'Where the GEMM's cycles go'), one barrier per K-tile, fragments double-buffered by K half (kk).  0.25 LDS reads per
// MFMA against 0.375 in gemm256_kernel.  Un-swapped MFMA operands + B rows interleaved in the LDS image (row 16 j + f of a wave's 128 holds
// weight row 8 f + j), so a lane ends with 8 consecutive output columns: 16-byte stores, 16 lanes = one 256-byte run of a row.
// M, N multiples of 256, K a multiple of 64, K >= 128.   Build + run: tools/r06/gemm4w/run.py
#include "common.h"

namespace {
constexpr int BK = 64;
constexpr int OPB = 256 * BK * 2;        // bytes of one operand tile (32 KiB)
constexpr int STAGEB = 2 * OPB;          // A tile | B tile (64 KiB)

struct Args {
    const bf16_t* A; const bf16_t* B; bf16_t* C;
    int M, N, K, lda, ldb, ldc, tiles_m, tiles_n;
};

#define MFMA_T(C, A_, B_) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(C) : "v"(A_), "v"(B_) : "memory")

__global__ __launch_bounds__(256, 1) void gemm4w_reg_kernel(Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;

    // XCD-aware tile walk (blocks b, b+8, ... share an XCD), GROUP_M = 4
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int swzid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int per_group = 4 * p.tiles_n;
    const int first_m = (swzid / per_group) * 4;
    const int gsz = min(p.tiles_m - first_m, 4);
    const int m0 = (first_m + (swzid % per_group) % gsz) * 256, n0 = ((swzid % per_group) / gsz) * 256;

    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- staging: wave w moves rows 64 w .. 64 w + 63 of the A tile and of the B tile; instruction i (0..7) = rows 8 i + (lane >> 3), 16-byte chunk lane & 7
    const int srow = lane >> 3, sch = lane & 7;
    const char* gA = reinterpret_cast<const char*>(p.A + (size_t)(m0 + wave * 64 + srow) * p.lda) + sch * 16;
    const char* gB = reinterpret_cast<const char*>(p.B + (size_t)(n0 + wave * 64 + srow) * p.ldb) + sch * 16;
    const size_t strideA = (size_t)8 * p.lda * 2, strideB = (size_t)8 * p.ldb * 2;
    // LDS byte offsets of the lane's pieces: A row as it is; B row g = 64 w + 8 i + srow of the tile -> LDS row (g & 128) + 16 (g & 7) + ((g & 127) >> 3)
    unsigned wa[8], wb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int ra = wave * 64 + i * 8 + srow;
        wa[i] = (unsigned)(ra * 128 + ((sch ^ (ra & 7)) << 4));
        const int g = wave * 64 + i * 8 + srow;
        const int rb = (g & 128) + 16 * (g & 7) + ((g & 127) >> 3);
        wb[i] = (unsigned)(OPB + rb * 128 + ((sch ^ (rb & 7)) << 4));
    }
    u32x4 sa[8], sb[8];
    const int nk = p.K / BK;
    auto gload = [&](int t, int i, bool isb) {
        const int tt = t < nk ? t : nk - 1;
        if (isb) sb[i] = *reinterpret_cast<const u32x4*>(gB + i * strideB + (size_t)tt * BK * 2);
        else sa[i] = *reinterpret_cast<const u32x4*>(gA + i * strideA + (size_t)tt * BK * 2);
    };
    auto swrite = [&](int stage, int i, bool isb) {
        char* s = smem + stage * STAGEB;
        if (isb) *reinterpret_cast<u32x4*>(s + wb[i]) = sb[i];
        else *reinterpret_cast<u32x4*>(s + wa[i]) = sa[i];
    };
    // ---- fragments: A rows wr * 128 + 16 i + fr, B LDS rows wc * 128 + 16 j + fr; chunk (4 kk + fq) ^ (row & 7), row & 7 == fr & 7
    const unsigned fa = (unsigned)((wr * 128 + fr) * 128), fb = (unsigned)(OPB + (wc * 128 + fr) * 128);
    const unsigned ck0 = (unsigned)((fq ^ (fr & 7)) << 4), ck1 = (unsigned)(((4 + fq) ^ (fr & 7)) << 4);
    bf16x8 a0[8], b0[8], a1[8], b1[8];
    auto fread = [&](int stage, int kk, int idx, bf16x8 (&af)[8], bf16x8 (&bf)[8]) {      // idx 0..15: A0 B0 A1 B1 ...
        const char* s = smem + stage * STAGEB + (kk ? ck1 : ck0);
        if (idx & 1) bf[idx >> 1] = *reinterpret_cast<const bf16x8*>(s + fb + (idx >> 1) * 2048);
        else af[idx >> 1] = *reinterpret_cast<const bf16x8*>(s + fa + (idx >> 1) * 2048);
    };

    // ---- prologue: tile 0 into stage 0, tile 1 into the registers, kk = 0 fragments of tile 0
#pragma unroll
    for (int i = 0; i < 8; ++i) { gload(0, i, false); gload(0, i, true); }
#pragma unroll
    for (int i = 0; i < 8; ++i) { swrite(0, i, false); swrite(0, i, true); }
#pragma unroll
    for (int i = 0; i < 8; ++i) { gload(1, i, false); gload(1, i, true); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int x = 0; x < 16; ++x) fread(0, 0, x, a0, b0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    for (int t = 0; t < nk; ++t) {
        const int st = t & 1;
        // ---- first half: kk = 0 products of tile t; beside them the kk = 1 fragments of tile t (even slots), the registers' tile t + 1 into the
        // other stage (its last reads returned before the previous mid barrier) and the loads of tile t + 2 into the registers (odd slots)
#pragma unroll
        for (int n = 0; n < 64; ++n) {
            MFMA_T(acc[n >> 3][n & 7], a0[n >> 3], b0[n & 7]);
#ifndef ABL_NOREAD
            if ((n & 3) == 0) fread(st, 1, n >> 2, a1, b1);
#endif
#ifndef ABL_NOSTAGE
            if ((n & 3) == 2 && n < 32) { const int k = n >> 2; swrite(st ^ 1, k, false); swrite(st ^ 1, k, true); }      // n = 2, 6, .., 30: 8 x (A, B)
            if ((n & 3) == 2 && n >= 32) { const int k = (n - 32) >> 2; gload(t + 2, k, false); gload(t + 2, k, true); }  // n = 34, .., 62
#endif
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef ABL_NOBAR
        __builtin_amdgcn_s_barrier();
#endif
        // ---- second half: kk = 1 products; beside them the kk = 0 fragments of tile t + 1
#pragma unroll
        for (int n = 0; n < 64; ++n) {
            MFMA_T(acc[n >> 3][n & 7], a1[n >> 3], b1[n & 7]);
#ifndef ABL_NOREAD
            if ((n & 3) == 0) fread(st ^ 1, 0, n >> 2, a0, b0);
#endif
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // nothing still landing; MFMA results readable

    // ---- epilogue: lane (fr, fq) owns rows wr * 128 + 16 i + 4 fq + e, columns wc * 128 + 8 fr + j (j = 0..7): 16 bytes per row
    bf16_t* c0 = p.C + (size_t)(m0 + wr * 128 + 4 * fq) * p.ldc + n0 + wc * 128 + 8 * fr;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const u32x4 v = u32x4{pack_bf2(acc[i][0][e], acc[i][1][e]), pack_bf2(acc[i][2][e], acc[i][3][e]),
                                  pack_bf2(acc[i][4][e], acc[i][5][e]), pack_bf2(acc[i][6][e], acc[i][7][e])};
            *reinterpret_cast<u32x4*>(c0 + (size_t)(i * 16 + e) * p.ldc) = v;
        }
}
}  // namespace

extern "C" int gemm4w_reg_nt(void* stream, const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc) {
    if (M % 256 || N % 256 || K % BK || K < 2 * BK) return 1;
    static bool set = false;
    if (!set) {
        (void)hipFuncSetAttribute((const void*)gemm4w_reg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGEB);
        set = true;
    }
    Args p{(const bf16_t*)A, (const bf16_t*)B, (bf16_t*)C, M, N, K, lda, ldb, ldc, M / 256, N / 256};
    hipLaunchKernelGGL(gemm4w_reg_kernel, dim3(p.tiles_m * p.tiles_n), dim3(256), 2 * STAGEB, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
