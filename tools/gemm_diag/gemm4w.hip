// EXPERIMENT (not part of the library): NT bf16 GEMM with ONE wave per SIMD — 256x256 block tile, 4 waves (2 x 2), each
// wave a 128x128 output tile (256 accumulator registers, 512-register budget), BK = 32, 5 LDS stages of 32 KiB, one
// barrier per K-step, fragments double-buffered in registers, MFMA / ds_read / LDS-DMA interleaved by
// sched_group_barrier.  Reads 2/3 of the LDS bytes per flop of the 8-wave kernel (DESIGN.md "Where the GEMM's cycles go").
// M, N multiples of 256, K a multiple of 32.   Build + run: tools/gemm_diag/run_gemm4w.py
#include "common.h"

namespace {
constexpr int BK = 32, NSTAGE = 5;
constexpr int OP = 256 * BK;             // elements of one operand tile (16 KiB)
constexpr int STAGE = 2 * OP;            // A tile | B tile

struct Args {
    const bf16_t* A; const bf16_t* B; bf16_t* C;
    int M, N, K, lda, ldb, ldc, tiles_m, tiles_n;
};

// swizzle of the [256 rows][32 k] LDS image (64-byte rows, 4 chunks of 16 B): slot (row, pos) holds chunk pos ^ f(row),
// f(row) = (-(row >> 2)) & 3 — every 16-lane group of a ds_read_b128 of 16 rows x 4 chunks then covers all 16 bank quads.
__device__ __forceinline__ int swz(int row) { return (-(row >> 2)) & 3; }

__device__ __forceinline__ void stage_op(const bf16_t* __restrict__ g, int ld, int row0, int k0, bf16_t* lds, int wave, int lane) {
    const int r_in = lane >> 2, pos = lane & 3;
    const int c_src = pos ^ swz(r_in);                       // (inst*16 + r_in) >> 2 has the low bits of r_in >> 2
    // wave-uniform 64-bit base per piece (scalar registers) + ONE 32-bit per-lane byte offset shared by all pieces
    const unsigned off = ((unsigned)r_in * (unsigned)ld + (unsigned)(c_src * 8)) * 2u;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int inst = wave * 4 + i;
        const char* base = reinterpret_cast<const char*>(g + (size_t)(row0 + inst * 16) * ld + k0);
        __builtin_amdgcn_global_load_lds(GLB_PTR(base + off), LDS_PTR(lds + inst * 512), 16, 0, 0);
    }
}

__global__ __launch_bounds__(256, 1) void gemm4w_kernel(Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;

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

    const int nk = p.K / BK;
    auto issue = [&](int t) {
        bf16_t* st = smem + (t % NSTAGE) * STAGE;
        stage_op(p.A, p.lda, m0, t * BK, st, wave, lane);
        stage_op(p.B, p.ldb, n0, t * BK, st + OP, wave, lane);
    };
    const int fr = lane & 15, fq = lane >> 4;
    const int lane_off = fr * BK + ((fq ^ swz(fr)) * 8);     // (r0 + fr) with r0 % 16 == 0: the swizzle only sees fr
    auto read = [&](int t, bf16x8 (&af)[8], bf16x8 (&bfr)[8]) {
        const bf16_t* sA = smem + (t % NSTAGE) * STAGE + wr * 128 * BK + lane_off;
        const bf16_t* sB = smem + (t % NSTAGE) * STAGE + OP + wc * 128 * BK + lane_off;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            af[i] = *reinterpret_cast<const bf16x8*>(sA + i * 16 * BK);
            bfr[i] = *reinterpret_cast<const bf16x8*>(sB + i * 16 * BK);
        }
    };
    // One K-step, hand-ordered.  The MFMAs are inline asm with the accumulator TIED in an AGPR ("+a"): hipcc's own MFMA
    // selection for a 512-register kernel writes each product to a fresh AGPR quad and copies accumulators through VGPRs
    // at the loop edge (hundreds of v_accvgpr_read/write per trip).  Every asm carries a memory clobber, so the ds_reads and
    // LDS-DMA builtins placed between them stay where the source puts them: 16 fragment reads beside the first 32 MFMAs, the
    // 8 LDS-DMA pieces beside the last 32.
#define MFMA_T(C, B_, A_) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(C) : "v"(B_), "v"(A_) : "memory")
    // Uniform pipeline: EVERY K-step issues 8 LDS-DMA pieces and 16 fragment reads (past the end of K they re-stage / re-read
    // the last K-step into slots nobody uses any more), so the loop body has no conditional paths and one counted wait.
    const int last = nk - 1;
#pragma unroll
    for (int t = 0; t < NSTAGE - 1; ++t) issue(min(t, last));
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");          // K-step 0 landed, three younger ones in flight
    __builtin_amdgcn_s_barrier();
    bf16x8 a0[8], b0[8], a1[8], b1[8];
    read(0, a0, b0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    auto step = [&](int t, const bf16x8 (&af)[8], const bf16x8 (&bfr)[8], bf16x8 (&naf)[8], bf16x8 (&nbf)[8]) {
        // K-step t+1 landed (own pieces; steps t+2, t+3 stay in flight), every wave's reads of step t-1's slot returned
#ifndef ABL_NODMA
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#endif
#ifndef ABL_NOBAR
        __builtin_amdgcn_s_barrier();
#endif
        const int tn = min(t + 1, last), tt = t + NSTAGE - 1, tl = min(tt, last);
        const bf16_t* sA = smem + (tn % NSTAGE) * STAGE + wr * 128 * BK + lane_off;
        const bf16_t* sB = smem + (tn % NSTAGE) * STAGE + OP + wc * 128 * BK + lane_off;
        bf16_t* st = smem + (tt % NSTAGE) * STAGE;
        const unsigned offa = ((unsigned)(lane >> 2) * (unsigned)p.lda + (unsigned)(((lane & 3) ^ swz(lane >> 2)) * 8)) * 2u;
        const unsigned offb = ((unsigned)(lane >> 2) * (unsigned)p.ldb + (unsigned)(((lane & 3) ^ swz(lane >> 2)) * 8)) * 2u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                MFMA_T(acc[i][j], bfr[j], af[i]);
                const int n = i * 8 + j;                       // MFMA number within the K-step
#ifndef ABL_NOREAD
                if (n < 32 && (n & 1)) {                       // after MFMAs 1, 3, ..., 31: one fragment read each
                    const int k = n >> 1;                      // 0..15: A0 B0 A1 B1 ...
                    if (k & 1) nbf[k >> 1] = *reinterpret_cast<const bf16x8*>(sB + (k >> 1) * 16 * BK);
                    else naf[k >> 1] = *reinterpret_cast<const bf16x8*>(sA + (k >> 1) * 16 * BK);
                }
#endif
#ifndef ABL_NODMA
                if (n >= 32 && (n & 3) == 3) {                 // after MFMAs 35, 39, ..., 63: one LDS-DMA piece each
                    const int k = (n - 32) >> 2;               // 0..7: A pieces 0..3, B pieces 0..3 of this wave
                    const int inst = wave * 4 + (k & 3);
                    if (k < 4) {
                        const char* base = reinterpret_cast<const char*>(p.A + (size_t)(m0 + inst * 16) * p.lda + tl * BK);
                        __builtin_amdgcn_global_load_lds(GLB_PTR(base + offa), LDS_PTR(st + inst * 512), 16, 0, 0);
                    } else {
                        const char* base = reinterpret_cast<const char*>(p.B + (size_t)(n0 + inst * 16) * p.ldb + tl * BK);
                        __builtin_amdgcn_global_load_lds(GLB_PTR(base + offb), LDS_PTR(st + OP + inst * 512), 16, 0, 0);
                    }
                }
#endif
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // nk even (K % 64 == 0 checked by the launcher): two K-steps per trip, the fragment sets swap roles statically
    for (int t = 0; t < nk; t += 2) {
        step(t, a0, b0, a1, b1);
        step(t + 1, a1, b1, a0, b0);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // nothing still landing; MFMA results readable

    // epilogue: lane owns C[m = .. + fr][n = .. + fq*4 + 0..3]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + wr * 128 + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int n = n0 + wc * 128 + j * 16 + fq * 4;
            const f32x4 v = acc[i][j];
            *reinterpret_cast<u32x2*>(p.C + (size_t)m * p.ldc + n) = u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        }
    }
}
}  // namespace

extern "C" int gemm4w_nt(void* stream, const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc) {
    if (M % 256 || N % 256 || K % (2 * BK)) return 1;
    static bool set = false;
    if (!set) {
        (void)hipFuncSetAttribute((const void*)gemm4w_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NSTAGE * STAGE * 2);
        set = true;
    }
    Args p{(const bf16_t*)A, (const bf16_t*)B, (bf16_t*)C, M, N, K, lda, ldb, ldc, M / 256, N / 256};
    hipLaunchKernelGGL(gemm4w_kernel, dim3(p.tiles_m * p.tiles_n), dim3(256), NSTAGE * STAGE * 2, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
