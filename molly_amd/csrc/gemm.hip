// bf16 MFMA GEMM for gfx950:  C[M,N] = A[M,K] · B[N,K]^T  (+bias[N]) (GELU) (+res[M,N]) (+= C)
//
// Both operands are K-contiguous ("NT").  This is the shape of every Linear forward on Molly's hot path
// (HF nn.Linear: y = x W^T; reference call sites src/model/omics_one.py:75-91,175) and — because the
// runtime keeps a transposed bf16 copy of each weight and transposes activations for wgrad — of every
// dgrad / wgrad as well (see DESIGN.md "GEMM forms").
//
// v1 structure (guide §5 "minimum 2-phase"): 128x128x64 tile, 4 waves (2x2, 64x64 each),
// mfma_f32_16x16x32_bf16, A/B tiles staged HBM->LDS by global_load_lds (16 B/lane, 1 KiB per wave
// instruction), double-buffered LDS (64 KiB -> 2 blocks/CU), XOR-swizzled 16-B chunks (conflict-free
// ds_read_b128), XCD-aware tile order.  MFMA operands are passed swapped (B-tile fragment as the A
// operand) so every lane owns 4 CONSECUTIVE n of one row m -> 8-byte bf16 stores / 16-byte fp32 stores.
#include "common.h"
#include "molly_hip.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_ELEMS = BM * BK;          // 8192 bf16 = 16 KiB per operand per stage

struct GemmArgs {
    const bf16_t* A; const bf16_t* B; void* C;
    const bf16_t* bias; const bf16_t* res;
    int M, N, K, lda, ldb, ldc, ldres;
    int flags;
    int tiles_m, tiles_n;
};

// stage one 128x64 bf16 tile: 16 wave-instructions of 1 KiB; wave w issues instructions w*4 .. w*4+3
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ g, int ld, int row0, int rows_total, int k0,
                                           bf16_t* lds_tile, int wave, int lane) {
    const int r_in = lane >> 3;                 // row inside the 8-row group
    const int c_src = (lane & 7) ^ r_in;        // swizzle on the SOURCE address (LDS dest is lane-linear)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int inst = wave * 4 + i;
        int row = row0 + inst * 8 + r_in;
        row = row < rows_total ? row : rows_total - 1;     // clamp: OOB rows re-read a valid row, masked at store
        const bf16_t* src = g + (size_t)row * ld + k0 + c_src * 8;
        __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds_tile + inst * 512), 16, 0, 0);
    }
}

__device__ __forceinline__ bf16x8 lds_frag(const bf16_t* lds_tile, int row, int chunk) {
    const int phys = chunk ^ (row & 7);
    return *reinterpret_cast<const bf16x8*>(lds_tile + row * BK + phys * 8);
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
    // layout: [stage][A|B][128*64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- XCD-aware tile mapping: blocks b and b+8 share an XCD (L2); give each XCD a contiguous
    // chunk of the tile list, and walk tiles in groups of 8 M-tiles per N sweep so neighbours share B.
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    constexpr int GROUP_M = 8;
    const int per_group = GROUP_M * p.tiles_n;
    const int grp = swz / per_group;
    const int first_m = grp * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int tm = first_m + (swz % per_group) % gsz;
    const int tn = (swz % per_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    stage_tile(p.A, p.lda, m0, p.M, 0, smem, wave, lane);
    stage_tile(p.B, p.ldb, n0, p.N, 0, smem + TILE_ELEMS, wave, lane);
    __syncthreads();   // hipcc emits s_waitcnt vmcnt(0) ahead of the barrier while LDS-DMA is in flight

    const int fr = lane & 15, fq = lane >> 4;
    int cur = 0;
    for (int t = 0; t < nk; ++t) {
        bf16_t* sA = smem + cur * 2 * TILE_ELEMS;
        bf16_t* sB = sA + TILE_ELEMS;
        if (t + 1 < nk) {
            bf16_t* nA = smem + (cur ^ 1) * 2 * TILE_ELEMS;
            stage_tile(p.A, p.lda, m0, p.M, (t + 1) * BK, nA, wave, lane);
            stage_tile(p.B, p.ldb, n0, p.N, (t + 1) * BK, nA + TILE_ELEMS, wave, lane);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = lds_frag(sA, wm * 64 + i * 16 + fr, kk * 4 + fq);
                bfr[i] = lds_frag(sB, wn * 64 + i * 16 + fr, kk * 4 + fq);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: lane owns C[m = .. + fr][n = .. + fq*4 + 0..3]
    const bool has_bias = p.flags & MOLLY_GEMM_BIAS, has_res = p.flags & MOLLY_GEMM_RESIDUAL;
    const bool gelu = p.flags & MOLLY_GEMM_GELU, accum = p.flags & MOLLY_GEMM_ACCUMULATE;
    const bool out_f32 = p.flags & MOLLY_GEMM_OUT_F32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + fq * 4;
            if (n >= p.N) continue;                       // N % 4 == 0 is checked by the host
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (has_bias) {
                const u32x2 b = *reinterpret_cast<const u32x2*>(p.bias + n);
                v[0] += bflo(b[0]); v[1] += bfhi(b[0]); v[2] += bflo(b[1]); v[3] += bfhi(b[1]);
            }
            if (gelu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
            }
            if (has_res) {
                const u32x2 b = *reinterpret_cast<const u32x2*>(p.res + (size_t)m * p.ldres + n);
                v[0] += bflo(b[0]); v[1] += bfhi(b[0]); v[2] += bflo(b[1]); v[3] += bfhi(b[1]);
            }
            if (out_f32) {
                float* c = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n;
                if (accum) {
                    const f32x4 o = *reinterpret_cast<const f32x4*>(c);
                    v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
                }
                *reinterpret_cast<f32x4*>(c) = f32x4{v[0], v[1], v[2], v[3]};
            } else {
                bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n;
                if (accum) {
                    const u32x2 o = *reinterpret_cast<const u32x2*>(c);
                    v[0] += bflo(o[0]); v[1] += bfhi(o[0]); v[2] += bflo(o[1]); v[3] += bfhi(o[1]);
                }
                *reinterpret_cast<u32x2*>(c) = u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
            }
        }
    }
}

}  // namespace

extern "C" int molly_gemm_nt_bf16(void* stream, const void* A, const void* B, void* C, const void* bias,
                                  const void* res, int M, int N, int K, int lda, int ldb, int ldc, int ldres,
                                  int flags) {
    MOLLY_CHECK(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
    MOLLY_CHECK(K % BK == 0, "gemm: K=%d must be a multiple of %d", K, BK);
    MOLLY_CHECK(N % 4 == 0, "gemm: N=%d must be a multiple of 4", N);
    MOLLY_CHECK(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, "gemm: lda/ldb must be multiples of 8, ldc of 4");
    MOLLY_CHECK(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)C % 16) == 0,
                "gemm: operands must be 16-byte aligned");
    MOLLY_CHECK(!(flags & MOLLY_GEMM_BIAS) || bias, "gemm: MOLLY_GEMM_BIAS without bias pointer");
    MOLLY_CHECK(!(flags & MOLLY_GEMM_RESIDUAL) || (res && ldres % 4 == 0), "gemm: bad residual");
    GemmArgs p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C;
    p.bias = (const bf16_t*)bias; p.res = (const bf16_t*)res;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldres = ldres; p.flags = flags;
    p.tiles_m = cdiv(M, BM); p.tiles_n = cdiv(N, BN);
    const int grid = p.tiles_m * p.tiles_n;
    const size_t lds = 2 * 2 * TILE_ELEMS * sizeof(bf16_t);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_nt_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_nt_kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, p);
    MOLLY_LAUNCH_CHECK();
    return 0;
}
