// Instruction-semantics probes: tiny kernels that expose the gfx950 MFMA operand maps and the
// ds_read_b64_tr_b16 gather exactly as the production kernels use them, so tests can pin them with
// exact integer data (guide §3 "A=I with asymmetric B").
#include "common.h"
#include "molly_hip.h"

namespace {
// D[16x16] = A[16x32] · B[16x32]^T with the NT operand convention of gemm.hip / attention.hip:
// lane l reads 8 consecutive k of row (l&15) at k = 8*(l>>4).
__global__ void probe_mfma16(const bf16_t* A, const bf16_t* B, float* D) {
    const int l = threadIdx.x, fr = l & 15, fq = l >> 4;
    bf16x8 a = *reinterpret_cast<const bf16x8*>(A + fr * 32 + fq * 8);
    bf16x8 b = *reinterpret_cast<const bf16x8*>(B + fr * 32 + fq * 8);
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    // documented C/D map: col = lane&15 (B-operand row index), row = (lane>>4)*4 + reg (A-operand row index)
#pragma unroll
    for (int r = 0; r < 4; ++r) D[(fq * 4 + r) * 16 + fr] = c[r];
}

// every lane of ONE wave issues ds_read_b64_tr_b16 at &tile[(l>>2)&3 ... ] the way the guide describes:
// 16-lane group g reads the 4x16 block whose rows are 4g..4g+3; lane 4q+p of the group supplies the address of
// row q, columns 4p..4p+3.  Output: the 4 u16 each lane received.
__global__ void probe_tr16(const bf16_t* tile, bf16_t* out, int stride) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    bf16_t* lds = reinterpret_cast<bf16_t*>(sm);
    const int l = threadIdx.x;
    for (int i = l; i < 16 * stride; i += 64) lds[i] = tile[i];
    __syncthreads();
    const int g = l >> 4, i = l & 15, q = i >> 2, pp = i & 3;
    const bf16_t* addr = lds + (4 * g + q) * stride + 4 * pp;
    bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)addr);
#pragma unroll
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = (bf16_t)v[e];
}
// A stand-in for a collective's kernel: `blocks` workgroups that each own a whole CU (all 160 KiB of LDS) and spin for `us`
// microseconds of the 100 MHz s_memrealtime clock.  tools/diag/gemm_beside_hog.py times the GEMM's launch shapes beside it.
__global__ __launch_bounds__(64) void probe_hog(unsigned long long ticks, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    unsigned long long t0, t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    do {
        __builtin_amdgcn_s_sleep(16);
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    } while (t - t0 < ticks);
    if (sink && threadIdx.x == 0 && t == 0) sink[0] = (unsigned)sm[0];       // keeps the LDS allocation alive
}
}  // namespace

extern "C" int molly_probe_hog(void* stream, int blocks, int us, void* sink) {
    MOLLY_ENTER();
    MOLLY_CHECK(blocks >= 1 && blocks <= 256 && us >= 1 && us <= 100000, "probe_hog: blocks=%d us=%d", blocks, us);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)probe_hog, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        attr = true;
    }
    hipLaunchKernelGGL(probe_hog, dim3(blocks), dim3(64), 163840, (hipStream_t)stream, (unsigned long long)us * 100ull, (unsigned*)sink);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_probe_mfma16(void* stream, const void* A, const void* B, float* D) {
    MOLLY_ENTER();
    hipLaunchKernelGGL(probe_mfma16, dim3(1), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)A, (const bf16_t*)B, D);
    MOLLY_LAUNCH_CHECK();
    return 0;
}
extern "C" int molly_probe_tr16(void* stream, const void* tile, void* out, int stride) {
    MOLLY_ENTER();
    MOLLY_CHECK(stride >= 16 && stride % 4 == 0 && stride <= 512, "probe_tr16: bad stride %d", stride);
    hipLaunchKernelGGL(probe_tr16, dim3(1), dim3(64), 16 * stride * 2, (hipStream_t)stream, (const bf16_t*)tile,
                       (bf16_t*)out, stride);
    MOLLY_LAUNCH_CHECK();
    return 0;
}
