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
}  // namespace

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
