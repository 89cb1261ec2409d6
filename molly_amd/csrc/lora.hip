// Elementwise pieces of the LoRA branch  y = W x + (alpha/r) * B (A dropout(x))
// (PEFT lora.Linear.forward as configured by the reference: src/utils/tools.py:379-389 — r = --lora_r, alpha 64, dropout 0.05).
// The rank-r contractions themselves run on the library's bf16 MFMA GEMM (gemm.hip); this file has the HBM-bound parts.
#include "common.h"
#include "molly_hip.h"

namespace {

// Philox4x32-10: common.h (element i's keep bit is a pure function of (seed, i): the backward regenerates the mask).

// out[i] (+)= keep_i ? x[i] / (1 - p) : 0, 8 elements (one Philox block = eight 16-bit uniforms) per thread.
// P(keep) = 1 - thr / 65536 with thr = round(p * 65536): p = 0.05 -> 0.0500031.
template <bool ACC>
__global__ __launch_bounds__(256) void dropout_kernel(const bf16_t* x, bf16_t* out, long nch,   // may alias (in place)
                                                      uint32_t thr, float inv_keep, uint32_t seed_lo, uint32_t seed_hi) {
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < nch; t += (long)gridDim.x * 256) {
        uint32_t rnd[4];
        philox4x32_10((uint32_t)t, (uint32_t)(t >> 32), 0u, 0u, seed_lo, seed_hi, rnd);
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + t * 8);
        u32x4 o;
        if (ACC) o = *reinterpret_cast<const u32x4*>(out + t * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // the dropped value is rounded to bf16 first (nn.Dropout returns bf16), then added
            float a = (rnd[e] & 0xffffu) >= thr ? bflo(v[e]) * inv_keep : 0.f;
            float b = (rnd[e] >> 16) >= thr ? bfhi(v[e]) * inv_keep : 0.f;
            if (ACC) {
                const uint32_t r = pack_bf2(a, b);
                a = bflo(r) + bflo(o[e]);
                b = bfhi(r) + bfhi(o[e]);
            }
            o[e] = pack_bf2(a, b);
        }
        *reinterpret_cast<u32x4*>(out + t * 8) = o;
    }
}

__global__ __launch_bounds__(256) void scale_kernel(bf16_t* __restrict__ x, long nch, float s) {
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < nch; t += (long)gridDim.x * 256) {
        u32x4 v = *reinterpret_cast<u32x4*>(x + t * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = pack_bf2(bflo(v[e]) * s, bfhi(v[e]) * s);
        *reinterpret_cast<u32x4*>(x + t * 8) = v;
    }
}

inline int grid_for(long items) { return (int)((items + 255) / 256 < 4096 ? (items + 255) / 256 : 4096); }

}  // namespace

extern "C" int molly_dropout_bf16(void* stream, const void* x, void* out, long n, float p, uint64_t seed, int accumulate) {
    MOLLY_ENTER();
    MOLLY_CHECK(n > 0 && n % 8 == 0, "dropout: n=%ld must be a positive multiple of 8", n);
    MOLLY_CHECK(p >= 0.f && p < 1.f, "dropout: p=%f not in [0,1)", (double)p);
    const uint32_t thr = (uint32_t)(p * 65536.f + 0.5f);
    if (accumulate)
        hipLaunchKernelGGL(dropout_kernel<true>, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                           (bf16_t*)out, n / 8, thr, 1.f / (1.f - p), (uint32_t)seed, (uint32_t)(seed >> 32));
    else
        hipLaunchKernelGGL(dropout_kernel<false>, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                           (bf16_t*)out, n / 8, thr, 1.f / (1.f - p), (uint32_t)seed, (uint32_t)(seed >> 32));
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_scale_bf16(void* stream, void* x, long n, float s) {
    MOLLY_ENTER();
    MOLLY_CHECK(n > 0 && n % 8 == 0, "scale: n=%ld must be a positive multiple of 8", n);
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, n / 8, s);
    MOLLY_LAUNCH_CHECK();
    return 0;
}
