// Shared device/host helpers for the Molly MI355X hot-path library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 storage; all arithmetic is fp32
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf2_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float bf2f(bf16_t v) { return __builtin_bit_cast(float, ((uint32_t)v) << 16); }
__device__ __forceinline__ float bflo(uint32_t v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bfhi(uint32_t v) { return __builtin_bit_cast(float, v & 0xffff0000u); }
// round-to-nearest-even, NaN preserving: hipcc lowers this to v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    bf2_t v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ bf16_t f2bf(float x) { return (bf16_t)(pack_bf2(x, 0.f) & 0xffffu); }

// Streamed-once global accesses of the HBM-bound kernels carry the nontemporal hint: measured on MI355X (tools/stream_diag/
// stream_bench.hip, same box, interleaved) a float4 copy moves 6.2 -> 6.65 TB/s, the AdamW shard step 5.9 -> 6.2 TB/s and the
// SwiGLU backward 5.75 -> 6.1 TB/s with it (and each of them 8-19 % less without a grid-stride loop: one chunk per thread).
// Small operands every block re-reads (gains, rotary tables) stay on plain loads.  -DMOLLY_NT_LOAD=0 / -DMOLLY_NT_STORE=0 = A/B builds.
#ifndef MOLLY_NT_LOAD
#define MOLLY_NT_LOAD 1
#endif
#ifndef MOLLY_NT_STORE
#define MOLLY_NT_STORE 1
#endif
template <typename T> __device__ __forceinline__ T ld_stream(const void* p) {
#if MOLLY_NT_LOAD
    return __builtin_nontemporal_load(reinterpret_cast<const T*>(p));
#else
    return *reinterpret_cast<const T*>(p);
#endif
}
template <typename T> __device__ __forceinline__ void st_stream(void* p, T v) {
#if MOLLY_NT_STORE
    __builtin_nontemporal_store(v, reinterpret_cast<T*>(p));
#else
    *reinterpret_cast<T*>(p) = v;
#endif
}

// SwiGLU backward on one packed bf16 pair (HF:models/qwen3/modeling_qwen3.py:82, act = silu(gate) * up):
// d(gate) = d * up * sigmoid(g) * (1 + g * (1 - sigmoid(g))), d(up) = d * g * sigmoid(g).  Shared by molly_swiglu_bwd and the
// MOLLY_GEMM_SWIGLU_BWD epilogue so that the two paths agree bit for bit.
// The logistic function of every SwiGLU path (forward and backward, fused epilogues and stand-alone kernels share it, so they agree
// bit for bit): 1 / (1 + e^-x) with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division — the division is ten
// vector instructions, and the fused epilogues (gate|up forward, down-projection dgrad) are bound by their vector arithmetic: 14 us
// of a 73 us tile in the dgrad (round 4).  The result is rounded to bf16 right after; against the fp32 reference nothing moves.
#ifndef MOLLY_DIAG_CHEAP_SIGMOID
#define MOLLY_DIAG_CHEAP_SIGMOID 0      // 1 (timing-only variant build): no transcendentals in the logistic — what do the fused SwiGLU epilogues pay for them?
#endif
__device__ __forceinline__ float sigmoid_fast(float x) { return MOLLY_DIAG_CHEAP_SIGMOID ? 0.5f + 0.25f * x : __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
// returns {d(gate) pair, d(up) pair}
__device__ __forceinline__ u32x2 swiglu_bwd_pair(uint32_t g, uint32_t u, uint32_t d) {
    const float ga = bflo(g), gb = bfhi(g);
    const float siga = sigmoid_fast(ga), sigb = sigmoid_fast(gb);
    const float da = bflo(d), db = bfhi(d);
    return u32x2{pack_bf2(da * bflo(u) * siga * (1.f + ga * (1.f - siga)), db * bfhi(u) * sigb * (1.f + gb * (1.f - sigb))),
                 pack_bf2(da * ga * siga, db * gb * sigb)};
}

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below a bf16 step): one v_rcp, one v_exp, a 5-term Horner —
// a third of libm erff's instruction count; the GELU epilogue of the ESM FFN GEMM is VALU-bound on it.
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    float poly = 1.061405429f;
    poly = poly * t - 1.453152027f;
    poly = poly * t + 1.421413741f;
    poly = poly * t - 0.284496736f;
    poly = poly * t + 0.254829592f;
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    const float r = 1.0f - poly * t * e;
    return copysignf(r, x);
}

// sum over an aligned group of LPR (8 | 16) lanes, result in every one of them: DPP adds (quad_perm, quad_perm, row_half_mirror,
// row_mirror) — one VALU instruction each, no LDS permute (first used by the decode attention, decode.hip)
template <int LPR>
__device__ __forceinline__ float dpp_row_sum(float d) {
    d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0xB1, 0xf, 0xf, true));       // quad_perm [1,0,3,2]
    d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x4E, 0xf, 0xf, true));       // quad_perm [2,3,0,1]
    d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x141, 0xf, 0xf, true));      // row_half_mirror
    if (LPR == 16) d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x140, 0xf, 0xf, true));   // row_mirror
    return d;
}

// One head's q/k-norm and rotary on the 4 + 4 elements a thread owns (i .. i+3 and their rotary partners i+half ..): the arithmetic that
// norm_rope_fwd_kernel, rows_tail_qkv_kernel and the decode attention's prologue (decode.hip) share, written once and with the contraction
// pinned — the KV-cache rows the three append must agree bit for bit, and under -ffp-contract=fast hipcc fused x1 c - x2 s into an fma for
// some elements and left mul + sub for others, differently per kernel (one cache element in 10^5 came out an ulp apart).
__device__ __forceinline__ float head_sumsq8(const float x1[4], const float x2[4]) {
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float t = __builtin_fmaf(x2[e], x2[e], x1[e] * x1[e]);
        ss = e ? ss + t : t;
    }
    return ss;
}
// sum over the tph lanes (a power of two <= 64, aligned) that hold one head
__device__ __forceinline__ float head_lanes_sum(float ss, int tph) {
    if (tph == 16) return dpp_row_sum<16>(ss);
    if (tph == 8) return dpp_row_sum<8>(ss);
    for (int o = tph >> 1; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    return ss;
}
__device__ __forceinline__ void head_norm8(float x1[4], float x2[4], float rstd, const bf16_t* w, int i, int half) {
    const u32x2 wa = *reinterpret_cast<const u32x2*>(w + i), wb = *reinterpret_cast<const u32x2*>(w + i + half);
    const float w1[4] = {bflo(wa[0]), bfhi(wa[0]), bflo(wa[1]), bfhi(wa[1])};
    const float w2[4] = {bflo(wb[0]), bfhi(wb[0]), bflo(wb[1]), bfhi(wb[1])};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        x1[e] = bf2f(f2bf(bf2f(f2bf(x1[e] * rstd)) * w1[e]));      // HF: bf16(x * rstd) then * gain (bf16)
        x2[e] = bf2f(f2bf(bf2f(f2bf(x2[e] * rstd)) * w2[e]));
    }
}
__device__ __forceinline__ void head_rope8(const float x1[4], const float x2[4], const f32x4 c, const f32x4 sn, float y1[4], float y2[4]) {
    // (explicit fmas: under -ffp-contract=fast the backend fuses whatever it likes, a contract(off) pragma notwithstanding)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        y1[e] = __builtin_fmaf(x1[e], c[e], -(x2[e] * sn[e]));
        y2[e] = __builtin_fmaf(x2[e], c[e], x1[e] * sn[e]);
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block reduction over <=16 waves; `red` = 16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): counter-based — the output is a pure
// function of (counter, key), so nothing has to be stored to regenerate it.
// ROUNDS: 10 is the generator's standard form; the dropout kernels take it as a build knob (MOLLY_DROPOUT_PHILOX_ROUNDS: 7 — the fewest
// rounds the paper lists as passing BigCrush — measured no faster there in round 5, the fused LoRA kernels are HBM-bound).
template <int ROUNDS = 10>
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// ---- host side -------------------------------------------------------------------------------
void molly_set_error(const char* fmt, ...);
#define MOLLY_CHECK(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            molly_set_error(__VA_ARGS__); \
            return 1;                     \
        }                                 \
    } while (0)
// Entry of every launching function: forget an error some OTHER library left in this thread's HIP error slot (found in round 2:
// after torch.distributed's device probing hipcub's radix sort — and our own launch check — reported "no ROCm-capable device is
// detected" for a launch that had worked), so that what MOLLY_LAUNCH_CHECK reports is ours.
#define MOLLY_ENTER() (void)hipGetLastError()
#define MOLLY_LAUNCH_CHECK()                                                   \
    do {                                                                       \
        hipError_t e_ = hipGetLastError();                                     \
        if (e_ != hipSuccess) {                                                \
            molly_set_error("%s:%d HIP launch error: %s", __FILE__, __LINE__, \
                            hipGetErrorString(e_));                            \
            return 2;                                                          \
        }                                                                      \
    } while (0)

// ---- MOLLY_HOST_DRY: the host-sanitizer build (python -m molly_amd.build --host-asan; tools/host_asan_driver.cpp).
// The HOST half of every translation unit — argument checks, launch_cfg's cost model, stream-K range arithmetic, grid / LDS sizing —
// compiled with -fsanitize=address,undefined and run on the CPU box; no launch and no HIP call reaches a runtime or a device.  A
// launch is RECORDED instead (kernel text, grid, block, dynamic LDS) and checked against the limits of gfx950 (a zero or oversized
// grid dimension, more than 1,024 threads, more than 160 KiB of LDS fail the call the way a launch error would).
#if defined(MOLLY_HOST_DRY)
extern "C" int molly_dry_record(const char* kernel, unsigned gx, unsigned gy, unsigned gz, unsigned bx, unsigned by, unsigned bz,
                                unsigned long lds);
extern "C" int molly_dry_failed(void);
static inline int molly_dry_launch_(const char* k, dim3 g, dim3 b, size_t lds) {
    return molly_dry_record(k, g.x, g.y, g.z, b.x, b.y, b.z, (unsigned long)lds);
}
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) (void)molly_dry_launch_(#kernel, dim3(grid), dim3(block), (size_t)(lds))
#define hipFuncSetAttribute(...) hipSuccess
#define hipMemset(p, v, n) hipSuccess
#define hipMemcpy(d, s, n, k) hipSuccess
#define hipDeviceSynchronize() hipSuccess
static inline hipError_t molly_dry_symbol_(void** p) { *p = (void*)(uintptr_t)0x7e0000000000ULL; return hipSuccess; }
#define hipGetSymbolAddress(pp, sym) molly_dry_symbol_((void**)(pp))
#undef MOLLY_ENTER
#undef MOLLY_LAUNCH_CHECK
#define MOLLY_ENTER() (void)0
#define MOLLY_LAUNCH_CHECK()                                                                  \
    do {                                                                                      \
        if (molly_dry_failed()) return 2;       /* molly_dry_record set the error text */      \
    } while (0)
#endif

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
