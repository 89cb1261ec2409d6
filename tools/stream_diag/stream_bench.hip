// Streaming-kernel variants timed side by side on one box (diagnostic, not part of the library):
//   copy (float4) as the box's calibration, then the AdamW shard step and the SwiGLU backward in several forms
//   (grid-stride vs one chunk per thread, chunks in flight per thread, nontemporal loads / stores, block count).
// Build:  hipcc -O3 --offload-arch=gfx950 -o tools/stream_diag/stream_bench tools/stream_diag/stream_bench.hip
// Run  :  tools/stream_diag/stream_bench            (prints GB/s per variant; 3 rounds, interleaved)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <algorithm>
#include <functional>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint16_t bf16_t;

__device__ __forceinline__ float bflo(uint32_t v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bfhi(uint32_t v) { return __builtin_bit_cast(float, v & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    uint32_t a = __builtin_bit_cast(uint32_t, lo), b = __builtin_bit_cast(uint32_t, hi);
    a += 0x7fffu + ((a >> 16) & 1u);
    b += 0x7fffu + ((b >> 16) & 1u);
    return (a >> 16) | (b & 0xffff0000u);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ f32x4 ld4(const float* p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return *reinterpret_cast<const f32x4*>(p);
}
template <bool NT> __device__ __forceinline__ void st4(float* p, f32x4 v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
    else *reinterpret_cast<f32x4*>(p) = v;
}
template <bool NT> __device__ __forceinline__ u32x2 ld2(const bf16_t* p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p));
    return *reinterpret_cast<const u32x2*>(p);
}
template <bool NT> __device__ __forceinline__ void st2(bf16_t* p, u32x2 v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(p));
    else *reinterpret_cast<u32x2*>(p) = v;
}
template <bool NT> __device__ __forceinline__ u32x4 ldq(const bf16_t* p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return *reinterpret_cast<const u32x4*>(p);
}
template <bool NT> __device__ __forceinline__ void stq(bf16_t* p, u32x4 v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
    else *reinterpret_cast<u32x4*>(p) = v;
}

template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_kernel(const float* __restrict__ in, float* __restrict__ out, long nch) {
    for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < nch; c += (long)gridDim.x * 256) st4<NTS>(out + c * 4, ld4<NTL>(in + c * 4));
}

struct AdamP { float lr, b1, b2, eps, wd, bc1, bc2s, gs; };

__device__ __forceinline__ void adam4(f32x4& p, f32x4& mm, f32x4& vv, u32x2 gr, const AdamP& a) {
    float g[4] = {bflo(gr[0]) * a.gs, bfhi(gr[0]) * a.gs, bflo(gr[1]) * a.gs, bfhi(gr[1]) * a.gs};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        p[e] *= (1.f - a.lr * a.wd);
        mm[e] = a.b1 * mm[e] + (1.f - a.b1) * g[e];
        vv[e] = a.b2 * vv[e] + (1.f - a.b2) * g[e] * g[e];
        const float denom = sqrtf(vv[e]) / a.bc2s + a.eps;
        p[e] -= (a.lr / a.bc1) * (mm[e] / denom);
    }
}

// U chunks of 4 elements in flight per thread; a block covers 256*U consecutive chunks per iteration (coalesced per u)
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ master, float* __restrict__ m, float* __restrict__ v,
                                                    const bf16_t* __restrict__ grad, bf16_t* __restrict__ param_out, long nch, AdamP a) {
    for (long c0 = (long)blockIdx.x * 256 * U + threadIdx.x; c0 < nch; c0 += (long)gridDim.x * 256 * U) {
        f32x4 p[U], mm[U], vv[U];
        u32x2 gr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long c = c0 + u * 256;
            if (c < nch) {
                p[u] = ld4<NTL>(master + c * 4);
                mm[u] = ld4<NTL>(m + c * 4);
                vv[u] = ld4<NTL>(v + c * 4);
                gr[u] = ld2<NTL>(grad + c * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long c = c0 + u * 256;
            if (c < nch) {
                adam4(p[u], mm[u], vv[u], gr[u], a);
                st4<NTS>(master + c * 4, p[u]);
                st4<NTS>(m + c * 4, mm[u]);
                st4<NTS>(v + c * 4, vv[u]);
                st2<NTS>(param_out + c * 4, u32x2{pack_bf2(p[u][0], p[u][1]), pack_bf2(p[u][2], p[u][3])});
            }
        }
    }
}

template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ gu, const bf16_t* __restrict__ dout,
                                                         bf16_t* __restrict__ dgu, long rows, int ff) {
    const int nch = ff >> 3;
    const long total = rows * nch;
    for (long t0 = (long)blockIdx.x * 256 * U + threadIdx.x; t0 < total; t0 += (long)gridDim.x * 256 * U) {
        u32x4 g[U], uu[U], d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long t = t0 + u * 256;
            if (t < total) {
                const long r = t / nch;
                const int c = (int)(t % nch);
                g[u] = ldq<NTL>(gu + (size_t)r * 2 * ff + c * 8);
                uu[u] = ldq<NTL>(gu + (size_t)r * 2 * ff + ff + c * 8);
                d[u] = ldq<NTL>(dout + (size_t)r * ff + c * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long t = t0 + u * 256;
            if (t < total) {
                const long r = t / nch;
                const int c = (int)(t % nch);
                u32x4 og, ou;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ga = bflo(g[u][e]), gb = bfhi(g[u][e]);
                    const float siga = 1.f / (1.f + __expf(-ga)), sigb = 1.f / (1.f + __expf(-gb));
                    const float da = bflo(d[u][e]), db = bfhi(d[u][e]);
                    og[e] = pack_bf2(da * bflo(uu[u][e]) * siga * (1.f + ga * (1.f - siga)),
                                     db * bfhi(uu[u][e]) * sigb * (1.f + gb * (1.f - sigb)));
                    ou[e] = pack_bf2(da * ga * siga, db * gb * sigb);
                }
                stq<NTS>(dgu + (size_t)r * 2 * ff + c * 8, og);
                stq<NTS>(dgu + (size_t)r * 2 * ff + ff + c * 8, ou);
            }
        }
    }
}

struct Variant { std::string name; double bytes; std::function<void()> run; std::vector<float> ms; };

int main(int argc, char** argv) {
    const long N = 19531264;                 // one AdamW bucket of the 1.7B step (2.03e9 / 104, multiple of 1024)
    const int NB = 12;                       // buckets swept round-robin (6.6 GB: nothing survives in the 256 MiB Infinity Cache)
    float *master, *m, *v; bf16_t *grad, *param;
    CK(hipMalloc(&master, N * NB * 4)); CK(hipMalloc(&m, N * NB * 4)); CK(hipMalloc(&v, N * NB * 4));
    CK(hipMalloc(&grad, N * NB * 2)); CK(hipMalloc(&param, N * NB * 2));
    CK(hipMemset(master, 0, N * NB * 4)); CK(hipMemset(m, 0, N * NB * 4)); CK(hipMemset(v, 0, N * NB * 4));
    CK(hipMemset(grad, 0x3c, N * NB * 2));
    const long rows = 16384; const int ff = 6144;
    const int NS = 4;
    bf16_t *gu, *dout, *dgu;
    CK(hipMalloc(&gu, rows * 2 * ff * 2 * NS)); CK(hipMalloc(&dout, rows * ff * 2 * NS)); CK(hipMalloc(&dgu, rows * 2 * ff * 2 * NS));
    CK(hipMemset(gu, 0x3c, rows * 2 * ff * 2 * NS)); CK(hipMemset(dout, 0x3c, rows * ff * 2 * NS));
    AdamP a{3e-5f, 0.9f, 0.999f, 1e-8f, 0.01f, 0.1f, 0.0316f, 1.f};
    int bucket = 0, slot = 0;
    std::vector<Variant> vs;
    auto grid_cap = [](long items, int per_block, int cap) { long g = (items + per_block - 1) / per_block; return (int)std::min<long>(std::max<long>(g, 1), cap); };
    const long nch = N / 4;
    const long cpn = 536870912 / 16;        // 512 MiB copy
#define ADD(NAME, BYTES, ...) vs.push_back(Variant{NAME, (double)(BYTES), [&]() { __VA_ARGS__; }, {}})
    float *cin = master, *cout = m;
    ADD("copy f32x4 grid2048", cpn * 32.0, hipLaunchKernelGGL((copy_kernel<false, false>), dim3(2048), dim3(256), 0, 0, cin, cout, cpn));
    ADD("copy f32x4 grid2048 nt-store", cpn * 32.0, hipLaunchKernelGGL((copy_kernel<false, true>), dim3(2048), dim3(256), 0, 0, cin, cout, cpn));
    ADD("copy f32x4 grid2048 nt-both", cpn * 32.0, hipLaunchKernelGGL((copy_kernel<true, true>), dim3(2048), dim3(256), 0, 0, cin, cout, cpn));
    ADD("copy f32x4 one-shot", cpn * 32.0, hipLaunchKernelGGL((copy_kernel<false, false>), dim3((int)(cpn / 256)), dim3(256), 0, 0, cin, cout, cpn));
#define ADAM(U, NTL, NTS, GRID) hipLaunchKernelGGL((adamw_kernel<U, NTL, NTS>), dim3(GRID), dim3(256), 0, 0, master + (long)bucket * N, m + (long)bucket * N, \
        v + (long)bucket * N, grad + (long)bucket * N, param + (long)bucket * N, nch, a); bucket = (bucket + 1) % NB
    ADD("adamw U1 grid2048 (product)", N * 28.0, ADAM(1, false, false, 2048));
    ADD("adamw U1 grid1024", N * 28.0, ADAM(1, false, false, 1024));
    ADD("adamw U1 grid4096", N * 28.0, ADAM(1, false, false, 4096));
    ADD("adamw U1 one-shot", N * 28.0, ADAM(1, false, false, (int)((nch + 255) / 256)));
    ADD("adamw U2 grid2048", N * 28.0, ADAM(2, false, false, 2048));
    ADD("adamw U2 grid1024", N * 28.0, ADAM(2, false, false, 1024));
    ADD("adamw U2 one-shot", N * 28.0, ADAM(2, false, false, (int)((nch + 511) / 512)));
    ADD("adamw U4 grid1024", N * 28.0, ADAM(4, false, false, 1024));
    ADD("adamw U1 grid2048 nt-store", N * 28.0, ADAM(1, false, true, 2048));
    ADD("adamw U1 grid2048 nt-both", N * 28.0, ADAM(1, true, true, 2048));
    ADD("adamw U2 grid2048 nt-both", N * 28.0, ADAM(2, true, true, 2048));
    ADD("adamw U2 grid2048 nt-store", N * 28.0, ADAM(2, false, true, 2048));
    ADD("adamw U1 one-shot nt-both", N * 28.0, ADAM(1, true, true, (int)((nch + 255) / 256)));
    ADD("adamw U2 one-shot nt-both", N * 28.0, ADAM(2, true, true, (int)((nch + 511) / 512)));
    ADD("adamw U1 one-shot nt-load", N * 28.0, ADAM(1, true, false, (int)((nch + 255) / 256)));
    ADD("adamw U1 one-shot nt-store", N * 28.0, ADAM(1, false, true, (int)((nch + 255) / 256)));
    ADD("adamw U1 grid8192 nt-both", N * 28.0, ADAM(1, true, true, 8192));
    ADD("adamw U1 grid16384 nt-both", N * 28.0, ADAM(1, true, true, 16384));
    ADD("copy f32x4 one-shot nt-both", cpn * 32.0, hipLaunchKernelGGL((copy_kernel<true, true>), dim3((int)(cpn / 256)), dim3(256), 0, 0, cin, cout, cpn));
    ADD("copy f32x4 grid16384", cpn * 32.0, hipLaunchKernelGGL((copy_kernel<false, false>), dim3(16384), dim3(256), 0, 0, cin, cout, cpn));
    const double sb = (double)rows * ff * 2 * 5;
    const long stot = rows * (ff / 8);
#define SWI(U, NTL, NTS, GRID) hipLaunchKernelGGL((swiglu_bwd_kernel<U, NTL, NTS>), dim3(GRID), dim3(256), 0, 0, gu + (size_t)slot * rows * 2 * ff, \
        dout + (size_t)slot * rows * ff, dgu + (size_t)slot * rows * 2 * ff, rows, ff); slot = (slot + 1) % NS
    ADD("swiglu_bwd U1 grid2048 (product)", sb, SWI(1, false, false, 2048));
    ADD("swiglu_bwd U1 one-shot", sb, SWI(1, false, false, (int)(stot / 256)));
    ADD("swiglu_bwd U2 grid2048", sb, SWI(2, false, false, 2048));
    ADD("swiglu_bwd U2 one-shot", sb, SWI(2, false, false, (int)(stot / 512)));
    ADD("swiglu_bwd U1 grid2048 nt-both", sb, SWI(1, true, true, 2048));
    ADD("swiglu_bwd U1 grid2048 nt-store", sb, SWI(1, false, true, 2048));
    ADD("swiglu_bwd U2 grid2048 nt-both", sb, SWI(2, true, true, 2048));
    ADD("swiglu_bwd U1 one-shot nt-both", sb, SWI(1, true, true, (int)(stot / 256)));
    ADD("swiglu_bwd U2 one-shot nt-both", sb, SWI(2, true, true, (int)(stot / 512)));
    ADD("swiglu_bwd U1 one-shot nt-load", sb, SWI(1, true, false, (int)(stot / 256)));
    ADD("swiglu_bwd U1 grid16384 nt-both", sb, SWI(1, true, true, 16384));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int REP = 12;
    for (int round = 0; round < 4; ++round) {
        for (auto& x : vs) {
            x.run(); x.run();
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < REP; ++i) x.run();
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (round) x.ms.push_back(ms / REP);
        }
    }
    CK(hipGetLastError());
    for (auto& x : vs) {
        std::sort(x.ms.begin(), x.ms.end());
        printf("%-40s  %8.1f us  %7.0f GB/s (best) %7.0f GB/s (median)\n", x.name.c_str(), x.ms[0] * 1e3, x.bytes / x.ms[0] * 1e-6,
               x.bytes / x.ms[x.ms.size() / 2] * 1e-6);
    }
    return 0;
}
