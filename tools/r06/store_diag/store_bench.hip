// How fast can ONE CU store a 256 x 256 bf16 tile (128 KB), and does it depend on the store flavour, the row pattern or on how many CUs store at
// the same moment?  (round 6: the GEMM epilogue costs 5-8 us per tile whatever the wave does around it: profiles/r06_logs/ab_se_*.log)
//   hipcc --offload-arch=gfx950 -O3 -o store_bench store_bench.hip && ./store_bench
// One workgroup of 8 waves per CU; wave (wr, wc) owns rows wr*128 .. +127, columns wc*64 .. +63 as in gemm256_kernel.  Per round every wave issues
// its 16 stores of 16 bytes per lane, then waits vmcnt(0); rounds go to different tiles of a 1 GiB buffer.  s_memtime around the round, averaged.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int FLAVOUR, int PATTERN>
__global__ __launch_bounds__(512) void store_kernel(unsigned short* C, long ld, int tiles_n, int ntiles, int rounds, int active_mod, int spin,
                                                    unsigned long long* out) {
    if ((int)(blockIdx.x % active_mod) != 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fq = lane >> 4;
    unsigned long long total = 0;
    u32x4 v = u32x4{(unsigned)tid, (unsigned)lane, 3u, 4u};
    for (int r = 0; r < rounds; ++r) {
        const int t = (int)((blockIdx.x + (long)r * gridDim.x) % ntiles);
        unsigned short* base = C + (long)(t / tiles_n) * 256 * ld + (long)(t % tiles_n) * 256;
        // ~spin x 64 cycles of nothing (the K loop between two epilogues), so that every round starts from an idle memory pipe
        for (int s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(1);
        __syncthreads();
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                unsigned short* p;
                if (PATTERN == 0) p = base + (long)(wr * 128 + i * 16 + fr) * ld + wc * 64 + sh * 32 + fq * 8;     // 16 rows x 64 B per instruction
                else if (PATTERN == 1) p = base + (long)(wr * 128 + i * 16 + sh * 8 + (lane >> 3)) * ld + wc * 64 + (lane & 7) * 8;  // 8 rows x 128 B per instruction
                else if (PATTERN == 3) p = base + (long)(wr * 128 + i * 16 + fr) * ld + wc * 64 + fq * 16 + sh * 8;      // 16 rows x 4 pieces of 16 B, 32 B apart
                else if (PATTERN == 6) p = base + (long)(wr * 128 + i * 16 + sh * 8 + (fr & 7)) * ld + wc * 64 + (fq + 4 * (fr >> 3)) * 8;  // 8 rows x 128 B, a row's lanes not adjacent
                else p = nullptr;
                v[0] += (unsigned)r;
                if (PATTERN == 2) {
                    // two 8-byte stores per (i, sh): 4 rows x 128 B each (lane = 16 x row + 4-column group)
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        unsigned short* q = base + (long)(wr * 128 + i * 16 + sh * 8 + h * 4 + fq) * ld + wc * 64 + fr * 4;
                        *reinterpret_cast<u32x2*>(q) = u32x2{v[0], v[1] + (unsigned)h};
                    }
                } else if (PATTERN == 5) {
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                    for (int h = 0; h < 2; ++h) {       // the round-1 epilogue: 16 rows x 32 B per 8-byte store
                        unsigned short* q = base + (long)(wr * 128 + i * 16 + fr) * ld + wc * 64 + (sh * 2 + h) * 16 + fq * 4;
                        *reinterpret_cast<u32x2*>(q) = u32x2{v[0], v[1] + (unsigned)h};
                    }
                } else
                if (FLAVOUR == 0) *reinterpret_cast<u32x4*>(p) = v;
                else if (FLAVOUR == 1) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
                else if (FLAVOUR == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
                else if (FLAVOUR == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
                else asm volatile("global_store_dwordx4 %0, %1, off nt sc0 sc1" :: "v"(p), "v"(v) : "memory");
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (r > 0) total += t1 - t0;
    }
    if (tid == 0) out[blockIdx.x] = total / (rounds - 1);
}

int main() {
    const long ld = 8192;                       // 32768 x 8192 bf16 = 512 MiB
    const long rows = 32768;
    unsigned short* C;
    hipMalloc(&C, rows * ld * 2);
    unsigned long long* out;
    hipMalloc(&out, 256 * 8);
    const int tiles_n = (int)(ld / 256), ntiles = (int)(rows / 256) * tiles_n;
    std::vector<unsigned long long> h(256);
    printf("%-34s %8s %10s %10s %10s\n", "flavour / pattern", "CUs", "median cyc", "min cyc", "B/cyc/CU");
#define RUN(F, P, NAME)                                                                                            \
    for (int mod : {1, 2, 8, 32}) {                                                                                \
        hipMemset(out, 0, 256 * 8);                                                                                \
        hipLaunchKernelGGL((store_kernel<F, P>), dim3(256), dim3(512), 0, 0, C, ld, tiles_n, ntiles, 20, mod, 400, out); \
        hipDeviceSynchronize();                                                                                    \
        hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost);                                                  \
        std::vector<unsigned long long> a;                                                                         \
        for (int b = 0; b < 256; b += mod) a.push_back(h[b]);                                                      \
        std::sort(a.begin(), a.end());                                                                             \
        printf("%-34s %8d %10llu %10llu %10.1f\n", NAME, (int)a.size(), a[a.size() / 2], a[0], 131072.0 / (double)a[a.size() / 2]); \
    }
    RUN(0, 0, "plain, 16 rows x 64 B");
    RUN(0, 1, "plain, 8 rows x 128 B");
    RUN(0, 2, "plain 8 B/lane, 4 rows x 128 B");
    RUN(0, 3, "plain, 16 rows x 4 x 16 B (32 apart)");
    RUN(0, 5, "plain 8 B/lane, 16 rows x 32 B");
    RUN(0, 6, "plain, 8 rows x 128 B, lanes apart");
    return 0;
    RUN(1, 0, "nontemporal, 16 rows x 64 B");
    RUN(1, 1, "nontemporal, 8 rows x 128 B");
    RUN(2, 0, "sc0 sc1, 16 rows x 64 B");
    RUN(2, 1, "sc0 sc1, 8 rows x 128 B");
    RUN(3, 1, "sc1, 8 rows x 128 B");
    RUN(4, 1, "nt sc0 sc1, 8 rows x 128 B");
    return 0;
}
