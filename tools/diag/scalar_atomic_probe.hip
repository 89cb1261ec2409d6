// Does s_atomic_add (SMEM atomic, returns through lgkmcnt) work on this chip?  512 workgroups x 8 waves each add 1 to one counter
// and record the value they got back: the values must be a permutation of 0 .. 4095 and the counter must read 4096.
// hipcc --offload-arch=gfx950 -O3 -o tools/diag/build/scalar_atomic_probe tools/diag/scalar_atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k(unsigned* ctr, unsigned* out) {
    unsigned v = 1;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(ctr) : "memory");
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = v;
}
__global__ void rd(unsigned* ctr, unsigned* out) {
    unsigned w;
    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(ctr) : "memory");
    if (threadIdx.x == 0) out[0] = w;
}
int main() {
    unsigned *ctr, *out, *o2;
    hipMalloc(&ctr, 64); hipMalloc(&out, 4096 * 4); hipMalloc(&o2, 64);
    hipMemset(ctr, 0, 64);
    hipLaunchKernelGGL(k, dim3(512), dim3(512), 0, 0, ctr, out);
    hipLaunchKernelGGL(rd, dim3(1), dim3(64), 0, 0, ctr, o2);
    std::vector<unsigned> h(4096); unsigned c = 0, c2 = 0;
    hipMemcpy(h.data(), out, 4096 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(&c, ctr, 4, hipMemcpyDeviceToHost); hipMemcpy(&c2, o2, 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    bool perm = true; for (unsigned i = 0; i < 4096; ++i) perm &= h[i] == i;
    printf("counter %u (s_load glc reads %u), returned values a permutation of 0..4095: %s (min %u max %u)\n", c, c2, perm ? "yes" : "NO", h[0], h[4095]);
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}
