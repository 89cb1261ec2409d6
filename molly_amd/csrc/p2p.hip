// Direct peer exchange for the ZeRO-2 step (SURVEY.md §5 option 3; reference role: DeepSpeed's reduce-scatter / all-gather of
// src/configs/ds_z2_config.json:18-27 behind deepspeed.initialize, src/train.py:606-614).
// xGMI is point to point: every GPU of the node has a link of its own to each of the 7 others.  The owner of a gradient chunk can
// therefore READ that chunk from the seven peers' gradient buffers directly — seven links carrying one chunk each at the same time,
// no ring, no staging buffer — and add the eight copies in fp32 in rank order (the arithmetic of molly_reduce_rows_bf16, i.e. of
// rs_algo = "a2a": one rounding per element, an order no collective algorithm can change); and it can WRITE its updated parameter
// chunk straight into the seven peers' parameter buffers.  The peers' buffers are mapped into this process by the runtime's IPC
// (molly_amd/trainer/p2p.py does that with torch's own CUDA-IPC handles); what the kernels need beyond plain loads and stores is
// the hand-shake: "my gradients of this bucket are final", "my parameter chunk has arrived in your buffer".
//   flags live in memory every peer has mapped; a flag only ever grows (the step's sequence number), so there is nothing to reset;
//   the producer's flag store is a system-scope release in a kernel of its own BEHIND the kernels that produced the data (data
//   written by earlier kernels of the stream is visible device-wide at that kernel boundary; the release orders the flag behind it),
//   the consumer spins on system-scope acquire loads — bounded by a WALL-CLOCK deadline (s_memrealtime, 100 MHz): a peer that never
//   arrives raises the error word instead of hanging the GPU — in a kernel of its own IN FRONT of the kernels that consume.
//   A raised error word is FATAL for the step (round 6): molly_p2p_reduce_bf16 then writes NaN instead of sums, so the
//   all-reduced gradient norm is not finite on every rank and the optimizer step is skipped everywhere (the existing non-finite-norm
//   skip) — no rank ever applies a sum with a stale term — and the host side (trainer/p2p.py) reads the word, which lives in
//   host-visible memory, at its next call and raises.
// Validated on ONE GPU with two and four processes (tests/test_gpu_two_ranks.py: bit-identical to rs_algo = "a2a"); never yet run
// over links — no speed is claimed for it.
#include "common.h"
#include "molly_hip.h"

namespace {

constexpr int P2P_MAX_WORLD = 16;
struct PeerPtrs { const void* p[P2P_MAX_WORLD]; };
struct PeerPtrsW { void* p[P2P_MAX_WORLD]; };

// out[i] = bf16( sum_r float(src_r[i]) ), r = 0 .. world-1 in that order (src_r = peer r's copy of this rank's chunk)
// (out is NOT restrict: it is this rank's own chunk, i.e. src.p[rank] — each 16-byte piece is read before it is written by the same lane)
__global__ __launch_bounds__(256) void p2p_reduce_kernel(PeerPtrs src, int world, long n, bf16_t* out, const int* err) {
    const long nch = n >> 3;
    // a wait in front of this launch gave up: some peer's copy may not be final — poison the result instead of summing stale gradients
    const bool poisoned = err != nullptr && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
    for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < nch; c += (long)gridDim.x * 256) {
        if (poisoned) {
            *reinterpret_cast<u32x4*>(out + c * 8) = u32x4{0x7FC07FC0u, 0x7FC07FC0u, 0x7FC07FC0u, 0x7FC07FC0u};     // bf16 NaN
            continue;
        }
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int r = 0; r < world; ++r) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(src.p[r]) + c * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[2 * j] += bflo(v[j]); acc[2 * j + 1] += bfhi(v[j]); }
        }
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = pack_bf2(acc[2 * j], acc[2 * j + 1]);
        *reinterpret_cast<u32x4*>(out + c * 8) = o;
    }
}

// dst_r[i] = src[i] for every peer r != skip (the rank's own buffer holds the chunk already)
__global__ __launch_bounds__(256) void p2p_push_kernel(const bf16_t* __restrict__ src, PeerPtrsW dst, int world, int skip, long n) {
    const long nch = n >> 3;
    for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < nch; c += (long)gridDim.x * 256) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + c * 8);
        for (int r = 0; r < world; ++r)
            if (r != skip) *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(dst.p[r]) + c * 8) = v;
    }
    // the peers' copies are other devices' memory: make this thread's writes visible system-wide before the kernel ends, not only at the
    // boundary (the flag kernel behind this one releases at system scope as well; never yet exercised over links — ADVICE r05)
    __threadfence_system();
}

__global__ void p2p_flag_set_kernel(int* flag, int value) {
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// one lane per peer: spin until flags[r][idx] >= value; a peer that does not arrive within `timeout_us` microseconds of WALL-CLOCK time
// (s_memrealtime: a constant 100 MHz, whatever the shader clock does) raises *err.  A poll count would be a different time on every
// box and under every clock.
__global__ void p2p_flag_wait_kernel(PeerPtrs flags, int world, int idx, int value, long timeout_us, int* err) {
    const int r = threadIdx.x;
    if (r >= world) return;
    const int* f = reinterpret_cast<const int*>(flags.p[r]) + idx;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long budget = (unsigned long long)timeout_us * 100ull;
    while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < value) {
        __builtin_amdgcn_s_sleep(8);
        if (__builtin_amdgcn_s_memrealtime() - t0 > budget) {
            __hip_atomic_store(err, 1 + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
    }
}

inline int grid_for(long items) { return (int)((items + 255) / 256 < 2048 ? (items + 255) / 256 : 2048); }

}  // namespace

extern "C" int molly_p2p_reduce_bf16(void* stream, const void* const* srcs, int world, long n, void* out, const int* err) {
    MOLLY_ENTER();
    MOLLY_CHECK(world >= 1 && world <= P2P_MAX_WORLD && n >= 0 && n % 8 == 0, "p2p_reduce: world=%d n=%ld (n %% 8 == 0, world <= 16)", world, n);
    if (n == 0) return 0;
    PeerPtrs s{};
    for (int r = 0; r < world; ++r) {
        MOLLY_CHECK(srcs[r] && ((uintptr_t)srcs[r] % 16) == 0, "p2p_reduce: peer %d's buffer is null or not 16-byte aligned", r);
        s.p[r] = srcs[r];
    }
    MOLLY_CHECK(((uintptr_t)out % 16) == 0, "p2p_reduce: out alignment");
    hipLaunchKernelGGL(p2p_reduce_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, s, world, n, (bf16_t*)out, err);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_p2p_push_bf16(void* stream, const void* src, void* const* dsts, int world, int skip, long n) {
    MOLLY_ENTER();
    MOLLY_CHECK(world >= 1 && world <= P2P_MAX_WORLD && n >= 0 && n % 8 == 0, "p2p_push: world=%d n=%ld", world, n);
    if (n == 0 || world == 1) return 0;
    PeerPtrsW d{};
    for (int r = 0; r < world; ++r) {
        MOLLY_CHECK(r == skip || (dsts[r] && ((uintptr_t)dsts[r] % 16) == 0), "p2p_push: peer %d's buffer is null or not 16-byte aligned", r);
        d.p[r] = dsts[r];
    }
    hipLaunchKernelGGL(p2p_push_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, d, world, skip, n);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_p2p_flag_set(void* stream, int* flag, int value) {
    MOLLY_ENTER();
    hipLaunchKernelGGL(p2p_flag_set_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, value);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_p2p_flag_wait(void* stream, const void* const* flags, int world, int idx, int value, long timeout_us, int* err) {
    MOLLY_ENTER();
    MOLLY_CHECK(world >= 1 && world <= P2P_MAX_WORLD && err && timeout_us > 0, "p2p_flag_wait: world=%d timeout_us=%ld", world, timeout_us);
    PeerPtrs f{};
    for (int r = 0; r < world; ++r) f.p[r] = flags[r];
    hipLaunchKernelGGL(p2p_flag_wait_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, f, world, idx, value, timeout_us, err);
    MOLLY_LAUNCH_CHECK();
    return 0;
}
