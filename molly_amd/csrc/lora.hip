// Elementwise pieces of the LoRA branch  y = W x + (alpha/r) * B (A dropout(x))
// (PEFT lora.Linear.forward as configured by the reference: src/utils/tools.py:379-389 — r = --lora_r, alpha 64, dropout 0.05).
// The rank-r contractions themselves run on the library's bf16 MFMA GEMM (gemm.hip); this file has the HBM-bound parts.
#include "common.h"
#include "molly_hip.h"

namespace {

// Philox4x32-10 (common.h): element i's keep bit is a pure function of (seed, i), the backward regenerates the mask.
#ifndef MOLLY_DROPOUT_PHILOX_ROUNDS
#define MOLLY_DROPOUT_PHILOX_ROUNDS 10     // (7, the fewest the paper lists as passing BigCrush, measured no faster: the fused kernels are HBM-bound)
#endif

// out[i] (+)= keep_i ? x[i] / (1 - p) : 0, 8 elements (one Philox block = eight 16-bit uniforms) per thread.
// P(keep) = 1 - thr / 65536 with thr = round(p * 65536): p = 0.05 -> 0.0500031.
template <bool ACC>
__global__ __launch_bounds__(256) void dropout_kernel(const bf16_t* x, bf16_t* out, long nch,   // may alias (in place)
                                                      uint32_t thr, float inv_keep, uint32_t seed_lo, uint32_t seed_hi) {
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < nch; t += (long)gridDim.x * 256) {
        uint32_t rnd[4];
        philox4x32<MOLLY_DROPOUT_PHILOX_ROUNDS>((uint32_t)t, (uint32_t)(t >> 32), 0u, 0u, seed_lo, seed_hi, rnd);
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

// ================================================================================================
// The two rank-r contractions that touch the dropout mask, each fused with it (round 5).  A LoRA step at Molly-1.7B, r = 64, p = 0.05,
// 16 x 2,048 tokens ran 392 stand-alone dropout launches (29.5 ms of a 336 ms step) beside 425 rank-64 GEMM launches on the 128 x 128
// kernel (28.8 ms: they are operand streams, not matrix work):
//   forward  t = s * dropout(x) A^T        was  dropout (x -> xd: read + write [M, in])  +  GEMM (read xd again)
//   backward dx += mask * (dt A)           was  GEMM (write tmp [M, in])  +  dropout-accumulate (read tmp, read + write dx)
// Here the mask is applied where the operand already sits in registers: on x's way into LDS (the forward also writes xd, which the
// dA weight gradient reads), and on the product's way out to dx.  Both kernels spend their time in Philox (one block per 8 elements),
// about what the stand-alone dropout kernel needs by itself.  The mask is the same function of (seed, element index) as
// dropout_kernel's, so fused and unfused launches can be mixed (tests/test_gpu_lora.py compares them).
// ================================================================================================

// keep bits of chunk t (8 consecutive elements): element 2e keeps iff (rnd[e] & 0xffff) >= thr, element 2e + 1 iff (rnd[e] >> 16) >= thr
__device__ __forceinline__ u32x4 drop_chunk(const u32x4& v, long t, uint32_t thr, float inv_keep, uint32_t seed_lo, uint32_t seed_hi) {
    uint32_t rnd[4];
    philox4x32<MOLLY_DROPOUT_PHILOX_ROUNDS>((uint32_t)t, (uint32_t)(t >> 32), 0u, 0u, seed_lo, seed_hi, rnd);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a = (rnd[e] & 0xffffu) >= thr ? bflo(v[e]) * inv_keep : 0.f;
        const float b = (rnd[e] >> 16) >= thr ? bfhi(v[e]) * inv_keep : 0.f;
        o[e] = pack_bf2(a, b);
    }
    return o;
}

// ---- forward: t[M, 64] = scale * (dropout(x)[M, K] A[64, K]^T), xd = dropout(x) written on the way (nullable).
// Block = 64 rows of x, 4 waves; K walked in tiles of 64.  A tile of x is loaded with 16-byte coalesced loads (8 rows x 128 B per
// wave-instruction) one tile ahead, masked in registers, stored to xd and to LDS (16-byte chunks XOR-swizzled by the row: the MFMA
// fragment reads — 32 rows, one chunk — are conflict-free, guide T2).  D'[r][m] = A[r][k] xd[m][k] (swapped operands: a lane owns one
// row m of t with four consecutive r per register quad, so t leaves in 8-byte pieces of its rows).
#ifndef MOLLY_LORA_DOWN_AHEAD
#define MOLLY_LORA_DOWN_AHEAD 4
#endif
#ifndef MOLLY_LORA_UP_AHEAD
#define MOLLY_LORA_UP_AHEAD 1
#endif
constexpr int LD_PD = MOLLY_LORA_DOWN_AHEAD;     // K-steps of x and A in flight per workgroup
constexpr int LD_BM = 64, LD_BK = 64;      // (32-row blocks — twice the workgroups, each wave a 16 x 32 piece — measured below)
template <bool WRITE_XD, bool DROP = true>
__global__ __launch_bounds__(256) void lora_down_drop_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ A, bf16_t* __restrict__ xd,
                                                             bf16_t* __restrict__ t, int M, int K, int ldx, int ldt, uint32_t thr, float inv_keep,
                                                             uint32_t seed_lo, uint32_t seed_hi, float scale, bf16_t* __restrict__ tT, int ldtT) {
    __shared__ __attribute__((aligned(16))) bf16_t xs[2][LD_BM * LD_BK], as[2][64 * LD_BK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * LD_BM;
    const int mb = wave & 1, rh = wave >> 1;                    // this wave: rows 32 mb .. of the block, adapter rows 32 rh ..
    // staging roles: chunk c = tid (+ 256): row c >> 3, 16-byte chunk c & 7 — the same for the x tile and the A tile
    int srow[2], sch[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) { const int c = tid + 256 * q; srow[q] = c >> 3; sch[q] = c & 7; }
    // x and A tiles travel LD_PD K-steps ahead in registers: with one tile in flight per workgroup (two workgroups per CU at 32k rows) the kernel
    // ran at the latency of its loads — 2.6 TB/s of a stream that has no other cost
    u32x4 xv[LD_PD][2], av[LD_PD][2];
    auto load = [&](int k0, u32x4 (&xq)[2], u32x4 (&aq)[2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int m = m0 + srow[q];
            xq[q] = m < M ? *reinterpret_cast<const u32x4*>(x + (size_t)m * ldx + k0 + 8 * sch[q]) : u32x4{0, 0, 0, 0};
            aq[q] = *reinterpret_cast<const u32x4*>(A + (size_t)srow[q] * K + k0 + 8 * sch[q]);
        }
    };
    auto put = [&](int buf, int k0, const u32x4 (&xq)[2], const u32x4 (&aq)[2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int m = m0 + srow[q];
            const u32x4 d = DROP ? drop_chunk(xq[q], ((long)m * K + k0) / 8 + sch[q], thr, inv_keep, seed_lo, seed_hi) : xq[q];
            if (WRITE_XD && m < M) *reinterpret_cast<u32x4*>(xd + (size_t)m * K + k0 + 8 * sch[q]) = d;
            const int sw = (sch[q] ^ (srow[q] & 7)) * 8;
            *reinterpret_cast<u32x4*>(&xs[buf][srow[q] * LD_BK + sw]) = d;
            *reinterpret_cast<u32x4*>(&as[buf][srow[q] * LD_BK + sw]) = aq[q];
        }
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int r = lane & 31, h = lane >> 5;
    const int nk = K / LD_BK;
#pragma unroll
    for (int u = 0; u < LD_PD; ++u)
        if (u < nk) load(u * LD_BK, xv[u], av[u]);
    for (int kt0 = 0; kt0 < nk; kt0 += LD_PD) {
#pragma unroll
        for (int u = 0; u < LD_PD; ++u) {
            const int kt = kt0 + u;
            if (kt >= nk) break;                                   // (block-uniform)
            const int buf = kt & 1;
            put(buf, kt * LD_BK, xv[u], av[u]);
            __syncthreads();                                       // tile kt visible; everyone is done with buffer buf ^ 1's reads (tile kt - 1)
            if (kt + LD_PD < nk) load((kt + LD_PD) * LD_BK, xv[u], av[u]);
            const bf16_t* ar = &as[buf][(32 * rh + r) * LD_BK];
            const bf16_t* xr = &xs[buf][(32 * mb + r) * LD_BK];
#pragma unroll
            for (int s4 = 0; s4 < LD_BK / 16; ++s4) {
                const int ch = ((2 * s4 + h) ^ (r & 7)) * 8;       // rows 32 j + r: (row & 7) == (r & 7)
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(ar + ch);
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(xr + ch);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, xf, acc, 0, 0, 0);
            }
        }
    }
    const int m = m0 + 32 * mb + r;
    if (m < M) {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
            *reinterpret_cast<u32x2*>(t + (size_t)m * ldt + 32 * rh + 8 * g4 + 4 * h) =
                u32x2{pack_bf2(acc[4 * g4] * scale, acc[4 * g4 + 1] * scale), pack_bf2(acc[4 * g4 + 2] * scale, acc[4 * g4 + 3] * scale)};
        // t^T [64][ldtT] as well (the adapter weight gradients' k-contiguous operand: dB^T = t^T dy, dA = dt^T dropout(x)) — the accumulators already
        // hold it with the row index across lanes: 32 lanes = 64 contiguous bytes of one rank row per store.  Saves a transpose launch per target.
        if (tT) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
                tT[(size_t)(32 * rh + 8 * (e >> 2) + 4 * h + (e & 3)) * ldtT + m] = f2bf(acc[e] * scale);
        }
    }
}

// ---- backward: dx[M, K] += mask * bf16(dt[M, 64] A[64, K]).  The product is the P.V half of attention with 64 "keys" (the adapter
// rank) and K / 128 "heads": dx^T[n][m] += A^T[n][r] dt^T[r][m], A's tile [64 r][128 n] staged row-major by LDS-DMA into the sub-tiled
// image whose transposed reads are conflict-free (attention.hip img_off / tr_frag, restated here for a 128-column tile), dt^T as the
// register operand in the k order those reads deliver.  A lane then owns one row m of the 32 x 128 result; the rows cross a
// wave-private LDS slab (bf16: the rounding the unfused GEMM applied to its output) and leave as 16-byte chunks — one Philox block
// each — added into dx.  Block = 128 rows x LU_NCH columns (4 waves x 32 rows; the column chunks of a row band are separate workgroups).
constexpr int LU_BN = 128, LU_NCH = 512;
__device__ __forceinline__ void lu_stage(const bf16_t* __restrict__ g, int ld, bf16_t* lds, int wave, int lane) {       // [64 rows][128 cols]
    constexpr int NCB = LU_BN / 16;
    const char* base = reinterpret_cast<const char*>(g);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int inst = wave * 4 + i;
        const int P = inst * 64 + lane;
        const int blk = P >> 3, cin = P & 7;
        const int rowblk = blk / NCB, cbs = blk % NCB;
        const int b0 = rowblk & 1, b1 = (rowblk >> 1) & 1;
        const int row = rowblk * 4 + (cin >> 1);
        const int col = ((cbs ^ b0) << 4) + (((cin & 1) ^ b1) << 3);
        const unsigned off = ((unsigned)row * (unsigned)ld + (unsigned)col) * 2u;
        const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(lds + inst * 512));
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds_addr) : "memory", "m0");
    }
}
__device__ __forceinline__ bf16x8 lu_tr_frag(const bf16_t* tile, int row0, int sp, int dt, int lane) {
    constexpr int NCB = LU_BN / 16;
    const int h = lane >> 5, g1 = (lane >> 4) & 1, gi = lane & 15, gq = gi >> 2, gp = gi & 3;
    const int base_a = (h * NCB + (g1 ^ h)) * 64 + (gq * 2 + (gp >> 1)) * 8 + (gp & 1) * 4;
    const int base_b = ((h + 2) * NCB + (g1 ^ h)) * 64 + (gq * 2 + ((gp >> 1) ^ 1)) * 8 + (gp & 1) * 4;
    const int cst = ((row0 >> 2) + 4 * sp) * NCB * 64 + dt * 128;
    const bf16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(tile + base_a + cst));
    const bf16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(tile + base_b + cst));
    return bf16x8{va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
}

// NTG targets in one pass over dx (NTG = 3: q | k | v of one input, 2: gate | up): dx = bf16(... bf16(bf16(dx + m_0 * z_0) + m_1 * z_1) ...) — the roundings of NTG
// launches one after the other, bit for bit, with dx read and written ONCE.  The A tiles of (column tile, target) pairs stream through the two LDS buffers.
struct UpDropArgs {
    const bf16_t* dt[3]; const bf16_t* A[3];
    int lddt[3];
    uint32_t seed_lo[3], seed_hi[3];
};
template <int NTG>
__global__ __launch_bounds__(256, 2) void lora_up_drop_acc_kernel(UpDropArgs q, bf16_t* __restrict__ dx, int M, int K, uint32_t thr, float inv_keep) {
    extern __shared__ __attribute__((aligned(16))) char lu_smem[];
    bf16_t* tiles = reinterpret_cast<bf16_t*>(lu_smem);                     // [2][64 x 128]
    constexpr int PITCH = LU_BN + 8;
    bf16_t* slab = tiles + 2 * 64 * LU_BN + (threadIdx.x >> 6) * 32 * PITCH;  // wave-private [32][128 + 8]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 128 + wave * 32;
    const int n_begin = blockIdx.y * LU_NCH, n_end = min(K, n_begin + LU_NCH);
    // dt^T fragments (the register operand), in the k order of the transposed reads: rank indices 16 sp + {4h .. 4h+3, 8 + 4h .. 8 + 4h+3}
    bf16x8 pf[NTG][2][2];
    {
        const int m = min(m0 + r, M - 1);
#pragma unroll
        for (int u = 0; u < NTG; ++u) {
            const bf16_t* dp = q.dt[u] + (size_t)m * q.lddt[u];
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) {
                    const u32x2 lo = *reinterpret_cast<const u32x2*>(dp + 32 * sub + 16 * sp + 4 * h);
                    const u32x2 hi = *reinterpret_cast<const u32x2*>(dp + 32 * sub + 16 * sp + 8 + 4 * h);
                    pf[u][sub][sp] = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
                }
        }
#pragma unroll
        for (int u = 0; u < NTG; ++u)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp) asm volatile("" ::"v"(pf[u][sub][sp]));    // arrived: their waits must not land among the DMA waits
    }
    lu_stage(q.A[0] + n_begin, K, tiles, wave, lane);
    int cur = 0;
    // dx's chunks of the NEXT column tile are requested before this tile's masks are computed (MOLLY_LORA_UP_AHEAD): requested where
    // they are used, every tile paid the latency of its eight loads on top of the Philox arithmetic
    const int lr = lane >> 4, lc = (lane & 15) * 8;
    u32x4 oq[8];
    auto load_dx = [&](int n0, u32x4 (&o)[8]) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int m = m0 + it * 4 + lr;
            o[it] = m < M ? *reinterpret_cast<const u32x4*>(dx + (size_t)m * K + n0 + lc) : u32x4{0, 0, 0, 0};
        }
    };
    if (MOLLY_LORA_UP_AHEAD) load_dx(n_begin, oq);
    for (int n0 = n_begin; n0 < n_end; n0 += LU_BN) {
        u32x4 ocur[8];
#pragma unroll
        for (int u = 0; u < NTG; ++u) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                   // tile (n0, u) landed for everyone; the other buffer's readers are done
            if (u + 1 < NTG) lu_stage(q.A[u + 1] + n0, K, tiles + (cur ^ 1) * 64 * LU_BN, wave, lane);
            else if (n0 + LU_BN < n_end) lu_stage(q.A[0] + n0 + LU_BN, K, tiles + (cur ^ 1) * 64 * LU_BN, wave, lane);
            const bf16_t* tile = tiles + cur * 64 * LU_BN;
            f32x16 acc[4];
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[d][e] = 0.f;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lu_tr_frag(tile, 32 * sub, sp, d, lane), pf[u][sub][sp], acc[d], 0, 0, 0);
            // rows of the wave's 32 x 128 result through its slab (bf16), then 16-byte chunks: mask, scale, add into dx
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    *reinterpret_cast<u32x2*>(slab + r * PITCH + 32 * d + 8 * g4 + 4 * h) =
                        u32x2{pack_bf2(acc[d][4 * g4], acc[d][4 * g4 + 1]), pack_bf2(acc[d][4 * g4 + 2], acc[d][4 * g4 + 3])};
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (u == 0) {
#pragma unroll
                for (int it = 0; it < 8; ++it) ocur[it] = oq[it];
                if (MOLLY_LORA_UP_AHEAD) {
                    if (n0 + LU_BN < n_end) load_dx(n0 + LU_BN, oq);
                } else {
                    load_dx(n0, ocur);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 4 + lr;
                const int m = m0 + row;
                if (m < M) {
                    const u32x4 o = ocur[it];
                    const u32x4 v = *reinterpret_cast<const u32x4*>(slab + row * PITCH + lc);
                    const u32x4 d = drop_chunk(v, ((long)m * K + n0 + lc) / 8, thr, inv_keep, q.seed_lo[u], q.seed_hi[u]);
                    u32x4 w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = pack_bf2(bflo(d[e]) + bflo(o[e]), bfhi(d[e]) + bfhi(o[e]));
                    ocur[it] = w;
                    if (u + 1 == NTG) *reinterpret_cast<u32x4*>(dx + (size_t)m * K + n0 + lc) = w;
                }
            }
            // (the slab is the wave's own: its reads above are ordered before the next target's writes by the LDS queue)
            cur ^= 1;
        }
    }
}

// ---- batched block copy: item i copies rows x 64 bf16 (one padded-rank adapter matrix, 128-byte rows) from src (row stride 64) into dst (row
// stride ld_dst) — the diagonal blocks of the stacked B matrices a fused projection's K-extended GEMM reads (molly_gemm_kx_bf16_ctx), refreshed
// once per forward for every layer in ONE launch (the adapters change at every optimizer step; the off-diagonal zeros never do).
struct PackItem { const bf16_t* src; bf16_t* dst; int rows; int ld_dst; };
__global__ __launch_bounds__(256) void lora_pack_kernel(const PackItem* __restrict__ items) {
    const PackItem it = items[blockIdx.y];
    const int row = blockIdx.x * 32 + (threadIdx.x >> 3), ch = threadIdx.x & 7;
    if (row < it.rows)
        *reinterpret_cast<u32x4*>(it.dst + (size_t)row * it.ld_dst + 8 * ch) = *reinterpret_cast<const u32x4*>(it.src + (size_t)row * 64 + 8 * ch);
}

// the same table format, transposing: src [rows][64] -> dst [64][ld_dst] (B^T of every target, the `A` operand of the backward's dt = s * dy B skinny
// product), all layers in one launch per forward instead of a transpose launch per target and layer in the backward
__global__ __launch_bounds__(256) void lora_pack_t_kernel(const PackItem* __restrict__ items) {
    const PackItem it = items[blockIdx.y];
    const int row = blockIdx.x * 32 + (threadIdx.x & 31), ch = threadIdx.x >> 5;      // 32 consecutive rows across a half-wave: 64-byte stores
    if (row < it.rows) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(it.src + (size_t)row * 64 + 8 * ch);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            it.dst[(size_t)(8 * ch + 2 * e) * it.ld_dst + row] = (bf16_t)(v[e] & 0xffffu);
            it.dst[(size_t)(8 * ch + 2 * e + 1) * it.ld_dst + row] = (bf16_t)(v[e] >> 16);
        }
    }
}

}  // namespace

extern "C" int molly_lora_pack_bt(void* stream, const void* items_dev, int n_items, int max_rows) {
    MOLLY_ENTER();
    MOLLY_CHECK(items_dev && n_items >= 1 && n_items <= 65535 && max_rows >= 1, "lora_pack_bt: %d items, max_rows=%d", n_items, max_rows);
    hipLaunchKernelGGL(lora_pack_t_kernel, dim3((max_rows + 31) / 32, n_items), dim3(256), 0, (hipStream_t)stream, (const PackItem*)items_dev);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_lora_pack_b(void* stream, const void* items_dev, int n_items, int max_rows) {
    MOLLY_ENTER();
    MOLLY_CHECK(items_dev && n_items >= 1 && n_items <= 65535 && max_rows >= 1, "lora_pack_b: %d items, max_rows=%d", n_items, max_rows);
    hipLaunchKernelGGL(lora_pack_kernel, dim3((max_rows + 31) / 32, n_items), dim3(256), 0, (hipStream_t)stream, (const PackItem*)items_dev);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

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

extern "C" int molly_lora_down_drop_t_bf16(void* stream, const void* x, const void* A, void* xd, void* t, int M, int K, int R, int ldx, int ldt,
                                           float p, uint64_t seed, float scale, void* tT, int ldtT);
extern "C" int molly_lora_down_drop_bf16(void* stream, const void* x, const void* A, void* xd, void* t, int M, int K, int R, int ldx, int ldt,
                                         float p, uint64_t seed, float scale) {
    return molly_lora_down_drop_t_bf16(stream, x, A, xd, t, M, K, R, ldx, ldt, p, seed, scale, nullptr, 0);
}
extern "C" int molly_lora_down_drop_t_bf16(void* stream, const void* x, const void* A, void* xd, void* t, int M, int K, int R, int ldx, int ldt,
                                           float p, uint64_t seed, float scale, void* tT, int ldtT) {
    MOLLY_ENTER();
    MOLLY_CHECK(!tT || ldtT >= M, "lora_down_drop: ldtT=%d < M=%d", ldtT, M);
    MOLLY_CHECK(M > 0 && K > 0 && K % 64 == 0, "lora_down_drop: M=%d K=%d (K must be a positive multiple of 64)", M, K);
    MOLLY_CHECK(R == 64, "lora_down_drop: padded rank %d (built for 64)", R);
    MOLLY_CHECK(ldt % 4 == 0 && ldt >= R && ldx % 8 == 0 && ldx >= K, "lora_down_drop: ldt=%d ldx=%d", ldt, ldx);
    MOLLY_CHECK(((uintptr_t)x % 16) == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)t % 8) == 0 && ((uintptr_t)xd % 16) == 0, "lora_down_drop: alignment");
    MOLLY_CHECK(p >= 0.f && p < 1.f, "lora_down_drop: p=%f not in [0,1)", (double)p);
    const uint32_t thr = (uint32_t)(p * 65536.f + 0.5f);
    const dim3 grid((M + LD_BM - 1) / LD_BM);
    if (p == 0.f && !xd)          // no mask: the plain skinny product t = scale * x A^T (the backward's dt = s * dy B with A = B^T)
        hipLaunchKernelGGL((lora_down_drop_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)A,
                           (bf16_t*)nullptr, (bf16_t*)t, M, K, ldx, ldt, 0u, 1.f, 0u, 0u, scale, (bf16_t*)tT, ldtT);
    else if (xd)
        hipLaunchKernelGGL(lora_down_drop_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)A, (bf16_t*)xd,
                           (bf16_t*)t, M, K, ldx, ldt, thr, 1.f / (1.f - p), (uint32_t)seed, (uint32_t)(seed >> 32), scale, (bf16_t*)tT, ldtT);
    else
        hipLaunchKernelGGL(lora_down_drop_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)A, (bf16_t*)nullptr,
                           (bf16_t*)t, M, K, ldx, ldt, thr, 1.f / (1.f - p), (uint32_t)seed, (uint32_t)(seed >> 32), scale, (bf16_t*)tT, ldtT);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_lora_up_drop_acc_multi_bf16(void* stream, int n, const void* const* dt, const void* const* A, void* dx, int M, int K, int R,
                                                 const int* lddt, float p, const uint64_t* seed) {
    MOLLY_ENTER();
    MOLLY_CHECK(n >= 1 && n <= 3 && dt && A && lddt && seed, "lora_up_drop_acc: %d targets (1..3)", n);
    MOLLY_CHECK(M > 0 && K > 0 && K % 128 == 0, "lora_up_drop_acc: M=%d K=%d (K must be a positive multiple of 128)", M, K);
    MOLLY_CHECK(R == 64, "lora_up_drop_acc: padded rank %d (built for 64)", R);
    MOLLY_CHECK(p >= 0.f && p < 1.f, "lora_up_drop_acc: p=%f not in [0,1)", (double)p);
    MOLLY_CHECK(((uintptr_t)dx % 16) == 0, "lora_up_drop_acc: alignment");
    UpDropArgs q{};
    for (int u = 0; u < n; ++u) {
        MOLLY_CHECK(lddt[u] % 4 == 0 && lddt[u] >= R, "lora_up_drop_acc: lddt=%d", lddt[u]);
        MOLLY_CHECK(dt[u] && A[u] && ((uintptr_t)dt[u] % 8) == 0 && ((uintptr_t)A[u] % 16) == 0, "lora_up_drop_acc: alignment (target %d)", u);
        q.dt[u] = (const bf16_t*)dt[u]; q.A[u] = (const bf16_t*)A[u]; q.lddt[u] = lddt[u];
        q.seed_lo[u] = (uint32_t)seed[u]; q.seed_hi[u] = (uint32_t)(seed[u] >> 32);
    }
    const uint32_t thr = (uint32_t)(p * 65536.f + 0.5f);
    const size_t lds = (2 * 64 * LU_BN + 4 * 32 * (LU_BN + 8)) * sizeof(bf16_t);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)lora_up_drop_acc_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)lora_up_drop_acc_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)lora_up_drop_acc_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    const dim3 grid((M + 127) / 128, (K + LU_NCH - 1) / LU_NCH);
    const float inv_keep = 1.f / (1.f - p);
    if (n == 1) hipLaunchKernelGGL(lora_up_drop_acc_kernel<1>, grid, dim3(256), lds, (hipStream_t)stream, q, (bf16_t*)dx, M, K, thr, inv_keep);
    else if (n == 2) hipLaunchKernelGGL(lora_up_drop_acc_kernel<2>, grid, dim3(256), lds, (hipStream_t)stream, q, (bf16_t*)dx, M, K, thr, inv_keep);
    else hipLaunchKernelGGL(lora_up_drop_acc_kernel<3>, grid, dim3(256), lds, (hipStream_t)stream, q, (bf16_t*)dx, M, K, thr, inv_keep);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_lora_up_drop_acc_bf16(void* stream, const void* dt, const void* A, void* dx, int M, int K, int R, int lddt, float p,
                                           uint64_t seed) {
    return molly_lora_up_drop_acc_multi_bf16(stream, 1, &dt, &A, dx, M, K, R, &lddt, p, &seed);
}

extern "C" int molly_scale_bf16(void* stream, void* x, long n, float s) {
    MOLLY_ENTER();
    MOLLY_CHECK(n > 0 && n % 8 == 0, "scale: n=%ld must be a positive multiple of 8", n);
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, n / 8, s);
    MOLLY_LAUNCH_CHECK();
    return 0;
}
