// Flash attention for gfx950 (MFMA 32x32x16 bf16, online softmax in fp32, LDS-staged K/V tiles).
//   forward : O = softmax(scale * Q K^T + mask) V        + LSE (log2 domain) for the backward
//   backward: dQ, dK, dV by recomputation from LSE (attention_bwd.hip)
// replaces FlashAttention-2 / SDPA / eager attention selected by `--attn_impl` (reference src/train.py:578-582):
//   Qwen3 causal GQA, hd 128  — HF:models/qwen3/modeling_qwen3.py:185-208 (scores*hd^-0.5, fp32 softmax)
//   ESM bidirectional, hd 64  — HF:models/esm/modeling_esm.py:292-317 (q pre-scaled, scale 1, key-padding mask)
// mask = causal (optional) AND key index in [kv_lo[b], kv_hi[b]) (padding, right- or left-padded).
//
// Orientation (guide §3 "An accumulator tile as the next MFMA's operand"): S^T = K·Q^T so that every lane owns ONE
// query column (lane&31) with its keys in registers -> the row max/sum are per-lane (+1 cross-half exchange), and P^T
// converted to bf16 is directly the B operand of O^T = V^T·P^T; V^T fragments come from the row-major V tile through
// ds_read_b64_tr_b16 (guide T10).  Q lives in registers for the whole kernel.
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "molly_hip.h"

namespace {

constexpr int BQ = 128;   // query rows per block (4 waves x 32)
constexpr int BKV = 64;   // keys per tile
constexpr float LOG2E = 1.4426950408889634f;
constexpr float RESCALE_THR = 8.0f;   // log2 domain: O/l are rescaled only when the running max grows by more than this
                                      // (guide T13 "defer-max"); P then stays <= 2^8, far inside bf16/fp32 range

// raw v_exp_f32 (2^x): inputs here are <= RESCALE_THR or -inf; no denormal range handling needed
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

struct AttnArgs {
    const bf16_t* Q; const bf16_t* K; const bf16_t* V; bf16_t* O; float* LSE;
    const int* kv_lo; const int* kv_hi;      // per-batch valid key range (nullable = [0,T))
    int T, nh, nkv, ldq, ldk, ldv, ldo;
    float scale_log2;                        // softmax scale * log2(e)
    int causal;
    int nblk, order_set;                     // block order (block_item): query blocks per head, (batch, kv head) pairs walked together
    int prio;                                // wave_priority(): 0 none, n: the odd wave slot of every SIMD runs at priority n
    bf16_t* OT; int ldot;                    // attn_fwd_kernel: O a second time, transposed [nh * hd][ldot >= B * T] (nullable) — the o-projection's weight-gradient operand
};

// One LDS image serves row reads (ds_read_b128: tile row on the lane) AND transposed reads (ds_read_b64_tr_b16: tile
// row = contraction index) — guide T10.  The image is SUB-TILED: blocks of 4 rows x 16 columns (128 B, row pitch 32 B),
// block (rowblk, colblk) stored at block index  rowblk*NCB + (colblk ^ (rowblk&1)),  NCB = HD/16, and inside a block the
// two 16-byte halves of a row are swapped when (rowblk>>1)&1.  Consequences:
//  * a transposed read (4 rows x 16 cols per 16-lane group, two adjacent column blocks per half-wave) touches 256
//    CONTIGUOUS bytes -> conflict-free; a b128 row read of 16 rows distinct mod 16 hits 16 distinct 16-B slots ->
//    conflict-free (the two swaps spread the four row-blocks of a lane group over the bank row);
//  * every address is  lane_base + compile-time constant  (the swaps only involve lane bits), so the unrolled reads
//    share a couple of base VGPRs and use DS immediate offsets instead of one hoisted address register each.
template <int HD> __device__ __forceinline__ int img_off(int row, int col) {      // element offset of (row, col..col+3)
    constexpr int NCB = HD / 16;
    const int rowblk = row >> 2, q = row & 3, b0 = rowblk & 1, b1 = (rowblk >> 1) & 1;
    const int colblk = col >> 4, half = (col >> 3) & 1;
    return (rowblk * NCB + (colblk ^ b0)) * 64 + (q * 2 + (half ^ b1)) * 8 + (col & 7);
}

// stage a [64 rows][HD] tile with global_load_lds; LDS image is lane-linear, so each lane decodes which (row, col chunk)
// its 16-byte destination slot holds and fetches that from global memory.
template <int HD, bool IS_V, int ROWS = BKV>
__device__ __forceinline__ void stage_kv(const bf16_t* __restrict__ g, int ld, int key0, int T, bf16_t* lds, int wave,
                                         int lane) {
    constexpr int NCB = HD / 16;
    constexpr int NINST = ROWS * HD * 2 / 1024;  // 1 KiB per wave-instruction
    const char* base = reinterpret_cast<const char*>(g + (size_t)key0 * ld);
#pragma unroll
    for (int i = 0; i < NINST / 4; ++i) {
        const int inst = wave * (NINST / 4) + i;
        const int P = inst * 64 + lane;            // destination 16-byte slot
        const int blk = P >> 3, cin = P & 7;
        const int rowblk = blk / NCB, cbs = blk % NCB;
        const int b0 = rowblk & 1, b1 = (rowblk >> 1) & 1;
        const int row = rowblk * 4 + (cin >> 1);
        const int col = ((cbs ^ b0) << 4) + (((cin & 1) ^ b1) << 3);
        // wave-uniform 64-bit base (tile row 0: scalar registers) + one 32-bit per-lane byte offset; rows past the end of
        // the sequence re-read the last valid row (masked by index later).  key0 < T at every call site.
        const int last = max(T - 1 - key0, 0);
        const int rel = row < last ? row : last;
        const unsigned off = ((unsigned)rel * (unsigned)ld + (unsigned)col) * 2u;
        // The LDS-DMA is issued from inline asm so that hipcc does not know an LDS write is in flight: with the builtin it
        // put `s_waitcnt vmcnt(0)` in front of the first ds_read_b64_tr_b16 of every half tile (the intrinsic read carries no
        // alias information, so it waits for every pending LDS-DMA) — i.e. the prefetch of tile t+1 was drained in the
        // middle of tile t and never overlapped the P·V work.  The wait is now ours: dma_wait() in front of the barrier that
        // ends the tile (guide §5.7 'Inline asm', §5 'Three .s-level traps').  hipcc pads hazards only between ITS instructions;
        // the one that could bite here (an SGPR base fresh from a VALU write, 5 wait states: §5.7 item 2) is ruled out on the
        // generated code by tools/check_asm_dma_hazards.py (tests/test_abi.py) instead of an `s_nop 4` per piece, which cost
        // 2-6 % in the GEMM.
        // Tried on top and dropped (same-box A/B, round 2): interior tiles as ONE 64-key step (both S^T halves, one max / rescale
        // decision, 32 exponentials, 16 P.V MFMAs; 252 VGPRs): 682-612 vs 692-652 TF/s — no gain; the cross-half maximum through
        // v_permlane32_swap: hipcc folded max(r[0], r[1]) of the swap of a value with itself to r[0] (the LOWER half's maximum
        // for both halves: correct O, but -inf LSE for rows whose only live keys sit in the upper half) — ds_bpermute stays.
        // An 8-wave ping-pong form of this kernel (one 512-thread workgroup per CU = 256 query rows, 3-stage K/V ring, the two
        // waves of a SIMD alternating a 16-MFMA segment [P.V of half h-1 + S^T of half h] with a load + softmax segment behind
        // workgroup barriers, group 1 one segment behind group 0; 208 VGPRs; passed every attention test): 510-520 TF/s against
        // 734-747 for this kernel in the same process — the softmax segment (24 LDS reads, ~86 vector instructions, 16
        // transcendentals, DMA issue) is longer than the 512-cycle matrix segment it is paired with, four 8-wave barriers per
        // tile pay for the slowest wave, and one workgroup per CU has nobody to cover its prologue and diagonal tail.  Removed.
        // A PERSISTENT form (512 resident workgroups walking a static snake-ordered item list; the next item's first K/V tile and Q
        // fragments in flight during the current item's last tile and epilogue, O through a swizzled 32 KB slab in the consumed
        // stage; passed every test): 213 us against 173.5 for this kernel in the same process (ESM shape 34.7 against 24.9) — the
        // second Q fragment set takes the kernel to 256 registers with spills, and a static list loses what the hardware dispatcher
        // gives for free (a finished workgroup's slot is refilled at once, whatever the item lengths).  Removed.
        const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(lds + inst * 512));
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                     :: "v"(off), "s"(base), "s"(lds_addr) : "memory", "m0");
    }
}

// ---- the same staging with the address arithmetic hoisted out of the tile loop (round 4).  Stamped (tools/r04/attn_stamp.py,
// profiles/r04_logs/attn_stamp.log), a wave of the forward spent ~470 cycles per 32-key half-step issuing the next tile's LDS-DMA:
// per piece a clamp, a 32-bit multiply (quarter rate), an add-shift and the address-space checks of the generic -> LDS pointer
// cast — against 512 cycles of MFMA.  For a tile that lies entirely inside the sequence nothing needs clamping, and the slot a
// lane fills differs between a wave's pieces only by a row step and two XOR-ed column bits:
//   piece i of wave w = instruction w * NP + i:  row = row0 + RSTEP * i,  col = col0 ^ cx(i)      (row0 / col0: stage_lane_const)
// so a piece costs one XOR, one three-operand add with a scalar, the m0 move and the load.
template <int HD> struct StageConst { unsigned rowb, colb; };          // per lane: row0 * ld * 2 bytes, col0 * 2 bytes
template <int HD, int ROWS = BKV>
__device__ __forceinline__ void stage_lane_const(int ld, int wave, int lane, unsigned& rowb, unsigned& colb) {
    constexpr int NCB = HD / 16, NP = ROWS * HD * 2 / 1024 / 4;
    const int P = wave * NP * 64 + lane;                  // piece 0 of this wave
    const int blk = P >> 3, cin = P & 7;
    const int rowblk = blk / NCB, cbs = blk % NCB;
    const int b0 = rowblk & 1, b1 = (rowblk >> 1) & 1;
    rowb = (unsigned)(rowblk * 4 + (cin >> 1)) * (unsigned)ld * 2u;
    colb = (unsigned)((((cbs ^ b0) << 4) + (((cin & 1) ^ b1) << 3)) * 2);
}
// base = tile row 0 of the (batch, head) slice (wave-uniform), lds_addr = LDS byte address of the tile image (wave-uniform, 32 bit)
template <int HD, int ROWS = BKV>
__device__ __forceinline__ void stage_tile_fast(const char* base, int ld, unsigned rowb, unsigned colb, unsigned lds_addr, int wave) {
    static_assert(ROWS == BKV || (ROWS == 32 && HD == 128), "a 32-row tile: hd 128 only (rowblk = 2 w + i: b1 = w & 1 is in colb, b0 = i & 1)");
    constexpr int NP = ROWS * HD * 2 / 1024 / 4;          // pieces per wave: 4 (hd 128) / 2 (hd 64; hd 128 at 32 rows)
    constexpr int RSTEP = HD == 128 ? 4 : 8;              // tile rows between two pieces of a wave
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        // hd 128: rowblk = instruction -> b0 = i & 1, b1 = (i >> 1) & 1 (w * NP is a multiple of 4); hd 64: rowblk = 2 * instruction +
        // (lane >> 5) -> b0 is a lane bit (in colb already), b1 = i & 1
        const unsigned cx = HD == 128 ? (unsigned)((((i & 1) << 4) ^ (((i >> 1) & 1) << 3)) * 2) : (unsigned)(((i & 1) << 3) * 2);
        const unsigned off = rowb + (colb ^ cx) + (unsigned)(i * RSTEP * 2) * (unsigned)ld;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_addr + (unsigned)((wave * NP + i) * 1024));
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(dst) : "memory", "m0");
    }
}

// "this register has arrived": an empty asm that READS v makes hipcc place the wait for v's pending global load HERE.  Used on
// the per-kernel operand fragments (Q, dO, K, V rows, row statistics) right after their loads: left to itself hipcc waits for
// them lazily inside the tile loop with counted `vmcnt(N)` that cannot see the asm-issued LDS-DMA — and those waits would then
// drain the DMA pipeline every tile (guide §5 'Three .s-level traps' (b)).
template <class T> __device__ __forceinline__ void arrived(const T& v) { asm volatile("" ::"v"(v)); }

// every LDS-DMA piece this wave issued has landed (the barrier behind it publishes the tile to the other waves)
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// row-read fragment: tile[row][16*s + 8*h .. +7]   (row = 32*sub + (lane&31); s is a compile-time constant at call sites)
template <int HD>
__device__ __forceinline__ bf16x8 row_frag(const bf16_t* tile, int row, int s, int h) {
    constexpr int NCB = HD / 16;
    const int rowblk = row >> 2, q = row & 3, b0 = rowblk & 1, b1 = (rowblk >> 1) & 1;
    // colblk = s, half = h:  (s ^ b0) written as (s & ~1) + ((s & 1) ^ b0) so that only the parity term is per-lane
    const int off = (rowblk * NCB + ((s & 1) ^ b0)) * 64 + (q * 2 + (h ^ b1)) * 8 + (s & ~1) * 64;
    return *reinterpret_cast<const bf16x8*>(tile + off);
}

// A operand (row index = column d of the tile, contraction = tile row) of a 32x32x16 MFMA via ds_read_b64_tr_b16:
// returns tile[row0 + 16*sp + perm(j,h)][32*dt + (lane&31)], j = 0..7, perm = the accumulator-as-operand k order
// (16 sp + 8 (j>>2) + 4 h + (j&3)).  row0 must be a multiple of 32.
template <int HD>
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* tile, int row0, int sp, int dt, int lane) {
    constexpr int NCB = HD / 16;
    const int h = lane >> 5, g1 = (lane >> 4) & 1, gi = lane & 15, gq = gi >> 2, gp = gi & 3;
    // rows 16sp + 4h + gq (+8): rowblk = row0/4 + 4sp + h (+2) -> b0 = h, b1 = 0 (+8: 1); colblk = 2dt + g1, half = gp>>1
    const int base_a = (h * NCB + (g1 ^ h)) * 64 + (gq * 2 + (gp >> 1)) * 8 + (gp & 1) * 4;
    const int base_b = ((h + 2) * NCB + (g1 ^ h)) * 64 + (gq * 2 + ((gp >> 1) ^ 1)) * 8 + (gp & 1) * 4;
    const int cst = ((row0 >> 2) + 4 * sp) * NCB * 64 + dt * 128;
    const bf16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(tile + base_a + cst));
    const bf16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)(tile + base_b + cst));
    return bf16x8{va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
}

__device__ __forceinline__ bf16x8 acc_to_frag(const f32x16& a, int base) {
    u32x4 w;
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = pack_bf2(a[base + 2 * j], a[base + 2 * j + 1]);
    return __builtin_bit_cast(bf16x8, w);
}

// Output rows of the attention kernels.  The accumulator layout gives a lane ONE row (query or key: r = lane & 31) with 4-column
// pieces of it spread over its registers (columns 32 d + 8 g4 + 4 h .. +3): stored directly that is 32 rows x 16 bytes per
// wave-instruction — 2,048 sixteen-byte requests per 32 KB block output.  The K/V (Q/dO) stages are dead once every wave has
// passed the loop's last barrier, so the wave parks its 32 rows in a private LDS slab (row pitch HD*2 + 16 bytes) and stores them
// back as whole rows: 16 bytes per lane, 4 full 256-byte rows per instruction (forward 189 -> 176 us at B8 T2048 H16 D128, same box).
// `mul` is per lane (the row's 1/l in the forward, the softmax scale in the backward); rows >= rows_valid are not stored.
// -DMOLLY_ATTN_ROWS_VIA_LDS=0 restores the direct stores (A/B).
// -DMOLLY_ATTN_PHASE_PRIO=1 (A/B build): the matrix segments of the forward run at raised priority (the GEMM's per-segment flips)
#ifndef MOLLY_ATTN_PHASE_PRIO
#define MOLLY_ATTN_PHASE_PRIO 0
#endif
// -DMOLLY_ATTN_STAMP=1 (diagnostic build, tools/r04/attn_stamp.py): s_memtime laps around the segments of the forward's half-step,
// summed per wave into g_attn_stamp[block][wave][8]; molly_exp_attn_stamps copies them out
#ifndef MOLLY_ATTN_STAMP
#define MOLLY_ATTN_STAMP 0
#endif
#if MOLLY_ATTN_STAMP
__device__ unsigned long long g_attn_stamp[32768 * 4 * 8];
#define ASTAMP(x) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(x) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define ALAP(i) do { ASTAMP(tq1_); ts_[i] += tq1_ - tq0_; tq0_ = tq1_; } while (0)
#else
#define ALAP(i) do { } while (0)
#endif
#ifndef MOLLY_ATTN_ROWS_VIA_LDS
#define MOLLY_ATTN_ROWS_VIA_LDS 1
#endif
// timing-only builds of attn_fwd_kernel (wrong results; tools/r05/attn_issue_diag.sh): 1 no row sum, 2 a multiply in place of every exp2,
// 4 neither fma nor exp2, 8 no LDS-DMA behind the first two tiles, 16 no tile barrier
#ifndef MOLLY_ATTN_DIAG
#define MOLLY_ATTN_DIAG 0
#endif
template <int HD, int ND>
__device__ __forceinline__ void store_rows(bf16_t* slab, const f32x16 (&acc)[ND], float mul, bf16_t* dst, size_t ld, int rows_valid,
                                           int lane) {
    const int r = lane & 31, h = lane >> 5;
#if MOLLY_ATTN_ROWS_VIA_LDS
    constexpr int PITCH = HD + 8;                                       // elements
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
            *reinterpret_cast<u32x2*>(slab + r * PITCH + 32 * d + 8 * g4 + 4 * h) =
                u32x2{pack_bf2(acc[d][4 * g4] * mul, acc[d][4 * g4 + 1] * mul), pack_bf2(acc[d][4 * g4 + 2] * mul, acc[d][4 * g4 + 3] * mul)};
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // the wave's own writes have landed (DS ops are in order)
    __builtin_amdgcn_sched_barrier(0);
    constexpr int LPR = HD / 8;                                          // lanes per row (16 bytes each)
    constexpr int RPI = 64 / LPR;                                        // rows per instruction
    const int lr = lane / LPR, lc = (lane % LPR) * 8;
#pragma unroll
    for (int it = 0; it < 32 / RPI; ++it) {
        const int row = it * RPI + lr;
        const u32x4 v = *reinterpret_cast<const u32x4*>(slab + row * PITCH + lc);
        if (row < rows_valid) *reinterpret_cast<u32x4*>(dst + (size_t)row * ld + lc) = v;
    }
#else
    if (r < rows_valid) {
        bf16_t* op = dst + (size_t)r * ld;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *reinterpret_cast<u32x2*>(op + 32 * d + 8 * g4 + 4 * h) =
                    u32x2{pack_bf2(acc[d][4 * g4] * mul, acc[d][4 * g4 + 1] * mul), pack_bf2(acc[d][4 * g4 + 2] * mul, acc[d][4 * g4 + 3] * mul)};
    }
#endif
}

// The rows store_rows has just parked in the wave's slab, a second time TRANSPOSED: dstT[col][0 .. 31] for the slab's HD columns (the
// caller points dstT at column 0 / the wave's first row).  The slab is read back through ds_read_b64_tr_b16 the way transpose64_kernel
// reads its tile (elementwise.hip): a 16-lane group takes 4 rows x 16 columns, a lane ends up with 8 consecutive rows of one column —
// 16 bytes of an output row; four groups = 64 contiguous bytes of each of 16 output rows per instruction.  Every one of the 32 rows
// must be valid (the host checks T % 128 == 0).  Replaces one transpose launch per layer (the o-projection's weight gradient reads
// attn^T): round 5.
template <int HD>
__device__ __forceinline__ void store_rows_t(const bf16_t* slab, bf16_t* dstT, size_t ldT, int lane) {
    constexpr int PITCH = HD + 8;
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
    const int rr = g * 8;
#pragma unroll
    for (int oc = 0; oc < HD; oc += 16) {
        const bf16_t* pa = slab + (rr + q) * PITCH + oc + 4 * pp;
        const bf16_t* pb = slab + (rr + 4 + q) * PITCH + oc + 4 * pp;
        const bf16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)pa);
        const bf16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)pb);
        const bf16x8 v = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
        *reinterpret_cast<bf16x8*>(dstT + (size_t)(oc + i) * ldT + rr) = v;
    }
}

// max over the two half-waves' copies of a row statistic without an LDS round trip: after the swap `a` holds the lower half's
// value in both halves and `b` the upper half's.  (asm with two registers: chained on one value the builtin is folded away by
// hipcc — the note in stage_kv; MOLLY_ATTN_XHALF_LDS=1 at compile time restores the ds_bpermute form for an A/B)
__device__ __forceinline__ float xhalf_max(float v) {
#if defined(MOLLY_ATTN_XHALF_LDS) && MOLLY_ATTN_XHALF_LDS
    return fmaxf(v, __shfl_xor(v, 32, 64));
#else
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return fmaxf(a, b);
#endif
}

// ---- which (batch, kv head, query head of the group, block) a workgroup works on.
// The 8 XCDs have an L2 each and the dispatcher deals workgroups over them round-robin (blocks b and b + 8 share one).  Round 3's
// grid (x = head + n_heads * batch fastest) therefore put the two query heads of a GQA group on DIFFERENT XCDs and gave every XCD
// 16 (batch, head) pairs' K/V at once — 16 MB against a 4 MB L2: the forward moved 760 MB per launch on the fabric side for 201 MB
// of Q + K + V + O (profiles/r03_pmc_hbm_traffic.csv).  Now: a 1-D grid; the workgroups of one XCD label take a contiguous chunk of
// the item order (the same chunking gemm.hip uses), and the item order is (batch, kv head) PAIR-major — every block of every
// query head of a pair before the next pair — so an XCD's 64 resident workgroups work on two or three pairs (1 MB of K/V each at
// T = 2048) and each K/V byte leaves HBM once.  Inside a pair: heaviest block first (causal: the last query block / the first key
// block), the heads of the group side by side.  `set` > 1 walks that many pairs interleaved (a knob for measurements);
// set = 0 is round 3's order (for A/B).  Speed only: any placement computes the same thing.
struct BlockItem { int b, kvh, g, blk; };
__device__ __forceinline__ BlockItem block_item(int bid, int nwg, int nkv, int group, int nblk, int set) {
    if (set <= 0) {                                    // round 3: x = (kv head * group + g) + n_heads * batch fastest, y = block slot
        const int nx = nwg / nblk, x = bid % nx, y = bid / nx;
        const int head = x % (nkv * group);
        return BlockItem{x / (nkv * group), head / group, head % group, y};
    }
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int bp = group * nblk;                       // workgroups per pair
    const int npairs = nwg / bp;
    const int s = v / (set * bp), pair0 = s * set;
    const int pc = min(set, npairs - pair0);
    const int w = v - s * set * bp;
    const int blk = w / (pc * group), rem = w - blk * (pc * group);
    const int pair = pair0 + rem / group;
    return BlockItem{pair / nkv, pair % nkv, rem % group, blk};
}

// Two workgroups share a CU: on every SIMD one wave of each, running the same loop — a matrix segment (S^T, then P.V: 16 MFMAs) and a
// vector segment (the softmax: ~70 vector instructions + 16 exponentials) of about the same length.  The matrix pipe is busy 43-48 %
// of the time (profiles/r03_pmc_sq_summary.csv): the two waves drift into the SAME segment and take turns on one pipe while the other
// idles.  A fixed priority difference between the two resident waves lets one of them finish its segment first whenever they
// collide, which puts them in opposite segments from then on.  The wave's slot on its SIMD (HW_ID bits 3:0) tells the two apart.
__device__ __forceinline__ void wave_priority(int prio) {
    if (prio <= 0) return;
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(hwid));
    if (hwid & 1) {
        if (prio == 1) __builtin_amdgcn_s_setprio(1);
        else if (prio == 2) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
    }
}

template <int HD>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);        // [2 stages][K tile | V tile]
    constexpr int TILE = BKV * HD;
    constexpr int NS = HD / 16;      // k-steps of the S product
    constexpr int ND = HD / 32;      // 32-wide d tiles of O^T

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // launched as the fall-back behind the fixed-reference kernel: nothing to do unless that kernel asked for it
    wave_priority(p.prio);
    // 1-D grid, XCD-aware item order (block_item); within a pair the heaviest causal blocks (largest query index) start first
    const BlockItem bi = block_item(blockIdx.x, gridDim.x, p.nkv, p.nh / p.nkv, p.nblk, p.order_set);
    const int qb = p.nblk - 1 - bi.blk;
    const int b = bi.b, kvh = bi.kvh, head = kvh * (p.nh / p.nkv) + bi.g;
    const int q0 = qb * BQ + wave * 32;                        // this wave's first query row
    const int T = p.T;
    const int lo = p.kv_lo ? p.kv_lo[b] : 0;
    const int hi = p.kv_hi ? p.kv_hi[b] : T;

    const bf16_t* Qb = p.Q + (size_t)b * T * p.ldq + head * HD;
    const bf16_t* Kb = p.K + (size_t)b * T * p.ldk + kvh * HD;
    const bf16_t* Vb = p.V + (size_t)b * T * p.ldv + kvh * HD;

    // Q fragments (B operand of S^T = K Q^T): lane (r,h) holds Q[q0+r][16s + 8h .. +7].  (Tried: the block's 128 query rows as two
    // 64-row tiles through the idle second K/V stage by LDS-DMA, fragments read back with row_frag — coalesced, but one more
    // barrier and 8 LDS reads per wave: 171.8 -> 174.1 us at hd 128, 24.9 -> 24.3 us on the ESM shape; not kept.)
    bf16x8 qf[NS];
    {
        int qrow = q0 + r;
        qrow = qrow < T ? qrow : T - 1;
        const bf16_t* qp = Qb + (size_t)qrow * p.ldq + 8 * h;
#pragma unroll
        for (int s = 0; s < NS; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
#pragma unroll
        for (int s = 0; s < NS; ++s) arrived(qf[s]);
    }

    f32x16 o[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // key tiles this BLOCK needs: [t_first, t_last]; this WAVE computes tiles up to tw_last (causal diagonal)
    const int blk_q_last = min(qb * BQ + BQ - 1, T - 1);
    const int kv_end = p.causal ? min(blk_q_last + 1, hi) : hi;       // exclusive
    const int t_first = lo / BKV;
    const int t_last = kv_end > lo ? (kv_end - 1) / BKV : t_first - 1;
    const int tw_last = p.causal ? min(t_last, (q0 + 31) / BKV) : t_last;
    const int qi = q0 + r;

#if MOLLY_ATTN_STAMP
    unsigned long long tq0_ = 0, tq1_ = 0, ts_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // S^T(32 keys x 32 queries) of one half tile
    auto qk_half = [&](const bf16_t* sK, int sub) {
        bf16x8 kf[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) kf[s] = row_frag<HD>(sK, 32 * sub + r, s, h);
        // every read in flight before the first MFMA: left alone hipcc recycles ONE fragment register and runs
        // read -> lgkmcnt(0) -> MFMA eight times in a row (an LDS round trip per 32-cycle MFMA)
        __builtin_amdgcn_sched_barrier(0);
        f32x16 sc;
#pragma unroll
        for (int e = 0; e < 16; ++e) sc[e] = 0.f;
        if (MOLLY_ATTN_PHASE_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < NS; ++s) sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s], qf[s], sc, 0, 0, 0);
        if (MOLLY_ATTN_PHASE_PRIO) __builtin_amdgcn_s_setprio(0);
        return sc;
    };
    auto v_half = [&](const bf16_t* sV, int sub, bf16x8 (&vf)[2][ND]) {
#pragma unroll
        for (int sp = 0; sp < 2; ++sp)
#pragma unroll
            for (int d = 0; d < ND; ++d) vf[sp][d] = tr_frag<HD>(sV, 32 * sub, sp, d, lane);
    };
    // online softmax of one half (keys kbase .. kbase+31) + O^T += V^T P^T
    auto softmax_pv = [&](f32x16& sc, int kbase, const bf16x8 (&vf)[2][ND]) {
        const bool need_mask = (p.causal && (kbase + 31 > q0)) || (kbase < lo) || (kbase + 32 > hi);
        // the scores stay RAW here: the softmax scale (> 0) commutes with max and is folded into the exponent's fma below
        float mx = -INFINITY;
        if (need_mask) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int key = kbase + (e & 3) + 8 * (e >> 2) + 4 * h;      // accumulator row -> key index
                const bool ok = key >= lo && key < hi && (!p.causal || key <= qi);
                sc[e] = ok ? sc[e] : -INFINITY;
                mx = fmaxf(mx, sc[e]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) mx = fmaxf(mx, sc[e]);
        }
        mx = xhalf_max(mx) * p.scale_log2;
        // deferred rescale (guide T13): keep the old reference max unless some row's max grew by > RESCALE_THR.  The
        // decision precedes every exponentiation of this half (textbook order), so nothing is ever half-scaled.
        if (!__all(mx - m_run <= RESCALE_THR)) {
            const float m_new = fmaxf(m_run, mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = fast_exp2(m_run - m_use);               // m_run = -inf -> 0
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
        }
        const float m_ref = (m_run == -INFINITY) ? 0.f : m_run;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (MOLLY_ATTN_DIAG == 4) sc[e] = sc[e] * m_ref;
            else if (MOLLY_ATTN_DIAG & 2) sc[e] = (sc[e] * p.scale_log2 - m_ref) * m_ref;
            else sc[e] = fast_exp2(sc[e] * p.scale_log2 - m_ref);       // one fma + one exp2 per score
        }
        // (tried, round 4: the fma and the row sum as packed fp32 — v_pk_fma_f32 / v_pk_add_f32, two scores per instruction, 15 % fewer
        // vector instructions in this loop: 164.0-164.9 us against 162.0-162.3 at B8 T2048 H16 D128, the backward unchanged too —
        // these loops are not bound by vector-instruction throughput: profiles/r04_logs/attn_pk.log)
        // row sum as a tree (a 16-deep dependent add chain would serialise on the add latency)
        float rs[4];
        if (MOLLY_ATTN_DIAG & 1) { l_run += sc[0]; } else {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) rs[g4] = (sc[4 * g4] + sc[4 * g4 + 1]) + (sc[4 * g4 + 2] + sc[4 * g4 + 3]);
        l_run += (rs[0] + rs[1]) + (rs[2] + rs[3]);                                                    // per-half-wave partial; merged at the end
        }
        const bf16x8 p0 = acc_to_frag(sc, 0), p1 = acc_to_frag(sc, 8);
#if MOLLY_ATTN_STAMP
        asm volatile("" ::"v"(p0), "v"(p1));
#endif
        ALAP(2);                                          // max, rescale decision, 16 exponentials, row sum, bf16 packing
        if (MOLLY_ATTN_PHASE_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int d = 0; d < ND; ++d) {
            o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0][d], p0, o[d], 0, 0, 0);
            o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1][d], p1, o[d], 0, 0, 0);
        }
        if (MOLLY_ATTN_PHASE_PRIO) __builtin_amdgcn_s_setprio(0);
    };

#if MOLLY_ATTN_STAMP
    ASTAMP(tq0_);
#endif
    // ---- main loop: double-buffered K/V tiles (64 keys), two 32-key halves per tile.  Two waves per SIMD (<= 256
    // registers) provide the matrix-pipe / VALU overlap; a source-level software pipeline (QK^T of the next half issued
    // before the softmax of the current one) was measured slower here — it needs > 256 registers at hd 128.
    // K/V tile `t` into stage `stg`: the hoisted form for tiles inside the sequence, the clamping one for a ragged last tile
    unsigned rbK, rbV, cbK, cbV;
    stage_lane_const<HD>(p.ldk, wave, lane, rbK, cbK);
    stage_lane_const<HD>(p.ldv, wave, lane, rbV, cbV);
    const unsigned smem_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(smem));
    auto stage_tile = [&](int t, int stg) {
        const int key0 = t * BKV;
        if (key0 + BKV <= T) {
            stage_tile_fast<HD>(reinterpret_cast<const char*>(Kb + (size_t)key0 * p.ldk), p.ldk, rbK, cbK, smem_lds + stg * 2 * TILE * 2, wave);
            stage_tile_fast<HD>(reinterpret_cast<const char*>(Vb + (size_t)key0 * p.ldv), p.ldv, rbV, cbV, smem_lds + (stg * 2 + 1) * TILE * 2, wave);
        } else {
            stage_kv<HD, false>(Kb, p.ldk, key0, T, smem + stg * 2 * TILE, wave, lane);
            stage_kv<HD, true>(Vb, p.ldv, key0, T, smem + stg * 2 * TILE + TILE, wave, lane);
        }
    };
    if (t_last >= t_first) stage_tile(t_first, 0);
    dma_wait();
    __syncthreads();
    int cur = 0;
    for (int t = t_first; t <= t_last; ++t) {
        const bf16_t* sK = smem + cur * 2 * TILE;
        const bf16_t* sV = sK + TILE;
        // (MOLLY_ATTN_DIAG & 8, timing only: no LDS-DMA behind the first two tiles — what does the staging's ISSUE cost?
        //  & 16: no tile barrier either)
        if (t + 1 <= t_last && !((MOLLY_ATTN_DIAG & 8) && t > t_first)) stage_tile(t + 1, cur ^ 1);
        const int k0 = t * BKV;
        if (t <= tw_last) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                if (p.causal && (k0 + 32 * sub > q0 + 31)) continue;      // half entirely above the diagonal (uniform)
                ALAP(0);                                  // loop overhead + LDS-DMA issue of the next tile (first half of a tile)
                f32x16 sc = qk_half(sK, sub);
                bf16x8 vf[2][ND];
                v_half(sV, sub, vf);
#if MOLLY_ATTN_STAMP
                { float probe = sc[0] + sc[15]; asm volatile("" ::"v"(probe)); ++ts_[7]; }   // the S^T chain has completed
#endif
                ALAP(1);                                  // K fragment reads + the 8 S^T MFMAs (+ V fragment reads issued)
                softmax_pv(sc, k0 + 32 * sub, vf);
                ALAP(3);                                  // rest of the softmax (after the lap inside) + P.V MFMA issue
            }
        }
        ALAP(0);
        if (!(MOLLY_ATTN_DIAG & 16)) {
        dma_wait();
        __syncthreads();
        }
        ALAP(4);                                          // wait for the next tile + the workgroup barrier
        cur ^= 1;
    }
    ALAP(0);

    // ---- epilogue: O[q][d] = o / l ; LSE2 = m + log2(l)
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    store_rows<HD, ND>(smem + wave * 32 * (HD + 8), o, inv, p.O + ((size_t)b * T + q0) * p.ldo + head * HD, p.ldo, T - q0, lane);
    if (MOLLY_ATTN_ROWS_VIA_LDS && p.OT)
        store_rows_t<HD>(smem + wave * 32 * (HD + 8), p.OT + (size_t)(head * HD) * p.ldot + (size_t)b * T + q0, (size_t)p.ldot, lane);
    if (qi < T && p.LSE && h == 0)
        p.LSE[((size_t)b * p.nh + head) * T + qi] = l_tot > 0.f ? m_run + log2f(l_tot) : -INFINITY;
#if MOLLY_ATTN_STAMP
    ALAP(5);                                              // epilogue
    if (lane == 0 && blockIdx.x < 32768) {
        unsigned long long* q = g_attn_stamp + ((size_t)blockIdx.x * 4 + wave) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) q[i] += ts_[i];
    }
#endif
}


// ================================================================================================
// Small head dims (8 <= hd <= 64, hd % 8 == 0, not 64): the mini encoders of the reference's plumbing config (ESM2-t6-8M:
// 320 hidden / 20 heads = 16).  Forward only, plain VALU: one thread per query row, K/V tiles of 64 keys staged in LDS as
// fp32 (every thread reads the same key row: LDS broadcast), online softmax in the exp2 domain.  Not a performance path.
// ================================================================================================
template <int HD>
__global__ __launch_bounds__(128) void attn_fwd_small_kernel(AttnArgs p) {
    __shared__ float sK[64][HD], sV[64][HD];
    const int head = blockIdx.x % p.nh, b = blockIdx.x / p.nh;
    const int kvh = head / (p.nh / p.nkv);
    const int T = p.T;
    const int lo = p.kv_lo ? p.kv_lo[b] : 0;
    const int hi = p.kv_hi ? p.kv_hi[b] : T;
    const int q = blockIdx.y * 128 + threadIdx.x;
    const bool q_ok = q < T;
    float qf[HD], acc[HD];
    {
        const bf16_t* qp = p.Q + ((size_t)b * T + (q_ok ? q : T - 1)) * p.ldq + head * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) { qf[d] = bf2f(qp[d]) * p.scale_log2; acc[d] = 0.f; }
    }
    float m = -INFINITY, l = 0.f;
    const int q_last = min(blockIdx.y * 128 + 127, T - 1);
    const int k_end = p.causal ? min(hi, q_last + 1) : hi;
    for (int k0 = lo - (lo % 64); k0 < k_end; k0 += 64) {
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * HD; i += 128) {
            const int r = i / HD, d = i % HD;
            const int key = min(k0 + r, T - 1);
            sK[r][d] = bf2f(p.K[((size_t)b * T + key) * p.ldk + kvh * HD + d]);
            sV[r][d] = bf2f(p.V[((size_t)b * T + key) * p.ldv + kvh * HD + d]);
        }
        __syncthreads();
        for (int r = 0; r < 64; ++r) {
            const int key = k0 + r;
            if (key < lo || key >= hi || (p.causal && key > q)) continue;      // per-thread only through `q`: cheap divergence
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) s += qf[d] * sK[r][d];
            const float mn = fmaxf(m, s);
            const float c = fast_exp2(m - mn), pe = fast_exp2(s - mn);
            l = l * c + pe;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] = acc[d] * c + pe * sV[r][d];
            m = mn;
        }
    }
    if (!q_ok) return;
    bf16_t* op = p.O + ((size_t)b * T + q) * p.ldo + head * HD;
    const float inv = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) op[d] = f2bf(acc[d] * inv);
    if (p.LSE) p.LSE[((size_t)b * p.nh + head) * T + q] = l > 0.f ? m + __log2f(l) : -INFINITY;
}

// ================================================================================================
// backward.  Two kernels, both recomputing P from Q,K and the forward's LSE (no atomics, bitwise reproducible):
//   dq kernel : grid like the forward (query blocks); per key tile  S^T = K Q^T, dP^T = V dO^T,
//               dS^T = P^T ⊙ (dP^T − δ),  dQ^T += K^T dS^T          (K tile read by rows AND transposed)
//   dkv kernel: grid over key blocks; per query tile  S = Q K^T, dP = dO V^T (key on the lane),
//               dV^T += dO^T P,  dK^T += Q^T dS                      (Q/dO tiles read by rows AND transposed)
// δ[q] = rowsum(dO ⊙ O) comes from a small preprocess kernel.  7 MFMA products instead of flash-bwd's 5, traded
// for no cross-workgroup dQ reduction (guide Appendix B "Attention backward": dQ atomics are rate-limited).
// ================================================================================================
// ---- q/k-norm + rotary BACKWARD inside the attention backward's row epilogues (round 6).  HF applies, per head row, RMSNorm with a gain and
// then the rotary embedding (HF:models/qwen3/modeling_qwen3.py:225-236: q_norm / k_norm, apply_rotary_pos_emb); its backward was a kernel of its
// own between the attention backward and the q | k | v dgrad (norm_rope_bwd_kernel: 154 us per layer at 32 k tokens, instruction-bound at
// 3.9 TB/s: one more read of dq | dk and of the saved pre-norm rows, one more write).  The dQ and dK kernels leave their rows through a
// wave-private LDS slab as WHOLE head rows (store_rows), so the same arithmetic runs there: a lane holds 8 consecutive elements of a row
// (16 lanes per 256-byte row), its rotary partners sit 8 lanes away in the same DPP row, the two row sums are DPP adds — and the rows go
// straight into d(q | k | v) as the dgrad GEMM reads them.  Gain gradients: per-lane sums over the wave's rows, lanes and waves combined
// through LDS, one row of HD floats per workgroup into `dw_part` (summed by the batched column reduction at the end of the backward).
struct RopeBwdFuse {
    const bf16_t* X; int ldx;        // the saved PRE-norm projection rows, pointing at head 0 of q (or of k); nullptr = off
    const bf16_t* w;                 // the norm's gain [HD]
    const float* cos; const float* sin;   // [positions][HD / 2] (position = token index within the sample)
    bf16_t* dX; int lddx;            // output: d(projection), pointing at head 0 of q (or of k)
    float* dw_part;                  // [gridDim.x][HD]
    float eps;
};

// the wave's 32 rows of acc * mul (rows row0 .. of the (batch x T) token axis, positions pos0 .., head columns head_col ..): rounded to bf16 as the
// unfused path stores dq / dk, then rotary^T, norm backward against the saved row, stored; dw[e] += this lane's share of the gain gradient
template <int HD, int ND>
__device__ __forceinline__ void store_rows_rope_bwd(bf16_t* slab, const f32x16 (&acc)[ND], float mul, const RopeBwdFuse& f, size_t row0, int pos0,
                                                    int head_col, int rows_valid, int lane, float (&dw)[8]) {
    static_assert(HD == 128, "the fused rotary backward is built for head dim 128");
    const int r = lane & 31, h = lane >> 5;
    constexpr int PITCH = HD + 8;
    // the saved pre-norm rows come from HBM (2 us away under load): all eight 16-byte pieces this lane needs are requested NOW, before the slab is
    // written — one round trip per workgroup instead of one per pass (the first version, with the load inside the loop, cost the dQ kernel 66 us
    // and the dK kernel 31 us per layer of the 154 the separate kernel took: profiles/r06_logs/fuse1_kernel_stats.csv)
    u32x4 xall[8];
    {
        const int lr_ = lane >> 4, c_ = lane & 15;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 4 + lr_;
            xall[it] = *reinterpret_cast<const u32x4*>(f.X + (row0 + (row < rows_valid ? row : 0)) * (size_t)f.ldx + head_col + 8 * c_);
        }
    }
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
            *reinterpret_cast<u32x2*>(slab + r * PITCH + 32 * d + 8 * g4 + 4 * h) =
                u32x2{pack_bf2(acc[d][4 * g4] * mul, acc[d][4 * g4 + 1] * mul), pack_bf2(acc[d][4 * g4 + 2] * mul, acc[d][4 * g4 + 3] * mul)};
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    const int lr = lane >> 4, c = lane & 15;                    // row of 4 per pass, 16-byte chunk of the 256-byte row
    const bool first = c < 8;                                   // elements 8c .. 8c+7 lie in the first half: their partners are 64 further
    const int i0 = (c & 7) * 8;                                 // index into the cos / sin row
    float wv[8];
    {
        const u32x4 wq = *reinterpret_cast<const u32x4*>(f.w + 8 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { wv[2 * e] = bflo(wq[e]); wv[2 * e + 1] = bfhi(wq[e]); }
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int row = it * 4 + lr;
        const bool ok = row < rows_valid;
        const int rowc = ok ? row : 0;
        const u32x4 gv = *reinterpret_cast<const u32x4*>(slab + row * PITCH + 8 * c);
        const u32x4 xv = xall[it];
        const float* cp = f.cos + (size_t)(pos0 + rowc) * (HD / 2) + i0;
        const float* sp = f.sin + (size_t)(pos0 + rowc) * (HD / 2) + i0;
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(cp), c1 = *reinterpret_cast<const f32x4*>(cp + 4);
        const f32x4 s0 = *reinterpret_cast<const f32x4*>(sp), s1 = *reinterpret_cast<const f32x4*>(sp + 4);
        float g[8], pg[8], x[8], t[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned pv = (unsigned)__builtin_amdgcn_update_dpp(0, (int)gv[e], 0x128, 0xf, 0xf, true);     // row_ror:8: the lane 8 away in the row
            g[2 * e] = bflo(gv[e]); g[2 * e + 1] = bfhi(gv[e]);
            pg[2 * e] = bflo(pv); pg[2 * e + 1] = bfhi(pv);
            x[2 * e] = bflo(xv[e]); x[2 * e + 1] = bfhi(xv[e]);
        }
        float ss = 0.f, dot = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float cs = e < 4 ? c0[e & 3] : c1[e & 3], sn = e < 4 ? s0[e & 3] : s1[e & 3];
            // rotary^T (the forward: y1 = x1 c - x2 s, y2 = x2 c + x1 s): first half g1 c + g2 s, second half g2 c - g1 s
            t[e] = first ? g[e] * cs + pg[e] * sn : g[e] * cs - pg[e] * sn;
            ss += x[e] * x[e];
            dot += t[e] * wv[e] * x[e];
        }
        ss = dpp_row_sum<16>(ss);
        dot = dpp_row_sum<16>(dot);
        const float rstd = rsqrtf(ss / (float)HD + f.eps);
        const float coef = dot * rstd * rstd * rstd / (float)HD;
        float dx[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            dx[e] = t[e] * wv[e] * rstd - x[e] * coef;
            if (ok) dw[e] += t[e] * x[e] * rstd;
        }
        if (ok)
            *reinterpret_cast<u32x4*>(f.dX + (row0 + row) * (size_t)f.lddx + head_col + 8 * c) =
                u32x4{pack_bf2(dx[0], dx[1]), pack_bf2(dx[2], dx[3]), pack_bf2(dx[4], dx[5]), pack_bf2(dx[6], dx[7])};
    }
}

// the workgroup's gain-gradient row: the four row-lanes of a wave (lanes l, l + 16, l + 32, l + 48 hold the same columns), then the four waves
template <int HD>
__device__ __forceinline__ void rope_bwd_dw_flush(float (&dw)[8], float* red /* LDS, 4 x HD floats, free */, float* dst, int wave, int lane, int tid) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        dw[e] += __shfl_xor(dw[e], 16, 64);
        dw[e] += __shfl_xor(dw[e], 32, 64);
    }
    __syncthreads();                                            // every wave is done with its slab (red aliases the staging memory)
    if (lane < 16) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[wave * HD + 8 * lane + e] = dw[e];
    }
    __syncthreads();
    if (tid < HD) dst[tid] = red[tid] + red[HD + tid] + red[2 * HD + tid] + red[3 * HD + tid];
}

struct AttnBwdArgs {
    const bf16_t* Q; const bf16_t* K; const bf16_t* V; const bf16_t* dO; const bf16_t* O;
    const float* LSE; float* delta;            // delta[b, head, q] = MINUS sum_d dO[q,d] * O[q,d]: written by the dQ kernel, read by dK's as the dP chain's initial value
    bf16_t* dQ; bf16_t* dK; bf16_t* dV;
    const int* kv_lo; const int* kv_hi;
    int T, nh, nkv, ldq, ldk, ldv, ldo, lddq, lddk, lddv;
    float scale, scale_log2;
    int causal;
    float* part;               // head-split dK/dV pass: per-(block, query head) accumulator images, register order (see SPLIT)
    int nblk, order_set;       // block order (block_item): query / key blocks per head, (batch, kv head) pairs walked together
    int prio;                  // wave_priority()
    RopeBwdFuse fq, fk;        // round 6: the q/k-norm + rotary backward inside the dQ / dK row epilogues (X == nullptr: off)
};

template <int HD>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);        // [2 stages][K tile | V tile]
    constexpr int TILE = BKV * HD;
    constexpr int NS = HD / 16, ND = HD / 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    wave_priority(p.prio);
    const BlockItem bi = block_item(blockIdx.x, gridDim.x, p.nkv, p.nh / p.nkv, p.nblk, p.order_set);   // see attn_fwd_kernel
    const int qb = p.nblk - 1 - bi.blk;                        // heaviest first
    const int b = bi.b, kvh = bi.kvh, head = kvh * (p.nh / p.nkv) + bi.g;
    const int q0 = qb * BQ + wave * 32;
    const int T = p.T;
    const int lo = p.kv_lo ? p.kv_lo[b] : 0;
    const int hi = p.kv_hi ? p.kv_hi[b] : T;
    const bf16_t* Kb = p.K + (size_t)b * T * p.ldk + kvh * HD;
    const bf16_t* Vb = p.V + (size_t)b * T * p.ldv + kvh * HD;

    const int qi = q0 + r;
    const int qrow = qi < T ? qi : T - 1;
    bf16x8 qf[NS], dof[NS];
    float dlt = 0.f;
    {
        const bf16_t* qp = p.Q + ((size_t)b * T + qrow) * p.ldq + head * HD + 8 * h;
        const bf16_t* dp = p.dO + ((size_t)b * T + qrow) * p.ldo + head * HD + 8 * h;
        const bf16_t* op = p.O + ((size_t)b * T + qrow) * p.ldo + head * HD + 8 * h;
        u32x4 of[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
            dof[s] = *reinterpret_cast<const bf16x8*>(dp + 16 * s);
            of[s] = *reinterpret_cast<const u32x4*>(op + 16 * s);
        }
        // delta = rowsum(dO * O) of this lane's query row (round 1: a kernel of its own, a second pass over O and dO): the lane
        // holds half of the row's dO already; the other half-wave holds the rest
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const u32x4 dv = __builtin_bit_cast(u32x4, dof[s]);
#pragma unroll
            for (int e = 0; e < 4; ++e) dlt += bflo(of[s][e]) * bflo(dv[e]) + bfhi(of[s][e]) * bfhi(dv[e]);
        }
        dlt += __shfl_xor(dlt, 32, 64);
    }
    float lse = p.LSE[((size_t)b * p.nh + head) * T + qrow];
    lse = (lse == -INFINITY) ? 0.f : lse;
    if (h == 0 && qi < T) p.delta[((size_t)b * p.nh + head) * T + qi] = -dlt;      // for the dK pass: NEGATED — it is the initial value of the dP accumulators there
#pragma unroll
    for (int s = 0; s < NS; ++s) { arrived(qf[s]); arrived(dof[s]); }
    arrived(lse); arrived(dlt);

    f32x16 dq[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) dq[d][e] = 0.f;

    const int blk_q_last = min(qb * BQ + BQ - 1, T - 1);
    const int kv_end = p.causal ? min(blk_q_last + 1, hi) : hi;
    const int t_first = lo / BKV;
    const int t_last = kv_end > lo ? (kv_end - 1) / BKV : t_first - 1;
    unsigned rbK, rbV, cbK, cbV;                               // see attn_fwd_kernel
    stage_lane_const<HD>(p.ldk, wave, lane, rbK, cbK);
    stage_lane_const<HD>(p.ldv, wave, lane, rbV, cbV);
    const unsigned smem_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(smem));
    auto stage_tile = [&](int t, int stg) {
        const int key0 = t * BKV;
        if (key0 + BKV <= T) {
            stage_tile_fast<HD>(reinterpret_cast<const char*>(Kb + (size_t)key0 * p.ldk), p.ldk, rbK, cbK, smem_lds + stg * 2 * TILE * 2, wave);
            stage_tile_fast<HD>(reinterpret_cast<const char*>(Vb + (size_t)key0 * p.ldv), p.ldv, rbV, cbV, smem_lds + (stg * 2 + 1) * TILE * 2, wave);
        } else {
            stage_kv<HD, false>(Kb, p.ldk, key0, T, smem + stg * 2 * TILE, wave, lane);
            stage_kv<HD, true>(Vb, p.ldv, key0, T, smem + stg * 2 * TILE + TILE, wave, lane);
        }
    };
    if (t_last >= t_first) stage_tile(t_first, 0);
    dma_wait();
    __syncthreads();
    int cur = 0;
    for (int t = t_first; t <= t_last; ++t) {
        const bf16_t* sK = smem + cur * 2 * TILE;
        const bf16_t* sV = sK + TILE;
        if (t + 1 <= t_last) stage_tile(t + 1, cur ^ 1);
        const int k0 = t * BKV;
        const bool skip = p.causal && (k0 > q0 + 31);
        if (!skip) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int kbase = k0 + 32 * sub;
                if (p.causal && kbase > q0 + 31) continue;           // half entirely above the diagonal (wave-uniform)
                // -delta as the initial value of the dP accumulators (guide, 'Attention backward': row constants as the initial accumulator): dP - delta
                // leaves the MFMA chain ready, sixteen subtractions per block gone
                f32x16 sT, dpT;
#pragma unroll
                for (int e = 0; e < 16; ++e) { sT[e] = 0.f; dpT[e] = -dlt; }
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    sT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(sK, 32 * sub + r, s, h), qf[s], sT, 0, 0, 0);
                    dpT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(sV, 32 * sub + r, s, h), dof[s], dpT, 0, 0, 0);
                }
                // masks are needed only where the half touches the causal diagonal or the [lo, hi) edges (wave-uniform test):
                // the interior halves — most of them — run 4 VALU per element (fma, exp2, sub, mul) instead of ~10
                const bool need_mask = (p.causal && kbase + 31 > q0) || kbase < lo || kbase + 32 > hi;
                if (need_mask) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int key = kbase + (e & 3) + 8 * (e >> 2) + 4 * h;
                        const bool ok = key >= lo && key < hi && (!p.causal || key <= qi);
                        const float pr = ok ? fast_exp2(sT[e] * p.scale_log2 - lse) : 0.f;
                        sT[e] = pr * dpT[e];                        // dS^T (unscaled)
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) sT[e] = fast_exp2(sT[e] * p.scale_log2 - lse) * dpT[e];
                }
                const bf16x8 f0 = acc_to_frag(sT, 0), f1 = acc_to_frag(sT, 8);
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HD>(sK, 32 * sub, 0, d, lane), f0, dq[d], 0, 0, 0);
                    dq[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HD>(sK, 32 * sub, 1, d, lane), f1, dq[d], 0, 0, 0);
                }
            }
        }
        dma_wait();
        __syncthreads();
        cur ^= 1;
    }
    if constexpr (HD == 128) {
        if (p.fq.X) {
            // q-norm + rotary backward on the rows as they leave (RopeBwdFuse above): d(q) goes straight into d(q | k | v)
            float dw[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            store_rows_rope_bwd<HD, ND>(smem + wave * 32 * (HD + 8), dq, p.scale, p.fq, (size_t)b * T + q0, q0, head * HD, T - q0, lane, dw);
            rope_bwd_dw_flush<HD>(dw, reinterpret_cast<float*>(smem_raw), p.fq.dw_part + (size_t)blockIdx.x * HD, wave, lane, tid);
            return;
        }
    }
    store_rows<HD, ND>(smem + wave * 32 * (HD + 8), dq, p.scale, p.dQ + ((size_t)b * T + q0) * p.lddq + head * HD, p.lddq, T - q0, lane);
}

// dK/dV: block = 128 keys (4 waves x 32 keys) of one (batch, kv head); loops over the group's query heads and query tiles.
// MODE 0: dK and dV in one pass (256 accumulator registers -> one wave per SIMD); MODE 1: dV only (S -> P -> dV);
// MODE 2: dK only (S, dP -> dS -> dK).  The two single-output passes recompute S (+25 % MFMAs in total) but fit two
// waves per SIMD, which is what the matrix-pipe / VALU overlap needs — measured faster than MODE 0 at hd 128.
// SPLIT: one block per QUERY head of the group instead of one per kv head.  At one sample per GPU (BASELINE configs 3 / 4:
// B = 1, 8 kv heads, T = 3072 / 4096) the grid over (kv head, key block) is 192-256 blocks for 512 slots, and the block of key
// block 0 walks group x T/64 = 192-256 query tiles alone: the launch is as long as that one block (346 us against 98 for the
// forward).  Split, there are group x as many blocks, each 1/group as long; a block leaves its accumulators, unscaled fp32 in
// register order (16 bytes per lane, coalesced), in `part`, and attn_dkv_reduce_kernel adds the group's images in head order
// (fixed: bitwise reproducible, no atomics) and writes the bf16 rows.
template <int HD, int MODE, bool SPLIT = false>
__global__ __launch_bounds__(256, MODE == 0 ? 1 : 2) void attn_bwd_dkv_kernel(AttnBwdArgs p) {
    constexpr bool DO_DV = MODE != 2, DO_DK = MODE != 1;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int TILE = BKV * HD;                               // 64 query rows per tile
    constexpr int NS = HD / 16, ND = HD / 32;
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);          // [2 stages][Q tile | dO tile]
    float* sStat = reinterpret_cast<float*>(smem_raw + 2 * 2 * TILE * sizeof(bf16_t));   // [2 stages][lse 64 | delta 64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    wave_priority(p.prio);
    // causal: key block 0 sees every query tile (heaviest) -> slot order = key-block order already is heaviest-first
    const int group = p.nh / p.nkv;
    const BlockItem bi = block_item(blockIdx.x, gridDim.x, p.nkv, SPLIT ? group : 1, p.nblk, p.order_set);
    const int kb = bi.blk, kvh = bi.kvh, b = bi.b, g0 = bi.g;
    const int bx = b * p.nkv + kvh;                              // the (batch, kv head) pair
    const int T = p.T;
    const int lo = p.kv_lo ? p.kv_lo[b] : 0;
    const int hi = p.kv_hi ? p.kv_hi[b] : T;
    const int key0 = kb * 128 + wave * 32;
    const int key = key0 + r;
    const int krow = key < T ? key : T - 1;
    const bool key_ok = key >= lo && key < hi;

    bf16x8 kf[NS], vf[NS];
    {
        const bf16_t* kp = p.K + ((size_t)b * T + krow) * p.ldk + kvh * HD + 8 * h;
        const bf16_t* vp = p.V + ((size_t)b * T + krow) * p.ldv + kvh * HD + 8 * h;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            kf[s] = *reinterpret_cast<const bf16x8*>(kp + 16 * s);
            if (DO_DK) vf[s] = *reinterpret_cast<const bf16x8*>(vp + 16 * s);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) { arrived(kf[s]); if (DO_DK) arrived(vf[s]); }
    }
    f32x16 dk[ND], dv[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dk[d][e] = 0.f; dv[d][e] = 0.f; }

    // query tiles this block needs: causal -> tiles whose last row >= first key of the block
    const int nqt = (T + BKV - 1) / BKV;
    const int qt_first = p.causal ? (kb * 128) / BKV : 0;
    const int n_tiles = nqt > qt_first ? nqt - qt_first : 0;
    const int total = SPLIT ? n_tiles : n_tiles * group;         // iteration it -> (head g, tile qt_first + j)

    unsigned rbQ, rbD, cbQ, cbD;
    stage_lane_const<HD>(p.ldq, wave, lane, rbQ, cbQ);
    stage_lane_const<HD>(p.ldo, wave, lane, rbD, cbD);
    const unsigned smem_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(smem));
    // iteration it = (head g of the group, query tile qt_first + j): both walked by counters (a division by n_tiles per staged
    // tile and another per computed one was ~100 scalar / vector instructions per iteration)
    int sg = 0, sj = 0;                                          // (g, j) of the NEXT tile to stage
    int cj = 0;                                                  // j of the tile being computed
    auto stage = [&](int /*it*/, int stg) {
        const int g = SPLIT ? g0 : sg, qt = qt_first + sj;
        if (++sj == n_tiles) { sj = 0; ++sg; }
        const int head = kvh * group + g;
        const bf16_t* Qb = p.Q + (size_t)b * T * p.ldq + head * HD;
        const bf16_t* Db = p.dO + (size_t)b * T * p.ldo + head * HD;
        bf16_t* sQ = smem + stg * 2 * TILE;
        if (qt * BKV + BKV <= T) {                                   // (the hoisted form: attn_fwd_kernel)
            stage_tile_fast<HD>(reinterpret_cast<const char*>(Qb + (size_t)qt * BKV * p.ldq), p.ldq, rbQ, cbQ, smem_lds + stg * 2 * TILE * 2, wave);
            stage_tile_fast<HD>(reinterpret_cast<const char*>(Db + (size_t)qt * BKV * p.ldo), p.ldo, rbD, cbD, smem_lds + (stg * 2 + 1) * TILE * 2, wave);
        } else {
            stage_kv<HD, false>(Qb, p.ldq, qt * BKV, T, sQ, wave, lane);
            stage_kv<HD, false>(Db, p.ldo, qt * BKV, T, sQ + TILE, wave, lane);
        }
        // row statistics through LDS-DMA as well (4 B/lane): an ordinary global_load here would make hipcc drain
        // vmcnt(0) — i.e. the whole tile prefetch — at its first use (guide §5 "Three .s-level traps" (b))
        if (wave < 2) {
            int q = qt * BKV + lane;
            q = q < T ? q : T - 1;
            const size_t idx = ((size_t)b * p.nh + head) * T + q;
            const float* src = wave == 0 ? p.LSE + idx : p.delta + idx;
            const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(sStat + stg * 128 + wave * 64));
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(src), "s"(lds_addr) : "memory", "m0");
        }
    };
    if (total > 0) stage(0, 0);
    dma_wait();
    __syncthreads();
    int cur = 0;
    for (int it = 0; it < total; ++it) {
        if (it + 1 < total) stage(it + 1, cur ^ 1);
        const int qt = qt_first + cj;
        if (++cj == n_tiles) cj = 0;
        const int qbase = qt * BKV;
        const bf16_t* sQ = smem + cur * 2 * TILE;
        const bf16_t* sD = sQ + TILE;
        const float* sL = sStat + cur * 128;
        const bool skip = p.causal && (qbase + BKV - 1 < key0);   // every query of the tile precedes this wave's keys
        if (!skip) {
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int qsub = qbase + 32 * sub;
                if (p.causal && qsub + 31 < key0) continue;       // these 32 queries all precede this wave's keys (uniform)
                f32x16 sA, dpA;                                   // [q rows (regs), key cols (lane)]
#pragma unroll
                for (int e = 0; e < 16; ++e) { sA[e] = 0.f; dpA[e] = 0.f; }
                if (DO_DK) {                                      // -delta (the dQ kernel stores it negated) of the 4 rows each accumulator group holds: the dP
#pragma unroll                                                    // chain's initial value, read straight into its registers
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const f32x4 nd = *reinterpret_cast<const f32x4*>(sL + 64 + 32 * sub + 8 * g4 + 4 * h);
                        dpA[4 * g4] = nd[0]; dpA[4 * g4 + 1] = nd[1]; dpA[4 * g4 + 2] = nd[2]; dpA[4 * g4 + 3] = nd[3];
                    }
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(sQ, 32 * sub + r, s, h), kf[s], sA, 0, 0, 0);
                    if (DO_DK)
                        dpA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(sD, 32 * sub + r, s, h), vf[s], dpA, 0, 0, 0);
                }
                // row statistics of the 4 consecutive query rows each accumulator group holds: one 16-byte LDS read each
                f32x4 lse4[4];
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) lse4[g4] = *reinterpret_cast<const f32x4*>(sL + 32 * sub + 8 * g4 + 4 * h);
                f32x16 pA;
                // masks only where the 32 x 32 block touches the causal diagonal, the [lo, hi) key edges or the end of the
                // sequence (wave-uniform test); interior blocks run the bare fma / exp2 / sub / mul
                const bool need_mask = (p.causal && qsub < key0 + 31) || key0 < lo || key0 + 32 > hi || qsub + 32 > T;
                if (need_mask) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int q = qsub + (e & 3) + 8 * (e >> 2) + 4 * h;
                        const bool ok = key_ok && q < T && (!p.causal || key <= q);
                        const float pr = ok ? fast_exp2(sA[e] * p.scale_log2 - lse4[e >> 2][e & 3]) : 0.f;
                        pA[e] = pr;
                        if (DO_DK) sA[e] = pr * dpA[e];                         // dS (unscaled)
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float pr = fast_exp2(sA[e] * p.scale_log2 - lse4[e >> 2][e & 3]);
                        pA[e] = pr;
                        if (DO_DK) sA[e] = pr * dpA[e];
                    }
                }
                if (DO_DV) {
                    const bf16x8 p0 = acc_to_frag(pA, 0), p1 = acc_to_frag(pA, 8);
#pragma unroll
                    for (int d = 0; d < ND; ++d) {
                        dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HD>(sD, 32 * sub, 0, d, lane), p0, dv[d], 0, 0, 0);
                        dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HD>(sD, 32 * sub, 1, d, lane), p1, dv[d], 0, 0, 0);
                    }
                }
                if (DO_DK) {
                    const bf16x8 d0 = acc_to_frag(sA, 0), d1 = acc_to_frag(sA, 8);
#pragma unroll
                    for (int d = 0; d < ND; ++d) {
                        dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HD>(sQ, 32 * sub, 0, d, lane), d0, dk[d], 0, 0, 0);
                        dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HD>(sQ, 32 * sub, 1, d, lane), d1, dk[d], 0, 0, 0);
                    }
                }
            }
        }
        dma_wait();
        __syncthreads();
        cur ^= 1;
    }
    if constexpr (SPLIT) {
        // image [(b, kvh, kb)][g][wave][d][g4][lane] f32x4; the dV pass owns the first half of `part`, the dK pass the second
        const size_t nimg = (size_t)gridDim.x;
        const size_t img = ((size_t)bx * p.nblk + kb) * group + g0 + (DO_DK ? nimg : 0);
        float* dst = p.part + img * (4 * ND * 16 * 64) + (size_t)wave * (ND * 16 * 64) + lane * 4;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x16& a = DO_DK ? dk[d] : dv[d];
                *reinterpret_cast<f32x4*>(dst + (d * 4 + g4) * 256) = f32x4{a[4 * g4], a[4 * g4 + 1], a[4 * g4 + 2], a[4 * g4 + 3]};
            }
    } else {
        bf16_t* slab = reinterpret_cast<bf16_t*>(smem_raw) + wave * 32 * (HD + 8);
        const int key0w = key - (lane & 31);                                           // the wave's first key
        bool fused_k = false;
        if constexpr (HD == 128 && DO_DK && !DO_DV) {
            if (p.fk.X) {
                float dw[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                store_rows_rope_bwd<HD, ND>(slab, dk, p.scale, p.fk, (size_t)b * T + key0w, key0w, kvh * HD, T - key0w, lane, dw);
                rope_bwd_dw_flush<HD>(dw, reinterpret_cast<float*>(smem_raw), p.fk.dw_part + (size_t)blockIdx.x * HD, wave, lane, tid);
                fused_k = true;
            }
        }
        if (DO_DK && !fused_k) store_rows<HD, ND>(slab, dk, p.scale, p.dK + ((size_t)b * T + key0w) * p.lddk + kvh * HD, p.lddk, T - key0w, lane);
        if (DO_DV) store_rows<HD, ND>(slab, dv, 1.0f, p.dV + ((size_t)b * T + key0w) * p.lddv + kvh * HD, p.lddv, T - key0w, lane);
    }
}

// dK AND dV in one pass at two waves per SIMD (7 executed matmuls per attention backward instead of 8, the exponentials of a
// block once instead of twice).  What made the one-pass form need a whole SIMD's registers (MODE 0: K and V fragments 64 + dK and dV accumulators
// 128 + two 32-query blocks in flight) is cut three ways: V is NOT held in registers — the workgroup's 128 V rows sit in LDS beside the stages and are
// read as row fragments per use (+8 KB of LDS reads per block: 40 KB per 32 MFMAs, the LDS-bandwidth cap moves from 1.0 to 0.8 of the matrix pipe's
// time; the two-pass kernels run at 0.45); the stages hold 32 query rows instead of 64 (2 x 16 KB + V 32 KB = 64 KB: two workgroups per CU still
// fit, which the 64-row stages + V would not), so one block is in flight per wave; the row statistics are read per 4-row group.
// SPLIT (one sample per GPU): one block per QUERY head, both accumulator images left in `part` for attn_dkv_reduce_kernel (see attn_bwd_dkv_kernel).
template <int HD, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_fused_kernel(AttnBwdArgs p) {
    static_assert(HD == 128, "built for head dim 128");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int QR = 32;                                       // query rows per stage
    constexpr int TILE = QR * HD;                                // elements of one 32-row image
    constexpr int NS = HD / 16, ND = HD / 32;
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);          // [2 stages][Q 32 rows | dO 32 rows]
    bf16_t* sV = smem + 2 * 2 * TILE;                            // [2 images of 64 keys][HD]
    float* sStat = reinterpret_cast<float*>(smem_raw + (2 * 2 * TILE + 2 * BKV * HD) * sizeof(bf16_t));   // [2 stages][lse 64 | delta 64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    wave_priority(p.prio);
    const int group = p.nh / p.nkv;
    const BlockItem bi = block_item(blockIdx.x, gridDim.x, p.nkv, SPLIT ? group : 1, p.nblk, p.order_set);
    const int kb = bi.blk, kvh = bi.kvh, b = bi.b, g0 = bi.g;
    const int T = p.T;
    const int lo = p.kv_lo ? p.kv_lo[b] : 0;
    const int hi = p.kv_hi ? p.kv_hi[b] : T;
    const int key0 = kb * 128 + wave * 32;
    const int key = key0 + r;
    const int krow = key < T ? key : T - 1;
    const bool key_ok = key >= lo && key < hi;

    // the workgroup's V rows -> LDS (two 64-row images); this wave's K rows -> registers
    {
        const bf16_t* Vb = p.V + (size_t)b * T * p.ldv + kvh * HD;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k0 = kb * 128 + 64 * j;
            stage_kv<HD, false>(Vb, p.ldv, k0 < T ? k0 : T - 1, T, sV + j * BKV * HD, wave, lane);
        }
    }
    bf16x8 kf[NS];
    {
        const bf16_t* kp = p.K + ((size_t)b * T + krow) * p.ldk + kvh * HD + 8 * h;
#pragma unroll
        for (int s = 0; s < NS; ++s) kf[s] = *reinterpret_cast<const bf16x8*>(kp + 16 * s);
#pragma unroll
        for (int s = 0; s < NS; ++s) arrived(kf[s]);
    }
    const bf16_t* sVw = sV + (wave >> 1) * BKV * HD;             // the image holding this wave's keys
    const int vrow = 32 * (wave & 1) + r;
    f32x16 dk[ND], dv[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dk[d][e] = 0.f; dv[d][e] = 0.f; }

    // 32-row query tiles this block needs: causal -> tiles whose last row >= first key of the block
    const int nqt = (T + QR - 1) / QR;
    const int qt_first = p.causal ? (kb * 128) / QR : 0;
    const int n_tiles = nqt > qt_first ? nqt - qt_first : 0;
    const int total = SPLIT ? n_tiles : n_tiles * group;

    unsigned rbQ, rbD, cbQ, cbD;
    stage_lane_const<HD, QR>(p.ldq, wave, lane, rbQ, cbQ);
    stage_lane_const<HD, QR>(p.ldo, wave, lane, rbD, cbD);
    const unsigned smem_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(smem));
    int sg = 0, sj = 0;                                          // (g, j) of the NEXT tile to stage
    int cj = 0;                                                  // j of the tile being computed
    auto stage = [&](int stg) {
        const int g = SPLIT ? g0 : sg, qt = qt_first + sj;
        if (++sj == n_tiles) { sj = 0; ++sg; }
        const int head = kvh * group + g;
        const bf16_t* Qb = p.Q + (size_t)b * T * p.ldq + head * HD;
        const bf16_t* Db = p.dO + (size_t)b * T * p.ldo + head * HD;
        bf16_t* sQ = smem + stg * 2 * TILE;
        if (qt * QR + QR <= T) {
            stage_tile_fast<HD, QR>(reinterpret_cast<const char*>(Qb + (size_t)qt * QR * p.ldq), p.ldq, rbQ, cbQ, smem_lds + stg * 2 * TILE * 2, wave);
            stage_tile_fast<HD, QR>(reinterpret_cast<const char*>(Db + (size_t)qt * QR * p.ldo), p.ldo, rbD, cbD, smem_lds + (stg * 2 + 1) * TILE * 2, wave);
        } else {
            stage_kv<HD, false, QR>(Qb, p.ldq, qt * QR, T, sQ, wave, lane);
            stage_kv<HD, false, QR>(Db, p.ldo, qt * QR, T, sQ + TILE, wave, lane);
        }
        if (wave < 2) {                                          // row statistics by LDS-DMA as well (the note in attn_bwd_dkv_kernel); lanes 32.. fetch rows nobody reads
            int q = qt * QR + lane;
            q = q < T ? q : T - 1;
            const size_t idx = ((size_t)b * p.nh + head) * T + q;
            const float* src = wave == 0 ? p.LSE + idx : p.delta + idx;
            const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(sStat + stg * 128 + wave * 64));
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(src), "s"(lds_addr) : "memory", "m0");
        }
    };
    if (total > 0) stage(0);
    dma_wait();
    __syncthreads();
    int cur = 0;
    for (int it = 0; it < total; ++it) {
        if (it + 1 < total) stage(cur ^ 1);
        const int qsub = (qt_first + cj) * QR;
        if (++cj == n_tiles) cj = 0;
        const bf16_t* sQ = smem + cur * 2 * TILE;
        const bf16_t* sD = sQ + TILE;
        const float* sL = sStat + cur * 128;
        if (!(p.causal && qsub + 31 < key0)) {                    // (else: these 32 queries all precede this wave's keys — wave-uniform)
            f32x16 sA, dpA;                                       // [q rows (regs), key cols (lane)]
#pragma unroll
            for (int e = 0; e < 16; ++e) sA[e] = 0.f;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {                      // dP starts from -delta (stored negated by the dQ kernel), read straight into the accumulator
                const f32x4 nd = *reinterpret_cast<const f32x4*>(sL + 64 + 8 * g4 + 4 * h);
                dpA[4 * g4] = nd[0]; dpA[4 * g4 + 1] = nd[1]; dpA[4 * g4 + 2] = nd[2]; dpA[4 * g4 + 3] = nd[3];
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                sA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(sQ, r, s, h), kf[s], sA, 0, 0, 0);
                dpA = __builtin_amdgcn_mfma_f32_32x32x16_bf16(row_frag<HD>(sD, r, s, h), row_frag<HD>(sVw, vrow, s, h), dpA, 0, 0, 0);
            }
            const bool need_mask = (p.causal && qsub < key0 + 31) || key0 < lo || key0 + 32 > hi || qsub + 32 > T;
            f32x16 pA;
            if (need_mask) {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(sL + 8 * g4 + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = 4 * g4 + j;
                        const int q = qsub + j + 8 * g4 + 4 * h;
                        const bool ok = key_ok && q < T && (!p.causal || key <= q);
                        const float pr = ok ? fast_exp2(sA[e] * p.scale_log2 - l4[j]) : 0.f;
                        pA[e] = pr;
                        sA[e] = pr * dpA[e];                     // dS (unscaled)
                    }
                }
            } else {
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(sL + 8 * g4 + 4 * h);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int e = 4 * g4 + j;
                        const float pr = fast_exp2(sA[e] * p.scale_log2 - l4[j]);
                        pA[e] = pr;
                        sA[e] = pr * dpA[e];
                    }
                }
            }
            {
                const bf16x8 p0 = acc_to_frag(pA, 0), p1 = acc_to_frag(pA, 8);
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HD>(sD, 0, 0, d, lane), p0, dv[d], 0, 0, 0);
                    dv[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HD>(sD, 0, 1, d, lane), p1, dv[d], 0, 0, 0);
                }
            }
            {
                const bf16x8 d0 = acc_to_frag(sA, 0), d1 = acc_to_frag(sA, 8);
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HD>(sQ, 0, 0, d, lane), d0, dk[d], 0, 0, 0);
                    dk[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag<HD>(sQ, 0, 1, d, lane), d1, dk[d], 0, 0, 0);
                }
            }
        }
        dma_wait();
        __syncthreads();
        cur ^= 1;
    }
    if constexpr (SPLIT) {
        // image [(b, kvh, kb)][g][wave][d][g4][lane] f32x4; dV images in the first half of `part`, dK images in the second (attn_dkv_reduce_kernel)
        const size_t nimg = (size_t)gridDim.x;
        const size_t img = ((size_t)(b * p.nkv + kvh) * p.nblk + kb) * group + g0;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            float* dst = p.part + (img + (which ? nimg : 0)) * (4 * ND * 16 * 64) + (size_t)wave * (ND * 16 * 64) + lane * 4;
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x16& a = which ? dk[d] : dv[d];
                    *reinterpret_cast<f32x4*>(dst + (d * 4 + g4) * 256) = f32x4{a[4 * g4], a[4 * g4 + 1], a[4 * g4 + 2], a[4 * g4 + 3]};
                }
        }
        return;
    }
    bf16_t* slab = reinterpret_cast<bf16_t*>(smem_raw) + wave * 32 * (HD + 8);
    const int key0w = key - (lane & 31);                                           // the wave's first key
    store_rows<HD, ND>(slab, dv, 1.0f, p.dV + ((size_t)b * T + key0w) * p.lddv + kvh * HD, p.lddv, T - key0w, lane);
    if (p.fk.X) {
        float dw[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        store_rows_rope_bwd<HD, ND>(slab, dk, p.scale, p.fk, (size_t)b * T + key0w, key0w, kvh * HD, T - key0w, lane, dw);
        rope_bwd_dw_flush<HD>(dw, reinterpret_cast<float*>(smem_raw), p.fk.dw_part + (size_t)blockIdx.x * HD, wave, lane, tid);
    } else {
        store_rows<HD, ND>(slab, dk, p.scale, p.dK + ((size_t)b * T + key0w) * p.lddk + kvh * HD, p.lddk, T - key0w, lane);
    }
}

// the head-split pass's second half: one WAVE per (batch, kv head, key block, 32 keys, dK | dV) — 100 MB of images at one sample
// per GPU want thousands of waves with loads in flight, not 192 workgroups (41 -> ~20 us at B = 1, T = 3072) — adds the group's
// images in head order, then the same row store as the unsplit kernel (dK scaled, dV not)
template <int HD>
__global__ __launch_bounds__(64) void attn_dkv_reduce_kernel(AttnBwdArgs p) {
    constexpr int ND = HD / 32;
    __shared__ __attribute__((aligned(16))) bf16_t slab[32 * (HD + 8)];
    const int lane = threadIdx.x;
    const int wave = blockIdx.x & 3, bx = blockIdx.x >> 2, which = blockIdx.z;        // which: 0 dV, 1 dK
    const int kb = blockIdx.y, kvh = bx % p.nkv, b = bx / p.nkv;
    const int group = p.nh / p.nkv;
    const size_t nimg = (size_t)(gridDim.x >> 2) * group * gridDim.y;
    const int key0w = kb * 128 + wave * 32;
    if (key0w >= p.T) return;
    const float* src = p.part + (((size_t)bx * gridDim.y + kb) * group + (which ? nimg : 0)) * (4 * ND * 16 * 64) +
                       (size_t)wave * (ND * 16 * 64) + lane * 4;
    f32x16 acc[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            f32x4 v = *reinterpret_cast<const f32x4*>(src + (d * 4 + g4) * 256);
            for (int g = 1; g < group; ++g) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(src + (size_t)g * (4 * ND * 16 * 64) + (d * 4 + g4) * 256);
                v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3];
            }
            acc[d][4 * g4] = v[0]; acc[d][4 * g4 + 1] = v[1]; acc[d][4 * g4 + 2] = v[2]; acc[d][4 * g4 + 3] = v[3];
        }
    if (which) store_rows<HD, ND>(slab, acc, p.scale, p.dK + ((size_t)b * p.T + key0w) * p.lddk + kvh * HD, p.lddk, p.T - key0w, lane);
    else store_rows<HD, ND>(slab, acc, 1.0f, p.dV + ((size_t)b * p.T + key0w) * p.lddv + kvh * HD, p.lddv, p.T - key0w, lane);
}

}  // namespace

// (batch, kv head) pairs an XCD label's workgroups walk together (block_item).  Default: ALL the pairs of the label's chunk, up to 8
// — inside an XCD the order is then heaviest block first over its pairs, the launch-wide order round 3 had, which matters more than
// the L2 footprint (r04 attn_order.log, B8 T2048 16/8 heads: one pair at a time 188.7 us forward, two 173.8, four 168.6, round 3's
// order 173.3; every XCD has the same work, so the labels finish together) — when the pairs divide evenly over the 8 labels;
// otherwise round 3's order (0).  MOLLY_ATTN_ORDER_SET (forward, dQ pass) / MOLLY_ATTN_ORDER_SET_DKV (dK / dV passes) pin a value.
static int wave_prio() {          // MOLLY_ATTN_PRIO: see wave_priority()
    static const int v = [] { const char* e = getenv("MOLLY_ATTN_PRIO"); return e ? atoi(e) : 0; }();
    return v;
}
static bool dkv_fused() {          // dK and dV in one pass (attn_bwd_dkv_fused_kernel) where the passes are not split by query head; MOLLY_ATTN_DKV_FUSED=0: the
    const char* e = getenv("MOLLY_ATTN_DKV_FUSED");          // two single-output passes (read per call: the A/B test switches it inside one process)
    return !e || atoi(e) != 0;
}
static int order_set(int dkv, int npairs) {
    static const int v[2] = {[] { const char* e = getenv("MOLLY_ATTN_ORDER_SET"); return e ? atoi(e) : -1; }(),
                             [] { const char* e = getenv("MOLLY_ATTN_ORDER_SET_DKV"); return e ? atoi(e) : -1; }()};
    if (v[dkv != 0] >= 0) return v[dkv != 0];
    if (npairs % 8 != 0) return 0;
    int s = npairs / 8;                              // pairs per XCD label; a set must not straddle two labels' chunks
    while (s > 8) s = (s % 2 == 0) ? s / 2 : (s % 3 == 0) ? s / 3 : (s % 5 == 0) ? s / 5 : 1;
    return s;
}

static bool attr_set_fwd() {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 65536);
        (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
        attr_set = true;
    }
    return true;
}

static int attn_fwd_impl(void* stream, const void* Q, const void* K, const void* V, void* O, float* lse2,
                              const int* kv_lo, const int* kv_hi, int B, int T, int n_heads, int n_kv_heads, int head_dim,
                              int ldq, int ldk, int ldv, int ldo, float scale, int causal, void* OT, int ldot) {
    MOLLY_ENTER();
    MOLLY_CHECK(head_dim == 128 || head_dim == 64 || head_dim == 16 || head_dim == 32 || head_dim == 8 || head_dim == 24 ||
                    head_dim == 40 || head_dim == 48,
                "attn_fwd: head_dim=%d not built (64 and 128 on the MFMA kernel; 8..48 in steps of 8 on the small-head kernel)",
                head_dim);
    MOLLY_CHECK(n_heads % n_kv_heads == 0, "attn_fwd: n_heads %% n_kv_heads != 0");
    MOLLY_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "attn_fwd: row strides must be multiples of 8");
    MOLLY_CHECK(((uintptr_t)Q % 16) == 0 && ((uintptr_t)K % 16) == 0 && ((uintptr_t)V % 16) == 0, "attn_fwd: alignment");
    MOLLY_CHECK(B > 0 && T > 0, "attn_fwd: empty problem");
    AttnArgs p{(const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, (bf16_t*)O, lse2, kv_lo, kv_hi, T, n_heads,
               n_kv_heads, ldq, ldk, ldv, ldo, scale * LOG2E, causal, cdiv(T, BQ), order_set(0, B * n_kv_heads), wave_prio(),
               (bf16_t*)OT, ldot};
    if (OT) {
        MOLLY_CHECK(head_dim == 128 || head_dim == 64, "attn_fwd: the transposed second store is built for head dims 64 and 128");
        MOLLY_CHECK(T % 128 == 0 && ldot >= B * T && ldot % 8 == 0 && ((uintptr_t)OT % 16) == 0,
                    "attn_fwd: transposed store needs T %% 128 == 0 (T=%d), ldot=%d >= B*T, 16-byte alignment", T, ldot);
    }
    if (head_dim < 64) {
        const dim3 g(n_heads * B, cdiv(T, 128));
#define MOLLY_SMALL(HD_) case HD_: hipLaunchKernelGGL(attn_fwd_small_kernel<HD_>, g, dim3(128), 0, (hipStream_t)stream, p); break
        switch (head_dim) { MOLLY_SMALL(8); MOLLY_SMALL(16); MOLLY_SMALL(24); MOLLY_SMALL(32); MOLLY_SMALL(40); MOLLY_SMALL(48); }
#undef MOLLY_SMALL
        MOLLY_LAUNCH_CHECK();
        return 0;
    }
    dim3 grid(n_heads * B * cdiv(T, BQ));                  // 1-D: block_item() decodes it
    // MOLLY_ATTN_LDS_PAD (diagnostic): extra dynamic LDS per workgroup, e.g. 40960 leaves room for ONE workgroup per CU — what the
    // second wave of every SIMD is worth is then the ratio of the two run times
    static const size_t pad = [] { const char* e = getenv("MOLLY_ATTN_LDS_PAD"); return e ? (size_t)atoi(e) : (size_t)0; }();
    const size_t lds = 2 * 2 * BKV * head_dim * sizeof(bf16_t) + pad;
    (void)attr_set_fwd();
    if (head_dim == 128)
        hipLaunchKernelGGL(attn_fwd_kernel<128>, grid, dim3(256), lds, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(attn_fwd_kernel<64>, grid, dim3(256), lds, (hipStream_t)stream, p);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_attn_fwd(void* stream, const void* Q, const void* K, const void* V, void* O, float* lse2,
                              const int* kv_lo, const int* kv_hi, int B, int T, int n_heads, int n_kv_heads, int head_dim,
                              int ldq, int ldk, int ldv, int ldo, float scale, int causal) {
    return attn_fwd_impl(stream, Q, K, V, O, lse2, kv_lo, kv_hi, B, T, n_heads, n_kv_heads, head_dim, ldq, ldk, ldv, ldo, scale, causal, nullptr, 0);
}
extern "C" int molly_attn_fwd_ot(void* stream, const void* Q, const void* K, const void* V, void* O, void* OT, float* lse2,
                                 const int* kv_lo, const int* kv_hi, int B, int T, int n_heads, int n_kv_heads, int head_dim,
                                 int ldq, int ldk, int ldv, int ldo, int ldot, float scale, int causal) {
    return attn_fwd_impl(stream, Q, K, V, O, lse2, kv_lo, kv_hi, B, T, n_heads, n_kv_heads, head_dim, ldq, ldk, ldv, ldo, scale, causal, OT, ldot);
}

#if MOLLY_ATTN_STAMP
extern "C" int molly_exp_attn_stamps(void* dst, int clear) {
    if (dst) (void)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_attn_stamp), sizeof(unsigned long long) * 32768 * 4 * 8);
    if (clear) (void)hipMemset((void*)nullptr, 0, 0);
    if (clear) { void* p = nullptr; (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_attn_stamp)); (void)hipMemset(p, 0, sizeof(unsigned long long) * 32768 * 4 * 8); }
    return 0;
}
#endif

// floats of scratch with which molly_attn_bwd_ws runs the dK / dV passes split by query head (0: the split does not apply —
// the grid over (kv head, key block) already fills the chip, or there is one query head per kv head, or head_dim != 128)
extern "C" int molly_attn_bwd_workspace(int B, int T, int n_heads, int n_kv_heads, int head_dim) {
    if (head_dim != 128 || n_kv_heads <= 0 || n_heads % n_kv_heads != 0 || n_heads == n_kv_heads || n_heads / n_kv_heads > 8) return 0;
    const long nkb = cdiv(T, 128);
    // (two blocks per CU: the split pays up to one round of the 512 slots — B = 2 x T = 4096: 1,373 -> 1,162 us; at 768 blocks it loses;
    // MOLLY_ATTN_SPLIT_MAX moves the bound for measurements: tools/bench_attn_b1.py)
    static const long max_blocks = [] { const char* e = getenv("MOLLY_ATTN_SPLIT_MAX"); return e ? atol(e) : 513L; }();
    if ((long)n_kv_heads * B * nkb >= max_blocks) return 0;
    return (int)(2L * n_heads * B * nkb * (4 * 4 * 16 * 64));          // dV and dK images: one per (query head, key block); < 2^28
}

static int attn_bwd_impl(void* stream, const void* Q, const void* K, const void* V, const void* O, const void* dO,
                              const float* lse2, float* delta_ws, void* dQ, void* dK, void* dV, const int* kv_lo,
                              const int* kv_hi, int B, int T, int n_heads, int n_kv_heads, int head_dim, int ldq, int ldk,
                              int ldv, int ldo, int lddo, int lddq, int lddk, int lddv, float scale, int causal,
                              float* workspace, long workspace_floats, const RopeBwdFuse* fq = nullptr, const RopeBwdFuse* fk = nullptr) {
    MOLLY_ENTER();
    MOLLY_CHECK(head_dim == 128 || head_dim == 64, "attn_bwd: head_dim=%d not built (64 and 128 are)", head_dim);
    MOLLY_CHECK(n_heads % n_kv_heads == 0, "attn_bwd: n_heads %% n_kv_heads != 0");
    MOLLY_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 && lddo % 8 == 0 && lddq % 4 == 0 &&
                    lddk % 4 == 0 && lddv % 4 == 0, "attn_bwd: row strides must be multiples of 8");
    MOLLY_CHECK(delta_ws && lse2, "attn_bwd: lse2 and a delta workspace of B*n_heads*T floats are required");
    MOLLY_CHECK(lddo == ldo, "attn_bwd: dO must share O's row stride");
    hipStream_t st = (hipStream_t)stream;
    AttnBwdArgs p{(const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, (const bf16_t*)dO, (const bf16_t*)O, lse2, delta_ws,
                  (bf16_t*)dQ, (bf16_t*)dK, (bf16_t*)dV, kv_lo, kv_hi, T, n_heads, n_kv_heads, ldq, ldk, ldv, ldo,
                  lddq, lddk, lddv, scale, scale * LOG2E, causal, nullptr, cdiv(T, BQ), order_set(0, B * n_kv_heads), wave_prio()};
    const size_t lds_dq = 2 * 2 * BKV * head_dim * sizeof(bf16_t);
    const size_t lds_dkv = lds_dq + 2 * 128 * sizeof(float);
    const size_t lds_fused = (2 * 2 * 32 * 128 + 2 * BKV * 128) * sizeof(bf16_t) + 2 * 128 * sizeof(float);   // attn_bwd_dkv_fused_kernel<128>
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
        (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<128, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 1024);
        (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<128, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 1024);
        (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<64, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768 + 1024);
        (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<128, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 1024);
        (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<128, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 1024);
        (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_fused_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fused);
        (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_fused_kernel<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fused);
        attr_set = true;
    }
    const dim3 gq(n_heads * B * cdiv(T, BQ)), gk(n_kv_heads * B, cdiv(T, 128));   // dQ pass 1-D; gk: the reduce kernel's 2-D shape
    const dim3 gk1(gk.x * gk.y);                                                    // dK / dV passes 1-D (block_item)
    if (fq) {
        // (the fused form never splits the dK / dV passes by query head: callers for whom molly_attn_bwd_workspace(..) > 0 — one sample per GPU —
        // are better served by molly_attn_bwd_ws + molly_norm_rope_bwd; qwen3.py decides that way)
        MOLLY_CHECK(fk && head_dim == 128, "attn_bwd_rope: the fused q/k-norm + rotary backward is built for head dim 128");
        p.fq = *fq;
        p.fk = *fk;
    }
    AttnBwdArgs pk = p;                                                             // their block order
    pk.nblk = cdiv(T, 128);
    pk.order_set = order_set(1, B * n_kv_heads);
    // head dim 128: dK and dV in ONE pass at two waves per SIMD (attn_bwd_dkv_fused_kernel: V read from LDS, 32-row stages; round 6: 1,289 -> 1,087 us
    // per backward at B 16) unless the passes are split by query head (one sample per GPU: two passes + reduce); head dim 64 (the encoders): one pass
    if (head_dim == 128) {
        hipLaunchKernelGGL(attn_bwd_dq_kernel<128>, gq, dim3(256), lds_dq, st, p);
        const long need = molly_attn_bwd_workspace(B, T, n_heads, n_kv_heads, head_dim);
        if (need > 0 && workspace && workspace_floats >= need) {
            p.part = pk.part = workspace;
            const dim3 gs(n_heads * B * cdiv(T, 128));
            if (dkv_fused()) {
                hipLaunchKernelGGL((attn_bwd_dkv_fused_kernel<128, true>), gs, dim3(256), lds_fused, st, pk);
            } else {
                hipLaunchKernelGGL((attn_bwd_dkv_kernel<128, 1, true>), gs, dim3(256), lds_dkv, st, pk);
                hipLaunchKernelGGL((attn_bwd_dkv_kernel<128, 2, true>), gs, dim3(256), lds_dkv, st, pk);
            }
            hipLaunchKernelGGL(attn_dkv_reduce_kernel<128>, dim3(gk.x * 4, gk.y, 2), dim3(64), 0, st, p);
        } else if (dkv_fused()) {
            hipLaunchKernelGGL(attn_bwd_dkv_fused_kernel<128>, gk1, dim3(256), lds_fused, st, pk);
        } else {
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<128, 1>), gk1, dim3(256), lds_dkv, st, pk);
            hipLaunchKernelGGL((attn_bwd_dkv_kernel<128, 2>), gk1, dim3(256), lds_dkv, st, pk);
        }
    } else {
        hipLaunchKernelGGL(attn_bwd_dq_kernel<64>, gq, dim3(256), lds_dq, st, p);
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<64, 0>), gk1, dim3(256), lds_dkv, st, pk);
    }
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_attn_bwd(void* stream, const void* Q, const void* K, const void* V, const void* O, const void* dO,
                              const float* lse2, float* delta_ws, void* dQ, void* dK, void* dV, const int* kv_lo,
                              const int* kv_hi, int B, int T, int n_heads, int n_kv_heads, int head_dim, int ldq, int ldk,
                              int ldv, int ldo, int lddo, int lddq, int lddk, int lddv, float scale, int causal) {
    return attn_bwd_impl(stream, Q, K, V, O, dO, lse2, delta_ws, dQ, dK, dV, kv_lo, kv_hi, B, T, n_heads, n_kv_heads, head_dim, ldq, ldk,
                         ldv, ldo, lddo, lddq, lddk, lddv, scale, causal, nullptr, 0);
}
extern "C" int molly_attn_bwd_rope_blocks(int B, int T, int n_heads, int n_kv_heads, int which) {
    return (which ? n_kv_heads : n_heads) * B * cdiv(T, 128);
}
extern "C" int molly_attn_bwd_rope(void* stream, const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse2,
                                   float* delta_ws, void* dV, const int* kv_lo, const int* kv_hi, int B, int T, int n_heads, int n_kv_heads,
                                   int head_dim, int ldq, int ldk, int ldv, int ldo, int lddo, int lddv, float scale, int causal, const void* X,
                                   int ldx, const void* qw, const void* kw, const float* cos_t, const float* sin_t, float eps, void* dX, int lddx,
                                   float* dwq_part, float* dwk_part) {
    MOLLY_ENTER();
    MOLLY_CHECK(X && qw && kw && cos_t && sin_t && dX && dwq_part && dwk_part, "attn_bwd_rope: null operand");
    MOLLY_CHECK(ldx % 8 == 0 && lddx % 8 == 0 && ((uintptr_t)X % 16) == 0 && ((uintptr_t)dX % 16) == 0 && ((uintptr_t)qw % 16) == 0 &&
                ((uintptr_t)kw % 16) == 0 && ((uintptr_t)cos_t % 16) == 0 && ((uintptr_t)sin_t % 16) == 0,
                "attn_bwd_rope: X / dX row strides must be multiples of 8 and every operand 16-byte aligned");
    const size_t kcol = (size_t)n_heads * head_dim;                  // k heads follow the q heads in the projection's row
    const RopeBwdFuse fq{(const bf16_t*)X, ldx, (const bf16_t*)qw, cos_t, sin_t, (bf16_t*)dX, lddx, dwq_part, eps};
    const RopeBwdFuse fk{(const bf16_t*)X + kcol, ldx, (const bf16_t*)kw, cos_t, sin_t, (bf16_t*)dX + kcol, lddx, dwk_part, eps};
    // (dQ / dK are not written: the rows leave as d(q | k | v); the strides passed for them only have to satisfy the checks)
    return attn_bwd_impl(stream, Q, K, V, O, dO, lse2, delta_ws, dX, (bf16_t*)dX + kcol, dV, kv_lo, kv_hi, B, T, n_heads, n_kv_heads, head_dim, ldq,
                         ldk, ldv, ldo, lddo, lddx, lddx, lddv, scale, causal, nullptr, 0, &fq, &fk);
}
extern "C" int molly_attn_bwd_ws(void* stream, const void* Q, const void* K, const void* V, const void* O, const void* dO,
                                 const float* lse2, float* delta_ws, void* dQ, void* dK, void* dV, const int* kv_lo,
                                 const int* kv_hi, int B, int T, int n_heads, int n_kv_heads, int head_dim, int ldq, int ldk,
                                 int ldv, int ldo, int lddo, int lddq, int lddk, int lddv, float scale, int causal,
                                 float* workspace, long workspace_floats) {
    return attn_bwd_impl(stream, Q, K, V, O, dO, lse2, delta_ws, dQ, dK, dV, kv_lo, kv_hi, B, T, n_heads, n_kv_heads, head_dim, ldq, ldk,
                         ldv, ldo, lddo, lddq, lddk, lddv, scale, causal, workspace, workspace_floats);
}
