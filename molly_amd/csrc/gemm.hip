// bf16 MFMA GEMM for gfx950:  C[M,N] = op(A)[M,K] · op(B)[K,N]  (+bias[N]) (GELU) (+res[M,N]) (+= C)
//
// Operand layouts (template flags AT / BT): an operand is either "k-contiguous" (stored [rows][K]: A as [M][K], B as
// [N][K]) or "k-major" (stored [K][rows]: A as [K][M], B as [K][N]).  The three forms on Molly's hot path:
//   forward  y  = x W^T      : A=x [M][K] k-contig,  B=W  [N][K] k-contig            (AT=0, BT=0, "NT")
//   dgrad    dx = dy W       : A=dy[M][N'] k-contig, B=W  [N'][K'] = [K][N] k-major   (AT=0, BT=1, "NN")
//   wgrad    dW = dy^T x     : A=dy[Mtok][N'] = [K][M] k-major, B=x [Mtok][K'] = [K][N] k-major (AT=1, BT=1, "TN")
// (HF nn.Linear: y = x W^T; reference call sites src/model/omics_one.py:75-91,175 and the autograd of those).
// k-major tiles are staged row-major [64 k][cols] and their MFMA fragments are gathered with ds_read_b64_tr_b16
// (guide T10), so no operand is ever transposed in HBM.
//
// Two kernels:
//   gemm256_kernel  256x256x64 tile, 8 waves, all 160 KiB of LDS, persistent, ping-pong wave groups (described at the
//                   kernel) — every grid that fills the chip, and through split-K the long-K / skinny ones.
//   gemm_kernel     128x128x64 tile, 4 waves (2x2, 64x64 each), 64 KiB LDS, 2 blocks per CU — what is left.
//                   (A 256x128 3-stage ring and a BK=32 variant of this template were built and measured slower than
//                   both on every step shape; they are not instantiated.)
// Common: mfma_f32_16x16x32_bf16, global_load_lds 16 B/lane staging with the swizzle on the SOURCE address (LDS image
// is lane-linear), XOR-swizzled chunks (conflict-free ds_read_b128 / tr reads), XCD-aware tile order, MFMA operands
// passed swapped so every lane owns 4 CONSECUTIVE n of one row m (8-byte bf16 / 16-byte fp32 stores).
#include <type_traits>
#include "common.h"
#include "molly_hip.h"

namespace {

constexpr int BN = 128;

struct GemmArgs {
    const bf16_t* A; const bf16_t* B; void* C;
    const bf16_t* bias; const bf16_t* res;
    int M, N, K, lda, ldb, ldc, ldres;
    int flags;
    int tiles_m, tiles_n;
    const bf16_t* zeros;      // >= 16 bytes of zeros: source of k-rows beyond K for k-major operands
    int splits;               // split-K factor of the 256x256 kernel (1 = none)
    int group_m;              // M-tiles per group in the tile walk of the 256x256 kernel
    float* ws;                // fp32 partial slabs [splits][M][N] when splits > 1
    // stream-K (gemm256_kernel<..., SKM = true>): the K-tiles of ALL tiles, tile after tile, are cut evenly over the blocks
    int sk;                   // K-tiles per unit (the granule a cut falls on, >= 2): every piece of a tile is >= sk K-tiles long
    int sk_tile_aligned;      // 1: the 8 XCD labels own whole tiles each (no piece of a tile crosses from one label to the next)
    unsigned* sk_flag;        // [tiles] units of the tile whose pieces have been written, 64 bytes apart, zero between launches
    float* sk_slab;           // [grid][2][8 waves][32 quads][64 lanes] f32x4: the accumulators of the block's (at most two) pieces
    // dynamic tile fetch (gemm256_kernel<..., DYN = true>): one ticket counter per XCD label, 64 bytes apart, zero between launches
    unsigned* dyn_cnt;
    // K-extension (gemm256_kernel<..., KX = true>, the NT form): k2 more K-tiles behind the K / 64 of (A, B), read from a SECOND operand pair —
    // C = A B^T + A2 B2^T in one accumulation (a LoRA branch's t B_lora^T riding in the base projection: A2 = t [M][64 k2], B2 = [N][64 k2])
    const bf16_t* A2; const bf16_t* B2;
    int lda2, ldb2, k2;
    // grouped launch (gemm256_kernel<..., GRP>): up to 16 problems sharing K, layouts and epilogue flags; one work list
    int ngroup;
    struct Group {
        const bf16_t* A; const bf16_t* B; void* C;
        int M, N, lda, ldb, ldc, tiles_m, tiles_n, trans_out, work0;
    } grp[16];
};

// One 1-KiB LDS-DMA piece (16 B per lane).  HIDE = issue it from inline asm, so that hipcc does not know an LDS write is in
// flight: knowing, it puts `s_waitcnt vmcnt(0)` in front of the first ds_read_b64_tr_b16 of every K-tile (the transposed-read
// intrinsic carries no alias information, so it is made to wait for EVERY pending LDS-DMA) — which drained the two-tiles-ahead
// prefetch of the k-major-operand forms (dgrad, wgrad) once per K-tile while the hand-placed counted waits sat unused beside it.
// With HIDE every wait is the kernel's own (the 256x256 kernel's schedule already names them all); the 128x128 kernel keeps
// the builtin because it relies on the waits hipcc derives.  (guide §5 'Three .s-level traps', §5.7)
// Hazards inside the string are ours: m0 is written by an SALU move one wait state before the load (s_nop 0); the SGPR base
// must not come from a VALU write in the 5 preceding instructions — checked on the generated code by
// tools/check_asm_dma_hazards.py rather than padded with `s_nop 4` (which measured -2...-6 % on every form).
#ifndef MOLLY_GEMM_RES_AHEAD
#define MOLLY_GEMM_RES_AHEAD 4     // row groups of the residual in flight ahead of the residual-add epilogue (of 8)
#endif
#ifndef MOLLY_GEMM_DIAG_NOEPI
#define MOLLY_GEMM_DIAG_NOEPI 0     // timing-only diagnostic builds (tools/build_variant.py): 1 = the plain 16-byte epilogue neither packs nor stores
#endif                              // (wrong results; the accumulators are kept live) — the upper bound of what hiding the epilogue can buy
#ifndef MOLLY_GEMM_SE_FORM
#define MOLLY_GEMM_SE_FORM 1        // streaming epilogue: 1 = the product; 2 = timing-only diagnostic, the row halves are packed but never stored
#endif
#ifndef MOLLY_GEMM_EPI_LDS
#define MOLLY_GEMM_EPI_LDS 1        // epilogues that read or write more than the plain tile (SwiGLU backward / forward, residual add): 1 = the wave's packed
#endif                              // accumulators go through a 4 KB slab of the free A stage and come back with lane = (row of 8, 16-byte chunk of 8), so every
                                    // load and store instruction covers whole 128-byte lines (2.4x the store rate of one CU: tools/r06/store_diag); 0 = round 2's
                                    // lane-row regrouping (16 rows x 64 bytes per instruction) — A/B
#ifndef MOLLY_GEMM_SBW_AHEAD
#define MOLLY_GEMM_SBW_AHEAD 2     // row groups of gate / up in flight ahead of the SwiGLU-backward epilogue's arithmetic (of 8)
#endif
#ifndef MOLLY_GEMM_ASM_DMA
#define MOLLY_GEMM_ASM_DMA 1
#endif
// one word through LDS (ds_write_b32 / ds_read_b32: a generic pointer would make these flat instructions, which count on vmcnt too)
__device__ __forceinline__ void lds_put(void* p, unsigned v) { *reinterpret_cast<volatile __attribute__((address_space(3))) unsigned*>(LDS_PTR(p)) = v; }
__device__ __forceinline__ unsigned lds_get(void* p) { return *reinterpret_cast<volatile __attribute__((address_space(3))) unsigned*>(LDS_PTR(p)); }

template <bool HIDE, bool NT = false>
__device__ __forceinline__ void dma16(const char* base, unsigned off, bf16_t* lds_dst) {
    if constexpr (HIDE && MOLLY_GEMM_ASM_DMA && NT) {
        // non-temporal: bytes ONE workgroup reads ONCE (the weight stream of a decode-row launch) do not displace what the chip re-reads
        const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(lds_dst));
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(off), "s"(base), "s"(lds_addr)
                     : "memory", "m0");
    } else if constexpr (HIDE && MOLLY_GEMM_ASM_DMA) {
        const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(lds_dst));
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds_addr)
                     : "memory", "m0");
    } else {
        __builtin_amdgcn_global_load_lds(GLB_PTR(base + off), LDS_PTR(lds_dst), 16, 0, NT ? 2 : 0);       // (aux 2 = nt)
    }
}
template <bool HIDE>
__device__ __forceinline__ void dma16_lane(const char* src, bf16_t* lds_dst) {        // per-lane 64-bit source address
    if constexpr (HIDE && MOLLY_GEMM_ASM_DMA) {
        const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)LDS_PTR(lds_dst));
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_addr) : "memory", "m0");
    } else {
        __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds_dst), 16, 0, 0);
    }
}

// ---- k-contiguous operand: tile [ROWS][64] bf16 (128-B LDS rows); one wave-instruction = 8 rows (1 KiB).
// chunk swizzle: ch ^ (row & 7)
// remap_ff > 0 (SwiGLU-fused gate|up projection, MOLLY_GEMM_SWIGLU): the half-tile's 128 rows are not 128 consecutive weight
// rows but, in 32-row groups, [gate c..c+31 | up c..c+31 | gate c+32..c+63 | up c+32..c+63] with c = row0 (an index into the
// `ff` activation columns; up rows live `remap_ff` rows below their gate rows).  A wave's two 32-column n-halves are then the
// gate and the up projection of the SAME 32 activation columns, so its epilogue holds both values of every element it owns.
template <int ROWS, int NW, int BK, bool HIDE = false, bool NT = false>
__device__ __forceinline__ void stage_kc(const bf16_t* __restrict__ g, int ld, int row0, int rows_total, int k0,
                                         bf16_t* lds_tile, int wave, int lane, int remap_ff = 0, int remap_shift = 5) {
    constexpr int CPR = BK / 8;                   // chunks per row (8: 128-B rows, 4: 64-B rows)
    constexpr int RPI = 64 / CPR;                 // rows per wave-instruction
    constexpr int PER = ROWS / RPI / NW;
    const int r_in = lane / CPR;
    // swizzle on the SOURCE chunk: BK=64: ch ^ (row&7) ; BK=32: ch ^ ((row>>2)&3)   (row = inst*RPI + r_in)
    const int c_src = BK == 64 ? ((lane & 7) ^ r_in) : ((lane & 3) ^ ((r_in >> 2) & 3));
    // address = wave-uniform 64-bit base (tile row 0, K offset: SGPRs) + one 32-bit per-lane byte offset: the K advance
    // of the main loop then lives in scalar registers and each LDS-DMA costs ONE address VGPR
    const int r0 = remap_ff ? 0 : (row0 < rows_total ? row0 : rows_total - 1);   // a half-tile may start past the last row
    const char* base = reinterpret_cast<const char*>(g + (size_t)r0 * ld + k0);
    const int last = rows_total - 1 - r0;         // clamp: OOB rows re-read a valid row, masked at store
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int inst = wave * PER + i;
        int rel = inst * RPI + r_in;
        // (groups of 1 << remap_shift rows: gate, up, gate, up ..; ff a multiple of the tile's columns: no ragged tile)
        if (remap_ff) rel = ((rel >> remap_shift) & 1) * remap_ff + row0 + ((rel >> (remap_shift + 1)) << remap_shift) + (rel & ((1 << remap_shift) - 1));
        else rel = rel < last ? rel : last;
        const unsigned off = ((unsigned)rel * (unsigned)ld + (unsigned)(c_src * 8)) * 2u;
        dma16<HIDE, NT>(base, off, lds_tile + inst * 512);
    }
}

template <int BK>
__device__ __forceinline__ bf16x8 frag_kc(const bf16_t* lds_tile, int row, int chunk) {
    const int phys = BK == 64 ? (chunk ^ (row & 7)) : (chunk ^ ((row >> 2) & 3));
    return *reinterpret_cast<const bf16x8*>(lds_tile + row * BK + phys * 8);
}

// ---- k-major operand: tile [BK k][COLS] bf16, stored SUB-TILED for the transposed reads: blocks of 4 k-rows x 16
// columns (128 B, row pitch 32 B); block (kblk = k>>2, cblk = col>>4) sits at block index kblk*NCB + (cblk ^ ((kblk>>1)&1)),
// NCB = COLS/16.  A ds_read_b64_tr_b16 of the 16x16x32 operand (half-wave = k-blocks 2g and 2g' of ONE column block)
// then touches two ADJACENT 128-byte blocks = 256 contiguous bytes (conflict-free), and every address is
// lane_base + compile-time constant (the XOR only involves lane bits), so the unrolled reads use DS immediates.
template <int COLS, int NW, int BK, bool HIDE = false>
__device__ __forceinline__ void stage_km(const bf16_t* __restrict__ g, int ld, int col0, int cols_total, int k0, int k_total,
                                         const bf16_t* zeros, bf16_t* lds_tile, int wave, int lane) {
    constexpr int NCB = COLS / 16;
    constexpr int NINST = BK * COLS * 2 / 1024;   // 1 KiB per wave-instruction = 8 blocks
    constexpr int PER = NINST / NW;
    const int c0 = col0 <= cols_total - 8 ? col0 : cols_total - 8;     // a half-tile may start past the last column
    const int last = cols_total - 8 - c0;         // clamp: OOB columns duplicate valid data, masked at store
    const bool full = k0 + BK <= k_total;         // wave-uniform: every K-tile but a ragged last one
    // full K-tile: wave-uniform 64-bit base + one 32-bit per-lane byte offset (see stage_kc)
    const char* base = reinterpret_cast<const char*>(g + (size_t)k0 * ld + c0);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int inst = wave * PER + i;
        const int P = inst * 64 + lane;            // destination 16-byte slot
        const int blk = P >> 3, cin = P & 7;
        const int kblk = blk / NCB, cbs = blk % NCB;
        const int kr = kblk * 4 + (cin >> 1);
        int rel = col0 - c0 + ((cbs ^ ((kblk >> 1) & 1)) << 4) + ((cin & 1) << 3);
        rel = rel < last ? rel : last;
        const unsigned off = ((unsigned)kr * (unsigned)ld + (unsigned)rel) * 2u;
        if (full) {
            dma16<HIDE>(base, off, lds_tile + inst * 512);
        } else {
            const char* src = (k0 + kr < k_total) ? base + off : reinterpret_cast<const char*>(zeros);   // rows past K: exact zeros
            dma16_lane<HIDE>(src, lds_tile + inst * 512);
        }
    }
}

// ---- The 256x256 kernel splits stage_kc / stage_km in two: the per-lane byte offsets of a half-tile (row / column clamps of a
// ragged edge included) depend on the work item, not on the K-tile, so they are computed when the staged work item changes
// (once per tile, ~10 vector instructions per half-tile) and every LDS-DMA piece of the K loop is then a wave-uniform base that
// advances by a constant per K-tile plus one of these registers: no vector arithmetic, no per-piece edge test, ONE code path.
// Returns the row / column the wave-uniform base has to point at.
template <int ROWS, int NW, int BK>
__device__ __forceinline__ int stage_offsets_kc(int ld, int row0, int rows_total, int wave, int lane, int remap_ff,
                                                unsigned (&off)[ROWS / (64 / (BK / 8)) / NW]) {
    constexpr int CPR = BK / 8, RPI = 64 / CPR, PER = ROWS / RPI / NW;
    const int r_in = lane / CPR;
    const int c_src = BK == 64 ? ((lane & 7) ^ r_in) : ((lane & 3) ^ ((r_in >> 2) & 3));
    const int r0 = remap_ff ? row0 : (row0 < rows_total ? row0 : rows_total - 1);       // a half-tile may start past the last row
    const int last = rows_total - 1 - r0;                                               // clamp: see stage_kc
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        int rel = (wave * PER + i) * RPI + r_in;
        if (remap_ff > 0) rel = ((rel >> 5) & 1) * remap_ff + (rel >> 6) * 32 + (rel & 31);   // ff % 128 == 0: no ragged tile
        // (remap_ff < 0, the streaming epilogue: LDS row 64 w + 16 j + f holds row 64 w + 4 f + j — whole tiles only)
        else if (remap_ff < 0) rel = (rel & 64) + ((rel & 15) << 2) + ((rel >> 4) & 3);
        else rel = rel < last ? rel : last;
        off[i] = ((unsigned)rel * (unsigned)ld + (unsigned)(c_src * 8)) * 2u;
    }
    return r0;
}
template <int COLS, int NW, int BK>
__device__ __forceinline__ int stage_offsets_km(int ld, int col0, int cols_total, int wave, int lane,
                                                unsigned (&off)[BK * COLS * 2 / 1024 / NW]) {
    constexpr int NCB = COLS / 16, PER = BK * COLS * 2 / 1024 / NW;
    const int c0 = col0 <= cols_total - 8 ? col0 : cols_total - 8;
    const int last = cols_total - 8 - c0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int P = (wave * PER + i) * 64 + lane;
        const int blk = P >> 3, cin = P & 7;
        const int kblk = blk / NCB, cbs = blk % NCB;
        const int kr = kblk * 4 + (cin >> 1);
        int rel = col0 - c0 + ((cbs ^ ((kblk >> 1) & 1)) << 4) + ((cin & 1) << 3);
        rel = rel < last ? rel : last;
        off[i] = ((unsigned)kr * (unsigned)ld + (unsigned)rel) * 2u;
    }
    return c0;
}
template <int PER, bool HIDE>
__device__ __forceinline__ void stage_pre(const char* base, const unsigned (&off)[PER], bf16_t* lds_tile, int wave) {
#pragma unroll
    for (int i = 0; i < PER; ++i) dma16<HIDE>(base, off[i], lds_tile + (wave * PER + i) * 512);
}

// fragment for mfma_16x16x32: lane (i = lane&15, g = lane>>4) gets tile[k = 32*kk + 8g + j][col16 + i], j = 0..7
// (col16 a multiple of 16; kk, col16 compile-time / wave-uniform at the call sites)
template <int COLS>
__device__ __forceinline__ bf16x8 frag_km(const bf16_t* lds_tile, int kk, int col16, int lane) {
    constexpr int NCB = COLS / 16;
    const int g = lane >> 4, gi = lane & 15, q = gi >> 2, pp = gi & 3;
    // rows 32kk + 8g + q (+4): kblk = 8kk + 2g (+1) -> (kblk>>1)&1 = g&1 for both reads
    const int cblk = col16 >> 4;
    const int lane_off = (2 * g * NCB) * 64 + q * 16 + (pp >> 1) * 8 + (pp & 1) * 4;
    const int cst = (8 * kk * NCB) * 64;
    // (cblk ^ (g&1)) written as (cblk & ~1) + ((cblk & 1) ^ (g & 1)): only the parity term is per-lane
    const int cb = ((cblk & ~1) + ((cblk & 1) ^ (g & 1))) * 64;
    const bf16_t* pa = lds_tile + lane_off + cst + cb;
    const bf16_t* pb = pa + NCB * 64;
    const bf16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)pa);
    const bf16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)pb);
    return bf16x8{va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752f)); }

// NTB: the k-contiguous B operand (a weight matrix) loaded non-temporal — launches of ONE row tile, where every weight byte is read once by one
// workgroup (the lm_head of a decode step: 1.2 GB)
template <bool AT, bool BT, int BM, int NSTAGE, int BK, bool NTB = false>
__global__ __launch_bounds__(BM * 2) void gemm_kernel(GemmArgs p) {
    constexpr int NW = BM / 32;                       // waves: (BM/64) x 2, each 64x64
    constexpr int A_ELEMS = BM * BK, B_ELEMS = BN * BK, STAGE = A_ELEMS + B_ELEMS;
    constexpr int NLOAD = (BM + BN) * BK * 2 / 1024 / NW;     // LDS-DMA instructions per wave per stage (both layouts)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);          // [stage][A tile | B tile]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    // ---- XCD-aware tile mapping: blocks b and b+8 share an XCD (L2); give each XCD a contiguous chunk of the tile
    // list, and walk tiles in groups of GROUP_M M-tiles per N sweep so neighbours share operand panels.
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    const int swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    constexpr int GROUP_M = BM == 256 ? 4 : 8;
    const int per_group = GROUP_M * p.tiles_n;
    const int grp = swz / per_group;
    const int first_m = grp * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int tm = first_m + (swz % per_group) % gsz;
    const int tn = (swz % per_group) / gsz;
    const int m0 = tm * BM, n0 = tn * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (p.K + BK - 1) / BK;
    // (the 3-stage ring names every wait itself: its LDS-DMA is issued from asm, invisible to hipcc — dma16's note; the 2-stage
    // loop relies on the vmcnt(0) hipcc puts in front of __syncthreads while an LDS-DMA is in flight)
    constexpr bool HIDE = NSTAGE != 2;
    auto stage = [&](int t, bf16_t* dst) {
        if (AT) stage_km<BM, NW, BK, HIDE>(p.A, p.lda, m0, p.M, t * BK, p.K, p.zeros, dst, wave, lane);
        else stage_kc<BM, NW, BK, HIDE>(p.A, p.lda, m0, p.M, t * BK, dst, wave, lane);
        if (BT) stage_km<BN, NW, BK, HIDE>(p.B, p.ldb, n0, p.N, t * BK, p.K, p.zeros, dst + A_ELEMS, wave, lane);
        else stage_kc<BN, NW, BK, HIDE, NTB>(p.B, p.ldb, n0, p.N, t * BK, dst + A_ELEMS, wave, lane);
    };
    const int fr = lane & 15, fq = lane >> 4;
    auto read_frags = [&](const bf16_t* sA, int kk, bf16x8 (&af)[4], bf16x8 (&bfr)[4]) {
        const bf16_t* sB = sA + A_ELEMS;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i] = AT ? frag_km<BM>(sA, kk, wm * 64 + i * 16, lane) : frag_kc<BK>(sA, wm * 64 + i * 16 + fr, kk * 4 + fq);
            bfr[i] = BT ? frag_km<BN>(sB, kk, wn * 64 + i * 16, lane) : frag_kc<BK>(sB, wn * 64 + i * 16 + fr, kk * 4 + fq);
        }
    };
    auto mma = [&](const bf16x8 (&af)[4], const bf16x8 (&bfr)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    };

    if constexpr (NSTAGE == 2) {
        stage(0, smem);
        __syncthreads();   // hipcc emits s_waitcnt vmcnt(0) ahead of the barrier while LDS-DMA is in flight
        int cur = 0;
        for (int t = 0; t < nk; ++t) {
            if (t + 1 < nk) stage(t + 1, smem + (cur ^ 1) * STAGE);
#pragma unroll
            for (int kk = 0; kk < BK / 32; ++kk) {
                bf16x8 af[4], bfr[4];
                read_frags(smem + cur * STAGE, kk, af, bfr);
                mma(af, bfr);
            }
            __syncthreads();
            cur ^= 1;
        }
    } else {
        // 3-stage LDS ring (tiles t+1, t+2 in flight behind a COUNTED vmcnt, raw barrier: guide §5 "Pipelining across
        // barriers") + register double-buffering of the MFMA fragments: the reads of the next 32-deep sub-step are
        // issued before the MFMAs of the current one, so the matrix pipe only idles across the one barrier per K-step.
        static_assert((NLOAD == 6 || NLOAD == 8) && BK == 64, "vmcnt immediates below: 6 or 8 LDS-DMA instructions per wave per stage");
        // s_waitcnt vmcnt(k * NLOAD) (+ lgkmcnt(0)): all but the k youngest stages' pieces have landed
        auto wait_stages = [&](auto k_tag, bool lgkm) {
            constexpr int N = decltype(k_tag)::value * NLOAD;
            if (lgkm) {
                if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
                else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            } else {
                if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            }
        };
        using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>; using K2 = std::integral_constant<int, 2>;
        stage(0, smem);
        if (nk > 1) stage(1, smem + STAGE);
        if (nk > 2) stage(2, smem + 2 * STAGE);
        if (nk > 2) wait_stages(K2{}, false);
        else if (nk > 1) wait_stages(K1{}, false);
        else wait_stages(K0{}, false);
        __builtin_amdgcn_s_barrier();
        bf16x8 a0[4], b0[4], a1[4], b1[4];
        read_frags(smem, 0, a0, b0);
        int s_cur = 0;
        for (int t = 0; t < nk; ++t) {
            const bf16_t* cur = smem + s_cur * STAGE;
            const int s_nxt = s_cur == 2 ? 0 : s_cur + 1;
            read_frags(cur, 1, a1, b1);
            mma(a0, b0);
            if (t + 1 < nk) {
                // every read of tile t must have returned before any wave may overwrite its stage (tile t+3)
                if (t + 2 < nk) wait_stages(K1{}, true);
                else wait_stages(K0{}, true);
                __builtin_amdgcn_s_barrier();
                if (t + 3 < nk) stage(t + 3, smem + s_cur * STAGE);
                read_frags(smem + s_nxt * STAGE, 0, a0, b0);
            }
            mma(a1, b1);
            s_cur = s_nxt;
        }
    }

    // ---- epilogue: lane owns C[m = .. + fr][n = .. + fq*4 + 0..3]
    const bool has_bias = p.flags & MOLLY_GEMM_BIAS, has_res = p.flags & MOLLY_GEMM_RESIDUAL;
    const bool gelu = p.flags & MOLLY_GEMM_GELU, accum = p.flags & MOLLY_GEMM_ACCUMULATE;
    const bool out_f32 = p.flags & MOLLY_GEMM_OUT_F32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + fq * 4;
            if (n >= p.N) continue;                       // N % 4 == 0 is checked by the host
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (has_bias) {
                const u32x2 b = *reinterpret_cast<const u32x2*>(p.bias + n);
                v[0] += bflo(b[0]); v[1] += bfhi(b[0]); v[2] += bflo(b[1]); v[3] += bfhi(b[1]);
            }
            if (gelu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
            }
            if (has_res) {
                const u32x2 b = *reinterpret_cast<const u32x2*>(p.res + (size_t)m * p.ldres + n);
                v[0] += bflo(b[0]); v[1] += bfhi(b[0]); v[2] += bflo(b[1]); v[3] += bfhi(b[1]);
            }
            if (out_f32) {
                float* c = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n;
                if (accum) {
                    const f32x4 o = *reinterpret_cast<const f32x4*>(c);
                    v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
                }
                *reinterpret_cast<f32x4*>(c) = f32x4{v[0], v[1], v[2], v[3]};
            } else {
                bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n;
                if (accum) {
                    const u32x2 o = *reinterpret_cast<const u32x2*>(c);
                    v[0] += bflo(o[0]); v[1] += bfhi(o[0]); v[2] += bflo(o[1]); v[3] += bfhi(o[1]);
                }
                *reinterpret_cast<u32x2*>(c) = u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
            }
        }
    }
}


// ================================================================================================
// 256x256x64 tile, 8 waves (2 x 4, each 128x64), all 160 KiB of LDS = 10 half-tile slots of 16 KiB (A x3, B x2 buffers).
// Structure (after guide §5 "The 256² 8-phase template", own schedule):
//  * each K-tile is 4 PHASES = the 4 quadrants (64x32) of the wave's output; a phase = LOAD segment (ds_read of the
//    operand sub-tiles it needs + ONE half-tile LDS-DMA prefetch) | barrier | COMPUTE segment (16 MFMAs) | barrier;
//  * the two wave groups (waves 0-3 / 4-7, one of each per SIMD) are STAGGERED by one barrier, so one group's MFMA
//    segment always runs beside the other's LDS/DMA segment (matrix pipe beside memory pipe on every SIMD);
//  * half-tile prefetch order makes every restage >= 2 phases after the slot's last ds_read (WAR) and every first read
//    >= 1 phase after the counted s_waitcnt vmcnt that retires it, behind a barrier both groups have passed (RAW):
//      K-tile T:  P0 reads B(n0) A(m0), issues nothing | P1 reads B(n1), issues A0(T+2) | P2 reads A(m1), issues A1(T+2)
//                 P3 issues B0,B1(T+2), waits vmcnt(8) => K-tile T+1 landed   (every piece is >= 4 phases ahead of its use)
// Arithmetic intensity 128 flop/B of L2->LDS traffic (2x the 128² tile): the per-CU vector-memory path (64 B/clk) and
// the matrix pipe are no longer at a 1:1 ridge.
// ================================================================================================
// GRP: grouped launch — the work list concatenates the tiles of up to 16 problems (p.grp) that share K, the operand layouts
// and the epilogue flags; each keeps its own pointers, sizes and transposed-output choice (TO is ignored).  The weight-
// gradient GEMMs of one decoder layer are 64 + 128 + 192 + 384 = 768 tiles = three full rounds of 256 CUs: one launch, no
// split-K slabs, no reduce launches.
// The epilogue's lane-row regrouping of a register pair (x0, x1) = quads (j even, j odd) of one accumulator row: swap the
// register index with lane bit 5 (v_permlane32_swap), then with lane bit 4 (v_permlane16_swap).  Written as asm with both
// registers read-write: the clang builtins return the pair by value, and hipcc (ROCm 7.2) folded chains of them over vector
// elements into wrong code (3 of 4 columns of the residual path; the same family as the permlane32_swap(u, u) fold recorded in
// attention.hip).
__device__ __forceinline__ void regroup_rows(unsigned& x0, unsigned& x1) {
    // (hipcc pads no hazards inside an asm string.  What it pads around the builtins, on a test kernel: `s_nop 1` between a VALU
    // write of an operand and the swap that reads it — the second swap reads the first one's results — and nothing behind a swap)
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x0), "+v"(x1));
}

// (nontemporal epilogue stores, tried on the 8-byte pieces before the regrouping: 3x the fixed cost per tile — written through
// as partial lines)
//
// SKM: STREAM-K.  A grid whose tiles do not fill the 256 CUs in whole rounds (the decoder GEMMs of a B = 1 micro-batch: 288 tiles
// = 1.125 rounds, 384 = 1.5) is cut by WORK instead of by tile: all tiles' K-tiles form one sequence (tile-major), every block
// takes an equal contiguous share of it, so a block's share is [the tail of one tile][whole tiles][the head of another].  A
// whole tile runs the ordinary epilogue.  A PIECE writes its accumulators to one of the block's two slabs (fp32, register order,
// write-through) and adds its length to the tile's counter; the piece whose addition completes the tile — the last arriver,
// whichever that is: nobody ever waits — sums the slabs of all the tile's pieces in ascending-K order and runs the epilogue (see
// the block at 'stream-K: a PIECE of a tile' below).  Same numbers on every run: the cut and the order of the additions are
// functions of the grid alone.  Blocks with the same XCD label (blockIdx & 7) take neighbouring shares, so a tile's pieces and
// its operand panels meet in one L2.  The hand-off costs 30-36 us per launch on this chip, so the launcher (launch_cfg) takes
// stream-K only where its cost model says the alternatives waste more: long contractions on grids just past a whole round.
//
// DYN: DYNAMIC TILE FETCH.  The static walk (block b runs items b, b + G, ...) assumes that all G blocks start together and run
// at one speed.  Beside a collective that is false: RCCL's kernels hold CUs, the blocks that found none start a round late and
// the launch takes two rounds for what fits in 1.07 (tools/diag/gemm_beside_hog.py: -40 % with 16 CUs held).  Here a block
// DRAWS its items: one returning atomic add on the counter of its XCD label (8 counters: a label's blocks keep walking their
// own chunk of the tile order, so the L2 sharing of the static walk stays), issued by one lane five K-tiles before the
// current tile ends and handed to the other waves through LDS, so that the next tile is known when the rolling prefetch needs it
// (two K-tiles before the end).  The first item is drawn too (one exposed round trip per launch, ~2 us): a block that starts late
// takes what is left instead of finding its share untouched.  Every block of a label draws exactly one ticket past the end; the
// block that draws the LAST one (n + blocks - 1) puts the counter back to zero — after it, nobody in this launch reads it again.
// Results are identical to the static walk's (a tile is computed the same way whoever computes it).
//
// SE: STREAMING EPILOGUE WITH WHOLE-LINE STORES (round 6; the NT form).  Two findings behind it (profiles/r06_logs/):
//  * what a tile's 128 KB of output cost is a property of ONE CU's store path, not of the chip: tools/r06/store_diag/store_bench.hip — 8,600 cycles
//    per tile for the regrouped 16-byte stores of the fast16 epilogue below, whether 8 or 256 CUs store at that moment, because the swapped MFMA
//    operands leave the ROW index in the low four lane bits: sixteen consecutive lanes touch sixteen different 128-byte lines.  The same bytes
//    as instructions whose sixteen consecutive lanes cover ONE line: 5,500 cycles at 8 bytes per lane, 3,550 at 16;
//  * with no epilogue at all (timing-only build) the K = 2048 forward shapes run 9-12 % faster, and packing / regrouping inside the K loop's
//    load segments is free (ab_se_2.log): the time is the stores'.
// So here the MFMA operands are NOT swapped (lane (fr, fq) owns rows 4 fq + e of column fr of each 16 x 16 block) and the B half-tiles are staged
// with their rows interleaved (the LDS-DMA source address is per lane: free) — LDS row 16 j + f of a wave's 64 holds weight row 4 f + j — so a
// lane's four column blocks j are four CONSECUTIVE output columns: 8 bytes per lane and row, sixteen lanes = one whole 128-byte line, four lines
// per store instruction, and no lane swaps at all.  The four-phase schedule finishes the tile's upper row half (m0) with P1 of the LAST K-tile and
// the lower (m1) with P3, so the upper half is packed and stored in that K-tile's P2 load segment and the lower half in P1 of the NEXT tile's
// first K-tile, beside the partner group's MFMA segments; the K loop runs on into the next work item (no closing / opening barriers, the wave
// groups keep their stagger, the staging ring keeps rolling; a tile's first K-tile takes a zero C operand instead of zeroed accumulators).  The
// stores sit between LDS-DMA pieces in the in-order vmcnt queue, so the counted waits next to a boundary are vmcnt(24).
// Launched for grids of whole interior tiles (M, N multiples of 256), one K slice, >= 3 K-tiles, no epilogue flag; bit-identical to the
// plain instantiation (same products in the same order, same roundings).
template <bool AT, bool BT, bool TO = false, bool P2 = false, bool GRP = false, bool SKM = false, bool DYN = false, bool KX = false, int SE = 0>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmArgs p) {
    static_assert(!SE || (!AT && !BT && !TO && !P2 && !GRP && !SKM && !DYN && !KX), "the streaming epilogue exists on the plain four-phase NT kernel");
    static_assert(!KX || (!AT && !BT && !TO && !P2 && !GRP && !SKM && !DYN), "the K-extension exists on the plain four-phase NT kernel");
    static_assert(!SKM || (!P2 && !GRP), "stream-K exists on the four-phase, single-problem kernel");
    static_assert(!DYN || (!P2 && !GRP && !SKM), "the dynamic tile fetch exists on the four-phase, single-problem kernel");
    constexpr int BK = 64, HT = 128 * BK;             // half-tile elements (16 KiB)
    constexpr int NWI = P2 ? 4 : 8;                   // waves that stage one operand
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);   // [buf][A0|A1|B0|B1][HT]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- persistent tile loop: block b runs virtual ids b, b+G, b+2G, ... (G = gridDim.x, a multiple of 8 whenever there
    // is more than one round, so a block's XCD label v&7 never changes).  The NEXT work item's first two K-tiles are staged
    // during the current item's last two (rolling prefetch, below); without it (two-phase schedule, grouped launches, one-K-tile
    // slices) they are issued after the K loop, before the epilogue's stores.
    // One work item: the problem it belongs to (never changes outside a grouped launch), its tile and its K-slice
    struct Tile { int gi, m0, n0, split, kt0, nk; };       // gi = problem of a grouped launch (0 otherwise)
    // (kept small: the current and the next item's are live across the K loop)
    auto tC = [&](const Tile& t) { return GRP ? p.grp[t.gi].C : p.C; };
    auto tM = [&](const Tile& t) { return GRP ? p.grp[t.gi].M : p.M; };
    auto tN = [&](const Tile& t) { return GRP ? p.grp[t.gi].N : p.N; };
    auto tldc = [&](const Tile& t) { return GRP ? p.grp[t.gi].ldc : p.ldc; };
    auto tto = [&](const Tile& t) { return GRP ? p.grp[t.gi].trans_out != 0 : TO; };
    const int nwork = GRP ? p.grp[p.ngroup - 1].work0 + p.grp[p.ngroup - 1].tiles_m * p.grp[p.ngroup - 1].tiles_n
                          : p.tiles_m * p.tiles_n * p.splits;
    const int nk_main = (p.K + BK - 1) / BK;                       // K-tiles of (A, B)
    const int nk_all = nk_main + (KX ? p.k2 : 0);
    // stream-K share of this block, in units of p.sk K-tiles (the last unit of a tile also takes the K-tiles a division leaves over)
    const int sk_upt = SKM ? nk_all / p.sk : 1;                    // units per tile (>= 1)
    // units [u0, u1) of block `b`: label x = b & 7 owns the units [L0, L1) — whole tiles when sk_tile_aligned — and its blocks
    // j = b >> 3 cut that range evenly
    auto sk_range = [&](int b, int& u0, int& u1) {
        // (32-bit unsigned arithmetic: tiles x units x 8 stays far below 2^32 — the host refuses stream-K beyond 8192 tiles)
        const unsigned x = b & 7, j = b >> 3, nl = gridDim.x >> 3;            // (the grid is a multiple of 8)
        const unsigned ntile = p.tiles_m * p.tiles_n, upt = sk_upt;
        unsigned L0, L1;
        if (p.sk_tile_aligned) { L0 = (ntile * x / 8u) * upt; L1 = (ntile * (x + 1u) / 8u) * upt; }
        else { L0 = ntile * upt * x / 8u; L1 = ntile * upt * (x + 1u) / 8u; }
        u0 = (int)(L0 + (L1 - L0) * j / nl);
        u1 = (int)(L0 + (L1 - L0) * (j + 1u) / nl);
    };
    int sk_u0 = 0, sk_u1 = 0;
    if constexpr (SKM) sk_range(blockIdx.x, sk_u0, sk_u1);
    // this block's work items: stream-K — its pieces, highest tile first; otherwise the virtual ids b, b + G, b + 2G, ...
    // (dynamic fetch: the items of this block's XCD label, indices 0 .. nitems - 1 of its chunk, drawn one at a time)
    const int nitems = SKM ? (sk_u1 > sk_u0 ? (sk_u1 - 1) / sk_upt - sk_u0 / sk_upt + 1 : 0)
                     : DYN ? (nwork >> 3) + ((int)(blockIdx.x & 7) < (nwork & 7) ? 1 : 0)
                           : (nwork - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (nitems <= 0) return;                                       // (block-uniform: more blocks than pieces of work)
    unsigned* const dcnt = DYN ? p.dyn_cnt + (blockIdx.x & 7) * 16 : nullptr;
    const int dyn_last = nitems + (int)(gridDim.x >> 3) - 1;       // the last ticket any block of this label can draw
    auto decode = [&](int it) -> Tile {
        Tile t;
        t.gi = 0;
        int ctm = p.tiles_m, ctn = p.tiles_n;
        int work;
        if constexpr (SKM) {
            work = (sk_u1 - 1) / sk_upt - it;                      // tile index (in the XCD-chunked order the labels already give)
        } else {
            const int v = DYN ? (int)(blockIdx.x & 7) + 8 * it : (int)(blockIdx.x + it * gridDim.x);
            const int q = nwork >> 3, r = nwork & 7, xcd = v & 7;
            work = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
        }
        if constexpr (GRP) {
            int gi = 0;
#pragma unroll
            for (int i = 1; i < 16; ++i)
                if (i < p.ngroup && work >= p.grp[i].work0) gi = i;
            const GemmArgs::Group& G = p.grp[gi];
            t.gi = gi;
            ctm = G.tiles_m; ctn = G.tiles_n;
            work -= G.work0;
        }
        const int ntile = ctm * ctn;
        // K-slice is the SLOWEST index: the work items that run side by side on one XCD then belong to one slice and
        // keep sharing operand panels through its L2 (slices of one tile share nothing: they read different K ranges)
        // (this runs twice per tile in every wave, once of them inside the K loop: without slices — every launch of the step's big
        // GEMMs — it is two unsigned 32-bit divisions; the general form below adds one more and two 64-bit ones, ~0.7 us per tile)
        const bool one_slice = SKM || p.splits == 1;
        t.split = one_slice ? 0 : work / ntile;
        const unsigned swz = (unsigned)(work - t.split * ntile);
        const int GROUP_M = p.group_m;
        const unsigned per_group = (unsigned)(GROUP_M * ctn);
        const int grp = (int)(swz / per_group);
        const unsigned in_grp = swz - (unsigned)grp * per_group;
        const int first_m = grp * GROUP_M;
        const unsigned gsz = (unsigned)min(ctm - first_m, GROUP_M);
        const unsigned nq = in_grp / gsz;
        t.m0 = (first_m + (int)(in_grp - nq * gsz)) * 256;
        t.n0 = (int)nq * 256;
        if constexpr (SKM) {
            // the piece of tile `work` inside [sk_u0, sk_u1); split = 0: the whole tile, 1: a piece (slab + ticket); gi = the tile
            const int ua = max(sk_u0, work * sk_upt) - work * sk_upt, ub = min(sk_u1, (work + 1) * sk_upt) - work * sk_upt;
            t.kt0 = ua * p.sk;
            t.nk = (ub == sk_upt ? nk_all : ub * p.sk) - t.kt0;
            t.split = t.nk == nk_all ? 0 : 1;
            t.gi = work;
        } else if (one_slice) {
            t.kt0 = 0;
            t.nk = nk_all;
        } else {
            t.kt0 = (int)((long)nk_all * t.split / p.splits);
            t.nk = (int)((long)nk_all * (t.split + 1) / p.splits) - t.kt0;      // K-tiles of this slice
        }
        return t;
    };
    int vcur = 0;                                                  // index of the current item in this block's list
    if constexpr (DYN) {
        // the first ticket, through LDS (A slot 2: the prologue stages slots 0 and 1 only)
        if (tid == 0) lds_put(smem + 4 * HT, __hip_atomic_fetch_add(dcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        __syncthreads();
        vcur = __builtin_amdgcn_readfirstlane((int)lds_get(smem + 4 * HT));
        if (vcur >= nitems) {
            if (tid == 0 && vcur == dyn_last) __hip_atomic_store(dcnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
    Tile cur = decode(vcur);
#ifdef MOLLY_GEMM_SE_SKEW_TICKS
    // timing-only diagnostic (tools/build_variant.py): the blocks start in eight groups MOLLY_GEMM_SE_SKEW_TICKS x 10 ns apart, so that their tile
    // boundaries — and the 128 KB of stores each brings — no longer fall on one moment chip-wide
    if constexpr (SE != 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long dl = (unsigned long long)((blockIdx.x >> 3) & 7) * MOLLY_GEMM_SE_SKEW_TICKS;
        while (__builtin_amdgcn_s_memrealtime() - t0 < dl) __builtin_amdgcn_s_sleep(16);
    }
#endif

    f32x4 acc[8][4];
    // LDS = 10 half-tile slots (all 160 KiB): A triple-buffered [3][A0|A1], then B double-buffered [2][B0|B1]
    constexpr int SPER = 128 / 8 / NWI;               // LDS-DMA pieces per wave per half-tile (2; two-phase schedule: 4)
    const bool swiglu_b = !AT && !BT && !TO && !GRP && (p.flags & MOLLY_GEMM_SWIGLU);
    // The STAGING STREAM: the work item whose K-tiles are being staged (under the rolling prefetch it runs two K-tiles ahead of
    // the one being computed and switches to the next work item in the middle of the K loop): its operands (launch constants
    // outside a grouped launch), the slice-local index of the K-tile it stages next, per half-tile the per-lane byte offsets
    // (off0..3: A0, A1, B0, B1) and a running wave-uniform source pointer that advances by a constant per K-tile.
    const bf16_t* s_A = p.A; const bf16_t* s_B = p.B;
    int s_M = p.M, s_N = p.N, s_lda = p.lda, s_ldb = p.ldb, s_m0 = 0, s_n0 = 0, s_kt0 = 0, s_ktl = 0;
    const char* sp0 = nullptr; const char* sp1 = nullptr; const char* sp2 = nullptr; const char* sp3 = nullptr;
    unsigned off0[SPER], off1[SPER], off2[SPER], off3[SPER];
    int stepA = AT ? BK * p.lda * 2 : BK * 2, stepB = BT ? BK * p.ldb * 2 : BK * 2;            // bytes per K-tile
    const bool k_full = p.K % BK == 0;                // a ragged last K-tile: both operands k-major only (host check)
    auto set_stream = [&](const Tile& t) {
        s_m0 = t.m0; s_n0 = t.n0; s_kt0 = t.kt0; s_ktl = 0;
        if constexpr (GRP) {                          // per-problem operands and strides
            const GemmArgs::Group& G = p.grp[t.gi];
            s_A = G.A; s_B = G.B; s_M = G.M; s_N = G.N; s_lda = G.lda; s_ldb = G.ldb;
            stepA = AT ? BK * s_lda * 2 : BK * 2; stepB = BT ? BK * s_ldb * 2 : BK * 2;
        }
        const int wi = P2 ? wc : wave;
        const long ka = AT ? (long)t.kt0 * BK * s_lda : (long)t.kt0 * BK, kb = BT ? (long)t.kt0 * BK * s_ldb : (long)t.kt0 * BK;
        int a0, a1, b0, b1;                           // row (k-contiguous) / column (k-major) the base points at
        if (AT) {
            a0 = stage_offsets_km<128, NWI, BK>(s_lda, t.m0, s_M, wi, lane, off0);
            a1 = stage_offsets_km<128, NWI, BK>(s_lda, t.m0 + 128, s_M, wi, lane, off1);
        } else {
            a0 = stage_offsets_kc<128, NWI, BK>(s_lda, t.m0, s_M, wi, lane, 0, off0);
            a1 = stage_offsets_kc<128, NWI, BK>(s_lda, t.m0 + 128, s_M, wi, lane, 0, off1);
        }
        if (BT) {
            b0 = stage_offsets_km<128, NWI, BK>(s_ldb, t.n0, s_N, wi, lane, off2);
            b1 = stage_offsets_km<128, NWI, BK>(s_ldb, t.n0 + 128, s_N, wi, lane, off3);
        } else if (swiglu_b) {
            b0 = stage_offsets_kc<128, NWI, BK>(s_ldb, t.n0 >> 1, s_N, wi, lane, s_N >> 1, off2);
            b1 = stage_offsets_kc<128, NWI, BK>(s_ldb, (t.n0 >> 1) + 64, s_N, wi, lane, s_N >> 1, off3);
        } else {
            b0 = stage_offsets_kc<128, NWI, BK>(s_ldb, t.n0, s_N, wi, lane, SE ? -1 : 0, off2);
            b1 = stage_offsets_kc<128, NWI, BK>(s_ldb, t.n0 + 128, s_N, wi, lane, SE ? -1 : 0, off3);
        }
        sp0 = reinterpret_cast<const char*>(s_A + ka + (AT ? (long)a0 : (long)a0 * s_lda));
        sp1 = reinterpret_cast<const char*>(s_A + ka + (AT ? (long)a1 : (long)a1 * s_lda));
        sp2 = reinterpret_cast<const char*>(s_B + kb + (BT ? (long)b0 : (long)b0 * s_ldb));
        sp3 = reinterpret_cast<const char*>(s_B + kb + (BT ? (long)b1 : (long)b1 * s_ldb));
    };
    // stage half-tile `which` (0 = A0, 1 = A1, 2 = B0, 3 = B1) of the stream's next K-tile into slots (ab, bb); the K-tile is
    // complete after which == 3
    // K-extension: the stream's K-tile nk_main is the first of the second operand pair — same tile rows / columns, other matrices and row strides:
    // the per-lane offsets and the four running pointers are re-derived once (splits == 1: slice-local index = K-tile index)
    auto ext_stream = [&]() {
        const int a0 = stage_offsets_kc<128, NWI, BK>(p.lda2, s_m0, s_M, wave, lane, 0, off0);
        const int a1 = stage_offsets_kc<128, NWI, BK>(p.lda2, s_m0 + 128, s_M, wave, lane, 0, off1);
        int b0, b1;
        if (swiglu_b) {
            b0 = stage_offsets_kc<128, NWI, BK>(p.ldb2, s_n0 >> 1, s_N, wave, lane, s_N >> 1, off2);
            b1 = stage_offsets_kc<128, NWI, BK>(p.ldb2, (s_n0 >> 1) + 64, s_N, wave, lane, s_N >> 1, off3);
        } else {
            b0 = stage_offsets_kc<128, NWI, BK>(p.ldb2, s_n0, s_N, wave, lane, 0, off2);
            b1 = stage_offsets_kc<128, NWI, BK>(p.ldb2, s_n0 + 128, s_N, wave, lane, 0, off3);
        }
        sp0 = reinterpret_cast<const char*>(p.A2 + (long)a0 * p.lda2);
        sp1 = reinterpret_cast<const char*>(p.A2 + (long)a1 * p.lda2);
        sp2 = reinterpret_cast<const char*>(p.B2 + (long)b0 * p.ldb2);
        sp3 = reinterpret_cast<const char*>(p.B2 + (long)b1 * p.ldb2);
    };
    auto stage_stream = [&](int which, int ab, int bb) {
        if constexpr (KX) {
            if (which == 0 && s_ktl == nk_main) ext_stream();
        }
        if (P2 && ((which < 2) != (wr == 0))) { if (which == 3) ++s_ktl; return; }
        const int wi = P2 ? wc : wave;
        bf16_t* dst = which < 2 ? smem + (ab * 2 + which) * HT : smem + (6 + bb * 2 + (which - 2)) * HT;
        bool done = false;
        if constexpr (AT && BT) {
            // a ragged last K-tile (contraction length not a multiple of 64): rows past K are read as exact zeros, lane by lane
            if (!k_full && (s_kt0 + s_ktl + 1) * BK > p.K) {
                const int kt = s_kt0 + s_ktl;
                if (which < 2) stage_km<128, NWI, BK, true>(s_A, s_lda, s_m0 + which * 128, s_M, kt * BK, p.K, p.zeros, dst, wi, lane);
                else stage_km<128, NWI, BK, true>(s_B, s_ldb, s_n0 + (which - 2) * 128, s_N, kt * BK, p.K, p.zeros, dst, wi, lane);
                done = true;
            }
        }
        if (!done) {
            if (which == 0) stage_pre<SPER, true>(sp0, off0, dst, wi);
            else if (which == 1) stage_pre<SPER, true>(sp1, off1, dst, wi);
            else if (which == 2) stage_pre<SPER, true>(sp2, off2, dst, wi);
            else stage_pre<SPER, true>(sp3, off3, dst, wi);
        }
        if (which == 0) sp0 += stepA; else if (which == 1) sp1 += stepA; else if (which == 2) sp2 += stepB; else sp3 += stepB;
        if (which == 3) ++s_ktl;
    };
    const int fr = lane & 15, fq = lane >> 4;
    auto readA = [&](int abuf, int mh, bf16x8 (&af)[2][4]) {
        const bf16_t* t = smem + (abuf * 2 + wr) * HT;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[kk][i] = AT ? frag_km<128>(t, kk, mh * 64 + i * 16, lane) : frag_kc<BK>(t, mh * 64 + i * 16 + fr, kk * 4 + fq);
    };
    auto readB = [&](int bbuf, int nh, bf16x8 (&bfr)[2][2]) {
        const bf16_t* t = smem + (6 + bbuf * 2 + (wc >> 1)) * HT;
        const int c0 = (wc & 1) * 64 + nh * 32;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                bfr[kk][j] = BT ? frag_km<128>(t, kk, c0 + j * 16, lane) : frag_kc<BK>(t, c0 + j * 16 + fr, kk * 4 + fq);
    };
#define MMA_QUAD(MH, NH, AF, BF)                                                                            \
    do {                                                                                                    \
        __builtin_amdgcn_s_setprio(1);                                                                      \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                    \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                       \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                       \
            acc[(MH) * 4 + i][(NH) * 2 + j] = TOQ                                                           \
                ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(AF[kk][i], BF[kk][j], acc[(MH) * 4 + i][(NH) * 2 + j], 0, 0, 0) \
                : __builtin_amdgcn_mfma_f32_16x16x32_bf16(BF[kk][j], AF[kk][i], acc[(MH) * 4 + i][(NH) * 2 + j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                      \
    } while (0)
#define SEG_BARRIER()                          \
    do {                                       \
        __builtin_amdgcn_sched_barrier(0);     \
        __builtin_amdgcn_s_barrier();          \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)

    // ---- prologue of the first tile: K-tiles 0 and 1 issued
    set_stream(cur);
    stage_stream(0, 0, 0); stage_stream(1, 0, 0); stage_stream(2, 0, 0); stage_stream(3, 0, 0);
    if (cur.nk > 1) { stage_stream(0, 1, 1); stage_stream(1, 1, 1); stage_stream(2, 1, 1); stage_stream(3, 1, 1); }
    // stores the previous epilogue left behind its loads, per lane, when that number is known (16: the fast paths' 16-byte
    // stores; 32: plain 8-byte quads; 48: the SwiGLU-fused gate, up and activation quads); 0 = unknown: conservative waits
    int pend_stores = 0;
    // s_waitcnt vmcnt(n) for a run-time n (the instruction takes an immediate); n is a multiple of 8 up to 56, anything else waits
    // for everything
    auto wait_vm = [&](int n) {
        if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (n == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (n == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (n == 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        else if (n == 40) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
        else if (n == 48) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
        else if (n == 56) asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // ROLLING prefetch (four-phase schedule, every slice >= 2 K-tiles): the staging pipeline does not drain at a tile boundary.
    // K-tile T+2 of the loop is the NEXT work item's K-tile T+2-nk once T+2 >= nk, staged into the slot the ring would use
    // anyway (abuf / bbuf keep rotating across tiles), so the next tile's first two K-tiles are in flight during this tile's last
    // two and its K-tile 0 has landed — waited for by the loop's own counted vmcnt — before the epilogue starts.  Before: both
    // were issued after the loop, every CU at the same moment (a 32 MB burst), and their latency was hidden only by the
    // ~1.5 us epilogue: ~9 us of fixed cost per 256x256 tile = 6 K-tiles' worth, 16 % of a K = 2048 tile
    // (tools/gemm_diag/run_kscan.py: time = fixed + per-K-tile * nk; run_tilestamp.py).
    // (not in a grouped launch: its slices are 256 K-tiles long; with the per-problem operands re-derived inside the K loop hipcc
    // no longer keeps the LDS-DMA's scalar base in SGPRs, and with the tile state forced uniform by readfirstlane the launch
    // measured 1,190 -> 1,209 us: the boundary saved less than the longer loop cost)
    const bool roll = !P2 && !GRP && nk_all / p.splits >= 2;
    bool landed0 = false;                     // K-tile 0 of the tile about to start has already been waited for
    int abuf = 0, bbuf = 0;                   // LDS slots of the current K-tile (A: 0..2, B: 0..1)

    unsigned tk = 0, tk_here = 0;             // dynamic fetch: the ticket in flight (lane 0 of wave 0) and its copy once it has arrived
    bool se_first = true;                     // SE: no tile of this block has been computed yet
    char* se_pc = nullptr;                    // SE: the previous tile's base (its lower row half is stored in this tile's first K-tile)
    // SE: the lane's byte offset inside any tile: row wr * 128 + 4 fq, column wc * 64 + 4 fr (rows x ldc stay far below 2^31 bytes: host check)
    const unsigned se_loff = SE ? (unsigned)(((wr * 128 + (lane >> 4) * 4) * p.ldc + wc * 64 + (lane & 15) * 4) * 2) : 0u;
    for (;;) {
    // (dynamic fetch: unknown until the ticket arrives at K-tile nk - 3 — both are only read from K-tile nk - 2 on)
    int vnext = vcur + 1;
    bool more = DYN ? false : vnext < nitems;
    const int nk = cur.nk;
    int t_stage_end = (roll && more) ? nk : nk - 2;             // loop iterations T < t_stage_end stage a K-tile (T + 2)
    const bool cto = tto(cur);
    // (SE: a tile after the first starts with its K-tile 0 landed behind the previous loop's last counted wait and inside the barriers of the loop it continues,
    // and its accumulators are never zeroed: the MFMAs of a tile's first K-tile take a zero C operand)
    if constexpr (!SE) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // wait for K-tile 0 only.  Outstanding, oldest first: [K-tile 0: 8][K-tile 1: 8 if nk > 1][previous epilogue stores].
    // vmcnt counts loads and stores together in issue order, so with exactly 32 stores behind the loads the counted wait
    // lets all of them (and K-tile 1) stay in flight; any other epilogue falls back to a conservative count.
    if (!landed0) wait_vm((nk > 1 ? 8 : 0) + pend_stores);
    // the first counted wait of a tile sits behind the previous epilogue's stores: [K-tile 1: 8][stores]
    // [K-tile 2: 8] — K-tile 1 has landed once all but the stores and K-tile 2's pieces are done
    const int first_wait = 8 + pend_stores;      // (SE: the counted waits are written into ktile_se)
    if (!SE || se_first) {
    SEG_BARRIER();
    if (wr == 1) SEG_BARRIER();               // stagger: group 1 runs one segment behind group 0
    }
    // SE: this tile's output as the lane sees it: rows m0 + wr * 128 + 4 fq (+ e, + 16 i), columns n0 + wc * 64 + 4 fr .. + 3
    // (a wave-uniform tile base + one per-lane byte offset that never changes: the stores take their base from scalar registers)
    char* se_c = nullptr;
    if constexpr (SE != 0) se_c = reinterpret_cast<char*>(p.C) + ((size_t)cur.m0 * p.ldc + cur.n0) * 2;
    // one finished row half (MH) of the wave's 128 x 64: 16 stores of 8 bytes per lane, each four whole 128-byte lines
    auto se_half = [&](auto mh_tag, char* cb) {
        constexpr int MH = decltype(mh_tag)::value;
        {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int i = MH * 4 + ii;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const u32x2 d = u32x2{pack_bf2(acc[i][0][e], acc[i][1][e]), pack_bf2(acc[i][2][e], acc[i][3][e])};
#if MOLLY_GEMM_SE_FORM == 2
                    asm volatile("" :: "v"(d));
#else
                    *reinterpret_cast<u32x2*>(cb + (size_t)(i * 16 + e) * p.ldc * 2 + se_loff) = d;
#endif
                }
            }
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // LDS-DMA schedule (one K-tile = 4 half-tiles = 8 instructions per wave), every piece >= 4 phases ahead of its use:
    //   P0 (most operand reads) issues nothing | P1 -> A0(T+2) | P2 -> A1(T+2) | P3 (no reads) -> B0(T+2), B1(T+2)
    //   A lives in 3 slots (abuf rotates 0,1,2 and keeps rotating across tiles): slot of T+2 = slot of T-1, last read at P2(T-1),
    //   restaged at P1(T): 3 phases later;  B lives in 2 slots (bbuf toggles): slots die after P1(T), restaged at P3(T).
    // The only in-loop wait is P3's counted vmcnt(8): everything older than K-tile T+2's eight pieces — i.e. all of
    // K-tile T+1 — has landed; its first ds_read happens in the next phase, behind a barrier both groups have passed.
    bf16x8 af[2][4], b0[2][2], b1[2][2];
    // staging of loop K-tile T+2 (this work item's, or under the rolling prefetch the next one's) into the slots two ahead
    auto stage2 = [&](int T, int which) {
        const int a2 = abuf == 0 ? 2 : abuf - 1;                 // (abuf + 2) % 3; B: same parity as T
        if (T < t_stage_end) stage_stream(which, a2, bbuf);
    };
    if constexpr (P2) {
        auto kloop = [&](auto to_tag) {
        constexpr bool TOQ = decltype(to_tag)::value;
        // TWO phases per K-tile (32 MFMAs per compute segment, 4 barriers per K-tile instead of 8):
        //   LA: read B(n0), B(n1), A(m0)                              | CA: quadrants (m0,n0), (m0,n1)
        //   LB: read A(m1); stage own operand of K-tile T+2; counted  | CB: quadrants (m1,n1), (m1,n0)
        //       wait -> own pieces of T+1 have landed
        // Group 0 stages A (slot (T+2)%3 = slot of T-1: its last reads, group 1's LB(T-1), returned two barriers ago), group 1
        // stages B (slot of T: last read in its own LA(T), returned before its CA(T), one barrier ago; group 0's a segment
        // earlier still).  Each piece is waited for by the wave that issued it one K-tile later, in front of a barrier every
        // reader passes before its first read.  Measured against the 4-phase schedule in one process (tools/bench_gemm.py
        // --sched): +3-4 % on the forms whose B operand is k-major (dgrad, wgrad; +13 % on TN), -1...-6 % on NT — the
        // launcher picks per form.  (Staging B from group 1's LA — more flight time — was slower and races with group 0's
        // reads of that slot, which only return during the same segment.)
        for (int T = 0; T < nk; ++T) {
            readB(bbuf, 0, b0);
            readB(bbuf, 1, b1);
            readA(abuf, 0, af);
            SEG_BARRIER();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            MMA_QUAD(0, 0, af, b0);
            MMA_QUAD(0, 1, af, b1);
            SEG_BARRIER();
            readA(abuf, 1, af);
            if (T + 2 < nk) {
                stage2(T, 0); stage2(T, 1); stage2(T, 2); stage2(T, 3);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            SEG_BARRIER();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            MMA_QUAD(1, 1, af, b1);
            MMA_QUAD(1, 0, af, b0);
            SEG_BARRIER();
            abuf = abuf == 2 ? 0 : abuf + 1;
            bbuf ^= 1;
        }
        };
        if constexpr (GRP) {
            if (cto) kloop(std::true_type{});
            else kloop(std::false_type{});
        } else {
            kloop(std::integral_constant<bool, TO>{});
        }
    } else {
    auto kloop4 = [&](auto to_tag) {
    constexpr bool TOQ = decltype(to_tag)::value;
    for (int T = 0; T < nk; ++T) {
        // from here on the staged K-tiles are the next work item's (decoded again at the tile's end: a Tile kept live across the
        // K loop costs more scalar registers than decoding twice)
        if (T + 2 == nk && t_stage_end == nk) set_stream(decode(vnext));
        // ---- P0: quadrant (m0,n0)
        readB(bbuf, 0, b0);
        readA(abuf, 0, af);
        unsigned mailv = 0;
        if constexpr (DYN) {
            // the ticket wave 0 left in the A1 half of the slot K-tile T-1 lived in (dead from its last read, group 1's P2(T-1),
            // until P2(T) restages it)
            if (T == nk - 3)
                mailv = lds_get(smem + ((abuf == 0 ? 2 : abuf - 1) * 2 + 1) * HT);
        }
        SEG_BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (DYN) {
            if (T == nk - 3) {
                vnext = __builtin_amdgcn_readfirstlane((int)mailv);
                more = vnext < nitems;
                if (more) t_stage_end = nk;
            }
        }
        MMA_QUAD(0, 0, af, b0);
        SEG_BARRIER();
        // ---- P1: quadrant (m0,n1)
        readB(bbuf, 1, b1);
        if constexpr (DYN) {
            // the ticket's wait, placed where nothing younger than it is in flight (hipcc waits vmcnt(0) for an ordinary result: it
            // does not count the LDS-DMA issued from asm) — K-tile T+1's pieces, the only things older, are due a phase later anyway
            // (copied by an asm move: a register hipcc never sees a load pending on — at the later store it would otherwise wait
            // again, for the path that did not come through here)
            if (T == nk - 4 && tid == 0) asm volatile("v_mov_b32 %0, %1" : "=v"(tk_here) : "v"(tk));
        }
        stage2(T, 0);
        SEG_BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        MMA_QUAD(0, 1, af, b1);
        SEG_BARRIER();
        // ---- P2: quadrant (m1,n1)
        readA(abuf, 1, af);
        stage2(T, 1);
        SEG_BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        MMA_QUAD(1, 1, af, b1);
        SEG_BARRIER();
        // ---- P3: quadrant (m1,n0); retire K-tile T+1
        if (T < t_stage_end) {
            stage2(T, 2); stage2(T, 3);
            if (T == 0 && first_wait != 8) wait_vm(first_wait);
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (DYN) {
            // draw the next ticket: the youngest vector-memory operation, behind the counted wait
            if (T == nk - 5 && tid == 0) tk = __hip_atomic_fetch_add(dcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        SEG_BARRIER();
        MMA_QUAD(1, 0, af, b0);
        if constexpr (DYN) {
            if (T == nk - 4) {
                // into the A1 half of THIS K-tile's slot: group 1's reads of it returned before its P2 compute segment, one barrier ago
                if (tid == 0) lds_put(smem + (abuf * 2 + 1) * HT, tk_here);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        }
        SEG_BARRIER();
        abuf = abuf == 2 ? 0 : abuf + 1;
        bbuf ^= 1;
    }
    };
    // ---- SE: the same four phases, written out three times — a tile's FIRST K-tile (MODE 0: its MFMAs take a zero C operand; the previous tile's
    // lower row half leaves in P1), its MIDDLE K-tiles (MODE 1) and its LAST (MODE 2: the upper row half leaves in P2).  Explicit bodies
    // instead of run-time tests inside one loop: with the accumulators written under a condition hipcc kept two homes for them and moved quadrants
    // through scratch memory between MFMAs.
#define MMA_QUAD0(MH, NH, AF, BF)                                                                           \
    do {                                                                                                    \
        __builtin_amdgcn_s_setprio(1);                                                                      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                       \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                       \
            acc[(MH) * 4 + i][(NH) * 2 + j] =                                                               \
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(AF[0][i], BF[0][j], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0); \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                       \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                       \
            acc[(MH) * 4 + i][(NH) * 2 + j] =                                                               \
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(AF[1][i], BF[1][j], acc[(MH) * 4 + i][(NH) * 2 + j], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                      \
    } while (0)
    auto ktile_se = [&](auto mode_tag, int T) {
        constexpr int MODE = decltype(mode_tag)::value;
        constexpr bool TOQ = true;                 // un-swapped operands: the lane owns four rows of one column per 16 x 16 block
        if (MODE == 1 && T + 2 == nk && t_stage_end == nk) set_stream(decode(vnext));
        // ---- P0: quadrant (m0,n0)
        readB(bbuf, 0, b0);
        readA(abuf, 0, af);
        SEG_BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE == 0) MMA_QUAD0(0, 0, af, b0); else MMA_QUAD(0, 0, af, b0);
        SEG_BARRIER();
        // ---- P1: quadrant (m0,n1)
        readB(bbuf, 1, b1);
        stage2(T, 0);
        if constexpr (MODE == 0) { if (!se_first) se_half(I1{}, se_pc); }      // the PREVIOUS tile's lower row half
        SEG_BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE == 0) MMA_QUAD0(0, 1, af, b1); else MMA_QUAD(0, 1, af, b1);
        SEG_BARRIER();
        // ---- P2: quadrant (m1,n1)
        readA(abuf, 1, af);
        stage2(T, 1);
        if constexpr (MODE == 2) se_half(I0{}, se_c);                            // rows m0: final since P1
        SEG_BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE == 0) MMA_QUAD0(1, 1, af, b1); else MMA_QUAD(1, 1, af, b1);
        SEG_BARRIER();
        // ---- P3: quadrant (m1,n0); retire K-tile T+1
        if (T < t_stage_end) {
            stage2(T, 2); stage2(T, 3);
            // MODE 0 behind a streamed tile: younger than K-tile 1's last piece are the sixteen stores of the previous tile's lower half (this K-tile's
            // P1) and K-tile 2's eight pieces; MODE 2: the sixteen stores of the upper half sit between the eight pieces staged in this iteration
            if ((MODE == 0 && !se_first) || MODE == 2) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        SEG_BARRIER();
        if constexpr (MODE == 0) MMA_QUAD0(1, 0, af, b0); else MMA_QUAD(1, 0, af, b0);
        SEG_BARRIER();
        abuf = abuf == 2 ? 0 : abuf + 1;
        bbuf ^= 1;
    };
    if constexpr (SE) {
        ktile_se(std::integral_constant<int, 0>{}, 0);
        for (int T = 1; T < nk - 1; ++T) ktile_se(std::integral_constant<int, 1>{}, T);
        ktile_se(std::integral_constant<int, 2>{}, nk - 1);
    } else if constexpr (GRP) {               // grouped launch: the output orientation is a per-problem property
        if (cto) kloop4(std::true_type{});
        else kloop4(std::false_type{});
    } else {
        kloop4(std::integral_constant<bool, TO>{});
    }
    }
    if constexpr (SE) {
        if (more) {
            // the loop continues into the next work item: Q(m1,n0) leaves in its first K-tile
            se_pc = se_c;
            se_first = false;
            landed0 = true;
            vcur = vnext;
            cur = decode(vnext);
            continue;
        }
        if (wr == 0) SEG_BARRIER();
        se_half(I1{}, se_c);
        break;
    }
    if (wr == 0) SEG_BARRIER();               // balance the stagger barrier (every LDS read of this tile has returned)

    // this tile's coordinates for the epilogue.  Without the rolling prefetch (two-phase schedule, one-K-tile slices) the next
    // tile's first two K-tiles are issued here, BEFORE the stores, into slots 0 and 1.
    const int em0 = cur.m0, en0 = cur.n0, esplit = cur.split;
    void* const eC = tC(cur);
    const int eM = tM(cur), eN = tN(cur), eldc = tldc(cur);
    const bool eto = cto;
    Tile nxt = cur;
    if (more) nxt = decode(vnext);
    if (more && !roll) {
        abuf = 0; bbuf = 0;
        set_stream(nxt);
        stage_stream(0, 0, 0); stage_stream(1, 0, 0); stage_stream(2, 0, 0); stage_stream(3, 0, 0);
        if (nxt.nk > 1) { stage_stream(0, 1, 1); stage_stream(1, 1, 1); stage_stream(2, 1, 1); stage_stream(3, 1, 1); }
    }
    landed0 = roll && more;
    // ---- the generic store of one accumulator quad (bias -> GELU -> residual -> accumulate, fp32 or bf16 output), as a function of
    // the quad's four values: used by the general epilogue on the accumulators and by stream-K's reducer on summed slab values.
    // Normal orientation: the lane owns C[m = .. + fr][n = .. + fq*4 + 0..3]
    auto store_quad = [&](int i, int j, float (&v)[4]) {
        const int m = em0 + wr * 128 + i * 16 + fr, n = en0 + wc * 64 + j * 16 + fq * 4;
        if (m >= eM || n >= eN) return;
        if (p.flags & MOLLY_GEMM_BIAS) {
            const u32x2 b = *reinterpret_cast<const u32x2*>(p.bias + n);
            v[0] += bflo(b[0]); v[1] += bfhi(b[0]); v[2] += bflo(b[1]); v[3] += bfhi(b[1]);
        }
        if (p.flags & MOLLY_GEMM_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
        }
        if (p.flags & MOLLY_GEMM_RESIDUAL) {
            const u32x2 b = *reinterpret_cast<const u32x2*>(p.res + (size_t)m * p.ldres + n);
            v[0] += bflo(b[0]); v[1] += bfhi(b[0]); v[2] += bflo(b[1]); v[3] += bfhi(b[1]);
        }
        if (p.flags & MOLLY_GEMM_OUT_F32) {
            float* c = reinterpret_cast<float*>(eC) + (size_t)m * eldc + n;
            if (p.flags & MOLLY_GEMM_ACCUMULATE) {
                const f32x4 o = *reinterpret_cast<const f32x4*>(c);
                v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
            }
            *reinterpret_cast<f32x4*>(c) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
            bf16_t* c = reinterpret_cast<bf16_t*>(eC) + (size_t)m * eldc + n;
            if (p.flags & MOLLY_GEMM_ACCUMULATE) {
                const u32x2 o = *reinterpret_cast<const u32x2*>(c);
                v[0] += bflo(o[0]); v[1] += bfhi(o[0]); v[2] += bflo(o[1]); v[3] += bfhi(o[1]);
            }
            *reinterpret_cast<u32x2*>(c) = u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        }
    };
    // Transposed output: operands were passed un-swapped, so the lane owns C[m = .. + fq*4 + 0..3][n = .. + fr]; C^T is stored as
    // [N][M] (ld = ldc), i.e. 4 consecutive m of one n -> 8-byte bf16 / 16-byte fp32 stores (no bias / GELU / residual: host check)
    auto store_quad_t = [&](int i, int j, float (&v)[4]) {
        const int m = em0 + wr * 128 + i * 16 + fq * 4, n = en0 + wc * 64 + j * 16 + fr;
        if (m >= eM || n >= eN) return;                                  // M % 4 == 0 checked by the host
        if (p.flags & MOLLY_GEMM_OUT_F32) {
            float* c = reinterpret_cast<float*>(eC) + (size_t)n * eldc + m;
            if (p.flags & MOLLY_GEMM_ACCUMULATE) {
                const f32x4 o = *reinterpret_cast<const f32x4*>(c);
                v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
            }
            *reinterpret_cast<f32x4*>(c) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
            bf16_t* c = reinterpret_cast<bf16_t*>(eC) + (size_t)n * eldc + m;
            if (p.flags & MOLLY_GEMM_ACCUMULATE) {
                const u32x2 o = *reinterpret_cast<const u32x2*>(c);
                v[0] += bflo(o[0]); v[1] += bfhi(o[0]); v[2] += bflo(o[1]); v[3] += bfhi(o[1]);
            }
            *reinterpret_cast<u32x2*>(c) = u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        }
    };
    // ---- stream-K: a PIECE of a tile (esplit == 1).  Its accumulators go to the block's slab as they stand (register order, 16
    // bytes per lane, write-through: no release fence needed — guide 6 G16 R1, 'publish-large'); every wave drains its stores, a
    // barrier, then ONE lane adds the piece's length to the tile's counter.  The block whose addition completes the tile (the
    // LAST ARRIVER, whichever piece that is: nobody ever waits, so no dispatch order or residency can stall the launch) takes one
    // agent-scope acquire and reduces: the slabs of all the tile's pieces in ascending-K order — so the sums do not depend on who
    // reduces — through the generic store above, and leaves the counter zero for the next launch.  (Keeping the reducer's own
    // piece in its accumulators and adding the others to them — by VALU or, exactly, through the matrix pipe as three bf16 terms
    // — was built first: any update of the accumulator tuple after the K loop made hipcc keep a second copy of its 128 registers
    // and spill 55-270 VGPRs, reloaded INSIDE the K loop.)
    bool sk_done = false;
    if constexpr (SKM) {
        if (esplit != 0) {
            const int tile = cur.gi, ua = cur.kt0 / p.sk;
            const int my_units = (cur.kt0 + cur.nk == nk_all ? sk_upt : (cur.kt0 + cur.nk) / p.sk) - ua;
            // a block can hold TWO pieces (the head of its share's last tile and the tail of its first): slab 0 of the block takes
            // the piece that contains the share's last unit (its first item), slab 1 the other
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                p.sk_slab + ((size_t)blockIdx.x * 2 + (vcur == 0 ? 0 : 1)) * 65536, 0, 262144, 0x00020000);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[i][j]), rs,
                                                           ((wave * 32 + i * 4 + j) * 64 + lane) * 16, 0, 16 /* sc1 */);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave (this also lands the prefetched K-tiles)
            SEG_BARRIER();
            // the A slot the K loop read last is dead until the next item's P1: its first word carries the ticket to all waves
            volatile unsigned* mail = reinterpret_cast<volatile unsigned*>(smem + ((abuf == 0 ? 2 : abuf - 1) * 2) * HT);
            if (tid == 0) {
                // the slab stores above are write-through (sc1) and drained by every wave before the barrier; the release makes
                // the ordering against the ticket explicit for pieces that meet across XCD L2s (sk_tile_aligned = 0) — one
                // buffer_wbl2 with nothing dirty behind write-through stores
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                *mail = __hip_atomic_fetch_add(p.sk_flag + (size_t)tile * 16, (unsigned)my_units, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SEG_BARRIER();
            const int before = __builtin_amdgcn_readfirstlane((int)*mail);
            if (before + my_units == sk_upt) {
                // the tile's pieces in ascending K: walk down to the block that holds the tile's first unit, then up to its last
                constexpr int SK_MAXC = 10;
                const int nl = (int)gridDim.x >> 3;
                const int tile_u0 = tile * sk_upt, tile_u1 = tile_u0 + sk_upt;
                int first = blockIdx.x, u0 = 0, u1 = 0;
                sk_range(first, u0, u1);
                while (u0 > tile_u0) {
                    first = (first >> 3) > 0 ? first - 8 : (nl - 1) * 8 + (first & 7) - 1;
                    sk_range(first, u0, u1);
                }
                int cb[SK_MAXC];
                int nc = 0;
                for (int b2 = first;; b2 = (b2 >> 3) < nl - 1 ? b2 + 8 : (b2 & 7) + 1) {
                    sk_range(b2, u0, u1);
                    // (a block without work wrote nothing)  slab index = 2 * block + (0: the piece holds the block's last unit)
                    if (u1 > u0) {
                        if (nc < SK_MAXC) cb[nc++] = b2 * 2 + ((u1 - 1) / sk_upt == tile ? 0 : 1);
                        // a tile cut into more pieces than the reducer's list holds would be stored with part of its K sum missing:
                        // never silently — the error word of the header (what molly_gemm_ctx_streamk_timeouts reads; the host's
                        // grid rule keeps a tile at <= 9 pieces, tests assert the word stays 0)
                        else if (tid == 0) atomicOr(p.sk_flag - 16, 1u);
                    }
                    if (u1 >= tile_u1) break;
                }
                if (wave == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                SEG_BARRIER();
                asm volatile("" ::: "memory");
                const float* sl0 = p.sk_slab + (size_t)(wave * 32 * 64 + lane) * 4;
                // eight quads at a time (the kernel's 256 registers are 128 accumulator + 128 vector registers, and ~45 of the
                // latter hold the K loop's lane constants: 64 is what fits without spilling into the K loop); the first TWO pieces'
                // loads are issued together — a tile cut once, the common case, costs four round trips of 16 KB per wave
#pragma unroll 1
                for (int h = 0; h < 4; ++h) {
                    f32x4 sum[8], v[8];
                    const float* sa = sl0 + (size_t)cb[0] * 65536 + h * 8 * 256;
                    const float* sb = sl0 + (size_t)cb[nc > 1 ? 1 : 0] * 65536 + h * 8 * 256;
#pragma unroll
                    for (int q = 0; q < 8; ++q) sum[q] = *reinterpret_cast<const f32x4*>(sa + q * 256);
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const f32x4*>(sb + q * 256);
                    if (nc > 1) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) sum[q] += v[q];
                    }
                    for (int ci = 2; ci < nc; ++ci) {
                        const float* sl = sl0 + (size_t)cb[ci] * 65536 + h * 8 * 256;
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const f32x4*>(sl + q * 256);
#pragma unroll
                        for (int q = 0; q < 8; ++q) sum[q] += v[q];
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        float vv[4] = {sum[q][0], sum[q][1], sum[q][2], sum[q][3]};
                        if (TO) store_quad_t(h * 2 + (q >> 2), q & 3, vv);
                        else store_quad(h * 2 + (q >> 2), q & 3, vv);
                    }
                }
                if (tid == 0) __hip_atomic_store(p.sk_flag + (size_t)tile * 16, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            sk_done = true;
        }
    }
    // Fast paths first: a full interior tile with no epilogue flag (every qkv / dgrad launch) or with the residual add alone
    // (o_proj, down_proj).  The accumulator layout gives a lane 4 consecutive columns of one row (8 bytes as bf16), and a
    // wave-wide store of those touches 16 rows x 32 bytes: 4,096 32-byte requests per CU and tile, every CU at the same moment
    // — stamped, the 32 stores of a tile took 4-7 us of a 55 us K = 2048 tile (tools/gemm_diag/run_tilestamp.py).  Two lane-row
    // swaps per register pair (v_permlane32_swap, then v_permlane16_swap: a 3-cycle of the bits {register j&1, lane>>5,
    // (lane>>4)&1}) regroup the quads so that lane-row fq holds columns s*32 + fq*8 .. +7 of its row for s = 0, 1: 16-byte
    // stores, 64 contiguous bytes per row and instruction, half the requests and half the instructions.
    // (Tried on top: whole 128-byte rows through a per-wave 2 KB slab in the A slot that is free during the epilogue — 4 ds_write_b64
    // + 2 ds_read_b128 per row-block instead of the swaps: K = 2048 launches 202.5 / 108.4 us against 200.1 / 101.5 with the swaps,
    // bit-identical; not kept.)
    // ---- whole-line access for the epilogues that touch memory beyond one plain store (round 6).  The swapped MFMA operands leave the ROW index in
    // the low four lane bits, so any access shaped like the accumulators touches 16 lines per 16 consecutive lanes — 8,600 cycles per 128 KB tile and
    // CU, against 3,550 when 8 consecutive lanes cover one line (tools/r06/store_diag/store_bench.hip).  The SwiGLU-backward epilogue moves four tiles
    // that way (gate, up in; d gate, d up out): 18 us per tile, most of what the fusion cost.  So the wave's packed bf16 results pass through a 4 KB
    // slab of the A stage that is free between two tiles (the one the last K-tile was read from; every read of it has returned: the closing barrier
    // above), 32 rows at a time: written as the lanes own them (8-byte quads, 16-byte chunks XOR-swizzled by the row), read back as
    // lane = (row l >> 3 of 8, chunk l & 7) — 16 bytes per lane, 8 rows x 128 bytes per instruction.  DS operations of one wave execute in order,
    // so the slab needs no barrier and no wait between its writes and reads beyond the register dependencies hipcc tracks.
    constexpr bool EPI_LDS = MOLLY_GEMM_EPI_LDS && !SKM && !P2;     // (not in the stream-K and two-phase instantiations: their register files are full)
    char* const epi_slab = reinterpret_cast<char*>(smem + ((abuf == 0 ? 2 : abuf - 1) * 2) * HT) + wave * 4096;
    const int epi_row = lane >> 3, epi_ch = lane & 7;
    // rows 32 c .. 32 c + 31 of the wave's 128 (row blocks i = 2c, 2c + 1), packed with pack_bf2: -> out[k] = row 32 c + 8 k + epi_row, columns 8 epi_ch .. + 7
    auto epi_transpose = [&](int cidx, u32x4 (&out)[4]) {
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = 2 * cidx + ib, row = ib * 16 + fr;
                const u32x2 q = u32x2{pack_bf2(acc[i][j][0], acc[i][j][1]), pack_bf2(acc[i][j][2], acc[i][j][3])};
                *reinterpret_cast<u32x2*>(epi_slab + row * 128 + (((2 * j + (fq >> 1)) ^ (fr & 7)) << 4) + (fq & 1) * 8) = q;
            }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            out[k] = *reinterpret_cast<const u32x4*>(epi_slab + (k * 8 + epi_row) * 128 + ((epi_ch ^ epi_row) << 4));
    };
    const bool interior = em0 + 256 <= eM && en0 + 256 <= eN && p.splits <= 1 && !(GRP ? eto : TO);
    // (+ MOLLY_GEMM_ACCUMULATE alone, bf16: the residual path with C itself as the second operand — the weight gradients of every micro-batch but
    // the first under gradient accumulation, which took the generic epilogue's 32 dependent 8-byte read-add-write round trips: +9 % on the grouped
    // launch of Qwen3-4B's layer at 3,072 rows, +14 % on Qwen3-8B's at 4,096: profiles/r05_logs/grouped_c3.log)
    const bool fast16 = interior && (p.flags == 0 || p.flags == MOLLY_GEMM_RESIDUAL || p.flags == MOLLY_GEMM_ACCUMULATE);
    const bool swiglu16 = !AT && !BT && !TO && !GRP && p.flags == MOLLY_GEMM_SWIGLU && em0 + 256 <= eM;      // (N % 256 == 0)
    const bool swiglu_bwd16 = !AT && BT && !TO && !GRP && p.flags == MOLLY_GEMM_SWIGLU_BWD && em0 + 256 <= eM && en0 + 256 <= eN;
    // transposed output (weight gradients), interior tile, plain bf16 store: the lane owns 4 consecutive m of one n
    const bool to16 = (GRP ? eto : TO) && em0 + 256 <= eM && en0 + 256 <= eN && p.splits <= 1 && p.flags == 0;
    const bool to16acc = (GRP ? eto : TO) && em0 + 256 <= eM && en0 + 256 <= eN && p.splits <= 1 && p.flags == MOLLY_GEMM_ACCUMULATE;
    pend_stores = (fast16 || to16 || to16acc) ? 16 : swiglu16 ? 24 : swiglu_bwd16 ? 32
                : (em0 + 256 <= eM && en0 + 256 <= eN &&
                   !(p.flags & (MOLLY_GEMM_BIAS | MOLLY_GEMM_RESIDUAL | MOLLY_GEMM_ACCUMULATE | MOLLY_GEMM_SWIGLU | MOLLY_GEMM_SWIGLU_BWD))) ? 32
                : 0;
    if (sk_done) {
        pend_stores = 0;                          // the dump drained everything; a reducer's stores behind it: conservative waits
    } else if (to16) {
        // C^T[n][m]: the quads of row-blocks (i even, i odd) regrouped so that lane-row fq holds m = s*32 + fq*8 .. +7 of its n
        bf16_t* c0 = reinterpret_cast<bf16_t*>(eC) + (size_t)(en0 + wc * 64 + fr) * eldc + em0 + wr * 128 + fq * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16_t* c = c0 + (size_t)j * 16 * eldc;
#pragma unroll
            for (int sh = 0; sh < 4; ++sh) {
                unsigned d0[2], d1[2];
                d0[0] = pack_bf2(acc[2 * sh][j][0], acc[2 * sh][j][1]);
                d0[1] = pack_bf2(acc[2 * sh][j][2], acc[2 * sh][j][3]);
                d1[0] = pack_bf2(acc[2 * sh + 1][j][0], acc[2 * sh + 1][j][1]);
                d1[1] = pack_bf2(acc[2 * sh + 1][j][2], acc[2 * sh + 1][j][3]);
                regroup_rows(d0[0], d1[0]);
                regroup_rows(d0[1], d1[1]);
                *reinterpret_cast<u32x4*>(c + sh * 32) = u32x4{d0[0], d0[1], d1[0], d1[1]};
            }
        }
    } else if (to16acc) {
        // the same store with C^T's old values added in fp32 first (one rounding, as the generic epilogue does): the fp32 quads regrouped, the old
        // values of column group j + TOA_AHEAD requested before group j is worked on
        bf16_t* c0 = reinterpret_cast<bf16_t*>(eC) + (size_t)(en0 + wc * 64 + fr) * eldc + em0 + wr * 128 + fq * 8;
        constexpr int TOA_AHEAD = 2;
        u32x4 ova[TOA_AHEAD][4];
        auto old_load = [&](int j, u32x4 (&ov)[4]) {
            const bf16_t* c = c0 + (size_t)j * 16 * eldc;
#pragma unroll
            for (int sh = 0; sh < 4; ++sh) ov[sh] = *reinterpret_cast<const u32x4*>(c + sh * 32);
        };
#pragma unroll
        for (int j = 0; j < TOA_AHEAD; ++j) old_load(j, ova[j]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16_t* c = c0 + (size_t)j * 16 * eldc;
            u32x4 (&ov)[4] = ova[j % TOA_AHEAD];
#pragma unroll
            for (int sh = 0; sh < 4; ++sh) {
                float v[2][4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float f0 = acc[2 * sh][j][e], f1 = acc[2 * sh + 1][j][e];
                    unsigned x0 = __float_as_uint(f0), x1 = __float_as_uint(f1);
                    regroup_rows(x0, x1);
                    v[0][e] = __uint_as_float(x0);
                    v[1][e] = __uint_as_float(x1);
                }
                const u32x4 o = ov[sh];
                *reinterpret_cast<u32x4*>(c + sh * 32) =
                    u32x4{pack_bf2(v[0][0] + bflo(o[0]), v[0][1] + bfhi(o[0])), pack_bf2(v[0][2] + bflo(o[1]), v[0][3] + bfhi(o[1])),
                          pack_bf2(v[1][0] + bflo(o[2]), v[1][1] + bfhi(o[2])), pack_bf2(v[1][2] + bflo(o[3]), v[1][3] + bfhi(o[3]))};
            }
            if (j + TOA_AHEAD < 4) {
                __builtin_amdgcn_sched_barrier(0);
                old_load(j + TOA_AHEAD, ov);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (MOLLY_GEMM_DIAG_NOEPI && fast16 && p.flags == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(acc[i][j]));
    } else if (fast16 && EPI_LDS) {
        // plain / residual-add / accumulate epilogue, whole-line form (round 6).  Without a second operand the packed result goes through the slab
        // once.  With one, the fp32 sums must be formed where the accumulators are, so the OPERAND travels first: it is loaded in whole lines (lane = row of 8, 16-byte chunk), written to the slab and read back as the accumulators
        // lie (8-byte quads), added in fp32 with one rounding exactly as before, and the packed sums return through the slab to whole-line stores.
        bf16_t* c0 = reinterpret_cast<bf16_t*>(eC) + (size_t)(em0 + wr * 128 + epi_row) * eldc + en0 + wc * 64 + epi_ch * 8;
        if (p.flags == 0) {
#pragma unroll
            for (int cidx = 0; cidx < 4; ++cidx) {
                u32x4 d[4];
                epi_transpose(cidx, d);
#pragma unroll
                for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(c0 + (size_t)(cidx * 32 + k * 8) * eldc) = d[k];
            }
        } else {
            const bool acc_c = p.flags == MOLLY_GEMM_ACCUMULATE;
            const int ldr = acc_c ? eldc : p.ldres;
            const bf16_t* r0 = (acc_c ? reinterpret_cast<const bf16_t*>(eC) : p.res) + (size_t)(em0 + wr * 128 + epi_row) * ldr + en0 + wc * 64 + epi_ch * 8;
            // the second operand of the next 32 rows is requested before this chunk is worked on (a ring of two sets of four 16-byte pieces)
            u32x4 rva[2][4];
            auto res_load4 = [&](int cidx, u32x4 (&rv)[4]) {
#pragma unroll
                for (int k = 0; k < 4; ++k) rv[k] = *reinterpret_cast<const u32x4*>(r0 + (size_t)(cidx * 32 + k * 8) * ldr);
            };
            res_load4(0, rva[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int cidx = 0; cidx < 4; ++cidx) {
                if (cidx + 1 < 4) {
                    res_load4(cidx + 1, rva[(cidx + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                u32x4 (&rv)[4] = rva[cidx & 1];
                // operand: whole-line layout -> slab -> accumulator layout
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    *reinterpret_cast<u32x4*>(epi_slab + (k * 8 + epi_row) * 128 + ((epi_ch ^ epi_row) << 4)) = rv[k];
                u32x2 rq[2][4];
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        rq[ib][j] = *reinterpret_cast<const u32x2*>(epi_slab + (ib * 16 + fr) * 128 + (((2 * j + (fq >> 1)) ^ (fr & 7)) << 4) + (fq & 1) * 8);
                // sums (fp32, one rounding) back through the slab
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int i = 2 * cidx + ib;
                        const u32x2 o = rq[ib][j];
                        const u32x2 q = u32x2{pack_bf2(acc[i][j][0] + bflo(o[0]), acc[i][j][1] + bfhi(o[0])),
                                              pack_bf2(acc[i][j][2] + bflo(o[1]), acc[i][j][3] + bfhi(o[1]))};
                        *reinterpret_cast<u32x2*>(epi_slab + (ib * 16 + fr) * 128 + (((2 * j + (fq >> 1)) ^ (fr & 7)) << 4) + (fq & 1) * 8) = q;
                    }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const u32x4 d = *reinterpret_cast<const u32x4*>(epi_slab + (k * 8 + epi_row) * 128 + ((epi_ch ^ epi_row) << 4));
                    *reinterpret_cast<u32x4*>(c0 + (size_t)(cidx * 32 + k * 8) * eldc) = d;
                }
            }
        }
    } else if (fast16) {
        bf16_t* c0 = reinterpret_cast<bf16_t*>(eC) + (size_t)(em0 + wr * 128 + fr) * eldc + en0 + wc * 64 + fq * 8;
        if (p.flags == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bf16_t* c = c0 + (size_t)i * 16 * eldc;
                unsigned d[4][2];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    d[j][0] = pack_bf2(acc[i][j][0], acc[i][j][1]);
                    d[j][1] = pack_bf2(acc[i][j][2], acc[i][j][3]);
                }
#pragma unroll
                for (int sh = 0; sh < 2; ++sh) {
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        regroup_rows(d[2 * sh][w], d[2 * sh + 1][w]);
                    }
                    *reinterpret_cast<u32x4*>(c + sh * 32) = u32x4{d[2 * sh][0], d[2 * sh][1], d[2 * sh + 1][0], d[2 * sh + 1][1]};
                }
            }
        } else {
            // the residual quads of row group i + RES_AHEAD are requested before group i is used (a ring of register sets: the fragment
            // registers are dead here).  As one load pair per row group the epilogue was eight dependent round trips — each wait also covered the
            // stores of the group before, older in the in-order vmcnt queue (generated code: load, load, ~100 instructions, store, store,
            // eight times over)
            const bool acc_c = p.flags == MOLLY_GEMM_ACCUMULATE;
            const int ldr = acc_c ? eldc : p.ldres;
            const bf16_t* r0 = (acc_c ? reinterpret_cast<const bf16_t*>(eC) : p.res) + (size_t)(em0 + wr * 128 + fr) * ldr + en0 + wc * 64 + fq * 8;
            constexpr int RES_AHEAD = MOLLY_GEMM_RES_AHEAD;
            u32x4 rva[RES_AHEAD][2];
            auto res_load = [&](int i, u32x4 (&rv)[2]) {
                const bf16_t* r = r0 + (size_t)i * 16 * ldr;
                rv[0] = *reinterpret_cast<const u32x4*>(r);
                rv[1] = *reinterpret_cast<const u32x4*>(r + 32);
            };
#pragma unroll
            for (int i = 0; i < RES_AHEAD; ++i) res_load(i, rva[i]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bf16_t* c = c0 + (size_t)i * 16 * eldc;
                const u32x4 rv0 = rva[i % RES_AHEAD][0], rv1 = rva[i % RES_AHEAD][1];
#pragma unroll
                for (int sh = 0; sh < 2; ++sh) {
                    float v[2][4];                               // fp32 sums regrouped (one rounding, as before the regrouping)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // (through named floats: __builtin_bit_cast applied to a vector ELEMENT read element 0 for every e)
                        const float f0 = acc[i][2 * sh][e], f1 = acc[i][2 * sh + 1][e];
                        unsigned x0 = __float_as_uint(f0), x1 = __float_as_uint(f1);
                        regroup_rows(x0, x1);
                        v[0][e] = __uint_as_float(x0);
                        v[1][e] = __uint_as_float(x1);
                    }
                    const u32x4 rv = sh ? rv1 : rv0;
                    *reinterpret_cast<u32x4*>(c + sh * 32) =
                        u32x4{pack_bf2(v[0][0] + bflo(rv[0]), v[0][1] + bfhi(rv[0])), pack_bf2(v[0][2] + bflo(rv[1]), v[0][3] + bfhi(rv[1])),
                              pack_bf2(v[1][0] + bflo(rv[2]), v[1][1] + bfhi(rv[2])), pack_bf2(v[1][2] + bflo(rv[3]), v[1][3] + bfhi(rv[3]))};
                }
                if (i + RES_AHEAD < 8) {
                    __builtin_amdgcn_sched_barrier(0);
                    res_load(i + RES_AHEAD, rva[i % RES_AHEAD]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    } else if (GRP ? eto : TO) {
        // transposed output, general form (ragged edge, accumulate, fp32, or a split-K slab)
        float* slab = p.splits > 1 ? p.ws + (size_t)esplit * eM * eN : nullptr;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if (slab) {
                    const int m = em0 + wr * 128 + i * 16 + fq * 4, n = en0 + wc * 64 + j * 16 + fr;
                    if (m < eM && n < eN) *reinterpret_cast<f32x4*>(slab + (size_t)n * eM + m) = f32x4{v[0], v[1], v[2], v[3]};
                } else {
                    store_quad_t(i, j, v);
                }
            }
        }
    } else if (p.splits > 1) {
        // split-K: plain fp32 partial slab of this slice; molly's splitk_reduce kernel sums the slabs (launch-boundary
        // reduce: cheaper than an in-launch combine at these slab sizes, guide §5 "Projection GEMM" item 2)
        float* slab = p.ws + (size_t)esplit * eM * eN;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = em0 + wr * 128 + i * 16 + (lane & 15);
            if (m >= eM) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = en0 + wc * 64 + j * 16 + (lane >> 4) * 4;
                if (n >= eN) continue;
                *reinterpret_cast<f32x4*>(slab + (size_t)m * eN + n) = acc[i][j];
            }
        }
    } else if (swiglu16) {
        // SwiGLU-fused gate|up projection, interior rows: as the branch below, with gate, up and activation regrouped to 16-byte
        // stores (the wave's 32 activation columns = 4 lane-rows x 8 columns: 64 contiguous bytes per row and instruction)
        const int ff = eN >> 1;
        const size_t col = (size_t)(en0 >> 1) + (wc >> 1) * 64 + (wc & 1) * 32 + fq * 8;
        bf16_t* g0 = reinterpret_cast<bf16_t*>(eC) + (size_t)(em0 + wr * 128 + fr) * eldc + col;
        bf16_t* a0 = const_cast<bf16_t*>(p.res) + (size_t)(em0 + wr * 128 + fr) * p.ldres + col;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            unsigned gq[2][2], uq[2][2], aq[2][2];            // [jj][dword]
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                gq[jj][0] = pack_bf2(acc[i][jj][0], acc[i][jj][1]);
                gq[jj][1] = pack_bf2(acc[i][jj][2], acc[i][jj][3]);
                uq[jj][0] = pack_bf2(acc[i][2 + jj][0], acc[i][2 + jj][1]);
                uq[jj][1] = pack_bf2(acc[i][2 + jj][2], acc[i][2 + jj][3]);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float ga = bflo(gq[jj][e]), gb = bfhi(gq[jj][e]);
                    const float sa = bf2f(f2bf(ga * sigmoid_fast(ga))), sb = bf2f(f2bf(gb * sigmoid_fast(gb)));
                    aq[jj][e] = pack_bf2(sa * bflo(uq[jj][e]), sb * bfhi(uq[jj][e]));
                }
            }
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                regroup_rows(gq[0][w], gq[1][w]);
                regroup_rows(uq[0][w], uq[1][w]);
                regroup_rows(aq[0][w], aq[1][w]);
            }
            bf16_t* gp = g0 + (size_t)i * 16 * eldc;
            *reinterpret_cast<u32x4*>(gp) = u32x4{gq[0][0], gq[0][1], gq[1][0], gq[1][1]};
            *reinterpret_cast<u32x4*>(gp + ff) = u32x4{uq[0][0], uq[0][1], uq[1][0], uq[1][1]};
            *reinterpret_cast<u32x4*>(a0 + (size_t)i * 16 * p.ldres) = u32x4{aq[0][0], aq[0][1], aq[1][0], aq[1][1]};
        }
    } else if (swiglu_bwd16 && EPI_LDS) {
        // SwiGLU backward in the down-projection's dgrad, interior tile, whole-line form: d(act) packed (the rounding the unfused path applies),
        // transposed through the slab, then gate / up loaded and d(gate) / d(up) stored 16 bytes per lane with 8 lanes per 128-byte line; the
        // arithmetic is swiglu_bwd_pair on the same bf16 pairs: bit-identical to dgrad + molly_swiglu_bwd and to the regrouped form below.
        const int ff = eN;
        const size_t col = (size_t)en0 + wc * 64 + epi_ch * 8;
        const bf16_t* g0 = p.res + (size_t)(em0 + wr * 128 + epi_row) * p.ldres + col;
        bf16_t* c0 = reinterpret_cast<bf16_t*>(eC) + (size_t)(em0 + wr * 128 + epi_row) * eldc + col;
        // gate / up of the NEXT 16 rows (two row groups) are requested before the current 16 are worked on: a ring of two sets of two
        u32x4 gva[2][2], uva[2][2];
        auto sbw_load2 = [&](int h, u32x4 (&gv)[2], u32x4 (&uv)[2]) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const bf16_t* gp = g0 + (size_t)(h * 16 + k * 8) * p.ldres;
                gv[k] = *reinterpret_cast<const u32x4*>(gp);
                uv[k] = *reinterpret_cast<const u32x4*>(gp + ff);
            }
        };
        sbw_load2(0, gva[0], uva[0]);
        __builtin_amdgcn_sched_barrier(0);
        u32x4 dq[4];
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            if (h + 1 < 8) {
                sbw_load2(h + 1, gva[(h + 1) & 1], uva[(h + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
            if ((h & 1) == 0) epi_transpose(h >> 1, dq);
            u32x4 (&gv)[2] = gva[h & 1], (&uv)[2] = uva[h & 1];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                u32x4 og, ou;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const u32x2 r = swiglu_bwd_pair(gv[k][e], uv[k][e], dq[(h & 1) * 2 + k][e]);
                    og[e] = r[0];
                    ou[e] = r[1];
                }
                bf16_t* c = c0 + (size_t)(h * 16 + k * 8) * eldc;
                *reinterpret_cast<u32x4*>(c) = og;
                *reinterpret_cast<u32x4*>(c + ff) = ou;
            }
        }
    } else if (swiglu_bwd16) {
        // SwiGLU backward in the down-projection's dgrad, interior tile: as the general branch below, with d(act) regrouped
        // first (packed bf16, the rounding the unfused path applies), gate / up loaded and d(gate) / d(up) stored 16 bytes per lane
        const int ff = eN;
        const size_t col = (size_t)en0 + wc * 64 + fq * 8;
        const bf16_t* g0 = p.res + (size_t)(em0 + wr * 128 + fr) * p.ldres + col;
        bf16_t* c0 = reinterpret_cast<bf16_t*>(eC) + (size_t)(em0 + wr * 128 + fr) * eldc + col;
        // gate / up of row group i + SBW_AHEAD are requested before group i is worked on (a ring of SBW_AHEAD register sets, 16 each).
        // One group at a time the epilogue was eight dependent round trips, each wait covering the previous group's stores as well
        // (in-order vmcnt): ~20 us per tile against ~50 of K loop
        constexpr int SBW_AHEAD = MOLLY_GEMM_SBW_AHEAD;
        u32x4 gva[SBW_AHEAD][2], uva[SBW_AHEAD][2];
        auto sbw_load = [&](int i, u32x4 (&gv)[2], u32x4 (&uv)[2]) {
            const bf16_t* gp = g0 + (size_t)i * 16 * p.ldres;
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                gv[sh] = *reinterpret_cast<const u32x4*>(gp + sh * 32);
                uv[sh] = *reinterpret_cast<const u32x4*>(gp + ff + sh * 32);
            }
        };
#pragma unroll
        for (int i = 0; i < SBW_AHEAD; ++i) sbw_load(i, gva[i], uva[i]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            bf16_t* c = c0 + (size_t)i * 16 * eldc;
            u32x4 (&gv)[2] = gva[i % SBW_AHEAD], (&uv)[2] = uva[i % SBW_AHEAD];
            unsigned d[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                d[j][0] = pack_bf2(acc[i][j][0], acc[i][j][1]);
                d[j][1] = pack_bf2(acc[i][j][2], acc[i][j][3]);
            }
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
                regroup_rows(d[2 * sh][0], d[2 * sh + 1][0]);
                regroup_rows(d[2 * sh][1], d[2 * sh + 1][1]);
                const unsigned dq[4] = {d[2 * sh][0], d[2 * sh][1], d[2 * sh + 1][0], d[2 * sh + 1][1]};
                u32x4 og, ou;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const u32x2 r = swiglu_bwd_pair(gv[sh][e], uv[sh][e], dq[e]);
                    og[e] = r[0];
                    ou[e] = r[1];
                }
                *reinterpret_cast<u32x4*>(c + sh * 32) = og;
                *reinterpret_cast<u32x4*>(c + ff + sh * 32) = ou;
            }
            if (i + SBW_AHEAD < 8) {
                __builtin_amdgcn_sched_barrier(0);
                sbw_load(i + SBW_AHEAD, gv, uv);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (!AT && !BT && !TO && !GRP && (p.flags & MOLLY_GEMM_SWIGLU)) {
        // SwiGLU-fused gate|up projection: acc[i][jj] = gate, acc[i][2 + jj] = up of the SAME activation columns (stage_kc's
        // row remap).  Stores gu = [gate | up] (what the backward reads) and act = silu(gate) * up with the roundings of the
        // two-kernel path (GEMM output rounded to bf16, silu rounded, product rounded: bit-identical to molly_swiglu_fwd).
        const int ff = eN >> 1;
        bf16_t* act = const_cast<bf16_t*>(p.res);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = em0 + wr * 128 + i * 16 + fr;
            if (m >= eM) continue;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int c = (en0 >> 1) + (wc >> 1) * 64 + (wc & 1) * 32 + jj * 16 + fq * 4;
                const u32x2 gq = u32x2{pack_bf2(acc[i][jj][0], acc[i][jj][1]), pack_bf2(acc[i][jj][2], acc[i][jj][3])};
                const u32x2 uq = u32x2{pack_bf2(acc[i][2 + jj][0], acc[i][2 + jj][1]), pack_bf2(acc[i][2 + jj][2], acc[i][2 + jj][3])};
                bf16_t* gp = reinterpret_cast<bf16_t*>(eC) + (size_t)m * eldc + c;
                *reinterpret_cast<u32x2*>(gp) = gq;
                *reinterpret_cast<u32x2*>(gp + ff) = uq;
                u32x2 o;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const float ga = bflo(gq[e]), gb = bfhi(gq[e]);
                    const float sa = bf2f(f2bf(ga * sigmoid_fast(ga))), sb = bf2f(f2bf(gb * sigmoid_fast(gb)));
                    o[e] = pack_bf2(sa * bflo(uq[e]), sb * bfhi(uq[e]));
                }
                *reinterpret_cast<u32x2*>(act + (size_t)m * p.ldres + c) = o;
            }
        }
    } else if (!AT && BT && !TO && !GRP && (p.flags & MOLLY_GEMM_SWIGLU_BWD)) {
        // SwiGLU backward fused into the down-projection's dgrad: acc = d(act)[m][n], n < ff = N.  res = gu = [gate | up]
        // [M][2ff] (ldres) is read, C = d(gu) [M][2ff] (ldc) is written: d(gate) at column n, d(up) at ff + n.  d(act) is rounded
        // to bf16 first and the arithmetic is swiglu_bwd_pair (common.h), so the result is bit-identical to the dgrad GEMM
        // followed by molly_swiglu_bwd — without d(act)'s round trip through HBM.
        const int ff = eN;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = em0 + wr * 128 + i * 16 + fr;
            if (m >= eM) continue;
            u32x2 gq[4], uq[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = en0 + wc * 64 + j * 16 + fq * 4;
                if (n >= eN) continue;
                const bf16_t* gp = p.res + (size_t)m * p.ldres + n;
                gq[j] = *reinterpret_cast<const u32x2*>(gp);
                uq[j] = *reinterpret_cast<const u32x2*>(gp + ff);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = en0 + wc * 64 + j * 16 + fq * 4;
                if (n >= eN) continue;
                const u32x2 dq = u32x2{pack_bf2(acc[i][j][0], acc[i][j][1]), pack_bf2(acc[i][j][2], acc[i][j][3])};
                u32x2 og, ou;
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const u32x2 r = swiglu_bwd_pair(gq[j][e], uq[j][e], dq[e]);
                    og[e] = r[0];
                    ou[e] = r[1];
                }
                bf16_t* c = reinterpret_cast<bf16_t*>(eC) + (size_t)m * eldc + n;
                *reinterpret_cast<u32x2*>(c) = og;
                *reinterpret_cast<u32x2*>(c + ff) = ou;
            }
        }
    } else {
        // ---- general epilogue: every flag, ragged edges
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                store_quad(i, j, v);
            }
    }   // epilogue variants

    if (!more) {
        if constexpr (DYN) {
            if (tid == 0 && vnext == dyn_last) __hip_atomic_store(dcnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        break;
    }
    vcur = vnext;
    cur = nxt;
    }   // persistent tile loop
#undef MMA_QUAD
#undef MMA_QUAD0
#undef SEG_BARRIER
}

// ================================================================================================
// Decode rows: y[M <= 64, N] = x[M, K] W[N, K]^T — the q_len = 1 projections of `generate` (HF:models/qwen3/modeling_qwen3.py:
// 76-83, 225-236 with one token per sample; reference src/model/omics_one.py:220-232).  A pure WEIGHT STREAM: every byte of W is
// read once, x (M x K, a few hundred KB) stays in L2, the arithmetic is free.  So no tiles, no LDS staging, no slabs:
//   * one workgroup = 16 rows of W (one MFMA tile of outputs), its 4 waves split K in four; a wave streams its quarter of the 16
//     rows straight into registers with 16-byte non-temporal loads — W is the A operand of mfma_16x16x32 (lane (n = lane & 15,
//     g = lane >> 4) holds W[n][k0 + 8g .. +7]), the two k-steps of a 128-byte line issued back to back, U line-pairs
//     (8 loads per lane) in flight before the first is used (guide 5 'glds vs register staging', GEMV row);
//   * x is the B operand (lane (m = lane & 15, g) holds x[m][k0 + 8g .. +7]): plain loads that hit L1 / L2;
//   * the four partial 16 x M tiles meet in 4 KB of LDS, wave 0..MT-1 apply the epilogue (bias -> GELU -> residual ->
//     accumulate; bf16 or fp32) and store 4 consecutive n per lane.
// N / 16 workgroups of 4 waves, up to 8 per CU: 256 KB of loads in flight per CU.  Replaces, for M <= 64, the 256x256 kernel
// split 8-32 ways over K + fp32 slabs + the reduce launch (2.3-3.1 TB/s on Qwen3-8B's matrices; two launches per projection).
// ================================================================================================
struct SkinnyArgs {
    const bf16_t* X; const bf16_t* W; void* C; const bf16_t* bias; const bf16_t* res;
    int M, N, K, ldx, ldw, ldc, ldres, flags;
};

// MT = 16-row tiles of x (M <= 16 MT), NT = 16-row tiles of W per wave, KW = waves per workgroup = ways K is split.
// What bounds it is the vector-memory path, not HBM: every byte — W from HBM, x from L2 — enters through 64-byte requests at
// ~17 B/clk per CU (measured: 9 TB/s of loads chip-wide whatever the mix).  With NT = 1 a wave loads 2 x-bytes per W-byte at
// M = 32 (2.9 TB/s of W); NT = 4 reuses each x fragment for four W fragments: 0.5 x-bytes per W-byte.
template <int MT, int NT, int KW>
__global__ __launch_bounds__(64 * KW) void gemm_skinny_kernel(SkinnyArgs p) {
    __shared__ float red[KW - 1][NT][MT][256];                  // [waves 1..][n-tile][m-tile][64 lanes x 4 floats]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * 16 * NT;
    const int fr = lane & 15, g = lane >> 4;
    const int kq = p.K / KW;                                    // K per wave (a multiple of 64: host check)
    const bf16_t* wp[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        int wrow = n0 + 16 * j + fr;
        wrow = wrow < p.N ? wrow : p.N - 1;                     // ragged last tile: re-read the last row, masked at the store
        wp[j] = p.W + (size_t)wrow * p.ldw + wave * kq + 8 * g;
    }
    const bf16_t* xp[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        int m = t * 16 + fr;
        m = m < p.M ? m : p.M - 1;
        xp[t] = p.X + (size_t)m * p.ldx + wave * kq + 8 * g;
    }
    f32x4 acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int U = NT * MT <= 2 ? 4 : 2;                     // line pairs (64 k each) in flight per wave
    const int npair = kq >> 6;
    // W AND x of a pair are issued U pairs ahead (x is an L2 round trip too: M x K does not stay in a 32 KB L1 beside 32 streams)
    bf16x8 wa[U][NT], wb[U][NT], xa[U][MT], xb[U][MT];
    auto issue = [&](int slot, int pair) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            wa[slot][j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp[j] + pair * 64));
            wb[slot][j] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp[j] + pair * 64 + 32));
        }
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            xa[slot][t] = *reinterpret_cast<const bf16x8*>(xp[t] + pair * 64);
            xb[slot][t] = *reinterpret_cast<const bf16x8*>(xp[t] + pair * 64 + 32);
        }
    };
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (u < npair) issue(u, u);
    for (int pr = 0; pr < npair; pr += U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (pr + u >= npair) break;
            bf16x8 a[NT], b[NT], ya[MT], yb[MT];
#pragma unroll
            for (int j = 0; j < NT; ++j) { a[j] = wa[u][j]; b[j] = wb[u][j]; }
#pragma unroll
            for (int t = 0; t < MT; ++t) { ya[t] = xa[u][t]; yb[t] = xb[u][t]; }
            if (pr + u + U < npair) issue(u, pr + u + U);      // refill the slot: U pairs stay in flight
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j], ya[t], acc[j][t], 0, 0, 0);
                    acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], yb[t], acc[j][t], 0, 0, 0);
                }
        }
    }
    // ---- the K parts meet in LDS; lane (m = fr, g) of wave 0 holds y[m][n0 + 16 j + 4g .. +3] of m-tile t
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int t = 0; t < MT; ++t) *reinterpret_cast<f32x4*>(&red[wave - 1][j][t][lane * 4]) = acc[j][t];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        f32x4 v = acc[j][t];
#pragma unroll
        for (int w = 0; w < KW - 1; ++w) v += *reinterpret_cast<const f32x4*>(&red[w][j][t][lane * 4]);
        const int m = t * 16 + fr, n = n0 + 16 * j + 4 * g;
        if (m >= p.M || n >= p.N) continue;                     // N % 4 == 0 (host check)
        float o[4] = {v[0], v[1], v[2], v[3]};
        if (p.flags & MOLLY_GEMM_BIAS) {
            const u32x2 bb = *reinterpret_cast<const u32x2*>(p.bias + n);
            o[0] += bflo(bb[0]); o[1] += bfhi(bb[0]); o[2] += bflo(bb[1]); o[3] += bfhi(bb[1]);
        }
        if (p.flags & MOLLY_GEMM_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = gelu_erf(o[e]);
        }
        if (p.flags & MOLLY_GEMM_RESIDUAL) {
            const u32x2 bb = *reinterpret_cast<const u32x2*>(p.res + (size_t)m * p.ldres + n);
            o[0] += bflo(bb[0]); o[1] += bfhi(bb[0]); o[2] += bflo(bb[1]); o[3] += bfhi(bb[1]);
        }
        if (p.flags & MOLLY_GEMM_OUT_F32) {
            float* c = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n;
            if (p.flags & MOLLY_GEMM_ACCUMULATE) {
                const f32x4 old = *reinterpret_cast<const f32x4*>(c);
                o[0] += old[0]; o[1] += old[1]; o[2] += old[2]; o[3] += old[3];
            }
            *reinterpret_cast<f32x4*>(c) = f32x4{o[0], o[1], o[2], o[3]};
        } else {
            bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n;
            if (p.flags & MOLLY_GEMM_ACCUMULATE) {
                const u32x2 old = *reinterpret_cast<const u32x2*>(c);
                o[0] += bflo(old[0]); o[1] += bfhi(old[0]); o[2] += bflo(old[1]); o[3] += bfhi(old[1]);
            }
            *reinterpret_cast<u32x2*>(c) = u32x2{pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
        }
    }
}

// ================================================================================================
// Decode rows, tiled: y[M <= 64, N] = x W^T for the 100-200 MB matrices at M = 17..64 (Qwen3-8B gate|up / down at the batch 32 of
// BASELINE config 5), where the register-streaming kernel above loses: its x fragments enter through the same 64-byte request
// path as W (2 x-bytes per W-byte at M = 32, NT = 1).  Here both operands go HBM / L2 -> LDS by LDS-DMA in whole 128-byte lines
// (8 rows x 128 B per wave-instruction), x ONCE per workgroup and K-tile: a tile is [16 MT rows of x | 128 rows of W] x 64 k =
// 4 + 16 KB (MT = 2), so x adds a quarter to the bytes the memory path carries instead of doubling them — the 128x128 kernel
// reaches 5 TB/s on the 1.2 GB lm_head this way while staging 128 rows of x per tile, 96 of them padding.
//   * 4 waves; wave w owns 32 of the tile's 128 output columns for all 16 MT rows (MT x 2 accumulator tiles);
//   * NSTAGE-deep LDS ring (20 KB per stage at MT = 2: 4 stages, two workgroups per CU = 120 KB in flight per CU), ONE barrier
//     per K-tile, counted vmcnt: the stage issued at K-tile t lands in the slot K-tile t-1 was read from;
//   * K split over `splits` workgroups per column tile so that >= 2 workgroups per CU exist whatever N is; partial tiles go to
//     fp32 slabs [split][M][N] and splitk_reduce_kernel applies the epilogue; splits == 1 stores directly.  (Stream-K's hand-off
//     inside the launch — images in register order, a ticket per column tile, the last slice adds them in slice order — was
//     built and measured 1-3 us SLOWER per launch than the reduce launch it saves: 19.7 / 16.9 / 29.1 against 17.8 / 14.2 / 28.0 us
//     on Qwen3-8B qkv / o / down at M = 32; a dependent launch in the same stream overlaps the tail the hand-off serialises.)
// ================================================================================================
#ifndef MOLLY_ROWS_W_NT
#define MOLLY_ROWS_W_NT 1
#endif
struct RowsArgs {
    const bf16_t* X; const bf16_t* W; void* C; const bf16_t* bias; const bf16_t* res; float* ws;
    int M, N, K, ldx, ldw, ldc, ldres, flags, tiles_n, splits;
    int tiles_m;               // 16 MT-row tiles of x (1 for decode rows; > 1: small grids of the encoders at one sample per GPU)
};

// BN: W rows per tile (128: a wave owns 32 output columns; 64: 16).  GU (decode rows of the gate | up projection, one K slice): the tile is BN / 2
// gate rows and the BN / 2 up rows of the SAME columns (rows ff apart in W, interleaved by the staging's source addresses), so silu(gate) * up is
// formed from the accumulators — no slabs, no combine launch.  BN = 128: a wave's two column blocks are gate | up of one 16-column block;
// BN = 64: waves 2, 3 hold the up halves of waves 0, 1's columns and hand them over through LDS after the K loop.
template <int MT, int NSTAGE, int BN = 128, bool GU = false>
__global__ __launch_bounds__(256) void gemm_rows_kernel(RowsArgs p) {
    constexpr int BM = 16 * MT, BK = 64, A_ELEMS = BM * BK, B_ELEMS = BN * BK, STAGE = A_ELEMS + B_ELEMS;
    constexpr int NLOAD = BN * BK * 2 / 1024 / 4 + BM * BK * 2 / 1024 / 4;          // LDS-DMA instructions per wave per stage: 2-4 of W, 1-2 of x
    constexpr int NJ = BN >= 64 ? BN / 64 : 1;                    // 16-column blocks per wave (BN = 32: one block, HALF of every K-tile — below)
    static_assert(MT == 2 || MT == 4, "16 MT rows: 32 or 64");
    static_assert(BN == 128 || BN == 64 || (BN == 32 && GU && MT == 2), "W rows per tile");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);          // [stage][x tile | W tile]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (column tile fastest, then row tile, then K slice: neighbours in the launch order share the x tile and stream different W tiles)
    const int tn = blockIdx.x % p.tiles_n, tm = (blockIdx.x / p.tiles_n) % p.tiles_m, sp = blockIdx.x / (p.tiles_n * p.tiles_m);
    const int n0 = tn * (GU ? BN / 2 : BN), m0 = tm * BM;       // GU: first of the tile's BN / 2 columns of [0, ff)
    const int ff = p.N >> 1;
    const int nk_all = p.K / BK;
    const int kt0 = (int)((long)nk_all * sp / p.splits), nk = (int)((long)nk_all * (sp + 1) / p.splits) - kt0;

    f32x4 acc[MT][NJ];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int t, int slot) {
        bf16_t* dst = smem + slot * STAGE;
        stage_kc<BM, 4, BK, true>(p.X, p.ldx, m0, p.M, (kt0 + t) * BK, dst, wave, lane);
        // GU: tile rows in groups of 32 — gate, up, (gate, up) — of columns n0 .. (stage_kc's remap: the training step's fused SwiGLU uses it too)
        // (W non-temporal at decode rows — MT = 2: one row tile, every W byte read once by one workgroup; x, which every workgroup re-reads, default)
        stage_kc<BN, 4, BK, true, MT == 2 && MOLLY_ROWS_W_NT>(p.W, p.ldw, n0, p.N, (kt0 + t) * BK, dst + A_ELEMS, wave, lane, GU ? ff : 0, BN == 32 ? 4 : 5);
    };
    // s_waitcnt vmcnt(k * NLOAD): everything but the k youngest stages has landed (k = 0 .. NSTAGE - 2)
    auto wait_younger = [&](int k) {
        if (NSTAGE >= 6 && k >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * NLOAD) : "memory");
        else if (NSTAGE >= 5 && k >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NLOAD) : "memory");
        else if (NSTAGE >= 4 && k >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLOAD) : "memory");
        else if (k == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLOAD) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    const int fr = lane & 15, fq = lane >> 4;
    // the wave's column blocks as rows of the W tile.  GU, BN = 128: block j = 0 gate, j = 1 up of columns n0 + 32 (wave >> 1) + 16 (wave & 1) ..
    const int wrow0 = GU && BN == 128 ? (wave >> 1) * 64 + (wave & 1) * 16 : wave * (BN / 4);
    constexpr int WROW_J = GU && BN == 128 ? 32 : 16;
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s)
        if (s < nk) stage(s, s);
    int slot = 0;
    for (int t = 0; t < nk; ++t) {
        wait_younger(min(NSTAGE - 2, nk - 1 - t));
        __builtin_amdgcn_s_barrier();
        // slot of K-tile t-1: every wave has passed the barrier, so its reads (consumed before that wave's MFMAs) are over
        if (t + NSTAGE - 1 < nk) stage(t + NSTAGE - 1, slot == 0 ? NSTAGE - 1 : slot - 1);
        const bf16_t* sA = smem + slot * STAGE;
        const bf16_t* sB = sA + A_ELEMS;
        if constexpr (BN == 32) {
            // 16 gate + 16 up rows of the same 16 columns: wave = (block: gate | up) x (half of the K-tile); the halves and the pair meet after the loop
            const int kk = wave >> 1;
            bf16x8 xh[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) xh[i] = frag_kc<BK>(sA, i * 16 + fr, kk * 4 + fq);
            const bf16x8 wh = frag_kc<BK>(sB, (wave & 1) * 16 + fr, kk * 4 + fq);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh[i], acc[i][0], 0, 0, 0);
            slot = slot == NSTAGE - 1 ? 0 : slot + 1;
            continue;
        }
        bf16x8 xf[2][MT], wf[2][NJ];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int i = 0; i < MT; ++i) xf[kk][i] = frag_kc<BK>(sA, i * 16 + fr, kk * 4 + fq);
#pragma unroll
            for (int j = 0; j < NJ; ++j) wf[kk][j] = frag_kc<BK>(sB, wrow0 + j * WROW_J + fr, kk * 4 + fq);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[kk][j], xf[kk][i], acc[i][j], 0, 0, 0);
        slot = slot == NSTAGE - 1 ? 0 : slot + 1;
    }
    if constexpr (GU) {
        // ---- silu(gate) * up from the accumulators (rows_tail_swiglu_kernel's roundings: gate and up to bf16 first).  p.C (may be NULL): gate | up
        // [M][2 ff]; p.res: the activation [M][ldres]
        bf16_t* act = const_cast<bf16_t*>(p.res);
        bf16_t* gu = reinterpret_cast<bf16_t*>(p.C);
        if constexpr (BN == 32) {
            __syncthreads();                                   // every wave is done with the ring: its first 8 KB carry the four partial blocks
            float* xch = reinterpret_cast<float*>(smem_raw);   // [4 waves][MT][64 lanes][4]
#pragma unroll
            for (int i = 0; i < MT; ++i) *reinterpret_cast<f32x4*>(xch + ((wave * MT + i) * 64 + lane) * 4) = acc[i][0];
            __syncthreads();
            if (wave >= MT) return;
            const int i = wave;                                // wave i finishes row block i
            auto part = [&](int w) { return *reinterpret_cast<const f32x4*>(xch + ((w * MT + i) * 64 + lane) * 4); };
            f32x4 g = part(0), u = part(1);
            g += part(2); u += part(3);                        // K-tile halves, first + second
            const int m = m0 + i * 16 + fr, n = n0 + fq * 4;
            if (m >= p.M) return;
            if (p.bias) {
                const u32x2 b = *reinterpret_cast<const u32x2*>(p.bias + n), b2 = *reinterpret_cast<const u32x2*>(p.bias + ff + n);
                g[0] += bflo(b[0]); g[1] += bfhi(b[0]); g[2] += bflo(b[1]); g[3] += bfhi(b[1]);
                u[0] += bflo(b2[0]); u[1] += bfhi(b2[0]); u[2] += bflo(b2[1]); u[3] += bfhi(b2[1]);
            }
            const u32x2 gq = u32x2{pack_bf2(g[0], g[1]), pack_bf2(g[2], g[3])}, uq = u32x2{pack_bf2(u[0], u[1]), pack_bf2(u[2], u[3])};
            if (gu) {
                *reinterpret_cast<u32x2*>(gu + (size_t)m * p.ldc + n) = gq;
                *reinterpret_cast<u32x2*>(gu + (size_t)m * p.ldc + ff + n) = uq;
            }
            u32x2 o;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float ga = bflo(gq[e]), gb = bfhi(gq[e]);
                const float sa = bf2f(f2bf(ga * sigmoid_fast(ga))), sb = bf2f(f2bf(gb * sigmoid_fast(gb)));
                o[e] = pack_bf2(sa * bflo(uq[e]), sb * bfhi(uq[e]));
            }
            *reinterpret_cast<u32x2*>(act + (size_t)m * p.ldres + n) = o;
            return;
        }
        if constexpr (BN == 64) {
            __syncthreads();                                   // every wave is done with the ring: its first bytes carry the up halves
            float* xch = reinterpret_cast<float*>(smem_raw);   // [2 waves][MT][64 lanes][4]
            if (wave >= 2) {
#pragma unroll
                for (int i = 0; i < MT; ++i) *reinterpret_cast<f32x4*>(xch + (((wave - 2) * MT + i) * 64 + lane) * 4) = acc[i][0];
            }
            __syncthreads();
            if (wave >= 2) return;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + i * 16 + fr;
            f32x4 g = acc[i][0], u;
            int n;
            if constexpr (BN == 128) { u = acc[i][1]; n = n0 + (wave >> 1) * 32 + (wave & 1) * 16 + fq * 4; }
            else { u = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(smem_raw) + ((wave * MT + i) * 64 + lane) * 4); n = n0 + wave * 16 + fq * 4; }
            if (m >= p.M) continue;
            if (p.bias) {
                const u32x2 b = *reinterpret_cast<const u32x2*>(p.bias + n), b2 = *reinterpret_cast<const u32x2*>(p.bias + ff + n);
                g[0] += bflo(b[0]); g[1] += bfhi(b[0]); g[2] += bflo(b[1]); g[3] += bfhi(b[1]);
                u[0] += bflo(b2[0]); u[1] += bfhi(b2[0]); u[2] += bflo(b2[1]); u[3] += bfhi(b2[1]);
            }
            const u32x2 gq = u32x2{pack_bf2(g[0], g[1]), pack_bf2(g[2], g[3])}, uq = u32x2{pack_bf2(u[0], u[1]), pack_bf2(u[2], u[3])};
            if (gu) {
                *reinterpret_cast<u32x2*>(gu + (size_t)m * p.ldc + n) = gq;
                *reinterpret_cast<u32x2*>(gu + (size_t)m * p.ldc + ff + n) = uq;
            }
            u32x2 o;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float ga = bflo(gq[e]), gb = bfhi(gq[e]);
                const float sa = bf2f(f2bf(ga * sigmoid_fast(ga))), sb = bf2f(f2bf(gb * sigmoid_fast(gb)));
                o[e] = pack_bf2(sa * bflo(uq[e]), sb * bfhi(uq[e]));
            }
            *reinterpret_cast<u32x2*>(act + (size_t)m * p.ldres + n) = o;
        }
        return;
    }
    // ---- lane owns C[m = 16 i + fr][n = n0 + 32 wave + 16 j + 4 fq + 0..3]
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + i * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + wave * (BN / 4) + j * 16 + fq * 4;
            if (n >= p.N) continue;                       // N % 4 == 0 (host check)
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (p.splits > 1) {
                *reinterpret_cast<f32x4*>(p.ws + ((size_t)sp * p.M + m) * p.N + n) = f32x4{v[0], v[1], v[2], v[3]};
                continue;
            }
            if (p.flags & MOLLY_GEMM_BIAS) {
                const u32x2 b = *reinterpret_cast<const u32x2*>(p.bias + n);
                v[0] += bflo(b[0]); v[1] += bfhi(b[0]); v[2] += bflo(b[1]); v[3] += bfhi(b[1]);
            }
            if (p.flags & MOLLY_GEMM_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]);
            }
            if (p.flags & MOLLY_GEMM_RESIDUAL) {
                const u32x2 b = *reinterpret_cast<const u32x2*>(p.res + (size_t)m * p.ldres + n);
                v[0] += bflo(b[0]); v[1] += bfhi(b[0]); v[2] += bflo(b[1]); v[3] += bfhi(b[1]);
            }
            if (p.flags & MOLLY_GEMM_OUT_F32) {
                float* c = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n;
                if (p.flags & MOLLY_GEMM_ACCUMULATE) {
                    const f32x4 o = *reinterpret_cast<const f32x4*>(c);
                    v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3];
                }
                *reinterpret_cast<f32x4*>(c) = f32x4{v[0], v[1], v[2], v[3]};
            } else {
                bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n;
                if (p.flags & MOLLY_GEMM_ACCUMULATE) {
                    const u32x2 o = *reinterpret_cast<const u32x2*>(c);
                    v[0] += bflo(o[0]); v[1] += bfhi(o[0]); v[2] += bflo(o[1]); v[3] += bfhi(o[1]);
                }
                *reinterpret_cast<u32x2*>(c) = u32x2{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
            }
        }
    }
}

// out[m,n] (+)= sum_s slab[s][m][n]   (split-K combine; 4 elements per thread)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, long MN, int M, int N,
                                                            void* C, int ldc, int out_f32, int accumulate,
                                                            const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res,
                                                            int ldres, int gelu) {
    // slab = [M rows][N cols] of the OUTPUT matrix as stored (for transposed output the caller passes rows = N_gemm);
    // the epilogue the one-pass kernel would have applied (bias -> GELU -> residual -> accumulate) is applied here
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < MN; i += (long)gridDim.x * 1024) {
        f32x4 a = *reinterpret_cast<const f32x4*>(ws + i);
        for (int s = 1; s < splits; ++s) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(ws + (size_t)s * MN + i);
            a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
        }
        const long m = i / N;
        const int n = (int)(i % N);
        if (bias) {
            const u32x2 b = *reinterpret_cast<const u32x2*>(bias + n);
            a[0] += bflo(b[0]); a[1] += bfhi(b[0]); a[2] += bflo(b[1]); a[3] += bfhi(b[1]);
        }
        if (gelu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = gelu_erf(a[e]);
        }
        if (res) {
            const u32x2 b = *reinterpret_cast<const u32x2*>(res + m * ldres + n);
            a[0] += bflo(b[0]); a[1] += bfhi(b[0]); a[2] += bflo(b[1]); a[3] += bfhi(b[1]);
        }
        if (out_f32) {
            float* c = reinterpret_cast<float*>(C) + m * ldc + n;
            if (accumulate) {
                const f32x4 o = *reinterpret_cast<const f32x4*>(c);
                a[0] += o[0]; a[1] += o[1]; a[2] += o[2]; a[3] += o[3];
            }
            *reinterpret_cast<f32x4*>(c) = a;
        } else {
            bf16_t* c = reinterpret_cast<bf16_t*>(C) + m * ldc + n;
            if (accumulate) {
                const u32x2 o = *reinterpret_cast<const u32x2*>(c);
                a[0] += bflo(o[0]); a[1] += bfhi(o[0]); a[2] += bflo(o[1]); a[3] += bfhi(o[1]);
            }
            *reinterpret_cast<u32x2*>(c) = u32x2{pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3])};
        }
    }
}

// ---- tails of a K-sliced decode-row GEMM (molly_gemm_rows_tail_bf16_ctx): the slab combine together with what the decode step
// launches right behind the projection, so that one launch replaces two or three of the ~16 per layer.
// RMSNorm tail: one workgroup of 1024 threads per output row (N <= 8192).  y = sum of the slices (+ bias) (+ residual), rounded to
// bf16 and stored — the row the unfused path would hand to the norm kernel — then yn = bf16(bf16(y * rstd) * gain), the roundings of
// rmsnorm_fwd_kernel (HF:models/qwen3/modeling_qwen3.py:50-63).
__global__ __launch_bounds__(1024) void rows_tail_norm_kernel(const float* __restrict__ ws, int splits, int M, int N, bf16_t* C, int ldc,
                                                              const bf16_t* __restrict__ bias, const bf16_t* __restrict__ res, int ldres,
                                                              const bf16_t* __restrict__ gain, float eps, bf16_t* out, int ldo) {
    __shared__ float red[16];
    const int m = blockIdx.x, tid = threadIdx.x;
    const long MN = (long)M * N;
    float v[2][4];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int n = (c * 1024 + tid) * 4;
        if (n >= N) continue;
        const float* src = ws + (size_t)m * N + n;
        f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
        int s2 = 0;
        for (; s2 + 4 <= splits; s2 += 4) {                       // four slices' loads in flight, added in slice order
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(src + (size_t)s2 * MN);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(src + (size_t)(s2 + 1) * MN);
            const f32x4 a2 = *reinterpret_cast<const f32x4*>(src + (size_t)(s2 + 2) * MN);
            const f32x4 a3 = *reinterpret_cast<const f32x4*>(src + (size_t)(s2 + 3) * MN);
            a += a0; a += a1; a += a2; a += a3;
        }
        for (; s2 < splits; ++s2) a += *reinterpret_cast<const f32x4*>(src + (size_t)s2 * MN);
        if (bias) {
            const u32x2 b = *reinterpret_cast<const u32x2*>(bias + n);
            a[0] += bflo(b[0]); a[1] += bfhi(b[0]); a[2] += bflo(b[1]); a[3] += bfhi(b[1]);
        }
        if (res) {
            const u32x2 b = *reinterpret_cast<const u32x2*>(res + (size_t)m * ldres + n);
            a[0] += bflo(b[0]); a[1] += bfhi(b[0]); a[2] += bflo(b[1]); a[3] += bfhi(b[1]);
        }
        const u32x2 y = u32x2{pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3])};
        *reinterpret_cast<u32x2*>(C + (size_t)m * ldc + n) = y;
        v[c][0] = bflo(y[0]); v[c][1] = bfhi(y[0]); v[c][2] = bflo(y[1]); v[c][3] = bfhi(y[1]);
#pragma unroll
        for (int e = 0; e < 4; ++e) ss += v[c][e] * v[c][e];
    }
    ss = wave_sum(ss);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) tot += red[w];                   // (every thread adds the 16 wave sums in the same order)
    const float rstd = rsqrtf(tot / (float)N + eps);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int n = (c * 1024 + tid) * 4;
        if (n >= N) continue;
        const u32x2 g = *reinterpret_cast<const u32x2*>(gain + n);
        const uint32_t t0 = pack_bf2(v[c][0] * rstd, v[c][1] * rstd), t1 = pack_bf2(v[c][2] * rstd, v[c][3] * rstd);
        *reinterpret_cast<u32x2*>(out + (size_t)m * ldo + n) =
            u32x2{pack_bf2(bflo(t0) * bflo(g[0]), bfhi(t0) * bfhi(g[0])), pack_bf2(bflo(t1) * bflo(g[1]), bfhi(t1) * bfhi(g[1]))};
    }
}

// SwiGLU tail: C = [gate | up] [M][2 ff] (the GEMM's output, bf16), out = silu(gate) * up [M][ff] with swiglu_fwd_kernel's roundings
// (silu rounded to bf16, then the product: HF:models/qwen3/modeling_qwen3.py:76-83); 4 activation columns per thread.
__global__ __launch_bounds__(256) void rows_tail_swiglu_kernel(const float* __restrict__ ws, int splits, int M, int N, bf16_t* C, int ldc,
                                                               const bf16_t* __restrict__ bias, bf16_t* out, int ldo) {
    const int ff = N >> 1;
    const long MN = (long)M * N;
    const long total = (long)M * (ff >> 2);
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int m = (int)(t / (ff >> 2)), n = (int)(t % (ff >> 2)) * 4;
        const float* src = ws + (size_t)m * N + n;
        f32x4 g = *reinterpret_cast<const f32x4*>(src), u = *reinterpret_cast<const f32x4*>(src + ff);
        for (int s2 = 1; s2 < splits; ++s2) {
            g += *reinterpret_cast<const f32x4*>(src + (size_t)s2 * MN);
            u += *reinterpret_cast<const f32x4*>(src + (size_t)s2 * MN + ff);
        }
        if (bias) {
            const u32x2 b = *reinterpret_cast<const u32x2*>(bias + n), b2 = *reinterpret_cast<const u32x2*>(bias + ff + n);
            g[0] += bflo(b[0]); g[1] += bfhi(b[0]); g[2] += bflo(b[1]); g[3] += bfhi(b[1]);
            u[0] += bflo(b2[0]); u[1] += bfhi(b2[0]); u[2] += bflo(b2[1]); u[3] += bfhi(b2[1]);
        }
        const u32x2 gq = u32x2{pack_bf2(g[0], g[1]), pack_bf2(g[2], g[3])}, uq = u32x2{pack_bf2(u[0], u[1]), pack_bf2(u[2], u[3])};
        if (C) {
            *reinterpret_cast<u32x2*>(C + (size_t)m * ldc + n) = gq;
            *reinterpret_cast<u32x2*>(C + (size_t)m * ldc + ff + n) = uq;
        }
        u32x2 o;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float ga = bflo(gq[e]), gb = bfhi(gq[e]);
            const float sa = bf2f(f2bf(ga * sigmoid_fast(ga))), sb = bf2f(f2bf(gb * sigmoid_fast(gb)));
            o[e] = pack_bf2(sa * bflo(uq[e]), sb * bfhi(uq[e]));
        }
        *reinterpret_cast<u32x2*>(out + (size_t)m * ldo + n) = o;
    }
}

// qkv tail: the fused q | k | v projection of one decode step (one row per sample) -> per-head RMSNorm of the q and k heads, rotary,
// q | k to `dst`, and the KV-cache append (k after norm + rotary, v as it is, to cache row slot[m]) — the arithmetic and roundings of
// norm_rope_fwd_kernel (elementwise.hip; HF:models/qwen3/modeling_qwen3.py:225-236, 148-170) on the slice sums rounded to bf16 first,
// as the GEMM's own bf16 output would be.  hd / 8 threads per head, each 4 consecutive (i, i + hd/2) pairs.
struct QkvTail {
    const bf16_t* qw; const bf16_t* kw; const float* cos; const float* sin; const int* pos;
    bf16_t* dst; bf16_t* kc; bf16_t* vc; const int* slot;
    int nq, nk, hd, ld_dst, ld_cache;
    float eps;
};
__global__ __launch_bounds__(256) void rows_tail_qkv_kernel(const float* __restrict__ ws, int splits, int M, int N,
                                                            const bf16_t* __restrict__ bias, QkvTail t) {
    const int half = t.hd >> 1, tph = half >> 2;
    const int i = (threadIdx.x % tph) * 4;
    const long item = (long)blockIdx.x * (256 / tph) + threadIdx.x / tph;
    const int nh = t.nq + 2 * t.nk;
    const bool live = item < (long)M * nh;
    const int m = live ? (int)(item / nh) : 0, head = live ? (int)(item % nh) : 0;
    const long MN = (long)M * N;
    float x1[4] = {0, 0, 0, 0}, x2[4] = {0, 0, 0, 0};
    if (live) {
        const float* src = ws + (size_t)m * N + head * t.hd + i;
        f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + half);
        for (int s2 = 1; s2 < splits; ++s2) {
            a += *reinterpret_cast<const f32x4*>(src + (size_t)s2 * MN);
            b += *reinterpret_cast<const f32x4*>(src + (size_t)s2 * MN + half);
        }
        if (bias) {
            const u32x2 ba = *reinterpret_cast<const u32x2*>(bias + head * t.hd + i), bb = *reinterpret_cast<const u32x2*>(bias + head * t.hd + i + half);
            a[0] += bflo(ba[0]); a[1] += bfhi(ba[0]); a[2] += bflo(ba[1]); a[3] += bfhi(ba[1]);
            b[0] += bflo(bb[0]); b[1] += bfhi(bb[0]); b[2] += bflo(bb[1]); b[3] += bfhi(bb[1]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { x1[e] = bf2f(f2bf(a[e])); x2[e] = bf2f(f2bf(b[e])); }      // the projection's bf16 output
    }
    if (head >= t.nq + t.nk) {                              // a v head (whole groups of tph lanes): to the cache as it is
        if (live) {
            bf16_t* d = t.vc + (size_t)t.slot[m] * t.ld_cache + (head - t.nq - t.nk) * t.hd;
            *reinterpret_cast<u32x2*>(d + i) = u32x2{pack_bf2(x1[0], x1[1]), pack_bf2(x1[2], x1[3])};
            *reinterpret_cast<u32x2*>(d + i + half) = u32x2{pack_bf2(x2[0], x2[1]), pack_bf2(x2[2], x2[3])};
        }
        return;
    }
    const bool isq = head < t.nq;
    const bf16_t* w = isq ? t.qw : t.kw;
    if (w) {
        const float ss = head_lanes_sum(head_sumsq8(x1, x2), tph);        // (common.h: norm_rope_fwd_kernel's arithmetic)
        head_norm8(x1, x2, rsqrtf(ss / (float)t.hd + t.eps), w, i, half);
    }
    float y1[4], y2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { y1[e] = x1[e]; y2[e] = x2[e]; }
    if (t.cos) {
        const int pos = t.pos ? t.pos[m] : 0;
        const f32x4 c = *reinterpret_cast<const f32x4*>(t.cos + (size_t)pos * half + i);
        const f32x4 sn = *reinterpret_cast<const f32x4*>(t.sin + (size_t)pos * half + i);
        head_rope8(x1, x2, c, sn, y1, y2);
    }
    if (live) {
        const u32x2 o1 = u32x2{pack_bf2(y1[0], y1[1]), pack_bf2(y1[2], y1[3])}, o2 = u32x2{pack_bf2(y2[0], y2[1]), pack_bf2(y2[2], y2[3])};
        bf16_t* d = t.dst + (size_t)m * t.ld_dst + head * t.hd;
        *reinterpret_cast<u32x2*>(d + i) = o1;
        *reinterpret_cast<u32x2*>(d + i + half) = o2;
        if (!isq) {
            bf16_t* c = t.kc + (size_t)t.slot[m] * t.ld_cache + (head - t.nq) * t.hd;
            *reinterpret_cast<u32x2*>(c + i) = o1;
            *reinterpret_cast<u32x2*>(c + i + half) = o2;
        }
    }
}

// ---- launch state.  Everything a launch decision reads lives in a CONTEXT: the tuning knobs, the scratch memory and the record
// of the last configuration.  molly_gemm_ctx_* create and edit contexts; the *_ctx entry points launch through one.  The entry
// points without a context use the calling thread's default context (thread_local: what one host thread sets, another never
// inherits), which is what the molly_gemm_set_* setters edit.
struct GemmCtx {
    int group_m = 4;
    int schedule = -1;             // 256x256 kernel: -1 / 0 = four phases per K-tile, 1 = two phases per K-tile (A/B knob)
    int persist_blocks = 256;      // 256x256 kernel: resident blocks (1 per CU); 0 = one block per tile; -t = t tiles per block
    int small_grid_tile = 128;     // with stream-K off: kernel for grids that fill the chip neither plain nor split (128 | 512)
    int min_ktiles = 16;           // split-K: shortest K-slice (in 64-wide K-tiles) of a grid that is not skinny
    int force_tile = 0;            // 0 heuristic | 128 | 512
    int small3 = 0;                // 1 = 128x128 grids of at most one tile per CU on the 3-stage ring; 0 (default): the 2-stage loop — measured equal
                                   // (tools/bench_esm_gemm.py: 22.4 / 24.4 / 27.5 us against 21.8 / 22.7 / 27.5 at M = 1024): these launches are not
                                   // bound by the LDS-DMA drain
    int rows_gu = 1;               // decode rows of gate | up with SwiGLU (tail 2, M <= 32): the ONE-slice kernel that forms the activation from its
                                   // accumulators — 1: on, W rows per tile chosen by the launcher; 32 | 64 | 128: that tile; 0 = K slices through slabs +
                                   // the combine launch (round 3's path; A/B)
    int rows_bn = 64;              // W rows per tile of the tiled decode-row kernel at M <= 32: 64 (48 KB of LDS: three workgroups per CU, twice the column
                                   // tiles, fewer K slices and slabs) | 128 (round 3).  M = 32, us at 128 | 64 (tools/bench_decode_gemm.py, profiles/r04_logs/
                                   // rows_bn_bench.log): 8B qkv 17.8 | 16.6, o 15.6 | 13.6, down 27.9 | 26.4; 4B down 21.7 | 19.6; 1.7B gate|up 17.6 | 16.5;
                                   // decode step 8B B = 32 6.76-6.81 | 6.63-6.65 ms, B = 20 6.52 | 6.29, 4B 5.55 | 5.42, 1.7B 3.81 | 3.76
    int rows_max_m = 1024;         // largest M the tiled decode-row kernel takes (64-row tiles; > 64 only where the 128x128 grid has <= 192 blocks:
                                   // 512 rows: 20.8 -> 15.8 us qkv, 30.0 -> 24.4 ffn2; at 2048 rows it loses to split-K: tools/bench_esm_gemm.py)
    int rows_tiled = 1;            // 1 = M <= 64 forward GEMMs the weight-streaming kernel does not take run on the tiled decode-row kernel; 0 = split-K
                                   // through the 256x256 kernel (round 2's path; A/B)
    int skinny = 1;                // 1 = M <= 64 forward GEMMs (decode rows) on the weight-streaming kernel; 0 = split-K through the tile kernel (A/B)
    int dynamic_min_work = 257;    // the dynamic fetch applies to launches of at least this many work items (default: more than one round)
    int stream_epi = [] { const char* e = getenv("MOLLY_GEMM_STREAM_EPI"); return e ? atoi(e) : 1; }();
                                   // 1 = plain NT launches of whole interior 256x256 tiles run gemm256_kernel<SE>: un-swapped MFMA operands, B rows interleaved
                                   // at staging, each row half stored as whole 128-byte lines from the K loop's load segments, the loop running on into the
                                   // next tile; 0 = the epilogue after the K loop (A/B).  Same-process A/B at M = 32,768 (profiles/r06_logs/ab_se_5.log):
                                   // qkv forward 401.3 -> 383.6 us (1,370 -> 1,433 TF/s; hipBLASLt 1,386), 12,288 x 2,048: 1,168 -> 1,128 us; bit-identical
    int small_split = 1;           // 1 = small grids with long contractions priced for split-K (see launch_cfg); 0 = round 2's rule (A/B)
    int dynamic = 0;               // 1 = plain 256x256 launches of more than one round draw their tiles (gemm256_kernel<DYN>): for GEMMs that
                                   // run beside a collective's kernels (ranks of a multi-GPU job)
    int streamk = 1;               // 1 = stream-K where its cost model says it wins (M, N >= 256); 2 = wherever it is able (tests); 0 = off
    int last_cfg = 0;              // 128 / 512 + 1000 * split-K factor (+ 50000: stream-K, + 100000 * problems: grouped)
    float* ws = nullptr;           // scratch: [stream-K header: error word + flags][fp32 slabs of split-K / stream-K]
    size_t ws_bytes = 0;
};
thread_local GemmCtx t_ctx;
inline GemmCtx& ctx_of(void* h) { return h ? *static_cast<GemmCtx*>(h) : t_ctx; }

constexpr int SK_MAX_BLOCKS = 2048;
constexpr int SK_MAX_TILES = 8192;                                     // one counter line per TILE of a stream-K launch
constexpr size_t DYN_CNT_OFF = 64 + (size_t)SK_MAX_TILES * 64;         // the dynamic fetch's 8 ticket counters, one line each
constexpr size_t SK_HDR_BYTES = DYN_CNT_OFF + 8 * 64;                  // error word line + one 64-byte line per counter
constexpr size_t SK_SLAB_BYTES = 2 * 262144;                           // per block: two pieces of 8 waves x 32 quads x 64 lanes x 16 B
inline float* ws_slabs(const GemmCtx& c) { return c.ws ? reinterpret_cast<float*>(reinterpret_cast<char*>(c.ws) + SK_HDR_BYTES) : nullptr; }
inline size_t ws_slab_bytes(const GemmCtx& c) { return c.ws_bytes > SK_HDR_BYTES ? c.ws_bytes - SK_HDR_BYTES : 0; }

// Per-form default.  Round 1 ran the k-major-B forms (dgrad, wgrad) on the two-phase schedule, measured +3-4 % — while hipcc
// was draining the LDS-DMA pipeline in front of their transposed reads (dma16 above).  With the DMA issued from asm the
// four-phase schedule wins on every form (same-box: gate|up dgrad 1387 vs 1320 TF/s, qkv dgrad 1296 vs 1279), so -1 now
// means four phases everywhere; 1 still selects two.
inline bool two_phase(const GemmCtx& c) { return c.schedule == 1; }
// grid of the 256x256 kernel for `nwork` work items.  n > 0: min(nwork, n) resident blocks that walk the work list (every CU
// must be free for the whole launch, or the blocks that found none run as a second round).  0: one block per item.  -t: blocks of
// at most t tiles each, a multiple of 256 of them (whole rounds when the chip is free; a block keeps its XCD) — the dispatcher
// places them on whatever CUs are free (a collective running beside the GEMM owns some), and t - 1 of every t tile boundaries
// still run under the rolling prefetch.
inline int grid256(const GemmCtx& c, int nwork) {
    if (c.persist_blocks > 0) return min(nwork, c.persist_blocks);
    if (c.persist_blocks == 0) return nwork;
    const int t = -c.persist_blocks;
    return min(nwork, 256 * cdiv(nwork, 256 * t));                  // whole rounds of 256 blocks, at most t tiles per block
}
// blocks of a STREAM-K launch over `ntile` tiles of `upt` units: the context's resident-block count (256 in the modes that have
// none) — always a multiple of 8 (the XCD labels), never more than one block per two units, and within what the scratch memory
// holds slabs for.
inline int grid_sk(const GemmCtx& c, int ntile, int upt) {
    // (never more than 256: the scratch every context is given holds 256 blocks' slabs, so the cut — and with it the fp32 order of
    // the sums — does not change when a caller later grows the workspace; found by the config-5 test under MOLLY_TEST_GEMM_BLOCKS=0,
    // where the second prefill of a batch was cut over 512 blocks and its argmax differed on two near-ties)
    int g = c.persist_blocks > 0 ? c.persist_blocks : 256;
    g = min(g, 256);
    g = min(g, (int)min((long)ntile * upt / 2, (long)(ws_slab_bytes(c) / SK_SLAB_BYTES)));
    g = min(g, 6 * ntile);                       // a tile in at most 6 + 2 pieces: the reducer's list holds 10, and it reads them one by one
    return g / 8 * 8;
}

__device__ bf16_t g_zero_page[64];      // zero-initialised device memory (k-rows beyond K)

template <bool AT, bool BT, bool TO = false>
int launch_cfg(hipStream_t st, GemmCtx& c, GemmArgs& p, int force_tile) {
    // force_tile: 0 heuristic | 128 = gemm_kernel<128, 2 stages, BK 64> | 512 = gemm256_kernel
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_kernel<AT, BT, 128, 2, 64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  2 * (128 + BN) * 64 * 2);
        (void)hipFuncSetAttribute((const void*)gemm_kernel<AT, BT, 128, 3, 64>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  3 * (128 + BN) * 64 * 2);
        (void)hipFuncSetAttribute((const void*)gemm256_kernel<AT, BT, TO, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        (void)hipFuncSetAttribute((const void*)gemm256_kernel<AT, BT, TO, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        (void)hipFuncSetAttribute((const void*)gemm256_kernel<AT, BT, TO, false, false, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        (void)hipFuncSetAttribute((const void*)gemm256_kernel<AT, BT, TO, false, false, false, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        if constexpr (!TO && !AT && !BT) {
            (void)hipFuncSetAttribute((const void*)gemm256_kernel<false, false, false, false, false, false, false, false, 1>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        }
        attr_set = true;
    }
    // heuristic.  Whole rounds of the 256 CUs: the persistent 256x256 kernel, one tile after another.  Anything else with both
    // sides at least one tile wide: the same kernel as STREAM-K (one launch, every CU an equal share of the K-tiles).  One
    // side narrower than a tile (rank-r adapters, decode rows): a pure operand stream — K split over the chip into fp32 slabs +
    // the reduce launch, or the 128x128 kernel.
    p.splits = 1;
    p.ws = nullptr;
    p.sk = 0;
    p.group_m = c.group_m;
    const long t256 = (long)cdiv(p.M, 256) * cdiv(p.N, 256);
    const int nk = cdiv(p.K, 64);
    auto eff = [](long items) { return (double)items / (double)(((items + 255) / 256) * 256); };
    bool use_sk = false;
    const bool sk_able = c.streamk && !two_phase(c) && p.M >= 256 && p.N >= 256 && nk >= 4 && t256 <= SK_MAX_TILES &&
                         ws_slab_bytes(c) >= 8 * SK_SLAB_BYTES && !(p.flags & (MOLLY_GEMM_SWIGLU | MOLLY_GEMM_SWIGLU_BWD));
    if (force_tile == 0) {
        if (t256 >= 200 && eff(t256) >= 0.8) {
            force_tile = 512;
        } else {
            const bool plain_epilogue = !(p.flags & (MOLLY_GEMM_BIAS | MOLLY_GEMM_GELU | MOLLY_GEMM_RESIDUAL));
            // "skinny": one side of the output is narrower than a tile (rank-r adapter GEMMs: N = r or M = r) and the
            // 128x128 grid would leave CUs empty.  Such a problem is bound by streaming the big operand once, so what
            // matters is blocks (and bytes in flight) on every CU: split K finer (>= 8 K-tiles per slice) and further.
            const long t128 = (long)cdiv(p.M, 128) * cdiv(p.N, BN);
            // (decode GEMMs, M = batch rows: a pure weight stream — slices down to 4 K-tiles)
            const bool skinny = (p.M < 256 || p.N < 256) && t128 < 256;
            const int min_kt = skinny ? ((p.M < 64 || p.N < 64) ? 4 : 8) : c.min_ktiles;
            int best = 0;
            double best_score = 0.0;
            for (int sp : {2, 3, 4, 6, 8, 12, 16, 24, 32}) {
                if (nk / sp < min_kt) break;                               // keep slices >= 1024 (skinny: 512) deep
                if ((size_t)sp * p.M * p.N * sizeof(float) > ws_slab_bytes(c)) break;
                if (t256 * sp >= 200 && eff(t256 * sp) >= 0.85) { best = sp; break; }
                if (skinny && eff(t256 * sp) > best_score + 1e-9) { best_score = eff(t256 * sp); best = -sp; }
            }
            if (best < 0) best = skinny && best_score > eff(t128) ? -best : 0;   // nothing fills the chip: the fullest
            // A SMALL grid with a long contraction (the encoders' ffn2 at one sample per GPU: 512 x 1280 x 5120 = 40 blocks of
            // 128x128 walking 80 K-tiles each: 105 us for 6.7 GFLOP): no slice count reaches the 200 work items the rule above
            // wants, so it fell through to the 128x128 kernel.  Priced instead, with the constants of the stream-K model below
            // (conservative for split-K: they overestimate it by ~15 % on the shapes measured): slices down to 4 K-tiles.
            if (best == 0 && !skinny && c.small_split && p.M >= 256 && p.N >= 256 && t128 <= 256) {
                double best_t = 0.9 * (5.0 + 1.25 * nk);
                for (int sp : {2, 3, 4, 6, 8, 12, 16, 24, 32}) {
                    if (nk / sp < 4) break;
                    if ((size_t)sp * p.M * p.N * sizeof(float) > ws_slab_bytes(c)) break;
                    const double t = 12.0 + cdiv(t256 * sp, 256) * (6.0 + 1.4 * nk / sp) + 4.0 + 8.0 * p.M * p.N * sp / 9e6;
                    if (t < best_t) { best_t = t; best = sp; }
                }
            }
            // ONE partial round instead of K slices (round 6): a grid of 128..199 tiles keeps at least half the CUs busy by itself, and a CU that runs
            // beside idle ones walks its K-tiles faster (1.2 us per K-tile measured against 1.4 on a full chip) — while the slices pay their slabs and
            // the reduce launch.  The encoders at the headline's 8,224 rows: ffn2 (165 tiles x 80 K-tiles) 107.4 us as one round against 125.7 as
            // three slices, o-proj (165 x 20) 47.6 against 54.9 on the 128x128 kernel (profiles/r06_logs/esm_gemm_a.log)
            if (best > 0 && !skinny && c.small_split && t256 >= 128 && t256 < 200 && p.M >= 256 && p.N >= 256) {
                const double t_one = 18.0 + 1.25 * nk;
                const double t_split = 12.0 + cdiv(t256 * best, 256) * (6.0 + 1.4 * nk / best) + 4.0 + 8.0 * p.M * p.N * best / 9e6;
                if (t_one < t_split) best = 0;
            }
            // the slab path applies bias/GELU/residual in its reduce kernel (not for the transposed-output form)
            if (best && (plain_epilogue || !TO) && p.N % 4 == 0 && ((p.M >= 256 && p.N >= 256) || skinny)) {
                force_tile = 512;
                p.splits = best;
                p.ws = ws_slabs(c);
            } else {
                force_tile = TO ? 512 : c.small_grid_tile;     // transposed output exists in the 256x256 kernel only
                // a grid of 128..199 tiles whose contraction is too short to slice (Qwen3-4B's o_proj dgrad at one sample per GPU: 3,072 x 4,096 x
                // 2,560 = 192 tiles x 40 K-tiles): ONE round of the 256x256 kernel on three quarters of the CUs beats 768 tiles of the 128x128
                // kernel (109.7 us = 587 TFLOP/s measured there, profiles/r04_logs/c3_gemm_table.txt) — priced with the constants used below
                static const int partial_round = [] { const char* e = getenv("MOLLY_GEMM_PARTIAL_ROUND"); return e ? atoi(e) : 1; }();
                // (the 128x128 kernel's last, partly filled round costs a whole one: 650 blocks x 20 K-tiles measured 54.9 us, 1,040 blocks 66.2 —
                // 1.3 us per K-tile and round, half a round on top: round 6, profiles/r06_logs/esm_gemm_a.log)
                if (partial_round && force_tile == 128 && t256 >= 128 && t256 < 256 && p.M >= 256 && p.N >= 256 &&
                    12.0 + 6.0 + 1.25 * nk < 5.0 + (t128 > 512 ? (t128 / 512.0 + 0.5) * 1.3 : 1.25) * nk)
                    force_tile = 512;
            }
        }
        {
            const long t128 = (long)cdiv(p.M, 128) * cdiv(p.N, BN);
            // ---- stream-K instead?  Its hand-off has a price on this chip: a piece's accumulators are 256 KiB written through to
            // memory (~21 GB/s per workgroup: guide price list 'publish-large') and read back by the reducer — measured 30-36 us
            // per launch (tools/bench_streamk.py) against ~1.4 us per K-tile of work.  So it pays where the alternatives waste more
            // than that: long contractions on grids just past a whole round (Qwen3-8B qkv at B = 1: 384 tiles, 162 us against 197
            // for split-K and its 200 MB of slabs; Qwen3-4B qkv: 288 tiles, 99 against 105), not on the encoders' 20-K-tile
            // projections.  Times in us from constants fitted to those measurements; stream-K must win by 7 %.
            if (sk_able && eff(t256) < 0.9) {
                const double kt = 1.4, launch = 12.0, tile_fix = 6.0;
                double t_other;
                if (force_tile == 512 && p.splits > 1)
                    t_other = launch + cdiv(t256 * p.splits, 256) * (tile_fix + kt * nk / p.splits) + 8.0 * p.M * p.N * p.splits / 9e6 + 4.0;
                else if (force_tile == 512)
                    t_other = launch + cdiv(t256, 256) * (tile_fix + kt * nk);
                else
                    t_other = 5.0 + (t128 > 512 ? (t128 / 512.0 + 0.35) * 1.17 : 1.25) * nk;      // 128x128 kernel, two blocks per CU
                // (14 us: a block's slab writes; 8 us per piece the reducer reads — two when the shares are a tile or more)
                const int g_sk = grid_sk(c, (int)t256, nk / 2);
                const double pieces = g_sk > t256 ? (double)g_sk / t256 + 1.0 : 2.0;
                const double t_sk = launch + tile_fix + kt * nk * (double)t256 / (g_sk > 0 ? g_sk : 1) + 14.0 + 8.0 * pieces;
                if (c.streamk == 2 || (g_sk >= 8 && (double)nk * t256 / g_sk >= 8.0 && t_sk < 0.93 * t_other)) {
                    force_tile = 512;
                    p.splits = 1;
                    p.ws = nullptr;
                    use_sk = true;
                }
            }
        }
    }
    if (force_tile == 512) {
        p.tiles_m = cdiv(p.M, 256); p.tiles_n = cdiv(p.N, 256);
        if (use_sk) {
            const int ntile = p.tiles_m * p.tiles_n;
            p.sk = 2;                                              // a cut falls on even K-tiles: every piece >= 2 K-tiles
            const int upt = nk / p.sk;
            const int grid = grid_sk(c, ntile, upt);
            if (grid >= 8) {
                // the 8 XCD labels own whole tiles when that costs < 3 % of balance; else their borders fall inside tiles too
                const double per = ntile / 8.0;
                p.sk_tile_aligned = (cdiv(ntile, 8) / per - 1.0) <= 0.03 ? 1 : 0;
                p.sk_flag = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(c.ws) + 64);
                p.sk_slab = ws_slabs(c);
                c.last_cfg = 512 + 1000 + 50000;
                hipLaunchKernelGGL((gemm256_kernel<AT, BT, TO, false, false, true>), dim3(grid), dim3(512), 163840, st, p);
                return 0;
            }
            p.sk = 0;                                              // (no room for slabs: the plain launch below)
        }
        c.last_cfg = force_tile + 1000 * p.splits;
        // persistent: at most persist_blocks (one per CU; a multiple of 8 so a block keeps its XCD across rounds)
        const int nwork = p.tiles_m * p.tiles_n * p.splits;
        const int grid = grid256(c, nwork);
        if (c.dynamic && !two_phase(c) && c.ws && c.ws_bytes > SK_HDR_BYTES /* counters exist and were cleared */ && nwork > 256 && nwork >= c.dynamic_min_work && nk / p.splits >= 8) {
            // 256 resident blocks (32 per XCD label) that draw their tiles
            p.dyn_cnt = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(c.ws) + DYN_CNT_OFF);
            c.last_cfg += 1;                                       // 513: the 256x256 kernel drawing its tiles
            hipLaunchKernelGGL((gemm256_kernel<AT, BT, TO, false, false, false, true>), dim3(256), dim3(512), 163840, st, p);
        } else if (two_phase(c)) hipLaunchKernelGGL((gemm256_kernel<AT, BT, TO, true>), dim3(grid), dim3(512), 163840, st, p);
        else {
            // the NT form on whole interior tiles, one K slice, the plain epilogue: the streaming-epilogue instantiation
            bool se = false;
            if constexpr (!TO && !AT && !BT) {
                se = c.stream_epi && p.flags == 0 && p.splits == 1 && p.M % 256 == 0 && p.N % 256 == 0 && p.K % 64 == 0 && nk >= 3 && p.ldc < (1 << 21);
            }
            if (se) {
                if constexpr (!TO && !AT && !BT) {
                    c.last_cfg += 2;                               // 514: the 256x256 kernel with the streaming epilogue
                    hipLaunchKernelGGL((gemm256_kernel<false, false, false, false, false, false, false, false, 1>), dim3(grid), dim3(512), 163840, st, p);
                }
            } else hipLaunchKernelGGL((gemm256_kernel<AT, BT, TO, false>), dim3(grid), dim3(512), 163840, st, p);
        }
        if (p.splits > 1) {
            const long MN = (long)p.M * p.N;
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)min((MN / 4 + 255) / 256, 4096L)), dim3(256), 0, st, p.ws,
                               p.splits, MN, TO ? p.N : p.M, TO ? p.M : p.N, p.C, p.ldc,
                               (p.flags & MOLLY_GEMM_OUT_F32) ? 1 : 0, (p.flags & MOLLY_GEMM_ACCUMULATE) ? 1 : 0,
                               (p.flags & MOLLY_GEMM_BIAS) ? p.bias : nullptr, (p.flags & MOLLY_GEMM_RESIDUAL) ? p.res : nullptr,
                               p.ldres, (p.flags & MOLLY_GEMM_GELU) ? 1 : 0);
        }
    } else if constexpr (TO) {
        molly_set_error("gemm: transposed output is only built into the 256x256 kernel");
        return 1;
    } else {
        c.last_cfg = 128 + 1000;
        p.tiles_m = cdiv(p.M, 128); p.tiles_n = cdiv(p.N, BN);
        // (knob, off by default: grids of at most one tile per CU on the 3-stage ring — tiles t+1, t+2 in flight behind a counted
        // vmcnt, fragments double-buffered in registers — instead of the 2-stage loop; measured equal, see GemmCtx::small3)
        if (c.small3 && nk >= 3 && p.tiles_m * p.tiles_n <= 256)
            hipLaunchKernelGGL((gemm_kernel<AT, BT, 128, 3, 64>), dim3(p.tiles_m * p.tiles_n), dim3(256),
                               3 * (128 + BN) * 64 * sizeof(bf16_t), st, p);
        else if (!AT && !BT && p.tiles_m == 1 && p.tiles_n >= 512) {
            // one row tile against a wide weight matrix (the lm_head at decode rows): the weights non-temporal
            static bool nt_attr = false;
            if (!nt_attr) {
                (void)hipFuncSetAttribute((const void*)gemm_kernel<AT, BT, 128, 2, 64, !AT && !BT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                          2 * (128 + BN) * 64 * 2);
                nt_attr = true;
            }
            hipLaunchKernelGGL((gemm_kernel<AT, BT, 128, 2, 64, !AT && !BT>), dim3(p.tiles_m * p.tiles_n), dim3(256),
                               2 * (128 + BN) * 64 * sizeof(bf16_t), st, p);
        } else
            hipLaunchKernelGGL((gemm_kernel<AT, BT, 128, 2, 64>), dim3(p.tiles_m * p.tiles_n), dim3(256),
                               2 * (128 + BN) * 64 * sizeof(bf16_t), st, p);
    }
    return 0;
}

int resolve_zero_page(const bf16_t** out) {
    static const bf16_t* zeros = nullptr;
    if (!zeros) {
        void* zp = nullptr;
        if (hipGetSymbolAddress(&zp, HIP_SYMBOL(g_zero_page)) != hipSuccess) {
            molly_set_error("gemm: cannot resolve the zero page");
            return 3;
        }
        zeros = (const bf16_t*)zp;
    }
    *out = zeros;
    return 0;
}


// ---- the tiled decode-row kernel's launcher.  tail: 0 none | 1 RMSNorm of the output rows (gain, eps -> tail_out [M][N]) | 2 SwiGLU
// of an output that is [gate | up] (-> tail_out [M][N / 2]); a tail runs inside the slab combine, so it forces >= 2 K slices.
// (round 3: not the lm_head — the 128x128 kernel streamed it at 5.0 TB/s, this one at 4.8; round 4, with 64-row tiles and non-temporal weights: rows_max_n())
// which of the two decode-row kernels takes an M <= 64 forward GEMM: the register-streaming one wherever launches, not bytes, set the
// time (matrices up to 9 M elements; M <= 16 on matrices of at most 4,096 rows), the tiled one everywhere else it applies
// (tools/bench_decode_gemm.py, gpurun_out/r03/rows_bench3.log) — ONE rule for launch_gemm and molly_gemm_rows_tail_supported
inline bool rows_applicable(const GemmCtx& c, int M, int N, int K, int flags);
inline bool streaming_rows(const GemmCtx& c, int M, int N, int K, int flags) {
    if (!(M <= 64 && c.skinny && c.force_tile == 0 && K % 256 == 0 && N % 4 == 0 && N >= 256) ||
        (flags & (MOLLY_GEMM_TRANS_OUT | MOLLY_GEMM_SWIGLU | MOLLY_GEMM_SWIGLU_BWD))) return false;
    if (!rows_applicable(c, M, N, K, flags)) return M <= 16 || (long)N * K <= (32L << 20);     // (round 3's first rule, where the tiled kernel is off)
    return (long)N * K <= (9L << 20) || (M <= 16 && N <= 4096);
}
inline int rows_max_n() {
    // (the lm_head of a decode step too since round 4: 64-row tiles + non-temporal weights stream its 1.2 GB at 5.25 TB/s, 237 us, against 4.85 TB/s,
    // 257 us, on the 128x128 kernel — round 3 measured the opposite, 4.8 against 5.0, before those two; MOLLY_ROWS_MAX_N=65536 restores)
    static const int v = [] { const char* e = getenv("MOLLY_ROWS_MAX_N"); return e ? atoi(e) : 1 << 20; }();
    return v;
}
inline bool rows_applicable(const GemmCtx& c, int M, int N, int K, int flags) {
    // (M > 64, up to the context's rows_max_m: grids that leave most CUs with less than one 128x128 block — the encoders' projections
    // at one sample per GPU — as 64-row tiles of this kernel: four times the workgroups, two per CU)
    const bool small_m = M <= 64 || (M <= c.rows_max_m && (long)cdiv(M, 128) * cdiv(N, 128) <= 192);
    return small_m && c.rows_tiled && c.force_tile == 0 && K % 64 == 0 && K >= 256 && N % 4 == 0 && N >= 128 && N < rows_max_n() &&
           !(flags & (MOLLY_GEMM_TRANS_OUT | MOLLY_GEMM_SWIGLU | MOLLY_GEMM_SWIGLU_BWD));
}
int launch_rows(GemmCtx& c, hipStream_t st, const void* A, const void* B, void* C, const void* bias, const void* res, int M, int N, int K,
                int lda, int ldb, int ldc, int ldres, int flags, int tail, const void* gain, float eps, void* tail_out, int ld_tail,
                const QkvTail* qt = nullptr) {
    RowsArgs q{(const bf16_t*)A, (const bf16_t*)B, C, (const bf16_t*)bias, (const bf16_t*)res, nullptr,
               M, N, K, lda, ldb, ldc, ldres, flags, cdiv(N, 128), 1, M <= 64 ? 1 : cdiv(M, 64)};
    if (tail == 2 && c.rows_gu && M <= 32 && (N / 2) % 64 == 0) {
        // gate | up with SwiGLU, one K slice: the activation from the accumulators, no slabs and no combine launch (gemm_rows_kernel<.., GU>)
        // W rows per tile (tools/r04/bench_rows_gu.py, profiles/r04_logs/rows_gu_bench2.log; M = 32, us at 32 | 64 | 128, K slices + combine launch last):
        // Qwen3-8B (201 MB) 41.6 | 47.5 | 45.5, 47.5; ff 16,384 x 4,096 (268 MB) 55.0 | 52.7 | 60.6, 63.6; ff 24,576 (403 MB) 79.0 | 84.4 | 86.9, 89.2;
        // Qwen3-4B (100 MB) 26.9 | 27.8 | 24.9, 27.3; 1.7B (50 MB) 14.6 | 13.3 | 16.3, 16.4 — the 150 MB+ matrices want the most workgroups (768+ of
        // 32 KB: up to five per CU), the smaller ones the fewest that still cover the chip
        const int bn = c.rows_gu > 1 ? c.rows_gu : ((long)N * K * 2 >= 150000000L && (N / 2) % 16 == 0 ? 32 : (N / 2) / 64 >= 128 ? 128 : 64);
        if (bn == 32 && (N / 2) % 16 != 0) { molly_set_error("gemm rows gate|up: ff=%d is no multiple of 16", N / 2); return 1; }
        q.tiles_n = (N / 2) / (bn / 2);
        q.res = (const bf16_t*)tail_out; q.ldres = ld_tail;
        q.bias = (flags & MOLLY_GEMM_BIAS) ? (const bf16_t*)bias : nullptr;
        static bool gu_attr = false;
        if (!gu_attr) {
            (void)hipFuncSetAttribute((const void*)gemm_rows_kernel<2, 4, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (32 + 128) * 64 * 2);
            (void)hipFuncSetAttribute((const void*)gemm_rows_kernel<2, 4, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (32 + 64) * 64 * 2);
            gu_attr = true;
        }
        if (bn == 128) hipLaunchKernelGGL((gemm_rows_kernel<2, 4, 128, true>), dim3(q.tiles_n), dim3(256), 4 * (32 + 128) * 64 * 2, st, q);
        else if (bn == 64) hipLaunchKernelGGL((gemm_rows_kernel<2, 4, 64, true>), dim3(q.tiles_n), dim3(256), 4 * (32 + 64) * 64 * 2, st, q);
        // (32-row tiles: a SIX-stage ring, 48 KB, still three workgroups per CU — 38.7 -> 37.3 us on Qwen3-8B's matrix against four stages; the 64-row
        // K-sliced kernel measured within +-3 % at 4 / 6 / 8 stages and stays at four: profiles/r04_logs/rows_stages.log)
        else hipLaunchKernelGGL((gemm_rows_kernel<2, 6, 32, true>), dim3(q.tiles_n), dim3(256), 6 * (32 + 32) * 64 * 2, st, q);
        c.last_cfg = 32 + 1000 + (bn == 32 ? 300 : 100 * (bn / 64));       // (one slice; + 100 / 200 / 300: the SwiGLU form at 64 / 128 / 32 W rows per tile)
        return 0;
    }
    const int bn = M <= 32 ? c.rows_bn : 128;
    const int slots = bn == 64 ? 768 : 512;              // resident workgroups: 48 KB of LDS each at 64 W rows per tile (three per CU), 80 KB at 128
    q.tiles_n = cdiv(N, bn);
    const int ntile = q.tiles_n * q.tiles_m;
    // K slices, priced in K-tile times of one workgroup (~0.75 us with two workgroups per CU): whole rounds of the 512 slots x
    // (slice length + ring fill) + the reduce launch and its slab traffic; slices of >= 4 K-tiles, within the scratch
    const int nk = K / 64;
    int splits = 1;
    double best_cost = 1e30;
    for (int sp = tail ? 2 : 1; sp <= 32 && (nk / sp >= 4 || (tail && sp == 2)); ++sp) {
        if (sp > 1 && (size_t)sp * M * N * sizeof(float) > ws_slab_bytes(c)) break;
        const double cost = cdiv(ntile * sp, slots) * ((double)nk / sp + 3.0) * (bn == 64 ? 0.75 : 1.0) +
                            (sp > 1 ? (5.0 + 8.0 * sp * M * N / 5e6) / 0.75 : 0.0);
        if (cost < best_cost) { best_cost = cost; splits = sp; }
    }
    if (M <= 32 && bn == 64) {
        // decode rows on 64-row tiles with non-temporal weights: as many K slices as keep the launch within ONE workgroup per CU (the largest count with
        // tiles x slices <= 256), slices of >= 4 K-tiles.  us of GEMM + combine at M = 32 by slice count (tools/r04/bench_rows_splits.py,
        // profiles/r04_logs/rows_splits.log): 8B o (64 tiles) 3: 12.0, 4: 11.2, 5: 13.3, 8: 13.6, 12: 15.6; 8B down 3: 26.1, 4: 22.4, 5: 29.9, 8: 24.2,
        // 12: 26.5; 8B qkv (96 tiles) 2: 15.9, 4: 15.8, 5: 15.4, 8: 16.8; 4B o (40 tiles) 4: 10.6, 5: 10.2, 6: 10.4, 8: 11.9, 16: 14.8; 4B down 4: 17.3,
        // 6: 15.2, 8: 18.2, 19: 20.2; 4B qkv 2: 11.8, 4: 13.2, 8: 14.7 — the priced model above (rounds of 768 resident workgroups) took 8-19 slices
        int sp = ntile > 0 ? 256 / ntile : 1;
        sp = sp > nk / 4 ? nk / 4 : sp;
        sp = sp < (tail ? 2 : 1) ? (tail ? 2 : 1) : sp;
        while (sp > 1 && (size_t)sp * M * N * sizeof(float) > ws_slab_bytes(c)) --sp;
        splits = sp;
    }
    static const int force_sp = [] { const char* e = getenv("MOLLY_ROWS_FORCE_SPLITS"); return e ? atoi(e) : 0; }();     // (A/B)
    if (force_sp > 0 && nk / force_sp >= 1 && (size_t)force_sp * M * N * sizeof(float) <= ws_slab_bytes(c)) splits = force_sp;
    if (tail && (splits < 2 || (size_t)splits * M * N * sizeof(float) > ws_slab_bytes(c))) {
        molly_set_error("gemm rows tail: the context has no scratch for %d x %d x %d slabs", splits, M, N);
        return 1;
    }
    q.splits = splits;
    q.ws = splits > 1 ? ws_slabs(c) : nullptr;
    static bool rows_attr = false;
    if (!rows_attr) {
        (void)hipFuncSetAttribute((const void*)gemm_rows_kernel<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (32 + 128) * 64 * 2);
        (void)hipFuncSetAttribute((const void*)gemm_rows_kernel<4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * (64 + 128) * 64 * 2);
        rows_attr = true;
    }
    if (M <= 32 && bn == 64) {
        static bool bn_attr = false;
        if (!bn_attr) {
            (void)hipFuncSetAttribute((const void*)gemm_rows_kernel<2, 4, 64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * (32 + 64) * 64 * 2);
            bn_attr = true;
        }
        hipLaunchKernelGGL((gemm_rows_kernel<2, 4, 64, false>), dim3(ntile * splits), dim3(256), 4 * (32 + 64) * 64 * 2, st, q);
    } else if (M <= 32) hipLaunchKernelGGL((gemm_rows_kernel<2, 4>), dim3(ntile * splits), dim3(256), 4 * (32 + 128) * 64 * 2, st, q);
    else hipLaunchKernelGGL((gemm_rows_kernel<4, 3>), dim3(ntile * splits), dim3(256), 3 * (64 + 128) * 64 * 2, st, q);
    const bf16_t* bp = (flags & MOLLY_GEMM_BIAS) ? (const bf16_t*)bias : nullptr;
    const bf16_t* rp = (flags & MOLLY_GEMM_RESIDUAL) ? (const bf16_t*)res : nullptr;
    if (tail == 1) {
        hipLaunchKernelGGL(rows_tail_norm_kernel, dim3(M), dim3(1024), 0, st, q.ws, splits, M, N, (bf16_t*)C, ldc, bp, rp, ldres,
                           (const bf16_t*)gain, eps, (bf16_t*)tail_out, ld_tail);
    } else if (tail == 3) {
        const int hpb = 256 / (qt->hd / 8);
        const long items = (long)M * (qt->nq + 2 * qt->nk);
        hipLaunchKernelGGL(rows_tail_qkv_kernel, dim3((unsigned)((items + hpb - 1) / hpb)), dim3(256), 0, st, q.ws, splits, M, N, bp, *qt);
    } else if (tail == 2) {
        const long total = (long)M * (N / 8);
        hipLaunchKernelGGL(rows_tail_swiglu_kernel, dim3((unsigned)min((total + 255) / 256, 4096L)), dim3(256), 0, st, q.ws, splits, M, N,
                           (bf16_t*)C, ldc, bp, (bf16_t*)tail_out, ld_tail);
    } else if (tail == 4) {
        // (the slabs are the result)
    } else if (splits > 1) {
        const long MN = (long)M * N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)min((MN / 4 + 255) / 256, 4096L)), dim3(256), 0, st, q.ws,
                           splits, MN, M, N, C, ldc, (flags & MOLLY_GEMM_OUT_F32) ? 1 : 0, (flags & MOLLY_GEMM_ACCUMULATE) ? 1 : 0,
                           bp, rp, ldres, (flags & MOLLY_GEMM_GELU) ? 1 : 0);
    }
    c.last_cfg = 32 + 1000 * splits;
    return 0;
}

int launch_gemm(void* ctx, void* stream, const void* A, const void* B, void* C, const void* bias, const void* res, int M, int N,
                int K, int lda, int ldb, int ldc, int ldres, int flags, bool at, bool bt) {
    MOLLY_ENTER();
    GemmCtx& c = ctx_of(ctx);
    MOLLY_CHECK(M > 0 && N > 0 && K > 0, "gemm: empty problem M=%d N=%d K=%d", M, N, K);
    MOLLY_CHECK((at && bt) || K % 64 == 0, "gemm: K=%d must be a multiple of %d when an operand is k-contiguous", K, 64);
    MOLLY_CHECK(N % 4 == 0 || (flags & MOLLY_GEMM_TRANS_OUT), "gemm: N=%d must be a multiple of 4", N);
    MOLLY_CHECK(!at || M % 8 == 0, "gemm: k-major A needs M %% 8 == 0 (M=%d)", M);
    MOLLY_CHECK(!bt || N % 8 == 0, "gemm: k-major B needs N %% 8 == 0 (N=%d)", N);
    MOLLY_CHECK(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, "gemm: lda/ldb must be multiples of 8, ldc of 4");
    MOLLY_CHECK(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)C % 16) == 0,
                "gemm: operands must be 16-byte aligned");
    MOLLY_CHECK(!(flags & MOLLY_GEMM_BIAS) || bias, "gemm: MOLLY_GEMM_BIAS without bias pointer");
    MOLLY_CHECK(!(flags & MOLLY_GEMM_RESIDUAL) || (res && ldres % 4 == 0), "gemm: bad residual");
    // ---- a grid a FEW 256 x 256 tiles past whole rounds of the chip (Qwen3-4B q | k | v at one sample per GPU: 3,072 x 6,144 x 2,560 = 288 tiles):
    // the last tile columns are carved into a launch of their own (K-sliced by the rules below) and the rest is whole rounds — 102.6 us against 128.4 for
    // stream-K on that shape (tools/r04/bench_cols_carve.py, profiles/r04_logs/cols_carve.log); every output column is still written by one launch
    if (!(flags & (MOLLY_GEMM_TRANS_OUT | MOLLY_GEMM_SWIGLU | MOLLY_GEMM_SWIGLU_BWD)) && M >= 256 && c.force_tile == 0) {
        static const int carve = [] { const char* e = getenv("MOLLY_GEMM_CARVE"); return e ? atoi(e) : 1; }();
        const long tm = cdiv(M, 256), tn = cdiv(N, 256), t = tm * tn, rem = t % 256;
        const long cols = (rem + tm - 1) / tm;
        if (carve && t > 256 && rem > 0 && rem <= 48 && cols < tn && N % 256 == 0) {
            const int n0 = (int)(tn - cols) * 256;
            const size_t esz = (flags & MOLLY_GEMM_OUT_F32) ? 4 : 2;
            const char* Bs = (const char*)B + (bt ? (size_t)n0 : (size_t)n0 * ldb) * 2;
            if (int rc = launch_gemm(ctx, stream, A, B, C, bias, res, M, n0, K, lda, ldb, ldc, ldres, flags, at, bt)) return rc;
            return launch_gemm(ctx, stream, A, Bs, (char*)C + (size_t)n0 * esz, bias ? (const char*)bias + (size_t)n0 * 2 : nullptr,
                               res ? (const char*)res + (size_t)n0 * 2 : nullptr, M, N - n0, K, lda, ldb, ldc, ldres, flags, at, bt);
        }
    }
    GemmArgs p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C;
    p.bias = (const bf16_t*)bias; p.res = (const bf16_t*)res;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldres = ldres; p.flags = flags;
    p.ngroup = 0;
    p.sk = 0; p.sk_tile_aligned = 0; p.sk_flag = nullptr; p.sk_slab = nullptr;
    if (int rc = resolve_zero_page(&p.zeros)) return rc;
    hipStream_t st = (hipStream_t)stream;
    // decode rows (M <= 64, the plain forward form): the weight-streaming kernel, one launch
    // Where it is used (tools/bench_decode_gemm.py, M = 32 / 8, same box): it wins wherever launches, not bytes, set the time —
    // matrices up to ~64 MB (every projection of the 1.7B model: 8.7 against 15.6 us for qkv; Qwen3-8B qkv / o: 19.5 / 14.2 against
    // 21.7 / 18.8) and any matrix at M <= 16 (8B gate|up at M = 8: 4.5 against 4.0 TB/s).  The 100-200 MB matrices at M = 32 stay on
    // the tile kernel (3.1-3.6 TB/s there, 5.0 on the 1.2 GB lm_head): x enters through the same 64-byte request path as W, and
    // at 32 rows it is the larger stream unless a wave keeps 64 rows of W — which leaves too few waves for the 4096-row matrices.
    if (!at && !bt && streaming_rows(c, M, N, K, flags)) {
        SkinnyArgs q{(const bf16_t*)A, (const bf16_t*)B, C, (const bf16_t*)bias, (const bf16_t*)res, M, N, K, lda, ldb, ldc, ldres, flags};
        // W rows per wave (16 NT) and K parts per workgroup (KW): as much reuse of x as still leaves >= ~1000 waves on the chip
        const int mt = M <= 16 ? 1 : M <= 32 ? 2 : 4;
        int nt = 4, kw = 4;
        auto waves = [&](int nt_, int kw_) { return (long)cdiv(N, 16 * nt_) * kw_; };
        if (waves(4, 4) < 1000) { nt = 2; kw = K % 512 == 0 ? 8 : 4; }
        if (mt == 4) nt = min(nt, 2);                        // (registers: 4 x 4 accumulator tiles + the fragments in flight)
        if (N <= 4096 || waves(nt, kw) < 600) { nt = 1; kw = 4; }   // (a 4096-row matrix: 256 workgroups of 4 waves measured best)
#define MOLLY_SKINNY(MT_, NT_, KW_) \
        hipLaunchKernelGGL((gemm_skinny_kernel<MT_, NT_, KW_>), dim3(cdiv(N, 16 * NT_)), dim3(64 * KW_), 0, st, q)
#define MOLLY_SKINNY_M(NT_, KW_) \
        do { if (mt == 1) MOLLY_SKINNY(1, NT_, KW_); else if (mt == 2) MOLLY_SKINNY(2, NT_, KW_); else MOLLY_SKINNY(4, NT_, KW_); } while (0)
        if (nt == 4) { if (mt == 1) MOLLY_SKINNY(1, 4, 4); else MOLLY_SKINNY(2, 4, 4); }
        else if (nt == 2 && kw == 8) MOLLY_SKINNY_M(2, 8);
        else if (nt == 2) MOLLY_SKINNY_M(2, 4);
        else if (kw == 8) MOLLY_SKINNY_M(1, 8);
        else MOLLY_SKINNY_M(1, 4);
#undef MOLLY_SKINNY_M
#undef MOLLY_SKINNY
        c.last_cfg = 16 + 1000;
        MOLLY_LAUNCH_CHECK();
        return 0;
    }
    // decode rows the streaming kernel left: the tiled decode-row kernel (x and W through LDS-DMA, x once per workgroup and K-tile)
    if (!at && !bt && rows_applicable(c, M, N, K, flags)) {
        const int rc = launch_rows(c, st, A, B, C, bias, res, M, N, K, lda, ldb, ldc, ldres, flags, 0, nullptr, 0.f, nullptr, 0);
        if (rc) return rc;
        MOLLY_LAUNCH_CHECK();
        return 0;
    }
    if (flags & MOLLY_GEMM_SWIGLU) {
        MOLLY_CHECK(!at && !bt && flags == MOLLY_GEMM_SWIGLU && res && N % 256 == 0 && ldres % 4 == 0,
                    "gemm: MOLLY_GEMM_SWIGLU is the plain NT form with N = 2*ff, ff %% 128 == 0 (N=%d), res = the activation output", N);
        if (launch_cfg<false, false>(st, c, p, 512)) return 1;    // the 256x256 kernel, one pass (no split-K slabs)
        MOLLY_LAUNCH_CHECK();
        return 0;
    }
    if (flags & MOLLY_GEMM_SWIGLU_BWD) {
        MOLLY_CHECK(!at && bt && flags == MOLLY_GEMM_SWIGLU_BWD && res && ldres % 4 == 0 && ldres >= 2 * N && ldc >= 2 * N,
                    "gemm: MOLLY_GEMM_SWIGLU_BWD is the dgrad form (k-contiguous A, k-major B) with N = ff (N=%d), res = [gate | up] "
                    "[M][2*ff], C = d[gate | up] [M][2*ff]", N);
        if (launch_cfg<false, true>(st, c, p, 512)) return 1;     // the 256x256 kernel, one pass (no split-K slabs)
        MOLLY_LAUNCH_CHECK();
        return 0;
    }
    if (flags & MOLLY_GEMM_TRANS_OUT) {
        MOLLY_CHECK(!at && bt, "gemm: MOLLY_GEMM_TRANS_OUT is built for the (k-contiguous A, k-major B) form only");
        MOLLY_CHECK(!(flags & (MOLLY_GEMM_BIAS | MOLLY_GEMM_GELU | MOLLY_GEMM_RESIDUAL)) && M % 4 == 0,
                    "gemm: MOLLY_GEMM_TRANS_OUT takes no bias/GELU/residual and needs M %% 4 == 0");
        if (launch_cfg<false, true, true>(st, c, p, c.force_tile == 0 ? 0 : 512)) return 1;
    } else if (!at && !bt) {
        if (launch_cfg<false, false>(st, c, p, c.force_tile)) return 1;
    } else if (!at && bt) {
        if (launch_cfg<false, true>(st, c, p, c.force_tile)) return 1;
    } else if (at && bt) {
        if (launch_cfg<true, true>(st, c, p, c.force_tile)) return 1;
    } else {
        molly_set_error("gemm: the (k-major A, k-contiguous B) form is not on the hot path and not built");
        return 1;
    }
    MOLLY_LAUNCH_CHECK();
    return 0;
}

// ---- C = A B^T + A2 B2^T (+ epilogue): the NT form of the 256x256 kernel with K2 / 64 more K-tiles read from a second operand pair.  Whole tiles of
// work only (the plain persistent launch: no K slices, no stream-K, no carve): the caller asks kx_supported() first and runs two launches otherwise.
bool kx_supported(const GemmCtx& c, int M, int N, int K, int K2, int flags) {
    const long t256 = (long)cdiv(M, 256) * cdiv(N, 256);
    return c.force_tile == 0 && !two_phase(c) && M >= 256 && N >= 256 && K % 64 == 0 && K >= 128 && K2 % 64 == 0 && K2 >= 64 && K2 <= 512 &&
           t256 >= 200 && !(flags & (MOLLY_GEMM_TRANS_OUT | MOLLY_GEMM_SWIGLU_BWD | MOLLY_GEMM_OUT_F32)) &&
           (!(flags & MOLLY_GEMM_SWIGLU) || (flags == MOLLY_GEMM_SWIGLU && N % 256 == 0));
}
int launch_gemm_kx(void* ctx, void* stream, const void* A, const void* B, void* C, const void* bias, const void* res, int M, int N, int K,
                   int lda, int ldb, int ldc, int ldres, int flags, const void* A2, const void* B2, int K2, int lda2, int ldb2) {
    MOLLY_ENTER();
    GemmCtx& c = ctx_of(ctx);
    MOLLY_CHECK(kx_supported(c, M, N, K, K2, flags), "gemm_kx: M=%d N=%d K=%d K2=%d flags=0x%x is not taken by the K-extended launch "
                "(molly_gemm_kx_supported)", M, N, K, K2, flags);
    MOLLY_CHECK(A && B && C && A2 && B2, "gemm_kx: null operand");
    MOLLY_CHECK(lda % 8 == 0 && ldb % 8 == 0 && lda2 % 8 == 0 && ldb2 % 8 == 0 && ldc % 4 == 0 && lda2 >= K2 && ldb2 >= K2,
                "gemm_kx: lda/ldb/lda2/ldb2 must be multiples of 8 (lda2, ldb2 >= K2), ldc of 4");
    MOLLY_CHECK(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0 && ((uintptr_t)C % 16) == 0 && ((uintptr_t)A2 % 16) == 0 &&
                ((uintptr_t)B2 % 16) == 0, "gemm_kx: operands must be 16-byte aligned");
    MOLLY_CHECK(!(flags & MOLLY_GEMM_BIAS) || bias, "gemm_kx: MOLLY_GEMM_BIAS without bias pointer");
    MOLLY_CHECK(!(flags & (MOLLY_GEMM_RESIDUAL | MOLLY_GEMM_SWIGLU)) || (res && ldres % 4 == 0), "gemm_kx: bad residual / activation output");
    GemmArgs p{};
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C;
    p.bias = (const bf16_t*)bias; p.res = (const bf16_t*)res;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.ldres = ldres; p.flags = flags;
    p.A2 = (const bf16_t*)A2; p.B2 = (const bf16_t*)B2; p.lda2 = lda2; p.ldb2 = ldb2; p.k2 = K2 / 64;
    p.splits = 1; p.group_m = c.group_m;
    p.tiles_m = cdiv(M, 256); p.tiles_n = cdiv(N, 256);
    if (int rc = resolve_zero_page(&p.zeros)) return rc;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm256_kernel<false, false, false, false, false, false, false, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        attr = true;
    }
    const int grid = grid256(c, p.tiles_m * p.tiles_n);
    c.last_cfg = 512 + 1000 + 7;                                   // (517: the K-extended launch)
    hipLaunchKernelGGL((gemm256_kernel<false, false, false, false, false, false, false, true>), dim3(grid), dim3(512), 163840, (hipStream_t)stream, p);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

int launch_grouped(void* ctx, void* stream, const molly_gemm_problem* problems, int count, int K, int flags) {
    MOLLY_ENTER();
    GemmCtx& c = ctx_of(ctx);
    MOLLY_CHECK(problems && count >= 1 && count <= 16, "gemm_grouped: 1..16 problems (count=%d)", count);
    MOLLY_CHECK(K > 0 && !(flags & ~(MOLLY_GEMM_ACCUMULATE | MOLLY_GEMM_OUT_F32)),
                "gemm_grouped: K=%d, flags 0x%x (accumulate / fp32 output only)", K, flags);
    GemmArgs p{};
    p.K = K; p.flags = flags; p.splits = 1; p.ws = nullptr; p.group_m = c.group_m; p.ngroup = count;
    int work = 0;
    for (int i = 0; i < count; ++i) {
        const molly_gemm_problem& q = problems[i];
        MOLLY_CHECK(q.A && q.B && q.C && q.M > 0 && q.N > 0, "gemm_grouped: problem %d is empty", i);
        MOLLY_CHECK(q.N % 8 == 0 && q.M % 4 == 0, "gemm_grouped: problem %d: N=%d must be a multiple of 8, M=%d of 4", i, q.N, q.M);
        MOLLY_CHECK(q.lda % 8 == 0 && q.ldb % 8 == 0 && q.ldc % 4 == 0, "gemm_grouped: problem %d: lda/ldb multiples of 8, ldc of 4", i);
        MOLLY_CHECK(((uintptr_t)q.A % 16) == 0 && ((uintptr_t)q.B % 16) == 0 && ((uintptr_t)q.C % 16) == 0,
                    "gemm_grouped: problem %d: operands must be 16-byte aligned", i);
        MOLLY_CHECK(K % 64 == 0, "gemm_grouped: K=%d must be a multiple of 64 (k-contiguous A)", K);
        GemmArgs::Group& G = p.grp[i];
        G.A = (const bf16_t*)q.A; G.B = (const bf16_t*)q.B; G.C = q.C;
        G.M = q.M; G.N = q.N; G.lda = q.lda; G.ldb = q.ldb; G.ldc = q.ldc;
        G.tiles_m = cdiv(q.M, 256); G.tiles_n = cdiv(q.N, 256); G.trans_out = q.trans_out; G.work0 = work;
        work += G.tiles_m * G.tiles_n;
    }
    if (int rc = resolve_zero_page(&p.zeros)) return rc;
    // the single-problem fields describe problem 0 (never read by the grouped kernel beyond its initial values)
    p.A = p.grp[0].A; p.B = p.grp[0].B; p.C = p.grp[0].C; p.M = p.grp[0].M; p.N = p.grp[0].N;
    p.lda = p.grp[0].lda; p.ldb = p.grp[0].ldb; p.ldc = p.grp[0].ldc; p.tiles_m = p.grp[0].tiles_m; p.tiles_n = p.grp[0].tiles_n;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)gemm256_kernel<false, true, false, true, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        (void)hipFuncSetAttribute((const void*)gemm256_kernel<false, true, false, false, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
        attr = true;
    }
    const int grid = grid256(c, work);
    c.last_cfg = 512 + 1000 + 100000 * count;
    if (two_phase(c))
        hipLaunchKernelGGL((gemm256_kernel<false, true, false, true, true>), dim3(grid), dim3(512), 163840, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((gemm256_kernel<false, true, false, false, true>), dim3(grid), dim3(512), 163840, (hipStream_t)stream, p);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

int ctx_set(GemmCtx& c, int key, long v) {
    switch (key) {
    case MOLLY_GEMM_KEY_PERSISTENT_BLOCKS:
        MOLLY_CHECK((v >= 0 && v % 8 == 0 && v <= SK_MAX_BLOCKS) || (v < 0 && v >= -64),
                    "gemm persistent_blocks: %ld must be a non-negative multiple of 8 (<= %d), or -t (t tiles per block, t <= 64)", v,
                    SK_MAX_BLOCKS);
        c.persist_blocks = (int)v;
        return 0;
    case MOLLY_GEMM_KEY_SCHEDULE:
        MOLLY_CHECK(v >= -1 && v <= 1, "gemm schedule: %ld not in {-1,0,1}", v);
        c.schedule = (int)v;
        return 0;
    case MOLLY_GEMM_KEY_FORCE_TILE:
        MOLLY_CHECK(v == 0 || v == 128 || v == 512, "gemm force_tile: %ld not in {0,128,512}", v);
        c.force_tile = (int)v;
        return 0;
    case MOLLY_GEMM_KEY_GROUP_M:
        MOLLY_CHECK(v >= 1 && v <= 64, "gemm group_m: %ld", v);
        c.group_m = (int)v;
        return 0;
    case MOLLY_GEMM_KEY_SMALL_GRID_TILE:
        MOLLY_CHECK(v == 128 || v == 512, "gemm small_grid_tile: 128 or 512 (got %ld)", v);
        c.small_grid_tile = (int)v;
        return 0;
    case MOLLY_GEMM_KEY_MIN_KTILES:
        MOLLY_CHECK(v >= 2 && v <= 64, "gemm min_ktiles: %ld not in 2..64", v);
        c.min_ktiles = (int)v;
        return 0;
    case MOLLY_GEMM_KEY_SMALL3:
        MOLLY_CHECK(v == 0 || v == 1, "gemm small3: %ld not in {0,1}", v);
        c.small3 = (int)v;
        return 0;
    case MOLLY_GEMM_KEY_DYNAMIC:
        MOLLY_CHECK(v == 0 || v == 1, "gemm dynamic: %ld not in {0,1}", v);
        c.dynamic = (int)v;
        return 0;
    case MOLLY_GEMM_KEY_STREAM_EPI:
        MOLLY_CHECK(v == 0 || v == 1, "gemm stream_epi: %ld not in {0,1}", v);
        c.stream_epi = (int)v;
        return 0;
    case MOLLY_GEMM_KEY_SMALL_SPLIT:
        MOLLY_CHECK(v == 0 || v == 1, "gemm small_split: %ld not in {0,1}", v);
        c.small_split = (int)v;
        return 0;
    case MOLLY_GEMM_KEY_ROWS_TILED:
        MOLLY_CHECK(v == 0 || v == 1, "gemm rows_tiled: %ld not in {0,1}", v);
        c.rows_tiled = (int)v;
        return 0;
    case MOLLY_GEMM_KEY_ROWS_MAX_M:
        MOLLY_CHECK(v >= 64 && v <= 8192, "gemm rows_max_m: %ld not in 64..8192", v);
        c.rows_max_m = (int)v;
        return 0;
    case MOLLY_GEMM_KEY_ROWS_BN:
        MOLLY_CHECK(v == 64 || v == 128, "gemm rows_bn: %ld not in {64, 128}", v);
        c.rows_bn = (int)v;
        return 0;
    case MOLLY_GEMM_KEY_ROWS_GU:
        MOLLY_CHECK(v == 0 || v == 1 || v == 32 || v == 64 || v == 128, "gemm rows_gu: %ld not in {0, 1, 32, 64, 128}", v);
        c.rows_gu = (int)v;
        return 0;
    case MOLLY_GEMM_KEY_DYNAMIC_MIN_WORK:
        MOLLY_CHECK(v >= 257 && v <= (1 << 30), "gemm dynamic_min_work: %ld < 257", v);
        c.dynamic_min_work = (int)v;
        return 0;
    case MOLLY_GEMM_KEY_SKINNY:
        MOLLY_CHECK(v == 0 || v == 1, "gemm skinny: %ld not in {0,1}", v);
        c.skinny = (int)v;
        return 0;
    case MOLLY_GEMM_KEY_STREAMK:
        MOLLY_CHECK(v >= 0 && v <= 2, "gemm streamk: %ld not in {0,1,2}", v);
        c.streamk = (int)v;
        return 0;
    default:
        molly_set_error("gemm_ctx_set: unknown key %d", key);
        return 1;
    }
}

int ctx_set_workspace(GemmCtx& c, void* ptr, long bytes) {
    MOLLY_CHECK(bytes >= 0 && ((uintptr_t)ptr % 256) == 0, "gemm_set_workspace: pointer must be 256-byte aligned");
    c.ws = (float*)ptr;
    c.ws_bytes = ptr ? (size_t)bytes : 0;
    if (ptr && (size_t)bytes > SK_HDR_BYTES) {
        // stream-K's flags must read zero before the first launch (afterwards every launch leaves them zero).  Not on the hot
        // path: a synchronous clear on the null stream, once per workspace.
        if (hipMemset(ptr, 0, SK_HDR_BYTES) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            (void)hipGetLastError();
            molly_set_error("gemm_set_workspace: cannot clear the stream-K header");
            return 2;
        }
    }
    return 0;
}

}  // namespace

extern "C" int molly_gemm_nt_bf16(void* stream, const void* A, const void* B, void* C, const void* bias,
                                  const void* res, int M, int N, int K, int lda, int ldb, int ldc, int ldres,
                                  int flags) {
    return launch_gemm(nullptr, stream, A, B, C, bias, res, M, N, K, lda, ldb, ldc, ldres, flags, false, false);
}

extern "C" int molly_gemm_bf16(void* stream, const void* A, const void* B, void* C, const void* bias, const void* res,
                               int M, int N, int K, int lda, int ldb, int ldc, int ldres, int flags, int a_kmajor,
                               int b_kmajor) {
    return launch_gemm(nullptr, stream, A, B, C, bias, res, M, N, K, lda, ldb, ldc, ldres, flags, a_kmajor != 0, b_kmajor != 0);
}

extern "C" int molly_gemm_bf16_ctx(void* ctx, void* stream, const void* A, const void* B, void* C, const void* bias,
                                   const void* res, int M, int N, int K, int lda, int ldb, int ldc, int ldres, int flags,
                                   int a_kmajor, int b_kmajor) {
    return launch_gemm(ctx, stream, A, B, C, bias, res, M, N, K, lda, ldb, ldc, ldres, flags, a_kmajor != 0, b_kmajor != 0);
}

// ---- decode-row GEMM + the kernel the decode step launches behind it (tail 1: RMSNorm of the output rows, the next block's input
// norm; tail 2: SwiGLU of a gate|up output), in the launch that combines the K slices
extern "C" int molly_gemm_kx_supported(void* ctx, int M, int N, int K, int K2, int flags) { return kx_supported(ctx_of(ctx), M, N, K, K2, flags) ? 1 : 0; }
extern "C" int molly_gemm_kx_bf16_ctx(void* ctx, void* stream, const void* A, const void* B, void* C, const void* bias, const void* res, int M, int N,
                                      int K, int lda, int ldb, int ldc, int ldres, int flags, const void* A2, const void* B2, int K2, int lda2,
                                      int ldb2) {
    return launch_gemm_kx(ctx, stream, A, B, C, bias, res, M, N, K, lda, ldb, ldc, ldres, flags, A2, B2, K2, lda2, ldb2);
}

extern "C" int molly_gemm_rows_tail_supported(void* ctx, int M, int N, int K, int tail) {
    const GemmCtx& c = ctx_of(ctx);
    if (!(tail >= 1 && tail <= 4) || !rows_applicable(c, M, N, K, 0) || !c.ws) return 0;
    if (tail == 1 && N > 8192) return 0;
    if (tail == 2 && N % 8 != 0) return 0;
    if (streaming_rows(c, M, N, K, 0)) return 0;          // a streaming-kernel shape stays there: its one launch is cheaper than slices + tail
    return (size_t)2 * M * N * sizeof(float) <= ws_slab_bytes(c) ? 1 : 0;
}
extern "C" int molly_gemm_rows_tail_bf16_ctx(void* ctx, void* stream, const void* A, const void* B, void* C, const void* bias,
                                             const void* res, int M, int N, int K, int lda, int ldb, int ldc, int ldres, int flags,
                                             int tail, const void* gain, float eps, void* tail_out, int ld_tail) {
    MOLLY_ENTER();
    GemmCtx& c = ctx_of(ctx);
    MOLLY_CHECK(molly_gemm_rows_tail_supported(ctx, M, N, K, tail), "gemm rows tail: M=%d N=%d K=%d tail=%d is not a shape of the tiled "
                "decode-row kernel (ask molly_gemm_rows_tail_supported first)", M, N, K, tail);
    MOLLY_CHECK(!(flags & ~(MOLLY_GEMM_BIAS | MOLLY_GEMM_RESIDUAL)) && (tail == 1 || !(flags & MOLLY_GEMM_RESIDUAL)),
                "gemm rows tail: flags %d (bias, and a residual in front of the norm, are what a tail takes)", flags);
    MOLLY_CHECK(lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && ld_tail % 4 == 0 && tail_out && (C || tail == 2) && (tail == 2 || gain),
                "gemm rows tail: strides / pointers");
    MOLLY_CHECK(!(flags & MOLLY_GEMM_BIAS) || bias, "gemm rows tail: MOLLY_GEMM_BIAS without bias pointer");
    MOLLY_CHECK(!(flags & MOLLY_GEMM_RESIDUAL) || (res && ldres % 4 == 0), "gemm rows tail: bad residual");
    const int rc = launch_rows(c, (hipStream_t)stream, A, B, C, bias, res, M, N, K, lda, ldb, ldc, ldres, flags, tail, gain, eps, tail_out, ld_tail);
    if (rc) return rc;
    MOLLY_LAUNCH_CHECK();
    return 0;
}

// decode rows, the K slices left as they are: A W^T as n fp32 slabs [n][M][N] in the context's scratch (n >= 2; their sum is the product), for a
// consumer that combines them itself (molly_attn_decode_qkv).  slabs_out[0] = device address, [1] = n; valid until this context's next launch.
extern "C" int molly_gemm_rows_slabs_bf16_ctx(void* ctx, void* stream, const void* A, const void* W, int M, int N, int K, int lda, int ldw,
                                              long* slabs_out) {
    MOLLY_ENTER();
    GemmCtx& c = ctx_of(ctx);
    MOLLY_CHECK(molly_gemm_rows_tail_supported(ctx, M, N, K, 4), "gemm rows slabs: M=%d N=%d K=%d is not a shape of the tiled decode-row "
                "kernel (ask molly_gemm_rows_tail_supported first)", M, N, K);
    MOLLY_CHECK(slabs_out && lda % 8 == 0 && ldw % 8 == 0, "gemm rows slabs: pointers / strides");
    const int rc = launch_rows(c, (hipStream_t)stream, A, W, nullptr, nullptr, nullptr, M, N, K, lda, ldw, 0, 0, 0, 4, nullptr, 0.f, nullptr, 0);
    if (rc) return rc;
    slabs_out[0] = (long)(uintptr_t)ws_slabs(c);
    slabs_out[1] = c.last_cfg / 1000;
    MOLLY_LAUNCH_CHECK();
    return 0;
}

// decode rows: the fused q | k | v projection with q/k-norm + rotary + the KV-cache append in the launch that combines the K slices
// (molly_gemm_rows_tail_supported(ctx, M, N, K, 3) says whether this context runs the shape that way)
extern "C" int molly_gemm_rows_qkv_bf16_ctx(void* ctx, void* stream, const void* A, const void* W, const void* bias, int M, int N, int K,
                                            int lda, int ldw, const void* q_norm_w, const void* k_norm_w, const float* cos, const float* sin,
                                            const int* positions, float eps, int n_q_heads, int n_k_heads, int head_dim, void* dst, int ld_dst,
                                            void* kcache, void* vcache, const int* slot, int ld_cache) {
    MOLLY_ENTER();
    GemmCtx& c = ctx_of(ctx);
    MOLLY_CHECK(molly_gemm_rows_tail_supported(ctx, M, N, K, 3), "gemm rows qkv: M=%d N=%d K=%d is not a shape of the tiled decode-row "
                "kernel (ask molly_gemm_rows_tail_supported first)", M, N, K);
    MOLLY_CHECK(head_dim >= 16 && head_dim <= 512 && (head_dim & (head_dim - 1)) == 0 && N == (n_q_heads + 2 * n_k_heads) * head_dim,
                "gemm rows qkv: N=%d is not (%d + 2 x %d) heads of %d", N, n_q_heads, n_k_heads, head_dim);
    MOLLY_CHECK((q_norm_w == nullptr) == (k_norm_w == nullptr) && (cos == nullptr) == (sin == nullptr), "gemm rows qkv: norm gains / cos, sin in pairs");
    MOLLY_CHECK(dst && kcache && vcache && slot && lda % 8 == 0 && ldw % 8 == 0 && ld_dst % 4 == 0 && ld_cache % 4 == 0, "gemm rows qkv: pointers / strides");
    QkvTail t{(const bf16_t*)q_norm_w, (const bf16_t*)k_norm_w, cos, sin, positions, (bf16_t*)dst, (bf16_t*)kcache, (bf16_t*)vcache, slot,
              n_q_heads, n_k_heads, head_dim, ld_dst, ld_cache, eps};
    const int rc = launch_rows(c, (hipStream_t)stream, A, W, nullptr, bias, nullptr, M, N, K, lda, ldw, 0, 0, bias ? MOLLY_GEMM_BIAS : 0, 3,
                               nullptr, 0.f, nullptr, 0, &t);
    if (rc) return rc;
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_gemm_grouped_bf16(void* stream, const molly_gemm_problem* problems, int count, int K, int flags) {
    return launch_grouped(nullptr, stream, problems, count, K, flags);
}

extern "C" int molly_gemm_grouped_bf16_ctx(void* ctx, void* stream, const molly_gemm_problem* problems, int count, int K,
                                           int flags) {
    return launch_grouped(ctx, stream, problems, count, K, flags);
}

// ---- contexts
extern "C" int molly_gemm_ctx_create(void** out) {
    MOLLY_CHECK(out, "gemm_ctx_create: NULL out pointer");
    *out = new GemmCtx();
    return 0;
}
extern "C" int molly_gemm_ctx_destroy(void* ctx) {
    delete static_cast<GemmCtx*>(ctx);
    return 0;
}
extern "C" int molly_gemm_ctx_set(void* ctx, int key, long value) { return ctx_set(ctx_of(ctx), key, value); }
extern "C" int molly_gemm_ctx_get(void* ctx, int key) {
    const GemmCtx& c = ctx_of(ctx);
    switch (key) {
    case MOLLY_GEMM_KEY_PERSISTENT_BLOCKS: return c.persist_blocks;
    case MOLLY_GEMM_KEY_SCHEDULE: return c.schedule;
    case MOLLY_GEMM_KEY_FORCE_TILE: return c.force_tile;
    case MOLLY_GEMM_KEY_GROUP_M: return c.group_m;
    case MOLLY_GEMM_KEY_SMALL_GRID_TILE: return c.small_grid_tile;
    case MOLLY_GEMM_KEY_MIN_KTILES: return c.min_ktiles;
    case MOLLY_GEMM_KEY_STREAMK: return c.streamk;
    case MOLLY_GEMM_KEY_SKINNY: return c.skinny;
    case MOLLY_GEMM_KEY_SMALL3: return c.small3;
    case MOLLY_GEMM_KEY_DYNAMIC: return c.dynamic;
    case MOLLY_GEMM_KEY_SMALL_SPLIT: return c.small_split;
    case MOLLY_GEMM_KEY_STREAM_EPI: return c.stream_epi;
    case MOLLY_GEMM_KEY_ROWS_TILED: return c.rows_tiled;
    case MOLLY_GEMM_KEY_DYNAMIC_MIN_WORK: return c.dynamic_min_work;
    case MOLLY_GEMM_KEY_ROWS_MAX_M: return c.rows_max_m;
    case MOLLY_GEMM_KEY_ROWS_GU: return c.rows_gu;
    case MOLLY_GEMM_KEY_ROWS_BN: return c.rows_bn;
    case MOLLY_GEMM_KEY_LAST_CONFIG: return c.last_cfg;
    default: return -1;
    }
}
extern "C" int molly_gemm_ctx_set_workspace(void* ctx, void* ptr, long bytes) { return ctx_set_workspace(ctx_of(ctx), ptr, bytes); }
// the error word of the stream-K hand-off (bit 0: a tile was cut into more pieces than the reducer lists — its sum would be short): a
// device read, for tests and diagnostics only
extern "C" int molly_gemm_ctx_streamk_timeouts(void* ctx) {
    const GemmCtx& c = ctx_of(ctx);
    if (!c.ws || c.ws_bytes <= SK_HDR_BYTES) return 0;
    unsigned v = 0;
    if (hipMemcpy(&v, c.ws, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return (int)v;
}

// ---- the calling thread's default context (what the entry points without a context launch through)
// fp32 scratch for split-K partial slabs and the stream-K slabs + flags (caller-owned device memory; NULL/0 disables both)
extern "C" int molly_gemm_set_workspace(void* ptr, long bytes) { return ctx_set_workspace(t_ctx, ptr, bytes); }
// which kernel configuration the most recent molly_gemm_* call of this thread's default context used
extern "C" int molly_gemm_last_config(void) { return t_ctx.last_cfg; }
extern "C" int molly_gemm_set_small_grid_tile(int tile) { return ctx_set(t_ctx, MOLLY_GEMM_KEY_SMALL_GRID_TILE, tile); }
extern "C" int molly_gemm_set_group_m(int g) { return ctx_set(t_ctx, MOLLY_GEMM_KEY_GROUP_M, g); }
extern "C" int molly_gemm_set_schedule(int mode) { return ctx_set(t_ctx, MOLLY_GEMM_KEY_SCHEDULE, mode); }
extern "C" int molly_gemm_set_min_ktiles(int n) { return ctx_set(t_ctx, MOLLY_GEMM_KEY_MIN_KTILES, n); }
extern "C" int molly_gemm_set_persistent_blocks(int n) { return ctx_set(t_ctx, MOLLY_GEMM_KEY_PERSISTENT_BLOCKS, n); }
extern "C" int molly_gemm_force_tile(int bm) { return ctx_set(t_ctx, MOLLY_GEMM_KEY_FORCE_TILE, bm); }
extern "C" int molly_gemm_set_streamk(int on) { return ctx_set(t_ctx, MOLLY_GEMM_KEY_STREAMK, on); }
