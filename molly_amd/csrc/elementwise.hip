// HBM-bound kernels of Molly's hot path (gfx950): norms, rotary, SwiGLU, embedding gather/scatter,
// cross-entropy, transposes, AdamW.  All of them stream bf16 as 16-byte vectors (guide G13), keep the
// math in fp32 and touch every byte once (or state why not).  Roofline for each: HBM (~6.3 TB/s achievable).
#include "common.h"
#include "molly_hip.h"

namespace {

// ------------------------------------------------------------------------------------------------
// transpose: out[C,R] = in[R,C]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int R,
                                                        int C, int ld_in, int ld_out) {
    __shared__ bf16_t tile[64][66];   // +2 pad: conflict-light column reads
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < R && c < C) ? in[(size_t)r * ld_in + c] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < C && r < R) out[(size_t)c * ld_out + r] = tile[tx][i];
    }
}

// fast path (R, C multiples of 64, 16-byte aligned rows): 16-byte coalesced loads, transposition by
// ds_read_b64_tr_b16 (guide T10), 16-byte stores that fill whole 64-byte segments of the output rows.  HBM-bound.
__global__ __launch_bounds__(256) void transpose64_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int ld_in,
                                                          int ld_out) {
    __shared__ __attribute__((aligned(16))) bf16_t tile[64 * 72];      // row pitch 144 B
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int t = threadIdx.x;
    {
        const int row = t >> 2, col = (t & 3) * 16;
        const u32x4* src = reinterpret_cast<const u32x4*>(in + (size_t)(r0 + row) * ld_in + c0 + col);
        const u32x4 a = src[0], b = src[1];      // (plain: with the nontemporal hint this kernel measured 41 us against 27)
        // pitch 144 B keeps 16-byte alignment only for even rows; write as 8-byte pieces
        u32x2* dst = reinterpret_cast<u32x2*>(tile + row * 72 + col);
        dst[0] = u32x2{a[0], a[1]}; dst[1] = u32x2{a[2], a[3]}; dst[2] = u32x2{b[0], b[1]}; dst[3] = u32x2{b[2], b[3]};
    }
    __syncthreads();
    const int lane = t & 63, w = t >> 6;
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
    const int oc = w * 16;                         // this wave's 16 input columns = 16 output rows
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int rr = pass * 32 + g * 8;          // 8 input rows -> 8 consecutive output columns
        const bf16_t* pa = tile + (rr + q) * 72 + oc + 4 * pp;
        const bf16_t* pb = tile + (rr + 4 + q) * 72 + oc + 4 * pp;
        const bf16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)pa);
        const bf16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)pb);
        const bf16x8 v = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
        *reinterpret_cast<bf16x8*>(out + (size_t)(c0 + oc + i) * ld_out + r0 + rr) = v;
    }
}

// ------------------------------------------------------------------------------------------------
// RMSNorm (HF:models/qwen3/modeling_qwen3.py:59-64): y = w * bf16(x * rsqrt(mean(x^2)+eps))
// one wave per row, 4 rows per block; row kept in registers (H <= 8192)
// ------------------------------------------------------------------------------------------------
constexpr int RN_MAXC = 8;   // 8 chunks * 64 lanes * 8 elems = 4096 (Qwen3-8B hidden size)

template <int NC>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          bf16_t* __restrict__ y, float* __restrict__ rstd_out, int rows,
                                                          int H, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = H >> 3;
    const u32x4* xr = reinterpret_cast<const u32x4*>(x + (size_t)row * H);
    u32x4 v[NC];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            v[i] = ld_stream<u32x4>(xr + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = bflo(v[i][e]), b = bfhi(v[i][e]);
                ss += a * a + b * b;
            }
        }
    }
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / (float)H + eps);
    if (rstd_out && lane == 0) rstd_out[row] = rstd;
    const u32x4* wr = reinterpret_cast<const u32x4*>(w);
    u32x4* yr = reinterpret_cast<u32x4*>(y + (size_t)row * H);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            const u32x4 wv = wr[c];
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // HF rounds the normalised value to bf16 BEFORE the gain multiply
                const uint32_t t = pack_bf2(bflo(v[i][e]) * rstd, bfhi(v[i][e]) * rstd);
                o[e] = pack_bf2(bflo(t) * bflo(wv[e]), bfhi(t) * bfhi(wv[e]));
            }
            st_stream<u32x4>(yr + c, o);
        }
    }
}

// ---- a block's 64 rows, written a second time TRANSPOSED (round 5).  The weight-gradient GEMMs want their narrow operand with the
// token index contiguous (dW^T = x^T dy with x^T k-contiguous), which cost one 5 TB/s transpose launch per operand and
// layer — 113 per step, 5.6 ms at 16 x 2,048 tokens, each re-reading a tensor its producer had just had in registers.  The producer
// now stores it both ways: the block keeps its 64 rows' results in registers (8 waves x 8 rows x NC 16-byte chunks), and per panel of
// 512 columns every wave puts its rows into an LDS tile [64][512 + pad], the tile is read back through ds_read_b64_tr_b16 exactly as
// transpose64_kernel reads its own, and leaves as whole 128-byte lines of the [H][rows] image.
constexpr int TP_COLS = 512, TP_PITCH = TP_COLS + 72;       // elements; 1,168-byte rows: 16-byte aligned, the transposed reads of transpose64_kernel's pitch class
// tile: [64][TP_PITCH]; the calling wave's rows 8 w .. 8 w + 7 hold o[j] = columns 8 lane .. 8 lane + 7 of the panel.  out_t = &yT[panel col 0][row0].
__device__ __forceinline__ void panel_transpose_store(bf16_t* tile, const u32x4 (&o)[8], bf16_t* __restrict__ out_t, size_t ld_t, int wave, int lane) {
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<u32x4*>(tile + (8 * wave + j) * TP_PITCH + 8 * lane) = o[j];
    __syncthreads();
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {                       // this wave's 64 panel columns, 16 at a time
        const int oc = 64 * wave + 16 * cb;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int rr = pass * 32 + g * 8;              // 8 tile rows -> 8 consecutive output columns
            const bf16_t* pa = tile + (rr + q) * TP_PITCH + oc + 4 * pp;
            const bf16_t* pb = tile + (rr + 4 + q) * TP_PITCH + oc + 4 * pp;
            const bf16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)pa);
            const bf16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)pb);
            const bf16x8 v = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
            *reinterpret_cast<bf16x8*>(out_t + (size_t)(oc + i) * ld_t + rr) = v;
        }
    }
    __syncthreads();                                       // the tile is free for the next panel
}

// RMSNorm forward with the transposed second store: y [rows][H] and yT [H][ld_t] (ld_t >= rows).  rows % 64 == 0, H % 512 == 0.
template <int NC>
__global__ __launch_bounds__(512) void rmsnorm_fwd_t_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, bf16_t* __restrict__ y,
                                                            bf16_t* __restrict__ yT, int ld_t, int H, float eps) {
    extern __shared__ __attribute__((aligned(16))) char tp_smem[];
    bf16_t* tile = reinterpret_cast<bf16_t*>(tp_smem);
    // (wave index made provably uniform: the row pointers are then scalar + one 32-bit lane offset; as a vector value hipcc kept a
    // 64-bit address pair per row, chunk and tensor — 192 registers in the backward form — and spilled)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row0 = blockIdx.x * 64;
    const u32x4* wr = reinterpret_cast<const u32x4*>(w);
    u32x4 o[NC][8];                                            // [chunk = panel][row of the wave]
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = row0 + 8 * wave + j;
        const u32x4* xr = reinterpret_cast<const u32x4*>(x + (size_t)row * H);
        u32x4 v[NC];
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            v[i] = ld_stream<u32x4>(xr + lane + i * 64);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = bflo(v[i][e]), b = bfhi(v[i][e]);
                ss += a * a + b * b;
            }
        }
        ss = wave_sum(ss);
        const float rstd = rsqrtf(ss / (float)H + eps);
        u32x4* yr = reinterpret_cast<u32x4*>(y + (size_t)row * H);
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const u32x4 wv = wr[lane + i * 64];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const uint32_t t = pack_bf2(bflo(v[i][e]) * rstd, bfhi(v[i][e]) * rstd);     // HF: round, then the gain
                o[i][j][e] = pack_bf2(bflo(t) * bflo(wv[e]), bfhi(t) * bfhi(wv[e]));
            }
            st_stream<u32x4>(yr + lane + i * 64, o[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) panel_transpose_store(tile, o[i], yT + (size_t)(i * TP_COLS) * ld_t + row0, (size_t)ld_t, wave, lane);
}

// backward: dx = rstd*(g*w - xhat*mean(g*w*xhat)) (+dres) ; dw partial per block (fp32) -> workspace[nblk][H]
// persistent over rows: block b handles rows b*4+wave, stride gridDim*4
template <int NC>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                          const bf16_t* __restrict__ g, const bf16_t* __restrict__ dres,
                                                          bf16_t* __restrict__ dx, float* __restrict__ dw_part, int rows,
                                                          int H, float eps) {
    extern __shared__ __attribute__((aligned(16))) char sm_raw[];
    float* sdw = reinterpret_cast<float*>(sm_raw);   // [4][H] would be big; instead waves accumulate in regs
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = H >> 3;
    float dwacc[NC][8];
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) dwacc[i][e] = 0.f;
    const u32x4* wr = reinterpret_cast<const u32x4*>(w);
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const u32x4* xr = reinterpret_cast<const u32x4*>(x + (size_t)row * H);
        const u32x4* gr = reinterpret_cast<const u32x4*>(g + (size_t)row * H);
        u32x4 xv[NC], gv[NC];
        float ss = 0.f, dot = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + i * 64;
            if (c < nch) {
                xv[i] = ld_stream<u32x4>(xr + c);
                gv[i] = ld_stream<u32x4>(gr + c);
                const u32x4 wv = wr[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = bflo(xv[i][e]), b = bfhi(xv[i][e]);
                    ss += a * a + b * b;
                    dot += bflo(gv[i][e]) * bflo(wv[e]) * a + bfhi(gv[i][e]) * bfhi(wv[e]) * b;
                }
            }
        }
        ss = wave_sum(ss);
        dot = wave_sum(dot);
        const float rstd = rsqrtf(ss / (float)H + eps);
        const float coef = dot * rstd * rstd * rstd / (float)H;   // mean(g*w*xhat)*rstd, xhat = x*rstd
        u32x4* dxr = reinterpret_cast<u32x4*>(dx + (size_t)row * H);
        const u32x4* rr = dres ? reinterpret_cast<const u32x4*>(dres + (size_t)row * H) : nullptr;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + i * 64;
            if (c < nch) {
                const u32x4 wv = wr[c];
                u32x4 rv = u32x4{0, 0, 0, 0};
                if (rr) rv = ld_stream<u32x4>(rr + c);
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xa = bflo(xv[i][e]), xb = bfhi(xv[i][e]);
                    const float ga = bflo(gv[i][e]), gb = bfhi(gv[i][e]);
                    float da = ga * bflo(wv[e]) * rstd - xa * coef;
                    float db = gb * bfhi(wv[e]) * rstd - xb * coef;
                    if (rr) { da += bflo(rv[e]); db += bfhi(rv[e]); }
                    o[e] = pack_bf2(da, db);
                    dwacc[i][2 * e] += ga * xa * rstd;
                    dwacc[i][2 * e + 1] += gb * xb * rstd;
                }
                st_stream<u32x4>(dxr + c, o);
            }
        }
    }
    // combine the 4 waves' partials through LDS (one chunk column at a time), then write the block partial
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + i * 64;
        __syncthreads();
        if (c < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sdw[(wave * 64 + lane) * 8 + e] = dwacc[i][e];
        }
        __syncthreads();
        if (wave == 0 && c < nch) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = sdw[(0 * 64 + lane) * 8 + e] + sdw[(1 * 64 + lane) * 8 + e] +
                                sdw[(2 * 64 + lane) * 8 + e] + sdw[(3 * 64 + lane) * 8 + e];
                dw_part[(size_t)blockIdx.x * H + c * 8 + e] = t;
            }
        }
    }
}

// (The BACKWARD with a transposed second store of dx — the down-projection's weight-gradient operand — was built the same way and
// removed in round 5: eight rows of dx held in registers beside the row's x, g, residual and the 32 gain-gradient accumulators is
// 228 live registers by count and 514 spilled ones as hipcc compiles it, 8 ms SLOWER per step than norm + transpose launch; the
// two-phase form that fits re-reads x and g and would have saved 0.6 ms.)
// out[j] (+)= sum_b part[b][j]  — deterministic column reduce of block partials.
// block = 32 columns x 8 row groups (fixed summation tree), grid = H/32 blocks
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ part, int nb, int H, int row_stride,
                                                     void* out, int out_f32, int accumulate) {
    // block = 32 columns as 8 float4 quads x 32 row groups (16-B loads, 128 B contiguous per row); H % 4 == 0
    __shared__ float red[32][33];
    const int cq = threadIdx.x & 7, rg = threadIdx.x >> 3;
    const int j = blockIdx.x * 32 + cq * 4;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    if (j < H)
        for (int b = rg; b < nb; b += 32) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(part + (size_t)b * row_stride + j);
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rg][cq * 4 + e] = s[e];
    __syncthreads();
    const int c = threadIdx.x, jj = blockIdx.x * 32 + c;
    if (c < 32 && jj < H) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 32; ++g) t += red[g][c];
        if (out_f32) {
            float* o = reinterpret_cast<float*>(out);
            o[jj] = accumulate ? o[jj] + t : t;
        } else {
            bf16_t* o = reinterpret_cast<bf16_t*>(out);
            o[jj] = f2bf(accumulate ? bf2f(o[jj]) + t : t);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// per-head RMSNorm (optional) + scale (optional) + rotary (optional) on the q/k heads of a fused
// projection buffer.  Qwen3: q_norm/k_norm then RoPE (HF:qwen3:252-257,148-170).  ESM-2: q*hd^-0.5 then
// fp32 rotary, no norm (HF:esm:374-378,56-79).  One thread owns the pair (i, i+hd/2) of one head.
// src [M, ld_src] heads at column head*hd ; dst [M, ld_dst] ; cos/sin fp32 [n_pos, hd/2]
// ------------------------------------------------------------------------------------------------
struct RopeArgs {
    const bf16_t* src; bf16_t* dst;
    const bf16_t* qw; const bf16_t* kw;      // per-head norm gains (nullable = no norm)
    const float* cos; const float* sin;      // nullable = no rotary
    const int* pos;                          // nullable: position = m % T
    int M, T, nq, nk, hd, ld_src, ld_dst;
    float eps, q_scale;
    // decode step: the k heads (after norm + rotary) and the v heads (as they are) of row m also go to row slot[m] of the KV caches
    bf16_t* kc; bf16_t* vc; const int* slot; int ld_cache, nv;
};

// each thread owns 4 consecutive (i, i+hd/2) pairs -> 8-byte loads/stores; hd/8 threads per head
__global__ __launch_bounds__(256) void norm_rope_fwd_kernel(RopeArgs p) {
    const int half = p.hd >> 1;
    const int tph = half >> 2;                          // threads per head
    const int heads_per_blk = 256 / tph;
    const int i = (threadIdx.x % tph) * 4;
    const long item = (long)blockIdx.x * heads_per_blk + threadIdx.x / tph;
    const int nh = p.nq + p.nk + p.nv;                  // (nv > 0: the decode step's cache append; v heads follow the k heads in src)
    const long total = (long)p.M * nh;
    const bool live = item < total;
    const int m = live ? (int)(item / nh) : 0, head = live ? (int)(item % nh) : 0;
    const bf16_t* s = p.src + (size_t)m * p.ld_src + head * p.hd;
    if (head >= p.nq + p.nk) {                          // a v head: copied as it is (whole groups of tph lanes take this branch)
        if (live) {
            bf16_t* d = p.vc + (size_t)p.slot[m] * p.ld_cache + (head - p.nq - p.nk) * p.hd;
            *reinterpret_cast<u32x2*>(d + i) = *reinterpret_cast<const u32x2*>(s + i);
            *reinterpret_cast<u32x2*>(d + i + half) = *reinterpret_cast<const u32x2*>(s + i + half);
        }
        return;
    }
    float x1[4] = {0, 0, 0, 0}, x2[4] = {0, 0, 0, 0};
    if (live) {
        const u32x2 a = ld_stream<u32x2>(s + i), b = ld_stream<u32x2>(s + i + half);
        x1[0] = bflo(a[0]); x1[1] = bfhi(a[0]); x1[2] = bflo(a[1]); x1[3] = bfhi(a[1]);
        x2[0] = bflo(b[0]); x2[1] = bfhi(b[0]); x2[2] = bflo(b[1]); x2[3] = bfhi(b[1]);
    }
    const bool isq = head < p.nq;
    const bf16_t* w = isq ? p.qw : p.kw;
    if (w) {
        const float ss = head_lanes_sum(head_sumsq8(x1, x2), tph);        // (common.h: shared with the decode step's kernels, contraction pinned)
        head_norm8(x1, x2, rsqrtf(ss / (float)p.hd + p.eps), w, i, half);
    }
    if (isq && p.q_scale != 1.0f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            x1[e] = bf2f(f2bf(x1[e] * p.q_scale));
            x2[e] = bf2f(f2bf(x2[e] * p.q_scale));
        }
    }
    float y1[4], y2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { y1[e] = x1[e]; y2[e] = x2[e]; }
    if (p.cos) {
        const int pos = p.pos ? p.pos[m] : (m % p.T);
        const f32x4 c = *reinterpret_cast<const f32x4*>(p.cos + (size_t)pos * half + i);
        const f32x4 sn = *reinterpret_cast<const f32x4*>(p.sin + (size_t)pos * half + i);
        head_rope8(x1, x2, c, sn, y1, y2);
    }
    if (live) {
        bf16_t* d = p.dst + (size_t)m * p.ld_dst + head * p.hd;
        const u32x2 o1 = u32x2{pack_bf2(y1[0], y1[1]), pack_bf2(y1[2], y1[3])}, o2 = u32x2{pack_bf2(y2[0], y2[1]), pack_bf2(y2[2], y2[3])};
        st_stream<u32x2>(d + i, o1);
        st_stream<u32x2>(d + i + half, o2);
        if (p.kc && !isq) {
            bf16_t* c = p.kc + (size_t)p.slot[m] * p.ld_cache + (head - p.nq) * p.hd;
            *reinterpret_cast<u32x2*>(c + i) = o1;
            *reinterpret_cast<u32x2*>(c + i + half) = o2;
        }
    }
}

// The training / prefill form of the above (no cache append) for a head_dim known at compile time and M * heads < 2^31.  The same arithmetic, bit
// for bit (the shared helpers of common.h, the same lanes per head), with a fraction of the index instructions: norm_rope_fwd_kernel runs ~300
// vector instructions per thread for 16 bytes in and 16 out — a 64-bit division and modulo by the head count, three more divisions by run-time
// powers of two — and is bound by them (97 us per Qwen3-1.7B layer at 32 k tokens = 4.1 TB/s).  Here: one 32-bit division per item, shifts for the
// rest, no branches in front of the loads (a dead item's address is clamped, only its store is masked), and TWO heads per thread whose loads are
// all issued ahead of the arithmetic (32 bytes per thread in flight instead of 16).
template <int HD>
__global__ __launch_bounds__(256) void norm_rope_fwd_fast_kernel(RopeArgs p) {
    constexpr int half = HD / 2, tph = half / 4, HPB = 256 / tph;
    const int i = (threadIdx.x % tph) * 4, sub = threadIdx.x / tph;
    const unsigned nh = (unsigned)(p.nq + p.nk), total = (unsigned)p.M * nh;
    const bool t_pow2 = (p.T & (p.T - 1)) == 0;
    unsigned m[2], head[2], pos[2];
    bool live[2];
    u32x2 a[2], b[2];
    f32x4 c[2], sn[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const unsigned it = blockIdx.x * (2 * HPB) + k * HPB + sub;
        live[k] = it < total;
        const unsigned itc = live[k] ? it : total - 1;
        m[k] = itc / nh;
        head[k] = itc - m[k] * nh;
        pos[k] = p.pos ? (unsigned)p.pos[m[k]] : t_pow2 ? (m[k] & (unsigned)(p.T - 1)) : m[k] % (unsigned)p.T;
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const bf16_t* s = p.src + (size_t)m[k] * p.ld_src + head[k] * HD;
        a[k] = ld_stream<u32x2>(s + i);
        b[k] = ld_stream<u32x2>(s + i + half);
    }
    if (p.cos) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            c[k] = *reinterpret_cast<const f32x4*>(p.cos + (size_t)pos[k] * half + i);
            sn[k] = *reinterpret_cast<const f32x4*>(p.sin + (size_t)pos[k] * half + i);
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float x1[4] = {bflo(a[k][0]), bfhi(a[k][0]), bflo(a[k][1]), bfhi(a[k][1])};
        float x2[4] = {bflo(b[k][0]), bfhi(b[k][0]), bflo(b[k][1]), bfhi(b[k][1])};
        const bool isq = head[k] < (unsigned)p.nq;
        const bf16_t* w = isq ? p.qw : p.kw;
        if (w) {
            const float ss = head_lanes_sum(head_sumsq8(x1, x2), tph);
            head_norm8(x1, x2, rsqrtf(ss / (float)HD + p.eps), w, i, half);
        }
        if (isq && p.q_scale != 1.0f) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x1[e] = bf2f(f2bf(x1[e] * p.q_scale));
                x2[e] = bf2f(f2bf(x2[e] * p.q_scale));
            }
        }
        float y1[4], y2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { y1[e] = x1[e]; y2[e] = x2[e]; }
        if (p.cos) head_rope8(x1, x2, c[k], sn[k], y1, y2);
        if (live[k]) {
            bf16_t* d = p.dst + (size_t)m[k] * p.ld_dst + head[k] * HD;
            st_stream<u32x2>(d + i, u32x2{pack_bf2(y1[0], y1[1]), pack_bf2(y1[2], y1[3])});
            st_stream<u32x2>(d + i + half, u32x2{pack_bf2(y2[0], y2[1]), pack_bf2(y2[2], y2[3])});
        }
    }
}

// Many rows (the decoder layer's launch at 16-32 k tokens): one group of hd/8 lanes walks ALL the heads of one token, so the token's cos / sin
// values are loaded once instead of once per head (per 256 bytes of a head the kernels above read 512 bytes of tables through the same L1 path) and
// nothing is divided at all; four heads' loads are in flight per thread.  Same arithmetic, bit for bit.  Used when the rows fill the chip.
template <int HD>
__global__ __launch_bounds__(256) void norm_rope_fwd_rows_kernel(RopeArgs p) {
    constexpr int half = HD / 2, tph = half / 4, RPB = 256 / tph, U = 4;
    const int i = (threadIdx.x % tph) * 4, sub = threadIdx.x / tph;
    const int nh = p.nq + p.nk;
    const int m_raw = blockIdx.x * RPB + sub;
    const bool row_live = m_raw < p.M;
    const int m = row_live ? m_raw : p.M - 1;
    f32x4 c = {1.f, 1.f, 1.f, 1.f}, sn = {0.f, 0.f, 0.f, 0.f};
    if (p.cos) {
        const unsigned pos = p.pos ? (unsigned)p.pos[m] : (unsigned)m % (unsigned)p.T;
        c = *reinterpret_cast<const f32x4*>(p.cos + (size_t)pos * half + i);
        sn = *reinterpret_cast<const f32x4*>(p.sin + (size_t)pos * half + i);
    }
    const bf16_t* s = p.src + (size_t)m * p.ld_src;
    bf16_t* d = p.dst + (size_t)m * p.ld_dst;
    for (int h0 = 0; h0 < nh; h0 += U) {
        u32x2 a[U], b[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int head = h0 + k < nh ? h0 + k : nh - 1;
            a[k] = ld_stream<u32x2>(s + head * HD + i);
            b[k] = ld_stream<u32x2>(s + head * HD + i + half);
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int head = h0 + k < nh ? h0 + k : nh - 1;
            float x1[4] = {bflo(a[k][0]), bfhi(a[k][0]), bflo(a[k][1]), bfhi(a[k][1])};
            float x2[4] = {bflo(b[k][0]), bfhi(b[k][0]), bflo(b[k][1]), bfhi(b[k][1])};
            const bool isq = head < p.nq;
            const bf16_t* w = isq ? p.qw : p.kw;
            if (w) {
                const float ss = head_lanes_sum(head_sumsq8(x1, x2), tph);
                head_norm8(x1, x2, rsqrtf(ss / (float)HD + p.eps), w, i, half);
            }
            if (isq && p.q_scale != 1.0f) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x1[e] = bf2f(f2bf(x1[e] * p.q_scale));
                    x2[e] = bf2f(f2bf(x2[e] * p.q_scale));
                }
            }
            float y1[4], y2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { y1[e] = x1[e]; y2[e] = x2[e]; }
            if (p.cos) head_rope8(x1, x2, c, sn, y1, y2);
            if (row_live && h0 + k < nh) {
                st_stream<u32x2>(d + head * HD + i, u32x2{pack_bf2(y1[0], y1[1]), pack_bf2(y1[2], y1[3])});
                st_stream<u32x2>(d + head * HD + i + half, u32x2{pack_bf2(y2[0], y2[1]), pack_bf2(y2[2], y2[3])});
            }
        }
    }
}

// backward of the above (Qwen3 form: norm + rotary; q_scale==1).  g = d(dst) [M, ld_dst];
// writes d(src) for the q/k heads [M, ld_out] and per-block fp32 partials of d(q_norm.w), d(k_norm.w):
// dw_part[blk][2][hd]
struct RopeBwdArgs {
    const bf16_t* src; const bf16_t* g; bf16_t* dsrc;
    const bf16_t* qw; const bf16_t* kw;
    const float* cos; const float* sin; const int* pos;
    float* dw_part;
    int M, T, nq, nk, hd, ld_src, ld_g, ld_out;
    float eps;
    long items_per_blk;
    float q_scale;                           // forward multiplied the q heads by this (ESM: hd^-0.5 before rotary)
};

__global__ __launch_bounds__(256) void norm_rope_bwd_kernel(RopeBwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) char sm_raw[];
    float* sdw = reinterpret_cast<float*>(sm_raw);          // [256 / (hd/8) sub-groups][2 * hd] = 4096 floats
    const int half = p.hd >> 1;
    const int tph = half >> 2;
    const int heads_per_it = 256 / tph;
    const int i = (threadIdx.x % tph) * 4, sub = threadIdx.x / tph;
    const int nh = p.nq + p.nk;
    const long total = (long)p.M * nh;
    float dwq1[4] = {0, 0, 0, 0}, dwq2[4] = {0, 0, 0, 0}, dwk1[4] = {0, 0, 0, 0}, dwk2[4] = {0, 0, 0, 0};
    const long begin = (long)blockIdx.x * p.items_per_blk;
    const long end = min(begin + p.items_per_blk, total);
    // (token, head) of this thread's item walk by counters: the 64-bit division and the modulo per iteration were most of this
    // kernel's instructions — it is bound by them, not by HBM (82.7 us at 16 k tokens whatever the access width or the grid:
    // profiles/r04_logs/rope_bwd_kernel.log, rope_bwd_blocks.log)
    int m_run = (int)((begin + sub) / nh), head_run = (int)((begin + sub) % nh);
    int pos_run = p.T > 0 ? m_run % p.T : 0;
    for (long base = begin; base < end; base += heads_per_it) {
        const long item = base + sub;
        const bool live = item < end;
        const int m = live ? m_run : 0, head = live ? head_run : 0;
        const int pos_m = live ? pos_run : 0;
        head_run += heads_per_it;
        while (head_run >= nh) {
            head_run -= nh;
            ++m_run;
            if (++pos_run == p.T) pos_run = 0;
        }
        const bool isq = head < p.nq;
        const bf16_t* s = p.src + (size_t)m * p.ld_src + head * p.hd;
        const bf16_t* gg = p.g + (size_t)m * p.ld_g + head * p.hd;
        float x1[4] = {0, 0, 0, 0}, x2[4] = {0, 0, 0, 0}, g1[4] = {0, 0, 0, 0}, g2[4] = {0, 0, 0, 0};
        if (live) {
            const u32x2 a = ld_stream<u32x2>(s + i), b = ld_stream<u32x2>(s + i + half);
            const u32x2 c = ld_stream<u32x2>(gg + i), d = ld_stream<u32x2>(gg + i + half);
            x1[0] = bflo(a[0]); x1[1] = bfhi(a[0]); x1[2] = bflo(a[1]); x1[3] = bfhi(a[1]);
            x2[0] = bflo(b[0]); x2[1] = bfhi(b[0]); x2[2] = bflo(b[1]); x2[3] = bfhi(b[1]);
            g1[0] = bflo(c[0]); g1[1] = bfhi(c[0]); g1[2] = bflo(c[1]); g1[3] = bfhi(c[1]);
            g2[0] = bflo(d[0]); g2[1] = bfhi(d[0]); g2[2] = bflo(d[1]); g2[3] = bfhi(d[1]);
        }
        if (p.cos) {
            const int pos = p.pos ? p.pos[m] : pos_m;
            const f32x4 c = *reinterpret_cast<const f32x4*>(p.cos + (size_t)pos * half + i);
            const f32x4 sn = *reinterpret_cast<const f32x4*>(p.sin + (size_t)pos * half + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t1 = g1[e] * c[e] + g2[e] * sn[e], t2 = g2[e] * c[e] - g1[e] * sn[e];
                g1[e] = t1; g2[e] = t2;
            }
        }
        if (isq && p.q_scale != 1.f) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { g1[e] *= p.q_scale; g2[e] *= p.q_scale; }
        }
        const bf16_t* w = isq ? p.qw : p.kw;
        float d1[4], d2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { d1[e] = g1[e]; d2[e] = g2[e]; }
        if (w) {
            const u32x2 wa = *reinterpret_cast<const u32x2*>(w + i), wb = *reinterpret_cast<const u32x2*>(w + i + half);
            const float w1[4] = {bflo(wa[0]), bfhi(wa[0]), bflo(wa[1]), bfhi(wa[1])};
            const float w2[4] = {bflo(wb[0]), bfhi(wb[0]), bflo(wb[1]), bfhi(wb[1])};
            float ss = 0.f, dot = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ss += x1[e] * x1[e] + x2[e] * x2[e];
                dot += g1[e] * w1[e] * x1[e] + g2[e] * w2[e] * x2[e];
            }
            if (tph == 16 || tph == 8) {
                // the 8 / 16 lanes of a head are a DPP row (or half of one): sums by VALU-only cross-lane adds instead of LDS permutes
                ss = tph == 16 ? dpp_row_sum<16>(ss) : dpp_row_sum<8>(ss);
                dot = tph == 16 ? dpp_row_sum<16>(dot) : dpp_row_sum<8>(dot);
            } else {
                for (int o = tph >> 1; o > 0; o >>= 1) {
                    ss += __shfl_xor(ss, o, 64);
                    dot += __shfl_xor(dot, o, 64);
                }
            }
            const float rstd = rsqrtf(ss / (float)p.hd + p.eps);
            const float coef = dot * rstd * rstd * rstd / (float)p.hd;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                d1[e] = g1[e] * w1[e] * rstd - x1[e] * coef;
                d2[e] = g2[e] * w2[e] * rstd - x2[e] * coef;
                if (live) {
                    if (isq) { dwq1[e] += g1[e] * x1[e] * rstd; dwq2[e] += g2[e] * x2[e] * rstd; }
                    else     { dwk1[e] += g1[e] * x1[e] * rstd; dwk2[e] += g2[e] * x2[e] * rstd; }
                }
            }
        }
        if (live) {
            bf16_t* d = p.dsrc + (size_t)m * p.ld_out + head * p.hd;
            st_stream<u32x2>(d + i, u32x2{pack_bf2(d1[0], d1[1]), pack_bf2(d1[2], d1[3])});
            st_stream<u32x2>(d + i + half, u32x2{pack_bf2(d2[0], d2[1]), pack_bf2(d2[2], d2[3])});
        }
    }
    // block partial of the gain gradients, in a FIXED order (no LDS atomics: their arrival order would make the fp32 sums —
    // and once in a while a bf16 rounding of the result — differ from run to run): every thread parks its 16 sums in its own
    // row [sub][2*hd], then 2*hd threads add the rows top to bottom
    __syncthreads();
    float* row = sdw + (size_t)sub * 2 * p.hd;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        row[i + e] = dwq1[e];
        row[i + e + half] = dwq2[e];
        row[p.hd + i + e] = dwk1[e];
        row[p.hd + i + e + half] = dwk2[e];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * p.hd; t += 256) {
        float acc = 0.f;
        for (int sb = 0; sb < heads_per_it; ++sb) acc += sdw[(size_t)sb * 2 * p.hd + t];
        p.dw_part[(size_t)blockIdx.x * 2 * p.hd + t] = acc;
    }
}

// ------------------------------------------------------------------------------------------------
// SwiGLU (HF:models/qwen3/modeling_qwen3.py:82): act = silu(gate) * up ; gu = [gate | up] per row
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* __restrict__ gu, bf16_t* __restrict__ out, long rows,
                                                         int ff) {
    const int nch = ff >> 3;
    const long total = rows * nch;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const long r = t / nch;
        const int c = (int)(t % nch);
        const u32x4 g = ld_stream<u32x4>(gu + (size_t)r * 2 * ff + c * 8);
        const u32x4 u = ld_stream<u32x4>(gu + (size_t)r * 2 * ff + ff + c * 8);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float ga = bflo(g[e]), gb = bfhi(g[e]);
            // HF computes silu in bf16 (rounds), then the product (rounds)
            const float sa = bf2f(f2bf(ga * sigmoid_fast(ga))), sb = bf2f(f2bf(gb * sigmoid_fast(gb)));
            o[e] = pack_bf2(sa * bflo(u[e]), sb * bfhi(u[e]));
        }
        st_stream<u32x4>(out + (size_t)r * ff + c * 8, o);
    }
}

__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ gu, const bf16_t* __restrict__ dout,
                                                         bf16_t* __restrict__ dgu, long rows, int ff) {
    const int nch = ff >> 3;
    const long total = rows * nch;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const long r = t / nch;
        const int c = (int)(t % nch);
        const u32x4 g = ld_stream<u32x4>(gu + (size_t)r * 2 * ff + c * 8);
        const u32x4 u = ld_stream<u32x4>(gu + (size_t)r * 2 * ff + ff + c * 8);
        const u32x4 d = ld_stream<u32x4>(dout + (size_t)r * ff + c * 8);
        u32x4 og, ou;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const u32x2 r = swiglu_bwd_pair(g[e], u[e], d[e]);
            og[e] = r[0];
            ou[e] = r[1];
        }
        st_stream<u32x4>(dgu + (size_t)r * 2 * ff + c * 8, og);
        st_stream<u32x4>(dgu + (size_t)r * 2 * ff + ff + c * 8, ou);
    }
}

// ------------------------------------------------------------------------------------------------
// row gather / scatter:  dst[dst_idx[i] or i, :] = src[src_idx[i] or i, :]   (H % 8 == 0), idx < 0 skips
// embedding lookup (reference src/model/omics_one.py:164) = gather with int64 ids;
// omic injection (src/model/omics_one.py:93-97) = scatter with an int32 row map.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy_rows_kernel(const bf16_t* __restrict__ src, const long* __restrict__ src_idx64,
                                                        const int* __restrict__ src_idx32, bf16_t* __restrict__ dst,
                                                        const int* __restrict__ dst_idx32, long n, int H, int ld_src,
                                                        int ld_dst, int accumulate) {
    const int nch = H >> 3;
    const long total = n * nch;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const long i = t / nch;
        const int c = (int)(t % nch);
        long s = src_idx64 ? src_idx64[i] : (src_idx32 ? (long)src_idx32[i] : i);
        long d = dst_idx32 ? (long)dst_idx32[i] : i;
        if (s < 0 || d < 0) continue;
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + (size_t)s * ld_src + c * 8);
        u32x4* dp = reinterpret_cast<u32x4*>(dst + (size_t)d * ld_dst + c * 8);
        if (accumulate) {
            const u32x4 o = *dp;
            u32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = pack_bf2(bflo(o[e]) + bflo(v[e]), bfhi(o[e]) + bfhi(v[e]));
            *dp = r;
        } else {
            *dp = v;
        }
    }
}

// embedding backward: dE[id] += g[row] for every row whose `keep[row]` != 0, summed in fp32 through a sorted
// index (rows grouped by id): seg_start[u]..seg_start[u+1] are positions in `order` of unique id `uid[u]`.
// One wave per unique id -> deterministic, no atomics.
__global__ __launch_bounds__(256) void embed_bwd_kernel(const bf16_t* __restrict__ g, const int* __restrict__ order,
                                                        const int* __restrict__ seg_start, const long* __restrict__ uid,
                                                        int n_unique, bf16_t* __restrict__ dE, int H, int ld_g,
                                                        const float* __restrict__ row_scale,
                                                        const int* __restrict__ n_unique_dev) {
    const int lane = threadIdx.x & 63;
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= n_unique || (n_unique_dev && u >= *n_unique_dev)) return;
    const int b = seg_start[u], e = seg_start[u + 1];
    const long id = uid[u];
    if (id < 0) return;
    const int nch = H >> 3;
    for (int c = lane; c < nch; c += 64) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = b; k < e; ++k) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(g + (size_t)order[k] * ld_g + c * 8);
            const float rs = row_scale ? row_scale[order[k]] : 1.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[2 * j] += rs * bflo(v[j]); acc[2 * j + 1] += rs * bfhi(v[j]); }
        }
        u32x4* dp = reinterpret_cast<u32x4*>(dE + (size_t)id * H + c * 8);
        const u32x4 o = *dp;
        u32x4 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = pack_bf2(bflo(o[j]) + acc[2 * j], bfhi(o[j]) + acc[2 * j + 1]);
        *dp = r;
    }
}

// Many column reductions in ONE launch: item t reduces part_t[nb_t][row_stride_t] (first H_t columns) into out_t.  The decoder
// backward leaves the per-block partials of its 2L+1 RMSNorm gain gradients and 2L q/k-norm gain gradients in their own
// workspaces and reduces them all here, once, at the end — instead of one 64-block colsum launch (12.8 us on an otherwise
// full timeline) behind every norm backward.  Same summation tree per item as colsum_kernel: bit-identical results.
struct ColsumItem { const float* part; void* out; int nb, H, row_stride, pad; };
__global__ __launch_bounds__(256) void colsum_batched_kernel(const ColsumItem* __restrict__ items, int out_f32, int accumulate) {
    __shared__ float red[32][33];
    const ColsumItem it = items[blockIdx.y];
    if (blockIdx.x * 32 >= it.H) return;
    const int cq = threadIdx.x & 7, rg = threadIdx.x >> 3;
    const int j = blockIdx.x * 32 + cq * 4;
    f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
    if (j < it.H)
        for (int b = rg; b < it.nb; b += 32) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(it.part + (size_t)b * it.row_stride + j);
            s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
        }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[rg][cq * 4 + e] = s[e];
    __syncthreads();
    const int c = threadIdx.x, jj = blockIdx.x * 32 + c;
    if (c < 32 && jj < it.H) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 32; ++g) t += red[g][c];
        if (out_f32) {
            float* o = reinterpret_cast<float*>(it.out);
            o[jj] = accumulate ? o[jj] + t : t;
        } else {
            bf16_t* o = reinterpret_cast<bf16_t*>(it.out);
            o[jj] = f2bf(accumulate ? bf2f(o[jj]) + t : t);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Classifier-head losses of the Enc-Head baselines (reference: baselines/model.py:196-204): logits [rows][V] bf16 with
// any V (row pitch ld >= V; columns V..ld-1 are padding and get zero gradient).  One wave per row.
//   mode 0: F.cross_entropy(logits, labels)          labels int64 [rows] (ignore_index rows: zero loss / gradient)
//   mode 1: F.binary_cross_entropy_with_logits(logits, targets)   targets fp32 [rows][V]
// row_loss[r] = the row's summed loss; the logits are overwritten by d(logits) = dloss/dlogit * (*scale).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void cls_loss_kernel(bf16_t* __restrict__ logits, const long* __restrict__ labels,
                                                      const float* __restrict__ targets, float* __restrict__ row_loss,
                                                      const float* __restrict__ scale, int V, int ld, int mode,
                                                      int ignore_index, int write_grad) {
    const long row = blockIdx.x;
    const int lane = threadIdx.x;
    bf16_t* lr = logits + (size_t)row * ld;
    const float sc = write_grad ? *scale : 0.f;
    float loss = 0.f;
    if (mode == 0) {
        const long lab = labels[row];
        if (lab == ignore_index || lab < 0 || lab >= V) {
            if (lane == 0) row_loss[row] = lab == ignore_index ? 0.f : __builtin_nanf("");   // torch asserts on a bad label
            if (write_grad && lab == ignore_index)
                for (int c = lane; c < ld; c += 64) lr[c] = 0;
            return;
        }
        float mx = -INFINITY;
        for (int c = lane; c < V; c += 64) mx = fmaxf(mx, bf2f(lr[c]));
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sm = 0.f;
        for (int c = lane; c < V; c += 64) sm += __expf(bf2f(lr[c]) - mx);
        for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
        const float lse = mx + __logf(sm);
        loss = lse - bf2f(lr[lab]);
        if (write_grad)
            for (int c = lane; c < ld; c += 64) {
                const float p = c < V ? __expf(bf2f(lr[c]) - lse) - (c == lab ? 1.f : 0.f) : 0.f;
                lr[c] = f2bf(p * sc);
            }
    } else {
        const float* t = targets + (size_t)row * V;
        for (int c = lane; c < ld; c += 64) {
            float g = 0.f;
            if (c < V) {
                const float x = bf2f(lr[c]), y = t[c];
                // max(x, 0) - x y + log(1 + exp(-|x|))  (torch's stable form)
                loss += fmaxf(x, 0.f) - x * y + log1pf(__expf(-fabsf(x)));
                g = 1.f / (1.f + __expf(-x)) - y;
            }
            if (write_grad) lr[c] = f2bf(g * sc);
        }
        for (int o = 32; o > 0; o >>= 1) loss += __shfl_xor(loss, o, 64);
    }
    if (lane == 0) row_loss[row] = loss;
}

// ------------------------------------------------------------------------------------------------
// cross-entropy on a chunk of bf16 logits, in place -> d(logits)   (HF:loss/loss_utils.py:32-71)
// labels are ALREADY shifted (row r predicts labels[r]); ignore_index rows get zero grad / zero loss.
// dlogits = (softmax - onehot) * (*scale)   with *scale = 1/n_valid computed on device beforehand.
// one block per row; pass 1 online (max,sum), pass 2 rewrite.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_fwd_bwd_kernel(bf16_t* __restrict__ logits, const long* __restrict__ labels,
                                                         float* __restrict__ row_loss, const float* __restrict__ scale,
                                                         int V, int ld, int ignore_index, int write_grad) {
    __shared__ float red[16];
    const long row = blockIdx.x;
    bf16_t* lr = logits + (size_t)row * ld;
    const long lab = labels[row];
    const int nch = V >> 3;
    if (lab == ignore_index) {
        if (threadIdx.x == 0) row_loss[row] = 0.f;
        if (write_grad) {
            for (int c = threadIdx.x; c < nch; c += 256) *reinterpret_cast<u32x4*>(lr + c * 8) = u32x4{0, 0, 0, 0};
        }
        return;
    }
    if (lab < 0 || lab >= V) {             // torch asserts here; we poison the loss instead of reading OOB
        if (threadIdx.x == 0) row_loss[row] = __builtin_nanf("");
        return;
    }
    float mx = -INFINITY, sm = 0.f;
    for (int c = threadIdx.x; c < nch; c += 256) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(lr + c * 8);
        float f[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { f[2 * e] = bflo(v[e]); f[2 * e + 1] = bfhi(v[e]); }
        float cm = f[0];
#pragma unroll
        for (int e = 1; e < 8; ++e) cm = fmaxf(cm, f[e]);
        const float nm = fmaxf(mx, cm);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += __expf(f[e] - nm);
        sm = sm * __expf(mx - nm) + s;
        mx = nm;
    }
    const float gmx = block_max(mx, red);
    sm = sm * __expf(mx - gmx);            // mx=-inf lanes (no chunk): exp(-inf)=0 and sm=0
    if (mx == -INFINITY) sm = 0.f;
    const float gsm = block_sum(sm, red);
    const float lse = gmx + __logf(gsm);
    const float xl = bf2f(lr[lab]);
    __syncthreads();
    if (threadIdx.x == 0) row_loss[row] = lse - xl;
    if (!write_grad) return;
    const float sc = *scale;
    for (int c = threadIdx.x; c < nch; c += 256) {
        u32x4* pv = reinterpret_cast<u32x4*>(lr + c * 8);
        const u32x4 v = *pv;
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a = __expf(bflo(v[e]) - lse), b = __expf(bfhi(v[e]) - lse);
            const long j = (long)c * 8 + 2 * e;
            if (j == lab) a -= 1.f;
            if (j + 1 == lab) b -= 1.f;
            o[e] = pack_bf2(a * sc, b * sc);
        }
        *pv = o;
    }
}

// n_valid = #(labels != ignore) ; scale_out = 1/max(n_valid,1) ; count_out = n_valid   (single block)
__global__ __launch_bounds__(1024) void count_valid_kernel(const long* __restrict__ labels, long n, int ignore_index,
                                                           float* scale_out, float* count_out) {
    __shared__ float red[16];
    float c = 0.f;
    for (long i = threadIdx.x; i < n; i += 1024) c += (labels[i] != ignore_index) ? 1.f : 0.f;
    c = block_sum(c, red);
    if (threadIdx.x == 0) {
        *count_out = c;
        // no scored token: the reference's mean over zero tokens is NaN (HF:loss/loss_utils.py:32-47, reduction="mean").
        // The LOSS reproduces that; the gradients stay exactly zero (ignored rows write literal zeros), where the reference
        // would hand NaN gradients to the optimizer — a deliberate, documented difference.
        *scale_out = c > 0.f ? 1.f / c : __builtin_nanf("");
    }
}

// deterministic sum of a float vector, optionally scaled by *scale: out = sum(x) * (*scale)  (single block)
__global__ __launch_bounds__(1024) void sum_f32_kernel(const float* __restrict__ x, long n, const float* scale, float* out,
                                                       int accumulate) {
    __shared__ float red[16];
    float c = 0.f;
    for (long i = threadIdx.x; i < n; i += 1024) c += x[i];
    c = block_sum(c, red);
    if (threadIdx.x == 0) {
        const float v = c * (scale ? *scale : 1.f);
        *out = accumulate ? *out + v : v;
    }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm forward (ESM: HF:models/esm/modeling_esm.py:429,518,552; nn.LayerNorm eps 1e-5, affine)
// ------------------------------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            const bf16_t* __restrict__ b, bf16_t* __restrict__ y, int rows,
                                                            int H, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nch = H >> 3;
    const u32x4* xr = reinterpret_cast<const u32x4*>(x + (size_t)row * H);
    u32x4 v[NC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            v[i] = xr[c];
#pragma unroll
            for (int e = 0; e < 4; ++e) s += bflo(v[i][e]) + bfhi(v[i][e]);
        }
    }
    const float mean = wave_sum(s) / (float)H;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = bflo(v[i][e]) - mean, bb = bfhi(v[i][e]) - mean;
                ss += a * a + bb * bb;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(ss) / (float)H + eps);
    const u32x4* wr = reinterpret_cast<const u32x4*>(w);
    const u32x4* br = reinterpret_cast<const u32x4*>(b);
    u32x4* yr = reinterpret_cast<u32x4*>(y + (size_t)row * H);
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = lane + i * 64;
        if (c < nch) {
            const u32x4 wv = wr[c], bv = br[c];
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                o[e] = pack_bf2((bflo(v[i][e]) - mean) * rstd * bflo(wv[e]) + bflo(bv[e]),
                                (bfhi(v[i][e]) - mean) * rstd * bfhi(wv[e]) + bfhi(bv[e]));
            yr[c] = o;
        }
    }
}

// LayerNorm backward (HF nn.LayerNorm in the ESM blocks, HF:models/esm/modeling_esm.py:417,449,552): one wave per row, the
// row stays in registers; dx = rstd * (g*w - mean(g*w) - xhat * mean(g*w*xhat)) (+ dres); per-block fp32 partials of
// dw = sum g*xhat and db = sum g, reduced by colsum_kernel (deterministic, no atomics).  Encoder backward: `--train-bio`.
template <int NC>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            const bf16_t* __restrict__ g, const bf16_t* __restrict__ dres,
                                                            bf16_t* __restrict__ dx, float* __restrict__ part, int rows, int H,
                                                            float eps, int nb) {
    extern __shared__ __attribute__((aligned(16))) char sm_raw[];
    float* sdw = reinterpret_cast<float*>(sm_raw);          // [4 waves][64 lanes][8]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nch = H >> 3;
    float dwacc[NC][8], dbacc[NC][8];
#pragma unroll
    for (int i = 0; i < NC; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { dwacc[i][e] = 0.f; dbacc[i][e] = 0.f; }
    const u32x4* wr = reinterpret_cast<const u32x4*>(w);
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const u32x4* xr = reinterpret_cast<const u32x4*>(x + (size_t)row * H);
        const u32x4* gr = reinterpret_cast<const u32x4*>(g + (size_t)row * H);
        u32x4 xv[NC], gv[NC];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + i * 64;
            if (c < nch) {
                xv[i] = xr[c];
                gv[i] = gr[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) s += bflo(xv[i][e]) + bfhi(xv[i][e]);
            }
        }
        const float mean = wave_sum(s) / (float)H;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + i * 64;
            if (c < nch) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a = bflo(xv[i][e]) - mean, b = bfhi(xv[i][e]) - mean;
                    ss += a * a + b * b;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(ss) / (float)H + eps);
        float s1 = 0.f, s2 = 0.f;                               // sum(g*w), sum(g*w*xhat)
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + i * 64;
            if (c < nch) {
                const u32x4 wv = wr[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ha = (bflo(xv[i][e]) - mean) * rstd, hb = (bfhi(xv[i][e]) - mean) * rstd;
                    const float ga = bflo(gv[i][e]) * bflo(wv[e]), gb = bfhi(gv[i][e]) * bfhi(wv[e]);
                    s1 += ga + gb;
                    s2 += ga * ha + gb * hb;
                }
            }
        }
        s1 = wave_sum(s1) / (float)H;
        s2 = wave_sum(s2) / (float)H;
        u32x4* dxr = reinterpret_cast<u32x4*>(dx + (size_t)row * H);
        const u32x4* rr = dres ? reinterpret_cast<const u32x4*>(dres + (size_t)row * H) : nullptr;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + i * 64;
            if (c < nch) {
                const u32x4 wv = wr[c];
                u32x4 rv = u32x4{0, 0, 0, 0};
                if (rr) rv = rr[c];
                u32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ha = (bflo(xv[i][e]) - mean) * rstd, hb = (bfhi(xv[i][e]) - mean) * rstd;
                    const float ga = bflo(gv[i][e]), gb = bfhi(gv[i][e]);
                    float da = rstd * (ga * bflo(wv[e]) - s1 - ha * s2);
                    float db = rstd * (gb * bfhi(wv[e]) - s1 - hb * s2);
                    if (rr) { da += bflo(rv[e]); db += bfhi(rv[e]); }
                    o[e] = pack_bf2(da, db);
                    dwacc[i][2 * e] += ga * ha; dwacc[i][2 * e + 1] += gb * hb;
                    dbacc[i][2 * e] += ga;      dbacc[i][2 * e + 1] += gb;
                }
                dxr[c] = o;
            }
        }
    }
    // 4 waves -> block partial through LDS; part = [dw: nb x H][db: nb x H]
#pragma unroll
    for (int which = 0; which < 2; ++which) {
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = lane + i * 64;
            __syncthreads();
            if (c < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e) sdw[(wave * 64 + lane) * 8 + e] = which ? dbacc[i][e] : dwacc[i][e];
            }
            __syncthreads();
            if (wave == 0 && c < nch) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    part[((size_t)which * nb + blockIdx.x) * H + c * 8 + e] =
                        sdw[(0 * 64 + lane) * 8 + e] + sdw[(1 * 64 + lane) * 8 + e] + sdw[(2 * 64 + lane) * 8 + e] +
                        sdw[(3 * 64 + lane) * 8 + e];
            }
        }
    }
}

// erf-GELU (HF ACT2FN["gelu"], the ESM intermediate activation: HF:models/esm/modeling_esm.py:56-60) on a stored
// pre-activation, and its derivative  0.5 (1 + erf(z/sqrt2)) + z exp(-z^2/2) / sqrt(2 pi)
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16_t* __restrict__ z, bf16_t* __restrict__ out, long nch) {
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < nch; t += (long)gridDim.x * 256) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(z + t * 8);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = bflo(v[e]), b = bfhi(v[e]);
            o[e] = pack_bf2(0.5f * a * (1.f + fast_erf(a * 0.70710678118654752f)), 0.5f * b * (1.f + fast_erf(b * 0.70710678118654752f)));
        }
        *reinterpret_cast<u32x4*>(out + t * 8) = o;
    }
}

__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* __restrict__ z, const bf16_t* __restrict__ dout,
                                                       bf16_t* __restrict__ dz, long nch) {
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < nch; t += (long)gridDim.x * 256) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(z + t * 8);
        const u32x4 d = *reinterpret_cast<const u32x4*>(dout + t * 8);
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = bflo(v[e]), b = bfhi(v[e]);
            const float ga = 0.5f * (1.f + fast_erf(a * 0.70710678118654752f)) + a * __expf(-0.5f * a * a) * 0.3989422804014327f;
            const float gb = 0.5f * (1.f + fast_erf(b * 0.70710678118654752f)) + b * __expf(-0.5f * b * b) * 0.3989422804014327f;
            o[e] = pack_bf2(bflo(d[e]) * ga, bfhi(d[e]) * gb);
        }
        *reinterpret_cast<u32x4*>(dz + t * 8) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// ESM embeddings (HF:models/esm/modeling_esm.py:224-271,1050-1063).  gridDim.y blocks per sequence, each rebuilding the
// sequence's statistics (K ids: nothing) and writing its share of the K x H output rows:
// mask = ids != pad ; token-dropout rescale (1-0.12)/(1-n_mask/n_valid) ; masked-token rows zeroed ;
// absolute position ids = cumsum(mask)*mask + pad ; output multiplied by mask.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void esm_embed_kernel(const long* __restrict__ ids, const bf16_t* __restrict__ wemb,
                                                        const bf16_t* __restrict__ pemb, bf16_t* __restrict__ out,
                                                        int* __restrict__ pos_out, int* __restrict__ klen_out, int K, int H,
                                                        int pad_id, int mask_id, int token_dropout) {
    extern __shared__ __attribute__((aligned(16))) char sm_raw[];
    int* spos = reinterpret_cast<int*>(sm_raw);     // [K] position ids
    __shared__ float red[16];
    const int seq = blockIdx.x;
    const long* sid = ids + (size_t)seq * K;
    float nv = 0.f, nm = 0.f;
    for (int t = threadIdx.x; t < K; t += 256) {
        const long id = sid[t];
        nv += (id != pad_id) ? 1.f : 0.f;
        nm += (id == mask_id) ? 1.f : 0.f;
    }
    nv = block_sum(nv, red);
    nm = block_sum(nm, red);
    for (int t = threadIdx.x; t < K; t += 256) spos[t] = sid[t] != pad_id;
    __syncthreads();
    if (threadIdx.x == 0) {
        // serial cumsum over LDS: K <= 8192, negligible next to the encoder GEMMs
        int run = 0;
        int last_valid = 0;
        for (int t = 0; t < K; ++t) {
            const int mk = spos[t];
            run += mk;
            spos[t] = mk ? run + pad_id : pad_id;
            if (mk) last_valid = t + 1;
        }
        if (klen_out && blockIdx.y == 0) klen_out[seq] = last_valid;
    }
    __syncthreads();
    const int nch = H >> 3;
    const int per = (K + gridDim.y - 1) / gridDim.y, t_lo = blockIdx.y * per, t_hi = min(K, t_lo + per);
    for (int t = t_lo * nch + threadIdx.x; t < t_hi * nch; t += 256) {
        const int tok = t / nch, c = t % nch;
        const long id = sid[tok];
        u32x4 o = u32x4{0, 0, 0, 0};
        if (id != pad_id) {
            float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            if (!(token_dropout && id == mask_id)) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(wemb + (size_t)id * H + c * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) { f[2 * e] = bflo(v[e]); f[2 * e + 1] = bfhi(v[e]); }
                if (token_dropout) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = bf2f(f2bf(f[e] * (1.f - 0.15f * 0.8f) / (1.f - nm / nv)));
                }
            }
            if (pemb) {
                const u32x4 pv = *reinterpret_cast<const u32x4*>(pemb + (size_t)spos[tok] * H + c * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) { f[2 * e] += bflo(pv[e]); f[2 * e + 1] += bfhi(pv[e]); }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = pack_bf2(f[2 * e], f[2 * e + 1]);
        }
        *reinterpret_cast<u32x4*>(out + ((size_t)seq * K + tok) * H + c * 8) = o;
    }
    if (pos_out && blockIdx.y == 0)
        for (int t = threadIdx.x; t < K; t += 256) pos_out[(size_t)seq * K + t] = spos[t];
}

// ------------------------------------------------------------------------------------------------
// optimizer: squared-norm partials, clip coefficient, AdamW shard step
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sqnorm_part_kernel(const bf16_t* __restrict__ g, long n, float* __restrict__ part) {
    __shared__ float red[16];
    float s = 0.f;
    const long nch = n >> 3;
    for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < nch; c += (long)gridDim.x * 256) {
        const u32x4 v = ld_stream<u32x4>(g + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float a = bflo(v[e]), b = bfhi(v[e]); s += a * a + b * b; }
    }
    if (blockIdx.x == 0)
        for (long i = (nch << 3) + threadIdx.x; i < n; i += 256) { const float a = bf2f(g[i]); s += a * a; }
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// out[i] = bf16( sum_r float(in[r][i]) ), r = 0 .. rows-1 in that fixed order: the local half of the all-to-all
// reduce-scatter (Zero2Optimizer rs_algo="a2a"): every rank receives its chunk from every peer over that peer's own xGMI link
// and sums the `world` copies in fp32 — one rounding, an order that does not depend on any collective algorithm.
__global__ __launch_bounds__(256) void reduce_rows_kernel(const bf16_t* __restrict__ in, int rows, long n, bf16_t* __restrict__ out) {
    const long nch = n >> 3;
    for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < nch; c += (long)gridDim.x * 256) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int r = 0; r < rows; ++r) {
            const u32x4 v = ld_stream<u32x4>(in + (size_t)r * n + c * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[2 * j] += bflo(v[j]); acc[2 * j + 1] += bfhi(v[j]); }
        }
        u32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = pack_bf2(acc[2 * j], acc[2 * j + 1]);
        *reinterpret_cast<u32x4*>(out + c * 8) = o;
    }
}

// norm_sq (device scalar, possibly all-reduced across ranks) -> total norm, clip coefficient
// coef = min(1, max_norm / (norm + 1e-6)) * pre_scale       (torch.nn.utils.clip_grad_norm_)
// A non-finite norm (an inf/NaN gradient somewhere) makes the coefficient NaN = "skip this step": adamw_kernel leaves the
// master, the moments and the parameters untouched for it, and *skipped (optional) counts such steps — what DeepSpeed's
// ZeRO step does on overflow (the reference survives a bad batch; DeepSpeed 0.16.9 stage_1_and_2.py step(), un-vendored).
__global__ void clip_coef_kernel(const float* norm_sq, float max_norm, float pre_scale, float* norm_out, float* coef_out,
                                 float* skipped) {
    const float n = sqrtf(*norm_sq) * pre_scale;
    *norm_out = n;
    if (!isfinite(n)) {
        *coef_out = __builtin_nanf("");
        if (skipped) *skipped += 1.f;
        return;
    }
    float c = max_norm > 0.f ? fminf(1.f, max_norm / (n + 1e-6f)) : 1.f;
    *coef_out = c * pre_scale;
}

// torch.optim.AdamW single-tensor math on an fp32 master shard; grad bf16 scaled by *gscale; emits bf16 param
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ master, float* __restrict__ m, float* __restrict__ v,
                                                    const bf16_t* __restrict__ grad, bf16_t* __restrict__ param_out, long n,
                                                    float lr, float b1, float b2, float eps, float wd, int step,
                                                    const float* __restrict__ gscale, const float* __restrict__ skipped) {
    const float gs = gscale ? *gscale : 1.f;
    if (gs != gs) return;                                   // skipped step (clip_coef_kernel): nothing changes
    // bias corrections of the steps actually TAKEN (skipped ones do not advance Adam's step count)
    const float t = (float)step - (skipped ? *skipped : 0.f);
    const float bc1 = 1.f - powf(b1, t);
    const float bc2_sqrt = sqrtf(1.f - powf(b2, t));
    const long nch = n >> 2;
    // two 4-element chunks in flight per thread (a block sweeps 512 consecutive chunks), one sweep per thread: the launch
    // covers the shard without a grid-stride loop (tools/stream_diag: 5.7 -> 6.2 TB/s with the nontemporal hint)
    for (long c0 = (long)blockIdx.x * 512 + threadIdx.x; c0 < nch; c0 += (long)gridDim.x * 512) {
        f32x4 p[2], mm[2], vv[2];
        u32x2 gr[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long c = c0 + u * 256;
            if (c < nch) {
                p[u] = ld_stream<f32x4>(master + c * 4);
                mm[u] = ld_stream<f32x4>(m + c * 4);
                vv[u] = ld_stream<f32x4>(v + c * 4);
                gr[u] = ld_stream<u32x2>(grad + c * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const long c = c0 + u * 256;
            if (c >= nch) continue;
            const float g[4] = {bflo(gr[u][0]) * gs, bfhi(gr[u][0]) * gs, bflo(gr[u][1]) * gs, bfhi(gr[u][1]) * gs};
            // Every operation is named (no contraction left to the compiler): under -ffp-contract=fast hipcc fused the four
            // elements of a chunk differently (packed fma for two of them), so an element's result depended on its position in
            // its chunk, i.e. on where a bucket or shard starts — ZeRO-2 and ZeRO-0 runs of the same step then differed in the
            // last fp32 bit of thousands of masters (tests/test_gpu_two_ranks.py caught the second-step norm moving by 1.4e-4).
            const float decay = 1.f - lr * wd, omb1 = 1.f - b1, omb2 = 1.f - b2, nstep = -(lr / bc1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pd = __fmul_rn(p[u][e], decay);
                mm[u][e] = __fmaf_rn(b1, mm[u][e], __fmul_rn(omb1, g[e]));
                vv[u][e] = __fmaf_rn(b2, vv[u][e], __fmul_rn(__fmul_rn(omb2, g[e]), g[e]));
                const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(vv[u][e]), bc2_sqrt), eps);
                p[u][e] = __fmaf_rn(nstep, __fdiv_rn(mm[u][e], denom), pd);
            }
            st_stream<f32x4>(master + c * 4, p[u]);
            st_stream<f32x4>(m + c * 4, mm[u]);
            st_stream<f32x4>(v + c * 4, vv[u]);
            st_stream<u32x2>(param_out + c * 4, u32x2{pack_bf2(p[u][0], p[u][1]), pack_bf2(p[u][2], p[u][3])});
        }
    }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long n) {
    const long nch = n >> 2;
    for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < nch; c += (long)gridDim.x * 256) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(in + c * 4);
        *reinterpret_cast<u32x2*>(out + c * 4) = u32x2{pack_bf2(p[0], p[1]), pack_bf2(p[2], p[3])};
    }
    if (blockIdx.x == 0)
        for (long i = (nch << 2) + threadIdx.x; i < n; i += 256) out[i] = f2bf(in[i]);
}

__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = bf2f(in[i]);
}

// column sums of a bf16 matrix, stage 1: part[p][j] = sum over the p-th row slab of x[:, j]
__global__ __launch_bounds__(256) void colsum_bf16_part_kernel(const bf16_t* __restrict__ x, int rows, int H, int ld,
                                                               float* __restrict__ part, int rows_per_part) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= H) return;
    const int r0 = blockIdx.y * rows_per_part, r1 = min(rows, r0 + rows_per_part);
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += bf2f(x[(size_t)r * ld + j]);
    part[(size_t)blockIdx.y * H + j] = s;
}


// One chunk per thread: a streaming launch covers its range without a grid-stride loop (the kernels keep the loop for ranges
// beyond the cap).  Measured against a 2,048-block grid-stride launch (tools/stream_diag): float4 copy 5.0 -> 6.2 TB/s,
// AdamW shard step 5.7 -> 5.9, SwiGLU backward 5.25 -> 5.75.
#ifndef MOLLY_GRID_CAP
#define MOLLY_GRID_CAP (1 << 24)
#endif
inline int grid_for(long items, int per_block = 256, int cap = MOLLY_GRID_CAP) {
    long g = (items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}

}  // namespace

#define ST ((hipStream_t)stream)
// pick the smallest register-resident chunk count that covers H (512 elements per chunk)
#define NC_DISPATCH(H, F)          \
    do {                           \
        if ((H) <= 512) F(1);      \
        else if ((H) <= 1024) F(2);\
        else if ((H) <= 1536) F(3);\
        else if ((H) <= 2048) F(4);\
        else if ((H) <= 2560) F(5);\
        else F(8);                 \
    } while (0)

extern "C" int molly_transpose_bf16(void* stream, const void* in, void* out, int R, int C, int ld_in, int ld_out) {
    MOLLY_ENTER();
    MOLLY_CHECK(R > 0 && C > 0 && ld_in >= C && ld_out >= R, "transpose: bad shape R=%d C=%d", R, C);
    if (R % 64 == 0 && C % 64 == 0 && ld_in % 8 == 0 && ld_out % 8 == 0 && ((uintptr_t)in % 16) == 0 &&
        ((uintptr_t)out % 16) == 0) {
        hipLaunchKernelGGL(transpose64_kernel, dim3(C / 64, R / 64), dim3(256), 0, ST, (const bf16_t*)in, (bf16_t*)out, ld_in,
                           ld_out);
        MOLLY_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(C, 64), cdiv(R, 64)), dim3(256), 0, ST, (const bf16_t*)in, (bf16_t*)out, R,
                       C, ld_in, ld_out);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_rmsnorm_fwd(void* stream, const void* x, const void* w, void* y, float* rstd, int rows, int H,
                                 float eps) {
    MOLLY_ENTER();
    MOLLY_CHECK(rows > 0 && H % 8 == 0 && H <= RN_MAXC * 512, "rmsnorm: H=%d must be a multiple of 8 and <= %d", H,
                RN_MAXC * 512);
#define RMS_FWD(NC)                                                                                              \
    hipLaunchKernelGGL(rmsnorm_fwd_kernel<NC>, dim3(cdiv(rows, 4)), dim3(256), 0, ST, (const bf16_t*)x, (const bf16_t*)w, \
                       (bf16_t*)y, rstd, rows, H, eps)
    NC_DISPATCH(H, RMS_FWD);
#undef RMS_FWD
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_rmsnorm_fwd_t(void* stream, const void* x, const void* w, void* y, void* yT, int rows, int H, int ld_t, float eps) {
    MOLLY_ENTER();
    MOLLY_CHECK(rows > 0 && rows % 64 == 0 && H % 512 == 0 && H <= 2048, "rmsnorm_fwd_t: rows=%d (a multiple of 64) H=%d (a multiple of 512, <= 2048)", rows, H);
    MOLLY_CHECK(ld_t >= rows && ld_t % 8 == 0 && ((uintptr_t)yT % 16) == 0, "rmsnorm_fwd_t: ld_t=%d / alignment", ld_t);
    const size_t lds = 64 * TP_PITCH * sizeof(bf16_t);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)rmsnorm_fwd_t_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)rmsnorm_fwd_t_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)rmsnorm_fwd_t_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)rmsnorm_fwd_t_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
#define RMS_FWD_T(NC) hipLaunchKernelGGL(rmsnorm_fwd_t_kernel<NC>, dim3(rows / 64), dim3(512), lds, ST, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, (bf16_t*)yT, ld_t, H, eps)
    switch (H / 512) { case 1: RMS_FWD_T(1); break; case 2: RMS_FWD_T(2); break; case 3: RMS_FWD_T(3); break; default: RMS_FWD_T(4); break; }
#undef RMS_FWD_T
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_rmsnorm_bwd_blocks(int rows) { return grid_for(rows, 4, 1024); }

extern "C" int molly_rmsnorm_bwd(void* stream, const void* x, const void* w, const void* g, const void* dres, void* dx,
                                 void* dw, int dw_f32, int dw_accumulate, float* workspace, int rows, int H, float eps) {
    MOLLY_ENTER();
    MOLLY_CHECK(rows > 0 && H % 8 == 0 && H <= RN_MAXC * 512, "rmsnorm_bwd: bad H=%d", H);
    MOLLY_CHECK(workspace, "rmsnorm_bwd: workspace of molly_rmsnorm_bwd_blocks(rows)*H floats required");
    const int nb = molly_rmsnorm_bwd_blocks(rows);
#define RMS_BWD(NC)                                                                                                  \
    hipLaunchKernelGGL(rmsnorm_bwd_kernel<NC>, dim3(nb), dim3(256), 4 * 64 * 8 * sizeof(float), ST, (const bf16_t*)x,   \
                       (const bf16_t*)w, (const bf16_t*)g, (const bf16_t*)dres, (bf16_t*)dx, workspace, rows, H, eps)
    NC_DISPATCH(H, RMS_BWD);
#undef RMS_BWD
    MOLLY_LAUNCH_CHECK();
    if (dw) {            // dw == NULL: the partials stay in `workspace` for a later molly_colsum_batched
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(H, 32)), dim3(256), 0, ST, workspace, nb, H, H, dw, dw_f32, dw_accumulate);
        MOLLY_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int molly_colsum_batched(void* stream, const void* items_dev, int n_items, int max_H, int out_f32, int accumulate) {
    MOLLY_ENTER();
    MOLLY_CHECK(items_dev && n_items >= 1 && n_items <= 65535 && max_H >= 1, "colsum_batched: %d items, max_H=%d", n_items, max_H);
    hipLaunchKernelGGL(colsum_batched_kernel, dim3(cdiv(max_H, 32), n_items), dim3(256), 0, ST, (const ColsumItem*)items_dev,
                       out_f32, accumulate);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_norm_rope_fwd(void* stream, const void* src, void* dst, const void* q_norm_w, const void* k_norm_w,
                                   const float* cos, const float* sin, const int* positions, int M, int T, int n_q_heads,
                                   int n_k_heads, int head_dim, int ld_src, int ld_dst, float eps, float q_scale) {
    MOLLY_ENTER();
    MOLLY_CHECK(head_dim >= 16 && head_dim <= 512 && (head_dim & (head_dim - 1)) == 0, "norm_rope: head_dim=%d", head_dim);
    MOLLY_CHECK(ld_src % 4 == 0 && ld_dst % 4 == 0 && ((uintptr_t)src % 8) == 0 && ((uintptr_t)dst % 8) == 0, "norm_rope: 8-byte alignment required");
    MOLLY_CHECK((q_norm_w == nullptr) == (k_norm_w == nullptr), "norm_rope: give both norm gains or neither");
    MOLLY_CHECK((cos == nullptr) == (sin == nullptr), "norm_rope: give both cos and sin or neither");
    RopeArgs p{(const bf16_t*)src, (bf16_t*)dst, (const bf16_t*)q_norm_w, (const bf16_t*)k_norm_w, cos, sin, positions,
               M, T, n_q_heads, n_k_heads, head_dim, ld_src, ld_dst, eps, q_scale, nullptr, nullptr, nullptr, 0, 0};
    const long items = (long)M * (n_q_heads + n_k_heads);
    const int hpb = 256 / (head_dim / 8);
    static const bool fast = [] { const char* e = getenv("MOLLY_ROPE_FWD_FAST"); return !e || atoi(e) != 0; }();
    if (fast && items > 0 && items < (1L << 31) && T > 0 && (head_dim == 128 || head_dim == 64)) {      // (norm_rope_fwd_fast_kernel / _rows_kernel)
        static const int rows_min = [] { const char* e = getenv("MOLLY_ROPE_FWD_ROWS_MIN"); return e ? atoi(e) : 1024; }();   // workgroups of the rows form
        const int rpb = hpb;                                 // token rows per workgroup: one group of hd/8 lanes each
        if ((M + rpb - 1) / rpb >= rows_min) {
            const unsigned nbr = (unsigned)((M + rpb - 1) / rpb);
            if (head_dim == 128) hipLaunchKernelGGL(norm_rope_fwd_rows_kernel<128>, dim3(nbr), dim3(256), 0, ST, p);
            else hipLaunchKernelGGL(norm_rope_fwd_rows_kernel<64>, dim3(nbr), dim3(256), 0, ST, p);
            MOLLY_LAUNCH_CHECK();
            return 0;
        }
        const unsigned nb = (unsigned)((items + 2 * hpb - 1) / (2 * hpb));
        if (head_dim == 128) hipLaunchKernelGGL(norm_rope_fwd_fast_kernel<128>, dim3(nb), dim3(256), 0, ST, p);
        else hipLaunchKernelGGL(norm_rope_fwd_fast_kernel<64>, dim3(nb), dim3(256), 0, ST, p);
        MOLLY_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(norm_rope_fwd_kernel, dim3((unsigned)((items + hpb - 1) / hpb)), dim3(256), 0, ST, p);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_norm_rope_cache_fwd(void* stream, const void* src, void* dst, const void* q_norm_w, const void* k_norm_w,
                                         const float* cos, const float* sin, const int* positions, int M, int T, int n_q_heads,
                                         int n_k_heads, int head_dim, int ld_src, int ld_dst, float eps, float q_scale,
                                         void* kcache, void* vcache, const int* slot, int ld_cache) {
    MOLLY_ENTER();
    MOLLY_CHECK(head_dim >= 16 && head_dim <= 512 && (head_dim & (head_dim - 1)) == 0, "norm_rope_cache: head_dim=%d", head_dim);
    MOLLY_CHECK(ld_src % 4 == 0 && ld_dst % 4 == 0 && ld_cache % 4 == 0 && ((uintptr_t)src % 8) == 0 && ((uintptr_t)dst % 8) == 0 &&
                    ((uintptr_t)kcache % 8) == 0 && ((uintptr_t)vcache % 8) == 0, "norm_rope_cache: 8-byte alignment required");
    MOLLY_CHECK((q_norm_w == nullptr) == (k_norm_w == nullptr), "norm_rope_cache: give both norm gains or neither");
    MOLLY_CHECK((cos == nullptr) == (sin == nullptr), "norm_rope_cache: give both cos and sin or neither");
    MOLLY_CHECK(kcache && vcache && slot && ld_src >= (n_q_heads + 2 * n_k_heads) * head_dim,
                "norm_rope_cache: caches, slots and a src row of q | k | v heads are required");
    RopeArgs p{(const bf16_t*)src, (bf16_t*)dst, (const bf16_t*)q_norm_w, (const bf16_t*)k_norm_w, cos, sin, positions,
               M, T, n_q_heads, n_k_heads, head_dim, ld_src, ld_dst, eps, q_scale, (bf16_t*)kcache, (bf16_t*)vcache, slot, ld_cache, n_k_heads};
    const long items = (long)M * (n_q_heads + 2 * n_k_heads);
    const int hpb = 256 / (head_dim / 8);
    hipLaunchKernelGGL(norm_rope_fwd_kernel, dim3((unsigned)((items + hpb - 1) / hpb)), dim3(256), 0, ST, p);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

// workgroups of the norm + rope backward (each leaves one row of gain-gradient partials).  What bounds the kernel is the bytes it keeps in
// flight: a thread has one item's four 8-byte loads outstanding, so 1,024 workgroups = 4 per CU = 32 KB per CU ran at 3.5-3.8 TB/s;
// MOLLY_ROPE_BWD_BLOCKS overrides (A/B)
extern "C" int molly_norm_rope_bwd_blocks(void) {
    static const int nb = [] { const char* e = getenv("MOLLY_ROPE_BWD_BLOCKS"); return e ? atoi(e) : 1024; }();
    return nb;
}

extern "C" int molly_norm_rope_bwd(void* stream, const void* src, const void* g, void* dsrc, const void* q_norm_w,
                                   const void* k_norm_w, const float* cos, const float* sin, const int* positions,
                                   void* dq_w, void* dk_w, int dw_f32, int dw_accumulate, float* workspace, int M, int T,
                                   int n_q_heads, int n_k_heads, int head_dim, int ld_src, int ld_g, int ld_out,
                                   float eps, float q_scale) {
    MOLLY_ENTER();
    MOLLY_CHECK(head_dim >= 16 && head_dim <= 512 && (head_dim & (head_dim - 1)) == 0, "norm_rope_bwd: head_dim=%d",
                head_dim);
    MOLLY_CHECK(workspace, "norm_rope_bwd: workspace of molly_norm_rope_bwd_blocks()*2*head_dim floats required");
    const int nb = molly_norm_rope_bwd_blocks();
    const long items = (long)M * (n_q_heads + n_k_heads);
    const int hpi = 256 / (head_dim / 8);
    long ipb = (items + nb - 1) / nb;
    ipb = (ipb + hpi - 1) / hpi * hpi;
    RopeBwdArgs p{(const bf16_t*)src, (const bf16_t*)g, (bf16_t*)dsrc, (const bf16_t*)q_norm_w, (const bf16_t*)k_norm_w,
                  cos, sin, positions, workspace, M, T, n_q_heads, n_k_heads, head_dim, ld_src, ld_g, ld_out, eps, ipb, q_scale};
    hipLaunchKernelGGL(norm_rope_bwd_kernel, dim3(nb), dim3(256), 4096 * sizeof(float), ST, p);
    MOLLY_LAUNCH_CHECK();
    if (q_norm_w && (dq_w || dk_w)) {    // both NULL: partials stay in `workspace` ([nb][2*head_dim]: q | k) for molly_colsum_batched
        MOLLY_CHECK(dq_w && dk_w, "norm_rope_bwd: give both gain-gradient outputs or neither");
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(head_dim, 32)), dim3(256), 0, ST, workspace, nb, head_dim, 2 * head_dim,
                           dq_w, dw_f32, dw_accumulate);
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(head_dim, 32)), dim3(256), 0, ST, workspace + head_dim, nb, head_dim,
                           2 * head_dim, dk_w, dw_f32, dw_accumulate);
        MOLLY_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int molly_swiglu_fwd(void* stream, const void* gate_up, void* out, long rows, int ff) {
    MOLLY_ENTER();
    MOLLY_CHECK(rows > 0 && ff % 8 == 0, "swiglu: ff=%d must be a multiple of 8", ff);
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for(rows * (ff / 8))), dim3(256), 0, ST, (const bf16_t*)gate_up,
                       (bf16_t*)out, rows, ff);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_swiglu_bwd(void* stream, const void* gate_up, const void* dout, void* dgate_up, long rows, int ff) {
    MOLLY_ENTER();
    MOLLY_CHECK(rows > 0 && ff % 8 == 0, "swiglu_bwd: ff=%d must be a multiple of 8", ff);
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for(rows * (ff / 8))), dim3(256), 0, ST, (const bf16_t*)gate_up,
                       (const bf16_t*)dout, (bf16_t*)dgate_up, rows, ff);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_copy_rows(void* stream, const void* src, const int64_t* src_idx64, const int* src_idx32, void* dst,
                               const int* dst_idx32, long n, int H, int ld_src, int ld_dst, int accumulate) {
    MOLLY_ENTER();
    MOLLY_CHECK(n >= 0 && H % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0, "copy_rows: H/ld must be multiples of 8");
    MOLLY_CHECK(!(src_idx64 && src_idx32), "copy_rows: give at most one source index");
    if (n == 0) return 0;
    hipLaunchKernelGGL(copy_rows_kernel, dim3(grid_for(n * (H / 8))), dim3(256), 0, ST, (const bf16_t*)src,
                       (const long*)src_idx64, src_idx32, (bf16_t*)dst, dst_idx32, n, H, ld_src, ld_dst, accumulate);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_embed_bwd(void* stream, const void* g, const int* order, const int* seg_start, const int64_t* uid,
                               int n_unique, void* dE, int H, int ld_g, const float* row_scale, const int* n_unique_dev) {
    MOLLY_ENTER();
    MOLLY_CHECK(H % 8 == 0 && ld_g % 8 == 0, "embed_bwd: H must be a multiple of 8");
    if (n_unique <= 0) return 0;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(cdiv(n_unique, 4)), dim3(256), 0, ST, (const bf16_t*)g, order, seg_start,
                       (const long*)uid, n_unique, (bf16_t*)dE, H, ld_g, row_scale, n_unique_dev);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_count_valid(void* stream, const int64_t* labels, long n, int ignore_index, float* scale_out,
                                 float* count_out) {
    MOLLY_ENTER();
    hipLaunchKernelGGL(count_valid_kernel, dim3(1), dim3(1024), 0, ST, (const long*)labels, n, ignore_index, scale_out,
                       count_out);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_ce_fwd_bwd(void* stream, void* logits, const int64_t* labels, float* row_loss, const float* scale,
                                int rows, int V, int ld, int ignore_index, int write_grad) {
    MOLLY_ENTER();
    MOLLY_CHECK(rows > 0 && V % 8 == 0 && ld % 8 == 0, "ce: V=%d and ld=%d must be multiples of 8", V, ld);
    hipLaunchKernelGGL(ce_fwd_bwd_kernel, dim3(rows), dim3(256), 0, ST, (bf16_t*)logits, (const long*)labels, row_loss,
                       scale, V, ld, ignore_index, write_grad);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_cls_loss_fwd_bwd(void* stream, void* logits, const int64_t* labels, const float* targets,
                                     float* row_loss, const float* scale, int rows, int V, int ld, int mode,
                                     int ignore_index, int write_grad) {
    MOLLY_ENTER();
    MOLLY_CHECK(rows > 0 && V > 0 && ld >= V, "cls_loss: rows=%d V=%d ld=%d", rows, V, ld);
    MOLLY_CHECK(mode == 0 ? labels != nullptr : (mode == 1 && targets != nullptr), "cls_loss: mode %d without its labels", mode);
    MOLLY_CHECK(!write_grad || scale, "cls_loss: gradient requested without a scale");
    hipLaunchKernelGGL(cls_loss_kernel, dim3(rows), dim3(64), 0, ST, (bf16_t*)logits, (const long*)labels, targets, row_loss,
                       scale, V, ld, mode, ignore_index, write_grad);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

// greedy token selection of HF generate (do_sample=False: torch.argmax over the vocabulary; the FIRST maximal index wins
// ties, a NaN counts as the maximum): one block per row of fp32 logits.
// PARTS > 1: block (row, part) scans the part's columns and leaves (value, index) in part_v / part_i [rows][PARTS]; argmax_merge_kernel picks
// per row (32 rows alone occupy 32 of 256 CUs: 30 us for the 0.6 MB rows of a 152 k vocabulary, against 8 as 32 x 8 blocks + the merge)
__global__ __launch_bounds__(1024) void argmax_f32_kernel(const float* __restrict__ x, long* __restrict__ out, int V, int ld, int parts,
                                                          float* __restrict__ part_v, int* __restrict__ part_i) {
    // 1,024 threads per row, 16-byte loads, four in flight per thread (the 0.6 MB row of a 152 k vocabulary: one 256-thread block with 4-byte
    // loads took 40+ us of the decode step; torch.argmax 55)
    __shared__ float sv[1024];
    __shared__ int si[1024];
    const int row = blockIdx.x / parts, part = blockIdx.x % parts;
    // the part's columns [c_lo, c_hi): multiples of 4,096 (whole rounds of the vector loop) except the last part's end
    const int per = ((V + parts - 1) / parts + 4095) / 4096 * 4096;
    const int c_lo = min(part * per, V), c_hi = min(c_lo + per, V);
    const float* r = x + (size_t)row * ld + c_lo;
    V = c_hi - c_lo;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    bool nan = false;
    auto see = [&](float v, int c) {
        if (v != v) { if (!nan) { nan = true; bi = c; } }
        else if (!nan && (v > best || (bi == 0x7fffffff))) { best = v; bi = c; }
    };
    const bool vec = (V & 3) == 0 && (ld & 3) == 0 && ((uintptr_t)x & 15) == 0;      // (c_lo is a multiple of 4)
    if (vec) {
        const int nv = V >> 2;
        int c = threadIdx.x;
        for (; c + 3 * 1024 < nv; c += 4 * 1024) {
            f32x4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = *reinterpret_cast<const f32x4*>(r + 4 * (size_t)(c + u * 1024));
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) see(q[u][e], 4 * (c + u * 1024) + e);
        }
        for (; c < nv; c += 1024) {
            const f32x4 q = *reinterpret_cast<const f32x4*>(r + 4 * (size_t)c);
#pragma unroll
            for (int e = 0; e < 4; ++e) see(q[e], 4 * c + e);
        }
    } else {
        for (int c = threadIdx.x; c < V; c += 1024) see(r[c], c);
    }
    sv[threadIdx.x] = nan ? INFINITY : best;
    si[threadIdx.x] = nan ? bi - 0x40000000 : bi;          // NaN lanes sort in front of every number, by index among themselves
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const float a = sv[threadIdx.x], b = sv[threadIdx.x + o];
            const int ia = si[threadIdx.x], ib = si[threadIdx.x + o];
            const bool na = ia < 0, nb = ib < 0;
            const bool take_b = (nb && !na) || (nb == na && (b > a || (b == a && ib < ia)));
            if (take_b) { sv[threadIdx.x] = b; si[threadIdx.x] = ib; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // (index: relative to the part; an empty part leaves 0x7fffffff and -inf, which loses to everything)
        const bool isnan = si[0] < 0;
        const int rel = isnan ? si[0] + 0x40000000 : si[0];
        if (parts == 1) out[row] = rel;
        else {
            part_v[blockIdx.x] = isnan ? NAN : sv[0];
            part_i[blockIdx.x] = rel == 0x7fffffff ? rel : rel + c_lo;
        }
    }
}
__global__ __launch_bounds__(64) void argmax_merge_kernel(const float* __restrict__ part_v, const int* __restrict__ part_i, long* __restrict__ out,
                                                          int parts, int rows) {
    // parts ascend in column order: the first NaN, else the first strictly larger value, wins
    const int row = blockIdx.x * 64 + threadIdx.x;
    if (row >= rows) return;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int p = 0; p < parts; ++p) {
        const float v = part_v[row * parts + p];
        const int i = part_i[row * parts + p];
        if (i == 0x7fffffff) continue;
        if (v != v) { bi = i; break; }
        if (v > best || bi == 0x7fffffff) { best = v; bi = i; }
    }
    out[row] = bi;
}

extern "C" int molly_argmax_f32(void* stream, const float* x, int64_t* out, int rows, int V, int ld) {
    MOLLY_ENTER();
    MOLLY_CHECK(rows > 0 && V > 0 && V < 0x40000000 && ld >= V, "argmax: rows=%d V=%d ld=%d", rows, V, ld);
    hipLaunchKernelGGL(argmax_f32_kernel, dim3(rows), dim3(1024), 0, ST, x, (long*)out, V, ld, 1, nullptr, nullptr);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

// the same with every row cut over up to 8 blocks (workspace: molly_argmax_workspace(rows) bytes)
extern "C" int molly_argmax_workspace(int rows) { return rows * 8 * 8 + 512; }
extern "C" int molly_argmax_f32_ws(void* stream, const float* x, int64_t* out, int rows, int V, int ld, void* workspace, long workspace_bytes) {
    MOLLY_ENTER();
    MOLLY_CHECK(rows > 0 && V > 0 && V < 0x40000000 && ld >= V, "argmax: rows=%d V=%d ld=%d", rows, V, ld);
    int parts = 256 / rows;
    parts = parts > 8 ? 8 : parts;
    if (parts > (V + 8191) / 8192) parts = (V + 8191) / 8192;
    if (parts <= 1 || workspace == nullptr || workspace_bytes < molly_argmax_workspace(rows))
        return molly_argmax_f32(stream, x, out, rows, V, ld);
    float* pv = (float*)workspace;
    int* pi = (int*)(pv + (size_t)rows * 8 + 64);
    hipLaunchKernelGGL(argmax_f32_kernel, dim3(rows * parts), dim3(1024), 0, ST, x, (long*)out, V, ld, parts, pv, pi);
    hipLaunchKernelGGL(argmax_merge_kernel, dim3((rows + 63) / 64), dim3(64), 0, ST, pv, pi, (long*)out, parts, rows);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_sum_f32(void* stream, const float* x, long n, const float* scale, float* out, int accumulate) {
    MOLLY_ENTER();
    hipLaunchKernelGGL(sum_f32_kernel, dim3(1), dim3(1024), 0, ST, x, n, scale, out, accumulate);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_layernorm_fwd(void* stream, const void* x, const void* w, const void* b, void* y, int rows, int H,
                                   float eps) {
    MOLLY_ENTER();
    MOLLY_CHECK(rows > 0 && H % 8 == 0 && H <= RN_MAXC * 512, "layernorm: bad H=%d", H);
#define LN_FWD(NC)                                                                                                 \
    hipLaunchKernelGGL(layernorm_fwd_kernel<NC>, dim3(cdiv(rows, 4)), dim3(256), 0, ST, (const bf16_t*)x, (const bf16_t*)w, \
                       (const bf16_t*)b, (bf16_t*)y, rows, H, eps)
    NC_DISPATCH(H, LN_FWD);
#undef LN_FWD
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_layernorm_bwd_blocks(int rows) { return grid_for(rows, 4, 1024); }

extern "C" int molly_layernorm_bwd(void* stream, const void* x, const void* w, const void* g, const void* dres, void* dx,
                                   void* dw, void* db, int dw_f32, int dw_accumulate, float* workspace, int rows, int H,
                                   float eps) {
    MOLLY_ENTER();
    MOLLY_CHECK(rows > 0 && H % 8 == 0 && H <= RN_MAXC * 512, "layernorm_bwd: bad H=%d", H);
    MOLLY_CHECK(workspace && dw && db, "layernorm_bwd: workspace of 2*molly_layernorm_bwd_blocks(rows)*H floats, dw, db required");
    const int nb = molly_layernorm_bwd_blocks(rows);
#define LN_BWD(NC)                                                                                                   \
    hipLaunchKernelGGL(layernorm_bwd_kernel<NC>, dim3(nb), dim3(256), 4 * 64 * 8 * sizeof(float), ST, (const bf16_t*)x, \
                       (const bf16_t*)w, (const bf16_t*)g, (const bf16_t*)dres, (bf16_t*)dx, workspace, rows, H, eps, nb)
    NC_DISPATCH(H, LN_BWD);
#undef LN_BWD
    MOLLY_LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(H, 32)), dim3(256), 0, ST, workspace, nb, H, H, dw, dw_f32, dw_accumulate);
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(H, 32)), dim3(256), 0, ST, workspace + (size_t)nb * H, nb, H, H, db, dw_f32,
                       dw_accumulate);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_gelu_fwd(void* stream, const void* z, void* out, long n) {
    MOLLY_ENTER();
    MOLLY_CHECK(n > 0 && n % 8 == 0, "gelu_fwd: n=%ld must be a positive multiple of 8", n);
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, ST, (const bf16_t*)z, (bf16_t*)out, n / 8);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_gelu_bwd(void* stream, const void* z, const void* dout, void* dz, long n) {
    MOLLY_ENTER();
    MOLLY_CHECK(n > 0 && n % 8 == 0, "gelu_bwd: n=%ld must be a positive multiple of 8", n);
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, ST, (const bf16_t*)z, (const bf16_t*)dout,
                       (bf16_t*)dz, n / 8);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_esm_embed(void* stream, const int64_t* ids, const void* word_emb, const void* pos_emb, void* out,
                               int* pos_ids_out, int* kv_len_out, int n_seq, int K, int H, int pad_id, int mask_id,
                               int token_dropout) {
    MOLLY_ENTER();
    MOLLY_CHECK(n_seq > 0 && K > 0 && K <= 8192 && H % 8 == 0, "esm_embed: bad shape n_seq=%d K=%d H=%d", n_seq, K, H);
    // enough blocks to fill the chip whatever the number of sequences (one block per sequence left 8 CUs busy)
    const int parts = max(1, min(K / 8, cdiv(1024, n_seq)));
    hipLaunchKernelGGL(esm_embed_kernel, dim3(n_seq, parts), dim3(256), K * sizeof(int), ST, (const long*)ids,
                       (const bf16_t*)word_emb, (const bf16_t*)pos_emb, (bf16_t*)out, pos_ids_out, kv_len_out, K, H, pad_id,
                       mask_id, token_dropout);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_sqnorm_blocks(void) { return 1024; }

extern "C" int molly_sqnorm_bf16(void* stream, const void* g, long n, float* workspace, float* out, int accumulate) {
    MOLLY_ENTER();
    MOLLY_CHECK(workspace && ((uintptr_t)g % 16) == 0, "sqnorm: workspace required, 16-byte aligned input");
    const int nb = molly_sqnorm_blocks();
    hipLaunchKernelGGL(sqnorm_part_kernel, dim3(nb), dim3(256), 0, ST, (const bf16_t*)g, n, workspace);
    MOLLY_LAUNCH_CHECK();
    hipLaunchKernelGGL(sum_f32_kernel, dim3(1), dim3(1024), 0, ST, workspace, (long)nb, (const float*)nullptr, out,
                       accumulate);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_reduce_rows_bf16(void* stream, const void* in, int rows, long n, void* out) {
    MOLLY_ENTER();
    MOLLY_CHECK(rows >= 1 && n >= 0 && n % 8 == 0, "reduce_rows: rows=%d, n=%ld must be a multiple of 8", rows, n);
    MOLLY_CHECK(((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0, "reduce_rows: buffers must be 16-byte aligned");
    if (n == 0) return 0;
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(grid_for(n / 8)), dim3(256), 0, ST, (const bf16_t*)in, rows, n, (bf16_t*)out);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_clip_coef(void* stream, const float* norm_sq, float max_norm, float pre_scale, float* norm_out,
                               float* coef_out, float* skipped_count_or_null) {
    MOLLY_ENTER();
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(1), 0, ST, norm_sq, max_norm, pre_scale, norm_out, coef_out,
                       skipped_count_or_null);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_adamw_step(void* stream, float* master, float* exp_avg, float* exp_avg_sq, const void* grad,
                                void* param_out, long n, float lr, float beta1, float beta2, float eps, float weight_decay,
                                int step, const float* grad_scale, const float* skipped_count_or_null) {
    MOLLY_ENTER();
    MOLLY_CHECK(n % 4 == 0 && step >= 1, "adamw: n=%ld must be a multiple of 4 and step >= 1", n);
    if (n == 0) return 0;
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4, 512)), dim3(256), 0, ST, master, exp_avg, exp_avg_sq,
                       (const bf16_t*)grad, (bf16_t*)param_out, n, lr, beta1, beta2, eps, weight_decay, step,
                       grad_scale, skipped_count_or_null);
    MOLLY_LAUNCH_CHECK();
    return 0;
}


extern "C" int molly_colsum_parts(int rows) { return rows >= 1024 ? 64 : (rows >= 64 ? 16 : 1); }

extern "C" int molly_colsum_bf16(void* stream, const void* x, int rows, int H, int ld, void* out, int out_f32,
                                 int accumulate, float* workspace) {
    MOLLY_ENTER();
    MOLLY_CHECK(rows > 0 && H > 0 && H % 4 == 0 && workspace,
                "colsum: H=%d must be a positive multiple of 4; workspace of molly_colsum_parts(rows)*H floats required", H);
    const int np = molly_colsum_parts(rows);
    const int rpp = cdiv(rows, np);
    hipLaunchKernelGGL(colsum_bf16_part_kernel, dim3(cdiv(H, 256), np), dim3(256), 0, ST, (const bf16_t*)x, rows, H, ld,
                       workspace, rpp);
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(H, 32)), dim3(256), 0, ST, workspace, np, H, H, out, out_f32, accumulate);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_cast_f32_to_bf16(void* stream, const float* in, void* out, long n) {
    MOLLY_ENTER();
    if (n == 0) return 0;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, ST, in, (bf16_t*)out, n);
    MOLLY_LAUNCH_CHECK();
    return 0;
}

extern "C" int molly_cast_bf16_to_f32(void* stream, const void* in, float* out, long n) {
    MOLLY_ENTER();
    if (n == 0) return 0;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n)), dim3(256), 0, ST, (const bf16_t*)in, out, n);
    MOLLY_LAUNCH_CHECK();
    return 0;
}
