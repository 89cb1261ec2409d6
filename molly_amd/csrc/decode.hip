// Single-query (decode-step) attention over a token-major KV cache — the per-token step of HF `generate` with a
// DynamicCache that the reference runs for inference (reference src/model/omics_one.py:220-232; attention arithmetic
// HF:models/qwen3/modeling_qwen3.py:185-208 with q_len = 1).
//
// HBM-bound: every step streams the whole K and V cache of each sample once (2 * kv_len * n_kv_heads * hd * 2 B).
//  * one block = (sample, KV head, key range): the G = n_heads / n_kv_heads query heads that share the KV head are
//    processed together, so K/V rows are read ONCE (GQA), 16 B per lane, whole rows coalesced (hd*2 B contiguous);
//  * split-KV ("flash-decoding"): the key range of a sample is cut into `splits` chunks so B * n_kv_heads * splits blocks
//    fill the chip; each writes an un-normalised partial (m, l, acc) and a small second kernel merges them;
//  * 8 independent 16-B loads in flight per lane (4 keys x {K, V}) and online softmax with one rescale per 4 keys;
//  * the arithmetic per key and query head is what bounds a group of 4 heads (Qwen3-8B: 3.45 TB/s against 5.0 for the groups of
//    2 of Qwen3-1.7B with the first form of this kernel), so it is cut to the bone: q . k as four v_dot2_f32_bf16 on the raw
//    bf16 pairs (no conversions, scale applied to the sum), the sum over the 8 / 16 lanes of a key row by DPP adds (quad_perm,
//    row_half_mirror, row_mirror: one VALU instruction each, no LDS permute), p . v as packed fp32 FMAs on V converted once per key.
#include "common.h"
#include "molly_hip.h"

namespace {

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// sum over the LPR (8 | 16) lanes that hold one key row, result in every one of them
template <int LPR>
__device__ __forceinline__ float row_sum(float d) {
    d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0xB1, 0xf, 0xf, true));       // quad_perm [1,0,3,2]
    d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x4E, 0xf, 0xf, true));       // quad_perm [2,3,0,1]
    d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x141, 0xf, 0xf, true));      // row_half_mirror
    if (LPR == 16) d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x140, 0xf, 0xf, true));   // row_mirror
    return d;
}

constexpr float NEG_BIG = -1e30f;       // finite "-inf": exp2(NEG_BIG - x) == 0 and NEG_BIG - NEG_BIG == 0 (no NaN)
constexpr int MAX_SPLITS = 16;

struct DecodeArgs {
    const bf16_t* q; const bf16_t* kc; const bf16_t* vc; bf16_t* out;
    const int* lo; const int* hi;
    float* part;                        // [B][n_heads][splits][HD + 2] fp32: acc[HD], m, l   (splits > 1)
    int Tmax, nh, nkv, ldq, splits;
    float qscale;                       // softmax scale * log2(e): scores live in the exp2 domain
    // FUSED (molly_attn_decode_qkv): q, and the step's new key / value, come from the K-slice slabs of the q | k | v projection
    const float* slabs; int n_slabs, ldn;                  // [n_slabs][B][ldn] fp32, ldn = (nh + 2 nkv) * HD
    const bf16_t* qw; const bf16_t* kw; const float* cos; const float* sin; const int* pos; const int* slot;
    bf16_t* kc_w; bf16_t* vc_w;                            // the caches again, writable
    float eps;
};

// NW waves per block: 4 with the key range of a (sample, KV head) cut over `splits` blocks, or 16 where B * n_kv_heads blocks fill the chip by
// themselves (one 1,024-thread block per CU keeps the same 16 waves and 8 loads per lane in flight as four 256-thread ones, and its 16
// partials meet in LDS: no partials in HBM, no merge launch).
// FUSED: the step's q | k | v projection arrives as the fp32 K-slice slabs the decode-row GEMM leaves (molly_gemm_rows_slabs_bf16_ctx); the
// block adds the slices of its G query heads and its key / value head, applies q/k-norm and rotary (same arithmetic, in the same order, as
// rows_tail_qkv_kernel / norm_rope_fwd_kernel), appends the new key and value to the caches and attends to them from LDS — the old keys
// [lo, hi - 1) stream from the cache as before.  One launch where the step had three (slab combine + norm + rope + append, attention, merge).
template <int HD, int G, int NW, bool FUSED>
__global__ __launch_bounds__(64 * NW) void attn_decode_kernel(DecodeArgs p) {
    constexpr int LPR = HD / 8;                       // lanes per key row (16 B each)
    constexpr int RPW = 64 / LPR;                     // rows per wave-instruction
    constexpr int STEP = NW * RPW;                    // rows per block step
    constexpr int U = 4;                              // keys per lane per iteration
    __shared__ float sm[NW][G][HD + 2];
    __shared__ __attribute__((aligned(16))) bf16_t sq[FUSED ? G + 2 : 1][HD];      // FUSED: the G query heads, the new key, the new value
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane % LPR, rg = lane / LPR;
    int bid = blockIdx.x;
    const int s = bid % p.splits; bid /= p.splits;
    const int kvh = bid % p.nkv;
    const int b = bid / p.nkv;
    const int lo = p.lo ? p.lo[b] : 0, hi = p.hi[b] - (FUSED ? 1 : 0);           // FUSED: kv_hi counts the new token, which is not in the cache yet
    const int n = max(hi - lo, 0);
    int chunk = (n + p.splits - 1) / p.splits;
    chunk = (chunk + STEP * U - 1) / (STEP * U) * (STEP * U);
    const int k0 = lo + s * chunk, k1 = min(hi, k0 + chunk);
    const int ldc = p.nkv * HD;
    const bf16_t* kb = p.kc + (size_t)b * p.Tmax * ldc + kvh * HD + sub * 8;
    const bf16_t* vb = p.vc + (size_t)b * p.Tmax * ldc + kvh * HD + sub * 8;

    static_assert(LPR == 8 || LPR == 16, "row_sum: a key row lives in 8 or 16 lanes");
    u32x4 qv[G];                                     // the lane's 8 q values of every head of the group, raw bf16 pairs
    f32x2 acc[G][4];
    float m[G], l[G];
    // the first keys' loads go out BEFORE q exists (FUSED: the slab combine, norm and rotary below run under their latency)
    u32x4 kv[U], vv[U];
    bool ok[U];
    auto load_keys = [&](int key) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ku = key + u * STEP;
            ok[u] = ku < k1;
            const int kc_ = ok[u] ? ku : key;                     // clamp: a valid row, masked below
            // (non-temporal: every cache byte is read once per step, by one block)
            kv[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kb + (size_t)kc_ * ldc));
            vv[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vb + (size_t)kc_ * ldc));
        }
    };
    const int key_first = k0 + wave * RPW + rg;
    if (FUSED && key_first < k1) load_keys(key_first);
    if constexpr (FUSED) {
        constexpr int TPH = HD / 8, half = HD / 2;   // a thread owns elements i .. i+3 and their rotary partners i+half .. of one head
        constexpr int NPRO = ((G + 2) * TPH + 63) / 64 * 64;         // whole waves: the head's sum of squares is a DPP reduction
        if (tid < NPRO) {
            const bool live = tid < (G + 2) * TPH;
            const int hh = live ? tid / TPH : G + 1, i = (tid % TPH) * 4;
            const int col = (hh < G ? kvh * G + hh : hh == G ? p.nh + kvh : p.nh + p.nkv + kvh) * HD;
            const float* src = p.slabs + (size_t)b * p.ldn + col + i;
            const size_t MN = (size_t)(gridDim.x / (p.nkv * p.splits)) * p.ldn;
            f32x4 a = *reinterpret_cast<const f32x4*>(src), c = *reinterpret_cast<const f32x4*>(src + half);
            for (int s2 = 1; s2 < p.n_slabs; ++s2) {
                a += *reinterpret_cast<const f32x4*>(src + s2 * MN);
                c += *reinterpret_cast<const f32x4*>(src + s2 * MN + half);
            }
            float x1[4], x2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { x1[e] = bf2f(f2bf(a[e])); x2[e] = bf2f(f2bf(c[e])); }      // the projection's bf16 output
            const bf16_t* w = hh < G ? p.qw : p.kw;
            const float ss = head_lanes_sum(head_sumsq8(x1, x2), TPH);      // (every lane of the wave: the v head's sum is not used)
            if (hh <= G && w) head_norm8(x1, x2, rsqrtf(ss / (float)HD + p.eps), w, i, half);
            float y1[4], y2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { y1[e] = x1[e]; y2[e] = x2[e]; }
            if (hh <= G && p.cos) {
                const int pos = p.pos ? p.pos[b] : 0;
                const f32x4 cs = *reinterpret_cast<const f32x4*>(p.cos + (size_t)pos * half + i);
                const f32x4 sn = *reinterpret_cast<const f32x4*>(p.sin + (size_t)pos * half + i);
                head_rope8(x1, x2, cs, sn, y1, y2);
            }
            const u32x2 o1 = u32x2{pack_bf2(y1[0], y1[1]), pack_bf2(y1[2], y1[3])}, o2 = u32x2{pack_bf2(y2[0], y2[1]), pack_bf2(y2[2], y2[3])};
            if (live) {
                *reinterpret_cast<u32x2*>(&sq[hh][i]) = o1;
                *reinterpret_cast<u32x2*>(&sq[hh][i + half]) = o2;
            }
            if (live && hh >= G && s == p.splits - 1) {        // the append (one block per (sample, KV head) does it)
                bf16_t* d = (hh == G ? p.kc_w : p.vc_w) + (size_t)p.slot[b] * ldc + kvh * HD;
                *reinterpret_cast<u32x2*>(d + i) = o1;
                *reinterpret_cast<u32x2*>(d + i + half) = o2;
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if constexpr (FUSED) qv[g] = *reinterpret_cast<const u32x4*>(&sq[g][sub * 8]);
        else qv[g] = *reinterpret_cast<const u32x4*>(p.q + (size_t)b * p.ldq + (kvh * G + g) * HD + sub * 8);
        m[g] = NEG_BIG; l[g] = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[g][e] = f32x2{0.f, 0.f};
    }

    auto attend = [&]() {
        f32x2 vf[U][4];                              // V converted once per key, used by every head of the group
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) vf[u][e] = f32x2{bflo(vv[u][e]), bfhi(vv[u][e])};
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float sc[U];
            float mx = m[g];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // (through named scalars: __builtin_bit_cast applied to a vector ELEMENT reads element 0 for every e — the hipcc
                    // fold recorded in gemm.hip's residual epilogue; here it turned the 16-byte loads into 4-byte ones)
                    const unsigned kx = kv[u][e], qx = qv[g][e];
                    d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, kx), __builtin_bit_cast(bf16x2_t, qx), d, false);
                }
                d = row_sum<LPR>(d);
                sc[u] = ok[u] ? d * p.qscale : NEG_BIG;
                mx = fmaxf(mx, sc[u]);
            }
            const float c = __builtin_amdgcn_exp2f(m[g] - mx);
            m[g] = mx;
            float lsum = l[g] * c;
            const f32x2 c2 = f32x2{c, c};
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[g][e] *= c2;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float pe = __builtin_amdgcn_exp2f(sc[u] - mx);
                lsum += pe;
                const f32x2 pe2 = f32x2{pe, pe};
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[g][e] = __builtin_elementwise_fma(pe2, vf[u][e], acc[g][e]);
            }
            l[g] = lsum;
        }
    };
    int key = key_first;
    if (FUSED && key < k1) {                         // (the iteration whose loads went out in front of the prologue)
        attend();
        key += STEP * U;
    }
#pragma unroll 1
    for (; key < k1; key += STEP * U) {
        load_keys(key);
        attend();
    }

    if constexpr (FUSED) {
        // the step's own key and value, from LDS: row group 0 of wave 0 in the block that holds the end of the key range
        if (wave == 0 && s == p.splits - 1) {
            const u32x4 kx4 = *reinterpret_cast<const u32x4*>(&sq[G][sub * 8]), vx4 = *reinterpret_cast<const u32x4*>(&sq[G + 1][sub * 8]);
            f32x2 vn[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) vn[e] = f32x2{bflo(vx4[e]), bfhi(vx4[e])};
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned kx = kx4[e], qx = qv[g][e];
                    d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, kx), __builtin_bit_cast(bf16x2_t, qx), d, false);
                }
                d = row_sum<LPR>(d);
                const float sc = rg == 0 ? d * p.qscale : NEG_BIG;
                const float mx = fmaxf(m[g], sc);
                const float c = __builtin_amdgcn_exp2f(m[g] - mx);
                const float pe = rg == 0 ? __builtin_amdgcn_exp2f(sc - mx) : 0.f;
                m[g] = mx;
                l[g] = l[g] * c + pe;
                const f32x2 c2 = f32x2{c, c}, pe2 = f32x2{pe, pe};
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[g][e] = __builtin_elementwise_fma(pe2, vn[e], acc[g][e] * c2);
            }
        }
    }
    // merge the RPW row groups of the wave (butterfly over the lane bits above the row), then the NW waves through LDS
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) {
            const float mo = __shfl_xor(m[g], o, 64), lo_ = __shfl_xor(l[g], o, 64);
            const float mn = fmaxf(m[g], mo);
            const float c1 = __builtin_amdgcn_exp2f(m[g] - mn), c2 = __builtin_amdgcn_exp2f(mo - mn);
            l[g] = l[g] * c1 + lo_ * c2;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[g][e][0] = acc[g][e][0] * c1 + __shfl_xor(acc[g][e][0], o, 64) * c2;
                acc[g][e][1] = acc[g][e][1] * c1 + __shfl_xor(acc[g][e][1], o, 64) * c2;
            }
            m[g] = mn;
        }
        if (rg == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { sm[wave][g][sub * 8 + 2 * e] = acc[g][e][0]; sm[wave][g][sub * 8 + 2 * e + 1] = acc[g][e][1]; }
            if (sub == 0) { sm[wave][g][HD] = m[g]; sm[wave][g][HD + 1] = l[g]; }
        }
    }
    __syncthreads();
    for (int t = tid; t < G * HD; t += 64 * NW) {
        const int g = t / HD, d = t % HD;
        float mn = NEG_BIG;
#pragma unroll
        for (int w = 0; w < NW; ++w) mn = fmaxf(mn, sm[w][g][HD]);
        float a = 0.f, ls = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float c = __builtin_amdgcn_exp2f(sm[w][g][HD] - mn);
            a += sm[w][g][d] * c;
            ls += sm[w][g][HD + 1] * c;
        }
        const int head = kvh * G + g;
        if (p.splits == 1) {
            p.out[(size_t)b * p.nh * HD + head * HD + d] = f2bf(ls > 0.f ? a / ls : 0.f);
        } else {
            float* pp = p.part + (((size_t)b * p.nh + head) * p.splits + s) * (HD + 2);
            pp[d] = a;
            if (d == 0) { pp[HD] = mn; pp[HD + 1] = ls; }
        }
    }
}

// out[b][head][:] = merge of the `splits` partials (one block per (b, head), HD threads)
template <int HD>
__global__ void attn_decode_merge_kernel(const float* __restrict__ part, bf16_t* __restrict__ out, int splits) {
    const float* pp = part + (size_t)blockIdx.x * splits * (HD + 2);
    const int d = threadIdx.x;
    float mn = NEG_BIG;
    for (int s = 0; s < splits; ++s) mn = fmaxf(mn, pp[s * (HD + 2) + HD]);
    float a = 0.f, ls = 0.f;
    for (int s = 0; s < splits; ++s) {
        const float c = __builtin_amdgcn_exp2f(pp[s * (HD + 2) + HD] - mn);
        a += pp[s * (HD + 2) + d] * c;
        ls += pp[s * (HD + 2) + HD + 1] * c;
    }
    out[(size_t)blockIdx.x * HD + d] = f2bf(ls > 0.f ? a / ls : 0.f);
}

template <int HD, int G, bool FUSED>
void launch(hipStream_t st, const DecodeArgs& p, int B, int nw) {
    if (nw == 16) hipLaunchKernelGGL((attn_decode_kernel<HD, G, 16, FUSED>), dim3(B * p.nkv * p.splits), dim3(1024), 0, st, p);
    else hipLaunchKernelGGL((attn_decode_kernel<HD, G, 4, FUSED>), dim3(B * p.nkv * p.splits), dim3(256), 0, st, p);
    if (p.splits > 1)
        hipLaunchKernelGGL(attn_decode_merge_kernel<HD>, dim3(B * p.nh), dim3(HD), 0, st, p.part, p.out, p.splits);
}

// blocks of 16 waves (caches of >= 512 keys): one per (sample, KV head) where those cover >= 3/4 of the 256 CUs, else the key range cut so that ~256
// blocks exist; short caches: blocks of 4 waves and the key range cut so that ~4 per CU exist; >= 256 keys per block either way (kv_len_hint =
// upper bound of the valid length, 0 = Tmax).  MOLLY_DECODE_NW = 4 | 16, MOLLY_DECODE_BLOCKS, MOLLY_DECODE_MIN_KEYS override (A/B).
void pick_shape(int B, int n_kv_heads, int len, bool have_ws, int* nw, int* splits) {
    static const int target = [] { const char* e = getenv("MOLLY_DECODE_BLOCKS"); return e ? atoi(e) : 1024; }();
    static const int min_keys = [] { const char* e = getenv("MOLLY_DECODE_MIN_KEYS"); return e ? atoi(e) : 256; }();
    static const int force_nw = [] { const char* e = getenv("MOLLY_DECODE_NW"); return e ? atoi(e) : 0; }();
    const int pairs = B * n_kv_heads;
    // (16 waves at every batch: Qwen3-8B decode step, 3,100 keys, ms at 4 | 16 waves: B = 4 4.98 | 4.85, 8 5.33 | 5.18, 16 6.16 | 5.99, 24 6.62 | 6.01,
    // 28 7.02 | 6.32 — the 192 / 224 blocks of B = 24 / 28 leave CUs empty and still win; profiles/r04_logs/decode_nw.log)
    *nw = force_nw == 4 || force_nw == 16 ? force_nw : (len >= 512 ? 16 : 4);
    if (*nw == 16 && !have_ws) { *splits = 1; return; }
    // (16-wave blocks: one per CU is resident, so never more than 256 of them — 160 pairs stay 160 blocks rather than 320 in two rounds)
    int sp = *nw == 16 ? 256 / pairs : (target + pairs - 1) / pairs;
    sp = sp < 1 ? 1 : sp;
    if (sp > len / min_keys) sp = len / min_keys > 0 ? len / min_keys : 1;
    if (sp > MAX_SPLITS) sp = MAX_SPLITS;
    if (!have_ws) sp = 1;
    *splits = sp;
}

template <bool FUSED>
int dispatch(hipStream_t st, DecodeArgs& p, int B, int head_dim, int kv_len_hint, float* workspace, long workspace_floats) {
    const int G = p.nh / p.nkv;
    const int len = kv_len_hint > 0 ? kv_len_hint : p.Tmax;
    int nw, splits;
    pick_shape(B, p.nkv, len, workspace != nullptr, &nw, &splits);
    if (splits > 1 && workspace_floats < (long)B * p.nh * splits * (head_dim + 2)) splits = 1;
    p.splits = splits;
#define MOLLY_DEC(HD_, G_) else if (head_dim == HD_ && G == G_) launch<HD_, G_, FUSED>(st, p, B, nw)
    if (false) {}
    MOLLY_DEC(128, 1); MOLLY_DEC(128, 2); MOLLY_DEC(128, 4); MOLLY_DEC(128, 8);
    MOLLY_DEC(64, 1); MOLLY_DEC(64, 2); MOLLY_DEC(64, 4); MOLLY_DEC(64, 8);
    else {
        molly_set_error("attn_decode: n_heads / n_kv_heads = %d (built for 1, 2, 4, 8)", G);
        return 1;
    }
#undef MOLLY_DEC
    return 0;
}

}  // namespace

extern "C" int molly_attn_decode_workspace(int B, int n_heads, int head_dim) {
    return B * n_heads * MAX_SPLITS * (head_dim + 2);
}

extern "C" int molly_attn_decode(void* stream, const void* q, const void* kcache, const void* vcache, void* out,
                                 const int* kv_lo, const int* kv_hi, int B, int Tmax, int n_heads, int n_kv_heads, int head_dim,
                                 int ldq, float scale, int kv_len_hint, float* workspace, long workspace_floats) {
    MOLLY_ENTER();
    MOLLY_CHECK(head_dim == 64 || head_dim == 128, "attn_decode: head_dim=%d (built for 64 and 128)", head_dim);
    MOLLY_CHECK(B > 0 && Tmax > 0 && kv_hi != nullptr, "attn_decode: kv_hi required, B=%d Tmax=%d", B, Tmax);
    MOLLY_CHECK(n_heads % n_kv_heads == 0, "attn_decode: n_heads=%d not a multiple of n_kv_heads=%d", n_heads, n_kv_heads);
    MOLLY_CHECK(ldq % 8 == 0 && ((uintptr_t)q % 16) == 0 && ((uintptr_t)kcache % 16) == 0 && ((uintptr_t)vcache % 16) == 0,
                "attn_decode: q / caches must be 16-byte aligned, ldq %% 8 == 0");
    DecodeArgs p{};
    p.q = (const bf16_t*)q; p.kc = (const bf16_t*)kcache; p.vc = (const bf16_t*)vcache; p.out = (bf16_t*)out;
    p.lo = kv_lo; p.hi = kv_hi; p.part = workspace;
    p.Tmax = Tmax; p.nh = n_heads; p.nkv = n_kv_heads; p.ldq = ldq;
    p.qscale = scale * 1.44269504088896341f;
    if (int rc = dispatch<false>((hipStream_t)stream, p, B, head_dim, kv_len_hint, workspace, workspace_floats)) return rc;
    MOLLY_LAUNCH_CHECK();
    return 0;
}

// The decode step's attention taken straight from the q | k | v projection's K-slice slabs (molly_gemm_rows_slabs_bf16_ctx): row b of the
// projection is the sum of qkv_slabs[s][b][:] over the n_slabs slices, rounded to bf16; its q and k heads get q/k-norm (gains NULL: none) and
// rotary at positions[b] (cos NULL: none); the new key and value go to cache row slot[b] (of the [B * Tmax] rows) and the n_heads query heads
// attend to the cache keys [kv_lo[b], kv_hi[b] - 1) and to the new one — kv_hi counts the new token, as it does for molly_attn_decode after
// molly_gemm_rows_qkv_bf16_ctx, whose results this call reproduces (the caches bit for bit).  out [B][n_heads * head_dim] bf16.
extern "C" int molly_attn_decode_qkv(void* stream, const float* qkv_slabs, int n_slabs, const void* q_norm_w, const void* k_norm_w,
                                     const float* cos, const float* sin, const int* positions, float eps, void* kcache, void* vcache,
                                     const int* slot, void* out, const int* kv_lo, const int* kv_hi, int B, int Tmax, int n_heads,
                                     int n_kv_heads, int head_dim, float scale, int kv_len_hint, float* workspace, long workspace_floats) {
    MOLLY_ENTER();
    MOLLY_CHECK(head_dim == 64 || head_dim == 128, "attn_decode_qkv: head_dim=%d (built for 64 and 128)", head_dim);
    MOLLY_CHECK(B > 0 && Tmax > 0 && kv_hi && slot && qkv_slabs && n_slabs >= 1 && out, "attn_decode_qkv: kv_hi, slot, slabs, out required, B=%d Tmax=%d",
                B, Tmax);
    MOLLY_CHECK(n_heads % n_kv_heads == 0, "attn_decode_qkv: n_heads=%d not a multiple of n_kv_heads=%d", n_heads, n_kv_heads);
    MOLLY_CHECK((q_norm_w == nullptr) == (k_norm_w == nullptr) && (cos == nullptr) == (sin == nullptr), "attn_decode_qkv: norm gains / cos, sin in pairs");
    MOLLY_CHECK(((uintptr_t)qkv_slabs % 16) == 0 && ((uintptr_t)kcache % 16) == 0 && ((uintptr_t)vcache % 16) == 0,
                "attn_decode_qkv: slabs / caches must be 16-byte aligned");
    DecodeArgs p{};
    p.kc = (const bf16_t*)kcache; p.vc = (const bf16_t*)vcache; p.out = (bf16_t*)out;
    p.kc_w = (bf16_t*)kcache; p.vc_w = (bf16_t*)vcache;
    p.lo = kv_lo; p.hi = kv_hi; p.part = workspace;
    p.Tmax = Tmax; p.nh = n_heads; p.nkv = n_kv_heads; p.ldq = 0;
    p.qscale = scale * 1.44269504088896341f;
    p.slabs = qkv_slabs; p.n_slabs = n_slabs; p.ldn = (n_heads + 2 * n_kv_heads) * head_dim;
    p.qw = (const bf16_t*)q_norm_w; p.kw = (const bf16_t*)k_norm_w; p.cos = cos; p.sin = sin; p.pos = positions; p.slot = slot; p.eps = eps;
    if (int rc = dispatch<true>((hipStream_t)stream, p, B, head_dim, kv_len_hint, workspace, workspace_floats)) return rc;
    MOLLY_LAUNCH_CHECK();
    return 0;
}
