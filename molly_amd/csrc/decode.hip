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
};

template <int HD, int G>
__global__ __launch_bounds__(256) void attn_decode_kernel(DecodeArgs p) {
    constexpr int LPR = HD / 8;                       // lanes per key row (16 B each)
    constexpr int RPW = 64 / LPR;                     // rows per wave-instruction
    constexpr int STEP = 4 * RPW;                     // rows per block step
    constexpr int U = 4;                              // keys per lane per iteration
    __shared__ float sm[4][G][HD + 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane % LPR, rg = lane / LPR;
    int bid = blockIdx.x;
    const int s = bid % p.splits; bid /= p.splits;
    const int kvh = bid % p.nkv;
    const int b = bid / p.nkv;
    const int lo = p.lo ? p.lo[b] : 0, hi = p.hi[b];
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
#pragma unroll
    for (int g = 0; g < G; ++g) {
        qv[g] = *reinterpret_cast<const u32x4*>(p.q + (size_t)b * p.ldq + (kvh * G + g) * HD + sub * 8);
        m[g] = NEG_BIG; l[g] = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[g][e] = f32x2{0.f, 0.f};
    }

    for (int key = k0 + wave * RPW + rg; key < k1; key += STEP * U) {
        u32x4 kv[U], vv[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ku = key + u * STEP;
            ok[u] = ku < k1;
            const int kc_ = ok[u] ? ku : key;                     // clamp: a valid row, masked below
            kv[u] = *reinterpret_cast<const u32x4*>(kb + (size_t)kc_ * ldc);
            vv[u] = *reinterpret_cast<const u32x4*>(vb + (size_t)kc_ * ldc);
        }
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
    }

    // merge the RPW row groups of the wave (butterfly over the lane bits above the row), then the 4 waves through LDS
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
    for (int t = tid; t < G * HD; t += 256) {
        const int g = t / HD, d = t % HD;
        float mn = NEG_BIG;
#pragma unroll
        for (int w = 0; w < 4; ++w) mn = fmaxf(mn, sm[w][g][HD]);
        float a = 0.f, ls = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
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

template <int HD, int G>
void launch(hipStream_t st, const DecodeArgs& p, int B) {
    hipLaunchKernelGGL((attn_decode_kernel<HD, G>), dim3(B * p.nkv * p.splits), dim3(256), 0, st, p);
    if (p.splits > 1)
        hipLaunchKernelGGL(attn_decode_merge_kernel<HD>, dim3(B * p.nh), dim3(HD), 0, st, p.part, p.out, p.splits);
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
    const int G = n_heads / n_kv_heads;
    DecodeArgs p;
    p.q = (const bf16_t*)q; p.kc = (const bf16_t*)kcache; p.vc = (const bf16_t*)vcache; p.out = (bf16_t*)out;
    p.lo = kv_lo; p.hi = kv_hi; p.part = workspace;
    p.Tmax = Tmax; p.nh = n_heads; p.nkv = n_kv_heads; p.ldq = ldq;
    p.qscale = scale * 1.44269504088896341f;
    // enough blocks for ~4 per CU, but keep >= 256 keys per split (kv_len_hint = upper bound of the valid length, 0 = Tmax)
    const int len = kv_len_hint > 0 ? kv_len_hint : Tmax;
    static const int target = [] { const char* e = getenv("MOLLY_DECODE_BLOCKS"); return e ? atoi(e) : 1024; }();
    static const int min_keys = [] { const char* e = getenv("MOLLY_DECODE_MIN_KEYS"); return e ? atoi(e) : 256; }();
    int splits = (target + B * n_kv_heads - 1) / (B * n_kv_heads);
    splits = splits < 1 ? 1 : splits;
    if (splits > len / min_keys) splits = len / min_keys > 0 ? len / min_keys : 1;
    if (splits > MAX_SPLITS) splits = MAX_SPLITS;
    if (splits > 1 && (workspace == nullptr || workspace_floats < (long)B * n_heads * splits * (head_dim + 2))) splits = 1;
    p.splits = splits;
    hipStream_t st = (hipStream_t)stream;
#define MOLLY_DEC(HD_, G_) else if (head_dim == HD_ && G == G_) launch<HD_, G_>(st, p, B)
    if (false) {}
    MOLLY_DEC(128, 1); MOLLY_DEC(128, 2); MOLLY_DEC(128, 4); MOLLY_DEC(128, 8);
    MOLLY_DEC(64, 1); MOLLY_DEC(64, 2); MOLLY_DEC(64, 4); MOLLY_DEC(64, 8);
    else {
        molly_set_error("attn_decode: n_heads / n_kv_heads = %d (built for 1, 2, 4, 8)", G);
        return 1;
    }
#undef MOLLY_DEC
    MOLLY_LAUNCH_CHECK();
    return 0;
}
