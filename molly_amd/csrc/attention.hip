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
#include "common.h"
#include "molly_hip.h"

namespace {

constexpr int BQ = 128;   // query rows per block (4 waves x 32)
constexpr int BKV = 64;   // keys per tile
constexpr float LOG2E = 1.4426950408889634f;

struct AttnArgs {
    const bf16_t* Q; const bf16_t* K; const bf16_t* V; bf16_t* O; float* LSE;
    const int* kv_lo; const int* kv_hi;      // per-batch valid key range (nullable = [0,T))
    int T, nh, nkv, ldq, ldk, ldv, ldo;
    float scale_log2;                        // softmax scale * log2(e)
    int causal;
};

// chunk swizzles (16-byte chunks inside one row of HD bf16): see DESIGN.md "attention LDS images"
template <int HD> __device__ __forceinline__ int k_swz(int row, int ch) {
    return HD == 128 ? (ch ^ (row & 15)) : (ch ^ ((row >> 1) & 7));
}
template <int HD> __device__ __forceinline__ int v_swz(int row, int ch) {
    return HD == 128 ? (ch ^ ((row & 3) << 2)) : (ch ^ (((row >> 1) & 1) << 2));
}

// stage a [64 keys][HD] tile with global_load_lds; LDS image is lane-linear, swizzle applied to the SOURCE chunk.
template <int HD, bool IS_V>
__device__ __forceinline__ void stage_kv(const bf16_t* __restrict__ g, int ld, int key0, int T, bf16_t* lds, int wave,
                                         int lane) {
    constexpr int CPR = HD / 8;                  // chunks per row
    constexpr int RPI = 64 / CPR;                // rows per wave-instruction (1 KiB)
    constexpr int NINST = BKV / RPI;             // instructions per tile
#pragma unroll
    for (int i = 0; i < NINST / 4; ++i) {
        const int inst = wave * (NINST / 4) + i;
        const int row = inst * RPI + lane / CPR;
        const int cpos = lane % CPR;
        const int csrc = IS_V ? v_swz<HD>(row, cpos) : k_swz<HD>(row, cpos);
        int key = key0 + row;
        key = key < T ? key : T - 1;             // clamp; out-of-range keys are masked by index
        const bf16_t* src = g + (size_t)key * ld + csrc * 8;
        __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds + inst * 512), 16, 0, 0);
    }
}

template <int HD>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);        // [2 stages][K tile | V tile]
    constexpr int TILE = BKV * HD;
    constexpr int NS = HD / 16;      // k-steps of the S product
    constexpr int ND = HD / 32;      // 32-wide d tiles of O^T

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int nqb = gridDim.x;
    const int qb = nqb - 1 - blockIdx.x;                       // heaviest (latest) causal blocks first
    const int head = blockIdx.y, b = blockIdx.z;
    const int kvh = head / (p.nh / p.nkv);
    const int q0 = qb * BQ + wave * 32;                        // this wave's first query row
    const int T = p.T;
    const int lo = p.kv_lo ? p.kv_lo[b] : 0;
    const int hi = p.kv_hi ? p.kv_hi[b] : T;

    const bf16_t* Qb = p.Q + (size_t)b * T * p.ldq + head * HD;
    const bf16_t* Kb = p.K + (size_t)b * T * p.ldk + kvh * HD;
    const bf16_t* Vb = p.V + (size_t)b * T * p.ldv + kvh * HD;

    // Q fragments (B operand of S^T = K Q^T): lane (r,h) holds Q[q0+r][16s + 8h .. +7]
    bf16x8 qf[NS];
    {
        int qrow = q0 + r;
        qrow = qrow < T ? qrow : T - 1;
        const bf16_t* qp = Qb + (size_t)qrow * p.ldq + 8 * h;
#pragma unroll
        for (int s = 0; s < NS; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
    }

    f32x16 o[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // key tiles this BLOCK needs: [first_tile, last_tile]
    const int blk_q_last = min(qb * BQ + BQ - 1, T - 1);
    int kv_end = p.causal ? min(blk_q_last + 1, hi) : hi;       // exclusive
    int kv_begin = lo;
    const int t_first = kv_begin / BKV;
    const int t_last = kv_end > kv_begin ? (kv_end - 1) / BKV : t_first - 1;

    if (t_last >= t_first) {
        stage_kv<HD, false>(Kb, p.ldk, t_first * BKV, T, smem, wave, lane);
        stage_kv<HD, true>(Vb, p.ldv, t_first * BKV, T, smem + TILE, wave, lane);
    }
    __syncthreads();

    int cur = 0;
    for (int t = t_first; t <= t_last; ++t) {
        const bf16_t* sK = smem + cur * 2 * TILE;
        const bf16_t* sV = sK + TILE;
        if (t + 1 <= t_last) {
            bf16_t* nK = smem + (cur ^ 1) * 2 * TILE;
            stage_kv<HD, false>(Kb, p.ldk, (t + 1) * BKV, T, nK, wave, lane);
            stage_kv<HD, true>(Vb, p.ldv, (t + 1) * BKV, T, nK + TILE, wave, lane);
        }
        const int k0 = t * BKV;
        // wave-level skip: tile entirely above this wave's causal diagonal
        const bool skip = p.causal && (k0 > q0 + 31);
        if (!skip) {
            // ---- S^T = K Q^T : two 32x32 accumulators (keys k0..k0+31, k0+32..k0+63)
            f32x16 s0, s1;
#pragma unroll
            for (int e = 0; e < 16; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int ch = 2 * s + h;
                const bf16x8 ka = *reinterpret_cast<const bf16x8*>(sK + r * HD + k_swz<HD>(r, ch) * 8);
                const bf16x8 kb = *reinterpret_cast<const bf16x8*>(sK + (r + 32) * HD + k_swz<HD>(r + 32, ch) * 8);
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[s], s0, 0, 0, 0);
                s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb, qf[s], s1, 0, 0, 0);
            }
            // ---- scale + mask.  reg e of half h is key row (e&3) + 8*(e>>2) + 4h
            const int qi = q0 + r;
            const bool need_mask = (p.causal && (k0 + BKV - 1 > q0)) || (k0 < lo) || (k0 + BKV > hi);
            float mx = -INFINITY;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float a = s0[e] * p.scale_log2, c = s1[e] * p.scale_log2;
                if (need_mask) {
                    const int ka_ = k0 + (e & 3) + 8 * (e >> 2) + 4 * h, kc_ = ka_ + 32;
                    const bool oka = ka_ >= lo && ka_ < hi && (!p.causal || ka_ <= qi);
                    const bool okc = kc_ >= lo && kc_ < hi && (!p.causal || kc_ <= qi);
                    a = oka ? a : -INFINITY;
                    c = okc ? c : -INFINITY;
                }
                s0[e] = a; s1[e] = c;
                mx = fmaxf(mx, fmaxf(a, c));
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
            const float alpha = exp2f(m_run - m_use);           // m_run=-inf -> 0
            float rs = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s0[e] = exp2f(s0[e] - m_use);
                s1[e] = exp2f(s1[e] - m_use);
                rs += s0[e] + s1[e];
            }
            l_run = l_run * alpha + rs;                         // per-half partial; halves merged at the end
            m_run = m_new;
#pragma unroll
            for (int d = 0; d < ND; ++d)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
            // ---- P^T -> bf16 B operands: k-step s' uses accumulator regs 8(s'&1)..+7 of s0 (s'<2) / s1
            bf16x8 pf[4];
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) {
                const f32x16& src = sp < 2 ? s0 : s1;
                const int base = 8 * (sp & 1);
                u32x4 w;
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = pack_bf2(src[base + 2 * j], src[base + 2 * j + 1]);
                pf[sp] = __builtin_bit_cast(bf16x8, w);
            }
            // ---- O^T += V^T P^T : A operand = V^T via transposed LDS reads.
            // lane group g = lane>>4: d column block 16*(g&1), key sub-block 4*(g>>1) == 4h ; lane 4q+pp of the group
            // supplies row q, columns 4pp..4pp+3 of the 4x16 block.
            const int gi = lane & 15, gq = gi >> 2, gp = gi & 3, gcol = 16 * ((lane >> 4) & 1);
#pragma unroll
            for (int sp = 0; sp < 4; ++sp) {
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    const int col = 32 * d + gcol + 4 * gp;          // element column of this lane's address
                    const int rowa = 16 * sp + 4 * h + gq;            // keys for elements 0..3
                    const int rowb = rowa + 8;                        // keys for elements 4..7
                    const bf16_t* pa = sV + rowa * HD + v_swz<HD>(rowa, col >> 3) * 8 + (col & 7);
                    const bf16_t* pb = sV + rowb * HD + v_swz<HD>(rowb, col >> 3) * 8 + (col & 7);
                    const bf16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)pa);
                    const bf16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4*)pb);
                    const bf16x8 vf = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[sp], o[d], 0, 0, 0);
                }
            }
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: O[q][d] = o / l ; LSE2 = m + log2(l)
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
    const int qi = q0 + r;
    if (qi < T) {
        bf16_t* op = p.O + ((size_t)b * T + qi) * p.ldo + head * HD;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int dd = 32 * d + 8 * g4 + 4 * h;
                *reinterpret_cast<u32x2*>(op + dd) =
                    u32x2{pack_bf2(o[d][4 * g4] * inv, o[d][4 * g4 + 1] * inv),
                          pack_bf2(o[d][4 * g4 + 2] * inv, o[d][4 * g4 + 3] * inv)};
            }
        if (p.LSE && h == 0)
            p.LSE[((size_t)b * p.nh + head) * T + qi] = l_tot > 0.f ? m_run + log2f(l_tot) : -INFINITY;
    }
}

}  // namespace

extern "C" int molly_attn_fwd(void* stream, const void* Q, const void* K, const void* V, void* O, float* lse2,
                              const int* kv_lo, const int* kv_hi, int B, int T, int n_heads, int n_kv_heads, int head_dim,
                              int ldq, int ldk, int ldv, int ldo, float scale, int causal) {
    MOLLY_CHECK(head_dim == 128 || head_dim == 64, "attn_fwd: head_dim=%d not built (64 and 128 are)", head_dim);
    MOLLY_CHECK(n_heads % n_kv_heads == 0, "attn_fwd: n_heads %% n_kv_heads != 0");
    MOLLY_CHECK(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 4 == 0, "attn_fwd: row strides must be multiples of 8");
    MOLLY_CHECK(((uintptr_t)Q % 16) == 0 && ((uintptr_t)K % 16) == 0 && ((uintptr_t)V % 16) == 0, "attn_fwd: alignment");
    MOLLY_CHECK(B > 0 && T > 0, "attn_fwd: empty problem");
    AttnArgs p{(const bf16_t*)Q, (const bf16_t*)K, (const bf16_t*)V, (bf16_t*)O, lse2, kv_lo, kv_hi, T, n_heads,
               n_kv_heads, ldq, ldk, ldv, ldo, scale * LOG2E, causal};
    dim3 grid(cdiv(T, BQ), n_heads, B);
    const size_t lds = 2 * 2 * BKV * head_dim * sizeof(bf16_t);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
        attr_set = true;
    }
    if (head_dim == 128)
        hipLaunchKernelGGL(attn_fwd_kernel<128>, grid, dim3(256), lds, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(attn_fwd_kernel<64>, grid, dim3(256), lds, (hipStream_t)stream, p);
    MOLLY_LAUNCH_CHECK();
    return 0;
}
