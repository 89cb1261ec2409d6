// Sampling logits processors + token draw of `generate(do_sample=True)` on the device.  The reference samples
// with repetition_penalty 1.1, temperature 0.8, top_k 20, top_p 0.95 (src/inference_lora.py:293-298,
// scripts/infer/inference_nt_lora.sh:28-30) through HF generate's processors, in HF's order
// (HF:generation/logits_process.py RepetitionPenaltyLogitsProcessor -> TemperatureLogitsWarper -> TopKLogitsWarper ->
// TopPLogitsWarper, then softmax + torch.multinomial, HF:generation/utils.py _sample):
//   1. repetition penalty on every token generated so far:  s = s < 0 ? s * pen : s / pen   (once per distinct token)
//   2. s /= temperature
//   3. top-k: everything below the k-th largest score -> -inf   (ties with the k-th stay, as `scores < kth` keeps them)
//   4. top-p: sort ascending, softmax, cumsum; drop tokens whose cumulative mass is <= 1 - top_p; the largest always stays
//   5. softmax over what is left, one draw.
// One block per row.  The k-th largest of V = 151936 scores comes from a 3-pass radix select on the order-preserving integer
// image of the floats (11 + 11 + 10 bits, LDS histograms); the <= 1024 survivors are sorted in LDS (bitonic) and finished by
// one thread (k = 20 in the reference's script).  The draw uses Philox4x32-10 keyed by (seed, step, row): the DISTRIBUTION is
// HF's, the random stream is not torch's (tests compare the distribution; the survivors and their probabilities can be
// written out for that).  The logits row is modified in place by step 1 (it is scratch: the next decode step overwrites it).
#include "common.h"
#include "molly_hip.h"

namespace {

constexpr int CAP = 1024;
constexpr int MAXV = 262144;                                   // vocabulary bound of the penalty's "seen" bitmap (32 KiB of LDS)

__device__ __forceinline__ unsigned fkey(float v) {            // larger float -> larger unsigned
    const unsigned u = __builtin_bit_cast(unsigned, v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(1024) void sample_kernel(float* __restrict__ logits, int V, int ld, const long* __restrict__ gen,
                                                      int n_gen, int ld_gen, float pen, float temperature, int top_k,
                                                      float top_p, unsigned seed_lo, unsigned seed_hi, unsigned step,
                                                      long* __restrict__ next_token, float* __restrict__ probs_out,
                                                      long* __restrict__ ids_out, int* __restrict__ n_out, int cap_out) {
    __shared__ unsigned hist[2048];
    __shared__ float s_val[CAP];
    __shared__ int s_idx[CAP];
    __shared__ unsigned s_prefix, s_need;
    __shared__ int s_cnt;
    __shared__ unsigned seen[MAXV / 32];
    __shared__ int s_wave[16];
    const int row = blockIdx.x, t = threadIdx.x;
    float* x = logits + (size_t)row * ld;

    // ---- 1. repetition penalty, once per DISTINCT generated token whatever their number: the thread that sets a token's bit in
    // the `seen` bitmap owns that token and is the only one to touch its score (HF gathers, rescales and scatters the same
    // value for every duplicate)
    if (pen != 1.0f && n_gen > 0) {
        const long* g = gen + (size_t)row * ld_gen;
        for (int i = t; i < (V + 31) / 32; i += 1024) seen[i] = 0u;
        __syncthreads();
        for (int i = t; i < n_gen; i += 1024) {
            const int id = (int)g[i];
            if (id < 0 || id >= V) continue;
            const unsigned bit = 1u << (id & 31);
            if (!(atomicOr(&seen[id >> 5], bit) & bit)) {
                const float v = x[id];
                x[id] = v < 0.f ? v * pen : v / pen;
            }
        }
        __syncthreads();
    }

    const int k = min(max(top_k, 1), V);
    // ---- 3 (fast form).  The k-th largest of the 1,024 threads' own maxima is a LOWER bound of the row's k-th largest score (they are 1,024 distinct
    // elements), so every survivor of top-k is among the scores >= that bound: one scan for the maxima, a sort of 1,024 values in LDS, one scan that
    // collects the few scores above the bound (k = 20: a few dozen of 151,936), a sort of those — and the k-th of them IS the row's k-th, ties and
    // all.  Two contention-free scans instead of the radix select's three histogram passes + one (LDS atomics of 1,024 threads into the few hot bins
    // of a score distribution: 271 us per step at 32 x 151,936, round 4); more than CAP candidates (a row of equal scores): the radix select below.
    bool have = false;
    {
        const bool vec = (V & 3) == 0 && (ld & 3) == 0 && ((uintptr_t)logits & 15) == 0;
        float mx = -INFINITY;
        if (vec) {
            for (int i = t * 4; i < V; i += 4096) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(x + i);
                mx = fmaxf(fmaxf(mx, fmaxf(q[0], q[1])), fmaxf(q[2], q[3]));
            }
        } else {
            for (int i = t; i < V; i += 1024) mx = fmaxf(mx, x[i]);
        }
        s_val[t] = mx;
        s_idx[t] = t;
        __syncthreads();
        for (int size = 2; size <= CAP; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                const int j = t ^ stride;
                if (j > t) {
                    const bool desc = (t & size) == 0;
                    const float a = s_val[t], b = s_val[j];
                    if (desc != (a > b || (a == b && t < j))) { s_val[t] = b; s_val[j] = a; }
                }
                __syncthreads();
            }
        const float bound = s_val[min(k, CAP) - 1];
        __syncthreads();
        if (t == 0) s_cnt = 0;
        s_val[t] = -INFINITY;
        s_idx[t] = 0x7fffffff;
        __syncthreads();
        auto take = [&](float v, int i) {
            if (v >= bound) {
                const int p = atomicAdd(&s_cnt, 1);
                if (p < CAP) { s_val[p] = v; s_idx[p] = i; }
            }
        };
        if (vec) {
            for (int i = t * 4; i < V; i += 4096) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(x + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) take(q[e], i + e);
            }
        } else {
            for (int i = t; i < V; i += 1024) take(x[i], i);
        }
        __syncthreads();
        have = s_cnt <= CAP && bound > -INFINITY;                    // block-uniform
    }
    if (have) {
        // sort the candidates descending (ties: lower token id first); the survivors are the prefix >= the k-th value
        for (int size = 2; size <= CAP; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                const int j = t ^ stride;
                if (j > t) {
                    const bool desc = (t & size) == 0;
                    const float a = s_val[t], b = s_val[j];
                    const int ia = s_idx[t], ib = s_idx[j];
                    const bool a_first = a > b || (a == b && ia < ib);
                    if (desc != a_first) { s_val[t] = b; s_val[j] = a; s_idx[t] = ib; s_idx[j] = ia; }
                }
                __syncthreads();
            }
        const int nc = s_cnt;
        const float kth = s_val[min(k, nc) - 1];
        __syncthreads();
        if (t < nc && s_val[t] >= kth && (t + 1 == nc || s_val[t + 1] < kth)) s_cnt = t + 1;
        __syncthreads();
    }
    // ---- 3a. k-th largest by radix select (temperature > 0 is monotonic: select on the unscaled scores)
    if (!have) {
    if (t == 0) { s_prefix = 0; s_need = (unsigned)k; }
    __syncthreads();
    const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
        for (int i = t; i < 2048; i += 1024) hist[i] = 0;
        __syncthreads();
        const unsigned prefix = s_prefix;
        const int sh = shifts[pass], nb = bits[pass];
        const int hi_sh = sh + nb;                                   // bits above this pass's field
        for (int i = t; i < V; i += 1024) {
            const unsigned key = fkey(x[i]);
            if (pass == 0 || (key >> hi_sh) == prefix) atomicAdd(&hist[(key >> sh) & ((1u << nb) - 1)], 1u);
        }
        __syncthreads();
        if (t == 0) {
            unsigned need = s_need, cum = 0;
            int b = (1 << nb) - 1;
            for (; b > 0; --b) {
                if (cum + hist[b] >= need) break;
                cum += hist[b];
            }
            s_need = need - cum;
            s_prefix = (prefix << nb) | (unsigned)b;
        }
        __syncthreads();
    }
    const unsigned thr = s_prefix;                                   // key of the k-th largest score

    // ---- 3b. survivors (score >= k-th) into LDS, then sorted descending (ties: lower token id first)
    if (t == 0) s_cnt = 0;
    s_val[t] = -INFINITY;
    s_idx[t] = 0x7fffffff;
    __syncthreads();
    for (int i = t; i < V; i += 1024) {
        const float v = x[i];
        if (fkey(v) >= thr) {
            const int p = atomicAdd(&s_cnt, 1);
            if (p < CAP) { s_val[p] = v; s_idx[p] = i; }
        }
    }
    __syncthreads();
    if (s_cnt > CAP) {
        // more than CAP scores tie with the k-th largest (block-uniform, rare): the atomic order above would decide which ties
        // survive.  Redo it deterministically: every score strictly above the threshold (fewer than k <= CAP of them), then the
        // ties by ascending token id until the list is full.
        __syncthreads();
        if (t == 0) s_cnt = 0;
        s_val[t] = -INFINITY;
        s_idx[t] = 0x7fffffff;
        __syncthreads();
        for (int i = t; i < V; i += 1024) {
            const float v = x[i];
            if (fkey(v) > thr) {
                const int p = atomicAdd(&s_cnt, 1);
                s_val[p] = v; s_idx[p] = i;
            }
        }
        __syncthreads();
        int filled = s_cnt;                                          // block-uniform from here on
        for (int i0 = 0; i0 < V && filled < CAP; i0 += 1024) {
            const int i = i0 + t;
            const float v = i < V ? x[i] : 0.f;
            const bool tie = i < V && fkey(v) == thr;
            const unsigned long long bal = __ballot(tie);
            const int lane = t & 63, w = t >> 6;
            if (lane == 0) s_wave[w] = __popcll(bal);
            __syncthreads();
            int before = 0, total = 0;
            for (int j = 0; j < 16; ++j) { if (j < w) before += s_wave[j]; total += s_wave[j]; }
            const int p = filled + before + __popcll(bal & ((1ull << lane) - 1ull));
            if (tie && p < CAP) { s_val[p] = v; s_idx[p] = i; }
            filled = min(filled + total, CAP);
            __syncthreads();
        }
        if (t == 0) s_cnt = filled;
        __syncthreads();
    }
    }   // (!have)
    const int n = min(s_cnt, CAP);
    for (int size = 2; size <= (have ? 1 : CAP); size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int j = t ^ stride;
            if (j > t) {
                const bool desc = (t & size) == 0;
                const float a = s_val[t], b = s_val[j];
                const int ia = s_idx[t], ib = s_idx[j];
                const bool a_first = a > b || (a == b && ia < ib);   // a belongs before b in descending order
                if (desc != a_first) { s_val[t] = b; s_val[j] = a; s_idx[t] = ib; s_idx[j] = ia; }
            }
            __syncthreads();
        }

    // ---- 2, 4, 5: temperature, top-p, softmax, draw — one thread over the n sorted survivors
    if (t == 0) {
        const float T = temperature > 0.f ? temperature : 1.f;
        const float mx = s_val[0] / T;
        float sum = 0.f;
        for (int j = n - 1; j >= 0; --j) { s_val[j] = __expf(s_val[j] / T - mx); sum += s_val[j]; }
        int n_keep = n;
        if (top_p < 1.f) {
            float cum = 0.f;                                         // ascending cumulative mass, as HF computes it
            for (int j = n - 1; j >= 1; --j) {
                cum += s_val[j] / sum;
                if (cum <= 1.f - top_p) n_keep = j; else break;
            }
        }
        float ksum = 0.f;
        for (int j = 0; j < n_keep; ++j) ksum += s_val[j];
        unsigned rnd[4];
        philox4x32_10((unsigned)row, step, 0u, 0u, seed_lo, seed_hi, rnd);
        const float u = (float)(rnd[0] >> 8) * (1.0f / 16777216.0f) * ksum;
        float acc = 0.f;
        int pick = n_keep - 1;
        for (int j = 0; j < n_keep; ++j) {
            acc += s_val[j];
            if (u < acc) { pick = j; break; }
        }
        next_token[row] = (long)s_idx[pick];
        if (n_out) n_out[row] = n_keep;
        if (probs_out)
            for (int j = 0; j < min(n_keep, cap_out); ++j) {
                probs_out[(size_t)row * cap_out + j] = s_val[j] / ksum;
                ids_out[(size_t)row * cap_out + j] = (long)s_idx[j];
            }
    }
}

}  // namespace

extern "C" int molly_sample_logits(void* stream, float* logits, int rows, int V, int ld, const int64_t* generated, int n_generated,
                                   int ld_generated, float repetition_penalty, float temperature, int top_k, float top_p,
                                   uint64_t seed, int step, int64_t* next_token, float* probs_out_or_null,
                                   int64_t* ids_out_or_null, int* n_out_or_null, int cap_out) {
    MOLLY_ENTER();
    MOLLY_CHECK(rows > 0 && V > 0 && ld >= V, "sample_logits: rows=%d V=%d ld=%d", rows, V, ld);
    MOLLY_CHECK(top_k >= 1 && top_k <= CAP, "sample_logits: top_k=%d outside 1..%d (no-top-k sampling is not built)", top_k, CAP);
    MOLLY_CHECK(temperature > 0.f && top_p > 0.f && repetition_penalty > 0.f, "sample_logits: temperature / top_p / penalty must be > 0");
    MOLLY_CHECK(V <= MAXV, "sample_logits: V=%d exceeds the %d-token bitmap of the repetition penalty", V, MAXV);
    MOLLY_CHECK(n_generated >= 0, "sample_logits: n_generated=%d", n_generated);
    MOLLY_CHECK(n_generated == 0 || generated, "sample_logits: %d generated tokens without their ids", n_generated);
    MOLLY_CHECK((probs_out_or_null == nullptr) == (ids_out_or_null == nullptr), "sample_logits: probs_out and ids_out go together");
    hipLaunchKernelGGL(sample_kernel, dim3(rows), dim3(1024), 0, (hipStream_t)stream, logits, V, ld, (const long*)generated,
                       n_generated, ld_generated, repetition_penalty, temperature, top_k, top_p, (unsigned)seed,
                       (unsigned)(seed >> 32), (unsigned)step, (long*)next_token, probs_out_or_null, (long*)ids_out_or_null,
                       n_out_or_null, cap_out);
    MOLLY_LAUNCH_CHECK();
    return 0;
}
