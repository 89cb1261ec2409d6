// Batch assembly on the device.  Replaces the host-side work the reference does per step around the model
// call — reference src/model/omics_one.py:69-72 (stack + mask + a device->host sync assert), :93-97 (one slice-copy launch per
// span), :104-118 (per-row .to(device)) — and the host argsort / index uploads this library itself used in round 1:
//   ONE packed int32 image (token ids, labels, key ranges, span table (b, start, group, row), omic ids) arrives through ONE
//   pinned host->device copy; these kernels then build everything the step needs, on the stream, with no host round trip:
//     * shifted labels (HF:loss/loss_utils.py:60-63: pad one ignore column, drop the first) and the ordered list of scored rows
//     * per modality group: int64 encoder ids and the scatter index  dst[row*K + j] = b*T + start + 1 + j  (j < k, else -1)
//     * the overwritten-row mask, and from it the sorted embedding-gradient index (rows grouped by token id, overwritten rows
//       dropped: the reference overwrites them in place, omics_one.py:97, so their embedding rows get no gradient):
//       order[], seg_start[], uid[], n_unique — stable LSD radix sort, so the summation order inside a segment is the row
//       order: bitwise reproducible (no atomics).
// The radix sort is rocPRIM's device primitive through hipcub (index plumbing, not a hot op); the rest is written here.
#include <hipcub/hipcub.hpp>

#include "common.h"
#include "molly_hip.h"

namespace {

// ---- single-block ordered compaction: emit(k, i) for the k-th index i in [0, n) with pred(i); returns the count in every
// thread.  1024 threads, thread t owns the contiguous chunk [t*per, (t+1)*per) -> output order = index order.
template <class Pred, class Emit>
__device__ __forceinline__ int block_compact(int n, Pred pred, Emit emit, int* s_wave /* 17 ints of LDS */) {
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int per = (n + 1023) / 1024;
    const int i0 = min(t * per, n), i1 = min(i0 + per, n);
    int cnt = 0;
    for (int i = i0; i < i1; ++i) cnt += pred(i) ? 1 : 0;
    int incl = cnt;                                                   // wave inclusive scan
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    if (lane == 63) s_wave[w] = incl;
    __syncthreads();
    if (t == 0) {
        int run = 0;
        for (int i = 0; i < 16; ++i) { const int v = s_wave[i]; s_wave[i] = run; run += v; }
        s_wave[16] = run;
    }
    __syncthreads();
    int k = s_wave[w] + incl - cnt;
    for (int i = i0; i < i1; ++i)
        if (pred(i)) emit(k++, i);
    return s_wave[16];
}

__global__ __launch_bounds__(256) void assemble_elementwise_kernel(
    const int* __restrict__ ids, const int* __restrict__ labels, int M, int T, int ignore_index,
    long* __restrict__ labels_shifted, unsigned* __restrict__ keys, int* __restrict__ vals, unsigned char* __restrict__ overwritten,
    const int* __restrict__ om0, long* __restrict__ om0_64, int n0, const int* __restrict__ om1, long* __restrict__ om1_64, int n1) {
    const long total = (long)M + n0 + n1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        if (i < M) {
            if (labels_shifted) labels_shifted[i] = ((int)(i % T) == T - 1) ? (long)ignore_index : (long)labels[i + 1];
            if (keys) { keys[i] = (unsigned)ids[i]; vals[i] = (int)i; }
            overwritten[i] = 0;
        } else if (i < M + n0) {
            om0_64[i - M] = om0[i - M];
        } else {
            om1_64[i - M - n0] = om1[i - M - n0];
        }
    }
}

// spans[s] = (b, start, group, row in the group's encoder batch).  start == -1: the row is encoded but never scattered.
__global__ __launch_bounds__(256) void spans_kernel(const int* __restrict__ spans, int n_spans, int T, int K0, int k0, int K1,
                                                    int k1, int* __restrict__ dst0, int* __restrict__ dst1,
                                                    unsigned* __restrict__ keys, unsigned sentinel,
                                                    unsigned char* __restrict__ overwritten) {
    const int s = blockIdx.x;
    if (s >= n_spans) return;
    const int b = spans[4 * s], start = spans[4 * s + 1], grp = spans[4 * s + 2], row = spans[4 * s + 3];
    const int K = grp ? K1 : K0, k = grp ? k1 : k0;
    int* dst = (grp ? dst1 : dst0) + (size_t)row * K;
    for (int j = threadIdx.x; j < K; j += 256) {
        const bool live = start >= 0 && j < k;
        const int d = b * T + start + 1 + j;
        dst[j] = live ? d : -1;
        if (live) {
            overwritten[d] = 1;
            if (keys) keys[d] = sentinel;
        }
    }
}

__global__ __launch_bounds__(1024) void compact_scored_kernel(const long* __restrict__ labels_shifted, int M, int ignore_index,
                                                              int* __restrict__ scored_rows, int* __restrict__ n_scored) {
    __shared__ int s_wave[17];
    const int n = block_compact(M, [&](int i) { return labels_shifted[i] != (long)ignore_index; },
                                [&](int k, int i) { scored_rows[k] = i; }, s_wave);
    if (threadIdx.x == 0) *n_scored = n;
}

__global__ __launch_bounds__(1024) void segments_kernel(const unsigned* __restrict__ ks, int M, unsigned sentinel,
                                                        int* __restrict__ seg_start, long* __restrict__ uid,
                                                        int* __restrict__ n_unique) {
    __shared__ int s_wave[17];
    __shared__ int s_valid;
    if (threadIdx.x == 0) s_valid = M;
    __syncthreads();
    const int n = block_compact(M, [&](int i) { return ks[i] != sentinel && (i == 0 || ks[i] != ks[i - 1]); },
                                [&](int k, int i) { seg_start[k] = i; uid[k] = (long)ks[i]; }, s_wave);
    // first sentinel position = number of live rows (sorted: sentinels are last)
    const int per = (M + 1023) / 1024;
    const int i0 = min((int)threadIdx.x * per, M), i1 = min(i0 + per, M);
    for (int i = i0; i < i1; ++i)
        if (ks[i] == sentinel && (i == 0 || ks[i - 1] != sentinel)) s_valid = i;
    __syncthreads();
    if (threadIdx.x == 0) { *n_unique = n; seg_start[n] = s_valid; }
}

inline int key_bits(int vocab) {            // bits to sort ids in [0, vocab] (vocab itself is the sentinel)
    int b = 1;
    while ((1L << b) <= (long)vocab) ++b;
    return b;
}

}  // namespace

extern "C" int molly_batch_sort_workspace(int M) {
    MOLLY_ENTER();
    size_t bytes = 0;
#if defined(MOLLY_HOST_DRY)      // (host-sanitizer build: the library's size query and sort talk to a runtime — not run there)
    bytes = (size_t)M * 16 + 4096;
#else
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const unsigned*)nullptr, (unsigned*)nullptr, (const int*)nullptr,
                                             (int*)nullptr, M, 0, 32, (hipStream_t)0);
#endif
    return (int)bytes;
}

extern "C" int molly_batch_assemble(void* stream, const int* ids32, const int* labels32, int B, int T, int vocab, int ignore_index,
                                    const int* spans, int n_spans, const int* omic32_g0, int n_rows0, int K0, int k0,
                                    const int* omic32_g1, int n_rows1, int K1, int k1, int64_t* labels_shifted,
                                    int* scored_rows, int* n_scored, int64_t* omic64_g0, int* dst_g0, int64_t* omic64_g1,
                                    int* dst_g1, unsigned char* overwritten, int* keys_tmp, int* vals_tmp, int* order,
                                    int* seg_start, int64_t* uid, int* n_unique, void* sort_ws, long sort_ws_bytes) {
    MOLLY_ENTER();
    const long M = (long)B * T;
    MOLLY_CHECK(M > 0 && M < (1L << 30) && overwritten, "batch_assemble: B*T = %ld rows", M);
    MOLLY_CHECK((labels32 != nullptr) == (labels_shifted != nullptr) && (!labels_shifted || (scored_rows && n_scored)),
                "batch_assemble: labels in and shifted labels / scored rows out go together");
    MOLLY_CHECK(n_spans == 0 || spans, "batch_assemble: %d spans without a table", n_spans);
    MOLLY_CHECK((n_rows0 == 0 || (omic32_g0 && omic64_g0 && dst_g0 && k0 <= K0)) &&
                    (n_rows1 == 0 || (omic32_g1 && omic64_g1 && dst_g1 && k1 <= K1)),
                "batch_assemble: a modality group has rows but no buffers (or k > K)");
    const bool sort = order != nullptr;
    MOLLY_CHECK(!sort || (ids32 && keys_tmp && vals_tmp && seg_start && uid && n_unique && sort_ws),
                "batch_assemble: the embedding-gradient index needs ids, key/value scratch, outputs and a sort workspace");
    hipStream_t st = (hipStream_t)stream;
    unsigned* keys_in = (unsigned*)keys_tmp;               // [M] unsorted, [M] sorted behind it
    unsigned* keys_out = keys_in + M;
    const long n0 = (long)n_rows0 * K0, n1 = (long)n_rows1 * K1;
    const long total = M + n0 + n1;
    hipLaunchKernelGGL(assemble_elementwise_kernel, dim3((unsigned)min((total + 255) / 256, 4096L)), dim3(256), 0, st, ids32,
                       labels32, (int)M, T, ignore_index, (long*)labels_shifted, sort ? keys_in : nullptr, vals_tmp, overwritten,
                       omic32_g0, (long*)omic64_g0, (int)n0, omic32_g1, (long*)omic64_g1, (int)n1);
    if (n_spans > 0)
        hipLaunchKernelGGL(spans_kernel, dim3(n_spans), dim3(256), 0, st, spans, n_spans, T, K0, k0, K1, k1, dst_g0, dst_g1,
                           sort ? keys_in : nullptr, (unsigned)vocab, overwritten);
    if (labels_shifted)
        hipLaunchKernelGGL(compact_scored_kernel, dim3(1), dim3(1024), 0, st, (const long*)labels_shifted, (int)M, ignore_index,
                           scored_rows, n_scored);
    if (sort) {
#if !defined(MOLLY_HOST_DRY)
        size_t bytes = (size_t)sort_ws_bytes;
        hipError_t e = hipcub::DeviceRadixSort::SortPairs(sort_ws, bytes, (const unsigned*)keys_in, keys_out, (const int*)vals_tmp,
                                                          order, (int)M, 0, key_bits(vocab), st);
        MOLLY_CHECK(e == hipSuccess, "batch_assemble: radix sort failed: %s", hipGetErrorString(e));
#else
        (void)keys_out; (void)key_bits(vocab);
#endif
        hipLaunchKernelGGL(segments_kernel, dim3(1), dim3(1024), 0, st, (const unsigned*)keys_out, (int)M, (unsigned)vocab,
                           seg_start, (long*)uid, n_unique);
    }
    MOLLY_LAUNCH_CHECK();
    return 0;
}
