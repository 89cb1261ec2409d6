// Host-sanitizer driver (CPU only; never on the GPU box).  Linked against the MOLLY_HOST_DRY build of every csrc translation unit
// (python -m molly_amd.build --host-asan): the HOST half of libmolly_hip — argument validation, launch_cfg's cost model, split-K /
// stream-K range arithmetic, grid and LDS sizing of every entry point — compiled with -fsanitize=address,undefined, launches
// recorded instead of issued (common.h).  This program walks the argument space those functions see:
//   * the GEMM shape generator of tools/gemm_diag/fuzz_gemm.py (forms, epilogue flags, ragged sizes, the step's real shapes, the
//     B = 1 shapes of the 8-GPU configs, decode rows) through every GEMM entry point and every launch knob of a context,
//   * the attention argument space of tools/fuzz_attn.py (head dims, GQA groups, ragged T, strides) incl. the head-split backward,
//   * the elementwise / optimizer / batch entry points at empty, tiny, ragged and 2^31-adjacent sizes.
// Device pointers are fake (never dereferenced by host code: a dereference is exactly what ASan would report).  Exit code 0 and the
// line "host-asan: ok" = no sanitizer report, no launch outside gfx950's limits, every rejected call left an error text.
// Reference roles of what is validated here: cuBLAS behind every nn.Linear (HF:models/qwen3/modeling_qwen3.py:76-83), flash-attn
// (reference src/train.py:578-582), DeepSpeed's step (src/configs/ds_z2_config.json:18-27).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "molly_hip.h"

extern "C" long molly_dry_launch_count(void);
extern "C" const char* molly_dry_last_launch(void);

namespace {
struct Rng {            // xorshift64*: the driver's shapes are a pure function of the seed
    uint64_t s;
    uint64_t next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 2685821237ULL * 1000003ULL + 0x9E3779B97F4A7C15ULL; }
    int range(int lo, int hi) { return lo + (int)(next() % (uint64_t)(hi - lo + 1)); }
    template <class T, size_t N> T pick(const T (&a)[N]) { return a[next() % N]; }
};
void* fake(uint64_t i) { return (void*)(uintptr_t)(0x7f0000000000ULL + i * 0x40000000ULL); }     // 1 GiB apart, 256-byte aligned
long calls = 0, accepted = 0, rejected = 0, silent = 0;
struct Stat { const char* what; long calls, ok; };
std::vector<Stat> stats;
void seen(int rc, const char* what) {
    ++calls;
    Stat* st = nullptr;
    for (Stat& s : stats)
        if (!strcmp(s.what, what)) st = &s;
    if (!st) { stats.push_back(Stat{what, 0, 0}); st = &stats.back(); }
    ++st->calls;
    if (rc == 0) { ++accepted; ++st->ok; return; }
    ++rejected;
    const char* e = molly_last_error();
    if (!e || !*e) { ++silent; fprintf(stderr, "rejected without an error text: %s (rc %d)\n", what, rc); }
}
}  // namespace

int main(int argc, char** argv) {
    const int n_gemm = argc > 1 ? atoi(argv[1]) : 6000, n_attn = argc > 2 ? atoi(argv[2]) : 3000;
    Rng r{0x243F6A8885A308D3ULL};
    void *A = fake(1), *B = fake(2), *C = fake(3), *bias = fake(4), *res = fake(5), *ws = fake(6), *st = nullptr;

    // ---- GEMM: contexts with every knob, scratch of several sizes
    std::vector<void*> ctxs;
    const long ws_sizes[] = {0, 4096, 600000, 64L << 20, 1L << 30, 8L << 30};
    for (int i = 0; i < 12; ++i) {
        void* c = nullptr;
        seen(molly_gemm_ctx_create(&c), "ctx_create");
        seen(molly_gemm_ctx_set_workspace(c, i % 6 == 0 ? nullptr : ws, ws_sizes[i % 6]), "ctx_set_workspace");
        ctxs.push_back(c);
    }
    const int real_mnk[][3] = {{16384, 4096, 2048}, {16384, 2048, 2048}, {16384, 12288, 2048}, {16384, 2048, 6144}, {4096, 151936, 2048},
                               {3072, 6144, 2560}, {3072, 2560, 4096}, {3072, 19456, 2560}, {3072, 2560, 9728}, {4096, 6144, 4096},
                               {4096, 4096, 4096}, {4096, 24576, 4096}, {4096, 4096, 12288}, {4096, 3840, 1280}, {4096, 1280, 5120},
                               {512, 1280, 5120}, {1024, 3840, 1280}, {32, 6144, 4096}, {8, 4096, 4096}, {64, 24576, 4096}, {1, 151936, 4096}};
    const int sizes[] = {1, 7, 8, 16, 17, 31, 32, 63, 64, 65, 127, 128, 129, 255, 256, 257, 384, 511, 512, 640, 1000, 1024, 1280, 2048, 2560,
                         3072, 4096, 5120, 6144, 8192, 9728, 12288, 16384, 151936};
    const int flag_sets[] = {0, MOLLY_GEMM_BIAS, MOLLY_GEMM_BIAS | MOLLY_GEMM_GELU, MOLLY_GEMM_RESIDUAL, MOLLY_GEMM_ACCUMULATE, MOLLY_GEMM_OUT_F32,
                             MOLLY_GEMM_OUT_F32 | MOLLY_GEMM_ACCUMULATE, MOLLY_GEMM_TRANS_OUT, MOLLY_GEMM_TRANS_OUT | MOLLY_GEMM_OUT_F32,
                             MOLLY_GEMM_SWIGLU, MOLLY_GEMM_SWIGLU_BWD, MOLLY_GEMM_BIAS | MOLLY_GEMM_RESIDUAL, 255, 1 << 20};
    const int keys[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14};
    const long key_vals[] = {-3, -1, 0, 1, 2, 3, 4, 8, 16, 64, 128, 256, 257, 512, 1024, 3000, 8192};
    for (int it = 0; it < n_gemm; ++it) {
        void* ctx = ctxs[r.next() % ctxs.size()];
        if (it % 5 == 0) (void)molly_gemm_ctx_set(ctx, r.pick(keys), r.pick(key_vals));        // (some combinations are refused: fine)
        int M, N, K;
        if (it % 3 == 0) { const int* s = real_mnk[r.next() % (sizeof(real_mnk) / sizeof(real_mnk[0]))]; M = s[0]; N = s[1]; K = s[2]; }
        else { M = r.pick(sizes); N = r.pick(sizes); K = r.pick(sizes); if (it % 7 == 0) { M += r.range(-3, 3); N += r.range(-3, 3); K += r.range(-3, 3); } }
        int ak = (int)(r.next() & 1), bk = (int)(r.next() & 1), flags = r.pick(flag_sets);
        int pad = r.pick((const int[]){0, 0, 0, 8, 64, 1});
        if (it % 2 == 0) {                                          // half of the walk: calls the launchers ACCEPT (the cost model's side)
            M = (M + 7) & ~7; N = (N + 7) & ~7; K = (K + 7) & ~7; pad &= ~7;
            if (ak && !bk) ak = 0;
            flags = !ak && !bk ? r.pick((const int[]){0, 0, MOLLY_GEMM_BIAS, MOLLY_GEMM_BIAS | MOLLY_GEMM_GELU, MOLLY_GEMM_RESIDUAL, MOLLY_GEMM_ACCUMULATE,
                                                      MOLLY_GEMM_OUT_F32, MOLLY_GEMM_BIAS | MOLLY_GEMM_RESIDUAL})
                  : !ak ? r.pick((const int[]){0, 0, MOLLY_GEMM_ACCUMULATE, MOLLY_GEMM_OUT_F32, MOLLY_GEMM_TRANS_OUT, MOLLY_GEMM_TRANS_OUT | MOLLY_GEMM_ACCUMULATE})
                        : r.pick((const int[]){0, MOLLY_GEMM_ACCUMULATE, MOLLY_GEMM_OUT_F32});
        }
        const int lda = (ak ? M : K) + pad, ldb = (bk ? N : K) + pad, ldc = ((flags & MOLLY_GEMM_TRANS_OUT) ? M : N) + pad, ldres = N + pad;
        seen(molly_gemm_bf16_ctx(ctx, st, A, B, C, (flags & MOLLY_GEMM_BIAS) ? bias : nullptr,
                                 (flags & (MOLLY_GEMM_RESIDUAL | MOLLY_GEMM_SWIGLU | MOLLY_GEMM_SWIGLU_BWD)) ? res : nullptr, M, N, K, lda, ldb, ldc,
                                 ldres, flags, ak, bk), "gemm_bf16_ctx");
        (void)molly_gemm_ctx_get(ctx, MOLLY_GEMM_KEY_LAST_CONFIG);
        if (it % 4 == 0) seen(molly_gemm_nt_bf16(st, A, B, C, nullptr, nullptr, M, N, K, K, K, N, N, 0), "gemm_nt_bf16");
        if (it % 6 == 0) {
            for (int tail = 1; tail <= 4; ++tail) (void)molly_gemm_rows_tail_supported(ctx, M, N, K, tail);
            long slabs[2] = {0, 0};
            seen(molly_gemm_rows_slabs_bf16_ctx(ctx, st, A, B, M, N, K, K, K, slabs), "gemm_rows_slabs");
            seen(molly_gemm_rows_tail_bf16_ctx(ctx, st, A, B, nullptr, nullptr, nullptr, M, N, K, K, K, 0, 0, 0, 2, nullptr, 0.f, fake(8), N / 2),
                 "gemm_rows_tail (no gate|up output)");
            seen(molly_gemm_rows_tail_bf16_ctx(ctx, st, A, B, C, nullptr, (it & 8) ? res : nullptr, M, N, K, K, K, N, N, (it & 8) ? MOLLY_GEMM_RESIDUAL : 0,
                                               1 + it % 2, fake(7), 1e-6f, fake(8), N), "gemm_rows_tail");
            seen(molly_gemm_rows_qkv_bf16_ctx(ctx, st, A, B, nullptr, M, N, K, K, K, fake(7), fake(8), (const float*)fake(9), (const float*)fake(10),
                                              (const int*)fake(11), 1e-6f, 32, 8, 128, C, 5120, fake(12), fake(13), (const int*)fake(14), 1024),
                 "gemm_rows_qkv");
        }
        if (it % 9 == 0) {                                          // grouped launches: 1..16 problems sharing K
            molly_gemm_problem pr[16];
            const int cnt = r.range(1, 16);
            for (int i = 0; i < cnt; ++i) {
                const int m = r.pick(sizes), n = r.pick(sizes), to = (int)(r.next() & 1);
                pr[i] = molly_gemm_problem{fake(10 + i), fake(30 + i), fake(50 + i), m, n, K, n, to ? m : n, to};
            }
            seen(molly_gemm_grouped_bf16_ctx(ctx, st, pr, cnt, K, r.pick((const int[]){0, MOLLY_GEMM_ACCUMULATE, MOLLY_GEMM_OUT_F32, MOLLY_GEMM_BIAS})),
                 "gemm_grouped");
        }
    }
    (void)molly_gemm_ctx_streamk_timeouts(ctxs[1]);
    for (void* c : ctxs) seen(molly_gemm_ctx_destroy(c), "ctx_destroy");
    // the thread-default context's setters
    (void)molly_gemm_set_workspace(ws, 1L << 30);
    (void)molly_gemm_set_group_m(4); (void)molly_gemm_set_persistent_blocks(-3); (void)molly_gemm_set_min_ktiles(16); (void)molly_gemm_set_schedule(1);
    (void)molly_gemm_force_tile(512); (void)molly_gemm_set_streamk(2); (void)molly_gemm_set_small_grid_tile(512);
    seen(molly_gemm_bf16(st, A, B, C, nullptr, nullptr, 3072, 6144, 2560, 2560, 2560, 6144, 6144, 0, 0, 0), "gemm_bf16 default ctx");
    (void)molly_gemm_last_config();
    (void)molly_gemm_set_workspace(nullptr, 0);

    // ---- attention
    const int hds[] = {8, 16, 24, 32, 40, 48, 64, 128, 96, 256};
    const int Ts[] = {1, 31, 64, 127, 128, 129, 500, 512, 1000, 1024, 2048, 3072, 4096, 8192};
    for (int it = 0; it < n_attn; ++it) {
        const int hd = r.pick(hds), nkv = r.pick((const int[]){1, 2, 4, 8, 20}), group = r.pick((const int[]){1, 1, 2, 4, 8, 3});
        const int nh = nkv * group + (it % 50 == 0), Bn = r.pick((const int[]){1, 1, 2, 8, 32}), T = r.pick(Ts), causal = (int)(r.next() & 1);
        const int fused = (int)(r.next() & 1), ld = fused ? (nh + 2 * nkv) * hd : nh * hd, ldk = fused ? ld : nkv * hd;
        const int* lo = (r.next() & 1) ? (const int*)fake(20) : nullptr;
        seen(molly_attn_fwd(st, A, B, C, fake(9), (float*)fake(11), lo, lo ? (const int*)fake(21) : nullptr, Bn, T, nh, nkv, hd, ld, ldk, ldk, nh * hd,
                            0.125f, causal), "attn_fwd");
        const int need = molly_attn_bwd_workspace(Bn, T, nh, nkv, hd);
        seen(molly_attn_bwd_ws(st, A, B, C, fake(9), fake(12), (const float*)fake(11), (float*)fake(13), fake(14), fake(15), fake(16), lo,
                               lo ? (const int*)fake(21) : nullptr, Bn, T, nh, nkv, hd, ld, ldk, ldk, nh * hd, nh * hd, ld, ldk, ldk, 0.125f, causal,
                               need > 0 && (it & 1) ? (float*)fake(17) : nullptr, need), "attn_bwd_ws");
        if (it % 3 == 0) {
            const int dw = molly_attn_decode_workspace(Bn, nh, hd);
            seen(molly_attn_decode(st, A, B, C, fake(9), lo, (const int*)fake(21), Bn, T, nh, nkv, hd, nh * hd, 0.125f, it % 2 ? T : 0,
                                   (it & 4) ? (float*)fake(17) : nullptr, dw), "attn_decode");
            seen(molly_attn_decode_qkv(st, (const float*)fake(18), 1 + it % 12, (it & 8) ? fake(7) : nullptr, (it & 8) ? fake(8) : nullptr,
                                       (it & 16) ? (const float*)fake(9) : nullptr, (it & 16) ? (const float*)fake(10) : nullptr, (const int*)fake(11),
                                       1e-6f, B, C, (const int*)fake(14), fake(9), lo, (const int*)fake(21), Bn, T, nh, nkv, hd, 0.125f,
                                       it % 2 ? T : 0, (it & 4) ? (float*)fake(17) : nullptr, dw), "attn_decode_qkv");
        }
    }

    // ---- elementwise / optimizer / batch entry points at edge sizes
    const long ns[] = {0, 1, 7, 8, 64, 1000, 4096, 1L << 20, (1L << 31) - 8, 1L << 31, 3L << 30};
    const int rows_[] = {0, 1, 3, 64, 4096, 16384, 65536};
    const int Hs[] = {8, 320, 1000, 1280, 2048, 2560, 4096, 5120, 16384, 7};
    for (int rows : rows_)
        for (int H : Hs) {
            seen(molly_rmsnorm_fwd(st, A, B, C, nullptr, rows, H, 1e-6f), "rmsnorm_fwd");
            seen(molly_rmsnorm_bwd(st, A, B, C, (rows & 1) ? nullptr : res, fake(9), (rows & 2) ? nullptr : fake(10), rows & 1, 0, (float*)fake(11), rows, H,
                                   1e-6f), "rmsnorm_bwd");
            (void)molly_rmsnorm_bwd_blocks(rows);
            seen(molly_layernorm_fwd(st, A, B, C, fake(9), rows, H, 1e-5f), "layernorm_fwd");
            (void)molly_layernorm_bwd_blocks(rows);
            seen(molly_swiglu_fwd(st, A, C, rows, H), "swiglu_fwd");
            seen(molly_swiglu_bwd(st, A, B, C, rows, H), "swiglu_bwd");
            seen(molly_transpose_bf16(st, A, C, rows, H, H, rows), "transpose");
            (void)molly_colsum_parts(rows);
            seen(molly_colsum_bf16(st, A, rows, H, H, C, rows & 1, 0, (float*)fake(9)), "colsum");
            seen(molly_ce_fwd_bwd(st, A, (const int64_t*)B, (float*)C, (const float*)fake(9), rows, H, H, -100, 1), "ce_fwd_bwd");
            seen(molly_argmax_f32(st, (const float*)A, (int64_t*)C, rows, H, H), "argmax");
            seen(molly_argmax_f32_ws(st, (const float*)A, (int64_t*)C, rows, H, H, (H & 64) ? fake(17) : nullptr, molly_argmax_workspace(rows)), "argmax_ws");
        }
    for (long n : ns) {
        seen(molly_sqnorm_bf16(st, A, n, (float*)fake(9), (float*)C, 0), "sqnorm");
        seen(molly_adamw_step(st, (float*)A, (float*)B, (float*)C, fake(9), fake(10), n, 3e-5f, 0.9f, 0.999f, 1e-8f, 0.01f, 3, (const float*)fake(11),
                              (const float*)fake(12)), "adamw");
        seen(molly_cast_f32_to_bf16(st, (const float*)A, C, n), "cast");
        seen(molly_cast_bf16_to_f32(st, A, (float*)C, n), "cast");
        seen(molly_gelu_fwd(st, A, C, n), "gelu_fwd");
        seen(molly_dropout_bf16(st, A, C, n, 0.05f, 42, 0), "dropout");
        seen(molly_scale_bf16(st, C, n, 0.5f), "scale");
        seen(molly_sum_f32(st, (const float*)A, n, nullptr, (float*)C, 0), "sum_f32");
        seen(molly_count_valid(st, (const int64_t*)A, n, -100, (float*)B, (float*)C), "count_valid");
        seen(molly_reduce_rows_bf16(st, A, 8, n, C), "reduce_rows");
    }
    (void)molly_sqnorm_blocks();
    for (int M : {0, 1, 512, 16384, 1 << 20}) {
        const int wsb = molly_batch_sort_workspace(M > 0 ? M : 1);
        seen(molly_batch_assemble(st, (const int*)A, (const int*)B, M ? 8 : 0, M / 8 > 0 ? M / 8 : 1, 151936, -100, (const int*)fake(9), 8, (const int*)fake(10), 8,
                                  512, 512, nullptr, 0, 0, 0, (int64_t*)fake(11), (int*)fake(12), (int*)fake(13), (int64_t*)fake(14), (int*)fake(15), nullptr,
                                  nullptr, (unsigned char*)fake(16), (int*)fake(17), (int*)fake(18), (int*)fake(19), (int*)fake(20), (int64_t*)fake(21),
                                  (int*)fake(22), fake(23), wsb), "batch_assemble");
    }
    seen(molly_probe_hog(st, 16, 100, C), "probe_hog");

    printf("host-asan: %ld calls (%ld accepted, %ld rejected with an error text), %ld dry launches; last: %s\n", calls, accepted, rejected - silent,
           molly_dry_launch_count(), molly_dry_last_launch());
    for (const Stat& s : stats) printf("  %-24s %6ld calls %6ld accepted\n", s.what, s.calls, s.ok);
    for (const Stat& s : stats)
        if (s.ok == 0 && strcmp(s.what, "ctx_set_workspace") != 0) { printf("host-asan: FAILED (no call of %s was accepted: the driver no longer matches the ABI)\n", s.what); return 1; }
    if (silent) { printf("host-asan: FAILED (%ld calls rejected without an error text)\n", silent); return 1; }
    printf("host-asan: ok\n");
    return 0;
}
