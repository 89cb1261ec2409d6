/* molly_hip.h — C ABI of the MI355X (gfx950) hot-path library `libmolly_hip.so`.
 *
 * The reference (SeedLLM/molly) has no FFI of its own: its hot path is the Python nn.Module surface of
 * `OmicsOne` (reference: src/model/omics_one.py:10-233) whose arithmetic executes inside HuggingFace
 * transformers / Liger / flash-attn / DeepSpeed CUDA kernels.  This header is the boundary a maintainer
 * binds instead of those kernels (ctypes stub: INTEGRATION.md).  Each entry point names the reference /
 * third-party op it replaces ("HF:" = transformers/, pinned 4.53.0 in reference requirements.txt:21).
 *
 * Conventions
 *  - plain pointers + sizes only; every pointer is DEVICE memory unless named `h_*`;
 *  - `stream` is a hipStream_t (0 = default stream); all work is asynchronous on it; no allocation,
 *    no synchronisation inside (graph-capture safe);
 *  - bf16 tensors are uint16 storage, row-major, innermost dimension contiguous; `ld*` are in ELEMENTS;
 *  - return 0 on success; non-zero on a rejected argument or launch failure, message via
 *    molly_last_error() (thread-local).  Nothing is silently ignored.
 *  - threading: the error message and the DEFAULT GEMM context (what molly_gemm_bf16 / molly_gemm_grouped_bf16 launch through
 *    and the molly_gemm_set_* setters edit) are thread-local: a host thread never inherits another's knobs or scratch memory.
 *    Everything a GEMM launch decision reads — launch shape, schedule, tile choice, scratch — lives in a context
 *    (molly_gemm_ctx_*); the *_ctx entry points launch through an explicit one, so two models (or a trainer and an evaluator)
 *    in one process keep separate launch state.  One context must not launch onto two streams at the same time (its scratch
 *    memory is shared by its launches); contexts with their own scratch may.
 */
#ifndef MOLLY_HIP_H
#define MOLLY_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* molly_last_error(void);
int molly_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * GEMM  C[M,N] = A[M,K] · B[N,K]^T (+bias[N]) (GELU) (+res[M,N]) (+= C)
 * replaces: every nn.Linear on the path — Qwen3 q/k/v/o/gate/up/down/lm_head (HF:models/qwen3/
 * modeling_qwen3.py:76-83,225-236,495), ESM query/key/value/dense (HF:models/esm/modeling_esm.py:336-338,
 * 402,445,456), Molly's projectors (reference src/model/omics_one.py:23-29,91); the dgrad/wgrad GEMMs that
 * autograd issues for them; erf-GELU epilogue = HF:models/esm/modeling_esm.py:82-86.
 * K % 64 == 0, N % 4 == 0, lda/ldb % 8 == 0, 16-byte aligned bases. */
enum {
    MOLLY_GEMM_BIAS = 1,        /* + bias[n] (bf16)                              */
    MOLLY_GEMM_GELU = 2,        /* exact erf GELU after bias                      */
    MOLLY_GEMM_RESIDUAL = 4,    /* + res[m,n] (bf16, ldres)                       */
    MOLLY_GEMM_ACCUMULATE = 8,  /* C += result (same dtype as C)                  */
    MOLLY_GEMM_OUT_F32 = 16,    /* C is fp32 instead of bf16                      */
    MOLLY_GEMM_TRANS_OUT = 32,  /* store C^T: the output buffer is [N][M] (ldc >= M); molly_gemm_bf16 with a_kmajor=0,
                                   b_kmajor=1 only — the wgrad form  dW[N',K'] = (x^T)[K',tok] dy[tok,N']  stored as dW */
    MOLLY_GEMM_SWIGLU = 64,     /* Qwen3MLP's gate|up projection with the activation fused (HF:models/qwen3/modeling_qwen3.py:
                                   76-83, Liger swiglu in the reference: src/train.py:130-132): B = [gate_proj | up_proj] weights
                                   ([2*ff][K]), C = [gate | up] [M][2*ff] as usual, and `res` is an OUTPUT: act[M][ff] (ldres)
                                   = silu(gate) * up, bit-identical to molly_swiglu_fwd on C.  NT form, ff % 128 == 0, no
                                   other flag. */
    MOLLY_GEMM_SWIGLU_BWD = 128 /* the backward of the same activation fused into the down-projection's dgrad (what autograd runs
                                   as a GEMM + Liger's swiglu backward): molly_gemm_bf16(A = d(out) [M][h], B = down_proj.weight
                                   [h][ff] k-major, a_kmajor = 0, b_kmajor = 1), N = ff; `res` = [gate | up] [M][2*ff] (ldres) is
                                   READ and C = d[gate | up] [M][2*ff] (ldc) is written — d(act) never reaches HBM.  Bit-identical
                                   to the dgrad GEMM followed by molly_swiglu_bwd.  No other flag. */
};
int molly_gemm_nt_bf16(void* stream, const void* A, const void* B, void* C, const void* bias, const void* res,
                       int M, int N, int K, int lda, int ldb, int ldc, int ldres, int flags);

/* General operand layouts: a_kmajor / b_kmajor = operand stored [K][rows] instead of [rows][K] (no HBM transposes):
 *   dgrad  dx[M,K'] = dy[M,N'] W[N',K']      -> molly_gemm_bf16(A=dy, B=W, a_kmajor=0, b_kmajor=1)
 *   wgrad  dW[N',K'] = dy[Mtok,N']^T x[Mtok,K'] -> molly_gemm_bf16(A=dy, B=x, a_kmajor=1, b_kmajor=1), K = Mtok (any value)
 * Same epilogue flags as molly_gemm_nt_bf16 (which is the a_kmajor=b_kmajor=0 case). */
int molly_gemm_bf16(void* stream, const void* A, const void* B, void* C, const void* bias, const void* res, int M, int N,
                    int K, int lda, int ldb, int ldc, int ldres, int flags, int a_kmajor, int b_kmajor);

/* Grouped launch of up to 16 GEMMs that share K, the operand layouts (A k-contiguous [M][K], B k-major [K][N]: the dgrad /
 * weight-gradient form) and the epilogue flags (MOLLY_GEMM_ACCUMULATE, MOLLY_GEMM_OUT_F32); each problem has its own
 * pointers, sizes and transposed-output choice (trans_out != 0: C is [N][M], ldc >= M).  One persistent launch walks the
 * concatenated tile list: the four weight-gradient GEMMs of a decoder layer (64 + 128 + 192 + 384 tiles of 256 x 256 =
 * three full rounds of the 256 CUs) need no split-K slabs and no reduce launches; the fourteen rank-r adapter gradients of a
 * layer (dA, dB of seven targets) are one launch instead of fourteen split-K pairs.  `problems` is a HOST array. */
typedef struct {
    const void* A; const void* B; void* C;
    int M, N, lda, ldb, ldc, trans_out;
} molly_gemm_problem;
int molly_gemm_grouped_bf16(void* stream, const molly_gemm_problem* problems, int count, int K, int flags);

/* ---- launch state: contexts.  A context holds the launch knobs (MOLLY_GEMM_KEY_*), the scratch memory and the record of the
 * last configuration; NULL = the calling thread's default context.  The reference has no such state: cuBLAS picks its kernels
 * per call (every nn.Linear, HF:models/qwen3/modeling_qwen3.py:76-83 et al.); this is the boundary's equivalent of a cuBLAS
 * handle + workspace. */
enum {
    MOLLY_GEMM_KEY_PERSISTENT_BLOCKS = 1, /* 256x256 kernel: resident blocks (default 256 = one per CU; a multiple of 8); 0 = one
                                             block per tile; -t = blocks of t tiles each (placed by the hardware dispatcher on
                                             whatever CUs are free: the setting for a GEMM that runs beside a collective).
                                             Stream-K launches take the same three forms (shares instead of tiles). */
    MOLLY_GEMM_KEY_SCHEDULE = 2,          /* barrier schedule of the 256x256 kernel: -1 / 0 = four phases (16 MFMAs each) per
                                             K-tile (default), 1 = two phases (32 MFMAs); an A/B knob */
    MOLLY_GEMM_KEY_FORCE_TILE = 3,        /* 0 = heuristic, 128 = force the 128x128 kernel, 512 = force the 256x256 kernel */
    MOLLY_GEMM_KEY_GROUP_M = 4,           /* M-tiles per group in the 256x256 kernel's tile walk (L2 locality; default 4) */
    MOLLY_GEMM_KEY_SMALL_GRID_TILE = 5,   /* with stream-K off: 128 (default) | 512 for grids that fill the chip neither plain
                                             nor split-K */
    MOLLY_GEMM_KEY_MIN_KTILES = 6,        /* split-K: shortest K-slice, in 64-wide K-tiles, for a grid that is not skinny (16) */
    MOLLY_GEMM_KEY_STREAMK = 7,           /* 1 (default): a grid with M, N >= 256 that does not fill whole rounds of the 256 CUs runs
                                             as ONE stream-K launch (every block an equal share of all tiles' K-tiles; the pieces of
                                             a cut tile combined inside the launch by whichever piece finishes last, in a fixed
                                             order: deterministic) WHERE that beats split-K slabs + reduce launch / the 128x128
                                             kernel by the launcher's cost model (long contractions just past a whole round);
                                             2: wherever stream-K is able to run (tests, tools); 0: never */
    MOLLY_GEMM_KEY_SKINNY = 8,            /* 1 (default): forward GEMMs with M <= 64 rows (the decode step of `generate`) run on the
                                             weight-streaming kernel — one launch, W read once straight into registers; 0: K split over
                                             the chip through the tile kernel + fp32 slabs + reduce launch (round 2's path; A/B) */
    MOLLY_GEMM_KEY_SMALL3 = 9,            /* A/B knob: 1 = 128x128 grids of at most one tile per CU run a 3-stage ring; 0 (default) = the 2-stage loop */
    MOLLY_GEMM_KEY_DYNAMIC = 10,          /* 1: plain 256x256 launches of more than one round run as 256 resident blocks that DRAW their tiles
                                             (one ticket counter per XCD label in the workspace header; the next ticket fetched five
                                             K-tiles ahead, so the rolling prefetch never waits for it) — for GEMMs that share the chip
                                             with a collective's kernels; same results as the static walk.  0 (default): static walk */
    MOLLY_GEMM_KEY_SMALL_SPLIT = 11,      /* 1 (default): a grid of at most 256 128x128 blocks with a long contraction (encoder ffn2 at one sample
                                             per GPU) is priced for split-K with slices down to 4 K-tiles; 0: round 2's rule (A/B) */
    MOLLY_GEMM_KEY_ROWS_TILED = 12,       /* 1 (default): forward GEMMs with M <= 64 rows that the weight-streaming kernel does not take (M > 16 on
                                             matrices beyond 64 MB: Qwen3-8B gate|up / down at batch 32) run on the tiled decode-row kernel —
                                             x and W through LDS-DMA in whole lines, x once per workgroup and K-tile, K split over the chip;
                                             0: split-K through the 256x256 kernel (round 2's path; A/B) */
    MOLLY_GEMM_KEY_DYNAMIC_MIN_WORK = 13, /* with DYNAMIC = 1: only launches of at least this many work items draw their tiles (default 257: every
                                             launch of more than one round) */
    MOLLY_GEMM_KEY_ROWS_MAX_M = 14,       /* largest M (64..8192) for which a forward GEMM whose 128x128 grid has at most 192 blocks runs as 64-row
                                             tiles of the tiled decode-row kernel (the encoders' projections at 512 / 1024 rows) */
    MOLLY_GEMM_KEY_ROWS_GU = 15,          /* decode rows of gate | up with SwiGLU (molly_gemm_rows_tail_bf16_ctx, tail 2, M <= 32): 1 (default) = the one-slice kernel
                                             that forms silu(gate) * up from its accumulators (no slabs, no combine launch), tile chosen by the launcher;
                                             32 | 64 | 128 = that many W rows per tile; 0 = K slices + the combine launch */
    MOLLY_GEMM_KEY_ROWS_BN = 16,          /* W rows per tile of the tiled decode-row kernel at M <= 32: 64 (default) | 128 */
    MOLLY_GEMM_KEY_STREAM_EPI = 17,       /* 1 (default; env MOLLY_GEMM_STREAM_EPI): plain NT launches of whole interior 256x256 tiles (M, N multiples of 256,
                                             one K slice, no epilogue flag) run the streaming-epilogue instantiation: whole-line stores from inside the K
                                             loop, which runs on into the next tile (bit-identical results; last_config 514); 0 = epilogue after the K loop */
    MOLLY_GEMM_KEY_LAST_CONFIG = 100      /* read-only: 16 (decode-row kernel) | 32 (tiled decode-row kernel) | 128 | 512 | 513 (512 drawing its tiles) | 514 (512 with the streaming epilogue) (+ 1000 * split-K factor, + 50000 stream-K, + 100000 * problems
                                             of a grouped launch) of the context's most recent launch */
};
/* C = A B^T + A2 B2^T (+ the epilogue `flags` of molly_gemm_bf16: bias, GELU, residual, accumulate, or MOLLY_GEMM_SWIGLU) in ONE accumulation:
 * the NT form with K2 / 64 more K-tiles read from a second operand pair (A2 [M][K2] row stride lda2, B2 [N][K2] row stride ldb2).
 * Replaces PEFT's `result = base_layer(x) + lora_B(lora_A(dropout(x))) * scaling` second half (`lora.Linear.forward`, injected by
 * src/utils/tools.py:378-389) as a read-modify-write pass over the projection's output: A2 = t = scaling * lora_A(dropout(x)), B2 = lora_B (for a
 * fused q|k|v or gate|up weight: the block-diagonal stack of its targets' lora_B).  One rounding of the fp32 sum.
 * molly_gemm_kx_supported: 1 when (M, N, K, K2, flags) is taken by this launch (whole 256 x 256 tiles' worth of work on the plain persistent walk;
 * K % 64 == 0, 64 <= K2 <= 512, K2 % 64 == 0, no transposed / fp32 output); otherwise run molly_gemm_bf16_ctx and an accumulating second launch. */
int molly_gemm_kx_supported(void* ctx, int M, int N, int K, int K2, int flags);
int molly_gemm_kx_bf16_ctx(void* ctx, void* stream, const void* A, const void* B, void* C, const void* bias, const void* res, int M, int N, int K,
                           int lda, int ldb, int ldc, int ldres, int flags, const void* A2, const void* B2, int K2, int lda2, int ldb2);

/* decode rows (M <= 64, the plain forward form) on the tiled decode-row kernel, with the kernel the decode step would launch next
 * folded into the launch that combines the K slices.  tail 1: C = A B^T (+ bias) (+ res) [M][N] and tail_out = RMSNorm(C) * gain
 * [M][N] (the next block's input norm, HF:models/qwen3/modeling_qwen3.py:50-63, 262-276); tail 2: C = [gate | up] [M][N] and
 * tail_out = silu(gate) * up [M][N / 2] (HF:models/qwen3/modeling_qwen3.py:76-83; C may be NULL: gate | up itself is not kept).  Same roundings as
 * the separate kernels.
 * molly_gemm_rows_tail_supported: 1 when this context would run M x N x K that way (else use the GEMM and the kernel). */
int molly_gemm_rows_tail_supported(void* ctx, int M, int N, int K, int tail);
/* tail 3 (a projection of its own because of its arguments): A W^T (+ bias) = the q | k | v row of one decode step -> q/k-norm + rotary
 * (molly_norm_rope_cache_fwd's arithmetic), q | k to dst [M][ld_dst], k and v appended to the caches at slot[m]
 * (HF:models/qwen3/modeling_qwen3.py:225-236; DynamicCache.update) — norm + qkv + rope + cache write as two launches instead of five. */
int molly_gemm_rows_qkv_bf16_ctx(void* ctx, void* stream, const void* A, const void* W, const void* bias, int M, int N, int K, int lda,
                                 int ldw, const void* q_norm_w, const void* k_norm_w, const float* cos, const float* sin,
                                 const int* positions, float eps, int n_q_heads, int n_k_heads, int head_dim, void* dst, int ld_dst,
                                 void* kcache, void* vcache, const int* slot, int ld_cache);
/* tail 4: no combine at all — A W^T left as n >= 2 fp32 K-slice slabs [n][M][N] in the context's scratch (their sum is the product) for a consumer
 * that adds them itself (molly_attn_decode_qkv).  slabs_out[0] = device address of the slabs, slabs_out[1] = n; valid until the context's next launch. */
int molly_gemm_rows_slabs_bf16_ctx(void* ctx, void* stream, const void* A, const void* W, int M, int N, int K, int lda, int ldw,
                                   long* slabs_out);
int molly_gemm_rows_tail_bf16_ctx(void* ctx, void* stream, const void* A, const void* B, void* C, const void* bias, const void* res,
                                  int M, int N, int K, int lda, int ldb, int ldc, int ldres, int flags, int tail, const void* gain,
                                  float eps, void* tail_out, int ld_tail);
int molly_gemm_ctx_create(void** out);
int molly_gemm_ctx_destroy(void* ctx);
int molly_gemm_ctx_set(void* ctx, int key, long value);
int molly_gemm_ctx_get(void* ctx, int key);                 /* the value (an answer, not a status); -1 = unknown key */
/* scratch of a context: device memory the caller owns, 256-byte aligned, valid until replaced; NULL / 0 = none (no split-K, no
 * stream-K).  Layout: a 512 KiB + 576 B header (stream-K's error word, one counter line per tile, the dynamic fetch's 8 ticket lines; cleared here, synchronously,
 * once) followed by fp32 slabs: split-K partial sums [splits][M][N], or stream-K's two 256 KiB accumulator images per block.
 * 129 MiB serve every stream-K launch of 256 blocks. */
int molly_gemm_ctx_set_workspace(void* ctx, void* ptr, long bytes);
int molly_gemm_ctx_streamk_timeouts(void* ctx);            /* diagnostics: non-zero = a stream-K wait gave up (device read) */
int molly_gemm_bf16_ctx(void* ctx, void* stream, const void* A, const void* B, void* C, const void* bias, const void* res,
                        int M, int N, int K, int lda, int ldb, int ldc, int ldres, int flags, int a_kmajor, int b_kmajor);
int molly_gemm_grouped_bf16_ctx(void* ctx, void* stream, const molly_gemm_problem* problems, int count, int K, int flags);

/* The same knobs on the calling thread's DEFAULT context (kept for tools and tests) */
int molly_gemm_set_workspace(void* ptr, long bytes);
int molly_gemm_last_config(void);
int molly_gemm_set_small_grid_tile(int tile);
int molly_gemm_set_group_m(int g);
int molly_gemm_set_persistent_blocks(int n);
int molly_gemm_set_min_ktiles(int n);
int molly_gemm_set_schedule(int mode);
int molly_gemm_force_tile(int bm);
int molly_gemm_set_streamk(int on);

/* out[C,R] = in[R,C]^T (bf16).  Used to keep W^T copies for dgrad and X^T / dY^T for wgrad. */
int molly_transpose_bf16(void* stream, const void* in, void* out, int R, int C, int ld_in, int ld_out);

/* ------------------------------------------------------------------------------------------------
 * RMSNorm — HF:models/qwen3/modeling_qwen3.py:59-64 (Liger rms_norm when --use_liger, reference
 * src/train.py:130-132): y = w * bf16(x * rsqrt(mean(x^2) + eps)).  H % 8 == 0, H <= 4096.
 * bwd: dx = norm-backward(g) (+ dres, the residual-branch gradient, fused); dw (+)= sum_rows g*xhat, reduced
 * deterministically through `workspace` (molly_rmsnorm_bwd_blocks(rows) * H floats). */
int molly_rmsnorm_fwd(void* stream, const void* x, const void* w, void* y, float* rstd_or_null, int rows, int H, float eps);
/* the same forward with a TRANSPOSED second store: yT[H][ld_t] (ld_t >= rows) = y^T — the k-contiguous operand the weight-gradient GEMM
 * of the projection behind the norm wants (it replaces a molly_transpose_bf16 launch per operand).  rows % 64 == 0, H % 512 == 0, H <= 2048. */
int molly_rmsnorm_fwd_t(void* stream, const void* x, const void* w, void* y, void* yT, int rows, int H, int ld_t, float eps);
int molly_rmsnorm_bwd_blocks(int rows);
int molly_rmsnorm_bwd(void* stream, const void* x, const void* w, const void* g, const void* dres_or_null, void* dx,
                      void* dw, int dw_f32, int dw_accumulate, float* workspace, int rows, int H, float eps);
/* Deferred gain gradients: molly_rmsnorm_bwd with dw == NULL and molly_norm_rope_bwd with dq_w == dk_w == NULL leave their
 * per-block partials in the workspace; ONE launch then reduces many of them: items_dev = device array of n_items records
 * {const float* part; void* out; int nb, H, row_stride, pad} (32 bytes), out_t[j] (+)= sum_b part_t[b * row_stride + j], j < H. */
int molly_colsum_batched(void* stream, const void* items_dev, int n_items, int max_H, int out_f32, int accumulate);

/* per-head RMSNorm (optional) + q pre-scale (optional) + rotary (optional) over the q|k heads of a fused
 * projection buffer.  Qwen3: q_norm/k_norm then RoPE — HF:models/qwen3/modeling_qwen3.py:252-257,148-170.
 * ESM-2: q *= hd^-0.5 then fp32 rotary, no norm — HF:models/esm/modeling_esm.py:374-378,56-79.
 * src[M, ld_src] holds n_q_heads then n_k_heads heads of head_dim; dst likewise; cos/sin fp32 [n_pos, head_dim/2];
 * positions int32 [M] or NULL (= m %% T, i.e. arange(T) per sample: HF:qwen3:386-389). */
int molly_norm_rope_fwd(void* stream, const void* src, void* dst, const void* q_norm_w, const void* k_norm_w,
                        const float* cos, const float* sin, const int* positions, int M, int T, int n_q_heads,
                        int n_k_heads, int head_dim, int ld_src, int ld_dst, float eps, float q_scale);
/* the same for one decode step (T = 1 row per sample), together with the KV-cache append HF's DynamicCache.update does behind it
 * (reference src/model/omics_one.py:220-232): src rows hold q | k | v heads; the k heads after norm + rotary and the v heads as they are
 * are also written to row slot[m] of kcache / vcache ([rows][ld_cache], n_k_heads * head_dim wide). */
int molly_norm_rope_cache_fwd(void* stream, const void* src, void* dst, const void* q_norm_w, const void* k_norm_w,
                              const float* cos, const float* sin, const int* positions, int M, int T, int n_q_heads,
                              int n_k_heads, int head_dim, int ld_src, int ld_dst, float eps, float q_scale,
                              void* kcache, void* vcache, const int* slot, int ld_cache);
int molly_norm_rope_bwd_blocks(void);
int molly_norm_rope_bwd(void* stream, const void* src, const void* g, void* dsrc, const void* q_norm_w,
                        const void* k_norm_w, const float* cos, const float* sin, const int* positions, void* dq_w,
                        void* dk_w, int dw_f32, int dw_accumulate, float* workspace, int M, int T, int n_q_heads,
                        int n_k_heads, int head_dim, int ld_src, int ld_g, int ld_out, float eps, float q_scale);

/* SwiGLU — HF:models/qwen3/modeling_qwen3.py:82 (Liger swiglu): out = silu(gate) * up, gate_up = [gate | up] per row. */
int molly_swiglu_fwd(void* stream, const void* gate_up, void* out, long rows, int ff);
int molly_swiglu_bwd(void* stream, const void* gate_up, const void* dout, void* dgate_up, long rows, int ff);

/* row gather/scatter: dst[dst_idx[i] | i] (+)= src[src_idx[i] | i]; negative index skips the row.
 * embedding lookup = reference src/model/omics_one.py:164 (int64 ids as src_idx64);
 * omic injection hs[b, start+1 : start+1+k] = emb[i, :k] = reference src/model/omics_one.py:93-97 (dst_idx32). */
int molly_copy_rows(void* stream, const void* src, const int64_t* src_idx64, const int* src_idx32, void* dst,
                    const int* dst_idx32, long n, int H, int ld_src, int ld_dst, int accumulate);
/* embedding backward through a sorted index: for unique id u, dE[uid[u]] += sum_k g[order[k]],
 * k in [seg_start[u], seg_start[u+1]); uid < 0 skips.  Deterministic (no atomics).  n_unique_dev (nullable): the number of
 * segments lives on the device (molly_batch_assemble wrote it); n_unique is then only the launch bound (>= the true count). */
int molly_embed_bwd(void* stream, const void* g, const int* order, const int* seg_start, const int64_t* uid,
                    int n_unique, void* dE, int H, int ld_g, const float* row_scale, const int* n_unique_dev);

/* cross-entropy on bf16 logits, in place -> d(logits) — HF:loss/loss_utils.py:32-71 (ForCausalLMLoss; Liger
 * fused-linear-CE when --use_liger).  `labels` are already shifted (row r is scored against labels[r]);
 * rows with ignore_index get zero loss/grad.  dlogits = (softmax - onehot) * (*scale). */
int molly_count_valid(void* stream, const int64_t* labels, long n, int ignore_index, float* scale_out, float* count_out);
int molly_ce_fwd_bwd(void* stream, void* logits, const int64_t* labels, float* row_loss, const float* scale, int rows,
                     int V, int ld, int ignore_index, int write_grad);
/* classifier-head losses of the Enc-Head baselines — reference baselines/model.py:196-204: mode 0 = F.cross_entropy
 * (int64 labels, ignore_index rows skipped), mode 1 = F.binary_cross_entropy_with_logits (fp32 targets [rows][V]).
 * logits bf16 [rows][ld], any V <= ld; row_loss[r] = summed loss of row r; logits are overwritten by
 * d(logits) * (*scale) when write_grad (padding columns V..ld-1 get 0). */
int molly_cls_loss_fwd_bwd(void* stream, void* logits, const int64_t* labels, const float* targets, float* row_loss,
                           const float* scale, int rows, int V, int ld, int mode, int ignore_index, int write_grad);
/* greedy token selection of generate(do_sample=False) — HF:generation/utils.py `next_tokens = torch.argmax(scores, -1)`
 * (reference src/model/omics_one.py:220-232 passes do_sample through): first maximal index per row of fp32 logits [rows][ld]. */
int molly_argmax_f32(void* stream, const float* x, int64_t* out, int rows, int V, int ld);
/* the same with every row cut over up to 8 workgroups + a merge launch (few rows of a wide vocabulary: 32 rows alone keep 32 of the 256 CUs busy);
 * workspace: molly_argmax_workspace(rows) bytes, NULL or too small = the one-launch form */
int molly_argmax_workspace(int rows);
int molly_argmax_f32_ws(void* stream, const float* x, int64_t* out, int rows, int V, int ld, void* workspace, long workspace_bytes);
/* sampling of generate(do_sample=True) — the reference's inference settings (src/inference_lora.py:293-298: temperature 0.8,
 * top_p 0.95, top_k 20, repetition_penalty 1.1) through HF's processors in HF's order (HF:generation/logits_process.py:
 * RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper; softmax + multinomial in
 * HF:generation/utils.py _sample).  fp32 logits [rows][ld] (modified in place by the penalty), generated int64
 * [rows][ld_generated] (first n_generated columns), one token per row into next_token.  The draw is Philox4x32-10 keyed by
 * (seed, step, row): HF's distribution, not torch's random stream.  probs_out / ids_out [rows][cap_out], n_out [rows]
 * (nullable): the surviving tokens in descending order with their final probabilities.  1 <= top_k <= 1024. */
int molly_sample_logits(void* stream, float* logits, int rows, int V, int ld, const int64_t* generated, int n_generated,
                        int ld_generated, float repetition_penalty, float temperature, int top_k, float top_p, uint64_t seed,
                        int step, int64_t* next_token, float* probs_out_or_null, int64_t* ids_out_or_null,
                        int* n_out_or_null, int cap_out);
int molly_sum_f32(void* stream, const float* x, long n, const float* scale_or_null, float* out, int accumulate);

/* LayerNorm forward (affine, eps 1e-5) — ESM pre-LN blocks and emb_layer_norm_after: HF:models/esm/
 * modeling_esm.py:429,518,552. */
int molly_layernorm_fwd(void* stream, const void* x, const void* w, const void* b, void* y, int rows, int H, float eps);
/* LayerNorm backward of the ESM blocks (encoder training, reference `--train-bio`: src/utils/tools.py:326-330):
 * dx = rstd (g w - mean(g w) - xhat mean(g w xhat)) (+ dres); dw (+)= sum_rows g xhat, db (+)= sum_rows g.
 * workspace: 2 * molly_layernorm_bwd_blocks(rows) * H floats. */
int molly_layernorm_bwd_blocks(int rows);
int molly_layernorm_bwd(void* stream, const void* x, const void* w, const void* g, const void* dres, void* dx, void* dw,
                        void* db, int dw_f32, int dw_accumulate, float* workspace, int rows, int H, float eps);
/* erf-GELU on a stored pre-activation and its backward (ESM intermediate activation, HF:models/esm/modeling_esm.py:56-60);
 * training keeps z, so the forward GEMM runs without the fused GELU epilogue. */
int molly_gelu_fwd(void* stream, const void* z, void* out, long n);
int molly_gelu_bwd(void* stream, const void* z, const void* dout, void* dz, long n);

/* ESM embeddings — HF:models/esm/modeling_esm.py:224-271 (+ position ids :1050-1063): word embedding,
 * token-dropout rescale, optional learned absolute positions (pos_emb NULL for rotary models), x mask.
 * Also emits the int32 position ids and per-sequence valid key length (last non-pad index + 1). */
int molly_esm_embed(void* stream, const int64_t* ids, const void* word_emb, const void* pos_emb_or_null, void* out,
                    int* pos_ids_out_or_null, int* kv_len_out_or_null, int n_seq, int K, int H, int pad_id, int mask_id,
                    int token_dropout);

/* Batch assembly on the device — replaces the per-step host work around the model call: reference
 * src/model/omics_one.py:69-72 (stack, mask, device->host sync assert), :93-97 (one slice copy per span), :104-118 (per-row
 * .to(device)), and HF's label shift (HF:loss/loss_utils.py:60-63).  Inputs are slices of ONE packed int32 image that arrived
 * by one pinned host->device copy: ids32/labels32 [B*T], spans [n_spans][4] = (b, start, group 0|1, row in the group's encoder
 * batch; start -1 = encoded but never scattered), omic32_g* [n_rows][K].  Outputs: labels_shifted int64 [B*T] and the ordered
 * scored_rows / *n_scored (label != ignore_index); per group the int64 encoder ids and dst[row*K + j] = b*T + start + 1 + j
 * for j < k (k = min(config token count, K)), else -1; overwritten[B*T] (bytes); and, when `order` is given, the sorted
 * embedding-gradient index over the NOT overwritten rows (stable radix sort by token id): order[], seg_start[n_unique + 1],
 * uid[n_unique], *n_unique — all on the device.  keys_tmp: 2*B*T ints, vals_tmp: B*T ints, sort_ws: molly_batch_sort_workspace
 * bytes.  labels32 / ids32 / order may be NULL (inference, process_omic_sequences). */
int molly_batch_sort_workspace(int M);
int molly_batch_assemble(void* stream, const int* ids32, const int* labels32, int B, int T, int vocab, int ignore_index,
                         const int* spans, int n_spans, const int* omic32_g0, int n_rows0, int K0, int k0,
                         const int* omic32_g1, int n_rows1, int K1, int k1, int64_t* labels_shifted, int* scored_rows,
                         int* n_scored, int64_t* omic64_g0, int* dst_g0, int64_t* omic64_g1, int* dst_g1,
                         unsigned char* overwritten, int* keys_tmp, int* vals_tmp, int* order, int* seg_start, int64_t* uid,
                         int* n_unique, void* sort_ws, long sort_ws_bytes);

/* optimizer shard step — torch.optim.AdamW (HF `adamw_torch`, reference src/trainer/omics_trainer.py:53-60) on an
 * fp32 master shard with bf16 grads; global-norm clip = torch.nn.utils.clip_grad_norm_ (reference
 * src/trainer/domain_loss.py:676-708; DeepSpeed gradient_clipping, src/configs/ds_z2_config.json:5). */
/* local half of the all-to-all gradient reduce-scatter (SURVEY.md §5 option 2; replaces the reduction inside DeepSpeed's / NCCL's
 * reduce-scatter, src/configs/ds_z2_config.json:23): out[i] = bf16(sum over r of in[r][i]) in fp32, r ascending. */
int molly_reduce_rows_bf16(void* stream, const void* in, int rows, long n, void* out);
int molly_sqnorm_blocks(void);
int molly_sqnorm_bf16(void* stream, const void* g, long n, float* workspace, float* out, int accumulate);
/* Overflow guard (DeepSpeed's ZeRO step skips the update when a gradient is inf/NaN): a non-finite norm makes the
 * coefficient NaN and adds 1 to *skipped_count; molly_adamw_step given a NaN *grad_scale leaves master / moments /
 * parameters untouched, and takes its bias corrections at (step - *skipped_count): skipped steps do not age Adam. */
int molly_clip_coef(void* stream, const float* norm_sq, float max_norm, float pre_scale, float* norm_out, float* coef_out,
                    float* skipped_count_or_null);
int molly_adamw_step(void* stream, float* master, float* exp_avg, float* exp_avg_sq, const void* grad, void* param_out,
                     long n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                     const float* grad_scale_or_null, const float* skipped_count_or_null);
/* out[j] (+)= sum_r x[r, j] — bias gradient of the projector nn.Linear (reference src/model/omics_one.py:23-29).
 * workspace: molly_colsum_parts(rows) * H floats. */
int molly_colsum_parts(int rows);
int molly_colsum_bf16(void* stream, const void* x, int rows, int H, int ld, void* out, int out_f32, int accumulate,
                      float* workspace);
int molly_cast_f32_to_bf16(void* stream, const float* in, void* out, long n);
int molly_cast_bf16_to_f32(void* stream, const void* in, float* out, long n);

/* ------------------------------------------------------------------------------------------------
 * flash attention — replaces FlashAttention-2 / SDPA / eager attention chosen by --attn_impl (reference
 * src/train.py:578-582): Qwen3 causal GQA hd=128 (HF:models/qwen3/modeling_qwen3.py:185-208) and ESM
 * bidirectional hd=64 with key-padding mask (HF:models/esm/modeling_esm.py:292-317).
 * Q/K/V/O are token-major: row (b*T + t) with row stride ld*, head h at column h*head_dim.
 * mask = (causal ? key <= query : all) AND kv_lo[b] <= key < kv_hi[b]  (NULL = whole sequence).
 * lse2[B, n_heads, T] = log2-domain log-sum-exp of the scaled scores (saved for the backward).
 * head_dim 64 | 128: MFMA kernel.  head_dim 8..48 (multiples of 8): a plain forward-only kernel for the mini encoders of the
 * reference's plumbing config (ESM2-t6-8M has 320 / 20 = 16); molly_attn_bwd exists for 64 and 128 only. */
int molly_attn_fwd(void* stream, const void* Q, const void* K, const void* V, void* O, float* lse2, const int* kv_lo,
                   const int* kv_hi, int B, int T, int n_heads, int n_kv_heads, int head_dim, int ldq, int ldk, int ldv,
                   int ldo, float scale, int causal);
/* the same forward with O stored a second time TRANSPOSED, OT[n_heads * head_dim][ldot >= B * T] = O^T (the k-contiguous operand of the
 * o-projection's weight gradient: replaces a molly_transpose_bf16 launch per layer).  head_dim 64 | 128, T % 128 == 0. */
int molly_attn_fwd_ot(void* stream, const void* Q, const void* K, const void* V, void* O, void* OT, float* lse2, const int* kv_lo,
                      const int* kv_hi, int B, int T, int n_heads, int n_kv_heads, int head_dim, int ldq, int ldk, int ldv, int ldo,
                      int ldot, float scale, int causal);

/* backward of molly_attn_fwd (what autograd runs through flash-attn's backward in the reference).  Recomputes P from
 * Q, K and lse2; writes dQ [.., n_heads*hd], dK/dV [.., n_kv_heads*hd] (GQA group already summed), bitwise
 * reproducible (no atomics).  delta_ws: B*n_heads*T floats of scratch. */
int molly_attn_bwd(void* stream, const void* Q, const void* K, const void* V, const void* O, const void* dO,
                   const float* lse2, float* delta_ws, void* dQ, void* dK, void* dV, const int* kv_lo, const int* kv_hi,
                   int B, int T, int n_heads, int n_kv_heads, int head_dim, int ldq, int ldk, int ldv, int ldo, int lddo,
                   int lddq, int lddk, int lddv, float scale, int causal);
/* the same with scratch: when the dK / dV grid over (kv head, key block) does not fill the chip (one sample per GPU: B = 1 of
 * scripts/train/examples/run_train_4B_z2_b1.sh:29 — 8 kv heads x T/128 key blocks = 192 blocks for 512 slots) the two passes run one
 * block per QUERY head, leave fp32 accumulator images in `workspace`, and a third launch adds a group's images in head order and
 * writes the rows: still no atomics, still bitwise reproducible.  molly_attn_bwd_workspace: the floats that takes (0 = the split
 * does not apply to these sizes); a smaller or NULL workspace runs the unsplit passes. */
int molly_attn_bwd_workspace(int B, int T, int n_heads, int n_kv_heads, int head_dim);
/* molly_attn_bwd with the q/k-norm + rotary BACKWARD (HF:models/qwen3/modeling_qwen3.py:225-236, what molly_norm_rope_bwd computes) inside the dQ and
 * dK kernels' row epilogues (round 6): Q / K are the normed + rotated rows the forward saved, X the PRE-norm projection rows [B*T][>= (nq + nk) * hd]
 * pointing at q head 0 (k heads follow), dX the gradient of that projection in the same layout (its q and k parts are written here, dV goes to dV as
 * before); qw / kw the gains, cos / sin [T][hd / 2] fp32 with position = token index within the sample.  Gain gradients leave as one row of hd floats
 * per workgroup: dwq_part [molly_attn_bwd_rope_blocks(.., 0)][hd], dwk_part [..(.., 1)][hd] (sum the rows: molly_colsum_batched).
 * Head dim 128; the dK / dV passes are never split by query head here: where molly_attn_bwd_workspace(..) > 0 (one sample per GPU) prefer
 * molly_attn_bwd_ws + molly_norm_rope_bwd. */
int molly_attn_bwd_rope_blocks(int B, int T, int n_heads, int n_kv_heads, int which);
int molly_attn_bwd_rope(void* stream, const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse2,
                        float* delta_ws, void* dV, const int* kv_lo, const int* kv_hi, int B, int T, int n_heads, int n_kv_heads,
                        int head_dim, int ldq, int ldk, int ldv, int ldo, int lddo, int lddv, float scale, int causal, const void* X,
                        int ldx, const void* qw, const void* kw, const float* cos_t, const float* sin_t, float eps, void* dX, int lddx,
                        float* dwq_part, float* dwk_part);
int molly_attn_bwd_ws(void* stream, const void* Q, const void* K, const void* V, const void* O, const void* dO,
                      const float* lse2, float* delta_ws, void* dQ, void* dK, void* dV, const int* kv_lo, const int* kv_hi,
                      int B, int T, int n_heads, int n_kv_heads, int head_dim, int ldq, int ldk, int ldv, int ldo, int lddo,
                      int lddq, int lddk, int lddv, float scale, int causal, float* workspace, long workspace_floats);

/* single-query attention over a KV cache — the decode step of HF `generate` with DynamicCache that the reference runs
 * for inference (reference src/model/omics_one.py:220-232).  q [B, ldq] (heads at column h*hd), out [B, n_heads*hd]; k/v
 * cache [B, Tmax, n_kv_heads*hd]; keys kv_lo[b] <= key < kv_hi[b] (kv_lo NULL = 0).  The query heads of one KV head are
 * processed together (K/V read once) and the key range is split over blocks (flash-decoding); `workspace` holds the fp32
 * partials of the splits: molly_attn_decode_workspace(B, n_heads, head_dim) floats (NULL = no split).  kv_len_hint: an
 * upper bound of kv_hi - kv_lo used only to choose the split count (0 = Tmax).  head_dim 64 | 128, n_heads/n_kv_heads 1|2|4|8. */
int molly_attn_decode_workspace(int B, int n_heads, int head_dim);
int molly_attn_decode(void* stream, const void* q, const void* kcache, const void* vcache, void* out, const int* kv_lo,
                      const int* kv_hi, int B, int Tmax, int n_heads, int n_kv_heads, int head_dim, int ldq, float scale,
                      int kv_len_hint, float* workspace, long workspace_floats);
/* The same attention taken straight from the q | k | v projection's K-slice slabs (molly_gemm_rows_slabs_bf16_ctx): row b of the projection is
 * the sum of qkv_slabs[s][b][:] ([n_slabs][B][(n_heads + 2 n_kv_heads) * head_dim] fp32) rounded to bf16; its q and k heads get q/k-norm (gains
 * NULL: none) and rotary at positions[b] (cos NULL: none) — molly_norm_rope_cache_fwd's arithmetic; the new key and value are appended at cache
 * row slot[b] (of the B * Tmax rows; HF DynamicCache.update) and the query heads attend to the cache keys kv_lo[b] <= key < kv_hi[b] - 1 and to
 * the new one: kv_hi counts the new token, as it does for molly_attn_decode after molly_gemm_rows_qkv_bf16_ctx, whose caches this call
 * reproduces bit for bit.  One launch where the step had three (slab combine + norm + rope + append | attention | merge of the splits). */
int molly_attn_decode_qkv(void* stream, const float* qkv_slabs, int n_slabs, const void* q_norm_w, const void* k_norm_w, const float* cos,
                          const float* sin, const int* positions, float eps, void* kcache, void* vcache, const int* slot, void* out,
                          const int* kv_lo, const int* kv_hi, int B, int Tmax, int n_heads, int n_kv_heads, int head_dim, float scale,
                          int kv_len_hint, float* workspace, long workspace_floats);

/* ------------------------------------------------------------------------------------------------
 * LoRA branch  y = W x + (alpha/r) * B (A dropout(x))  — PEFT lora.Linear.forward as the reference configures it
 * (src/utils/tools.py:379-389: r = --lora_r, lora_alpha 64, lora_dropout 0.05).  The rank-r contractions are
 * molly_gemm_bf16 calls; these are the HBM-bound parts.
 * dropout: out[i] = keep_i ? x[i] / (1-p) : 0 with keep_i a pure function of (seed, i) (Philox4x32-10), so the backward
 * regenerates the mask from the seed instead of storing it.  n % 8 == 0.  In place (out == x) is allowed.
 * accumulate != 0: out[i] += dropped value (the branch's contribution to d(x): the same mask applied to the gradient). */
int molly_dropout_bf16(void* stream, const void* x, void* out, long n, float p, uint64_t seed, int accumulate);
/* x[i] *= s (bf16, rounds once): the alpha/r scaling of the rank-r intermediate and of its gradient. */
int molly_scale_bf16(void* stream, void* x, long n, float s);
/* The two rank-r contractions next to the dropout mask, fused with it (PEFT lora.Linear.forward, reference src/utils/tools.py:379-389;
 * same mask function of (seed, flat element index of the [M, K] operand) as molly_dropout_bf16, so fused and unfused calls mix):
 * down: t[M, R] = scale * (dropout(x)[M, K] A[R, K]^T), xd (nullable) = dropout(x) written on the way (the dA weight gradient reads it).
 *       x [M, K] with row stride ldx (the mask is indexed by the logical [M, K] position), xd contiguous [M, K]; A [R, K] contiguous; t row
 *       stride ldt.  K % 64 == 0, R == 64 (the padded rank).  p == 0 and xd == NULL:
 *       the plain rank-R product (no mask is generated) — the backward's dt = scale * dy B runs through it with A = B^T.
 * up:   dx[M, K] += mask * bf16(dt[M, R] A[R, K])  — the branch's contribution to d(x).  dx contiguous, dt row stride lddt.  K % 128 == 0. */
int molly_lora_down_drop_bf16(void* stream, const void* x, const void* A, void* xd, void* t, int M, int K, int R, int ldx, int ldt, float p,
                              uint64_t seed, float scale);
int molly_lora_up_drop_acc_bf16(void* stream, const void* dt, const void* A, void* dx, int M, int K, int R, int lddt, float p, uint64_t seed);
/* n (1..3) targets that share their input in ONE pass over dx (q | k | v, gate | up): dx = bf16(... bf16(dx + mask_0 * (dt_0 A_0)) ... + mask_{n-1} * (dt_{n-1} A_{n-1})) —
 * the roundings of n molly_lora_up_drop_acc_bf16 launches one after the other, bit for bit.  dt, A, lddt, seed: host arrays of n entries. */
int molly_lora_up_drop_acc_multi_bf16(void* stream, int n, const void* const* dt, const void* const* A, void* dx, int M, int K, int R,
                                      const int* lddt, float p, const uint64_t* seed);
/* items_dev: n_items records { const void* src; void* dst; int rows; int ld_dst; } (24 bytes, device memory): src [rows][64] bf16 contiguous is
 * copied into dst (row stride ld_dst, 16-byte aligned) — the diagonal blocks of the stacked lora_B a fused projection hands to
 * molly_gemm_kx_bf16_ctx, all layers in one launch.  max_rows: the largest `rows`. */
int molly_lora_pack_b(void* stream, const void* items_dev, int n_items, int max_rows);
/* the same table, transposing: dst [64][ld_dst] = src^T (lora_B^T of every target, the `A` operand of the backward's dt = scaling * dy lora_B through
 * molly_lora_down_drop_bf16's p = 0 form) */
int molly_lora_pack_bt(void* stream, const void* items_dev, int n_items, int max_rows);
/* molly_lora_down_drop_bf16 that also leaves t^T [64][ldtT] (ldtT >= M; NULL: not wanted): the k-contiguous operand of the adapter weight gradients
 * (dB^T = t^T dy, dA = dt^T dropout(x)) without a transpose launch */
int molly_lora_down_drop_t_bf16(void* stream, const void* x, const void* A, void* xd, void* t, int M, int K, int R, int ldx, int ldt, float p,
                                uint64_t seed, float scale, void* tT, int ldtT);

/* ------------------------------------------------------------------------------------------------
 * Direct peer exchange for the ZeRO-2 step (SURVEY.md §5 option 3; reference role: DeepSpeed ZeRO-2's reduce-scatter / all-gather,
 * src/configs/ds_z2_config.json:18-27, src/train.py:606-614).  The peers' gradient / parameter / flag buffers are mapped into this
 * process by IPC (the caller does that); srcs / dsts / flags are HOST arrays of `world` device pointers in rank order.
 * reduce: out[i] = bf16(sum_r float(srcs[r][i])) in rank order (molly_reduce_rows_bf16's arithmetic on copies that stay where they are).
 * push:   dsts[r][i] = src[i] for every r != skip.   n % 8 == 0, 16-byte aligned buffers, world <= 16.
 * flag_set: *flag = value, a system-scope release behind everything the stream has launched so far.
 * flag_wait: returns (in stream order) when flags[r][idx] >= value for every r; a peer that does not arrive within timeout_us
 *            microseconds of wall-clock time (s_memrealtime) writes 1 + r into *err instead of hanging the GPU.  *err is an int the
 *            device can write and the host can read without a synchronisation (pinned host memory in trainer/p2p.py).
 * A raised *err is fatal for the step: reduce (given err) then writes bf16 NaN instead of sums, so the gradient norm — all-reduced over
 * the ranks — is not finite anywhere and every rank skips the optimizer step; the host raises at its next call. err = NULL: no check. */
int molly_p2p_reduce_bf16(void* stream, const void* const* srcs, int world, long n, void* out, const int* err);
int molly_p2p_push_bf16(void* stream, const void* src, void* const* dsts, int world, int skip, long n);
int molly_p2p_flag_set(void* stream, int* flag, int value);
int molly_p2p_flag_wait(void* stream, const void* const* flags, int world, int idx, int value, long timeout_us, int* err);

/* ------------------------------------------------------------------------------------------------
 * layout / instruction probes (used by tests/test_gpu_kernels.py::test_probe_* to pin the gfx950 operand maps the
 * kernels rely on; not part of the product path) */
int molly_probe_mfma16(void* stream, const void* A16x32, const void* B16x32, float* D16x16);
int molly_probe_tr16(void* stream, const void* tile_in, void* lanes_out /* [64][4] u16 */, int row_stride_elems);
/* diagnostic: `blocks` workgroups that each own one CU (all of its LDS) for `us` microseconds — a stand-in for a collective's
 * kernel when timing the GEMM's launch shapes beside it (tools/diag/gemm_beside_hog.py); sink: any device word (never written) */
int molly_probe_hog(void* stream, int blocks, int us, void* sink);

#ifdef __cplusplus
}
#endif
#endif
