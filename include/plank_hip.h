/*
 * plank_hip.h -- C ABI of libplank_hip.so, the MI355X (gfx950) implementation of the
 * PlankAssembly encoder-decoder hot path.
 *
 * The reference has no FFI: its hot path (reference plankassembly/models.py) is Python glue
 * over torch.nn modules.  The drop-in boundary towards the reference's callers is therefore
 * the Python surface in plankassembly_amd/models.py (build_model / forward / state_dict,
 * SURVEY.md section 8b); THIS header is the boundary between that host code and the device
 * code.  Each entry point names the reference lines (and the torch op behind them) it
 * replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every pointer is a device pointer owned by the caller (PyTorch's caching allocator);
 *     the library never allocates, frees or synchronises; scratch is passed in explicitly;
 *   - `stream` is a hipStream_t passed as void*; calls only enqueue work on it (capturable
 *     in a hipGraph);
 *   - return value: 0 on success, a hipError_t (>0) from the launch, or a negative
 *     PA_E* code for a bad argument.  No exceptions cross the ABI;
 *   - dtypes: PA_F32 (parity path, exact-f32 MFMA) and PA_BF16 (throughput path, bf16 MFMA,
 *     f32 accumulate).  Parameters, LayerNorm statistics, losses and gradients of
 *     parameters are always f32.
 *   - one process per GPU; calls are made from that process' host thread.
 */
#ifndef PLANK_HIP_H
#define PLANK_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PA_F32 0
#define PA_BF16 1

#define PA_EINVAL (-1)   /* bad argument (null pointer, unsupported size/dtype) */
#define PA_EALIGN (-2)   /* pointer / leading dimension not 16-byte aligned where required */
#define PA_ESHAPE (-3)   /* unsupported shape (e.g. head dim) */

int pa_version(void);

/* ------------------------------------------------------------------------------------------
 * GEMM with fused epilogue:  C[b] = epi( alpha * A[b] x B[b] ),  A: M x K, B: K x N.
 * Replaces every nn.Linear / torch.bmm on the path: MHA in_proj / out_proj
 * (torch F.multi_head_attention_forward), FFN linear1/linear2 (torch transformer.py
 * _ff_block), vocab_head / pointer_head (reference models.py:145-149) and their backward
 * GEMMs (dX = dY W, dW = dY^T X).
 *   a_kcontig: 1 -> A stored [M][K] (lda = row stride); 0 -> stored [K][M]
 *   b_kcontig: 1 -> B stored [N][K] (torch Linear weight layout); 0 -> stored [K][N]
 * epilogue order: *alpha, +bias[n], relu, relu-backward gate (aux[m][n] > 0 ? v*aux_scale : 0),
 * dropout(drop_p, drop_seed), +R[m][n].  Dropout decisions are counter-based and separable: element (row = b*M+m, col = n) is
 * kept iff the low 32 bits of A[row] * C[col] are >= drop_p * 2^32, A / C = 24-bit odd hashes of (drop_seed, row) / (drop_seed,
 * col) (csrc/pa_device.h drop_keep_rc; restated in tests/dropout_masks.py linear_keep); survivors are scaled by 1/(1-p).
 * splitk > 1: contraction split into `splitk` slices, partial products go through `ws`
 * (f32, splitk*batch*M*N elements) and a reduce pass applies the epilogue.
 */
typedef struct {
    const void* A; const void* B; void* C;
    const float* bias;          /* [N] f32 or NULL */
    const void* R;              /* residual [M][ldr], dtype = out_dtype, or NULL (may alias C) */
    const void* aux;            /* relu-backward gate [M][ldaux], dtype = in_dtype, or NULL */
    void* ws;                   /* split-K workspace or NULL */
    int32_t M, N, K;
    int32_t lda, ldb, ldc, ldr, ldaux;
    int64_t sA, sB, sC, sR, sAux;   /* batch strides in elements */
    int32_t batch;
    int32_t a_kcontig, b_kcontig;
    int32_t in_dtype, out_dtype;
    float alpha;
    int32_t relu;
    float aux_scale;
    float drop_p; uint32_t drop_seed;
    int32_t splitk;
    int32_t splitk_defer;   /* 1: only write the f32 slabs to `ws`; the caller reduces them later (pa_splitk_reduce_many) */
    int64_t sBias;          /* batch stride of `bias` in elements (0: one bias for every batch member) */
    void* C_lp;             /* optional bf16 copy [M][ldc_lp] of an f32 output (out_dtype PA_F32): the next Linear's matrix operand when
                             * the residual stream is kept in f32 (bf16 greedy decode).  Launches of at most 512 rows that take the
                             * skinny kernel only (PA_EINVAL otherwise) */
    int32_t ldc_lp; int32_t pad_lp_;
} pa_gemm_args;
int pa_gemm(const pa_gemm_args* a, void* stream);
/* Leave `n` of the 256 CUs free in every persistent GEMM launch from now on (0 <= n <= 192; 0 restores the full grid): room for
 * the RCCL kernels that all-reduce finished gradient slices while the backward continues (the data-parallel exchange the
 * reference gets from Lightning's `strategy: ddp`, configs/train_complete.yaml:18).  plankassembly_amd.distributed.GradSync
 * sets it (PA_RESERVE_CUS, default 0) when the first slice of a backward is sent and clears it once the last has been waited for. */
int pa_set_reserved_cus(int32_t n);
int pa_get_reserved_cus(void);
/* Slices pa_gemm really uses for a requested `splitk` (= number of slabs it writes; <= splitk). */
int pa_gemm_effective_splitk(int32_t K, int32_t in_dtype, int32_t splitk);
/* bf16x3 ("split") mode: while `on`, every pa_gemm call with in_dtype PA_F32 - the Linears of torch's Transformer layers and of
 * the heads, reference plankassembly/models.py:60-69,85-88 (nn.TransformerEncoder/DecoderLayer's linear1/2, in_proj, out_proj;
 * vocab / pointer / switch heads) and every gradient product of their backward - runs on the bf16 matrix pipe as
 * hi*hi + hi*lo + lo*hi of the operands' bf16 hi / lo parts with f32 accumulation (~2^-17 relative per product; operands, outputs,
 * epilogues and the split-K slabs stay f32).  `ws` / `bytes`: caller-owned, 256-byte aligned scratch for the split operands of ONE
 * GEMM at a time (launches are stream-ordered): 6 bytes per operand element, e.g. 256 MB for the 16 x 1024-row headline step.
 * A GEMM whose shape the bf16 fast paths do not take (contraction length not a multiple of 64 between two k-contiguous
 * operands, rows / leading dimensions not multiples of 8 / 4, scratch too small) silently runs exact f32 instead; pa_gemm_split_stats
 * counts both.  on = 2 ("retain", the backward segments): the cut image of every k-contiguous A operand - the dY of a dX GEMM -
 * stays in the scratch buffer until the next pa_gemm_split_config call, and pa_gemm_group reuses it for the weight gradients of the
 * same segment instead of cutting dY a second time.  on = 3 (the forward of a train step): the cut image of every k-contiguous A
 * operand - the input X of a Linear - stays at the front of the buffer through the following on = 0 / on = 2 calls, until the next
 * on = 3 or on = 1 call (or until the buffer is needed: at most 3/4 of it is used this way), and the weight gradients find X already
 * cut as well; images are keyed by (address, rows, cols, leading dimension), so the caller must not rewrite a Linear input
 * between its forward GEMM and its weight gradient - which a backward pass cannot do anyway.  on = 0 switches the mode off and
 * leaves the images alone.
 * WHOSE mode: every pa_gemm_split_* / pa_attn_split_* call (and every pa_gemm / pa_attn_* / pa_layernorm_* launch) made by a host
 * thread acts on the bf16x3 CONTEXT that thread has entered - one per model: mode flags, scratch, retained images, weight cache,
 * counters (round 6; rounds 4-5 kept one process-global state).  plankassembly_amd.models.PlankModel(compute_dtype="x3") creates
 * its own context and scratch and enters it around its library calls, so models of different compute modes can step from different
 * threads / on different streams, or alternate on one.  Outside any context the calls act on a process-default context (the
 * behaviour of rounds 4-5: kernel tests and tools that never create a context). */
int pa_split_ctx_create(void** out);
void pa_split_ctx_destroy(void* ctx);
int pa_split_ctx_enter(void* ctx);                            /* this host thread: ctx until the next enter; NULL = leave */
void* pa_split_ctx_current(void);
int pa_gemm_split_config(int32_t on, void* ws, int64_t bytes);
int pa_gemm_split_active(void);                               /* 1 while the mode is on */
int64_t pa_gemm_split_reused(void);                           /* weight-gradient operands found already cut (on = 2 / 3) since the last stats reset */
int pa_gemm_split_stats(int64_t* out2, int32_t reset);        /* out2[0] GEMMs run as bf16x3, out2[1] asked but run exact */
/* Images written by the PRODUCER of an operand (bf16x3 mode, retain modes 2 / 3).  pa_gemm_split_reserve: the kernel about to write
 * the f32 matrix `src` ([rows][cols], leading dimension ld, cols % 8 == 0) also writes its cut image - [rows][3 cols] bf16, parts
 * pattern *pat: 0 = (hi, hi, lo), 1 = (hi, lo, hi), hi = bf16(x), lo = bf16(x - hi) - to the returned address, and the pa_gemm call
 * that consumes `src` as its k-contiguous A operand (and, in a backward segment, the grouped weight-gradient launch) skips its own
 * cut.  NULL: no image wanted (mode off, no retain mode, no room); the consumer then cuts as before.  Used by
 * pa_layernorm_fwd_img (csrc/runtime.hip Ctx::ln_fwd). */
void* pa_gemm_split_reserve(const void* src, int32_t rows, int32_t cols, int32_t ld, int32_t* pat);
int64_t pa_gemm_split_made_hits(void);                        /* A operands found written by their producer since the last stats reset */
/* Images of the constant operands (the weights), owned by one model: a cache over a caller-owned device buffer (256-byte aligned;
 * 12 bytes per parameter + 32 KiB holds W and W^T of every Linear).  While a cache is in use (pa_gemm_split_cache_use, inside the
 * model's own pa_gemm_split_config bracket) every unbatched k-contiguous B operand of a bf16x3 GEMM is looked up in it; a miss is
 * cut INTO the cache (the first step learns the list).  pa_gemm_split_cache_refresh cuts every image again from its source in one
 * launch: call it whenever the parameters (or their transposed copies) changed.  The sources must stay alive and in place for as
 * long as the cache exists; destroy it before freeing them. */
int pa_gemm_split_cache_create(void* buf, int64_t bytes, void** out);
void pa_gemm_split_cache_destroy(void* cache);
int pa_gemm_split_cache_use(void* cache);                     /* NULL: none */
int32_t pa_gemm_split_cache_entries(void* cache);
int pa_gemm_split_cache_refresh(void* cache, void* stream);
int64_t pa_gemm_split_cache_hits(void);
/* Measurement hook (bench.py's roofline census; no reference counterpart): pa_gemm_record(1) starts appending every
 * pa_gemm() argument block to a host-side list; pa_gemm_record(0) returns the count so far; pa_gemm_recorded() copies
 * up to `cap` recorded blocks out and stops recording.  Replaying the blocks re-launches the same GEMMs. */
int pa_gemm_record(int32_t enable);
int pa_gemm_recorded(pa_gemm_args* out, int32_t cap);
/* Kernel each recorded launch was dispatched to (valid until the next pa_gemm_record(1)):
 * PA_GEMM_KIND_RING = gemm3_kernel (one block per CU, 4-stage LDS ring; single-round bf16 launches),
 * PA_GEMM_KIND_PAIR = gemm_kernel (two blocks per CU; everything else). */
#define PA_GEMM_KIND_PAIR 0
#define PA_GEMM_KIND_RING 1
#define PA_GEMM_KIND_WIDE 2   /* gemm3w_kernel: 128 x 256 tiles for the large multi-round Linears (opt-in); 192 x 128 ("tall") tiles for N <= 512 Linears a few 128 x 128 tiles past one round of the CUs */
#define PA_GEMM_KIND_SMALL 3  /* gemm3s_kernel: 64 x 64 tiles for launches that cover at most half of the CUs */
#define PA_GEMM_KIND_SKINNY 4 /* gemm_skinny_kernel: <= 512 rows against a whole weight (greedy decode), 32 x 32 tiles, K resident */
#define PA_GEMM_KIND_BIG 5    /* gemm8_kernel: eight waves on 256 x 256 / 256 x 128 tiles (plain k-contiguous Linears of many rows) */
int pa_gemm_recorded_kinds(int32_t* out, int32_t cap);
/* -1 for launches made by pa_gemm; members of one pa_gemm_group launch share an id >= 0 (consecutive entries). */
int pa_gemm_recorded_groups(int32_t* out, int32_t cap);

/* Linear + bias (+ dropout) + residual + LayerNorm in one launch, for the post-norm sublayer tails of the reference's
 * encoder / decoder layers (torch nn/modules/transformer.py `x = norm(x + dropout(sublayer(x)))`, used by
 * plankassembly/models.py:76-112 through nn.TransformerEncoderLayer / nn.TransformerDecoderLayer):
 *     Z[M][512] = R + drop(A[M][K] W[512][K]^T + bias),   Y = LayerNorm(Z; gamma, beta, eps),   mean / rstd per row.
 * bf16 operands and outputs, f32 bias / gamma / beta / statistics; N must be 512, K a multiple of 64, rows 16-byte aligned.
 * Z, R, bias, mean, rstd may be NULL (Z is what pa_layernorm_bwd needs; the greedy-decode step does not keep it).
 * Bit-identical to pa_gemm (same arguments) followed by pa_layernorm_fwd.  One block per 32 rows: meant for
 * M <= pa_gemm_ln_max_rows() (one round of blocks); beyond that the two separate launches are faster. */
typedef struct {
    const void* A; const void* W; const float* bias; const void* R;
    void* Z; void* Y; const float* gamma; const float* beta; float* mean; float* rstd;
    int32_t M, N, K, lda, ldw, ldr, ldz, ldy;
    float eps, drop_p; uint32_t drop_seed; int32_t pad_;
} pa_gemm_ln_args;
int pa_gemm_ln(const pa_gemm_ln_args* a, void* stream);
int pa_gemm_ln_max_rows(void);

/* Linear on LayerNorm(Z) without a LayerNorm launch - for the post-norm chains of the greedy-decode step, where a sublayer's
 * output `x = norm(z)` (torch nn/modules/transformer.py, used by plankassembly/models.py:293-294 once per generated token) feeds the
 * next Linear and the next residual add and a LayerNorm on B rows is pure launch latency.  With y = (z - mean) rstd gamma + beta:
 *     y W^T + b  =  rstd (z (W gamma)^T - mean u) + v,     u[n] = sum_k W[n][k] gamma[k],   v[n] = b[n] + sum_k W[n][k] beta[k].
 * pa_ln_fold_weights prepares Wf = bf16(W gamma) (from the f32 master weight), u (summed over the ROUNDED Wf) and v once per
 * decode; pa_gemm_norm_a runs C = epi(rstd (A Wf^T - mean u) + v) on the raw rows A = Z (bf16, k-contiguous, K % 64 == 0,
 * N % 32 == 0; `args->B` = Wf, `args->bias` = v; no residual / gate / dropout / split-K), every block computing the row
 * statistics of its own 64 rows, and - when `y` is given - writes LayerNorm(Z) [M][K] for the later residual add.
 * 64 x 64 tiles, at most two per CU (PA_ESHAPE beyond). */
typedef struct {
    const float* u; const float* gamma; const float* beta;
    void* y; int32_t ldy; float eps;
    /* f32 residual stream (optional; <= 512 rows, K = 512 only - PA_ESHAPE otherwise): `zf` = the f32 rows Z [M][ldzf] of which
     * `args->A` is the bf16 copy.  The row statistics and the materialised LayerNorm(Z) are then computed from zf, and `y` is
     * written as f32 when y_f32 != 0. */
    const float* zf; int32_t ldzf; int32_t y_f32;
} pa_gemm_norm_ext;
int pa_ln_fold_weights(void* Wf, float* u, float* v, const float* W, const float* bias, const float* gamma, const float* beta,
                       int32_t N, int32_t K, void* stream);
int pa_gemm_norm_a(const pa_gemm_args* args, const pa_gemm_norm_ext* ext, void* stream);
/* The same fold in exact f32 (the f32 greedy-decode step, round 4): pa_ln_fold_weights_f32 writes Wf = W gamma as f32; pa_gemm_norm_a
 * with in_dtype = out_dtype = PA_F32 takes f32 rows A (ext->zf = the same rows: the statistics source), Wf, u, v and writes f32.
 * At most 512 rows, K = 512 (PA_ESHAPE otherwise). */
int pa_ln_fold_weights_f32(float* Wf, float* u, float* v, const float* W, const float* bias, const float* gamma, const float* beta,
                           int32_t N, int32_t K, void* stream);

/* Several weight-gradient GEMMs (dW = dY^T X: bf16 operands, contraction index strided in both, f32 output, no
 * epilogue, batch 1; splitk > 1 only with splitk_defer) in one launch of the ring kernel: its unit stream runs through
 * all members.  PA_EINVAL when a member does not qualify - the caller then launches them one by one with pa_gemm. */
#define PA_MAX_GROUP 8
int pa_gemm_group(const pa_gemm_args* args, int32_t n, void* stream);

/* Column sums of several matrices in one launch: out[n] += sum_m X[m][n] (f32 atomics; outputs are accumulated
 * into).  The backward pass queues the bias gradients of a segment (dY buffers stay live until its end) and sums
 * them together.  Rows must be 16-byte aligned vectors (PA_EALIGN otherwise - use pa_colsum). */
#define PA_MAX_COLSUM 8
typedef struct {
    const void* X; float* out;
    int32_t M, N, ldx, pad_;
} pa_colsum_desc;
int pa_colsum_many(const pa_colsum_desc* descs, int32_t n_desc, int32_t dtype, void* stream);

/* Deferred split-K reduction for plain f32 outputs (weight gradients): out[m][n] = sum_s ws[s][m][n], one launch for
 * up to PA_MAX_REDUCE launches of pa_gemm(splitk_defer = 1).  The backward pass queues every dW of a segment and
 * reduces them together (13 launches per step instead of 69).  No reference counterpart (torch accumulates dW
 * inside its GEMM). */
#define PA_MAX_REDUCE 16
typedef struct {
    const float* ws; float* out;
    int32_t rows, cols, ld_out, splitk;
} pa_reduce_desc;
int pa_splitk_reduce_many(const pa_reduce_desc* descs, int32_t n_desc, void* stream);

/* Batched 2-D transposes dst[c][r] = src[r][c] (one launch for a table of matrices; descriptors live in device
 * memory, tile_begin = prefix sum of ceil(rows/64)*ceil(cols/64)).  Keeps the transposed shadow of the Linear
 * weights current so the backward GEMM dX = dY W runs with both operands k-contiguous. */
typedef struct {
    const void* src; void* dst;
    int32_t rows, cols, ld_src, ld_dst;
    int32_t tile_begin, pad_;
} pa_tr_desc;
int pa_transpose_many(const pa_tr_desc* descs_dev, int32_t n_desc, int32_t total_tiles, int32_t dtype, void* stream);

/* column sums: out[n] (+)= sum_m X[m][n]  (bias gradients).  f32 out. `partial` is scratch of
 * pa_colsum_ws_floats(M,N) floats. */
int64_t pa_colsum_ws_floats(int32_t M, int32_t N);
int pa_colsum(const void* X, int32_t dtype, int32_t M, int32_t N, int32_t ldx, float* out,
              int32_t accumulate, float* partial, void* stream);

/* ------------------------------------------------------------------------------------------
 * Embeddings.
 * pa_embed_input_fwd: reference models.py:103-112 (_embed_input): out[t] = sum_k table_k[idx_k[t]].
 *   tables: up to 5 f32 tables [rows_k][d]; idx: int64 [n_tok] each; a NULL idx skips the table
 *   (sideface batches have no input_type).
 * pa_embed_output_fwd: reference models.py:114-138 (_embed_output): out[b][0] = 0,
 *   out[b][t] = value[tok[b][t-1]] + coord[(t-1) % dof] + pos[(t-1) / dof], t in [1, T).
 *   tok has row stride tok_ld (the reference passes output_value[:, :-1]).
 * pa_embed_*_bwd: scatter-add of d_out into the f32 table gradients (atomic; tables with <= 8 rows, whose
 *   row count the caller passes in table_rows (host array), are pre-reduced per block in LDS).
 */
int pa_embed_input_fwd(void* out, int32_t out_dtype, const float* const* tables, const int64_t* const* idx,
                       const int32_t* rowmap, int32_t n_tables, int64_t n_tok, int32_t d, void* stream);
int pa_embed_input_bwd(const void* dout, int32_t dtype, float* const* dtables, const int64_t* const* idx,
                       const int32_t* rowmap, const int32_t* table_rows, int32_t n_tables, int64_t n_tok, int32_t d,
                       void* stream);
/* Embedding-table gradients by sorted segments: the token rows are grouped by table row once per batch - order[k] =
 * row indices (into dout) sorted by id, seg[k][r] .. seg[k][r+1] = the rows that use table row r (int32, seg has
 * table_rows[k] + 1 entries) - and each table row sums its segment: dtable[r] += sum (tables with few rows split
 * their long segments over several blocks and combine with one atomic per column; in larger tables a row used by more
 * than 64 tokens is split over 8 blocks the same way).  n_rows = rows of dout (sizes the
 * splitting).  The grouping depends only on the batch, so it is built when the batch is prepared
 * (PlankModel.prepare_batch); without it the atomic scatter-add kernels (pa_embed_input_bwd / _output_bwd) run. */
#define PA_MAX_SEG_TABLES 5
int pa_embed_segment_bwd(const void* dout, int32_t dtype, float* const* dtables, const int32_t* const* order,
                         const int32_t* const* seg, const int32_t* table_rows, int32_t n_tables, int64_t n_rows,
                         int32_t d, void* stream);
/* Row packing ("unpadding"): mask uint8 [B][S] (1 = PAD) -> cu int32 [2B+1] (cu[b] = #valid rows before batch element
 * b, cu[B] = total; entries B+1..2B = the batch elements by descending row count, the dispatch order of the
 * variable-length attention launches, pa_attn_args.order) and rowmap int32 [B*S] (rowmap[packed row] = b*S + s).  Padded
 * encoder positions never reach the loss (they are masked as keys everywhere), so the encoder stack can run on the
 * packed rows only; rowmap (optional, NULL = identity) lets the embedding kernels gather / scatter packed rows. */
int pa_pack_rows(const uint8_t* mask, int32_t B, int32_t S, int32_t* cu, int32_t* rowmap, void* stream);

/* Token rows grouped by embedding-table row (the `order` / `seg` inputs of pa_embed_segment_bwd) for up to
 * PA_MAX_GROUP_TABLES tables in ONE launch (one block per table; stable counting sort, so segment sums add in token order).
 * Batch-only information like pa_pack_rows: belongs where the batch is built (the reference builds its batches in the
 * dataloader, plankassembly/datasets/line_data.py:34-109; the gradients these groupings serve are those of
 * models.py:103-138).  kind 0: an input table - entry i (0 <= i < n) uses table row idx[rowmap ? rowmap[i] : i] and
 * reads gradient row i (the packed encoder rows).  kinds 1 / 2 / 3: the decoder's value / coordinate / position tables -
 * entry i = b * (T-1) + t1 reads gradient row b*T + t1 + 1 (decoder row t embeds token t-1) and uses table row
 * idx[b * tok_ld + t1] / t1 % dof / t1 / dof; n must be B * (T-1).
 * Outputs: order int32 [n] (gradient rows sorted by table row, ties in entry order), seg int32 [rows + 1]. */
#define PA_MAX_GROUP_TABLES 8
typedef struct {
    const int64_t* idx; const int32_t* rowmap;
    int32_t n, rows, kind, T, dof, tok_ld;
    int32_t* order; int32_t* seg;
} pa_group_desc;
int pa_group_rows(const pa_group_desc* descs, int32_t n_tables, void* stream);
int pa_embed_output_fwd(void* out, int32_t out_dtype, const float* value, const float* coord, const float* pos,
                        const int64_t* tok, int32_t tok_ld, int32_t B, int32_t T, int32_t d, int32_t dof,
                        void* stream);
int pa_embed_output_bwd(const void* dout, int32_t dtype, float* dvalue, float* dcoord, float* dpos,
                        const int64_t* tok, int32_t tok_ld, int32_t B, int32_t T, int32_t d, int32_t dof,
                        void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the last dim (biased variance), torch nn.LayerNorm as used by
 * TransformerEncoderLayer/DecoderLayer post-norm branch (torch nn/modules/transformer.py) with
 * eps = 1.0 per layer (reference models.py:60-61,66-67: normalize_before lands in the eps slot)
 * and 1e-5 for encoder.norm / decoder.norm (models.py:62,68).
 * fwd: y = (z - mean) * rstd * gamma + beta; saves mean/rstd (f32 [rows]).
 * bwd: dz from dy; dgamma/dbeta are ACCUMULATED into f32 outputs through `partial` scratch
 *      (pa_layernorm_ws_floats(rows, d) floats).  If drop_p > 0, also writes
 *      ddrop = dz * dropout_mask(seed, row*d+col) / (1-p)  (gradient of the sub-layer output
 *      that went through dropout before the residual add).  dzsum (optional) accumulates the
 *      column sums of that sub-layer-output gradient (ddrop if drop_p > 0, else dz) = the bias
 *      gradient of the Linear that produced it.
 */
int64_t pa_layernorm_ws_floats(int64_t rows, int32_t d);
int pa_layernorm_fwd(void* y, const void* z, const float* gamma, const float* beta, float* mean, float* rstd,
                     int64_t rows, int32_t d, float eps, int32_t dtype, void* stream);
int pa_layernorm_bwd(void* dz, void* ddrop, const void* dy, const void* z, const float* gamma,
                     const float* mean, const float* rstd, float* dgamma, float* dbeta, float* dzsum,
                     float* partial, int64_t rows, int32_t d, int32_t dtype,
                     float drop_p, uint32_t drop_seed, void* stream);
/* The same in two steps, so that several LayerNorm backward passes share ONE finishing launch: _partial writes only the
 * per-block partial sums (pa_layernorm_bwd_nparts(rows) blocks x 3 x d floats, a distinct `partial` buffer per pass until
 * finished); pa_layernorm_finish_many adds them to dgamma / dbeta / dzsum (dzsum may be NULL). */
int pa_layernorm_bwd_partial(void* dz, void* ddrop, const void* dy, const void* z, const float* gamma,
                             const float* mean, const float* rstd, int32_t want_dzsum, float* partial,
                             int64_t rows, int32_t d, int32_t dtype, float drop_p, uint32_t drop_seed, void* stream);
/* pa_layernorm_fwd that also writes the bf16x3 image of y (f32 only; img from pa_gemm_split_reserve, NULL = none) */
int pa_layernorm_fwd_img(void* y, const void* z, const float* gamma, const float* beta, float* mean, float* rstd,
                         int64_t rows, int32_t d, float eps, int32_t dtype, void* img, int32_t img_pat, void* stream);
/* pa_layernorm_bwd_partial that also writes the bf16x3 image of its output (ddrop, or dz when drop_p == 0) for the dX GEMM that
 * consumes it; only where pa_layernorm_bwd_can_img(d, dtype) says so (f32, d = 512). */
int pa_layernorm_bwd_can_img(int32_t d, int32_t dtype);
int pa_layernorm_bwd_partial_img(void* dz, void* ddrop, const void* dy, const void* z, const float* gamma,
                                 const float* mean, const float* rstd, int32_t want_dzsum, float* partial,
                                 int64_t rows, int32_t d, int32_t dtype, float drop_p, uint32_t drop_seed,
                                 void* img, int32_t img_pat, void* stream);
int32_t pa_layernorm_bwd_nparts(int64_t rows);
#define PA_MAX_LN_FINISH 4
typedef struct {
    const float* partial; float* dgamma; float* dbeta; float* dzsum;
    int32_t nparts, pad_;
} pa_ln_finish_desc;
int pa_layernorm_finish_many(const pa_ln_finish_desc* descs, int32_t n, int32_t d, void* stream);
/* The three batched end-of-segment reductions (pa_layernorm_finish_many, pa_colsum_many, pa_splitk_reduce_many) in
 * ONE launch: they are independent of each other, their blocks are concatenated.  Any of the three lists may be empty. */
int pa_segment_tail(const pa_ln_finish_desc* ln, int32_t n_ln, int32_t d_model, const pa_colsum_desc* cs, int32_t n_cs,
                    int32_t dtype, const pa_reduce_desc* rd, int32_t n_rd, void* stream);


/* ------------------------------------------------------------------------------------------
 * Multi-head attention core: softmax(Q K^T * scale + mask) V per (batch, head), flash-style
 * (scores never reach HBM).  Replaces torch F.multi_head_attention_forward's
 * bmm/softmax/dropout/bmm (torch 1.10) resp. scaled_dot_product_attention (torch 2.x) for the
 * encoder self-attention (key padding mask), decoder self-attention (causal + key padding,
 * reference models.py:85-89,209-214) and decoder cross-attention (memory key padding).
 *   Q/K/V/O: element (b, l, h, c) at ptr[(b*L + l)*ld + h*dh + c]  (head = contiguous dh slice,
 *   so Q/K/V may be views into the packed in_proj output);
 *   kpm: uint8 [B][Lk], 1 = masked key (PAD), or NULL;  causal: key j allowed iff j <= i;
 *   lse: f32 [B][H][Lq] log-sum-exp of the scaled, masked scores (saved for backward);
 *   dropout on the attention probabilities (torch MHA `dropout`): counter-based and separable - probability (row, key) with
 *   row = (b*H+h)*Lq + i is kept iff the low 32 bits of A[row] * C[key] are >= drop_p * 2^32 (csrc/pa_device.h drop_keep2, the
 *   same function the Linear epilogues use; tests/dropout_masks.py attn_keep); survivors are scaled by 1/(1-p); the backward
 *   kernels regenerate the same decisions.
 * bwd: dq/dk/dv have the layouts of q/k/v; delta is f32 scratch [B][H][Lq].
 */
typedef struct {
    const void* q; const void* k; const void* v; void* o;
    float* lse;
    const uint8_t* kpm;
    int32_t B, H, Lq, Lk, dh;
    int32_t ldq, ldk, ldv, ldo;
    int32_t causal;
    float scale;
    float drop_p; uint32_t drop_seed;
    int32_t dtype;
    /* backward only */
    const void* dout; void* dq; void* dk; void* dv; float* delta;
    int32_t lddo, lddq, lddk, lddv;
    /* variable-length ("unpadded") batches: int32 [B+1] row offsets of the packed Q resp. K/V rows of each batch
     * element, or NULL for the dense [B][L] layout.  With offsets, Lq/Lk are the per-batch maxima (grid size and
     * layout of lse/delta [B][H][Lq]); no padding mask is needed because padded tokens are simply not there. */
    const int32_t* cu_q; const int32_t* cu_k;
    /* optional dispatch order: int32 [B], a permutation of the batch elements by descending length (pa_pack_rows
     * leaves it in cu[B+1 .. 2B]).  All blocks of a launch start together, a few per CU, so a variable-length launch
     * lasts as long as the CU that drew the longest elements; in this order every CU gets a long, a medium and a short
     * block.  Results do not depend on it.  NULL = batch order. */
    const int32_t* order;
    /* optional scratch (device, 256-byte aligned; NULL = none) for packed bf16 self-attention launches (cu_q == cu_k, order given,
     * H = 8, dh = 64): the key range of a long batch element - resp. its query range in the dK / dV launch - is then cut across
     * several blocks whose partial results meet in this buffer (csrc/attention.hip decode_unit_split), so that the launch lasts
     * as long as its work instead of as long as its longest element's serial chain.  OPT-IN (PA_ATTN_SPLIT=1): on MI355X every
     * extra block costs ~6 us of slot time (lookup, first tile, publish + merge) and the split launches measured 15-20 % SLOWER
     * than one block per tile (profiles/r06_attention_launch_shape.txt).  Size: pa_attn_ws_bytes() (0 while the split is off).  Contract: its
     * first pa_attn_ws_ticket_bytes(ws_bytes) bytes are ZERO before the first launch that uses the buffer; every launch leaves them
     * zero.  Launches that share a buffer must be ordered on one stream.  Results do not depend on it beyond f32 summation order. */
    void* ws; int64_t ws_bytes;
} pa_attn_args;
int pa_attn_fwd(const pa_attn_args* a, void* stream);
int pa_attn_bwd(const pa_attn_args* a, void* stream);
/* scratch for pa_attn_args.ws: rows_total packed rows of B batch elements, the longest L_max rows (0 bytes when no launch of that
 * shape would use it: split not enabled - PA_ATTN_SPLIT=1 -, H != 8, or L_max short enough for one block per tile) */
int64_t pa_attn_ws_bytes(int32_t rows_total, int32_t B, int32_t H, int32_t L_max);
int64_t pa_attn_ws_ticket_bytes(int64_t ws_bytes);
/* bf16x3 ("split") attention, the companion of pa_gemm_split_config: while `on`, f32 launches with dh = 64 compute every matrix
 * product of torch's F.multi_head_attention_forward (reference plankassembly/models.py:60-69: S = Q K^T, O = P V) and of its
 * backward as hi*hi + hi*lo + lo*hi of the operands' bf16 hi / lo parts with f32 accumulation (csrc/attention_x3.h); inputs,
 * outputs, softmax statistics, masks, lse / delta and dropout decisions are the exact-f32 kernels'.  Other head sizes run exact.
 * pa_attn_split_taken: launches (forward or backward) that ran split since the last reset. */
int pa_attn_split_config(int32_t on);
int64_t pa_attn_split_taken(int32_t reset);

/* ------------------------------------------------------------------------------------------
 * Output heads + mixture NLL (training): reference models.py:140-166,186 (_create_dist training
 * branch) and 219-227 (nll_loss(ignore_index=PAD), argmax accuracy).
 * Inputs: vocab logits [rows][ldv] f32 (rows = B*T), pointer logits [B][T][T] f32 (already
 * scaled by 1/d_model), switch logit [rows] f32, labels int64 [rows] in [0, V+T).
 * fwd accumulates stats[0] += sum of -logp(label) over non-PAD rows, stats[1] += #non-PAD rows,
 * stats[2] += #correct argmax; saves per-row (lse_vocab, lse_ptr) for backward.
 * The training mask quirk is reproduced: pointer logits j >= i are REPLACED by the value 1e-6
 * and stay in the softmax.
 * bwd writes d(vocab logits), d(pointer logits) (zero where j >= i) in `out_dtype` (they feed the
 * backward GEMMs) and d(switch logit) in f32, for loss = stats[0] / stats[1]; the upstream
 * gradient is gscale * stats[3] (stats[3] is device resident so autograd's grad_output never
 * needs a host sync).
 * pa_switch_fwd: s[row] = h[row] . w + b (reference models.py:153), pa_switch_bwd its gradient
 * (dh += ds * w fused into the caller's GEMM epilogue is not possible, so dh_out is written
 * and dw/db accumulated).  `partial`: scratch of pa_layernorm_bwd_nparts(rows) x 2 x d floats.
 */
/* ACTIVATION: gelu (reference models.py:60-61,66-67 -> torch F.gelu, exact erf form).  The GEMM epilogues are ReLU-only; in GELU
 * mode the FFN's first Linear writes its pre-activation and these two element-wise launches do the rest:
 *   pa_gelu_fwd: out[r][c] = keep(seed, r, c) ? gelu(pre[r][c]) / (1 - p) : 0        (may run in place)
 *   pa_gelu_bwd: dpre[r][c] = dh[r][c] * gelu'(pre[r][c]) * (keep(seed, r, c) ? 1 / (1 - p) : 0)   (dpre may alias dh)
 * keep = pa_gemm's Linear-output dropout decision for (row r, column c) (csrc/pa_device.h drop_keep_rc), so a step under dropout takes
 * the same decisions as the fused ReLU epilogue would.  rows x cols elements, row stride ld, cols % 4 == 0, 8 / 16-byte aligned rows. */
int pa_gelu_fwd(void* out, const void* pre, int64_t rows, int32_t cols, int32_t ld, int32_t dtype, float drop_p, uint32_t drop_seed,
                void* stream);
int pa_gelu_bwd(void* dpre, const void* dh, const void* pre, int64_t rows, int32_t cols, int32_t ld, int32_t dtype, float drop_p,
                uint32_t drop_seed, void* stream);
int pa_switch_fwd(float* s, const void* h, int32_t dtype, const float* w, const float* b, int64_t rows,
                  int32_t d, void* stream);
int pa_switch_bwd(void* dh, int32_t accumulate, float* dw, float* db, const float* ds, const void* h,
                  int32_t dtype, const float* w, float* partial, int64_t rows, int32_t d, void* stream);
int pa_mixture_nll_fwd(float* stats, float* row_lse, const float* vocab, int32_t ldv, const float* ptr,
                       const float* sw, const int64_t* label, int32_t B, int32_t T, int32_t V, int32_t pad,
                       void* stream);
int pa_mixture_nll_bwd(void* dvocab, void* dptr, int32_t out_dtype, float* dsw, const float* stats,
                       const float* row_lse, const float* vocab, int32_t ldv, const float* ptr, const float* sw,
                       const int64_t* label, int32_t B, int32_t T, int32_t V, int32_t pad, float gscale,
                       void* stream);
/* The same forward on an EIGHT-float statistics block that the launch zeroes itself: [0] summed NLL, [1] unmasked rows, [2] correct
 * rows, [3] upstream d(loss) (armed with 1.0), [4] loss = [0] / [1], [5] accuracy = [2] / ([1] + 1e-10) - reference
 * plankassembly/models.py:226-231 - written by a one-wave finishing launch behind the forward kernel, [6], [7] spare.  The training step returns
 * views of [4] / [5]: no element-wise launches behind the forward.  pa_mixture_nll_bwd_up: `upstream` (device f32 scalar, or NULL =
 * stats[3]) is d(loss) from the caller's autograd - read in place instead of being copied into stats[3]. */
int pa_mixture_nll_fwd_fin(float* stats8, float* row_lse, const float* vocab, int32_t ldv, const float* ptr,
                           const float* sw, const int64_t* label, int32_t B, int32_t T, int32_t V, int32_t pad,
                           void* stream);
int pa_mixture_nll_bwd_up(void* dvocab, void* dptr, int32_t out_dtype, float* dsw, const float* stats,
                       const float* row_lse, const float* vocab, int32_t ldv, const float* ptr, const float* sw,
                       const int64_t* label, int32_t B, int32_t T, int32_t V, int32_t pad, float gscale,
                       const float* upstream, void* stream);

/* ------------------------------------------------------------------------------------------
 * Adam (torch.optim.Adam defaults; reference trainer_complete.py:127-129), fused over one flat
 * parameter buffer.  Optionally refreshes the bf16 shadow copy of the parameters.
 * `gscale` multiplies the gradient first (1/world_size for a summed all-reduce).
 */
int pa_adam_step(float* p, const float* g, float* m, float* v, void* p_bf16, int64_t n, float lr, float b1,
                 float b2, float eps, int32_t step, float gscale, void* stream);
int pa_cast(void* dst, int32_t dst_dtype, const void* src, int32_t src_dtype, int64_t n, void* stream);
/* One-GPU rehearsal of the data-parallel gradient exchange (the reference's `strategy: ddp`, configs/train_complete.yaml:18-21):
 * a stand-in for one ring all-reduce.  `blocks` workgroups (<= 256) stay resident for at least `min_us` microseconds and stream
 * `buf` (16-byte aligned, `bytes` long; contents unchanged) through HBM at least `passes` times.  Launched on a side stream by
 * plankassembly_amd.distributed.GradSync when PLANK_FAKE_COLLECTIVE is set and the process group has one rank, so that the CU
 * reservation of the persistent GEMM grids (pa_set_reserved_cus) can be tuned without an 8-GPU node. */
int pa_fake_collective(void* buf, int64_t bytes, int32_t blocks, int32_t passes, float min_us, void* stream);

/* ------------------------------------------------------------------------------------------
 * Model-level runtime: one call enqueues the whole training forward of reference
 * plankassembly/models.py:190-233 (train_step: _embed_input, encoder, _embed_output, decoder,
 * _create_dist, nll_loss, accuracy) resp. its backward, on the caller's stream, using only the
 * caller's workspace.  `pa_model` is a small HOST-side object (configuration, parameter pointer
 * tables, workspace layout); it owns no device memory.
 *
 * Parameters are bound as three pointer tables in the canonical order of the reference
 * state_dict (plankassembly_amd/models.py PARAM_ORDER):
 *   params_f32 : f32 master parameters (embedding tables, biases, LayerNorm affine are read here)
 *   params_lp  : the same tensors in the compute dtype (GEMM B operands); = params_f32 for PA_F32
 *   params_lpT : (pa_model_bind_transposed, optional) transposed copies W^T [in][out] of the 2-D Linear weights
 *                at the same table positions (NULL entries = not available): dX = dY W then runs as a
 *                k-contiguous GEMM against W^T.  The caller refreshes them after each optimizer step
 *                (pa_transpose_many).
 *   grads      : f32 gradients (may be NULL for inference).  Gradients that are produced by
 *                accumulation (embedding tables, LayerNorm affine, out_proj/linear2 biases, switch
 *                head) are ADDED to: the caller zeroes the gradient buffer before backward.
 * The torch argument-order slip of the reference is explicit here: eps_layer = float(NORMALIZE_BEFORE)
 * (1.0), eps_final = 1e-5, has_enc_norm = bool(NORMALIZE_BEFORE), layers are post-norm.
 */
typedef struct pa_model pa_model;
typedef struct {
    int32_t d_model, n_head, d_ff, n_enc, n_dec, vocab;
    int32_t out_dof;
    int32_t in_table_rows[5];      /* rows of input_value / input_pos / input_coord / input_view / input_type */
    float eps_layer, eps_final;
    int32_t has_enc_norm;
    float dropout;
    int32_t pad, end;
    int32_t dtype;
    int32_t activation;            /* FFN activation: 1 ReLU (every shipped config), 2 GELU (torch's exact erf form); the reference hands
                                    * cfg.MODEL.ACTIVATION to nn.TransformerEncoderLayer / DecoderLayer (models.py:60-61,66-67).  0 = 1. */
} pa_model_cfg;
typedef struct {
    const int64_t* input_idx[5];   /* input_value, input_pos, input_coord, input_view, input_type (NULL ok) : [B][S] */
    const uint8_t* input_mask;     /* [B][S], 1 = PAD */
    const int64_t* output_value;   /* [B][T]  (NULL: run the encoder only) */
    const int64_t* output_label;   /* [B][T] */
    const uint8_t* output_mask;    /* [B][T] */
    int32_t B, S, T;
    /* optional packed-encoder mode: cu_in (int32 [2B+1]) and rowmap exactly as pa_pack_rows wrote them (the runtime also
     * reads the dispatch order it leaves in cu_in[B+1 .. 2B]), or cu_in == NULL for the dense path.  n_valid is the
     * host copy of cu_in[B] (the one device->host read of a training step). */
    const int32_t* cu_in; const int32_t* rowmap; int32_t n_valid;
    /* optional grouping of the token rows by table row (pa_embed_segment_bwd; NULL = atomic scatter-add kernels):
     * in_* for the five input tables over the (packed) encoder rows; out_* for the output embedding's value / coord /
     * pos tables over the B*T decoder rows (row (b,t) uses token t-1; rows with t = 0 are left out) */
    const int32_t* in_order[5]; const int32_t* in_seg[5];
    const int32_t* out_order[3]; const int32_t* out_seg[3];
} pa_batch;

#define PA_T_MEMORY 0
#define PA_T_HIDDENS 1
#define PA_T_VOCAB_LOGITS 2
#define PA_T_PTR_LOGITS 3
/* FFN hidden activation relu(linear1(x)) (after its dropout, when training with dropout) of encoder layer l / decoder layer l:
 * [encoder rows processed (packed: valid rows)][d_ff] resp. [B*T][d_ff], compute dtype.  Its sign pattern (> 0) is the ReLU
 * branch the backward pass differentiates (torch transformer.py _ff_block, reference models.py:60-61,66-67 activation=relu);
 * the parity tests evaluate the float64 oracle on exactly these branches. */
#define PA_T_ENC_FFN(l) (16 + (l))
#define PA_T_DEC_FFN(l) (80 + (l))

int pa_model_create(const pa_model_cfg* cfg, pa_model** out);
void pa_model_destroy(pa_model* m);
int pa_model_num_params(const pa_model* m);
int pa_model_bind(pa_model* m, void* const* params_f32, void* const* params_lp, void* const* grads);
int pa_model_bind_transposed(pa_model* m, void* const* params_lpT);
/* Optional (bf16): the cross-attention K/V rows of every decoder layer's in_proj_weight, transposed and packed side by
 * side: kvT_all[k][l * 2d + n] = in_proj_weight_l[d + n][k]  ([d][n_dec * 2d], low-precision dtype).  With it bound the
 * backward pass forms d(memory) with ONE GEMM over all layers instead of n_dec accumulating ones.  NULL unbinds. */
int pa_model_bind_cross_kv_t(pa_model* m, const void* kvT_all);
int64_t pa_model_train_ws_bytes(pa_model* m, int32_t B, int32_t S, int32_t T);
/* stats (device, f32[PA_MODEL_STATS_FLOATS] = f32[8]; the forward zeroes all eight itself - pa_mixture_nll_fwd_fin):
 * [0] sum of -log p(label) over non-PAD labels, [1] #non-PAD, [2] #correct.
 * loss = stats[0]/stats[1] (reference models.py:221), accuracy = stats[2]/(stats[1]+1e-10) (:227);
 * [3] upstream gradient d(objective)/d(loss), initialised to 1 by the forward and read (on the
 * device, no host sync) by the backward; [4] loss, [5] accuracy (written by the forward's last launch), [6], [7] spare.
 * A caller that sized the block for the four-float layout of rounds 1-4 gets a 16-byte out-of-bounds write: size it with
 * pa_model_stats_floats().
 * `seed` keys this step's dropout masks (training != 0 and cfg.dropout > 0). */
#define PA_MODEL_STATS_FLOATS 8
int32_t pa_model_stats_floats(void);                           /* = PA_MODEL_STATS_FLOATS of the loaded library */
int pa_model_train_fwd(pa_model* m, const pa_batch* batch, void* ws, int64_t ws_bytes, uint32_t seed,
                       int32_t training, float* stats, void* stream);
/* Backward in execution-ordered segments [seg_lo, seg_hi): 0 = heads + decoder.norm,
 * 1..n_dec = decoder layers (last layer first), then output embedding, encoder.norm, encoder
 * layers (last first), input embedding.  After segment s returns, the gradients of that
 * segment's parameters are final on `stream` (the shared value-embedding table only after the
 * last segment) - the host can launch their all-reduce on a side stream. */
/* d(loss) of the next pa_model_train_bwd calls is read from `upstream` (device f32 scalar; NULL = the stats block's own slot). */
int pa_model_set_upstream(pa_model* m, const float* upstream);
int pa_model_train_num_segments(const pa_model* m);
int pa_model_train_bwd(pa_model* m, int32_t seg_lo, int32_t seg_hi, float gscale, void* stream);
/* Gradient finality lag in segments.  With PA_SIDE_STREAM=1 (experimental, off by default: slower on MI355X) the model
 * owns one extra HIP stream: the work queued at
 * the end of a layer's backward segment (grouped weight-gradient GEMM, split-K reductions, bias column sums,
 * LayerNorm finishes - all independent of the dX chain) runs there, fenced with events against `stream`, while
 * `stream` continues with the next segment.  The gradients of segment s are then final, in `stream` order, once
 * segment s+2 has been enqueued or the last segment has returned (which joins everything): returns 2; 0 otherwise. */
int pa_model_grad_lag(const pa_model* m);
/* introspection for parity tests: device pointer + element count of an activation of the last forward */
int pa_model_tensor(pa_model* m, int32_t which, void** ptr, int64_t* numel);

/* ------------------------------------------------------------------------------------------
 * Greedy decode: reference plankassembly/models.py:267-307 (eval_step loop), 168-186 (_create_dist
 * eval branch, last row only), 235-256 (_sample).  K/V-cached (the reference recomputes the whole
 * prefix and the cross-attention K/V projection of `memory` every step); token-exact w.r.t. the
 * reference in PA_F32.
 *   1. run the encoder: pa_model_train_fwd with batch.output_value == NULL (T = 1);
 *   2. pa_decode_begin: lays out `ws` (pa_decode_ws_bytes), projects the cross-attention K/V of
 *      `memory` for every decoder layer once, resets the device-side step counter / outputs;
 *   3. pa_decode_step x Tmax: one token for every sequence.  All step kernels read the step index
 *      from device memory, so ONE captured hipGraph of a step can be replayed Tmax times;
 *   4. pa_decode_buffers: device pointers of tokens int64 [B][Tmax], attach int64 [B][Tmax]
 *      (-1 = no pointer), first_end int32 [B] (step at which END was first emitted, -1 = never),
 *      t_dev int32 (steps done).  The reference's early stop (all rows contain END) is the
 *      host checking first_end between replays and truncating to max(first_end)+1 columns.
 */
int64_t pa_decode_ws_bytes(pa_model* m, int32_t B, int32_t S, int32_t Tmax);
int pa_decode_begin(pa_model* m, void* ws, int64_t ws_bytes, int32_t Tmax, void* stream);
int pa_decode_step(pa_model* m, void* stream);
/* One step of TWO half-batches of the same decode (two handles over the same parameters, each after its own
 * pa_decode_begin) on two streams, attention launches strictly alternating between them so that one lane's K/V streaming
 * overlaps the other lane's latency-bound launches.  Under stream capture stream_b must already belong to stream_a's
 * capture (fork before, join after).  Host-side scheduling only: each lane computes exactly what pa_decode_step computes.
 * Experimental: on MI355X / ROCm 7.2 the cross-queue event edges cost more than the overlap. */
int pa_decode_step_pair(pa_model* a, pa_model* b, void* stream_a, void* stream_b);
int pa_decode_buffers(pa_model* m, void** tokens, void** attach, void** first_end, void** t_dev);
/* Cross-attention of one decode step in absorbed ("multi-query") form (reference plankassembly/models.py:284-307, the
 * cross-attention of nn.TransformerDecoderLayer with K = W_k memory + b_k, V = W_v memory + b_v): per batch element and head
 * ctx[b][h][:] = sum_s softmax_s(qt[b][h] . mem[s]) mem[s] over the element's memory rows, where the caller has put
 * qt_h = scale log2(e) W_k,h^T q_h into `qt` and applies W_v,h (+ b_v,h) to ctx afterwards.  bf16, d == 512, H <= 8
 * (PA_ESHAPE otherwise).  qt, ctx: [B][H][512]; mem: dense [B][S][512] with optional kpm [B][S] (1 = PAD) or packed rows
 * with cu [B + 1]. */
int pa_dec_cross_mq(void* ctx, const void* qt, const void* mem, const uint8_t* kpm, const int32_t* cu, int32_t B, int32_t S,
                    int32_t H, int32_t d, void* stream);
/* The same in exact f32 (f32 ctx / qt / mem; v_mfma_f32_16x16x4_f32): the cross-attention of the parity (token-exact) decode. */
/* pa_dec_cross_mq with scratch for RANGE BLOCKS: below ~256 batch elements one block per element leaves most CUs idle (one CU streams
 * an element's S x 512 memory rows at ~22 GB/s); with scratch an element's keys are walked by up to 256 / B blocks whose partial
 * (O, m, l) are merged by the last one to arrive (csrc/decode_mq.h, csrc/split_merge.h).  ws: pa_dec_cross_mq_ws_bytes(B, S) bytes,
 * 256-byte aligned, its first ceil(4 B / 256) * 256 bytes ZERO before the first launch (launches leave them zero). */
int64_t pa_dec_cross_mq_ws_bytes(int32_t B, int32_t S);
int pa_dec_cross_mq_ws(void* ctx, const void* qt, const void* mem, const uint8_t* kpm, const int32_t* cu, int32_t B,
                       int32_t S, int32_t H, int32_t d, void* ws, int64_t ws_bytes, void* stream);
int pa_dec_cross_mq32_ws(float* ctx, const float* qt, const float* mem, const uint8_t* kpm, const int32_t* cu, int32_t B,
                         int32_t S, int32_t H, int32_t d, void* ws, int64_t ws_bytes, void* stream);    /* the exact-f32 form (same scratch size) */
int pa_dec_cross_mq32(float* ctx, const float* qt, const float* mem, const uint8_t* kpm, const int32_t* cu, int32_t B, int32_t S,
                      int32_t H, int32_t d, void* stream);
/* The self-attention form of the same launch (exact f32; what the token-exact decode step runs on its cache of layer-input rows):
 * rows [B][Tmax][512], of which element b attends over rows 0 .. *t_dev (the key count t + 1 is read on the device, so the launch can
 * sit in a captured graph). */
int pa_dec_self_mq32(float* ctx, const float* qt, const float* xcache, const int32_t* t_dev, int32_t B, int32_t Tmax, int32_t H,
                     int32_t d, void* stream);

#ifdef __cplusplus
}
#endif
#endif
