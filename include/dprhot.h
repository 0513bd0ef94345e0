/*
 * dprhot.h -- C ABI of libdprhot.so: the MI355X (gfx950) implementation of dpr-scale's in-batch
 * contrastive hot path.  Reference boundary (paths relative to the dpr-scale tree):
 *
 *   dpr_scale/task/dpr_task.py:153-214  DenseRetrieverTask.training_step   (the step)
 *   dpr_scale/task/dpr_task.py:98-105   DenseRetrieverTask.sim_score       (Q x C^T, masked fill)
 *   dpr_scale/task/dpr_task.py:46,212   nn.CrossEntropyLoss()              (row softmax CE, mean)
 *   dpr_scale/task/dpr_task.py:235-246  compute_rank_metrics               (rank of the gold ctx)
 *   dpr_scale/run_retrieval_pytorch.py:141-176 search_index                (einsum + topk)  ["next" row]
 *
 * The reference is 100 % Python and has no FFI of its own; these entry points are what a binding for this
 * path binds (INTEGRATION.md shows the ctypes stub and the dpr_task.py patch).
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer owned by the caller (e.g. torch's caching allocator) unless its
 *     name starts with h_.  Nothing is allocated, freed or retained by the library.
 *   - All matrices are row-major and contiguous; rows are 16-byte aligned: d % 8 == 0, Nc % 8 == 0.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Kernels are enqueued
 *     asynchronously; no entry point synchronises with the host.
 *   - bf16 values travel as uint16_t bit patterns (round-to-nearest-even from fp32).
 *   - Return value: 0 = OK, <0 = error (DPRHOT_E_*); dprhot_last_error() returns a thread-local message.
 *     Functions are re-entrant; the only process-wide state is the explicit option table below (test / A-B switches that
 *     production never touches) -- the library reads no environment variable.
 *   - Shapes: B = rows (queries) held by this rank, Nc = all columns (contexts) after the gather,
 *     d = hidden size.
 */
#ifndef DPRHOT_H
#define DPRHOT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPRHOT_VERSION 172 /* 0.1.72: + dprhot_fwd_one_pass; options nl128, nl128_max_tiles, loss_with_dq, dq_cap_few, pair128, pair128_slices */

#define DPRHOT_OK 0
#define DPRHOT_E_INVALID (-1)     /* bad argument (NULL pointer, non-positive or misaligned size) */
#define DPRHOT_E_HIP (-2)         /* a HIP runtime call failed; message has hipGetErrorString */
#define DPRHOT_E_UNSUPPORTED (-3) /* shape outside what the kernels were built for */
#define DPRHOT_E_WORKSPACE (-4)   /* workspace too small */

typedef uint16_t dprhot_bf16;

int dprhot_version(void);
const char* dprhot_last_error(void);

/* Process-wide test / A-B switches of the shape plans (all default to "plan decides"; production never calls this).  Names:
 *   tile (-1 | 0..5)   no_tr   unfused_bwd   big_min (256)   no_nl   no_big_bwd   no_skinny   no_small_step   no_short
 *   sk_cols (0 | 64 | 128)   search_unfused   no_8pb   no_wide   no_8p_store   no_wide_bwd   nt_stores (1)   sk_dq_slices (0)
 *   sk_fused (1: the few-rows step without its dScores launch from 2^19 scores up | 2: wherever it exists | 0: never)
 *   sk_w8 (1) / sk_sim_w8 (1): eight waves per workgroup in the few-rows backward / sim launch   sk_pair (0): one kind of backward unit
 *   sk_dbg (0; timing experiments only)
 * Setting one changes the plans of every later call on every thread (workspace sizes included: query them after setting).
 * DPRHOT_E_INVALID for an unknown name. */
int dprhot_set_option(const char* name, int value);
/* Bumped by every dprhot_set_option: whoever caches a plan fact of this library (dprhot_step_wants_g, dprhot_train_dq_slabs,
 * dprhot_workspace_bytes) keys the cache on it -- options set through this C ABI from any binding invalidate every such cache. */
long long dprhot_options_epoch(void);
int dprhot_get_option(const char* name, int* h_value);

/* Bytes of scratch the fused entry points (dprhot_inbatch_fwd/_bwd, dprhot_dq) need for this shape. */
int dprhot_workspace_bytes(int B, int Nc, int d, size_t* h_out);

/* fp32 -> bf16 (RNE) of n contiguous values.  Producer side of the gather: the encoder output
 * (hf_model.py:36-41, fp32) is written straight into this rank's slot of the gathered bf16 buffer.
 * n % 8 == 0. */
int dprhot_cast_bf16(const float* src, dprhot_bf16* dst, size_t n, void* stream);

/* Both producer-side casts of one step in ONE launch: Qb <- q (nq values), Cdst <- c (nc values), fp32 -> bf16
 * RNE.  Cdst is this rank's slot of the gathered context buffer (W == 1) or the all-gather send buffer. */
int dprhot_prep(const float* q, size_t nq, dprhot_bf16* Qb, const float* c, size_t nc, dprhot_bf16* Cdst, void* stream);

/* Multi-rank gather in ONE collective (the reference issues four all_gathers, dpr_task.py:169-176).
 * dprhot_pack_ctx writes this rank's all-gather send buffer [rows_c, d] bf16: rows [0, n_ctx) = the context rows
 * (fp32 -> bf16 RNE), followed by rows whose bytes carry the dummy-context mask (mask may be NULL = no dummies).
 * rows_c = dprhot_packed_rows(n_ctx, d) (a multiple of 8; of 64 from n_ctx = 2048 on).  After all-gathering W such buffers back to back, the
 * result IS the context matrix C [W*rows_c, d] of every other entry point -- the mask rows are just extra columns
 * that dprhot_unpack_mask marks as masked in the column mask it builds ([W*rows_c] bytes).  The label offset of
 * rank r becomes r * rows_c, and rank r's gradient is the first n_ctx rows of its reduce-scatter chunk. */
int dprhot_packed_rows(int n_ctx, int d, int* h_rows);
int dprhot_pack_ctx(const float* c, const uint8_t* mask, int n_ctx, int d, dprhot_bf16* send, void* stream);
int dprhot_unpack_mask(const dprhot_bf16* gathered, int W, int n_ctx, int d, uint8_t* colmask, void* stream);

/* sim_score (dpr_task.py:98-105) + the temperature scale (:211), one kernel:
 *   S[i][j] = inv_T * sum_k Q[i][k] * C[j][k]      (bf16 MFMA, fp32 accumulate)
 *   S[i][j] = -inf where colmask[j] != 0            (colmask may be NULL; it is the row that
 *                                                    `mask.repeat(Nq, 1)` at :197 broadcasts)
 * Q [B,d], C [Nc,d] bf16; S [B,Nc] fp32. */
int dprhot_sim_fwd(const dprhot_bf16* Q, int B, const dprhot_bf16* C, int Nc, int d, const uint8_t* colmask,
                   float inv_T, float* S, void* stream);

/* Row softmax cross-entropy (dpr_task.py:46,212) fused with its backward into dScores, one pass over S:
 *   row_lse[i]  = logsumexp_j S[i][j]
 *   row_loss[i] = row_lse[i] - S[i][y[i]]
 *   G[i][j]     = (exp(S[i][j] - row_lse[i]) - [j == y[i]]) * grad_scale        (bf16, 0 at -inf columns)
 * The gold column of row i is y[i] + y_offset: y holds the rank-local positive indices from the batch
 * (dpr_transform.py:164-166) and y_offset = rank * ctx_per_rank is the label offset the reference adds
 * in its gather loop (dpr_task.py:189-190).  row_win_start is relative to the same offset.
 * grad_scale carries 1/(Nq_global * T); the incoming grad_output is applied later by dprhot_dq/_dc.
 * row_win_start/win_len (optional, NULL/0): the in_batch_negatives=False branch (:198-207) -- row i only
 * sees columns [row_win_start[i], row_win_start[i] + win_len).
 * row_loss, row_lse, G may each be NULL (not written). */
int dprhot_softmax_ce_fwd_bwd(const float* S, int B, int Nc, const int64_t* y, int64_t y_offset, float grad_scale,
                              const int64_t* row_win_start, int win_len, float* row_loss, float* row_lse,
                              dprhot_bf16* G, void* stream);

/* out[0] = scale * sum_i x[i]  (deterministic single-workgroup tree; the `mean` of CrossEntropyLoss). */
int dprhot_reduce_sum(const float* x, int n, float scale, float* out, void* stream);

/* Backward of sim_score into the local query rows:  dQ[B,d] = s * G[B,Nc] x C[Nc,d]   (fp32 out),
 * s = h_scale * (d_scale ? *d_scale : 1).  d_scale is the autograd grad_output scalar, read on the
 * device so that no host sync is needed (AMP loss scale).  Split-K over Nc; `workspace` of
 * dprhot_workspace_bytes() is required. */
int dprhot_dq(const dprhot_bf16* G, const dprhot_bf16* C, int B, int Nc, int d, float h_scale,
              const float* d_scale, float* dQ, void* workspace, size_t workspace_bytes, void* stream);

/* Backward into ALL context columns (this rank's partial; summed over ranks by reduce-scatter):
 *   dC_part[Nc,d] = s * G^T[Nc,B] x Q[B,d]   (fp32 out) */
int dprhot_dc(const dprhot_bf16* G, const dprhot_bf16* Q, int B, int Nc, int d, float h_scale,
              const float* d_scale, float* dC_part, void* stream);

/* compute_rank_metrics (dpr_task.py:235-246) without the sort or the per-row host syncs:
 *   rank[i] = 1 + #{j : S[i][j] > S[i][y[i]]} + #{j < y[i] : S[i][j] == S[i][y[i]]}
 * (== position of y[i] in torch.sort(S[i], descending=True, stable=True); bit-exact integer result). */
int dprhot_rank_of_gold(const float* S, int rows, int cols, const int64_t* y, int64_t y_offset, int64_t* rank,
                        void* stream);

/* Whole forward of the step for this rank's rows in TWO launches:
 *   1. sim GEMM whose epilogue applies mask and 1/T, stores the fp32 logits and leaves, per row and column
 *      tile, the (max, sum-exp) pair plus the gold logit;
 *   2. logsumexp from those pairs, then one streaming pass over the logits producing G (bf16), row_loss,
 *      row_lse and loss_sum[0] = sum_i row_loss[i] (fixed-point integer atomics: deterministic).
 * S_out may be NULL (logits live in the workspace) or a [B,Nc] fp32 buffer (debug / parity / eval).
 * row_loss, row_lse, G may be NULL.  `workspace` must hold dprhot_workspace_bytes(B, Nc, d) bytes.
 * LARGE shapes (>= 256 tiles of 256 x 256, B > 128, d % 128 == 0) with S_out == NULL run the NO-LOGITS plan instead: the
 * logits are never stored (at B = 8192, Nc = 65536 they would be 2 GiB, written once and read once) --
 *   1. sim GEMM whose epilogue keeps only (max, sum-exp) per row and 64-column strip plus the gold logit,
 *   2. logsumexp / loss from those,
 *   3. the same GEMM again (bit-identical accumulators), its epilogue writing G = (softmax - onehot) * grad_scale as bf16.
 * The workspace of such a shape holds no logit buffer.  Passing S_out selects the two-launch plan above at any shape. */
int dprhot_inbatch_fwd(const dprhot_bf16* Q, int B, const dprhot_bf16* C, int Nc, int d, const int64_t* y,
                       int64_t y_offset, const uint8_t* colmask, float inv_T, float grad_scale, float* S_out, float* row_loss,
                       float* row_lse, float* loss_sum, dprhot_bf16* G, void* workspace,
                       size_t workspace_bytes, void* stream);

/* Whole backward: dQ (local rows) and dC_part (all columns) from G; see dprhot_dq / dprhot_dc.  The two
 * GEMMs share only their input, so they run side by side in ONE launch (plus the split-K combine of dQ
 * when Nc is long). */
int dprhot_inbatch_bwd(const dprhot_bf16* G, const dprhot_bf16* Q, const dprhot_bf16* C, int B, int Nc, int d,
                       float h_scale, const float* d_scale, float* dQ, float* dC_part, void* workspace,
                       size_t workspace_bytes, void* stream);

/* The two launches of dprhot_inbatch_fwd, individually (profiling, or a caller that wants the logits
 * before deciding on the softmax): dprhot_sim_stats leaves logits (S_out, or the workspace when NULL) and
 * the per-tile statistics in `workspace`; dprhot_softmax_finish consumes them (S_in NULL = the workspace
 * logits).  Same B, Nc, d and workspace for both calls.
 * Short rows (Nc <= 4096 and B <= 64): dprhot_sim_stats writes up to 4 split-K slabs of PARTIAL logits into the
 * workspace and leaves S_out alone; the summed logits exist only after dprhot_softmax_finish, which writes
 * them to its S_in argument when that is non-NULL (there S_in is an output).  Callers that want logits from
 * one call use dprhot_sim_fwd.
 * No-logits shapes (see dprhot_inbatch_fwd) with S_out / S_in == NULL: dprhot_sim_stats leaves only statistics,
 * dprhot_softmax_finish turns them into row_lse / row_loss / loss_sum and must be given G == NULL (there are no logits to
 * stream; DPRHOT_E_UNSUPPORTED otherwise), and dprhot_dscores is the third launch. */
int dprhot_sim_stats(const dprhot_bf16* Q, int B, const dprhot_bf16* C, int Nc, int d, const int64_t* y,
                     int64_t y_offset, const uint8_t* colmask, float inv_T, float* S_out, void* workspace,
                     size_t workspace_bytes, void* stream);
int dprhot_softmax_finish(const float* S_in, int B, int Nc, int d, const int64_t* y, int64_t y_offset,
                          float grad_scale, float* row_loss, float* row_lse, float* loss_sum, dprhot_bf16* G,
                          void* workspace, size_t workspace_bytes, void* stream);

/* Third launch of the no-logits forward: G[B,Nc] (bf16) = (softmax(S) - onehot(y + y_offset)) * grad_scale with
 * S = (Q x C^T) * inv_T recomputed tile by tile on the MFMA (never stored).  row_lse [B]: the rows' logsumexp, or NULL to use
 * what dprhot_softmax_finish left in `workspace`.  DPRHOT_E_UNSUPPORTED at shapes that are not no-logits shapes. */
int dprhot_dscores(const dprhot_bf16* Q, int B, const dprhot_bf16* C, int Nc, int d, const int64_t* y, int64_t y_offset,
                   const uint8_t* colmask, float inv_T, float grad_scale, const float* row_lse, dprhot_bf16* G,
                   void* workspace, size_t workspace_bytes, void* stream);

/* compute_rank_metrics (dpr_task.py:235-246) straight from the embeddings: rank as dprhot_rank_of_gold of
 * S = (Q x C^T) * inv_T with masked columns at -inf, bit-exact.  At no-logits shapes S is never written: the gold logits
 * come from a gathered mini GEMM that runs the identical MFMA chain, the count-greater happens in the similarity GEMM's
 * epilogue.  Smaller problems score into the workspace and count there. */
int dprhot_sim_rank(const dprhot_bf16* Q, int B, const dprhot_bf16* C, int Nc, int d, const int64_t* y, int64_t y_offset,
                    const uint8_t* colmask, float inv_T, int64_t* rank, void* workspace, size_t workspace_bytes,
                    void* stream);

/* Both halves of validation from the embeddings in one call (dpr_task.py:224-227, :296-299: compute_rank_metrics and self.loss read
 * the same score matrix): rank as dprhot_sim_rank, row_loss / row_lse (either may be NULL) and loss_sum = sum_i (lse_i - S[i][y_i]) as
 * dprhot_inbatch_fwd with grad_scale unused.  At no-logits shapes the similarity GEMM runs ONCE: its epilogue counts and keeps the
 * softmax statistics (no score matrix); smaller problems compose the two existing calls. */
int dprhot_sim_rank_loss(const dprhot_bf16* Q, int B, const dprhot_bf16* C, int Nc, int d, const int64_t* y, int64_t y_offset,
                         const uint8_t* colmask, float inv_T, int64_t* rank, float* row_loss, float* row_lse, float* loss_sum,
                         void* workspace, size_t workspace_bytes, void* stream);

/* The same forward reading the encoder outputs as they are (fp32), so that no separate cast launch and no extra
 * round trip through HBM is needed: q [B,d] fp32; c [Nc,d] fp32 when the rank holds every column (world size
 * 1), or NULL when Cb is the already gathered bf16 buffer (world size > 1: the all-gather ships bf16).
 * The sim kernel rounds to bf16 (RNE) while staging and writes the bf16 images to Qb (and to Cb when c is
 * given) for the backward GEMMs.  Everything else as dprhot_sim_stats / dprhot_inbatch_fwd. */
int dprhot_sim_stats_f32(const float* q, const float* c, dprhot_bf16* Qb, dprhot_bf16* Cb, int B, int Nc, int d,
                         const int64_t* y, int64_t y_offset, const uint8_t* colmask, float inv_T, float* S_out,
                         void* workspace, size_t workspace_bytes, void* stream);
int dprhot_inbatch_fwd_f32(const float* q, const float* c, dprhot_bf16* Qb, dprhot_bf16* Cb, int B, int Nc, int d,
                           const int64_t* y, int64_t y_offset, const uint8_t* colmask, float inv_T, float grad_scale,
                           float* S_out, float* row_loss, float* row_lse, float* loss_sum, dprhot_bf16* G,
                           void* workspace, size_t workspace_bytes, void* stream);

/* One training step -- forward and backward of dpr_task.py:197-212 -- in a single call, for callers that know a
 * backward will follow (autograd: any input requires grad).  Arguments as dprhot_inbatch_fwd_f32 followed by those
 * of dprhot_inbatch_bwd; loss_sum, dQ and dC_part are required, G unless dprhot_step_wants_g says 0.  dQ / dC_part are scaled by
 * h_scale * (d_scale ? *d_scale : 1): pass the autograd grad_output there if it is known at forward time, or 1 and
 * multiply later.  At the latency-bound shapes (B <= 32, Nc <= 1152, d % 16 == 0) this is TWO launches -- the sim GEMM,
 * then one kernel doing softmax-CE, dScores and both backward GEMMs from an LDS-resident G; otherwise it equals
 * dprhot_inbatch_fwd_f32 + dprhot_inbatch_bwd (three launches). */
/* Whether the step entry points below need a G buffer at this shape: *h_wants = 1 -- the plan materialises the dScores (G required);
 * 0 -- G may be NULL, and passing NULL (with S_out NULL) selects the form of the step that never writes them: the few-rows plan's sim
 * launch then leaves every 128-column tile's own softmax (bf16) and the backward units derive the row logsumexp and one factor per
 * (row, tile) themselves -- three launches instead of four at BASELINE cfg3 per rank (dpr_task.py:197-212 and its backward).  A
 * non-NULL G is always honoured (and costs the dScores launch). */
int dprhot_step_wants_g(int B, int Nc, int d, int* h_wants);
/* *h_nl = 1 when the forward of this shape never stores the logits (the "no-logits" plan: >= 256 tiles of 256 x 256 -- or the option
 * big_min -- K % 128 == 0; dprhot_softmax_finish then yields logsumexp / loss only and G comes from dprhot_inbatch_fwd / dprhot_dscores),
 * 0 when the logits live in the workspace.  A pure function of the shape and the options, like every plan. */
int dprhot_fwd_no_logits(int B, int Nc, int d, int* h_nl);
/* How dprhot_inbatch_fwd / the one-call steps form G when the caller wants the dScores and not the logits (S_out == NULL, G != NULL):
 * *h_kind = 0 the logits are stored (workspace) and a streaming softmax turns them into G; 1 ONE pass of the 256 x 256 GEMM (strip
 * statistics + fp16 softmax numerators, then a row kernel that rescales them into G in place); 2 the same on the 128 x 128 LDS-DMA tile
 * (the shapes whose 256-wide tiles would leave most of the chip idle: a few hundred query rows against thousands of contexts). */
int dprhot_fwd_one_pass(int B, int Nc, int d, int* h_kind);
int dprhot_inbatch_step_f32(const float* q, const float* c, dprhot_bf16* Qb, dprhot_bf16* Cb, int B, int Nc, int d,
                            const int64_t* y, int64_t y_offset, const uint8_t* colmask, float inv_T, float grad_scale,
                            float h_scale, const float* d_scale, float* S_out, float* row_loss, float* row_lse,
                            float* loss_sum, dprhot_bf16* G, float* dQ, float* dC_part, void* workspace,
                            size_t workspace_bytes, void* stream);

/* The same step for world size > 1, everything between the all-gather and the reduce-scatter in ONE call
 * (dpr_task.py:177-212 and its backward on this rank's rows).  `gathered` is the all-gathered packed buffer
 * [W * rows_c, d] bf16 (dprhot_pack_ctx on every rank); it is used as the context matrix as it is:
 *   - the column mask is read from its mask rows inside the sim kernel (no dprhot_unpack_mask launch, no colmask),
 *   - labels are y + rank * rows_c,
 *   - dC_part [W * rows_c, d] is what the caller reduce-scatters; in every rank chunk k its element
 *     [k * rows_c + n_ctx][0] (first mask row, a dead gradient) carries THIS rank's loss numerator, so the rank's
 *     reduce-scatter output holds the global sum of the loss numerators at [n_ctx][0]: no all-reduce for the loss.
 * loss_sum still receives the local numerator.  G as in dprhot_inbatch_step_f32 (dprhot_step_wants_g(B, W * rows_c, d)). */
int dprhot_inbatch_step_packed_f32(const float* q, const dprhot_bf16* gathered, dprhot_bf16* Qb, int B, int W, int rank,
                                   int n_ctx, int d, const int64_t* y, float inv_T, float grad_scale, float h_scale,
                                   const float* d_scale, float* row_loss, float* row_lse, float* loss_sum,
                                   dprhot_bf16* G, float* dQ, float* dC_part, void* workspace, size_t workspace_bytes,
                                   void* stream);

/* Brute-force retrieval epilogue (run_retrieval_pytorch.py:149-150): per row, the k largest scores and
 * their column indices, descending, ties by lower column index.  k <= 4096 (the reference's recipes use --topk 100 and 1000; up to
 * 256 and up to 1024 run on 12 / 48 KB of LDS per row, beyond that on 96 KB: 8192 sort slots), k <= cols. */
int dprhot_topk(const float* S, int rows, int cols, int k, float* values, int64_t* indices, void* stream);

/* The same as a streaming update, for a corpus that is scored in pieces (the shard loop of
 * run_retrieval_pytorch.py:196-243 and its "sort the score again if shard > 1" re-merge at :272-277):
 * folds the scores S[rows][0..cols) (row stride ld, global column index = col_offset + j) into the running
 * per-row top-k state values/indices [rows,k] (sorted; total order score desc, column asc).  first != 0 starts
 * from an empty state (slots not yet filled hold -inf / -1).  The result after any sequence of updates equals
 * dprhot_topk of the concatenated score matrix. */
int dprhot_topk_update(const float* S, int rows, int cols, int64_t ld, int64_t col_offset, int k, float* values,
                       int64_t* indices, int first, void* stream);

/* The same update for ANY k (run_retrieval_pytorch.py:69 takes any --topk): the state lives in HBM, nothing has to fit in LDS
 * (csrc/wideselect.h: exact radix select over state + chunk, then only the k winners are sorted -- blocks of 4096 in LDS, merge
 * passes over HBM; ties exact at any multiplicity: the select continues over the ids).  workspace:
 * dprhot_topk_wide_workspace_bytes(rows, k) bytes; its first rows * 8 32-bit words are one record per row whose word 5 is an error
 * word -- a row with a non-zero one keeps its old state (no such condition is defined at present: always 0). */
int dprhot_topk_wide_workspace_bytes(int rows, int k, size_t* h_bytes);
int dprhot_topk_update_wide(const float* S, int rows, int cols, int64_t ld, int64_t col_offset, int k, float* values, int64_t* indices,
                            int first, void* workspace, size_t workspace_bytes, void* stream);

/* search_index (run_retrieval_pytorch.py:141-166) for one resident corpus shard: scores = Q x C^T on bf16 MFMA
 * (fp32 scores; the reference scores in fp16), chunk columns at a time, each chunk folded into the running
 * top-k -- the [nq, n_ctx] score matrix never exists.  Q [nq,d], C [n_ctx,d] bf16; passage ids are
 * id_offset + row.  n_ctx and chunk multiples of 8 (a ragged tail goes through sim_fwd + topk_update with
 * cols < ld).  first as in dprhot_topk_update.  The first chunk of an empty state (at most 65536 passages of it) is scored into the workspace
 * and selected from; for every later chunk the GEMM epilogue compares each score with the row's current k-th best
 * and only appends the few that beat it to a candidate list, which a merge kernel folds into the state.
 * workspace: dprhot_search_workspace_bytes(nq, chunk). */
int dprhot_search_workspace_bytes(int nq, int chunk, size_t* h_out);
int dprhot_search(const dprhot_bf16* Q, int nq, const dprhot_bf16* C, int64_t n_ctx, int d, int64_t id_offset,
                  int k, int chunk, float* values, int64_t* indices, int first, void* workspace,
                  size_t workspace_bytes, void* stream);

/* Pairwise scores of the CITADEL router / distillation path (dpr_scale/task/citadel_task.py:137-146, sim_score(...,
 * pairwise=True)): query b against its own M contexts,  S[b][j] = <q[b], c[b*M + j]>, -inf where mask[b*M + j] != 0
 * (mask may be NULL).  q [B,d], c [B*M,d] fp32 (router vectors are vocabulary-wide: d = 30522), S [B,M] fp32.  Any d. */
int dprhot_pairwise_fwd(const float* q, const float* c, const uint8_t* mask, int B, int M, int d, float* S, void* stream);
/* Its backward: dq[b] = sum_j g[b][j] c[b*M+j] and dc[b*M+j] = g[b][j] q[b]; g [B,M] fp32 with 0 at masked pairs.
 * dq / dc may each be NULL. */
int dprhot_pairwise_bwd(const float* g, const float* q, const float* c, int B, int M, int d, float* dq, float* dc, void* stream);

/* The step as the autograd operator of dpr_scale_amd/hotpath.py runs it (dpr_task.py:197-212 + its backward inside the forward call):
 * dprhot_inbatch_step_f32 / _packed_f32 with
 *   loss_scale   loss_out[0] = loss_scale * sum_i row_loss[i]: pass 1 / Nq_global and the mean of dpr_task.py:212 leaves the kernel
 *                ready (and the stamp of the packed form carries that value)
 *   d_scale      DEVICE scalar the gradients are scaled by (required): the grad_output the caller expects backward() to deliver
 *   dq_part      NULL, or -- when dprhot_train_dq_slabs(B, Nc, d) = n > 0 -- a caller buffer [n][B][d] fp32: the step then leaves
 *                dQ as n unscaled split-K partial slabs there (dQ itself untouched) and skips its own reduction launch; the
 *                caller's dprhot_rescale_grads forms dQ from them (the operator's backward launches that anyway)
 *   dC_part      fp32 (dc_kind = 2), or bf16 (dc_kind = 0: the wire format of the reduce-scatter written by the dC epilogue itself,
 *                no loss stamp; DPRHOT_E_UNSUPPORTED at shapes whose plan has no such epilogue -- the caller then asks for fp32)
 * No S_out, no h_scale (= 1). */
int dprhot_train_dq_slabs(int B, int Nc, int d, int* h_nslabs);
int dprhot_train_step_f32(const float* q, const float* c, dprhot_bf16* Qb, dprhot_bf16* Cb, int B, int Nc, int d, const int64_t* y,
                          int64_t y_offset, const uint8_t* colmask, float inv_T, float grad_scale, float loss_scale, const float* d_scale,
                          float* row_loss, float* row_lse, float* loss_out, dprhot_bf16* G, float* dQ, float* dq_part, void* dC_part,
                          int dc_kind, void* workspace, size_t workspace_bytes, void* stream);
int dprhot_train_step_packed_f32(const float* q, const dprhot_bf16* gathered, dprhot_bf16* Qb, int B, int W, int rank, int n_ctx, int d,
                                 const int64_t* y, float inv_T, float grad_scale, float loss_scale, const float* d_scale, float* row_loss,
                                 float* row_lse, float* loss_out, dprhot_bf16* G, float* dQ, float* dq_part, void* dC_part, int dc_kind,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* backward() of that operator: the gradients were computed for grad_output = *used; the autograd engine delivers *go.
 *   nslabs > 0         dQ[0..n_dq) = *go * sum of the nslabs slabs of dq_part, in slab order (bit-reproducible)
 *   if (*go != *used)  dQ (nslabs == 0) and dC[0..n_dc) are multiplied by *go / *used   (AMP's loss scale changes once in thousands
 *                      of steps; otherwise their workgroups read two floats and leave)
 *   out2[0] = *go   (what the gradients are now scaled by)
 *   out2[1] = *go if it is a finite, normal, non-zero number, else 1   (the d_scale the next forward should expect)
 * out2 is a fresh 2-float buffer (never go / used).  dQ fp32, dC fp32 (dc_kind 2) or bf16 (0); either may be NULL with count 0;
 * counts are multiples of 8. */
int dprhot_rescale_grads(float* dQ, size_t n_dq, const float* dq_part, int nslabs, void* dC, size_t n_dc, int dc_kind, const float* go,
                         const float* used, float* out2, void* stream);

/* Gradient all-reduce of the encoder towers (reference: dpr_scale/task/dpr_task.py:90-92 registers torch's fp16_compress_hook on
 * the DDP model: bucket -> fp16 -> ONE ring all-reduce -> fp32).  The hook of dpr_scale_amd/comm_hooks.py moves a bucket as
 *   pack -> all-to-all (every pair of GPUs owns an xGMI link) -> sum of the W received shards in fp32 -> all-gather -> unpack;
 * these are its three local legs, one pass over HBM each.  Wire kinds: 0 = bf16 (RNE), 1 = fp16 (RNE, the reference's), 2 = fp32.
 *   dprhot_grad_pack        send[i] = wire(bucket[i] * scale) for i < n, 0 for n <= i < n_padded (n_padded % 8 == 0; = W * shard)
 *   dprhot_grad_sum_shards  out[i] = sum_r recv[r * shard + i], fp32 accumulation in rank order r = 0 .. W-1 (deterministic), ONE
 *                           rounding into out_kind (= wire kind, or 2 for an fp32 return leg); shard % 8 == 0
 *   dprhot_grad_unpack      bucket[i] = float(full[i]) for i < n (kind of `full` = the return leg's) */
int dprhot_grad_pack(const float* bucket, size_t n, float scale, int wire, void* send, size_t n_padded, void* stream);
int dprhot_grad_sum_shards(const void* recv, int W, size_t shard, int wire, int out_kind, void* out, void* stream);
int dprhot_grad_unpack(const void* full, int kind, float* bucket, size_t n, void* stream);

/* Optional communicator: the collectives of the path (dpr_task.py:166-176 gathers; the autograd split of :192-195)
 * issued directly on the caller's stream through the RCCL library already present in the process (dlopen; no
 * link-time dependency) instead of through torch.distributed -- no stream hand-over, ~3x less host time per call.
 *   dprhot_comm_unique_id   rank 0: 128-byte id to be handed to every rank (any side channel)
 *   dprhot_comm_init        COLLECTIVE over the W ranks, on the current HIP device; *h is the communicator handle
 *   dprhot_allgather_ctx    recv[r * bytes_per_rank ...] = rank r's send  (the packed context buffer of dprhot_pack_ctx)
 *   dprhot_reducescatter_dc recv[0 .. count_per_rank) = sum over ranks of send[rank * count_per_rank ...]  (fp32 dC partials)
 *   dprhot_reducescatter_rows the same for dC partials of any storage kind (0 bf16, 1 fp16, 2 fp32: the wire formats of the backward)
 *   dprhot_allreduce_sum    in place, fp32 (the loss numerator)
 *   dprhot_allgather_allpairs      the same result as dprhot_allgather_ctx by a DIRECT ALL-PAIRS exchange: one grouped send / receive of
 *                                  the one send buffer to / from every peer (SURVEY 8(e) "Topology": on the fully connected node every
 *                                  pair of GPUs owns an xGMI link, so the W - 1 transfers use W - 1 links at once instead of a ring's
 *                                  W - 1 sequential hops)
 *   dprhot_reducescatter_allpairs  the same result as dprhot_reducescatter_rows: chunk k of send goes to rank k, the W chunks received
 *                                  land in tmp [W * count_per_rank] (caller-owned, storage kind `kind`), then ONE kernel adds them in
 *                                  fp32 in rank order (deterministic; a half-width wire is rounded once per partial, never inside the
 *                                  sum) into recv, stored as `out_kind` (= kind, or 2 for fp32)
 *   dprhot_comm_has_allpairs       LOCAL, no communication: 1 when this process's RCCL has ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd
 *                                  (the two all-pairs entry points would run), 0 when not -- what a rank checks BEFORE its peers enter a
 *                                  send / receive group (dist.try_direct_comm, dist.choose_path_collectives)
 * One communicator per rank process, used from one thread; every rank issues the same calls in the same order. */
int dprhot_comm_unique_id(void* id128);
int dprhot_comm_init(const void* id128, int W, int rank, void** h);
int dprhot_comm_destroy(void* h);
int dprhot_allgather_ctx(void* h, const void* send, void* recv, size_t bytes_per_rank, void* stream);
int dprhot_reducescatter_dc(void* h, const float* send, float* recv, size_t count_per_rank, void* stream);
int dprhot_reducescatter_rows(void* h, const void* send, void* recv, size_t count_per_rank, int kind, void* stream);
int dprhot_allreduce_sum(void* h, float* buf, size_t count, void* stream);
int dprhot_allgather_allpairs(void* h, const void* send, void* recv, size_t bytes_per_rank, void* stream);
int dprhot_reducescatter_allpairs(void* h, const void* send, void* tmp, void* recv, size_t count_per_rank, int kind, int out_kind, void* stream);
int dprhot_comm_has_allpairs(void* h);

#ifdef __cplusplus
}
#endif
#endif /* DPRHOT_H */
