/* mmt_layer.h -- C ABI of the fused residual-block kernels around the attention core
 * (SURVEY.md 8(f) rank 1).
 *
 * They replace the element-wise chain of etcmodel's `ResidualBlock` / `DenseLayers` as
 * instantiated by `RelativeTransformerLayers` (reference ctor src/modeling/models/mmt_encoder.py:124-135;
 * math SURVEY.md App. A.3, pre-activation order `y = x + Dropout(inner(LayerNorm(x)))`,
 * LayerNorm eps 1e-12, tanh-GELU mmt_encoder.py:53-54), i.e. what TF executes as separate
 * BiasAdd / Dropout / Add / LayerNormalization / Gelu kernels and their gradients.
 *
 * Conventions as in mmt_attn.h: device pointers, caller-owned buffers + workspace, caller's
 * hipStream_t, 0 / negative MMT_E_* return codes, message via mmt_last_error().
 * Activations are [rows, H] row-major in `dtype` (MMT_F32 | MMT_BF16); parameters (bias, gamma,
 * beta) and their gradients are fp32 (master weights); mean / rstd are fp32 [rows].
 * Dropout keeps element (row, col) iff hash(seed, row*H + col) >= p * 2^16 on 16 bits; the
 * backward regenerates the mask from the same seed (nothing is stored).
 */
#ifndef MMT_LAYER_H_
#define MMT_LAYER_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mmt_rows_desc {
  int64_t rows;        /* B * S                                         */
  int32_t H;           /* row length; multiple of 8, <= 8192            */
  int32_t dtype;       /* MMT_F32 | MMT_BF16                            */
  float eps;           /* LayerNorm epsilon (1e-12 in the reference)    */
  float dropout_p;     /* hidden_dropout_prob; 0 disables               */
  uint64_t dropout_seed;
  int32_t accumulate;  /* backward: != 0 adds dbias / dgamma / dbeta INTO the given buffers (fp32
                          master gradients) instead of overwriting them                       */
  int32_t defer_reduce; /* backward: != 0 leaves the column-sum partials in the workspace and does NOT launch
                           the fixed-order reduce; the caller finishes with mmt_colsum_reduce (same desc,
                           same workspace) on a stream of its choice -- the parameter gradients are off the
                           critical path of backward                                                      */
  const uint64_t* dropout_epoch; /* ABI 4: DEVICE uint64 added to dropout_seed by the kernel, or NULL (mmt_attn.h) */
} mmt_rows_desc;

/* Bytes of scratch the *_bwd entry points need (column-sum partials). */
size_t mmt_layer_workspace_bytes(const mmt_rows_desc* desc);

/* y = LayerNorm(x) * gamma + beta; mean, rstd saved for the backward. */
int mmt_ln_fwd(const mmt_rows_desc* desc, const void* x, const float* gamma, const float* beta,
               void* y, float* mean, float* rstd, void* stream);

/* dx, dgamma[H], dbeta[H] of mmt_ln_fwd (dgamma / dbeta are overwritten). */
/* mmt_ln_bwd with a second incoming gradient: dx = dx_in + LayerNormBackward(dy) (dx_in may be NULL) -- the input of
 * the first pre-activation block feeds both its LayerNorm and its residual sum (mmt_encoder.py:124-135); one kernel
 * instead of autograd's extra add over [B*S, H]. */
int mmt_ln_bwd_add(const mmt_rows_desc* desc, const void* dy, const void* x, const float* gamma,
                   const float* mean, const float* rstd, const void* dx_in, void* dx, float* dgamma, float* dbeta,
                   void* workspace, size_t workspace_bytes, void* stream);
int mmt_ln_bwd(const mmt_rows_desc* desc, const void* dy, const void* x, const float* gamma,
               const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta,
               void* workspace, size_t workspace_bytes, void* stream);

/* Residual block tail + next block's LayerNorm:
 *   x_new = x + Dropout(o + bias);   h = LayerNorm(x_new) * gamma + beta
 * gamma == NULL: no LayerNorm (h, mean, rstd unused) -- the last block of a pre-activation stack. */
int mmt_residual_block_fwd(const mmt_rows_desc* desc, const void* o, const float* bias,
                           const void* x, const float* gamma, const float* beta, void* x_new,
                           void* h, float* mean, float* rstd, void* stream);

/* Backward: dX = dx_new_in + LayerNormBwd(dh);  dx = dX;  do = DropoutBwd(dX);
 * dbias[H] = colsum(do), dgamma, dbeta overwritten.  dx_new_in may be NULL (treated as 0);
 * dh / gamma NULL together when the forward had no LayerNorm. */
int mmt_residual_block_bwd(const mmt_rows_desc* desc, const void* dx_new_in, const void* dh,
                           const void* x_new, const float* gamma, const float* mean,
                           const float* rstd, void* d_o, void* dx, float* dbias, float* dgamma,
                           float* dbeta, void* workspace, size_t workspace_bytes, void* stream);

/* Second half of a *_bwd call made with desc->defer_reduce: sums the partial column sums the first
 * half left in `workspace` into (o0, o1, o2) in fixed order (overwrite, or add with desc->accumulate).
 * kind: 0 = mmt_ln_bwd (dgamma, dbeta), 1 = mmt_residual_block_bwd with LayerNorm (dbias, dgamma, dbeta),
 * 2 = mmt_residual_block_bwd without LayerNorm (dbias), 3 = mmt_bias_gelu_bwd (dbias). */
int mmt_colsum_reduce(const mmt_rows_desc* desc, int32_t kind, const void* workspace, float* o0, float* o1,
                      float* o2, void* stream);

/* The same for up to 48 deferred reduces in ONE launch (a host that needs no parameter gradient before the backward
 * pass ends queues them: ~32 five-microsecond launches per step otherwise).  rows / H / kind / accumulate as in the
 * desc and call they stand for. */
typedef struct mmt_colsum_item {
  const void* workspace;
  float* o0; float* o1; float* o2;
  int64_t rows;
  int32_t H, kind, accumulate, reserved;
} mmt_colsum_item;
int mmt_colsum_reduce_batch(int32_t n, const mmt_colsum_item* items, void* stream);

/* y = gelu_tanh(u + bias)   (DenseLayers hidden activation; rows x H with H = intermediate_size) */
int mmt_bias_gelu_fwd(const mmt_rows_desc* desc, const void* u, const float* bias, void* y,
                      void* stream);

/* du = dy * gelu_tanh'(u + bias);  dbias[H] = colsum(du) (overwritten). */
int mmt_bias_gelu_bwd(const mmt_rows_desc* desc, const void* dy, const void* u, const float* bias,
                      void* du, float* dbias, void* workspace, size_t workspace_bytes, void* stream);

/* acc[i] += g[i] for i < n: fp32 master-gradient accumulation of a low-precision gradient
 * (the `AccumulateGrad` step behind `optimizer.apply_gradients`, src/tasks/pretraining.py:262-273).
 * g_dtype: MMT_F32 | MMT_BF16; acc must be 16-byte aligned, g 8-byte aligned. */
/* out[c] (+)= sum_rows x[row, c] for a [rows, C] matrix (fp32 | bf16) with row stride ld >= C in elements and no
 * alignment requirement beyond the element's: the bias gradient dy.sum(0) of a Dense layer (`tape.gradient`,
 * src/tasks/pretraining.py:292-296) -- e.g. the output bias of the 30522-way MLM logits, whose rows start on odd
 * 4-byte boundaries.  Fixed-order sums (16 row groups, then the groups in order). */
size_t mmt_colsum_workspace_bytes(int64_t rows, int32_t C);
int mmt_colsum(int64_t rows, int32_t C, int32_t dtype, const void* x, int64_t ld, float* out, int32_t accumulate,
               void* workspace, size_t workspace_bytes, void* stream);
int mmt_accumulate_grad(float* acc, const void* g, int32_t g_dtype, int64_t n, void* stream);

/* Global-norm clip factor of the gradient slabs (`optimizer_config.gradient_clip_norm`, the reference's trainer
 * config src/configs/pretraining_experiments.py:24-47, applied after the all-reduce of pretraining.py:273):
 *   norm = pending_scale * sqrt(sum over all slabs of g^2),   *scale_out = min(1, max_norm / (norm + 1e-6)) * pending_scale
 * in one streaming pass + a fixed-order sum of 2048 per-block partials (bitwise reproducible).  `pending_scale` is a
 * factor (e.g. 1/replicas) the slabs have not been multiplied with yet; `scale_out` (device float) is what
 * mmt_adamw_step takes as grad_scale; `norm_out` (device float) may be NULL.  Up to 16 slabs, each 16-byte aligned
 * with a multiple of 4 elements; workspace: 2048 floats. */
int mmt_grad_clip_scale(int32_t n_slabs, const float* const* slabs, const int64_t* sizes, float max_norm,
                        float pending_scale, float* scale_out, float* norm_out, void* workspace,
                        size_t workspace_bytes, void* stream);

/* One AdamW step over a flat parameter slab (the optimizer of the reference's trainer config,
 * src/configs/pretraining_experiments.py:24-47: adamw, weight_decay_rate 0.01 with the
 * LayerNorm/bias exclusion list, applied after the gradient all-reduce of
 * src/tasks/pretraining.py:273).  Per element i of chunk c = i / 1024:
 *   g = grad[i] * (*grad_scale)                      (global-norm clip factor, device scalar, may be NULL)
 *   p = param[i] * (1 - lr * chunk_wd[c]);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2
 *   p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps)            (torch.optim.AdamW semantics)
 * writes param, exp_avg, exp_avg_sq (fp32), the bf16 shadow copy (nullable) and, with
 * zero_grad != 0, clears grad.  n must be a multiple of 1024; all buffers 16-byte aligned. */
typedef struct mmt_adamw_desc {
  int64_t n;
  float lr, beta1, beta2, eps;
  float bias_correction1, bias_correction2;   /* 1 - beta^t */
  int32_t zero_grad;
  int32_t reserved;
  const float* hyper;    /* ABI 4: DEVICE {lr, bias_correction1, bias_correction2} read by the kernel INSTEAD of the three
                            fields above, or NULL (mmt_attn.h: mmt_write_step_scalars)                                 */
} mmt_adamw_desc;

int mmt_adamw_step(const mmt_adamw_desc* desc, float* param, float* grad, float* exp_avg,
                   float* exp_avg_sq, void* param_bf16, const float* chunk_wd,
                   const float* grad_scale, void* stream);

/* Weight gradient of a Dense layer, accumulated into the fp32 master gradient:
 *   dw[M,N] += dy[K,M]^T . x[K,N]        (bf16 operands, fp32 accumulation, float atomics)
 * = the per-layer `tape.gradient` product + `AccumulateGrad` of src/tasks/pretraining.py:262-296.
 * Requires M % 128 == 0, N % 256 == 0 (any K), 16-byte aligned operands, ld* in elements
 * (ldy, ldx multiples of 8); returns MMT_E_UNSUPPORTED otherwise (callers fall back to a
 * library GEMM).  With a workspace of mmt_wgrad_workspace_bytes() the split-K partials are
 * written as plain fp32 slabs and summed in fixed order (bitwise reproducible); without one they
 * are added with float atomics (order not fixed). */
size_t mmt_wgrad_workspace_bytes(int32_t M, int32_t N, int64_t K);
/* Compute units one weight-gradient GEMM may fill (default 256 = the whole MI355X, clamped to [32, 256]).
 * Process-wide; a data-parallel host lowers it so that the split-K grid and the collective kernels
 * overlapping backward fit the chip together.  Affects mmt_wgrad_workspace_bytes: query after setting. */
void mmt_wgrad_set_cu_budget(int32_t cus);
int mmt_wgrad_accumulate(float* dw, int64_t ldw, const void* dy, int64_t ldy, const void* x,
                         int64_t ldx, int32_t M, int32_t N, int64_t K, void* workspace,
                         size_t workspace_bytes, void* stream);
/* Same, and additionally dbias[M] += column sums of dy (the Dense layer's bias gradient, fp32): the
 * workgroups of the first column tile add up the dy chunks they stage for the product anyway.
 * dbias == NULL: exactly mmt_wgrad_accumulate. */
int mmt_wgrad_bias_accumulate(float* dw, int64_t ldw, float* dbias, const void* dy, int64_t ldy,
                              const void* x, int64_t ldx, int32_t M, int32_t N, int64_t K,
                              void* workspace, size_t workspace_bytes, void* stream);

/* Up to 28 such products in ONE launch -- the weight gradients of the four Dense layers of one to seven encoder blocks
 * (`tape.gradient`, src/tasks/pretraining.py:292-296), which contract the same K = B*S rows.  Alone, the small
 * products need a deep split of K to fill the chip and write one fp32 slab per slice; together two slices
 * suffice for one block (108 tiles at BERT-base dims: 4x less slab traffic) and for two blocks K is not split
 * at all: every tile belongs to one workgroup, which adds it into dw with plain stores (no slabs, no reduce).
 * With more tiles than compute units (three blocks and up) the tiles go in whole rounds of one workgroup per tile over
 * the whole K, and only the tail that does not fill a round is split -- into compact per-tile slabs, reduced by a small
 * second launch (seven blocks: 756 tiles = 2 rounds + 244 unsplit; five: 2 rounds + 28 tiles x 8 slices).
 * Every problem needs M % 256 == 0, N % 256 == 0; K % 64 == 0; the workspace (mmt_wgrad_group_workspace_bytes, may be
 * 0 bytes) is required.  dbias may be NULL per problem.  Results are bitwise those of a fixed-order sum over the slices. */
typedef struct mmt_wgrad_problem {
  float* dw;        /* [M, N] fp32, row stride ldw, accumulated into */
  int64_t ldw;
  float* dbias;     /* [M] fp32 += column sums of dy, or NULL */
  const void* dy;   /* [K, M] bf16, row stride ldy */
  int64_t ldy;
  const void* x;    /* [K, N] bf16, row stride ldx */
  int64_t ldx;
  int32_t M, N;
} mmt_wgrad_problem;
size_t mmt_wgrad_group_workspace_bytes(int32_t n, const mmt_wgrad_problem* problems, int64_t K);
int mmt_wgrad_grouped(int32_t n, const mmt_wgrad_problem* problems, int64_t K, void* workspace,
                      size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Embedding assembly of MmtEncoder.call (src/modeling/models/mmt_encoder.py:189-218):
 *   we  = Dropout(LayerNorm_eps(WordEmb[word_ids]))          LN + dropout on the WORD embeddings only
 *   out = we + SegEmb[segment_ids] (+ PosEmb[s]) (+ patch_proj[b, s - patch_start] + patch_bias
 *                                                 for patch_start <= s < patch_start + n_patch)
 * in one pass per row (one wave per row; tables fp32, `out` in `dtype`).  What TF runs as gather /
 * one-hot matmul, LayerNormalization, Dropout, three Adds, a Pad and a Cast.
 * Ids outside [0, vocab) contribute a zero row (one-hot lookup semantics, mmt_encoder.py:110).
 * patch_proj is the [B, n_patch, H] output of the patch projection GEMM WITHOUT its bias, in `dtype`. */
typedef struct mmt_embed_desc {
  int64_t rows;          /* B * S                                                        */
  int32_t S;             /* sequence length: row = b * S + s                             */
  int32_t H;             /* hidden size = embedding size; multiple of 8, <= 2048         */
  int32_t dtype;         /* MMT_F32 | MMT_BF16: out / patch_proj / dout / dpatch         */
  int32_t vocab;         /* rows of word_table                                           */
  int32_t seg_vocab;     /* rows of seg_table                                            */
  int32_t patch_start;   /* 2 in the reference ([CLS], [PATCH] first; mmt_encoder.py:213-218) */
  int32_t n_patch;       /* 0: no patch term                                             */
  float eps;             /* 1e-12                                                        */
  float dropout_p;       /* hidden_dropout_prob; 0 disables                              */
  int32_t accumulate;    /* backward: != 0 adds dgamma / dbeta INTO the given buffers     */
  uint64_t dropout_seed;
  const uint64_t* dropout_epoch; /* ABI 4: DEVICE uint64 added to dropout_seed by the kernel, or NULL (mmt_attn.h) */
} mmt_embed_desc;

int mmt_embed_fwd(const mmt_embed_desc* desc, const int32_t* word_ids, const int32_t* seg_ids,
                  const float* word_table, const float* seg_table, const float* pos_table /* nullable */,
                  const float* gamma, const float* beta, const void* patch_proj /* nullable */,
                  const float* patch_bias /* nullable */, void* out, float* mean, float* rstd,
                  void* stream);

/* Bytes of scratch mmt_embed_bwd needs. */
size_t mmt_embed_workspace_bytes(const mmt_embed_desc* desc);

/* Backward of mmt_embed_fwd for the word path and the patch slice:
 *   dword_table[id] += LayerNormBwd(DropoutBwd(dout[row]))   summed over the rows holding id,
 *   dgamma / dbeta (overwritten, or added to with desc->accumulate),
 *   dpatch[b, j] = dout[b, patch_start + j]                   (compact copy for the projection's wgrad GEMM; nullable).
 * `order` = the permutation that sorts word_ids ascending (stable: ties in row order) and
 * `sorted_ids[i] = word_ids[order[i]]` -- rows with the same id are then adjacent and are summed in that fixed order by ONE wave per id (runs longer
 * than 32 go through per-32 partial sums), so the scatter needs no atomics and is bitwise reproducible.
 * dword_table is fp32 [vocab, H] and is ACCUMULATED into (the master gradient).  The segment /
 * position table gradients are plain column sums of dout and are left to the caller. */
int mmt_embed_bwd(const mmt_embed_desc* desc, const void* dout, const int32_t* sorted_ids,
                  const int32_t* order, const float* word_table, const float* gamma, const float* mean,
                  const float* rstd, float* dword_table, float* dgamma, float* dbeta, void* dpatch,
                  void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Per-row softmax cross-entropy of wide logits (the tied 30522-way MLM head):
 *   loss[row] = logsumexp(logits[row, :]) - logits[row, labels[row]]       (natural log; lse saved)
 * = the `unweighted` term of src/modeling/losses/weighted_sparse_categorical_crossentropy_loss.py:17-43;
 * the weighting and divide_no_nan reduction over a few hundred rows stay with the caller.
 * logits are read once in their storage dtype (MMT_F32 | MMT_BF16), row stride `ld` elements.
 * A label outside [0, C) means "no target": loss 0, zero gradient.
 * Backward: dlogits[row, i] = (softmax(logits[row])[i] - [i == label]) * coef[row], written in `dtype`. */
int mmt_xent_fwd(int64_t rows, int32_t C, int32_t dtype, const void* logits, int64_t ld,
                 const int32_t* labels, float* loss, float* lse, void* stream);
/* The same pass also reporting argmax[row] = the FIRST index of the row's largest logit -- what tf.argmax, hence
 * tf.keras.metrics.SparseCategoricalAccuracy (src/tasks/pretraining.py:183-222), compares with the label -- so that the
 * accuracy metrics cost no second sweep over the 30522-way logits.  `argmax` may be NULL. */
int mmt_xent_fwd_argmax(int64_t rows, int32_t C, int32_t dtype, const void* logits, int64_t ld,
                        const int32_t* labels, float* loss, float* lse, int32_t* argmax, void* stream);
int mmt_xent_bwd(int64_t rows, int32_t C, int32_t dtype, const void* logits, int64_t ld,
                 const int32_t* labels, const float* lse, const float* coef, void* dlogits, int64_t ldd,
                 void* stream);
/* The same with one more factor read on the device: dlogits *= gscale[0] (gscale may be NULL) -- the upstream
 * gradient of a scalar loss, so that the host never reads it. */
int mmt_xent_bwd_scaled(int64_t rows, int32_t C, int32_t dtype, const void* logits, int64_t ld,
                        const int32_t* labels, const float* lse, const float* coef, const float* gscale,
                        void* dlogits, int64_t ldd, void* stream);
/* The reduction of `weighted_sparse_categorical_crossentropy_loss`
 * (src/modeling/losses/weighted_sparse_categorical_crossentropy_loss.py:36-43) and of the loss bookkeeping around it
 * (src/tasks/pretraining.py:95-140: MLM / MPP weights masked by the example's ITM label) in one launch:
 *   w_i = weight[i] * (mask ? mask[i / mask_div] : 1),   l_i = loss[i] * (lmul ? lmul[i] : 1)
 *   num = sum_i w_i l_i,  den = sum_i w_i,  out3 = { divide_no_nan(num, den), num, den }
 *   coef[i] = d out3[0] / d loss[i] = den != 0 ? w_i lmul_i / den : 0        (coef may be NULL)
 * All arrays fp32 on the device; one workgroup, sums in a fixed order (bitwise reproducible). */
int mmt_weighted_loss(int64_t rows, const float* loss, const float* weight, const float* lmul, const float* mask,
                      int64_t mask_div, float* out3, float* coef, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Feed-forward GEMMs with the GELU in the epilogue (bf16 operands, fp32 accumulate; K11, csrc/ffn_gemm.hip).
 * They replace, for the intermediate Dense of the encoder block (reference: the `inner_activation` Dense pair
 * of every TransformerEncoderBlock the encoder stacks, mmt_encoder.py:53-54 + 221-238), the sequence
 * "library GEMM -> mmt_bias_gelu_fwd" and, in the tape's backward, "library GEMM -> mmt_bias_gelu_bwd":
 *   mmt_ffn_gelu_gemm :  u[M,N] = x[M,K] . w[N,K]^T + bias[N]   (u optional, rounded to bf16)
 *                        g[M,N] = gelu_tanh(u)                    (of the ROUNDED u, so that the backward, which
 *                                                                  only has the stored u, sees the same function)
 *   mmt_ffn_dgelu_gemm:  du[M,N] = (dy[M,K] . w[K,N]) * gelu_tanh'(u[M,N] + bias[N])      (bias may be NULL)
 * All matrices row-major with row strides ld* in elements (multiples of 8), 16-byte aligned; bias fp32.
 * Needs M % 256 == 0, N % 256 == 0, K % 64 == 0, otherwise MMT_E_UNSUPPORTED (nothing launched).
 * The kernels are persistent, one workgroup per compute unit: mmt_ffn_set_cu_budget (default 256, clamped to
 * [32, 256], process-wide) sizes the grid, e.g. to leave units to collective kernels that overlap backward. */
void mmt_ffn_set_cu_budget(int32_t cus);
int mmt_ffn_gelu_gemm(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias, void* u,
                      int64_t ldu, void* g, int64_t ldg, int64_t M, int64_t N, int64_t K, void* stream);
int mmt_ffn_dgelu_gemm(const void* dy, int64_t lddy, const void* w, int64_t ldw, const void* u, int64_t ldu,
                       const float* bias, void* du, int64_t lddu, int64_t M, int64_t N, int64_t K, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMT_LAYER_H_ */
