/* mmt_attn.h -- C ABI of the MI355X (gfx950) relative-attention hot path.
 *
 * Drop-in boundary for ONE path of googleinterns/multimodal-long-transformer-2021:
 * the relative attention that `MmtEncoder` runs through
 * `etc_layers.RelativeTransformerLayers(inputs, att_mask, relative_att_ids)`
 * (src/modeling/models/mmt_encoder.py:124-135, call :220-224) and the integer
 * side inputs that feed it (src/data/data_utils.py:285-380, src/feature_utils.py:29).
 * The reference has no FFI of its own (it is pure Python/TF); these are the entry
 * points a binding for that path would call -- see INTEGRATION.md for the ctypes
 * stub and the tf.py_function / custom-op shape of the call.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory unless it says host;
 *  - the caller owns every buffer; the library allocates nothing persistent;
 *  - every launch goes to the caller's hipStream_t (passed as void*), no implicit sync;
 *  - return 0 on success, a negative MMT_E_* code on failure; the message of the last
 *    failure on the calling thread is available from mmt_last_error(); nothing throws
 *    or aborts across this boundary;
 *  - re-entrant across threads and streams: no global mutable state.  Everything a call depends on travels in its
 *    descriptor and arguments (ABI 4: the device-resident step scalars and the kernel-selection switches too; the
 *    library reads no environment variable).  The one exception is speed-only: the two atomic compute-unit budgets
 *    of mmt_layer.h (mmt_wgrad_set_cu_budget / mmt_ffn_set_cu_budget), which size grids and never change a result.
 */
#ifndef MMT_ATTN_H_
#define MMT_ATTN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMT_ABI_VERSION 4

enum {
  MMT_OK = 0,
  MMT_E_INVALID = -1,     /* bad descriptor / null pointer / unsupported shape */
  MMT_E_UNSUPPORTED = -2, /* valid request this build has no kernel for         */
  MMT_E_WORKSPACE = -3,   /* workspace missing or too small                     */
  MMT_E_LAUNCH = -4       /* HIP reported a launch error                        */
};

enum { MMT_F32 = 0, MMT_BF16 = 1 };

/* How relative ids are produced for the structured (no dense [S,S] input) path. */
enum {
  MMT_IDS_NONE = 0, /* no relative term                                                   */
  MMT_IDS_1D = 1,   /* etcmodel RelativePositionGenerator over the whole sequence
                       (data_utils.py:300-301): id = min(j-i,m) | m+min(i-j,m)           */
  MMT_IDS_2D = 2    /* MmtRelativePositionGenerator (feature_utils.py:114-184): 2-D ids
                       for the first P*P positions, part ids across, 1-D ids for the rest */
};

/* flags */
#define MMT_FLAG_SCALE_BEFORE_ADD 1u /* s = content*scale + rel  (default: (content+rel)*scale) */
#define MMT_FLAG_ACCUM_REL_GRADS 2u  /* backward: drel_emb / drel_bias += (fp32 master gradients) instead of = */

/* Kernel-selection switches (mmt_attn_desc.tuning; ABI 4).  0 = the library's defaults.  They choose between kernels
 * that compute the same result (parity tests flip them to reach every kernel; they replace the MMT_* environment
 * variables the library read until ABI 3). */
#define MMT_TUNE_FWD_WALK 0x01u          /* forward: the plane-walk kernel (attn_fwd_walk.hip) where it covers the shape  */
#define MMT_TUNE_FWD_PWIN 0x100u         /* forward: the persistent sliding-window kernel (attn_fwd_pwin.hip), likewise */
#define MMT_TUNE_FWD_ROWS_ONE_WG 0x200u   /* forward, window kernel: one workgroup per (plane, 8 global rows) even with `sync` */
#define MMT_TUNE_FWD_NO_WIN 0x02u        /* forward: not the window kernel (attn_fwd_win.hip) either: per-wave staging  */
#define MMT_TUNE_FWD_FORCE_WIN 0x04u     /* forward: window kernel whenever the shape is covered, whatever its LDS need */
#define MMT_TUNE_BWD_NO_HANDOVER 0x08u   /* backward: the dK/dV pass recomputes P instead of reading the dQ pass's      */
#define MMT_TUNE_BWD_HO_PER_WAVE 0x10u   /* backward: hand-over dK/dV pass in its per-wave form (no workgroup window)   */
#define MMT_TUNE_BWD_NO_PEEL_DQ 0x20u    /* backward: global keys as a tile visit of the dQ pass, not a peeled step     */
#define MMT_TUNE_BWD_NO_PEEL_DKV 0x40u   /* backward: likewise for the global query rows of the recomputing dK/dV pass  */
#define MMT_TUNE_BWD_DQ_PLANE_MAJOR 0x80u /* backward: plane-major block placement of the dQ pass                       */

/* Attention pattern + id generator.  With local_radius >= S and n_global == 0 the
 * pattern is exactly the reference's segmented mask (data_utils.py:321-322):
 *   mask(q,k) = (q < valid_len[b]) == (k < valid_len[b])
 * otherwise (build-defined):
 *   mask(q,k) = segmented(q,k) && (|q-k| <= local_radius || global(q) || global(k)),
 *   global(x) = global_start <= x < global_start + n_global, or -- with global_index --
 *   x is one of the n_global listed positions.  The structured kernels take the contiguous
 *   form only; a listed set is served by materialising the mask (mmt_side_inputs with
 *   materialize_pattern) and calling the dense operator with it.                       */
typedef struct mmt_mask_desc {
  const int32_t* valid_len; /* [B] device ints (num_image_wordpieces + num_text_wordpieces),
                               NULL = every position valid                            */
  int32_t local_radius;     /* >= 0; values >= S mean "no band restriction"          */
  int32_t global_start;
  int32_t n_global;         /* contiguous range of global tokens; 0 = none           */
  int32_t id_mode;          /* MMT_IDS_*                                             */
  int32_t max_dist;         /* relative_pos_max_distance m (encoders.py:60)          */
  int32_t patches_per_row;  /* P  = image_size // patch_size        (MMT_IDS_2D)     */
  int32_t core_layers;      /* r  = relative_att_num_core_layers    (MMT_IDS_2D)     */
  const int32_t* global_index; /* NULL: the contiguous range above; else n_global ascending,
                               distinct positions in [0, S) on the device (global_start
                               is ignored).  ABI 2.                                     */
} mmt_mask_desc;

/* One attention call: q,k,v,out are [B,S,N,D] views with element strides (D contiguous). */
typedef struct mmt_attn_desc {
  int32_t B, S, N, D;   /* D must be 64                                               */
  int32_t R;            /* rows of relative_emb_table (relative_vocab_size); 0 = none; at most 128 (above 64: the general kernels) */
  int32_t dtype;        /* MMT_F32 | MMT_BF16 : element type of q,k,v,out,rel tables  */
  int64_t q_stride[3];  /* element strides of (b, s, n) for q                         */
  int64_t k_stride[3];
  int64_t v_stride[3];
  int64_t o_stride[3];  /* out, dout                                                  */
  float scale;          /* 1/sqrt(D)                                                  */
  float mask_value;     /* additive value for masked keys: -10000.0 (App. A.3)        */
  uint32_t flags;       /* MMT_FLAG_*                                                 */
  float dropout_p;      /* attention_probs_dropout_prob; 0 disables                   */
  uint64_t dropout_seed;
  mmt_mask_desc mask;   /* used when att_mask / rel_ids pointers are NULL             */
  /* ---- ABI 4 ---- */
  const uint64_t* dropout_epoch; /* DEVICE uint64 the kernels add to dropout_seed when they start, or NULL.  For callers
                                    that record a train step once as a HIP graph and replay it: kernel arguments are
                                    frozen in a graph, this word is not (mmt_write_step_scalars).  Forward and backward
                                    of one call must see the same value.                                              */
  uint32_t tuning;               /* MMT_TUNE_* switches; 0 = defaults                                                 */
  uint32_t sync_words;           /* length of `sync` in 32-bit words                                                   */
  uint32_t* sync;                /* DEVICE scratch of arrival counters for kernels whose workgroups combine partial
                                    results inside ONE launch (the forward's plane walk: B*N words), or NULL (such
                                    kernels are then not taken).  Caller-owned like every buffer; must be all zero
                                    before its first use; every call leaves it all zero again; calls that share it
                                    must be ordered on one stream.                                                    */
} mmt_attn_desc;

int mmt_abi_version(void);

/* Message of the last failure on this thread ("" if none).  Never NULL. */
const char* mmt_last_error(void);

/* Per-step scalars that live in DEVICE memory, for callers that record a train step once as a HIP graph and replay it
 * (kernel arguments are frozen in a graph; these words are read by the kernels when they start):
 *   dropout_epoch : one uint64, named by the `dropout_epoch` field of mmt_attn_desc / mmt_rows_desc / mmt_embed_desc:
 *                   the kernel adds it to the descriptor's dropout_seed -- forward and backward alike, so a backward
 *                   still regenerates its forward's mask.
 *   adamw_hyper   : three floats {lr, bias_correction1, bias_correction2}, named by mmt_adamw_desc.hyper, that
 *                   mmt_adamw_step uses INSTEAD of the fields of its descriptor.
 * The library keeps no registration of them (ABI 3's process-wide mmt_set_step_scalars is gone): two models, devices or
 * threads in one process each pass their own words.  Replaces nothing in the reference: its train step
 * (src/tasks/pretraining.py:224-298) is a tf.function whose step-dependent values are tf.Variables for the same reason.
 *
 * mmt_write_step_scalars writes one step's values into two such locations with ONE small launch on `stream` (the values
 * travel as kernel arguments, so the host may be any number of steps ahead of the device).  Either destination may be NULL. */
int mmt_write_step_scalars(uint64_t* dropout_epoch, float* adamw_hyper, uint64_t epoch, float lr,
                           float bias_correction1, float bias_correction2, void* stream);

/* Bytes of scratch mmt_attn_fwd / mmt_attn_bwd need for this descriptor (the larger of
 * the two).  Host-only computation. */
size_t mmt_workspace_bytes(const mmt_attn_desc* desc);

/* Forward of QkvRelativeAttention (replaces the einsum/one-hot/softmax
 * chain behind mmt_encoder.py:220-224):
 *   s = (q.k + relall[q, id(q,k)]) * scale + (1 - mask(q,k)) * mask_value,
 *   relall[q,r] = q.rel_emb[r] (+ rel_bias[r]),   ids >= R contribute 0,
 *   out = softmax_k(s) . v,   lse[b,n,q] = log sum_k exp(s)  (fp32).
 * rel_emb [R,N,D], rel_bias [R,N] (nullable) in desc->dtype.
 * att_mask / rel_ids: dense int32 [B,S,S] exactly as the reference feeds them (the
 * literal operator); pass NULL for both to use desc->mask (structured fast path, ids and
 * mask generated in-kernel, never materialised).  lse may be NULL. */
int mmt_attn_fwd(const mmt_attn_desc* desc, const void* q, const void* k, const void* v,
                 const void* rel_emb, const void* rel_bias, const int32_t* att_mask,
                 const int32_t* rel_ids, void* out, float* lse, void* workspace,
                 size_t workspace_bytes, void* stream);

/* Backward of the same operator.  dq,dk,dv in desc->dtype with q/k/v strides;
 * drel_emb [R,N,D] and drel_bias [R,N] are fp32 and are OVERWRITTEN, or added to when
 * desc->flags has MMT_FLAG_ACCUM_REL_GRADS (gradient accumulation into the master gradients).
 * drel_bias / drel_emb may be NULL when R == 0. */
int mmt_attn_bwd(const mmt_attn_desc* desc, const void* q, const void* k, const void* v,
                 const void* rel_emb, const void* rel_bias, const int32_t* att_mask,
                 const int32_t* rel_ids, const void* out, const void* dout, const float* lse,
                 void* dq, void* dk, void* dv, float* drel_emb, float* drel_bias,
                 void* workspace, size_t workspace_bytes, void* stream);

/* Integer side inputs, bit-exact with the reference's tf.data stage
 * (data_utils.py:335-379: segment_ids :350-361, att_mask :321-322, relative_att_ids
 * :326-329 via feature_utils.py:114-184 or the etcmodel 1-D generator).
 * num_image_wordpieces / num_text_wordpieces: [B] device ints.  Any output may be NULL.
 * att_mask_out / rel_ids_out: [B,S,S] int32, segment_ids_out: [B,S] int32.
 * mask->valid_len is ignored (valid = img + txt); with materialize_pattern != 0 the
 * band/global pattern of `mask` is intersected into att_mask_out (for equivalence tests). */
int mmt_side_inputs(const mmt_mask_desc* mask, int32_t B, int32_t S,
                    const int32_t* num_image_wordpieces, const int32_t* num_text_wordpieces,
                    int32_t materialize_pattern, int32_t* att_mask_out, int32_t* rel_ids_out,
                    int32_t* segment_ids_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MMT_ATTN_H_ */
