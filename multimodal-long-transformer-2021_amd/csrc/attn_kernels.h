// Host <-> kernel parameter blocks and launcher prototypes (internal; the public
// boundary is include/mmt_attn.h).
#pragma once
#include "mmt_common.h"

namespace mmt {

enum { kBand = 0, kDense = 1, kRows = 2 };

struct FwdParams {
  const void *q, *k, *v, *emb, *bias;
  void* out;
  float* lse;
  const int32_t *att_mask, *rel_ids;  // dense mode
  const int32_t* valid_len;
  int B, S, N, R;
  long qs[3], ks[3], vs[3], os[3];    // element strides of (b, s, n)
  float sscale;    // scale * log2(e)                          (multiplies q.k)
  float tscale;    // factor folded into the relative table (scale*log2e, or log2e)
  float mask_add;  // mask_value * log2(e)
  PatternDev pat;
  int skip_global_rows;  // kBand: rows of global tokens are produced by the kRows pass
  // dropout
  uint32_t drop_thresh;  // 16-bit threshold: keep iff bits16 >= thresh; 0 disables
  uint32_t seed_lo, seed_hi;
  const unsigned long long* epoch;      // device-resident addend of the seed (mmt_attn_desc.dropout_epoch) or NULL
  float inv_keep;
  // kRows
  float* part_o;
  float* part_ml;
  float part_scale;   // combine: factor on the summed partial rows (1 / keep when the lean kernel left it to the epilogue)
  int n_chunks, chunk_tiles, n_rowblk;
  int n_band_blocks;  // B*N*ceil(S/128): blocks before the global-row items
  int perm_1d;        // 1-D ids with R >= 2m+1: table columns permuted, fast path allowed
  int lean_rp;        // lean 2-D path: table width (32 | 64) that holds every id that can contribute
  int rows_only;      // lean band kernel: launch holds the global-row items only (the window kernel produced the band rows)
  int rows_parts;     // window kernel: workgroups per (plane, 8 global rows), each walking 1 / rows_parts of the keys (> 1 needs
                      // `sync` and `walk_part`: the plane's last arriver merges the parts)
  int tstride;        // window / walk kernels: row stride (floats) of the per-wave relative-score table
  // plane-walk kernel (attn_fwd_walk.hip): runs per plane (walk_nseg, + 1 for the first walk_nhi planes of every XCD
  // group), XCD groups (8 | 1), partial slots per plane; partials of the global rows [B*N][walk_maxseg][8][66] floats;
  // arrival counters [B*N] (mmt_attn_desc.sync)
  int walk_groups, walk_nseg, walk_nhi, walk_maxseg;
  int pw_walk;        // sliding-window kernel (attn_fwd_pwin.hip): consecutive 128-row blocks per workgroup
  float* walk_part;
  unsigned* sync;
  long long* dbg;     // -DMMT_STAMP diagnostic builds only: in-kernel s_memtime stamps (never set in the product)
  int dbg_mode, dbg_sleep;   // -DMMT_STAMP builds only: ablations (1 = no tile loop, 2 = no DMA), start delay of the second resident round
};

hipError_t launch_attn_fwd(const FwdParams& p, int mode, bool bf16, hipStream_t st);
hipError_t launch_rows_combine(const FwdParams& p, bool bf16, hipStream_t st);
hipError_t launch_attn_fwd_band_bf16(const FwdParams& p, hipStream_t st);   // attn_fwd_band.hip
hipError_t launch_attn_fwd_win_bf16(const FwdParams& p, hipStream_t st);    // attn_fwd_win.hip
int fwd_win_lds_bytes(int ng, int tstride);
hipError_t launch_attn_fwd_walk_bf16(const FwdParams& p, int grid_size, hipStream_t st);   // attn_fwd_walk.hip
int fwd_walk_lds_bytes(int ng, int tstride, bool rel);
int fwd_walk_plan(FwdParams& p, int target_wgs);
size_t fwd_walk_workspace_bytes(int B, int N, int S);
hipError_t launch_attn_fwd_pwin_bf16(const FwdParams& p, int grid_size, hipStream_t st);   // attn_fwd_pwin.hip
int fwd_pwin_plan(FwdParams& p, int target_wgs);
size_t fwd_pwin_workspace_bytes(int B, int N, int S, int target_wgs);

struct BwdParams {
  const void *q, *k, *v, *emb, *bias, *out, *dout;
  const float* lse;
  const int32_t *att_mask, *rel_ids;
  const int32_t* valid_len;
  void *dq, *dk, *dv;
  float *drel_emb, *drel_bias;        // outputs [R,N,64], [R,N] fp32
  int drel_accum;                     // != 0: add to them instead of overwriting
  int red_per_plane, red_live;        // dE partials per (b,n) plane in part_red: slots, and how many of them are written
  int dstride;                        // lean dQ kernel: row stride (floats) of the per-wave dRel table in LDS
  int comb_in_next;                   // lean path: dQ combine rides in the dK/dV launch, dK/dV combine in the dE reduce
  int B, S, N, R, Rp;
  long qs[3], ks[3], vs[3], os[3];    // q/dq, k/dk, v/dv, out/dout
  float sscale, tscale, mask_add;     // as FwdParams (log2 domain)
  float gscale;                       // d(content) = ds * gscale        (= scale)
  float rel_gscale;                   // d(rel)     = ds * rel_gscale    (= scale, or 1)
  PatternDev pat;
  int perm_1d;
  int lean2d;                         // 2-D ids on the lean path (Rp already narrowed to the ids that can contribute)
  int skip_global;                    // band items leave global rows / keys to the split items
  uint32_t drop_thresh, seed_lo, seed_hi;
  const unsigned long long* epoch;      // as FwdParams
  float inv_keep;
  // workspace
  float* delta;      // [B,N,S]       rowsum(dO * O)
  float* relfar;     // [B,N,S,2]     clipped-id relative scores of each row (log2 domain), lean path
  float* drel;       // [B*N, n_global, Rp]  d(relall) rows of the global tokens, id order
  float* part_dq;    // [B*N, n_gblk, n_chunks, 32, 64]   global-row partials
  float* part_dtab;  // [B*N, n_gblk, n_chunks, 32, Rp]
  float* part_dkv;   // [B*N, n_gblk, dkv_slots, 2, 32, 64] global-key partials
  float* part_red;   // [B*N * ceil(S/128) * 4 waves, Rp*64 + Rp]  per-wave dE^T / dbias partials
  int n_band_blocks, n_chunks, chunk_tiles, n_gblk, n_split;
  int peel_gkeys;       // bit 0, dQ pass: the (<= 8) global keys outside a wave's band tiles as a peeled quarter-tile step; bit 1, dK/dV pass: the global query rows likewise
  // P hand-over (lean bf16 path, attn_bwd_band.hip): the dQ pass stores every tile's probabilities as bf16 (sign bit =
  // dropped) and the dK/dV pass (attn_bwd_dkv_ho_kernel) rebuilds P' and dS from them instead of recomputing scores,
  // exponentials, masks and keep hashes.  Regions of `ho` (bytes): band tiles [B*N][n_tiles q blocks][ho_slots][2 KiB],
  // then global-key strips [B*N][n_tiles q blocks][512] (32 rows x 8 keys), then global-row tiles
  // [B*N][n_tiles key blocks][512] (8 rows x 32 keys).  NULL = off.
  unsigned char* ho;
  int ho_slots;         // band key tiles per 32-row q block: 2 * ceil(radius / 32) + 1
  int dkv_slots;        // partial slots per (plane, global block) in part_dkv: n_chunks, + 1 with the hand-over (the
                        // band part of the global keys, written by the band key waves)
  int ho_per_wave;      // hand-over dK/dV pass in its per-wave form even where the window form fits (MMT_TUNE_BWD_HO_PER_WAVE)
  int dq_plane_major;   // dQ pass: plane-major block placement (attn_lean.h) instead of long-items-first + XCD remap
  long long* dbg;    // -DMMT_STAMP diagnostic builds only (see FwdParams)
  int dbg_mode;
};

hipError_t launch_attn_bwd(const BwdParams& p, int mode, bool bf16, hipStream_t st);
hipError_t launch_attn_bwd_band_bf16(const BwdParams& p, hipStream_t st);   // attn_bwd_band.hip
hipError_t launch_bwd_dq_combine(const BwdParams& p, bool bf16, hipStream_t st);
hipError_t launch_bwd_dkv_combine(const BwdParams& p, bool bf16, hipStream_t st);
hipError_t launch_drel_reduce(const BwdParams& p, bool bf16, hipStream_t st);

struct SideParams {
  PatternDev pat;
  const int32_t* gidx;     // listed global positions (ascending, pat.ng of them) or NULL: the range [pat.g0, pat.g0 + pat.ng)
  int B, S;
  const int32_t *img_wp, *txt_wp;
  int materialize_pattern;
  int32_t *att_mask, *rel_ids, *segment_ids;
};
hipError_t launch_side_inputs(const SideParams& p, hipStream_t st);
// mmt_write_step_scalars: one thread writes the step's epoch and {lr, bias corrections} (side_inputs.hip)
hipError_t launch_write_step_scalars(unsigned long long* epoch_dst, float* hyper_dst, unsigned long long epoch, float lr,
                                     float bc1, float bc2, hipStream_t st);

}  // namespace mmt
