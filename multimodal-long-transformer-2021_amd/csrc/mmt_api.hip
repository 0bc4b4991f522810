// C ABI of the hot path (include/mmt_attn.h): validation, descriptor -> kernel parameter
// translation, launches on the caller's stream.  No allocation, no synchronisation, no
// global mutable state (the error message is thread-local; kernel-selection switches and device-resident step
// scalars travel in the descriptor).
#include "../../include/mmt_attn.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "attn_kernels.h"
#include "mmt_err.h"

namespace {
thread_local char g_err[512] = "";
}

namespace mmt {
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
}  // namespace mmt

namespace {
using mmt::fail;

constexpr int kChunkTiles = 8;  // kRows pass: 8 tiles = 256 keys per partial

struct Plan {
  bool dense;        // literal operator: att_mask / rel_ids from HBM
  bool split_rows;   // structured pattern with global ROWS handled by the kRows pass
  int n_rowblk, n_chunks;
  size_t fwd_ws;     // bytes
  size_t bwd_ws;
  int n_split;
  size_t off_delta, off_relfar, off_drel, off_pdq, off_pdtab, off_pdkv, off_red;  // float offsets
  size_t off_ho;     // float offset of the P / dS hand-over region (attn_kernels.h), 0 bytes when the shape has none
  int ho_slots;      // band key tiles per q block there (0 = no hand-over for this shape)
};

// P / dS hand-over between the two backward passes (lean bf16 kernels, 1-D or no relative ids): shapes it is built for.
// The global tokens, if any, must be the peeled kind (<= 8, contiguous); the band at most 8 tiles wide.
int handover_slots(const mmt_attn_desc* d, bool dense) {
  if (dense || d->dtype != MMT_BF16) return 0;
  if (d->mask.global_index || d->mask.n_global > 8) return 0;
  const int W = d->mask.local_radius > d->S ? d->S : d->mask.local_radius;
  const int slots = 2 * ((W + 31) / 32) + 1;
  return slots <= 8 ? slots : 0;
}

int check_desc(const mmt_attn_desc* d) {
  if (!d) return fail(MMT_E_INVALID, "desc is NULL");
  if (d->B <= 0 || d->S <= 0 || d->N <= 0) return fail(MMT_E_INVALID, "B,S,N must be positive");
  if (d->D != 64) return fail(MMT_E_UNSUPPORTED, "D=%d: only head size 64 is built", d->D);
  if (d->R < 0 || d->R > 128) return fail(MMT_E_UNSUPPORTED, "R=%d: relative vocab must be in [0,128]", d->R);
  if (d->dtype != MMT_F32 && d->dtype != MMT_BF16) return fail(MMT_E_INVALID, "bad dtype %d", d->dtype);
  const int64_t* st[4] = {d->q_stride, d->k_stride, d->v_stride, d->o_stride};
  const int align = d->dtype == MMT_BF16 ? 8 : 4;  // 16-byte row loads
  for (int t = 0; t < 4; ++t)
    for (int i = 0; i < 3; ++i)
      if (st[t][i] < 0 || st[t][i] % align) return fail(MMT_E_INVALID, "strides must be non-negative multiples of %d elements", align);
  for (int t = 0; t < 4; ++t)
    if ((int64_t)d->S * st[t][1] >= (int64_t)1 << 31) return fail(MMT_E_UNSUPPORTED, "S * stride_s must stay below 2^31 elements");
  if (!(d->dropout_p >= 0.f && d->dropout_p < 1.f)) return fail(MMT_E_INVALID, "dropout_p must be in [0,1)");
  const mmt_mask_desc& m = d->mask;
  if (m.local_radius < 0) return fail(MMT_E_INVALID, "local_radius must be >= 0");
  if (m.n_global < 0 || (!m.global_index && (m.global_start < 0 || m.global_start + m.n_global > d->S))) return fail(MMT_E_INVALID, "global range outside the sequence");
  if (m.id_mode < MMT_IDS_NONE || m.id_mode > MMT_IDS_2D) return fail(MMT_E_INVALID, "bad id_mode");
  if (m.id_mode != MMT_IDS_NONE && m.max_dist < 0) return fail(MMT_E_INVALID, "max_dist must be >= 0");
  if (m.id_mode == MMT_IDS_2D) {
    if (m.patches_per_row <= 0 || m.core_layers <= 0) return fail(MMT_E_INVALID, "2-D ids need patches_per_row > 0 and core_layers > 0");
    if ((int64_t)m.patches_per_row * m.patches_per_row > d->S) return fail(MMT_E_INVALID, "image part longer than the sequence");
  }
  return MMT_OK;
}

mmt::PatternDev make_pattern(const mmt_mask_desc& m, int S) {
  mmt::PatternDev p;
  p.radius = m.local_radius > S ? S : m.local_radius;
  p.g0 = m.global_start;
  p.ng = m.n_global;
  p.id_mode = m.id_mode;
  p.m = m.max_dist;
  p.P = m.patches_per_row > 0 ? m.patches_per_row : 1;
  p.magicP = (unsigned)((1ull << 32) / (unsigned)p.P) + 1u;      // exact for x * P < 2^32: x < S (checked: S * stride < 2^31, P <= S)
  p.r = m.core_layers;
  p.I = m.id_mode == MMT_IDS_2D ? m.patches_per_row * m.patches_per_row : 0;
  p.image_part = m.patches_per_row * m.patches_per_row + 8 + 2 * m.max_dist + 1;
  p.text_part = p.image_part + 1;
  return p;
}

Plan make_plan(const mmt_attn_desc* d, bool dense) {
  Plan pl{};
  pl.dense = dense;
  const int n_tiles = (d->S + 31) / 32;
  pl.split_rows = !dense && d->mask.n_global > 0 && d->mask.local_radius < d->S;
  pl.n_rowblk = pl.split_rows ? (d->mask.n_global + 31) / 32 : 0;
  pl.n_chunks = pl.split_rows ? (n_tiles + kChunkTiles - 1) / kChunkTiles : 0;
  pl.fwd_ws = (size_t)d->B * d->N * pl.n_rowblk * pl.n_chunks * (32 * 64 + 64) * sizeof(float);
  if (pl.split_rows && d->mask.n_global <= 16)     // window kernel: the row groups' parts (<= 2 groups x 4 parts of 8 x 66 floats per plane)
    pl.fwd_ws = std::max(pl.fwd_ws, (size_t)d->B * d->N * 2 * 4 * 8 * 66 * sizeof(float));
  if (pl.split_rows && d->mask.n_global <= 8) {    // plane-walk / sliding-window kernels: partials of the global rows per run
    pl.fwd_ws = std::max(pl.fwd_ws, mmt::fwd_walk_workspace_bytes(d->B, d->N, d->S));
    pl.fwd_ws = std::max(pl.fwd_ws, mmt::fwd_pwin_workspace_bytes(d->B, d->N, d->S, 2 * 256));
  }
  // backward: delta, dRel, global-row / global-key partials, dE partials (floats)
  const size_t bn = (size_t)d->B * d->N, Rp = d->R <= 32 ? 32 : (d->R <= 64 ? 64 : 128);
  pl.n_split = (int)std::min<size_t>(256, ((size_t)d->B * d->S + 255) / 256);
  pl.off_delta = 0;
  pl.off_relfar = pl.off_delta + bn * d->S;
  pl.off_drel = pl.off_relfar + bn * d->S * 2;
  pl.off_pdq = pl.off_drel + bn * (size_t)d->mask.n_global * Rp;
  pl.off_pdtab = pl.off_pdq + bn * pl.n_rowblk * pl.n_chunks * (32 * 64);
  pl.off_pdkv = pl.off_pdtab + bn * pl.n_rowblk * pl.n_chunks * (32 * Rp);
  pl.off_red = pl.off_pdkv + bn * pl.n_rowblk * (pl.n_chunks + 1) * (2 * 32 * 64);      // (+ 1: the hand-over's band slot)
  pl.off_ho = (pl.off_red + bn * ((d->S + 127) / 128) * 4 * (Rp * 64 + Rp) + 3) & ~(size_t)3;
  pl.ho_slots = handover_slots(d, dense);
  const size_t ho_bytes = pl.ho_slots ? bn * n_tiles * ((size_t)pl.ho_slots * 2048 + 1024) : 0;
  pl.bwd_ws = pl.off_ho * sizeof(float) + ho_bytes;
  return pl;
}

// 2-D ids on the lean (bf16, structured pattern) kernels: table width that holds every id that can contribute --
// image ids < (2r+1)^2 + 8, text ids <= 2m, and the two cross-modal part ids P^2 + 8 + 2m + 1 (+ 1) WHEN they are
// below R (small images: P = 4, m = 3 gives 31 / 32 against R = 49); never more than R (ids >= R contribute 0 under
// the one-hot lookup).  0 = not eligible (the general kernels of attn_fwd.hip / attn_bwd.hip take
// the call).
int lean2d_width(const mmt::PatternDev& pat, int R, bool dense) {
  if (dense || pat.id_mode != MMT_IDS_2D || R <= 0) return 0;
  const int d = 2 * pat.r + 1, n2 = d + 2;
  if (n2 * n2 > 256) return 0;                       // look-up table of the clamped (dx, dy) grid
  int need = std::max(d * d + 8, 2 * pat.m + 1);
  if (pat.image_part < R) need = std::max(need, pat.text_part + 1);       // the part ids index real table rows
  need = std::min(R, need);
  return need <= 32 ? 32 : (need <= 64 ? 64 : 0);
}

void fill_common(mmt::FwdParams& p, const mmt_attn_desc* d) {
  std::memset(&p, 0, sizeof(p));
  p.rows_parts = 1;
  p.B = d->B; p.S = d->S; p.N = d->N; p.R = d->R;
  for (int i = 0; i < 3; ++i) {
    p.qs[i] = d->q_stride[i]; p.ks[i] = d->k_stride[i];
    p.vs[i] = d->v_stride[i]; p.os[i] = d->o_stride[i];
  }
  p.sscale = d->scale * mmt::kLog2e;
  p.tscale = (d->flags & MMT_FLAG_SCALE_BEFORE_ADD) ? mmt::kLog2e : d->scale * mmt::kLog2e;
  p.mask_add = d->mask_value * mmt::kLog2e;
  p.pat = make_pattern(d->mask, d->S);
  p.valid_len = d->mask.valid_len;
  if (d->dropout_p > 0.f) {
    unsigned t = (unsigned)((double)d->dropout_p * 65536.0 + 0.5);
    p.drop_thresh = t < 1 ? 1 : (t > 65535 ? 65535 : t);
    p.inv_keep = 65536.f / (65536.f - (float)p.drop_thresh);   // exact keep probability of the 16-bit test
    p.seed_lo = (uint32_t)d->dropout_seed;
    p.seed_hi = (uint32_t)(d->dropout_seed >> 32);
    p.epoch = reinterpret_cast<const unsigned long long*>(d->dropout_epoch);
  }
}

}  // namespace

extern "C" {

int mmt_abi_version(void) { return MMT_ABI_VERSION; }

int mmt_write_step_scalars(uint64_t* dropout_epoch, float* adamw_hyper, uint64_t epoch, float lr,
                           float bias_correction1, float bias_correction2, void* stream) {
  if (!dropout_epoch && !adamw_hyper) return fail(MMT_E_INVALID, "mmt_write_step_scalars: both destinations are NULL");
  const hipError_t e = mmt::launch_write_step_scalars(reinterpret_cast<unsigned long long*>(dropout_epoch), adamw_hyper,
                                                      (unsigned long long)epoch, lr, bias_correction1, bias_correction2,
                                                      reinterpret_cast<hipStream_t>(stream));
  return e == hipSuccess ? MMT_OK : fail(MMT_E_LAUNCH, "mmt_write_step_scalars: %s", hipGetErrorString(e));
}

const char* mmt_last_error(void) { return g_err; }

size_t mmt_workspace_bytes(const mmt_attn_desc* desc) {
  if (check_desc(desc) != MMT_OK) return 0;
  Plan pl = make_plan(desc, false);   // the structured plan is a superset of the dense one
  return pl.fwd_ws > pl.bwd_ws ? pl.fwd_ws : pl.bwd_ws;
}

int mmt_attn_fwd(const mmt_attn_desc* desc, const void* q, const void* k, const void* v,
                 const void* rel_emb, const void* rel_bias, const int32_t* att_mask,
                 const int32_t* rel_ids, void* out, float* lse, void* workspace,
                 size_t workspace_bytes, void* stream) {
  if (int rc = check_desc(desc)) return rc;
  if (!q || !k || !v || !out) return fail(MMT_E_INVALID, "q, k, v, out must not be NULL");
  if (desc->R > 0 && !rel_emb) return fail(MMT_E_INVALID, "R > 0 but rel_emb is NULL");
  const bool dense = att_mask != nullptr || rel_ids != nullptr;
  if (!dense && desc->mask.global_index && desc->mask.n_global > 0)
    return fail(MMT_E_UNSUPPORTED, "a listed global-token set has no structured kernel: materialise att_mask with mmt_side_inputs(materialize_pattern = 1) and pass it (dense operator)");
  const Plan pl = make_plan(desc, dense);
  if (pl.fwd_ws > 0 && (!workspace || workspace_bytes < pl.fwd_ws))
    return fail(MMT_E_WORKSPACE, "workspace too small: need %zu bytes, got %zu", pl.fwd_ws, workspace_bytes);

  mmt::FwdParams p;
  fill_common(p, desc);
  p.q = q; p.k = k; p.v = v; p.emb = rel_emb; p.bias = rel_bias; p.out = out; p.lse = lse;
  p.att_mask = att_mask; p.rel_ids = rel_ids;
  if (desc->R == 0) { p.pat.id_mode = 0; p.rel_ids = nullptr; }
  p.n_band_blocks = desc->B * desc->N * ((desc->S + 127) / 128);
  p.perm_1d = (!dense && p.pat.id_mode == MMT_IDS_1D && desc->R >= 2 * p.pat.m + 1) ? 1 : 0;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool bf16 = desc->dtype == MMT_BF16;

  hipError_t e;
  if (dense) {
    e = mmt::launch_attn_fwd(p, mmt::kDense, bf16, st);
    if (e != hipSuccess) return fail(MMT_E_LAUNCH, "dense forward launch: %s", hipGetErrorString(e));
    return MMT_OK;
  }
  p.skip_global_rows = pl.split_rows ? 1 : 0;
  if (pl.split_rows) {
    p.n_rowblk = pl.n_rowblk; p.n_chunks = pl.n_chunks; p.chunk_tiles = kChunkTiles;
    p.part_o = reinterpret_cast<float*>(workspace);
    p.part_ml = p.part_o + (size_t)desc->B * desc->N * pl.n_rowblk * pl.n_chunks * (32 * 64);
  }
  p.lean_rp = lean2d_width(p.pat, desc->R, dense);
  const bool lean = bf16 && (p.pat.id_mode == 0 || (p.perm_1d && desc->R <= 64) || p.lean_rp);   // attn_fwd_band.hip (tables up to 64 wide)
  p.part_scale = (lean && p.drop_thresh) ? p.inv_keep : 1.f;
  // window kernel (attn_fwd_win.hip): K / V staged once per workgroup, global keys as a peeled quarter-tile step,
  // rows of up to 16 global tokens by flipped-orientation workgroups of the same launch (no workspace, no combine
  // launch).  Shapes it does not cover, or whose LDS need leaves one workgroup per CU, stay with the per-wave staging
  // kernel.  desc->tuning: MMT_TUNE_FWD_NO_WIN turns it off, MMT_TUNE_FWD_FORCE_WIN takes it whenever the shape is
  // covered (the tests run both).
  const int win_mode = (desc->tuning & MMT_TUNE_FWD_NO_WIN) ? 0 : ((desc->tuning & MMT_TUNE_FWD_FORCE_WIN) ? 2 : 1);
  p.tstride = p.pat.id_mode == 0 ? 0 : (2 * p.pat.m + 1 <= 25 ? 26 : 34);
#ifdef MMT_STAMP
  if (const char* v = std::getenv("MMT_DBG_PTR")) p.dbg = reinterpret_cast<long long*>(std::strtoull(v, nullptr, 0));
  if (const char* v = std::getenv("MMT_DBG_MODE")) p.dbg_mode = std::atoi(v);
  if (const char* v = std::getenv("MMT_DBG_SLEEP")) p.dbg_sleep = std::atoi(v);
#endif
  // plane-walk kernel (attn_fwd_walk.hip): persistent workgroups walking runs of row blocks down the band, K / V
  // sliding through a two-slot LDS ring, the rows of <= 8 global tokens merged from the runs' partials by the last
  // arriver of each plane (needs the caller's arrival counters, desc->sync).  OPT-IN (MMT_TUNE_FWD_WALK): measured
  // slower than the window / per-wave kernels at every BASELINE shape -- at per-GPU
  // batch 4 a plane walk has 12 row blocks per run, and filling / draining the diagonal costs 3 of its 9 super-steps.
  const bool walk_shape = lean && !p.lean_rp && desc->R <= 32 && p.pat.radius <= 64 && p.pat.ng <= 8 &&
                          (p.pat.ng == 0 || pl.split_rows) && desc->S > 32;
  const bool walk_sync = p.pat.ng == 0 || (desc->sync && desc->sync_words >= (uint32_t)(desc->B * desc->N));
  if (walk_shape && walk_sync && (desc->tuning & MMT_TUNE_FWD_WALK) &&
      mmt::fwd_walk_lds_bytes(p.pat.ng, p.tstride, p.pat.id_mode != 0) <= 81920) {
    const int grid = mmt::fwd_walk_plan(p, 2 * 256);      // two resident workgroups per compute unit of an MI355X
    p.walk_part = reinterpret_cast<float*>(workspace);
    p.sync = desc->sync;
    e = mmt::launch_attn_fwd_walk_bf16(p, grid, st);
    if (e != hipSuccess) return fail(MMT_E_LAUNCH, "plane-walk forward launch: %s", hipGetErrorString(e));
    return MMT_OK;
  }
  // sliding-window kernel (attn_fwd_pwin.hip): the window kernel made persistent -- <= 512 resident workgroups walking
  // consecutive 128-row blocks, four new K / V tiles per block by LDS-DMA, the next block's Q under the merge, the rows
  // of <= 8 global tokens by the pairs' second waves beside the table build and merged by the plane's last arriver.
  // OPT-IN (MMT_TUNE_FWD_PWIN): correct on every case of the forward tests, measured slower than the window kernel
  // (config 3: 41.9 vs 37.9 us without global tokens, 96 vs 43 us with 8; stamps in profiles/r04_pwin_stamps_*.txt,
  // DESIGN.md section 4, round 4).
  const bool pwin_shape = lean && !p.lean_rp && desc->R <= 32 && p.pat.radius <= 64 && p.pat.ng <= 8 && p.tstride <= 26 &&
                          (p.pat.ng == 0 || (pl.split_rows && p.pat.radius > 32));
  if (pwin_shape && walk_sync && (desc->tuning & MMT_TUNE_FWD_PWIN)) {
    const int grid = mmt::fwd_pwin_plan(p, 2 * 256);
    if (p.pat.ng == 0 || p.walk_maxseg <= 51) {            // (the last arriver's merge keeps a (max, sum) pair per partial in LDS)
      p.walk_part = reinterpret_cast<float*>(workspace);
      p.sync = desc->sync;
      e = mmt::launch_attn_fwd_pwin_bf16(p, grid, st);
      if (e != hipSuccess) return fail(MMT_E_LAUNCH, "sliding-window forward launch: %s", hipGetErrorString(e));
      return MMT_OK;
    }
  }
  const bool win_ok = lean && !p.lean_rp && desc->R <= 32 && p.pat.radius <= 64 && p.pat.ng <= 128;
  // The window kernel's flipped-rows workgroups walk the key tiles of their plane, 8 waves x S / 256 tiles each when one
  // workgroup takes a (plane, 8 rows) group alone: under the band workgroups it shares its CU with it then lives about
  // as long as the launch at S = 4096 and longer at S = 8192 (config 5, g = 8: window 52.7 us, per-wave 47.1 us per
  // call).  With the caller's arrival counters (desc->sync) the group is split over the keys -- S / 2048 workgroups, at
  // most four, merged by the plane's last arriver -- and the window kernel wins at both lengths (config 3: 41.6 against
  // 44.0 us unsplit and 45.8 per-wave; config 5: 41.8 against 45.9).  Without counters it keeps S <= 4096 only.
  // Without global tokens the window kernel has nothing to win -- what it made cheaper is the global tokens -- and the
  // per-wave kernel, whose waves never meet at a barrier, is 5-7 % faster (config 3 shape, dropout 0.1: 34.8-35.4
  // against 36.9-37.9 us, two boxes).
  const bool can_split_rows = desc->sync && desc->sync_words >= (uint32_t)(desc->B * desc->N) && workspace &&
                              !(desc->tuning & MMT_TUNE_FWD_ROWS_ONE_WG);
  const bool win = win_ok && win_mode != 0 &&
                   (win_mode == 2 || (mmt::fwd_win_lds_bytes(p.pat.ng, p.tstride) <= 81920 && p.pat.ng > 0 &&
                                      (desc->S <= 4096 || (can_split_rows && p.pat.ng <= 16))));
  if (win) {
    // rows of the global tokens: at most 16 -> extra workgroups of the window launch (8 rows each, no workspace, no
    // combine launch); more -> the 32-row items of the per-wave kernel + combine, as a launch of their own
    const bool rows_in_win = pl.split_rows && p.pat.ng <= 16;
    const int n_rowblk_items = p.n_rowblk;
    p.n_rowblk = rows_in_win ? (p.pat.ng + 7) / 8 : 0;
    // each (plane, 8 rows) group by up to four workgroups, a quarter of the keys each, merged by the plane's last arriver:
    // needs the caller's arrival counters (desc->sync); without them one workgroup walks all keys, as until round 4
    p.rows_parts = 1;
    if (rows_in_win && can_split_rows) {
      const int n_tiles = (desc->S + 31) / 32;
      p.rows_parts = std::max(1, std::min(4, n_tiles / 64));
      if ((size_t)desc->B * desc->N * p.n_rowblk * p.rows_parts * 8 * 66 * sizeof(float) > workspace_bytes) p.rows_parts = 1;
      p.walk_part = reinterpret_cast<float*>(workspace);
      p.sync = desc->sync;
    }
    e = mmt::launch_attn_fwd_win_bf16(p, st);
    if (e != hipSuccess) return fail(MMT_E_LAUNCH, "window forward launch: %s", hipGetErrorString(e));
    if (!pl.split_rows || rows_in_win) return MMT_OK;
    p.n_rowblk = n_rowblk_items;
    p.rows_only = 1;
  }
  e = lean ? mmt::launch_attn_fwd_band_bf16(p, st) : mmt::launch_attn_fwd(p, mmt::kBand, bf16, st);
  if (e != hipSuccess) return fail(MMT_E_LAUNCH, "band forward launch: %s", hipGetErrorString(e));
  if (pl.split_rows) {
    e = mmt::launch_rows_combine(p, bf16, st);
    if (e != hipSuccess) return fail(MMT_E_LAUNCH, "global-rows combine launch: %s", hipGetErrorString(e));
  }
  return MMT_OK;
}

int mmt_attn_bwd(const mmt_attn_desc* desc, const void* q, const void* k, const void* v,
                 const void* rel_emb, const void* rel_bias, const int32_t* att_mask,
                 const int32_t* rel_ids, const void* out, const void* dout, const float* lse,
                 void* dq, void* dk, void* dv, float* drel_emb, float* drel_bias,
                 void* workspace, size_t workspace_bytes, void* stream) {
  if (int rc = check_desc(desc)) return rc;
  if (!q || !k || !v || !out || !dout || !lse || !dq || !dk || !dv)
    return fail(MMT_E_INVALID, "q, k, v, out, dout, lse, dq, dk, dv must not be NULL");
  if (desc->R > 0 && (!rel_emb || !drel_emb)) return fail(MMT_E_INVALID, "R > 0 but rel_emb / drel_emb is NULL");
  const bool dense = att_mask != nullptr || rel_ids != nullptr;
  if (!dense && desc->mask.global_index && desc->mask.n_global > 0)
    return fail(MMT_E_UNSUPPORTED, "a listed global-token set has no structured kernel: materialise att_mask with mmt_side_inputs(materialize_pattern = 1) and pass it (dense operator)");
  const Plan pl = make_plan(desc, dense);
  if (!workspace || workspace_bytes < pl.bwd_ws)
    return fail(MMT_E_WORKSPACE, "workspace too small: need %zu bytes, got %zu", pl.bwd_ws, workspace_bytes);

  mmt::FwdParams f;
  fill_common(f, desc);
  mmt::BwdParams p;
  std::memset(&p, 0, sizeof(p));
  p.q = q; p.k = k; p.v = v; p.emb = rel_emb; p.bias = rel_bias; p.out = out; p.dout = dout; p.lse = lse;
  p.att_mask = att_mask; p.rel_ids = rel_ids; p.valid_len = f.valid_len;
  p.dq = dq; p.dk = dk; p.dv = dv; p.drel_emb = drel_emb; p.drel_bias = rel_bias ? drel_bias : nullptr;
  p.B = f.B; p.S = f.S; p.N = f.N; p.R = f.R; p.Rp = desc->R <= 32 ? 32 : (desc->R <= 64 ? 64 : 128);
  for (int i = 0; i < 3; ++i) { p.qs[i] = f.qs[i]; p.ks[i] = f.ks[i]; p.vs[i] = f.vs[i]; p.os[i] = f.os[i]; }
  p.sscale = f.sscale; p.tscale = f.tscale; p.mask_add = f.mask_add;
  p.gscale = desc->scale;
  p.rel_gscale = (desc->flags & MMT_FLAG_SCALE_BEFORE_ADD) ? 1.f : desc->scale;
  p.drel_accum = (desc->flags & MMT_FLAG_ACCUM_REL_GRADS) ? 1 : 0;
  p.pat = f.pat;
  if (desc->R == 0) { p.pat.id_mode = 0; p.rel_ids = nullptr; }
  p.perm_1d = (!dense && p.pat.id_mode == MMT_IDS_1D && desc->R >= 2 * p.pat.m + 1) ? 1 : 0;
  p.drop_thresh = f.drop_thresh; p.seed_lo = f.seed_lo; p.seed_hi = f.seed_hi; p.inv_keep = f.inv_keep; p.epoch = f.epoch;
  if (desc->dtype == MMT_BF16) {
    if (const int w2 = lean2d_width(p.pat, desc->R, dense)) {      // lean 2-D path: the kernels run at the narrowed table width
      p.lean2d = 1;
      p.Rp = w2;
    }
  }
  float* ws = reinterpret_cast<float*>(workspace);
  p.delta = ws + pl.off_delta; p.relfar = ws + pl.off_relfar; p.drel = ws + pl.off_drel; p.part_dq = ws + pl.off_pdq;
  p.part_dtab = ws + pl.off_pdtab; p.part_dkv = ws + pl.off_pdkv; p.part_red = ws + pl.off_red;
  p.n_band_blocks = desc->B * desc->N * ((desc->S + 127) / 128);
  p.n_split = pl.n_split;
  if (pl.split_rows) {
    p.skip_global = 1; p.n_gblk = pl.n_rowblk; p.n_chunks = pl.n_chunks; p.chunk_tiles = kChunkTiles;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (p.lean2d && p.R > p.Rp && !p.drel_accum) {
    // ids in [Rp, R) cannot occur (lean2d_width): their gradient rows are zero, and the dE reduce covers ids < Rp only
    (void)hipMemsetAsync(drel_emb + (size_t)p.Rp * p.N * 64, 0, (size_t)(p.R - p.Rp) * p.N * 64 * sizeof(float), st);
    if (p.drel_bias) (void)hipMemsetAsync(p.drel_bias + (size_t)p.Rp * p.N, 0, (size_t)(p.R - p.Rp) * p.N * sizeof(float), st);
  }
  p.dq_plane_major = (desc->tuning & MMT_TUNE_BWD_DQ_PLANE_MAJOR) ? 1 : 0;
  p.ho_per_wave = (desc->tuning & MMT_TUNE_BWD_HO_PER_WAVE) ? 1 : 0;
  // peeled global keys need clipped relative ids only: every peeled key lies beyond the radius, hence beyond max_dist
  p.peel_gkeys = (!dense && pl.split_rows && p.pat.ng <= 8 && (p.pat.id_mode == 0 || (p.perm_1d && p.pat.radius >= p.pat.m))) ? 3 : 0;
  if (!dense && pl.split_rows && p.pat.ng <= 8 && p.lean2d) p.peel_gkeys = 1;      // 2-D ids: the dQ pass's peeled step looks its columns up (the recomputing dK/dV pass keeps its tile visit)
  if (desc->tuning & MMT_TUNE_BWD_NO_PEEL_DQ) p.peel_gkeys &= ~1;       // bit 0: dQ pass, bit 1: dK/dV pass
  if (desc->tuning & MMT_TUNE_BWD_NO_PEEL_DKV) p.peel_gkeys &= ~2;
  p.dkv_slots = p.n_chunks;
  {   // P / dS hand-over: the dK/dV pass reads what the dQ pass computed (needs the peeled kind of global tokens, if any)
    const bool on = !(desc->tuning & MMT_TUNE_BWD_NO_HANDOVER);
    const bool lean = desc->dtype == MMT_BF16 && !dense && (p.pat.id_mode == 0 || (p.perm_1d && p.Rp <= 64) || p.lean2d);
    if (on && lean && pl.ho_slots > 0 && (p.pat.ng == 0 || !pl.split_rows || (p.peel_gkeys & 1))) {
      p.ho = reinterpret_cast<unsigned char*>(ws + pl.off_ho);
      p.ho_slots = pl.ho_slots;
      if (pl.split_rows) p.dkv_slots = p.n_chunks + 1;
    }
  }
#ifdef MMT_STAMP
  if (const char* v = std::getenv("MMT_DBG_PTR")) p.dbg = reinterpret_cast<long long*>(std::strtoull(v, nullptr, 0));
  if (const char* v = std::getenv("MMT_DBG_MODE")) p.dbg_mode = std::atoi(v);
#endif
  hipError_t e = mmt::launch_attn_bwd(p, dense ? mmt::kDense : mmt::kBand, desc->dtype == MMT_BF16, st);
  if (e != hipSuccess) return fail(MMT_E_LAUNCH, "backward launch: %s", hipGetErrorString(e));
  return MMT_OK;
}

int mmt_side_inputs(const mmt_mask_desc* mask, int32_t B, int32_t S,
                    const int32_t* num_image_wordpieces, const int32_t* num_text_wordpieces,
                    int32_t materialize_pattern, int32_t* att_mask_out, int32_t* rel_ids_out,
                    int32_t* segment_ids_out, void* stream) {
  if (!mask) return fail(MMT_E_INVALID, "mask desc is NULL");
  if (B <= 0 || S <= 0) return fail(MMT_E_INVALID, "B and S must be positive");
  if (mask->id_mode < MMT_IDS_NONE || mask->id_mode > MMT_IDS_2D) return fail(MMT_E_INVALID, "bad id_mode");
  if (mask->id_mode == MMT_IDS_2D) {
    // same argument errors as MmtRelativePositionGenerator.__init__ (feature_utils.py:60-65)
    if (mask->patches_per_row <= 0) return fail(MMT_E_INVALID, "`num_patch_per_row` must be positive.");
    if (mask->core_layers <= 0) return fail(MMT_E_INVALID, "`num_core_layers` must be positive.");
    if ((int64_t)mask->patches_per_row * mask->patches_per_row > S) return fail(MMT_E_INVALID, "image part longer than the sequence");
  }
  if (mask->id_mode != MMT_IDS_NONE && mask->max_dist < 0) return fail(MMT_E_INVALID, "`text_relative_pos_max_distance` must be positive.");
  if (rel_ids_out && mask->id_mode == MMT_IDS_NONE) return fail(MMT_E_INVALID, "rel_ids_out requested with id_mode NONE");
  if (materialize_pattern && (mask->local_radius < 0 || mask->n_global < 0 ||
                              (!mask->global_index && (mask->global_start < 0 || mask->global_start + mask->n_global > S))))
    return fail(MMT_E_INVALID, "bad pattern");
  mmt::SideParams p;
  p.pat = make_pattern(*mask, S);
  p.B = B; p.S = S;
  p.img_wp = num_image_wordpieces; p.txt_wp = num_text_wordpieces;
  p.materialize_pattern = materialize_pattern;
  p.gidx = mask->n_global > 0 ? mask->global_index : nullptr;
  p.att_mask = att_mask_out; p.rel_ids = rel_ids_out; p.segment_ids = segment_ids_out;
  hipError_t e = mmt::launch_side_inputs(p, reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(MMT_E_LAUNCH, "side inputs launch: %s", hipGetErrorString(e));
  return MMT_OK;
}

}  // extern "C"
