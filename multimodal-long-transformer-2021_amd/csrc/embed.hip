// Embedding assembly of MmtEncoder (C ABI: include/mmt_layer.h, mmt_embed_*).
//
// Forward: one wave per row gathers the word-table row (fp32, kept in registers between the
// statistics pass and the normalisation pass), applies LayerNorm + dropout, adds the segment row,
// the position row and the projected patch row, and writes the sum once in the compute dtype --
// the reference's gather / LayerNorm / Dropout / Add x3 / Pad / Cast chain (mmt_encoder.py:192-218)
// in a single HBM pass.
//
// Backward: the word-table gradient is a scatter-add of per-row LayerNorm gradients.  Rows are
// visited in the order that sorts their ids (computed by the caller), so equal ids are adjacent:
// the wave at the head of a run of equal ids sums the run in registers and adds it to the table
// row it alone owns -- no atomics, fixed summation order.  Runs are cut at multiples of 32 sorted
// positions so that a heavily repeated id (padding, [MASK]) is spread over many waves; a run that
// crosses a cut leaves per-piece sums in a scratch slab and its head wave adds them up in a
// second launch.  dgamma / dbeta ride along as per-wave column sums (partial slab + fixed-order
// reduce, as in fused_layer.hip); the patch slice of dout is copied out compactly for the
// projection's weight-gradient GEMM.
#include "../../include/mmt_attn.h"
#include "../../include/mmt_layer.h"

#include "layer_common.h"
#include "mmt_err.h"

namespace mmt {

constexpr int kEmbedCut = 32;      // sorted positions per piece of a run
constexpr int kEmbedBlocks = 1024; // blocks of the forward kernel (4 rows in flight each)

struct EmbedParams {
  long rows;
  int S, H, vocab, seg_vocab, patch_start, n_patch;
  float eps, inv_keep;
  uint32_t thresh16, seed_lo, seed_hi;
  const unsigned long long* epoch;     // device-resident addend of the seed (mmt_embed_desc.dropout_epoch) or NULL
  const int *word_ids, *seg_ids, *order, *sorted_ids;   // sorted_ids[i] = word_ids[order[i]]
  const float *word_table, *seg_table, *pos_table, *gamma, *beta, *patch_bias;
  const void* patch;
  void* out;
  float *mean_out, *rstd_out;
  // backward
  const void* dout;
  const float *mean, *rstd;
  float* dword_table;
  float* part;       // [nblocks][2][H] column-sum partials (dgamma, dbeta)
  float* pieces;     // [rows][H] piece sums of runs that cross a cut (only those slots are written)
  void* dpatch;
  int nblocks;
};

template <typename T, int NCH>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const EmbedParams p) {
  const SeedPair sd = effective_seed(p.seed_lo, p.seed_hi, p.epoch);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = p.H >> 3;
  const float invH = 1.f / (float)p.H;
  for (long row = (long)blockIdx.x * 4 + wave; row < p.rows; row += (long)gridDim.x * 4) {
    const int id = p.word_ids[row], sg = p.seg_ids[row];
    const bool id_ok = (unsigned)id < (unsigned)p.vocab, sg_ok = (unsigned)sg < (unsigned)p.seg_vocab;
    const int b = (int)(row / p.S), s = (int)(row - (long)b * p.S);
    const int pj = s - p.patch_start;
    const bool has_patch = p.patch != nullptr && (unsigned)pj < (unsigned)p.n_patch;
    float v[NCH][8];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
#pragma unroll
      for (int i = 0; i < 8; ++i) v[j][i] = 0.f;
      if (c < nch && id_ok) load_param(p.word_table + (long)id * p.H + c * 8, v[j]);
#pragma unroll
      for (int i = 0; i < 8; ++i) sum += v[j][i];
    }
    const float mean = wave_sum(sum) * invH;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j)
      if (lane + 64 * j < nch)
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = v[j][i] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) * invH + p.eps);
    if (lane == 0) { p.mean_out[row] = mean; p.rstd_out[row] = rstd; }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      if (c >= nch) continue;
      const long off = row * p.H + c * 8;
      float g[8], bt[8], y[8], t[8];
      load_param(p.gamma + c * 8, g);
      load_param(p.beta + c * 8, bt);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float w = (v[j][i] - mean) * rstd * g[i] + bt[i];
        if (p.thresh16) w = layer_drop_bits16(layer_drop_row(sd.lo, sd.hi, row), (uint32_t)(c * 8 + i)) >= p.thresh16 ? w * p.inv_keep : 0.f;
        y[i] = w;
      }
      if (sg_ok) {
        load_param(p.seg_table + (long)sg * p.H + c * 8, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] += t[i];
      }
      if (p.pos_table) {
        load_param(p.pos_table + (long)s * p.H + c * 8, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] += t[i];
      }
      if (has_patch) {
        Chunk<T>::load(reinterpret_cast<const T*>(p.patch) + ((long)b * p.n_patch + pj) * p.H + c * 8, t);
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] += t[i];
        if (p.patch_bias) {
          load_param(p.patch_bias + c * 8, t);
#pragma unroll
          for (int i = 0; i < 8; ++i) y[i] += t[i];
        }
      }
      Chunk<T>::store(reinterpret_cast<T*>(p.out) + off, y);
    }
  }
}

// One wave per sorted position i.  Heads of pieces (first of a run, or i % kEmbedCut == 0) sum
// their piece [i, end): end = first position with another id, or the next cut.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const EmbedParams p) {
  const SeedPair sd = effective_seed(p.seed_lo, p.seed_hi, p.epoch);
  __shared__ float red[4][64 * NCH * 8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = p.H >> 3;
  const float invH = 1.f / (float)p.H;
  float acc_g[NCH][8], acc_bt[NCH][8], gam[NCH][8];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc_g[j][i] = 0.f; acc_bt[j][i] = 0.f; gam[j][i] = 0.f; }
    if (lane + 64 * j < nch) load_param(p.gamma + (lane + 64 * j) * 8, gam[j]);
  }
  for (long pos = (long)blockIdx.x * 4 + wave; pos < p.rows; pos += (long)gridDim.x * 4) {
    const int id = p.sorted_ids[pos];
    const bool first = pos == 0 || p.sorted_ids[pos - 1] != id;
    if (!first && (pos % kEmbedCut) != 0) continue;
    const long cut = (pos / kEmbedCut + 1) * kEmbedCut;
    const long lim = cut < p.rows ? cut : p.rows;
    const bool id_ok = (unsigned)id < (unsigned)p.vocab;
    float xw[NCH][8], acc[NCH][8];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { xw[j][i] = 0.f; acc[j][i] = 0.f; }
      if (id_ok && lane + 64 * j < nch) load_param(p.word_table + (long)id * p.H + (lane + 64 * j) * 8, xw[j]);
    }
    // Rows with one id share the LayerNorm input (the id's table row), hence mean, rstd and x_hat: the LayerNorm
    // gradient is linear in dout, so the piece needs only T = sum of its rows' (dropout-masked) dout and ONE
    // evaluation of the LayerNorm backward -- no wave reductions and no dependent loads inside the row loop.
    // Lane l looks at sorted position pos + l: piece length by ballot, row indices by readlane, four rows of dout
    // in flight per step.
    int n_rows, rowv = 0;
    {
      const long me = pos + (lane & 31);
      const bool in_piece = me < lim && p.sorted_ids[me] == id;
      if (me < lim) rowv = p.order[me];
      const unsigned long long mism = ~__ballot(in_piece) & 0xFFFFFFFFull;      // first position that leaves the run
      n_rows = mism ? __builtin_ctzll(mism) : 32;
    }
    const long e = pos + n_rows;
    const int row0 = __builtin_amdgcn_readfirstlane(rowv);
    const float mean = p.mean[row0], rstd = p.rstd[row0];
    float tsum[NCH][8];
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) tsum[j][i] = 0.f;
    for (int k0 = 0; k0 < n_rows; k0 += 4) {
      float t[4][NCH][8];
      int rows4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int k = min(k0 + u, n_rows - 1);
        rows4[u] = __builtin_amdgcn_readlane(rowv, k);
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
          const int c = lane + 64 * j;
#pragma unroll
          for (int i = 0; i < 8; ++i) t[u][j][i] = 0.f;
          if (c < nch && k0 + u < n_rows) Chunk<T>::load(reinterpret_cast<const T*>(p.dout) + (long)rows4[u] * p.H + c * 8, t[u][j]);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (k0 + u >= n_rows) break;
        const int row = rows4[u];
        const int b = row / p.S, s = row - b * p.S, pj = s - p.patch_start;
        const bool to_patch = p.dpatch != nullptr && (unsigned)pj < (unsigned)p.n_patch;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
          const int c = lane + 64 * j;
          if (c >= nch) continue;
          if (to_patch) Chunk<T>::store(reinterpret_cast<T*>(p.dpatch) + ((long)b * p.n_patch + pj) * p.H + c * 8, t[u][j]);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            float v = t[u][j][i];
            if (p.thresh16) v = layer_drop_bits16(layer_drop_row(sd.lo, sd.hi, (long)row), (uint32_t)(c * 8 + i)) >= p.thresh16 ? v * p.inv_keep : 0.f;
            tsum[j][i] += v;
          }
        }
      }
    }
    {
      float xh[NCH][8];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const bool live = lane + 64 * j < nch;
          xh[j][i] = live ? (xw[j][i] - mean) * rstd : 0.f;
          const float dyh = tsum[j][i] * gam[j][i];
          s1 += dyh;
          s2 += dyh * xh[j][i];
          acc_g[j][i] += tsum[j][i] * xh[j][i];
          acc_bt[j][i] += tsum[j][i];
        }
      const float c1 = wave_sum(s1) * invH, c2 = wave_sum(s2) * invH;
#pragma unroll
      for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = rstd * (tsum[j][i] * gam[j][i] - c1 - xh[j][i] * c2);
    }
    // whole run inside this piece -> add to the table row; else park the piece sum in the slab
    const bool run_ends = e >= p.rows || p.sorted_ids[e] != id;
    const bool whole = first && run_ends;
    if (!id_ok) continue;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      if (c >= nch) continue;
      if (whole) {
        float* dst = p.dword_table + (long)id * p.H + c * 8;
        float cur[8];
        load_param(dst, cur);
#pragma unroll
        for (int i = 0; i < 8; ++i) cur[i] += acc[j][i];
        Chunk<float>::store(dst, cur);
      } else {
        Chunk<float>::store(p.pieces + pos * p.H + c * 8, acc[j]);
      }
    }
  }
  // block-level column sums -> partial slab [block][2][H]
  int set = 0;
  auto flush = [&](float (&a)[NCH][8]) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
      for (int i = 0; i < 8; ++i) red[wave][(lane + 64 * j) * 8 + i] = a[j][i];
    __syncthreads();
    for (int col = threadIdx.x; col < p.H; col += 256)
      p.part[((long)blockIdx.x * 2 + set) * p.H + col] = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
    ++set;
  };
  flush(acc_g);
  flush(acc_bt);
}

// Second launch: the head of every run that crosses a cut adds up its pieces (its own, then one per
// cut inside the run, in order) and adds the total to the table row.
template <int NCH>
__global__ __launch_bounds__(256) void embed_bwd_runs_kernel(const EmbedParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = p.H >> 3;
  for (long pos = (long)blockIdx.x * 4 + wave; pos < p.rows; pos += (long)gridDim.x * 4) {
    const int id = p.sorted_ids[pos];
    const bool first = pos == 0 || p.sorted_ids[pos - 1] != id;
    const long cut = (pos / kEmbedCut + 1) * kEmbedCut;
    if (!first || cut >= p.rows || p.sorted_ids[cut] != id || (unsigned)id >= (unsigned)p.vocab) continue;
    float acc[NCH][8];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[j][i] = 0.f;
      if (lane + 64 * j < nch) load_param(p.pieces + pos * p.H + (lane + 64 * j) * 8, acc[j]);
    }
    for (long q = cut; q < p.rows && p.sorted_ids[q] == id; q += kEmbedCut) {
#pragma unroll
      for (int j = 0; j < NCH; ++j)
        if (lane + 64 * j < nch) {
          float t[8];
          load_param(p.pieces + q * p.H + (lane + 64 * j) * 8, t);
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[j][i] += t[i];
        }
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      if (c >= nch) continue;
      float* dst = p.dword_table + (long)id * p.H + c * 8;
      float cur[8];
      load_param(dst, cur);
#pragma unroll
      for (int i = 0; i < 8; ++i) cur[i] += acc[j][i];
      Chunk<float>::store(dst, cur);
    }
  }
}

}  // namespace mmt

namespace {

int check_embed(const mmt_embed_desc* d) {
  if (!d) return mmt::fail(MMT_E_INVALID, "embed desc is NULL");
  if (d->rows <= 0 || d->S <= 0 || (d->rows % d->S)) return mmt::fail(MMT_E_INVALID, "rows must be a positive multiple of S");
  if (d->rows > 0x7fffffffL) return mmt::fail(MMT_E_UNSUPPORTED, "rows exceeds int32");
  if (d->H <= 0 || (d->H & 7)) return mmt::fail(MMT_E_INVALID, "H must be a positive multiple of 8");
  if (d->H > 2048) return mmt::fail(MMT_E_UNSUPPORTED, "H=%d exceeds the built maximum 2048", d->H);
  if (d->dtype != MMT_F32 && d->dtype != MMT_BF16) return mmt::fail(MMT_E_INVALID, "bad dtype %d", d->dtype);
  if (d->vocab <= 0 || d->seg_vocab <= 0) return mmt::fail(MMT_E_INVALID, "vocab sizes must be positive");
  if (d->n_patch < 0 || d->patch_start < 0 || (long)d->patch_start + d->n_patch > d->S)
    return mmt::fail(MMT_E_INVALID, "patch range [%d, %d) does not fit the sequence", d->patch_start, d->patch_start + d->n_patch);
  if (!(d->dropout_p >= 0.f && d->dropout_p < 1.f)) return mmt::fail(MMT_E_INVALID, "dropout_p must be in [0,1)");
  return MMT_OK;
}

void fill_embed(mmt::EmbedParams& p, const mmt_embed_desc* d) {
  p = mmt::EmbedParams{};
  p.rows = d->rows; p.S = d->S; p.H = d->H; p.vocab = d->vocab; p.seg_vocab = d->seg_vocab;
  p.patch_start = d->patch_start; p.n_patch = d->n_patch; p.eps = d->eps;
  p.thresh16 = mmt::dropout_thresh16(d->dropout_p);
  if (p.thresh16) {
    p.inv_keep = mmt::dropout_inv_keep(p.thresh16);
    p.seed_lo = (uint32_t)d->dropout_seed; p.seed_hi = (uint32_t)(d->dropout_seed >> 32);
    p.epoch = reinterpret_cast<const unsigned long long*>(d->dropout_epoch);
  }
}

int embed_blocks(const mmt_embed_desc* d) {
  const long need = (d->rows + 3) / 4;
  return (int)(need < mmt::kEmbedBlocks ? need : mmt::kEmbedBlocks);
}

}  // namespace

extern "C" {

int mmt_embed_fwd(const mmt_embed_desc* d, const int32_t* word_ids, const int32_t* seg_ids,
                  const float* word_table, const float* seg_table, const float* pos_table,
                  const float* gamma, const float* beta, const void* patch_proj, const float* patch_bias,
                  void* out, float* mean, float* rstd, void* stream) {
  if (int rc = check_embed(d)) return rc;
  if (!word_ids || !seg_ids || !word_table || !seg_table || !gamma || !beta || !out || !mean || !rstd)
    return mmt::fail(MMT_E_INVALID, "mmt_embed_fwd: NULL argument");
  if (d->n_patch > 0 && !patch_proj) return mmt::fail(MMT_E_INVALID, "mmt_embed_fwd: n_patch > 0 but patch_proj is NULL");
  mmt::EmbedParams p; fill_embed(p, d);
  p.word_ids = word_ids; p.seg_ids = seg_ids; p.word_table = word_table; p.seg_table = seg_table;
  p.pos_table = pos_table; p.gamma = gamma; p.beta = beta; p.patch = d->n_patch > 0 ? patch_proj : nullptr;
  p.patch_bias = patch_bias; p.out = out; p.mean_out = mean; p.rstd_out = rstd;
  const int blocks = embed_blocks(d), nchl = ((d->H >> 3) + 63) / 64;
  hipStream_t st = (hipStream_t)stream;
#define MMT_EF(T, N) hipLaunchKernelGGL((mmt::embed_fwd_kernel<T, N>), dim3(blocks), dim3(256), 0, st, p)
  if (d->dtype == MMT_BF16) { if (nchl <= 1) MMT_EF(__bf16, 1); else if (nchl <= 2) MMT_EF(__bf16, 2); else MMT_EF(__bf16, 4); }
  else { if (nchl <= 1) MMT_EF(float, 1); else if (nchl <= 2) MMT_EF(float, 2); else MMT_EF(float, 4); }
#undef MMT_EF
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MMT_OK : mmt::fail(MMT_E_LAUNCH, "mmt_embed_fwd: %s", hipGetErrorString(e));
}

size_t mmt_embed_workspace_bytes(const mmt_embed_desc* d) {
  if (!d || d->H <= 0 || d->rows <= 0) return 0;
  return ((size_t)mmt::kEmbedBlocks * 2 * d->H + (size_t)d->rows * d->H) * sizeof(float);
}

int mmt_embed_bwd(const mmt_embed_desc* d, const void* dout, const int32_t* sorted_ids, const int32_t* order,
                  const float* word_table, const float* gamma, const float* mean, const float* rstd,
                  float* dword_table, float* dgamma, float* dbeta, void* dpatch, void* ws, size_t ws_bytes,
                  void* stream) {
  if (int rc = check_embed(d)) return rc;
  if (!dout || !sorted_ids || !order || !word_table || !gamma || !mean || !rstd || !dword_table || !dgamma || !dbeta)
    return mmt::fail(MMT_E_INVALID, "mmt_embed_bwd: NULL argument");
  if (!ws || ws_bytes < mmt_embed_workspace_bytes(d)) return mmt::fail(MMT_E_WORKSPACE, "mmt_embed_bwd: workspace too small");
  mmt::EmbedParams p; fill_embed(p, d);
  p.dout = dout; p.sorted_ids = sorted_ids; p.order = order; p.word_table = word_table; p.gamma = gamma;
  p.mean = mean; p.rstd = rstd; p.dword_table = dword_table; p.dpatch = d->n_patch > 0 ? dpatch : nullptr;
  p.part = (float*)ws; p.pieces = p.part + (size_t)mmt::kEmbedBlocks * 2 * d->H;
  p.nblocks = embed_blocks(d);
  const int nchl = ((d->H >> 3) + 63) / 64;
  hipStream_t st = (hipStream_t)stream;
#define MMT_EB(T, N) hipLaunchKernelGGL((mmt::embed_bwd_kernel<T, N>), dim3(p.nblocks), dim3(256), 0, st, p)
  if (d->dtype == MMT_BF16) { if (nchl <= 1) MMT_EB(__bf16, 1); else if (nchl <= 2) MMT_EB(__bf16, 2); else MMT_EB(__bf16, 4); }
  else { if (nchl <= 1) MMT_EB(float, 1); else if (nchl <= 2) MMT_EB(float, 2); else MMT_EB(float, 4); }
#undef MMT_EB
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mmt::fail(MMT_E_LAUNCH, "mmt_embed_bwd: %s", hipGetErrorString(e));
  if (nchl <= 1) hipLaunchKernelGGL(mmt::embed_bwd_runs_kernel<1>, dim3(p.nblocks), dim3(256), 0, st, p);
  else if (nchl <= 2) hipLaunchKernelGGL(mmt::embed_bwd_runs_kernel<2>, dim3(p.nblocks), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(mmt::embed_bwd_runs_kernel<4>, dim3(p.nblocks), dim3(256), 0, st, p);
  if ((e = hipGetLastError()) != hipSuccess) return mmt::fail(MMT_E_LAUNCH, "mmt_embed_bwd runs: %s", hipGetErrorString(e));
  e = mmt::launch_colsum_reduce(p.part, p.nblocks, 2, d->H, dgamma, dbeta, nullptr, d->accumulate, st);
  return e == hipSuccess ? MMT_OK : mmt::fail(MMT_E_LAUNCH, "mmt_embed_bwd reduce: %s", hipGetErrorString(e));
}

}  // extern "C"
