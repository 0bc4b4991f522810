// Fused residual-block kernels for gfx950 (C ABI: include/mmt_layer.h).
//
// All of them are HBM-bound streaming kernels: 16-byte loads/stores per lane, one wave per
// row for the LayerNorm-bearing ones (row kept in registers between the statistics pass and
// the normalisation pass, wave reductions by DPP/shuffle, no LDS in the row loop), per-wave
// register accumulators for the column sums (dbias / dgamma / dbeta) that are flushed once per
// block into a partial slab and summed in fixed order by a second tiny kernel (bitwise
// reproducible, no atomics).
#include "../../include/mmt_attn.h"
#include "../../include/mmt_layer.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>

#include "layer_common.h"
#include <cstring>
#include "mmt_err.h"

namespace mmt {

struct LayerParams {
  long rows;
  int H;
  float eps, inv_keep;
  uint32_t thresh16, seed_lo, seed_hi;
  const unsigned long long* epoch;     // device-resident addend of the seed (mmt_rows_desc.dropout_epoch) or NULL
  const void *a, *b, *c;          // inputs (meaning per kernel)
  const float *p0, *p1, *p2;      // fp32 parameters / statistics
  const float *mean, *rstd;
  void *o0, *o1;                  // outputs
  float *s0, *s1;                 // fp32 outputs (mean/rstd)
  float* part;                    // partial column sums [nblocks][K][H]
  int nblocks;
};

// =============================================================================================
// x_new = x + Dropout(o + bias);  h = LN(x_new)      (RESID)     |    h = LN(x)    (!RESID)
//   a = o, b = x, p0 = bias, p1 = gamma, p2 = beta, o0 = x_new, o1 = h, s0 = mean, s1 = rstd
// =============================================================================================
template <typename T, int W, int NCH, bool RESID, bool HAS_LN>
__global__ __launch_bounds__(256) void resid_ln_fwd_kernel(const LayerParams p) {
  // Persistent: a wave walks rows blockIdx * 4 + wave, + 4 gridDim, ... with the NEXT row's inputs requested before the
  // current row's reductions (raw registers, converted when their turn comes) and bias / gamma / beta loaded once per wave
  // (round 4; one row per wave and nine parameter loads per row until then: 23-26 us for the 101 MB of a residual block).
  const SeedPair sd = effective_seed(p.seed_lo, p.seed_hi, p.epoch);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = p.H / W;
  const float invH = 1.f / (float)p.H;
  float bs[NCH][W], g[NCH][W], bt[NCH][W];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    const int c = lane + 64 * j;
#pragma unroll
    for (int i = 0; i < W; ++i) { bs[j][i] = 0.f; g[j][i] = 0.f; bt[j][i] = 0.f; }
    if (c < nch) {
      if (RESID) load_param_w<W>(p.p0 + c * W, bs[j]);
      if (HAS_LN) { load_param_w<W>(p.p1 + c * W, g[j]); load_param_w<W>(p.p2 + c * W, bt[j]); }
    }
  }
  const long stride = (long)gridDim.x * 4;
  long row = (long)blockIdx.x * 4 + wave;
  RawW<T, W> ra[NCH], rb[NCH];             // o, x as loaded
  auto request = [&](long r) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
      ra[j].zero(); rb[j].zero();
      if (c < nch) {
        const long off = r * p.H + c * W;
        if (RESID) ra[j].load(reinterpret_cast<const T*>(p.a) + off);
        rb[j].load(reinterpret_cast<const T*>(p.b) + off);
      }
    }
  };
  if (row < p.rows) request(row);
  for (; row < p.rows; row += stride) {
    float v[NCH][W];
    const uint32_t drow = layer_drop_row(sd.lo, sd.hi, row);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int c = lane + 64 * j;
#pragma unroll
      for (int i = 0; i < W; ++i) {
        if (RESID) {
          float t = ra[j].get(i) + bs[j][i];
          if (p.thresh16) t = layer_drop_bits16(drow, (uint32_t)(c * W + i)) >= p.thresh16 ? t * p.inv_keep : 0.f;
          v[j][i] = rb[j].get(i) + t;
          if (sizeof(T) == 2) v[j][i] = (float)(__bf16)v[j][i];      // LayerNorm sees the stored (rounded) x_new
        } else {
          v[j][i] = rb[j].get(i);
        }
      }
      if (c >= nch) {
#pragma unroll
        for (int i = 0; i < W; ++i) v[j][i] = 0.f;
      }
    }
    if (row + stride < p.rows) request(row + stride);          // in flight under this row's reductions and stores
    if (RESID) {
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int c = lane + 64 * j;
        if (c < nch) ChunkW<T, W>::store(reinterpret_cast<T*>(p.o0) + row * p.H + c * W, v[j]);
      }
    }
    if (HAS_LN) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int i = 0; i < W; ++i) s += v[j][i];
      const float mean = wave_sum(s) * invH;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < NCH; ++j)
        if (lane + 64 * j < nch)
#pragma unroll
          for (int i = 0; i < W; ++i) { const float d = v[j][i] - mean; q += d * d; }
      const float rstd = rsqrtf(wave_sum(q) * invH + p.eps);
      if (lane == 0) { p.s0[row] = mean; p.s1[row] = rstd; }
#pragma unroll
      for (int j = 0; j < NCH; ++j) {
        const int c = lane + 64 * j;
        if (c < nch) {
          float y[W];
#pragma unroll
          for (int i = 0; i < W; ++i) y[i] = (v[j][i] - mean) * rstd * g[j][i] + bt[j][i];
          ChunkW<T, W>::store(reinterpret_cast<T*>(p.o1) + row * p.H + c * W, y);
        }
      }
    }
  }
}

// =============================================================================================
// dX = dx_in + LNbwd(dh);  dx = dX;  do = DropBwd(dX);  partial column sums {dbias, dgamma, dbeta}
//   a = dx_in (nullable), b = dh, c = x_new (LN input), p1 = gamma, o0 = do, o1 = dx
// =============================================================================================
template <typename T, int W, int NCH, bool RESID, bool HAS_LN>
__global__ __launch_bounds__(256, (W * NCH <= 12 ? 4 : 1)) void resid_ln_bwd_kernel(const LayerParams p) {
  const SeedPair sd = effective_seed(p.seed_lo, p.seed_hi, p.epoch);
  __shared__ float red[4][64 * NCH * W];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = p.H / W;
  const float invH = 1.f / (float)p.H;
  const bool has_in = p.a != nullptr;
  float acc_b[NCH][W], acc_g[NCH][W], acc_bt[NCH][W];
  float gam[NCH][W];
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
#pragma unroll
    for (int i = 0; i < W; ++i) { acc_b[j][i] = 0.f; acc_g[j][i] = 0.f; acc_bt[j][i] = 0.f; gam[j][i] = 0.f; }
    if (HAS_LN && lane + 64 * j < nch) load_param_w<W>(p.p1 + (lane + 64 * j) * W, gam[j]);
  }
  // The row's three inputs stay in registers AS LOADED (bf16: W/2 registers per chunk) across the
  // two wave reductions and are converted twice.
  for (long row = (long)blockIdx.x * 4 + wave; row < p.rows; row += (long)gridDim.x * 4) {
    RawW<T, W> ra[NCH], rb[NCH], rc[NCH];      // dx_in, dh, x_new
    const uint32_t drow = layer_drop_row(sd.lo, sd.hi, row);
    float mean = 0.f, rstd = 0.f;
    if (HAS_LN) { mean = p.mean[row]; rstd = p.rstd[row]; }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int ch = lane + 64 * j;
      ra[j].zero(); rb[j].zero(); rc[j].zero();
      if (ch < nch) {
        const long off = row * p.H + ch * W;
        if (has_in) ra[j].load(reinterpret_cast<const T*>(p.a) + off);
        if (HAS_LN) { rb[j].load(reinterpret_cast<const T*>(p.b) + off); rc[j].load(reinterpret_cast<const T*>(p.c) + off); }
      }
    }
    float c1 = 0.f, c2 = 0.f;
    if (HAS_LN) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int i = 0; i < W; ++i) {
          const float dh = rb[j].get(i);
          const float xh = (rc[j].get(i) - mean) * rstd;
          const float dyh = dh * gam[j][i];
          s1 += dyh;
          s2 += dyh * xh;
          acc_g[j][i] += dh * xh;       // lanes past the row hold zeros (dh = 0)
          acc_bt[j][i] += dh;
        }
      c1 = wave_sum(s1) * invH; c2 = wave_sum(s2) * invH;
#pragma unroll
      for (int j = 0; j < NCH; ++j) { rb[j].opaque(); rc[j].opaque(); }
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      const int ch = lane + 64 * j;
      if (ch < nch) {
        const long off = row * p.H + ch * W;
        float dx[W];
#pragma unroll
        for (int i = 0; i < W; ++i) {
          float g = ra[j].get(i);
          if (HAS_LN) {
            const float xh = (rc[j].get(i) - mean) * rstd;
            g += rstd * (rb[j].get(i) * gam[j][i] - c1 - xh * c2);
          }
          dx[i] = g;
        }
        if (RESID) {
          float d_o[W];
#pragma unroll
          for (int i = 0; i < W; ++i) {
            float t = dx[i];
            if (p.thresh16) t = layer_drop_bits16(drow, (uint32_t)(ch * W + i)) >= p.thresh16 ? t * p.inv_keep : 0.f;
            d_o[i] = t;
            acc_b[j][i] += t;
          }
          ChunkW<T, W>::store(reinterpret_cast<T*>(p.o0) + off, d_o);
        }
        ChunkW<T, W>::store(reinterpret_cast<T*>(p.o1) + off, dx);
      }
    }
  }
  // block-level sums of the three accumulators -> partial slab [block][k][H]
  constexpr int kSets = (RESID ? 1 : 0) + (HAS_LN ? 2 : 0);
  int set = 0;
  auto flush = [&](float (&acc)[NCH][W]) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
      for (int i = 0; i < W; ++i) red[wave][(lane + 64 * j) * W + i] = acc[j][i];
    __syncthreads();
    for (int col = threadIdx.x; col < p.H; col += 256)
      p.part[((long)blockIdx.x * kSets + set) * p.H + col] = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
    ++set;
  };
  if (RESID) flush(acc_b);
  if (HAS_LN) { flush(acc_g); flush(acc_bt); }
}

// out[j][col] = sum_blocks part[block][j][col]   (fixed order).  Workgroup = 16 columns x 64 block groups: every
// thread sums nblocks / 64 partials (independent loads), LDS combines the groups in group order.  (64 columns x 16
// groups had 36 workgroups for the 3 x 768 sums of a residual block -- 36 of 256 CUs reading a 9.4 MB slab:
// 12 us; 144 narrower workgroups take a third of that.)
constexpr int kCsCols = 16, kCsGroups = 64;
__device__ __forceinline__ void colsum_reduce_body(int blk, const float* part, int nblocks, int ksets, int H,
                                                   float* o0, float* o1, float* o2, int accumulate, float (*red)[kCsCols]) {
  const int c = threadIdx.x & (kCsCols - 1), grp = threadIdx.x / kCsCols;
  const int idx = blk * kCsCols + c;
  const int total = ksets * H;
  const int per = (nblocks + kCsGroups - 1) / kCsGroups;
  const int lo = grp * per, hi = min(nblocks, lo + per);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (idx < total) {
    int b = lo;
    for (; b + 3 < hi; b += 4) {
      a0 += part[(long)b * total + idx];
      a1 += part[(long)(b + 1) * total + idx];
      a2 += part[(long)(b + 2) * total + idx];
      a3 += part[(long)(b + 3) * total + idx];
    }
    for (; b < hi; ++b) a0 += part[(long)b * total + idx];
  }
  red[grp][c] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (grp == 0 && idx < total) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < kCsGroups; ++g) s += red[g][c];
    const int set = idx / H, col = idx - set * H;
    float* dst = set == 0 ? o0 : (set == 1 ? o1 : o2);
    dst[col] = accumulate ? dst[col] + s : s;
  }
}
__global__ __launch_bounds__(1024) void colsum_reduce_kernel(const float* part, int nblocks, int ksets, int H,
                                                             float* o0, float* o1, float* o2, int accumulate) {
  __shared__ float red[kCsGroups][kCsCols];
  colsum_reduce_body(blockIdx.x, part, nblocks, ksets, H, o0, o1, o2, accumulate, red);
}
// The reduces of many *_bwd calls (defer_reduce) in ONE launch: workgroups [blocks_end[j-1], blocks_end[j]) belong to
// item j.  A replayed step holds ~32 of these 5-us launches (0.18 ms); a host with nothing waiting for the parameter
// gradients before backward ends queues them and launches once.
constexpr int kCsBatchMax = 48;
struct ColsumBatch {
  const float* part[kCsBatchMax];
  float* o0[kCsBatchMax]; float* o1[kCsBatchMax]; float* o2[kCsBatchMax];
  int nblocks[kCsBatchMax], H[kCsBatchMax], blocks_end[kCsBatchMax];
  unsigned char ksets[kCsBatchMax], accumulate[kCsBatchMax];
  int n;
};
__global__ __launch_bounds__(1024) void colsum_reduce_batch_kernel(const ColsumBatch g) {
  __shared__ float red[kCsGroups][kCsCols];
  int j = 0;
  while (j + 1 < g.n && (int)blockIdx.x >= g.blocks_end[j]) ++j;
  const int blk = (int)blockIdx.x - (j ? g.blocks_end[j - 1] : 0);
  colsum_reduce_body(blk, g.part[j], g.nblocks[j], g.ksets[j], g.H[j], g.o0[j], g.o1[j], g.o2[j], g.accumulate[j], red);
}

// Column sums of a [rows, C] matrix with any row stride (the bias gradient of a Dense layer whose output has no
// 16-byte row alignment, e.g. the 30522-way MLM logits): thread = two adjacent columns (4-byte bf16 / 8-byte fp32
// loads), blockIdx.y = row group; eight rows in flight per thread; partials [row groups][C] -> colsum_reduce_kernel.
template <typename T>
__global__ __launch_bounds__(256) void colsum_pairs_kernel(const T* x, long ld, long rows, int C, float* part) {
  const int c = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (c >= C) return;
  const bool two = c + 1 < C;
  float a0 = 0.f, a1 = 0.f;
  const long r0 = blockIdx.y, step = gridDim.y;
  auto load2 = [&](long row, float& u, float& v) {
    const T* q = x + row * ld + c;
    if (sizeof(T) == 2 && two && ((reinterpret_cast<uintptr_t>(q) & 3) == 0)) {
      const uint32_t w = *reinterpret_cast<const uint32_t*>(q);
      u = __uint_as_float(w << 16); v = __uint_as_float(w & 0xFFFF0000u);
    } else {
      u = (float)q[0]; v = two ? (float)q[1] : 0.f;
    }
  };
  long row = r0;
  for (; row + 7 * step < rows; row += 8 * step) {
    float u[8], v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) load2(row + j * step, u[j], v[j]);
#pragma unroll
    for (int j = 0; j < 8; ++j) { a0 += u[j]; a1 += v[j]; }
  }
  for (; row < rows; row += step) { float u, v; load2(row, u, v); a0 += u; a1 += v; }
  part[(long)blockIdx.y * C + c] = a0;
  if (two) part[(long)blockIdx.y * C + c + 1] = a1;
}

hipError_t launch_colsum_reduce_batch(const ColsumBatch& g, hipStream_t st) {
  hipLaunchKernelGGL(colsum_reduce_batch_kernel, dim3(g.blocks_end[g.n - 1]), dim3(1024), 0, st, g);
  return hipGetLastError();
}

hipError_t launch_colsum_reduce(const float* part, int nblocks, int ksets, int H, float* o0, float* o1,
                                float* o2, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(colsum_reduce_kernel, dim3((ksets * H + kCsCols - 1) / kCsCols), dim3(1024), 0, st, part, nblocks, ksets, H, o0, o1, o2, accumulate);
  return hipGetLastError();
}

// =============================================================================================
// GELU (tanh approximation, mmt_encoder.py:53-54) with the dense layer's bias folded in.
// =============================================================================================
// gelu_tanh(): layer_common.h

// thread = one 8-column chunk (blockIdx.x * 256 + tid), rows strided by gridDim.y
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void bias_gelu_kernel(const LayerParams p) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int nch = p.H >> 3;
  if (c >= nch) return;
  float bs[8], acc[8];
  load_param(p.p0 + c * 8, bs);
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 0.f;
  for (long row = blockIdx.y; row < p.rows; row += gridDim.y) {
    const long off = row * p.H + c * 8;
    float u[8], y[8];
    Chunk<T>::load(reinterpret_cast<const T*>(p.a) + off, u);
    if (BWD) {
      float dy[8];
      Chunk<T>::load(reinterpret_cast<const T*>(p.b) + off, dy);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float dz;
        (void)gelu_tanh(u[i] + bs[i], dz);
        y[i] = dy[i] * dz;
        acc[i] += y[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) { float dz; y[i] = gelu_tanh(u[i] + bs[i], dz); }
    }
    Chunk<T>::store(reinterpret_cast<T*>(p.o0) + off, y);
  }
  if (BWD) {
    float* dst = p.part + (long)blockIdx.y * p.H + c * 8;
    *reinterpret_cast<f32x4*>(dst) = f32x4{acc[0], acc[1], acc[2], acc[3]};
    *reinterpret_cast<f32x4*>(dst + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
  }
}

// Forward only: the [rows, H] matrix as one flat list of 8-column chunks, grid-stride, four chunks in flight per
// thread.  The grid is sized so that the stride is a multiple of the chunks per row: a thread keeps its column
// (bias in registers) and no lane idles (the (column block, row split) grid left a quarter of them idle at
// H = 3072 and kept one load in flight per thread: 43 us for 200 MB).
template <typename T>
__global__ __launch_bounds__(256) void bias_gelu_flat_kernel(const LayerParams p) {
  const int nch = p.H >> 3;
  const long total = p.rows * nch, stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  float bs[8];
  load_param(p.p0 + (int)(i % nch) * 8, bs);
  const T* src = reinterpret_cast<const T*>(p.a);
  T* dst = reinterpret_cast<T*>(p.o0);
  for (; i + 3 * stride < total; i += 4 * stride) {
    float u[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k) Chunk<T>::load(src + (i + k * stride) * 8, u[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int j = 0; j < 8; j += 2) {
        f32x2 dz;
        const f32x2 g = gelu_tanh_x2(f32x2{u[k][j] + bs[j], u[k][j + 1] + bs[j + 1]}, dz);
        u[k][j] = g[0]; u[k][j + 1] = g[1];
      }
      Chunk<T>::store(dst + (i + k * stride) * 8, u[k]);
    }
  }
  for (; i < total; i += stride) {
    float u[8];
    Chunk<T>::load(src + i * 8, u);
#pragma unroll
    for (int j = 0; j < 8; ++j) { float dz; u[j] = gelu_tanh(u[j] + bs[j], dz); }
    Chunk<T>::store(dst + i * 8, u);
  }
}

// acc[i] += g[i]: 8 elements per thread, grid-stride.
template <typename T>
__global__ __launch_bounds__(256) void accumulate_grad_kernel(float* acc, const T* g, long n) {
  const long nch = n >> 3;
  for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < nch; c += (long)gridDim.x * 256) {
    float a[8], b[8];
    Chunk<float>::load(acc + c * 8, a);
    Chunk<T>::load(g + c * 8, b);
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] += b[i];
    Chunk<float>::store(acc + c * 8, a);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
    const long i = (nch << 3) + threadIdx.x;
    acc[i] += (float)g[i];
  }
}

// Global gradient norm + clip factor (`optimizer_config.gradient_clip_norm`, applied to the all-reduced gradients
// before AdamW): one streaming pass over the flat fp32 gradient slabs (16 bytes per lane, 4 loads in flight), a
// partial sum of squares per block, and a one-block kernel that adds the partials in a fixed order and writes
//   scale = min(1, max_norm / (pending * sqrt(sum) + 1e-6)) * pending
// (pending = a mean over replicas that has not been applied to the slabs yet) as the device scalar AdamW reads.
constexpr int kMaxNormSlabs = 16, kNormBlocks = 2048;
struct NormSlabs {
  const float* ptr[kMaxNormSlabs];
  long n4[kMaxNormSlabs];       // float4 elements
  int n;
};
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const NormSlabs a, float* partials) {
  __shared__ float red[4];
  float acc = 0.f;
  for (int sidx = 0; sidx < a.n; ++sidx) {
    const f32x4* x = reinterpret_cast<const f32x4*>(a.ptr[sidx]);
    const long n4 = a.n4[sidx], stride = (long)gridDim.x * 256;
    long c = (long)blockIdx.x * 256 + threadIdx.x;
    for (; c + 3 * stride < n4; c += 4 * stride) {
      const f32x4 v0 = x[c], v1 = x[c + stride], v2 = x[c + 2 * stride], v3 = x[c + 3 * stride];
      acc += (v0[0] * v0[0] + v0[1] * v0[1]) + (v0[2] * v0[2] + v0[3] * v0[3]);
      acc += (v1[0] * v1[0] + v1[1] * v1[1]) + (v1[2] * v1[2] + v1[3] * v1[3]);
      acc += (v2[0] * v2[0] + v2[1] * v2[1]) + (v2[2] * v2[2] + v2[3] * v2[3]);
      acc += (v3[0] * v3[0] + v3[1] * v3[1]) + (v3[2] * v3[2] + v3[3] * v3[3]);
    }
    for (; c < n4; c += stride) {
      const f32x4 v = x[c];
      acc += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void clip_scale_kernel(const float* partials, float max_norm, float pending,
                                                         float* scale_out, float* norm_out) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < kNormBlocks; i += 256) acc += (double)partials[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt((red[0] + red[1]) + (red[2] + red[3])) * pending;
    if (norm_out) *norm_out = norm;
    *scale_out = fminf(max_norm / (norm + 1e-6f), 1.f) * pending;
  }
}

// AdamW over a flat slab; one block = one 1024-element chunk (4 elements per thread).
struct AdamwParams {
  float lr, beta1, beta2, eps, inv_bc1, inv_sqrt_bc2;
  int zero_grad;
};
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ prm, float* __restrict__ grd,
                                                    float* __restrict__ m1, float* __restrict__ m2,
                                                    __bf16* __restrict__ shadow, const float* __restrict__ chunk_wd,
                                                    const float* __restrict__ grad_scale, long n_chunks, AdamwParams a,
                                                    const float* __restrict__ hyper) {
  const float gs = grad_scale ? *grad_scale : 1.f;
  if (hyper) {          // device-resident {lr, bias_correction1, bias_correction2} (mmt_adamw_desc.hyper)
    a.lr = hyper[0];
    a.inv_bc1 = 1.f / hyper[1];
    a.inv_sqrt_bc2 = 1.f / sqrtf(hyper[2]);
  }
  // two chunks per iteration: eight 16-byte loads in flight per thread before the first store
  for (long c0 = blockIdx.x; c0 < n_chunks; c0 += 2L * gridDim.x) {
    const long c1 = c0 + gridDim.x;
    const bool two = c1 < n_chunks;
    const long i0 = c0 * 1024 + threadIdx.x * 4, i1 = (two ? c1 : c0) * 1024 + threadIdx.x * 4;
    const float decay0 = 1.f - a.lr * chunk_wd[c0], decay1 = 1.f - a.lr * chunk_wd[two ? c1 : c0];
    f32x4 p4[2], g4[2], a4[2], b4[2];
    p4[0] = *reinterpret_cast<const f32x4*>(prm + i0); g4[0] = *reinterpret_cast<const f32x4*>(grd + i0);
    a4[0] = *reinterpret_cast<const f32x4*>(m1 + i0); b4[0] = *reinterpret_cast<const f32x4*>(m2 + i0);
    p4[1] = *reinterpret_cast<const f32x4*>(prm + i1); g4[1] = *reinterpret_cast<const f32x4*>(grd + i1);
    a4[1] = *reinterpret_cast<const f32x4*>(m1 + i1); b4[1] = *reinterpret_cast<const f32x4*>(m2 + i1);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (u == 1 && !two) break;
      const long i = u ? i1 : i0;
      const float decay = u ? decay1 : decay0;
      f32x4 o4;
      bf16x4 s4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float g = g4[u][j] * gs;
        const float mm = a.beta1 * a4[u][j] + (1.f - a.beta1) * g;
        const float vv = a.beta2 * b4[u][j] + (1.f - a.beta2) * g * g;
        const float denom = sqrtf(vv) * a.inv_sqrt_bc2 + a.eps;
        const float pn = p4[u][j] * decay - a.lr * a.inv_bc1 * (mm / denom);
        a4[u][j] = mm; b4[u][j] = vv; o4[j] = pn; s4[j] = (__bf16)pn;
      }
      *reinterpret_cast<f32x4*>(prm + i) = o4;
      *reinterpret_cast<f32x4*>(m1 + i) = a4[u];
      *reinterpret_cast<f32x4*>(m2 + i) = b4[u];
      if (shadow) *reinterpret_cast<bf16x4*>(shadow + i) = s4;
      if (a.zero_grad) *reinterpret_cast<f32x4*>(grd + i) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
}

}  // namespace mmt

// =============================================================================================
// C ABI
// =============================================================================================
namespace {

#define lfail mmt::fail

constexpr int kRowBlocks = 1024;   // blocks of the row-wise kernels (4 rows in flight each)
constexpr int kGeluRowSplit = 512; // gridDim.y of the GELU kernels

int check_rows(const mmt_rows_desc* d, int max_h) {
  if (!d) return lfail(MMT_E_INVALID, "desc is NULL");
  if (d->rows <= 0 || d->H <= 0 || (d->H & 7)) return lfail(MMT_E_INVALID, "rows must be positive and H a positive multiple of 8");
  if (d->H > max_h) return lfail(MMT_E_UNSUPPORTED, "H=%d exceeds the built maximum %d", d->H, max_h);
  if (d->dtype != MMT_F32 && d->dtype != MMT_BF16) return lfail(MMT_E_INVALID, "bad dtype %d", d->dtype);
  if (!(d->dropout_p >= 0.f && d->dropout_p < 1.f)) return lfail(MMT_E_INVALID, "dropout_p must be in [0,1)");
  return MMT_OK;
}

// forward kernels have no per-block partial slab: one row per wave up to 4096 blocks
int row_blocks_fwd(const mmt_rows_desc* d) {      // persistent forward kernels: at most the waves an MI355X holds at once
  const long need = (d->rows + 3) / 4;
  return (int)(need < 1024 ? need : 1024);
}

int row_blocks(const mmt_rows_desc* d) {
  const long need = (d->rows + 3) / 4;
  return (int)(need < kRowBlocks ? need : kRowBlocks);
}

void fill(mmt::LayerParams& p, const mmt_rows_desc* d) {
  p = mmt::LayerParams{};
  p.rows = d->rows; p.H = d->H; p.eps = d->eps;
  if (d->dropout_p > 0.f) {
    p.thresh16 = mmt::dropout_thresh16(d->dropout_p);
    p.inv_keep = mmt::dropout_inv_keep(p.thresh16);          // exact keep probability of the 16-bit test
    p.seed_lo = (uint32_t)d->dropout_seed; p.seed_hi = (uint32_t)(d->dropout_seed >> 32);
    p.epoch = reinterpret_cast<const unsigned long long*>(d->dropout_epoch);
  }
}

// (chunk width, chunks per lane) for a row length: 4-wide chunks when they tile the row with fewer
// idle lanes (H = 768: 3 x 4 per lane instead of 2 x 8 with half the lanes idle in the second).
void row_tiling(int H, int& W, int& nchl) {
  const int n8 = ((H >> 3) + 63) / 64, n4 = ((H >> 2) + 63) / 64;
  if (n4 <= 3 && n4 * 256 < n8 * 512) { W = 4; nchl = n4; }
  else { W = 8; nchl = n8 <= 2 ? n8 : (n8 <= 4 ? 4 : 8); }
}

template <bool RESID, bool HAS_LN>
hipError_t launch_fwd(const mmt::LayerParams& p, bool bf16, hipStream_t st, int blocks) {
  int W, nchl; row_tiling(p.H, W, nchl);
#define MMT_FWD(T, WW, N) hipLaunchKernelGGL((mmt::resid_ln_fwd_kernel<T, WW, N, RESID, HAS_LN>), dim3(blocks), dim3(256), 0, st, p)
#define MMT_FWD_T(T) \
  if (W == 4) { if (nchl <= 1) MMT_FWD(T, 4, 1); else if (nchl <= 2) MMT_FWD(T, 4, 2); else MMT_FWD(T, 4, 3); } \
  else { if (nchl <= 1) MMT_FWD(T, 8, 1); else if (nchl <= 2) MMT_FWD(T, 8, 2); else if (nchl <= 4) MMT_FWD(T, 8, 4); else MMT_FWD(T, 8, 8); }
  if (bf16) { MMT_FWD_T(__bf16) } else { MMT_FWD_T(float) }
#undef MMT_FWD_T
#undef MMT_FWD
  return hipGetLastError();
}

template <bool RESID, bool HAS_LN>
hipError_t launch_bwd(const mmt::LayerParams& p, bool bf16, hipStream_t st, int blocks) {
  int W, nchl; row_tiling(p.H, W, nchl);
#define MMT_BWD(T, WW, N) hipLaunchKernelGGL((mmt::resid_ln_bwd_kernel<T, WW, N, RESID, HAS_LN>), dim3(blocks), dim3(256), 0, st, p)
#define MMT_BWD_T(T) \
  if (W == 4) { if (nchl <= 1) MMT_BWD(T, 4, 1); else if (nchl <= 2) MMT_BWD(T, 4, 2); else MMT_BWD(T, 4, 3); } \
  else { if (nchl <= 1) MMT_BWD(T, 8, 1); else if (nchl <= 2) MMT_BWD(T, 8, 2); else MMT_BWD(T, 8, 4); }
  if (bf16) { MMT_BWD_T(__bf16) } else { MMT_BWD_T(float) }
#undef MMT_BWD_T
#undef MMT_BWD
  return hipGetLastError();
}

constexpr int kMaxLnH = 2048;   // row kept in registers: 4 chunks of 8 per lane

}  // namespace

extern "C" {

size_t mmt_layer_workspace_bytes(const mmt_rows_desc* d) {
  if (!d || d->H <= 0) return 0;
  const size_t a = (size_t)kRowBlocks * 3 * d->H * sizeof(float);
  const size_t b = (size_t)kGeluRowSplit * d->H * sizeof(float);
  return a > b ? a : b;
}

int mmt_ln_fwd(const mmt_rows_desc* d, const void* x, const float* gamma, const float* beta,
               void* y, float* mean, float* rstd, void* stream) {
  if (int rc = check_rows(d, kMaxLnH)) return rc;
  if (!x || !gamma || !beta || !y || !mean || !rstd) return lfail(MMT_E_INVALID, "mmt_ln_fwd: NULL argument");
  mmt::LayerParams p; fill(p, d);
  p.b = x; p.p1 = gamma; p.p2 = beta; p.o1 = y; p.s0 = mean; p.s1 = rstd;
  hipError_t e = launch_fwd<false, true>(p, d->dtype == MMT_BF16, (hipStream_t)stream, row_blocks_fwd(d));
  return e == hipSuccess ? MMT_OK : lfail(MMT_E_LAUNCH, "mmt_ln_fwd: %s", hipGetErrorString(e));
}

int mmt_ln_bwd(const mmt_rows_desc* d, const void* dy, const void* x, const float* gamma,
               const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta,
               void* ws, size_t ws_bytes, void* stream) {
  return mmt_ln_bwd_add(d, dy, x, gamma, mean, rstd, nullptr, dx, dgamma, dbeta, ws, ws_bytes, stream);
}

int mmt_ln_bwd_add(const mmt_rows_desc* d, const void* dy, const void* x, const float* gamma,
                   const float* mean, const float* rstd, const void* dx_in, void* dx, float* dgamma, float* dbeta,
                   void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_rows(d, kMaxLnH)) return rc;
  if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta) return lfail(MMT_E_INVALID, "mmt_ln_bwd: NULL argument");
  if (!ws || ws_bytes < mmt_layer_workspace_bytes(d)) return lfail(MMT_E_WORKSPACE, "mmt_ln_bwd: workspace too small");
  mmt::LayerParams p; fill(p, d);
  p.a = dx_in; p.b = dy; p.c = x; p.p1 = gamma; p.mean = mean; p.rstd = rstd; p.o1 = dx; p.part = (float*)ws;
  p.nblocks = row_blocks(d);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = launch_bwd<false, true>(p, d->dtype == MMT_BF16, st, p.nblocks);
  if (e != hipSuccess) return lfail(MMT_E_LAUNCH, "mmt_ln_bwd: %s", hipGetErrorString(e));
  if (d->defer_reduce) return MMT_OK;
  hipLaunchKernelGGL(mmt::colsum_reduce_kernel, dim3((2 * d->H + mmt::kCsCols - 1) / mmt::kCsCols), dim3(1024), 0, st, p.part, p.nblocks, 2, d->H, dgamma, dbeta, (float*)nullptr, d->accumulate);
  e = hipGetLastError();
  return e == hipSuccess ? MMT_OK : lfail(MMT_E_LAUNCH, "mmt_ln_bwd reduce: %s", hipGetErrorString(e));
}

int mmt_residual_block_fwd(const mmt_rows_desc* d, const void* o, const float* bias, const void* x,
                           const float* gamma, const float* beta, void* x_new, void* h, float* mean,
                           float* rstd, void* stream) {
  if (int rc = check_rows(d, kMaxLnH)) return rc;
  if (!o || !bias || !x || !x_new) return lfail(MMT_E_INVALID, "mmt_residual_block_fwd: NULL argument");
  if (gamma && (!beta || !h || !mean || !rstd)) return lfail(MMT_E_INVALID, "mmt_residual_block_fwd: LayerNorm outputs missing");
  mmt::LayerParams p; fill(p, d);
  p.a = o; p.b = x; p.p0 = bias; p.p1 = gamma; p.p2 = beta; p.o0 = x_new; p.o1 = h; p.s0 = mean; p.s1 = rstd;
  const bool bf16 = d->dtype == MMT_BF16;
  hipError_t e = gamma ? launch_fwd<true, true>(p, bf16, (hipStream_t)stream, row_blocks_fwd(d))
                       : launch_fwd<true, false>(p, bf16, (hipStream_t)stream, row_blocks_fwd(d));
  return e == hipSuccess ? MMT_OK : lfail(MMT_E_LAUNCH, "mmt_residual_block_fwd: %s", hipGetErrorString(e));
}

int mmt_residual_block_bwd(const mmt_rows_desc* d, const void* dx_new_in, const void* dh,
                           const void* x_new, const float* gamma, const float* mean, const float* rstd,
                           void* d_o, void* dx, float* dbias, float* dgamma, float* dbeta,
                           void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_rows(d, kMaxLnH)) return rc;
  const bool has_ln = gamma != nullptr;
  if (!d_o || !dx || !dbias) return lfail(MMT_E_INVALID, "mmt_residual_block_bwd: NULL output");
  if (has_ln && (!dh || !x_new || !mean || !rstd || !dgamma || !dbeta)) return lfail(MMT_E_INVALID, "mmt_residual_block_bwd: LayerNorm arguments missing");
  if (!has_ln && !dx_new_in) return lfail(MMT_E_INVALID, "mmt_residual_block_bwd: no incoming gradient");
  if (!ws || ws_bytes < mmt_layer_workspace_bytes(d)) return lfail(MMT_E_WORKSPACE, "mmt_residual_block_bwd: workspace too small");
  mmt::LayerParams p; fill(p, d);
  p.a = dx_new_in; p.b = dh; p.c = x_new; p.p1 = gamma; p.mean = mean; p.rstd = rstd; p.o0 = d_o; p.o1 = dx;
  p.part = (float*)ws; p.nblocks = row_blocks(d);
  hipStream_t st = (hipStream_t)stream;
  const bool bf16 = d->dtype == MMT_BF16;
  hipError_t e = has_ln ? launch_bwd<true, true>(p, bf16, st, p.nblocks) : launch_bwd<true, false>(p, bf16, st, p.nblocks);
  if (e != hipSuccess) return lfail(MMT_E_LAUNCH, "mmt_residual_block_bwd: %s", hipGetErrorString(e));
  const int ksets = has_ln ? 3 : 1;
  if (d->defer_reduce) return MMT_OK;
  hipLaunchKernelGGL(mmt::colsum_reduce_kernel, dim3((ksets * d->H + mmt::kCsCols - 1) / mmt::kCsCols), dim3(1024), 0, st, p.part, p.nblocks, ksets, d->H, dbias, dgamma, dbeta, d->accumulate);
  e = hipGetLastError();
  return e == hipSuccess ? MMT_OK : lfail(MMT_E_LAUNCH, "mmt_residual_block_bwd reduce: %s", hipGetErrorString(e));
}

int mmt_colsum_reduce(const mmt_rows_desc* d, int32_t kind, const void* ws, float* o0, float* o1, float* o2,
                      void* stream) {
  if (int rc = check_rows(d, 8192)) return rc;
  if (!ws || !o0 || kind < 0 || kind > 3) return lfail(MMT_E_INVALID, "mmt_colsum_reduce: bad argument");
  int nblocks, ksets;
  if (kind == 3) { nblocks = (int)(d->rows < kGeluRowSplit ? d->rows : kGeluRowSplit); ksets = 1; }
  else { nblocks = row_blocks(d); ksets = kind == 0 ? 2 : (kind == 1 ? 3 : 1); }
  if ((ksets >= 2 && !o1) || (ksets >= 3 && !o2)) return lfail(MMT_E_INVALID, "mmt_colsum_reduce: output missing");
  hipError_t e = mmt::launch_colsum_reduce((const float*)ws, nblocks, ksets, d->H, o0, o1, o2, d->accumulate, (hipStream_t)stream);
  return e == hipSuccess ? MMT_OK : lfail(MMT_E_LAUNCH, "mmt_colsum_reduce: %s", hipGetErrorString(e));
}

int mmt_colsum_reduce_batch(int32_t n, const mmt_colsum_item* items, void* stream) {
  if (n < 1 || n > mmt::kCsBatchMax || !items) return lfail(MMT_E_INVALID, "mmt_colsum_reduce_batch: needs 1..%d items", mmt::kCsBatchMax);
  mmt::ColsumBatch g;
  std::memset(&g, 0, sizeof(g));
  g.n = n;
  int end = 0;
  for (int i = 0; i < n; ++i) {
    const mmt_colsum_item& it = items[i];
    if (!it.workspace || !it.o0 || it.kind < 0 || it.kind > 3 || it.rows <= 0 || it.H <= 0 || (it.H & 7) || it.H > 8192)
      return lfail(MMT_E_INVALID, "mmt_colsum_reduce_batch: bad item %d", i);
    const int ksets = it.kind == 0 ? 2 : (it.kind == 1 ? 3 : 1);
    if ((ksets >= 2 && !it.o1) || (ksets >= 3 && !it.o2)) return lfail(MMT_E_INVALID, "mmt_colsum_reduce_batch: output missing in item %d", i);
    mmt_rows_desc d{}; d.rows = it.rows; d.H = it.H;
    g.part[i] = (const float*)it.workspace; g.o0[i] = it.o0; g.o1[i] = it.o1; g.o2[i] = it.o2;
    g.nblocks[i] = it.kind == 3 ? (int)(it.rows < kGeluRowSplit ? it.rows : kGeluRowSplit) : row_blocks(&d);
    g.H[i] = it.H; g.ksets[i] = (unsigned char)ksets; g.accumulate[i] = it.accumulate ? 1 : 0;
    end += (ksets * it.H + mmt::kCsCols - 1) / mmt::kCsCols;
    g.blocks_end[i] = end;
  }
  hipError_t e = mmt::launch_colsum_reduce_batch(g, (hipStream_t)stream);
  return e == hipSuccess ? MMT_OK : lfail(MMT_E_LAUNCH, "mmt_colsum_reduce_batch: %s", hipGetErrorString(e));
}

int mmt_grad_clip_scale(int32_t n_slabs, const float* const* slabs, const int64_t* sizes, float max_norm,
                        float pending_scale, float* scale_out, float* norm_out, void* workspace,
                        size_t workspace_bytes, void* stream) {
  if (n_slabs < 1 || n_slabs > mmt::kMaxNormSlabs || !slabs || !sizes || !scale_out)
    return lfail(MMT_E_INVALID, "mmt_grad_clip_scale: needs 1..%d slabs and an output", mmt::kMaxNormSlabs);
  if (!workspace || workspace_bytes < (size_t)mmt::kNormBlocks * sizeof(float))
    return lfail(MMT_E_WORKSPACE, "mmt_grad_clip_scale: workspace of %zu bytes needed", (size_t)mmt::kNormBlocks * sizeof(float));
  mmt::NormSlabs a;
  a.n = n_slabs;
  for (int i = 0; i < n_slabs; ++i) {
    if (!slabs[i] || sizes[i] <= 0 || (sizes[i] & 3) || ((uintptr_t)slabs[i] & 15))
      return lfail(MMT_E_INVALID, "mmt_grad_clip_scale: slabs must be 16-byte aligned with a multiple of 4 elements");
    a.ptr[i] = slabs[i]; a.n4[i] = sizes[i] >> 2;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mmt::sumsq_partial_kernel, dim3(mmt::kNormBlocks), dim3(256), 0, st, a, (float*)workspace);
  hipLaunchKernelGGL(mmt::clip_scale_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, max_norm, pending_scale, scale_out, norm_out);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MMT_OK : lfail(MMT_E_LAUNCH, "mmt_grad_clip_scale: %s", hipGetErrorString(e));
}

int mmt_accumulate_grad(float* acc, const void* g, int32_t g_dtype, int64_t n, void* stream) {
  if (!acc || !g || n <= 0) return lfail(MMT_E_INVALID, "mmt_accumulate_grad: NULL argument or n <= 0");
  if (g_dtype != MMT_F32 && g_dtype != MMT_BF16) return lfail(MMT_E_INVALID, "mmt_accumulate_grad: bad dtype");
  if (((uintptr_t)acc & 15) || ((uintptr_t)g & (g_dtype == MMT_BF16 ? 15 : 15)))
    return lfail(MMT_E_INVALID, "mmt_accumulate_grad: buffers must be 16-byte aligned");
  long blocks = ((n >> 3) + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  if (g_dtype == MMT_BF16) hipLaunchKernelGGL(mmt::accumulate_grad_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, acc, (const __bf16*)g, (long)n);
  else hipLaunchKernelGGL(mmt::accumulate_grad_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, acc, (const float*)g, (long)n);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MMT_OK : lfail(MMT_E_LAUNCH, "mmt_accumulate_grad: %s", hipGetErrorString(e));
}

int mmt_adamw_step(const mmt_adamw_desc* d, float* param, float* grad, float* exp_avg, float* exp_avg_sq,
                   void* param_bf16, const float* chunk_wd, const float* grad_scale, void* stream) {
  if (!d || !param || !grad || !exp_avg || !exp_avg_sq || !chunk_wd) return lfail(MMT_E_INVALID, "mmt_adamw_step: NULL argument");
  if (d->n <= 0 || (d->n & 1023)) return lfail(MMT_E_INVALID, "mmt_adamw_step: n must be a positive multiple of 1024");
  if (!(d->bias_correction1 > 0.f) || !(d->bias_correction2 > 0.f)) return lfail(MMT_E_INVALID, "mmt_adamw_step: bias corrections must be positive");
  mmt::AdamwParams a{d->lr, d->beta1, d->beta2, d->eps, 1.f / d->bias_correction1, 1.f / sqrtf(d->bias_correction2), d->zero_grad};
  const long n_chunks = d->n >> 10;
  const long blocks = n_chunks < 8192 ? n_chunks : 8192;
  hipLaunchKernelGGL(mmt::adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, (__bf16*)param_bf16, chunk_wd, grad_scale, n_chunks, a, d->hyper);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MMT_OK : lfail(MMT_E_LAUNCH, "mmt_adamw_step: %s", hipGetErrorString(e));
}

size_t mmt_colsum_workspace_bytes(int64_t rows, int32_t C) {
  if (rows <= 0 || C <= 0) return 0;
  return (size_t)(rows < 16 ? rows : 16) * C * sizeof(float);
}

int mmt_colsum(int64_t rows, int32_t C, int32_t dtype, const void* x, int64_t ld, float* out, int32_t accumulate,
               void* ws, size_t ws_bytes, void* stream) {
  if (!x || !out) return lfail(MMT_E_INVALID, "mmt_colsum: NULL argument");
  if (rows <= 0 || C <= 0 || ld < C) return lfail(MMT_E_INVALID, "mmt_colsum: bad shape");
  if (dtype != MMT_F32 && dtype != MMT_BF16) return lfail(MMT_E_INVALID, "mmt_colsum: bad dtype %d", dtype);
  if (!ws || ws_bytes < mmt_colsum_workspace_bytes(rows, C)) return lfail(MMT_E_WORKSPACE, "mmt_colsum: workspace too small");
  const int gy = (int)(rows < 16 ? rows : 16);
  dim3 grid(((C + 1) / 2 + 255) / 256, gy);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MMT_BF16) hipLaunchKernelGGL(mmt::colsum_pairs_kernel<__bf16>, grid, dim3(256), 0, st, (const __bf16*)x, (long)ld, (long)rows, C, (float*)ws);
  else hipLaunchKernelGGL(mmt::colsum_pairs_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (long)ld, (long)rows, C, (float*)ws);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return lfail(MMT_E_LAUNCH, "mmt_colsum: %s", hipGetErrorString(e));
  e = mmt::launch_colsum_reduce((const float*)ws, gy, 1, C, out, nullptr, nullptr, accumulate, st);
  return e == hipSuccess ? MMT_OK : lfail(MMT_E_LAUNCH, "mmt_colsum reduce: %s", hipGetErrorString(e));
}

int mmt_bias_gelu_fwd(const mmt_rows_desc* d, const void* u, const float* bias, void* y, void* stream) {
  if (int rc = check_rows(d, 8192)) return rc;
  if (!u || !bias || !y) return lfail(MMT_E_INVALID, "mmt_bias_gelu_fwd: NULL argument");
  mmt::LayerParams p; fill(p, d);
  p.a = u; p.p0 = bias; p.o0 = y;
  const int nch = d->H >> 3;
  // blocks: a multiple of nch / gcd(nch, 256) (so that blocks * 256 is a multiple of nch), about 16 chunks per thread
  int a = nch, b = 256;
  while (b) { const int t = a % b; a = b; b = t; }
  const long unit = nch / a;
  const long total = d->rows * (long)nch;
  long blocks = (total / (256 * 16) + unit - 1) / unit * unit;
  if (blocks < unit) blocks = unit;
  if (blocks > 8192 / unit * unit && 8192 >= unit) blocks = 8192 / unit * unit;
  if (d->dtype == MMT_BF16) hipLaunchKernelGGL((mmt::bias_gelu_flat_kernel<__bf16>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((mmt::bias_gelu_flat_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MMT_OK : lfail(MMT_E_LAUNCH, "mmt_bias_gelu_fwd: %s", hipGetErrorString(e));
}

int mmt_bias_gelu_bwd(const mmt_rows_desc* d, const void* dy, const void* u, const float* bias, void* du,
                      float* dbias, void* ws, size_t ws_bytes, void* stream) {
  if (int rc = check_rows(d, 8192)) return rc;
  if (!dy || !u || !bias || !du || !dbias) return lfail(MMT_E_INVALID, "mmt_bias_gelu_bwd: NULL argument");
  if (!ws || ws_bytes < mmt_layer_workspace_bytes(d)) return lfail(MMT_E_WORKSPACE, "mmt_bias_gelu_bwd: workspace too small");
  mmt::LayerParams p; fill(p, d);
  p.a = u; p.b = dy; p.p0 = bias; p.o0 = du; p.part = (float*)ws;
  const int nch = d->H >> 3;
  const long gy = d->rows < kGeluRowSplit ? d->rows : kGeluRowSplit;
  dim3 grid((nch + 255) / 256, (unsigned)gy);
  hipStream_t st = (hipStream_t)stream;
  if (d->dtype == MMT_BF16) hipLaunchKernelGGL((mmt::bias_gelu_kernel<__bf16, true>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((mmt::bias_gelu_kernel<float, true>), grid, dim3(256), 0, st, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return lfail(MMT_E_LAUNCH, "mmt_bias_gelu_bwd: %s", hipGetErrorString(e));
  if (d->defer_reduce) return MMT_OK;
  hipLaunchKernelGGL(mmt::colsum_reduce_kernel, dim3((d->H + mmt::kCsCols - 1) / mmt::kCsCols), dim3(1024), 0, st, p.part, (int)gy, 1, d->H, dbias, (float*)nullptr, (float*)nullptr, d->accumulate);
  e = hipGetLastError();
  return e == hipSuccess ? MMT_OK : lfail(MMT_E_LAUNCH, "mmt_bias_gelu_bwd reduce: %s", hipGetErrorString(e));
}

}  // extern "C"
