// Weight-gradient GEMM for gfx950:  dW[M,N] += dY[K,M]^T . X[K,N]   (bf16 in, fp32 accumulate)
//
// This is the `tape.gradient` product behind every Dense layer of the encoder
// (src/tasks/pretraining.py:292-296): K = B*S rows (16384 at BASELINE config 3) are contracted,
// M = output features, N = input features, both operands row-major with the contracted index as
// the slow dimension -- the layout hipBLASLt handles worst (175-490 TF/s measured).
//
// Structure: workgroup tile 128 (M) x 256 (N), 4 waves of 64 x 128 (128 accumulator registers),
// split-K over the rows.  Per 32-row step the dY and X slabs are staged row-major in LDS
// (register-staged, double-buffered) in the 32 x 128-byte tile format of the attention kernels;
// BOTH MFMA operands are then "column reads" of those tiles, fetched with ds_read_b64_tr_b16 in
// the same k-permuted order (attn_tile.h: mma_xt), so no transposed copy of either matrix ever
// exists.  The epilogue adds the fp32 tile into the master-gradient buffer with float atomics
// (one 128-byte row segment per half-wave, adders = split-K factor), i.e. the gradient
// accumulation (`AccumulateGrad`) is fused and dW is never rounded to bf16.
#include "../../include/mmt_attn.h"
#include "../../include/mmt_layer.h"
#include "attn_tile.h"
#include "mmt_err.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>

namespace mmt {

constexpr int kWgTN = 256;
// MW = wave rows along M: workgroup = 2*MW waves, tile (64*MW) x 256
template <int MW> struct WgCfg {
  static constexpr int kTM = 64 * MW, kThreads = 128 * MW, kBPer = 8 / MW;
  static constexpr int kStageBytes = (kTM + kWgTN) * 32 * 2;
};

struct WgradParams {
  const __bf16* dy;   // [K, M] row stride ldy
  const __bf16* x;    // [K, N] row stride ldx
  float* dw;          // [M, N] row stride ldw (fp32, accumulated into)
  float* slabs;       // [split, M, N] fp32 partials (plain stores) or NULL -> float atomics into dw
  float* dbias;       // [M] fp32 += column sums of dy (NULL: not wanted); written by the tn == 0 tiles
  float* bias_part;   // [split, M] partial column sums (slab mode) or NULL -> float atomics into dbias
  long ldy, ldx, ldw;
  int M, N, K;
  int tiles_n, tiles_m, k_per_split;   // rows per split-K slice (multiple of 32)
};

__device__ __forceinline__ int vtile_off(int row, int ch) {      // byte offset inside one 32x128B tile
  return row * 128 + ((((ch >> 2) ^ ((row >> 1) & 1))) << 6) + (ch & 3) * 16;
}

template <int MW>
__global__ __launch_bounds__(128 * MW, 2) void wgrad_kernel(const WgradParams p) {
  using C = WgCfg<MW>;
  constexpr int kWgTM = C::kTM, kWgStageBytes = C::kStageBytes;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, li = lane & 15, cb = (lane >> 4) & 1, r = lane & 31;
  const int tile = blockIdx.x, ks = blockIdx.y;
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int m0 = tm * kWgTM, n0 = tn * kWgTN;
  const int k_begin = ks * p.k_per_split;
  const int k_end = min(p.K, k_begin + p.k_per_split);
  const int n_steps = (k_end - k_begin + 31) >> 5;      // the last step may be ragged: rows >= K stage as zeros
  if (n_steps <= 0) return;

  // ---- staging map: 16-byte chunks; A slab 32 x 128 cols = 512 chunks (2 per thread),
  //      B slab 32 x 256 cols = 1024 chunks (4 per thread) -------------------------------------
  constexpr int kAChunksRow = kWgTM / 8;      // 16-byte chunks per A-slab row
  const __bf16* ga[2]; int la[2], ka[2];
  const __bf16* gb[C::kBPer]; int lb[C::kBPer], kb[C::kBPer];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int c = tid + C::kThreads * u, row = c / kAChunksRow, ch = c % kAChunksRow;
    ga[u] = p.dy + (long)(k_begin + row) * p.ldy + m0 + ch * 8;
    la[u] = (ch >> 3) * 4096 + vtile_off(row, ch & 7);
    ka[u] = k_begin + row;
  }
#pragma unroll
  for (int u = 0; u < C::kBPer; ++u) {
    const int c = tid + C::kThreads * u, row = c >> 5, ch = c & 31;
    gb[u] = p.x + (long)(k_begin + row) * p.ldx + n0 + ch * 8;
    lb[u] = kWgTM * 64 + (ch >> 3) * 4096 + vtile_off(row, ch & 7);
    kb[u] = k_begin + row;
  }
  const long astep = 32 * p.ldy, bstep = 32 * p.ldx;

  // ---- fragment read offsets (k-step s adds 16 rows = 2048 B; second read is 8 rows = 1024 B on)
  const int frow = 4 * h + (li >> 2);
  int fo[2];      // column block db = 0/1 inside a tile
#pragma unroll
  for (int db = 0; db < 2; ++db) fo[db] = frow * 128 + ((db ^ ((frow >> 1) & 1)) << 6) + 32 * cb + 8 * (li & 3);
  const int wm = wave % MW, wn = wave / MW;
  const int a_tile = wm * 4096;                                  // this wave's 64 M columns
  const int b_tile = kWgTM * 64 + wn * 2 * 4096;                 // this wave's 128 N columns

  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x16{0};
  // bias gradient = column sums of dy: the tn == 0 workgroups add up the dy chunks they stage anyway
  // (VALU work beside an LDS/MFMA-bound loop); thread -> fixed 8 columns, rows tid/kAChunksRow (+16)
  const bool do_bias = p.dbias != nullptr && tn == 0;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  auto add_bias = [&](const bf16x8 (&ra_)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) bsum[j] += (float)ra_[u][j];
  };

  bf16x8 ra[2], rb[C::kBPer];
#pragma unroll
  for (int u = 0; u < 2; ++u) ra[u] = ka[u] < k_end ? *reinterpret_cast<const bf16x8*>(ga[u]) : bf16x8{0};
#pragma unroll
  for (int u = 0; u < C::kBPer; ++u) rb[u] = kb[u] < k_end ? *reinterpret_cast<const bf16x8*>(gb[u]) : bf16x8{0};
#pragma unroll
  for (int u = 0; u < 2; ++u) *reinterpret_cast<bf16x8*>(smem + la[u]) = ra[u];
#pragma unroll
  for (int u = 0; u < C::kBPer; ++u) *reinterpret_cast<bf16x8*>(smem + lb[u]) = rb[u];
  if (do_bias) add_bias(ra);
  __syncthreads();

  for (int step = 0; step < n_steps; ++step) {
    unsigned char* cur = smem + (step & 1) * kWgStageBytes;
    unsigned char* nxt = smem + ((step + 1) & 1) * kWgStageBytes;
    const bool more = step + 1 < n_steps;
    if (more) {          // next slab: global -> registers, in flight under this step's MFMAs
      const int kleft = k_end - 32 * (step + 1);
#pragma unroll
      for (int u = 0; u < 2; ++u) ra[u] = ka[u] < kleft ? *reinterpret_cast<const bf16x8*>(ga[u] + (long)(step + 1) * astep) : bf16x8{0};
#pragma unroll
      for (int u = 0; u < C::kBPer; ++u) rb[u] = kb[u] < kleft ? *reinterpret_cast<const bf16x8*>(gb[u] + (long)(step + 1) * bstep) : bf16x8{0};
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[2], bfr[4];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const unsigned char* base = cur + a_tile + fo[a] + s * 2048;
        const bf16x4 lo = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base)));
        const bf16x4 hi = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 1024)));
#pragma unroll
        for (int j = 0; j < 4; ++j) { af[a][j] = lo[j]; af[a][4 + j] = hi[j]; }
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const unsigned char* base = cur + b_tile + (b >> 1) * 4096 + fo[b & 1] + s * 2048;
        const bf16x4 lo = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base)));
        const bf16x4 hi = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 1024)));
#pragma unroll
        for (int j = 0; j < 4; ++j) { bfr[b][j] = lo[j]; bfr[b][4 + j] = hi[j]; }
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
    }
    if (more) {
#pragma unroll
      for (int u = 0; u < 2; ++u) *reinterpret_cast<bf16x8*>(nxt + la[u]) = ra[u];
#pragma unroll
      for (int u = 0; u < C::kBPer; ++u) *reinterpret_cast<bf16x8*>(nxt + lb[u]) = rb[u];
      if (do_bias) add_bias(ra);
    }
    __syncthreads();     // nxt is complete, cur is free (one barrier per step: writes go to the other buffer)
  }

  if (do_bias) {       // 16 row groups x kWgTM columns through the (now idle) staging LDS, fixed order
    float* red = reinterpret_cast<float*>(smem);
    const int grp = tid / kAChunksRow, ch = tid % kAChunksRow;
#pragma unroll
    for (int j = 0; j < 8; ++j) red[grp * kWgTM + ch * 8 + j] = bsum[j];
    __syncthreads();
    if (tid < kWgTM) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) t += red[g * kWgTM + tid];
      if (p.bias_part) p.bias_part[(long)ks * p.M + m0 + tid] = t;
      else atomicAdd(p.dbias + m0 + tid, t);
    }
  }

  // ---- epilogue: dW[m][n] += acc   (accumulator register i of lane (r,h): row kap(i,h), col r) ----
  if (p.slabs) {
    float* out = p.slabs + (long)ks * p.M * p.N + (long)(m0 + wm * 64) * p.N + n0 + wn * 128 + r;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          out[(long)(32 * a + kap(i, h)) * p.N + 32 * b] = acc[a][b][i];
    return;
  }
  float* out = p.dw + (long)(m0 + wm * 64) * p.ldw + n0 + wn * 128 + r;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        atomicAdd(out + (long)(32 * a + kap(i, h)) * p.ldw + 32 * b, acc[a][b][i]);
}

// ---------------------------------------------------------------------------------------------------
// LDS-DMA variant of the 256 x 256 tile (8 waves of 64 x 128): the slabs go global -> LDS with
// global_load_lds_dwordx4 (no staging registers, no ds_write pass in front of the barrier -- the
// measured cost of that pass was ~40 % of the loop), 64 rows per step (one barrier per 32 MFMAs per
// wave instead of 16).  One DMA wave-instruction fills 1 KiB of LDS lane-linearly = 8 rows of one
// 32 x 128-byte tile, so the tile's swizzle (64-byte halves swapped when bit 1 of the row is set) is
// applied on the SOURCE side: lane l fetches the 16-byte chunk that belongs at LDS position l.
// Stage image: [k-slab 0|1][A tiles 0-3 | B tiles 0-3][32 rows x 128 B]  = 64 KiB, double buffered.
// Needs every split-K slice to be a multiple of 64 rows.
// ---------------------------------------------------------------------------------------------------
constexpr int kDmaStageBytes = 2 * 8 * 4096;

__device__ __forceinline__ void wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// XCD-aware order: blocks are dealt round-robin over the 8 XCDs (block b -> XCD b % 8); XCD x takes the
// x-th contiguous range of the list ordered (split-K slice, tile), so that the workgroups sharing one L2 read
// the same K rows and mostly the same dy / x column slabs (each slab ~once per L2 instead of once per
// workgroup -- the L2-miss traffic, not the MFMA rate, bounded this kernel).
__device__ __forceinline__ int xcd_list_index() {
  const int nwg = gridDim.x, b = blockIdx.x, x = b & 7;
  const int base = nwg >> 3, rem = nwg & 7;                       // XCD y holds base + (y < rem) blocks
  return x * base + min(x, rem) + (b >> 3);
}

__device__ __forceinline__ void wgrad_dma_body(const WgradParams& p, int ks, int tm, int tn) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, li = lane & 15, cb = (lane >> 4) & 1, r = lane & 31;
  const int m0 = tm * 256, n0 = tn * 256;
  const int k_begin = ks * p.k_per_split;
  const int k_end = min(p.K, k_begin + p.k_per_split);
  const int n_steps = (k_end - k_begin) >> 6;
  if (n_steps <= 0) return;

  // ---- DMA map: the staging LDS is a ring of four 32-row SLABS (8 tiles of 32 x 128 B each: dy tiles 0-3, x tiles
  //      4-7).  Wave w loads tile w of every slab: four 1 KiB instructions (row groups of 8) per slab.
  const int drow = lane >> 3, dpos = lane & 7;                 // row inside the 8-row group, 16-byte position in the LDS row
  const __bf16* gsrc[4];
  int ldst[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = j * 8 + drow;                              // row inside the slab
    const int ch = (((dpos >> 2) ^ ((row >> 1) & 1)) << 2) | (dpos & 3);    // logical chunk stored at this LDS position
    const long krow = k_begin + row;
    gsrc[j] = wave < 4 ? p.dy + krow * p.ldy + m0 + wave * 64 + ch * 8
                       : p.x + krow * p.ldx + n0 + (wave - 4) * 64 + ch * 8;
    ldst[j] = (wave * 4 + j) * 1024;
  }
  const long slab_step = 32 * (wave < 4 ? p.ldy : p.ldx);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  constexpr int kSlabBytes = 8 * 4096;
  // instructions 2*part, 2*part+1 of slab `sl` into ring slot sl & 3 (part < 0: all four).  In the main loop they go
  // out two per k-substep, between the MFMA groups (back to back they overrun the memory pipe's queue)
  auto dma = [&](int sl, int part) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (part >= 0 && (j >> 1) != part) continue;
      glds16(gsrc[j] + (long)sl * slab_step, lds0 + (sl & 3) * kSlabBytes + ldst[j]);
    }
  };

  const int frow = 4 * h + (li >> 2);
  int fo[2];
#pragma unroll
  for (int db = 0; db < 2; ++db) fo[db] = frow * 128 + ((db ^ ((frow >> 1) & 1)) << 6) + 32 * cb + 8 * (li & 3);
  const int wm = wave & 3, wn = wave >> 2;
  const int a_tile = wm * 4096, b_tile = 4 * 4096 + wn * 2 * 4096;

  f32x16 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x16{0};
  // bias gradient: the tn == 0 workgroups sum the dy slab out of LDS; thread -> logical chunk (tid & 31)
  // of the 256 columns, rows (tid >> 5) + 16 v
  const bool do_bias = p.dbias != nullptr && tn == 0;
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int bch = tid & 31, brow0 = tid >> 5;

  // Ring schedule: three slabs are in flight ahead of the one being consumed (1.5 steps of 64 rows instead of 1 with
  // two 64-row buffers: an HBM miss takes longer than one step).  Per slab: a COUNTED wait for this wave's own four
  // instructions of the slab (the 8 of the two later slabs stay in flight), one raw barrier -- every wave's part has
  // landed and every wave is done reading the slot that is refilled next -- then the refill and 16 MFMAs per wave.
  const int n_slabs = 2 * n_steps;
  dma(0, -1);
  if (n_slabs > 1) dma(1, -1);
  if (n_slabs > 2) dma(2, -1);
  for (int sl = 0; sl < n_slabs; ++sl) {
    const int behind = n_slabs - 1 - sl;                   // slabs issued after this one (at most 2 right now)
    if (behind >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (behind == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's LDS reads of the previous slab have returned
    __builtin_amdgcn_s_barrier();
    const unsigned char* cur = smem + (sl & 3) * kSlabBytes;
    const bool more = sl + 3 < n_slabs;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      if (more) dma(sl + 3, s);                            // into the slot of slab sl - 1: every wave has left it
      const unsigned char* slab = cur + s * 2048;
      bf16x8 af[2], bfr[4];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const unsigned char* base = slab + a_tile + fo[a];
        const bf16x4 lo = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base)));
        const bf16x4 hi = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 1024)));
#pragma unroll
        for (int j = 0; j < 4; ++j) { af[a][j] = lo[j]; af[a][4 + j] = hi[j]; }
      }
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const unsigned char* base = slab + b_tile + (b >> 1) * 4096 + fo[b & 1];
        const bf16x4 lo = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base)));
        const bf16x4 hi = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 1024)));
#pragma unroll
        for (int j = 0; j < 4; ++j) { bfr[b][j] = lo[j]; bfr[b][4 + j] = hi[j]; }
      }
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
    }
    if (do_bias) {
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const int rt = brow0 + 16 * v;                   // 0..31: row inside the slab
        const int t4 = bch >> 3, c8 = bch & 7;
        const unsigned char* src = cur + t4 * 4096 + rt * 128 + ((((c8 >> 2) ^ ((rt >> 1) & 1))) << 6) + (c8 & 3) * 16;
        const bf16x8 t = *reinterpret_cast<const bf16x8*>(src);
#pragma unroll
        for (int j = 0; j < 8; ++j) bsum[j] += (float)t[j];
      }
    }
  }
  __syncthreads();

  if (do_bias) {       // 16 row groups x 256 columns through the (now idle) staging LDS, fixed order
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < 8; ++j) red[brow0 * 256 + bch * 8 + j] = bsum[j];
    __syncthreads();
    if (tid < 256) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 16; ++g) t += red[g * 256 + tid];
      if (p.bias_part) p.bias_part[(long)ks * p.M + m0 + tid] = t;
      else if (p.k_per_split >= p.K) p.dbias[m0 + tid] += t;          // one slice: this workgroup owns the rows
      else atomicAdd(p.dbias + m0 + tid, t);
    }
  }

  if (p.slabs) {
    float* out = p.slabs + (long)ks * p.M * p.N + (long)(m0 + wm * 64) * p.N + n0 + wn * 128 + r;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          out[(long)(32 * a + kap(i, h)) * p.N + 32 * b] = acc[a][b][i];
    return;
  }
  float* out = p.dw + (long)(m0 + wm * 64) * p.ldw + n0 + wn * 128 + r;
  if (p.k_per_split >= p.K) {      // one slice of K: the tile belongs to this workgroup alone -- plain read-add-store
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float* o = out + (long)(32 * a + kap(i, h)) * p.ldw + 32 * b;
          *o += acc[a][b][i];
        }
    return;
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        atomicAdd(out + (long)(32 * a + kap(i, h)) * p.ldw + 32 * b, acc[a][b][i]);
}

__global__ __launch_bounds__(512, 2) void wgrad_dma_kernel(const WgradParams p) {
  const int L = xcd_list_index();
  const int tiles = p.tiles_n * p.tiles_m;
  const int ks = L / tiles, t = L - ks * tiles;
  const int tm = t / p.tiles_n;
  wgrad_dma_body(p, ks, tm, t - tm * p.tiles_n);
}

// Up to eight weight-gradient products in ONE launch (the four Dense layers of one or two encoder blocks: same K rows,
// same 256 x 256 tiles).  Alone, a 768 x 768 product has 9 tiles and needs a 24-way split of K to fill the
// chip -- 24 fp32 slabs written and read back for one gradient; together the block's four products have 108
// tiles, two K slices fill the chip, and the slab traffic falls from ~230 MB to ~57 MB per block.
struct WgradGroup {
  WgradParams p[8];
  int tiles_end[8];      // running tile count
  int n, tiles_total;
};
__global__ __launch_bounds__(512, 2) void wgrad_dma_group_kernel(const WgradGroup g) {
  const int L = xcd_list_index();
  const int ks = L / g.tiles_total;
  int t = L - ks * g.tiles_total;
  int j = 0;
  while (j + 1 < g.n && t >= g.tiles_end[j]) ++j;
  if (j > 0) t -= g.tiles_end[j - 1];
  const WgradParams p = g.p[j];
  // Tile order inside a problem: the SHORTER tile dimension runs fastest, so that the ~27 consecutive tiles one XCD
  // works on at a time form a block as square as the problem allows -- e.g. the 3 x 12 tiles of the FFN output
  // weight: 3 dy column blocks + 9 x column blocks per 27 tiles instead of 3 + 12 (fewer distinct operand
  // blocks per 64-row step = fewer L2 fills).
  int tm, tn;
  if (p.tiles_m < p.tiles_n) { tn = t / p.tiles_m; tm = t - tn * p.tiles_m; }
  else { tm = t / p.tiles_n; tn = t - tm * p.tiles_n; }
  wgrad_dma_body(p, ks, tm, tn);
}

// ---- more tiles than compute units: whole rounds of UNSPLIT tiles plus a split tail ---------------------------------------
// A block's four products have 108 tiles: two K slices make 216 workgroups for 256 CUs (84 % of the chip, and every
// tile goes through two fp32 slabs and a reduce).  With the products of SEVERAL blocks in one launch -- a host that has
// no gradient exchange to overlap can queue a whole backward pass -- the tiles come in rounds: the first
// floor(tiles / CUs) * CUs tiles are taken one per workgroup over the whole K and added into dw in place (no slab, no
// reduce; XCD x owns a contiguous eighth of them, marching K in step as before), the remaining R < CUs tiles are split
// floor(CUs / R)-way (at most 8) into compact per-tile slabs and reduced by a small launch.  12 blocks: 1296 tiles = 5
// rounds + 16 tiles x 8 slices; 5 blocks: 540 = 2 rounds + 28 x 8; 7 blocks: 756 = 2 rounds + 244 x 1.
struct WgradProb { const __bf16* dy; const __bf16* x; float* dw; float* dbias; int ldy, ldx, ldw, M, N, pad; };
constexpr int kBigMax = 28;
struct WgradBig {
  WgradProb q[kBigMax];
  int tiles_end[kBigMax];
  int n, K;
  int full_tiles;                 // tiles [0, full_tiles): one workgroup each, whole K (multiple of 8)
  int tail_tiles, tail_pad;       // the tiles behind them; tail_tiles * tail_split padded to a multiple of 8 workgroups
  int tail_split, tail_kps;
  float* tail_slabs;              // [tail tile][slice][256 x 256]
  float* tail_bias;               // [tail tile][slice][256]
};
__device__ __forceinline__ void big_tile(const WgradBig& g, int t, int& j, int& tm, int& tn) {
  j = 0;
  while (j + 1 < g.n && t >= g.tiles_end[j]) ++j;
  if (j > 0) t -= g.tiles_end[j - 1];
  const int tiles_m = g.q[j].M >> 8, tiles_n = g.q[j].N >> 8;
  if (tiles_m < tiles_n) { tn = t / tiles_m; tm = t - tn * tiles_m; }      // (the shorter dimension fastest, as wgrad_dma_group_kernel)
  else { tm = t / tiles_n; tn = t - tm * tiles_n; }
}
__global__ __launch_bounds__(512, 2) void wgrad_dma_big_kernel(const WgradBig g) {
  // blockIdx -> (XCD x, position i in the XCD's list): the XCD's share of the unsplit tiles first, then its share of the tail
  const int x = blockIdx.x & 7, i = blockIdx.x >> 3;
  const int full8 = g.full_tiles >> 3, tail8 = g.tail_pad >> 3;
  int t, ks = 0, tt = -1;
  if (i < full8) {
    t = x * full8 + i;
  } else {
    const int u = x * tail8 + (i - full8);
    if (u >= g.tail_tiles * g.tail_split) return;
    ks = u / g.tail_tiles; tt = u - ks * g.tail_tiles;
    t = g.full_tiles + tt;
  }
  int j, tm, tn;
  big_tile(g, t, j, tm, tn);
  const WgradProb& q = g.q[j];
  WgradParams p;
  p.dy = q.dy; p.x = q.x; p.dw = q.dw; p.dbias = q.dbias;
  p.ldy = q.ldy; p.ldx = q.ldx; p.ldw = q.ldw; p.K = g.K;
  p.tiles_n = q.N >> 8; p.tiles_m = q.M >> 8;
  if (tt < 0 || g.tail_split == 1) {       // (a tail that fills most of the chip is not split: in place like the rounds)
    p.M = q.M; p.N = q.N; p.k_per_split = g.K; p.slabs = nullptr; p.bias_part = nullptr;
  } else {
    // the tile's own compact slabs: the body addresses [slice][M x N] slabs by (m0, n0) -- with M = N = 256 and the base
    // moved back by the tile's origin it lands in [tail tile][slice][256 x 256]
    p.M = 256; p.N = 256; p.k_per_split = g.tail_kps;
    p.slabs = g.tail_slabs + (size_t)tt * g.tail_split * 65536 - ((long)tm * 65536 + (long)tn * 256);
    p.bias_part = q.dbias ? g.tail_bias + (size_t)tt * g.tail_split * 256 - (long)tm * 256 : nullptr;
  }
  wgrad_dma_body(p, ks, tm, tn);
}
// dw tile += its slices in slice order (and dbias of the tn == 0 tiles); grid = 16 x tail tiles, 16 rows each (with 4 x
// tail tiles -- 112 workgroups walking their slices one dependent load after the other -- the launch took 47 us for 57 MB)
__global__ __launch_bounds__(256) void wgrad_tail_reduce_kernel(const WgradBig g) {
  const int tt = blockIdx.x >> 4, part = blockIdx.x & 15;
  int j, tm, tn;
  big_tile(g, g.full_tiles + tt, j, tm, tn);
  const WgradProb& q = g.q[j];
  const float* slabs = g.tail_slabs + (size_t)tt * g.tail_split * 65536;
  const int n_slices = (g.K + g.tail_kps - 1) / g.tail_kps;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int e = it * 256 + threadIdx.x;
    const int row = part * 16 + (e >> 6), c4 = (e & 63) * 4;
    float* o = q.dw + (long)(tm * 256 + row) * q.ldw + tn * 256 + c4;
    f32x4 a = *reinterpret_cast<const f32x4*>(o);
    f32x4 b[8];                                                    // (tail_split <= 8: every slice's load in flight)
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2)
      b[s2] = s2 < n_slices ? *reinterpret_cast<const f32x4*>(slabs + (size_t)s2 * 65536 + row * 256 + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) { a[0] += b[s2][0]; a[1] += b[s2][1]; a[2] += b[s2][2]; a[3] += b[s2][3]; }
    *reinterpret_cast<f32x4*>(o) = a;
  }
  if (part == 0 && tn == 0 && q.dbias) {
    const float* bp = g.tail_bias + (size_t)tt * g.tail_split * 256;
    float a = q.dbias[tm * 256 + threadIdx.x];
    for (int s2 = 0; s2 < n_slices; ++s2) a += bp[s2 * 256 + threadIdx.x];
    q.dbias[tm * 256 + threadIdx.x] = a;
  }
}

// dw[m][n] += sum_s slabs[s][m][n]   (fixed order: bitwise reproducible)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(float* dw, long ldw, const float* slabs, int split, int M, int N,
                                                           float* dbias, const float* bias_part) {
  const long n4 = (long)M * N / 4;
  if (dbias) {           // bias partials: one column per thread, independent loads
    for (int c = blockIdx.x * 256 + threadIdx.x; c < M; c += gridDim.x * 256) {
      float a = dbias[c];
#pragma unroll 8
      for (int s2 = 0; s2 < split; ++s2) a += bias_part[(long)s2 * M + c];
      dbias[c] = a;
    }
  }
  for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < n4; c += (long)gridDim.x * 256) {
    const long e = c * 4;
    const int m = (int)(e / N), n = (int)(e - (long)m * N);
    f32x4 a = *reinterpret_cast<const f32x4*>(dw + (long)m * ldw + n);
    for (int s2 = 0; s2 < split; ++s2) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(slabs + (long)s2 * M * N + e);
      a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
    }
    *reinterpret_cast<f32x4*>(dw + (long)m * ldw + n) = a;
  }
}

// The reduces of a grouped launch in ONE launch: workgroup ranges [blocks_end[j-1], blocks_end[j]) belong to problem j.
struct ReduceGroup {
  float* dw[8]; const float* slabs[8]; float* dbias[8]; const float* bias_part[8];
  long ldw[8];
  int M[8], N[8], blocks_end[8];
  int n, split;
};
__global__ __launch_bounds__(256) void wgrad_reduce_group_kernel(const ReduceGroup g) {
  int j = 0;
  while (j + 1 < g.n && (int)blockIdx.x >= g.blocks_end[j]) ++j;
  const int b0 = j ? g.blocks_end[j - 1] : 0, nb = g.blocks_end[j] - b0, bid = blockIdx.x - b0;
  const int M = g.M[j], N = g.N[j], split = g.split;
  float* dw = g.dw[j]; const float* slabs = g.slabs[j]; const long ldw = g.ldw[j];
  if (g.dbias[j]) {
    for (int c = bid * 256 + threadIdx.x; c < M; c += nb * 256) {
      float a = g.dbias[j][c];
      for (int s2 = 0; s2 < split; ++s2) a += g.bias_part[j][(long)s2 * M + c];
      g.dbias[j][c] = a;
    }
  }
  const long n4 = (long)M * N / 4;
  for (long c = (long)bid * 256 + threadIdx.x; c < n4; c += (long)nb * 256) {
    const long e = c * 4;
    const int m = (int)(e / N), n = (int)(e - (long)m * N);
    f32x4 a = *reinterpret_cast<const f32x4*>(dw + (long)m * ldw + n);
    for (int s2 = 0; s2 < split; ++s2) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(slabs + (long)s2 * M * N + e);
      a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; a[3] += b[3];
    }
    *reinterpret_cast<f32x4*>(dw + (long)m * ldw + n) = a;
  }
}

}  // namespace mmt

namespace {
// Compute units the GEMM may fill with one round of workgroups (mmt_wgrad_set_cu_budget): 256 on a GPU of
// its own; a data-parallel run leaves some to the collective kernels that overlap backward -- the 256-row
// kernels hold a whole CU per workgroup, so a full-chip grid next to RCCL's resident workgroups would need
// a second, nearly empty round.
std::atomic<int> g_cu_budget{256};
// 256-row tiles (8 waves, one workgroup per CU) when M allows, else 128-row tiles (4 waves, 2 per CU)
int wgrad_tile_m(int M) { return (M % 256) == 0 ? 256 : 128; }
// split-K factor: one full round of workgroups, slices of >= 256 rows
int wgrad_split(int tiles, long K, int tile_m) {
  const int cus = g_cu_budget.load(std::memory_order_relaxed);
  int split = (tile_m == 256 ? cus : 2 * cus) / tiles;
  const int max_split = (int)((K + 255) / 256);
  if (split > max_split) split = max_split;
  return split < 1 ? 1 : split;
}
}  // namespace

extern "C" void mmt_wgrad_set_cu_budget(int32_t cus) {
  g_cu_budget.store(cus < 32 ? 32 : (cus > 256 ? 256 : cus), std::memory_order_relaxed);
}

extern "C" size_t mmt_wgrad_workspace_bytes(int32_t M, int32_t N, int64_t K) {
  if (M <= 0 || N <= 0 || K <= 0 || (M % 128) || (N % mmt::kWgTN)) return 0;
  const int tm = wgrad_tile_m(M);
  return (size_t)wgrad_split((M / tm) * (N / mmt::kWgTN), K, tm) * ((size_t)M * N + M) * sizeof(float);
}

extern "C" int mmt_wgrad_accumulate(float* dw, int64_t ldw, const void* dy, int64_t ldy, const void* x,
                                    int64_t ldx, int32_t M, int32_t N, int64_t K, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  return mmt_wgrad_bias_accumulate(dw, ldw, nullptr, dy, ldy, x, ldx, M, N, K, workspace, workspace_bytes, stream);
}

extern "C" int mmt_wgrad_bias_accumulate(float* dw, int64_t ldw, float* dbias, const void* dy, int64_t ldy,
                                         const void* x, int64_t ldx, int32_t M, int32_t N, int64_t K,
                                         void* workspace, size_t workspace_bytes, void* stream) {
  if (!dw || !dy || !x) return mmt::fail(MMT_E_INVALID, "mmt_wgrad_accumulate: NULL argument");
  if (M <= 0 || N <= 0 || K <= 0 || (M % 128) || (N % mmt::kWgTN))
    return mmt::fail(MMT_E_UNSUPPORTED, "mmt_wgrad_accumulate: needs M %% 128 == 0, N %% 256 == 0 (got %d, %d, K = %lld)", M, N, (long long)K);
  if ((ldy % 8) || (ldx % 8) || ldy < M || ldx < N || ldw < N || (ldw % 4)) return mmt::fail(MMT_E_INVALID, "mmt_wgrad_accumulate: bad leading dimensions");
  if (((uintptr_t)dy & 15) || ((uintptr_t)x & 15) || ((uintptr_t)dw & 15)) return mmt::fail(MMT_E_INVALID, "mmt_wgrad_accumulate: operands must be 16-byte aligned");
  mmt::WgradParams p;
  p.dy = (const __bf16*)dy; p.x = (const __bf16*)x; p.dw = dw;
  p.ldy = ldy; p.ldx = ldx; p.ldw = ldw; p.M = M; p.N = N; p.K = (int)K;
  const int tile_m = wgrad_tile_m(M);
  const int tiles_m = M / tile_m;
  p.tiles_n = N / mmt::kWgTN; p.tiles_m = tiles_m;
  const int tiles = tiles_m * p.tiles_n;
  int split = wgrad_split(tiles, K, tile_m);
  const bool dma = tile_m == 256 && (K % 64) == 0;
  const int kq = dma ? 64 : 32;            // rows per main-loop step
  const int kps = (int)(((K + split - 1) / split + kq - 1) / kq * kq);
  split = (int)((K + kps - 1) / kps);
  p.k_per_split = kps;
  const size_t need = (size_t)split * ((size_t)M * N + M) * sizeof(float);
  p.slabs = (workspace && workspace_bytes >= need && split > 1) ? (float*)workspace : nullptr;
  p.dbias = dbias;
  p.bias_part = (p.slabs && dbias) ? p.slabs + (size_t)split * M * N : nullptr;
  hipStream_t st = (hipStream_t)stream;
  if (dma) {
    const int lds = 2 * mmt::kDmaStageBytes;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mmt::wgrad_dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(mmt::wgrad_dma_kernel, dim3(tiles * split), dim3(512), lds, st, p);
  } else if (tile_m == 256) {
    const int lds = 2 * mmt::WgCfg<4>::kStageBytes;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mmt::wgrad_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(mmt::wgrad_kernel<4>, dim3(tiles, split), dim3(512), lds, st, p);
  } else {
    hipLaunchKernelGGL(mmt::wgrad_kernel<2>, dim3(tiles, split), dim3(256), 2 * mmt::WgCfg<2>::kStageBytes, st, p);
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && p.slabs) {
    long blocks = ((long)M * N / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(mmt::wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dw, (long)ldw, p.slabs, split, M, N, dbias, p.bias_part);
    e = hipGetLastError();
  }
  return e == hipSuccess ? MMT_OK : mmt::fail(MMT_E_LAUNCH, "mmt_wgrad_accumulate: %s", hipGetErrorString(e));
}

// ---- grouped launch (see wgrad_dma_group_kernel) --------------------------------------------------------
namespace {
bool group_ok(int32_t n, const mmt_wgrad_problem* pr, int64_t K) {
  if (n < 1 || n > mmt::kBigMax || !pr || K <= 0 || (K % 64)) return false;
  for (int i = 0; i < n; ++i) {
    const mmt_wgrad_problem& q = pr[i];
    if (!q.dw || !q.dy || !q.x || q.M <= 0 || q.N <= 0 || (q.M % 256) || (q.N % 256)) return false;
    if ((q.ldy % 8) || (q.ldx % 8) || q.ldy < q.M || q.ldx < q.N || q.ldw < q.N || (q.ldw % 4)) return false;
    if (((uintptr_t)q.dy & 15) || ((uintptr_t)q.x & 15) || ((uintptr_t)q.dw & 15)) return false;
  }
  return true;
}
// split-K factor and rows per slice of a group with `tiles` 256 x 256 tiles in all
void group_split(int tiles, int64_t K, int& split, int& kps) {
  const int cus = g_cu_budget.load(std::memory_order_relaxed);
  split = cus / tiles;
  const int max_split = (int)((K + 255) / 256);
  if (split > max_split) split = max_split;
  if (split < 1) split = 1;
  kps = (int)(((K + split - 1) / split + 63) / 64 * 64);
  split = (int)((K + kps - 1) / kps);
}
}  // namespace

namespace {
// Rounds of unsplit tiles + a split tail (wgrad_dma_big_kernel): taken when the group has more tiles than compute units
// (or more problems than the eight of the small form).
struct BigPlan { int tiles, full, tail, split, kps, pad; };
BigPlan big_plan(int32_t n, const mmt_wgrad_problem* pr, int64_t K) {
  BigPlan b{};
  for (int i = 0; i < n; ++i) b.tiles += (pr[i].M / 256) * (pr[i].N / 256);
  const int cus = g_cu_budget.load(std::memory_order_relaxed) & ~7;
  b.full = (b.tiles / cus) * cus;
  b.tail = b.tiles - b.full;
  b.split = 1; b.kps = (int)K;
  if (b.tail > 0) {
    int split = std::min(8, std::max(1, cus / b.tail));
    const int max_split = (int)((K + 255) / 256);
    if (split > max_split) split = max_split;
    b.kps = (int)(((K + split - 1) / split + 63) / 64 * 64);
    b.split = (int)((K + b.kps - 1) / b.kps);
  }
  b.pad = (b.tail * b.split + 7) & ~7;
  return b;
}
bool use_big(int32_t n, const mmt_wgrad_problem* pr) {
  int tiles = 0;
  for (int i = 0; i < n; ++i) tiles += (pr[i].M / 256) * (pr[i].N / 256);
  return n > 8 || tiles > (g_cu_budget.load(std::memory_order_relaxed) & ~7);
}
}  // namespace

extern "C" size_t mmt_wgrad_group_workspace_bytes(int32_t n, const mmt_wgrad_problem* problems, int64_t K) {
  if (!group_ok(n, problems, K)) return 0;
  if (use_big(n, problems)) {
    const BigPlan b = big_plan(n, problems, K);
    return b.split > 1 ? (size_t)b.tail * b.split * (65536 + 256) * sizeof(float) : 0;
  }
  int tiles = 0;
  size_t elems = 0;
  for (int i = 0; i < n; ++i) { tiles += (problems[i].M / 256) * (problems[i].N / 256); elems += (size_t)problems[i].M * problems[i].N + problems[i].M; }
  int split, kps;
  group_split(tiles, K, split, kps);
  return split > 1 ? (size_t)split * elems * sizeof(float) : 0;      // an unsplit K needs no slabs
}

extern "C" int mmt_wgrad_grouped(int32_t n, const mmt_wgrad_problem* problems, int64_t K, void* workspace,
                                 size_t workspace_bytes, void* stream) {
  if (!group_ok(n, problems, K))
    return mmt::fail(MMT_E_UNSUPPORTED, "mmt_wgrad_grouped: needs 1..28 problems with M %% 256 == 0, N %% 256 == 0, K %% 64 == 0, 16-byte aligned operands");
  const size_t need = mmt_wgrad_group_workspace_bytes(n, problems, K);
  if (need > 0 && (!workspace || workspace_bytes < need)) return mmt::fail(MMT_E_WORKSPACE, "mmt_wgrad_grouped: workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
  if (use_big(n, problems)) {
    const BigPlan b = big_plan(n, problems, K);
    mmt::WgradBig g;
    std::memset(&g, 0, sizeof(g));
    g.n = n; g.K = (int)K;
    int run = 0;
    for (int i = 0; i < mmt::kBigMax; ++i) {
      const mmt_wgrad_problem& q = problems[i < n ? i : n - 1];
      mmt::WgradProb& w = g.q[i];
      w.dy = (const __bf16*)q.dy; w.x = (const __bf16*)q.x; w.dw = q.dw; w.dbias = q.dbias;
      w.ldy = (int)q.ldy; w.ldx = (int)q.ldx; w.ldw = (int)q.ldw; w.M = q.M; w.N = q.N;
      if (i < n) run += (q.M / 256) * (q.N / 256);
      g.tiles_end[i] = run;
    }
    g.full_tiles = b.full; g.tail_tiles = b.tail; g.tail_pad = b.pad; g.tail_split = b.split; g.tail_kps = b.kps;
    g.tail_slabs = b.split > 1 ? (float*)workspace : nullptr;
    g.tail_bias = g.tail_slabs ? g.tail_slabs + (size_t)b.tail * b.split * 65536 : nullptr;
    hipStream_t st = (hipStream_t)stream;
    const int lds = 2 * mmt::kDmaStageBytes;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mmt::wgrad_dma_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(mmt::wgrad_dma_big_kernel, dim3(b.full + b.pad), dim3(512), lds, st, g);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && b.tail > 0 && b.split > 1) {
      hipLaunchKernelGGL(mmt::wgrad_tail_reduce_kernel, dim3(16 * b.tail), dim3(256), 0, st, g);
      e = hipGetLastError();
    }
    return e == hipSuccess ? MMT_OK : mmt::fail(MMT_E_LAUNCH, "mmt_wgrad_grouped: %s", hipGetErrorString(e));
  }
  mmt::WgradGroup g;
  g.n = n;
  int tiles = 0;
  for (int i = 0; i < n; ++i) tiles += (problems[i].M / 256) * (problems[i].N / 256);
  int split, kps;
  group_split(tiles, K, split, kps);
  float* ws = (float*)workspace;
  int run = 0;
  for (int i = 0; i < 8; ++i) {
    const mmt_wgrad_problem& q = problems[i < n ? i : n - 1];
    mmt::WgradParams& p = g.p[i];
    p.dy = (const __bf16*)q.dy; p.x = (const __bf16*)q.x; p.dw = q.dw;
    p.ldy = q.ldy; p.ldx = q.ldx; p.ldw = q.ldw; p.M = q.M; p.N = q.N; p.K = (int)K;
    p.tiles_n = q.N / 256; p.tiles_m = q.M / 256; p.k_per_split = kps;
    p.dbias = q.dbias;
    p.slabs = nullptr; p.bias_part = nullptr;
    if (i < n) {
      if (split > 1) {
        p.slabs = ws; ws += (size_t)split * q.M * q.N;
        if (q.dbias) { p.bias_part = ws; }
        ws += (size_t)split * q.M;
      }
      run += p.tiles_n * p.tiles_m;
    }
    g.tiles_end[i] = run;
  }
  g.tiles_total = tiles;
  hipStream_t st = (hipStream_t)stream;
  const int lds = 2 * mmt::kDmaStageBytes;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mmt::wgrad_dma_group_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(mmt::wgrad_dma_group_kernel, dim3(tiles * split), dim3(512), lds, st, g);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && split > 1) {         // the slabs of every problem -> dw, one launch, slice order (as wgrad_reduce_kernel)
    mmt::ReduceGroup r;
    r.n = n; r.split = split;
    int end = 0;
    for (int i = 0; i < 8; ++i) {
      const mmt::WgradParams& p = g.p[i < n ? i : n - 1];
      long blocks = ((long)p.M * p.N / 4 + 255) / 256;
      if (blocks > 1024) blocks = 1024;
      if (i < n) end += (int)blocks;
      r.dw[i] = p.dw; r.slabs[i] = p.slabs; r.ldw[i] = p.ldw; r.M[i] = p.M; r.N[i] = p.N;
      r.dbias[i] = p.bias_part ? p.dbias : nullptr; r.bias_part[i] = p.bias_part;
      r.blocks_end[i] = end;
    }
    hipLaunchKernelGGL(mmt::wgrad_reduce_group_kernel, dim3((unsigned)end), dim3(256), 0, st, r);
    e = hipGetLastError();
  }
  return e == hipSuccess ? MMT_OK : mmt::fail(MMT_E_LAUNCH, "mmt_wgrad_grouped: %s", hipGetErrorString(e));
}
