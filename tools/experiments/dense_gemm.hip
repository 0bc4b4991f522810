// Dense-layer GEMMs of the encoder stack with their bias / activation in the epilogue (gfx950, bf16 in, fp32 accumulate):
//
//   mmt_dense_fwd  :  y[M,N] = x[M,K] . W[N,K]^T + bias[N]   (+ g = gelu_tanh(y), y kept for the backward)
//   mmt_dense_dgrad:  dx[M,N] = dy[M,K] . W[K,N]              (the tape's product back through a Dense layer)
//
// = the projections of etcmodel's RelativeAttention / DenseLayers as instantiated at src/modeling/models/
// mmt_encoder.py:124-135 (QKV, attention output, intermediate, output) and `tape.gradient` through them
// (src/tasks/pretraining.py:292-296).  The library GEMMs they replace run the K = 768 products (QKV, attention output)
// at 0.7-0.9 PFLOP/s inside the train step: 256 x 256 tiles leave a quarter of the chip idle at N = 768 and N = 2304
// (192 / 576 tiles on 256 CUs), and a 12-step main loop is mostly prologue.
//
// Structure: the weight-gradient kernel's pipeline (wgrad_gemm.hip) for k-contiguous operands --
//   * output tile 256 x BN, BN = 192 | 256 chosen so that the tile count is a multiple of the CU count (N = 768,
//     2304: 192; N = 3072: 256); 8 waves as 4 (M) x 2 (N), 64 x BN/2 per wave = 2 x NB accumulators of 32 x 32;
//   * K advances in HALF-STAGES of 32: A [256 rows][64 B] + B [BN rows][64 B] (k-contiguous W) or [32 k][BN] (n-
//     contiguous W: column reads with ds_read_b64_tr_b16), 28-32 KiB, in a ring of FOUR LDS slots filled by LDS-DMA
//     (1 KiB pieces; the conflict-free 16-byte-slot swizzle on the source address): three half-stages are in flight
//     ahead of the one being consumed, each wave waits for its own pieces with a COUNTED vmcnt and there is one raw
//     barrier per half-stage (12-16 MFMAs per wave);
//   * persistent workgroups: one per CU, consecutive tiles of the row-major tile list as ONE stream of half-stages --
//     the DMA of the next tile's first three half-stages runs under the current tile's last steps and its epilogue,
//     whose stores in turn drain under the next tile's main loop (the counted waits skip over them);
//   * epilogue through a wave-private 4 KiB park (32 x 32 accumulator block -> row-wise 16-byte stores), bias and
//     tanh-GELU (packed fp32 math) applied on the way out.
#include "../../include/mmt_attn.h"      // (paths from tools/experiments/: build with -I multimodal-long-transformer-2021_amd/csrc)
#include "../../include/mmt_layer.h"
#include "attn_tile.h"
#include "layer_common.h"
#include "mmt_err.h"
#include <cstdarg>
#include <cstdio>

namespace mmt {
int fail(int code, const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); fputc('\n', stderr); va_end(ap); return code; }      // stand-alone build

struct DenseParams {
  const __bf16* a;     // [M, K] row stride lda (k-contiguous)
  const __bf16* b;     // BT == 1: [N, K] row stride ldb;  BT == 0: [K, N] row stride ldb
  const float* bias;   // [N] or NULL
  __bf16* y;           // [M, N] row stride ldy: a.b (+ bias); with GELU: the pre-activation (may be NULL)
  __bf16* g;           // GELU epilogue: gelu(y) [M, N] row stride ldg
  long lda, ldb, ldy, ldg;
  int M, N, K, tiles_m, tiles_n, tiles_per_wg;
};

constexpr int kDenseSlots = 4;
enum { kDenseBias = 0, kDenseGelu = 1 };

template <int NB> struct DenseCfg {
  static constexpr int kBN = 64 * NB;                        // 192 | 256
  static constexpr int kABytes = 256 * 64;                   // one half-stage of A: 256 rows x 32 k
  static constexpr int kBBytes = kBN * 64;
  static constexpr int kSlot = kABytes + kBBytes;            // 28,672 | 32,768
  static constexpr int kPieces = kSlot / 1024;               // 28 | 32: piece i belongs to wave i & 7
  static constexpr int kPark = kDenseSlots * kSlot;          // wave-private 4 KiB parks behind the ring
  static constexpr int kLds = kPark + 8 * 4096;              // 147,456 | 163,840
};

template <int N> __device__ __forceinline__ void dense_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int NB, int BT, int EPI>
__global__ __launch_bounds__(512, 2) void dense_gemm_kernel(const DenseParams p) {
  using C = DenseCfg<NB>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, li = lane & 15, cb = (lane >> 4) & 1, r = lane & 31;
  // XCD-aware order (block b runs on XCD b % 8): XCD x takes the x-th contiguous range of workgroups, so the
  // workgroups sharing an L2 work on neighbouring row blocks of A and sweep B together.
  int t_begin, t_end;
  {
    const int nwg = gridDim.x, b = blockIdx.x, x = b & 7;
    const int base = nwg >> 3, rem = nwg & 7;
    const int L = x * base + min(x, rem) + (b >> 3);
    t_begin = L * p.tiles_per_wg;
    t_end = min(p.tiles_m * p.tiles_n, t_begin + p.tiles_per_wg);
  }
  if (t_begin >= t_end) return;
  const int HS = p.K >> 5;                                   // half-stages per tile
  const int G = (t_end - t_begin) * HS;                      // ... of this workgroup's whole stream

  // ---- DMA map.  Piece i (1 KiB = one wave-instruction) of a half-stage: i < 16: A rows 16 i .. 16 i + 15 (64 B each,
  //      lane l -> row l >> 2, 16-byte slot l & 3 holding source chunk slot ^ ((row >> 2) & 3)); i >= 16: B, the same
  //      shape for a k-contiguous W, or 8 k-rows x 128 B of a 32 x 64-column tile for an n-contiguous one (the tile
  //      image of the weight-gradient kernel: 64-byte halves swapped on odd row pairs).  Wave w issues pieces w, w + 8, ...
  const int npc = (C::kPieces - wave + 7) >> 3;              // 4, or 3 for waves 4..7 of the 192-wide tile
  const __bf16* gsrc[4];
  long kstep[4], tm_step[4], tn_step[4];
  int ldst[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = wave + 8 * j;
    gsrc[j] = p.a; kstep[j] = 0; tm_step[j] = 0; tn_step[j] = 0; ldst[j] = 0;
    if (i >= C::kPieces) continue;
    ldst[j] = i * 1024;
    if (i < 16 || BT == 1) {
      const int row = (i & 15) * 16 + (lane >> 2), slot = lane & 3;       // (B: i - 16 < kBN / 16 <= 16 pieces)
      const int ch = slot ^ ((row >> 2) & 3);
      if (i < 16) { gsrc[j] = p.a + (long)row * p.lda + ch * 8; kstep[j] = 32; tm_step[j] = 256 * p.lda; }
      else { gsrc[j] = p.b + (long)row * p.ldb + ch * 8; kstep[j] = 32; tn_step[j] = (long)C::kBN * p.ldb; }
    } else {
      const int pb = i - 16, rg = pb & 3, ct = pb >> 2;
      const int drow = lane >> 3, dpos = lane & 7, row = rg * 8 + drow;
      const int ch = (((dpos >> 2) ^ ((row >> 1) & 1)) << 2) | (dpos & 3);
      gsrc[j] = p.b + (long)row * p.ldb + ct * 64 + ch * 8;
      kstep[j] = 32 * p.ldb; tn_step[j] = C::kBN;
      ldst[j] = 16 * 1024 + ct * 4096 + rg * 1024;
    }
  }
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  const int tiles_n = p.tiles_n;
  // pieces [2 part, 2 part + 1] of the NEXT stream half-stage to fetch into ring slot dg & 3.  The source pointers
  // are RUNNING pointers (one 64-bit add per piece and half-stage; a jump at tile ends): computed from (tile, step)
  // each time they cost ~20 scalar instructions per piece -- 400 issue cycles per half-stage and wave, beside 16 MFMAs.
  int dg = 0, dhs = 0, dtn = t_begin - (t_begin / tiles_n) * tiles_n;
  long jump_n[4], jump_m[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    gsrc[j] += (long)(t_begin / tiles_n) * tm_step[j] + (long)dtn * tn_step[j];
    jump_n[j] = tn_step[j] - (long)HS * kstep[j];                                  // next tile of the same row block
    jump_m[j] = tm_step[j] - (long)(tiles_n - 1) * tn_step[j] - (long)HS * kstep[j];   // first tile of the next row block
  }
  auto dma = [&](int part) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if ((j >> 1) != part || j >= npc) continue;
      glds16(gsrc[j], lds0 + (dg & 3) * C::kSlot + ldst[j]);
    }
    if (part == 1) {
      ++dg;
      if (++dhs == HS) {
        dhs = 0;
        const bool row_end = ++dtn == tiles_n;
        if (row_end) dtn = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) gsrc[j] += kstep[j] + (row_end ? jump_m[j] : jump_n[j]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) gsrc[j] += kstep[j];
      }
    }
  };

  // ---- fragment offsets
  const int wm = wave & 3, wn = wave >> 2;
  // k-contiguous operand: row (32 i + r) of the region, 16-byte slot (2 s + h) ^ ((r >> 2) & 3)
  const int krow = r * 64, kx = (r >> 2) & 3;
  // n-contiguous B: 4-row transposed reads, rows 8 h + (li >> 2) (+ 4 for the upper half, + 16 s)
  const int frow = 8 * h + (li >> 2);
  int fo[2];
#pragma unroll
  for (int db = 0; db < 2; ++db) fo[db] = frow * 128 + ((db ^ ((frow >> 1) & 1)) << 6) + 32 * cb + 8 * (li & 3);
  float* park = reinterpret_cast<float*>(smem + C::kPark + wave * 4096);

  dma(0); dma(1);
  if (G > 1) { dma(0); dma(1); }
  if (G > 2) { dma(0); dma(1); }

  f32x16 acc[2][NB];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[a][b] = f32x16{0};
  constexpr int kStores = 2 * NB * 2;      // store instructions of one epilogue per wave (the GELU form with y has twice as many: then the wait below is merely stricter)

  bf16x8 af0[2], bf0[NB], af1[2], bf1[NB];
  auto read_frags = [&](const unsigned char* cur, int s, bf16x8 (&af)[2], bf16x8 (&bfr)[NB]) {
    const int coff = ((2 * s + h) ^ kx) << 4;
#pragma unroll
    for (int a = 0; a < 2; ++a)
      af[a] = *reinterpret_cast<const bf16x8*>(cur + (wm * 64 + a * 32) * 64 + krow + coff);
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      if (BT == 1) {
        bfr[b] = *reinterpret_cast<const bf16x8*>(cur + C::kABytes + (wn * NB * 32 + b * 32) * 64 + krow + coff);
      } else {
        const int ci = wn * NB + b;                        // 32-column block of the tile
        const unsigned char* base = cur + C::kABytes + (ci >> 1) * 4096 + fo[ci & 1] + s * 2048;
        const bf16x4 lo = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base)));
        const bf16x4 hi = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 512)));
#pragma unroll
        for (int j = 0; j < 4; ++j) { bfr[b][j] = lo[j]; bfr[b][4 + j] = hi[j]; }
      }
    }
  };
  auto mma_block = [&](const bf16x8 (&af)[2], const bf16x8 (&bfr)[NB]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b)
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  int hs = 0, t = t_begin;
  int since_epi = 1 << 20;                                   // half-stages consumed since this wave's last epilogue
  for (int g = 0; g < G; ++g) {
    // Wait for THIS half-stage's pieces.  Vector-memory operations retire in issue order, so everything this wave
    // issued after them may stay in flight: the pieces of the (at most two) later half-stages and, for the first
    // three half-stages after an epilogue, that epilogue's stores (issued between the pieces of g + 2 and g + 3).
    const int behind = G - 1 - g;
    if (behind >= 2) {
      if (since_epi < 3) { if (npc == 4) dense_vmcnt<8 + kStores>(); else dense_vmcnt<6 + kStores>(); }
      else if (npc == 4) dense_vmcnt<8>(); else dense_vmcnt<6>();
    } else if (behind == 1) {
      if (npc == 4) dense_vmcnt<4>(); else dense_vmcnt<3>();
    } else dense_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's LDS reads of the previous half-stage are back
    __builtin_amdgcn_s_barrier();
    ++since_epi;
    const unsigned char* cur = smem + (g & 3) * C::kSlot;
    const bool more = g + 3 < G;
    // Software pipeline over the k16 sub-steps: the fragments of sub-step u + 1 are requested BEFORE the MFMAs of
    // sub-step u are issued -- across the barrier too (the second sub-step of half-stage g - 1 is contracted after
    // this half-stage's barrier and first reads), so that the matrix pipe has 8 MFMAs to chew on while the barrier
    // skew and the LDS latency of a fresh half-stage pass.
    if (more) dma(0);                                        // half-stage g + 3 into the slot of g - 1: every wave has left it
    read_frags(cur, 0, af0, bf0);
    if (hs > 0) mma_block(af1, bf1);                         // (g - 1).s1
    if (more) dma(1);
    read_frags(cur, 1, af1, bf1);
    mma_block(af0, bf0);                                     // g.s0
    if (++hs < HS) continue;
    mma_block(af1, bf1);                                     // the tile's last sub-step: nothing left to hide behind

    // ---- epilogue of tile t: 32 x 32 accumulator blocks through this wave's park, bias / GELU on the way out
    {
      const int tm = t / tiles_n, tn = t - tm * tiles_n;
      const int m0 = tm * 256 + wm * 64, n0 = tn * C::kBN + wn * (32 * NB);
      const int erow = lane >> 2, ec = (lane & 3) * 8;       // read-back: 16 rows per pass, 8 columns per lane
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          float bs[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) bs[j] = 0.f;
          const int ncol = n0 + 32 * b + ec;
          if (p.bias) load_param(p.bias + ncol, bs);
#pragma unroll
          for (int i = 0; i < 16; ++i) park[kap(i, h) * 32 + r] = acc[a][b][i];
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int ps = 0; ps < 2; ++ps) {
            const int row = ps * 16 + erow;
            float v[8];
            Chunk<float>::load(park + row * 32 + ec, v);
            const long orow = (long)(m0 + 32 * a + row);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += bs[j];
            if (EPI == kDenseGelu) {
              if (p.y) Chunk<__bf16>::store(p.y + orow * p.ldy + ncol, v);     // rounds v: gelu of the stored value
#pragma unroll
              for (int j = 0; j < 8; j += 2) {
                f32x2 dz;
                const f32x2 gz = gelu_tanh_x2(f32x2{v[j], v[j + 1]}, dz);
                v[j] = gz[0]; v[j + 1] = gz[1];
              }
              Chunk<__bf16>::store(p.g + orow * p.ldg + ncol, v);
            } else {
              Chunk<__bf16>::store(p.y + orow * p.ldy + ncol, v);
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          acc[a][b] = f32x16{0};
        }
    }
    hs = 0; ++t; since_epi = 0;
  }
}

template <int NB, int BT, int EPI>
static hipError_t launch_dense(const DenseParams& p_in, int cus, hipStream_t st) {
  DenseParams p = p_in;
  using C = DenseCfg<NB>;
  p.tiles_m = p.M / 256; p.tiles_n = p.N / C::kBN;
  const int tiles = p.tiles_m * p.tiles_n;
  p.tiles_per_wg = (tiles + cus - 1) / cus;
  const int grid = (tiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dense_gemm_kernel<NB, BT, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, C::kLds);
  hipLaunchKernelGGL((dense_gemm_kernel<NB, BT, EPI>), dim3(grid), dim3(512), C::kLds, st, p);
  return hipGetLastError();
}

// 192-wide tiles when they fill the chip better: fewer idle CU-rounds for the tile count
static int dense_pick_nb(long M, long N, int cus) {
  const long tm = M / 256;
  if (N % 192) return 4;
  if (N % 256) return 3;
  auto rounds_waste = [&](long tiles) { const long rounds = (tiles + cus - 1) / cus; return (double)(rounds * cus) / (double)tiles; };
  const double w3 = rounds_waste(tm * (N / 192)), w4 = rounds_waste(tm * (N / 256));
  return w3 < w4 - 1e-9 ? 3 : 4;
}

static int dense_check(const char* who, const void* a, const void* b, const void* d, int64_t M, int64_t N, int64_t K,
                       int64_t lda, int64_t ldb, int64_t ldb_min, int64_t ldd) {
  if (!a || !b || !d) return fail(MMT_E_INVALID, "%s: NULL argument", who);
  if (M <= 0 || N <= 0 || K < 128 || (M % 256) || ((N % 256) && (N % 192)) || (K % 32) || M > (1 << 30) || N > (1 << 30) || K > (1 << 30))
    return fail(MMT_E_UNSUPPORTED, "%s: needs M %% 256 == 0, N %% 192 == 0 or N %% 256 == 0, K %% 32 == 0, K >= 128 (got %lld, %lld, %lld)", who,
                (long long)M, (long long)N, (long long)K);
  if ((lda % 8) || (ldb % 8) || (ldd % 8) || lda < K || ldb < ldb_min || ldd < N) return fail(MMT_E_INVALID, "%s: bad leading dimensions", who);
  if (((uintptr_t)a & 15) || ((uintptr_t)b & 15) || ((uintptr_t)d & 15)) return fail(MMT_E_INVALID, "%s: operands must be 16-byte aligned", who);
  return MMT_OK;
}

}  // namespace mmt

extern "C" int mmt_dense_fwd(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias, void* y,
                             int64_t ldy, void* g, int64_t ldg, int64_t M, int64_t N, int64_t K, int32_t cu_budget,
                             void* stream) {
  int rc = mmt::dense_check("mmt_dense_fwd", x, w, g ? g : y, M, N, K, ldx, ldw, K, g ? ldg : ldy);
  if (rc != MMT_OK) return rc;
  if (y && ((ldy % 8) || ldy < N || ((uintptr_t)y & 15))) return mmt::fail(MMT_E_INVALID, "mmt_dense_fwd: bad y");
  if (bias && ((uintptr_t)bias & 15)) return mmt::fail(MMT_E_INVALID, "mmt_dense_fwd: bias must be 16-byte aligned");
  const int cus = cu_budget < 32 ? (cu_budget <= 0 ? 256 : 32) : (cu_budget > 256 ? 256 : cu_budget);
  mmt::DenseParams p{};
  p.a = (const __bf16*)x; p.b = (const __bf16*)w; p.bias = bias; p.y = (__bf16*)y; p.g = (__bf16*)g;
  p.lda = ldx; p.ldb = ldw; p.ldy = ldy; p.ldg = ldg; p.M = (int)M; p.N = (int)N; p.K = (int)K;
  const int nb = mmt::dense_pick_nb(M, N, cus);
  hipError_t e;
  if (g) e = nb == 3 ? mmt::launch_dense<3, 1, mmt::kDenseGelu>(p, cus, (hipStream_t)stream) : mmt::launch_dense<4, 1, mmt::kDenseGelu>(p, cus, (hipStream_t)stream);
  else e = nb == 3 ? mmt::launch_dense<3, 1, mmt::kDenseBias>(p, cus, (hipStream_t)stream) : mmt::launch_dense<4, 1, mmt::kDenseBias>(p, cus, (hipStream_t)stream);
  return e == hipSuccess ? MMT_OK : mmt::fail(MMT_E_LAUNCH, "mmt_dense_fwd: %s", hipGetErrorString(e));
}

extern "C" int mmt_dense_dgrad(const void* dy, int64_t lddy, const void* w, int64_t ldw, void* dx, int64_t lddx, int64_t M,
                               int64_t N, int64_t K, int32_t cu_budget, void* stream) {
  int rc = mmt::dense_check("mmt_dense_dgrad", dy, w, dx, M, N, K, lddy, ldw, N, lddx);
  if (rc != MMT_OK) return rc;
  const int cus = cu_budget < 32 ? (cu_budget <= 0 ? 256 : 32) : (cu_budget > 256 ? 256 : cu_budget);
  mmt::DenseParams p{};
  p.a = (const __bf16*)dy; p.b = (const __bf16*)w; p.y = (__bf16*)dx;
  p.lda = lddy; p.ldb = ldw; p.ldy = lddx; p.M = (int)M; p.N = (int)N; p.K = (int)K;
  const int nb = mmt::dense_pick_nb(M, N, cus);
  const hipError_t e = nb == 3 ? mmt::launch_dense<3, 0, mmt::kDenseBias>(p, cus, (hipStream_t)stream)
                               : mmt::launch_dense<4, 0, mmt::kDenseBias>(p, cus, (hipStream_t)stream);
  return e == hipSuccess ? MMT_OK : mmt::fail(MMT_E_LAUNCH, "mmt_dense_dgrad: %s", hipGetErrorString(e));
}
