// Feed-forward GEMMs with the GELU fused into the epilogue (gfx950, bf16 in, fp32 accumulate):
//
//   mmt_ffn_gelu_gemm :  u = x . W1^T + b1,  g = gelu(u)                 (mmt_encoder.py:53-54 activation of the
//                        intermediate Dense of every TransformerEncoderBlock; u is kept for the backward)
//   mmt_ffn_dgelu_gemm:  du = (dy . W2) * gelu'(u)                       (the tape's product through the output
//                        Dense followed by the activation's derivative, src/tasks/pretraining.py:292-296)
//
// Without the fusion the activation is a pass of its own over the [B*S, 4H] intermediate (read u, write g:
// 200 MB per layer; read dg and u, write du: 300 MB) between two library GEMMs; hipBLASLt's own GELU_AUX / DGELU
// epilogues never return from the heuristic query in this ROCm build.  Here the accumulator tile is finished in
// registers and leaves the kernel once.
//
// Structure = the weight-gradient kernel's (wgrad_gemm.hip): 256 x 256 output tile, 8 waves of 64 x 128, 64
// contracted elements per step, operands global -> LDS by LDS-DMA into a double-buffered 64 KiB stage image,
// one barrier per step, XCD-aware tile order.  Operand A is always k-contiguous ([M, K] row-major): its 32-row
// x 128-byte tiles hold 64 k per row, 16-byte chunks XOR-swizzled by (row >> 1) & 7 so that the 16 lanes of one
// ds_read_b128 pass hit 16 different bank groups; the swizzle is applied on the DMA's source address.  Operand
// B is either n-contiguous ([K, N], the backward's W2: 32 k-rows x 64 columns per tile, column reads with
// ds_read_b64_tr_b16 as in the weight-gradient kernel, rows chosen so that the fragment's k order is the
// natural one A's b128 read delivers) or k-contiguous ([N, K], the forward's W1: same tile format as A).
// Epilogue: each wave parks 32 x 64 accumulator blocks in its slice of the stage just consumed, reads them
// back row-wise (8 columns per lane), applies bias / GELU / GELU' (packed fp32 math) against 16-byte coalesced
// loads of u, and writes 16-byte coalesced rows.  The kernel is persistent (one workgroup per CU of the budget).
#include "../../include/mmt_attn.h"
#include "../../include/mmt_layer.h"
#include "attn_tile.h"
#include "layer_common.h"
#include "mmt_err.h"

#include <atomic>

namespace mmt {

struct FfnGemmParams {
  const __bf16* a;     // [M, K] row stride lda
  const __bf16* b;     // BT == 0: [K, N] row stride ldb;  BT == 1: [N, K] row stride ldb
  const float* bias;   // [N] or NULL
  const __bf16* u_in;  // dgelu: pre-activation [M, N] row stride ldu
  __bf16* u_out;       // gelu: pre-activation out (may be NULL)
  __bf16* d;           // [M, N] row stride ldd
  long lda, ldb, ldu, ldd;
  int M, N, K, tiles_m, tiles_n, tiles_per_wg;
};

// Compute units the persistent grid may fill (mmt_ffn_set_cu_budget).
static std::atomic<int> g_ffn_cus{256};
static int cu_budget() { return g_ffn_cus.load(std::memory_order_relaxed); }

constexpr int kFfnStage = 64 * 1024;
enum { kEpiBias = 0, kEpiGelu = 1, kEpiDgelu = 2 };

__device__ __forceinline__ void ffn_wait_dma() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Persistent: workgroup w walks tiles_per_wg consecutive tiles of the row-major tile list as ONE stream of
// K/64-step main loops -- the DMA of the next tile's first stage is issued during the current tile's last
// step, so only the very first stage's latency is exposed, and a tile's output stores drain under the next
// tile's main loop.
template <int BT, int EPI>
__global__ __launch_bounds__(512, 2) void ffn_gemm_kernel(const FfnGemmParams p) {
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
  const int n_steps = p.K >> 6;

  // ---- DMA map: 64 wave-instructions of 1 KiB per stage; wave w issues 8w .. 8w+7 (waves 0-3: A, 4-7: B).
  //      A: tile (idx >> 2), row group idx & 3.  Pointers are for tile (0, 0), step 0.
  const bool is_a = wave < 4;
  const int drow = lane >> 3, dpos = lane & 7;
  const __bf16* gsrc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int idx = wave * 8 + j;
    const int rg = idx & 3, row = rg * 8 + drow;
    if (idx < 32 || BT == 1) {
      const int t8 = (idx >> 2) & 7;
      const int ch = dpos ^ ((row >> 1) & 7);
      gsrc[j] = (idx < 32 ? p.a + (long)(t8 * 32 + row) * p.lda : p.b + (long)(t8 * 32 + row) * p.ldb) + ch * 8;
    } else {
      const int i2 = idx - 32, kslab = i2 >> 4, t4 = (i2 >> 2) & 3;
      const int ch = (((dpos >> 2) ^ ((row >> 1) & 1)) << 2) | (dpos & 3);
      gsrc[j] = p.b + (long)(kslab * 32 + row) * p.ldb + t4 * 64 + ch * 8;
    }
  }
  const long sstep = is_a ? 64 : (BT == 1 ? 64 : 64 * p.ldb);
  const long tstep_m = is_a ? 256 * p.lda : 0, tstep_n = is_a ? 0 : (BT == 1 ? 256 * p.ldb : 256);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;
  auto dma = [&](unsigned stage, int tm, int tn, int step) {
    const long off = tm * tstep_m + tn * tstep_n + step * sstep;
#pragma unroll
    for (int j = 0; j < 8; ++j) glds16(gsrc[j] + off, stage + (wave * 8 + j) * 1024);
  };

  // ---- fragment offsets
  const int wm = wave & 3, wn = wave >> 2;
  const int arow = r * 128, ax = (r >> 1) & 7;
  const int frow = 8 * h + (li >> 2);
  int fo[2];
#pragma unroll
  for (int db = 0; db < 2; ++db) fo[db] = frow * 128 + ((db ^ ((frow >> 1) & 1)) << 6) + 32 * cb + 8 * (li & 3);
  const int erow = lane >> 3, ech = lane & 7;               // epilogue read-back: 8 rows per pass, 8 columns per lane

  int tm = t_begin / p.tiles_n, tn = t_begin - tm * p.tiles_n;
  dma(lds0, tm, tn, 0);
  ffn_wait_dma();
  __syncthreads();
  int gs = 0;                                               // steps done so far: stage parity

  for (int t = t_begin; t < t_end; ++t) {
    int tm_next = tm, tn_next = tn + 1;
    if (tn_next == p.tiles_n) { tn_next = 0; ++tm_next; }
    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x16{0};

    // dgelu: the tile's u is fetched 32 x 64 block by block, two blocks ahead of its use (the first two go out
    // before the last main-loop step), so that the epilogue does not sit out one HBM latency per block
    const int m0 = tm * 256, n0 = tn * 256;
    bf16x8 uq[3][4];
    auto load_u = [&](int blk, bf16x8 (&dst)[4]) {          // blk = 2 * hb + a
      const __bf16* up = p.u_in + (long)(m0 + wm * 64 + (blk & 1) * 32 + erow) * p.ldu + n0 + wn * 128 + (blk >> 1) * 64 + ech * 8;
#pragma unroll
      for (int it = 0; it < 4; ++it) dst[it] = *reinterpret_cast<const bf16x8*>(up + (long)it * 8 * p.ldu);
    };

    for (int step = 0; step < n_steps; ++step, ++gs) {
      const unsigned char* cur = smem + (gs & 1) * kFfnStage;
      const unsigned nxt = lds0 + ((gs + 1) & 1) * kFfnStage;
      if (step + 1 < n_steps) dma(nxt, tm, tn, step + 1);
      else if (t + 1 < t_end) dma(nxt, tm_next, tn_next, 0);
      if (EPI == kEpiDgelu && step == n_steps - 1) { load_u(0, uq[0]); load_u(1, uq[1]); }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 af[2], bfr[4];
        const int coff = ((2 * s + h) ^ ax) << 4;
#pragma unroll
        for (int a = 0; a < 2; ++a)
          af[a] = *reinterpret_cast<const bf16x8*>(cur + (wm * 2 + a) * 4096 + arow + coff);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          if (BT == 1) {
            bfr[b] = *reinterpret_cast<const bf16x8*>(cur + 32768 + (wn * 4 + b) * 4096 + arow + coff);
          } else {
            const unsigned char* base = cur + 32768 + ((s >> 1) * 4 + wn * 2 + (b >> 1)) * 4096 + fo[b & 1] + (s & 1) * 2048;
            const bf16x4 lo = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base)));
            const bf16x4 hi = __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 512)));
#pragma unroll
            for (int j = 0; j < 4; ++j) { bfr[b][j] = lo[j]; bfr[b][4 + j] = hi[j]; }
          }
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bfr[b], acc[a][b], 0, 0, 0);
      }
      ffn_wait_dma();
      __syncthreads();
    }

    // ---- epilogue: 32 x 64 accumulator blocks through a wave-private 8 KiB of the stage just consumed
    //      (the other stage already holds the next tile's first slab)
    float* park = reinterpret_cast<float*>(smem + ((gs - 1) & 1) * kFfnStage + wave * 8192);
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      const int hb = blk >> 1, a = blk & 1;
      const int ncol = n0 + wn * 128 + hb * 64 + ech * 8;
      const int mrow0 = m0 + wm * 64 + a * 32;
      float bs[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) bs[j] = 0.f;
      if (p.bias) load_param(p.bias + ncol, bs);
      if (EPI == kEpiDgelu && blk + 2 < 4) load_u(blk + 2, uq[(blk + 2) % 3]);
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
        for (int i = 0; i < 16; ++i) park[kap(i, h) * 64 + 32 * b2 + r] = acc[a][2 * hb + b2][i];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + erow;
        float v[8];
        Chunk<float>::load(park + row * 64 + ech * 8, v);
        const long orow = (long)(mrow0 + row);
        if (EPI == kEpiDgelu) {
          const bf16x8 uu = uq[blk % 3][it];
#pragma unroll
          for (int j = 0; j < 8; j += 2) {
            f32x2 dz;
            (void)gelu_tanh_x2(f32x2{(float)uu[j] + bs[j], (float)uu[j + 1] + bs[j + 1]}, dz);
            v[j] *= dz[0]; v[j + 1] *= dz[1];
          }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += bs[j];
          if (EPI == kEpiGelu) {
            if (p.u_out) Chunk<__bf16>::store(p.u_out + orow * p.ldu + ncol, v);   // rounds v: gelu of the stored value
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
              f32x2 dz;
              const f32x2 gz = gelu_tanh_x2(f32x2{v[j], v[j + 1]}, dz);
              v[j] = gz[0]; v[j + 1] = gz[1];
            }
          }
        }
        Chunk<__bf16>::store(p.d + orow * p.ldd + ncol, v);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    if (t + 1 < t_end) __syncthreads();       // the next tile's first step DMAs into the stage the parks used
    tm = tm_next; tn = tn_next;
  }
}

template <int BT, int EPI>
static hipError_t launch_ffn(const FfnGemmParams& p_in, hipStream_t st) {
  FfnGemmParams p = p_in;
  const int tiles = p.tiles_m * p.tiles_n, cus = cu_budget();
  p.tiles_per_wg = (tiles + cus - 1) / cus;
  const int grid = (tiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
  const int lds = 2 * kFfnStage;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_gemm_kernel<BT, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((ffn_gemm_kernel<BT, EPI>), dim3(grid), dim3(512), lds, st, p);
  return hipGetLastError();
}


static int ffn_check(const char* who, const void* a, const void* b, const void* d, int64_t M, int64_t N, int64_t K,
                     int64_t lda, int64_t ldb, int64_t ldb_min, int64_t ldd) {
  if (!a || !b || !d) return fail(MMT_E_INVALID, "%s: NULL argument", who);
  if (M <= 0 || N <= 0 || K <= 0 || (M % 256) || (N % 256) || (K % 64) || M > (1 << 30) || N > (1 << 30) || K > (1 << 30))
    return fail(MMT_E_UNSUPPORTED, "%s: needs M %% 256 == 0, N %% 256 == 0, K %% 64 == 0 (got %lld, %lld, %lld)", who,
                (long long)M, (long long)N, (long long)K);
  if ((lda % 8) || (ldb % 8) || (ldd % 8) || lda < K || ldb < ldb_min || ldd < N) return fail(MMT_E_INVALID, "%s: bad leading dimensions", who);
  if (((uintptr_t)a & 15) || ((uintptr_t)b & 15) || ((uintptr_t)d & 15)) return fail(MMT_E_INVALID, "%s: operands must be 16-byte aligned", who);
  return MMT_OK;
}

}  // namespace mmt

extern "C" void mmt_ffn_set_cu_budget(int32_t cus) {
  mmt::g_ffn_cus.store(cus < 32 ? 32 : (cus > 256 ? 256 : cus), std::memory_order_relaxed);
}

extern "C" int mmt_ffn_gelu_gemm(const void* x, int64_t ldx, const void* w, int64_t ldw, const float* bias, void* u,
                                 int64_t ldu, void* g, int64_t ldg, int64_t M, int64_t N, int64_t K, void* stream) {
  int rc = mmt::ffn_check("mmt_ffn_gelu_gemm", x, w, g, M, N, K, ldx, ldw, K, ldg);
  if (rc != MMT_OK) return rc;
  if (u && ((ldu % 8) || ldu < N || ((uintptr_t)u & 15))) return mmt::fail(MMT_E_INVALID, "mmt_ffn_gelu_gemm: bad u");
  if (bias && ((uintptr_t)bias & 15)) return mmt::fail(MMT_E_INVALID, "mmt_ffn_gelu_gemm: bias must be 16-byte aligned");
  mmt::FfnGemmParams p{};
  p.a = (const __bf16*)x; p.b = (const __bf16*)w; p.bias = bias; p.u_out = (__bf16*)u; p.d = (__bf16*)g;
  p.lda = ldx; p.ldb = ldw; p.ldu = ldu; p.ldd = ldg; p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.tiles_m = (int)(M / 256); p.tiles_n = (int)(N / 256);
  const hipError_t e = mmt::launch_ffn<1, mmt::kEpiGelu>(p, (hipStream_t)stream);
  return e == hipSuccess ? MMT_OK : mmt::fail(MMT_E_LAUNCH, "mmt_ffn_gelu_gemm: %s", hipGetErrorString(e));
}

extern "C" int mmt_ffn_dgelu_gemm(const void* dy, int64_t lddy, const void* w, int64_t ldw, const void* u, int64_t ldu,
                                  const float* bias, void* du, int64_t lddu, int64_t M, int64_t N, int64_t K, void* stream) {
  int rc = mmt::ffn_check("mmt_ffn_dgelu_gemm", dy, w, du, M, N, K, lddy, ldw, N, lddu);
  if (rc != MMT_OK) return rc;
  if (!u || (ldu % 8) || ldu < N || ((uintptr_t)u & 15)) return mmt::fail(MMT_E_INVALID, "mmt_ffn_dgelu_gemm: bad u");
  if (bias && ((uintptr_t)bias & 15)) return mmt::fail(MMT_E_INVALID, "mmt_ffn_dgelu_gemm: bias must be 16-byte aligned");
  mmt::FfnGemmParams p{};
  p.a = (const __bf16*)dy; p.b = (const __bf16*)w; p.bias = bias; p.u_in = (const __bf16*)u; p.d = (__bf16*)du;
  p.lda = lddy; p.ldb = ldw; p.ldu = ldu; p.ldd = lddu; p.M = (int)M; p.N = (int)N; p.K = (int)K;
  p.tiles_m = (int)(M / 256); p.tiles_n = (int)(N / 256);
  const hipError_t e = mmt::launch_ffn<0, mmt::kEpiDgelu>(p, (hipStream_t)stream);
  return e == hipSuccess ? MMT_OK : mmt::fail(MMT_E_LAUNCH, "mmt_ffn_dgelu_gemm: %s", hipGetErrorString(e));
}
