// Shared device-side definitions for the gfx950 relative-attention kernels.
//
// Fragment conventions used throughout (CDNA4 32x32 MFMA, wave64):
//   lane l -> r = l & 31 (row/col owned by the lane), h = l >> 5 (half).
//   A 32x32 accumulator register i of lane (r,h) is element
//       [row = kap(i,h)][col = r],   kap(i,h) = (i & 3) + 8 * (i >> 2) + 4 * h.
//   Scores are computed "swapped" (S^T = K . Q^T) so that col = query row: every lane
//   owns ONE query row and its 16 registers walk 16 keys; the softmax state (m, l) and the
//   O^T accumulator (col = query row as well) are then lane-local.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mmt {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

__device__ __forceinline__ int kap(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

// ---------------------------------------------------------------------------------------
// Pattern + relative-id descriptor, device copy of mmt_mask_desc with derived constants.
// ---------------------------------------------------------------------------------------
struct PatternDev {
  int radius;      // local radius W (>= S: no band)
  int g0, ng;      // global tokens [g0, g0+ng)
  int id_mode;     // 0 none, 1 = 1-D, 2 = 2-D
  int m;           // relative_pos_max_distance
  int P, r;        // patches per row, core layers (2-D)
  int I;           // P*P image positions (2-D), 0 otherwise
  unsigned magicP; // floor(2^32 / P) + 1: x / P == umulhi(x, magicP) for x * P < 2^32 (no integer-division sequence per element)
  int image_part;  // P*P + 8 + 2m + 1   (feature_utils.py:78-79)
  int text_part;   // image_part + 1     (feature_utils.py:82)
};

// 1-D clipped id (etcmodel RelativePositionGenerator).
__device__ __forceinline__ int id_1d(int q, int k, int m) {
  int d = k - q;
  int a = min(abs(d), m);
  return d >= 0 ? a : m + a;
}

// 2-D id: value of the reference's base tensor at [P + dx, P + dy]
// (feature_utils.py:89-112, 164-170) in closed form.
__device__ __forceinline__ int id_2d(int dx, int dy, int r) {
  const int d = 2 * r + 1;
  const int vert = dx < -r ? 0 : (dx > r ? 2 : 1);
  const int horz = dy < -r ? 0 : (dy > r ? 2 : 1);
  if (vert == 1 && horz == 1) {
    int c = dx * d + dy;
    return c < 0 ? c + d * d : c;
  }
  // {top, top_right, right, right_bottom, bottom, bottom_left, left, top_left} = d*d + 0..7
  // packed as nibbles indexed by vert*3+horz: (0,0)=7 (0,1)=0 (0,2)=1 (1,0)=6 (1,1)=x (1,2)=2
  // (2,0)=5 (2,1)=4 (2,2)=3
  const unsigned long long lut = 0x345206107ull;
  return d * d + (int)((lut >> (4 * (vert * 3 + horz))) & 0xF);
}

__device__ __forceinline__ int rel_id(const PatternDev& p, int q, int k) {
  if (p.id_mode == 1) return id_1d(q, k, p.m);
  // id_mode == 2  (feature_utils.py:172-184)
  const bool qi = q < p.I, ki = k < p.I;
  if (qi && ki) {
    const int xq = (int)__umulhi((unsigned)q, p.magicP), yq = q - xq * p.P;
    const int xk = (int)__umulhi((unsigned)k, p.magicP), yk = k - xk * p.P;
    return id_2d(xk - xq, yk - yq, p.r);
  }
  if (qi) return p.text_part;
  if (ki) return p.image_part;
  return id_1d(q, k, p.m);
}

__device__ __forceinline__ bool is_global(const PatternDev& p, int x) {
  return (unsigned)(x - p.g0) < (unsigned)p.ng;
}

// mask(q,k) = segmented(q,k) && (|q-k| <= W || global(q) || global(k))   
__device__ __forceinline__ bool pattern_mask(const PatternDev& p, int valid_len, int q, int k) {
  const bool seg = (q < valid_len) == (k < valid_len);
  const bool near = abs(q - k) <= p.radius;
  return seg && (near || is_global(p, q) || is_global(p, k));
}

// ---------------------------------------------------------------------------------------
// Dropout keep decision shared by forward and backward (and restated on the CPU in the
// tests): one 32-bit mix per (b, n, q, k).  keep iff hash >= threshold.
// ---------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// Attention-probability dropout: 16 random bits per (b*N+n, q, k); one hash serves the key pair
// (k & ~1, k | 1); keep iff bits >= thresh16 (restated in oracle/attention.py: dropout_keep_mask).
// The row base is a full mix32; the per-pair finalizer is ONE multiply round (x ^= x>>16, x = lo24(x) * M24, x ^= x>>15):
// the pair term is already a product with a large odd constant, and the keep-mask statistics (keep rate, row /
// column spread, neighbour correlations < 0.002: tests/test_oracle_attention.py) match the two-round mixer's.  The
// hash is the biggest single item of a one-id tile's VALU work; the multiply is the 24-bit one (v_mul_u32_u24, full
// rate -- a 32-bit v_mul_lo_u32 runs at a quarter of it): the fold before it has already carried bits 24..31 of x into
// bits 8..15, so the product loses nothing the 32-bit one mixed in.
constexpr uint32_t kDropPairMul = 0xC2B2AE35u;
// the dropout seed a kernel works with: the descriptor's, plus the device-resident epoch when one is set
// (the descriptor's dropout_epoch; one scalar load per wave)
struct SeedPair { uint32_t lo, hi; };
__device__ __forceinline__ SeedPair effective_seed(uint32_t lo, uint32_t hi, const unsigned long long* epoch) {
  if (epoch) {
    const unsigned long long s = (((unsigned long long)hi << 32) | lo) + *epoch;
    lo = (uint32_t)s; hi = (uint32_t)(s >> 32);
  }
  return SeedPair{lo, hi};
}
__host__ __device__ __forceinline__ uint32_t drop_row_base(uint32_t seed_lo, uint32_t seed_hi,
                                                           uint32_t bn, uint32_t q) {
  return mix32(seed_lo ^ (bn * 0x9E3779B9u)) + seed_hi + q * 0x85EBCA6Bu;
}
// `pair_term` = (k >> 1) * kDropPairMul (the kernels build it with one multiply per tile plus constants)
__host__ __device__ __forceinline__ uint32_t drop_pair_finish(uint32_t row_base, uint32_t pair_term) {
  uint32_t x = row_base ^ pair_term;
  x ^= x >> 16;
#if defined(__HIP_DEVICE_COMPILE__)
  x = __umul24(x, 0xeb352du);
#else
  x = (x & 0xFFFFFFu) * 0xeb352du;
#endif
  x ^= x >> 15;
  return x;
}
__host__ __device__ __forceinline__ uint32_t drop_pair_hash(uint32_t row_base, uint32_t k) {
  return drop_pair_finish(row_base, (k >> 1) * kDropPairMul);
}
__host__ __device__ __forceinline__ uint32_t drop_bits16(uint32_t row_base, uint32_t k) {
  const uint32_t hsh = drop_pair_hash(row_base, k);
  return (k & 1) ? (hsh >> 16) : (hsh & 0xFFFFu);
}

// XCD-aware remap of a 1-D grid: consecutive logical ids land on the same XCD (blocks are
// dealt round-robin over the 8 XCDs).  Speed only; identity when the grid is not a multiple
// of 8.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  if (nwg & 7) return bid;
  const int cpx = nwg >> 3;
  return (bid & 7) * cpx + (bid >> 3);
}

// One global_load_lds_dwordx4: lane l's 16 bytes at `gsrc` -> LDS byte address lds_dst + 16 l (M0 = wave-
// uniform destination, saved and restored).  Issued as asm so that hipcc does not order every later LDS
// read behind it with a vmcnt(0); completion is counted by hand (vmcnt(0) before the step's barrier).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

}  // namespace mmt
