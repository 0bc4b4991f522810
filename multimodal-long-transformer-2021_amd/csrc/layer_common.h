// Helpers shared by the row-wise fused kernels (fused_layer.hip, embed.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "mmt_common.h"

namespace mmt {

// ---- 8-element chunk I/O --------------------------------------------------------------------
template <typename T> struct Chunk;
template <> struct Chunk<__bf16> {
  static __device__ __forceinline__ void load(const __bf16* p, float (&v)[8]) {
    const bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
  }
  static __device__ __forceinline__ void store(__bf16* p, float (&v)[8]) {   // rounds v in place
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) { t[i] = (__bf16)v[i]; v[i] = (float)t[i]; }
    *reinterpret_cast<bf16x8*>(p) = t;
  }
};
template <> struct Chunk<float> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
  }
  static __device__ __forceinline__ void store(float* p, float (&v)[8]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
  }
};
// ---- W-element chunks (W = 4 | 8) for the row kernels: W is chosen per row length so that the
// 64 lanes tile the row without idle lanes (H = 768 -> 3 chunks of 4 per lane). ----------------------
template <typename T, int W> struct ChunkW;
template <int W> struct ChunkW<__bf16, W> {
  typedef __attribute__((ext_vector_type(W))) __bf16 vec;
  static __device__ __forceinline__ void load(const __bf16* p, float (&v)[W]) {
    const vec t = *reinterpret_cast<const vec*>(p);
#pragma unroll
    for (int i = 0; i < W; ++i) v[i] = (float)t[i];
  }
  static __device__ __forceinline__ void store(__bf16* p, const float (&v)[W]) {
    vec t;
#pragma unroll
    for (int i = 0; i < W; ++i) t[i] = (__bf16)v[i];
    *reinterpret_cast<vec*>(p) = t;
  }
};
template <int W> struct ChunkW<float, W> {
  static __device__ __forceinline__ void load(const float* p, float (&v)[W]) {
#pragma unroll
    for (int q = 0; q < W; q += 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(p + q);
      v[q] = a[0]; v[q + 1] = a[1]; v[q + 2] = a[2]; v[q + 3] = a[3];
    }
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[W]) {
#pragma unroll
    for (int q = 0; q < W; q += 4) *reinterpret_cast<f32x4*>(p + q) = f32x4{v[q], v[q + 1], v[q + 2], v[q + 3]};
  }
};
// A chunk held as loaded (bf16: W/2 registers instead of W floats) so that a row can stay resident
// across a reduction at half the register cost.
template <typename T, int W> struct RawW;
template <int W> struct RawW<__bf16, W> {
  typedef __attribute__((ext_vector_type(W))) __bf16 vec;
  vec v;
  __device__ __forceinline__ void load(const __bf16* p) { v = *reinterpret_cast<const vec*>(p); }
  __device__ __forceinline__ void zero() { v = vec{0}; }
  __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
  // compiler barrier: forget what is known about the converted values, so that they are converted
  // again after it instead of being kept live as W floats
  __device__ __forceinline__ void opaque() { asm volatile("" : "+v"(v)); }
};
template <int W> struct RawW<float, W> {
  float v[W];
  __device__ __forceinline__ void load(const float* p) { ChunkW<float, W>::load(p, v); }
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < W; ++i) v[i] = 0.f;
  }
  __device__ __forceinline__ float get(int i) const { return v[i]; }
  __device__ __forceinline__ void opaque() {}
};
template <int W> __device__ __forceinline__ void load_param_w(const float* p, float (&v)[W]) { ChunkW<float, W>::load(p, v); }

__device__ __forceinline__ void load_param(const float* p, float (&v)[8]) { Chunk<float>::load(p, v); }

// Sum over the 64 lanes, the same value in every lane.  Butterfly order (lane ^ 1, ^ 2, ^ 4, ^ 8, ^ 16, ^ 32) as DPP adds
// and gfx950's lane-row swaps: no LDS round trip (six ds_bpermute, ~100 cycles each in a dependent chain, until round 4).
__device__ __forceinline__ float wave_sum(float x) {
#define MMT_DPP_ADD(ctrl) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, true))
  MMT_DPP_ADD(0xB1);     // quad_perm [1,0,3,2]
  MMT_DPP_ADD(0x4E);     // quad_perm [2,3,0,1]
  MMT_DPP_ADD(0x141);    // row_half_mirror (quads are uniform by now: lane ^ 4)
  MMT_DPP_ADD(0x140);    // row_mirror (groups of 8 are uniform: lane ^ 8)
#undef MMT_DPP_ADD
  // (inline asm with two distinct registers: given the same value for both operands, hipcc's builtin folds the two
  //  results into one -- `v_add_f32 v, v, v` after the swap)
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));       // rows (0,1) and (2,3) exchanged
  x = a + b;
  a = x; b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));       // halves exchanged
  return a + b;
}

// Dropout of the row-wise kernels: 16 random bits per element from the attention kernels' hash (mmt_common.h:
// drop_row_base / drop_pair_finish) on (row, column) -- per row one scalar mix, per element PAIR one xor, two
// shift-xors and one 24-bit multiply (all full rate); the pair terms (column >> 1) * kDropPairMul do not depend on
// the row.  (Until round 4: two 32-bit mixes of the flat 64-bit element index per pair -- 62 quarter-rate
// multiplies per row and lane in the residual kernel.)  Restated in oracle/layer_ops.py.
__host__ __device__ __forceinline__ uint32_t layer_drop_row(uint32_t seed_lo, uint32_t seed_hi, long row) {
  return drop_row_base(seed_lo, seed_hi, (uint32_t)((uint64_t)row >> 32), (uint32_t)row);
}
__host__ __device__ __forceinline__ uint32_t layer_drop_bits16(uint32_t row_base, uint32_t col) {
  return drop_bits16(row_base, col);
}

// GELU, tanh approximation (mmt_encoder.py:53-54): returns gelu(z), dz = gelu'(z).
__device__ __forceinline__ float gelu_tanh(float z, float& dz) {
  const float k = 0.7978845608028654f, c = 0.044715f;
  const float z2 = z * z;
  const float inner = k * z * fmaf(c, z2, 1.f);
  // tanh(a) = 1 - 2 / (1 + 2^(2a*log2e)): v_exp_f32 + v_rcp_f32, saturates cleanly at +-1
  const float e = __builtin_amdgcn_exp2f(inner * (2.f * kLog2e));
  const float t = 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + e);
  dz = 0.5f * (1.f + t) + 0.5f * z * (1.f - t * t) * k * fmaf(3.f * c, z2, 1.f);
  return 0.5f * z * (1.f + t);
}

// The same function on a pair of values with packed fp32 arithmetic (v_pk_mul/fma_f32), written in terms of
// s = 1 / (1 + 2^(2 a log2e)) = (1 - tanh a) / 2:   gelu = z (1 - s),   gelu' = (1 - s)(1 + 2 k z s (1 + 3 c z^2)).
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 gelu_tanh_x2(f32x2 z, f32x2& dz) {
  const float k = 0.7978845608028654f, c = 0.044715f;
  const f32x2 z2 = z * z;
  const f32x2 a = (z * (2.f * k * kLog2e)) * (z2 * c + 1.f);
  const f32x2 e = f32x2{__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])} + 1.f;
  const f32x2 sg = f32x2{__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
  const f32x2 q = 1.f - sg;
  dz = q * (((z * sg) * (z2 * (3.f * c) + 1.f)) * (2.f * k) + 1.f);
  return z * q;
}

// out[j][col] (+)= sum_blocks part[block][j][col] in fixed order (fused_layer.hip).
hipError_t launch_colsum_reduce(const float* part, int nblocks, int ksets, int H, float* o0, float* o1,
                                float* o2, int accumulate, hipStream_t st);

// 16-bit dropout threshold and the exact inverse keep probability of the 16-bit test.
inline unsigned dropout_thresh16(float p) {
  if (!(p > 0.f)) return 0;
  const unsigned t = (unsigned)(p * 65536.0 + 0.5);
  return t < 1 ? 1 : (t > 65535 ? 65535 : t);
}
inline float dropout_inv_keep(unsigned thresh16) { return 65536.f / (65536.f - (float)thresh16); }

}  // namespace mmt
