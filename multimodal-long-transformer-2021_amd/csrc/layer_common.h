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
__device__ __forceinline__ void load_param(const float* p, float (&v)[8]) { Chunk<float>::load(p, v); }

__device__ __forceinline__ float wave_sum(float x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}

// 16 random bits per element, one 32-bit mix per element pair (restated in oracle/layer_ops.py).
__host__ __device__ __forceinline__ uint32_t drop_bits16(uint32_t seed_lo, uint32_t seed_hi, uint64_t idx) {
  const uint64_t pair = idx >> 1;
  const uint32_t hsh = mix32((uint32_t)pair * 0x9E3779B9u + mix32((uint32_t)(pair >> 32) ^ seed_hi) + seed_lo);
  return (idx & 1) ? (hsh >> 16) : (hsh & 0xFFFFu);
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
