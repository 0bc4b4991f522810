// Tile-level building blocks shared by the forward and backward attention kernels (gfx950).
#pragma once
#include "attn_kernels.h"

namespace mmt {

template <typename T> struct Frag;

// ------------------------------- bf16: 32x32x16 MFMA ---------------------------------
// MFMA k-index (8h + j) of step s is mapped to head-dim d = 32h + 8s + j, so each lane
// loads 64 contiguous bytes of its row (4 x 16 B).
template <> struct Frag<__bf16> {
  bf16x8 v[4];
  __device__ __forceinline__ void load_row(const __bf16* row, int h) {
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = *reinterpret_cast<const bf16x8*>(row + 32 * h + 8 * s);
  }
};
__device__ __forceinline__ f32x16 mma_rows(const Frag<__bf16>& a, const Frag<__bf16>& b, f32x16 c) {
#pragma unroll
  for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v[s], b.v[s], c, 0, 0, 0);
  return c;
}

// ------------------------------- f32: 32x32x2 MFMA (exact f32) -----------------------
// MFMA k-index h of step s is mapped to d = 32h + s: each lane loads 128 contiguous bytes.
template <> struct Frag<float> {
  float v[32];
  __device__ __forceinline__ void load_row(const float* row, int h) {
#pragma unroll
    for (int s = 0; s < 32; s += 4) {
      f32x4 t = *reinterpret_cast<const f32x4*>(row + 32 * h + s);
      v[s] = t[0]; v[s + 1] = t[1]; v[s + 2] = t[2]; v[s + 3] = t[3];
    }
  }
};
__device__ __forceinline__ f32x16 mma_rows(const Frag<float>& a, const Frag<float>& b, f32x16 c) {
#pragma unroll
  for (int s = 0; s < 32; ++s) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[s], b.v[s], c, 0, 0, 0);
  return c;
}

// V rows of one tile held in registers between the global load and their use.
template <typename T> struct VTile;
template <> struct VTile<__bf16> {   // 4 x 16-B chunks per lane -> written to the LDS tile
  bf16x8 c[4];
  __device__ __forceinline__ void load(const __bf16* V, unsigned vs1, int k0, int S, int lane, int) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ci = lane + 64 * u, row = ci >> 3, ch = ci & 7;
      const unsigned kk = (unsigned)min(k0 + row, S - 1);  // rows past the end repeat the last row; their p is 0
      c[u] = *reinterpret_cast<const bf16x8*>(V + (kk * vs1 + (unsigned)ch * 8u));
    }
  }
  // 32 rows x 128 B; the two 64-B halves of a row are swapped when bit 1 of the row is set,
  // which makes the 4-row transposed reads below bank-conflict free.
  __device__ __forceinline__ void to_lds(unsigned char* vlds, int lane) const {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ci = lane + 64 * u, row = ci >> 3, ch = ci & 7;
      const int off = row * 128 + ((((ch >> 2) ^ ((row >> 1) & 1))) << 6) + (ch & 3) * 16;
      *reinterpret_cast<bf16x8*>(vlds + off) = c[u];
    }
  }
};
template <> struct VTile<float> {    // A operand of the 32x32x2 PV product, straight from L2
  float a0[16], a1[16];
  __device__ __forceinline__ void load(const float* V, unsigned vs1, int k0, int S, int lane, int) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const unsigned kk = (unsigned)min(k0 + kap(s, h), S - 1);
      const float* vr = V + kk * vs1;
      a0[s] = vr[r]; a1[s] = vr[32 + r];
    }
  }
  __device__ __forceinline__ void to_lds(unsigned char*, int) const {}
};

// Row fragment (MFMA A/B operand, row = lane & 31) read back from a wave-private LDS tile that
// was written with VTile<__bf16>::to_lds: saves the second, fragment-shaped global load.
__device__ __forceinline__ void frag_from_tile(Frag<__bf16>& f, const unsigned char* lds, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const unsigned char* row = lds + r * 128 + ((h ^ ((r >> 1) & 1)) << 6);
#pragma unroll
  for (int s = 0; s < 4; ++s) f.v[s] = *reinterpret_cast<const bf16x8*>(row + s * 16);
}

__device__ __forceinline__ float half_xchg(float x) { return __shfl_xor(x, 32, 64); }
// Reductions over the lane pair (l, l ^ 32) with gfx950's v_permlane32_swap: after the swap
// `a` holds the low half's value and `b` the high half's in every lane -- one VALU op and no LDS
// round trip (ds_bpermute) in the per-tile dependency chain.
__device__ __forceinline__ void half_pair(float x, float& a, float& b) {
  a = x; b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float half_max(float x) { float a, b; half_pair(x, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float half_sum(float x) { float a, b; half_pair(x, a, b); return a + b; }

// Row stride of the per-wave relative-score tables (floats).  Two access shapes matter: a column read by
// 32 lanes (address r * stride + c) and the DIAGONAL gather of a mixed-id tile, where lane r reads column
// c - r (address r * (stride - 1) + c).  Rp + 1 made the second a 32-way bank conflict (32 r + c hits two
// banks); with Rp + 2 the first walks 34 r (32 distinct even banks) and the second 33 r (all distinct).
constexpr int kTStride(int Rp) { return Rp + 2; }
constexpr float kRescaleThr = 6.0f;

// LDS carve per wave: T table [32][kTStride] f32, then (bf16 only) V tile 32 x 128 B.
template <typename T, int Rp> struct WaveLds {
  static constexpr int kTBytes = 32 * kTStride(Rp) * 4;
  static constexpr int kTBytesAligned = (kTBytes + 15) & ~15;
  static constexpr int kVBytes = sizeof(T) == 2 ? 32 * 128 : 0;
  static constexpr int kBytes = kTBytesAligned + kVBytes;
};

// Column of relative id `id` inside the LDS table.  For the 1-D generator the columns are
// permuted so that column = clamp(k - q, -m, m) + m: the hot loop then needs no sign
// handling (id <= m  <->  d = id;  m < id <= 2m  <->  d = m - id).
__device__ __forceinline__ int tcol(int perm_1d, int m, int id) {
  const int pc = id <= m ? m + id : 2 * m - id;
  return ((perm_1d != 0) & (id <= 2 * m)) ? pc : id;
}
// Inverse of tcol for the permuted (1-D) layout: the relative id whose score lives in table column c.  The lean
// kernels load E row icol(m, r) into fragment row r, so that the table product comes out in COLUMN order and is
// stored without a per-element tcol().
__device__ __forceinline__ int icol(int m, int c) {
  return c > 2 * m ? c : (c >= m ? c - m : 2 * m - c);
}


// acc^T[d x col] += X^T[d x row] . vals[row x col] for a 32-row tile X staged as a VTile:
// the 32x32 accumulator-layout values `vals` (16 per lane) are the B operand as they stand
// (guide: "an accumulator tile as the next MFMA's operand"); X^T fragments come from the
// wave-private LDS tile through ds_read_b64_tr_b16 (bf16) or straight from registers (f32).
// Eight f32 -> eight bf16 (round to nearest even) as FOUR v_cvt_pk_bf16_f32: a vector conversion.  Converted element by
// element, hipcc emits one conversion per value and merges the halves with v_perm_b32 -- 24 instructions for a 16-value
// fragment where 8 do.  (Not inline asm: an MFMA that reads a register an `asm` has just written gets no hazard padding
// from the compiler -- a hand-written v_cvt_pk in front of the peeled step's products gave infinities in 4-lane groups.)
typedef __attribute__((ext_vector_type(8))) float f32x8_t;
__device__ __forceinline__ bf16x8 pack8_bf16(const float* v) {
  const f32x8_t x = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
  return __builtin_convertvector(x, bf16x8);
}

__device__ __forceinline__ void mma_xt(f32x16& a0, f32x16& a1, const VTile<__bf16>&,
                                       const unsigned char* xlds, const float (&vals)[16], int lane) {
  const int h = lane >> 5, li = lane & 15, cb = (lane >> 4) & 1;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const bf16x8 pf = pack8_bf16(vals + 8 * s);
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int row = 16 * s + 4 * h + (li >> 2);
      const int within = 32 * cb + 8 * (li & 3);  // byte offset inside the 64-B half
      const int off0 = row * 128 + ((db ^ ((row >> 1) & 1)) << 6) + within;
      const int row1 = row + 8;
      const int off1 = row1 * 128 + ((db ^ ((row1 >> 1) & 1)) << 6) + within;
      s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(xlds + off0));
      s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
          (__attribute__((address_space(3))) s16x4*)(xlds + off1));
      bf16x8 vf;
      bf16x4 lo4 = __builtin_bit_cast(bf16x4, lo), hi4 = __builtin_bit_cast(bf16x4, hi);
#pragma unroll
      for (int j = 0; j < 4; ++j) { vf[j] = lo4[j]; vf[4 + j] = hi4[j]; }
      if (db == 0) a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, a0, 0, 0, 0);
      else a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, a1, 0, 0, 0);
    }
  }
}
__device__ __forceinline__ void mma_xt(f32x16& a0, f32x16& a1, const VTile<float>& x,
                                       const unsigned char*, const float (&vals)[16], int) {
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x.a0[s], vals[s], a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x.a1[s], vals[s], a1, 0, 0, 0);
  }
}

// Same product with `vals` split into bf16 hi + lo parts (about 16 mantissa bits): used where
// the right-hand values are sums that must not be rounded to bf16 (dRel . E).
__device__ __forceinline__ void mma_xt_hilo(f32x16& a0, f32x16& a1, const VTile<__bf16>& x,
                                            const unsigned char* xlds, const float (&vals)[16], int lane) {
  float hi[16], lo[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { hi[i] = (float)(__bf16)vals[i]; lo[i] = vals[i] - hi[i]; }
  mma_xt(a0, a1, x, xlds, hi, lane);
  mma_xt(a0, a1, x, xlds, lo, lane);
}
__device__ __forceinline__ void mma_xt_hilo(f32x16& a0, f32x16& a1, const VTile<float>& x,
                                            const unsigned char* xlds, const float (&vals)[16], int lane) {
  mma_xt(a0, a1, x, xlds, vals, lane);
}

}  // namespace mmt
