// Helpers shared by the VALU-lean bf16 kernels (attn_fwd_band.hip, attn_bwd_band.hip).
#pragma once
#include "attn_tile.h"

namespace mmt {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) const float* lds_cfp;

__device__ __forceinline__ bf16x8 buf16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ int med3i(int x, int lo, int hi) {
  int r;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
  return r;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}


// Tiles a 32-row block visits: [a0, a0+lenA) U [b0, b0+lenB) U [c0, c0+lenC), ascending.
struct TileWalkLean {
  int a0 = 0, lenA = 0, b0 = 0, lenB = 0, c0 = 0, lenC = 0;
  __device__ __forceinline__ void set_chunk(int lo, int hi) { b0 = lo; lenB = hi - lo; }
  __device__ __forceinline__ void set_band(const PatternDev& pat, int x0, int S) {
    const int lo = max(x0 - pat.radius, 0), hi = min(x0 + 31 + pat.radius, S - 1);
    b0 = lo >> 5;
    const int b1 = hi >> 5;
    lenB = b1 - b0 + 1;
    if (pat.ng > 0) {
      const int g_lo = pat.g0 >> 5, g_hi = (pat.g0 + pat.ng - 1) >> 5;
      a0 = g_lo; lenA = max(0, min(g_hi, b0 - 1) - g_lo + 1);
      c0 = max(g_lo, b1 + 1); lenC = max(0, g_hi - c0 + 1);
    }
  }
  __device__ __forceinline__ int count() const { return lenA + lenB + lenC; }
  __device__ __forceinline__ int at(int it) const {
    return it < lenA ? a0 + it : (it < lenA + lenB ? b0 + (it - lenA) : c0 + (it - lenA - lenB));
  }
};

// Block -> (plane bn, block-in-plane) map shared by the lean kernels.  Blocks are dealt round-robin
// over the 8 XCDs, so block bid runs on XCD group bid % 8; each group walks its share of the (b,n)
// planes one after the other and, per plane, first the `per_bn` split blocks (global rows / keys)
// and then the `nblk` band blocks: every reader of one plane's Q/K/V shares one L2 and runs back
// to back.  Speed only -- any placement is correct.  blk < per_bn <=> split block.
__device__ __forceinline__ void plane_major_map(int bid, int BN, int per_bn, int nblk, int& bn, int& blk) {
  if ((BN & 7) == 0) {
    const int per_plane = per_bn + nblk, planes_per_xcd = BN >> 3;
    const int x = bid & 7, i = bid >> 3;
    const int pl = i / per_plane;
    bn = x * planes_per_xcd + pl;
    blk = i - pl * per_plane;
  } else {                              // fallback: all split blocks first, then the band blocks
    const int n_split = per_bn * BN;
    if (bid < n_split) { bn = bid / per_bn; blk = bid - bn * per_bn; }
    else { const int wg = bid - n_split; bn = wg / nblk; blk = per_bn + wg - bn * nblk; }
  }
}

// Wave-uniform class of one 32x32 tile of (row-block x0, other-block y0), d = key - query.
//   dmin / dmax : extreme key-query distances inside the tile
struct TileClass {
  bool plain;      // every pair exists and is unmasked
  bool edge;       // only the |d| <= W test can mask a pair (no pad boundary, no global keys/rows)
  bool far_neg;    // d <= -m for every pair  (clipped column 0)
  bool far_pos;    // d >=  m for every pair  (clipped column 2m)
  bool outside;    // the tile lies wholly outside the band: only global keys (rows) of it are visible
};
__device__ __forceinline__ TileClass classify_tile(int q0, int k0, int S, int valid_len, int W, int m,
                                                   bool ignore_band, bool no_global_in_tile) {
  TileClass t;
  const int dmin = k0 - (q0 + 31), dmax = k0 + 31 - q0;
  const bool in_range = (k0 + 31 < S) && (q0 + 31 < S);
  const bool seg_all = (q0 + 31 < valid_len && k0 + 31 < valid_len) || (q0 >= valid_len && k0 >= valid_len);
  const bool band_all = ignore_band || (dmin >= -W && dmax <= W);
  t.plain = in_range && seg_all && band_all;
  t.edge = in_range && seg_all && no_global_in_tile && !ignore_band;
  t.far_neg = dmax <= -m;
  t.far_pos = dmin >= m;
  t.outside = in_range && seg_all && !ignore_band && (dmin > W || dmax < -W);
  return t;
}

}  // namespace mmt
