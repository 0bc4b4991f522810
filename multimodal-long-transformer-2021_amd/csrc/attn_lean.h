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


// ---- 2-D relative ids (feature_utils.py:114-184) in the lean kernels ------------------------------------------------
// The table keeps its columns in id order (column = id); ids that cannot contribute (>= R under the one-hot
// lookup, or >= Rp) read the zero column Rp + 1 of the padded row.  For a tile whose 32 x 32 pairs are all
// image x image, the id is a function of (dx, dy) = patch-grid offset key - query; clamped to +-(r + 1) the
// offsets index a (2r + 3)^2 look-up table of column BYTE offsets kept in LDS, so that an element costs seven
// VALU instructions and two LDS reads instead of two integer divisions and the region logic of id_2d():
// the lane owns one end of the pair (its grid position is fixed for the kernel), the 16 registers walk 28
// consecutive positions of the other end, which cross a patch-row boundary at most once when P >= 32.
constexpr int kZeroCol(int Rp) { return Rp + 1; }
__device__ __forceinline__ int lut2d_entries(const PatternDev& pat) { const int n2 = 2 * pat.r + 3; return n2 * n2; }
template <int Rp>
__device__ __forceinline__ void build_lut2d(int* lut, const PatternDev& pat, int R, int t, int nthreads) {
  const int n2 = 2 * pat.r + 3, lim = pat.r + 1;
  for (int e = t; e < n2 * n2; e += nthreads) {
    const int cx = e / n2, cy = e - cx * n2;
    const int id = id_2d(cx - lim, cy - lim, pat.r);
    lut[e] = 4 * ((id < Rp && id < R) ? id : kZeroCol(Rp));
  }
}
template <int Rp>
__device__ __forceinline__ int col2d(const PatternDev& pat, int R, int q, int k) {    // general (any pair) column
  const int id = rel_id(pat, q, k);
  return ((unsigned)id < (unsigned)Rp && id < R) ? id : kZeroCol(Rp);
}
struct Ids2dTile { int g0, g1, dy0, dy1, wth; };
// SGN = +1: the registers walk keys vb + ci and the lane owns the query (xfix, yfix);  SGN = -1: the registers walk
// query rows and the lane owns the key.  vb = first walked position of this half-wave.
template <int SGN>
__device__ __forceinline__ Ids2dTile ids2d_tile(const PatternDev& pat, int lut_addr, int vb, int xfix, int yfix) {
  const int xvb = (int)__umulhi((unsigned)vb, pat.magicP), yvb = vb - xvb * pat.P;
  const int n2 = 2 * pat.r + 3, lim = pat.r + 1;
  const int dx0 = SGN > 0 ? xvb - xfix : xfix - xvb;
  const int dx1 = dx0 + SGN;
  Ids2dTile t;
  t.g0 = lut_addr + 4 * ((min(max(dx0, -lim), lim) + lim) * n2 + lim);
  t.g1 = lut_addr + 4 * ((min(max(dx1, -lim), lim) + lim) * n2 + lim);
  t.dy0 = SGN > 0 ? yvb - yfix : yfix - yvb;
  t.dy1 = t.dy0 - SGN * pat.P;
  t.wth = pat.P - yvb;
  return t;
}
template <int SGN>
__device__ __forceinline__ int ids2d_col4(const Ids2dTile& t, int ci, int nlim, int lim) {   // column byte offset of element ci
  const bool wrap = ci >= t.wth;
  const int dy = (wrap ? t.dy1 : t.dy0) + SGN * ci;
  const int a = (wrap ? t.g1 : t.g0) + 4 * med3i(dy, nlim, lim);
  return *(__attribute__((address_space(3))) const int*)(size_t)(unsigned)a;
}
__device__ __forceinline__ int lds_addr(const void* p) {
  return (int)(unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
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
