// VALU-lean backward kernels for the common case (bf16, band + global pattern, relative ids
// none or 1-D with the permuted table).  Same math, work items, workspace layout and outputs as
// attn_bwd_dq_kernel / attn_bwd_dkv_kernel<kBand> (attn_bwd.hip, which stays the general path:
// fp32, 2-D ids, dense inputs); see attn_fwd_band.hip for the tile classes.  Differences:
//   * P needs no running maximum (LSE is known): p = exp2(fma(c, s, rel - lse)) in class A;
//   * gscale / rel_gscale are folded into the epilogues, not applied per element;
//   * dRel: clipped columns accumulate in two registers, near columns are plain stores;
//   * row fragments of a tile that is staged in LDS anyway are read back from LDS
//     (frag_from_tile) instead of being loaded a second time in fragment shape.
#include <cstdlib>
#include "attn_lean.h"
#include "attn_combine.h"

namespace mmt {

template <int Rp> struct LeanLds {
  static constexpr int kTab = (32 * kTStride(Rp) * 4 + 15) & ~15;
  static constexpr int kTile = 32 * 128;
  static constexpr int kDq = 2 * kTab + kTile;            // T, dT, K/Q/E tile (bias row aliases the tile)
  static constexpr int kDkv = kTab + 2 * kTile + Rp * 4 + 512;  // T, Q tile, dO tile, bias row, per-row constants
};

__device__ __forceinline__ void tile_to_lds(unsigned char* lds, const bf16x8 (&v)[4], int lane) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int ci = lane + 64 * u, row = ci >> 3, ch = ci & 7;
    *reinterpret_cast<bf16x8*>(lds + row * 128 + ((((ch >> 2) ^ ((row >> 1) & 1))) << 6) + (ch & 3) * 16) = v[u];
  }
}
// E rows in table-column order (row c <- relative id icol(m, c); ids >= R read as zeros through the buffer
// range check), staged ONCE per workgroup as Rp/32 swizzled 32 x 128-byte tile images at the start of LDS.  The
// table products (E . Q^T), the dQ += E^T . dRel^T product and the mixed-id table rebuilds of the dK/dV pass
// read their fragments from it instead of waiting for L2 once per item / per mixed-id tile.
template <int Rp, int REL>
__device__ __forceinline__ void stage_e_image(unsigned char* elds, const void* emb, int n, int N, int R, int m, int tid) {
  using T = __bf16;
  const T* Eb = reinterpret_cast<const T*>(emb) + (long)n * 64;
  const unsigned es1b = (unsigned)N * 128;
  const auto re = make_rsrc(Eb, (unsigned)(R - 1) * es1b + 128);
#pragma unroll
  for (int c0 = 0; c0 < Rp * 8; c0 += 256) {
    const int ci = c0 + tid, row = ci >> 3, ch = ci & 7, rt = row & 31;
    const bf16x8 e = buf16(re, (unsigned)(REL == 2 ? row : icol(m, row)) * es1b + ch * 16, 0u);   // 2-D ids: columns in id order
    *reinterpret_cast<bf16x8*>(elds + (row >> 5) * 4096 + rt * 128 + ((((ch >> 2) ^ ((rt >> 1) & 1))) << 6) + (ch & 3) * 16) = e;
  }
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// LDS float += v (no return value).  The rows of the per-wave dRel table are lane-private, so the only accesses
// that meet on one address are this lane's own, and LDS executes a wave's instructions in order.
__device__ __forceinline__ void lds_add_f32(int addr, float v) {
  asm volatile("ds_add_f32 %0, %1" : : "v"(addr), "v"(v) : "memory");
}

// ---- P hand-over between the passes (BwdParams::ho) ----------------------------------------------------------------
// The dQ pass stores every tile's probabilities as bf16, the SIGN bit carrying the dropout decision (set = dropped; P is
// never negative), and the dK/dV pass rebuilds P' = keep ? P / keep_prob : 0 and dS = P (dP' - delta) from them: dP is one
// more MFMA group there, the scores, the exponentials, the relative ids, the masks and the keep hashes are not redone.
// A 32 x 32 tile is stored the way the dQ pass holds it (lane = query row r, half h, registers 8 g .. 8 g + 7 = keys
// 16 g + 4 h + {0..3, 8..11}): four blocks (2 g + h) of 512 bytes, row r of a block in the 16-byte unit r ^ (block << 2),
// its keys +0..3 in the first and +8..11 in the second 8 bytes.  One store instruction writes a contiguous KiB, and the
// dK/dV pass's transposed LDS reads (four rows x four 4-key chunks per 16 lanes) find their chunks in different banks.
// Global-row tiles (8 rows x 32 keys): [block][row][16 bytes]; global-key strips (32 rows x 8 keys): [h][row][8 bytes].
constexpr int kHoTile = 2048, kHoStrip = 512;
__device__ __forceinline__ int ho_unit_off(int blk, int q) { return blk * 512 + ((q ^ (blk << 2)) << 4); }
__device__ __forceinline__ size_t ho_band_plane(const BwdParams& p, int n_tiles) { return (size_t)n_tiles * p.ho_slots * kHoTile; }
__device__ __forceinline__ unsigned char* ho_band_base(const BwdParams& p, int n_tiles, int bn) {          // [q block][slot] tiles of a plane
  return p.ho + (size_t)bn * ho_band_plane(p, n_tiles);
}
__device__ __forceinline__ unsigned char* ho_strip(const BwdParams& p, int n_tiles, int bn, int qb) {      // 32 rows x 8 global keys
  return p.ho + (size_t)p.B * p.N * ho_band_plane(p, n_tiles) + ((size_t)bn * n_tiles + qb) * kHoStrip;
}
__device__ __forceinline__ unsigned char* ho_grow_base(const BwdParams& p, int n_tiles, int bn) {          // [key block] 8-row tiles of a plane
  return p.ho + (size_t)p.B * p.N * (ho_band_plane(p, n_tiles) + (size_t)n_tiles * kHoStrip) + (size_t)bn * n_tiles * kHoStrip;
}
__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d) { return bf16x4{(__bf16)a, (__bf16)b, (__bf16)c, (__bf16)d}; }
#ifndef MMT_HO_AUX
#define MMT_HO_AUX 2          // nt: written once, read once by the next kernel -- keep it out of the way of the K / V / Q lines in L2
#endif
__device__ __forceinline__ void buf_store16(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, const bf16x8& v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, voff, soff, MMT_HO_AUX);
}

// =========================================================================================
// dQ, delta, dRel (lane = query row).
// =========================================================================================
template <int Rp, int REL>         // REL: 0 none, 1 = 1-D ids (permuted table), 2 = 2-D ids (columns in id order, attn_lean.h)
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_band_bf16_kernel(const BwdParams p) {
  using T = __bf16;
  using L = LeanLds<Rp>;
  constexpr bool HAS_REL = REL != 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  constexpr int kEImg = HAS_REL ? Rp * 128 : 0;
  unsigned char* elds = smem;
  // per wave: T table | dRel table (row stride p.dstride floats: only columns 0 .. 2m are ever written, and with
  // the narrow stride three workgroups fit the LDS of a CU) | K / Q tile
  const int dstride = p.dstride;
  const int kWave = L::kTab + 32 * dstride * 4 + 2 * L::kTile;       // + V tile
  unsigned char* wl = smem + kEImg + wave * kWave;
  float* tab = reinterpret_cast<float*>(wl);
  float* dtab = reinterpret_cast<float*>(wl + L::kTab);
  unsigned char* xlds = wl + L::kTab + 32 * dstride * 4;
  unsigned char* vlds = xlds + L::kTile;
  int* lut = reinterpret_cast<int*>(smem + kEImg + 4 * kWave);        // REL == 2: one per workgroup

#ifdef MMT_STAMP
  long long* dbg = (p.dbg && (p.dbg_mode & 1) == 0 && blockIdx.x == 900) ? p.dbg + wave * 32 : nullptr;
#define QSTAMP(i) do { if (dbg && lane == 0) dbg[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define QSTAMP(i) do { } while (0)
#endif
  QSTAMP(0);
  const int n_tiles = (p.S + 31) >> 5, nqb = (p.S + 127) >> 7;
  const int per_bn = (p.n_chunks * p.n_gblk + 3) >> 2;
  // long (global-row) items first, then the band blocks with the XCD remap: measured 8 % faster
  // for this kernel than the plane-major order the forward and dK/dV kernels use
  const int n_split_blocks = per_bn * p.B * p.N;
  bool split_item = (int)blockIdx.x < n_split_blocks;
  int bn, q0, chunk = 0, gblk = 0, band_wg = 0;
  bool live = true;
  if (p.dq_plane_major) {              // MMT_DQ_PLANE_MAJOR=1: the forward's / dK-dV pass's plane-major placement -- 26 % less HBM
                                       // fetch in this pass (98 -> 72 MB per call) for +2 % time: off by default
    int blk;
    plane_major_map(blockIdx.x, p.B * p.N, per_bn, nqb, bn, blk);
    split_item = blk < per_bn;
    if (split_item) {
      const int item = blk * 4 + wave;
      live = item < p.n_chunks * p.n_gblk;
      gblk = live ? item / p.n_chunks : 0;
      chunk = item - gblk * p.n_chunks;
      q0 = p.pat.g0 + gblk * 32;
    } else {
      band_wg = bn * nqb + (blk - per_bn);
      q0 = (blk - per_bn) * 128 + wave * 32;
      live = q0 < p.S;
    }
  } else if (split_item) {
    bn = blockIdx.x / per_bn;
    const int item = (blockIdx.x - bn * per_bn) * 4 + wave;
    live = item < p.n_chunks * p.n_gblk;
    gblk = live ? item / p.n_chunks : 0;
    chunk = item - gblk * p.n_chunks;
    q0 = p.pat.g0 + gblk * 32;
  } else {
    band_wg = xcd_remap(blockIdx.x - n_split_blocks, p.n_band_blocks);
    bn = band_wg / nqb;
    q0 = (band_wg - bn * nqb) * 128 + wave * 32;
    live = q0 < p.S;
  }
  const int b = bn / p.N, n = bn - b * p.N;
  const int q = q0 + r;
  const bool q_ok = q < p.S;
  const int valid_len = p.valid_len ? p.valid_len[b] : p.S;
  const int W = p.pat.radius, m = p.pat.m;
  const bool ignore_band = split_item;

  const unsigned qs1b = (unsigned)p.qs[1] * 2, ks1b = (unsigned)p.ks[1] * 2, vs1b = (unsigned)p.vs[1] * 2, os1b = (unsigned)p.os[1] * 2;
  const T* Qb = reinterpret_cast<const T*>(p.q) + (long)b * p.qs[0] + (long)n * p.qs[2];
  const T* Kb = reinterpret_cast<const T*>(p.k) + (long)b * p.ks[0] + (long)n * p.ks[2];
  const T* Vb = reinterpret_cast<const T*>(p.v) + (long)b * p.vs[0] + (long)n * p.vs[2];
  const T* Ob = reinterpret_cast<const T*>(p.out) + (long)b * p.os[0] + (long)n * p.os[2];
  const T* DOb = reinterpret_cast<const T*>(p.dout) + (long)b * p.os[0] + (long)n * p.os[2];
  const auto rq = make_rsrc(Qb, (unsigned)(p.S - 1) * qs1b + 128);
  const auto rk = make_rsrc(Kb, (unsigned)(p.S - 1) * ks1b + 128);
  const auto rv = make_rsrc(Vb, (unsigned)(p.S - 1) * vs1b + 128);
  const auto ro = make_rsrc(Ob, (unsigned)(p.S - 1) * os1b + 128);
  const auto rdo = make_rsrc(DOb, (unsigned)(p.S - 1) * os1b + 128);
  const unsigned voff_kc = (unsigned)(lane >> 3) * ks1b + (lane & 7) * 16;    // tile (coalesced) shape
  const unsigned voff_vc = (unsigned)(lane >> 3) * vs1b + (lane & 7) * 16;    // V rows: tile (coalesced) shape as well --
  // fragment-shaped loads touch every 128-byte line of the tile four times (4 instructions x 32 lines)

  TileWalkLean w;
  if (split_item) w.set_chunk(chunk * p.chunk_tiles, min(n_tiles, (chunk + 1) * p.chunk_tiles));
  else w.set_band(p.pat, live ? q0 : 0, p.S);
  // The (<= 8) global keys outside this wave's band tiles: not a sixth tile of the walk but a PEELED step before it
  // (registers 0..3 only: a quarter of a tile's VALU work, half of its dQ MFMAs, one K and one V load instruction).
  const bool peel = (p.peel_gkeys & 1) && !split_item && live &&
                    !(p.pat.g0 >= w.b0 * 32 && p.pat.g0 + p.pat.ng - 1 <= (w.b0 + w.lenB) * 32 - 1);
  if ((p.peel_gkeys & 1) && !split_item) { w.lenA = 0; w.lenC = 0; }
  const int n_it = live ? w.count() : 0;

  Frag<T> qf, dof;
  bf16x8 kt[4], vt[4], kg, vg;
  float delta;
  if (peel) {              // rows g0 .. g0 + 7 of K and V, tile shape (rows past the end read as zeros)
    kg = buf16(rk, voff_kc, (unsigned)p.pat.g0 * ks1b);
    vg = buf16(rv, voff_vc, (unsigned)p.pat.g0 * vs1b);
  }
  {
    // the item's own rows and its first K / V tile are requested BEFORE the workgroup stages the E image: the two
    // memory round trips overlap instead of following each other (waves past the end read zeros and leave below)
    Frag<T> of;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf.v[s] = buf16(rq, (unsigned)r * qs1b + 64 * h + 16 * s, (unsigned)q0 * qs1b);
      dof.v[s] = buf16(rdo, (unsigned)r * os1b + 64 * h + 16 * s, (unsigned)q0 * os1b);
      of.v[s] = buf16(ro, (unsigned)r * os1b + 64 * h + 16 * s, (unsigned)q0 * os1b);
    }
    const unsigned k0 = (unsigned)w.at(0) * 32;
#pragma unroll
    for (int u = 0; u < 4; ++u) kt[u] = buf16(rk, voff_kc, (k0 + 8 * u) * ks1b);
#pragma unroll
    for (int u = 0; u < 4; ++u) vt[u] = buf16(rv, voff_vc, (k0 + 8 * u) * vs1b);
    QSTAMP(1);
    if (HAS_REL) {           // every wave of the workgroup takes part (the plane, hence E, is the same for all four)
      stage_e_image<Rp, REL>(elds, p.emb, n, p.N, p.R, p.pat.m, threadIdx.x);
      if (REL == 2) build_lut2d<Rp>(lut, p.pat, p.R, threadIdx.x, 256);
      __syncthreads();
    }
    QSTAMP(2);
    if (!live) return;
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc = fmaf((float)of.v[s][j], (float)dof.v[s][j], acc);
    delta = half_sum(acc);
  }
  const long row_id = ((long)b * p.N + n) * p.S + min(q, p.S - 1);
  if (!split_item && q_ok && h == 0) p.delta[row_id] = delta;
  const float lse2 = p.lse[row_id] * kLog2e;

  float relfn = 0.f, relfp = 0.f;
  for (int i = lane; i < 32 * dstride; i += 64) dtab[i] = 0.f;
  if (HAS_REL) {
    float* bias_ts = reinterpret_cast<float*>(xlds);
    if (lane < Rp) {
      const int idc = REL == 2 ? lane : icol(m, lane);
      bias_ts[lane] = (p.bias && idc < p.R) ? (float)reinterpret_cast<const T*>(p.bias)[(long)idc * p.N + n] * p.tscale : 0.f;   // by column
    }
    if (REL == 2) tab[r * kTStride(Rp) + kZeroCol(Rp)] = 0.f;
    wave_lds_sync();
#pragma unroll
    for (int rb = 0; rb < Rp / 32; ++rb) {
      Frag<T> ef;
      frag_from_tile(ef, elds + rb * 4096, lane);   // row r <- E row of table column rb*32 + r
      f32x16 c = {0};
      c = mma_rows(ef, qf, c);
      float bv[16];                      // the bias values in one batch, then the stores (see the dK/dV pass's rebuild)
#pragma unroll
      for (int i = 0; i < 16; ++i) bv[i] = bias_ts[rb * 32 + kap(i, h)];
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(bv[i]));
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int col = rb * 32 + kap(i, h);
        tab[r * kTStride(Rp) + col] = fmaf(c[i], p.tscale, bv[i]);
      }
    }
    wave_lds_sync();
    if (REL == 1) {
      relfn = tab[r * kTStride(Rp)];
      relfp = tab[r * kTStride(Rp) + 2 * m];
      if (!split_item && q_ok && h == 0) {   // reused by the dK/dV pass for its one-id tiles
        p.relfar[row_id * 2] = relfn;
        p.relfar[row_id * 2 + 1] = relfp;
      }
    }
  }
  wave_lds_sync();

  QSTAMP(3);
  f32x16 a0 = {0}, a1 = {0};
  float far_neg_acc = 0.f, far_pos_acc = 0.f;
  const float* trow = tab + r * kTStride(Rp);
  float* dtrow = dtab + r * dstride;
  const int trow_addr = (int)(unsigned)(size_t)(__attribute__((address_space(3))) float*)(tab + r * kTStride(Rp));
  const SeedPair sd = effective_seed(p.seed_lo, p.seed_hi, p.epoch);
  const uint32_t drop_base = drop_row_base(sd.lo, sd.hi, (uint32_t)bn, (uint32_t)q);
  // REL == 2: this lane's query on the patch grid, LDS addresses of the look-up table and of the lane's dRel row
  const int xq2 = (int)__umulhi((unsigned)q, p.pat.magicP), yq2 = q - xq2 * p.pat.P;
  const int lut_addr = lds_addr(lut), dtrow_addr = lds_addr(dtrow);
  const int lim2 = p.pat.r + 1, nlim2 = -lim2;

  // P hand-over: a tile's image waits in registers and is stored in the NEXT tile step, ahead of that step's prefetch
  // loads: vmcnt counts loads and stores in issue order and hipcc waits for the prefetched K / V tile with vmcnt(0) at
  // the loop top, so stores issued behind the prefetch would be waited for there, one tile step after they were issued.
  bf16x8 hold_p[2];
  unsigned hold_so = 0;          // byte offset of the held tile inside the plane's region
  // band waves: the plane's [q block][slot] tiles; global-row items: its [key block] 8-row tiles (rows past the last
  // global token stay unwritten: the dK/dV pass reads them as zeros of its own)
  const auto rho = make_rsrc(p.ho ? (split_item ? ho_grow_base(p, n_tiles, bn) : ho_band_base(p, n_tiles, bn)) : nullptr,
                             p.ho ? (unsigned)(split_item ? n_tiles * kHoStrip : ho_band_plane(p, n_tiles)) : 0u);
  unsigned hoff0 = split_item ? (r < p.pat.ng ? h * 128 + r * 16 : 0x80000000u) : ho_unit_off(h, r);          // g = 0; (out of range: no store)
  unsigned hoff1 = split_item ? (r < p.pat.ng ? (2 + h) * 128 + r * 16 : 0x80000000u) : ho_unit_off(2 + h, r);  // g = 1
  auto ho_flush = [&]() {
    buf_store16(rho, hoff0, hold_so, hold_p[0]);
    buf_store16(rho, hoff1, hold_so, hold_p[1]);
  };

  if (peel) {
    {   // the 8 rows into the tile images (rows 0..7); rows 8..15 of the K image as zeros (dS is 0 there, 0 x garbage is not)
      const int row = lane >> 3, ch = lane & 7;
      const int off = row * 128 + ((((ch >> 2) ^ ((row >> 1) & 1))) << 6) + (ch & 3) * 16;
      *reinterpret_cast<bf16x8*>(xlds + off) = kg;
      *reinterpret_cast<bf16x8*>(vlds + off) = vg;
      const bf16x8 z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
      *reinterpret_cast<bf16x8*>(xlds + 1024 + off) = z;
    }
    wave_lds_sync();
    const int lo_k = w.b0 * 32, hi_k = (w.b0 + w.lenB) * 32 - 1;          // keys of this wave's band tiles
    const int kg0 = p.pat.g0, n_here = p.pat.ng;
    Frag<T> kf, vf;
    {
      const int rr = r & 7;                              // rows 8..31 of the "tile" are not used: any finite row
      const unsigned char* krow = xlds + rr * 128 + ((h ^ ((rr >> 1) & 1)) << 6);
      const unsigned char* vrow = vlds + rr * 128 + ((h ^ ((rr >> 1) & 1)) << 6);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        kf.v[s] = *reinterpret_cast<const bf16x8*>(krow + s * 16);
        vf.v[s] = *reinterpret_cast<const bf16x8*>(vrow + s * 16);
      }
    }
    f32x16 c = {0}, dp = {0};
    c = mma_rows(kf, qf, c);
    dp = mma_rows(vf, dof, dp);
    const bool qv = q < valid_len;
    float ds[16], pd4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kk = kg0 + i + 4 * h;
      const bool present = (i + 4 * h < n_here) && !(kk >= lo_k && kk <= hi_k) && q_ok && kk < p.S;
      const bool neg = kk < q;                             // beyond the radius on that side: clipped id (1-D)
      int col4_i = 0;                                      // 2-D ids: byte offset of the pair's table column
      float rel_i = HAS_REL ? (neg ? relfn : relfp) : 0.f;
      if (REL == 2) {
        col4_i = 4 * col2d<Rp>(p.pat, p.R, q, min(kk, p.S - 1));
        rel_i = *(lds_cfp)(size_t)(unsigned)(trow_addr + col4_i);
      }
      float sc = fmaf(c[i], p.sscale, rel_i);
      sc = ((kk < valid_len) == qv) ? sc : sc + p.mask_add;
      const float pr = present ? __builtin_amdgcn_exp2f(sc - lse2) : 0.f;
      float f = 1.f;
      if (p.drop_thresh) f = drop_bits16(drop_base, (uint32_t)kk) >= p.drop_thresh ? p.inv_keep : 0.f;
      ds[i] = pr * (dp[i] * f - delta);
      pd4[i] = f != 0.f ? pr : -pr;
      if (REL == 2) lds_add_f32(dtrow_addr + col4_i, ds[i] * p.rel_gscale);       // (0 when the pair is not present)
      else if (HAS_REL) { far_neg_acc += neg ? ds[i] : 0.f; far_pos_acc += neg ? 0.f : ds[i]; }
    }
    if (p.ho) {              // the strip of this q block: rows x the 8 global keys (zeros where a band tile holds the pair)
      unsigned char* sp = ho_strip(p, n_tiles, bn, q0 >> 5) + lane * 8;
      *reinterpret_cast<bf16x4*>(sp) = pack4(pd4[0], pd4[1], pd4[2], pd4[3]);
    }
#pragma unroll
    for (int i = 4; i < 16; ++i) ds[i] = 0.f;
    {   // dQ^T += K_g^T . dS^T over rows 0..15 of the image (mma_xt's first key step)
      const int li = lane & 15, cb = (lane >> 4) & 1;
      bf16x8 pf;
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = (__bf16)ds[j];
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const int row = 4 * h + (li >> 2);
        const int within = 32 * cb + 8 * (li & 3);
        const int off0 = row * 128 + ((db ^ ((row >> 1) & 1)) << 6) + within;
        const int row1 = row + 8;
        const int off1 = row1 * 128 + ((db ^ ((row1 >> 1) & 1)) << 6) + within;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(xlds + off0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(xlds + off1));
        bf16x8 xf;
        bf16x4 lo4 = __builtin_bit_cast(bf16x4, lo), hi4 = __builtin_bit_cast(bf16x4, hi);
#pragma unroll
        for (int j = 0; j < 4; ++j) { xf[j] = lo4[j]; xf[4 + j] = hi4[j]; }
        if (db == 0) a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, pf, a0, 0, 0, 0);
        else a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf, pf, a1, 0, 0, 0);
      }
    }
    wave_lds_sync();                 // the first band tile overwrites the images
  } else if (p.ho && p.n_gblk > 0 && !split_item) {       // every global key sits in this wave's band tiles: an empty strip
    unsigned char* sp = ho_strip(p, n_tiles, bn, q0 >> 5) + lane * 8;
    *reinterpret_cast<bf16x4*>(sp) = pack4(0.f, 0.f, 0.f, 0.f);
  }

  for (int it = 0; it < n_it; ++it) {
    const int k0 = w.at(it) * 32;
    tile_to_lds(xlds, kt, lane);
    tile_to_lds(vlds, vt, lane);
    wave_lds_sync();
    Frag<T> kf, vf;
    frag_from_tile(kf, xlds, lane);
    frag_from_tile(vf, vlds, lane);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) asm volatile("" : "+v"(kf.v[s4]), "+v"(vf.v[s4]));   // all eight reads in flight before the first MFMA
    f32x16 c = {0}, dp = {0};
    c = mma_rows(kf, qf, c);      // S^T  [key x q]
    dp = mma_rows(vf, dof, dp);   // dP^T [key x q]
    if (p.ho && it > 0) ho_flush();
    if (it + 1 < n_it) {
      const unsigned k1 = (unsigned)w.at(it + 1) * 32;
#pragma unroll
      for (int u = 0; u < 4; ++u) kt[u] = buf16(rk, voff_kc, (k1 + 8 * u) * ks1b);
#pragma unroll
      for (int u = 0; u < 4; ++u) vt[u] = buf16(rv, voff_vc, (k1 + 8 * u) * vs1b);
    } else if (HAS_REL && !split_item) {
      // last tile: the K staging registers are free -- fetch this block's Q rows in tile shape for the dE
      // contraction of the epilogue now, so that the wave does not end on an exposed memory round trip
#pragma unroll
      for (int u = 0; u < 4; ++u) kt[u] = buf16(rq, (unsigned)(lane >> 3) * qs1b + (lane & 7) * 16, (unsigned)(q0 + 8 * u) * qs1b);
    }
    const bool no_gkey = p.pat.ng == 0 || k0 + 31 < p.pat.g0 || k0 >= p.pat.g0 + p.pat.ng;
    const TileClass tc = classify_tile(q0, k0, p.S, valid_len, W, m, ignore_band, no_gkey);
    const bool one_id = REL != 2 && (!HAS_REL || tc.far_neg || tc.far_pos);
    const float relc = HAS_REL ? (tc.far_neg ? relfn : relfp) : 0.f;
    const int dbase = k0 - q + 4 * h;

    float pr[16];
    int col4[16];                                               // REL == 2: byte offset of each element's table column
    if (REL == 2) {
      if (q0 + 31 < p.pat.I && k0 + 31 < p.pat.I && p.pat.P >= 32) {     // image x image: look-up table
        const Ids2dTile t2 = ids2d_tile<1>(p.pat, lut_addr, k0 + 4 * h, xq2, yq2);
#pragma unroll
        for (int i = 0; i < 16; ++i) col4[i] = ids2d_col4<1>(t2, (i & 3) + 8 * (i >> 2), nlim2, lim2);
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) col4[i] = 4 * col2d<Rp>(p.pat, p.R, q, k0 + 4 * h + (i & 3) + 8 * (i >> 2));
      }
      float sc[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) sc[i] = fmaf(c[i], p.sscale, *(lds_cfp)(size_t)(unsigned)(trow_addr + col4[i])) - lse2;
      if (tc.plain) {
#pragma unroll
        for (int i = 0; i < 16; ++i) pr[i] = __builtin_amdgcn_exp2f(sc[i]);
      } else if (tc.edge) {
        const unsigned W2 = 2u * (unsigned)W;
#pragma unroll
        for (int i = 0; i < 16; ++i)
          pr[i] = __builtin_amdgcn_exp2f(sc[i] + ((unsigned)(dbase + (i & 3) + 8 * (i >> 2) + W) <= W2 ? 0.f : p.mask_add));
      } else if (tc.outside) {
        const unsigned gb = (unsigned)(k0 + 4 * h - p.pat.g0), ng = (unsigned)p.pat.ng;
#pragma unroll
        for (int i = 0; i < 16; ++i)
          pr[i] = __builtin_amdgcn_exp2f(sc[i] + (gb + (unsigned)((i & 3) + 8 * (i >> 2)) < ng ? 0.f : p.mask_add));
      } else {
        const int kb = k0 + 4 * h;
        const bool qv = q < valid_len;
        const unsigned W2 = 2u * (unsigned)W;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int ci = (i & 3) + 8 * (i >> 2);
          const int kk = kb + ci, d = dbase + ci;
          const bool near = ignore_band | ((unsigned)(d + W) <= W2);
          const bool gk = (unsigned)(kk - p.pat.g0) < (unsigned)p.pat.ng;
          const bool seg = (kk < valid_len) == qv;
          const bool keep = (int)seg & ((int)near | (int)gk);
          pr[i] = (kk < p.S && q_ok) ? __builtin_amdgcn_exp2f(keep ? sc[i] : sc[i] + p.mask_add) : 0.f;
        }
      }
    } else if (tc.plain && one_id) {                            // class A
      const float rc = relc - lse2;
#pragma unroll
      for (int i = 0; i < 16; ++i) pr[i] = __builtin_amdgcn_exp2f(fmaf(c[i], p.sscale, rc));
    } else if (tc.plain) {                                      // class B
      const int abase = trow_addr + 4 * (m + dbase), alo = trow_addr, ahi = trow_addr + 8 * m;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int a = med3i(abase + 4 * ((i & 3) + 8 * (i >> 2)), alo, ahi);
        pr[i] = __builtin_amdgcn_exp2f(fmaf(c[i], p.sscale, *(lds_cfp)(size_t)(unsigned)a) - lse2);
      }
    } else if (tc.edge && one_id) {                             // class D
      const unsigned W2 = 2u * (unsigned)W;
      const float rc = relc - lse2;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const unsigned dd = (unsigned)(dbase + (i & 3) + 8 * (i >> 2) + W);
        pr[i] = __builtin_amdgcn_exp2f(fmaf(c[i], p.sscale, rc) + (dd <= W2 ? 0.f : p.mask_add));
      }
    } else if (tc.outside && one_id) {                          // class G: only the tile's global keys are visible
      const unsigned gb = (unsigned)(k0 + 4 * h - p.pat.g0), ng = (unsigned)p.pat.ng;
      const float rc = relc - lse2;
#pragma unroll
      for (int i = 0; i < 16; ++i)
        pr[i] = __builtin_amdgcn_exp2f(fmaf(c[i], p.sscale, rc) + (gb + (unsigned)((i & 3) + 8 * (i >> 2)) < ng ? 0.f : p.mask_add));
    } else {                                                    // class C
      const int kb = k0 + 4 * h;
      const bool qv = q < valid_len;
      const unsigned W2 = 2u * (unsigned)W;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ci = (i & 3) + 8 * (i >> 2);
        const int kk = kb + ci, d = dbase + ci;
        const bool near = ignore_band | ((unsigned)(d + W) <= W2);
        const bool gk = (unsigned)(kk - p.pat.g0) < (unsigned)p.pat.ng;
        const bool seg = (kk < valid_len) == qv;
        const bool keep = (int)seg & ((int)near | (int)gk);
        float rel = 0.f;
        if (HAS_REL) rel = trow[min(max(d, -m), m) + m];
        float s = fmaf(c[i], p.sscale, rel);
        s = keep ? s : s + p.mask_add;
        pr[i] = (kk < p.S && q_ok) ? __builtin_amdgcn_exp2f(s - lse2) : 0.f;
      }
    }
    // dS = P o (dP' - delta)
    float ds[16];
    if (p.drop_thresh) {
      const uint32_t kc = ((uint32_t)(k0 >> 1) + 2u * (uint32_t)h) * kDropPairMul;   // pair index of kap(0, h)
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const uint32_t hsh = drop_pair_finish(drop_base, kc + (uint32_t)(4 * (i >> 2) + ((i & 3) >> 1)) * kDropPairMul);
        const float f0 = (hsh & 0xFFFFu) >= p.drop_thresh ? p.inv_keep : 0.f;
        const float f1 = (hsh >> 16) >= p.drop_thresh ? p.inv_keep : 0.f;
        ds[i] = pr[i] * (dp[i] * f0 - delta);
        ds[i + 1] = pr[i + 1] * (dp[i + 1] * f1 - delta);
        if (p.ho) { pr[i] = f0 != 0.f ? pr[i] : -pr[i]; pr[i + 1] = f1 != 0.f ? pr[i + 1] : -pr[i + 1]; }
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) ds[i] = pr[i] * (dp[i] - delta);
    }
    if (p.ho) {        // hand the tile to the dK/dV pass: P as bf16, sign = dropped -- stored one tile later (ho_flush)
#pragma unroll
      for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int j = 0; j < 8; ++j) hold_p[g][j] = (__bf16)pr[8 * g + j];
      hold_so = split_item ? (unsigned)(k0 >> 5) * kHoStrip : (unsigned)((q0 >> 5) * p.ho_slots + it) * kHoTile;
    }
    // dRel (unscaled; rel_gscale applied at the flush / in the stores)
    if (REL == 2) {                 // many keys of a row share an id: accumulate in the lane's dRel row
      // LDS float atomics retire about one lane every three cycles on this chip (a tile's 16 x 64 updates cost more
      // than the rest of the tile), so equal neighbours are merged first: along a row the ids come in runs -- a
      // direction id for every key left of the core window, the 2r + 1 core ids, another direction id to the
      // right -- and a lane's 16 keys are four groups of four consecutive ones.  Only the last element of a run
      // issues its (exec-masked) update: two or three lanes-worth of updates per lane and tile instead of 16.
      float run = ds[0] * p.rel_gscale;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const bool last = i == 15 || col4[i] != col4[i + 1];
        if (last) lds_add_f32(dtrow_addr + col4[i], run);
        if (i < 15) run = fmaf(ds[i + 1], p.rel_gscale, last ? 0.f : run);
      }
    } else if (HAS_REL) {
      if (one_id) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += ds[i];
        if (tc.far_neg) far_neg_acc += t; else far_pos_acc += t;
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int d = dbase + (i & 3) + 8 * (i >> 2);
          const bool neg = d <= -m, pos = (d >= m) && !neg;
          far_neg_acc += neg ? ds[i] : 0.f;
          far_pos_acc += pos ? ds[i] : 0.f;
          // no branch per element: a far element's store goes to the row's spare last column (zeroed after the loop);
          // sixteen exec-masked stores were 1.2 k of a mixed-id tile's 6 k cycles (profiles/r03_bwd_stamps.txt)
          dtrow[(neg | pos) ? dstride - 1 : m + d] = ds[i] * p.rel_gscale;
        }
      }
    }
    mma_xt(a0, a1, VTile<T>{}, xlds, ds, lane);   // dQ^T += K^T . dS^T   (gscale in the epilogue)
    wave_lds_sync();
    QSTAMP(4 + min(it, 7));
  }
  QSTAMP(12);
  if (p.ho && n_it > 0) ho_flush();
#pragma unroll
  for (int i = 0; i < 16; ++i) { a0[i] *= p.gscale; a1[i] *= p.gscale; }
  if (REL == 1) {
    const float fn = half_sum(far_neg_acc) * p.rel_gscale;
    const float fp = half_sum(far_pos_acc) * p.rel_gscale;
    wave_lds_sync();                  // (the spare column's last stores of both half-waves are behind this)
    if (h == 0) {
      dtrow[dstride - 1] = 0.f;       // the spare column that took the far elements' stores
      if (m == 0) dtrow[0] = fn + fp;
      else { dtrow[0] = fn; dtrow[2 * m] = fp; }
    }
  }
  wave_lds_sync();

  if (split_item) {
    const long slot = ((long)bn * p.n_gblk + gblk) * p.n_chunks + chunk;
    float* po = p.part_dq + slot * (32 * 64) + r * 64;
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      *reinterpret_cast<f32x4*>(po + 8 * gi + 4 * h) = f32x4{a0[4 * gi], a0[4 * gi + 1], a0[4 * gi + 2], a0[4 * gi + 3]};
      *reinterpret_cast<f32x4*>(po + 32 + 8 * gi + 4 * h) = f32x4{a1[4 * gi], a1[4 * gi + 1], a1[4 * gi + 2], a1[4 * gi + 3]};
    }
    float* pt = p.part_dtab + slot * (32 * Rp);
    for (int i = lane; i < 32 * Rp; i += 64) {
      const int rr = i / Rp, id = i - rr * Rp;
      const int col = REL == 2 ? id : tcol(1, m, id);
      pt[i] = col < dstride ? dtab[rr * dstride + col] : 0.f;
    }
    return;
  }

  if (HAS_REL) {
    // (1) dQ^T += E^T[d x column] . dRel^T[column x q], both in table-column order: the E image of the workgroup
    //     serves the transposed reads as it stands (columns that carry no relative id hold zero rows of E and
    //     zero dRel)
#pragma unroll
    for (int rb = 0; rb < Rp / 32; ++rb) {
      // (unconditional reads from a clamped column, then the select: sixteen exec-masked reads compiled to sixteen
      //  serialised LDS round trips, 2 k cycles of every workgroup's epilogue -- profiles/r03_bwd_stamps.txt)
      float vals[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) vals[i] = dtrow[min(rb * 32 + kap(i, h), dstride - 1)];
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(vals[i]));      // (or hipcc sinks each read back under its select)
#pragma unroll
      for (int i = 0; i < 16; ++i) vals[i] = rb * 32 + kap(i, h) < dstride ? vals[i] : 0.f;
      mma_xt(a0, a1, VTile<T>{}, elds + rb * 4096, vals, lane);
    }
  }
  if (q_ok && !(p.skip_global && is_global(p.pat, q))) {
    T* DQ = reinterpret_cast<T*>(p.dq) + (long)b * p.qs[0] + (long)q * p.qs[1] + (long)n * p.qs[2];
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      const int d = 8 * gi + 4 * h;
      bf16x4 x, y;
#pragma unroll
      for (int j = 0; j < 4; ++j) { x[j] = (__bf16)a0[4 * gi + j]; y[j] = (__bf16)a1[4 * gi + j]; }
      *reinterpret_cast<bf16x4*>(DQ + d) = x;
      *reinterpret_cast<bf16x4*>(DQ + 32 + d) = y;
    }
  }
  QSTAMP(13);
  if (!HAS_REL) return;

  // (2) the workgroup's share of dE^T[d x id] = Q^T . dRel and dbias[id] (one partial per workgroup for the
  //     fixed-order reduce K4c).  Every wave contracts ITS 32 query rows at once (its Q tile and dRel table are
  //     in its own LDS), parks the 64 x 32 result over its -- now dead -- tables, and after one workgroup
  //     barrier each wave adds a quarter of the ids over the (up to) four parked partials, in wave order, and
  //     stores it.  Before, wave 0 did all of this for the four waves behind the barrier: 14 of the kernel's
  //     93 us (tile-cap / no-dE ablations, profiles/r02 notes in DESIGN.md).
  if (n_it == 0) {             // (no tile visited: nothing was prefetched)
#pragma unroll
    for (int u = 0; u < 4; ++u) kt[u] = buf16(rq, (unsigned)(lane >> 3) * qs1b + (lane & 7) * 16, (unsigned)(q0 + 8 * u) * qs1b);
  }
  tile_to_lds(xlds, kt, lane);
  wave_lds_sync();
  // byte offset of the parked bias sums inside the wave's LDS: the V tile (dead by now).  (They used to sit at the
  // start of the Q tile for Rp = 64, where the first id block's sums overwrote row 0 of the tile the second block's
  // contraction still reads.)
  const int kParkBias = L::kTab + 32 * dstride * 4 + L::kTile;
#pragma unroll
  for (int rb = 0; rb < Rp / 32; ++rb) {
    const int id = rb * 32 + r;
    const int col = REL == 2 ? id : tcol(1, m, id);
    f32x16 e0 = {0}, e1 = {0};
    float bsum = 0.f;
    float vals[16];
    const float* dcol = dtab + 4 * h * dstride + min(col, dstride - 1);     // this lane's column, rows 4h + ...
    const bool col_ok = id < p.R && col < dstride;
#pragma unroll
    for (int i = 0; i < 16; ++i) vals[i] = dcol[((i & 3) + 8 * (i >> 2)) * dstride];   // unconditional reads, then the selects (as above)
#pragma unroll
    for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(vals[i]));
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int qq = q0 + kap(i, h);
      const bool use = col_ok && qq < p.S && !(p.skip_global && is_global(p.pat, qq));
      vals[i] = use ? vals[i] : 0.f;
      bsum += vals[i];
    }
    mma_xt(e0, e1, VTile<T>{}, xlds, vals, lane);
    bsum = half_sum(bsum);
    wave_lds_sync();               // this block's table and tile reads are done: the park may overwrite them
    float* park = reinterpret_cast<float*>(wl + rb * L::kTab) + r * 64;
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      *reinterpret_cast<f32x4*>(park + 8 * gi + 4 * h) = f32x4{e0[4 * gi], e0[4 * gi + 1], e0[4 * gi + 2], e0[4 * gi + 3]};
      *reinterpret_cast<f32x4*>(park + 32 + 8 * gi + 4 * h) = f32x4{e1[4 * gi], e1[4 * gi + 1], e1[4 * gi + 2], e1[4 * gi + 3]};
    }
    if (h == 0) reinterpret_cast<float*>(wl + kParkBias)[id] = bsum;
  }
  QSTAMP(14);
  __syncthreads();                 // waves past the end of the sequence have exited and are not waited for
  QSTAMP(15);
  {
    const int q0_wg = q0 - 32 * wave;
    float* pe = p.part_red + (long)band_wg * (Rp * 64 + Rp);
    const int n_live = min(4, (p.S - q0_wg + 31) >> 5);               // waves of this workgroup that exist
    const int d0 = (lane & 7) * 8;
    for (int g = wave; g < 4; g += n_live) {                          // id groups of 8, dealt over the live waves
      const int idl = 8 * g + (lane >> 3);                            // this lane: one id of the group, 8 head dims
#pragma unroll
      for (int rb = 0; rb < Rp / 32; ++rb) {
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
        float bs = 0.f;
        for (int w2 = 0; w2 < n_live; ++w2) {
          const unsigned char* wl2 = smem + kEImg + w2 * kWave;
          const float* src = reinterpret_cast<const float*>(wl2 + rb * L::kTab) + idl * 64 + d0;
          const f32x4 x0 = *reinterpret_cast<const f32x4*>(src), x1 = *reinterpret_cast<const f32x4*>(src + 4);
          s0 += x0; s1 += x1;
          if (lane < 8) bs += reinterpret_cast<const float*>(wl2 + kParkBias)[rb * 32 + 8 * g + lane];
        }
        float* row = pe + (long)(rb * 32 + idl) * 64 + d0;
        *reinterpret_cast<f32x4*>(row) = s0;
        *reinterpret_cast<f32x4*>(row + 4) = s1;
        if (lane < 8) pe[Rp * 64 + rb * 32 + 8 * g + lane] = bs;
      }
    }
  }
  QSTAMP(16);
}

// =========================================================================================
// dK, dV (lane = key, registers = query rows).
// =========================================================================================
template <int Rp, int REL>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_band_bf16_kernel(const BwdParams p) {
  using T = __bf16;
  using L = LeanLds<Rp>;
  constexpr bool HAS_REL = REL != 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  constexpr int kEImg = HAS_REL ? Rp * 128 : 0;
  unsigned char* elds = smem;
  unsigned char* wl = smem + kEImg + wave * L::kDkv;
  float* tab = reinterpret_cast<float*>(wl);
  unsigned char* qlds = wl + L::kTab;
  unsigned char* dolds = qlds + L::kTile;
  float* bias_ts = reinterpret_cast<float*>(dolds + L::kTile);
  float* rowc = bias_ts + Rp;            // per row of the current q tile: [0,32) lse*log2e, [32,64) delta,
                                         // [64,96) rel(clipped, d<=-m) - lse2, [96,128) rel(clipped, d>=m) - lse2
  int* lut = reinterpret_cast<int*>(smem + kEImg + 4 * L::kDkv);       // REL == 2: one per workgroup

  if (p.comb_in_next) {          // trailing blocks: the dQ combine of the global rows (one row per wave)
    const int per_bn0 = (p.n_chunks * p.n_gblk + 3) >> 2;
    const int n_main = p.n_band_blocks + per_bn0 * p.B * p.N;
    if ((int)blockIdx.x >= n_main) {
      const int pair = ((int)blockIdx.x - n_main) * 4 + wave;
      if (pair < p.pat.ng * p.B * p.N)
        dq_combine_row<T>(p, pair / p.pat.ng, pair % p.pat.ng, lane, reinterpret_cast<float*>(smem) + wave * 64,
                          [] { wave_lds_sync(); });
      return;
    }
  }
#ifdef MMT_STAMP
  long long* dbg = (p.dbg && (p.dbg_mode & 1) == 1 && blockIdx.x == 900) ? p.dbg + wave * 32 : nullptr;
#define KSTAMP(i) do { if (dbg && lane == 0) dbg[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define KSTAMP(i) do { } while (0)
#endif
  KSTAMP(0);
  const int n_tiles = (p.S + 31) >> 5, nkb = (p.S + 127) >> 7;
  const int per_bn = (p.n_chunks * p.n_gblk + 3) >> 2;
  int bn, k0, chunk = 0, gblk = 0, blk;
  plane_major_map(blockIdx.x, p.B * p.N, per_bn, nkb, bn, blk);
  const bool split_item = blk < per_bn;
  bool live = true;
  if (split_item) {
    const int item = blk * 4 + wave;
    live = item < p.n_chunks * p.n_gblk;
    gblk = live ? item / p.n_chunks : 0;
    chunk = item - gblk * p.n_chunks;
    k0 = p.pat.g0 + gblk * 32;
  } else {
    k0 = (blk - per_bn) * 128 + wave * 32;
    live = k0 < p.S;
  }
  const int b = bn / p.N, n = bn - b * p.N;
  const int k = k0 + r;
  const bool k_ok = k < p.S;
  const int valid_len = p.valid_len ? p.valid_len[b] : p.S;
  const int W = p.pat.radius, m = p.pat.m;
  const bool ignore_band = split_item;

  const unsigned qs1b = (unsigned)p.qs[1] * 2, ks1b = (unsigned)p.ks[1] * 2, vs1b = (unsigned)p.vs[1] * 2, os1b = (unsigned)p.os[1] * 2;
  const T* Qb = reinterpret_cast<const T*>(p.q) + (long)b * p.qs[0] + (long)n * p.qs[2];
  const T* Kb = reinterpret_cast<const T*>(p.k) + (long)b * p.ks[0] + (long)n * p.ks[2];
  const T* Vb = reinterpret_cast<const T*>(p.v) + (long)b * p.vs[0] + (long)n * p.vs[2];
  const T* DOb = reinterpret_cast<const T*>(p.dout) + (long)b * p.os[0] + (long)n * p.os[2];
  const auto rq = make_rsrc(Qb, (unsigned)(p.S - 1) * qs1b + 128);
  const auto rk = make_rsrc(Kb, (unsigned)(p.S - 1) * ks1b + 128);
  const auto rv = make_rsrc(Vb, (unsigned)(p.S - 1) * vs1b + 128);
  const auto rdo = make_rsrc(DOb, (unsigned)(p.S - 1) * os1b + 128);
  const float* lse_bn = p.lse + ((long)b * p.N + n) * p.S;
  const float* delta_bn = p.delta + ((long)b * p.N + n) * p.S;
  const float* relfar_bn = p.relfar + ((long)b * p.N + n) * p.S * 2;
  const unsigned voff_qc = (unsigned)(lane >> 3) * qs1b + (lane & 7) * 16;
  const unsigned voff_oc = (unsigned)(lane >> 3) * os1b + (lane & 7) * 16;

  TileWalkLean w;
  if (split_item) w.set_chunk(chunk * p.chunk_tiles, min(n_tiles, (chunk + 1) * p.chunk_tiles));
  else w.set_band(p.pat, live ? k0 : 0, p.S);
  // The (<= 8) global query ROWS outside this wave's band q tiles: a PEELED step before the walk instead of a sixth
  // q tile (registers 0..3 only; one load instruction each for their Q and dO rows) -- the mirror of the dQ pass's
  // peeled global keys.
  const bool peel = REL != 2 && (p.peel_gkeys & 2) && !split_item && live &&
                    !(p.pat.g0 >= w.b0 * 32 && p.pat.g0 + p.pat.ng - 1 <= (w.b0 + w.lenB) * 32 - 1);
  if (REL != 2 && (p.peel_gkeys & 2) && !split_item) { w.lenA = 0; w.lenC = 0; }
  const int n_it = live ? w.count() : 0;

  Frag<T> kf, vf;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    kf.v[s] = buf16(rk, (unsigned)r * ks1b + 64 * h + 16 * s, (unsigned)k0 * ks1b);
    vf.v[s] = buf16(rv, (unsigned)r * vs1b + 64 * h + 16 * s, (unsigned)k0 * vs1b);
  }
  bf16x8 qt[4], dot[4];
  // per-row constants of a q tile (one row per lane): lse, delta, clipped relative score -- fetched ONE TILE AHEAD with
  // the tile's Q / dO rows.  (They used to be loaded at the top of the tile they belong to, right behind the next
  // tile's prefetch: two dependent global round trips per tile, each behind a vmcnt(0) that also drained the
  // prefetch just issued -- the stamps' 6-10 k cycles per tile, profiles/r03_bwd_stamps.txt.)
  float rc_l = 0.f, rc_d = 0.f, rc_r = 0.f;
  auto load_rowc = [&](int q0t) {
    const int qq = min(q0t + r, p.S - 1);
    rc_l = lse_bn[qq];
    rc_d = delta_bn[qq];
    if (REL == 1) rc_r = relfar_bn[2 * qq + h];
  };
  {
    const unsigned q0 = (unsigned)w.at(0) * 32;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      qt[u] = buf16(rq, voff_qc, (q0 + 8 * u) * qs1b);
      dot[u] = buf16(rdo, voff_oc, (q0 + 8 * u) * os1b);
    }
    load_rowc((int)q0);
  }
  KSTAMP(1);
  if (HAS_REL) {           // E image after the item's loads are in flight (see the dQ kernel)
    stage_e_image<Rp, REL>(elds, p.emb, n, p.N, p.R, p.pat.m, threadIdx.x);
    if (REL == 2) build_lut2d<Rp>(lut, p.pat, p.R, threadIdx.x, 256);
    __syncthreads();
  }
  KSTAMP(2);
  if (!live) return;
  if (HAS_REL) {
    if (lane < Rp) {
      const int idc = REL == 2 ? lane : icol(m, lane);
      bias_ts[lane] = (p.bias && idc < p.R) ? (float)reinterpret_cast<const T*>(p.bias)[(long)idc * p.N + n] * p.tscale : 0.f;   // by column
    }
  }
  wave_lds_sync();
  // REL == 2: this lane's key on the patch grid
  const int xk2 = (int)__umulhi((unsigned)k, p.pat.magicP), yk2 = k - xk2 * p.pat.P;
  const int lut_addr = lds_addr(lut);
  const int lim2 = p.pat.r + 1, nlim2 = -lim2;

  f32x16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};
  const int tab_addr = (int)(unsigned)(size_t)(__attribute__((address_space(3))) float*)tab;
  const SeedPair sd = effective_seed(p.seed_lo, p.seed_hi, p.epoch);
  const uint32_t bn_seed = mix32(sd.lo ^ ((uint32_t)bn * 0x9E3779B9u)) + sd.hi;
  const uint32_t drop_kterm = (uint32_t)(k >> 1) * kDropPairMul, drop_ksh = (k & 1) ? 16u : 0u;   // this lane's key

  if (peel) {
    // rows g0 .. g0 + 7 of Q and dO (tile shape; rows past the end read as zeros) and their row constants
    const bf16x8 qg = buf16(rq, voff_qc, (unsigned)p.pat.g0 * qs1b);
    const bf16x8 dog = buf16(rdo, voff_oc, (unsigned)p.pat.g0 * os1b);
    const int qq = min(p.pat.g0 + r, p.S - 1);
    const float g_l2 = lse_bn[qq] * kLog2e, g_dl = delta_bn[qq];
    const float g_rf = REL == 1 ? relfar_bn[2 * qq + h] : 0.f;
    {
      const int row = lane >> 3, ch = lane & 7;
      const int off = row * 128 + ((((ch >> 2) ^ ((row >> 1) & 1))) << 6) + (ch & 3) * 16;
      const bf16x8 z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
      *reinterpret_cast<bf16x8*>(qlds + off) = qg;           // rows 0..7
      *reinterpret_cast<bf16x8*>(dolds + off) = dog;
      *reinterpret_cast<bf16x8*>(qlds + 1024 + off) = z;     // rows 8..15: P and dS are 0 there, 0 x garbage is not
      *reinterpret_cast<bf16x8*>(dolds + 1024 + off) = z;
      rowc[lane] = h == 0 ? g_l2 : g_dl;
      if (REL == 1) rowc[64 + lane] = g_rf - g_l2;
    }
    wave_lds_sync();
    const int lo_q = w.b0 * 32, hi_q = (w.b0 + w.lenB) * 32 - 1;          // q rows of this wave's band tiles
    const int qg0 = p.pat.g0, n_here = p.pat.ng;
    Frag<T> qgf, dogf;
    {
      const int rr = r & 7;
      const unsigned char* qrow = qlds + rr * 128 + ((h ^ ((rr >> 1) & 1)) << 6);
      const unsigned char* drow = dolds + rr * 128 + ((h ^ ((rr >> 1) & 1)) << 6);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        qgf.v[s] = *reinterpret_cast<const bf16x8*>(qrow + s * 16);
        dogf.v[s] = *reinterpret_cast<const bf16x8*>(drow + s * 16);
      }
    }
    f32x16 c = {0}, dp = {0};
    c = mma_rows(qgf, kf, c);            // S  [q x key]: registers 0..3 = global rows 4h + i
    dp = mma_rows(dogf, vf, dp);         // dP [q x key]
    // a peeled row lies beyond the radius (>= max_dist) from every key of this wave: clipped id, by the side it is on
    const bool kv = k < valid_len;
    float pr[16], g[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = i + 4 * h, qrow_abs = qg0 + row;
      const bool present = row < n_here && !(qrow_abs >= lo_q && qrow_abs <= hi_q) && k_ok;
      const float rl = HAS_REL ? rowc[(k < qrow_abs ? 64 : 96) + row] : -rowc[row];     // key < query: d <= -m
      float sc = fmaf(c[i], p.sscale, rl);
      sc = (kv == (qrow_abs < valid_len)) ? sc : sc + p.mask_add;
      float pv = present ? __builtin_amdgcn_exp2f(sc) : 0.f;
      float df = 1.f;
      if (p.drop_thresh) {
        const uint32_t hsh = drop_pair_finish(bn_seed + (uint32_t)qrow_abs * 0x85EBCA6Bu, drop_kterm);
        df = ((hsh >> drop_ksh) & 0xFFFFu) >= p.drop_thresh ? p.inv_keep : 0.f;
      }
      g[i] = pv * (dp[i] * df - rowc[32 + row]);
      pr[i] = pv * df;
    }
#pragma unroll
    for (int i = 4; i < 16; ++i) { pr[i] = 0.f; g[i] = 0.f; }
    {   // dV^T += dO_g^T . P and dK^T += Q_g^T . dS over rows 0..15 of the images (mma_xt's first q step)
      const int li = lane & 15, cb = (lane >> 4) & 1;
      bf16x8 pf, gf;
#pragma unroll
      for (int j = 0; j < 8; ++j) { pf[j] = (__bf16)pr[j]; gf[j] = (__bf16)g[j]; }
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const int row = 4 * h + (li >> 2);
        const int within = 32 * cb + 8 * (li & 3);
        const int off0 = row * 128 + ((db ^ ((row >> 1) & 1)) << 6) + within;
        const int row1 = row + 8;
        const int off1 = row1 * 128 + ((db ^ ((row1 >> 1) & 1)) << 6) + within;
        auto tr = [&](const unsigned char* img) {
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + off0));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + off1));
          bf16x8 xf;
          bf16x4 lo4 = __builtin_bit_cast(bf16x4, lo), hi4 = __builtin_bit_cast(bf16x4, hi);
#pragma unroll
          for (int j = 0; j < 4; ++j) { xf[j] = lo4[j]; xf[4 + j] = hi4[j]; }
          return xf;
        };
        const bf16x8 dof_t = tr(dolds), qf_t = tr(qlds);
        if (db == 0) { dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof_t, pf, dv0, 0, 0, 0); dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf_t, gf, dk0, 0, 0, 0); }
        else { dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof_t, pf, dv1, 0, 0, 0); dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf_t, gf, dk1, 0, 0, 0); }
      }
    }
    wave_lds_sync();                 // the first band tile overwrites the images and the row constants
  }

  for (int it = 0; it < n_it; ++it) {
    const int q0 = w.at(it) * 32;
    // 4h as a value hipcc must treat as loop-variant: every per-element LDS address below is then
    // "one base per tile + a literal" -- left to itself the compiler hoisted ~40 lane-dependent addresses
    // (row * stride, permuted table columns, hash constants) out of the loop, spilled them, and reloaded
    // them from scratch one by one behind vmcnt(0) waits that also drained the Q/dO prefetch
    int h4 = 4 * h;
    asm volatile("" : "+v"(h4));
    tile_to_lds(qlds, qt, lane);
    tile_to_lds(dolds, dot, lane);
    // per-row constants of this q tile (one row per lane) -> LDS, from the registers filled one tile ago
    {
      const float l2 = rc_l * kLog2e;
      rowc[lane] = h == 0 ? l2 : rc_d;
      if (REL == 1) rowc[64 + lane] = rc_r - l2;
    }
    if (it + 1 < n_it) {
      const unsigned q1 = (unsigned)w.at(it + 1) * 32;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        qt[u] = buf16(rq, voff_qc, (q1 + 8 * u) * qs1b);
        dot[u] = buf16(rdo, voff_oc, (q1 + 8 * u) * os1b);
      }
      load_rowc((int)q1);
    }
    const bool no_gq = p.pat.ng == 0 || q0 + 31 < p.pat.g0 || q0 >= p.pat.g0 + p.pat.ng;
    const TileClass tc = classify_tile(q0, k0, p.S, valid_len, W, m, ignore_band, no_gq);
    const bool one_id = REL != 2 && (!HAS_REL || tc.far_neg || tc.far_pos);
    wave_lds_sync();
    Frag<T> qf;
    frag_from_tile(qf, qlds, lane);
    if (HAS_REL && !one_id) {           // mixed ids: T rows = this q tile, with -lse2[row] folded in
      const float nl = -rowc[r];
      if (REL == 2) tab[r * kTStride(Rp) + kZeroCol(Rp)] = nl;
#pragma unroll
      for (int rb = 0; rb < Rp / 32; ++rb) {
        Frag<T> ef;                     // E rows of the table columns, from the workgroup's LDS image
        frag_from_tile(ef, elds + rb * 4096, lane);
        f32x16 c = {0};
        c = mma_rows(ef, qf, c);
        // the sixteen bias values first, then the sixteen stores: interleaved, hipcc kept every read behind the store
        // before it (it cannot tell the two LDS arrays apart) -- eight serialised LDS round trips per rebuilt tile
        float bv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) bv[i] = bias_ts[rb * 32 + (i & 3) + 8 * (i >> 2) + h4];
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(bv[i]));
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int col = rb * 32 + (i & 3) + 8 * (i >> 2) + h4;
          tab[r * kTStride(Rp) + col] = fmaf(c[i], p.tscale, bv[i]) + nl;
        }
      }
      wave_lds_sync();
    }
    f32x16 c = {0}, dp = {0};
    c = mma_rows(qf, kf, c);      // S  [q x key]
    {
      Frag<T> dof;
      frag_from_tile(dof, dolds, lane);
      dp = mma_rows(dof, vf, dp); // dP [q x key]
    }
    const int dbase = k - q0 - h4;                     // d_i = dbase - ci
    const float* relrow = rowc + (tc.far_neg ? 64 : 96);   // clipped rel - lse2, per row

    float pr[16];
    if (REL == 2) {
      // table value (rel - lse2 of the row) of each element, then the mask by tile class
      float sc[16];
      const int rowbase = tab_addr + h4 * (kTStride(Rp) * 4);
      if (q0 + 31 < p.pat.I && k0 + 31 < p.pat.I && p.pat.P >= 32) {     // image x image: look-up table
        const Ids2dTile t2 = ids2d_tile<-1>(p.pat, lut_addr, q0 + h4, xk2, yk2);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int ci = (i & 3) + 8 * (i >> 2);
          sc[i] = fmaf(c[i], p.sscale, *(lds_cfp)(size_t)(unsigned)(rowbase + ci * (kTStride(Rp) * 4) + ids2d_col4<-1>(t2, ci, nlim2, lim2)));
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int ci = (i & 3) + 8 * (i >> 2);
          sc[i] = fmaf(c[i], p.sscale, *(lds_cfp)(size_t)(unsigned)(rowbase + ci * (kTStride(Rp) * 4) + 4 * col2d<Rp>(p.pat, p.R, q0 + h4 + ci, k)));
        }
      }
      if (tc.plain) {
#pragma unroll
        for (int i = 0; i < 16; ++i) pr[i] = __builtin_amdgcn_exp2f(sc[i]);
      } else if (tc.edge) {
        const unsigned W2 = 2u * (unsigned)W;
#pragma unroll
        for (int i = 0; i < 16; ++i)
          pr[i] = __builtin_amdgcn_exp2f(sc[i] + ((unsigned)(dbase - ((i & 3) + 8 * (i >> 2)) + W) <= W2 ? 0.f : p.mask_add));
      } else if (tc.outside) {
        const unsigned gb = (unsigned)(q0 + h4 - p.pat.g0), ng = (unsigned)p.pat.ng;
#pragma unroll
        for (int i = 0; i < 16; ++i)
          pr[i] = __builtin_amdgcn_exp2f(sc[i] + (gb + (unsigned)((i & 3) + 8 * (i >> 2)) < ng ? 0.f : p.mask_add));
      } else {
        const bool kv = k < valid_len;
        const unsigned W2 = 2u * (unsigned)W;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int ci = (i & 3) + 8 * (i >> 2);
          const int qq = q0 + ci + h4, d = dbase - ci;
          const bool near = ignore_band | ((unsigned)(d + W) <= W2);
          const bool gq = (unsigned)(qq - p.pat.g0) < (unsigned)p.pat.ng;
          const bool seg = kv == (qq < valid_len);
          const bool keep = (int)seg & ((int)near | (int)gq);
          pr[i] = (qq < p.S && k_ok) ? __builtin_amdgcn_exp2f(keep ? sc[i] : sc[i] + p.mask_add) : 0.f;
        }
      }
    } else if (tc.plain && one_id) {                      // class A
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + h4;      // q row inside the tile
        const float rl = HAS_REL ? relrow[row] : -rowc[row];   // rel - lse2
        pr[i] = __builtin_amdgcn_exp2f(fmaf(c[i], p.sscale, rl));
      }
    } else if (tc.plain) {                                // class B
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ci = (i & 3) + 8 * (i >> 2), row = ci + h4;
        const int lo = tab_addr + row * kTStride(Rp) * 4;
        const int a = med3i(lo + 4 * (m + dbase - ci), lo, lo + 8 * m);
        pr[i] = __builtin_amdgcn_exp2f(fmaf(c[i], p.sscale, *(lds_cfp)(size_t)(unsigned)a));
      }
    } else if (tc.edge && one_id) {                       // class D
      const unsigned W2 = 2u * (unsigned)W;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ci = (i & 3) + 8 * (i >> 2), row = ci + h4;
        const float rl = HAS_REL ? relrow[row] : -rowc[row];
        const unsigned dd = (unsigned)(dbase - ci + W);
        pr[i] = __builtin_amdgcn_exp2f(fmaf(c[i], p.sscale, rl) + (dd <= W2 ? 0.f : p.mask_add));
      }
    } else if (tc.outside && one_id) {                    // class G: only the tile's global query rows see these keys
      const unsigned gb = (unsigned)(q0 + h4 - p.pat.g0), ng = (unsigned)p.pat.ng;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ci = (i & 3) + 8 * (i >> 2), row = ci + h4;
        const float rl = HAS_REL ? relrow[row] : -rowc[row];
        pr[i] = __builtin_amdgcn_exp2f(fmaf(c[i], p.sscale, rl) + (gb + (unsigned)ci < ng ? 0.f : p.mask_add));
      }
    } else {                                              // class C
      const bool kv = k < valid_len;
      const unsigned W2 = 2u * (unsigned)W;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ci = (i & 3) + 8 * (i >> 2), row = ci + h4;
        const int qq = q0 + row, d = dbase - ci;
        const bool near = ignore_band | ((unsigned)(d + W) <= W2);
        const bool gq = (unsigned)(qq - p.pat.g0) < (unsigned)p.pat.ng;
        const bool seg = kv == (qq < valid_len);
        const bool keep = (int)seg & ((int)near | (int)gq);
        const float rl = !HAS_REL ? -rowc[row] : (one_id ? relrow[row] : tab[row * kTStride(Rp) + min(max(d, -m), m) + m]);
        float s = fmaf(c[i], p.sscale, rl);
        s = keep ? s : s + p.mask_add;
        pr[i] = (qq < p.S && k_ok) ? __builtin_amdgcn_exp2f(s) : 0.f;
      }
    }
    float g[16];
    if (p.drop_thresh) {
      // row base of query q0 + ci + 4h = tb + ci * C: one multiply per tile and a literal per element.  tb is
      // made opaque so that hipcc does not re-associate it into 16 loop-invariant (ci + 4h) * C registers
      // (it did, and spilled them: 16 scratch reloads per tile inside this loop)
      const uint32_t tb = bn_seed + (uint32_t)(q0 + h4) * 0x85EBCA6Bu;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const uint32_t hsh = drop_pair_finish(tb + (uint32_t)((i & 3) + 8 * (i >> 2)) * 0x85EBCA6Bu, drop_kterm);
        const float df = ((hsh >> drop_ksh) & 0xFFFFu) >= p.drop_thresh ? p.inv_keep : 0.f;
        g[i] = pr[i] * (dp[i] * df - rowc[32 + (i & 3) + 8 * (i >> 2) + h4]);
        pr[i] *= df;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) g[i] = pr[i] * (dp[i] - rowc[32 + (i & 3) + 8 * (i >> 2) + h4]);
    }
    mma_xt(dv0, dv1, VTile<T>{}, dolds, pr, lane);   // dV^T[d x key] += dO^T[d x q] . P[q x key]
    mma_xt(dk0, dk1, VTile<T>{}, qlds, g, lane);     // dK^T[d x key] += Q^T[d x q] . dS[q x key]
    wave_lds_sync();
    KSTAMP(4 + min(it, 7));
  }
  KSTAMP(12);
#pragma unroll
  for (int i = 0; i < 16; ++i) { dk0[i] *= p.gscale; dk1[i] *= p.gscale; }

  if (split_item) {
    const long slot = ((long)bn * p.n_gblk + gblk) * p.dkv_slots + chunk;
    float* pk = p.part_dkv + slot * (2 * 32 * 64) + r * 64;
    float* pv2 = pk + 32 * 64;
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      *reinterpret_cast<f32x4*>(pk + 8 * gi + 4 * h) = f32x4{dk0[4 * gi], dk0[4 * gi + 1], dk0[4 * gi + 2], dk0[4 * gi + 3]};
      *reinterpret_cast<f32x4*>(pk + 32 + 8 * gi + 4 * h) = f32x4{dk1[4 * gi], dk1[4 * gi + 1], dk1[4 * gi + 2], dk1[4 * gi + 3]};
      *reinterpret_cast<f32x4*>(pv2 + 8 * gi + 4 * h) = f32x4{dv0[4 * gi], dv0[4 * gi + 1], dv0[4 * gi + 2], dv0[4 * gi + 3]};
      *reinterpret_cast<f32x4*>(pv2 + 32 + 8 * gi + 4 * h) = f32x4{dv1[4 * gi], dv1[4 * gi + 1], dv1[4 * gi + 2], dv1[4 * gi + 3]};
    }
    return;
  }
  if (!k_ok || (p.skip_global && is_global(p.pat, k))) return;
  T* DK = reinterpret_cast<T*>(p.dk) + (long)b * p.ks[0] + (long)k * p.ks[1] + (long)n * p.ks[2];
  T* DV = reinterpret_cast<T*>(p.dv) + (long)b * p.vs[0] + (long)k * p.vs[1] + (long)n * p.vs[2];
#pragma unroll
  for (int gi = 0; gi < 4; ++gi) {
    const int d = 8 * gi + 4 * h;
    bf16x4 x, y, z, u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x[j] = (__bf16)dk0[4 * gi + j]; y[j] = (__bf16)dk1[4 * gi + j];
      z[j] = (__bf16)dv0[4 * gi + j]; u[j] = (__bf16)dv1[4 * gi + j];
    }
    *reinterpret_cast<bf16x4*>(DK + d) = x; *reinterpret_cast<bf16x4*>(DK + 32 + d) = y;
    *reinterpret_cast<bf16x4*>(DV + d) = z; *reinterpret_cast<bf16x4*>(DV + 32 + d) = u;
  }
  KSTAMP(13);
}

// =========================================================================================
// dK, dV from the hand-over (lane = key, registers = query rows).  Per (q tile, key tile) pair the dQ pass left the
// probabilities as a bf16 image, sign = dropped; here dP = dO . V^T is one MFMA group, P' = max(x, 0) (scaled by
// 1 / keep_prob once, at the end), dS = |x| (dP' - delta): no scores, no exponentials, no relative ids, no masks, no
// keep hashes.
//   band key wave  : its (up to ho_slots) band q tiles + the 8-row image of the global query rows for its keys;
//                    the rows of its global keys go to partial slot n_chunks of part_dkv instead of dK / dV
//   global-key item: the strips (32 rows x 8 keys) of the q blocks of its chunk
// Rows of global queries inside band tiles / strips are zeroed here (the dQ pass computes and discards them: those
// pairs belong to the global-row items).
// The pass is a stream of independent loads with little arithmetic behind them, so what it needs is bytes in flight:
//   WIN = false: every wave stages its own Q / dO tile per step, one step ahead;
//   WIN = true : (ho_slots <= 5) the band workgroup stages the Q / dO rows of its whole q window -- at most 8 tiles,
//                64 KiB -- ONCE, and every wave requests all of its (<= 5) images before anything else.
// =========================================================================================
typedef short s16x8 __attribute__((ext_vector_type(8)));
constexpr int kHoWaveLds = 2 * 4096 + 2048 + 128;       // per-wave form: Q tile, dO tile, P image, delta of the tile's rows
constexpr int kHoWinLds = 16 * 4096 + 1024 + 4 * 3072;  // window form: Q window, dO window, delta of its rows, 3 KiB per wave

template <bool WIN>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_ho_kernel(const BwdParams p) {
  using T = __bf16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5, li = lane & 15, cb = (lane >> 4) & 1;

  if (p.comb_in_next) {          // trailing blocks: the dQ combine of the global rows (one row per wave)
    const int per_bn0 = (p.n_chunks * p.n_gblk + 3) >> 2;
    const int n_main = p.n_band_blocks + per_bn0 * p.B * p.N;
    if ((int)blockIdx.x >= n_main) {
      const int pair = ((int)blockIdx.x - n_main) * 4 + wave;
      if (pair < p.pat.ng * p.B * p.N)
        dq_combine_row<T>(p, pair / p.pat.ng, pair % p.pat.ng, lane, reinterpret_cast<float*>(smem) + wave * 64,
                          [] { wave_lds_sync(); });
      return;
    }
  }
  const int n_tiles = (p.S + 31) >> 5, nkb = (p.S + 127) >> 7;
  const int per_bn = (p.n_chunks * p.n_gblk + 3) >> 2;
  int bn, k0, chunk = 0, blk;
  plane_major_map(blockIdx.x, p.B * p.N, per_bn, nkb, bn, blk);
  const bool split_item = blk < per_bn;
  bool live = true;
  if (split_item) {              // (one global block with the hand-over: at most 8 global tokens)
    chunk = blk * 4 + wave;
    live = chunk < p.n_chunks;
    k0 = p.pat.g0;
  } else {
    k0 = (blk - per_bn) * 128 + wave * 32;
    live = k0 < p.S;
  }
  const bool win_item = WIN && !split_item;
  if (!live && !win_item) return;
#ifdef MMT_STAMP
  long long* dbg = (p.dbg && (p.dbg_mode & 1) == 1 && blockIdx.x == 900) ? p.dbg + wave * 32 : nullptr;
#define HSTAMP(i) do { if (dbg && lane == 0) dbg[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define HSTAMP(i) do { } while (0)
#endif
  HSTAMP(0);
  const int b = bn / p.N, n = bn - b * p.N;
  const int k = k0 + r;
  const int W = p.pat.radius;
  const int g0 = p.pat.g0, ng = p.n_gblk > 0 ? p.pat.ng : 0;     // global tokens handled by split items (0: none)
  const float ik = p.drop_thresh ? p.inv_keep : 1.f;

  const unsigned qs1b = (unsigned)p.qs[1] * 2, vs1b = (unsigned)p.vs[1] * 2, os1b = (unsigned)p.os[1] * 2;
  const T* Qb = reinterpret_cast<const T*>(p.q) + (long)b * p.qs[0] + (long)n * p.qs[2];
  const T* Vb = reinterpret_cast<const T*>(p.v) + (long)b * p.vs[0] + (long)n * p.vs[2];
  const T* DOb = reinterpret_cast<const T*>(p.dout) + (long)b * p.os[0] + (long)n * p.os[2];
  const auto rq = make_rsrc(Qb, (unsigned)(p.S - 1) * qs1b + 128);
  const auto rv = make_rsrc(Vb, (unsigned)(p.S - 1) * vs1b + 128);
  const auto rdo = make_rsrc(DOb, (unsigned)(p.S - 1) * os1b + 128);
  const float* delta_bn = p.delta + ((long)b * p.N + n) * p.S;
  const unsigned voff_qc = (unsigned)(lane >> 3) * qs1b + (lane & 7) * 16;
  const unsigned voff_oc = (unsigned)(lane >> 3) * os1b + (lane & 7) * 16;

  int t0 = 0, n_it = 0;          // q tiles of this item
  if (split_item) { t0 = chunk * p.chunk_tiles; n_it = min(n_tiles, t0 + p.chunk_tiles) - t0; }
  else if (live) { t0 = max(k0 - W, 0) >> 5; n_it = (min(k0 + 31 + W, p.S - 1) >> 5) - t0 + 1; }
  const int kb = k0 >> 5;
  const unsigned char* band_base = ho_band_base(p, n_tiles, bn);
  auto band_tile = [&](int qb) {         // this key block's image in q block qb
    const int slot = kb - (max(qb * 32 - W, 0) >> 5);
    return band_base + (size_t)(qb * p.ho_slots + slot) * kHoTile + lane * 16;
  };
  Frag<T> vf;                    // this wave's V rows (rows past the end read as zeros)
#pragma unroll
  for (int s = 0; s < 4; ++s) vf.v[s] = buf16(rv, (unsigned)r * vs1b + 64 * h + 16 * s, (unsigned)k0 * vs1b);

  auto tr = [&](const unsigned char* a) {
    return __builtin_bit_cast(bf16x4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a));
  };
  auto join = [&](bf16x4 lo, bf16x4 hi) {
    bf16x8 x;
#pragma unroll
    for (int j = 0; j < 4; ++j) { x[j] = lo[j]; x[4 + j] = hi[j]; }
    return x;
  };
  const bf16x4 z4 = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
  f32x16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};
  // one 16-row step of both products: dV^T[d x key] += dO^T[d x q] . P'[q x key], dK^T += Q^T . dS, A operands by
  // transposed reads of the tile images (rows 16 s + 4 h + j, + 8: the order of the B registers).  hi = false: rows
  // 8..15 of the step do not exist (the 8 global rows): zeros instead of the second read.
  auto step = [&](const unsigned char* qimg, const unsigned char* doimg, int s, const bf16x8& pf, const bf16x8& gf, bool hi) {
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int row = 16 * s + 4 * h + (li >> 2), row1 = row + 8;
      const int within = 32 * cb + 8 * (li & 3);
      const int off0 = row * 128 + ((db ^ ((row >> 1) & 1)) << 6) + within;
      const int off1 = row1 * 128 + ((db ^ ((row1 >> 1) & 1)) << 6) + within;
      const bf16x8 dof_t = join(tr(doimg + off0), hi ? tr(doimg + off1) : z4);
      const bf16x8 qf_t = join(tr(qimg + off0), hi ? tr(qimg + off1) : z4);
      if (db == 0) { dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof_t, pf, dv0, 0, 0, 0); dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf_t, gf, dk0, 0, 0, 0); }
      else { dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof_t, pf, dv1, 0, 0, 0); dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf_t, gf, dk1, 0, 0, 0); }
    }
  };
  // the signed-P registers of step s (element j <-> row 16 s + 4 h + (j & 3) + 8 (j >> 2), the accumulator order) from a
  // band image / a strip in LDS, rows of global queries zeroed
  auto x_regs = [&](const unsigned char* plds, int s, int q0, bool strip) {
    const int qr = 16 * s + 4 * h + (li >> 2);
    bf16x8 px;
    if (strip) {          // [key half][row][4 keys]: lanes past key 7 read copies (their columns are not used)
      const int off = (li & 1) * 256 + qr * 8;
      px = join(tr(plds + off), tr(plds + off + 64));
    } else {
      const int bk = 2 * cb + (li & 1), sub = ((li >> 1) & 1) * 8;      // chunk li & 3 = keys 16 cb + 4 (li & 3) .. = block bk, half sub
      px = join(tr(plds + ho_unit_off(bk, qr) + sub), tr(plds + ho_unit_off(bk, qr + 8) + sub));
    }
    if (ng > 0 && q0 + 31 >= g0 && q0 < g0 + ng) {       // rows of global queries in this tile
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int qq = q0 + 16 * s + 4 * h + (j & 3) + 8 * (j >> 2);
        if ((unsigned)(qq - g0) < (unsigned)ng) px[j] = (__bf16)0.f;
      }
    }
    return px;
  };
  // P' (dropped -> 0; the 1 / keep_prob factor is applied to dV once, at the end) and dS = |x| (dP' - delta) as the B
  // registers of a step; dl = the delta of the step's rows, dp8 = its dP registers
  auto finish = [&](const bf16x8& px, const float (&dp8)[8], const float (&dl)[8], bf16x8& pf, bf16x8& gf) {
    pf = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, px), (s16x8)(short)0));   // bf16 >= 0 <=> its bits as int16 >= 0
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = (float)px[j];
      float t = fmaf(dp8[j], ik, -dl[j]);
      t = x < 0.f ? -dl[j] : t;
      gf[j] = (__bf16)(__builtin_fabsf(x) * t);
    }
  };
  // a whole q tile: images qimg / doimg, P image (or strip) at plds, the rows' delta at drow (LDS floats)
  auto tile = [&](const unsigned char* qimg, const unsigned char* doimg, const unsigned char* plds, const float* drow, int q0, bool strip) {
    Frag<T> dof;
    frag_from_tile(dof, doimg, lane);
    f32x16 dp = {0};
    dp = mma_rows(dof, vf, dp);          // dP [q x key]
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bf16x8 px = x_regs(plds, s, q0, strip);
      const f32x4 dlo = *reinterpret_cast<const f32x4*>(drow + 16 * s + 4 * h), dhi = *reinterpret_cast<const f32x4*>(drow + 16 * s + 8 + 4 * h);
      float dp8[8], dl[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) { dp8[j] = dp[8 * s + j]; dl[j] = j < 4 ? dlo[j] : dhi[j - 4]; }
      bf16x8 pf, gf;
      finish(px, dp8, dl, pf, gf);
      step(qimg, doimg, s, pf, gf, true);
    }
  };
  // The global query rows against this wave's keys: Q_g / dO_g rows 0..7 (tile shape) and the 8-row image the
  // global-row items of the dQ pass left for this key tile ([block 4][row 8][16 bytes]; rows past the last global token
  // were not written: zeros here).  `area`: 2.5 KiB of this wave's LDS.
  auto global_rows = [&](unsigned char* area) {
    const bf16x8 qg = buf16(rq, voff_qc, (unsigned)g0 * qs1b);
    const bf16x8 dog = buf16(rdo, voff_oc, (unsigned)g0 * os1b);
    const unsigned char* gp = ho_grow_base(p, n_tiles, bn) + (size_t)kb * kHoStrip + lane * 8;
    const bool gl = ((lane >> 1) & 7) < ng;
    const bf16x4 gp4 = gl ? *reinterpret_cast<const bf16x4*>(gp) : z4;
    float dl[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { dl[j] = delta_bn[min(g0 + 4 * h + j, p.S - 1)]; dl[4 + j] = 0.f; }
    const int row = lane >> 3, ch = lane & 7;
    const int off = row * 128 + ((((ch >> 2) ^ ((row >> 1) & 1))) << 6) + (ch & 3) * 16;
    *reinterpret_cast<bf16x8*>(area + off) = qg;
    *reinterpret_cast<bf16x8*>(area + 1024 + off) = dog;
    *reinterpret_cast<bf16x4*>(area + 2048 + lane * 8) = gp4;
    wave_lds_sync();
    Frag<T> dogf;
    {
      const int rr = r & 7;
      const unsigned char* drow = area + 1024 + rr * 128 + ((h ^ ((rr >> 1) & 1)) << 6);
#pragma unroll
      for (int s = 0; s < 4; ++s) dogf.v[s] = *reinterpret_cast<const bf16x8*>(drow + s * 16);
    }
    f32x16 dp = {0};
    dp = mma_rows(dogf, vf, dp);         // registers 0..3 = global rows 4 h + i
    const int boff = (2 * cb + (li & 1)) * 128 + (4 * h + (li >> 2)) * 16 + ((li >> 1) & 1) * 8;    // chunk li & 3 = keys 16 cb + 4 (li & 3) ..
    const bf16x8 px = join(tr(area + 2048 + boff), z4);
    float dp8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) dp8[j] = j < 4 ? dp[j] : 0.f;
    bf16x8 pf, gf;
    finish(px, dp8, dl, pf, gf);
    step(area, area + 1024, 0, pf, gf, false);
    wave_lds_sync();
  };

  if (win_item) {
    // ---- band workgroup, window form ----
    const int k0wg = (blk - per_bn) * 128;
    const int t0w = max(k0wg - W, 0) >> 5;
    const int n_win = (min(k0wg + 127 + W, p.S - 1) >> 5) - t0w + 1;          // <= 8 (launcher)
    unsigned char* qwin = smem;
    unsigned char* dowin = smem + 8 * 4096;
    float* dwin = reinterpret_cast<float*>(smem + 16 * 4096);
    unsigned char* area = smem + 16 * 4096 + 1024 + wave * 3072;
    bf16x8 pt[5][2];
#pragma unroll
    for (int it = 0; it < 5; ++it)
      if (it < n_it) {
        const unsigned char* tp = band_tile(t0 + it);
        pt[it][0] = *reinterpret_cast<const bf16x8*>(tp); pt[it][1] = *reinterpret_cast<const bf16x8*>(tp + 1024);
      }
    {
      const int tid = threadIdx.x, row = tid >> 3, ch = tid & 7;
      const int dst = row * 128 + ((((ch >> 2) ^ ((row >> 1) & 1))) << 6) + (ch & 3) * 16;
      bf16x8 wq[8], wd[8];
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (t < n_win) {
          wq[t] = buf16(rq, (unsigned)row * qs1b + ch * 16, (unsigned)(t0w + t) * 32 * qs1b);
          wd[t] = buf16(rdo, (unsigned)row * os1b + ch * 16, (unsigned)(t0w + t) * 32 * os1b);
        }
      const float dv = delta_bn[min(t0w * 32 + tid, p.S - 1)];
#pragma unroll
      for (int t = 0; t < 8; ++t)
        if (t < n_win) {
          *reinterpret_cast<bf16x8*>(qwin + t * 4096 + dst) = wq[t];
          *reinterpret_cast<bf16x8*>(dowin + t * 4096 + dst) = wd[t];
        }
      HSTAMP(1);
      dwin[tid] = dv;
    }
    __syncthreads();
    HSTAMP(2);
    if (!live) return;
    if (ng > 0) global_rows(area);
    HSTAMP(3);
#pragma unroll
    for (int it = 0; it < 5; ++it)
      if (it < n_it) {
        const int qb = t0 + it, ws = qb - t0w;
        *reinterpret_cast<bf16x8*>(area + lane * 16) = pt[it][0]; *reinterpret_cast<bf16x8*>(area + 1024 + lane * 16) = pt[it][1];
        wave_lds_sync();
        tile(qwin + ws * 4096, dowin + ws * 4096, area, dwin + ws * 32, qb * 32, false);
        wave_lds_sync();
        HSTAMP(4 + it);
      }
    HSTAMP(12);
  } else {
    // ---- per-wave form: band key waves (WIN = false) and the global-key items ----
    unsigned char* qlds = smem + wave * kHoWaveLds;
    unsigned char* dolds = qlds + 4096;
    unsigned char* plds = dolds + 4096;
    float* drow = reinterpret_cast<float*>(plds + 2048);
    // one q tile's operands: Q / dO rows in tile shape, the image as it lies in memory, the rows' delta -- one tile ahead.
    // (Two tiles ahead, in two register sets, was slower: 58 -> 66 us per call.  The pass is not waiting for single
    // round trips; with more bytes in flight per wave the queues in front of L2 only get longer.)
    bf16x8 qt[4], dot[4], pt[2];
    float rc_d = 0.f;
    auto fetch = [&](int qb) {
      const unsigned q0 = (unsigned)qb * 32;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        qt[u] = buf16(rq, voff_qc, (q0 + 8 * u) * qs1b);
        dot[u] = buf16(rdo, voff_oc, (q0 + 8 * u) * os1b);
      }
      rc_d = delta_bn[min((int)q0 + r, p.S - 1)];
      if (split_item) {            // strip: 512 bytes, 8 per lane
        const bf16x4 a = *reinterpret_cast<const bf16x4*>(ho_strip(p, n_tiles, bn, qb) + lane * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) pt[0][j] = a[j];
      } else {
        const unsigned char* tp = band_tile(qb);
        pt[0] = *reinterpret_cast<const bf16x8*>(tp); pt[1] = *reinterpret_cast<const bf16x8*>(tp + 1024);
      }
    };
    fetch(t0);
    if (!split_item && ng > 0) global_rows(qlds);
    for (int it = 0; it < n_it; ++it) {
      const int qb = t0 + it;
      tile_to_lds(qlds, qt, lane);
      tile_to_lds(dolds, dot, lane);
      if (split_item) {
        *reinterpret_cast<bf16x4*>(plds + lane * 8) = bf16x4{pt[0][0], pt[0][1], pt[0][2], pt[0][3]};
      } else {
        *reinterpret_cast<bf16x8*>(plds + lane * 16) = pt[0]; *reinterpret_cast<bf16x8*>(plds + 1024 + lane * 16) = pt[1];
      }
      if (h == 0) drow[r] = rc_d;
      if (it + 1 < n_it) fetch(qb + 1);
      wave_lds_sync();
      tile(qlds, dolds, plds, drow, qb * 32, split_item);
      wave_lds_sync();
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) { dk0[i] *= p.gscale; dk1[i] *= p.gscale; dv0[i] *= ik; dv1[i] *= ik; }

  // fp32 rows of the global keys: partial slot `chunk` (items) or n_chunks (band waves) of part_dkv
  const bool k_glob = ng > 0 && (unsigned)(k - g0) < (unsigned)ng;
  if (split_item || k_glob) {
    const int row = split_item ? r : k - g0;
    const long slot = (long)bn * p.n_gblk * p.dkv_slots + (split_item ? chunk : p.n_chunks);
    float* pk = p.part_dkv + slot * (2 * 32 * 64) + row * 64;
    float* pv2 = pk + 32 * 64;
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      *reinterpret_cast<f32x4*>(pk + 8 * gi + 4 * h) = f32x4{dk0[4 * gi], dk0[4 * gi + 1], dk0[4 * gi + 2], dk0[4 * gi + 3]};
      *reinterpret_cast<f32x4*>(pk + 32 + 8 * gi + 4 * h) = f32x4{dk1[4 * gi], dk1[4 * gi + 1], dk1[4 * gi + 2], dk1[4 * gi + 3]};
      *reinterpret_cast<f32x4*>(pv2 + 8 * gi + 4 * h) = f32x4{dv0[4 * gi], dv0[4 * gi + 1], dv0[4 * gi + 2], dv0[4 * gi + 3]};
      *reinterpret_cast<f32x4*>(pv2 + 32 + 8 * gi + 4 * h) = f32x4{dv1[4 * gi], dv1[4 * gi + 1], dv1[4 * gi + 2], dv1[4 * gi + 3]};
    }
    return;
  }
  if (k >= p.S) return;
  T* DK = reinterpret_cast<T*>(p.dk) + (long)b * p.ks[0] + (long)k * p.ks[1] + (long)n * p.ks[2];
  T* DV = reinterpret_cast<T*>(p.dv) + (long)b * p.vs[0] + (long)k * p.vs[1] + (long)n * p.vs[2];
#pragma unroll
  for (int gi = 0; gi < 4; ++gi) {
    const int d = 8 * gi + 4 * h;
    bf16x4 x, y, z, u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      x[j] = (__bf16)dk0[4 * gi + j]; y[j] = (__bf16)dk1[4 * gi + j];
      z[j] = (__bf16)dv0[4 * gi + j]; u[j] = (__bf16)dv1[4 * gi + j];
    }
    *reinterpret_cast<bf16x4*>(DK + d) = x; *reinterpret_cast<bf16x4*>(DK + 32 + d) = y;
    *reinterpret_cast<bf16x4*>(DV + d) = z; *reinterpret_cast<bf16x4*>(DV + 32 + d) = u;
  }
  HSTAMP(13);
}

// ------------------------------------ launcher --------------------------------------------
template <int Rp, int REL>
static hipError_t launch_lean(const BwdParams& p_in, hipStream_t st) {
  constexpr bool HAS_REL = REL != 0;
  BwdParams p = p_in;
  p.comb_in_next = 0;
  p.red_per_plane = p.red_live = (p.S + 127) >> 7;        // one dE partial per 128-row workgroup
  const int per_bn = (p.n_chunks * p.n_gblk + 3) / 4;
  dim3 grid(p.n_band_blocks + per_bn * p.B * p.N);
  const int e_img = HAS_REL ? Rp * 128 : 0;     // the workgroup's E image
  p.dstride = (REL != 2 && Rp == 32 && 2 * p.pat.m + 1 <= 27) ? 28 : kTStride(Rp);     // 27 r: conflict-free diagonal stores
  const int n2 = 2 * p.pat.r + 3, lut_bytes = REL == 2 ? 4 * ((n2 * n2 + 15) & ~15) : 0;   // (dx, dy) look-up table, one per workgroup
  const int lds_a = 4 * (LeanLds<Rp>::kTab + 32 * p.dstride * 4 + 2 * LeanLds<Rp>::kTile) + e_img + lut_bytes, lds_b = 4 * LeanLds<Rp>::kDkv + e_img + lut_bytes;
  if (lds_a > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_band_bf16_kernel<Rp, REL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_a);
  if (lds_b > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_band_bf16_kernel<Rp, REL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_b);
  hipLaunchKernelGGL((attn_bwd_dq_band_bf16_kernel<Rp, REL>), grid, dim3(256), lds_a, st, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  // With a dE reduce to follow, the two combines of the global-token partials ride in the next launches
  // (attn_combine.h); without one they stay launches of their own.
  const bool ride = p.n_gblk > 0 && p.R > 0;
  if (p.n_gblk > 0 && !ride && (e = launch_bwd_dq_combine(p, true, st)) != hipSuccess) return e;
  p.comb_in_next = ride ? 1 : 0;
  dim3 grid_kv(grid.x + (ride ? (p.pat.ng * p.B * p.N + 3) / 4 : 0));
  if (p.ho) {
    const bool win = p.ho_slots <= 5 && !p.ho_per_wave;      // q window of a 128-key workgroup: 4 + ho_slots - 1 <= 8 tiles
    if (win) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_ho_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kHoWinLds);
      hipLaunchKernelGGL(attn_bwd_dkv_ho_kernel<true>, grid_kv, dim3(256), kHoWinLds, st, p);
    } else {
      hipLaunchKernelGGL(attn_bwd_dkv_ho_kernel<false>, grid_kv, dim3(256), 4 * kHoWaveLds, st, p);
    }
  } else hipLaunchKernelGGL((attn_bwd_dkv_band_bf16_kernel<Rp, REL>), grid_kv, dim3(256), lds_b, st, p);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if (p.n_gblk > 0 && !ride && (e = launch_bwd_dkv_combine(p, true, st)) != hipSuccess) return e;
  if (p.R > 0) e = launch_drel_reduce(p, true, st);
  return e;
}

hipError_t launch_attn_bwd_band_bf16(const BwdParams& p, hipStream_t st) {
  if (p.lean2d) return p.Rp == 32 ? launch_lean<32, 2>(p, st) : launch_lean<64, 2>(p, st);
  const bool has_rel = p.pat.id_mode == 1 && p.R > 0;
  if (p.Rp == 32) return has_rel ? launch_lean<32, 1>(p, st) : launch_lean<32, 0>(p, st);
  return has_rel ? launch_lean<64, 1>(p, st) : launch_lean<64, 0>(p, st);
}

}  // namespace mmt
