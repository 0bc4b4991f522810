// Backward of QkvRelativeAttention for gfx950 (MI355X).  Math: SURVEY.md App. A.3
// differentiated (checked against oracle/attention.py::relative_attention_bwd):
//
//   P = softmax(s),  s = (q.k + relall[q,id]) * scale + maskadd,  O = drop(P) V
//   dV = drop(P)^T dO;  dP = drop'(dO V^T);  dS = P o (dP - delta),  delta = rowsum(dO o O)
//   dQ = gscale * dS K + dRel E;   dK = gscale * dS^T Q
//   dRel[q,r] = rel_gscale * sum_k dS[q,k] [id(q,k) = r];  dE = dRel^T Q;  dbias = colsum(dRel)
//
// Two deterministic, atomic-free passes that recompute P from the forward's LSE:
//   attn_bwd_dq_kernel  : one wave = 32 query rows (row on the lane, as the forward);
//                         walks key tiles; produces dQ, delta, dRel.                     [K4a]
//   attn_bwd_dkv_kernel : one wave = 32 keys (key on the lane, query rows in registers);
//                         walks query tiles; produces dK, dV.                            [K4b]
// Global rows (K4a) / global keys (K4b) see the whole sequence: they are split into chunks
// handled by extra work items of the same launch, summed by small combine kernels.
// dE / dbias: each K4a wave emits its Q^T.dRel partial (MFMA); one fixed-order pass sums them
// (no atomics, bitwise reproducible).
#include "attn_tile.h"
#include "attn_combine.h"

namespace mmt {

// LDS per wave: T table, dT table (K4a) / second tile (K4b), one or two 32 x 128 B tiles.
template <typename T, int Rp> struct BwdLds {
  static constexpr int kTab = (32 * kTStride(Rp) * 4 + 15) & ~15;
  static constexpr int kTile = sizeof(T) == 2 ? 32 * 128 : 0;
  static constexpr int kBias = Rp * 4;                // bias[id] * tscale
  static constexpr int kDq = 2 * kTab + kTile + kBias;   // tab, dtab, K/E tile, bias
  static constexpr int kDkv = kTab + 2 * kTile + kBias;  // tab, Q tile, dO tile, bias
};

struct TileWalk {
  int a0, lenA, b0, lenB, c0, lenC;
  __device__ __forceinline__ int count() const { return lenA + lenB + lenC; }
  __device__ __forceinline__ int at(int it) const {
    return it < lenA ? a0 + it : (it < lenA + lenB ? b0 + (it - lenA) : c0 + (it - lenA - lenB));
  }
};

// Tiles a 32-row block starting at x0 must visit: the band around it plus the tiles that hold
// global tokens (the pattern is symmetric in (q,k), so this serves both passes).
__device__ __forceinline__ TileWalk band_walk(const PatternDev& pat, int x0, int S) {
  TileWalk w{0, 0, 0, 0, 0, 0};
  const int lo = max(x0 - pat.radius, 0), hi = min(x0 + 31 + pat.radius, S - 1);
  w.b0 = lo >> 5;
  const int b1 = hi >> 5;
  w.lenB = b1 - w.b0 + 1;
  if (pat.ng > 0) {
    const int g_lo = pat.g0 >> 5, g_hi = (pat.g0 + pat.ng - 1) >> 5;
    w.a0 = g_lo; w.lenA = max(0, min(g_hi, w.b0 - 1) - g_lo + 1);
    w.c0 = max(g_lo, b1 + 1); w.lenC = max(0, g_hi - w.c0 + 1);
  }
  return w;
}

// Mask bit and relative-table column of one (q,k) pair (col < 0: no relative term).
template <int MODE, bool GEN, typename P>
__device__ __forceinline__ void pair_mask_col(const P& p, int b, int valid_len, int q, int k,
                                              bool& keep, int& col) {
  int id = -1;
  col = -1;
  if (MODE == kDense) {
    const long off = ((long)b * p.S + min(q, p.S - 1)) * p.S + min(k, p.S - 1);
    keep = p.att_mask ? p.att_mask[off] != 0 : true;
    if (p.rel_ids) id = p.rel_ids[off];
    if ((unsigned)id < (unsigned)p.R) col = id;
  } else if (GEN) {
    keep = pattern_mask(p.pat, valid_len, q, k);
    if (p.pat.id_mode) id = rel_id(p.pat, q, k);
    if ((unsigned)id < (unsigned)p.R) col = id;
  } else {
    const int d = k - q;
    const unsigned W = (unsigned)p.pat.radius;
    const bool near = (unsigned)(d + (int)W) <= 2u * W;
    const bool seg = (k < valid_len) == (q < valid_len);
    keep = (int)seg & ((int)near | (int)is_global(p.pat, k) | (int)is_global(p.pat, q));
    if (p.pat.id_mode == 1) col = min(max(d, -p.pat.m), p.pat.m) + p.pat.m;
  }
}

template <typename T, typename P>
__device__ __forceinline__ float drop_factor(const P& p, int bn, int q, int k) {
  if (!p.drop_thresh) return 1.f;
  const uint32_t bits = drop_bits16(drop_row_base(effective_seed(p.seed_lo, p.seed_hi, p.epoch).lo, effective_seed(p.seed_lo, p.seed_hi, p.epoch).hi, (uint32_t)bn, (uint32_t)q), (uint32_t)k);
  return bits >= p.drop_thresh ? p.inv_keep : 0.f;
}

// bias_ts[id] = bias[id] * tscale, once per wave.
template <typename T, int Rp, typename P>
__device__ __forceinline__ void fill_bias(const P& p, int n, float* bias_ts, int lane) {
  for (int id = lane; id < Rp; id += 64)
    bias_ts[id] = p.bias ? (float)reinterpret_cast<const T*>(p.bias)[(long)min(id, max(p.R - 1, 0)) * p.N + n] * p.tscale : 0.f;
}

// Builds T[row][col(id)] = (x_row . E[id]) * tscale + bias_ts[id] for the 32 rows whose fragments
// are in `xf` (row = lane & 31).
template <typename T, int Rp, bool IDENT, typename P>
__device__ __forceinline__ void build_table(const P& p, int n, const Frag<T>& xf, float* tab,
                                            const float* bias_ts, int lane) {
  const int r = lane & 31, h = lane >> 5;
  const T* E = reinterpret_cast<const T*>(p.emb) + (long)n * 64;
#pragma unroll
  for (int rb = 0; rb < Rp / 32; ++rb) {
    const int rr = rb * 32 + r;
    Frag<T> ef;
    ef.load_row(E + (long)min(rr, p.R - 1) * p.N * 64, h);
    f32x16 c = {0};
    c = mma_rows(ef, xf, c);  // [id x row]
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int id = rb * 32 + kap(i, h);
      const int col = IDENT ? id : tcol(p.perm_1d, p.pat.m, id);
      tab[r * kTStride(Rp) + col] = fmaf(c[i], p.tscale, bias_ts[id]);
    }
  }
}

// =========================================================================================
// K4a: dQ, delta, dRel.  Lane (r,h) owns query row q0 + r; registers walk keys.
// =========================================================================================
template <typename T, int MODE, int Rp, bool GEN>
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? 2 : 1)) void attn_bwd_dq_kernel(const BwdParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  using L = BwdLds<T, Rp>;
  unsigned char* wl = smem + wave * L::kDq;
  float* tab = reinterpret_cast<float*>(wl);
  float* dtab = reinterpret_cast<float*>(wl + L::kTab);
  unsigned char* xlds = wl + 2 * L::kTab;
  float* bias_ts = reinterpret_cast<float*>(wl + 2 * L::kTab + L::kTile);
  constexpr bool IDENT = MODE == kDense || GEN;

  const int n_tiles = (p.S + 31) >> 5, nqb = (p.S + 127) >> 7;
  const bool split_item = MODE == kBand && (int)blockIdx.x >= p.n_band_blocks;
  int bn, q0, chunk = 0, gblk = 0;
  if (split_item) {
    const int per_bn = (p.n_chunks * p.n_gblk + 3) >> 2;
    const int rb = blockIdx.x - p.n_band_blocks;
    bn = rb / per_bn;
    const int item = (rb - bn * per_bn) * 4 + wave;
    if (item >= p.n_chunks * p.n_gblk) return;
    gblk = item / p.n_chunks;
    chunk = item - gblk * p.n_chunks;
    q0 = p.pat.g0 + gblk * 32;
  } else {
    const int wg = xcd_remap(blockIdx.x, p.n_band_blocks);
    bn = wg / nqb;
    q0 = (wg - bn * nqb) * 128 + wave * 32;
    if (q0 >= p.S) return;
  }
  const int b = bn / p.N, n = bn - b * p.N;
  const int q = q0 + r;
  const bool q_ok = q < p.S;
  const unsigned qc = (unsigned)min(q, p.S - 1);
  const int valid_len = p.valid_len ? p.valid_len[b] : p.S;
  const T* Q = reinterpret_cast<const T*>(p.q) + (long)b * p.qs[0] + (long)n * p.qs[2];
  const T* K = reinterpret_cast<const T*>(p.k) + (long)b * p.ks[0] + (long)n * p.ks[2];
  const T* V = reinterpret_cast<const T*>(p.v) + (long)b * p.vs[0] + (long)n * p.vs[2];
  const T* O = reinterpret_cast<const T*>(p.out) + (long)b * p.os[0] + (long)n * p.os[2];
  const T* DO = reinterpret_cast<const T*>(p.dout) + (long)b * p.os[0] + (long)n * p.os[2];
  const unsigned qs1 = (unsigned)p.qs[1], ks1 = (unsigned)p.ks[1], vs1 = (unsigned)p.vs[1], os1 = (unsigned)p.os[1];

  TileWalk w{0, 0, 0, n_tiles, 0, 0};
  if (MODE == kBand) {
    if (split_item) { w.b0 = chunk * p.chunk_tiles; w.lenB = min(n_tiles, w.b0 + p.chunk_tiles) - w.b0; }
    else w = band_walk(p.pat, q0, p.S);
  }

  Frag<T> qf, dof;
  qf.load_row(Q + qc * qs1, h);
  dof.load_row(DO + qc * os1, h);
  float delta;
  {
    Frag<T> of;
    of.load_row(O + qc * os1, h);
    float acc = 0.f;
    constexpr int kN = sizeof(T) == 2 ? 4 : 32;
#pragma unroll
    for (int s = 0; s < kN; ++s) {
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = fmaf((float)of.v[s][j], (float)dof.v[s][j], acc);
      } else {
        acc = fmaf(of.v[s], dof.v[s], acc);
      }
    }
    delta = half_sum(acc);
  }
  const long row_id = ((long)b * p.N + n) * p.S + qc;
  if (!split_item && q_ok && h == 0) p.delta[row_id] = delta;
  const float lse2 = p.lse[row_id] * kLog2e;

  fill_bias<T, Rp>(p, n, bias_ts, lane);
  for (int i = lane; i < 32 * kTStride(Rp); i += 64) dtab[i] = 0.f;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  if (p.R > 0) build_table<T, Rp, IDENT>(p, n, qf, tab, bias_ts, lane);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  f32x16 a0 = {0}, a1 = {0};
  float far_neg = 0.f, far_pos = 0.f;
  const float* trow = tab + r * kTStride(Rp);
  float* dtrow = dtab + r * kTStride(Rp);
  const int n_it = w.count();
  Frag<T> kf, vf;
  VTile<T> kt;
  {
    const int k0 = w.at(0) * 32;
    kf.load_row(K + (unsigned)min(k0 + r, p.S - 1) * ks1, h);
    vf.load_row(V + (unsigned)min(k0 + r, p.S - 1) * vs1, h);
    kt.load(K, ks1, k0, p.S, lane, 0);
  }
  for (int it = 0; it < n_it; ++it) {
    const int k0 = w.at(it) * 32;
    kt.to_lds(xlds, lane);
    VTile<T> kcur;
    if constexpr (sizeof(T) == 4) kcur = kt;
    f32x16 c = {0}, dp = {0};
    c = mma_rows(kf, qf, c);     // S^T  [key x q]
    dp = mma_rows(vf, dof, dp);  // dP^T [key x q]
    if (it + 1 < n_it) {         // next tile's operands arrive under this tile's math
      const int k1 = w.at(it + 1) * 32;
      kf.load_row(K + (unsigned)min(k1 + r, p.S - 1) * ks1, h);
      vf.load_row(V + (unsigned)min(k1 + r, p.S - 1) * vs1, h);
      kt.load(K, ks1, k1, p.S, lane, 0);
    }
    // phase 1: mask bits, table columns and the gathered relative scores (reads only)
    int cols[16];
    float rel[16];
    unsigned keepm = 0, existm = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int kk = k0 + kap(i, h);
      bool keep;
      pair_mask_col<MODE, GEN>(p, b, valid_len, q, kk, keep, cols[i]);
      keepm |= (unsigned)keep << i;
      existm |= (unsigned)(kk < p.S && q_ok) << i;
      rel[i] = cols[i] >= 0 ? trow[cols[i]] : 0.f;
    }
    // phase 2: P, dS
    float g[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float s2 = fmaf(c[i], p.sscale, rel[i]);
      s2 = (keepm >> i) & 1 ? s2 : s2 + p.mask_add;
      const float pr = (existm >> i) & 1 ? __builtin_amdgcn_exp2f(s2 - lse2) : 0.f;
      float dpi = dp[i];
      if (p.drop_thresh) dpi *= drop_factor<T>(p, bn, q, k0 + kap(i, h));
      g[i] = pr * (dpi - delta);          // dS
    }
    // phase 3: dRel scatter.
    if constexpr (!IDENT) {
      // permuted 1-D table: column m+d with |d| < m receives exactly ONE key (k = q+d) -> plain
      // store; the two clipped columns (0 and 2m) are summed in registers.  No atomics.
      if (p.pat.id_mode == 1) {
        const int m = p.pat.m, d0 = k0 - q + 4 * h;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int d = d0 + (i & 3) + 8 * (i >> 2);
          const float x = g[i] * p.rel_gscale;
          const bool neg = d <= -m, pos = (d >= m) && !neg;
          far_neg += neg ? x : 0.f;
          far_pos += pos ? x : 0.f;
          if (!neg && !pos) dtrow[m + d] = x;
        }
      }
    } else {
      // arbitrary ids: wave-private LDS float atomics (the two halves share rows)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (cols[i] >= 0) atomicAdd(&dtrow[cols[i]], g[i] * p.rel_gscale);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) g[i] *= p.gscale;
    if constexpr (sizeof(T) == 2) mma_xt(a0, a1, kt, xlds, g, lane);   // dQ^T += K^T . dS^T
    else mma_xt(a0, a1, kcur, xlds, g, lane);
  }
  if (!IDENT && p.pat.id_mode == 1) {       // flush the clipped columns (both halves of the row)
    const float fn = half_sum(far_neg), fp = half_sum(far_pos);
    if (h == 0) {
      if (p.pat.m == 0) dtrow[0] = fn + fp;
      else { dtrow[0] = fn; dtrow[2 * p.pat.m] = fp; }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();

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
      pt[i] = dtab[rr * kTStride(Rp) + (IDENT ? id : tcol(p.perm_1d, p.pat.m, id))];
    }
    return;
  }

  if (p.R > 0) {
    // (1) this wave's share of dE^T[d x id] = sum_q Q[q][d] * dRel[q][id] and dbias[id] (lane = id).
    //     Rows of global tokens are excluded here when they are produced by the split items.
    {
      VTile<T> qt;
      qt.load(Q, qs1, q0, p.S, lane, 0);
      qt.to_lds(xlds, lane);
      const int widx = (xcd_remap(blockIdx.x, p.n_band_blocks)) * 4 + wave;
      float* pe = p.part_red + (long)widx * (Rp * 64 + Rp);
#pragma unroll
      for (int rb = 0; rb < Rp / 32; ++rb) {
        const int id = rb * 32 + r;
        const int col = IDENT ? id : tcol(p.perm_1d, p.pat.m, id);
        float vals[16], bsum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int qq = q0 + kap(i, h);
          const bool use = id < p.R && qq < p.S && !(p.skip_global && is_global(p.pat, qq));
          vals[i] = use ? dtab[kap(i, h) * kTStride(Rp) + col] : 0.f;
          bsum += vals[i];
        }
        bsum = half_sum(bsum);
        f32x16 e0 = {0}, e1 = {0};
        mma_xt_hilo(e0, e1, qt, xlds, vals, lane);
        float* row = pe + (long)id * 64;
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
          *reinterpret_cast<f32x4*>(row + 8 * gi + 4 * h) = f32x4{e0[4 * gi], e0[4 * gi + 1], e0[4 * gi + 2], e0[4 * gi + 3]};
          *reinterpret_cast<f32x4*>(row + 32 + 8 * gi + 4 * h) = f32x4{e1[4 * gi], e1[4 * gi + 1], e1[4 * gi + 2], e1[4 * gi + 3]};
        }
        if (h == 0) pe[Rp * 64 + id] = bsum;
      }
    }
    // (2) dQ^T += E^T[d x id] . dRel^T[id x q]   (dRel at ~16 mantissa bits: hi/lo bf16 split)
    const T* E = reinterpret_cast<const T*>(p.emb) + (long)n * 64;
#pragma unroll
    for (int rb = 0; rb < Rp / 32; ++rb) {
      float vals[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int id = rb * 32 + kap(i, h);
        vals[i] = id < p.R ? dtrow[IDENT ? id : tcol(p.perm_1d, p.pat.m, id)] : 0.f;
      }
      VTile<T> et;   // rows = ids rb*32 .. rb*32+31 of E (clamped past R: their dRel is 0)
      et.load(E + (long)(rb * 32) * p.N * 64, (unsigned)(p.N * 64), 0, max(p.R - rb * 32, 1), lane, 0);
      et.to_lds(xlds, lane);
      mma_xt_hilo(a0, a1, et, xlds, vals, lane);
    }
  }
  if (!q_ok || (p.skip_global && is_global(p.pat, q))) return;
  T* DQ = reinterpret_cast<T*>(p.dq) + (long)b * p.qs[0] + (long)q * p.qs[1] + (long)n * p.qs[2];
#pragma unroll
  for (int gi = 0; gi < 4; ++gi) {
    const int d = 8 * gi + 4 * h;
    if constexpr (sizeof(T) == 2) {
      bf16x4 x, y;
#pragma unroll
      for (int j = 0; j < 4; ++j) { x[j] = (__bf16)a0[4 * gi + j]; y[j] = (__bf16)a1[4 * gi + j]; }
      *reinterpret_cast<bf16x4*>(DQ + d) = x;
      *reinterpret_cast<bf16x4*>(DQ + 32 + d) = y;
    } else {
      *reinterpret_cast<f32x4*>(DQ + d) = f32x4{a0[4 * gi], a0[4 * gi + 1], a0[4 * gi + 2], a0[4 * gi + 3]};
      *reinterpret_cast<f32x4*>(DQ + 32 + d) = f32x4{a1[4 * gi], a1[4 * gi + 1], a1[4 * gi + 2], a1[4 * gi + 3]};
    }
  }
}

// Global rows: sum the chunk partials, add dRel.E, write dQ and the row's dRel (for dE/dbias).
template <typename T>
__global__ __launch_bounds__(64) void attn_bwd_dq_combine_kernel(const BwdParams p) {
  __shared__ float dr_s[128];
  dq_combine_row<T>(p, blockIdx.y, blockIdx.x, threadIdx.x, dr_s, [] { __syncthreads(); });
}

// =========================================================================================
// K4b: dK, dV.  Lane (r,h) owns key k0 + r; registers walk query rows.
// =========================================================================================
template <typename T, int MODE, int Rp, bool GEN>
__global__ __launch_bounds__(256, (sizeof(T) == 2 ? 2 : 1)) void attn_bwd_dkv_kernel(const BwdParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  using L = BwdLds<T, Rp>;
  unsigned char* wl = smem + wave * L::kDkv;
  float* tab = reinterpret_cast<float*>(wl);
  unsigned char* qlds = wl + L::kTab;
  unsigned char* dolds = qlds + L::kTile;
  float* bias_ts = reinterpret_cast<float*>(dolds + L::kTile);
  constexpr bool IDENT = MODE == kDense || GEN;

  const int n_tiles = (p.S + 31) >> 5, nkb = (p.S + 127) >> 7;
  const bool split_item = MODE == kBand && (int)blockIdx.x >= p.n_band_blocks;
  int bn, k0, chunk = 0, gblk = 0;
  if (split_item) {
    const int per_bn = (p.n_chunks * p.n_gblk + 3) >> 2;
    const int rb = blockIdx.x - p.n_band_blocks;
    bn = rb / per_bn;
    const int item = (rb - bn * per_bn) * 4 + wave;
    if (item >= p.n_chunks * p.n_gblk) return;
    gblk = item / p.n_chunks;
    chunk = item - gblk * p.n_chunks;
    k0 = p.pat.g0 + gblk * 32;
  } else {
    const int wg = xcd_remap(blockIdx.x, p.n_band_blocks);
    bn = wg / nkb;
    k0 = (wg - bn * nkb) * 128 + wave * 32;
    if (k0 >= p.S) return;
  }
  const int b = bn / p.N, n = bn - b * p.N;
  const int k = k0 + r;
  const bool k_ok = k < p.S;
  const unsigned kc = (unsigned)min(k, p.S - 1);
  const int valid_len = p.valid_len ? p.valid_len[b] : p.S;
  const T* Q = reinterpret_cast<const T*>(p.q) + (long)b * p.qs[0] + (long)n * p.qs[2];
  const T* K = reinterpret_cast<const T*>(p.k) + (long)b * p.ks[0] + (long)n * p.ks[2];
  const T* V = reinterpret_cast<const T*>(p.v) + (long)b * p.vs[0] + (long)n * p.vs[2];
  const T* DO = reinterpret_cast<const T*>(p.dout) + (long)b * p.os[0] + (long)n * p.os[2];
  const unsigned qs1 = (unsigned)p.qs[1], ks1 = (unsigned)p.ks[1], vs1 = (unsigned)p.vs[1], os1 = (unsigned)p.os[1];
  const float* lse_bn = p.lse + ((long)b * p.N + n) * p.S;
  const float* delta_bn = p.delta + ((long)b * p.N + n) * p.S;

  TileWalk w{0, 0, 0, n_tiles, 0, 0};
  if (MODE == kBand) {
    if (split_item) { w.b0 = chunk * p.chunk_tiles; w.lenB = min(n_tiles, w.b0 + p.chunk_tiles) - w.b0; }
    else w = band_walk(p.pat, k0, p.S);
  }

  Frag<T> kf, vf;
  kf.load_row(K + kc * ks1, h);
  vf.load_row(V + kc * vs1, h);
  f32x16 dk0 = {0}, dk1 = {0}, dv0 = {0}, dv1 = {0};
  fill_bias<T, Rp>(p, n, bias_ts, lane);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int n_it = w.count();
  VTile<T> qt, dot;
  if constexpr (sizeof(T) == 2) {            // first tile's rows (prefetch registers)
    qt.load(Q, qs1, w.at(0) * 32, p.S, lane, 0);
    dot.load(DO, os1, w.at(0) * 32, p.S, lane, 0);
  }
  for (int it = 0; it < n_it; ++it) {
    const int q0 = w.at(it) * 32;
    Frag<T> qf, dof;
    if constexpr (sizeof(T) == 2) {
      qt.to_lds(qlds, lane);
      dot.to_lds(dolds, lane);
      if (it + 1 < n_it) {                   // next tile's rows arrive under this tile's math
        qt.load(Q, qs1, w.at(it + 1) * 32, p.S, lane, 0);
        dot.load(DO, os1, w.at(it + 1) * 32, p.S, lane, 0);
      }
    } else {
      qf.load_row(Q + (unsigned)min(q0 + r, p.S - 1) * qs1, h);
      dof.load_row(DO + (unsigned)min(q0 + r, p.S - 1) * os1, h);
      qt.load(Q, qs1, q0, p.S, lane, 0);
      dot.load(DO, os1, q0, p.S, lane, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if constexpr (sizeof(T) == 2) frag_from_tile(qf, qlds, lane);
    if (p.R > 0) build_table<T, Rp, IDENT>(p, n, qf, tab, bias_ts, lane);   // T rows = this q tile
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    f32x16 c = {0}, dp = {0};
    c = mma_rows(qf, kf, c);     // S  [q x key]
    if constexpr (sizeof(T) == 2) frag_from_tile(dof, dolds, lane);
    dp = mma_rows(dof, vf, dp);  // dP [q x key]
    int cols[16];
    float rel[16];
    unsigned keepm = 0, existm = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int qq = q0 + kap(i, h);
      bool keep;
      pair_mask_col<MODE, GEN>(p, b, valid_len, qq, k, keep, cols[i]);
      keepm |= (unsigned)keep << i;
      existm |= (unsigned)(qq < p.S && k_ok) << i;
      rel[i] = cols[i] >= 0 ? tab[kap(i, h) * kTStride(Rp) + cols[i]] : 0.f;
    }
    float pv[16], g[16];
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
      // per-row constants (LSE, delta) of 4 consecutive query rows
      float l4[4], d4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int qq = min(q0 + 8 * gi + 4 * h + j, p.S - 1);
        l4[j] = lse_bn[qq] * kLog2e;
        d4[j] = delta_bn[qq];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int i = 4 * gi + j;
        float s2 = fmaf(c[i], p.sscale, rel[i]);
        s2 = (keepm >> i) & 1 ? s2 : s2 + p.mask_add;
        const float pr = (existm >> i) & 1 ? __builtin_amdgcn_exp2f(s2 - l4[j]) : 0.f;
        float df = 1.f;
        if (p.drop_thresh) df = drop_factor<T>(p, bn, q0 + kap(i, h), k);
        pv[i] = pr * df;
        g[i] = pr * (dp[i] * df - d4[j]) * p.gscale;
      }
    }
    mma_xt(dv0, dv1, dot, dolds, pv, lane);  // dV^T[d x key] += dO^T[d x q] . P[q x key]
    mma_xt(dk0, dk1, qt, qlds, g, lane);     // dK^T[d x key] += Q^T[d x q] . dS[q x key]
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

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
    if constexpr (sizeof(T) == 2) {
      bf16x4 x, y, z, u;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x[j] = (__bf16)dk0[4 * gi + j]; y[j] = (__bf16)dk1[4 * gi + j];
        z[j] = (__bf16)dv0[4 * gi + j]; u[j] = (__bf16)dv1[4 * gi + j];
      }
      *reinterpret_cast<bf16x4*>(DK + d) = x; *reinterpret_cast<bf16x4*>(DK + 32 + d) = y;
      *reinterpret_cast<bf16x4*>(DV + d) = z; *reinterpret_cast<bf16x4*>(DV + 32 + d) = u;
    } else {
      *reinterpret_cast<f32x4*>(DK + d) = f32x4{dk0[4 * gi], dk0[4 * gi + 1], dk0[4 * gi + 2], dk0[4 * gi + 3]};
      *reinterpret_cast<f32x4*>(DK + 32 + d) = f32x4{dk1[4 * gi], dk1[4 * gi + 1], dk1[4 * gi + 2], dk1[4 * gi + 3]};
      *reinterpret_cast<f32x4*>(DV + d) = f32x4{dv0[4 * gi], dv0[4 * gi + 1], dv0[4 * gi + 2], dv0[4 * gi + 3]};
      *reinterpret_cast<f32x4*>(DV + 32 + d) = f32x4{dv1[4 * gi], dv1[4 * gi + 1], dv1[4 * gi + 2], dv1[4 * gi + 3]};
    }
  }
}

template <typename T>
__global__ __launch_bounds__(64) void attn_bwd_dkv_combine_kernel(const BwdParams p) {
  dkv_combine_row<T>(p, blockIdx.y, blockIdx.x, threadIdx.x);
}

// =========================================================================================
// K4c: dE[id,n,:] = sum over the per-wave partials of head n (+ the rows of global tokens),
// dbias likewise.  Fixed summation order: bitwise reproducible.  grid (Rp, N), 64 threads = d.
// =========================================================================================
template <typename T>
__global__ __launch_bounds__(1024) void drel_reduce_kernel(const BwdParams p) {
  __shared__ float red[16][64], redb[16];
  const int id = blockIdx.x, n = blockIdx.y, d = threadIdx.x & 63, part = threadIdx.x >> 6;
  if (id >= p.Rp) {            // lean path: the dK/dV combine of the global rows rides along (16 rows per block)
    const int pair = ((id - p.Rp) * p.N + n) * 16 + part;
    if (pair < p.pat.ng * p.B * p.N) dkv_combine_row<T>(p, pair / p.pat.ng, pair % p.pat.ng, d);
    return;
  }
  if (id >= p.R) return;
  const int per = p.Rp * 64 + p.Rp;
  const int waves_per_bn = p.red_per_plane;       // partial slots per plane (one per wave, or per workgroup)
  const int live = p.red_live;                    // slots past the end of the sequence were not written
  const int total = p.B * live;
  const int chunk = (total + 15) >> 4;
  const int lo = part * chunk, hi = min(total, lo + chunk);
  float acc = 0.f, bs = 0.f;
#pragma unroll 8
  for (int i = lo; i < hi; ++i) {
    const int b = i / live, w = i - b * live;
    const float* src = p.part_red + ((long)(b * p.N + n) * waves_per_bn + w) * per;
    acc += src[id * 64 + d];
    if (d == 0) bs += src[p.Rp * 64 + id];
  }
  if (p.n_gblk > 0) {                             // rows of global tokens (from the combine): dealt over the 16
    const int pairs = p.B * p.pat.ng;             // parts, so that no part walks a long chain of dependent loads
#pragma unroll 2
    for (int gi = part; gi < pairs; gi += 16) {
      const int b = gi / p.pat.ng, g = gi - b * p.pat.ng;
      const float x = p.drel[((long)(b * p.N + n) * p.pat.ng + g) * p.Rp + id];
      const T* Q = reinterpret_cast<const T*>(p.q) + (long)b * p.qs[0] + (long)n * p.qs[2];
      acc = fmaf(x, (float)Q[(long)(p.pat.g0 + g) * p.qs[1] + d], acc);
      if (d == 0) bs += x;
    }
  }
  red[part][d] = acc;
  if (d == 0) redb[part] = bs;
  __syncthreads();
  if (part == 0) {
    float a = 0.f, bsum = 0.f;
    for (int j = 0; j < 16; ++j) { a += red[j][d]; bsum += redb[j]; }
    float* de = p.drel_emb + ((long)id * p.N + n) * 64 + d;
    *de = p.drel_accum ? *de + a : a;
    if (d == 0 && p.drel_bias) {
      float* db = p.drel_bias + (long)id * p.N + n;
      *db = p.drel_accum ? *db + bsum : bsum;
    }
  }
}

// ------------------------------------ launchers -----------------------------------------
template <typename K>
static void allow_lds(K kernel, int bytes) {
  if (bytes > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

template <typename T, int MODE, int Rp, bool GEN>
static hipError_t launch_bwd_one(const BwdParams& p_in, hipStream_t st) {
  BwdParams p = p_in;
  p.red_per_plane = ((p.S + 127) >> 7) * 4;               // one dE partial per wave (32 rows)
  p.red_live = (p.S + 31) >> 5;
  const int per_bn = (p.n_chunks * p.n_gblk + 3) / 4;
  dim3 grid(p.n_band_blocks + (MODE == kBand ? per_bn * p.B * p.N : 0));
  const int lds_a = 4 * BwdLds<T, Rp>::kDq, lds_b = 4 * BwdLds<T, Rp>::kDkv;
  allow_lds(attn_bwd_dq_kernel<T, MODE, Rp, GEN>, lds_a);
  allow_lds(attn_bwd_dkv_kernel<T, MODE, Rp, GEN>, lds_b);
  hipLaunchKernelGGL((attn_bwd_dq_kernel<T, MODE, Rp, GEN>), grid, dim3(256), lds_a, st, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (MODE == kBand && p.n_gblk > 0) {
    hipLaunchKernelGGL(attn_bwd_dq_combine_kernel<T>, dim3(p.pat.ng, p.B * p.N), dim3(64), 0, st, p);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<T, MODE, Rp, GEN>), grid, dim3(256), lds_b, st, p);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if (MODE == kBand && p.n_gblk > 0) {
    hipLaunchKernelGGL(attn_bwd_dkv_combine_kernel<T>, dim3(p.pat.ng, p.B * p.N), dim3(64), 0, st, p);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  if (p.R > 0) {
    hipLaunchKernelGGL(drel_reduce_kernel<T>, dim3(p.Rp, p.N), dim3(1024), 0, st, p);
    e = hipGetLastError();
  }
  return e;
}

template <typename T, int MODE, bool GEN>
static hipError_t launch_bwd_rp(const BwdParams& p, hipStream_t st) {
  if (p.Rp == 32) return launch_bwd_one<T, MODE, 32, GEN>(p, st);
  if (p.Rp == 64) return launch_bwd_one<T, MODE, 64, GEN>(p, st);
  return launch_bwd_one<T, MODE, 128, GEN>(p, st);
}

template <typename T>
static hipError_t launch_bwd_t(const BwdParams& p, int mode, hipStream_t st) {
  if (mode == kDense) return launch_bwd_rp<T, kDense, true>(p, st);
  const bool gen = !(p.pat.id_mode == 0 || p.perm_1d);
  return gen ? launch_bwd_rp<T, kBand, true>(p, st) : launch_bwd_rp<T, kBand, false>(p, st);
}

hipError_t launch_bwd_dq_combine(const BwdParams& p, bool bf16, hipStream_t st) {
  dim3 grid(p.pat.ng, p.B * p.N);
  if (bf16) hipLaunchKernelGGL(attn_bwd_dq_combine_kernel<__bf16>, grid, dim3(64), 0, st, p);
  else hipLaunchKernelGGL(attn_bwd_dq_combine_kernel<float>, grid, dim3(64), 0, st, p);
  return hipGetLastError();
}
hipError_t launch_bwd_dkv_combine(const BwdParams& p, bool bf16, hipStream_t st) {
  dim3 grid(p.pat.ng, p.B * p.N);
  if (bf16) hipLaunchKernelGGL(attn_bwd_dkv_combine_kernel<__bf16>, grid, dim3(64), 0, st, p);
  else hipLaunchKernelGGL(attn_bwd_dkv_combine_kernel<float>, grid, dim3(64), 0, st, p);
  return hipGetLastError();
}
hipError_t launch_drel_reduce(const BwdParams& p, bool bf16, hipStream_t st) {
  int extra = 0;               // blocks (x >= Rp) that combine the dK/dV partials of the global rows, 16 rows each
  if (p.comb_in_next && p.n_gblk > 0) extra = (p.pat.ng * p.B * p.N + 16 * p.N - 1) / (16 * p.N);
  if (bf16) hipLaunchKernelGGL(drel_reduce_kernel<__bf16>, dim3(p.Rp + extra, p.N), dim3(1024), 0, st, p);
  else hipLaunchKernelGGL(drel_reduce_kernel<float>, dim3(p.Rp + extra, p.N), dim3(1024), 0, st, p);
  return hipGetLastError();
}

hipError_t launch_attn_bwd(const BwdParams& p, int mode, bool bf16, hipStream_t st) {
  if (mode == kBand && bf16 && (p.pat.id_mode == 0 || (p.perm_1d && p.Rp <= 64) || p.lean2d)) return launch_attn_bwd_band_bf16(p, st);
  return bf16 ? launch_bwd_t<__bf16>(p, mode, st) : launch_bwd_t<float>(p, mode, st);
}

}  // namespace mmt
