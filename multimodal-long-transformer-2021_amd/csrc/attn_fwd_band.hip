// VALU-lean forward kernel for the common case: bf16 I/O, band + global pattern, relative ids
// none or 1-D with the permuted table (R >= 2m+1).  Same math and work decomposition as
// attn_fwd_kernel<kBand> (attn_fwd.hip) -- that one stays the general path (fp32, 2-D ids,
// dense inputs) -- but with everything wave-uniform hoisted out of the per-element code:
//
//   * tile classes chosen by scalar code, one straight-line element loop per class:
//       A  no mask, one clipped id for the whole tile   p = exp2(fma(c, s, rel_const - m))
//       B  no mask, mixed ids                           gather T[q][clamp(k-q)] with one v_med3
//       D  band edge (|k-q| <= W test only), clipped id
//       C  anything else (pad boundary, global keys, sequence end): branch-free general mask
//   * K / V / Q / E through raw buffer loads: per-lane offset computed once, the tile offset
//     rides in the scalar offset, rows past the end read as zeros (no clamps, no 64-bit VALU);
//   * compile-time HAS_REL; relative bias row staged through LDS once per q-block;
//   * dropout: 16 random bits per element, one 32-bit mix per key pair.
#include "attn_lean.h"

namespace mmt {

template <int Rp, int REL>        // REL: 0 no relative term, 1 = 1-D ids (permuted table), 2 = 2-D ids (columns in id order)
__global__ __launch_bounds__(256, 3) void attn_fwd_band_bf16_kernel(const FwdParams p) {
  using T = __bf16;
  constexpr bool HAS_REL = REL != 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  constexpr int kWaveBytes = WaveLds<T, Rp>::kBytes;             // T table, one tile buffer: the K tile, then (its row
  unsigned char* wl = smem + wave * kWaveBytes;                  // fragments read) the V tile -- LDS serves a wave's
  float* tab = reinterpret_cast<float*>(wl);                     // accesses in order, so the two never meet, and three
  unsigned char* vlds = wl + WaveLds<T, Rp>::kTBytesAligned;     // workgroups fit a compute unit at the 64-wide table too
  unsigned char* klds = vlds;
  int* lut = reinterpret_cast<int*>(smem + 4 * kWaveBytes) + wave * ((lut2d_entries(p.pat) + 15) & ~15);   // REL == 2: wave-private

  // ---- work item ------------------------------------------------------------------------
  const int n_tiles = (p.S + 31) >> 5, nqb = (p.S + 127) >> 7;
  const int per_bn = (p.n_chunks * p.n_rowblk + 3) >> 2;     // global-row blocks per plane
  int bn, q0, chunk = 0, rowblk = 0;
  bool rows_item;
  {
    int blk;
    plane_major_map(blockIdx.x, p.B * p.N, per_bn, p.rows_only ? 0 : nqb, bn, blk);
    rows_item = blk < per_bn;
    if (rows_item) {
      const int item = blk * 4 + wave;
      if (item >= p.n_chunks * p.n_rowblk) return;
      rowblk = item / p.n_chunks;
      chunk = item - rowblk * p.n_chunks;
      q0 = p.pat.g0 + rowblk * 32;
    } else {
      q0 = (blk - per_bn) * 128 + wave * 32;
      if (q0 >= p.S) return;
    }
  }
  const int b = bn / p.N, n = bn - b * p.N;
  const int q = q0 + r;
  const bool q_ok = q < p.S;
  const int valid_len = p.valid_len ? p.valid_len[b] : p.S;
  const int W = p.pat.radius, m = p.pat.m;
  const bool ignore_band = rows_item;      // rows of global tokens see every key of their segment

  // ---- buffer descriptors (wave-uniform bases; rows past the end read as zeros) ------------
  const unsigned qs1b = (unsigned)p.qs[1] * 2, ks1b = (unsigned)p.ks[1] * 2, vs1b = (unsigned)p.vs[1] * 2;
  const T* Qb = reinterpret_cast<const T*>(p.q) + (long)b * p.qs[0] + (long)n * p.qs[2];
  const T* Kb = reinterpret_cast<const T*>(p.k) + (long)b * p.ks[0] + (long)n * p.ks[2];
  const T* Vb = reinterpret_cast<const T*>(p.v) + (long)b * p.vs[0] + (long)n * p.vs[2];
  const auto rq = make_rsrc(Qb, (unsigned)(p.S - 1) * qs1b + 128);
  const auto rk = make_rsrc(Kb, (unsigned)(p.S - 1) * ks1b + 128);
  const auto rv = make_rsrc(Vb, (unsigned)(p.S - 1) * vs1b + 128);
  const unsigned voff_q = (unsigned)r * qs1b + 64 * h;
  // K tiles are loaded coalesced (8 rows x 128 B per instruction) and their row fragments read back from a
  // wave-private LDS tile: a fragment-shaped global load touches every 128-byte line of the tile four times
  const unsigned voff_k = (unsigned)(lane >> 3) * ks1b + (lane & 7) * 16;
  const unsigned voff_v = (unsigned)(lane >> 3) * vs1b + (lane & 7) * 16;

  // ---- tile walk --------------------------------------------------------------------------
  int a0 = 0, lenA = 0, b0 = 0, lenB = n_tiles, c0 = 0, lenC = 0;
  if (rows_item) {
    b0 = chunk * p.chunk_tiles;
    lenB = min(n_tiles, b0 + p.chunk_tiles) - b0;
  } else {
    const int lo = max(q0 - W, 0), hi = min(q0 + 31 + W, p.S - 1);
    b0 = lo >> 5;
    const int b1 = hi >> 5;
    lenB = b1 - b0 + 1;
    if (p.pat.ng > 0) {
      const int g_lo = p.pat.g0 >> 5, g_hi = (p.pat.g0 + p.pat.ng - 1) >> 5;
      a0 = g_lo; lenA = max(0, min(g_hi, b0 - 1) - g_lo + 1);
      c0 = max(g_lo, b1 + 1); lenC = max(0, g_hi - c0 + 1);
    }
  }
  const int n_it = lenA + lenB + lenC;
  auto tile_at = [&](int it) {
    return it < lenA ? a0 + it : (it < lenA + lenB ? b0 + (it - lenA) : c0 + (it - lenA - lenB));
  };

  Frag<T> qf;
  bf16x8 kt[4], vt[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) qf.v[s] = buf16(rq, voff_q + 16 * s, (unsigned)q0 * qs1b);
  {
    const unsigned k0 = (unsigned)tile_at(0) * 32;
#pragma unroll
    for (int u = 0; u < 4; ++u) kt[u] = buf16(rk, voff_k, (k0 + 8 * u) * ks1b);
#pragma unroll
    for (int u = 0; u < 4; ++u) vt[u] = buf16(rv, voff_v, (k0 + 8 * u) * vs1b);
  }

  // ---- relative-score table (log2 domain), bias row through LDS -----------------------------
  float relfn = 0.f, relfp = 0.f;     // the two clipped columns of this lane's row
  if (HAS_REL) {
    // all global loads of the prologue are issued before anything waits: E fragments first, then the bias
    // row (its LDS round trip used to sit in front of the E loads: one extra memory latency per wave)
    const T* Eb = reinterpret_cast<const T*>(p.emb) + (long)n * 64;
    const unsigned es1b = (unsigned)p.N * 128;
    const auto re = make_rsrc(Eb, (unsigned)(p.R - 1) * es1b + 128);
    Frag<T> ef[Rp / 32];
#pragma unroll
    for (int rb = 0; rb < Rp / 32; ++rb)
#pragma unroll
      for (int s = 0; s < 4; ++s) ef[rb].v[s] = buf16(re, (unsigned)(REL == 2 ? rb * 32 + r : icol(m, rb * 32 + r)) * es1b + 64 * h + 16 * s, 0u);   // row r <- id of column rb*32 + r
    float* bias_ts = reinterpret_cast<float*>(vlds);
    if (lane < Rp) {
      const int idc = REL == 2 ? lane : icol(m, lane);
      bias_ts[lane] = (p.bias && idc < p.R) ? (float)reinterpret_cast<const T*>(p.bias)[(long)idc * p.N + n] * p.tscale : 0.f;   // by column
    }
    if (REL == 2) {
      build_lut2d<Rp>(lut, p.pat, p.R, lane, 64);
      tab[r * kTStride(Rp) + kZeroCol(Rp)] = 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int rb = 0; rb < Rp / 32; ++rb) {
      f32x16 c = {0};
      c = mma_rows(ef[rb], qf, c);   // [id x q]
      float bv[16];                  // the sixteen bias values in one batch, then the stores: interleaved, hipcc keeps every
#pragma unroll                       // read behind the store before it (one LDS round trip each: it cannot tell the arrays apart)
      for (int i = 0; i < 16; ++i) bv[i] = bias_ts[rb * 32 + kap(i, h)];
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(bv[i]));
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int col = rb * 32 + kap(i, h);
        tab[r * kTStride(Rp) + col] = fmaf(c[i], p.tscale, bv[i]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (REL == 1) {
      relfn = tab[r * kTStride(Rp)];
      relfp = tab[r * kTStride(Rp) + 2 * m];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  f32x16 o0 = {0}, o1 = {0};
  float m_run = -INFINITY, l_run = 0.f;
  const float* trow = tab + r * kTStride(Rp);
  const int trow_addr = (int)(unsigned)(size_t)(__attribute__((address_space(3))) float*)(tab + r * kTStride(Rp));
  const bool qblk_valid = q0 + 31 < valid_len, qblk_pad = q0 >= valid_len, qblk_in = q0 + 31 < p.S;
  const SeedPair sd = effective_seed(p.seed_lo, p.seed_hi, p.epoch);
  const uint32_t drop_base = drop_row_base(sd.lo, sd.hi, (uint32_t)bn, (uint32_t)q);
  // REL == 2: this lane's query on the patch grid, and the LDS address of the look-up table
  const int xq2 = (int)__umulhi((unsigned)q, p.pat.magicP), yq2 = q - xq2 * p.pat.P;
  const int lut_addr = lds_addr(lut);
  const int lim2 = p.pat.r + 1, nlim2 = -lim2;

  for (int it = 0; it < n_it; ++it) {
    const int k0 = tile_at(it) * 32;
    // K rows of this tile -> the wave-private LDS tile, row fragments back; then the V rows into the same tile (read back
    // transposed after the softmax)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ci = lane + 64 * u, row = ci >> 3, ch = ci & 7;
      const int off = row * 128 + ((((ch >> 2) ^ ((row >> 1) & 1))) << 6) + (ch & 3) * 16;
      *reinterpret_cast<bf16x8*>(klds + off) = kt[u];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    Frag<T> kf;
    frag_from_tile(kf, klds, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ci = lane + 64 * u, row = ci >> 3, ch = ci & 7;
      const int off = row * 128 + ((((ch >> 2) ^ ((row >> 1) & 1))) << 6) + (ch & 3) * 16;
      *reinterpret_cast<bf16x8*>(vlds + off) = vt[u];
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    f32x16 c = {0};
    c = mma_rows(kf, qf, c);     // S^T [key x q]
    if (it + 1 < n_it) {         // prefetch the next tile under this tile's math
      const unsigned k1 = (unsigned)tile_at(it + 1) * 32;
#pragma unroll
      for (int u = 0; u < 4; ++u) kt[u] = buf16(rk, voff_k, (k1 + 8 * u) * ks1b);
#pragma unroll
      for (int u = 0; u < 4; ++u) vt[u] = buf16(rv, voff_v, (k1 + 8 * u) * vs1b);
    }

    // ---- wave-uniform tile class -------------------------------------------------------------
    const int dmin = k0 - (q0 + 31), dmax = k0 + 31 - q0;
    const bool in_range = (k0 + 31 < p.S) && qblk_in;
    const bool seg_all = (qblk_valid && k0 + 31 < valid_len) || (qblk_pad && k0 >= valid_len);
    const bool band_all = ignore_band || (dmin >= -W && dmax <= W);
    const bool plain = in_range && seg_all && band_all;
    const bool far_neg = dmax <= -m, far_pos = dmin >= m;
    const bool one_id = REL != 2 && (!HAS_REL || far_neg || far_pos);
    const bool no_gkey = p.pat.ng == 0 || k0 + 31 < p.pat.g0 || k0 >= p.pat.g0 + p.pat.ng;
    const float relc = HAS_REL ? (far_neg ? relfn : relfp) : 0.f;
    const int dbase = k0 - q + 4 * h;

    float pr[16], s2[16];
    if (REL == 2) {
      // ---- 2-D ids: the relative term first (image x image tiles through the look-up table, anything else with
      //      the general id function), then the mask by tile class
      float rel[16];
      if (q0 + 31 < p.pat.I && k0 + 31 < p.pat.I && p.pat.P >= 32) {
        const Ids2dTile t2 = ids2d_tile<1>(p.pat, lut_addr, k0 + 4 * h, xq2, yq2);
#pragma unroll
        for (int i = 0; i < 16; ++i)
          rel[i] = *(lds_cfp)(size_t)(unsigned)(trow_addr + ids2d_col4<1>(t2, (i & 3) + 8 * (i >> 2), nlim2, lim2));
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) rel[i] = trow[col2d<Rp>(p.pat, p.R, q, k0 + 4 * h + (i & 3) + 8 * (i >> 2))];
      }
      if (plain) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s2[i] = fmaf(c[i], p.sscale, rel[i]);
      } else if (in_range && seg_all && no_gkey && !ignore_band) {          // band edge
        const unsigned W2 = 2u * (unsigned)W;
#pragma unroll
        for (int i = 0; i < 16; ++i)
          s2[i] = fmaf(c[i], p.sscale, rel[i]) + ((unsigned)(dbase + (i & 3) + 8 * (i >> 2) + W) <= W2 ? 0.f : p.mask_add);
      } else if (in_range && seg_all && !ignore_band && (dmin > W || dmax < -W)) {   // only the tile's global keys
        const unsigned gb = (unsigned)(k0 + 4 * h - p.pat.g0), ng = (unsigned)p.pat.ng;
#pragma unroll
        for (int i = 0; i < 16; ++i)
          s2[i] = fmaf(c[i], p.sscale, rel[i]) + (gb + (unsigned)((i & 3) + 8 * (i >> 2)) < ng ? 0.f : p.mask_add);
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
          float s = fmaf(c[i], p.sscale, rel[i]);
          s = keep ? s : s + p.mask_add;
          s2[i] = kk < p.S ? s : -INFINITY;
        }
      }
    } else if (plain && one_id) {                              // ---- class A
#pragma unroll
      for (int i = 0; i < 16; ++i) s2[i] = fmaf(c[i], p.sscale, relc);
    } else if (plain) {                                        // ---- class B (HAS_REL, mixed ids)
      // Register i holds key offset ci = (i & 3) + 8 (i >> 2) (+ 4 in the upper half-wave) of the tile:
      // d = o + ci (+4) - r with o = k0 - q0 and r = 0..31.  In the two tiles next to the diagonal one, half of the
      // registers are beyond the clip distance for EVERY lane and take the clipped constant without a gather:
      // o >= m + 15 -> registers 8..15 (ci >= 16) have d >= m;  o <= -(m + 15) -> registers 0..7 (ci <= 11) have d <= -m.
      const int abase = trow_addr + 4 * (m + dbase), alo = trow_addr, ahi = trow_addr + 8 * m;
      const int o = k0 - q0;
      if (o >= m + 15) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int a = med3i(abase + 4 * ((i & 3) + 8 * (i >> 2)), alo, ahi);
          s2[i] = fmaf(c[i], p.sscale, *(lds_cfp)(size_t)(unsigned)a);
        }
#pragma unroll
        for (int i = 8; i < 16; ++i) s2[i] = fmaf(c[i], p.sscale, relfp);
      } else if (o <= -(m + 15)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) s2[i] = fmaf(c[i], p.sscale, relfn);
#pragma unroll
        for (int i = 8; i < 16; ++i) {
          const int a = med3i(abase + 4 * ((i & 3) + 8 * (i >> 2)), alo, ahi);
          s2[i] = fmaf(c[i], p.sscale, *(lds_cfp)(size_t)(unsigned)a);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int a = med3i(abase + 4 * ((i & 3) + 8 * (i >> 2)), alo, ahi);
          s2[i] = fmaf(c[i], p.sscale, *(lds_cfp)(size_t)(unsigned)a);
        }
      }
    } else if (in_range && seg_all && no_gkey && one_id) {     // ---- class D (band edge)
      // masked <=> d = dbase + ci outside [-W, W]; when only one side can fail in this tile: one compare of the
      // literal ci against a per-lane bound, one select between the row constant with and without the additive
      // mask, one fma
      const float relm = relc + p.mask_add;
      if (dmin >= -W) {                    // only d > W can fail: masked iff ci > W - dbase
        const int bound = W - dbase;
#pragma unroll
        for (int i = 0; i < 16; ++i) s2[i] = fmaf(c[i], p.sscale, ((i & 3) + 8 * (i >> 2)) > bound ? relm : relc);
      } else if (dmax <= W) {              // only d < -W can fail: masked iff ci < -W - dbase
        const int bound = -W - dbase;
#pragma unroll
        for (int i = 0; i < 16; ++i) s2[i] = fmaf(c[i], p.sscale, ((i & 3) + 8 * (i >> 2)) < bound ? relm : relc);
      } else {                             // a radius below the tile size: both sides
        const unsigned W2 = 2u * (unsigned)W;
#pragma unroll
        for (int i = 0; i < 16; ++i)
          s2[i] = fmaf(c[i], p.sscale, (unsigned)(dbase + (i & 3) + 8 * (i >> 2) + W) <= W2 ? relc : relm);
      }
    } else if (in_range && seg_all && one_id && !ignore_band && (dmin > W || dmax < -W)) {   // ---- class G
      // a tile that lies wholly outside the band: only its global keys are visible (every band wave meets one)
      const unsigned gb = (unsigned)(k0 + 4 * h - p.pat.g0), ng = (unsigned)p.pat.ng;
      const float relm = relc + p.mask_add;
#pragma unroll
      for (int i = 0; i < 16; ++i)
        s2[i] = fmaf(c[i], p.sscale, gb + (unsigned)((i & 3) + 8 * (i >> 2)) < ng ? relc : relm);
    } else {                                                   // ---- class C (general)
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
        s2[i] = kk < p.S ? s : -INFINITY;
      }
    }
    // one definition of s2 / pr for every class: no register copies at the joins
    float tmax = fmaxf(fmaxf(s2[0], s2[1]), s2[2]);
#pragma unroll
    for (int i = 3; i < 15; i += 2) tmax = fmaxf(fmaxf(tmax, s2[i]), s2[i + 1]);
    tmax = fmaxf(tmax, s2[15]);
    tmax = half_max(tmax);
    if (__any(tmax > m_run + kRescaleThr)) {                   // deferred rescale, one site
      const float m_new = fmaxf(m_run, tmax);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) pr[i] = __builtin_amdgcn_exp2f(s2[i] - m_run);
    float psum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) psum += pr[i];
    l_run += psum;

    if (p.drop_thresh) {                   // 16 bits per element, one hash per key pair; 1 / keep in the epilogue
      const uint32_t t16 = p.drop_thresh;
      const uint32_t kc = ((uint32_t)(k0 >> 1) + 2u * (uint32_t)h) * kDropPairMul;   // pair index of kap(0, h)
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const uint32_t hsh = drop_pair_finish(drop_base, kc + (uint32_t)(4 * (i >> 2) + ((i & 3) >> 1)) * kDropPairMul);
        pr[i] = (hsh & 0xFFFFu) >= t16 ? pr[i] : 0.f;
        pr[i + 1] = (hsh >> 16) >= t16 ? pr[i + 1] : 0.f;
      }
    }
    mma_xt(o0, o1, VTile<T>{}, vlds, pr, lane);   // O^T[d x q] += V^T[d x key] . P^T[key x q]
  }

  // ---- epilogue ---------------------------------------------------------------------------------
  const float l_tot = half_sum(l_run);
  if (rows_item) {
    const long slot = ((long)bn * p.n_rowblk + rowblk) * p.n_chunks + chunk;
    float* po = p.part_o + slot * (32 * 64) + r * 64;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      *reinterpret_cast<f32x4*>(po + 8 * g + 4 * h) = f32x4{o0[4 * g], o0[4 * g + 1], o0[4 * g + 2], o0[4 * g + 3]};
      *reinterpret_cast<f32x4*>(po + 32 + 8 * g + 4 * h) = f32x4{o1[4 * g], o1[4 * g + 1], o1[4 * g + 2], o1[4 * g + 3]};
    }
    if (h == 0) {
      p.part_ml[slot * 64 + r] = m_run;
      p.part_ml[slot * 64 + 32 + r] = l_tot;
    }
    return;
  }
  if (!q_ok) return;
  if (p.skip_global_rows && is_global(p.pat, q)) return;
  const float inv = (p.drop_thresh ? p.inv_keep : 1.f) / l_tot;
  T* O = reinterpret_cast<T*>(p.out) + (long)b * p.os[0] + (long)q * p.os[1] + (long)n * p.os[2];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int d = 8 * g + 4 * h;
    bf16x4 x, y;
#pragma unroll
    for (int j = 0; j < 4; ++j) { x[j] = (__bf16)(o0[4 * g + j] * inv); y[j] = (__bf16)(o1[4 * g + j] * inv); }
    *reinterpret_cast<bf16x4*>(O + d) = x;
    *reinterpret_cast<bf16x4*>(O + 32 + d) = y;
  }
  if (p.lse && h == 0) p.lse[((long)b * p.N + n) * p.S + q] = (m_run + log2f(l_tot)) * kLn2;
}

hipError_t launch_attn_fwd_band_bf16(const FwdParams& p, hipStream_t st) {
  const int per_bn = (p.n_chunks * p.n_rowblk + 3) / 4;
  dim3 grid((p.rows_only ? 0 : p.n_band_blocks) + per_bn * p.B * p.N);
  const int rel = p.R > 0 ? p.pat.id_mode : 0;
  if (rel == 2) {                    // 2-D ids: table width chosen by the host (lean_rp), one look-up table per wave
    const int n2 = 2 * p.pat.r + 3, lut_bytes = 4 * 4 * ((n2 * n2 + 15) & ~15);
    if (p.lean_rp == 32) {
      hipLaunchKernelGGL((attn_fwd_band_bf16_kernel<32, 2>), grid, dim3(256), 4 * (WaveLds<__bf16, 32>::kBytes) + lut_bytes, st, p);
    } else {
      const int lds = 4 * (WaveLds<__bf16, 64>::kBytes) + lut_bytes;
      if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_band_bf16_kernel<64, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      hipLaunchKernelGGL((attn_fwd_band_bf16_kernel<64, 2>), grid, dim3(256), lds, st, p);
    }
    return hipGetLastError();
  }
  if (p.R <= 32) {
    const int lds = 4 * (WaveLds<__bf16, 32>::kBytes);
    if (rel) hipLaunchKernelGGL((attn_fwd_band_bf16_kernel<32, 1>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((attn_fwd_band_bf16_kernel<32, 0>), grid, dim3(256), lds, st, p);
  } else {
    const int lds = 4 * (WaveLds<__bf16, 64>::kBytes);
    if (rel) hipLaunchKernelGGL((attn_fwd_band_bf16_kernel<64, 1>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((attn_fwd_band_bf16_kernel<64, 0>), grid, dim3(256), lds, st, p);
  }
  return hipGetLastError();
}

}  // namespace mmt
