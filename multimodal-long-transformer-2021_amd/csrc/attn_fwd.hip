// Forward of QkvRelativeAttention for gfx950 (MI355X).
//
// Replaces the dense einsum / one-hot / softmax chain the reference runs behind
// `RelativeTransformerLayers(inputs, att_mask, relative_att_ids)`
// (src/modeling/models/mmt_encoder.py:220-224; math: SURVEY.md App. A.3) with a
// flash-style kernel that never materialises [S,S] tensors:
//
//   one wave = 32 query rows of one (batch, head); it walks 32-key tiles.
//   S^T = K.Q^T by MFMA (keys in registers, query row on the lane), relative scores are
//   gathered from a per-wave LDS table T[q][col(id)] = (q.E[id] + bias[id]) built once per
//   q-block by MFMA, softmax state is lane-local, O^T += V^T.P^T by MFMA with the P
//   accumulator reused as the B operand (no LDS round trip for P) and V^T fragments read
//   with ds_read_b64_tr_b16 from a wave-private, bank-swizzled LDS tile.  The next tile's
//   K fragments and V rows are prefetched into registers while the current tile computes.
//
// Work items of one launch
//   band items : 32 query rows x (band tiles U global-key tiles), final output        [K1]
//   rows items : 32 global query rows x one chunk of keys, partial (O,m,l) output     [K2]
//   (kDense)   : literal reference operator, att_mask / rel_ids int32 [B,S,S] from HBM [K3]
#include "attn_tile.h"

namespace mmt {

// GEN = true: ids/mask through the generic per-element generators (2-D ids, or 1-D ids whose
// vocabulary is smaller than 2m+1); GEN = false: no ids or 1-D ids with the permuted table.
template <typename T, int MODE, int Rp, bool GEN>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const FwdParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  unsigned char* wl = smem + wave * WaveLds<T, Rp>::kBytes;
  float* tab = reinterpret_cast<float*>(wl);
  unsigned char* vlds = wl + WaveLds<T, Rp>::kTBytesAligned;

  // ---- which 32 query rows does this wave own? ------------------------------------
  const int n_tiles = (p.S + 31) >> 5;
  const int nqb = (p.S + 127) >> 7;
  const bool rows_item = MODE == kBand && (int)blockIdx.x >= p.n_band_blocks;
  int bn, q0, chunk = 0, rowblk = 0;
  if (rows_item) {
    const int per_bn = (p.n_chunks * p.n_rowblk + 3) >> 2;
    const int rb = blockIdx.x - p.n_band_blocks;
    bn = rb / per_bn;
    const int item = (rb - bn * per_bn) * 4 + wave;
    if (item >= p.n_chunks * p.n_rowblk) return;
    rowblk = item / p.n_chunks;
    chunk = item - rowblk * p.n_chunks;
    q0 = p.pat.g0 + rowblk * 32;
  } else {
    const int wg = xcd_remap(blockIdx.x, p.n_band_blocks);
    bn = wg / nqb;
    q0 = (wg - bn * nqb) * 128 + wave * 32;
    if (q0 >= p.S) return;
  }
  const int b = bn / p.N, n = bn - b * p.N;
  const int q = q0 + r;
  const bool q_ok = q < p.S;
  const int valid_len = p.valid_len ? p.valid_len[b] : p.S;

  const T* Q = reinterpret_cast<const T*>(p.q) + (long)b * p.qs[0] + (long)n * p.qs[2];
  const T* K = reinterpret_cast<const T*>(p.k) + (long)b * p.ks[0] + (long)n * p.ks[2];
  const T* V = reinterpret_cast<const T*>(p.v) + (long)b * p.vs[0] + (long)n * p.vs[2];
  // row offsets inside one (b, n) plane fit 32 bits (checked on the host)
  const unsigned qs1 = (unsigned)p.qs[1], ks1 = (unsigned)p.ks[1], vs1 = (unsigned)p.vs[1];

  // ---- tile walk: [a0, a0+lenA) U [b0, b0+lenB) U [c0, c0+lenC), ascending ------------
  int a0 = 0, lenA = 0, b0 = 0, lenB = n_tiles, c0 = 0, lenC = 0;
  if (MODE == kBand) {
    if (rows_item) {
      b0 = chunk * p.chunk_tiles;
      lenB = min(n_tiles, b0 + p.chunk_tiles) - b0;
    } else {
      const int lo = max(q0 - p.pat.radius, 0), hi = min(q0 + 31 + p.pat.radius, p.S - 1);
      b0 = lo >> 5;
      const int b1 = hi >> 5;
      lenB = b1 - b0 + 1;
      if (p.pat.ng > 0) {
        const int g_lo = p.pat.g0 >> 5, g_hi = (p.pat.g0 + p.pat.ng - 1) >> 5;
        a0 = g_lo; lenA = max(0, min(g_hi, b0 - 1) - g_lo + 1);
        c0 = max(g_lo, b1 + 1); lenC = max(0, g_hi - c0 + 1);
      }
    }
  }
  const int n_it = lenA + lenB + lenC;
  auto tile_at = [&](int it) {
    return it < lenA ? a0 + it : (it < lenA + lenB ? b0 + (it - lenA) : c0 + (it - lenA - lenB));
  };

  Frag<T> qf;
  qf.load_row(Q + (unsigned)min(q, p.S - 1) * qs1, h);

  // prefetch of the first tile overlaps the table construction
  Frag<T> kf;
  VTile<T> vt;
  {
    const int k0 = tile_at(0) * 32;
    kf.load_row(K + (unsigned)min(k0 + r, p.S - 1) * ks1, h);
    vt.load(V, vs1, k0, p.S, lane, 0);
  }

  // ---- relative-score table T[q][col(id)] = (q.E[id] + bias[id]) * tscale  (log2 domain) --
  const int id_mode = p.pat.id_mode, mdist = p.pat.m;
  if (p.R > 0) {
    const T* E = reinterpret_cast<const T*>(p.emb) + (long)n * 64;
#pragma unroll
    for (int rb = 0; rb < Rp / 32; ++rb) {
      const int rr = rb * 32 + r;
      Frag<T> ef;
      ef.load_row(E + (long)min(rr, p.R - 1) * p.N * 64, h);   // columns >= R are never read
      f32x16 c = {0};
      c = mma_rows(ef, qf, c);  // [id x q]
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int id = rb * 32 + kap(i, h);
        float bias = 0.f;
        if (p.bias) bias = (float)reinterpret_cast<const T*>(p.bias)[(long)min(id, p.R - 1) * p.N + n];
        const int col = (MODE == kDense || GEN) ? id : tcol(p.perm_1d, mdist, id);
        tab[r * kTStride(Rp) + col] = (c[i] + bias) * p.tscale;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  f32x16 o0 = {0}, o1 = {0};
  float m_run = -INFINITY, l_run = 0.f;
  const float* trow = tab + r * kTStride(Rp);
  // whole q-block on one side of valid_len?  (needed for the fast path)
  const bool qblk_valid = q0 + 31 < valid_len, qblk_pad = q0 >= valid_len;
  const bool qblk_plain = q0 + 31 < p.S && !(p.pat.ng > 0 && q0 + 31 >= p.pat.g0 && q0 < p.pat.g0 + p.pat.ng);

  for (int it = 0; it < n_it; ++it) {
    const int k0 = tile_at(it) * 32;
    vt.to_lds(vlds, lane);

    f32x16 c = {0};
    c = mma_rows(kf, qf, c);

    VTile<T> vcur;
    if constexpr (sizeof(T) == 4) vcur = vt;
    if (it + 1 < n_it) {   // prefetch the next tile (registers) under this tile's math
      const int k1 = tile_at(it + 1) * 32;
      kf.load_row(K + (unsigned)min(k1 + r, p.S - 1) * ks1, h);
      vt.load(V, vs1, k1, p.S, lane, 0);
    }

    // ---- scores in the log2 domain --------------------------------------------------------
    float s2[16];
    float tmax = -INFINITY;
    bool fast = false;
    if (MODE == kBand) {
      // wave-uniform classification: every (q,k) of the tile unmasked and 1-D (or no) ids
      const bool seg_all = (qblk_valid && k0 + 31 < valid_len) || (qblk_pad && k0 >= valid_len);
      const bool band_all = (k0 - (q0 + 31) >= -p.pat.radius) && (k0 + 31 - q0 <= p.pat.radius);
      fast = !GEN && seg_all && band_all && qblk_plain && k0 + 31 < p.S;
    }
    constexpr bool cheap_ids = MODE == kBand && !GEN;
    if (!GEN && fast) {
      const int d0 = k0 - q + 4 * h;
      if (id_mode == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int d = d0 + (i & 3) + 8 * (i >> 2);
          const int col = min(max(d, -mdist), mdist) + mdist;  // < R: perm_1d implies R >= 2m+1
          s2[i] = fmaf(c[i], p.sscale, trow[col]);
          tmax = fmaxf(tmax, s2[i]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) { s2[i] = c[i] * p.sscale; tmax = fmaxf(tmax, s2[i]); }
      }
    } else if constexpr (cheap_ids) {
      // branch-free pattern mask, 1-D (permuted table) or no ids
      const int kb = k0 + 4 * h, d0 = kb - q;
      const unsigned W = (unsigned)p.pat.radius;
      const bool qv = q < valid_len, gq = is_global(p.pat, q);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ci = (i & 3) + 8 * (i >> 2);
        const int kk = kb + ci, d = d0 + ci;
        const bool near = (unsigned)(d + (int)W) <= 2u * W;
        const bool gk = (unsigned)(kk - p.pat.g0) < (unsigned)p.pat.ng;
        const bool seg = (kk < valid_len) == qv;
        const bool keep = (int)seg & ((int)near | (int)gk | (int)gq);
        float rel = 0.f;
        if (id_mode == 1) rel = trow[min(max(d, -mdist), mdist) + mdist];
        float s = fmaf(c[i], p.sscale, rel);
        s = keep ? s : s + p.mask_add;
        s = kk < p.S ? s : -INFINITY;
        s2[i] = s;
        tmax = fmaxf(tmax, s);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int kk = k0 + kap(i, h);
        bool keep;
        int id = -1;
        if (MODE == kDense) {
          const long off = ((long)b * p.S + (q_ok ? q : 0)) * p.S + (kk < p.S ? kk : 0);
          keep = p.att_mask ? p.att_mask[off] != 0 : true;
          if (p.rel_ids) id = p.rel_ids[off];
        } else {
          keep = pattern_mask(p.pat, valid_len, q, kk);
          if (id_mode) id = rel_id(p.pat, q, kk);
        }
        float rel = 0.f;
        if ((unsigned)id < (unsigned)p.R) rel = trow[id];
        float s = fmaf(c[i], p.sscale, rel);
        if (!keep) s += p.mask_add;
        if (kk >= p.S) s = -INFINITY;
        s2[i] = s;
        tmax = fmaxf(tmax, s);
      }
    }
    tmax = half_max(tmax);
    // Deferred rescale: the running reference m_run only moves when some row's tile maximum
    // exceeds it by more than kRescaleThr (log2 units), so p stays <= 2^kRescaleThr; O, l and
    // p always share one reference, hence the normalised result is unchanged.
    if (__any(tmax > m_run + kRescaleThr)) {
      const float m_new = fmaxf(m_run, tmax);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
    }
    float psum = 0.f;
    float pr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      pr[i] = __builtin_amdgcn_exp2f(s2[i] - m_run);
      psum += pr[i];
    }
    l_run += psum;

    if (p.drop_thresh) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const uint32_t bits = drop_bits16(drop_row_base(effective_seed(p.seed_lo, p.seed_hi, p.epoch).lo, effective_seed(p.seed_lo, p.seed_hi, p.epoch).hi, (uint32_t)bn, (uint32_t)q),
                                          (uint32_t)(k0 + kap(i, h)));
        pr[i] = bits >= p.drop_thresh ? pr[i] * p.inv_keep : 0.f;
      }
    }

    // ---- O^T[d x q] += V^T[d x key] . P^T[key x q] -------------------------------------------
    if constexpr (sizeof(T) == 2) mma_xt(o0, o1, vt, vlds, pr, lane);
    else mma_xt(o0, o1, vcur, vlds, pr, lane);
  }

  // ---- epilogue --------------------------------------------------------------------------
  const float l_tot = half_sum(l_run);
  if (rows_item) {
    // partial, unnormalised: part_o[bn][rowblk][chunk][q 32][d 64], part_ml[...][2][32]
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
  if (MODE == kBand && p.skip_global_rows && is_global(p.pat, q)) return;
  const float inv = 1.f / l_tot;
  T* O = reinterpret_cast<T*>(p.out) + (long)b * p.os[0] + (long)q * p.os[1] + (long)n * p.os[2];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int d = 8 * g + 4 * h;
    if constexpr (sizeof(T) == 2) {
      bf16x4 a, c2;
#pragma unroll
      for (int j = 0; j < 4; ++j) { a[j] = (__bf16)(o0[4 * g + j] * inv); c2[j] = (__bf16)(o1[4 * g + j] * inv); }
      *reinterpret_cast<bf16x4*>(O + d) = a;
      *reinterpret_cast<bf16x4*>(O + 32 + d) = c2;
    } else {
      *reinterpret_cast<f32x4*>(O + d) = f32x4{o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv};
      *reinterpret_cast<f32x4*>(O + 32 + d) = f32x4{o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv};
    }
  }
  if (p.lse && h == 0) p.lse[((long)b * p.N + n) * p.S + q] = (m_run + log2f(l_tot)) * kLn2;
}

// Combine the per-chunk partials of the global rows: one wave per (row, bn), lane = d.
// The chunk statistics are fetched one chunk per LANE (one load instruction, wave reductions) and the
// partial rows with all loads of a group of eight in flight -- a loop of dependent scalar loads made this
// 2 KB-per-wave kernel take 10 us, one L2 latency per chunk.
template <typename T>
__global__ __launch_bounds__(64) void attn_rows_combine_kernel(const FwdParams p) {
  const int bn = blockIdx.y;
  const int row = blockIdx.x;  // 0 .. ng-1
  const int d = threadIdx.x;
  const int rowblk = row >> 5, rr = row & 31;
  const int b = bn / p.N, n = bn - b * p.N;
  const long slot0 = ((long)bn * p.n_rowblk + rowblk) * p.n_chunks;
  float L = 0.f, acc = 0.f, M = -INFINITY;
  for (int c0 = 0; c0 < p.n_chunks; c0 += 64) {             // 64 chunks per pass (one pass in practice)
    const int c = c0 + d;
    const bool live = c < p.n_chunks;
    const float mc = live ? p.part_ml[(slot0 + c) * 64 + rr] : -INFINITY;
    const float lc = live ? p.part_ml[(slot0 + c) * 64 + 32 + rr] : 0.f;
    float Mn = mc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) Mn = fmaxf(Mn, __shfl_xor(Mn, o, 64));
    Mn = fmaxf(Mn, M);
    const float rescale = exp2f(M - Mn);                    // 0 on the first pass (M = -inf)
    const float wc = live ? exp2f(mc - Mn) : 0.f;
    float ls = wc * lc;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ls += __shfl_xor(ls, o, 64);
    L = L * rescale + ls;
    acc *= rescale;
    M = Mn;
    const int cnt = min(64, p.n_chunks - c0);
    const float* po = p.part_o + (slot0 + c0) * (32 * 64) + rr * 64 + d;
    int i = 0;
    for (; i + 8 <= cnt; i += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = po[(long)(i + u) * (32 * 64)];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = fmaf(__shfl(wc, i + u, 64), v[u], acc);
    }
    for (; i < cnt; ++i) acc = fmaf(__shfl(wc, i, 64), po[(long)i * (32 * 64)], acc);
  }
  const int q = p.pat.g0 + row;
  T* O = reinterpret_cast<T*>(p.out) + (long)b * p.os[0] + (long)q * p.os[1] + (long)n * p.os[2];
  O[d] = (T)(acc * p.part_scale / L);
  if (p.lse && d == 0) p.lse[((long)b * p.N + n) * p.S + q] = (M + log2f(L)) * kLn2;
}

// ------------------------------------ launchers -----------------------------------------
template <typename T, int MODE, int Rp, bool GEN>
static hipError_t launch_one(const FwdParams& p, dim3 grid, hipStream_t st) {
  const int lds = 4 * WaveLds<T, Rp>::kBytes;
  if (lds > 64 * 1024)               // (the 128-wide table: relative vocabularies of 65..128 ids)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<T, MODE, Rp, GEN>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((attn_fwd_kernel<T, MODE, Rp, GEN>), grid, dim3(256), lds, st, p);
  return hipGetLastError();
}

template <typename T, int MODE, bool GEN>
static hipError_t launch_rp(const FwdParams& p, dim3 grid, hipStream_t st) {
  if (p.R <= 32) return launch_one<T, MODE, 32, GEN>(p, grid, st);
  if (p.R <= 64) return launch_one<T, MODE, 64, GEN>(p, grid, st);
  return launch_one<T, MODE, 128, GEN>(p, grid, st);
}

template <typename T>
static hipError_t launch_t(const FwdParams& p, int mode, dim3 grid, hipStream_t st) {
  if (mode == kDense) return launch_rp<T, kDense, true>(p, grid, st);
  const bool gen = !(p.pat.id_mode == 0 || p.perm_1d);
  return gen ? launch_rp<T, kBand, true>(p, grid, st) : launch_rp<T, kBand, false>(p, grid, st);
}

hipError_t launch_attn_fwd(const FwdParams& p, int mode, bool bf16, hipStream_t st) {
  // band items first, then (kBand only) the global-row items of the same launch
  const int per_bn = (p.n_chunks * p.n_rowblk + 3) / 4;
  dim3 grid(p.n_band_blocks + (mode == kBand ? per_bn * p.B * p.N : 0));
  return bf16 ? launch_t<__bf16>(p, mode, grid, st) : launch_t<float>(p, mode, grid, st);
}

hipError_t launch_rows_combine(const FwdParams& p, bool bf16, hipStream_t st) {
  dim3 grid(p.pat.ng, p.B * p.N);
  if (bf16) hipLaunchKernelGGL(attn_rows_combine_kernel<__bf16>, grid, dim3(64), 0, st, p);
  else hipLaunchKernelGGL(attn_rows_combine_kernel<float>, grid, dim3(64), 0, st, p);
  return hipGetLastError();
}

}  // namespace mmt
