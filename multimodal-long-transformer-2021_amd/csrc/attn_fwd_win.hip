// Forward kernel with a workgroup-shared K / V WINDOW (bf16, band + contiguous global range, relative ids none
// or 1-D with the permuted table, R <= 32, radius <= 64).  Same operator, outputs and tile classes as
// attn_fwd_band_bf16_kernel (attn_fwd_band.hip, which stays the path for every other shape) -- what changes is
// how K and V reach the matrix cores:
//
//   * a workgroup = 128 query rows of one (batch, head) plane = 8 waves: TWO waves per 32-row block.  The pair
//     splits the block's key tiles (wave A: table build, peeled global keys, the first two band tiles; wave B: the
//     other three) and merges its two (m, l, O) partials through LDS at the end -- each wave normalises and stores
//     one 32-column half of the rows.  Without a global load in the tile loop a wave is bound by its own
//     instruction issue (one vector instruction per ~4 cycles per wave): sixteen waves per CU instead of eight
//     fill the issue slots of a SIMD that two waves left 60 % idle (profiles/r03 notes in DESIGN.md).
//   * The 8 key tiles the block's band can touch
//     (rows q0 - 64 .. q0 + 191, 2 x 32 KiB) are fetched ONCE per workgroup by LDS-DMA (global_load_lds_dwordx4,
//     1 KiB per wave-instruction, the tile image's 64-byte swizzle applied on the source address) while the waves
//     load their Q rows and build their relative-score tables; after one barrier the tile loop of a wave has no
//     global load, no LDS write and no wait on memory left in it -- the old kernel staged every tile once per WAVE
//     (2.5 x the bytes through the texture path and LDS write port) one tile ahead, and its waves spent their
//     life waiting for that prefetch.
//   * the global keys are not a tile of the walk any more: their K / V rows (8 per group) sit in LDS next to the
//     window and every band wave runs one PEELED step per group before its band tiles -- lane = query row as
//     everywhere, but only the 4 accumulator registers that hold a group's keys are computed (1/4 of a tile's
//     VALU work, half of its PV MFMAs).  Global keys that fall inside a wave's band tiles are handled there
//     (class C) and masked out of the peeled step, so nothing is counted twice.
//   * the rows of the global tokens (dense rows) are produced by extra workgroups of the same launch, 8 rows each,
//     in the flipped orientation (fwd_rows_body below) when there are at most 16 of them; more go through the
//     32-row items of attn_fwd_band.hip and its combine launch.
//
// LDS per workgroup: 64 KiB window (re-used for the pair merge) + 4 relative-score tables (row stride 26 floats when 2m + 1 <= 25) + 3 KiB per
// group of 8 global keys = 81,920 bytes = 64 granules of 1,280 B for the BASELINE pattern: two workgroups per CU.
#include "attn_lean.h"

namespace mmt {

constexpr int kWinTiles = 8;

__device__ __forceinline__ unsigned lds_u32(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}


__device__ __forceinline__ void tile_image_to_lds(unsigned char* lds, const bf16x8 (&v)[4], int lane) {
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int ci = lane + 64 * u, row = ci >> 3, ch = ci & 7;
    *reinterpret_cast<bf16x8*>(lds + row * 128 + ((((ch >> 2) ^ ((row >> 1) & 1))) << 6) + (ch & 3) * 16) = v[u];
  }
}

// =========================================================================================================
// Rows of the global tokens (dense rows: every key of the row's segment), 8 query rows per workgroup.
// Orientation FLIPPED against the band waves: S = Q_g . K^T with the 8 query rows as the A operand, so that a lane
// owns one KEY and registers 0..3 of its accumulator hold that key's scores against queries 4h .. 4h + 3 -- every
// lane works (the band orientation, lane = query row, would idle 24 of 32 lanes and walk 16 registers for 4).
// The contraction of P . V runs over keys, i.e. over lanes here: P crosses LDS once per tile (8 x 32 bf16) and
// comes back as the B operand of O^T += V^T . P^T.  The 8 waves split the keys; their (m, l, O) partials are
// merged through LDS by the workgroup itself (wave w finishes query w): no workspace, no combine launch.
// Row maxima are per-query values shared by all lanes of a half-wave; they move only when some score exceeds
// them by kRescaleThr (then by a cross-lane maximum), and start at a large negative FINITE value so that a
// wave without a single visible key contributes exactly 0.
// =========================================================================================================
__device__ __forceinline__ float half32_max(float x) {      // maximum over the 32 lanes of this lane's half-wave
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
  return x;
}
__device__ __forceinline__ float half32_sum(float x) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}

template <int REL, bool DROP>
__device__ __forceinline__ void fwd_rows_body(const FwdParams& p, unsigned char* smem, int bn, int gq, int part) {
  using T = __bf16;
#ifdef MMT_STAMP
  long long* rdbg = (p.dbg && blockIdx.x == 0 && ((threadIdx.x >> 6) & 1) == 0) ? p.dbg + (3 * 4 + (threadIdx.x >> 7)) * 128 : nullptr;
#define RSTAMP(i) do { if (rdbg && (threadIdx.x & 63) == 0) rdbg[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define RSTAMP(i) do { } while (0)
#endif
  RSTAMP(0);
  constexpr bool HAS_REL = REL != 0;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int tstride = p.tstride;
  // LDS: per wave two V tiles (LDS-DMA double buffer: no staging registers) + P tile (512) + alpha row (32 B, padded
  // to 64) | table of the 8 rows.  The merge area of the epilogue re-uses the V tiles.
  constexpr int kWaveB = 2 * 4096 + 512 + 64;
  unsigned char* vbuf = smem + wave * kWaveB;
  __bf16* pbuf = reinterpret_cast<__bf16*>(vbuf + 8192);
  float* abuf = reinterpret_cast<float*>(vbuf + 8192 + 512);
  float* tabg = reinterpret_cast<float*>(smem + 8 * kWaveB);                 // [8][tstride]
  unsigned char* qimg = smem + 8 * kWaveB + 8 * 34 * 4;                      // the 8 query rows as a fragment image (4 KiB)
  float* comb = reinterpret_cast<float*>(smem);                              // [8 waves][16 (m, l) + 8 x 64], after a barrier
  constexpr int kCombW = 16 + 8 * 64;

  const int b = bn / p.N, n = bn - b * p.N;
  const int valid_len = p.valid_len ? p.valid_len[b] : p.S;
  const int m = p.pat.m;
  const int qg0 = p.pat.g0 + 8 * gq, n_q = min(8, p.pat.ng - 8 * gq);
  const int n_tiles = (p.S + 31) >> 5;
  // this workgroup's share of the keys (rows_parts workgroups per plane and row group: alone, one lived 62-66 k cycles --
  // under the band workgroups it shares its CU with, as long as the whole launch), then the wave's share of that
  const int n_parts = p.rows_parts, per_part = (n_tiles + n_parts - 1) / n_parts;
  const int pt0 = min(n_tiles, part * per_part), pt1 = min(n_tiles, pt0 + per_part);
  const int per_wave = (pt1 - pt0 + 7) >> 3;
  const int tw0 = min(pt1, pt0 + wave * per_wave), tw1 = min(pt1, tw0 + per_wave);

  const unsigned qs1b = (unsigned)p.qs[1] * 2, ks1b = (unsigned)p.ks[1] * 2, vs1b = (unsigned)p.vs[1] * 2;
  const T* Qb = reinterpret_cast<const T*>(p.q) + (long)b * p.qs[0] + (long)n * p.qs[2];
  const T* Kb = reinterpret_cast<const T*>(p.k) + (long)b * p.ks[0] + (long)n * p.ks[2];
  const T* Vb = reinterpret_cast<const T*>(p.v) + (long)b * p.vs[0] + (long)n * p.vs[2];
  const auto rk = make_rsrc(Kb, (unsigned)(p.S - 1) * ks1b + 128);
  // the 8 query rows as a fragment: lane r < n_q holds row qg0 + r, every other lane zeros (the range check of a
  // descriptor that ends behind the last of the rows)
  const auto rqg = make_rsrc(Qb + (long)qg0 * p.qs[1], (unsigned)(n_q - 1) * qs1b + 128);
  Frag<T> qf;
#pragma unroll
  for (int s = 0; s < 4; ++s) qf.v[s] = buf16(rqg, (unsigned)r * qs1b + 64 * h + 16 * s, 0u);
  const unsigned voff_kf = (unsigned)r * ks1b + 64 * h;                      // K rows in fragment shape (B operand)
  // V tile t -> LDS buffer (t - tw0) & 1 by LDS-DMA (4 pieces of 8 rows; the image's swizzle on the source address;
  // rows past the end: the last row, their p is 0).  Issued one tile ahead, and always BEFORE the K loads of the
  // same tile: vector-memory loads complete in order, so the wait hipcc places in front of a tile's S product (for
  // that tile's K rows) also covers the tile's V image.
  const int drow = lane >> 3, dpos = lane & 7;
  const int vch = (((dpos >> 2) ^ ((drow >> 1) & 1)) << 2) | (dpos & 3);
  const unsigned vb0 = lds_u32(vbuf);
  auto dma_v = [&](int tile) {
    const unsigned dst = vb0 + (unsigned)(((tile - tw0) & 1) * 4096);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const unsigned grow = (unsigned)min(tile * 32 + 8 * u + drow, p.S - 1);
      glds16(reinterpret_cast<const unsigned char*>(Vb) + (size_t)grow * vs1b + vch * 16,
             (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + 1024u * u)));
    }
  };
  Frag<T> kf;
  if (tw0 < tw1) {
    dma_v(tw0);
#pragma unroll
    for (int s = 0; s < 4; ++s) kf.v[s] = buf16(rk, voff_kf + 16 * s, (unsigned)(tw0 * 32) * ks1b);
  }

  // The fragment is needed once per tile; kept in registers it pushed this path over the 128-VGPR budget of the
  // launch (spilled and re-read from scratch every tile, each reload a full vmcnt(0) drain of the prefetches):
  // wave 0 parks it in LDS and every wave re-reads it per tile.
  if (wave == 0) {
#pragma unroll
    for (int s = 0; s < 4; ++s) *reinterpret_cast<bf16x8*>(qimg + lane * 64 + s * 16) = qf.v[s];
  }
  // ---- relative-score rows of the 8 queries (wave 0), log2 domain, columns permuted as everywhere ------------
  if (HAS_REL) {
    if (wave == 0) {
      const T* Eb = reinterpret_cast<const T*>(p.emb) + (long)n * 64;
      const unsigned es1b = (unsigned)p.N * 128;
      const auto re = make_rsrc(Eb, (unsigned)(p.R - 1) * es1b + 128);
      Frag<T> ef;
#pragma unroll
      for (int s = 0; s < 4; ++s) ef.v[s] = buf16(re, (unsigned)icol(m, r) * es1b + 64 * h + 16 * s, 0u);
      // bias row by table column: one load per lane, handed round through this wave's P tile (128 floats)
      float* bts = reinterpret_cast<float*>(pbuf);
      {
        const int idc = icol(m, r);
        const bool use = p.bias != nullptr && idc < p.R;
        const T* bp = reinterpret_cast<const T*>(p.bias ? p.bias : p.emb) + (use ? (long)idc * p.N + n : 0);
        const unsigned raw = *reinterpret_cast<const unsigned short*>(bp);
        if (h == 0) bts[r] = use ? __builtin_bit_cast(float, raw << 16) * p.tscale : 0.f;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      f32x16 c = {0};
      c = mma_rows(ef, qf, c);                                               // [column x query]
      float tv[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) tv[i] = fmaf(c[i], p.tscale, bts[kap(i, h)]);
      if (r < 8) {                                                           // lanes 8..31 hold no query
#pragma unroll
        for (int i = 0; i < 16; ++i) tabg[r * tstride + min(kap(i, h), 2 * m + 1)] = tv[i];   // columns past 2m: the row's spare column
      }
    }
  }
  __syncthreads();
  // this lane's four queries: 4h + i
  const SeedPair sdr = effective_seed(p.seed_lo, p.seed_hi, p.epoch);
  bool qvalid[4], qin[4];
  uint32_t dbase[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int qi = 4 * h + i;
    qin[i] = qi < n_q;
    qvalid[i] = qg0 + qi < valid_len;
    dbase[i] = drop_row_base(sdr.lo, sdr.hi, (uint32_t)bn, (uint32_t)(qg0 + qi));
  }
  const uint32_t t16 = p.drop_thresh;

  RSTAMP(1);
  f32x16 o0 = {0}, o1 = {0};
  float m_ref[4], l_loc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { m_ref[i] = -1.0e30f; l_loc[i] = 0.f; }
  const int tabg_addr = (int)lds_u32(tabg) + 4 * h * tstride * 4;

  for (int tile = tw0; tile < tw1; ++tile) {
    const int k0 = tile * 32, k = k0 + r;
    f32x16 c = {0};
    {
      Frag<T> qg;
#pragma unroll
      for (int s = 0; s < 4; ++s) qg.v[s] = *reinterpret_cast<const bf16x8*>(qimg + lane * 64 + s * 16);
      c = mma_rows(qg, kf, c);                                               // S [query x key]: registers 0..3
    }
    const unsigned char* vlds = vbuf + ((tile - tw0) & 1) * 4096;
    if (tile + 1 < tw1) {
      dma_v(tile + 1);
#pragma unroll
      for (int s = 0; s < 4; ++s) kf.v[s] = buf16(rk, voff_kf + 16 * s, (unsigned)(k0 + 32) * ks1b);
    }
    const bool kv = k < valid_len, kin = k < p.S;
    float s2[4], relv[4] = {0.f, 0.f, 0.f, 0.f};
    // (one gather per element whatever the distance: for a far key the clamp lands on column 0 or 2m, the same
    // address in every lane of the half-wave -- a broadcast read.  All four are issued before any is used: left
    // alone, hipcc sinks each under its element's validity test and waits for it there -- four LDS round trips.)
    if (HAS_REL) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        relv[i] = *(lds_cfp)(size_t)(unsigned)(tabg_addr + i * tstride * 4 + 4 * med3i(k - (qg0 + 4 * h + i) + m, 0, 2 * m));
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(relv[i]));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float rel = relv[i];
      float sc = fmaf(c[i], p.sscale, rel);
      sc = (kv == qvalid[i]) ? sc : sc + p.mask_add;
      s2[i] = (kin && qin[i]) ? sc : -INFINITY;
    }
    bool grow = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) grow |= s2[i] > m_ref[i] + kRescaleThr;
    if (__any(grow)) {
      float alpha[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float m_new = fmaxf(m_ref[i], half32_max(s2[i]));
        alpha[i] = __builtin_amdgcn_exp2f(m_ref[i] - m_new);
        m_ref[i] = m_new;
        l_loc[i] *= alpha[i];
        if (r == 0) abuf[4 * h + i] = alpha[i];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const float a = abuf[r & 7];                                           // O^T columns are queries (lane & 31)
#pragma unroll
      for (int i = 0; i < 16; ++i) { o0[i] *= a; o1[i] *= a; }
    }
    float pr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      pr[i] = __builtin_amdgcn_exp2f(s2[i] - m_ref[i]);
      l_loc[i] += pr[i];
    }
    if (DROP) {
      const uint32_t pterm = ((uint32_t)k >> 1) * kDropPairMul;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t hsh = drop_pair_finish(dbase[i], pterm);
        pr[i] = ((k & 1) ? (hsh >> 16) : (hsh & 0xFFFFu)) >= t16 ? pr[i] : 0.f;
      }
    }
    // P -> LDS as [query][key] bf16, back as the B operand (lane = query; keys in the order mma_xt's V^T reads use)
#pragma unroll
    for (int i = 0; i < 4; ++i) pbuf[(4 * h + i) * 32 + r] = (__bf16)pr[i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // (this tile's V image: see dma_v; at most the next tile's 4 + 4 stay in flight)
    {
      const int li = lane & 15, cb = (lane >> 4) & 1;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 pf;
        {
          const bf16x4 z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
          const __bf16* prow = pbuf + (r & 7) * 32 + 16 * s + 4 * h;         // keys {0..3, 8..11} + 4h + 16s
          s16x4 lo_r = *reinterpret_cast<const s16x4*>(prow), hi_r = *reinterpret_cast<const s16x4*>(prow + 8);   // every lane reads
          asm volatile("" : "+v"(lo_r), "+v"(hi_r));                         // (a valid row); lanes 8..31 then take zeros
          const bf16x4 lo = r < 8 ? __builtin_bit_cast(bf16x4, lo_r) : z;
          const bf16x4 hi = r < 8 ? __builtin_bit_cast(bf16x4, hi_r) : z;
#pragma unroll
          for (int j = 0; j < 4; ++j) { pf[j] = lo[j]; pf[4 + j] = hi[j]; }
        }
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const int row = 16 * s + 4 * h + (li >> 2);
          const int within = 32 * cb + 8 * (li & 3);
          const int off0 = row * 128 + ((db ^ ((row >> 1) & 1)) << 6) + within;
          const int row1 = row + 8;
          const int off1 = row1 * 128 + ((db ^ ((row1 >> 1) & 1)) << 6) + within;
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vlds + off0));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vlds + off1));
          bf16x8 vf;
          bf16x4 lo4 = __builtin_bit_cast(bf16x4, lo), hi4 = __builtin_bit_cast(bf16x4, hi);
#pragma unroll
          for (int j = 0; j < 4; ++j) { vf[j] = lo4[j]; vf[4 + j] = hi4[j]; }
          if (db == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o0, 0, 0, 0);
          else o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o1, 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();                                         // V / P tiles are rewritten next tile
    RSTAMP(2 + min(tile - tw0, 11));
  }
  RSTAMP(14);

  // ---- merge of the 8 waves' partials: wave w finishes query w ------------------------------------------------
  __syncthreads();                                        // the merge area re-uses the V tiles
  {
    float* mine = comb + wave * kCombW;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float ls = half32_sum(l_loc[i]);
      if (r == 0) { mine[4 * h + i] = m_ref[i]; mine[8 + 4 * h + i] = ls; }
    }
    if (r < 8) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        mine[16 + r * 64 + kap(i, h)] = o0[i];
        mine[16 + r * 64 + 32 + kap(i, h)] = o1[i];
      }
    }
  }
  __syncthreads();
  float M = -INFINITY, L = 0.f, acc = 0.f;
  const int qq = wave, d = lane;
  if (wave < n_q) {
#pragma unroll
    for (int w = 0; w < 8; ++w) M = fmaxf(M, comb[w * kCombW + qq]);
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const float wgt = __builtin_amdgcn_exp2f(comb[w * kCombW + qq] - M);
      L = fmaf(wgt, comb[w * kCombW + 8 + qq], L);
      acc = fmaf(wgt, comb[w * kCombW + 16 + qq * 64 + d], acc);
    }
  }
  if (n_parts == 1) {
    if (wave < n_q) {
      const int q = qg0 + qq;
      T* O = reinterpret_cast<T*>(p.out) + (long)b * p.os[0] + (long)q * p.os[1] + (long)n * p.os[2];
      O[d] = (T)(acc * (DROP ? p.inv_keep : 1.f) / L);
      if (p.lse && d == 0) p.lse[((long)b * p.N + n) * p.S + q] = (M + log2f(L)) * kLn2;
    }
    return;
  }
  // ---- several workgroups per row group: this one's (max, sum, O) of every row goes to the workspace, the plane's last
  //      arriver (agent-scope ticket on the caller's counter of the plane, left zero) merges the parts of all row groups
  const int n_groups = p.n_rowblk;
  float* base = p.walk_part + (size_t)bn * n_groups * n_parts * (8 * 66);
  if (wave < n_q) {
    float* mine = base + ((size_t)gq * n_parts + part) * (8 * 66) + qq * 66;
    __hip_atomic_store(mine + d, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (d == 0) {
      __hip_atomic_store(mine + 64, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(mine + 65, L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  int* flag = reinterpret_cast<int*>(smem + 8 * kWaveB);        // (the table of the rows: dead)
  if (threadIdx.x == 0) {
    const unsigned total = (unsigned)(n_groups * n_parts);
    const unsigned old = __hip_atomic_fetch_add(p.sync + bn, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = old + 1u == total;
    if (last) {
      __hip_atomic_store(p.sync + bn, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // left zero for the next call
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    flag[0] = last;
  }
  __syncthreads();
  if (!flag[0]) return;
  for (int g = 0; g < n_groups; ++g) {
    const int nq_g = min(8, p.pat.ng - 8 * g);
    if (wave >= nq_g) continue;
    const float* src = base + (size_t)g * n_parts * (8 * 66) + qq * 66;
    float ms[4], ls[4], as[4];                                   // (rows_parts <= 4)
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      const float* ps = src + (size_t)min(s2, n_parts - 1) * (8 * 66);
      ms[s2] = __hip_atomic_load(ps + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ls[s2] = __hip_atomic_load(ps + 65, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      as[s2] = __hip_atomic_load(ps + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    float Mt = -INFINITY;
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) if (s2 < n_parts) Mt = fmaxf(Mt, ms[s2]);
    float Lt = 0.f, At = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      const float wgt = s2 < n_parts ? __builtin_amdgcn_exp2f(ms[s2] - Mt) : 0.f;
      Lt = fmaf(wgt, ls[s2], Lt);
      At = fmaf(wgt, as[s2], At);
    }
    const int q = p.pat.g0 + 8 * g + qq;
    T* O = reinterpret_cast<T*>(p.out) + (long)b * p.os[0] + (long)q * p.os[1] + (long)n * p.os[2];
    O[d] = (T)(At * (DROP ? p.inv_keep : 1.f) / Lt);
    if (p.lse && d == 0) p.lse[((long)b * p.N + n) * p.S + q] = (Mt + log2f(Lt)) * kLn2;
  }
}

template <int REL, bool DROP>        // REL: 0 no relative term, 1 = 1-D ids (permuted table, Rp = 32)
__global__ __launch_bounds__(512, 4) void attn_fwd_win_bf16_kernel(const FwdParams p) {
  using T = __bf16;
  constexpr bool HAS_REL = REL != 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int ngrp = (p.pat.ng + 7) >> 3;
  const int tstride = p.tstride;
  unsigned char* kwin = smem;
  unsigned char* vwin = smem + kWinTiles * 4096;
  unsigned char* gk = smem + 2 * kWinTiles * 4096;          // 8 rows per group
  unsigned char* gv = gk + ngrp * 1024;                     // 8 rows per group + 8 finite rows behind the last one
  const int rb = wave & 3, part = wave >> 2;                // row block of the workgroup; 0 = wave A, 1 = wave B of its pair
  float* tab = reinterpret_cast<float*>(gv + (ngrp ? (ngrp + 1) * 1024 : 0)) + rb * 32 * tstride;

#ifdef MMT_STAMP
  long long* dbg = nullptr;
  {
    const int sel = blockIdx.x == 8 ? 0 : (blockIdx.x == 601 ? 1 : (blockIdx.x == 1203 ? 2 : -1));
    if (p.dbg && sel >= 0 && (wave & 1) == 0) dbg = p.dbg + (sel * 4 + (wave >> 1)) * 128;
  }
#define STAMP(i) do { if (dbg && lane == 0) dbg[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define SUB(j) do { if (p.dbg_mode & 4) STAMP(16 + min(tile - t_begin, 7) * 8 + (j)); } while (0)
#else
#define STAMP(i) do { } while (0)
#define SUB(j) do { } while (0)
#endif
  STAMP(0);
#ifdef MMT_STAMP
  if (p.dbg_mode & 8) return;
  if (p.dbg_sleep > 0 && blockIdx.x >= 256 && blockIdx.x < 512)
    for (int i = 0; i < p.dbg_sleep; ++i) __builtin_amdgcn_s_sleep(64);     // 64 x 64 cycles each
#endif
  // ---- work item: plane-major over the XCDs like the other lean kernels (attn_lean.h) ------------------
  const int nqb = (p.S + 127) >> 7;
  int bn, blk;
  // The flipped-rows workgroups (n_rowblk per plane) come FIRST in the grid, padded to a multiple of 8 blocks so that
  // the band blocks keep their XCD groups: a rows workgroup lives about as long as half the launch (its 8 waves
  // walk all keys of the plane), so it has to start at time 0 to stay off the tail.
  const int n_rows_wg = p.n_rowblk * p.rows_parts * p.B * p.N, n_rows_pad = (n_rows_wg + 7) & ~7;
  if ((int)blockIdx.x < n_rows_pad) {
#ifdef MMT_STAMP
    if (p.dbg_mode & 32) return;
#endif
    if ((int)blockIdx.x < n_rows_wg) {
      // rows workgroup i -> (plane, group of 8 rows): on the XCD group that walks the plane's band blocks
      // (plane_major_map: group x owns planes [x * BN / 8, (x + 1) * BN / 8)), so that the plane's K / V pass through ONE
      // L2 for both; any assignment is correct
      const int BN = p.B * p.N, i = (int)blockIdx.x;
      int item = i;
      if ((BN & 7) == 0 && (n_rows_wg & 7) == 0) item = (i & 7) * (n_rows_wg >> 3) + (i >> 3);
      const int per_plane = p.n_rowblk * p.rows_parts, in_plane = item % per_plane;
      fwd_rows_body<REL, DROP>(p, smem, item / per_plane, in_plane / p.rows_parts, in_plane % p.rows_parts);
    }
    return;
  }
#ifdef MMT_STAMP
  if (p.dbg_mode & 16) return;
#endif
  plane_major_map((int)blockIdx.x - n_rows_pad, p.B * p.N, 0, nqb, bn, blk);
  const int q0w = blk * 128, q0 = q0w + rb * 32;
  const int b = bn / p.N, n = bn - b * p.N;
  const int q = q0 + r;
  const int valid_len = p.valid_len ? p.valid_len[b] : p.S;
  const int W = p.pat.radius, m = p.pat.m;
  const int t0w = max(q0w - W, 0) >> 5;                                   // first tile of the window
  const int t1w = min(min(q0w + 127, p.S - 1) + W, p.S - 1) >> 5;         // last tile any wave of the block visits

  const unsigned qs1b = (unsigned)p.qs[1] * 2, ks1b = (unsigned)p.ks[1] * 2, vs1b = (unsigned)p.vs[1] * 2;
  const T* Qb = reinterpret_cast<const T*>(p.q) + (long)b * p.qs[0] + (long)n * p.qs[2];
  const unsigned char* Kb = reinterpret_cast<const unsigned char*>(reinterpret_cast<const T*>(p.k) + (long)b * p.ks[0] + (long)n * p.ks[2]);
  const unsigned char* Vb = reinterpret_cast<const unsigned char*>(reinterpret_cast<const T*>(p.v) + (long)b * p.vs[0] + (long)n * p.vs[2]);

  const auto rq = make_rsrc(Qb, (unsigned)(p.S - 1) * qs1b + 128);
  Frag<T> qf;
#pragma unroll
  for (int s = 0; s < 4; ++s) qf.v[s] = buf16(rq, (unsigned)r * qs1b + 64 * h + 16 * s, (unsigned)q0 * qs1b);
  // the pair's wave A also fetches the E rows (row r <- relative id of table column r) and the bias row it builds
  // the table from.  All of these are issued BEFORE the LDS-DMA below: they return first and their latency runs
  // under the DMA issue.
  Frag<T> ef;
  unsigned braw = 0;                                         // bias of this lane's table column, raw bf16 bits
  bool buse = false;
  if (HAS_REL && part == 0) {
    const T* Eb = reinterpret_cast<const T*>(p.emb) + (long)n * 64;
    const unsigned es1b = (unsigned)p.N * 128;
    const auto re = make_rsrc(Eb, (unsigned)(p.R - 1) * es1b + 128);
#pragma unroll
    for (int s = 0; s < 4; ++s) ef.v[s] = buf16(re, (unsigned)icol(m, r) * es1b + 64 * h + 16 * s, 0u);
    const int idc = icol(m, r);
    const bool use = p.bias != nullptr && idc < p.R;
    const T* bp = reinterpret_cast<const T*>(p.bias ? p.bias : p.emb) + (use ? (long)idc * p.N + n : 0);
    braw = *reinterpret_cast<const unsigned short*>(bp);     // always a valid address; made opaque until the table build
    buse = use;                                              // below, or hipcc converts (and waits for) it right here
  }

  // ---- staging of the window and of the global keys' rows.  Piece = 8 rows x 128 B = one wave-instruction; lane l
  //      holds LDS bytes [16 l, 16 l + 16) of the piece: row l >> 3, and the chunk that the tile image keeps at that
  //      position (the image's 64-byte swizzle applied on the SOURCE address).  Register-staged (buffer_load_dwordx4
  //      -> ds_write_b128; rows past the end read as zeros): measured against LDS-DMA, whose issue rate (one 1-KiB
  //      piece per ~48 cycles per CU) made the 134 pieces of two resident workgroups the longest item of a
  //      workgroup's life.  Wave w stages row group w >> 1 of every window tile of K (w even) or V (w odd), and
  //      one piece of the global keys' rows per pass.
  const int drow = lane >> 3, dpos = lane & 7;
  const int ch = (((dpos >> 2) ^ ((drow >> 1) & 1)) << 2) | (dpos & 3);
  const bool stg_v = (wave & 1) != 0;
  const unsigned st1b = stg_v ? vs1b : ks1b;
  const auto rkv = make_rsrc(stg_v ? Vb : Kb, (unsigned)(p.S - 1) * st1b + 128);
  const unsigned voff_st = (unsigned)drow * st1b + ch * 16;
  int n_win = t1w - t0w + 1;
  int n_gp = ngrp ? 2 * ngrp + 1 : 0;                                 // global pieces: K groups, then V groups + pad
#ifdef MMT_STAMP
  if (p.dbg_mode & 2) n_win = n_gp = 0;
#endif
  bf16x8 stg[kWinTiles], stg_g;
#pragma unroll
  for (int j = 0; j < kWinTiles; ++j)
    if (j < n_win) stg[j] = buf16(rkv, voff_st, (unsigned)((t0w + j) * 32 + (wave >> 1) * 8) * st1b);
  auto gpiece = [&](int pg) {                                           // rows of global piece pg for this lane
    const bool gv_ = pg >= ngrp;
    const unsigned grow = (unsigned)min(p.pat.g0 + (gv_ ? pg - ngrp : pg) * 8 + drow, p.S - 1);
    return *reinterpret_cast<const bf16x8*>((gv_ ? Vb + (size_t)grow * vs1b : Kb + (size_t)grow * ks1b) + ch * 16);
  };
  if (wave < n_gp) stg_g = gpiece(wave);

  STAMP(1);

  // ---- relative-score table (log2 domain) of this wave's 32 rows ------------------------------------------
  float relfn = 0.f, relfp = 0.f;
  if (HAS_REL && part == 0) {
    // bias row by table column: one load per lane, handed round through the (not yet written) table area
    float bcol[16];
    {
      asm volatile("" : "+v"(braw));
      if (h == 0) tab[r] = buse ? __builtin_bit_cast(float, braw << 16) * p.tscale : 0.f;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 16; ++i) bcol[i] = tab[kap(i, h)];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    f32x16 c = {0};
    c = mma_rows(ef, qf, c);   // [column x q]
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      // columns past 2m carry no id: they all land in the row's spare column 2m + 1 (< tstride) -- no branches
      tab[r * tstride + min(kap(i, h), 2 * m + 1)] = fmaf(c[i], p.tscale, bcol[i]);
    }
  }
  STAMP(2);
#pragma unroll
  for (int j = 0; j < kWinTiles; ++j)
    if (j < n_win) *reinterpret_cast<bf16x8*>(smem + (stg_v ? kWinTiles * 4096 : 0) + j * 4096 + (wave >> 1) * 1024 + lane * 16) = stg[j];
  if (wave < n_gp) *reinterpret_cast<bf16x8*>(gk + wave * 1024 + lane * 16) = stg_g;
  for (int pg = wave + 8; pg < n_gp; pg += 8)                          // more than 8 global pieces: many global keys
    *reinterpret_cast<bf16x8*>(gk + pg * 1024 + lane * 16) = gpiece(pg);
  STAMP(3);
  __syncthreads();                                       // ... and everybody else's; the pair's table is complete
  STAMP(4);
  const bool live = q0 < p.S;                            // (waves past the end still meet the barriers below)
  if (HAS_REL && live) {
    relfn = tab[r * tstride];
    relfp = tab[r * tstride + 2 * m];
  }

  f32x16 o0 = {0}, o1 = {0};
  float m_run = -INFINITY, l_run = 0.f;
  const float* trow = tab + r * tstride;
  const int trow_addr = (int)lds_u32(trow);
  const bool q_ok = q < p.S;
  const bool qblk_valid = q0 + 31 < valid_len, qblk_pad = q0 >= valid_len, qblk_in = q0 + 31 < p.S;
  const SeedPair sd = effective_seed(p.seed_lo, p.seed_hi, p.epoch);
  const uint32_t drop_base = drop_row_base(sd.lo, sd.hi, (uint32_t)bn, (uint32_t)q);
  const uint32_t t16 = p.drop_thresh;

  // the block's band tiles b0 .. b1; wave A takes the first two (after the peeled global keys), wave B the rest
  const int b0 = max(q0 - W, 0) >> 5;
  int b1 = min(q0 + 31 + W, p.S - 1) >> 5;
#ifdef MMT_STAMP
  if (p.dbg_mode & 1) b1 = b0 - 1;
#endif
  const int nA = min(2, b1 - b0 + 1);
  const int t_begin = !live ? 1 : (part == 0 ? b0 : b0 + nA);
  const int t_end = !live ? 0 : (part == 0 ? b0 + nA - 1 : b1);           // inclusive

  // ---- peeled steps: the global keys outside this wave's band tiles, 8 per step, registers 0..3 only ----------
  bool fresh = true;
  for (int g = 0; g < (live && part == 0 ? ngrp : 0); ++g) {
    const int kg0 = p.pat.g0 + 8 * g;                      // keys kg0 .. kg0 + 7 (past the range: masked)
    const int n_here = min(8, p.pat.ng - 8 * g);
    if (kg0 >= b0 * 32 && kg0 + n_here - 1 <= b1 * 32 + 31) continue;     // all of them are band-tile keys
    Frag<T> kf;
    {
      const int rr = r & 7;                                // rows 8..31 of the "tile" do not exist: any finite row
      const unsigned char* row = gk + g * 1024 + rr * 128 + ((h ^ ((rr >> 1) & 1)) << 6);
#pragma unroll
      for (int s = 0; s < 4; ++s) kf.v[s] = *reinterpret_cast<const bf16x8*>(row + s * 16);
    }
    f32x16 c = {0};
    c = mma_rows(kf, qf, c);                               // registers 0..3: key kg0 + i + 4h
    float s2[4];
    const bool qv = q < valid_len;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kk = kg0 + i + 4 * h;
      const bool present = (i + 4 * h < n_here) && !(kk >= b0 * 32 && kk <= b1 * 32 + 31);
      float rel = 0.f;
      if (HAS_REL) rel = trow[min(max(kk - q, -m), m) + m];
      float s = fmaf(c[i], p.sscale, rel);
      s = ((kk < valid_len) == qv) ? s : s + p.mask_add;
      s2[i] = present ? s : -INFINITY;
    }
    float tmax = half_max(fmaxf(fmaxf(s2[0], s2[1]), fmaxf(s2[2], s2[3])));
    if (fresh) {                                           // nothing accumulated yet: no rescale of O / l
      m_run = tmax;
      fresh = false;
    } else if (__any(tmax > m_run + kRescaleThr)) {
      const float m_new = fmaxf(m_run, tmax);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
    }
    float pr[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      pr[i] = __builtin_amdgcn_exp2f(s2[i] - m_run);
      l_run += pr[i];
      if (DROP) pr[i] = drop_bits16(drop_base, (uint32_t)(kg0 + i + 4 * h)) >= t16 ? pr[i] : 0.f;
    }
#pragma unroll
    for (int i = 4; i < 16; ++i) pr[i] = 0.f;
    // O^T += V^T . P^T over the first 16 rows of the group's "tile" (rows 8..15: the next group's, or the pad rows; p = 0)
    {
      const unsigned char* xlds = gv + g * 1024;
      const int li = lane & 15, cb = (lane >> 4) & 1;
      bf16x8 pf;
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[j] = (__bf16)pr[j];
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        const int row = 4 * h + (li >> 2);
        const int within = 32 * cb + 8 * (li & 3);
        const int off0 = row * 128 + ((db ^ ((row >> 1) & 1)) << 6) + within;
        const int row1 = row + 8;
        const int off1 = row1 * 128 + ((db ^ ((row1 >> 1) & 1)) << 6) + within;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(xlds + off0));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(xlds + off1));
        bf16x8 vf;
        bf16x4 lo4 = __builtin_bit_cast(bf16x4, lo), hi4 = __builtin_bit_cast(bf16x4, hi);
#pragma unroll
        for (int j = 0; j < 4; ++j) { vf[j] = lo4[j]; vf[4 + j] = hi4[j]; }
        if (db == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o0, 0, 0, 0);
        else o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o1, 0, 0, 0);
      }
    }
  }

  STAMP(5);
  // ---- band tiles, straight out of the window ------------------------------------------------------------------
  for (int tile = t_begin; tile <= t_end; ++tile) {
    const int k0 = tile * 32;
    const unsigned char* klds = kwin + (tile - t0w) * 4096;
    const unsigned char* vlds = vwin + (tile - t0w) * 4096;
    Frag<T> kf;
    frag_from_tile(kf, klds, lane);
    f32x16 c = {0};
    c = mma_rows(kf, qf, c);     // S^T [key x q]
    SUB(0);

    const int dmin = k0 - (q0 + 31), dmax = k0 + 31 - q0;
    const bool in_range = (k0 + 31 < p.S) && qblk_in;
    const bool seg_all = (qblk_valid && k0 + 31 < valid_len) || (qblk_pad && k0 >= valid_len);
    const bool band_all = dmin >= -W && dmax <= W;
    const bool plain = in_range && seg_all && band_all;
    const bool far_neg = dmax <= -m, far_pos = dmin >= m;
    const bool one_id = !HAS_REL || far_neg || far_pos;
    const bool no_gkey = p.pat.ng == 0 || k0 + 31 < p.pat.g0 || k0 >= p.pat.g0 + p.pat.ng;
    const float relc = HAS_REL ? (far_neg ? relfn : relfp) : 0.f;
    const int dbase = k0 - q + 4 * h;

    float pr[16], s2[16];
    if (plain && one_id) {                                     // ---- class A
#pragma unroll
      for (int i = 0; i < 16; ++i) s2[i] = fmaf(c[i], p.sscale, relc);
    } else if (plain) {                                        // ---- class B (mixed ids): see attn_fwd_band.hip
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
      const float relm = relc + p.mask_add;
      if (dmin >= -W) {
        const int bound = W - dbase;
#pragma unroll
        for (int i = 0; i < 16; ++i) s2[i] = fmaf(c[i], p.sscale, ((i & 3) + 8 * (i >> 2)) > bound ? relm : relc);
      } else if (dmax <= W) {
        const int bound = -W - dbase;
#pragma unroll
        for (int i = 0; i < 16; ++i) s2[i] = fmaf(c[i], p.sscale, ((i & 3) + 8 * (i >> 2)) < bound ? relm : relc);
      } else {
        const unsigned W2 = 2u * (unsigned)W;
#pragma unroll
        for (int i = 0; i < 16; ++i)
          s2[i] = fmaf(c[i], p.sscale, (unsigned)(dbase + (i & 3) + 8 * (i >> 2) + W) <= W2 ? relc : relm);
      }
    } else {                                                   // ---- class C (general)
      const int kb = k0 + 4 * h;
      const bool qv = q < valid_len;
      const unsigned W2 = 2u * (unsigned)W;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ci = (i & 3) + 8 * (i >> 2);
        const int kk = kb + ci, d = dbase + ci;
        const bool near = (unsigned)(d + W) <= W2;
        const bool gkey = (unsigned)(kk - p.pat.g0) < (unsigned)p.pat.ng;
        const bool seg = (kk < valid_len) == qv;
        const bool keep = (int)seg & ((int)near | (int)gkey);
        float rel = 0.f;
        if (HAS_REL) rel = trow[min(max(d, -m), m) + m];
        float s = fmaf(c[i], p.sscale, rel);
        s = keep ? s : s + p.mask_add;
        s2[i] = kk < p.S ? s : -INFINITY;
      }
    }
    SUB(1);
    float tmax = fmaxf(fmaxf(s2[0], s2[1]), s2[2]);
#pragma unroll
    for (int i = 3; i < 15; i += 2) tmax = fmaxf(fmaxf(tmax, s2[i]), s2[i + 1]);
    tmax = fmaxf(tmax, s2[15]);
    tmax = half_max(tmax);
    SUB(2);
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
    SUB(3);

    if (DROP) {                            // 16 bits per element, one hash per key pair; 1 / keep in the epilogue
      const uint32_t kc = ((uint32_t)(k0 >> 1) + 2u * (uint32_t)h) * kDropPairMul;   // pair index of kap(0, h)
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const uint32_t hsh = drop_pair_finish(drop_base, kc + (uint32_t)(4 * (i >> 2) + ((i & 3) >> 1)) * kDropPairMul);
        pr[i] = (hsh & 0xFFFFu) >= t16 ? pr[i] : 0.f;
        pr[i + 1] = (hsh >> 16) >= t16 ? pr[i + 1] : 0.f;
      }
    }
    SUB(4);
    mma_xt(o0, o1, VTile<T>{}, vlds, pr, lane);   // O^T[d x q] += V^T[d x key] . P^T[key x q]
    STAMP(6 + min(tile - t_begin, 7));
  }

  // ---- pair merge + epilogue --------------------------------------------------------------------
  // Each wave keeps the 32-column half of O^T it will store (A: d < 32, B: d >= 32) and hands the other half, with
  // its row maximum and row sum, to its partner through the window area (dead after the first barrier below).
  const float l_own = half_sum(l_run);
  __syncthreads();
  {
    float* mine = reinterpret_cast<float*>(smem + (rb * 2 + part) * 4352);
    if (h == 0) { mine[r] = m_run; mine[32 + r] = l_own; }
#pragma unroll
    for (int i = 0; i < 16; ++i) mine[64 + i * 64 + lane] = part == 0 ? o1[i] : o0[i];
  }
  __syncthreads();
  STAMP(14);
  if (!q_ok) return;
  if (p.skip_global_rows && is_global(p.pat, q)) return;
  const float* theirs = reinterpret_cast<const float*>(smem + (rb * 2 + (part ^ 1)) * 4352);
  const float m_o = theirs[r], l_o = theirs[32 + r];
  const float m_tot = fmaxf(m_run, m_o);                   // wave A always holds a finite maximum
  const float a_own = __builtin_amdgcn_exp2f(m_run - m_tot), a_oth = __builtin_amdgcn_exp2f(m_o - m_tot);
  const float l_tot = l_own * a_own + l_o * a_oth;
  const float inv = (DROP ? p.inv_keep : 1.f) / l_tot;
  const float f_own = a_own * inv, f_oth = a_oth * inv;
  T* O = reinterpret_cast<T*>(p.out) + (long)b * p.os[0] + (long)q * p.os[1] + (long)n * p.os[2] + 32 * part;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    bf16x4 x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float own = part == 0 ? o0[4 * g + j] : o1[4 * g + j];
      x[j] = (__bf16)(own * f_own + theirs[64 + (4 * g + j) * 64 + lane] * f_oth);
    }
    *reinterpret_cast<bf16x4*>(O + 8 * g + 4 * h) = x;
  }
  if (p.lse && h == 0 && part == 0) p.lse[((long)b * p.N + n) * p.S + q] = (m_tot + log2f(l_tot)) * kLn2;
}

// LDS bytes of one workgroup (host side; `tstride` as chosen by the caller)
int fwd_win_lds_bytes(int ng, int tstride) {
  const int ngrp = (ng + 7) / 8;
  const int band = 2 * kWinTiles * 4096 + (ngrp ? (2 * ngrp + 1) * 1024 : 0) + 4 * 32 * tstride * 4;
  const int rows = 8 * (2 * 4096 + 512 + 64) + 8 * 34 * 4 + 4096;          // fwd_rows_body's carve (used when 0 < ng <= 16)
  return (ng > 0 && ng <= 16 && rows > band) ? rows : band;
}

hipError_t launch_attn_fwd_win_bf16(const FwdParams& p, hipStream_t st) {
  dim3 grid(p.n_band_blocks + ((p.n_rowblk * p.rows_parts * p.B * p.N + 7) & ~7));
  const int lds = fwd_win_lds_bytes(p.pat.ng, p.tstride);
  const bool rel = p.R > 0 && p.pat.id_mode == 1;
  const bool drop = p.drop_thresh != 0;
  auto go = [&](auto kern) {
    // the attribute is per kernel and per device: set on every call (host-side, no launch)
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, p);
  };
  if (rel) { if (drop) go(attn_fwd_win_bf16_kernel<1, true>); else go(attn_fwd_win_bf16_kernel<1, false>); }
  else     { if (drop) go(attn_fwd_win_bf16_kernel<0, true>); else go(attn_fwd_win_bf16_kernel<0, false>); }
  return hipGetLastError();
}

}  // namespace mmt
