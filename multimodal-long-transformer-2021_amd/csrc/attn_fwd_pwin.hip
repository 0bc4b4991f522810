// Forward kernel with a SLIDING workgroup-shared K / V window: the window kernel (attn_fwd_win.hip) made persistent.
// Same operator, outputs, tile classes and pair split (two waves per 32-row block, merged through LDS); what changes:
//
//   * a workgroup walks `walk` CONSECUTIVE 128-row blocks (linear order over (plane, block); <= 512 workgroups = two per
//     CU, all resident: one round instead of three).  Inside a plane the 8-tile window SLIDES: block i + 1 keeps tiles
//     4i + 2 .. 4i + 5 of block i and fetches only the four new ones into the ring slots (tile & 7) of the four that
//     died -- half the staging bytes, and the fetch is issued BEFORE the pair merge of block i, so that its latency runs
//     under the merge and the store instead of in front of the tile loop;
//   * the next block's Q rows are fetched into the registers of the current block's (dead after its last S product);
//     kernel arguments, plane pointers, descriptors and the global keys' rows are set up once per plane, not per block;
//   * the rows of the (<= 8) global tokens have no workgroups of their own (those lived as long as the whole launch
//     should take): while the pair's wave A builds the next block's relative-score table, its wave B -- idle until the
//     tiles start -- contracts the 8 global rows against ONE of the four KEPT window tiles in the flipped orientation
//     (lane = key; P crosses LDS through 256 B of scratch per wave).  Each wave B keeps a running
//     (m, l, O) per walk in the workspace; when the plane's last block is done the workgroup that arrives last at the
//     plane's counter (mmt_attn_desc.sync) merges the plane's partials and writes the 8 rows: no combine launch, no
//     spinning (cdna_hip_programming.md, Guideline 16: write-through partials, every storing wave drained, one
//     agent-scope add; the last arriver acquires).
//
// LDS per workgroup as in attn_fwd_win.hip: 64 KiB ring + 4 tables + 3 KiB of global-key rows = 81,920 B (two per CU).
#include "attn_lean.h"

namespace mmt {

namespace {

constexpr int kPwState = 512 + 256 + 16;      // floats of one rows stream: O^T of 8 rows (16 lanes x 32) | per-lane row sums | 8 maxima (+ pad)

__device__ __forceinline__ unsigned pw_lds_u32(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ float pw_h32_max(float x) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
  return x;
}
__device__ __forceinline__ float pw_h32_sum(float x) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}
__device__ __forceinline__ void pw_wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

}  // namespace

template <int REL, bool DROP>        // REL: 0 no relative term, 1 = 1-D ids (permuted table, Rp = 32)
__global__ __launch_bounds__(512, 4) void attn_fwd_pwin_bf16_kernel(const FwdParams p) {
  using T = __bf16;
  constexpr bool HAS_REL = REL != 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ng = p.pat.ng;
  const int tstride = p.tstride;
  unsigned char* kring = smem;
  unsigned char* vring = smem + 8 * 4096;
  unsigned char* gv = smem + 16 * 4096;                     // 8 V rows of the global keys | their 8 K rows (which double as the
  unsigned char* gk = gv + 1024;                            //   8 finite rows behind the V rows the peeled step reads) | 4 x 256 B
  unsigned char* scr = gv + 2048 + (wave & 3) * 256;        //   of scratch for the rows steps (P crosses LDS there)
  const int rb = wave & 3, part = wave >> 2;                // row block of the workgroup; 0 = wave A, 1 = wave B of its pair
  float* tab = reinterpret_cast<float*>(smem + 16 * 4096 + 3072) + rb * 32 * tstride;

  const int nqb = (p.S + 127) >> 7, NT = (p.S + 31) >> 5;
  const int total = p.B * p.N * nqb;
  const int L = xcd_remap((int)blockIdx.x, (int)gridDim.x);  // walk index: consecutive walks share an XCD (and a plane)
  const int k_begin = L * p.pw_walk, k_end = min(total, k_begin + p.pw_walk);
  if (k_begin >= k_end) return;
  const int W = p.pat.radius, m = p.pat.m;
  const SeedPair sd = effective_seed(p.seed_lo, p.seed_hi, p.epoch);
  const uint32_t t16 = p.drop_thresh;
  // workspace of the global rows (ng > 0): per walk: this workgroup's copy of the 8 rows' relative-score table and the
  // running state of its four tile streams; per plane: the compact partials the last arriver merges
  float* ws_tabg = p.walk_part + (size_t)L * (8 * 34 + 4 * kPwState);
  float* ws_state = ws_tabg + 8 * 34 + (size_t)rb * kPwState;
  float* ws_part = p.walk_part + (size_t)gridDim.x * (8 * 34 + 4 * kPwState);

#ifdef MMT_STAMP
  long long* dbg = nullptr;
  {
    const int sel = blockIdx.x == 8 ? 0 : (blockIdx.x == 301 ? 1 : -1);
    if (p.dbg && sel >= 0) dbg = p.dbg + (sel * 8 + wave) * 64;
  }
#define PSTAMP(i) do { if (dbg && (threadIdx.x & 63) == 0 && (i) < 64) dbg[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif
  Frag<T> qf;
  int cur_bn = -1, plane_blocks = 0;                        // plane of the previous block, blocks of it this walk has done
  bool rows_fresh = true;                                   // this walk has no running rows state for the plane yet

  for (int kb = k_begin; kb < k_end; ++kb) {
    // Everything lane-dependent is re-derived per block from an opaque copy of the lane id (hipcc hoists the dozens of
    // lane-dependent LDS addresses of the block body out of this loop otherwise -- and spills them).
    int lane = (int)(threadIdx.x & 63);
    asm volatile("" : "+v"(lane));
    const int r = lane & 31, h = lane >> 5;
    const int drow = lane >> 3, dpos = lane & 7;
    const int dch = (((dpos >> 2) ^ ((drow >> 1) & 1)) << 2) | (dpos & 3);
    const int sbase = (kb - k_begin) * 10;
    (void)sbase;
    PSTAMP(sbase + 0);

    const int bn = kb / nqb, blk = kb - bn * nqb;
    const bool new_plane = bn != cur_bn;
    const int b = bn / p.N, n = bn - b * p.N;
    const int valid_len = p.valid_len ? p.valid_len[b] : p.S;
    const int q0w = blk * 128, q0 = q0w + rb * 32, q = q0 + r;
    const int t0w = max(q0w - W, 0) >> 5;
    const int t1w = min(min(q0w + 127, p.S - 1) + W, p.S - 1) >> 5;
    const unsigned qs1b = (unsigned)p.qs[1] * 2, ks1b = (unsigned)p.ks[1] * 2, vs1b = (unsigned)p.vs[1] * 2;
    const T* Qb = reinterpret_cast<const T*>(p.q) + (long)b * p.qs[0] + (long)n * p.qs[2];
    const unsigned char* Kb = reinterpret_cast<const unsigned char*>(reinterpret_cast<const T*>(p.k) + (long)b * p.ks[0] + (long)n * p.ks[2]);
    const unsigned char* Vb = reinterpret_cast<const unsigned char*>(reinterpret_cast<const T*>(p.v) + (long)b * p.vs[0] + (long)n * p.vs[2]);
    const bool stg_v = (wave & 1) != 0;
    const unsigned st1b = stg_v ? vs1b : ks1b;
    // tiles this block has to fetch: the whole window at the start of a walk / plane, the part above the previous
    // block's window otherwise (4 tiles for a radius of 64)
    const int t_new0 = new_plane ? t0w : max(t0w, (min(min(q0w - 1, p.S - 1) + W, p.S - 1) >> 5) + 1);
    const int n_new = t1w - t_new0 + 1;
    const bool live = q0 < p.S;

    // ---- walk start / plane change: nothing was prefetched for this block --------------------------------------
    if (new_plane) {
      if (kb > k_begin) __syncthreads();                    // (the previous plane's last reads of gk / gv)
      const auto rq = make_rsrc(Qb, (unsigned)(p.S - 1) * qs1b + 128);
#pragma unroll
      for (int s = 0; s < 4; ++s) qf.v[s] = buf16(rq, (unsigned)r * qs1b + 64 * h + 16 * s, (unsigned)q0 * qs1b);
      if (ng > 0 && wave < 2) {                             // rows of the global keys: V (wave 0), K (wave 1)
        const unsigned grow = (unsigned)min(p.pat.g0 + drow, p.S - 1);
        glds16((wave == 1 ? Kb + (size_t)grow * ks1b : Vb + (size_t)grow * vs1b) + dch * 16,
               (unsigned)__builtin_amdgcn_readfirstlane((int)(pw_lds_u32(gv) + wave * 1024)));
      }
      if (ng > 0 && HAS_REL && wave == 7) {
        // this workgroup's copy of the 8 global rows' relative-score rows (read by the rows steps through L2)
        const T* Eb = reinterpret_cast<const T*>(p.emb) + (long)n * 64;
        const unsigned es1b = (unsigned)p.N * 128;
        const auto re = make_rsrc(Eb, (unsigned)(p.R - 1) * es1b + 128);
        const auto rqg = make_rsrc(Qb + (long)p.pat.g0 * p.qs[1], (unsigned)(min(8, ng) - 1) * qs1b + 128);
        Frag<T> ef, qg;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          ef.v[s] = buf16(re, (unsigned)icol(m, r) * es1b + 64 * h + 16 * s, 0u);
          qg.v[s] = buf16(rqg, (unsigned)r * qs1b + 64 * h + 16 * s, 0u);
        }
        f32x16 c = {0};
        c = mma_rows(ef, qg, c);                             // [column x query]
        if (r < 8) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int col = kap(i, h), idc = icol(m, col);
            const bool use = p.bias != nullptr && idc < p.R;
            const T* bp = reinterpret_cast<const T*>(p.bias ? p.bias : p.emb) + (use ? (long)idc * p.N + n : 0);
            const float bv = use ? (float)*bp * p.tscale : 0.f;
            if (col <= 2 * m) ws_tabg[r * 34 + col] = fmaf(c[i], p.tscale, bv);
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (visible to the workgroup's other waves after the next barrier)
      }
      rows_fresh = true;
      plane_blocks = 0;
      cur_bn = bn;
    }

    // ---- the block's new tiles by LDS-DMA (1-KiB pieces: 8 rows x 128 B, the tile image's swizzle on the source address;
    //      rows past the end repeat the last row -- their keys are masked) into the ring slots the previous block's merge
    //      used until barrier (4).  No staging registers: held across the merge they were spilled, load by load.  The
    //      transfer runs under the table build / rows step below; everybody waits for its own pieces before barrier (1). ---
    {
      const unsigned char* src = stg_v ? Vb : Kb;
      const unsigned ring0 = pw_lds_u32(stg_v ? vring : kring) + (unsigned)(wave >> 1) * 1024u;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j < n_new) {
          const unsigned grow = (unsigned)min((t_new0 + j) * 32 + (wave >> 1) * 8 + drow, p.S - 1);
          glds16(src + (size_t)grow * st1b + dch * 16, (unsigned)__builtin_amdgcn_readfirstlane((int)(ring0 + (unsigned)((t_new0 + j) & 7) * 4096u)));
        }
      }
    }

    // ---- (A) the pair's relative-score table  ||  (B) the global rows against one KEPT tile -------------------------
    float relfn = 0.f, relfp = 0.f;
    const float* trow = tab + r * tstride;
    const int trow_addr = (int)pw_lds_u32(trow);
    if (HAS_REL && part == 0 && live) {
      const T* Eb = reinterpret_cast<const T*>(p.emb) + (long)n * 64;
      const unsigned es1b = (unsigned)p.N * 128;
      const auto re = make_rsrc(Eb, (unsigned)(p.R - 1) * es1b + 128);
      Frag<T> ef;
#pragma unroll
      for (int s = 0; s < 4; ++s) ef.v[s] = buf16(re, (unsigned)icol(m, r) * es1b + 64 * h + 16 * s, 0u);
      const int idc = icol(m, r);
      const bool use = p.bias != nullptr && idc < p.R;
      const T* bp = reinterpret_cast<const T*>(p.bias ? p.bias : p.emb) + (use ? (long)idc * p.N + n : 0);
      unsigned braw = *reinterpret_cast<const unsigned short*>(bp);
      float bcol[16];
      {
        asm volatile("" : "+v"(braw));
        if (h == 0) tab[r] = use ? __builtin_bit_cast(float, braw << 16) * p.tscale : 0.f;
        pw_wave_fence();
#pragma unroll
        for (int i = 0; i < 16; ++i) bcol[i] = tab[kap(i, h)];
        pw_wave_fence();
      }
      f32x16 c = {0};
      c = mma_rows(ef, qf, c);   // [column x q]
#pragma unroll
      for (int i = 0; i < 16; ++i) tab[r * tstride + min(kap(i, h), 2 * m + 1)] = fmaf(c[i], p.tscale, bcol[i]);
    }
    // rows step of wave B (and, for tiles that were not kept, of both phases below): tile `tr` of the ring against the 8
    // global rows, flipped orientation; running state of stream `rb` in the workspace
    auto rows_step = [&](int tr) {
      const int n_q = min(8, ng), qg0 = p.pat.g0;
      const unsigned char* klds = kring + (tr & 7) * 4096;
      const unsigned char* vlds = vring + (tr & 7) * 4096;
      const int k = tr * 32 + r;
      const auto rqg = make_rsrc(Qb + (long)qg0 * p.qs[1], (unsigned)(n_q - 1) * qs1b + 128);
      Frag<T> kf, qg;
#pragma unroll
      for (int s = 0; s < 4; ++s) qg.v[s] = buf16(rqg, (unsigned)r * qs1b + 64 * h + 16 * s, 0u);
      frag_from_tile(kf, klds, lane);
      float m_ref[4], l_loc[4], alpha4[4] = {1.f, 1.f, 1.f, 1.f};
      if (rows_fresh) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { m_ref[i] = -1.0e30f; l_loc[i] = 0.f; }
      } else {
        const f32x4 lv = *reinterpret_cast<const f32x4*>(ws_state + 512 + lane * 4);
        const f32x4 mv = *reinterpret_cast<const f32x4*>(ws_state + 768 + 4 * h);
#pragma unroll
        for (int i = 0; i < 4; ++i) { l_loc[i] = lv[i]; m_ref[i] = mv[i]; }
      }
      float relv[4] = {0.f, 0.f, 0.f, 0.f};
      if (HAS_REL) {
#pragma unroll
        for (int i = 0; i < 4; ++i)      // (written by another wave of this workgroup, possibly over an earlier plane's copy: past L1)
          relv[i] = __hip_atomic_load(ws_tabg + (4 * h + i) * 34 + min(max(k - (qg0 + 4 * h + i), -m), m) + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      f32x16 c = {0};
      c = mma_rows(qg, kf, c);                                               // S [query x key]: registers 0..3
      const bool kv = k < valid_len, kin = k < p.S;
      float s2[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int qi = 4 * h + i;
        float sc = fmaf(c[i], p.sscale, relv[i]);
        sc = (kv == (qg0 + qi < valid_len)) ? sc : sc + p.mask_add;
        s2[i] = (kin && qi < n_q) ? sc : -INFINITY;
      }
      bool grow = false;
#pragma unroll
      for (int i = 0; i < 4; ++i) grow |= s2[i] > m_ref[i] + kRescaleThr;
      const bool grew = __any(grow);
      if (grew) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float m_new = fmaxf(m_ref[i], pw_h32_max(s2[i]));
          alpha4[i] = __builtin_amdgcn_exp2f(m_ref[i] - m_new);
          m_ref[i] = m_new;
          l_loc[i] *= alpha4[i];
        }
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
          const uint32_t hsh = drop_pair_finish(drop_row_base(sd.lo, sd.hi, (uint32_t)bn, (uint32_t)(qg0 + 4 * h + i)), pterm);
          pr[i] = ((k & 1) ? (hsh >> 16) : (hsh & 0xFFFFu)) >= t16 ? pr[i] : 0.f;
        }
      }
      // P -> LDS as [query][key] bf16, 16 keys at a time (256 B of scratch), back as the B operand of O^T += V^T . P^T
      __bf16* pbuf = reinterpret_cast<__bf16*>(scr);
      f32x16 g0a = {0}, g1a = {0};
      {
        const int li = lane & 15, cb = (lane >> 4) & 1;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          if ((r >> 4) == s) {
#pragma unroll
            for (int i = 0; i < 4; ++i) pbuf[(4 * h + i) * 16 + (r & 15)] = (__bf16)pr[i];
          }
          pw_wave_fence();
          bf16x8 pf;
          {
            const bf16x4 z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
            const __bf16* prow = pbuf + (r & 7) * 16 + 4 * h;                // keys {0..3, 8..11} + 4h + 16s
            s16x4 lo_r = *reinterpret_cast<const s16x4*>(prow), hi_r = *reinterpret_cast<const s16x4*>(prow + 8);
            asm volatile("" : "+v"(lo_r), "+v"(hi_r));
            const bf16x4 lo = r < 8 ? __builtin_bit_cast(bf16x4, lo_r) : z;
            const bf16x4 hi = r < 8 ? __builtin_bit_cast(bf16x4, hi_r) : z;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { pf[jj] = lo[jj]; pf[4 + jj] = hi[jj]; }
          }
          pw_wave_fence();
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
            for (int jj = 0; jj < 4; ++jj) { vf[jj] = lo4[jj]; vf[4 + jj] = hi4[jj]; }
            if (db == 0) g0a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, g0a, 0, 0, 0);
            else g1a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, g1a, 0, 0, 0);
          }
        }
      }
      // state = alpha . state + this tile's P . V  (lanes r < 8 hold the 8 rows' O^T columns)
      {
        f32x4 lv, mv;
#pragma unroll
        for (int i = 0; i < 4; ++i) { lv[i] = l_loc[i]; mv[i] = m_ref[i]; }
        *reinterpret_cast<f32x4*>(ws_state + 512 + lane * 4) = lv;
        if (r == 0) *reinterpret_cast<f32x4*>(ws_state + 768 + 4 * h) = mv;
        float a = 1.f;
        if (grew && !rows_fresh) {                                         // (wave-uniform)
          float* abuf = reinterpret_cast<float*>(scr);                     // (the P tile has been consumed)
          pw_wave_fence();
          if (r == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) abuf[4 * h + i] = alpha4[i];
          }
          pw_wave_fence();
          a = abuf[r & 7];                                                 // O^T columns are queries (lane & 31)
        }
        if (r < 8) {
          float* dstp = ws_state + (r + 8 * h) * 32;
#pragma unroll
          for (int i = 0; i < 16; i += 4) {
            f32x4 x0 = {0.f, 0.f, 0.f, 0.f}, x1 = {0.f, 0.f, 0.f, 0.f};
            if (!rows_fresh) { x0 = *reinterpret_cast<const f32x4*>(dstp + i); x1 = *reinterpret_cast<const f32x4*>(dstp + 16 + i); }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { x0[jj] = fmaf(x0[jj], a, g0a[i + jj]); x1[jj] = fmaf(x1[jj], a, g1a[i + jj]); }
            *reinterpret_cast<f32x4*>(dstp + i) = x0;
            *reinterpret_cast<f32x4*>(dstp + 16 + i) = x1;
          }
        }
      }
      rows_fresh = false;
    };
    // Ownership of the key tiles for the rows steps: block `blk` takes the lower half of its window, tiles 4 blk - 2 ..
    // 4 blk + 1 (stream rb <-> tile 4 blk - 2 + rb), and the plane's last block the tiles above as well.  When the
    // previous block of the walk left them in the ring (no new plane) they are contracted HERE, beside the table build.
    const int tr_low = 4 * blk - 2 + rb;
    const bool rows_low = ng > 0 && part == 1 && tr_low >= 0 && tr_low < NT;
    if (rows_low && !new_plane) rows_step(tr_low);

    PSTAMP(sbase + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces have landed
    PSTAMP(sbase + 2);
    __syncthreads();          // (1) the window is complete, the pair's table is written
    PSTAMP(sbase + 3);
    if (HAS_REL && live) { relfn = trow[0]; relfp = trow[2 * m]; }

    // rows steps on tiles that were not in the ring before: the low tiles of a walk's / plane's first block, and the
    // two tiles above the window's lower half that nobody else owns in the plane's last block
    if (ng > 0 && part == 1) {
      const int tr_up = (blk == nqb - 1 && rb < 2 && 4 * blk + 2 + rb < NT) ? 4 * blk + 2 + rb : -1;
#pragma unroll 1
      for (int cnd = 0; cnd < 2; ++cnd) {
        const int tr = cnd == 0 ? ((rows_low && new_plane) ? tr_low : -1) : tr_up;
        if (tr >= 0) rows_step(tr);
      }
    }

    PSTAMP(sbase + 4);
    f32x16 o0 = {0}, o1 = {0};
    float m_run = -INFINITY, l_run = 0.f;
    const bool q_ok = q < p.S;
    const bool qblk_valid = q0 + 31 < valid_len, qblk_pad = q0 >= valid_len, qblk_in = q0 + 31 < p.S;
    const uint32_t drop_base = drop_row_base(sd.lo, sd.hi, (uint32_t)bn, (uint32_t)q);

    // the block's band tiles b0 .. b1; wave A takes the first two (after the peeled global keys), wave B the rest
    const int b0 = max(q0 - W, 0) >> 5;
    const int b1 = min(q0 + 31 + W, p.S - 1) >> 5;
    const int nA = min(2, b1 - b0 + 1);
    const int t_begin = !live ? 1 : (part == 0 ? b0 : b0 + nA);
    const int t_end = !live ? 0 : (part == 0 ? b0 + nA - 1 : b1);           // inclusive

    // ---- peeled step: the global keys outside this wave's band tiles, registers 0..3 only ----------------------------
    if (live && part == 0 && ng > 0 && !(p.pat.g0 >= b0 * 32 && p.pat.g0 + ng - 1 <= b1 * 32 + 31)) {
      Frag<T> kf;
      {
        const int rr = r & 7;
        const unsigned char* row = gk + rr * 128 + ((h ^ ((rr >> 1) & 1)) << 6);
#pragma unroll
        for (int s = 0; s < 4; ++s) kf.v[s] = *reinterpret_cast<const bf16x8*>(row + s * 16);
      }
      f32x16 c = {0};
      c = mma_rows(kf, qf, c);                               // registers 0..3: key g0 + i + 4h
      float s2[4];
      const bool qv = q < valid_len;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int kk = p.pat.g0 + i + 4 * h;
        const bool present = (i + 4 * h < ng) && !(kk >= b0 * 32 && kk <= b1 * 32 + 31);
        float rel = 0.f;
        if (HAS_REL) rel = trow[min(max(kk - q, -m), m) + m];
        float sv = fmaf(c[i], p.sscale, rel);
        sv = ((kk < valid_len) == qv) ? sv : sv + p.mask_add;
        s2[i] = present ? sv : -INFINITY;
      }
      m_run = half_max(fmaxf(fmaxf(s2[0], s2[1]), fmaxf(s2[2], s2[3])));      // nothing accumulated yet: no rescale
      float pr[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        pr[i] = __builtin_amdgcn_exp2f(s2[i] - m_run);
        l_run += pr[i];
        if (DROP) pr[i] = drop_bits16(drop_base, (uint32_t)(p.pat.g0 + i + 4 * h)) >= t16 ? pr[i] : 0.f;
      }
#pragma unroll
      for (int i = 4; i < 8; ++i) pr[i] = 0.f;
      {
        const int li = lane & 15, cb = (lane >> 4) & 1;
        const bf16x8 pf = pack8_bf16(pr);
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const int row = 4 * h + (li >> 2);
          const int within = 32 * cb + 8 * (li & 3);
          const int off0 = row * 128 + ((db ^ ((row >> 1) & 1)) << 6) + within;
          const int row1 = row + 8;
          const int off1 = row1 * 128 + ((db ^ ((row1 >> 1) & 1)) << 6) + within;
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(gv + off0));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(gv + off1));
          bf16x8 vf;
          bf16x4 lo4 = __builtin_bit_cast(bf16x4, lo), hi4 = __builtin_bit_cast(bf16x4, hi);
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) { vf[jj] = lo4[jj]; vf[4 + jj] = hi4[jj]; }
          if (db == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o0, 0, 0, 0);
          else o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o1, 0, 0, 0);
        }
      }
    }

    // ---- band tiles, straight out of the ring -----------------------------------------------------------------------
    for (int tile = t_begin; tile <= t_end; ++tile) {
      const int k0 = tile * 32;
      const unsigned char* klds = kring + (tile & 7) * 4096;
      const unsigned char* vlds = vring + (tile & 7) * 4096;
      Frag<T> kf;
      frag_from_tile(kf, klds, lane);
      f32x16 c = {0};
      c = mma_rows(kf, qf, c);     // S^T [key x q]

      const int dmin = k0 - (q0 + 31), dmax = k0 + 31 - q0;
      const bool in_range = (k0 + 31 < p.S) && qblk_in;
      const bool seg_all = (qblk_valid && k0 + 31 < valid_len) || (qblk_pad && k0 >= valid_len);
      const bool band_all = dmin >= -W && dmax <= W;
      const bool plain = in_range && seg_all && band_all;
      const bool far_neg = dmax <= -m, far_pos = dmin >= m;
      const bool one_id = !HAS_REL || far_neg || far_pos;
      const bool no_gkey = ng == 0 || k0 + 31 < p.pat.g0 || k0 >= p.pat.g0 + ng;
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
        const int kb2 = k0 + 4 * h;
        const bool qv = q < valid_len;
        const unsigned W2 = 2u * (unsigned)W;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int ci = (i & 3) + 8 * (i >> 2);
          const int kk = kb2 + ci, d = dbase + ci;
          const bool near = (unsigned)(d + W) <= W2;
          const bool gkey = (unsigned)(kk - p.pat.g0) < (unsigned)ng;
          const bool segm = (kk < valid_len) == qv;
          const bool keep = (int)segm & ((int)near | (int)gkey);
          float rel = 0.f;
          if (HAS_REL) rel = trow[min(max(d, -m), m) + m];
          float sv = fmaf(c[i], p.sscale, rel);
          sv = keep ? sv : sv + p.mask_add;
          s2[i] = kk < p.S ? sv : -INFINITY;
        }
      }
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
      if (DROP) {                            // 16 bits per element, one hash per key pair; 1 / keep in the epilogue
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

    PSTAMP(sbase + 5);
    // ---- the next block's Q rows are requested NOW (into the registers of this block's, dead since the last S product):
    //      their latency runs under the merge and the store ----
    const bool next_same = kb + 1 < k_end && (kb + 1) / nqb == bn;
    if (next_same) {
      const int q0n = q0 + 128;
      const auto rq = make_rsrc(Qb, (unsigned)(p.S - 1) * qs1b + 128);
#pragma unroll
      for (int s = 0; s < 4; ++s) qf.v[s] = buf16(rq, (unsigned)r * qs1b + 64 * h + 16 * s, (unsigned)q0n * qs1b);
    }

    // ---- pair merge + epilogue: each wave keeps the 32-column half of O^T it stores (A: d < 32, B: d >= 32) and hands
    //      the other half to its partner through a ring slot the NEXT block's new tiles will take (slots (t1w + 1 + rb) & 7
    //      of the K / V rings: free, or holding a tile that died with this block); row maxima and sums through the
    //      pair's table area --------------------------
    const float l_own = half_sum(l_run);
    __syncthreads();          // (2) every wave is done with the tiles (and with the tables)
    PSTAMP(sbase + 6);
    float* mine = reinterpret_cast<float*>((wave >> 2 ? vring : kring) + ((t1w + 1 + rb) & 7) * 4096);      // (free, or dead with this block)
    float* mlx = HAS_REL ? tab : reinterpret_cast<float*>(smem + 16 * 4096 + 3072) + rb * 128;   // (no tables without a relative term)
    float* mine_ml = mlx + part * 64;
    {
      if (h == 0) { mine_ml[r] = m_run; mine_ml[32 + r] = l_own; }
#pragma unroll
      for (int i = 0; i < 16; ++i) mine[i * 64 + lane] = part == 0 ? o1[i] : o0[i];
    }
    __syncthreads();          // (3)
    PSTAMP(sbase + 7);
    if (q_ok && !(p.skip_global_rows && is_global(p.pat, q))) {
      const float* theirs = reinterpret_cast<const float*>((wave >> 2 ? kring : vring) + ((t1w + 1 + rb) & 7) * 4096);
      const float* their_ml = mlx + (part ^ 1) * 64;
      const float m_o = their_ml[r], l_o = their_ml[32 + r];
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
          x[j] = (__bf16)(own * f_own + theirs[(4 * g + j) * 64 + lane] * f_oth);
        }
        *reinterpret_cast<bf16x4*>(O + 8 * g + 4 * h) = x;
      }
      if (p.lse && h == 0 && part == 0) p.lse[((long)b * p.N + n) * p.S + q] = (m_tot + log2f(l_tot)) * kLn2;
    }
    PSTAMP(sbase + 8);
    ++plane_blocks;

    // ---- plane (or walk) end: this walk's rows partials, the plane's ticket, the last arriver's merge -------------------
    if (ng > 0 && !next_same) {
      const int n_q = min(8, ng);
      const int L0 = (bn * nqb) / p.pw_walk;                   // first walk that holds blocks of this plane
      if (part == 1) {
        // compact form of stream rb: (O[8][64], m[8], l[8]) -> 8 x 66 floats, write-through
        float* part_out = ws_part + (((size_t)bn * p.walk_maxseg + (L - L0)) * 4 + rb) * (8 * 66);
        const bool empty = rows_fresh;                         // this walk contracted no tile of the plane
        float lsum[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) lsum[i] = empty ? 0.f : pw_h32_sum(ws_state[512 + lane * 4 + i]);
        if (r == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            __hip_atomic_store(part_out + (4 * h + i) * 66 + 64, empty ? -1.0e30f : ws_state[768 + 4 * h + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(part_out + (4 * h + i) * 66 + 65, lsum[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        if (r < 8) {
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            __hip_atomic_store(part_out + r * 66 + kap(i, h), empty ? 0.f : ws_state[(r + 8 * h) * 32 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(part_out + r * 66 + 32 + kap(i, h), empty ? 0.f : ws_state[(r + 8 * h) * 32 + 16 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      int* flag = reinterpret_cast<int*>(smem + 16 * 4096 + 3072);      // pair 0's table area (dead: the next block, if any, rebuilds it)
      if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(p.sync + bn, (unsigned)plane_blocks, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old + (unsigned)plane_blocks == (unsigned)nqb;
        if (last) {
          __hip_atomic_store(p.sync + bn, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // left zero for the next call
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        flag[0] = last;
      }
      __syncthreads();
      const int is_last = flag[0];
      const int L1 = ((bn + 1) * nqb - 1) / p.pw_walk;         // last walk with blocks of this plane
      const int n_parts = (L1 - L0 + 1) * 4;
      const float* pbase = ws_part + (size_t)bn * p.walk_maxseg * 4 * (8 * 66);
      float* ml = reinterpret_cast<float*>(flag) + 16;         // (max, sum) of every (partial, row): one parallel round trip
      if (is_last) {
        for (int idx = (int)threadIdx.x; idx < n_parts * 8; idx += 512) {
          const float* ps = pbase + (size_t)(idx >> 3) * (8 * 66) + (idx & 7) * 66;
          ml[2 * idx] = __hip_atomic_load(ps + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ml[2 * idx + 1] = __hip_atomic_load(ps + 65, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      __syncthreads();                                         // (flag read by everybody before the tables are rebuilt)
      if (is_last && wave < n_q) {
        const int qq = wave, d = lane;
        const float* base = pbase + qq * 66 + d;
        float Mx = -INFINITY;
        for (int s = 0; s < n_parts; ++s) Mx = fmaxf(Mx, ml[2 * (s * 8 + qq)]);
        float Lt = 0.f, At = 0.f;
        for (int s0 = 0; s0 < n_parts; s0 += 16) {            // 16 loads in flight per lane (the serial form cost a round trip each)
          float x[16];
#pragma unroll
          for (int j = 0; j < 16; ++j)
            x[j] = __hip_atomic_load(base + (size_t)min(s0 + j, n_parts - 1) * (8 * 66), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int s = min(s0 + j, n_parts - 1);
            const float wgt = s0 + j < n_parts ? __builtin_amdgcn_exp2f(ml[2 * (s * 8 + qq)] - Mx) : 0.f;
            Lt = fmaf(wgt, ml[2 * (s * 8 + qq) + 1], Lt);
            At = fmaf(wgt, x[j], At);
          }
        }
        const int qg = p.pat.g0 + qq;
        T* O = reinterpret_cast<T*>(p.out) + (long)b * p.os[0] + (long)qg * p.os[1] + (long)n * p.os[2];
        O[d] = (T)(At * (DROP ? p.inv_keep : 1.f) / Lt);
        if (p.lse && d == 0) p.lse[((long)b * p.N + n) * p.S + qg] = (Mx + log2f(Lt)) * kLn2;
      }
    }
    PSTAMP(sbase + 9);
    __syncthreads();          // (4) everybody has read the merge area and the tables: the next block may overwrite them
  }
}

// ---- host side ----------------------------------------------------------------------------------------------------
// Blocks per walk so that all workgroups are resident at once (two per CU); fills pw_walk / walk_maxseg; returns the grid.
int fwd_pwin_plan(FwdParams& p, int target_wgs) {
  const int nqb = (p.S + 127) / 128, total = p.B * p.N * nqb;
  int walk = (total + target_wgs - 1) / target_wgs;
  if (walk < 1) walk = 1;
  p.pw_walk = walk;
  p.walk_maxseg = (nqb + walk - 1) / walk + 1;               // walks that can hold blocks of one plane
  int grid = (total + walk - 1) / walk;
  return grid;
}
size_t fwd_pwin_workspace_bytes(int B, int N, int S, int target_wgs) {
  const int nqb = (S + 127) / 128, total = B * N * nqb;
  const int walk = std::max(1, (total + target_wgs - 1) / target_wgs);
  const size_t grid = (size_t)(total + walk - 1) / walk;
  const size_t maxseg = (size_t)(nqb + walk - 1) / walk + 1;
  return (grid * (8 * 34 + 4 * kPwState) + (size_t)B * N * maxseg * 4 * 8 * 66) * sizeof(float);
}

hipError_t launch_attn_fwd_pwin_bf16(const FwdParams& p, int grid_size, hipStream_t st) {
  const bool rel = p.R > 0 && p.pat.id_mode == 1;
  const bool drop = p.drop_thresh != 0;
  const int lds = 16 * 4096 + 3072 + std::max(rel ? 4 * 32 * p.tstride * 4 : 2048, 64 + p.walk_maxseg * 4 * 64);      // ring | global-key rows + scratch | tables (or just the merge's row maxima / sums)
  auto go = [&](auto kern) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, dim3(grid_size), dim3(512), lds, st, p);
  };
  if (rel) { if (drop) go(attn_fwd_pwin_bf16_kernel<1, true>); else go(attn_fwd_pwin_bf16_kernel<1, false>); }
  else     { if (drop) go(attn_fwd_pwin_bf16_kernel<0, true>); else go(attn_fwd_pwin_bf16_kernel<0, false>); }
  return hipGetLastError();
}

}  // namespace mmt
