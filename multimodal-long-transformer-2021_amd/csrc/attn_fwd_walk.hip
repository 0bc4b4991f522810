// Forward kernel as a PLANE WALK (bf16, band + at most 8 contiguous global tokens, relative ids none or 1-D with the
// permuted table, R <= 32, radius <= 64): persistent workgroups, each walking a run of consecutive 32-row blocks of ONE
// (batch, head) plane DOWN THE DIAGONAL of the band.  Same operator, outputs and tile classes as the per-wave kernel
// (attn_fwd_band.hip) and the window kernel (attn_fwd_win.hip); what changes is the decomposition in time:
//
//   * The walk advances in SUPER-STEPS of two key tiles (64 keys).  Super-step T has tiles 2T, 2T + 1 of K and V in one
//     slot of a two-slot LDS ring; row block j needs tiles j - 2 .. j + 2, i.e. three consecutive super-steps.  A
//     workgroup = 8 waves; wave w owns the blocks j = w (mod 8) of its run, so at every super-step SIX waves work on
//     the ring's tiles (each on its own row block: 2 + 2 + 1 or 1 + 2 + 2 tiles over its three steps) and the pair
//     (2p, 2p + 1), p = (T + 2) & 3, is SPARE: it stores the block it has just finished, fetches Q of its next block,
//     builds that block's relative-score table, runs the peeled global-key step, issues the LDS-DMA of the next
//     super-step's tiles and does the workgroup's share of the global tokens' rows (below).  So the per-block fixed
//     work (Q fetch, table, store: more than half of a window-kernel workgroup's life) runs BESIDE
//     the tile work of the other six waves instead of in front of it, K / V slide by one super-tile per step (every
//     tile is fetched once per run, 4 KiB + 4 KiB per 32 rows), and kernel arguments, the E rows, the bias row and the
//     global keys' rows are set up once per run.
//   * One barrier per super-step (raw s_barrier: it must not drain the LDS-DMA that is in flight across it).  The DMA
//     of super-tile T + 1 is issued by the spare pair at the start of step T into the slot step T - 1 has left, and
//     waited for (vmcnt(0)) by the same two waves right before the barrier that ends step T.
//   * Rows of the global tokens (dense rows): no workgroups of their own.  The spare waves take them in the flipped
//     orientation of attn_fwd_win.hip (lane = key, the 8 rows in 4 accumulator registers) against the tile their
//     workgroup has in the ring anyway -- one tile per spare wave and super-step, the run's blocks [jb, je) <-> key
//     tiles [jb, je) -- with the running (m, l, O) of the two tile streams in LDS.  At the end of the run the
//     workgroup writes ONE partial per global row to the workspace (write-through stores), draws a ticket from the
//     plane's arrival counter (mmt_attn_desc.sync), and the workgroup that arrives last merges the plane's partials:
//     no combine launch, no spinning, any placement (cdna_hip_programming.md, Guideline 16: sc1 stores, every
//     storing wave drained, one agent-scope add; the last arriver acquires, then loads with sc1).
//
// LDS per workgroup (1-D ids with 2m + 1 <= 25, 8 global tokens): 32 KiB ring + 8 tables of 3,328 B + E image 4 KiB +
// global keys' rows 3 KiB + rows state 6.3 KiB + small = 75.8 KiB: two workgroups per CU, 16 waves, <= 128 VGPRs.
#include "attn_lean.h"

namespace mmt {

namespace {

constexpr int kSlotBytes = 16384;        // K tile 2T | K tile 2T + 1 | V tile 2T | V tile 2T + 1
constexpr int kRowsState = 2048 + 1024 + 64;      // O^T of 8 rows (16 lanes x 32 floats) | per-lane row sums | 8 maxima

__device__ __forceinline__ unsigned walk_lds_u32(const void* p) {
  return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ float h32_max(float x) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
  return x;
}
__device__ __forceinline__ float h32_sum(float x) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
  return x;
}
__device__ __forceinline__ void step_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

struct WalkLds {
  int tab, eimg, btab, gk, gv, rst, pbuf, tabg, qimg, flag, total;
};
__host__ __device__ inline WalkLds walk_lds(int ng, int tstride, bool rel) {
  const int ngrp = (ng + 7) / 8;
  WalkLds L;
  int o = 2 * kSlotBytes;
  L.tab = o; o += rel ? 8 * 32 * tstride * 4 : 0;
  L.eimg = o; o += rel ? 4096 : 0;
  L.btab = o; o += rel ? 128 : 0;
  L.gk = o; o += ngrp * 1024;
  L.gv = o; o += ngrp ? (ngrp + 1) * 1024 : 0;
  L.rst = o; o += ng ? 2 * kRowsState : 0;
  L.pbuf = o; o += ng ? 2 * 512 : 0;
  L.tabg = o; o += (ng && rel) ? 8 * tstride * 4 : 0;
  L.qimg = o; o += ng ? 1024 : 0;
  L.flag = o; o += 16 + 32;
  L.total = o;
  return L;
}

}  // namespace

template <int REL, bool DROP>        // REL: 0 no relative term, 1 = 1-D ids (permuted table, Rp = 32)
__global__ __launch_bounds__(512, 4) void attn_fwd_walk_bf16_kernel(const FwdParams p) {
  using T = __bf16;
  constexpr bool HAS_REL = REL != 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int pair = wave >> 1, e = wave & 1;
  const int tstride = p.tstride;
  const int ng = p.pat.ng, ngrp = (ng + 7) >> 3;
  const WalkLds L = walk_lds(ng, tstride, HAS_REL);
  unsigned char* ring = smem;
  float* tab = reinterpret_cast<float*>(smem + L.tab) + wave * 32 * tstride;
  unsigned char* eimg = smem + L.eimg;
  float* btab = reinterpret_cast<float*>(smem + L.btab);
  unsigned char* gk = smem + L.gk;
  unsigned char* gv = smem + L.gv;
  float* rst = reinterpret_cast<float*>(smem + L.rst + e * kRowsState);            // this wave's tile stream (parity e)
  __bf16* pbuf = reinterpret_cast<__bf16*>(smem + L.pbuf + e * 512);
  float* tabg = reinterpret_cast<float*>(smem + L.tabg);
  unsigned char* qimg = smem + L.qimg;
  int* flag = reinterpret_cast<int*>(smem + L.flag);
  uint32_t* gdrop = reinterpret_cast<uint32_t*>(smem + L.flag + 16);     // dropout row bases of the 8 global rows
#ifdef MMT_STAMP
  long long* dbg = nullptr;
  {
    const int sel = blockIdx.x == 8 ? 0 : (blockIdx.x == 301 ? 1 : -1);
    if (p.dbg && sel >= 0) dbg = p.dbg + (sel * 8 + wave) * 64;
  }
#define WSTAMP(i) do { if (dbg && (threadIdx.x & 63) == 0 && (i) < 64) dbg[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define WSTAMP(i) do { } while (0)
#endif
  WSTAMP(0);

  // ---- work item: (plane, run of row blocks).  Groups of workgroups that share an XCD (blockIdx % 8) walk a
  //      contiguous range of planes; the runs of one plane are as equal as whole pairs of blocks allow. ----
  int bn, seg, nseg;
  {
    const int ngroups = p.walk_groups, ppg = (p.B * p.N) / ngroups;
    const int x = ngroups == 8 ? ((int)blockIdx.x & 7) : 0, i = ngroups == 8 ? ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int hi_items = p.walk_nhi * (p.walk_nseg + 1);
    int pl;
    if (i < hi_items) { pl = i / (p.walk_nseg + 1); seg = i - pl * (p.walk_nseg + 1); nseg = p.walk_nseg + 1; }
    else { const int i2 = i - hi_items; pl = i2 / p.walk_nseg; seg = i2 - pl * p.walk_nseg; pl += p.walk_nhi; nseg = p.walk_nseg; }
    bn = x * ppg + pl;
  }
  const int NT = (p.S + 31) >> 5, U = (NT + 1) >> 1;
  const int jb = 2 * ((seg * U) / nseg), je = min(2 * (((seg + 1) * U) / nseg), NT);
  const int b = bn / p.N, n = bn - b * p.N;
  const int valid_len = p.valid_len ? p.valid_len[b] : p.S;
  const int W = p.pat.radius, m = p.pat.m;

  const unsigned qs1b = (unsigned)p.qs[1] * 2, ks1b = (unsigned)p.ks[1] * 2, vs1b = (unsigned)p.vs[1] * 2;
  const T* Qb = reinterpret_cast<const T*>(p.q) + (long)b * p.qs[0] + (long)n * p.qs[2];
  const unsigned char* Kb = reinterpret_cast<const unsigned char*>(reinterpret_cast<const T*>(p.k) + (long)b * p.ks[0] + (long)n * p.ks[2]);
  const unsigned char* Vb = reinterpret_cast<const unsigned char*>(reinterpret_cast<const T*>(p.v) + (long)b * p.vs[0] + (long)n * p.vs[2]);
  const auto rq = make_rsrc(Qb, (unsigned)(p.S - 1) * qs1b + 128);
  const SeedPair sd = effective_seed(p.seed_lo, p.seed_hi, p.epoch);
  const uint32_t t16 = p.drop_thresh;

  // position of this lane's 16 bytes inside a staged 8-row piece (tile image: 64-byte halves swapped on odd row pairs)
  const int drow = lane >> 3, dpos = lane & 7;
  const int dch = (((dpos >> 2) ^ ((drow >> 1) & 1)) << 2) | (dpos & 3);

  // ================= set-up of the run (once per workgroup) =================
  const int qg0 = p.pat.g0, n_q = min(8, ng);
  if (wave == 0 && HAS_REL) {
    // E rows in table-column order as a tile image (row c <- relative id icol(m, c)); bias row by table column
    const T* Eb = reinterpret_cast<const T*>(p.emb) + (long)n * 64;
    const unsigned es1b = (unsigned)p.N * 128;
    const auto re = make_rsrc(Eb, (unsigned)(p.R - 1) * es1b + 128);
    Frag<T> ef;
#pragma unroll
    for (int s = 0; s < 4; ++s) ef.v[s] = buf16(re, (unsigned)icol(m, r) * es1b + 64 * h + 16 * s, 0u);
    const int idc = icol(m, r);
    const bool use = p.bias != nullptr && idc < p.R;
    const T* bp = reinterpret_cast<const T*>(p.bias ? p.bias : p.emb) + (use ? (long)idc * p.N + n : 0);
    const unsigned braw = *reinterpret_cast<const unsigned short*>(bp);
    unsigned char* row = eimg + r * 128 + ((h ^ ((r >> 1) & 1)) << 6);
#pragma unroll
    for (int s = 0; s < 4; ++s) *reinterpret_cast<bf16x8*>(row + s * 16) = ef.v[s];
    if (h == 0) btab[r] = use ? __builtin_bit_cast(float, braw << 16) * p.tscale : 0.f;
    if (ng > 0) {
      // the 8 global query rows (fragment image, 16 lanes x 64 B) and their relative-score rows
      const auto rqg = make_rsrc(Qb + (long)qg0 * p.qs[1], (unsigned)(n_q - 1) * qs1b + 128);
      Frag<T> qf;
#pragma unroll
      for (int s = 0; s < 4; ++s) qf.v[s] = buf16(rqg, (unsigned)r * qs1b + 64 * h + 16 * s, 0u);
      if (r < 8) {
#pragma unroll
        for (int s = 0; s < 4; ++s) *reinterpret_cast<bf16x8*>(qimg + (r * 2 + h) * 64 + s * 16) = qf.v[s];
      }
      wave_fence();
      f32x16 c = {0};
      c = mma_rows(ef, qf, c);                                               // [column x query]
      if (r < 8) {
#pragma unroll
        for (int i = 0; i < 16; ++i) tabg[r * tstride + min(kap(i, h), 2 * m + 1)] = fmaf(c[i], p.tscale, btab[kap(i, h)]);
      }
    }
  } else if (wave == 0 && ng > 0) {        // no relative term: the query image only
    const auto rqg = make_rsrc(Qb + (long)qg0 * p.qs[1], (unsigned)(n_q - 1) * qs1b + 128);
    if (r < 8) {
#pragma unroll
      for (int s = 0; s < 4; ++s) *reinterpret_cast<bf16x8*>(qimg + (r * 2 + h) * 64 + s * 16) = buf16(rqg, (unsigned)r * qs1b + 64 * h + 16 * s, 0u);
    }
  }
  if (ng > 0) {
    // rows of the global keys: K groups, then V groups + 8 finite rows behind the last one
    const int n_gp = 2 * ngrp + 1;
    for (int pg = wave - 1; pg >= 0 && pg < n_gp; pg += 7) {
      const bool isv = pg >= ngrp;
      const unsigned grow = (unsigned)min(p.pat.g0 + (isv ? pg - ngrp : pg) * 8 + drow, p.S - 1);
      const bf16x8 x = *reinterpret_cast<const bf16x8*>((isv ? Vb + (size_t)grow * vs1b : Kb + (size_t)grow * ks1b) + dch * 16);
      *reinterpret_cast<bf16x8*>(gk + pg * 1024 + lane * 16) = x;
    }
    if (threadIdx.x < 8) gdrop[threadIdx.x] = drop_row_base(sd.lo, sd.hi, (uint32_t)bn, (uint32_t)(p.pat.g0 + (int)threadIdx.x));
    // running state of the two tile streams of the global rows: nothing seen yet
    for (int i = (int)threadIdx.x; i < 2 * kRowsState / 4; i += 512) {
      const int w = i % (kRowsState / 4);
      reinterpret_cast<float*>(smem + L.rst)[i] = w >= (2048 + 1024) / 4 ? -1.0e30f : 0.f;
    }
  }

  // ================= per-block state of this wave =================
  Frag<T> qf;
  f32x16 o0 = {0}, o1 = {0};
  float m_run = -INFINITY, l_run = 0.f, relfn = 0.f, relfp = 0.f;
  int cur_j = -1;                          // the row block this wave works on (-1: none)
  uint32_t drop_base = 0;

  const int T0 = (jb >> 1) - 2, T1 = ((je - 1) >> 1) + 1;      // first prologue step .. last block's last tiles (it stores there)
  __syncthreads();                         // set-up complete (nothing is in flight yet)
  WSTAMP(1);

  for (int Ts = T0; Ts <= T1; ++Ts) {
    WSTAMP(2 + 4 * (Ts - T0));
    // Everything that depends on the lane is re-derived per super-step from an opaque copy of the lane id: left to
    // itself, hipcc hoists the ~90 lane-dependent LDS addresses and masks of the step bodies out of this loop and
    // spills them (344 bytes of scratch per lane, reloaded inside the tile loops behind vmcnt waits).
    int lane = (int)(threadIdx.x & 63);
    asm volatile("" : "+v"(lane));
    const int r = lane & 31, h = lane >> 5;
    const int drow = lane >> 3, dpos = lane & 7;
    const int dch = (((dpos >> 2) ^ ((drow >> 1) & 1)) << 2) | (dpos & 3);
    const float* trow = tab + r * tstride;
    const int trow_addr = (int)walk_lds_u32(trow);
    const int phase = (Ts - pair + 1) & 3;
    if (phase == 3) {
      // ======================= SPARE super-step of this pair =======================
      // (1) LDS-DMA of the tile this wave owes the next super-step: tile 2 (Ts + 1) + e into slot (Ts + 1) & 1
      {
        const int tl = 2 * (Ts + 1) + e;
        if (tl >= max(jb - 2, 0) && tl <= min(je + 1, NT - 1)) {
          const unsigned dst = walk_lds_u32(ring + ((Ts + 1) & 1) * kSlotBytes + e * 4096);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const unsigned grow = (unsigned)min(tl * 32 + 8 * u + drow, p.S - 1);
            glds16(Kb + (size_t)grow * ks1b + dch * 16, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + 1024u * u)));
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const unsigned grow = (unsigned)min(tl * 32 + 8 * u + drow, p.S - 1);
            glds16(Vb + (size_t)grow * vs1b + dch * 16, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + 8192u + 1024u * u)));
          }
        }
      }
      // (the accumulators are dead from the block's last active step to (5): say so, or they stay allocated under the rows step)
#pragma unroll
      for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
      m_run = -INFINITY; l_run = 0.f;
      // (3) next block of this wave: its Q rows are on their way since the end of the previous block (or fetched here for
      //     the first blocks of the run)
      const int jn = 2 * (Ts + 2) + e;
      const bool has_next = jn >= jb && jn < je;
      const bool q_on_its_way = cur_j >= 0;                 // fetched at the end of the previous block (cur_j + 8 == jn)
      cur_j = has_next ? jn : -1;
      const int q0 = jn * 32, q = q0 + r;
      if (has_next && !q_on_its_way) {
#pragma unroll
        for (int s = 0; s < 4; ++s) qf.v[s] = buf16(rq, (unsigned)r * qs1b + 64 * h + 16 * s, (unsigned)q0 * qs1b);
      }
      // The rest of the step is three dependency chains -- (R) one tile of the global tokens' rows, flipped orientation;
      // (T) the next block's relative-score table; (P) its peeled global-key step -- written INTERLEAVED, front halves
      // (LDS reads + S products) first, so that the matrix-pipe and LDS latencies of one chain run under the vector
      // work of the others: back to back they made the spare waves the last to reach the barrier (6.7 k against
      // 5.5 k cycles for two band tiles; profiles/r04_walk_stamps*.txt).
      const int tr = 2 * Ts + e;
      const bool do_rows = ng > 0 && tr >= jb && tr < je;
      const unsigned char* rk_lds = ring + (Ts & 1) * kSlotBytes + e * 4096;
      const int b0n = max(q0 - W, 0) >> 5, b1n = min(q0 + 31 + W, p.S - 1) >> 5;
      const bool do_peel = has_next && ng > 0 && !(p.pat.g0 >= b0n * 32 && p.pat.g0 + ng - 1 <= b1n * 32 + 31);   // else: all band-tile keys
      // ---- (R) front: S = Q_g . K^T of tile tr  [query x key]: registers 0..3
      f32x16 c_r = {0};
      float m_ref[4], l_loc[4];
      u32x4_t gb = {0, 0, 0, 0};
      if (do_rows) {
        Frag<T> kf, qg;
        frag_from_tile(kf, rk_lds, lane);
#pragma unroll
        for (int s = 0; s < 4; ++s) qg.v[s] = *reinterpret_cast<const bf16x8*>(qimg + ((r & 7) * 2 + h) * 64 + s * 16);
        c_r = mma_rows(qg, kf, c_r);
        const f32x4 lv = *reinterpret_cast<const f32x4*>(rst + 512 + lane * 4);
        const f32x4 mv = *reinterpret_cast<const f32x4*>(rst + 768 + 4 * h);
#pragma unroll
        for (int i = 0; i < 4; ++i) { l_loc[i] = lv[i]; m_ref[i] = mv[i]; }
        if (DROP) gb = *reinterpret_cast<const u32x4_t*>(gdrop + 4 * h);
      }
      // ---- (T) the table of the next block: T[q][col] = (q . E[id(col)] + bias) * scale * log2e
      if (has_next) drop_base = drop_row_base(sd.lo, sd.hi, (uint32_t)bn, (uint32_t)q);
      if (HAS_REL && has_next) {
        Frag<T> ef;
        frag_from_tile(ef, eimg, lane);
        float bcol[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 bv = *reinterpret_cast<const f32x4*>(btab + 8 * g + 4 * h);
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) bcol[4 * g + jj] = bv[jj];
        }
        f32x16 c = {0};
        c = mma_rows(ef, qf, c);   // [column x q]
#pragma unroll
        for (int i = 0; i < 16; ++i) tab[r * tstride + min(kap(i, h), 2 * m + 1)] = fmaf(c[i], p.tscale, bcol[i]);
      }
      // ---- (P) front: S^T of the 8 global keys against the next block's rows: registers 0..3 = key g0 + i + 4h
      f32x16 c_g = {0};
      if (do_peel) {
        Frag<T> kf;
        const int rr = r & 7;
        const unsigned char* row = gk + rr * 128 + ((h ^ ((rr >> 1) & 1)) << 6);
#pragma unroll
        for (int s = 0; s < 4; ++s) kf.v[s] = *reinterpret_cast<const bf16x8*>(row + s * 16);
        c_g = mma_rows(kf, qf, c_g);
      }
      // ---- (R) back: softmax of the tile against the running maxima, P . V into fresh accumulators, merge into the
      //      LDS copy of the stream's state (read again by the spare wave of the same parity one super-step on)
      if (do_rows) {
        const unsigned char* vlds = rk_lds + 8192;
        const int k = tr * 32 + r;
        const bool kv = k < valid_len, kin = k < p.S;
        float alpha4[4] = {1.f, 1.f, 1.f, 1.f};
        float s2[4], relv[4] = {0.f, 0.f, 0.f, 0.f};
        if (HAS_REL) {
          const int tabg_addr = (int)walk_lds_u32(tabg) + 4 * h * tstride * 4;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            relv[i] = *(lds_cfp)(size_t)(unsigned)(tabg_addr + i * tstride * 4 + 4 * med3i(k - (qg0 + 4 * h + i) + m, 0, 2 * m));
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(relv[i]));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int qi = 4 * h + i;
          float sc = fmaf(c_r[i], p.sscale, relv[i]);
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
            const float m_new = fmaxf(m_ref[i], h32_max(s2[i]));
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
            const uint32_t hsh = drop_pair_finish(gb[i], pterm);
            pr[i] = ((k & 1) ? (hsh >> 16) : (hsh & 0xFFFFu)) >= t16 ? pr[i] : 0.f;
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) pbuf[(4 * h + i) * 32 + r] = (__bf16)pr[i];
        wave_fence();
        f32x16 g0a = {0}, g1a = {0};
        {
          const int li = lane & 15, cb = (lane >> 4) & 1;
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            bf16x8 pf;
            {
              const bf16x4 z = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
              const __bf16* prow = pbuf + (r & 7) * 32 + 16 * s + 4 * h;   // keys {0..3, 8..11} + 4h + 16s
              s16x4 lo_r = *reinterpret_cast<const s16x4*>(prow), hi_r = *reinterpret_cast<const s16x4*>(prow + 8);
              asm volatile("" : "+v"(lo_r), "+v"(hi_r));
              const bf16x4 lo = r < 8 ? __builtin_bit_cast(bf16x4, lo_r) : z;
              const bf16x4 hi = r < 8 ? __builtin_bit_cast(bf16x4, hi_r) : z;
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) { pf[jj] = lo[jj]; pf[4 + jj] = hi[jj]; }
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
              for (int jj = 0; jj < 4; ++jj) { vf[jj] = lo4[jj]; vf[4 + jj] = hi4[jj]; }
              if (db == 0) g0a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, g0a, 0, 0, 0);
              else g1a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, g1a, 0, 0, 0);
            }
          }
        }
        // O^T_state = alpha . O^T_state + this tile's P . V, column = query (lanes r < 8 hold the 8 rows)
        {
          f32x4 lv, mv;
#pragma unroll
          for (int i = 0; i < 4; ++i) { lv[i] = l_loc[i]; mv[i] = m_ref[i]; }
          *reinterpret_cast<f32x4*>(rst + 512 + lane * 4) = lv;
          if (r == 0) *reinterpret_cast<f32x4*>(rst + 768 + 4 * h) = mv;
          float a = 1.f;
          if (grew) {                                                      // (wave-uniform)
            float* abuf = reinterpret_cast<float*>(pbuf);                  // (the P tile has been consumed)
            wave_fence();
            if (r == 0) {
#pragma unroll
              for (int i = 0; i < 4; ++i) abuf[4 * h + i] = alpha4[i];
            }
            wave_fence();
            a = abuf[r & 7];                                               // O^T columns are queries (lane & 31)
          }
          if (r < 8) {
            float* dstp = rst + (r + 8 * h) * 32;
#pragma unroll
            for (int i = 0; i < 16; i += 4) {
              f32x4 x0 = *reinterpret_cast<const f32x4*>(dstp + i), x1 = *reinterpret_cast<const f32x4*>(dstp + 16 + i);
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) { x0[jj] = fmaf(x0[jj], a, g0a[i + jj]); x1[jj] = fmaf(x1[jj], a, g1a[i + jj]); }
              *reinterpret_cast<f32x4*>(dstp + i) = x0;
              *reinterpret_cast<f32x4*>(dstp + 16 + i) = x1;
            }
          }
        }
      }
      WSTAMP(3 + 4 * (Ts - T0));
      // ---- (T) back, (P) back: the table is complete; the global keys outside the block's band tiles open its softmax
      if (HAS_REL && has_next) {
        wave_fence();
        relfn = trow[0];
        relfp = trow[2 * m];
      }
      if (do_peel) {
        float s2[4];
        const bool qv = q < valid_len;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int kk = p.pat.g0 + i + 4 * h;
          const bool present = (i + 4 * h < ng) && !(kk >= b0n * 32 && kk <= b1n * 32 + 31);
          float rel = 0.f;
          if (HAS_REL) rel = trow[min(max(kk - q, -m), m) + m];
          float sv = fmaf(c_g[i], p.sscale, rel);
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
        {   // O^T += V^T . P^T over the first 16 rows of the group's "tile" (rows 8..15: the finite pad rows; p = 0)
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
      // (6) the DMA issued in (1) has landed (and this wave's stores have left) before anybody reads the slot
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (cur_j >= 0) {
      // ======================= ACTIVE super-step: this wave's block against tiles 2 Ts, 2 Ts + 1 =======================
      const int q0 = cur_j * 32, q = q0 + r;
      const int b0 = max(q0 - W, 0) >> 5, b1 = min(q0 + 31 + W, p.S - 1) >> 5;
      const bool qblk_valid = q0 + 31 < valid_len, qblk_pad = q0 >= valid_len, qblk_in = q0 + 31 < p.S;
#pragma unroll 1
      for (int u = 0; u < 2; ++u) {
        const int tile = 2 * Ts + u;
        if (tile < b0 || tile > b1) continue;
        const int k0 = tile * 32;
        const unsigned char* klds = ring + (Ts & 1) * kSlotBytes + u * 4096;
        const unsigned char* vlds = klds + 8192;
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
          const int kb = k0 + 4 * h;
          const bool qv = q < valid_len;
          const unsigned W2 = 2u * (unsigned)W;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int ci = (i & 3) + 8 * (i >> 2);
            const int kk = kb + ci, d = dbase + ci;
            const bool near = (unsigned)(d + W) <= W2;
            const bool gkey = (unsigned)(kk - p.pat.g0) < (unsigned)ng;
            const bool segm = (kk < valid_len) == qv;
            const bool keep = (int)segm & ((int)near | (int)gkey);
            float rel = 0.f;
            if (HAS_REL) rel = trow[min(max(d, -m), m) + m];
            float s = fmaf(c[i], p.sscale, rel);
            s = keep ? s : s + p.mask_add;
            s2[i] = kk < p.S ? s : -INFINITY;
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
      WSTAMP(3 + 4 * (Ts - T0));
      // the block's Q rows are dead after its last S product: the next block's (8 further down the run) are fetched
      // into the same registers NOW, under the store below, the barrier and the spare step's other work
      if (phase == 2 && cur_j + 8 < je) {
#pragma unroll
        for (int s = 0; s < 4; ++s) qf.v[s] = buf16(rq, (unsigned)r * qs1b + 64 * h + 16 * s, (unsigned)((cur_j + 8) * 32) * qs1b);
      }
      // last super-step of the block: normalise, store (16-byte stores after a half-wave exchange), LSE -- here and not
      // in the spare step, where 32 live accumulators under the Q fetch and the rows step push the kernel into scratch
      if (phase == 2) {
        const int q = cur_j * 32 + r;
        const float l_tot = half_sum(l_run);
        const float inv = (DROP ? p.inv_keep : 1.f) / l_tot;
        const bool st_ok = q < p.S && !(p.skip_global_rows && is_global(p.pat, q));
        unsigned char* O = reinterpret_cast<unsigned char*>(reinterpret_cast<T*>(p.out) + (long)b * p.os[0] + (long)q * p.os[1] + (long)n * p.os[2]) + 16 * h;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
          for (int g = 0; g < 4; g += 2) {
            bf16x4 xa, xb;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              xa[jj] = (__bf16)((half ? o1[4 * g + jj] : o0[4 * g + jj]) * inv);
              xb[jj] = (__bf16)((half ? o1[4 * g + 4 + jj] : o0[4 * g + 4 + jj]) * inv);
            }
            uint2 a = __builtin_bit_cast(uint2, xa), bb = __builtin_bit_cast(uint2, xb);
            auto s0 = __builtin_amdgcn_permlane32_swap(a.x, bb.x, false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(a.y, bb.y, false, false);
            if (st_ok) *reinterpret_cast<uint4*>(O + 64 * half + 16 * g) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
          }
        }
        if (p.lse && h == 0 && st_ok) p.lse[((long)b * p.N + n) * p.S + q] = (m_run + log2f(l_tot)) * kLn2;
      }
    }
    WSTAMP(4 + 4 * (Ts - T0));
    step_barrier();
    WSTAMP(5 + 4 * (Ts - T0));
  }

  // ================= rows of the global tokens: this run's partial, the plane's ticket, the last arriver's merge =================
  if (ng == 0) return;
  {
    // wave w finishes query w: merge of the two tile streams
    const int qq = wave, d = lane;
    const float* st0 = reinterpret_cast<const float*>(smem + L.rst);
    const float* st1 = reinterpret_cast<const float*>(smem + L.rst + kRowsState);
    const int hq = qq >> 2, iq = qq & 3;
    const float m0 = st0[768 + qq], m1 = st1[768 + qq];
    const float M = fmaxf(m0, m1);
    const float w0 = __builtin_amdgcn_exp2f(m0 - M), w1 = __builtin_amdgcn_exp2f(m1 - M);
    // row sums: lanes (r, hq) of each stream hold partial sums of query 4 hq + iq
    float ls = 0.f;
    if (lane < 32) ls = w0 * st0[512 + (lane + 32 * hq) * 4 + iq] + w1 * st1[512 + (lane + 32 * hq) * 4 + iq];
    ls = h32_sum(ls);
    ls = __shfl(ls, 0, 64);
    const int dd = d & 31, hh = (dd >> 2) & 1, ii = (dd & 3) + 4 * (dd >> 3) + 16 * (d >> 5);
    const float acc = w0 * st0[(qq + 8 * hh) * 32 + ii] + w1 * st1[(qq + 8 * hh) * 32 + ii];
    // partial of (plane, run, query): 64 floats O, then m, l  (write-through stores; every storing wave drains)
    float* part = p.walk_part + (((size_t)bn * p.walk_maxseg + seg) * 8 + qq) * 66;
    if (qq < n_q) {
      __hip_atomic_store(part + d, acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (d == 0) {
        __hip_atomic_store(part + 64, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(part + 65, ls, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned ticket = __hip_atomic_fetch_add(p.sync + bn, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == (unsigned)(nseg - 1);
      if (last) {
        __hip_atomic_store(p.sync + bn, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // left zero for the next call
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      flag[0] = last;
    }
    __syncthreads();
    if (!flag[0] || qq >= n_q) return;
    const float* base = p.walk_part + ((size_t)bn * p.walk_maxseg * 8 + qq) * 66;
    float Mx = -INFINITY;
    for (int s = 0; s < nseg; ++s) Mx = fmaxf(Mx, __hip_atomic_load(base + (size_t)s * 8 * 66 + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    float Lt = 0.f, At = 0.f;
    for (int s = 0; s < nseg; ++s) {
      const float* ps = base + (size_t)s * 8 * 66;
      const float wgt = __builtin_amdgcn_exp2f(__hip_atomic_load(ps + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - Mx);
      Lt = fmaf(wgt, __hip_atomic_load(ps + 65, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), Lt);
      At = fmaf(wgt, __hip_atomic_load(ps + d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), At);
    }
    const int q = qg0 + qq;
    T* O = reinterpret_cast<T*>(p.out) + (long)b * p.os[0] + (long)q * p.os[1] + (long)n * p.os[2];
    O[d] = (T)(At * (DROP ? p.inv_keep : 1.f) / Lt);
    if (p.lse && d == 0) p.lse[((long)b * p.N + n) * p.S + q] = (Mx + log2f(Lt)) * kLn2;
  }
}

// ---- host side ----------------------------------------------------------------------------------------------------
int fwd_walk_lds_bytes(int ng, int tstride, bool rel) { return walk_lds(ng, tstride, rel).total; }

// Runs per plane: as many workgroups as fit the chip at once (2 per CU), shared out over the planes; a run is at least
// one pair of row blocks.  Fills the walk_* fields of `p`; returns the grid size.
int fwd_walk_plan(FwdParams& p, int target_wgs) {
  const int BN = p.B * p.N, NT = (p.S + 31) / 32, U = (NT + 1) / 2;
  const int ngroups = (BN % 8) == 0 ? 8 : 1, ppg = BN / ngroups;
  int per_group = target_wgs / ngroups;
  if (per_group > ppg * U) per_group = ppg * U;
  if (per_group < ppg) per_group = ppg;
  p.walk_groups = ngroups;
  p.walk_nseg = per_group / ppg;
  p.walk_nhi = per_group % ppg;
  p.walk_maxseg = p.walk_nseg + (p.walk_nhi ? 1 : 0);
  return ngroups * per_group;
}
size_t fwd_walk_workspace_bytes(int B, int N, int S) {      // upper bound over every plan: U runs per plane
  const int NT = (S + 31) / 32, U = (NT + 1) / 2;
  return (size_t)B * N * U * 8 * 66 * sizeof(float);
}

hipError_t launch_attn_fwd_walk_bf16(const FwdParams& p, int grid_size, hipStream_t st) {
  const bool rel = p.R > 0 && p.pat.id_mode == 1;
  const bool drop = p.drop_thresh != 0;
  const int lds = fwd_walk_lds_bytes(p.pat.ng, p.tstride, rel);
  auto go = [&](auto kern) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, dim3(grid_size), dim3(512), lds, st, p);
  };
  if (rel) { if (drop) go(attn_fwd_walk_bf16_kernel<1, true>); else go(attn_fwd_walk_bf16_kernel<1, false>); }
  else     { if (drop) go(attn_fwd_walk_bf16_kernel<0, true>); else go(attn_fwd_walk_bf16_kernel<0, false>); }
  return hipGetLastError();
}

}  // namespace mmt
