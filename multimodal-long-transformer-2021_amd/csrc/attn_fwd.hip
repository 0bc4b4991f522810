// Forward of QkvRelativeAttention for gfx950 (MI355X).
//
// Replaces the dense einsum / one-hot / softmax chain the reference runs behind
// `RelativeTransformerLayers(inputs, att_mask, relative_att_ids)`
// (src/modeling/models/mmt_encoder.py:220-224; math: SURVEY.md App. A.3) with a
// flash-style kernel that never materialises [S,S] tensors:
//
//   one wave = 32 query rows of one (batch, head); it walks 32-key tiles.
//   S^T = K.Q^T by MFMA (keys in registers, query row on the lane), relative scores are
//   gathered from a per-wave LDS table T[q][id] = (q.E[id] + bias[id]) built once per
//   q-block by MFMA, softmax state is lane-local, O^T += V^T.P^T by MFMA with the P
//   accumulator reused as the B operand (no LDS round trip for P) and V^T fragments read
//   with ds_read_b64_tr_b16 from a wave-private, bank-swizzled LDS tile.
//
// Modes
//   kBand : structured pattern (band + global keys), ids/mask generated in-kernel   [K1]
//   kDense: literal reference operator, att_mask / rel_ids int32 [B,S,S] from HBM   [K3]
//   kRows : selected (global) query rows x a chunk of keys, partial (O,m,l) out     [K2]
#include "attn_kernels.h"

namespace mmt {

template <typename T> struct Frag;

// ------------------------------- bf16: 32x32x16 MFMA ---------------------------------
// MFMA k-index (8h + j) of step s is mapped to head-dim d = 32h + 8s + j, so each lane
// loads 64 contiguous bytes of its row (4 x 16 B).
template <> struct Frag<__bf16> {
  static constexpr int kSteps = 4;
  bf16x8 v[4];
  __device__ __forceinline__ void load_row(const __bf16* row, int h, bool ok) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (ok) v[s] = *reinterpret_cast<const bf16x8*>(row + 32 * h + 8 * s);
      else v[s] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
    }
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
  static constexpr int kSteps = 32;
  float v[32];
  __device__ __forceinline__ void load_row(const float* row, int h, bool ok) {
#pragma unroll
    for (int s = 0; s < 32; s += 4) {
      f32x4 t = ok ? *reinterpret_cast<const f32x4*>(row + 32 * h + s) : f32x4{0, 0, 0, 0};
      v[s] = t[0]; v[s + 1] = t[1]; v[s + 2] = t[2]; v[s + 3] = t[3];
    }
  }
};
__device__ __forceinline__ f32x16 mma_rows(const Frag<float>& a, const Frag<float>& b, f32x16 c) {
#pragma unroll
  for (int s = 0; s < 32; ++s) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[s], b.v[s], c, 0, 0, 0);
  return c;
}

__device__ __forceinline__ float half_xchg(float x) {  // value held by lane ^ 32
  return __shfl_xor(x, 32, 64);
}

constexpr int kTStride(int Rp) { return Rp + 1; }

// LDS carve per wave: T table [32][Rp+1] f32, then (bf16 only) V tile 32 x 128 B.
template <typename T, int Rp> struct WaveLds {
  static constexpr int kTBytes = 32 * kTStride(Rp) * 4;
  static constexpr int kTBytesAligned = (kTBytes + 15) & ~15;
  static constexpr int kVBytes = sizeof(T) == 2 ? 32 * 128 : 0;
  static constexpr int kBytes = kTBytesAligned + kVBytes;
};

template <typename T, int MODE, int Rp>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const FwdParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int r = lane & 31, h = lane >> 5;
  unsigned char* wl = smem + wave * WaveLds<T, Rp>::kBytes;
  float* tab = reinterpret_cast<float*>(wl);
  unsigned char* vlds = wl + WaveLds<T, Rp>::kTBytesAligned;
  (void)vlds;

  // ---- which 32 query rows does this wave own? ------------------------------------
  int bn, q0, chunk = 0, rowblk = 0;
  if (MODE == kRows) {
    // grid: ((n_chunks * n_rowblk + 3) / 4, B*N); one wave per (rowblk, chunk)
    const int item = blockIdx.x * 4 + wave;
    if (item >= p.n_chunks * p.n_rowblk) return;
    rowblk = item / p.n_chunks;
    chunk = item - rowblk * p.n_chunks;
    bn = blockIdx.y;
    q0 = p.pat.g0 + rowblk * 32;
  } else {
    const int nqb = (p.S + 127) >> 7;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
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

  Frag<T> qf;
  qf.load_row(Q + (long)(q_ok ? q : 0) * p.qs[1], h, q_ok);

  // ---- relative-score table T[q][id] = (q.E[id] + bias[id]) * tscale  (log2 domain) ---
  if (p.R > 0) {
    const T* E = reinterpret_cast<const T*>(p.emb) + (long)n * 64;
#pragma unroll
    for (int rb = 0; rb < Rp / 32; ++rb) {
      const int rr = rb * 32 + r;
      Frag<T> ef;
      ef.load_row(E + (long)(rr < p.R ? rr : 0) * p.N * 64, h, rr < p.R);
      f32x16 c = {0};
      c = mma_rows(ef, qf, c);  // [id x q]
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int id = rb * 32 + kap(i, h);
        float bias = 0.f;
        if (p.bias && id < p.R) bias = (float)reinterpret_cast<const T*>(p.bias)[(long)id * p.N + n];
        tab[r * kTStride(Rp) + id] = (c[i] + bias) * p.tscale;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // ---- tile walk -----------------------------------------------------------------------
  f32x16 o0 = {0}, o1 = {0};
  float m_run = -INFINITY, l_run = 0.f;
  const int n_tiles = (p.S + 31) >> 5;
  int t_lo, t_hi, g_lo = 0, g_hi = -1;
  if (MODE == kBand) {
    const int lo = q0 - p.pat.radius, hi = q0 + 31 + p.pat.radius;
    t_lo = (lo < 0 ? 0 : lo) >> 5;
    t_hi = (hi >= p.S || hi < 0 /*overflow*/ ? p.S - 1 : hi) >> 5;
    if (p.pat.ng > 0) { g_lo = p.pat.g0 >> 5; g_hi = (p.pat.g0 + p.pat.ng - 1) >> 5; }
  } else if (MODE == kRows) {
    t_lo = chunk * p.chunk_tiles;
    t_hi = min(n_tiles, t_lo + p.chunk_tiles) - 1;
  } else {
    t_lo = 0; t_hi = n_tiles - 1;
  }
  const int n_band = t_hi - t_lo + 1;
  const int n_glob = g_hi - g_lo + 1;

  for (int it = 0; it < n_band + n_glob; ++it) {
    int t;
    if (it < n_band) t = t_lo + it;
    else { t = g_lo + (it - n_band); if (t >= t_lo && t <= t_hi) continue; }
    const int k0 = t * 32;

    // K fragment of key (k0 + r) and S^T = K.Q^T
    Frag<T> kf;
    {
      const int kk = k0 + r;
      kf.load_row(K + (long)(kk < p.S ? kk : 0) * p.ks[1], h, kk < p.S);
    }
    // V tile -> wave-private LDS (bf16) while the QK^T MFMAs run
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int ci = lane + 64 * u, row = ci >> 3, ch = ci & 7;
        const int kk = k0 + row;
        bf16x8 t8 = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (kk < p.S) t8 = *reinterpret_cast<const bf16x8*>(V + (long)kk * p.vs[1] + ch * 8);
        const int off = row * 128 + ((((ch >> 2) ^ ((row >> 1) & 1))) << 6) + (ch & 3) * 16;
        *reinterpret_cast<bf16x8*>(vlds + off) = t8;
      }
    }
    f32x16 c = {0};
    c = mma_rows(kf, qf, c);

    // scores in the log2 domain: s2 = (qk [+ rel]) * scale*log2e [+ rel'] + mask_add
    float s2[16];
    float tmax = -INFINITY;
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
        if (p.pat.id_mode) id = rel_id(p.pat, q, kk);
      }
      float rel = 0.f;
      if ((unsigned)id < (unsigned)p.R) rel = tab[r * kTStride(Rp) + id];
      float s = fmaf(c[i], p.sscale, rel);
      if (!keep) s += p.mask_add;
      if (kk >= p.S) s = -INFINITY;
      s2[i] = s;
      tmax = fmaxf(tmax, s);
    }
    tmax = fmaxf(tmax, half_xchg(tmax));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = exp2f(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
    float pr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      pr[i] = exp2f(s2[i] - m_new);
      psum += pr[i];
    }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }

    if (p.drop_thresh) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const uint32_t hsh = dropout_hash(p.seed_lo, p.seed_hi, (uint32_t)bn, (uint32_t)q,
                                          (uint32_t)(k0 + kap(i, h)));
        pr[i] = hsh >= p.drop_thresh ? pr[i] * p.inv_keep : 0.f;
      }
    }

    // O^T[d x q] += V^T[d x key] . P^T[key x q]
    if constexpr (sizeof(T) == 2) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const int li = lane & 15, cb = (lane >> 4) & 1;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (__bf16)pr[8 * s + j];
#pragma unroll
        for (int db = 0; db < 2; ++db) {
          const int row = 16 * s + 4 * h + (li >> 2);
          const int within = 32 * cb + 8 * (li & 3);  // byte offset inside the 64-B half
          const int off0 = row * 128 + ((db ^ ((row >> 1) & 1)) << 6) + within;
          const int row1 = row + 8;
          const int off1 = row1 * 128 + ((db ^ ((row1 >> 1) & 1)) << 6) + within;
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(vlds + off0));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4*)(vlds + off1));
          bf16x8 vf;
          bf16x4 lo4 = __builtin_bit_cast(bf16x4, lo), hi4 = __builtin_bit_cast(bf16x4, hi);
#pragma unroll
          for (int j = 0; j < 4; ++j) { vf[j] = lo4[j]; vf[4 + j] = hi4[j]; }
          if (db == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o0, 0, 0, 0);
          else o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o1, 0, 0, 0);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        const int kk = k0 + kap(s, h);
        float v0 = 0.f, v1 = 0.f;
        if (kk < p.S) {
          const float* vr = reinterpret_cast<const float*>(V) + (long)kk * p.vs[1];
          v0 = vr[r]; v1 = vr[32 + r];
        }
        o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, pr[s], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, pr[s], o1, 0, 0, 0);
      }
    }
  }

  // ---- epilogue --------------------------------------------------------------------------
  const float l_tot = l_run + half_xchg(l_run);
  if (MODE == kRows) {
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

// Combine the per-chunk partials of the global rows: one thread per (row, d).
template <typename T>
__global__ __launch_bounds__(64) void attn_rows_combine_kernel(const FwdParams p) {
  const int bn = blockIdx.y;
  const int row = blockIdx.x;  // 0 .. ng-1
  const int d = threadIdx.x;
  const int rowblk = row >> 5, rr = row & 31;
  const int b = bn / p.N, n = bn - b * p.N;
  const long slot0 = ((long)bn * p.n_rowblk + rowblk) * p.n_chunks;
  float M = -INFINITY;
  for (int c = 0; c < p.n_chunks; ++c) M = fmaxf(M, p.part_ml[(slot0 + c) * 64 + rr]);
  float L = 0.f, acc = 0.f;
  for (int c = 0; c < p.n_chunks; ++c) {
    const float w = exp2f(p.part_ml[(slot0 + c) * 64 + rr] - M);
    L += w * p.part_ml[(slot0 + c) * 64 + 32 + rr];
    acc += w * p.part_o[(slot0 + c) * (32 * 64) + rr * 64 + d];
  }
  const int q = p.pat.g0 + row;
  T* O = reinterpret_cast<T*>(p.out) + (long)b * p.os[0] + (long)q * p.os[1] + (long)n * p.os[2];
  O[d] = (T)(acc / L);
  if (p.lse && d == 0) p.lse[((long)b * p.N + n) * p.S + q] = (M + log2f(L)) * kLn2;
}

// ------------------------------------ launchers -----------------------------------------
template <typename T, int MODE, int Rp>
static hipError_t launch_one(const FwdParams& p, dim3 grid, hipStream_t st) {
  const int lds = 4 * WaveLds<T, Rp>::kBytes;
  hipLaunchKernelGGL((attn_fwd_kernel<T, MODE, Rp>), grid, dim3(256), lds, st, p);
  return hipGetLastError();
}

template <typename T, int MODE>
static hipError_t launch_rp(const FwdParams& p, dim3 grid, hipStream_t st) {
  if (p.R <= 32) return launch_one<T, MODE, 32>(p, grid, st);
  return launch_one<T, MODE, 64>(p, grid, st);
}

template <typename T>
static hipError_t launch_mode(const FwdParams& p, int mode, dim3 grid, hipStream_t st) {
  switch (mode) {
    case kBand: return launch_rp<T, kBand>(p, grid, st);
    case kDense: return launch_rp<T, kDense>(p, grid, st);
    default: return launch_rp<T, kRows>(p, grid, st);
  }
}

hipError_t launch_attn_fwd(const FwdParams& p, int mode, bool bf16, hipStream_t st) {
  dim3 grid;
  if (mode == kRows) grid = dim3((p.n_chunks * p.n_rowblk + 3) / 4, p.B * p.N);
  else grid = dim3(p.B * p.N * ((p.S + 127) / 128));
  return bf16 ? launch_mode<__bf16>(p, mode, grid, st) : launch_mode<float>(p, mode, grid, st);
}

hipError_t launch_rows_combine(const FwdParams& p, bool bf16, hipStream_t st) {
  dim3 grid(p.pat.ng, p.B * p.N);
  if (bf16) hipLaunchKernelGGL(attn_rows_combine_kernel<__bf16>, grid, dim3(64), 0, st, p);
  else hipLaunchKernelGGL(attn_rows_combine_kernel<float>, grid, dim3(64), 0, st, p);
  return hipGetLastError();
}

}  // namespace mmt
