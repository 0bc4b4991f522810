// Bodies of the backward combines of the global-row / global-key chunk partials, shared by the stand-alone
// kernels (attn_bwd.hip, general path) and by the lean path, which runs them as extra workgroups of the
// NEXT big launch (dK/dV kernel, dE reduce) instead of as launches of their own: a stream already orders
// the producer kernel before the consumer, so the extra blocks see complete partials and the two tiny,
// latency-bound launches (9 + 7 us per layer call, more when they queue behind co-running GEMM
// workgroups) disappear.
#pragma once
#include "attn_kernels.h"

namespace mmt {

// dQ row of global token `row` of plane bn: sum of the chunk partials + E^T . dRel; also publishes the row's
// dRel for the dE reduce.  64 threads (d = 0..63) cooperate; dr_s = 64 floats of shared memory for them;
// `sync` = a barrier over (at least) those 64 threads.
template <typename T, typename Sync>
__device__ __forceinline__ void dq_combine_row(const BwdParams& p, int bn, int row, int d, float* dr_s, Sync sync) {
  const int gblk = row >> 5, rr = row & 31;
  const int b = bn / p.N, n = bn - b * p.N;
  const int q = p.pat.g0 + row;
  const long slot0 = ((long)bn * p.n_gblk + gblk) * p.n_chunks;
  float acc = 0.f, dr = 0.f;
  for (int c = 0; c < p.n_chunks; ++c) {
    acc += p.part_dq[(slot0 + c) * (32 * 64) + rr * 64 + d];
    if (d < p.Rp) dr += p.part_dtab[(slot0 + c) * (32 * p.Rp) + rr * p.Rp + d];
  }
  if (d >= p.R) dr = 0.f;
  dr_s[d] = dr;
  if (d < p.Rp) p.drel[((long)bn * p.pat.ng + row) * p.Rp + d] = dr;
  sync();
  const T* E = reinterpret_cast<const T*>(p.emb) + (long)n * 64;
  for (int id = 0; id < p.R; ++id) acc = fmaf(dr_s[id], (float)E[(long)id * p.N * 64 + d], acc);
  T* DQ = reinterpret_cast<T*>(p.dq) + (long)b * p.qs[0] + (long)q * p.qs[1] + (long)n * p.qs[2];
  DQ[d] = (T)acc;
}

// dK / dV rows of global token `row` of plane bn: sum of the chunk partials.
template <typename T>
__device__ __forceinline__ void dkv_combine_row(const BwdParams& p, int bn, int row, int d) {
  const int gblk = row >> 5, rr = row & 31;
  const int b = bn / p.N, n = bn - b * p.N;
  const int k = p.pat.g0 + row;
  const long slot0 = ((long)bn * p.n_gblk + gblk) * p.n_chunks;
  float ak = 0.f, av = 0.f;
  for (int c = 0; c < p.n_chunks; ++c) {
    const float* base = p.part_dkv + (slot0 + c) * (2 * 32 * 64) + rr * 64 + d;
    ak += base[0];
    av += base[32 * 64];
  }
  reinterpret_cast<T*>(p.dk)[(long)b * p.ks[0] + (long)k * p.ks[1] + (long)n * p.ks[2] + d] = (T)ak;
  reinterpret_cast<T*>(p.dv)[(long)b * p.vs[0] + (long)k * p.vs[1] + (long)n * p.vs[2] + d] = (T)av;
}

}  // namespace mmt
