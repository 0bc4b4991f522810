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
  // thread d sums output column d of dQ and table columns d (and 64 + d when the table is 128 wide: dr_s then holds
  // 128 floats -- only the general kernels' combine launch runs at that width)
  float acc = 0.f, dr = 0.f, dr2 = 0.f;
  const bool wide = p.Rp > 64;
  const float* pq = p.part_dq + slot0 * (32 * 64) + rr * 64 + d;
  const float* pt = p.part_dtab + slot0 * (32 * p.Rp) + rr * p.Rp + (d < p.Rp ? d : 0);
  const long tq = 32 * 64, tt = 32 * p.Rp;
  int c = 0;
  for (; c + 4 <= p.n_chunks; c += 4) {          // four chunks' loads in flight, summed in chunk order
    float a[4], t[4], t2[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a[u] = pq[(c + u) * tq]; t[u] = pt[(c + u) * tt]; t2[u] = wide ? pt[(c + u) * tt + 64] : 0.f; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { acc += a[u]; dr += t[u]; dr2 += t2[u]; }
  }
  for (; c < p.n_chunks; ++c) { acc += pq[c * tq]; dr += pt[c * tt]; if (wide) dr2 += pt[c * tt + 64]; }
  if (d >= p.Rp) dr = 0.f;
  if (d >= p.R) dr = 0.f;
  if (64 + d >= p.R) dr2 = 0.f;
  dr_s[d] = dr;
  if (d < p.Rp) p.drel[((long)bn * p.pat.ng + row) * p.Rp + d] = dr;
  if (wide) {
    dr_s[64 + d] = dr2;
    p.drel[((long)bn * p.pat.ng + row) * p.Rp + 64 + d] = dr2;
  }
  sync();
  const T* E = reinterpret_cast<const T*>(p.emb) + (long)n * 64;
  for (int id = 0; id < p.R; ++id) acc = fmaf(dr_s[id], (float)E[(long)id * p.N * 64 + d], acc);
  T* DQ = reinterpret_cast<T*>(p.dq) + (long)b * p.qs[0] + (long)q * p.qs[1] + (long)n * p.qs[2];
  DQ[d] = (T)acc;
}

// dK / dV rows of global token `row` of plane bn: sum of the chunk partials (p.dkv_slots of them, in slot order).
template <typename T>
__device__ __forceinline__ void dkv_combine_row(const BwdParams& p, int bn, int row, int d) {
  const int gblk = row >> 5, rr = row & 31;
  const int b = bn / p.N, n = bn - b * p.N;
  const int k = p.pat.g0 + row;
  const long slot0 = ((long)bn * p.n_gblk + gblk) * p.dkv_slots;
  float ak = 0.f, av = 0.f;
  const float* base = p.part_dkv + slot0 * (2 * 32 * 64) + rr * 64 + d;
  int c = 0;
  for (; c + 4 <= p.dkv_slots; c += 4) {
    float a[4], t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a[u] = base[(long)(c + u) * (2 * 32 * 64)]; t[u] = base[(long)(c + u) * (2 * 32 * 64) + 32 * 64]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { ak += a[u]; av += t[u]; }
  }
  for (; c < p.dkv_slots; ++c) { ak += base[(long)c * (2 * 32 * 64)]; av += base[(long)c * (2 * 32 * 64) + 32 * 64]; }
  reinterpret_cast<T*>(p.dk)[(long)b * p.ks[0] + (long)k * p.ks[1] + (long)n * p.ks[2] + d] = (T)ak;
  reinterpret_cast<T*>(p.dv)[(long)b * p.vs[0] + (long)k * p.vs[1] + (long)n * p.vs[2] + d] = (T)av;
}

}  // namespace mmt
