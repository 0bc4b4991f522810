// Host-only stand-ins for the kernel launchers, for the sanitizer build of the C-ABI shim (`make asan`): they launch
// nothing, check that every workspace pointer the shim derived lies inside the caller's workspace, and record what
// was asked for so that the driver can assert on it.  CPU only -- never built into libmmt_attn.so.
#include <cstdio>
#include <cstdlib>

#include "../attn_kernels.h"

extern "C" {
const unsigned char* g_ws_lo = nullptr;
const unsigned char* g_ws_hi = nullptr;
int g_launches = 0;
int g_last_kind = 0;   // 1 fwd general, 2 fwd band, 3 fwd window, 4 rows combine, 5 bwd, 6 side inputs, 7 step scalars, 8 fwd plane walk, 9 fwd sliding window
int g_last_handover = 0;   // the last backward asked for the P hand-over
const void* g_last_epoch = nullptr;   // the device epoch word the last attention launch was given
}

namespace {
void inside(const void* p, size_t bytes, const char* what) {
  if (!p) return;
  const unsigned char* b = static_cast<const unsigned char*>(p);
  if (b < g_ws_lo || b + bytes > g_ws_hi) {
    std::fprintf(stderr, "asan driver: %s [%p, +%zu) outside the workspace [%p, %p)\n", what, p, bytes, (const void*)g_ws_lo, (const void*)g_ws_hi);
    std::abort();
  }
}
}  // namespace

namespace mmt {
hipError_t launch_attn_fwd(const FwdParams& p, int, bool, hipStream_t) {
  ++g_launches; g_last_kind = 1; g_last_epoch = p.epoch;
  const size_t slots = (size_t)p.B * p.N * p.n_rowblk * p.n_chunks;
  inside(p.part_o, slots * 32 * 64 * 4, "part_o");
  inside(p.part_ml, slots * 64 * 4, "part_ml");
  return hipSuccess;
}
hipError_t launch_attn_fwd_band_bf16(const FwdParams& p, hipStream_t st) { hipError_t e = launch_attn_fwd(p, 0, true, st); g_last_kind = 2; return e; }
hipError_t launch_attn_fwd_win_bf16(const FwdParams& p, hipStream_t) {
  ++g_launches; g_last_kind = 3; g_last_epoch = p.epoch;
  if (p.rows_parts < 1 || p.rows_parts > 4) { std::fprintf(stderr, "asan driver: rows_parts %d\n", p.rows_parts); std::abort(); }
  if (p.rows_parts > 1) {                     // the row groups' parts live in the caller's workspace, the ticket in its counters
    inside(p.walk_part, (size_t)p.B * p.N * p.n_rowblk * p.rows_parts * 8 * 66 * 4, "rows parts");
    if (!p.sync) { std::fprintf(stderr, "asan driver: split row groups without arrival counters\n"); std::abort(); }
  }
  return hipSuccess;
}
hipError_t launch_attn_fwd_walk_bf16(const FwdParams& p, int grid, hipStream_t) {
  ++g_launches; g_last_kind = 8; g_last_epoch = p.epoch;
  if (p.pat.ng > 0) inside(p.walk_part, (size_t)p.B * p.N * p.walk_maxseg * 8 * 66 * 4, "walk_part");
  const int per_group = grid / p.walk_groups, ppg = p.B * p.N / p.walk_groups;
  if (grid <= 0 || per_group != ppg * p.walk_nseg + p.walk_nhi || p.walk_nseg < 1 || p.walk_maxseg > ((p.S + 31) / 32 + 1) / 2) {
    std::fprintf(stderr, "asan driver: inconsistent plane-walk plan\n"); std::abort();
  }
  return hipSuccess;
}
hipError_t launch_attn_fwd_pwin_bf16(const FwdParams& p, int grid, hipStream_t) {
  ++g_launches; g_last_kind = 9; g_last_epoch = p.epoch;
  const int nqb = (p.S + 127) / 128, total = p.B * p.N * nqb;
  if (grid <= 0 || p.pw_walk < 1 || (long)grid * p.pw_walk < total || (long)(grid - 1) * p.pw_walk >= total) { std::fprintf(stderr, "asan driver: inconsistent sliding-window plan\n"); std::abort(); }
  if (p.pat.ng > 0) inside(p.walk_part, ((size_t)grid * (8 * 34 + 4 * 784) + (size_t)p.B * p.N * p.walk_maxseg * 4 * 8 * 66) * 4, "pwin workspace");
  return hipSuccess;
}
int fwd_pwin_plan(FwdParams& p, int target_wgs) {
  const int nqb = (p.S + 127) / 128, total = p.B * p.N * nqb;
  int walk = (total + target_wgs - 1) / target_wgs; if (walk < 1) walk = 1;
  p.pw_walk = walk; p.walk_maxseg = (nqb + walk - 1) / walk + 1;
  return (total + walk - 1) / walk;
}
size_t fwd_pwin_workspace_bytes(int B, int N, int S, int target_wgs) {
  const int nqb = (S + 127) / 128, total = B * N * nqb;
  const int walk = (total + target_wgs - 1) / target_wgs < 1 ? 1 : (total + target_wgs - 1) / target_wgs;
  const size_t grid = (size_t)(total + walk - 1) / walk, maxseg = (size_t)(nqb + walk - 1) / walk + 1;
  return (grid * (8 * 34 + 4 * 784) + (size_t)B * N * maxseg * 4 * 8 * 66) * sizeof(float);
}
int fwd_walk_lds_bytes(int ng, int tstride, bool rel) { return 32768 + (rel ? 1024 * tstride + 4224 : 0) + (ng ? 3072 + 6272 + 2048 + 32 * tstride : 0) + 16; }
int fwd_walk_plan(FwdParams& p, int target_wgs) {
  const int BN = p.B * p.N, NT = (p.S + 31) / 32, U = (NT + 1) / 2;
  const int ngroups = (BN % 8) == 0 ? 8 : 1, ppg = BN / ngroups;
  int per_group = target_wgs / ngroups;
  if (per_group > ppg * U) per_group = ppg * U;
  if (per_group < ppg) per_group = ppg;
  p.walk_groups = ngroups; p.walk_nseg = per_group / ppg; p.walk_nhi = per_group % ppg;
  p.walk_maxseg = p.walk_nseg + (p.walk_nhi ? 1 : 0);
  return ngroups * per_group;
}
size_t fwd_walk_workspace_bytes(int B, int N, int S) { return (size_t)B * N * (((S + 31) / 32 + 1) / 2) * 8 * 66 * sizeof(float); }
int fwd_win_lds_bytes(int ng, int tstride) { return 65536 + (ng ? (2 * ((ng + 7) / 8) + 1) * 1024 : 0) + 512 * tstride; }
hipError_t launch_rows_combine(const FwdParams&, bool, hipStream_t) { ++g_launches; g_last_kind = 4; return hipSuccess; }
hipError_t launch_attn_bwd(const BwdParams& p, int, bool, hipStream_t) {
  ++g_launches; g_last_kind = 5; g_last_handover = p.ho != nullptr; g_last_epoch = p.epoch;
  const size_t bn = (size_t)p.B * p.N, slots = bn * p.n_gblk * p.n_chunks;
  inside(p.delta, bn * p.S * 4, "delta");
  inside(p.relfar, bn * p.S * 2 * 4, "relfar");
  inside(p.drel, bn * (size_t)p.pat.ng * p.Rp * 4, "drel");
  inside(p.part_dq, slots * 32 * 64 * 4, "part_dq");
  inside(p.part_dtab, slots * 32 * p.Rp * 4, "part_dtab");
  inside(p.part_dkv, bn * p.n_gblk * (size_t)p.dkv_slots * 2 * 32 * 64 * 4, "part_dkv");
  if (p.ho) {          // P hand-over: band tiles, global-key strips, global-row tiles (attn_kernels.h)
    const size_t n_tiles = (size_t)(p.S + 31) / 32;
    inside(p.ho, bn * n_tiles * ((size_t)p.ho_slots * 2048 + 512 + 512), "ho");
    if (p.dkv_slots != p.n_chunks + (p.n_gblk > 0 ? 1 : 0)) { std::fprintf(stderr, "asan driver: hand-over without its partial slot\n"); std::abort(); }
  }
  inside(p.part_red, bn * ((p.S + 127) / 128) * 4 * ((size_t)p.Rp * 64 + p.Rp) * 4, "part_red");
  return hipSuccess;
}
hipError_t launch_side_inputs(const SideParams&, hipStream_t) { ++g_launches; g_last_kind = 6; return hipSuccess; }
hipError_t launch_write_step_scalars(unsigned long long*, float*, unsigned long long, float, float, float, hipStream_t) { ++g_launches; g_last_kind = 7; return hipSuccess; }
}  // namespace mmt
