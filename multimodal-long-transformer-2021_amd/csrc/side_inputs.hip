// Integer side inputs on the device, bit-exact with the reference's tf.data stage
// (src/data/data_utils.py:335-379; src/feature_utils.py:114-184; etcmodel 1-D ids and
// make_segmented_att_mask per SURVEY.md App. A.2).  HBM-write-bound: one int4 store per
// lane per output, rows walked by a grid-stride loop.
#include "attn_kernels.h"

namespace mmt {

// x is one of the n ascending positions of `list` (binary search; n is a few tens)
__device__ __forceinline__ bool in_list(const int32_t* list, int n, int x) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (list[mid] < x) lo = mid + 1; else hi = mid;
  }
  return lo < n && list[lo] == x;
}

__global__ __launch_bounds__(256) void side_inputs_kernel(const SideParams p) {
  const int S = p.S;
  const long row_chunks = (S + 3) >> 2;                 // int4 chunks per row
  const long total = (long)p.B * S * row_chunks;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const long rowid = idx / row_chunks;
    const int k0 = (int)(idx - rowid * row_chunks) * 4;
    const int b = (int)(rowid / S), q = (int)(rowid - (long)b * S);
    const int img = p.img_wp ? p.img_wp[b] : S, txt = p.txt_wp ? p.txt_wp[b] : 0;
    const int valid = img + txt;
    int mv[4], iv[4];
    const bool gq = p.gidx && p.materialize_pattern && in_list(p.gidx, p.pat.ng, q);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = k0 + j;
      bool keep = (q < valid) == (k < valid);            // data_utils.py:321-322
      if (p.materialize_pattern) {
        if (p.gidx) keep = keep && (abs(q - k) <= p.pat.radius || gq || in_list(p.gidx, p.pat.ng, k));   // listed global set
        else keep = pattern_mask(p.pat, valid, q, k);
      }
      mv[j] = keep ? 1 : 0;
      iv[j] = p.pat.id_mode ? rel_id(p.pat, q, k) : 0;
    }
    const long off = rowid * S + k0;
    if ((S & 3) == 0) {
      if (p.att_mask) *reinterpret_cast<i32x4*>(p.att_mask + off) = i32x4{mv[0], mv[1], mv[2], mv[3]};
      if (p.rel_ids) *reinterpret_cast<i32x4*>(p.rel_ids + off) = i32x4{iv[0], iv[1], iv[2], iv[3]};
    } else {
      for (int j = 0; j < 4 && k0 + j < S; ++j) {
        if (p.att_mask) p.att_mask[off + j] = mv[j];
        if (p.rel_ids) p.rel_ids[off + j] = iv[j];
      }
    }
    if (p.segment_ids && k0 == 0) {                      // data_utils.py:350-361
      p.segment_ids[rowid] = (q < img ? 1 : 0) + ((q > img && q < img + txt) ? 2 : 0);
    }
  }
}

hipError_t launch_side_inputs(const SideParams& p, hipStream_t st) {
  const long total = (long)p.B * p.S * ((p.S + 3) / 4);
  long blocks = (total + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(side_inputs_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
  return hipGetLastError();
}

__global__ void write_step_scalars_kernel(unsigned long long* e, float* h, unsigned long long epoch, float lr, float bc1, float bc2) {
  if (threadIdx.x == 0) {
    if (e) *e = epoch;
    if (h) { h[0] = lr; h[1] = bc1; h[2] = bc2; }
  }
}
hipError_t launch_write_step_scalars(unsigned long long* epoch_dst, float* hyper_dst, unsigned long long epoch, float lr,
                                     float bc1, float bc2, hipStream_t st) {
  hipLaunchKernelGGL(write_step_scalars_kernel, dim3(1), dim3(64), 0, st, epoch_dst, hyper_dst, epoch, lr, bc1, bc2);
  return hipGetLastError();
}

}  // namespace mmt
