// Softmax cross-entropy over wide rows (the 30522-way MLM logits): the per-row
// loss -log softmax(logits)[label] of `weighted_sparse_categorical_crossentropy_loss`
// (src/modeling/losses/weighted_sparse_categorical_crossentropy_loss.py:17-43) and its gradient, each
// in ONE pass over the logits in their storage dtype -- TF / torch run cast, max, exp-sum, log, gather
// and the backward's softmax as separate fp32 passes.  One workgroup per row, online (max, sum) per
// thread, one LDS combine.  C ABI: include/mmt_layer.h (mmt_xent_fwd / mmt_xent_bwd).
#include "../../include/mmt_attn.h"
#include "../../include/mmt_layer.h"

#include "layer_common.h"
#include "mmt_err.h"

namespace mmt {

template <typename T> __device__ __forceinline__ float ldval(const T* p, long i) { return (float)p[i]; }

struct OnlineLse {
  float m = -INFINITY, s = 0.f;
  int am = 0x7fffffff;          // index of the FIRST element equal to m (tf.argmax / SparseCategoricalAccuracy's tie rule)
  __device__ __forceinline__ void add(float x, int i) {
    if (x > m) { s = s * __expf(m - x) + 1.f; m = x; am = i; }       // a thread meets its indices in ascending order
    else s += __expf(x - m);
  }
  __device__ __forceinline__ void merge(float m2, float s2, int am2) {
    const float mm = fmaxf(m, m2);
    if (mm == -INFINITY) return;
    am = m2 > m ? am2 : (m2 == m ? min(am, am2) : am);
    s = s * __expf(m - mm) + s2 * __expf(m2 - mm);
    m = mm;
  }
};

// rows x C logits (row stride ld elements) -> loss[row] = lse - logits[label], lse[row] (natural log)
template <typename T>
__global__ __launch_bounds__(256) void xent_fwd_kernel(const T* logits, long ld, int C, const int* labels,
                                                       float* loss, float* lse_out, int* amax_out) {
  __shared__ float rm[4], rs[4];
  __shared__ int ra[4];
  const long row = blockIdx.x;
  const T* x = logits + row * ld;
  OnlineLse acc;
  const bool pairs = sizeof(T) == 2 && ((ld & 1) == 0) && ((reinterpret_cast<uintptr_t>(logits) & 3) == 0);
  if (pairs) {                                    // two bf16 per 4-byte load
    const int np = C >> 1;
    const uint32_t* xp = reinterpret_cast<const uint32_t*>(x);
    for (int i = threadIdx.x; i < np; i += 256) {
      const uint32_t w = xp[i];
      acc.add(__uint_as_float(w << 16), 2 * i);
      acc.add(__uint_as_float(w & 0xFFFF0000u), 2 * i + 1);
    }
    if (C & 1) {                      // the odd last element: folded in by thread 0 through a merge (its own indices
      OnlineLse tail;                 // would no longer be ascending)
      if (threadIdx.x == 0) { tail.add(ldval(x, C - 1), C - 1); acc.merge(tail.m, tail.s, tail.am); }
    }
  } else {
    for (int i = threadIdx.x; i < C; i += 256) acc.add(ldval(x, i), i);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc.merge(__shfl_xor(acc.m, o, 64), __shfl_xor(acc.s, o, 64), __shfl_xor(acc.am, o, 64));
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { rm[wave] = acc.m; rs[wave] = acc.s; ra[wave] = acc.am; }
  __syncthreads();
  if (threadIdx.x == 0) {
    OnlineLse t;
    for (int w = 0; w < 4; ++w) t.merge(rm[w], rs[w], ra[w]);
    if (amax_out) amax_out[row] = t.am;
    const float lse = t.m + logf(t.s);
    const int lab = labels[row];
    lse_out[row] = lse;
    loss[row] = ((unsigned)lab < (unsigned)C) ? lse - ldval(x, lab) : 0.f;   // label outside [0, C): no target
  }
}

// dlogits[row][i] = (exp(logits - lse) - [i == label]) * coef[row] (* gscale[0])
template <typename T>
__global__ __launch_bounds__(256) void xent_bwd_kernel(const T* logits, long ld, int C, const int* labels,
                                                       const float* lse, const float* coef, const float* gscale,
                                                       T* dlogits, long ldd) {
  const long row = blockIdx.y;
  const T* x = logits + row * ld;
  T* dx = dlogits + row * ldd;
  const int lab = labels[row];
  const bool has = (unsigned)lab < (unsigned)C;
  const float l = lse[row], c = has ? coef[row] * (gscale ? gscale[0] : 1.f) : 0.f;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < C; i += gridDim.x * 256) {
    const float p = __expf(ldval(x, i) - l);
    dx[i] = (T)((p - (i == lab ? 1.f : 0.f)) * c);
  }
}

// Weighted mean of per-row losses with divide_no_nan, and its derivative per row -- one workgroup, fixed-order
// sums (bitwise reproducible).  w_i = weight[i] * (mask ? mask[i / mask_div] : 1);  l_i = loss[i] * (lmul ? lmul[i] : 1)
//   num = sum w_i l_i,  den = sum w_i,  out = {den != 0 ? num / den : 0, num, den},
//   coef[i] = den != 0 ? w_i (lmul_i) / den : 0
__global__ __launch_bounds__(256) void weighted_loss_kernel(long rows, const float* loss, const float* weight,
                                                            const float* lmul, const float* mask, long mask_div,
                                                            float* out3, float* coef) {
  __shared__ float rn[256], rd[256];
  const int t = threadIdx.x;
  float num = 0.f, den = 0.f;
  for (long i = t; i < rows; i += 256) {
    const float w = weight[i] * (mask ? mask[i / mask_div] : 1.f);
    num += w * loss[i] * (lmul ? lmul[i] : 1.f);
    den += w;
  }
  rn[t] = num; rd[t] = den;
  __syncthreads();
#pragma unroll
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) { rn[t] += rn[t + o]; rd[t] += rd[t + o]; }
    __syncthreads();
  }
  num = rn[0]; den = rd[0];
  const float inv = den != 0.f ? 1.f / den : 0.f;
  if (t == 0) { out3[0] = den != 0.f ? num / den : 0.f; out3[1] = num; out3[2] = den; }
  if (coef)
    for (long i = t; i < rows; i += 256)
      coef[i] = weight[i] * (mask ? mask[i / mask_div] : 1.f) * (lmul ? lmul[i] : 1.f) * inv;
}

}  // namespace mmt

extern "C" {

int mmt_weighted_loss(int64_t rows, const float* loss, const float* weight, const float* lmul, const float* mask,
                      int64_t mask_div, float* out3, float* coef, void* stream) {
  if (!loss || !weight || !out3) return mmt::fail(MMT_E_INVALID, "mmt_weighted_loss: NULL argument");
  if (rows <= 0 || (mask && mask_div <= 0)) return mmt::fail(MMT_E_INVALID, "mmt_weighted_loss: bad shape");
  hipLaunchKernelGGL(mmt::weighted_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (long)rows, loss, weight, lmul, mask,
                     (long)(mask ? mask_div : 1), out3, coef);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MMT_OK : mmt::fail(MMT_E_LAUNCH, "mmt_weighted_loss: %s", hipGetErrorString(e));
}


int mmt_xent_fwd(int64_t rows, int32_t C, int32_t dtype, const void* logits, int64_t ld, const int32_t* labels,
                 float* loss, float* lse, void* stream) {
  return mmt_xent_fwd_argmax(rows, C, dtype, logits, ld, labels, loss, lse, nullptr, stream);
}

int mmt_xent_fwd_argmax(int64_t rows, int32_t C, int32_t dtype, const void* logits, int64_t ld, const int32_t* labels,
                        float* loss, float* lse, int32_t* argmax, void* stream) {
  if (!logits || !labels || !loss || !lse) return mmt::fail(MMT_E_INVALID, "mmt_xent_fwd: NULL argument");
  if (rows <= 0 || C <= 0 || ld < C) return mmt::fail(MMT_E_INVALID, "mmt_xent_fwd: bad shape");
  if (dtype != MMT_F32 && dtype != MMT_BF16) return mmt::fail(MMT_E_INVALID, "mmt_xent_fwd: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MMT_BF16) hipLaunchKernelGGL(mmt::xent_fwd_kernel<__bf16>, dim3((unsigned)rows), dim3(256), 0, st, (const __bf16*)logits, (long)ld, C, labels, loss, lse, argmax);
  else hipLaunchKernelGGL(mmt::xent_fwd_kernel<float>, dim3((unsigned)rows), dim3(256), 0, st, (const float*)logits, (long)ld, C, labels, loss, lse, argmax);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MMT_OK : mmt::fail(MMT_E_LAUNCH, "mmt_xent_fwd: %s", hipGetErrorString(e));
}

int mmt_xent_bwd(int64_t rows, int32_t C, int32_t dtype, const void* logits, int64_t ld, const int32_t* labels,
                 const float* lse, const float* coef, void* dlogits, int64_t ldd, void* stream) {
  return mmt_xent_bwd_scaled(rows, C, dtype, logits, ld, labels, lse, coef, nullptr, dlogits, ldd, stream);
}

int mmt_xent_bwd_scaled(int64_t rows, int32_t C, int32_t dtype, const void* logits, int64_t ld, const int32_t* labels,
                        const float* lse, const float* coef, const float* gscale, void* dlogits, int64_t ldd, void* stream) {
  if (!logits || !labels || !lse || !coef || !dlogits) return mmt::fail(MMT_E_INVALID, "mmt_xent_bwd: NULL argument");
  if (rows <= 0 || rows > 65535 || C <= 0 || ld < C || ldd < C) return mmt::fail(MMT_E_INVALID, "mmt_xent_bwd: bad shape");
  if (dtype != MMT_F32 && dtype != MMT_BF16) return mmt::fail(MMT_E_INVALID, "mmt_xent_bwd: bad dtype %d", dtype);
  hipStream_t st = (hipStream_t)stream;
  int bx = (C + 256 * 8 - 1) / (256 * 8);
  bx = bx < 1 ? 1 : (bx > 64 ? 64 : bx);
  dim3 grid(bx, (unsigned)rows);
  if (dtype == MMT_BF16) hipLaunchKernelGGL(mmt::xent_bwd_kernel<__bf16>, grid, dim3(256), 0, st, (const __bf16*)logits, (long)ld, C, labels, lse, coef, gscale, (__bf16*)dlogits, (long)ldd);
  else hipLaunchKernelGGL(mmt::xent_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)logits, (long)ld, C, labels, lse, coef, gscale, (float*)dlogits, (long)ldd);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MMT_OK : mmt::fail(MMT_E_LAUNCH, "mmt_xent_bwd: %s", hipGetErrorString(e));
}

}  // extern "C"
