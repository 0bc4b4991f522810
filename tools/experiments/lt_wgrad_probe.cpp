// Experiment (not part of the product): the library's own weight-gradient GEMM -- dW[M, N] (fp32, += : beta = 1) =
// dY[K, M]^T . X[K, N], bf16 operands, K = 16384 -- on the four shapes of an encoder block, every heuristic algorithm timed.
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/lt_wgrad tools/experiments/lt_wgrad_probe.cpp -lhipblaslt && /tmp/lt_wgrad
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { auto e_ = (x); if (e_ != 0) { std::printf("error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); std::exit(1); } } while (0)
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static uint16_t* rnd(size_t n) {
  std::vector<uint16_t> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = f2bf((float)std::rand() / (float)RAND_MAX * 2.f - 1.f);
  uint16_t* p; CK(hipMalloc(&p, n * 2)); CK(hipMemcpy(p, h.data(), n * 2, hipMemcpyHostToDevice));
  return p;
}
int main() {
  hipblasLtHandle_t h; CK(hipblasLtCreate(&h));
  const int K = 16384;
  const int shapes[4][2] = {{768, 3072}, {3072, 768}, {768, 768}, {2304, 768}};
  void* ws; const uint64_t wsmax = 512u << 20; CK(hipMalloc(&ws, wsmax));
  double total_us = 0, total_flop = 0;
  for (auto& sh : shapes) {
    const int M = sh[0], N = sh[1];
    uint16_t* dy = rnd((size_t)K * M); uint16_t* x = rnd((size_t)K * N);
    float* dw; CK(hipMalloc(&dw, (size_t)M * N * 4)); CK(hipMemset(dw, 0, (size_t)M * N * 4));
    // column-major: D [N x M] (ld N) = A [N x K] (X memory, ld N, op N) . B^T with B [M x K] (dY memory, ld M, op T)
    hipblasLtMatmulDesc_t md; CK(hipblasLtMatmulDescCreate(&md, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    hipblasOperation_t ta = HIPBLAS_OP_N, tb = HIPBLAS_OP_T;
    CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
    CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
    hipblasLtMatrixLayout_t la, lb, lc;
    CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, N, K, N));
    CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, M, K, M));
    CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_32F, N, M, N));
    hipblasLtMatmulPreference_t pref; CK(hipblasLtMatmulPreferenceCreate(&pref));
    CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsmax, sizeof(wsmax)));
    hipblasLtMatmulHeuristicResult_t res[64]; int got = 0;
    auto st = hipblasLtMatmulAlgoGetHeuristic(h, md, la, lb, lc, lc, pref, 64, res, &got);
    if (st != 0 || got == 0) { std::printf("dW[%d x %d]: no algorithm (status %d)\n", M, N, (int)st); continue; }
    const float alpha = 1.f, beta = 1.f;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int i = 0; i < got; ++i) {
      if (res[i].state != 0) continue;
      bool ok = true;
      for (int w = 0; w < 2 && ok; ++w) ok = hipblasLtMatmul(h, md, &alpha, x, la, dy, lb, &beta, dw, lc, dw, lc, &res[i].algo, ws, wsmax, 0) == 0;
      if (!ok) continue;
      CK(hipEventRecord(e0, 0));
      for (int w = 0; w < 10; ++w) hipblasLtMatmul(h, md, &alpha, x, la, dy, lb, &beta, dw, lc, dw, lc, &res[i].algo, ws, wsmax, 0);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms / 10 * 1e3f < best) best = ms / 10 * 1e3f;
    }
    const double flop = 2.0 * M * N * (double)K;
    std::printf("dW[%4d x %4d] += dY^T . X (K = %d, fp32 out, beta = 1): %d algorithms, best %.1f us = %.2f PFLOP/s\n", M, N, K, got, best, flop / (best * 1e-6) / 1e15);
    total_us += best; total_flop += flop;
    CK(hipFree(dy)); CK(hipFree(x)); CK(hipFree(dw));
  }
  std::printf("one block: %.1f us = %.2f PFLOP/s (hand-written grouped launch: ~225-260 us per block)\n", total_us, total_flop / (total_us * 1e-6) / 1e15);
  return 0;
}
