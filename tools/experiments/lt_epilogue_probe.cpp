// Experiment (not part of the product): hipBLASLt's GELU_AUX_BIAS / DGELU epilogues on the feed-forward shapes of config 3,
// every heuristic algorithm timed with HIP events, a few hundred elements checked against a host computation.
//   hipcc -O2 --offload-arch=gfx950 -o /tmp/lt_probe tools/experiments/lt_epilogue_probe.cpp -lhipblaslt && /tmp/lt_probe
#include <hip/hip_runtime.h>
#include <hip/hip_bfloat16.h>
#include <hipblaslt/hipblaslt.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { auto e_ = (x); if (e_ != 0) { std::printf("error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); std::exit(1); } } while (0)

static float bf2f(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float gelu_tanh(float z) { return 0.5f * z * (1.f + std::tanh(0.7978845608f * (z + 0.044715f * z * z * z))); }
static float dgelu_tanh(float z) {
  const float k = 0.7978845608f, c = 0.044715f, t = std::tanh(k * (z + c * z * z * z));
  return 0.5f * (1.f + t) + 0.5f * z * (1.f - t * t) * k * (1.f + 3.f * c * z * z);
}

struct Dev { uint16_t* p; std::vector<uint16_t> h; };
static Dev rnd(size_t n, float scale) {
  Dev d; d.h.resize(n);
  for (size_t i = 0; i < n; ++i) d.h[i] = f2bf(scale * ((float)std::rand() / RAND_MAX * 2.f - 1.f));
  CK(hipMalloc(&d.p, n * 2)); CK(hipMemcpy(d.p, d.h.data(), n * 2, hipMemcpyHostToDevice));
  return d;
}

// D (m x n, column-major, ld = m) = op(A) . op(B) with the given epilogue; returns the best time in us over the heuristic's algorithms
static float run(hipblasLtHandle_t h, int m, int n, int k, hipblasOperation_t ta, const void* A, int lda, const void* B, int ldb, void* D,
                 hipblasLtEpilogue_t epi, const void* bias, void* aux, const char* what) {
  hipblasLtMatmulDesc_t md; CK(hipblasLtMatmulDescCreate(&md, HIPBLAS_COMPUTE_32F, HIP_R_32F));
  hipblasOperation_t tb = HIPBLAS_OP_N;
  CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
  CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
  CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)));
  if (bias) {
    hipDataType bt = HIP_R_32F;
    CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
    CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)));
  }
  if (aux) {
    int64_t ld = m; hipDataType at = HIP_R_16BF;
    CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_EPILOGUE_AUX_POINTER, &aux, sizeof(aux)));
    CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_EPILOGUE_AUX_LD, &ld, sizeof(ld)));
    CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_EPILOGUE_AUX_DATA_TYPE, &at, sizeof(at)));
  }
  hipblasLtMatrixLayout_t la, lb, ld_;
  if (ta == HIPBLAS_OP_T) CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, k, m, lda)); else CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, m, k, lda));
  CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, k, n, ldb));
  CK(hipblasLtMatrixLayoutCreate(&ld_, HIP_R_16BF, m, n, m));
  hipblasLtMatmulPreference_t pref; CK(hipblasLtMatmulPreferenceCreate(&pref));
  uint64_t wsmax = 256u << 20;
  CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsmax, sizeof(wsmax)));
  hipblasLtMatmulHeuristicResult_t res[32]; int got = 0;
  auto st = hipblasLtMatmulAlgoGetHeuristic(h, md, la, lb, ld_, ld_, pref, 32, res, &got);
  if (st != 0 || got == 0) { std::printf("%-28s no algorithm (status %d)\n", what, (int)st); return -1.f; }
  void* ws; CK(hipMalloc(&ws, wsmax));
  const float alpha = 1.f, beta = 0.f;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f; int besti = -1;
  for (int i = 0; i < got; ++i) {
    if (res[i].state != 0) continue;
    bool ok = true;
    for (int w = 0; w < 3 && ok; ++w) ok = hipblasLtMatmul(h, md, &alpha, A, la, B, lb, &beta, D, ld_, D, ld_, &res[i].algo, ws, wsmax, 0) == 0;
    if (!ok) continue;
    CK(hipEventRecord(e0, 0));
    for (int w = 0; w < 20; ++w) hipblasLtMatmul(h, md, &alpha, A, la, B, lb, &beta, D, ld_, D, ld_, &res[i].algo, ws, wsmax, 0);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms / 20 * 1e3f < best) { best = ms / 20 * 1e3f; besti = i; }
  }
  if (besti >= 0) hipblasLtMatmul(h, md, &alpha, A, la, B, lb, &beta, D, ld_, D, ld_, &res[besti].algo, ws, wsmax, 0);
  CK(hipDeviceSynchronize());
  std::printf("%-28s %d algorithms, best %.1f us = %.2f PFLOP/s\n", what, got, best, 2.0 * m * n * (double)k / (best * 1e-6) / 1e15);
  CK(hipFree(ws));
  return best;
}

int main() {
  hipblasLtHandle_t h; CK(hipblasLtCreate(&h));
  const int M = 16384, H = 768, I = 3072;
  // forward: u[M, I] = x[M, H] . W1^T + b1 (W1 [I, H] row-major); g = gelu(u).  Column-major: D^T [I x M] = W1(col-major [H x I])^T . x(col-major [H x M])
  Dev x = rnd((size_t)M * H, 1.f), w1 = rnd((size_t)I * H, 0.05f), dy = rnd((size_t)M * H, 1.f), w2 = rnd((size_t)H * I, 0.05f);
  std::vector<float> b1(I); for (auto& v : b1) v = 0.1f * ((float)std::rand() / RAND_MAX - 0.5f);
  float* b1d; CK(hipMalloc(&b1d, I * 4)); CK(hipMemcpy(b1d, b1.data(), I * 4, hipMemcpyHostToDevice));
  uint16_t *g, *u, *du; CK(hipMalloc(&g, (size_t)M * I * 2)); CK(hipMalloc(&u, (size_t)M * I * 2)); CK(hipMalloc(&du, (size_t)M * I * 2));
  run(h, I, M, H, HIPBLAS_OP_T, w1.p, H, x.p, H, g, HIPBLASLT_EPILOGUE_DEFAULT, nullptr, nullptr, "FFN1 plain");
  run(h, I, M, H, HIPBLAS_OP_T, w1.p, H, x.p, H, g, HIPBLASLT_EPILOGUE_BIAS, b1d, nullptr, "FFN1 + bias");
  run(h, I, M, H, HIPBLAS_OP_T, w1.p, H, x.p, H, g, HIPBLASLT_EPILOGUE_GELU_BIAS, b1d, nullptr, "FFN1 + bias + GELU");
  float tf = run(h, I, M, H, HIPBLAS_OP_T, w1.p, H, x.p, H, g, HIPBLASLT_EPILOGUE_GELU_AUX_BIAS, b1d, u, "FFN1 + bias + GELU + aux");
  if (tf > 0) {
    std::vector<uint16_t> gh(1000), uh(1000);
    double eg = 0, eu = 0;
    for (int s = 0; s < 200; ++s) {
      const int r = std::rand() % M, c = std::rand() % I;
      double acc = b1[c];
      for (int kk = 0; kk < H; ++kk) acc += (double)bf2f(x.h[(size_t)r * H + kk]) * bf2f(w1.h[(size_t)c * H + kk]);
      uint16_t gv, uv; CK(hipMemcpy(&gv, g + (size_t)r * I + c, 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(&uv, u + (size_t)r * I + c, 2, hipMemcpyDeviceToHost));
      eg = std::fmax(eg, std::fabs(bf2f(gv) - gelu_tanh((float)acc))); eu = std::fmax(eu, std::fabs(bf2f(uv) - acc));
    }
    std::printf("   forward check: max |g - gelu_tanh(ref)| = %.4f   max |aux - (x.W1^T + b1)| = %.4f\n", eg, eu);
  }
  // backward: du[M, I] = (dy[M, H] . W2[H, I]) * gelu'(u).  Column-major: D^T [I x M] = W2(col-major [I x H]) . dy(col-major [H x M])
  run(h, I, M, H, HIPBLAS_OP_N, w2.p, I, dy.p, H, du, HIPBLASLT_EPILOGUE_DEFAULT, nullptr, nullptr, "FFN2 dgrad plain");
  float tb = run(h, I, M, H, HIPBLAS_OP_N, w2.p, I, dy.p, H, du, HIPBLASLT_EPILOGUE_DGELU, nullptr, u, "FFN2 dgrad + DGELU(aux)");
  if (tb > 0 && tf > 0) {
    double ed = 0, mx = 0;
    for (int s = 0; s < 200; ++s) {
      const int r = std::rand() % M, c = std::rand() % I;
      double acc = 0;
      for (int kk = 0; kk < H; ++kk) acc += (double)bf2f(dy.h[(size_t)r * H + kk]) * bf2f(w2.h[(size_t)kk * I + c]);
      uint16_t dv, uv; CK(hipMemcpy(&dv, du + (size_t)r * I + c, 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(&uv, u + (size_t)r * I + c, 2, hipMemcpyDeviceToHost));
      const double want = acc * dgelu_tanh(bf2f(uv));
      ed = std::fmax(ed, std::fabs(bf2f(dv) - want)); mx = std::fmax(mx, std::fabs(want));
    }
    std::printf("   backward check: max |du - dgrad * gelu_tanh'(aux)| = %.4f (max |du| %.3f)\n", ed, mx);
  }
  return 0;
}
