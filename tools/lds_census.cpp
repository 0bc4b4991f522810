// Dev tool: how many 256-thread workgroups with a given dynamic-LDS size (and a 168-VGPR-like register budget) are
// resident per CU on this chip?  768 = 3 x 256 blocks spin ~40 us each; the launch takes ~40 us if they are all
// co-resident and ~80 us if only two fit per CU.   hipcc --offload-arch=gfx950 -O3 tools/lds_census.cpp -o /tmp/census
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256, 3) void spin(long long ticks, int* sink) {
  extern __shared__ char smem[];
  smem[threadIdx.x] = 1;
  const long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < ticks) {}
  if (smem[(threadIdx.x + 1) & 255] == 7) *sink = 1;
}
int main() {
  int* sink; hipMalloc(&sink, 4);
  const int sizes[] = {50176, 50688, 51200, 52224, 53248, 53760, 54272, 54784, 55296, 65536};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int lds : sizes) {
    if (lds > 65536) continue;
    hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(spin, dim3(768), dim3(256), lds, 0, 100000LL, sink);   // s_memtime ticks are shader cycles: ~45 us
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("lds %6d B/WG: %.1f us for 768 blocks\n", lds, ms * 1e3);
    }
  }
  return 0;
}
