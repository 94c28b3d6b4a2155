// Which workgroups share a CU?  Launches 1024 workgroups of 256 threads with 76 KB of LDS each (2 per CU, like the
// matcher) and records XCC_ID / HW_ID and start/end timestamps of each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
__global__ void __launch_bounds__(256, 2) k(unsigned* ids, long long* t) {
  extern __shared__ unsigned lds[];
  const int wg = blockIdx.y * gridDim.x + blockIdx.x;
  long long t0 = wall_clock64();
  unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);     // HW_REG_HW_ID
  unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);    // HW_REG_XCC_ID
  // burn ~50 us
  unsigned acc = threadIdx.x;
  for (int i = 0; i < 20000; ++i) { acc = acc * 1664525u + 1013904223u; lds[(acc >> 8) & 1023] = acc; }
  if (threadIdx.x == 0) { ids[2 * wg] = hw; ids[2 * wg + 1] = xcc | (acc & 0x80000000u ? 0 : 0); t[2 * wg] = t0; t[2 * wg + 1] = wall_clock64(); }
}
int main() {
  const int gx = 4, gy = 256, n = gx * gy;
  unsigned* d; long long* dt; hipMalloc(&d, n * 8); hipMalloc(&dt, n * 16);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 77 * 1024);
  hipLaunchKernelGGL(k, dim3(gx, gy), dim3(256), 77 * 1024, 0, d, dt);
  hipDeviceSynchronize();
  std::vector<unsigned> h(2 * n); std::vector<long long> ht(2 * n);
  hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), dt, n * 16, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cu;
  long long tmin = ht[0];
  for (int i = 0; i < n; ++i) tmin = std::min(tmin, ht[2 * i]);
  for (int i = 0; i < n; ++i) {
    unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    unsigned cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    cu[(xcc << 16) | (se << 8) | (sh << 4) | cu_id].push_back(i);
  }
  printf("distinct CUs: %zu\n", cu.size());
  int shown = 0;
  for (auto& kv : cu) {
    if (shown++ >= 6) break;
    printf("cu %05x:", kv.first);
    for (int w : kv.second) printf(" wg%d[%.1f-%.1f us]", w, (ht[2 * w] - tmin) / 100.0, (ht[2 * w + 1] - tmin) / 100.0);
    printf("\n");
  }
  return 0;
}
