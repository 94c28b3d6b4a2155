// ubench_streams.hip — do short kernels queued by different host threads on different streams overlap on the device?
// T threads, each with its own non-blocking stream, each queue M kernels of G workgroups that spin for D microseconds; the wall time of
// the whole batch against M * D tells how many chains the GPU advanced at once.  Variants: a stream synchronisation every S kernels (the
// tile loop's host round trips), a 256-byte pinned H2D copy before every kernel (its table uploads).
// Build: hipcc --offload-arch=gfx950 -O3 -pthread tools/ubench_streams.hip -o tools/build/ubench_streams
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void spin(long long ticks, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = 1;
}

static double run(int T, int M, int G, double us, int sync_every, int copies) {
  std::vector<hipStream_t> st(T);
  std::vector<void*> hbuf(T), dbuf(T);
  for (int i = 0; i < T; ++i) {
    (void)hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
    (void)hipHostMalloc(&hbuf[i], 4096, hipHostMallocDefault);
    (void)hipMalloc(&dbuf[i], 4096);
  }
  auto body = [&](int i, int m) {
    for (int k = 0; k < m; ++k) {
      if (copies) (void)hipMemcpyAsync(dbuf[i], hbuf[i], 256, hipMemcpyHostToDevice, st[i]);
      hipLaunchKernelGGL(spin, dim3(G), dim3(256), 0, st[i], (long long)(us * 100.0), (int*)nullptr);
      if (sync_every && (k + 1) % sync_every == 0) (void)hipStreamSynchronize(st[i]);
    }
    (void)hipStreamSynchronize(st[i]);
  };
  { std::vector<std::thread> th; for (int i = 0; i < T; ++i) th.emplace_back(body, i, 8); for (auto& t : th) t.join(); }     // warm up
  const auto t0 = std::chrono::steady_clock::now();
  { std::vector<std::thread> th; for (int i = 0; i < T; ++i) th.emplace_back(body, i, M); for (auto& t : th) t.join(); }
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (int i = 0; i < T; ++i) { (void)hipStreamDestroy(st[i]); (void)hipHostFree(hbuf[i]); (void)hipFree(dbuf[i]); }
  return wall * 1e6;
}

int main() {
  const int M = 400;
  for (int G : {1, 64, 1024})
    for (double us : {5.0, 20.0})
      for (int sync_every : {0, 8, 1})
        for (int copies : {0, 1}) {
          printf("G %4d workgroups, %4.0f us kernels, sync every %d, copies %d:", G, us, sync_every, copies);
          for (int T : {1, 2, 4, 8}) {
            const double w = run(T, M, G, us, sync_every, copies);
            printf("  T=%d %.1f us/kernel/thread (%.2f chains at once)", T, w / M, T * M * us / w);
          }
          printf("\n");
        }
  return 0;
}
