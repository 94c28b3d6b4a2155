// A chain of N small dependent kernels: stream launches vs a captured hipGraph replayed (is the MGM front loop — 2047 launches of ~5.5 us — worth a graph?).
// build: hipcc --offload-arch=gfx950 -O2 tools/ubench_graph.hip -o tools/build/ubench_graph
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k(unsigned* p, int f, int iters) {
  unsigned v = p[blockIdx.x * blockDim.x + threadIdx.x] + f;
  for (int i = 0; i < iters; ++i) v = v * 1664525u + 1013904223u;
  p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
int main() {
  const int N = 2000;
  unsigned* d; hipMalloc(&d, 4096 * 256 * 4); hipMemset(d, 0, 4096 * 256 * 4);
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  for (int blocks : {64, 1024, 4096}) {
    for (int iters : {10, 400}) {
      auto run_stream = [&]() { for (int f = 0; f < N; ++f) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, st, d, f, iters); };
      run_stream(); hipStreamSynchronize(st);
      auto t0 = std::chrono::steady_clock::now();
      run_stream(); hipStreamSynchronize(st);
      const double us_stream = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
      hipGraph_t g; hipGraphExec_t ge;
      hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
      run_stream();
      hipStreamEndCapture(st, &g);
      auto tc = std::chrono::steady_clock::now();
      hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
      const double ms_inst = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc).count();
      hipGraphLaunch(ge, st); hipStreamSynchronize(st);
      t0 = std::chrono::steady_clock::now();
      hipGraphLaunch(ge, st); hipStreamSynchronize(st);
      const double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
      printf("%5d workgroups x %3d iterations: stream %.2f us per kernel, graph %.2f us per kernel (instantiate %.1f ms)\n", blocks, iters, us_stream, us_graph, ms_inst);
      hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
  }
  return 0;
}
