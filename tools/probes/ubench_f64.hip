// ubench_f64.hip — issue rate and dependent-chain latency of the float64 / conversion instructions the zone matcher (bm_zones.hip) is built
// from.  Throughput: 16 independent instances per lane, 8 wavefronts per SIMD.  Latency: one dependent chain, ONE wavefront per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_f64.hip -o tools/build/ubench_f64
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr int ITER = 2048, UNROLL = 16;

template <int OP, int U>
__global__ void __launch_bounds__(256) bench(double* out, double seed) {
  double q[U]; float f[U]; typedef float f2 __attribute__((ext_vector_type(2))); f2 p[U];
  const double w = seed + 1e-9 * threadIdx.x, w2 = 1.0 - 1e-7 * seed;
  const float fw = (float)w; int idx[U];
#pragma unroll
  for (int i = 0; i < U; ++i) { q[i] = w + i; f[i] = fw + i; p[i] = f2{fw, fw + i}; idx[i] = i; }
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < U; ++i) {
      if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(q[i]) : "v"(w));
      if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(q[i]) : "v"(w2));
      if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(q[i]) : "v"(w2), "v"(w));
      if (OP == 3) asm volatile("v_max_f64 %0, %0, %1" : "+v"(q[i]) : "v"(w));
      if (OP == 4) asm volatile("v_min_f64 %0, %0, %1" : "+v"(q[i]) : "v"(w));
      if (OP == 5) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(q[i]) : "v"(f[i])); }
      if (OP == 6) { asm volatile("v_cvt_f64_f32 %0, %1\n v_cvt_f32_f64 %1, %0" : "+v"(q[i]), "+v"(f[i])); }
      if (OP == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[0]));
      if (OP == 8) asm volatile("v_cmp_gt_f64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(idx[i]) : "v"(q[i]), "v"(w), "v"(it) : "vcc");
      if (OP == 9) asm volatile("v_rsq_f64 %0, %0" : "+v"(q[i]));
      if (OP == 10) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(fw));
      if (OP == 11) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(q[i]));
      if (OP == 12) asm volatile("v_rcp_f64 %0, %0" : "+v"(q[i]));
      if (OP == 13) asm volatile("v_sqrt_f64 %0, %0" : "+v"(q[i]));
    }
  }
  double r = 0;
#pragma unroll
  for (int i = 0; i < U; ++i) r += q[i] + f[i] + p[i].x + p[i].y + idx[i];
  if (r == 12345.678) out[threadIdx.x] = r;
}

template <int OP> void run(const char* name, double* d, int per_iter = 1) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  {
    const int blocks = 256 * 8;
    hipLaunchKernelGGL((bench<OP, UNROLL>), dim3(blocks), dim3(256), 0, 0, d, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((bench<OP, UNROLL>), dim3(blocks), dim3(256), 0, 0, d, 2.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * ITER * UNROLL * per_iter;         // wave instructions
    printf("%-28s throughput: %6.2f clk per wavefront instruction per SIMD @2.4GHz", name, ms * 1e-3 * 2.4e9 * 1024 / winstr);
  }
  {
    const int blocks = 256;                                                       // one workgroup of 4 wavefronts per CU: one per SIMD
    hipLaunchKernelGGL((bench<OP, 1>), dim3(blocks), dim3(256), 0, 0, d, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((bench<OP, 1>), dim3(blocks), dim3(256), 0, 0, d, 2.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf(" | dependent chain: %6.2f clk per instruction\n", ms * 1e-3 * 2.4e9 / ((double)ITER * per_iter));
  }
}

int main() {
  double* d; hipMalloc(&d, 4096);
  run<0>("v_add_f64", d); run<1>("v_mul_f64", d); run<2>("v_fma_f64", d); run<3>("v_max_f64", d); run<4>("v_min_f64", d);
  run<5>("v_cvt_f64_f32", d); run<6>("v_cvt_f64_f32 + v_cvt_f32_f64", d, 2); run<11>("v_cvt_f32_f64", d);
  run<7>("v_pk_mul_f32", d); run<10>("v_mul_f32", d); run<8>("v_cmp_gt_f64 + v_cndmask", d, 2);
  run<9>("v_rsq_f64", d); run<12>("v_rcp_f64", d); run<13>("v_sqrt_f64", d);
  return 0;
}
