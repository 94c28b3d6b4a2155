// ubench_valu.hip — issue-rate microbenchmark of the VALU instructions the packed-u8 matcher is built from.
// Each kernel runs N_ITER x UNROLL independent instances of one instruction per lane, on 8 waves/SIMD-equivalent
// occupancy (256 CUs x 8 blocks x 256 threads), and reports lane-ops/clk/CU (peak 128 if the op issues
// 32 lanes/clk/SIMD).  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o ubench_valu
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef uint32_t u32; typedef uint64_t u64;
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

constexpr int ITER = 2048, UNROLL = 16;

template <int OP>
__global__ void __launch_bounds__(256) bench(u32* out, u32 seed) {
  u32 a[UNROLL]; u64 q[UNROLL];
  u32 b = seed * 2654435761u + threadIdx.x, c = b ^ 0x9e3779b9u;
  u64 w = ((u64)b << 32) | c;
#pragma unroll
  for (int i = 0; i < UNROLL; ++i) { a[i] = b + i * 77; q[i] = a[i]; }
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
      if (OP == 0) asm volatile("v_qsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(q[i]) : "v"(w), "v"(a[i]));
      if (OP == 1) asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 2) asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 3) asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 4) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 5) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 6) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 7) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 8) asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %0" : "+v"(q[i]) : "v"(w), "v"(a[i]));
      if (OP == 9) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 10) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(a[i]) : "v"(c));
      if (OP == 11) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 12) asm volatile("v_pk_minimum3_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 13) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 14) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 15) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 16) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 17) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 18) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 19) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(q[i]) : "v"(w));
      if (OP == 21) asm volatile("v_sub_u16_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 22) asm volatile("v_sub_u16_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 23) asm volatile("v_sub_u16 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 30) asm volatile("v_min_u32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 31) asm volatile("v_max_u32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 32) asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 33) asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 34) asm volatile("v_lshlrev_b32_e32 %0, 1, %0" : "+v"(a[i]));
      if (OP == 35) asm volatile("v_sub_u32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 36) asm volatile("v_add_u16_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 37) asm volatile("v_min_u16_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 38) asm volatile("v_max_u16_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 39) asm volatile("v_mov_b32_e32 %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 40) asm volatile("v_add_u32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 41) asm volatile("v_sub_u16_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 42) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 43) asm volatile("v_mul_f32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 44) asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 45) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 46) asm volatile("v_min_u32_e64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
      if (OP == 47) asm volatile("v_sub_u16_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1" : "+v"(a[i]) : "v"(c));
      if (OP == 20) asm volatile("v_add_f64 %0, %0, %1" : "+v"(q[i]) : "v"(w));
    }
  }
  u32 r = 0;
#pragma unroll
  for (int i = 0; i < UNROLL; ++i) r ^= a[i] ^ (u32)q[i] ^ (u32)(q[i] >> 32);
  if (r == 0x12345678u) out[threadIdx.x] = r;
}

template <int OP> void run(const char* name, u32* d) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8;
  hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double ops = (double)blocks * 256 * ITER * UNROLL;
  double per_clk_cu = ops / (ms * 1e-3) / 256 / 2.4e9;
  printf("%-22s %8.3f ms  %7.1f lane-ops/clk/CU @2.4GHz (128 = 32 lanes/clk/SIMD)\n", name, ms, per_clk_cu);
}

int main() {
  u32* d; hipMalloc(&d, 4096);
  run<6>("v_add_u32", d); run<9>("v_fma_f32", d); run<19>("v_pk_fma_f32", d); run<20>("v_add_f64", d);
  run<0>("v_qsad_pk_u16_u8", d); run<8>("v_mqsad_pk_u16_u8", d); run<7>("v_sad_u8", d); run<18>("v_dot4_u32_u8", d);
  run<1>("v_pk_sub_u16", d); run<13>("v_pk_add_u16", d); run<2>("v_pk_max_u16", d); run<11>("v_pk_min_u16", d);
  run<12>("v_pk_minimum3_f16", d); run<16>("v_pk_mad_u16", d);
  run<3>("v_min3_u32", d); run<17>("v_min_u32", d); run<4>("v_lshl_or_b32", d); run<5>("v_bfi_b32", d);
  run<21>("v_sub_u16_sdwa W0->W1", d); run<22>("v_sub_u16_sdwa W1->W1", d); run<23>("v_sub_u16", d);
  run<45>("v_add_u32_e32", d); run<40>("v_add_u32_e64", d); run<35>("v_sub_u32_e32", d);
  run<30>("v_min_u32_e32", d); run<46>("v_min_u32_e64", d); run<31>("v_max_u32_e32", d); run<32>("v_and_b32_e32", d); run<33>("v_xor_b32_e32", d);
  run<34>("v_lshlrev_b32_e32", d); run<39>("v_mov_b32_e32", d);
  run<36>("v_add_u16_e32", d); run<41>("v_sub_u16_e32(dep)", d); run<37>("v_min_u16_e32", d); run<38>("v_max_u16_e32", d);
  run<47>("v_sub_u16_sdwa(dword dst)", d);
  run<42>("v_add_f32_e32", d); run<43>("v_mul_f32_e32", d); run<44>("v_fmac_f32_e32", d);
  run<15>("v_and_or_b32", d); run<14>("v_perm_b32", d); run<10>("v_alignbyte_b32", d);
  return 0;
}
