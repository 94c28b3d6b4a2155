// ubench_mfma_corr.hip — the bilinear term of SSD / NCC block matching, S(x, y, d) = sum over a KX x KY window of L(x+i, y+j) * R(x+i+d, y+j),
// for one 256 x 16 tile and 128 disparities, as two inner loops (the experiment the round-2 review asked for):
//   (a) dot4   the formulation of bm_corr_u8.hip: lane <-> column, LEFT window words in registers, RIGHT words from an LDS array that holds the
//              32-bit word at every byte offset, quads of disparities {d, d+4, d+8, d+12} sharing words, chains of v_dot4_u32_u8 down the rows
//              (vertical prefix sums, window = P[r] - P[r-KY]);
//   (b) mfma   per 16 columns x 16 disparities x one row ONE v_mfma_i32_16x16x32_i8:  C[d][x] += A[d][k] * B[k][x]  with
//              A[d][k] = R(x0 + d0 + d + k)   (a Hankel matrix of the right row: lane (d, kg) reads the 8 bytes at offset d + 8 kg),
//              B[k][x] = L(x0 + k) if 0 <= k - x < KX else 0   (the banded left row; 16 + KX - 1 <= 32 columns),
//              values centred (v - 128: the instruction multiplies SIGNED bytes; a product kernel adds 128 (sum L + sum R) - 16384 KX KY back),
//              the accumulator chained down the rows, a ring of KY + 1 accumulators gives P[r] - P[r-KY]; two disparity tiles interleaved.
// Both kernels store S of workgroup 0 (checked on the host against a direct evaluation) and fold every S into a checksum, so the loops are
// not optimised away.  Only the inner loops are compared: staging, B2 / A2 tables and the per-evaluation finishing (score, key, best /
// runner-up: ~6 instruction slots per evaluation in bm_corr_u8) are the same work in both formulations and are left out.
// Build + run (GPU box): hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_corr.hip -o /tmp/ubench_mfma_corr && /tmp/ubench_mfma_corr
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32;
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int TW = 256, TY = 16, KX = 11, KY = 11, NR = TY + KY - 1, SX = 128;
constexpr int NW = (KX + 3) / 4;                       // dot4 words per window row
constexpr int RB = TW + SX + 32;                       // bytes per right row held in LDS
constexpr int LB = TW + 32;                            // bytes per left row
constexpr int URP = RB - 4;                            // words of the every-byte array per row

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

__host__ __device__ inline int pix(int img, int tile, int r, int c) {     // synthetic byte image (the same on host and device)
  u32 x = (u32)img * 0x9e3779b9u + (u32)tile * 0x85ebca6bu + (u32)r * 0xc2b2ae35u + (u32)c * 0x27d4eb2fu;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return (int)(x & 255u);
}

// ---- (a) dot4 ----------------------------------------------------------------------------------------------------------------------
template <bool STORE>
__global__ void __launch_bounds__(256, 2) dot4_kernel(int* __restrict__ s_out, unsigned long long* __restrict__ sums, int reps) {
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  u32* UR = lds;                                       // [NR][URP]: word at every byte offset of the right rows
  unsigned char* Lb = reinterpret_cast<unsigned char*>(UR + NR * URP);   // [NR][LB]
  unsigned char* Rb = Lb + NR * LB;                    // [NR][RB]
  const int tid = threadIdx.x, tile = blockIdx.x;
  for (int i = tid; i < NR * LB; i += 256) Lb[i] = (unsigned char)pix(0, tile, i / LB, i % LB);
  for (int i = tid; i < NR * RB; i += 256) Rb[i] = (unsigned char)pix(1, tile, i / RB, i % RB);
  __syncthreads();
  for (int i = tid; i < NR * URP; i += 256) {
    const int r = i / URP, b = i - r * URP;
    const unsigned char* p = Rb + r * RB + b;
    UR[i] = (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24);
  }
  u32 lwn[NR][NW];
  constexpr u32 KMASK = (KX % 4 == 0) ? 0xffffffffu : ((1u << (8 * (KX % 4))) - 1u);
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int n = 0; n < NW; ++n) {
      const unsigned char* p = Lb + r * LB + tid + 4 * n;
      lwn[r][n] = (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24);
      if (n == NW - 1) lwn[r][n] &= KMASK;
    }
  __syncthreads();
  unsigned long long chk = 0;
  const u32* ur0 = UR + tid;
  constexpr int Q = 4, NWQ = Q + NW - 1, PF = 3;
  for (int rep = 0; rep < reps; ++rep)
  for (int t = 0; t < 4; ++t)
    for (int a0 = 0; a0 < SX / 4; a0 += Q) {
      const int d0 = 4 * a0 + t;
      u32 Wd[NR][NWQ], P[Q][NR], acc[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) acc[q] = 0;
      auto fetch = [&](int r) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NWQ; ++j) Wd[r][j] = ur0[r * URP + d0 + 4 * j];
      };
#pragma unroll
      for (int r = 0; r < PF; ++r) fetch(r);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (r + PF < NR) fetch(r + PF);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NW; ++n)
#pragma unroll
          for (int q = 0; q < Q; ++q) acc[q] = __builtin_amdgcn_udot4(lwn[r][n], Wd[r][q + n], acc[q], false);
#pragma unroll
        for (int q = 0; q < Q; ++q) P[q][r] = acc[q];
        if (r >= KY - 1) {
          const int y = r - (KY - 1);
#pragma unroll
          for (int q = 0; q < Q; ++q) {
            const u32 s = r >= KY ? P[q][r] - P[q][r - KY] : P[q][r];
            chk += s;
            if (STORE) s_out[((size_t)y * TW + tid) * SX + d0 + 4 * q] = (int)s;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  atomicAdd(sums, chk);
}

// ---- (b) mfma ----------------------------------------------------------------------------------------------------------------------
template <bool STORE>
__global__ void __launch_bounds__(256, 2) mfma_kernel(int* __restrict__ s_out, unsigned long long* __restrict__ sums, int reps) {
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  u32* UR = lds;                                       // [NR][URP]: word at every byte offset of the centred right rows
  u32* LW = UR + NR * URP;                             // [NR][LB / 4]: centred left rows, aligned words
  unsigned char* Rb = reinterpret_cast<unsigned char*>(LW + NR * (LB / 4));   // [NR][RB] staging
  const int tid = threadIdx.x, tile = blockIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < NR * RB; i += 256) Rb[i] = (unsigned char)(pix(1, tile, i / RB, i % RB) - 128);
  for (int i = tid; i < NR * (LB / 4); i += 256) {
    const int r = i / (LB / 4), c = (i - r * (LB / 4)) * 4;
    u32 w = 0;
    for (int b = 0; b < 4; ++b) w |= (u32)(unsigned char)(pix(0, tile, r, c + b) - 128) << (8 * b);
    LW[i] = w;
  }
  __syncthreads();
  for (int i = tid; i < NR * URP; i += 256) {
    const int r = i / URP, b = i - r * URP;
    const unsigned char* p = Rb + r * RB + b;
    UR[i] = (u32)p[0] | ((u32)p[1] << 8) | ((u32)p[2] << 16) | ((u32)p[3] << 24);
  }
  __syncthreads();
  // operand layouts of v_mfma_i32_16x16x32_i8: A: lane l holds row l % 16, bytes k = 8 (l / 16) + 0..7; B: lane l holds column l % 16, the same
  // k; C: lane l holds column l % 16, rows 4 (l / 16) + 0..3
  const int mn = lane & 15, kg = lane >> 4;
  unsigned long long band = 0;                         // byte b of the lane's B operand is inside the window of column mn
#pragma unroll
  for (int b = 0; b < 8; ++b) { const int k = 8 * kg + b; if (k - mn >= 0 && k - mn < KX) band |= 0xffull << (8 * b); }
  unsigned long long chk = 0;
  for (int rep = 0; rep < reps; ++rep)
  for (int xb = wave * 4; xb < wave * 4 + 4; ++xb) {
    const int x0 = xb * 16;
    long Bop[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const u32* p = LW + r * (LB / 4) + x0 / 4 + 2 * kg;
      Bop[r] = (long)((((unsigned long long)p[1] << 32) | p[0]) & band);
    }
    for (int dt = 0; dt < SX / 16; dt += 2) {          // two disparity tiles interleaved: two independent accumulator chains
      v4i C[2][KY + 1];
      const u32* ua = UR + x0 + dt * 16 + mn + 8 * kg;
#pragma unroll
      for (int r = 0; r < NR; ++r) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const u32* pa = ua + r * URP + h * 16;
          const long Aop = (long)(((unsigned long long)pa[4] << 32) | pa[0]);
          const v4i zero = {0, 0, 0, 0};
          C[h][r % (KY + 1)] = __builtin_amdgcn_mfma_i32_16x16x32_i8(Aop, Bop[r], r == 0 ? zero : C[h][(r + KY) % (KY + 1)], 0, 0, 0);
        }
        if (r >= KY - 1) {
          const int y = r - (KY - 1);
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int s = r >= KY ? C[h][r % (KY + 1)][i] - C[h][(r + 1) % (KY + 1)][i] : C[h][r % (KY + 1)][i];
              chk += (unsigned long long)(long long)s;
              if (STORE) s_out[((size_t)y * TW + x0 + mn) * SX + (dt + h) * 16 + 4 * kg + i] = s;
            }
        }
        if (r % 2 == 1) __builtin_amdgcn_sched_barrier(0);       // (keeps the operand reads of later rows from piling up in registers)
      }
    }
  }
  atomicAdd(sums, chk);
}

int main() {
  const int tiles = 4096, reps = 4;
  int* d_s; unsigned long long* d_sum;
  const size_t ns = (size_t)TY * TW * SX;
  CHECK(hipMalloc(&d_s, ns * 4)); CHECK(hipMalloc(&d_sum, 8));
  std::vector<int> got(ns), want(ns);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const size_t lds_a = (size_t)NR * URP * 4 + (size_t)NR * LB + (size_t)NR * RB;
  const size_t lds_b = (size_t)NR * URP * 4 + (size_t)NR * LB + (size_t)NR * RB;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dot4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dot4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b));
  for (int variant = 0; variant < 2; ++variant) {
    const int centre = variant ? 128 : 0;
    for (int y = 0; y < TY; ++y)
      for (int x = 0; x < TW; ++x)
        for (int d = 0; d < SX; ++d) {
          int s = 0;
          for (int j = 0; j < KY; ++j)
            for (int i = 0; i < KX; ++i) s += (pix(0, 0, y + j, x + i) - centre) * (pix(1, 0, y + j, x + i + d) - centre);
          want[((size_t)y * TW + x) * SX + d] = s;
        }
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
      CHECK(hipMemset(d_sum, 0, 8));
      CHECK(hipEventRecord(e0));
      if (variant == 0) hipLaunchKernelGGL(dot4_kernel<false>, dim3(tiles), dim3(256), lds_a, 0, d_s, d_sum, reps);
      else hipLaunchKernelGGL(mfma_kernel<false>, dim3(tiles), dim3(256), lds_b, 0, d_s, d_sum, reps);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      CHECK(hipGetLastError());
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    // the same loops once more on tile 0 with every S stored
    CHECK(hipMemset(d_s, 0xff, ns * 4));
    if (variant == 0) hipLaunchKernelGGL(dot4_kernel<true>, dim3(1), dim3(256), lds_a, 0, d_s, d_sum, 1);
    else hipLaunchKernelGGL(mfma_kernel<true>, dim3(1), dim3(256), lds_b, 0, d_s, d_sum, 1);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(got.data(), d_s, ns * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < ns; ++i) bad += got[i] != want[i];
    const double evals = (double)tiles * reps * TW * TY * SX;
    printf("%-5s %8.3f ms for %d tiles x %d  = %.1f G evaluations/s  (inner loop only), %.3f clk per evaluation and CU at 2.4 GHz, mismatches in tile 0: %zu of %zu\n",
           variant ? "mfma" : "dot4", best, tiles, reps, evals / (best * 1e-3) / 1e9, best * 1e-3 * 2.4e9 * 256 / evals, bad, ns);
  }
  return 0;
}
