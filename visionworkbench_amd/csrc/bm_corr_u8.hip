// bm_corr_u8.hip — SSD and NCC block matching for integer-valued imagery in [0,255], register-blocked.
//
// Replaces best_of_search_convolution + fast_box_sum + SquaredCost / NCCCost (src/vw/Stereo/Correlation.cc:33-137,
// src/vw/Stereo/Algorithms.h:43-129, src/vw/Stereo/CostFunctions.h:94-141,179-236) on the domain of the packed SAD path:
// there every quantity the reference accumulates in float64 is an exactly representable integer,
//     S(x,y,d) = sum_window L * R,   A2(x,y) = sum_window L^2,   B2(x,y) = sum_window R^2,
//     SSD cost = A2(x,y) + B2(x+d,y) - 2 S                               compared as integers,
//     NCC cost = double(S) * sqrt((1.0 / A2(x,y)) * (1.0 / B2(x+d,y)))   the reference's float64 sequence, maximised.
//
//   mapping   lane <-> one output column, TY output rows; workgroup = 4 waves = 256 columns x TY rows.
//   products  the lane's LEFT window words (kx bytes per row, all TY+ky-1 rows) stay in registers; the RIGHT words come
//             from an LDS array holding the 32-bit word at EVERY byte offset of the staged rows (consecutive lanes read
//             consecutive dwords: conflict free), so one disparity step is a chain of v_dot4_u32_u8 down the rows whose
//             accumulator is the vertical prefix sum; the ky-row window sum is P[r] - P[r-ky].
//   SSD       the lane folds B2 (from an LDS table, one read per evaluation) and the disparity index into one 32-bit key
//             ((B2 + A2max - 2S) << 8 | d: SSD >= 0 bounds the field to 24 bits for kx*ky <= 129) and keeps min / max keys
//             with v_min3 / v_max3 over disparity pairs; valid <=> min cost != max cost (Correlation.cc:121-133).
//   NCC       the float64 score is needed only where it can decide.  The lane tracks the two largest keys — the fp32 score
//             S * fl32(1/sqrt(B2)) (A2 is a per-pixel constant) with the disparity in its low 8 mantissa bits: when the
//             runner-up's score is more than 2^-13 below, the reference's float64 sequence orders them the same way, so the
//             winner's disparity is the reference's and the pixel is valid.  The other pixels (near ties, flat patches) are queued
//             and ncc_full_kernel evaluates the float64 sequence for every disparity of them (Correlation.cc:91-133).
// Inputs that are not integers in [0,255], or an all-zero window under NCC (1/0), raise the device flag and the float64
// kernel recomputes the image (the protocol of bm_sad_u8.hip).  One search row (sy == 1).
//
// Roofline: HBM bound by the task's definition (20 B per output pixel), VALU-issue bound in fact: per evaluation
// (TY + ky - 1) / TY * ceil(kx / 4) dot4 + 4 ops (SSD) or + 4 ops per sweep (NCC).
#include <algorithm>
#include <cmath>

#include "vwgpu_internal.h"
#include "u8_tile.h"

namespace {

using namespace vwgpu_u8;
typedef uint64_t u64;

constexpr int CTW = 256;          // output columns per workgroup (lane = column)
constexpr int CTHREADS = 256;

struct CorrGeom {
  int sx;
  int nbx;       // right window origins per row = CTW + sx - 1
  int rwd;       // aligned dwords per staged right row
  int urp;       // dwords per row of the every-byte word array
};

__device__ __forceinline__ u32 umin3(u32 a, u32 b, u32 c) { u32 r; asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ u32 umax3(u32 a, u32 b, u32 c) { u32 r; asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float fmax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float fmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

template <int COST, int KX, int KY, int TY>
__global__ void __launch_bounds__(CTHREADS, 2)
bm_corr_u8_kernel(const float* __restrict__ L, ptrdiff_t ls, int lw, int lh,
                  const float* __restrict__ R, ptrdiff_t rs, int rcw, int rch, CorrGeom g,
                  int32_t* __restrict__ out, ptrdiff_t os, int ow, int oh,
                  int* __restrict__ flag_set, int* __restrict__ flag_clear,
                  u32* __restrict__ a2img, u32* __restrict__ b2img, int b2w,
                  u32* __restrict__ full_list, u32* __restrict__ full_count, u32 cap) {
  constexpr bool NCC = (COST == VWGPU_CROSS_CORRELATION);
  constexpr int NW = (KX + 3) / 4, NR = TY + KY - 1;
  constexpr int LWD = CTW / 4 + NW + 1;                         // aligned dwords per staged left row
  constexpr u32 KMASK = (KX % 4 == 0) ? 0xffffffffu : ((1u << (8 * (KX % 4))) - 1u);
  constexpr u32 OFFK = (u32)KX * KY * 65025u;                   // >= A2: SSD >= 0  =>  B2 - 2S + OFFK >= 0
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  const int sx = g.sx, nbx = g.nbx, RWD = g.rwd, URP = g.urp;
  u32* UR = lds;                                                 // [NR][URP]  word at every byte offset of the right rows
  u32* XR = UR + (size_t)NR * URP;                               // [NR][RWD] aligned right words, then [TY][nbx] B2 table
  const size_t xr_dw = (size_t)NR * RWD > (size_t)TY * nbx ? (size_t)NR * RWD : (size_t)TY * nbx;
  u32* LW = XR + xr_dw;                                          // [NR][LWD] aligned left words
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * CTW, y0 = blockIdx.y * TY;
  const int x = x0 + tid;
  if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) *flag_clear = 0;     // the NEXT call's flag

  // ---- stage both tiles as packed u8 (checks the domain) ----
  u32 bad_acc = 0;
  stage_u8_rows<NR>(L, ls, lw, lh, x0, y0, LWD, LWD, LW, tid, CTHREADS, bad_acc);
  stage_u8_rows<NR>(R, rs, rcw, rch, x0, y0, RWD, RWD, XR, tid, CTHREADS, bad_acc);
  __syncthreads();

  // ---- LEFT window words of this lane's column; NCC: A2 of its TY pixels ----
  u32 lwn[NR][NW];
  {
    const int w0 = tid >> 2, sh = tid & 3;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      u32 a[NW + 1];
#pragma unroll
      for (int n = 0; n <= NW; ++n) a[n] = LW[r * LWD + w0 + n];
#pragma unroll
      for (int n = 0; n < NW; ++n) lwn[r][n] = __builtin_amdgcn_alignbyte(a[n + 1], a[n], sh);
      lwn[r][NW - 1] &= KMASK;
    }
  }
  bool zero_window = false;
  if (NCC) {
    u32 h[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      u32 s = 0;
#pragma unroll
      for (int n = 0; n < NW; ++n) s = __builtin_amdgcn_udot4(lwn[r][n], lwn[r][n], s, false);
      h[r] = s;
    }
    u32 a2 = 0;
#pragma unroll
    for (int r = 0; r < KY - 1; ++r) a2 += h[r];
#pragma unroll
    for (int y = 0; y < TY; ++y) {
      a2 += h[y + KY - 1];
      if (x < ow && y0 + y < oh) { a2img[(size_t)(y0 + y) * ow + x] = a2; zero_window |= (a2 == 0); }
      a2 -= h[y];
    }
  }
  // ---- every-byte word array of the right rows ----
  for (int i = tid; i < NR * URP; i += CTHREADS) {
    const int r = i / URP, b = i - r * URP, w = b >> 2;
    UR[i] = __builtin_amdgcn_alignbyte(XR[r * RWD + w + 1], XR[r * RWD + w], b & 3);
  }
  __syncthreads();                                               // UR complete, aligned right words dead
  // ---- B2 table over [TY][nbx]: SSD key part (B2 + OFFK) << 8, NCC fp32 1/sqrt(B2) ----
  u32* B2K = XR;
  for (int xp = tid; xp < nbx; xp += CTHREADS) {
    u32 h[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      u32 s = 0;
#pragma unroll
      for (int n = 0; n < NW; ++n) {
        u32 v = UR[r * URP + xp + 4 * n];
        if (n == NW - 1) v &= KMASK;
        s = __builtin_amdgcn_udot4(v, v, s, false);
      }
      h[r] = s;
    }
    u32 b2 = 0;
#pragma unroll
    for (int r = 0; r < KY - 1; ++r) b2 += h[r];
#pragma unroll
    for (int y = 0; y < TY; ++y) {
      b2 += h[y + KY - 1];
      if (NCC) {
        B2K[y * nbx + xp] = __float_as_uint((float)(1.0 / sqrt((double)b2)));
        const bool inside = (x0 + xp < rcw - KX + 1) && (y0 + y < rch - KY + 1);
        if (inside) { b2img[(size_t)(y0 + y) * b2w + x0 + xp] = b2; zero_window |= (b2 == 0); }
      } else {
        B2K[y * nbx + xp] = (b2 + OFFK) << 8;
      }
      b2 -= h[y];
    }
  }
  __syncthreads();

  // ---- disparity sweeps ----
  // Disparities are walked in quads {d0, d0+4, d0+8, d0+12}: word n of disparity d is the word n-1 of disparity d+4, so a
  // quad needs NW + 3 right words per row instead of 4 NW — the LDS pipe (32 dwords per clock and CU) cannot feed one fresh
  // word to every dot4 (64 lanes per clock and CU).  Each chain's accumulator is the vertical prefix sum; the ky-row window
  // sum is P[r] - P[r-ky].  LDS reads are issued PF rows ahead and pinned there (sched_barrier): left to itself the
  // scheduler sinks every read next to its use and the chains wait out the LDS latency row by row.
  const u32* ur0 = UR + tid;
  const u32* bk0 = B2K + tid;
  constexpr int Q = 4, NWQ = Q + NW - 1, PF = 3;
  auto quad = [&](int d0, auto&& fn) __attribute__((always_inline)) {
    u32 Wd[NR][NWQ], Bq[TY][Q], P[Q][NR], acc[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = 0;
    auto fetch = [&](int r) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < NWQ; ++j) Wd[r][j] = ur0[r * URP + d0 + 4 * j];
      if (r >= KY - 1) {
#pragma unroll
        for (int q = 0; q < Q; ++q) Bq[r - (KY - 1)][q] = bk0[(r - (KY - 1)) * nbx + d0 + 4 * q];
      }
    };
#pragma unroll
    for (int r = 0; r < PF && r < NR; ++r) fetch(r);
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
        u32 sq[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) sq[q] = r >= KY ? P[q][r] - P[q][r - KY] : P[q][r];
        fn(r - (KY - 1), sq, Bq[r - (KY - 1)]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // the quads of one sweep: phase t = d mod 4, then steps of 16; the last quad of a phase may hold fewer than 4 disparities
  auto sweep = [&](auto&& full, auto&& tail) __attribute__((always_inline)) {
    for (int t = 0; t < 4; ++t) {
      const int nt = (sx - t + 3) >> 2;                          // disparities congruent to t
      int a0 = 0;
      for (; a0 + Q <= nt; a0 += Q) full(4 * a0 + t);
      if (a0 < nt) tail(4 * a0 + t, nt - a0);
    }
  };

  if (!NCC) {
    u32 K[TY], Wk[TY];
#pragma unroll
    for (int y = 0; y < TY; ++y) { K[y] = 0xffffffffu; Wk[y] = 0u; }
    sweep(
        [&](int d0) __attribute__((always_inline)) {
          quad(d0, [&](int y, const u32* sq, const u32* bq) __attribute__((always_inline)) {
            u32 k[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) k[q] = bq[q] + (u32)(d0 + 4 * q) - (sq[q] << 9);   // ((B2 + OFFK - 2 S) << 8) | d
            K[y] = umin3(umin3(K[y], k[0], k[1]), k[2], k[3]);
            Wk[y] = umax3(umax3(Wk[y], k[0], k[1]), k[2], k[3]);
          });
        },
        [&](int d0, int nv) __attribute__((always_inline)) {
          quad(d0, [&](int y, const u32* sq, const u32* bq) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < Q; ++q) {
              const u32 k = bq[q] + (u32)(d0 + 4 * q) - (sq[q] << 9);
              if (q < nv) { K[y] = K[y] < k ? K[y] : k; Wk[y] = Wk[y] > k ? Wk[y] : k; }
            }
          });
        });
    if (x < ow) {
#pragma unroll
      for (int y = 0; y < TY; ++y) {
        if (y0 + y < oh) {
          int32_t* o = out + ((ptrdiff_t)(y0 + y) * os + x) * 3;
          o[0] = (int32_t)(K[y] & 0xffu); o[1] = 0;
          o[2] = ((K[y] >> 8) == (Wk[y] >> 8)) ? 0 : 0x7fffffff;   // best == worst (Correlation.cc:121-133)
        }
      }
    }
  } else {
    // One sweep: the two largest keys K1 >= K2 of every pixel, key = fp32 score S * fl32(1/sqrt(B2)) with its low 8 mantissa bits
    // replaced by 255 - d (scores are >= 0, so the bit patterns order like the values; of equal truncated scores the smaller
    // disparity has the larger key).  A key's score part is within 2^-15 (truncation) + 2^-22 (fp32 arithmetic) of the exact
    // score, relatively.  When the runner-up's score part stays below (1 - 2^-13) of the winner's, the reference's float64
    // sequence cannot order the two differently, so the winner's disparity IS the reference's and the pixel is valid
    // (worst < best).  Otherwise — near ties, flat patches, a single disparity — the pixel is queued for ncc_full_kernel.
    u32 K1[TY], K2[TY];
#pragma unroll
    for (int y = 0; y < TY; ++y) { K1[y] = 0u; K2[y] = 0u; }
    auto take = [&](int y, u32 sv, u32 bv, u32 dcode) __attribute__((always_inline)) {
      const float v = (float)sv * __uint_as_float(bv);
      const u32 key = (__float_as_uint(v) & 0xffffff00u) | dcode;
      u32 m2;
      asm("v_med3_u32 %0, %1, %2, %3" : "=v"(m2) : "v"(K1[y]), "v"(K2[y]), "v"(key));   // second largest so far
      K2[y] = m2;
      K1[y] = K1[y] > key ? K1[y] : key;
    };
    sweep(
        [&](int d0) __attribute__((always_inline)) {
          quad(d0, [&](int y, const u32* sq, const u32* bq) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < Q; ++q) take(y, sq[q], bq[q], (u32)(255 - (d0 + 4 * q)));
          });
        },
        [&](int d0, int nv) __attribute__((always_inline)) {
          quad(d0, [&](int y, const u32* sq, const u32* bq) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < Q; ++q)
              if (q < nv) take(y, sq[q], bq[q], (u32)(255 - (d0 + 4 * q)));
          });
        });
    if (x < ow) {
#pragma unroll
      for (int y = 0; y < TY; ++y) {
        if (y0 + y < oh) {
          const size_t p = (size_t)(y0 + y) * ow + x;
          const float m1 = __uint_as_float(K1[y] & 0xffffff00u), m2 = __uint_as_float(K2[y] & 0xffffff00u);
          if (sx == 1 || m2 >= m1 * 0.99987793f) {                // 1 - 2^-13 (also: all scores zero); one disparity: best == worst
            const u32 i = atomicAdd(full_count, 1u);
            if (i < cap) full_list[i] = (u32)p;
          } else {
            int32_t* o = out + ((ptrdiff_t)(y0 + y) * os + x) * 3;
            o[0] = (int32_t)(255u - (K1[y] & 0xffu)); o[1] = 0; o[2] = 0x7fffffff;
          }
        }
      }
    }
  }
  if (bad_acc != 0u || zero_window) atomicOr(flag_set, 1);
}

// NCC, every disparity of the queued pixels: one wave per pixel, lane <-> disparity.  More queued pixels than `cap`
// (a largely flat image) raise the device flag instead: the float64 kernel then recomputes the image.
__global__ void __launch_bounds__(256)
ncc_full_kernel(const float* __restrict__ L, ptrdiff_t ls, const float* __restrict__ R, ptrdiff_t rs, int kx, int ky, int sx,
                const u32* __restrict__ a2img, const u32* __restrict__ b2img, int b2w,
                int32_t* __restrict__ out, ptrdiff_t os, int ow, int* __restrict__ flag,
                const u32* __restrict__ full_list, const u32* __restrict__ full_count, u32 cap) {
  if (*flag) return;
  const u32 total = *full_count;
  if (total > cap) {
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(flag, 1);
    return;
  }
  const int lane = threadIdx.x & 63;
  const u32 nwaves = gridDim.x * 4u;
  // Round 6: the ky rows of the right image a pixel's windows can reach (kx + sx - 1 values each) and its left window go to a wave-private LDS
  // area once, as integers; the 129 x 121 products then read LDS (the left value of a product is one address for the whole wave: a broadcast)
  // instead of asking the L1 for every operand again (130 -> 6x us for the ~2 % of the bench pair's pixels that are queued).
  extern __shared__ u32 full_lds[];
  const int rwid = kx + sx - 1;
  u32* const Rw = full_lds + (size_t)(threadIdx.x >> 6) * ((size_t)ky * rwid + (size_t)ky * kx);
  u32* const Lw = Rw + (size_t)ky * rwid;
  for (u32 e = blockIdx.x * 4u + (threadIdx.x >> 6); e < total; e += nwaves) {
    const u32 p = full_list[e];
    const int y = (int)(p / (u32)ow), x = (int)(p - (u32)y * (u32)ow);
    const double pl = 1.0 / (double)a2img[p];
    double best = 0.0, worst = 0.0;
    int bd = 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the previous pixel's reads are done (a wave's LDS operations complete in order)
    for (int j = 0; j < ky; ++j) {
      const float* rp = R + (ptrdiff_t)(y + j) * rs + x;
      for (int c = lane; c < rwid; c += 64) Rw[j * rwid + c] = (u32)rp[c];
      if (lane < kx) Lw[j * kx + lane] = (u32)L[(ptrdiff_t)(y + j) * ls + x + lane];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int d0 = 0; d0 < sx; d0 += 64) {
      const int d = d0 + lane;
      const bool act = d < sx;
      u32 S = 0;
      if (act)
        for (int j = 0; j < ky; ++j) {
          const u32* lp = Lw + j * kx;
          const u32* rp = Rw + j * rwid + d;
          for (int i = 0; i < kx; ++i) S += lp[i] * rp[i];
        }
      double v = act ? (double)S * sqrt(pl * (1.0 / (double)b2img[(size_t)y * b2w + x + d])) : -1.0;   // scores are >= 0
      double w = act ? v : INFINITY;
      int vd = act ? d : 0x7fffffff;
      for (int o = 32; o > 0; o >>= 1) {                          // (maximum, smallest index) and minimum across the lanes
        const double ov = __shfl_xor(v, o), owv = __shfl_xor(w, o);
        const int od = __shfl_xor(vd, o);
        if (ov > v || (ov == v && od < vd)) { v = ov; vd = od; }
        if (owv < w) w = owv;
      }
      if (d0 == 0) { best = v; bd = vd; worst = w; }
      else { if (v > best) { best = v; bd = vd; } if (w < worst) worst = w; }
    }
    if (lane == 0) {
      int32_t* o = out + ((ptrdiff_t)y * os + x) * 3;
      o[0] = bd; o[1] = 0; o[2] = (best == worst) ? 0 : 0x7fffffff;   // Correlation.cc:121-133
    }
  }
}

typedef void (*CorrFn)(const float*, ptrdiff_t, int, int, const float*, ptrdiff_t, int, int, CorrGeom, int32_t*, ptrdiff_t, int, int, int*, int*,
                       u32*, u32*, int, u32*, u32*, u32);
struct CorrLaunch { int cost, kx, ky, ty; CorrFn fn; };
#define VW_CORR(C, KX, KY, TY) CorrLaunch{C, KX, KY, TY, bm_corr_u8_kernel<C, KX, KY, TY>}
const CorrLaunch kCorr[] = {
    VW_CORR(VWGPU_SQUARED_DIFFERENCE, 3, 3, 16), VW_CORR(VWGPU_SQUARED_DIFFERENCE, 5, 5, 16), VW_CORR(VWGPU_SQUARED_DIFFERENCE, 7, 7, 16),
    VW_CORR(VWGPU_SQUARED_DIFFERENCE, 9, 9, 16), VW_CORR(VWGPU_SQUARED_DIFFERENCE, 11, 11, 16),
    VW_CORR(VWGPU_CROSS_CORRELATION, 3, 3, 16), VW_CORR(VWGPU_CROSS_CORRELATION, 5, 5, 16), VW_CORR(VWGPU_CROSS_CORRELATION, 7, 7, 16),
    VW_CORR(VWGPU_CROSS_CORRELATION, 9, 9, 16), VW_CORR(VWGPU_CROSS_CORRELATION, 11, 11, 16),
};
#undef VW_CORR

const CorrLaunch* find_corr(int cost, int kx, int ky) {
  for (const CorrLaunch& l : kCorr)
    if (l.cost == cost && l.kx == kx && l.ky == ky) return &l;
  return nullptr;
}

CorrGeom corr_geom(int kx, int sx) {
  CorrGeom g;
  const int nw = (kx + 3) / 4;
  g.sx = sx;
  g.nbx = CTW + sx - 1;
  g.urp = g.nbx + 4 * nw + 16;                                  // + the words a clamped tail quad may touch
  g.rwd = (g.urp + 3) / 4 + 1;
  return g;
}

size_t corr_lds_bytes(const CorrLaunch& l, const CorrGeom& g) {
  const int nw = (l.kx + 3) / 4, nr = l.ty + l.ky - 1;
  const size_t xr = std::max((size_t)nr * g.rwd, (size_t)l.ty * g.nbx);
  return ((size_t)nr * g.urp + xr + (size_t)nr * (CTW / 4 + nw + 1)) * sizeof(u32);
}

}  // namespace

bool vwgpu_bm_corr_u8_supported(int cost_type, int kx, int ky, int sx, int sy) {
  const CorrLaunch* l = find_corr(cost_type, kx, ky);
  if (!l || sy != 1 || sx > 256) return false;
  return corr_lds_bytes(*l, corr_geom(kx, sx)) <= 80 * 1024;     // two workgroups per CU
}

int vwgpu_launch_bm_corr_u8(vwgpu_ctx* ctx, int cost_type, const float* left, int lw, int lh, ptrdiff_t ls,
                            const float* right, int rw, int rh, ptrdiff_t rs, int kx, int ky, int sx, int sy,
                            int32_t* out, ptrdiff_t os, int** d_fallback_flag) {
  (void)rw; (void)rh; (void)sy;
  const CorrLaunch* l = find_corr(cost_type, kx, ky);
  if (!l) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "no register-blocked SSD / NCC kernel for %dx%d", kx, ky);
  const int ow = lw - kx + 1, oh = lh - ky + 1;
  const int rcw = lw + sx - 1, rch = lh;
  const CorrGeom g = corr_geom(kx, sx);
  int* flag_set = nullptr; int* flag_clear = nullptr;
  int rc = vwgpu_next_flags(ctx, 0, &flag_set, &flag_clear, nullptr);
  if (rc) return rc;
  *d_fallback_flag = flag_set;
  const bool ncc = cost_type == VWGPU_CROSS_CORRELATION;
  u32 *a2 = nullptr, *b2 = nullptr, *full_list = nullptr, *full_count = nullptr;
  const int b2w = rcw - kx + 1;
  // pixels evaluated over every disparity by ncc_full_kernel: at most ~3 % of the image, beyond that the float64 kernel is faster
  const u32 cap = (u32)std::max<size_t>(4096, (size_t)ow * oh / 32);
  if (ncc) {
    if ((size_t)ow * oh >= 0xffffffffull) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "bm_corr_u8: image too large");
    const size_t na = vwgpu_align_up((size_t)ow * oh * 4, 256), nb = vwgpu_align_up((size_t)b2w * oh * 4, 256),
                 nl = vwgpu_align_up((size_t)cap * 4 + 256, 256);
    rc = vwgpu_arena_reserve(ctx, &ctx->scratch, na + nb + nl);
    if (rc) return rc;
    char* base = static_cast<char*>(ctx->scratch.base);
    a2 = reinterpret_cast<u32*>(base); b2 = reinterpret_cast<u32*>(base + na);
    full_count = reinterpret_cast<u32*>(base + na + nb); full_list = full_count + 64;
    VWGPU_HIP(ctx, hipMemsetAsync(full_count, 0, 4, ctx->stream));
  }
  const size_t shmem = corr_lds_bytes(*l, g);
  if (shmem > 64 * 1024)
    VWGPU_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(l->fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  {
    vwgpu_prof_scope ps(ctx, "bm_corr_u8");
    hipLaunchKernelGGL(l->fn, dim3((ow + CTW - 1) / CTW, (oh + l->ty - 1) / l->ty), dim3(CTHREADS), shmem, ctx->stream,
                       left, ls, lw, lh, right, rs, rcw, rch, g, out, os, ow, oh, flag_set, flag_clear, a2, b2, b2w, full_list, full_count, cap);
  }
  VWGPU_HIP(ctx, hipGetLastError());
  if (ncc) return vwgpu_launch_ncc_full(ctx, left, ls, right, rs, kx, ky, sx, a2, b2, b2w, out, os, ow, flag_set, full_list, full_count, cap);
  return VWGPU_OK;
}

// The float64 sequence for the queued pixels of an integer NCC matcher (bm_corr_u8.hip, bm_corr_u16.hip: products and window sums below 2^32).
int vwgpu_launch_ncc_full(vwgpu_ctx* ctx, const float* left, ptrdiff_t ls, const float* right, ptrdiff_t rs, int kx, int ky, int sx,
                          const uint32_t* a2, const uint32_t* b2, int b2w, int32_t* out, ptrdiff_t os, int ow, int* flag,
                          const uint32_t* full_list, const uint32_t* full_count, uint32_t cap) {
  vwgpu_prof_scope ps(ctx, "ncc_full");
  const size_t full_lds = (size_t)4 * ((size_t)ky * (kx + sx - 1) + (size_t)ky * kx) * sizeof(uint32_t);      // kx, ky <= 11 (64 lanes cover a window row), sx <= 256: <= 49 KB
  hipLaunchKernelGGL(ncc_full_kernel, dim3(2048), dim3(256), full_lds, ctx->stream, left, ls, right, rs, kx, ky, sx, a2, b2, b2w,
                     out, os, ow, flag, full_list, full_count, cap);
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}
