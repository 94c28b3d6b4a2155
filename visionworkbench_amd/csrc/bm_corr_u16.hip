// bm_corr_u16.hip — SSD and NCC block matching for integer-valued imagery in [0,4095] (11- / 12-bit sensors stored in 16 bits).
//
// Replaces best_of_search_convolution + fast_box_sum + SquaredCost / NCCCost (src/vw/Stereo/Correlation.cc:33-137,
// src/vw/Stereo/Algorithms.h:43-129, src/vw/Stereo/CostFunctions.h:94-141,179-236) on the domain where the reference's
// arithmetic is integer arithmetic.  The reference forms every cost ELEMENT in float32 — (L - R)^2, L * R — and sums the
// elements in float64; below 2^12 the elements stay below 2^24 (exact in float32) and the window sums below 2^31, so
//     S(x,y,d) = sum_window L * R,   A2(x,y) = sum_window L^2,   B2(x,y) = sum_window R^2        are 32-bit integers,
//     SSD cost = A2(x,y) + B2(x+d,y) - 2 S,     NCC cost = double(S) * sqrt((1.0 / A2(x,y)) * (1.0 / B2(x+d,y))).
// From 2^12 upwards a float32 product rounds and no integer pipeline reproduces the sums: such imagery stays with the float64 kernel.
//
//   mapping   lane <-> one output column x TY rows, workgroup = 4 waves = 256 columns (the layout of bm_corr_u8.hip / bm_sad_u16.hip)
//   products  v_dot2_u32_u16 on pixel PAIRS: the lane's LEFT pairs (kx pixels per row, all TY+ky-1 rows) live in registers, the
//             RIGHT pairs come from an LDS array holding the pair that starts at EVERY pixel; a chain down the rows accumulates the
//             vertical prefix sum, the ky-row window sum is P[r] - P[r-ky]
//   quads     disparities {d0, d0+2, d0+4, d0+6} share right pairs (pair n of d is pair n-1 of d+2): kx/2 + 4 LDS reads per row
//             instead of four windows; the last pair of a window holds one live pixel, its LEFT half is cleared
//   SSD       (cost, d) is one 64-bit key (cost = B2 + A2max - 2 S < 2^32), kept with a 64-bit compare; the largest cost alongside;
//             valid <=> they differ (Correlation.cc:121-133)
//   NCC       the two-largest-keys scheme of bm_corr_u8.hip (fp32 score with the disparity in the low mantissa bits; pixels whose
//             runner-up is within 2^-13 are evaluated in float64 by ncc_full_kernel)
// Inputs outside the domain, or an all-zero window under NCC, raise the device flag; the caller then runs the float64 kernel.
// One search row (sy == 1).
//
// Roofline: HBM bound by the task's definition (20 B per output pixel); VALU-issue bound in fact: (TY+ky-1)/TY * (kx/2 + 1) dot2 +
// 5 (SSD) / 9 (NCC) instruction slots per (pixel, disparity).
#include <algorithm>
#include <cmath>
#include <type_traits>

#include "vwgpu_internal.h"

namespace {

typedef uint32_t u32;
typedef uint64_t u64;
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

constexpr int UTW = 256;          // output columns per workgroup
constexpr int UTHREADS = 256;
constexpr u32 UMAXV = 4095u;

__device__ __forceinline__ u32 dot2(u32 a, u32 b, u32 c) {
  return __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b), c, false);
}

// float -> u16 with the exactness test of the path: integer-valued and inside [0,4095]
__device__ __forceinline__ u32 to_u12(float v, bool& bad) {
  const float r = rintf(v);
  bad |= !(r == v && v >= 0.0f && v <= (float)UMAXV);
  return (u32)(int)fminf(fmaxf(r, 0.0f), (float)UMAXV);
}

// dst[r][g] = pixels (x0 + 2g, x0 + 2g + 1) of row y0 + r as a u16 pair, zero outside the image
template <int NROWS>
__device__ __forceinline__ void stage_u12_rows(const float* __restrict__ img, ptrdiff_t stride, int w, int h, int x0, int y0,
                                               int npairs, u32* __restrict__ dst, int tid, bool& bad) {
  for (int i = tid; i < NROWS * npairs; i += UTHREADS) {
    const int r = i / npairs, gq = i - r * npairs;
    const int x = x0 + 2 * gq, y = y0 + r;
    u32 p = 0;
    if (y < h) {
      const float* row = img + (ptrdiff_t)y * stride;
      if (x < w) p = to_u12(row[x], bad);
      if (x + 1 < w) p |= to_u12(row[x + 1], bad) << 16;
    }
    dst[i] = p;
  }
}

struct U16Geom {
  int sx;
  int nbx;       // right window origins per row = UTW + sx - 1
  int urp;       // dwords per row of the pair-at-every-pixel array
  int rpd;       // aligned pairs per staged right row
};

template <int COST, int KX, int KY, int TY>
__global__ void __launch_bounds__(UTHREADS, 2)
bm_corr_u16_kernel(const float* __restrict__ L, ptrdiff_t ls, int lw, int lh,
                   const float* __restrict__ R, ptrdiff_t rs, int rcw, int rch, U16Geom g,
                   int32_t* __restrict__ out, ptrdiff_t os, int ow, int oh,
                   int* __restrict__ flag_set, int* __restrict__ flag_clear,
                   u32* __restrict__ a2img, u32* __restrict__ b2img, int b2w,
                   u32* __restrict__ full_list, u32* __restrict__ full_count, u32 cap) {
  constexpr bool NCC = (COST == VWGPU_CROSS_CORRELATION);
  constexpr int NWF = KX / 2;                                    // full pairs of a window row; one more half pair (kx is odd)
  constexpr int NR = TY + KY - 1;
  constexpr int LPD = UTW / 2 + NWF + 2;                         // aligned pairs per staged left row
  constexpr u32 OFFK = (u32)KX * KY * UMAXV * UMAXV;             // >= A2: SSD >= 0  =>  B2 - 2S + OFFK >= 0; B2 + OFFK < 2^32
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  const int sx = g.sx, nbx = g.nbx, urp = g.urp, rpd = g.rpd;
  u32* UR = lds;                                                 // [NR][urp]  the pair starting at every pixel of the right rows
  u32* XR = UR + (size_t)NR * urp;                               // [NR][rpd]  aligned right pairs (staging), then [TY][nbx] B2 table
  const size_t xr_dw = (size_t)NR * rpd > (size_t)TY * nbx ? (size_t)NR * rpd : (size_t)TY * nbx;
  u32* LW = XR + xr_dw;                                          // [NR][LPD]  aligned left pairs
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * UTW, y0 = blockIdx.y * TY;
  const int x = x0 + tid;
  if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) *flag_clear = 0;     // the NEXT call's flag

  bool bad = false;
  stage_u12_rows<NR>(L, ls, lw, lh, x0, y0, LPD, LW, tid, bad);
  stage_u12_rows<NR>(R, rs, rcw, rch, x0, y0, rpd, XR, tid, bad);
  __syncthreads();
  // ---- LEFT pairs of this lane's column: pixels x + 2n, x + 2n + 1; the dead pixel of the last pair is cleared ----
  u32 lwn[NR][NWF + 1];
  {
    const int w0 = tid >> 1;
    const u32 sh = (u32)(tid & 1) * 16u;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      u32 a[NWF + 2];
#pragma unroll
      for (int n = 0; n <= NWF + 1; ++n) a[n] = LW[r * LPD + w0 + n];
#pragma unroll
      for (int n = 0; n <= NWF; ++n) lwn[r][n] = __builtin_amdgcn_alignbit(a[n + 1], a[n], sh);
      lwn[r][NWF] &= 0xffffu;
    }
  }
  bool zero_window = false;
  if (NCC) {
    u32 h[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      u32 s = 0;
#pragma unroll
      for (int n = 0; n <= NWF; ++n) s = dot2(lwn[r][n], lwn[r][n], s);
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
  // ---- the pair starting at every pixel of the right rows ----
  for (int i = tid; i < NR * urp; i += UTHREADS) {
    const int r = i / urp, b = i - r * urp, w = b >> 1;
    UR[i] = __builtin_amdgcn_alignbit(XR[r * rpd + w + 1], XR[r * rpd + w], (u32)(b & 1) * 16u);
  }
  __syncthreads();                                               // UR complete, aligned right pairs dead
  // ---- B2 table over [TY][nbx]: SSD B2 + OFFK, NCC fp32 1/sqrt(B2) ----
  u32* B2K = XR;
  for (int xp = tid; xp < nbx; xp += UTHREADS) {
    u32 h[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      u32 s = 0;
#pragma unroll
      for (int n = 0; n <= NWF; ++n) {
        u32 v = UR[r * urp + xp + 2 * n];
        if (n == NWF) v &= 0xffffu;
        s = dot2(v, v, s);
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
        B2K[y * nbx + xp] = b2 + OFFK;
      }
      b2 -= h[y];
    }
  }
  __syncthreads();

  // ---- disparity sweeps ----
  const u32* ur0 = UR + tid;
  const u32* bk0 = B2K + tid;
  constexpr int Q = 4, NWQ = Q + NWF, PF = 3;                    // words d0 + 2 j, j < NWQ: pair n of chain q is word q + n
  auto quad = [&](int d0, auto&& fn) __attribute__((always_inline)) {
    u32 Wd[NR][NWQ], Bq[TY][Q], P[Q][NR], acc[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = 0;
    auto fetch = [&](int r) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < NWQ; ++j) Wd[r][j] = ur0[r * urp + d0 + 2 * j];
      if (r >= KY - 1) {
#pragma unroll
        for (int q = 0; q < Q; ++q) Bq[r - (KY - 1)][q] = bk0[(r - (KY - 1)) * nbx + d0 + 2 * q];
      }
    };
#pragma unroll
    for (int r = 0; r < PF && r < NR; ++r) fetch(r);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if (r + PF < NR) fetch(r + PF);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n <= NWF; ++n)
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[q] = dot2(lwn[r][n], Wd[r][q + n], acc[q]);
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
  // the quads of one sweep: phase t = d mod 2, then steps of 8; the last quad of a phase may hold fewer than 4 disparities
  auto sweep = [&](auto&& full, auto&& tail) __attribute__((always_inline)) {
    for (int t = 0; t < 2; ++t) {
      const int nt = (sx - t + 1) >> 1;                          // disparities congruent to t
      int a0 = 0;
      for (; a0 + Q <= nt; a0 += Q) full(2 * a0 + t);
      if (a0 < nt) tail(2 * a0 + t, nt - a0);
    }
  };

  if (!NCC) {
    u64 K[TY];
    u32 Wc[TY];
#pragma unroll
    for (int y = 0; y < TY; ++y) { K[y] = ~0ull; Wc[y] = 0u; }
    auto take = [&](int y, u32 sv, u32 bv, int d) __attribute__((always_inline)) {
      const u32 c = bv - (sv << 1);                              // B2 + OFFK - 2 S
      const u64 k = ((u64)c << 32) | (u32)d;
      K[y] = k < K[y] ? k : K[y];
      Wc[y] = c > Wc[y] ? c : Wc[y];
    };
    sweep(
        [&](int d0) __attribute__((always_inline)) {
          quad(d0, [&](int y, const u32* sq, const u32* bq) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < Q; ++q) take(y, sq[q], bq[q], d0 + 2 * q);
          });
        },
        [&](int d0, int nv) __attribute__((always_inline)) {
          quad(d0, [&](int y, const u32* sq, const u32* bq) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < Q; ++q)
              if (q < nv) take(y, sq[q], bq[q], d0 + 2 * q);
          });
        });
    if (x < ow) {
#pragma unroll
      for (int y = 0; y < TY; ++y) {
        if (y0 + y < oh) {
          int32_t* o = out + ((ptrdiff_t)(y0 + y) * os + x) * 3;
          o[0] = (int32_t)(u32)K[y]; o[1] = 0;
          o[2] = ((u32)(K[y] >> 32) == Wc[y]) ? 0 : 0x7fffffff;     // best == worst (Correlation.cc:121-133)
        }
      }
    }
  } else {
    // The two largest keys K1 >= K2 of every pixel (see bm_corr_u8.hip): fp32 score S * fl32(1/sqrt(B2)) with 255 - d in the low 8
    // mantissa bits.  float(S) rounds (S < 2^31): 2^-24 more relative error, far inside the 2^-13 margin.
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
            for (int q = 0; q < Q; ++q) take(y, sq[q], bq[q], (u32)(255 - (d0 + 2 * q)));
          });
        },
        [&](int d0, int nv) __attribute__((always_inline)) {
          quad(d0, [&](int y, const u32* sq, const u32* bq) __attribute__((always_inline)) {
#pragma unroll
            for (int q = 0; q < Q; ++q)
              if (q < nv) take(y, sq[q], bq[q], (u32)(255 - (d0 + 2 * q)));
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
  if (__syncthreads_or(bad || zero_window) && tid == 0) atomicOr(flag_set, 1);
}

typedef void (*Corr16Fn)(const float*, ptrdiff_t, int, int, const float*, ptrdiff_t, int, int, U16Geom, int32_t*, ptrdiff_t, int, int, int*, int*,
                         u32*, u32*, int, u32*, u32*, u32);
struct Corr16Launch { int cost, kx, ky, ty; Corr16Fn fn; };
// rows per workgroup: the LEFT pairs of TY + ky - 1 rows are registers (kx/2 + 1 each); the tallest TY that compiles without scratch
// (measured at 4096^2 x 129: 9x9 at TY 16 spills 324 B and takes 2.43 ms, at TY 12 1.46 ms; 11x11 at TY 12 2.16 ms, at TY 8 1.54 ms)
#define VW_C16(C, KX, KY, TY) Corr16Launch{C, KX, KY, TY, bm_corr_u16_kernel<C, KX, KY, TY>}
const Corr16Launch kCorr16[] = {
    VW_C16(VWGPU_SQUARED_DIFFERENCE, 3, 3, 16), VW_C16(VWGPU_SQUARED_DIFFERENCE, 5, 5, 16), VW_C16(VWGPU_SQUARED_DIFFERENCE, 7, 7, 14),
    VW_C16(VWGPU_SQUARED_DIFFERENCE, 9, 9, 12), VW_C16(VWGPU_SQUARED_DIFFERENCE, 11, 11, 8),
    VW_C16(VWGPU_CROSS_CORRELATION, 3, 3, 16), VW_C16(VWGPU_CROSS_CORRELATION, 5, 5, 16), VW_C16(VWGPU_CROSS_CORRELATION, 7, 7, 16),
    VW_C16(VWGPU_CROSS_CORRELATION, 9, 9, 12), VW_C16(VWGPU_CROSS_CORRELATION, 11, 11, 8),
};
#undef VW_C16

const Corr16Launch* find_corr16(int cost, int kx, int ky) {
  for (const Corr16Launch& l : kCorr16)
    if (l.cost == cost && l.kx == kx && l.ky == ky) return &l;
  return nullptr;
}

U16Geom corr16_geom(int kx, int sx) {
  U16Geom g;
  g.sx = sx;
  g.nbx = UTW + sx - 1;
  g.urp = g.nbx + kx + 1 + 16;                                   // + the pairs a clamped tail quad may touch
  g.rpd = (g.urp + 1) / 2 + 2;
  return g;
}

size_t corr16_lds(const Corr16Launch& l, const U16Geom& g) {
  const int nr = l.ty + l.ky - 1;
  const size_t xr = std::max((size_t)nr * g.rpd, (size_t)l.ty * g.nbx);
  return ((size_t)nr * g.urp + xr + (size_t)nr * (UTW / 2 + l.kx / 2 + 2)) * sizeof(u32);
}

}  // namespace

bool vwgpu_bm_corr_u16_supported(int cost_type, int kx, int ky, int sx, int sy) {
  const Corr16Launch* l = find_corr16(cost_type, kx, ky);
  if (!l || sy != 1 || sx > 256) return false;
  return corr16_lds(*l, corr16_geom(kx, sx)) <= 80 * 1024;       // two workgroups per CU
}

int vwgpu_launch_bm_corr_u16(vwgpu_ctx* ctx, int cost_type, const float* left, int lw, int lh, ptrdiff_t ls,
                             const float* right, int rw, int rh, ptrdiff_t rs, int kx, int ky, int sx, int sy,
                             int32_t* out, ptrdiff_t os, int** d_fallback_flag) {
  (void)rw; (void)rh; (void)sy;
  const Corr16Launch* l = find_corr16(cost_type, kx, ky);
  if (!l) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "no packed-u16 SSD / NCC kernel for %dx%d", kx, ky);
  const int ow = lw - kx + 1, oh = lh - ky + 1;
  const int rcw = lw + sx - 1, rch = lh;
  const U16Geom g = corr16_geom(kx, sx);
  int* flag_set = nullptr; int* flag_clear = nullptr;
  int rc = vwgpu_next_flags(ctx, 0, &flag_set, &flag_clear, nullptr);
  if (rc) return rc;
  *d_fallback_flag = flag_set;
  const bool ncc = cost_type == VWGPU_CROSS_CORRELATION;
  u32 *a2 = nullptr, *b2 = nullptr, *full_list = nullptr, *full_count = nullptr;
  const int b2w = rcw - kx + 1;
  const u32 cap = (u32)std::max<size_t>(4096, (size_t)ow * oh / 32);     // as bm_corr_u8.hip
  if (ncc) {
    if ((size_t)ow * oh >= 0xffffffffull) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "bm_corr_u16: image too large");
    const size_t na = vwgpu_align_up((size_t)ow * oh * 4, 256), nb = vwgpu_align_up((size_t)b2w * oh * 4, 256),
                 nl = vwgpu_align_up((size_t)cap * 4 + 256, 256);
    rc = vwgpu_arena_reserve(ctx, &ctx->scratch, na + nb + nl);
    if (rc) return rc;
    char* base = static_cast<char*>(ctx->scratch.base);
    a2 = reinterpret_cast<u32*>(base); b2 = reinterpret_cast<u32*>(base + na);
    full_count = reinterpret_cast<u32*>(base + na + nb); full_list = full_count + 64;
    VWGPU_HIP(ctx, hipMemsetAsync(full_count, 0, 4, ctx->stream));
  }
  const size_t shmem = corr16_lds(*l, g);
  if (shmem > 64 * 1024)
    VWGPU_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(l->fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  {
    vwgpu_prof_scope ps(ctx, "bm_corr_u16");
    hipLaunchKernelGGL(l->fn, dim3((ow + UTW - 1) / UTW, (oh + l->ty - 1) / l->ty), dim3(UTHREADS), shmem, ctx->stream,
                       left, ls, lw, lh, right, rs, rcw, rch, g, out, os, ow, oh, flag_set, flag_clear, a2, b2, b2w, full_list, full_count, cap);
  }
  VWGPU_HIP(ctx, hipGetLastError());
  if (ncc) return vwgpu_launch_ncc_full(ctx, left, ls, right, rs, kx, ky, sx, a2, b2, b2w, out, os, ow, flag_set, full_list, full_count, cap);
  return VWGPU_OK;
}
