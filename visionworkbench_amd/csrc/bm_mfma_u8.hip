// bm_mfma_u8.hip — SSD and NCC block matching for integer-valued imagery in [0,255] with the products on the matrix cores.
//
// Same domain, same results and same protocol as bm_corr_u8.hip (best_of_search_convolution + fast_box_sum + SquaredCost / NCCCost,
// src/vw/Stereo/Correlation.cc:33-137, Algorithms.h:43-129, CostFunctions.h:94-141,179-236: on bytes every sum the reference forms in
// float64 is an exact integer).  What changes is who multiplies.  The bilinear term S(x, y, d) = sum_window L * R is, for 16 columns x
// 16 disparities x ONE image row, a product of two structured matrices:
//     C[d][x] += A[d][k] * B[k][x],   A[d][k] = R(x0 + d0 + d + k)                      a Hankel matrix of the right row,
//                                     B[k][x] = L(x0 + k) if 0 <= k - x < kx, else 0    the banded left row (16 + kx - 1 <= 32 columns)
// — one v_mfma_i32_16x16x32_i8 (8192 multiplies, 34 % of them useful at kx = 11) instead of 768 lane-slots of v_dot4_u32_u8; the accumulator
// is chained down the rows, a ring of ky + 1 accumulators gives the window sum P[r] - P[r-ky] (tools/ubench_mfma_corr.hip measured the two
// inner loops: 0.105 vs 0.164 clk per evaluation and CU, identical sums).  The instruction multiplies SIGNED bytes, so both images are
// centred (v - 128 = v ^ 0x80):
//   SSD   is invariant under a common shift: (B2' + 16384 n - 2 S') on the centred values orders the disparities of a pixel exactly as
//         A2 + B2 - 2 S does and is equal for two disparities iff that is; key = cost << 8 | d as in bm_corr_u8.hip;
//   NCC   needs the true S = S' + 128 (sum_window L + sum_window R) - 16384 n: the right part comes from a table next to 1 / sqrt(B2), the left
//         part is a per-pixel constant; then the two-largest-keys scheme of bm_corr_u8.hip unchanged (same fp32 scores, same queue for
//         ncc_full_kernel).
//   layout  A: lane l holds row l % 16 (a disparity), bytes k = 8 (l / 16) + 0..7 — 8 bytes at offset d + 8 kg of the word-at-every-byte array
//           (one ds_read2_b32); B: lane l holds column l % 16 (a pixel), the same k — two aligned words of the left row under a per-lane band
//           mask; C: lane l holds column l % 16, rows 4 (l / 16) + 0..3: FOUR DISPARITIES OF ONE PIXEL, so the per-pixel state (best / worst
//           key, or best / runner-up) is a register per row and the four lane groups are merged once per 16 columns.
//   mapping workgroup = 256 columns x TY rows (staging and tables as bm_corr_u8.hip); wave w owns columns [64 w, 64 w + 64) in 4 blocks of 16;
//           two disparity tiles are interleaved (two independent accumulator chains).
// Inputs outside the domain, or an all-zero window under NCC, raise the device flag; the caller then runs the float64 kernel.
// One search row (sy == 1), windows up to 17 wide.
//
// MEASURED (4096^2 x 129, tools/time_corr.py): identical results (tests/test_bm_gpu.py runs every case through both kernels) and SLOWER than
// bm_corr_u8.hip — 1.13 ms (SSD 7x7), 1.13 (NCC 7x7), 1.20 (SSD 11x11), 1.16 (NCC 11x11) against 0.69 / 0.83 / 0.78 / 0.94: the time does not
// depend on the window, the products are not what it spends.  A v_mfma_i32_16x16x32_i8 occupies the matrix pipe for 8 passes and an
// accumulator chain is a dependent instruction per row (two interleaved chains: ~66 clk per instruction in the loop of
// tools/ubench_mfma_corr.hip = 6.5 slot-equivalents per evaluation, twice what the v_dot4 chains with shared words cost at 7x7), the ring of
// ky + 1 accumulators takes 96 registers of the 256, and the finishing (table reads at lane-dependent offsets, key, tracking) is the same ~6
// slots per evaluation in a layout that holds four disparities per lane.  Kept behind VWGPU_OPT_CORR_MFMA = 1 as the measured experiment.
//
// Roofline: HBM bound by the task's definition (20 B per output pixel); matrix-pipe latency and finishing bound in fact.
#include <algorithm>
#include <cmath>

#include "vwgpu_internal.h"
#include "u8_tile.h"

namespace {

using namespace vwgpu_u8;
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int MTW = 256;          // output columns per workgroup
constexpr int MTHREADS = 256;

struct MGeom {
  int sx;
  int ndt;       // disparity tiles of 16, rounded up to an even count (they run in pairs)
  int nbx;       // right window origins per row = MTW + sx - 1
  int rwd;       // aligned dwords per staged right row
  int urp;       // dwords per row of the every-byte word array
};

__device__ __forceinline__ u32 umin3(u32 a, u32 b, u32 c) { u32 r; asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ u32 umax3(u32 a, u32 b, u32 c) { u32 r; asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ u32 umed3(u32 a, u32 b, u32 c) { u32 r; asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

template <int COST, int KX, int KY, int TY>
__global__ void __launch_bounds__(MTHREADS, 2)
bm_mfma_u8_kernel(const float* __restrict__ L, ptrdiff_t ls, int lw, int lh,
                  const float* __restrict__ R, ptrdiff_t rs, int rcw, int rch, MGeom g,
                  int32_t* __restrict__ out, ptrdiff_t os, int ow, int oh,
                  int* __restrict__ flag_set, int* __restrict__ flag_clear,
                  u32* __restrict__ a2img, u32* __restrict__ b2img, int b2w,
                  u32* __restrict__ full_list, u32* __restrict__ full_count, u32 cap) {
  constexpr bool NCC = (COST == VWGPU_CROSS_CORRELATION);
  constexpr int NW = (KX + 3) / 4, NR = TY + KY - 1;
  constexpr int LWD = MTW / 4 + 8;                               // aligned dwords per staged left row
  constexpr u32 KMASK = (KX % 4 == 0) ? 0xffffffffu : ((1u << (8 * (KX % 4))) - 1u);
  constexpr int NWIN = KX * KY;
  constexpr u32 OFFK = 16384u * NWIN;                            // >= A2' (centred): B2' - 2 S' + OFFK >= 0, < 2^24 for n <= 206
  static_assert(KX + 15 <= 32, "the banded left operand must fit the 32 columns of one instruction");
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  const int sx = g.sx, nbx = g.nbx, RWD = g.rwd, URP = g.urp;
  u32* UR = lds;                                                 // [NR][URP]  CENTRED word at every byte offset of the right rows
  u32* XR = UR + (size_t)NR * URP;                               // [NR][RWD] aligned right words (staging), then the tables over [TY][nbx]
  const size_t tab_dw = (size_t)(NCC ? 2 : 1) * TY * nbx;
  const size_t xr_dw = (size_t)NR * RWD > tab_dw ? (size_t)NR * RWD : tab_dw;
  u32* LW = XR + xr_dw;                                          // [NR][LWD] aligned left words (raw)
  int* CL = reinterpret_cast<int*>(LW + (size_t)NR * LWD);       // NCC: [TY][MTW]  128 * sum_window L - 16384 n
  const int tid = threadIdx.x;
  const int X0 = blockIdx.x * MTW, y0 = blockIdx.y * TY;
  if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) *flag_clear = 0;     // the NEXT call's flag

  // ---- stage both tiles as packed u8 (checks the domain) ----
  u32 bad_acc = 0;
  stage_u8_rows<NR>(L, ls, lw, lh, X0, y0, LWD, LWD, LW, tid, MTHREADS, bad_acc);
  stage_u8_rows<NR>(R, rs, rcw, rch, X0, y0, RWD, RWD, XR, tid, MTHREADS, bad_acc);
  __syncthreads();

  // ---- NCC: A2 and the left part of the centring correction, lane <-> column ----
  bool zero_window = false;
  if (NCC) {
    const int x = X0 + tid;
    const int w0 = tid >> 2, sh = tid & 3;
    u32 h2[NR], h1[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      u32 a[NW + 1];
#pragma unroll
      for (int n = 0; n <= NW; ++n) a[n] = LW[r * LWD + w0 + n];
      u32 s2 = 0, s1 = 0;
#pragma unroll
      for (int n = 0; n < NW; ++n) {
        u32 w = __builtin_amdgcn_alignbyte(a[n + 1], a[n], sh);
        if (n == NW - 1) w &= KMASK;
        s2 = __builtin_amdgcn_udot4(w, w, s2, false);
        s1 = __builtin_amdgcn_udot4(w, 0x01010101u, s1, false);
      }
      h2[r] = s2; h1[r] = s1;
    }
    u32 a2 = 0, sl = 0;
#pragma unroll
    for (int r = 0; r < KY - 1; ++r) { a2 += h2[r]; sl += h1[r]; }
#pragma unroll
    for (int y = 0; y < TY; ++y) {
      a2 += h2[y + KY - 1]; sl += h1[y + KY - 1];
      if (x < ow && y0 + y < oh) { a2img[(size_t)(y0 + y) * ow + x] = a2; zero_window |= (a2 == 0); }
      CL[y * MTW + tid] = (int)(128u * sl) - 16384 * NWIN;
      a2 -= h2[y]; sl -= h1[y];
    }
  }
  // ---- centred every-byte word array of the right rows ----
  for (int i = tid; i < NR * URP; i += MTHREADS) {
    const int r = i / URP, b = i - r * URP, w = b >> 2;
    UR[i] = __builtin_amdgcn_alignbyte(XR[r * RWD + w + 1], XR[r * RWD + w], b & 3) ^ 0x80808080u;
  }
  __syncthreads();                                               // UR complete, aligned right words dead
  // ---- tables over [TY][nbx]: SSD key part (B2' + OFFK) << 8; NCC {fp32 1 / sqrt(B2), 128 * sum_window R} ----
  u32* TAB = XR;
  for (int xp = tid; xp < nbx; xp += MTHREADS) {
    u32 h2[NR], h1[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      u32 s2 = 0, s1 = 0;
#pragma unroll
      for (int n = 0; n < NW; ++n) {
        u32 v = UR[r * URP + xp + 4 * n] ^ 0x80808080u;          // raw bytes again
        if (n == NW - 1) v &= KMASK;
        s2 = __builtin_amdgcn_udot4(v, v, s2, false);
        s1 = __builtin_amdgcn_udot4(v, 0x01010101u, s1, false);
      }
      h2[r] = s2; h1[r] = s1;
    }
    u32 b2 = 0, sr = 0;
#pragma unroll
    for (int r = 0; r < KY - 1; ++r) { b2 += h2[r]; sr += h1[r]; }
#pragma unroll
    for (int y = 0; y < TY; ++y) {
      b2 += h2[y + KY - 1]; sr += h1[y + KY - 1];
      if (NCC) {
        TAB[2 * (y * nbx + xp)] = __float_as_uint((float)(1.0 / sqrt((double)b2)));
        TAB[2 * (y * nbx + xp) + 1] = 128u * sr;
        const bool inside = (X0 + xp < rcw - KX + 1) && (y0 + y < rch - KY + 1);
        if (inside) { b2img[(size_t)(y0 + y) * b2w + X0 + xp] = b2; zero_window |= (b2 == 0); }
      } else {
        TAB[y * nbx + xp] = (b2 - 256u * sr + 16384u * NWIN + OFFK) << 8;       // B2' = sum (R - 128)^2
      }
      b2 -= h2[y]; sr -= h1[y];
    }
  }
  __syncthreads();

  // ---- the disparity sweep of this wave's four 16-column blocks ----
  const int lane = tid & 63, wave = tid >> 6;
  const int mn = lane & 15, kg = lane >> 4;
  unsigned long long band = 0;                                   // byte b of the lane's B operand lies inside the window of column mn
#pragma unroll
  for (int b = 0; b < 8; ++b) { const int k = 8 * kg + b; if (k - mn >= 0 && k - mn < KX) band |= 0xffull << (8 * b); }
  const int ndt = g.ndt;
  for (int xb = wave * 4; xb < wave * 4 + 4; ++xb) {
    const int x0 = xb * 16;
    long Bop[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const u32* p = LW + r * LWD + x0 / 4 + 2 * kg;
      Bop[r] = (long)(((((unsigned long long)p[1] << 32) | p[0]) ^ 0x8080808080808080ull) & band);
    }
    // per-pixel state of this lane's pixel (column x0 + mn) over the disparities 4 kg + i of every tile
    u32 S1[TY], S2[TY];                                          // SSD: smallest / largest key; NCC: largest / second largest key
    int cl[TY];
#pragma unroll
    for (int y = 0; y < TY; ++y) {
      S1[y] = NCC ? 0u : 0xffffffffu; S2[y] = 0u;
      cl[y] = NCC ? CL[y * MTW + x0 + mn] : 0;
    }
    const u32* ua = UR + x0 + mn + 8 * kg;
    const u32* tb = TAB + (NCC ? 2 : 1) * (x0 + mn + 4 * kg);
    for (int dt = 0; dt < ndt; dt += 2) {
      const bool full = (dt + 2) * 16 <= sx;                     // wave-uniform: every disparity of both tiles exists
      v4i C[2][KY + 1];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const u32* pa = ua + r * URP + (dt + h) * 16;
          const long Aop = (long)(((unsigned long long)pa[4] << 32) | pa[0]);
          const v4i zero = {0, 0, 0, 0};
          C[h][r % (KY + 1)] = __builtin_amdgcn_mfma_i32_16x16x32_i8(Aop, Bop[r], r == 0 ? zero : C[h][(r + KY) % (KY + 1)], 0, 0, 0);
        }
        if (r >= KY - 1) {
          const int y = r - (KY - 1);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int dbase = (dt + h) * 16 + 4 * kg;
            const u32* te = tb + (NCC ? 2 : 1) * (y * nbx + (dt + h) * 16);
            u32 key[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int sp = r >= KY ? C[h][r % (KY + 1)][i] - C[h][(r + 1) % (KY + 1)][i] : C[h][r % (KY + 1)][i];
              const int d = dbase + i;
              if (NCC) {
                const int s = sp + (int)te[2 * i + 1] + cl[y];   // the true sum_window L * R
                const float v = (float)s * __uint_as_float(te[2 * i]);
                key[i] = (__float_as_uint(v) & 0xffffff00u) | (u32)(255 - d);
                if (!full && d >= sx) key[i] = 0u;
              } else {
                key[i] = te[i] + (u32)d - ((u32)sp << 9);        // ((B2' + OFFK - 2 S') << 8) | d
              }
            }
            if (NCC) {
#pragma unroll
              for (int i = 0; i < 4; ++i) { S2[y] = umed3(S1[y], S2[y], key[i]); S1[y] = S1[y] > key[i] ? S1[y] : key[i]; }
            } else if (full) {
              S1[y] = umin3(umin3(S1[y], key[0], key[1]), key[2], key[3]);
              S2[y] = umax3(umax3(S2[y], key[0], key[1]), key[2], key[3]);
            } else {
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (dbase + i < sx) { S1[y] = S1[y] < key[i] ? S1[y] : key[i]; S2[y] = S2[y] > key[i] ? S2[y] : key[i]; }
            }
          }
        }
        if (r % 2 == 1) __builtin_amdgcn_sched_barrier(0);       // (keeps the operand reads of later rows from piling up in registers)
      }
    }
    // merge the four lane groups (the disparities 4 kg + i of every tile), then lane group 0 finishes the 16 pixels
    const int x = X0 + x0 + mn;
#pragma unroll
    for (int y = 0; y < TY; ++y) {
#pragma unroll
      for (int o = 16; o <= 32; o <<= 1) {
        const u32 o1 = (u32)__shfl_xor((int)S1[y], o), o2 = (u32)__shfl_xor((int)S2[y], o);
        if (NCC) {
          const u32 hi = S1[y] > o1 ? S1[y] : o1, lo = S1[y] > o1 ? o1 : S1[y];
          S2[y] = umax3(lo, S2[y], o2); S1[y] = hi;
        } else {
          S1[y] = S1[y] < o1 ? S1[y] : o1; S2[y] = S2[y] > o2 ? S2[y] : o2;
        }
      }
      if (kg == 0 && x < ow && y0 + y < oh) {
        int32_t* o = out + ((ptrdiff_t)(y0 + y) * os + x) * 3;
        if (!NCC) {
          o[0] = (int32_t)(S1[y] & 0xffu); o[1] = 0;
          o[2] = ((S1[y] >> 8) == (S2[y] >> 8)) ? 0 : 0x7fffffff;   // best == worst (Correlation.cc:121-133)
        } else {
          // as bm_corr_u8.hip: the runner-up within 2^-13 of the winner (or a single disparity) -> the float64 sequence decides
          const float m1 = __uint_as_float(S1[y] & 0xffffff00u), m2 = __uint_as_float(S2[y] & 0xffffff00u);
          if (sx == 1 || m2 >= m1 * 0.99987793f) {
            const u32 i = atomicAdd(full_count, 1u);
            if (i < cap) full_list[i] = (u32)((size_t)(y0 + y) * ow + x);
          } else {
            o[0] = (int32_t)(255u - (S1[y] & 0xffu)); o[1] = 0; o[2] = 0x7fffffff;
          }
        }
      }
    }
  }
  if (__syncthreads_or(bad_acc != 0u || zero_window) && tid == 0) atomicOr(flag_set, 1);
}

typedef void (*MfmaFn)(const float*, ptrdiff_t, int, int, const float*, ptrdiff_t, int, int, MGeom, int32_t*, ptrdiff_t, int, int, int*, int*,
                       u32*, u32*, int, u32*, u32*, u32);
struct MfmaLaunch { int cost, kx, ky, ty; MfmaFn fn; };
#define VW_MF(C, KX, KY, TY) MfmaLaunch{C, KX, KY, TY, bm_mfma_u8_kernel<C, KX, KY, TY>}
// rows per workgroup: SSD keeps the 16 of bm_corr_u8.hip; NCC has two table words per (row, column) and a per-pixel constant in LDS: 8 rows
const MfmaLaunch kMfma[] = {
    VW_MF(VWGPU_SQUARED_DIFFERENCE, 3, 3, 16), VW_MF(VWGPU_SQUARED_DIFFERENCE, 5, 5, 16), VW_MF(VWGPU_SQUARED_DIFFERENCE, 7, 7, 16),
    VW_MF(VWGPU_SQUARED_DIFFERENCE, 9, 9, 16), VW_MF(VWGPU_SQUARED_DIFFERENCE, 11, 11, 16),
    VW_MF(VWGPU_CROSS_CORRELATION, 3, 3, 8), VW_MF(VWGPU_CROSS_CORRELATION, 5, 5, 8), VW_MF(VWGPU_CROSS_CORRELATION, 7, 7, 8),
    VW_MF(VWGPU_CROSS_CORRELATION, 9, 9, 8), VW_MF(VWGPU_CROSS_CORRELATION, 11, 11, 8),
};
#undef VW_MF

const MfmaLaunch* find_mfma(int cost, int kx, int ky) {
  for (const MfmaLaunch& l : kMfma)
    if (l.cost == cost && l.kx == kx && l.ky == ky) return &l;
  return nullptr;
}

MGeom mfma_geom(int sx) {
  MGeom g;
  g.sx = sx;
  g.ndt = ((sx + 15) / 16 + 1) & ~1;
  g.nbx = MTW + sx - 1;
  g.urp = MTW + 16 * g.ndt + 32;                                 // the last tile's operand reads end at 240 + 16 ndt + 15 + 24 + 4 bytes
  g.rwd = g.urp / 4 + 2;
  return g;
}

size_t mfma_lds_bytes(const MfmaLaunch& l, const MGeom& g) {
  const bool ncc = l.cost == VWGPU_CROSS_CORRELATION;
  const int nr = l.ty + l.ky - 1;
  const size_t tab = (size_t)(ncc ? 2 : 1) * l.ty * g.nbx;
  const size_t xr = std::max((size_t)nr * g.rwd, tab);
  return ((size_t)nr * g.urp + xr + (size_t)nr * (MTW / 4 + 8) + (ncc ? (size_t)l.ty * MTW : 0)) * sizeof(u32);
}

}  // namespace

bool vwgpu_bm_mfma_u8_supported(int cost_type, int kx, int ky, int sx, int sy) {
  const MfmaLaunch* l = find_mfma(cost_type, kx, ky);
  if (!l || sy != 1 || sx > 240) return false;
  return mfma_lds_bytes(*l, mfma_geom(sx)) <= 80 * 1024;          // two workgroups per CU
}

int vwgpu_launch_bm_mfma_u8(vwgpu_ctx* ctx, int cost_type, const float* left, int lw, int lh, ptrdiff_t ls,
                            const float* right, int rw, int rh, ptrdiff_t rs, int kx, int ky, int sx, int sy,
                            int32_t* out, ptrdiff_t os, int** d_fallback_flag) {
  (void)rw; (void)rh; (void)sy;
  const MfmaLaunch* l = find_mfma(cost_type, kx, ky);
  if (!l) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "no matrix-core SSD / NCC kernel for %dx%d", kx, ky);
  const int ow = lw - kx + 1, oh = lh - ky + 1;
  const int rcw = lw + sx - 1, rch = lh;
  const MGeom g = mfma_geom(sx);
  int* flag_set = nullptr; int* flag_clear = nullptr;
  int rc = vwgpu_next_flags(ctx, 0, &flag_set, &flag_clear, nullptr);
  if (rc) return rc;
  *d_fallback_flag = flag_set;
  const bool ncc = cost_type == VWGPU_CROSS_CORRELATION;
  u32 *a2 = nullptr, *b2 = nullptr, *full_list = nullptr, *full_count = nullptr;
  const int b2w = rcw - kx + 1;
  const u32 cap = (u32)std::max<size_t>(4096, (size_t)ow * oh / 32);     // as bm_corr_u8.hip
  if (ncc) {
    if ((size_t)ow * oh >= 0xffffffffull) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "bm_mfma_u8: image too large");
    const size_t na = vwgpu_align_up((size_t)ow * oh * 4, 256), nb = vwgpu_align_up((size_t)b2w * oh * 4, 256),
                 nl = vwgpu_align_up((size_t)cap * 4 + 256, 256);
    rc = vwgpu_arena_reserve(ctx, &ctx->scratch, na + nb + nl);
    if (rc) return rc;
    char* base = static_cast<char*>(ctx->scratch.base);
    a2 = reinterpret_cast<u32*>(base); b2 = reinterpret_cast<u32*>(base + na);
    full_count = reinterpret_cast<u32*>(base + na + nb); full_list = full_count + 64;
    VWGPU_HIP(ctx, hipMemsetAsync(full_count, 0, 4, ctx->stream));
  }
  const size_t shmem = mfma_lds_bytes(*l, g);
  if (shmem > 64 * 1024)
    VWGPU_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(l->fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  {
    vwgpu_prof_scope ps(ctx, "bm_mfma_u8");
    hipLaunchKernelGGL(l->fn, dim3((ow + MTW - 1) / MTW, (oh + l->ty - 1) / l->ty), dim3(MTHREADS), shmem, ctx->stream,
                       left, ls, lw, lh, right, rs, rcw, rch, g, out, os, ow, oh, flag_set, flag_clear, a2, b2, b2w, full_list, full_count, cap);
  }
  VWGPU_HIP(ctx, hipGetLastError());
  if (ncc) return vwgpu_launch_ncc_full(ctx, left, ls, right, rs, kx, ky, sx, a2, b2, b2w, out, os, ow, flag_set, full_list, full_count, cap);
  return VWGPU_OK;
}
