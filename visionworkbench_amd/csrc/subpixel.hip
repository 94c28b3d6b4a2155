// subpixel.hip — parabola sub-pixel refinement (vw::stereo::ParabolaSubpixelView::evaluate,
// src/vw/Stereo/ParabolaSubpixelView.cc:31-274).
//
// For each pixel with integer disparity D the reference gathers the 3x3 patch of SAD costs at D + {-1,0,1}^2
// (always AbsoluteCost, :49-51; float |a-b| summed in float64 by fast_box_sum and stored as float, :187-205), fits a
// 2-D parabola with the pre-computed pseudo-inverse (ParabolaSubpixelView.h:83-88) and moves the disparity by
// ((c*e - 2*b*d)/den, (c*d - 2*a*e)/den), den = 4ab - c^2, if the shift is shorter than 5 px (:242-257).
// The reference reaches the per-pixel costs through zones of similar disparity (:72-218); the costs themselves are
// per-pixel quantities, so this kernel computes them directly: one thread per pixel, nine windowed SADs in float64
// over pre-filtered, pre-cropped rasters (the same `left_raster` / `right_raster` the reference rasterises, :44-45).
// The float solve below is the reference's expression order with contraction off.
// Exact on integer-valued imagery with PREFILTER_NONE (all sums exact); after LoG / mean-subtraction the reference's
// running sums are position dependent and agreement is to float rounding (tests: 1e-5 abs).
//
// Roofline: 9*kx*ky abs-diffs per pixel against 36 B of compulsory traffic per pixel (disparity in, two images,
// disparity out) — VALU / L1 bound; the windows of neighbouring pixels overlap and are served by the vector L1.
#include <algorithm>
#include <climits>

#include "vwgpu_internal.h"

namespace {

// get_disparity_range over ALL pixels (invalid ones included, src/vw/Stereo/DisparityMap.h:52-66) of the disparity truncated to int
// (the PixelMask<Vector2f> -> PixelMask<Vector2i> conversion of ParabolaSubpixelView.cc:283): parabola_prepass_kernel, blockIdx.z = 0.

// The class of the imagery in the SAME launch as the disparity range (blockIdx.z = 1, 2: the left / the right image): lowest set mantissa
// bit, largest exponent, "non-finite" (bit 0) and "negative" (bit 1) over all pixels, as float_grain_kernel (bm_exact.hip) measures them —
// cell[0] = min over non-zero pixels of the exponent of the lowest set bit, cell[1] = max exponent, cell[2] = flags.  Separate launches
// of the two measurements took 82 + 50 us at 4096^2 (each well under the HBM rate); together they overlap.
__device__ __forceinline__ void grain_take(float v, int& lo, int& hi, int& bad) {
  const unsigned u = __float_as_uint(v);
  const int e = (int)((u >> 23) & 0xffu);
  unsigned m = u & 0x7fffffu;
  if (e == 0xff) { bad |= 1; return; }
  if (e == 0 && m == 0) return;
  if (u >> 31) bad |= 2;
  int base;
  if (e == 0) base = -149; else { m |= 0x800000u; base = e - 127 - 23; }
  lo = min(lo, base + (__ffs((int)m) - 1));
  hi = max(hi, base + (31 - __clz((int)m)));
}
__global__ void __launch_bounds__(256)
parabola_prepass_kernel(const float* __restrict__ d, int w, int h, ptrdiff_t stride_px, int* __restrict__ out4,
                        const float* __restrict__ L, int lw, int lh, ptrdiff_t ls, const float* __restrict__ R, int rw, int rh, ptrdiff_t rs,
                        int* __restrict__ cell) {
  __shared__ int part[4][4];
  const int t = threadIdx.y * blockDim.x + threadIdx.x;
  const int xs = gridDim.x * blockDim.x, xf = blockIdx.x * blockDim.x + threadIdx.x;
  if (blockIdx.z == 0) {
    // A row of the disparity is 3 w floats {dx, dy, valid, dx, ...}: four pixels = three float4 (dx at elements 0, 3, 6, 9, dy at 1, 4, 7,
    // 10), two such groups in flight per thread where the rows are 16-byte aligned; pixel by pixel otherwise.
    int mnx = INT_MAX, mny = INT_MAX, mxx = INT_MIN, mxy = INT_MIN;
    const bool vec = ((reinterpret_cast<uintptr_t>(d) & 15) == 0) && (stride_px % 4 == 0);
    const int n12 = vec ? w / 4 : 0;                            // whole groups of four pixels
    auto tx = [&](float v) __attribute__((always_inline)) { const int iv = (int)v; mnx = min(mnx, iv); mxx = max(mxx, iv); };
    auto ty = [&](float v) __attribute__((always_inline)) { const int iv = (int)v; mny = min(mny, iv); mxy = max(mxy, iv); };
    // (the loads in flight are two ROWS of the thread's column group — a 4096-wide row is one group per thread of the 1024-wide grid row)
    const int ys = gridDim.y * blockDim.y;
    for (int y = blockIdx.y * blockDim.y + threadIdx.y; y < h; y += 2 * ys) {
      const int y1 = y + ys < h ? y + ys : y;
      const float* row = d + (ptrdiff_t)y * stride_px * 3;
      const float* rowb = d + (ptrdiff_t)y1 * stride_px * 3;
      const float4* row4 = reinterpret_cast<const float4*>(row);
      const float4* rowb4 = reinterpret_cast<const float4*>(rowb);
      for (int j = xf; j < n12; j += xs) {
        const float4 a0 = row4[3 * j], a1 = row4[3 * j + 1], a2 = row4[3 * j + 2];
        const float4 b0 = rowb4[3 * j], b1 = rowb4[3 * j + 1], b2 = rowb4[3 * j + 2];
        tx(a0.x); ty(a0.y); tx(a0.w); ty(a1.x); tx(a1.z); ty(a1.w); tx(a2.y); ty(a2.z);
        tx(b0.x); ty(b0.y); tx(b0.w); ty(b1.x); tx(b1.z); ty(b1.w); tx(b2.y); ty(b2.z);
      }
      for (int x = 4 * n12 + xf; x < w; x += xs) {
        tx(row[(ptrdiff_t)x * 3]); ty(row[(ptrdiff_t)x * 3 + 1]); tx(rowb[(ptrdiff_t)x * 3]); ty(rowb[(ptrdiff_t)x * 3 + 1]);
      }
    }
    for (int o = 32; o > 0; o >>= 1) {
      mnx = min(mnx, __shfl_xor(mnx, o)); mny = min(mny, __shfl_xor(mny, o));
      mxx = max(mxx, __shfl_xor(mxx, o)); mxy = max(mxy, __shfl_xor(mxy, o));
    }
    // one set of atomics per workgroup, and few workgroups: atomics on the same four words serialise at the L2
    if ((t & 63) == 0) { part[t >> 6][0] = mnx; part[t >> 6][1] = mny; part[t >> 6][2] = mxx; part[t >> 6][3] = mxy; }
    __syncthreads();
    if (t == 0) {
      for (int k = 1; k < 4; ++k) { mnx = min(mnx, part[k][0]); mny = min(mny, part[k][1]); mxx = max(mxx, part[k][2]); mxy = max(mxy, part[k][3]); }
      // Same-address atomics serialise at the L2 (~12 ns each: the 10 k of this launch were most of its time): a workgroup whose values
      // cannot move a word — the words only ever move one way, so a stale read errs on the side of an atomic — skips it.
      if (mnx < __hip_atomic_load(out4 + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(out4 + 0, mnx);
      if (mny < __hip_atomic_load(out4 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(out4 + 1, mny);
      if (mxx > __hip_atomic_load(out4 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out4 + 2, mxx);
      if (mxy > __hip_atomic_load(out4 + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(out4 + 3, mxy);
    }
    return;
  }
  if (!cell) return;
  const float* img = blockIdx.z == 1 ? L : R;
  const int iw = blockIdx.z == 1 ? lw : rw, ih = blockIdx.z == 1 ? lh : rh;
  const ptrdiff_t is = blockIdx.z == 1 ? ls : rs;
  int lo = INT_MAX, hi = INT_MIN, bad = 0;
  const bool vec = ((reinterpret_cast<uintptr_t>(img) & 15) == 0) && (is % 4 == 0);
  const int n4 = iw / 4;
  const int ys = gridDim.y * blockDim.y;
  for (int y = blockIdx.y * blockDim.y + threadIdx.y; y < ih; y += 4 * ys) {
    const float* rows[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) rows[k] = img + (ptrdiff_t)(y + k * ys < ih ? y + k * ys : y) * is;      // four rows in flight
    if (vec) {
      for (int i = xf; i < n4; i += xs) {
        float4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = reinterpret_cast<const float4*>(rows[k])[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) { grain_take(v[k].x, lo, hi, bad); grain_take(v[k].y, lo, hi, bad); grain_take(v[k].z, lo, hi, bad); grain_take(v[k].w, lo, hi, bad); }
      }
      for (int x = 4 * n4 + xf; x < iw; x += xs)
#pragma unroll
        for (int k = 0; k < 4; ++k) grain_take(rows[k][x], lo, hi, bad);
    } else {
      for (int x = xf; x < iw; x += xs)
#pragma unroll
        for (int k = 0; k < 4; ++k) grain_take(rows[k][x], lo, hi, bad);
    }
  }
  for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); bad |= __shfl_xor(bad, o); }
  if ((t & 63) == 0) { part[t >> 6][0] = lo; part[t >> 6][1] = hi; part[t >> 6][2] = bad; }
  __syncthreads();
  if (t == 0) {
    for (int k = 1; k < 4; ++k) { lo = min(lo, part[k][0]); hi = max(hi, part[k][1]); bad |= part[k][2]; }
    if (lo != INT_MAX) {
      if (lo < __hip_atomic_load(&cell[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&cell[0], lo);
      if (hi > __hip_atomic_load(&cell[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&cell[1], hi);
    }
    if (bad & ~__hip_atomic_load(&cell[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&cell[2], bad);
  }
}

// KX > 0: the window width is a compile-time constant and the nine costs are formed in one sweep over the (ky + 2) rows of
// the right neighbourhood: every right value is loaded once (kx + 2 per row) and every left value once (three rows kept in
// registers), instead of 9 * kx * ky loads of each.  The float64 accumulation order of each of the nine sums is unchanged
// (rows outer, columns inner).  KX == 0: any width, the plain loops.
// INT: every pixel of both rasters is an integer of magnitude < 2^21 (measured by the caller): the nine sums are then exact
// integers below 2^31 in any arithmetic, so they are formed with v_sad_u32 (one instruction per abs-diff instead of
// subtract + widen + float64 add) and converted once — the same float the reference's float64 sum rounds to.
__device__ __forceinline__ unsigned sad_u32(unsigned a, unsigned b, unsigned c) {
  unsigned r;
  asm("v_sad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

// KY > 0 (round 6, byte form only): the window height is a compile-time constant too — the loop over the right rows is unrolled, the three
// left rows rotate by renaming instead of nine moves per row, the row tests fold (470 -> 4xx us at 11 x 11).
template <int KX, int INT, int KY = 0>      // INT: 0 float64 sums, 1 integers below 2^21 (v_sad_u32), 2 integers in [0,255] (v_sad_u8 on packed bytes)
__global__ void __launch_bounds__(256)
parabola_kernel(const float* __restrict__ disp, int w, int h, ptrdiff_t dstride_px,
                const float* __restrict__ lras, int lrw, const float* __restrict__ rras, int rrw,
                int range_minx, int range_miny, int kx, int ky,
                float* __restrict__ out, ptrdiff_t ostride_px) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  const float* dp = disp + ((ptrdiff_t)y * dstride_px + x) * 3;
  float* op = out + ((ptrdiff_t)y * ostride_px + x) * 3;
  if (dp[2] == 0.0f) { op[0] = 0.0f; op[1] = 0.0f; op[2] = 0.0f; return; }   // PixelMask<Vector2f>()  (:262-264)
  const int Dx = (int)dp[0], Dy = (int)dp[1];

  // patch[(dy+1)*3 + (dx+1)] = cost at D + (dx,dy)   (:187-205)
  float patch[9];
  const float* lbase = lras + (ptrdiff_t)y * lrw + x;   // left_region starts at (-hx,-hy): window top-left == (x,y)
  if (KX == 0) {
#pragma unroll
    for (int ddy = -1; ddy <= 1; ++ddy) {
#pragma unroll
      for (int ddx = -1; ddx <= 1; ++ddx) {
        const float* rbase = rras + (ptrdiff_t)(y + Dy + ddy - range_miny) * rrw + (x + Dx + ddx - range_minx);
        double s = 0.0;
        for (int j = 0; j < ky; ++j) {
          const float* lp = lbase + (ptrdiff_t)j * lrw;
          const float* rp = rbase + (ptrdiff_t)j * rrw;
          for (int i = 0; i < kx; ++i) s += (double)fabsf(lp[i] - rp[i]);
        }
        patch[(ddy + 1) * 3 + (ddx + 1)] = (float)s;
      }
    }
  } else if (INT == 2) {
    // Bytes: a row of the window is NW dwords (the last one masked); the three x shifts of the right row are byte alignments
    // of the same kx + 2 bytes.  One v_sad_u8 covers four abs-diffs.
    constexpr int K = KX > 0 ? KX : 1;
    constexpr int NW = (K + 3) / 4, NB = (K + 2 + 3) / 4 + 1;       // dwords of a left row / of the kx + 2 right bytes (+1 for the alignment)
    constexpr unsigned LAST = (K % 4 == 0) ? 0xffffffffu : ((1u << (8 * (K % 4))) - 1u);
    unsigned s9[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) s9[a][b] = 0u;
    // The rasters are bytes here (f32_to_u8_raster_kernel, pitches lrw / rrw in bytes): nw + 1 aligned dwords and one
    // v_alignbyte per dword give the packed bytes starting at any x — 9 loads and 7 ALU ops per row pair of the window instead of
    // 24 float loads and 24 conversions (the float form ran into the L1: 21 GB through it for a 4096^2 image).
    // (the pitches are multiples of 4 — vwgpu_parabola_u8_pitch — so the byte phase of a window is the same in every row: the aligned dword
    // pointer and the shift are formed ONCE per image and the rows are a pitch apart; round 6: the per-row (address & 3, subtract) pair was
    // a seventh of the kernel's instructions)
    auto pack_row = [&](const unsigned* a, unsigned sh, unsigned* w, int nw) __attribute__((always_inline)) {
      unsigned raw[NB + 1];
#pragma unroll
      for (int j = 0; j <= nw; ++j) raw[j] = a[j];
#pragma unroll
      for (int j = 0; j < nw; ++j) w[j] = __builtin_amdgcn_alignbyte(raw[j + 1], raw[j], sh);
    };
    const uint8_t* lbase8 = reinterpret_cast<const uint8_t*>(lras) + (ptrdiff_t)y * lrw + x;
    const uint8_t* rras8 = reinterpret_cast<const uint8_t*>(rras);
    const unsigned lsh = (unsigned)(reinterpret_cast<uintptr_t>(lbase8) & 3u);
    const unsigned* const lrow32 = reinterpret_cast<const unsigned*>(lbase8 - lsh);
    const int lpd = lrw >> 2, rpd = rrw >> 2;
    unsigned la[NW], lb[NW], lc[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) { la[j] = 0u; lb[j] = 0u; }
    pack_row(lrow32, lsh, lc, NW);
    lc[NW - 1] &= LAST;
    const uint8_t* rrow8 = rras8 + (ptrdiff_t)(y + Dy - 1 - range_miny) * rrw + (x + Dx - 1 - range_minx);
    const unsigned rsh = (unsigned)(reinterpret_cast<uintptr_t>(rrow8) & 3u);
    const unsigned* rrow = reinterpret_cast<const unsigned*>(rrow8 - rsh);
    const int kyy = KY > 0 ? KY : ky;
    auto row_step = [&](int q) __attribute__((always_inline)) {
      unsigned rb[NB];
      pack_row(rrow, rsh, rb, NB);
      unsigned rs[3][NW];                                  // the window bytes at x shift b = 0, 1, 2
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          rs[b][j] = b == 0 ? rb[j] : __builtin_amdgcn_alignbyte(rb[j + 1], rb[j], b);
          if (j == NW - 1) rs[b][j] &= LAST;
        }
      if (q - 1 >= 0) {
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int j = 0; j < NW; ++j) s9[2][b] = __builtin_amdgcn_sad_u8(la[j], rs[b][j], s9[2][b]);
      }
      if (q >= 0 && q < kyy) {
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int j = 0; j < NW; ++j) s9[1][b] = __builtin_amdgcn_sad_u8(lb[j], rs[b][j], s9[1][b]);
      }
      if (q + 1 < kyy) {
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int j = 0; j < NW; ++j) s9[0][b] = __builtin_amdgcn_sad_u8(lc[j], rs[b][j], s9[0][b]);
      }
#pragma unroll
      for (int j = 0; j < NW; ++j) { la[j] = lb[j]; lb[j] = lc[j]; }
      if (q + 2 < kyy) {
        pack_row(lrow32 + (ptrdiff_t)(q + 2) * lpd, lsh, lc, NW);
        lc[NW - 1] &= LAST;
      }
      rrow += rpd;
    };
    if constexpr (KY > 0) {
#pragma unroll
      for (int q = -1; q <= KY; ++q) row_step(q);
    } else {
      for (int q = -1; q <= ky; ++q) row_step(q);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) patch[a * 3 + b] = (float)s9[a][b];
  } else if (INT == 1) {
    constexpr int K = KX > 0 ? KX : 1;
    constexpr int OFF = 1 << 21;
    unsigned s9[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) s9[a][b] = 0u;
    unsigned la[K], lb[K], lc[K];
#pragma unroll
    for (int i = 0; i < K; ++i) { la[i] = 0u; lb[i] = 0u; lc[i] = (unsigned)((int)lbase[i] + OFF); }
    const float* rrow = rras + (ptrdiff_t)(y + Dy - 1 - range_miny) * rrw + (x + Dx - 1 - range_minx);
    const int kyy = KY > 0 ? KY : ky;
    auto row_step = [&](int q) __attribute__((always_inline)) {
      unsigned r[K + 2];
#pragma unroll
      for (int i = 0; i < K + 2; ++i) r[i] = (unsigned)((int)rrow[i] + OFF);
      if (q - 1 >= 0) {
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int i = 0; i < K; ++i) s9[2][b] = sad_u32(la[i], r[i + b], s9[2][b]);
      }
      if (q >= 0 && q < kyy) {
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int i = 0; i < K; ++i) s9[1][b] = sad_u32(lb[i], r[i + b], s9[1][b]);
      }
      if (q + 1 < kyy) {
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int i = 0; i < K; ++i) s9[0][b] = sad_u32(lc[i], r[i + b], s9[0][b]);
      }
#pragma unroll
      for (int i = 0; i < K; ++i) { la[i] = lb[i]; lb[i] = lc[i]; }
      if (q + 2 < kyy) {
        const float* lp = lbase + (ptrdiff_t)(q + 2) * lrw;
#pragma unroll
        for (int i = 0; i < K; ++i) lc[i] = (unsigned)((int)lp[i] + OFF);
      }
      rrow += rrw;
    };
    if constexpr (KY > 0) {
#pragma unroll
      for (int q = -1; q <= KY; ++q) row_step(q);
    } else {
      for (int q = -1; q <= ky; ++q) row_step(q);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) patch[a * 3 + b] = (float)s9[a][b];
  } else {
    constexpr int K = KX > 0 ? KX : 1;
    double s9[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) s9[a][b] = 0.0;
    float la[K], lb[K], lc[K];                           // left rows q-1, q, q+1
#pragma unroll
    for (int i = 0; i < K; ++i) { la[i] = 0.0f; lb[i] = 0.0f; lc[i] = lbase[i]; }          // q = -1: row q+1 = 0
    const float* rrow = rras + (ptrdiff_t)(y + Dy - 1 - range_miny) * rrw + (x + Dx - 1 - range_minx);
    const int kyy = KY > 0 ? KY : ky;
    auto row_step = [&](int q) __attribute__((always_inline)) {                     // right row y + Dy + q pairs with left row q - ddy
      float r[K + 2];
#pragma unroll
      for (int i = 0; i < K + 2; ++i) r[i] = rrow[i];
      if (q - 1 >= 0) {                                  // ddy = +1: left row q-1
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int i = 0; i < K; ++i) s9[2][b] += (double)fabsf(la[i] - r[i + b]);
      }
      if (q >= 0 && q < kyy) {                            // ddy = 0: left row q
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int i = 0; i < K; ++i) s9[1][b] += (double)fabsf(lb[i] - r[i + b]);
      }
      if (q + 1 < kyy) {                                  // ddy = -1: left row q+1
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int i = 0; i < K; ++i) s9[0][b] += (double)fabsf(lc[i] - r[i + b]);
      }
      // slide the three left rows
#pragma unroll
      for (int i = 0; i < K; ++i) { la[i] = lb[i]; lb[i] = lc[i]; }
      if (q + 2 < kyy) {
        const float* lp = lbase + (ptrdiff_t)(q + 2) * lrw;
#pragma unroll
        for (int i = 0; i < K; ++i) lc[i] = lp[i];
      }
      rrow += rrw;
    };
    if constexpr (KY > 0) {
#pragma unroll
      for (int q = -1; q <= KY; ++q) row_step(q);
    } else {
      for (int q = -1; q <= ky; ++q) row_step(q);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) patch[a * 3 + b] = (float)s9[a][b];
  }

  float rx = (float)Dx, ry = (float)Dy;
  bool all_equal = true;
#pragma unroll
  for (int c = 1; c < 9; ++c) all_equal = all_equal && (patch[c] == patch[c - 1]);
  if (!all_equal) {                                      // std::equal guard (:236-237)
    // pinvA rows a..f (ParabolaSubpixelView.h:83-88), float values of the double literals
    const float s6 = (float)(1.0 / 6), t3 = (float)(-1.0 / 3), q4 = (float)(1.0 / 4), n4 = (float)(-1.0 / 4), n6 = (float)(-1.0 / 6);
    const float A[6][9] = {
        {s6, t3, s6, s6, t3, s6, s6, t3, s6},
        {s6, s6, s6, t3, t3, t3, s6, s6, s6},
        {q4, 0.0f, n4, 0.0f, 0.0f, 0.0f, n4, 0.0f, q4},
        {n6, 0.0f, s6, n6, 0.0f, s6, n6, 0.0f, s6},
        {n6, n6, n6, 0.0f, 0.0f, 0.0f, s6, s6, s6},
        {(float)(-1.0 / 9), (float)(2.0 / 9), (float)(-1.0 / 9), (float)(2.0 / 9), (float)(5.0 / 9), (float)(2.0 / 9),
         (float)(-1.0 / 9), (float)(2.0 / 9), (float)(-1.0 / 9)}};
    float xv[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) {                        // Matrix<float,6,9> * Vector<float,9>: sequential dot_prod
      float acc = 0.0f;
#pragma unroll
      for (int c = 0; c < 9; ++c) acc += A[r][c] * patch[c];
      xv[r] = acc;
    }
    const float denom = 4 * xv[0] * xv[1] - (xv[2] * xv[2]);                 // 4ab - c^2  (:250)
    const float ox = (xv[2] * xv[4] - 2 * xv[1] * xv[3]) / denom;             // (:251)
    const float oy = (xv[2] * xv[3] - 2 * xv[0] * xv[4]) / denom;             // (:252)
    double n2 = 0.0;                                                         // norm_2 (src/vw/Math/Vector.h:1591-1604)
    n2 += ox * ox;
    n2 += oy * oy;
    if (sqrt((double)(float)n2) < 5.0f) { rx += ox; ry += oy; }              // MAX_SUBPIXEL_SHIFT (:253-257)
  }
  op[0] = rx; op[1] = ry; op[2] = 1.0f;
}

}  // namespace

// Disparity range and (d_cell != nullptr) the class of both images in one launch.  d_out4: 4 ints; d_cell: 3 ints, initialised here.
int vwgpu_launch_parabola_prepass(vwgpu_ctx* ctx, const float* disp3f, int w, int h, ptrdiff_t stride_px, int* d_out4,
                                  const float* L, int lw, int lh, ptrdiff_t ls, const float* R, int rw, int rh, ptrdiff_t rs, int* d_cell) {
  const int init[8] = {INT_MAX, INT_MAX, INT_MIN, INT_MIN, INT_MAX, INT_MIN, 0, 0};
  if (d_cell && d_cell != d_out4 + 4) return vwgpu_fail(ctx, VWGPU_ERR_LOGIC, "parabola prepass: the cell follows the range");
  VWGPU_HIP(ctx, hipMemcpyAsync(d_out4, init, d_cell ? sizeof init : 4 * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
  const int mw = std::max(w, std::max(lw, rw)), mh = std::max(h, std::max(lh, rh));
  dim3 blk(64, 4), grd(std::min((mw + 63) / 64, 16), std::min((mh + 3) / 4, 64), d_cell ? 3 : 1);
  vwgpu_prof_scope ps(ctx, "parabola_prepass");
  hipLaunchKernelGGL(parabola_prepass_kernel, grd, blk, 0, ctx->stream, disp3f, w, h, stride_px, d_out4, L, lw, lh, ls, R, rw, rh, rs, d_cell);
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}

// Byte raster of the region [x0, x0 + bw) x [y0, y0 + bh) of a float image of integers in [0,255], constant edge extension — the crop
// the reference's prerasterize takes (ParabolaSubpixelView.cc:302-327) written as bytes directly (the float crop + its conversion
// were two passes: 8 + 5 B per pixel instead of 4 + 1).
__global__ void f32_ext_to_u8_raster_kernel(const float* __restrict__ src, ptrdiff_t stride, int w, int h, int x0, int y0, int bw, int bh,
                                            uint8_t* __restrict__ dst, int pitch) {
  const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x4 >= pitch || y >= bh) return;
  int sy = y0 + y; sy = sy < 0 ? 0 : (sy >= h ? h - 1 : sy);
  const float* s = src + (ptrdiff_t)sy * stride;
  unsigned v = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    int sx = x0 + x4 + e; sx = sx < 0 ? 0 : (sx >= w ? w - 1 : sx);
    v = __builtin_amdgcn_cvt_pk_u8_f32(x4 + e < bw ? s[sx] : 0.0f, e, v);
  }
  *reinterpret_cast<unsigned*>(dst + (size_t)y * pitch + x4) = v;
}
void vwgpu_launch_f32_ext_to_u8_raster(vwgpu_ctx* ctx, const float* src, ptrdiff_t stride, int w, int h, int x0, int y0, int bw, int bh,
                                       uint8_t* dst, int pitch) {
  vwgpu_prof_scope ps(ctx, "parabola_u8_raster");
  hipLaunchKernelGGL(f32_ext_to_u8_raster_kernel, dim3((pitch / 4 + 63) / 64, (bh + 3) / 4), dim3(64, 4), 0, ctx->stream, src, stride, w, h, x0, y0, bw, bh, dst, pitch);
}

int vwgpu_parabola_u8_pitch(int w) { return (w + 3) / 4 * 4 + 32; }

int vwgpu_launch_parabola(vwgpu_ctx* ctx, const float* disp3f, int w, int h, ptrdiff_t dstride_px,
                          const float* lras, int lrw, const float* rras, int rrw, int range_minx, int range_miny,
                          int kx, int ky, float* out3f, ptrdiff_t ostride_px, int integer_class) {
  dim3 blk(64, 4), grd((w + 63) / 64, (h + 3) / 4);
  vwgpu_prof_scope ps(ctx, integer_class == 2 ? "parabola_subpixel_u8" : (integer_class == 1 ? "parabola_subpixel_int" : "parabola_subpixel"));
#define VW_PARABOLA(K) do { if (integer_class == 2 && K > 0 && ky == K) hipLaunchKernelGGL((parabola_kernel<K, 2, K>), grd, blk, 0, ctx->stream, disp3f, w, h, dstride_px, \
                                          lras, lrw, rras, rrw, range_minx, range_miny, kx, ky, out3f, ostride_px); \
                            else if (integer_class == 2 && K > 0) hipLaunchKernelGGL((parabola_kernel<K, 2>), grd, blk, 0, ctx->stream, disp3f, w, h, dstride_px, \
                                          lras, lrw, rras, rrw, range_minx, range_miny, kx, ky, out3f, ostride_px); \
                            else if (integer_class == 1 && K > 0 && ky == K) hipLaunchKernelGGL((parabola_kernel<K, 1, K>), grd, blk, 0, ctx->stream, disp3f, w, h, dstride_px, \
                                          lras, lrw, rras, rrw, range_minx, range_miny, kx, ky, out3f, ostride_px); \
                            else if (integer_class == 1 && K > 0) hipLaunchKernelGGL((parabola_kernel<K, 1>), grd, blk, 0, ctx->stream, disp3f, w, h, dstride_px, \
                                          lras, lrw, rras, rrw, range_minx, range_miny, kx, ky, out3f, ostride_px); \
                            else if (K > 0 && ky == K) hipLaunchKernelGGL((parabola_kernel<K, 0, K>), grd, blk, 0, ctx->stream, disp3f, w, h, dstride_px, lras, lrw, rras, rrw, \
                                          range_minx, range_miny, kx, ky, out3f, ostride_px); \
                            else hipLaunchKernelGGL((parabola_kernel<K, 0>), grd, blk, 0, ctx->stream, disp3f, w, h, dstride_px, lras, lrw, rras, rrw, \
                                          range_minx, range_miny, kx, ky, out3f, ostride_px); } while (0)
  switch (kx) {
    case 3: VW_PARABOLA(3); break;   case 5: VW_PARABOLA(5); break;   case 7: VW_PARABOLA(7); break;   case 9: VW_PARABOLA(9); break;
    case 11: VW_PARABOLA(11); break; case 13: VW_PARABOLA(13); break; case 15: VW_PARABOLA(15); break; default: VW_PARABOLA(0); break;
  }
#undef VW_PARABOLA
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}
