// pyramid.hip — one tile of pyramid_correlate (block matching): the coarse-to-fine level loop of
// vw::stereo::PyramidCorrelationView::prerasterize (src/vw/Stereo/CorrelationView.cc:273-886) with every image
// (pyramids, masks, per-level disparity) resident in HBM; only the per-level disparity crosses to the host, for the
// data-dependent zone scheduler (zones.h).
//
//   build_image_pyramids      CorrelationView.cc:67-239   edge-extended crops, nodata mean fill, [1 4 6 4 1]/16
//                                                          smoothing + decimation (filters.hip), mask decimation,
//                                                          per-level prefilter
//   zone loop                 CorrelationView.cc:596-700  calc_disparity per zone (bm_*.hip), R->L run + L/R check at
//                                                          level 0, += zone.disparity_range().min()
//   clean-up                  CorrelationView.cc:702-744  rm_outliers_using_thresh / disparity_cleanup_using_thresh
//                                                          (src/vw/Stereo/DisparityMap.h:318-441) + disparity_mask (:97-253)
//   zone refinement           CorrelationView.cc:754-799  subdivide_regions (zones.hip), x2, expand(2), crop
//   result                    CorrelationView.cc:876-885  + search_region.min(), cast to PixelMask<Vector2f>
//   SGM branch                CorrelationView.cc:391-595  per-level calc_disparity_sgm seeded by the previous level (sgm.hip),
//                                                          R->L run, sub-pixel view; blob filter (:242-271), lr_disp_diff
// Inputs whose box sums would round (prefiltered imagery, mean-filled nodata, deep levels with SSD / NCC) are matched in
// the reference's own summation order (bm_exact.hip); the class of every level is measured on the device.
// VW_CORRELATION_MGM runs every level with use_mgm, _FINAL_MGM only level 0 (:365-366).  collar_size is applied by the caller
// (PyramidCorrelationView::rasterize, CorrelationView.h:123-133: a larger tile is rasterised and cropped).
#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <map>
#include <string>
#include <vector>

#include "vwgpu_internal.h"
#include "zones.h"

namespace {

using vwgpu::IBox;
using vwgpu::SearchZone;

// ---- small kernels ----------------------------------------------------------------------------------------------

// The six base crops of a tile (two float images, four mask crops) in ONE launch: job = blockIdx.z.
struct CropJob { const void* src; ptrdiff_t stride; int w, h, x0, y0; void* dst; int dw, dh; int is_float, edge; };
struct CropJobs { CropJob j[6]; };
__global__ void crop_jobs_kernel(CropJobs jobs) {
  const CropJob J = jobs.j[blockIdx.z];
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= J.dw || y >= J.dh) return;
  int sx = J.x0 + x, sy = J.y0 + y;
  const bool outside = sx < 0 || sy < 0 || sx >= J.w || sy >= J.h;
  sx = sx < 0 ? 0 : (sx >= J.w ? J.w - 1 : sx);
  sy = sy < 0 ? 0 : (sy >= J.h ? J.h - 1 : sy);
  if (J.is_float) {
    static_cast<float*>(J.dst)[(size_t)y * J.dw + x] = (J.edge == 1 && outside) ? 0.0f : static_cast<const float*>(J.src)[(ptrdiff_t)sy * J.stride + sx];
  } else {
    const uint8_t v = (J.edge == 1 && outside) ? (uint8_t)0 : (J.src ? static_cast<const uint8_t*>(J.src)[(ptrdiff_t)sy * J.stride + sx] : (uint8_t)255);
    static_cast<uint8_t*>(J.dst)[(size_t)y * J.dw + x] = v;
  }
}

// dst(x,y) = src(ext(x0+x, y0+y)); EDGE 0 = clamp, 1 = zero.  src == nullptr means "an all-255 mask".
template <class T, int EDGE>
__global__ void crop_ext_kernel(const T* __restrict__ src, ptrdiff_t stride, int w, int h, int x0, int y0,
                                T* __restrict__ dst, int dw, int dh) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= dw || y >= dh) return;
  int sx = x0 + x, sy = y0 + y;
  T v;
  if (EDGE == 1 && (sx < 0 || sy < 0 || sx >= w || sy >= h)) {
    v = T(0);
  } else {
    sx = sx < 0 ? 0 : (sx >= w ? w - 1 : sx);
    sy = sy < 0 ? 0 : (sy >= h ? h - 1 : sy);
    v = src ? src[(ptrdiff_t)sy * stride + sx] : T(255);
  }
  dst[(size_t)y * dw + x] = v;
}

// sum / count (double) of img over the valid pixels of every second row and column (mean_pixel_value(subsample(.,2)),
// CorrelationView.cc:137-149; MeanAccumulator sums in double, src/vw/Math/Functors.h:469-487).  Every workgroup writes ONE
// partial {sum, count} formed in a fixed order, and cell[] receives the lowest set bit / largest exponent of the summed
// pixels: when count * 2^(hi+1) / 2^lo < 2^53 every partial sum is exactly representable and the host adds the partials
// (any order returns the reference's serial sum); otherwise masked_mean_serial_kernel walks the pixels in raster order.
__global__ void __launch_bounds__(256)
masked_mean_kernel(const float* __restrict__ img, const uint8_t* __restrict__ mask, int w, int h,
                   double* __restrict__ part, int* __restrict__ cell) {
  __shared__ double ws[4][2];
  double s = 0.0, n = 0.0;
  int lo = INT_MAX, hi = INT_MIN, bad = 0;
  for (int y = 2 * (blockIdx.y * blockDim.y + threadIdx.y); y < h; y += 2 * gridDim.y * blockDim.y)
    for (int x = 2 * (blockIdx.x * blockDim.x + threadIdx.x); x < w; x += 2 * gridDim.x * blockDim.x)
      if (mask[(size_t)y * w + x]) {
        const float v = img[(size_t)y * w + x];
        s += (double)v; n += 1.0;
        const unsigned u = __float_as_uint(v);
        const int e = (int)((u >> 23) & 0xffu);
        unsigned m = u & 0x7fffffu;
        if (e == 0xff) bad = 1;
        else if (e != 0 || m != 0) {
          int base;
          if (e == 0) base = -149; else { m |= 0x800000u; base = e - 127 - 23; }
          lo = min(lo, base + (__ffs((int)m) - 1));
          hi = max(hi, base + (31 - __clz((int)m)));
        }
      }
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o); n += __shfl_xor(n, o);
    lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); bad |= __shfl_xor(bad, o);
  }
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  if ((tid & 63) == 0) {
    ws[tid >> 6][0] = s; ws[tid >> 6][1] = n;
    if (lo != INT_MAX) { atomicMin(&cell[0], lo); atomicMax(&cell[1], hi); }
    if (bad) atomicOr(&cell[2], 1);
  }
  __syncthreads();
  if (tid == 0) {
    double* p = part + 2 * (size_t)(blockIdx.y * gridDim.x + blockIdx.x);
    p[0] = ((ws[0][0] + ws[1][0]) + ws[2][0]) + ws[3][0];
    p[1] = ((ws[0][1] + ws[1][1]) + ws[2][1]) + ws[3][1];
  }
}

// The reference's serial accumulation (m_accum += value in raster order), one wave: 64 pixels are fetched at a time and
// added one by one.  Only for images whose sum is not exactly representable along the way (see above).
__global__ void __launch_bounds__(64)
masked_mean_serial_kernel(const float* __restrict__ img, const uint8_t* __restrict__ mask, int w, int h, double* __restrict__ acc2) {
  const int ws = (w + 1) / 2, hs = (h + 1) / 2;
  const long long total = (long long)ws * hs;
  double s = 0.0, n = 0.0;
  for (long long i0 = 0; i0 < total; i0 += 64) {
    const long long i = i0 + threadIdx.x;
    float v = 0.0f;
    int ok = 0;
    if (i < total) {
      const int y = 2 * (int)(i / ws), x = 2 * (int)(i % ws);
      ok = mask[(size_t)y * w + x] != 0;
      v = img[(size_t)y * w + x];
    }
    const unsigned long long okm = __ballot(ok);
#pragma unroll 8
    for (int l = 0; l < 64; ++l) {
      if ((okm >> l) & 1ull) { s += (double)__shfl(v, l); n += 1.0; }
    }
  }
  if (threadIdx.x == 0) { acc2[0] = s; acc2[1] = n; }
}

__global__ void fill_masked_kernel(float* __restrict__ img, const uint8_t* __restrict__ mask, size_t n, float value) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && !mask[i]) img[i] = value;
}

// child += (ax, ay) on a crop of a PixelMask<Vector2i> image; validity untouched (PixelMask arithmetic computes the
// children regardless, src/vw/Image/PixelMask.h:322-360).
__global__ void add_offset_kernel(int32_t* __restrict__ d, ptrdiff_t stride_px, int w, int h, int ax, int ay) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  int32_t* p = d + ((ptrdiff_t)y * stride_px + x) * 3;
  p[0] += ax; p[1] += ay;
}

// RmOutliersUsingThreshFunc (src/vw/Stereo/DisparityMap.h:357-385) evaluated over the output domain
// [ox0, ox0+ow) x [oy0, oy0+oh) of the constant-edge-extended input.  A 32 x 8 output tile and its (2 hh + 1) x (2 hv + 1)
// neighbourhood rim are staged in LDS once (the direct version read 121 x 12 B per pixel through the L1: 0.3 ms per 1024^2 level).
// `fabs((double)(a - b)) <= pthr` on int32 differences is the unsigned compare |a - b| <= floor(pthr) (thr; `never` for a
// negative or NaN threshold); the match ratio is compared in float64 as the reference does.
// Packed form: when every valid disparity of the tile is below 2^13 in magnitude and thr < 2^13 (always, in a pyramid) the tile is
// ALSO staged as one dword per pixel — (dx + 2^14) | (dy + 2^14) << 16, 0xffffffff for an invalid pixel — and a neighbour is one LDS
// read and four instructions: t = n + (thr - c) per half (v_pk_add_u16: |n - c| <= thr  <=>  t <= 2 thr as unsigned 16-bit values —
// a negative n - c + thr wraps to more than 2^15, and the invalid code gives 65535 - (c + 2^14) + thr, between 2^15 and 2^16: it
// never passes), max with 2 thr per half, compare with the splat, add the carry.  The three-array form took
// ten instructions and three LDS reads per neighbour: 160 us of an 870 us SAD tile went into this filter.
// HH > 0: hh == hv == HH known at compile time (the loops unroll: the small levels of a pyramid are latency bound here).
template <int HH>
__global__ void __launch_bounds__(256)
rm_outliers_kernel(const int32_t* __restrict__ src, int w, int h, int hh, int hv, unsigned thr, int never, double rthr,
                   int32_t* __restrict__ dst, int ow, int oh, int ox0, int oy0, size_t src_tile = 0, size_t dst_tile = 0) {
  extern __shared__ int32_t rm_sm[];
  src += blockIdx.z * src_tile; dst += blockIdx.z * dst_tile;     // tile groups: blockIdx.z = the tile, its images at a fixed stride (ints)
  if (HH > 0) { hh = HH; hv = HH; }
  const int tw = 32 + 2 * hh, th = 8 + 2 * hv, tn = tw * th;
  int32_t* sx = rm_sm;
  int32_t* sy = rm_sm + tn;
  unsigned* pk = reinterpret_cast<unsigned*>(rm_sm + 2 * tn);
  uint8_t* sv = reinterpret_cast<uint8_t*>(rm_sm + 3 * tn);
  const int tid = threadIdx.y * 32 + threadIdx.x;
  const int bx = blockIdx.x * 32 + ox0 - hh, by = blockIdx.y * 8 + oy0 - hv;      // source coordinates of the tile's corner
  bool big = thr >= 8192u;
  // (three pixels per thread and batch: their nine loads are requested together — the plain loop waited for every pixel's three words)
  for (int i0 = tid; i0 < tn; i0 += 3 * 256) {
    int32_t d0[3], d1[3], d2[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int i = i0 + b * 256;
      const int ty = i / tw, tx = i - ty * tw;
      int x = bx + tx, y = by + ty;
      x = x < 0 ? 0 : (x >= w ? w - 1 : x);
      y = y < 0 ? 0 : (y >= h ? h - 1 : y);
      const int32_t* c = src + ((size_t)y * w + x) * 3;
      const bool in = i < tn;
      d0[b] = in ? c[0] : 0; d1[b] = in ? c[1] : 0; d2[b] = in ? c[2] : 0;
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const int i = i0 + b * 256;
      if (i >= tn) break;
      const bool v = d2[b] != 0;
      sx[i] = d0[b]; sy[i] = d1[b]; sv[i] = v;
      if (v) big = big || d0[b] <= -8192 || d0[b] >= 8192 || d1[b] <= -8192 || d1[b] >= 8192;
      pk[i] = v ? ((unsigned)(d0[b] + 16384) & 0xffffu) | ((unsigned)(d1[b] + 16384) << 16) : 0xffffffffu;
    }
  }
  const bool wide = __syncthreads_or(big ? 1 : 0) != 0;         // (also the barrier behind the staging)
  const int ox = blockIdx.x * 32 + threadIdx.x, oy = blockIdx.y * 8 + threadIdx.y;
  if (ox >= ow || oy >= oh) return;
  const int ci = (threadIdx.y + hv) * tw + threadIdx.x + hh;
  int32_t r0 = sx[ci], r1 = sy[ci], r2 = 0;
  if (sv[ci]) {
    // the valid flag is written back as the source has it
    const int x = min(max(ox + ox0, 0), w - 1), y = min(max(oy + oy0, 0), h - 1);
    r2 = src[((size_t)y * w + x) * 3 + 2];
    int matched = 0;
    if (!never && !wide) {
      typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
      const unsigned cb = ((thr - (unsigned)(r0 + 16384)) & 0xffffu) | ((thr - (unsigned)(r1 + 16384)) << 16);      // thr - c per half (mod 2^16)
      const unsigned lim = (2u * thr) | ((2u * thr) << 16);
      auto one = [&](unsigned n) __attribute__((always_inline)) {
        const us2_t t = __builtin_bit_cast(us2_t, n) + __builtin_bit_cast(us2_t, cb);
        const us2_t m = __builtin_elementwise_max(t, __builtin_bit_cast(us2_t, lim));
        matched += __builtin_bit_cast(unsigned, m) == lim ? 1 : 0;
      };
      if (HH > 0) {
#pragma unroll
        for (int yk = 0; yk <= 2 * HH; ++yk) {
          const unsigned* row = pk + (threadIdx.y + yk) * tw + threadIdx.x;
#pragma unroll
          for (int xk = 0; xk <= 2 * HH; ++xk) one(row[xk]);
        }
      } else {
        for (int yk = 0; yk <= 2 * hv; ++yk) {
          const unsigned* row = pk + (threadIdx.y + yk) * tw + threadIdx.x;
          for (int xk = 0; xk <= 2 * hh; ++xk) one(row[xk]);
        }
      }
    } else if (!never) {
      for (int yk = 0; yk <= 2 * hv; ++yk) {
        const int row = (threadIdx.y + yk) * tw + threadIdx.x;
        for (int xk = 0; xk <= 2 * hh; ++xk) {
          const int i = row + xk;
          const int d0 = r0 - sx[i], d1 = r1 - sy[i];                         // int32 wrap-around, as the reference's subtraction
          const unsigned a0 = d0 < 0 ? 0u - (unsigned)d0 : (unsigned)d0, a1 = d1 < 0 ? 0u - (unsigned)d1 : (unsigned)d1;
          matched += (sv[i] && a0 <= thr && a1 <= thr) ? 1 : 0;
        }
      }
    }
    const int total = (2 * hh + 1) * (2 * hv + 1);
    if (((double)matched / (double)total) < rthr) { r0 = r1 = r2 = 0; }
  }
  int32_t* o = dst + ((size_t)oy * ow + ox) * 3;
  o[0] = r0; o[1] = r1; o[2] = r2;
}

// The same filter without the LDS tile, for neighbourhoods too large for it (64 KB).
__global__ void rm_outliers_direct_kernel(const int32_t* __restrict__ src, int w, int h, int hh, int hv, double pthr, double rthr,
                                   int32_t* __restrict__ dst, int ow, int oh, int ox0, int oy0) {
  const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y * blockDim.y + threadIdx.y;
  if (ox >= ow || oy >= oh) return;
  auto at = [&](int x, int y) -> const int32_t* {
    x = x < 0 ? 0 : (x >= w ? w - 1 : x);
    y = y < 0 ? 0 : (y >= h ? h - 1 : y);
    return src + ((size_t)y * w + x) * 3;
  };
  const int x = ox + ox0, y = oy + oy0;
  const int32_t* c = at(x, y);
  int32_t r0 = c[0], r1 = c[1], r2 = c[2];
  if (r2) {
    int matched = 0, total = 0;
    for (int yk = -hv; yk <= hv; ++yk)
      for (int xk = -hh; xk <= hh; ++xk) {
        const int32_t* n = at(x + xk, y + yk);
        if (n[2] && fabs((double)(r0 - n[0])) <= pthr && fabs((double)(r1 - n[1])) <= pthr) matched++;
        total++;
      }
    if (((double)matched / (double)total) < rthr) { r0 = r1 = r2 = 0; }
  }
  int32_t* o = dst + ((size_t)oy * ow + ox) * 3;
  o[0] = r0; o[1] = r1; o[2] = r2;
}

// second pass of disparity_cleanup_using_thresh: functor (1,1,3.0,0.20) over the inner VIEW, which is given here on
// the domain padded by one pixel (DisparityMap.h:427-441).
__global__ void cleanup_outer_kernel(const int32_t* __restrict__ inner, int w, int h, int32_t* __restrict__ dst) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  const int pw = w + 2;
  const int32_t* c = inner + ((size_t)(y + 1) * pw + (x + 1)) * 3;
  int32_t r0 = c[0], r1 = c[1], r2 = c[2];
  if (r2) {
    int matched = 0;
    for (int yk = -1; yk <= 1; ++yk)
      for (int xk = -1; xk <= 1; ++xk) {
        const int32_t* n = inner + ((size_t)(y + 1 + yk) * pw + (x + 1 + xk)) * 3;
        if (n[2] && fabs((double)(r0 - n[0])) <= 3.0 && fabs((double)(r1 - n[1])) <= 3.0) matched++;
      }
    if (((double)matched / 9.0) < 0.20) { r0 = r1 = r2 = 0; }
  }
  int32_t* o = dst + ((size_t)y * w + x) * 3;
  o[0] = r0; o[1] = r1; o[2] = r2;
}

// DisparityMaskView::operator() (DisparityMap.h:132-155), in place.
__global__ void disparity_mask_kernel(int32_t* __restrict__ d, int w, int h, const uint8_t* __restrict__ m1,
                                      const uint8_t* __restrict__ m2, int m2w, int m2h) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= w || j >= h) return;
  int32_t* p = d + ((size_t)j * w + i) * 3;
  bool keep = m1[(size_t)j * w + i] != 0 && p[2] != 0;
  if (keep) {
    const int x = i + p[0], y = j + p[1];
    keep = !(x < 0 || x >= m2w || y < 0 || y >= m2h || m2[(size_t)y * m2w + x] == 0);
  }
  if (!keep) { p[0] = 0; p[1] = 0; p[2] = 0; }
}

// disparity_mask_kernel + finish_kernel in one pass, for the last level of a block-matching tile when nothing else reads the masked image
__global__ void finish_masked_kernel(const int32_t* __restrict__ d, int w, int h, const uint8_t* __restrict__ m1, const uint8_t* __restrict__ m2,
                                     int m2w, int m2h, int ax, int ay, float* __restrict__ out, ptrdiff_t ostride_px) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= w || j >= h) return;
  const int32_t* p = d + ((size_t)j * w + i) * 3;
  int p0 = p[0], p1 = p[1], p2 = p[2];
  bool keep = m1[(size_t)j * w + i] != 0 && p2 != 0;
  if (keep) {
    const int x = i + p0, y = j + p1;
    keep = !(x < 0 || x >= m2w || y < 0 || y >= m2h || m2[(size_t)y * m2w + x] == 0);
  }
  if (!keep) { p0 = 0; p1 = 0; p2 = 0; }
  float* o = out + ((ptrdiff_t)j * ostride_px + i) * 3;
  o[0] = (float)(p0 + ax); o[1] = (float)(p1 + ay); o[2] = p2 ? 1.0f : 0.0f;
}

// out = PixelMask<Vector2f>(disparity + search.min)  (CorrelationView.cc:879-881)
__global__ void finish_kernel(const int32_t* __restrict__ d, int w, int h, int ax, int ay, float* __restrict__ out, ptrdiff_t ostride_px) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  const int32_t* p = d + ((size_t)y * w + x) * 3;
  float* o = out + ((ptrdiff_t)y * ostride_px + x) * 3;
  o[0] = (float)(p[0] + ax); o[1] = (float)(p[1] + ay); o[2] = p[2] ? 1.0f : 0.0f;
}

// ---- blob filter (disparity_blob_filter, CorrelationView.cc:242-271): 8-connected components of the valid pixels by a
// lock-free union-find (roots = smallest pixel index), component sizes by atomics, components of <= area pixels erased.
__device__ __forceinline__ int cc_find(int* L, int x) {
  int p = L[x];
  while (p != x) { const int g = L[p]; L[x] = g; x = p; p = g; }     // path halving; racy writes only ever shorten paths
  return x;
}
__device__ __forceinline__ void cc_unite(int* L, int a, int b) {
  while (true) {
    a = cc_find(L, a); b = cc_find(L, b);
    if (a == b) return;
    if (a < b) { const int t = a; a = b; b = t; }                     // hang the larger root under the smaller one
    const int old = atomicMin(&L[a], b);
    if (old == a) return;
    a = old;
  }
}
__global__ void cc_init_kernel(const int32_t* __restrict__ d, int n, int* __restrict__ label, int* __restrict__ size) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  label[i] = d[(size_t)i * 3 + 2] ? i : -1;
  size[i] = 0;
}
__global__ void cc_union_kernel(int* label, int w, int h) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  const int i = y * w + x;
  if (label[i] < 0) return;
  if (x > 0 && label[i - 1] >= 0) cc_unite(label, i, i - 1);
  if (y > 0) {
    if (label[i - w] >= 0) cc_unite(label, i, i - w);
    if (x > 0 && label[i - w - 1] >= 0) cc_unite(label, i, i - w - 1);
    if (x + 1 < w && label[i - w + 1] >= 0) cc_unite(label, i, i - w + 1);
  }
}
// the forest is final after the union kernel: read-only root walks (a compressing find here would race with the readers)
__device__ __forceinline__ int cc_root(const int* L, int x) {
  int p = L[x];
  while (p != x) { x = p; p = L[x]; }
  return x;
}
__global__ void cc_count_kernel(const int* __restrict__ label, int n, int* __restrict__ size) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || label[i] < 0) return;
  atomicAdd(&size[cc_root(label, i)], 1);
}
__global__ void cc_erase_kernel(int32_t* __restrict__ d, const int* __restrict__ label, const int* __restrict__ size, int n, int area) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || label[i] < 0) return;
  if (size[cc_root(label, i)] <= area) { d[(size_t)i * 3] = 0; d[(size_t)i * 3 + 1] = 0; d[(size_t)i * 3 + 2] = 0; }
}

// SGM result: sub-pixel view + search.min, invalid where the filtered integer disparity is (CorrelationView.cc:862-875)
__global__ void finish_sgm_kernel(const int32_t* __restrict__ d, const float* __restrict__ sub, int w, int h, float ax, float ay,
                                  float* __restrict__ out, ptrdiff_t ostride_px) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  const size_t i = ((size_t)y * w + x) * 3;
  float* o = out + ((ptrdiff_t)y * ostride_px + x) * 3;
  o[0] = sub[i] + ax; o[1] = sub[i + 1] + ay; o[2] = (d[i + 2] && sub[i + 2] != 0.0f) ? 1.0f : 0.0f;
}

// pixels the filters removed lose their L-R / R-L discrepancy as well (CorrelationView.cc:846-855)
__global__ void lr_diff_invalidate_kernel(const int32_t* __restrict__ d, int w, int h, float* __restrict__ diff2, ptrdiff_t dstride, int ulx, int uly) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  if (!d[((size_t)y * w + x) * 3 + 2]) diff2[((ptrdiff_t)(y + uly) * dstride + (x + ulx)) * 2 + 1] = 0.0f;
}

// Zeroes `count` 16-byte words at the start of every tile's buffer (blockIdx.y = the tile; buffers at a fixed stride).  hipMemset2DAsync over
// slices ~100 MB apart took 170 us per call (its fill kernel walks the pitch) — as long as the level's matcher launches.
__global__ void __launch_bounds__(256)
zero_tiles_kernel(uint4* __restrict__ base, size_t tile_words, size_t count) {
  uint4* p = base + blockIdx.y * tile_words;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

// finish_masked_kernel / finish_kernel / zero_out_kernel for the tiles of a group (blockIdx.z): per-tile images at a fixed stride, per-tile
// destinations from a table.  mode[t]: 0 = nothing to write, 1 = finish, 2 = mask + finish, 3 = zeros.
constexpr int VWGPU_MAX_GROUP = 16;
struct GroupOuts { float* out[VWGPU_MAX_GROUP]; ptrdiff_t os[VWGPU_MAX_GROUP]; int mode[VWGPU_MAX_GROUP]; };
__global__ void finish_group_kernel(GroupOuts g, const int32_t* __restrict__ d, size_t d_tile, int w, int h, const uint8_t* __restrict__ m1,
                                    const uint8_t* __restrict__ m2, size_t mask_tile, int m2w, int m2h, int ax, int ay) {
  const int t = blockIdx.z, mode = g.mode[t];
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (mode == 0 || i >= w || j >= h) return;
  float* o = g.out[t] + ((ptrdiff_t)j * g.os[t] + i) * 3;
  if (mode == 3) { o[0] = o[1] = o[2] = 0.0f; return; }
  const int32_t* p = d + t * d_tile + ((size_t)j * w + i) * 3;
  int p0 = p[0], p1 = p[1], p2 = p[2];
  if (mode == 2) {
    bool keep = m1[t * mask_tile + (size_t)j * w + i] != 0 && p2 != 0;
    if (keep) {
      const int x = i + p0, y = j + p1;
      keep = !(x < 0 || x >= m2w || y < 0 || y >= m2h || m2[t * mask_tile + (size_t)y * m2w + x] == 0);
    }
    if (!keep) { p0 = 0; p1 = 0; p2 = 0; }
  }
  o[0] = (float)(p0 + ax); o[1] = (float)(p1 + ay); o[2] = p2 ? 1.0f : 0.0f;
}

__global__ void zero_out_kernel(float* __restrict__ out, ptrdiff_t ostride_px, int w, int h) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  float* o = out + ((ptrdiff_t)y * ostride_px + x) * 3;
  o[0] = o[1] = o[2] = 0.0f;
}

inline dim3 grid2(int w, int h) { return dim3((w + 63) / 64, (h + 3) / 4); }
const dim3 kBlk(64, 4);

// Bump allocator over one arena; everything of a tile lives until the call returns.
struct Bump {
  char* base; size_t cap, off = 0;
  template <class T> T* take(size_t n) {
    off = vwgpu_align_up(off, 256);
    T* p = reinterpret_cast<T*>(base + off);
    off += n * sizeof(T);
    return off <= cap ? p : nullptr;
  }
};

struct DevImg { float* p = nullptr; int w = 0, h = 0; };
struct DevMask { uint8_t* p = nullptr; int w = 0, h = 0; };

}  // namespace

int vwgpu_launch_disparity_filter(vwgpu_ctx* ctx, const int32_t* src, int w, int h, int hh, int hv, double pthr, double rthr,
                                  bool cleanup, int32_t* tmp_padded, int32_t* dst, bool inner_only, int tiles, size_t tile_ints) {
  // tiles > 1 (the tile groups of vwgpu_pyramid_group_impl): the same filter over `tiles` images at a fixed stride of tile_ints ints, for the
  // source, the padded intermediate and the destination alike — one launch (blockIdx.z); the second pass is the caller's (inner_only)
  if (hh < 0 || hv < 0) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "disparity filter: negative half kernel");
  const size_t rm_lds = (size_t)(32 + 2 * hh) * (8 + 2 * hv) * 13;
  const bool direct = rm_lds > 64 * 1024;
  if (tiles > 1 && (direct || (cleanup && !inner_only))) return vwgpu_fail(ctx, VWGPU_ERR_LOGIC, "disparity filter: this form is not built for tile groups");
  const size_t ts = tiles > 1 ? tile_ints : 0;
  // fabs((double)int32 difference) <= pthr  <=>  |difference| <= floor(pthr)
  const int never = !(pthr >= 0.0);                                   // negative or NaN: nothing matches
  const unsigned thr = never ? 0u : (pthr >= 4294967295.0 ? 0xffffffffu : (unsigned)std::floor(pthr));
  auto rm_grid = [tiles](int ww, int hh2) { return dim3((unsigned)((ww + 31) / 32), (unsigned)((hh2 + 7) / 8), (unsigned)std::max(tiles, 1)); };
  if (!cleanup) {
    vwgpu_prof_scope ps(ctx, "rm_outliers");
    if (direct) hipLaunchKernelGGL(rm_outliers_direct_kernel, grid2(w, h), kBlk, 0, ctx->stream, src, w, h, hh, hv, pthr, rthr, dst, w, h, 0, 0);
    else if (hh == 5 && hv == 5) hipLaunchKernelGGL(rm_outliers_kernel<5>, rm_grid(w, h), dim3(32, 8), rm_lds, ctx->stream, src, w, h, hh, hv, thr, never, rthr, dst, w, h, 0, 0, ts, ts);
    else hipLaunchKernelGGL(rm_outliers_kernel<0>, rm_grid(w, h), dim3(32, 8), rm_lds, ctx->stream, src, w, h, hh, hv, thr, never, rthr, dst, w, h, 0, 0, ts, ts);
  } else {
    {
      vwgpu_prof_scope ps(ctx, "rm_outliers");
      if (direct) hipLaunchKernelGGL(rm_outliers_direct_kernel, grid2(w + 2, h + 2), kBlk, 0, ctx->stream, src, w, h, hh, hv, pthr, rthr,
                                     tmp_padded, w + 2, h + 2, -1, -1);
      else if (hh == 5 && hv == 5) hipLaunchKernelGGL(rm_outliers_kernel<5>, rm_grid(w + 2, h + 2), dim3(32, 8), rm_lds, ctx->stream, src, w, h, hh, hv, thr, never, rthr,
                                                      tmp_padded, w + 2, h + 2, -1, -1, ts, ts);
      else hipLaunchKernelGGL(rm_outliers_kernel<0>, rm_grid(w + 2, h + 2), dim3(32, 8), rm_lds, ctx->stream, src, w, h, hh, hv, thr, never, rthr,
                              tmp_padded, w + 2, h + 2, -1, -1, ts, ts);
    }
    if (!inner_only) {                                              // (inner_only: the caller applies the second pass itself, see zone_extent_fused_kernel)
      vwgpu_prof_scope ps(ctx, "disparity_cleanup_outer");
      hipLaunchKernelGGL(cleanup_outer_kernel, grid2(w, h), kBlk, 0, ctx->stream, tmp_padded, w, h, dst);
    }
  }
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}

// scratch: 2 x w*h ints
int vwgpu_launch_blob_filter(vwgpu_ctx* ctx, int32_t* d, int w, int h, int area, int* scratch) {
  if (area < 1) return VWGPU_OK;
  const int n = w * h;
  int* label = scratch; int* size = scratch + n;
  vwgpu_prof_scope ps(ctx, "blob_filter");
  const dim3 g1((n + 255) / 256), b1(256);
  hipLaunchKernelGGL(cc_init_kernel, g1, b1, 0, ctx->stream, d, n, label, size);
  hipLaunchKernelGGL(cc_union_kernel, grid2(w, h), kBlk, 0, ctx->stream, label, w, h);
  hipLaunchKernelGGL(cc_count_kernel, g1, b1, 0, ctx->stream, label, n, size);
  hipLaunchKernelGGL(cc_erase_kernel, g1, b1, 0, ctx->stream, d, label, size, n, area);
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}

int vwgpu_launch_disparity_mask(vwgpu_ctx* ctx, int32_t* d, int w, int h, const uint8_t* m1, const uint8_t* m2, int m2w, int m2h) {
  vwgpu_prof_scope ps(ctx, "disparity_mask");
  hipLaunchKernelGGL(disparity_mask_kernel, grid2(w, h), kBlk, 0, ctx->stream, d, w, h, m1, m2, m2w, m2h);
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}

// Zone scheduler, device part: one wave per leaf box of the quad tree measures the element-wise min / max of the VALID
// disparities inside the box and inside the box grown by 1 px (clipped to the image) — what subdivide_regions reads
// (Correlation.cc:149-171).  out: 10 ints per leaf (vwgpu::LeafExtent).
__global__ void __launch_bounds__(64)
zone_extent_kernel(const int32_t* __restrict__ disp, int w, int h, const int4* __restrict__ rects, int n, int32_t* __restrict__ out, size_t disp_tile = 0) {
  const int leaf = blockIdx.x;
  if (leaf >= n) return;
  disp += blockIdx.y * disp_tile; out += (size_t)blockIdx.y * n * 10;      // tile groups: blockIdx.y = the tile
  const int4 r = rects[leaf];                       // x0, y0, x1, y1
  const int ax0 = max(r.x - 1, 0), ay0 = max(r.y - 1, 0), ax1 = min(r.z + 1, w), ay1 = min(r.w + 1, h);
  const int aw = ax1 - ax0, count = aw * (ay1 - ay0);
  int any = 0, lox = INT_MAX, loy = INT_MAX, hix = INT_MIN, hiy = INT_MIN;
  int anya = 0, loxa = INT_MAX, loya = INT_MAX, hixa = INT_MIN, hiya = INT_MIN;
  for (int i = threadIdx.x; i < count; i += 64) {
    const int yy = i / aw, x = ax0 + (i - yy * aw), y = ay0 + yy;
    const int32_t* p = disp + ((size_t)y * w + x) * 3;
    if (!p[2]) continue;
    const int dx = p[0], dy = p[1];
    anya = 1; loxa = min(loxa, dx); hixa = max(hixa, dx); loya = min(loya, dy); hiya = max(hiya, dy);
    if (x >= r.x && x < r.z && y >= r.y && y < r.w) {
      any = 1; lox = min(lox, dx); hix = max(hix, dx); loy = min(loy, dy); hiy = max(hiy, dy);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    any |= __shfl_xor(any, o); anya |= __shfl_xor(anya, o);
    lox = min(lox, __shfl_xor(lox, o)); loy = min(loy, __shfl_xor(loy, o));
    hix = max(hix, __shfl_xor(hix, o)); hiy = max(hiy, __shfl_xor(hiy, o));
    loxa = min(loxa, __shfl_xor(loxa, o)); loya = min(loya, __shfl_xor(loya, o));
    hixa = max(hixa, __shfl_xor(hixa, o)); hiya = max(hiya, __shfl_xor(hiya, o));
  }
  if (threadIdx.x == 0) {
    int32_t* o = out + (size_t)leaf * 10;
    o[0] = any; o[1] = lox; o[2] = loy; o[3] = hix; o[4] = hiy;
    o[5] = anya; o[6] = loxa; o[7] = loya; o[8] = hixa; o[9] = hiya;
  }
}

// The same measurement for an intermediate level of the block-matching branch, straight from the first pass of the clean-up filter: the
// level's finished disparity image — second pass of disparity_cleanup_using_thresh (cleanup_outer_kernel), then disparity_mask
// (disparity_mask_kernel) — is read by nothing but this scheduler, so the two point-wise passes are applied to the pixels a leaf looks
// at as they are read, and the image itself is never written (two launches per level less; the small levels are launch bound).
__global__ void __launch_bounds__(64)
zone_extent_fused_kernel(const int32_t* __restrict__ inner, int w, int h, const uint8_t* __restrict__ m1, const uint8_t* __restrict__ m2,
                         int m2w, int m2h, const int4* __restrict__ rects, int n, int32_t* __restrict__ out,
                         size_t inner_tile = 0, size_t mask_tile = 0) {
  const int leaf = blockIdx.x;
  if (leaf >= n) return;
  inner += blockIdx.y * inner_tile; m1 += blockIdx.y * mask_tile; m2 += blockIdx.y * mask_tile; out += (size_t)blockIdx.y * n * 10;      // tile groups
  const int4 r = rects[leaf];                       // x0, y0, x1, y1
  const int ax0 = max(r.x - 1, 0), ay0 = max(r.y - 1, 0), ax1 = min(r.z + 1, w), ay1 = min(r.w + 1, h);
  const int aw = ax1 - ax0, count = aw * (ay1 - ay0);
  const int pw = w + 2;
  int any = 0, lox = INT_MAX, loy = INT_MAX, hix = INT_MIN, hiy = INT_MIN;
  int anya = 0, loxa = INT_MAX, loya = INT_MAX, hixa = INT_MIN, hiya = INT_MIN;
  for (int i = threadIdx.x; i < count; i += 64) {
    const int yy = i / aw, x = ax0 + (i - yy * aw), y = ay0 + yy;
    const int32_t* c = inner + ((size_t)(y + 1) * pw + (x + 1)) * 3;
    const int dx = c[0], dy = c[1];
    if (!c[2]) continue;
    int matched = 0;                                 // cleanup_outer_kernel
    for (int yk = -1; yk <= 1; ++yk)
      for (int xk = -1; xk <= 1; ++xk) {
        const int32_t* q = inner + ((size_t)(y + 1 + yk) * pw + (x + 1 + xk)) * 3;
        if (q[2] && fabs((double)(dx - q[0])) <= 3.0 && fabs((double)(dy - q[1])) <= 3.0) matched++;
      }
    if (((double)matched / 9.0) < 0.20) continue;
    if (m1[(size_t)y * w + x] == 0) continue;        // disparity_mask_kernel
    const int tx = x + dx, ty = y + dy;
    if (tx < 0 || tx >= m2w || ty < 0 || ty >= m2h || m2[(size_t)ty * m2w + tx] == 0) continue;
    anya = 1; loxa = min(loxa, dx); hixa = max(hixa, dx); loya = min(loya, dy); hiya = max(hiya, dy);
    if (x >= r.x && x < r.z && y >= r.y && y < r.w) {
      any = 1; lox = min(lox, dx); hix = max(hix, dx); loy = min(loy, dy); hiy = max(hiy, dy);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    any |= __shfl_xor(any, o); anya |= __shfl_xor(anya, o);
    lox = min(lox, __shfl_xor(lox, o)); loy = min(loy, __shfl_xor(loy, o));
    hix = max(hix, __shfl_xor(hix, o)); hiy = max(hiy, __shfl_xor(hiy, o));
    loxa = min(loxa, __shfl_xor(loxa, o)); loya = min(loya, __shfl_xor(loya, o));
    hixa = max(hixa, __shfl_xor(hixa, o)); hiya = max(hiya, __shfl_xor(hiya, o));
  }
  if (threadIdx.x == 0) {
    int32_t* o = out + (size_t)leaf * 10;
    o[0] = any; o[1] = lox; o[2] = loy; o[3] = hix; o[4] = hiy;
    o[5] = anya; o[6] = loxa; o[7] = loya; o[8] = hixa; o[9] = hiya;
  }
}

// All pointers are device pointers (masks may be null).  out: bw x bh x 3 floats.
int vwgpu_pyramid_correlate_impl(vwgpu_ctx* ctx, const float* left, int lw, int lh, ptrdiff_t ls,
                                 const float* right, int rw, int rh, ptrdiff_t rs,
                                 const uint8_t* lmask, ptrdiff_t lms, const uint8_t* rmask, ptrdiff_t rms,
                                 const vwgpu_pyramid_params* P, int bx, int by, int bw, int bh,
                                 float* out, ptrdiff_t os, float* lr_diff) {
  const int kx = P->kernel_x, ky = P->kernel_y;
  // lr_diff: DEVICE copy of P->lr_disp_diff (or null); its geometry comes from P
  const ptrdiff_t lr_stride = P->lr_disp_diff_stride ? P->lr_disp_diff_stride : P->lr_disp_diff_cols;
  const int lr_ulx = bx - P->region_ul_x, lr_uly = by - P->region_ul_y;       // tile origin inside the diff image
  const IBox search(P->search_min_x, P->search_min_y, P->search_max_x, P->search_max_y);
  const IBox bbox(bx, by, bx + bw, by + bh);
  hipStream_t st = ctx->stream;

  // constructor + 1.0): number of levels (CorrelationView.h:99-105, CorrelationView.cc:301-310)
  const int largest_search = std::max(search.dx(), search.dy());
  int by_search = (int)(std::floor(std::log(float(largest_search)) / std::log(2.0f)) - 1);
  by_search = std::max(0, std::min(by_search, P->max_pyramid_levels));
  int L = (int)std::floor(std::log((double)std::min(bw, bh)) / std::log(2.0f) - std::log((double)std::max(kx, ky)) / std::log(2.0f));
  L = std::min(L, by_search);
  if (L < 1) L = 0;
  const int hkx = kx / 2, hky = ky / 2, up = 1 << L;

  // geometry of level 0
  IBox lg = bbox; lg.x0 -= hkx * up; lg.x1 += hkx * up; lg.y0 -= hky * up; lg.y1 += hky * up;
  const IBox rg(lg.x0 + search.x0, lg.y0 + search.y0, lg.x1 + search.x0 + search.dx(), lg.y1 + search.y0 + search.dy());
  const IBox rmb(bbox.x0 + search.x0, bbox.y0 + search.y0, bbox.x1 + search.x0 + search.dx(), bbox.y1 + search.y0 + search.dy());

  // arena: pyramids (4/3 of the bases, twice for the prefiltered copies), masks, disparities, scratch
  const size_t nl = (size_t)lg.dx() * lg.dy(), nr = (size_t)rg.dx() * rg.dy();
  const size_t need = 4 * (nl + nr) * 3 + (nl + nr) * 2 + (size_t)rmb.dx() * rmb.dy() * 2 + (size_t)bw * bh * (12 * 3 + 2) +
                      (size_t)(bw + 2) * (bh + 2) * 12 + (nl + nr) * 4 * 2 + (size_t)(rg.dx() + 2 * search.dx()) * (rg.dy() + 2 * search.dy()) * 20 +
                      (size_t)(lg.dx() + 2 * search.dx() + 2) * (lg.dy() + 2 * search.dy() + 2) * 4 + (1 << 20) +
                      // per-level tables taken inside the level loop: zone flags of both passes and the R->L need records (8 ints per
                      // zone; zones <= 4 per 16 x 16 px of the tile + 64) with one cell flag per 16 x 16 px of every zone's R->L image (a zone widened
                      // by its search range: at most (sdx / 16 + 3)(sdy / 16 + 3) cells for a leaf-sized zone)
                      (size_t)(L + 1) * ((size_t)bw * bh / 64 + 64) * (48 + (size_t)(search.dx() / 16 + 3) * (search.dy() / 16 + 3)) + 4096;
  // SGM: R->L crops (left image grown by twice the search), R->L masks, disparities of both directions and their history
  const size_t rl_px = (size_t)(bw + search.dx() + 8) * (bh + search.dy() + 8);
  const size_t lrev_px = (size_t)(lg.dx() + 2 * search.dx() + 8) * (lg.dy() + 2 * search.dy() + 8);
  const size_t need_blob = P->blob_filter_area > 0 ? (size_t)(bw + search.dx() + 8) * (bh + search.dy() + 8) * 8 + 1024 : 0;
  const size_t need_sgm = P->algorithm != 0 ? lrev_px * 4 + rl_px * (12 * 5 + 1) + lrev_px + (size_t)bw * bh * (12 + 12) + (1 << 16) : 0;
  int rc = vwgpu_arena_reserve(ctx, &ctx->pyr, need + need_sgm + need_blob);
  if (rc) return rc;
  Bump A{static_cast<char*>(ctx->pyr.base), ctx->pyr.cap};
  std::vector<DevImg> lp(L + 1), rp(L + 1);
  std::vector<DevMask> lmp(L + 1), rmp(L + 1);
  auto fail_mem = [&]() { return vwgpu_fail(ctx, VWGPU_ERR_LOGIC, "pyramid_correlate: internal arena too small"); };

  lp[0].w = lg.dx(); lp[0].h = lg.dy(); lp[0].p = A.take<float>(nl);
  rp[0].w = rg.dx(); rp[0].h = rg.dy(); rp[0].p = A.take<float>(nr);
  uint8_t* lmx = A.take<uint8_t>(nl);
  uint8_t* rmx = A.take<uint8_t>(nr);
  lmp[0].w = bw; lmp[0].h = bh; lmp[0].p = A.take<uint8_t>((size_t)bw * bh);
  rmp[0].w = rmb.dx(); rmp[0].h = rmb.dy(); rmp[0].p = A.take<uint8_t>((size_t)rmb.dx() * rmb.dy());
  double* d_acc = A.take<double>(4);
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  if (!lp[0].p || !rp[0].p || !lmx || !rmx || !lmp[0].p || !rmp[0].p || !d_acc) return fail_mem();
  {
    vwgpu_prof_scope ps(ctx, "pyramid_base_crops");
    CropJobs cj;
    cj.j[0] = CropJob{left, ls, lw, lh, lg.x0, lg.y0, lp[0].p, lp[0].w, lp[0].h, 1, 0};
    cj.j[1] = CropJob{right, rs, rw, rh, rg.x0, rg.y0, rp[0].p, rp[0].w, rp[0].h, 1, 0};
    cj.j[2] = CropJob{lmask, lms, lw, lh, lg.x0, lg.y0, lmx, lp[0].w, lp[0].h, 0, 0};
    cj.j[3] = CropJob{rmask, rms, rw, rh, rg.x0, rg.y0, rmx, rp[0].w, rp[0].h, 0, 0};
    cj.j[4] = CropJob{lmask, lms, lw, lh, bbox.x0, bbox.y0, lmp[0].p, bw, bh, 0, 1};
    cj.j[5] = CropJob{rmask, rms, rw, rh, rmb.x0, rmb.y0, rmp[0].p, rmp[0].w, rmp[0].h, 0, 1};
    const int cmw = std::max(std::max(lp[0].w, rp[0].w), std::max(bw, rmp[0].w)), cmh = std::max(std::max(lp[0].h, rp[0].h), std::max(bh, rmp[0].h));
    dim3 cgrd((cmw + 63) / 64, (cmh + 3) / 4, 6);
    hipLaunchKernelGGL(crop_jobs_kernel, cgrd, kBlk, 0, st, cj);
  }
  // nodata mean fill (:130-149).  Without a user mask the mask of an image crop is all-valid (it is edge-extended like the image — only the
  // bbox crops lmp / rmp are zero outside the image): its mean is not needed and nothing is filled; when that is true of both images the two
  // reductions, the read-back and its host round trip are skipped altogether.
  const bool l_all = !lmask, r_all = !rmask;
  if (l_all && r_all) {
    acc[0] = acc[1] = acc[2] = acc[3] = 1.0;
  } else {
    constexpr int NB = 256;                         // workgroups per image
    double* d_part = A.take<double>(2 * 2 * NB);
    int* d_cell = A.take<int>(8);
    if (!d_part || !d_cell) return fail_mem();
    const int init[8] = {INT_MAX, INT_MIN, 0, 0, INT_MAX, INT_MIN, 0, 0};
    VWGPU_HIP(ctx, hipMemcpyAsync(d_cell, init, sizeof init, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(masked_mean_kernel, dim3(16, 16), kBlk, 0, st, lp[0].p, lmx, lp[0].w, lp[0].h, d_part, d_cell);
    hipLaunchKernelGGL(masked_mean_kernel, dim3(16, 16), kBlk, 0, st, rp[0].p, rmx, rp[0].w, rp[0].h, d_part + 2 * NB, d_cell + 4);
    double part[2 * 2 * NB];
    int cell[8];
    VWGPU_HIP(ctx, hipMemcpyAsync(part, d_part, sizeof part, hipMemcpyDeviceToHost, st));
    VWGPU_HIP(ctx, hipMemcpyAsync(cell, d_cell, sizeof cell, hipMemcpyDeviceToHost, st));
    VWGPU_HIP(ctx, hipStreamSynchronize(st));
    bool serial[2];
    for (int im = 0; im < 2; ++im) {
      double s = 0.0, n = 0.0;
      for (int b = 0; b < NB; ++b) { s += part[(im * NB + b) * 2]; n += part[(im * NB + b) * 2 + 1]; }
      acc[2 * im] = s; acc[2 * im + 1] = n;
      const int lo = cell[4 * im], hi = cell[4 * im + 1];
      int lgn = 0;
      while ((1LL << lgn) < (long long)n + 1) ++lgn;
      // every partial sum of <= n pixels < 2^(hi+1), all multiples of 2^lo, is exact iff it needs <= 53 bits
      serial[im] = cell[4 * im + 2] != 0 || (lo != INT_MAX && (long long)hi + 1 + lgn - lo > 53);
    }
    if (serial[0] || serial[1]) {
      if (serial[0]) hipLaunchKernelGGL(masked_mean_serial_kernel, dim3(1), dim3(64), 0, st, lp[0].p, lmx, lp[0].w, lp[0].h, d_acc);
      if (serial[1]) hipLaunchKernelGGL(masked_mean_serial_kernel, dim3(1), dim3(64), 0, st, rp[0].p, rmx, rp[0].w, rp[0].h, d_acc + 2);
      double got[4];
      VWGPU_HIP(ctx, hipMemcpyAsync(got, d_acc, sizeof got, hipMemcpyDeviceToHost, st));
      VWGPU_HIP(ctx, hipStreamSynchronize(st));
      if (serial[0]) { acc[0] = got[0]; acc[1] = got[1]; }
      if (serial[1]) { acc[2] = got[2]; acc[3] = got[3]; }
    }
  }
  if (acc[1] == 0.0 || acc[3] == 0.0) {            // a fully masked image: the tile has no data (:318-327)
    hipLaunchKernelGGL(zero_out_kernel, grid2(bw, bh), kBlk, 0, st, out, os, bw, bh);
    VWGPU_HIP(ctx, hipGetLastError());
    return VWGPU_OK;
  }
  if (!l_all) hipLaunchKernelGGL(fill_masked_kernel, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, st, lp[0].p, lmx, nl, (float)(acc[0] / acc[1]));
  if (!r_all) hipLaunchKernelGGL(fill_masked_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, st, rp[0].p, rmx, nr, (float)(acc[2] / acc[3]));

  // smoothing + decimation chain and mask decimation (:205-216)
  const float k5[5] = {(float)(1.0 / 16.0), (float)(4.0 / 16.0), (float)(6.0 / 16.0), (float)(4.0 / 16.0), (float)(1.0 / 16.0)};
  for (int i = 1; i <= L; ++i) {
    lp[i].w = 1 + (lp[i - 1].w - 1) / 2; lp[i].h = 1 + (lp[i - 1].h - 1) / 2; lp[i].p = A.take<float>((size_t)lp[i].w * lp[i].h);
    rp[i].w = 1 + (rp[i - 1].w - 1) / 2; rp[i].h = 1 + (rp[i - 1].h - 1) / 2; rp[i].p = A.take<float>((size_t)rp[i].w * rp[i].h);
    lmp[i].w = 1 + (lmp[i - 1].w - 1) / 2; lmp[i].h = 1 + (lmp[i - 1].h - 1) / 2; lmp[i].p = A.take<uint8_t>((size_t)lmp[i].w * lmp[i].h);
    rmp[i].w = 1 + (rmp[i - 1].w - 1) / 2; rmp[i].h = 1 + (rmp[i - 1].h - 1) / 2; rmp[i].p = A.take<uint8_t>((size_t)rmp[i].w * rmp[i].h);
    if (!lp[i].p || !rp[i].p || !lmp[i].p || !rmp[i].p) return fail_mem();
    // both images of the level in one launch, both masks in another (four launches of ~12 us on these sizes otherwise)
    const vwgpu_img_job sj[4] = {{lp[i - 1].p, lp[i - 1].w, lp[i - 1].w, lp[i - 1].h, lp[i].p, lp[i].w, lp[i].w, lp[i].h, 0, 0, nullptr, 0},
                                 {rp[i - 1].p, rp[i - 1].w, rp[i - 1].w, rp[i - 1].h, rp[i].p, rp[i].w, rp[i].w, rp[i].h, 0, 0, nullptr, 0},
                                 {lmp[i - 1].p, lmp[i - 1].w, lmp[i - 1].w, lmp[i - 1].h, lmp[i].p, lmp[i].w, lmp[i].w, lmp[i].h, 0, 0, nullptr, VWGPU_JOB_MASK_BY_TWO},
                                 {rmp[i - 1].p, rmp[i - 1].w, rmp[i - 1].w, rmp[i - 1].h, rmp[i].p, rmp[i].w, rmp[i].w, rmp[i].h, 0, 0, nullptr, VWGPU_JOB_MASK_BY_TWO}};
    if ((rc = vwgpu_launch_sepconv_jobs(ctx, sj, 4, k5, 5, 2, k5, 5, 2, 0, 2))) return rc;      // both images and both masks of the level in one launch
  }
  // prefilter every level (:232-236); levels stay unfiltered sources of the next level, so filter into copies
  const bool use_sgm = P->algorithm != 0;
  // SGM/MGM run without a prefilter (CorrelationView.h:96-97)
  const bool filtered = !use_sgm && (P->prefilter_mode == VWGPU_PREFILTER_LOG || P->prefilter_mode == VWGPU_PREFILTER_MEANSUB);
  if (filtered) {
    // every level of both images through the same two launches (gaussian, then laplacian / subtraction)
    std::vector<const float*> fs; std::vector<float*> fd; std::vector<int> fw, fh;
    for (int i = 0; i <= L; ++i) {
      float* lf = A.take<float>((size_t)lp[i].w * lp[i].h);
      float* rf = A.take<float>((size_t)rp[i].w * rp[i].h);
      if (!lf || !rf) return fail_mem();
      fs.push_back(lp[i].p); fd.push_back(lf); fw.push_back(lp[i].w); fh.push_back(lp[i].h);
      fs.push_back(rp[i].p); fd.push_back(rf); fw.push_back(rp[i].w); fh.push_back(rp[i].h);
      lp[i].p = lf; rp[i].p = rf;
    }
    if ((rc = vwgpu_prefilter_images_dev(ctx, (int)fs.size(), fs.data(), fw.data(), fh.data(), P->prefilter_mode, P->prefilter_width, fd.data()))) return rc;
  }

  // Class of every level: when the box sums of a level could round, its zones are matched in the reference's own
  // summation order (bm_exact.hip); integer imagery and most float imagery are order free and take the tile kernels.
  std::vector<char> exact_level(L + 1, 0), f32_level(L + 1, 0);      // f32_level: the window sums are exact in float32 as well (bm_zones.hip)
  // cert_hi[level] != INT_MIN: the level is not order free but finite — its zones go through the tile kernels first, which certify every pixel
  // against a bound on the difference to the reference's running sums (bm_zones.hip) and flag the zones that hold a pixel they cannot
  // certify; only those are redone in the reference's order (VWGPU_OPT_CERTIFY = 0: every zone of such a level, as in round 3)
  std::vector<int> cert_hi(L + 1, INT_MIN);
  if (!use_sgm) {
    constexpr int CS = 16;                                          // a cache line per level: same-line atomics queue up at the L2
    int* d_cells = A.take<int>(CS * (size_t)(L + 1));
    if (!d_cells) return fail_mem();
    std::vector<int> cells(CS * (size_t)(L + 1), 0);
    for (int i = 0; i <= L; ++i) { cells[CS * i] = INT_MAX; cells[CS * i + 1] = INT_MIN; cells[CS * i + 2] = 0; cells[CS * i + 3] = 0; }
    VWGPU_HIP(ctx, hipMemcpyAsync(d_cells, cells.data(), cells.size() * sizeof(int), hipMemcpyHostToDevice, st));
    {
      std::vector<const float*> gi; std::vector<int> gw, gh; std::vector<ptrdiff_t> gs; std::vector<int*> gc;
      for (int i = 0; i <= L; ++i)
        for (DevImg const* im : {&lp[i], &rp[i]}) { gi.push_back(im->p); gw.push_back(im->w); gh.push_back(im->h); gs.push_back(im->w); gc.push_back(d_cells + CS * i); }
      vwgpu_launch_float_grain(ctx, (int)gi.size(), gi.data(), gw.data(), gh.data(), gs.data(), gc.data());
    }
    VWGPU_HIP(ctx, hipMemcpyAsync(cells.data(), d_cells, cells.size() * sizeof(int), hipMemcpyDeviceToHost, st));
    VWGPU_HIP(ctx, hipStreamSynchronize(st));
    for (int i = 0; i <= L; ++i) {
      exact_level[i] = !vwgpu_sums_order_free(P->cost_type, kx, ky, cells[CS * i], cells[CS * i + 1], cells[CS * i + 2]);
      f32_level[i] = vwgpu_sums_bits(P->cost_type, kx, ky, cells[CS * i], cells[CS * i + 1], cells[CS * i + 2]) <= 24;      // (byte imagery under SAD)
      if (exact_level[i] && ctx->certify && (cells[CS * i + 2] & 1) == 0 && cells[CS * i] != INT_MAX      // (bit 0: a non-finite pixel; bit 1 only says "negative pixels")
          && cells[CS * i + 1] < 60 && cells[CS * i + 1] > -60)
        cert_hi[i] = cells[CS * i + 1];
    }
  }

  // certification statistics (VWGPU_OPT_TRACE bit 2): pixels in certified tiles / in tiles that flagged their zone
  unsigned long long* d_cert_stats = nullptr;
  if (ctx->trace & 4) {
    d_cert_stats = A.take<unsigned long long>(4);
    if (!d_cert_stats) return fail_mem();
    VWGPU_HIP(ctx, hipMemsetAsync(d_cert_stats, 0, 32, st));
  }
  // level loop
  int32_t* disp = A.take<int32_t>((size_t)bw * bh * 3);
  int32_t* disp2 = A.take<int32_t>((size_t)bw * bh * 3);
  int32_t* padded = A.take<int32_t>((size_t)(bw + 2) * (bh + 2) * 3);
  int32_t* rl = A.take<int32_t>((size_t)(rg.dx() + 2 * search.dx()) * (rg.dy() + 2 * search.dy()) * 3 + 64);
  // tmp_a also receives the edge-extended LEFT crop of an R->L run: (zone + 2 * search - 1) per axis, at most the level size + 2 * search
  const size_t tmp_a_px = std::max(nl + nr, (size_t)(lg.dx() + 2 * search.dx() + 2) * (lg.dy() + 2 * search.dy() + 2));
  const size_t tmp_b_px = (size_t)(rg.dx() + 2 * search.dx()) * (rg.dy() + 2 * search.dy());
  float* tmp_a = A.take<float>(tmp_a_px);
  float* tmp_b = A.take<float>(tmp_b_px);
  if (!disp || !disp2 || !padded || !rl || !tmp_a || !tmp_b) return fail_mem();
  int* blob_scratch = nullptr;
  if (P->blob_filter_area > 0) {
    blob_scratch = A.take<int>((size_t)(bw + search.dx() + 8) * (bh + search.dy() + 8) * 2);
    if (!blob_scratch) return fail_mem();
  }
  float* sgm_b = nullptr; float* sub = nullptr; uint8_t* rl_rmask = nullptr; uint8_t* rl_lmask = nullptr;
  int32_t *prev_disp = nullptr, *rl_a = nullptr, *rl_b = nullptr, *rl_pad = nullptr, *prev_rl = nullptr;
  if (use_sgm) {
    sgm_b = A.take<float>(lrev_px); sub = A.take<float>((size_t)bw * bh * 3);
    rl_rmask = A.take<uint8_t>(rl_px); rl_lmask = A.take<uint8_t>(lrev_px);
    prev_disp = A.take<int32_t>((size_t)bw * bh * 3);
    rl_a = A.take<int32_t>(rl_px * 3); rl_b = A.take<int32_t>(rl_px * 3); rl_pad = A.take<int32_t>(rl_px * 3 + 64); prev_rl = A.take<int32_t>(rl_px * 3);
    if (!sgm_b || !sub || !rl_rmask || !rl_lmask || !prev_disp || !rl_a || !rl_b || !rl_pad || !prev_rl) return fail_mem();
  }
  int pdw = 0, pdh = 0, prlw = 0, prlh = 0;
  bool have_prev_rl = false;
  vwgpu_sgm_params SP;
  SP.cost_type = P->cost_type; SP.use_mgm = 0; SP.kernel_size = kx; SP.subpixel_mode = P->sgm_subpixel_mode;
  SP.search_buffer_x = P->sgm_search_buffer_x; SP.search_buffer_y = P->sgm_search_buffer_y; SP.memory_limit_mb = P->memory_limit_mb;
  SP.p1 = 0; SP.p2 = 0; SP.ternary_census_threshold = 5; SP.num_threads = P->sgm_num_threads > 0 ? P->sgm_num_threads : 1;
  SP.allow_block_cost = 0;        // (the pyramid view only ever hands census costs to SGM, CorrelationView.cc:391-403)
  std::vector<vwgpu::LeafExtent> leaf_ext;
  std::vector<SearchZone> zones;
  zones.push_back(SearchZone{IBox(0, 0, lmp[L].w, lmp[L].h), IBox(0, 0, search.width() / up + 1, search.height() / up + 1)});
  // corr_timeout accounting (CorrelationView.cc:285-287,354-357,620-637): the estimate seconds_per_op * search volume is
  // replaced by the wall clock whenever it has advanced by more than measure_spacing (2 s) since the last measurement
  // (std::time: whole seconds), so a seconds_per_op calibrated for the CPU cannot make the GPU quit after milliseconds.
  double estim = 0.0, prev_estim = 0.0;
  const auto t_start = std::chrono::steady_clock::now();
  const bool dbg_time = (ctx->trace & 1) != 0;     // development aid: host-side timeline of a tile on stderr
  auto stamp = [&](const char* what, int level) {
    if (dbg_time) fprintf(stderr, "  [%8.1f us] level %d %s\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_start).count(), level, what);
  };
  auto remeasure = [&]() {
    if (P->corr_timeout > 0 && estim - prev_estim > 2.0) {
      estim = std::floor(std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count());
      prev_estim = estim;
    }
  };
  const int saved_force = ctx->forced_path;
  int dw = 0, dh = 0;
  bool fused_finish = false;
  for (int level = L; level >= 0; --level) {
    const bool last = (level == 0);
    const int scaling = 1 << level;
    stamp("begin", level);
    dw = lmp[level].w; dh = lmp[level].h;
    VWGPU_HIP(ctx, hipMemsetAsync(disp, 0, (size_t)dw * dh * 12, st));
    const int rox = up * hkx / scaling, roy = up * hky / scaling;
    // cheapest zones first (CorrelationView.cc:601-606) — which only matters when a time budget may cut the list: the zones are disjoint
    // boxes, so without one every order gives the same image (and sorting 2000 leaf zones is 80 us with the device waiting)
    if (P->corr_timeout > 0)
      std::stable_sort(zones.begin(), zones.end(), [](SearchZone const& a, SearchZone const& b) { return a.volume() < b.volume(); });
    // dyadic / prefiltered data is not integer-valued: go straight to the float64 matcher there, or to the exact-order
    // kernels when the level's sums could round
    const int level_path = exact_level[level] ? VWGPU_PATH_EXACT_ORDER : ((level > 0 || filtered) ? VWGPU_PATH_GENERIC_F64 : VWGPU_PATH_NONE);
    ctx->forced_path = (saved_force != VWGPU_PATH_NONE) ? saved_force : level_path;
    const DevImg Lv = lp[level], Rv = rp[level];
    const bool lr_active = P->consistency_threshold >= 0 && last;
    // One launch for all zones of the level (bm_zones.hip) unless the kernel is too large for its LDS tiles, or the
    // level is a single big zone (max_pyramid_levels = 0: that is plain calc_disparity and has faster kernels).
    bool check_rl = false;
    int rlw = 0, rlh = 0, lrm_w = 0, lrm_h = 0;
    if (use_sgm) {                                                      // SGM branch (CorrelationView.cc:391-595)
      ctx->forced_path = saved_force;
      const int sx = search.width() / scaling, sy = search.height() / scaling;   // zone.disparity_range().size(), inclusive for SGM
      const IBox lr(rox - hkx, roy - hky, dw + rox + hkx, dh + roy + hky);
      const IBox rr(lr.x0, lr.y0, lr.x1 + sx, lr.y1 + sy);
      if ((size_t)rr.dx() * rr.dy() > tmp_b_px || (size_t)lr.dx() * lr.dy() > tmp_a_px) return fail_mem();
      hipLaunchKernelGGL((crop_ext_kernel<float, 0>), grid2(lr.dx(), lr.dy()), kBlk, 0, st, Lv.p, Lv.w, Lv.w, Lv.h, lr.x0, lr.y0, tmp_a, lr.dx(), lr.dy());
      hipLaunchKernelGGL((crop_ext_kernel<float, 0>), grid2(rr.dx(), rr.dy()), kBlk, 0, st, Rv.p, Rv.w, Rv.w, Rv.h, rr.x0, rr.y0, tmp_b, rr.dx(), rr.dy());
      const bool have_prev = level < L;
      SP.use_mgm = (P->algorithm == 2 || (P->algorithm == 3 && level == 0)) ? 1 : 0;      // CorrelationView.cc:365-366
      int ow = 0, oh = 0;
      rc = vwgpu_sgm_impl(ctx, &SP, tmp_a, lr.dx(), lr.dy(), lr.dx(), tmp_b, rr.dx(), rr.dy(), rr.dx(), sx, sy,
                          lmp[level].p, lmp[level].w, lmp[level].h, rmp[level].p, rmp[level].w, rmp[level].h,
                          have_prev ? prev_disp : nullptr, pdw, pdh, disp, last ? sub : nullptr, (size_t)dw * dh, &ow, &oh);
      if (rc) return rc;
      if (ow != dw || oh != dh) return vwgpu_fail(ctx, VWGPU_ERR_LOGIC, "pyramid_correlate: SGM output %d x %d does not match the level size %d x %d", ow, oh, dw, dh);
      if (P->consistency_threshold >= 0.0f && level >= P->min_consistency_level) {
        check_rl = true;
        const IBox lrev(lr.x0 - sx, lr.y0 - sy, lr.x1 + sx, lr.y1 + sy);          // (left_region - size).max += 2 size
        rlw = rr.dx() - 2 * hkx; rlh = rr.dy() - 2 * hky;
        lrm_w = lrev.dx() - 2 * hkx; lrm_h = lrev.dy() - 2 * hky;
        if ((size_t)lrev.dx() * lrev.dy() > lrev_px || (size_t)rlw * rlh > rl_px) return fail_mem();
        hipLaunchKernelGGL((crop_ext_kernel<uint8_t, 1>), grid2(rlw, rlh), kBlk, 0, st, rmp[level].p, rmp[level].w, rmp[level].w, rmp[level].h, 0, 0, rl_rmask, rlw, rlh);
        hipLaunchKernelGGL((crop_ext_kernel<uint8_t, 1>), grid2(lrm_w, lrm_h), kBlk, 0, st, lmp[level].p, lmp[level].w, lmp[level].w, lmp[level].h, -sx, -sy, rl_lmask, lrm_w, lrm_h);
        hipLaunchKernelGGL((crop_ext_kernel<float, 0>), grid2(lrev.dx(), lrev.dy()), kBlk, 0, st, Lv.p, Lv.w, Lv.w, Lv.h, lrev.x0, lrev.y0, sgm_b, lrev.dx(), lrev.dy());
        int row = 0, roh = 0;
        rc = vwgpu_sgm_impl(ctx, &SP, tmp_b, rr.dx(), rr.dy(), rr.dx(), sgm_b, lrev.dx(), lrev.dy(), lrev.dx(), sx, sy,
                            rl_rmask, rlw, rlh, rl_lmask, lrm_w, lrm_h, (have_prev && have_prev_rl) ? prev_rl : nullptr, prlw, prlh,
                            rl_a, nullptr, rl_px, &row, &roh);
        if (rc) return rc;
        if (row != rlw || roh != rlh) return vwgpu_fail(ctx, VWGPU_ERR_LOGIC, "pyramid_correlate: SGM R->L output size mismatch");
        hipLaunchKernelGGL(add_offset_kernel, grid2(rlw, rlh), kBlk, 0, st, rl_a, rlw, rlw, rlh, -sx, -sy);
        rc = vwgpu_launch_lr_check_diff(ctx, disp, dw, dh, dw, rl_a, rlw, rlh, rlw, P->consistency_threshold, last ? lr_diff : nullptr, lr_stride,
                                        lr_ulx, lr_uly);
        if (rc) return rc;
        hipLaunchKernelGGL(add_offset_kernel, grid2(rlw, rlh), kBlk, 0, st, rl_a, rlw, rlw, rlh, sx, sy);
      }
    }
    bool batched = !use_sgm && vwgpu_bm_zones_supported(kx, ky) && saved_force == VWGPU_PATH_NONE;
    if (batched && zones.size() == 1 && (double)zones[0].region.dx() * zones[0].region.dy() * zones[0].range.dx() * zones[0].range.dy() > 3.2e7)
      batched = false;
    if (batched) {
      std::vector<vwgpu_zone_task> t1, t2, t3;
      t1.reserve(zones.size());
      if (lr_active) { t2.reserve(zones.size()); t3.reserve(zones.size()); }
      size_t rl_pixels = 0;
      for (SearchZone const& z : zones) {
        const IBox lr(z.region.x0 + rox - hkx, z.region.y0 + roy - hky, z.region.x1 + rox + hkx, z.region.y1 + roy + hky);
        const IBox rr(lr.x0 + z.range.x0, lr.y0 + z.range.y0, lr.x1 + z.range.x0 + z.range.dx(), lr.y1 + z.range.y0 + z.range.dy());
        const double next = P->seconds_per_op * ((double)lr.width() * lr.height() * z.range.width() * z.range.height());
        if (P->corr_timeout > 0 && estim + next > P->corr_timeout) break;
        estim += next;
        remeasure();
        const int zw = z.region.dx(), zh = z.region.dy(), sx = z.range.dx(), sy = z.range.dy();
        if (zw <= 0 || zh <= 0 || sx <= 0 || sy <= 0) continue;
        if (zw > 65535 * 32 || zh > 65535 * 32) { ctx->forced_path = saved_force; return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "pyramid_correlate: zone too large"); }
        vwgpu_zone_task a{lr.x0, lr.y0, rr.x0, rr.y0, zw, zh, sx, sy, z.region.y0 * dw + z.region.x0, dw,
                          lr_active ? 0 : z.range.x0, lr_active ? 0 : z.range.y0};
        t1.push_back(a);
        if (lr_active) {
          const double next2 = P->seconds_per_op * ((double)rr.width() * rr.height() * z.range.width() * z.range.height());
          if (P->corr_timeout > 0 && estim + next2 > P->corr_timeout) break;
          estim += next2;
          const int rlw = rr.dx() - kx + 1, rlh = rr.dy() - ky + 1;
          vwgpu_zone_task b{rr.x0, rr.y0, lr.x0 - sx, lr.y0 - sy, rlw, rlh, sx, sy, (int)rl_pixels, rlw, -sx, -sy};
          t2.push_back(b);
          // (sx, sy) slots of an lr task = the zone's origin in the lr_disp_diff image (ul_corner_offset, :683-687)
          vwgpu_zone_task c{(int)rl_pixels, 0, rlw, rlh, zw, zh, z.region.x0 + lr_ulx, z.region.y0 + lr_uly, a.out_off, dw, z.range.x0, z.range.y0};
          t3.push_back(c);
          rl_pixels += (size_t)rlw * rlh;
          if (rl_pixels > (size_t)INT32_MAX / 2) { ctx->forced_path = saved_force; return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "pyramid_correlate: tile too large for the L/R check buffers"); }
        }
      }
      ctx->forced_path = saved_force;
      stamp("zone tables built", level);
      bool exact = exact_level[level] != 0;
      for (vwgpu_zone_task const& z : t1) exact = exact && vwgpu_bm_exact_supported(z.sx, z.sy);
      if (ctx->trace & 2) {           // development aid: the shape of a level's work
        size_t px = 0, ev = 0; int maxd = 0, maxw = 0, maxh = 0;
        for (vwgpu_zone_task const& z : t1) { px += (size_t)z.zw * z.zh; ev += (size_t)z.zw * z.zh * z.sx * z.sy; maxd = std::max(maxd, z.sx * z.sy); maxw = std::max(maxw, z.zw); maxh = std::max(maxh, z.zh); }
        fprintf(stderr, "level %d: %zu zones, %zu px, %zu evaluations, max D %d, max zone %d x %d, exact %d\n", level, t1.size(), px, ev, maxd, maxw, maxh, (int)exact);
        // histogram of zone shapes: (width bucket, height bucket, disparities bucket) -> zones, evaluations
        std::map<std::string, std::pair<size_t, size_t>> hist;
        for (vwgpu_zone_task const& z : t1) {
          char key[96];
          snprintf(key, sizeof key, "w<=%d h<=%d sx<=%d sy<=%d", (z.zw + 15) / 16 * 16, (z.zh + 15) / 16 * 16, (z.sx + 3) / 4 * 4, (z.sy + 3) / 4 * 4);
          hist[key].first += 1; hist[key].second += (size_t)z.zw * z.zh * z.sx * z.sy;
        }
        for (auto const& kv : hist) fprintf(stderr, "    %-34s %6zu zones %12zu evaluations\n", kv.first.c_str(), kv.second.first, kv.second.second);
      }
      if (lr_active && rl_pixels) { if ((rc = vwgpu_arena_reserve(ctx, &ctx->zrl, rl_pixels * 12))) return rc; }
      // One matching pass over a zone list: the tile kernels when the level is order free; the tile kernels WITH certification plus the
      // exact-order kernels gated by the zone flags when it is finite but not order free; the exact-order kernels alone otherwise.
      // The exact-order launches behind a certified pass are queued only if some zone was flagged: the two directions' "any zone
      // flagged" words come back in one small copy (one more host round trip per certified level, hidden by the other tile threads;
      // twelve gated launches of ~6 us that find nothing to do are not — measured 0.25 ms of a 2 ms LoG + NCC tile).
      struct Pending { const float* a; int aw, ah; const float* b; int bw, bh; const std::vector<vwgpu_zone_task>* tz; int32_t* dst; int* zflag; };
      std::vector<Pending> pending;
      int* d_any = nullptr;
      const unsigned char* need_cells = nullptr;
      // The "cannot matter" certificate of the certified passes (bm_zones.hip, ZEdge) and its premises.  L->R: a mask pass follows the
      // clean-up filter (both sit behind filter_half_kernel > 0, CorrelationView.cc:702-744) and erases a pixel that points outside the
      // right image; the filters before it compare disparities within 3 px over +- half kernel, hence the margin; no lr_disp_diff image
      // (its entries keep the discrepancy of pixels the filters removed).  R->L: read by the L/R check alone.
      const int edge_k = rox - hkx;
      // Margin.  Last level: ONE filter pass (rm_outliers, hk x hk, 3 px) — a far pixel M must not be counted by a pixel that
      // can survive the mask, i.e. |dx(M) - dx(N)| > 3 for every N within hk columns whose target lies inside: hk + 3 + 1.  Intermediate
      // levels: disparity_cleanup_using_thresh is TWO passes (inner (hk, hk, 3, 0.5), then outer (1, 1, 3, 0.2) on the inner pass's output,
      // DisparityMap.h:427-441): a kept pixel K (target inside) has an outer-pass neighbour N within 1 column and 3 px — N's target up to 4
      // columns outside — whose inner-pass count sees M within hk columns and 3 px: M's target up to hk + 7 columns outside must still be
      // treated as "can matter", so far starts at hk + 8.
      const int edge_m_lr = (P->filter_half_kernel > 0 && !(last && lr_diff)) ? P->filter_half_kernel + (last ? 4 : 8) : 0;
      const int edge_m_rl = (P->consistency_threshold >= 0 && P->consistency_threshold < 1e6) ? (int)std::floor(P->consistency_threshold) + 2 : 0;
      // columns of the right mask of this level that can be non-zero: the part of the crop inside the right image, halved per level the way
      // subsample_mask_by_two does (a column is kept when one of its two source columns is)
      const int rv0 = std::max(0, -rmb.x0) >> level, rv1 = (std::min(rmp[0].w, rw - rmb.x0) + (1 << level) - 1) >> level;
      const int lr_lo = rv0 - edge_m_lr + 1, lr_hi = rv1 + edge_m_lr - 2;
      const int rl_lo = -edge_m_rl + 1, rl_hi = dw + edge_m_rl - 2;           // the left pixels of the check lie in the tile: [0, dw)
      auto match = [&](const float* a, int aw_, int ah_, const float* b, int bw_, int bh_, std::vector<vwgpu_zone_task> const& tz, int32_t* dst,
                       const int* need, int edge_m, int edge_lo, int edge_hi) -> int {
        if (tz.empty()) return VWGPU_OK;
        if (!exact) return vwgpu_launch_bm_zones(ctx, P->cost_type, a, aw_, ah_, b, bw_, bh_, kx, ky, tz.data(), (int)tz.size(), dst, f32_level[level],
                                                 INT_MIN, nullptr, nullptr, nullptr, need, need ? need_cells : nullptr);
        if (cert_hi[level] != INT_MIN) {
          if (!d_any) {                                            // the two "any zone flagged" words and the zone flags of both passes: one block, one fill
            const size_t nflag = 64 + t1.size() + t2.size();
            d_any = A.take<int>(nflag);
            if (!d_any) return fail_mem();
            VWGPU_HIP(ctx, hipMemsetAsync(d_any, 0, nflag * sizeof(int), st));
          }
          int* zflag = d_any + 64 + (pending.empty() ? 0 : t1.size());
          int rc2 = vwgpu_launch_bm_zones(ctx, P->cost_type, a, aw_, ah_, b, bw_, bh_, kx, ky, tz.data(), (int)tz.size(), dst, 0, cert_hi[level], zflag, d_cert_stats,
                                          d_any + pending.size(), need, need ? need_cells : nullptr, edge_m, edge_k, edge_lo, edge_hi);
          if (rc2) return rc2;
          pending.push_back(Pending{a, aw_, ah_, b, bw_, bh_, &tz, dst, zflag});
          return VWGPU_OK;
        }
        return vwgpu_launch_bm_exact(ctx, P->cost_type, a, aw_, ah_, aw_, b, bw_, bh_, bw_, kx, ky, tz.data(), (int)tz.size(), dst);
      };
      int32_t* rlbuf = static_cast<int32_t*>(ctx->zrl.base);
      if ((rc = match(Lv.p, Lv.w, Lv.h, Rv.p, Rv.w, Rv.h, t1, disp, nullptr, edge_m_lr, lr_lo, lr_hi))) return rc;
      if (lr_active && !t2.empty()) {
        // The R->L pass only has to cover what the L/R check will look at: the positions the L->R disparities point to — about the zone
        // itself instead of the zone widened by its search range (3x the pixels for a 16 x 16 leaf with 32 disparities).  Only where the
        // tile kernels match (their pixels are independent of each other); the exact-order kernels need the whole zone for their sums.
        int* need = nullptr;
        if (!exact || cert_hi[level] != INT_MIN) {
          const size_t ncells = vwgpu_zone_need_cells(t3.data(), (int)t3.size());
          need = A.take<int>(8 * t3.size() + (ncells + 16 + 3) / 4);      // records, then the cell flags: one block, one fill
          if (!need) return fail_mem();
          unsigned char* cells = reinterpret_cast<unsigned char*>(need + 8 * t3.size());
          if ((rc = vwgpu_launch_zone_need(ctx, t3.data(), (int)t3.size(), disp, pending.empty() ? nullptr : pending[0].zflag, need, cells, ncells))) return rc;
          need_cells = cells;
        }
        if ((rc = match(Rv.p, Rv.w, Rv.h, Lv.p, Lv.w, Lv.h, t2, rlbuf, need, edge_m_rl, rl_lo, rl_hi))) return rc;
      }
      if (!pending.empty()) {
        int any[2] = {0, 0};
        VWGPU_HIP(ctx, hipMemcpyAsync(any, d_any, sizeof any, hipMemcpyDeviceToHost, st));
        VWGPU_HIP(ctx, hipStreamSynchronize(st));
        // The flagged zones of a pass as a list of their own (their flags come back in a second small copy, only when there are any): the
        // exact-order launches are then sized by the flagged zones — gated launches over all 2000 zones of a border tile spent 0.1 ms
        // each on workgroups that read a zero and left (18 launches, 2 ms for 3 % of the tile's pixels).
        std::vector<int> hflag;
        for (size_t k = 0; k < pending.size(); ++k) {
          if (!any[k]) continue;
          const Pending& q = pending[k];
          hflag.resize(q.tz->size());
          VWGPU_HIP(ctx, hipMemcpyAsync(hflag.data(), q.zflag, hflag.size() * sizeof(int), hipMemcpyDeviceToHost, st));
          VWGPU_HIP(ctx, hipStreamSynchronize(st));
          std::vector<vwgpu_zone_task> redo;
          for (size_t i = 0; i < hflag.size(); ++i) if (hflag[i]) redo.push_back((*q.tz)[i]);
          if (redo.empty()) continue;
          if (ctx->trace & 4) {
            size_t px = 0, ev = 0;
            for (auto const& zt : redo) { px += (size_t)zt.zw * zt.zh; ev += (size_t)zt.zw * zt.zh * zt.sx * zt.sy; }
            fprintf(stderr, "  level %d pass %zu: %zu of %zu zones to the exact-order kernels, %zu px, %zu evaluations; largest:", level, k, redo.size(), q.tz->size(), px, ev);
            std::vector<vwgpu_zone_task> big(redo);
            std::sort(big.begin(), big.end(), [](vwgpu_zone_task const& a2, vwgpu_zone_task const& b2) { return (size_t)a2.zw * a2.zh * a2.sx * a2.sy > (size_t)b2.zw * b2.zh * b2.sx * b2.sy; });
            for (size_t i = 0; i < std::min<size_t>(4, big.size()); ++i) fprintf(stderr, " %dx%d x %dx%d", big[i].zw, big[i].zh, big[i].sx, big[i].sy);
            fprintf(stderr, "\n");
          }
          if ((rc = vwgpu_launch_bm_exact(ctx, P->cost_type, q.a, q.aw, q.ah, q.aw, q.b, q.bw, q.bh, q.bw, kx, ky, redo.data(), (int)redo.size(), q.dst))) return rc;
        }
      }
      if (lr_active && (rc = vwgpu_launch_zone_lr(ctx, t3.data(), (int)t3.size(), disp, rlbuf, P->consistency_threshold, lr_diff, lr_stride))) return rc;
    } else if (!use_sgm)
    for (SearchZone const& z : zones) {
      const IBox lr(z.region.x0 + rox - hkx, z.region.y0 + roy - hky, z.region.x1 + rox + hkx, z.region.y1 + roy + hky);
      const IBox rr(lr.x0 + z.range.x0, lr.y0 + z.range.y0, lr.x1 + z.range.x0 + z.range.dx(), lr.y1 + z.range.y0 + z.range.dy());
      const double next = P->seconds_per_op * ((double)lr.width() * lr.height() * z.range.width() * z.range.height());
      if (P->corr_timeout > 0 && estim + next > P->corr_timeout) break;
      estim += next;
      remeasure();
      const int zw = z.region.dx(), zh = z.region.dy(), sx = z.range.dx(), sy = z.range.dy();
      if (zw <= 0 || zh <= 0 || sx <= 0 || sy <= 0) continue;
      // the crops normally lie inside the level images; edge-extend into scratch when rounding makes them stick out
      const float* lptr; ptrdiff_t lstr; const float* rptr; ptrdiff_t rstr;
      const int rneed_w = lr.dx() + sx - 1, rneed_h = lr.dy() + sy - 1;
      if ((size_t)lr.dx() * lr.dy() > tmp_a_px || (size_t)rneed_w * rneed_h > tmp_b_px) { ctx->forced_path = saved_force; return fail_mem(); }
      if (lr.x0 >= 0 && lr.y0 >= 0 && lr.x1 <= Lv.w && lr.y1 <= Lv.h) { lptr = Lv.p + (size_t)lr.y0 * Lv.w + lr.x0; lstr = Lv.w; }
      else {
        hipLaunchKernelGGL((crop_ext_kernel<float, 0>), grid2(lr.dx(), lr.dy()), kBlk, 0, st, Lv.p, Lv.w, Lv.w, Lv.h, lr.x0, lr.y0, tmp_a, lr.dx(), lr.dy());
        lptr = tmp_a; lstr = lr.dx();
      }
      if (rr.x0 >= 0 && rr.y0 >= 0 && rr.x0 + rneed_w <= Rv.w && rr.y0 + rneed_h <= Rv.h) { rptr = Rv.p + (size_t)rr.y0 * Rv.w + rr.x0; rstr = Rv.w; }
      else {
        hipLaunchKernelGGL((crop_ext_kernel<float, 0>), grid2(rneed_w, rneed_h), kBlk, 0, st, Rv.p, Rv.w, Rv.w, Rv.h, rr.x0, rr.y0, tmp_b, rneed_w, rneed_h);
        rptr = tmp_b; rstr = rneed_w;
      }
      int32_t* zout = disp + ((size_t)z.region.y0 * dw + z.region.x0) * 3;
      if (saved_force == VWGPU_PATH_NONE && level_path == VWGPU_PATH_EXACT_ORDER)
        ctx->forced_path = vwgpu_bm_exact_supported(sx, sy) ? VWGPU_PATH_EXACT_ORDER : VWGPU_PATH_GENERIC_F64;
      rc = vwgpu_calc_disparity_dev(ctx, P->cost_type, lptr, lr.dx(), lr.dy(), lstr, rptr, rneed_w, rneed_h, rstr, kx, ky, sx, sy, zout, dw);
      if (rc) { ctx->forced_path = saved_force; return rc; }
      if (P->consistency_threshold >= 0 && last) {                    // R->L run + L/R check (:654-694)
        const double next2 = P->seconds_per_op * ((double)rr.width() * rr.height() * z.range.width() * z.range.height());
        if (P->corr_timeout > 0 && estim + next2 > P->corr_timeout) break;
        estim += next2;
        const int aw = rr.dx(), ah = rr.dy();                          // "left" of this run: the right crop (edge-extended)
        const int bw2 = aw + sx - 1, bh2 = ah + sy - 1;                // "right": the left image from (lr.min - s)
        if ((size_t)aw * ah > tmp_b_px || (size_t)bw2 * bh2 > tmp_a_px) { ctx->forced_path = saved_force; return fail_mem(); }
        hipLaunchKernelGGL((crop_ext_kernel<float, 0>), grid2(aw, ah), kBlk, 0, st, Rv.p, Rv.w, Rv.w, Rv.h, rr.x0, rr.y0, tmp_b, aw, ah);
        hipLaunchKernelGGL((crop_ext_kernel<float, 0>), grid2(bw2, bh2), kBlk, 0, st, Lv.p, Lv.w, Lv.w, Lv.h, lr.x0 - sx, lr.y0 - sy, tmp_a, bw2, bh2);
        const int rlw = aw - kx + 1, rlh = ah - ky + 1;
        rc = vwgpu_calc_disparity_dev(ctx, P->cost_type, tmp_b, aw, ah, aw, tmp_a, bw2, bh2, bw2, kx, ky, sx, sy, rl, rlw);
        if (rc) { ctx->forced_path = saved_force; return rc; }
        hipLaunchKernelGGL(add_offset_kernel, grid2(rlw, rlh), kBlk, 0, st, rl, rlw, rlw, rlh, -sx, -sy);
        rc = vwgpu_launch_lr_check_diff(ctx, zout, zw, zh, dw, rl, rlw, rlh, rlw, P->consistency_threshold, lr_diff, lr_stride,
                                        z.region.x0 + lr_ulx, z.region.y0 + lr_uly);
        if (rc) { ctx->forced_path = saved_force; return rc; }
      }
      hipLaunchKernelGGL(add_offset_kernel, grid2(zw, zh), kBlk, 0, st, zout, dw, zw, zh, z.range.x0, z.range.y0);
    }
    ctx->forced_path = saved_force;
    stamp("matchers queued", level);
    // clean-up filters (:702-744)
    // (an intermediate block-matching level: the second filter pass and the mask are applied by the zone scheduler's kernel as it reads)
    const bool fused_extents = !last && !use_sgm && P->filter_half_kernel > 0 && P->blob_filter_area <= 0;
    // (the last level: the mask is applied by the kernel that writes the tile when nothing else reads the masked image)
    fused_finish = last && !use_sgm && P->filter_half_kernel > 0 && P->blob_filter_area <= 0 && !lr_diff;
    if (P->filter_half_kernel > 0) {
      rc = vwgpu_launch_disparity_filter(ctx, disp, dw, dh, P->filter_half_kernel, P->filter_half_kernel, 3.0, 0.5, !last, padded, disp2, fused_extents);
      if (rc) return rc;
      if (!fused_extents) {
        std::swap(disp, disp2);
        if (!fused_finish) {
          rc = vwgpu_launch_disparity_mask(ctx, disp, dw, dh, lmp[level].p, rmp[level].p, rmp[level].w, rmp[level].h);
          if (rc) return rc;
        }
      }
      if (!last && check_rl && use_sgm) {           // the R->L result seeds the next level's R->L run (:722-730)
        rc = vwgpu_launch_disparity_filter(ctx, rl_a, rlw, rlh, P->filter_half_kernel, P->filter_half_kernel, 3.0, 0.5, true, rl_pad, rl_b);
        if (rc) return rc;
        std::swap(rl_a, rl_b);
        rc = vwgpu_launch_disparity_mask(ctx, rl_a, rlw, rlh, rl_rmask, rl_lmask, lrm_w, lrm_h);
        if (rc) return rc;
      }
    }
    // the kernel based filtering tends to leave isolated blobs behind (:746-750)
    if (P->blob_filter_area > 0) {
      const int area = P->blob_filter_area / scaling;
      if ((rc = vwgpu_launch_blob_filter(ctx, disp, dw, dh, area, blob_scratch))) return rc;
      if (check_rl && !last && use_sgm && (rc = vwgpu_launch_blob_filter(ctx, rl_a, rlw, rlh, area, blob_scratch))) return rc;
    }
    if (use_sgm && !last) {                         // prev_disparity / prev_disparity_rl of the next level (:368-371)
      VWGPU_HIP(ctx, hipMemcpyAsync(prev_disp, disp, (size_t)dw * dh * 12, hipMemcpyDeviceToDevice, st));
      pdw = dw; pdh = dh;
      have_prev_rl = check_rl;
      if (check_rl) { std::swap(prev_rl, rl_a); prlw = rlw; prlh = rlh; }
    }
    // zone refinement (:754-799): the scheduler is data dependent host logic
    if (!last && !use_sgm) {
      // The quad tree of boxes is fixed by (dw, dh): its leaves are measured on the device (box + 1-px neighbourhood),
      // only that table comes back, and the accept / retry / merge recursion runs on it (zones.hip).
      const std::vector<vwgpu::IBox>& leaves = vwgpu::cached_leaves(dw, dh);
      const size_t nleaf = leaves.size();
      static_assert(sizeof(vwgpu::IBox) == sizeof(int4), "leaf boxes upload as int4");
      // the leaf boxes of a level size go to the device once per context
      int4* d_rects = nullptr;
      for (auto& lr : ctx->leaf_rects)
        if (lr.w == dw && lr.h == dh && lr.n == nleaf) d_rects = static_cast<int4*>(lr.d_rects);
      if (!d_rects) {
        void* p = nullptr;
        VWGPU_HIP(ctx, hipMalloc(&p, std::max<size_t>(nleaf, 1) * sizeof(int4)));
        VWGPU_HIP(ctx, hipMemcpyAsync(p, leaves.data(), nleaf * sizeof(int4), hipMemcpyHostToDevice, st));
        VWGPU_HIP(ctx, hipStreamSynchronize(st));                 // the host list may be evicted from its cache later
        if (ctx->leaf_rects.size() >= 64) { (void)hipFree(ctx->leaf_rects.front().d_rects); ctx->leaf_rects.erase(ctx->leaf_rects.begin()); }
        ctx->leaf_rects.push_back({dw, dh, nleaf, p});
        d_rects = static_cast<int4*>(p);
      }
      const size_t ext_bytes = nleaf * sizeof(vwgpu::LeafExtent);
      if ((rc = vwgpu_arena_reserve(ctx, &ctx->zext, ext_bytes + 256))) return rc;
      int32_t* d_ext = static_cast<int32_t*>(ctx->zext.base);
      {
        vwgpu_prof_scope ps(ctx, "zone_extents");
        if (fused_extents)
          hipLaunchKernelGGL(zone_extent_fused_kernel, dim3((unsigned)nleaf), dim3(64), 0, st, padded, dw, dh, lmp[level].p, rmp[level].p, rmp[level].w, rmp[level].h,
                             d_rects, (int)nleaf, d_ext);
        else
          hipLaunchKernelGGL(zone_extent_kernel, dim3((unsigned)nleaf), dim3(64), 0, st, disp, dw, dh, d_rects, (int)nleaf, d_ext);
      }
      const vwgpu::LeafExtent* h_ext = static_cast<const vwgpu::LeafExtent*>(vwgpu_host_ring(ctx, ext_bytes));
      if (!h_ext) { leaf_ext.resize(nleaf); h_ext = leaf_ext.data(); }
      VWGPU_HIP(ctx, hipMemcpyAsync(const_cast<vwgpu::LeafExtent*>(h_ext), d_ext, ext_bytes, hipMemcpyDeviceToHost, st));
      stamp("level queued, waiting", level);
      VWGPU_HIP(ctx, hipStreamSynchronize(st));
      stamp("leaf extents on the host", level);
      zones.clear();
      vwgpu::subdivide_regions_from_leaves(dw, dh, kx, ky, h_ext, nleaf, zones);
      const IBox scale_search(0, 0, rp[level - 1].w - lp[level - 1].w, rp[level - 1].h - lp[level - 1].h);
      const IBox next_size(0, 0, lmp[level - 1].w, lmp[level - 1].h);
      for (SearchZone& z : zones) {
        z.region.scale(2);
        z.region.clip(next_size);
        z.range.scale(2);
        z.range.expand(2);
        z.range.clip(scale_search);
        if (z.range.empty()) z.range = IBox(0, 0, search.width(), search.height());
      }
      stamp("zones of the next level ready", level);
    }
  }
  if (dw != bw || dh != bh) return vwgpu_fail(ctx, VWGPU_ERR_LOGIC, "PyramidCorrelation: Solved disparity doesn't match requested bbox size.");
  if (lr_diff) hipLaunchKernelGGL(lr_diff_invalidate_kernel, grid2(bw, bh), kBlk, 0, st, disp, bw, bh, lr_diff, lr_stride, lr_ulx, lr_uly);
  if (use_sgm)
    hipLaunchKernelGGL(finish_sgm_kernel, grid2(bw, bh), kBlk, 0, st, disp, sub, bw, bh, (float)search.x0, (float)search.y0, out, os);
  else if (fused_finish)
    hipLaunchKernelGGL(finish_masked_kernel, grid2(bw, bh), kBlk, 0, st, disp, bw, bh, lmp[0].p, rmp[0].p, rmp[0].w, rmp[0].h, search.x0, search.y0, out, os);
  else
    hipLaunchKernelGGL(finish_kernel, grid2(bw, bh), kBlk, 0, st, disp, bw, bh, search.x0, search.y0, out, os);
  VWGPU_HIP(ctx, hipGetLastError());
  if (d_cert_stats) {
    unsigned long long got[4] = {0, 0, 0, 0};
    VWGPU_HIP(ctx, hipMemcpyAsync(got, d_cert_stats, sizeof got, hipMemcpyDeviceToHost, st));
    VWGPU_HIP(ctx, hipStreamSynchronize(st));
    ctx->cert_px[0] += got[0]; ctx->cert_px[1] += got[1]; ctx->cert_px[2] += got[2];
    fprintf(stderr, "certification: %llu pixels in certified tiles, %llu in tiles that sent their zone to the exact-order kernels (%.2f %%); %llu in tiles "
            "the fp32 tier passed on to float64 (%.2f %%)\n",
            got[0], got[1], got[0] + got[1] ? 100.0 * (double)got[1] / (double)(got[0] + got[1]) : 0.0, got[2],
            got[0] + got[1] ? 100.0 * (double)got[2] / (double)(got[0] + got[1]) : 0.0);
  }
  return VWGPU_OK;
}

// ---- helper threads of a context ----------------------------------------------------------------------------------------
// The host side of a tile group is per-tile work on tables the device sent back (the zone scheduler's accept / retry / merge walk, the zone
// task lists of the next level: ~30 000 zones for 16 tiles at level 0, milliseconds on one thread with the device idle).  A context keeps a
// few helper threads for it: persistent (the zone scheduler caches its trees per thread), parked on a condition variable between jobs.
namespace {
struct HostPool {
  std::vector<std::thread> workers;
  std::mutex m;
  std::condition_variable wake, done;
  void (*fn)(void*, int) = nullptr;
  void* arg = nullptr;
  int n = 0;
  std::atomic<int> next{0};
  int running = 0;                   // workers still inside the current job
  unsigned long long job = 0;        // generation counter
  bool quit = false;
  void work() {
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= n) return;
      fn(arg, i);
    }
  }
  void loop() {
    unsigned long long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(m);
        wake.wait(lk, [&] { return quit || job != seen; });
        if (quit) return;
        seen = job;
      }
      work();
      {
        std::lock_guard<std::mutex> lk(m);
        if (--running == 0) done.notify_one();
      }
    }
  }
};
}  // namespace

void vwgpu_pool_run(vwgpu_ctx* ctx, int n, void (*fn)(void*, int), void* arg) {
  if (n <= 0) return;
  HostPool* P = static_cast<HostPool*>(ctx->host_pool);
  if (!P && n >= 4) {
    P = new HostPool();
    const unsigned hc = std::thread::hardware_concurrency();
    const int nw = hc >= 32 ? 7 : (hc >= 8 ? 3 : 0);               // + the calling thread
    for (int i = 0; i < nw; ++i) P->workers.emplace_back([P] { P->loop(); });
    ctx->host_pool = P;
  }
  if (!P || P->workers.empty() || n < 4) { for (int i = 0; i < n; ++i) fn(arg, i); return; }
  {
    std::lock_guard<std::mutex> lk(P->m);
    P->fn = fn; P->arg = arg; P->n = n; P->next.store(0);
    P->running = (int)P->workers.size();
    ++P->job;
  }
  P->wake.notify_all();
  P->work();
  std::unique_lock<std::mutex> lk(P->m);
  P->done.wait(lk, [&] { return P->running == 0; });
}

void vwgpu_pool_destroy(vwgpu_ctx* ctx) {
  HostPool* P = static_cast<HostPool*>(ctx->host_pool);
  if (!P) return;
  {
    std::lock_guard<std::mutex> lk(P->m);
    P->quit = true;
  }
  P->wake.notify_all();
  for (std::thread& t : P->workers) t.join();
  delete P;
  ctx->host_pool = nullptr;
}

// ---- tile groups (round 5) ------------------------------------------------------------------------------------------
// vwgpu_pyramid_correlate_batch: several tiles of ONE image pair, of equal size, through the level loop TOGETHER.  The reference runs one
// tile per thread (src/vw/Image/ImageIO.h:228-251, BlockProcessor.h:52-176); here a tile is ~50 dependent launches of which the coarse levels
// are pure latency (images of 40 .. 300 pixels), and the device advances only 3 - 4 such chains at once however many tile threads feed it
// (profiles/r04_ubench_streams.txt).  A group of tiles shares every launch: the tiles' buffers have the same sizes (same bbox size, search
// range and kernel) and live in per-tile slices of one arena at a FIXED STRIDE, so a kernel finds tile t's image at base + t * slice
// (blockIdx.z / .y = the tile) and the zone matcher takes the zones of all tiles in one table (vwgpu_zone_group).  One host round trip per
// level serves the whole group: the leaf extents of all tiles come back together and the zone scheduler (zones.hip) runs per tile on them.
// Every tile's result is what vwgpu_pyramid_correlate_impl computes for it alone: same kernels, same arithmetic, same zone lists — the tiles
// never read each other's data.  Eligible: block matching without an lr_disp_diff image, blob filter or time budget; anything else runs
// tile by tile through vwgpu_pyramid_correlate_impl.
namespace {

struct GroupLayout {                       // offsets (bytes) inside a tile's slice of the arena; sizes per level
  std::vector<size_t> lp, rp, lpf, rpf, lmp, rmp;
  std::vector<int> lpw, lph, rpw, rph, lmw, lmh, rmw, rmh;
  size_t lmx = 0, rmx = 0, d_acc = 0, d_part = 0, d_cell = 0, d_cells = 0, disp = 0, disp2 = 0, padded = 0;
  size_t slice = 0;
};

struct GroupTile {
  IBox bbox, lg, rg, rmb;
  bool alive = true;                       // false: a fully masked image (the tile is zeros, CorrelationView.cc:318-327)
  std::vector<SearchZone> zones;
  std::vector<char> exact_level, f32_level;
  std::vector<int> cert_hi;
};

}  // namespace

namespace {
template <class F> void pool_for(vwgpu_ctx* ctx, int n, F& f) {
  vwgpu_pool_run(ctx, n, [](void* a, int i) { (*static_cast<F*>(a))(i); }, &f);
}
}  // namespace

bool vwgpu_pyramid_group_eligible(const vwgpu_ctx* ctx, const vwgpu_pyramid_params* P, int n, const int* bw, const int* bh) {
  if (n < 2 || n > VWGPU_MAX_GROUP) return false;
  if (P->algorithm != 0 || P->lr_disp_diff || P->blob_filter_area > 0 || P->corr_timeout > 0) return false;
  if (ctx->forced_path != VWGPU_PATH_NONE || !vwgpu_bm_zones_supported(P->kernel_x, P->kernel_y)) return false;
  if ((size_t)(32 + 2 * P->filter_half_kernel) * (8 + 2 * P->filter_half_kernel) * 13 > 64 * 1024) return false;      // (the LDS form of the clean-up filter)
  for (int i = 1; i < n; ++i) if (bw[i] != bw[0] || bh[i] != bh[0]) return false;
  return bw[0] > 0 && bh[0] > 0;
}

// All pointers are device pointers (masks may be null).  outs[t]: bw x bh x 3 floats with row stride oss[t] pixels.
int vwgpu_pyramid_group_impl(vwgpu_ctx* ctx, const float* left, int lw, int lh, ptrdiff_t ls, const float* right, int rw, int rh, ptrdiff_t rs,
                             const uint8_t* lmask, ptrdiff_t lms, const uint8_t* rmask, ptrdiff_t rms, const vwgpu_pyramid_params* P,
                             int n, const int* bxs, const int* bys, int bw, int bh, float* const* outs, const ptrdiff_t* oss) {
  const int kx = P->kernel_x, ky = P->kernel_y;
  const IBox search(P->search_min_x, P->search_min_y, P->search_max_x, P->search_max_y);
  hipStream_t st = ctx->stream;
  // number of levels (CorrelationView.h:99-105, CorrelationView.cc:301-310): the same for every tile of the group
  const int largest_search = std::max(search.dx(), search.dy());
  int by_search = (int)(std::floor(std::log(float(largest_search)) / std::log(2.0f)) - 1);
  by_search = std::max(0, std::min(by_search, P->max_pyramid_levels));
  int L = (int)std::floor(std::log((double)std::min(bw, bh)) / std::log(2.0f) - std::log((double)std::max(kx, ky)) / std::log(2.0f));
  L = std::min(L, by_search);
  if (L < 1) L = 0;
  const int hkx = kx / 2, hky = ky / 2, up = 1 << L;
  const bool filtered = P->prefilter_mode == VWGPU_PREFILTER_LOG || P->prefilter_mode == VWGPU_PREFILTER_MEANSUB;
  constexpr int CS = 16, NB = 256;

  const auto t_start = std::chrono::steady_clock::now();
  const bool dbg_time = (ctx->trace & 1) != 0;     // development aid: host-side timeline of a group on stderr
  auto stamp = [&](const char* what, int level) {
    if (dbg_time) fprintf(stderr, "  [%8.1f us] group of %d, level %d: %s\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_start).count(), n, level, what);
  };
  std::vector<GroupTile> T((size_t)n);
  for (int t = 0; t < n; ++t) {
    GroupTile& g = T[t];
    g.bbox = IBox(bxs[t], bys[t], bxs[t] + bw, bys[t] + bh);
    g.lg = g.bbox; g.lg.x0 -= hkx * up; g.lg.x1 += hkx * up; g.lg.y0 -= hky * up; g.lg.y1 += hky * up;
    g.rg = IBox(g.lg.x0 + search.x0, g.lg.y0 + search.y0, g.lg.x1 + search.x0 + search.dx(), g.lg.y1 + search.y0 + search.dy());
    g.rmb = IBox(g.bbox.x0 + search.x0, g.bbox.y0 + search.y0, g.bbox.x1 + search.x0 + search.dx(), g.bbox.y1 + search.y0 + search.dy());
    g.exact_level.assign(L + 1, 0); g.f32_level.assign(L + 1, 0); g.cert_hi.assign(L + 1, INT_MIN);
  }
  // the layout of a tile's slice: the geometry of tile 0 is every tile's
  GroupLayout Y;
  {
    Bump A{nullptr, SIZE_MAX};
    auto off = [&](void* p) { return (size_t)reinterpret_cast<uintptr_t>(p); };
    const IBox &lg = T[0].lg, &rg = T[0].rg, &rmb = T[0].rmb;
    Y.lp.resize(L + 1); Y.rp.resize(L + 1); Y.lpf.resize(L + 1); Y.rpf.resize(L + 1); Y.lmp.resize(L + 1); Y.rmp.resize(L + 1);
    Y.lpw.resize(L + 1); Y.lph.resize(L + 1); Y.rpw.resize(L + 1); Y.rph.resize(L + 1); Y.lmw.resize(L + 1); Y.lmh.resize(L + 1); Y.rmw.resize(L + 1); Y.rmh.resize(L + 1);
    Y.lpw[0] = lg.dx(); Y.lph[0] = lg.dy(); Y.rpw[0] = rg.dx(); Y.rph[0] = rg.dy();
    Y.lmw[0] = bw; Y.lmh[0] = bh; Y.rmw[0] = rmb.dx(); Y.rmh[0] = rmb.dy();
    const size_t nl = (size_t)lg.dx() * lg.dy(), nr = (size_t)rg.dx() * rg.dy();
    A.take<char>(256);                                            // (offset 0 stays unused: a null offset means "not there")
    Y.lp[0] = off(A.take<float>(nl)); Y.rp[0] = off(A.take<float>(nr));
    Y.lmx = off(A.take<uint8_t>(nl)); Y.rmx = off(A.take<uint8_t>(nr));
    Y.lmp[0] = off(A.take<uint8_t>((size_t)bw * bh)); Y.rmp[0] = off(A.take<uint8_t>((size_t)rmb.dx() * rmb.dy()));
    Y.d_acc = off(A.take<double>(4)); Y.d_part = off(A.take<double>(2 * 2 * NB)); Y.d_cell = off(A.take<int>(8));
    for (int i = 1; i <= L; ++i) {
      Y.lpw[i] = 1 + (Y.lpw[i - 1] - 1) / 2; Y.lph[i] = 1 + (Y.lph[i - 1] - 1) / 2; Y.rpw[i] = 1 + (Y.rpw[i - 1] - 1) / 2; Y.rph[i] = 1 + (Y.rph[i - 1] - 1) / 2;
      Y.lmw[i] = 1 + (Y.lmw[i - 1] - 1) / 2; Y.lmh[i] = 1 + (Y.lmh[i - 1] - 1) / 2; Y.rmw[i] = 1 + (Y.rmw[i - 1] - 1) / 2; Y.rmh[i] = 1 + (Y.rmh[i - 1] - 1) / 2;
      Y.lp[i] = off(A.take<float>((size_t)Y.lpw[i] * Y.lph[i])); Y.rp[i] = off(A.take<float>((size_t)Y.rpw[i] * Y.rph[i]));
      Y.lmp[i] = off(A.take<uint8_t>((size_t)Y.lmw[i] * Y.lmh[i])); Y.rmp[i] = off(A.take<uint8_t>((size_t)Y.rmw[i] * Y.rmh[i]));
    }
    for (int i = 0; i <= L; ++i) {
      Y.lpf[i] = filtered ? off(A.take<float>((size_t)Y.lpw[i] * Y.lph[i])) : Y.lp[i];
      Y.rpf[i] = filtered ? off(A.take<float>((size_t)Y.rpw[i] * Y.rph[i])) : Y.rp[i];
    }
    Y.d_cells = off(A.take<int>(CS * (size_t)(L + 1)));
    Y.disp = off(A.take<int32_t>((size_t)bw * bh * 3)); Y.disp2 = off(A.take<int32_t>((size_t)bw * bh * 3));
    Y.padded = off(A.take<int32_t>((size_t)(bw + 2) * (bh + 2) * 3));
    Y.slice = vwgpu_align_up(A.off + 256, 768);                  // a multiple of 12 (a disparity pixel), of 8 and of 256
  }
  // group-level tables behind the slices: the certified passes' flag words, the R->L need records and cells (bounds as in the single-tile arena)
  const size_t tables = (size_t)n * ((size_t)(L + 1) * ((size_t)bw * bh / 64 + 64) * (48 + (size_t)(search.dx() / 16 + 3) * (search.dy() / 16 + 3)) + 8192) + (1 << 16);
  int rc = vwgpu_arena_reserve(ctx, &ctx->pyr, (size_t)n * Y.slice + tables);
  if (rc) return rc;
  char* base = static_cast<char*>(ctx->pyr.base);
  Bump GA{base + (size_t)n * Y.slice, tables};
  auto fail_mem = [&]() { return vwgpu_fail(ctx, VWGPU_ERR_LOGIC, "pyramid_correlate (group): internal arena too small"); };
  auto at = [&](int t, size_t off) -> char* { return base + (size_t)t * Y.slice + off; };
  auto fimg = [&](int t, size_t off) { return reinterpret_cast<float*>(at(t, off)); };
  auto bimg = [&](int t, size_t off) { return reinterpret_cast<uint8_t*>(at(t, off)); };
  const size_t slice_f = Y.slice / 4, slice_i = Y.slice / 4, slice_px = Y.slice / 12;

  // ---- base crops, nodata mean fill (CorrelationView.cc:67-149) ----
  for (int t = 0; t < n; ++t) {
    const GroupTile& g = T[t];
    vwgpu_prof_scope ps(ctx, "pyramid_base_crops");
    CropJobs cj;
    cj.j[0] = CropJob{left, ls, lw, lh, g.lg.x0, g.lg.y0, fimg(t, Y.lp[0]), Y.lpw[0], Y.lph[0], 1, 0};
    cj.j[1] = CropJob{right, rs, rw, rh, g.rg.x0, g.rg.y0, fimg(t, Y.rp[0]), Y.rpw[0], Y.rph[0], 1, 0};
    cj.j[2] = CropJob{lmask, lms, lw, lh, g.lg.x0, g.lg.y0, bimg(t, Y.lmx), Y.lpw[0], Y.lph[0], 0, 0};
    cj.j[3] = CropJob{rmask, rms, rw, rh, g.rg.x0, g.rg.y0, bimg(t, Y.rmx), Y.rpw[0], Y.rph[0], 0, 0};
    cj.j[4] = CropJob{lmask, lms, lw, lh, g.bbox.x0, g.bbox.y0, bimg(t, Y.lmp[0]), bw, bh, 0, 1};
    cj.j[5] = CropJob{rmask, rms, rw, rh, g.rmb.x0, g.rmb.y0, bimg(t, Y.rmp[0]), Y.rmw[0], Y.rmh[0], 0, 1};
    const int cmw = std::max(std::max(Y.lpw[0], Y.rpw[0]), std::max(bw, Y.rmw[0])), cmh = std::max(std::max(Y.lph[0], Y.rph[0]), std::max(bh, Y.rmh[0]));
    hipLaunchKernelGGL(crop_jobs_kernel, dim3((cmw + 63) / 64, (cmh + 3) / 4, 6), kBlk, 0, st, cj);
  }
  const bool l_all = !lmask, r_all = !rmask;
  const size_t nl = (size_t)Y.lpw[0] * Y.lph[0], nr = (size_t)Y.rpw[0] * Y.rph[0];
  if (!(l_all && r_all)) {
    // the reductions of every tile are queued, ONE synchronisation brings all partial sums back (pinned ring; the pageable fallback waits per copy)
    std::vector<double> part((size_t)n * 4 * NB);
    std::vector<int> cell((size_t)n * 8);
    const int init[8] = {INT_MAX, INT_MIN, 0, 0, INT_MAX, INT_MIN, 0, 0};
    for (int t = 0; t < n; ++t) {
      double* d_part = reinterpret_cast<double*>(at(t, Y.d_part));
      int* d_cell = reinterpret_cast<int*>(at(t, Y.d_cell));
      VWGPU_HIP(ctx, hipMemcpyAsync(d_cell, init, sizeof init, hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(masked_mean_kernel, dim3(16, 16), kBlk, 0, st, fimg(t, Y.lp[0]), bimg(t, Y.lmx), Y.lpw[0], Y.lph[0], d_part, d_cell);
      hipLaunchKernelGGL(masked_mean_kernel, dim3(16, 16), kBlk, 0, st, fimg(t, Y.rp[0]), bimg(t, Y.rmx), Y.rpw[0], Y.rph[0], d_part + 2 * NB, d_cell + 4);
      VWGPU_HIP(ctx, hipMemcpyAsync(part.data() + (size_t)t * 4 * NB, d_part, 4 * NB * sizeof(double), hipMemcpyDeviceToHost, st));
      VWGPU_HIP(ctx, hipMemcpyAsync(cell.data() + (size_t)t * 8, d_cell, 8 * sizeof(int), hipMemcpyDeviceToHost, st));
    }
    VWGPU_HIP(ctx, hipStreamSynchronize(st));
    std::vector<double> acc((size_t)n * 4, 0.0);
    std::vector<char> serial((size_t)n * 2, 0);
    bool any_serial = false;
    for (int t = 0; t < n; ++t)
      for (int im = 0; im < 2; ++im) {
        double s_ = 0.0, n_ = 0.0;
        for (int b = 0; b < NB; ++b) { s_ += part[(size_t)t * 4 * NB + (im * NB + b) * 2]; n_ += part[(size_t)t * 4 * NB + (im * NB + b) * 2 + 1]; }
        acc[t * 4 + 2 * im] = s_; acc[t * 4 + 2 * im + 1] = n_;
        const int lo = cell[t * 8 + 4 * im], hi = cell[t * 8 + 4 * im + 1];
        int lgn = 0;
        while ((1LL << lgn) < (long long)n_ + 1) ++lgn;
        serial[t * 2 + im] = cell[t * 8 + 4 * im + 2] != 0 || (lo != INT_MAX && (long long)hi + 1 + lgn - lo > 53);
        any_serial = any_serial || serial[t * 2 + im];
      }
    if (any_serial) {
      std::vector<double> got((size_t)n * 4, 0.0);
      for (int t = 0; t < n; ++t) {
        double* d_acc = reinterpret_cast<double*>(at(t, Y.d_acc));
        if (serial[t * 2]) hipLaunchKernelGGL(masked_mean_serial_kernel, dim3(1), dim3(64), 0, st, fimg(t, Y.lp[0]), bimg(t, Y.lmx), Y.lpw[0], Y.lph[0], d_acc);
        if (serial[t * 2 + 1]) hipLaunchKernelGGL(masked_mean_serial_kernel, dim3(1), dim3(64), 0, st, fimg(t, Y.rp[0]), bimg(t, Y.rmx), Y.rpw[0], Y.rph[0], d_acc + 2);
        if (serial[t * 2] || serial[t * 2 + 1]) VWGPU_HIP(ctx, hipMemcpyAsync(got.data() + (size_t)t * 4, d_acc, 4 * sizeof(double), hipMemcpyDeviceToHost, st));
      }
      VWGPU_HIP(ctx, hipStreamSynchronize(st));
      for (int t = 0; t < n; ++t) {
        if (serial[t * 2]) { acc[t * 4] = got[t * 4]; acc[t * 4 + 1] = got[t * 4 + 1]; }
        if (serial[t * 2 + 1]) { acc[t * 4 + 2] = got[t * 4 + 2]; acc[t * 4 + 3] = got[t * 4 + 3]; }
      }
    }
    for (int t = 0; t < n; ++t) {
      // (without a mask an image's crop mask is all valid: its mean is not needed, as in the single-tile path)
      if ((!l_all && acc[t * 4 + 1] == 0.0) || (!r_all && acc[t * 4 + 3] == 0.0)) { T[t].alive = false; continue; }
      if (!l_all) hipLaunchKernelGGL(fill_masked_kernel, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, st, fimg(t, Y.lp[0]), bimg(t, Y.lmx), nl, (float)(acc[t * 4] / acc[t * 4 + 1]));
      if (!r_all) hipLaunchKernelGGL(fill_masked_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, st, fimg(t, Y.rp[0]), bimg(t, Y.rmx), nr, (float)(acc[t * 4 + 2] / acc[t * 4 + 3]));
    }
  }
  // ---- smoothing + decimation chain, mask decimation (:205-216): the four jobs of every tile of a level, VWGPU_MAX_IMG_JOBS per launch ----
  const float k5[5] = {(float)(1.0 / 16.0), (float)(4.0 / 16.0), (float)(6.0 / 16.0), (float)(4.0 / 16.0), (float)(1.0 / 16.0)};
  for (int i = 1; i <= L; ++i) {
    std::vector<vwgpu_img_job> jobs;
    for (int t = 0; t < n; ++t) {
      jobs.push_back({fimg(t, Y.lp[i - 1]), Y.lpw[i - 1], Y.lpw[i - 1], Y.lph[i - 1], fimg(t, Y.lp[i]), Y.lpw[i], Y.lpw[i], Y.lph[i], 0, 0, nullptr, 0});
      jobs.push_back({fimg(t, Y.rp[i - 1]), Y.rpw[i - 1], Y.rpw[i - 1], Y.rph[i - 1], fimg(t, Y.rp[i]), Y.rpw[i], Y.rpw[i], Y.rph[i], 0, 0, nullptr, 0});
      jobs.push_back({bimg(t, Y.lmp[i - 1]), Y.lmw[i - 1], Y.lmw[i - 1], Y.lmh[i - 1], bimg(t, Y.lmp[i]), Y.lmw[i], Y.lmw[i], Y.lmh[i], 0, 0, nullptr, VWGPU_JOB_MASK_BY_TWO});
      jobs.push_back({bimg(t, Y.rmp[i - 1]), Y.rmw[i - 1], Y.rmw[i - 1], Y.rmh[i - 1], bimg(t, Y.rmp[i]), Y.rmw[i], Y.rmw[i], Y.rmh[i], 0, 0, nullptr, VWGPU_JOB_MASK_BY_TWO});
    }
    for (size_t j0 = 0; j0 < jobs.size(); j0 += VWGPU_MAX_IMG_JOBS)
      if ((rc = vwgpu_launch_sepconv_jobs(ctx, jobs.data() + j0, (int)std::min<size_t>(VWGPU_MAX_IMG_JOBS, jobs.size() - j0), k5, 5, 2, k5, 5, 2, 0, 2))) return rc;
  }
  // ---- prefilter every level into copies (:232-236), level by level so that a launch holds images of one size ----
  if (filtered) {
    std::vector<const float*> fs; std::vector<float*> fd; std::vector<int> fw, fh;
    for (int i = 0; i <= L; ++i)
      for (int t = 0; t < n; ++t) {
        fs.push_back(fimg(t, Y.lp[i])); fd.push_back(fimg(t, Y.lpf[i])); fw.push_back(Y.lpw[i]); fh.push_back(Y.lph[i]);
        fs.push_back(fimg(t, Y.rp[i])); fd.push_back(fimg(t, Y.rpf[i])); fw.push_back(Y.rpw[i]); fh.push_back(Y.rph[i]);
      }
    if ((rc = vwgpu_prefilter_images_dev(ctx, (int)fs.size(), fs.data(), fw.data(), fh.data(), P->prefilter_mode, P->prefilter_width, fd.data()))) return rc;
  }
  // ---- class of every level of every tile: ONE read-back for the group ----
  {
    std::vector<int> cells((size_t)n * CS * (L + 1), 0);
    for (int t = 0; t < n; ++t)
      for (int i = 0; i <= L; ++i) { int* c = cells.data() + ((size_t)t * (L + 1) + i) * CS; c[0] = INT_MAX; c[1] = INT_MIN; }
    std::vector<const float*> gi; std::vector<int> gw, gh; std::vector<ptrdiff_t> gs; std::vector<int*> gc;
    for (int t = 0; t < n; ++t) {
      int* d_cells = reinterpret_cast<int*>(at(t, Y.d_cells));
      VWGPU_HIP(ctx, hipMemcpyAsync(d_cells, cells.data() + (size_t)t * (L + 1) * CS, (size_t)(L + 1) * CS * sizeof(int), hipMemcpyHostToDevice, st));
    }
    for (int i = 0; i <= L; ++i)
      for (int t = 0; t < n; ++t) {
        int* d_cells = reinterpret_cast<int*>(at(t, Y.d_cells));
        gi.push_back(fimg(t, Y.lpf[i])); gw.push_back(Y.lpw[i]); gh.push_back(Y.lph[i]); gs.push_back(Y.lpw[i]); gc.push_back(d_cells + CS * i);
        gi.push_back(fimg(t, Y.rpf[i])); gw.push_back(Y.rpw[i]); gh.push_back(Y.rph[i]); gs.push_back(Y.rpw[i]); gc.push_back(d_cells + CS * i);
      }
    vwgpu_launch_float_grain(ctx, (int)gi.size(), gi.data(), gw.data(), gh.data(), gs.data(), gc.data());
    for (int t = 0; t < n; ++t)
      VWGPU_HIP(ctx, hipMemcpyAsync(cells.data() + (size_t)t * (L + 1) * CS, at(t, Y.d_cells), (size_t)(L + 1) * CS * sizeof(int), hipMemcpyDeviceToHost, st));
    stamp("pyramids queued, waiting for the level classes", L);
    VWGPU_HIP(ctx, hipStreamSynchronize(st));
    stamp("level classes on the host", L);
    for (int t = 0; t < n; ++t)
      for (int i = 0; i <= L; ++i) {
        const int* c = cells.data() + ((size_t)t * (L + 1) + i) * CS;
        T[t].exact_level[i] = !vwgpu_sums_order_free(P->cost_type, kx, ky, c[0], c[1], c[2]);
        T[t].f32_level[i] = vwgpu_sums_bits(P->cost_type, kx, ky, c[0], c[1], c[2]) <= 24;
        if (T[t].exact_level[i] && ctx->certify && (c[2] & 1) == 0 && c[0] != INT_MAX && c[1] < 60 && c[1] > -60) T[t].cert_hi[i] = c[1];
      }
  }
  unsigned long long* d_cert_stats = nullptr;
  if (ctx->trace & 4) {
    d_cert_stats = GA.take<unsigned long long>(4);
    if (!d_cert_stats) return fail_mem();
    VWGPU_HIP(ctx, hipMemsetAsync(d_cert_stats, 0, 32, st));
  }
  for (int t = 0; t < n; ++t)
    if (T[t].alive) T[t].zones.push_back(SearchZone{IBox(0, 0, Y.lmw[L], Y.lmh[L]), IBox(0, 0, search.width() / up + 1, search.height() / up + 1)});

  // ---- level loop ----
  size_t disp_off = Y.disp, disp2_off = Y.disp2;                   // (the two disparity buffers of every tile swap together)
  int dw = 0, dh = 0;
  for (int level = L; level >= 0; --level) {
    const bool last = (level == 0);
    const int scaling = 1 << level;
    dw = Y.lmw[level]; dh = Y.lmh[level];
    {
      const size_t words = ((size_t)dw * dh * 12 + 15) / 16;      // (the buffers are 256-byte aligned and sized for the largest level; rounding up stays inside)
      hipLaunchKernelGGL(zero_tiles_kernel, dim3((unsigned)std::min<size_t>(512, (words + 255) / 256), (unsigned)n), dim3(256), 0, st,
                         reinterpret_cast<uint4*>(at(0, disp_off)), Y.slice / 16, words);
    }
    int32_t* disp0 = reinterpret_cast<int32_t*>(at(0, disp_off));
    const int rox = up * hkx / scaling, roy = up * hky / scaling;
    const bool lr_active = P->consistency_threshold >= 0 && last;
    const int aw_ = Y.lpw[level], ah_ = Y.lph[level], bw_ = Y.rpw[level], bh_ = Y.rph[level];
    const float* A0 = fimg(0, Y.lpf[level]);
    const float* B0 = fimg(0, Y.rpf[level]);
    // zone tasks of every tile, sorted into the kernel classes of this level: 0 = order free with float32 sums, 1 = order free, 2 = certified,
    // 3 = the reference's order only (a non-finite pixel, or VWGPU_OPT_CERTIFY = 0)
    std::vector<vwgpu_zone_task> t1[4], t2[4], t3[4];
    std::vector<int> cls((size_t)n, -1);
    size_t rl_pixels = 0;
    // per tile on the helper threads (R->L offsets local to the tile), then strung together per class
    struct TileTasks { std::vector<vwgpu_zone_task> a, b, c; size_t rl = 0; int err = 0; };
    std::vector<TileTasks> TT((size_t)n);
    auto build_tile = [&](int t) {
      GroupTile& g = T[t];
      TileTasks& tt = TT[t];
      if (!g.alive) return;
      bool exact = g.exact_level[level] != 0;
      for (SearchZone const& z : g.zones)
        if (z.range.dx() > 0 && z.range.dy() > 0) exact = exact && vwgpu_bm_exact_supported(z.range.dx(), z.range.dy());      // (as the single-tile path: the tile kernels otherwise)
      cls[t] = exact ? (g.cert_hi[level] != INT_MIN ? 2 : 3) : (g.f32_level[level] ? 0 : 1);
      tt.a.reserve(g.zones.size());
      if (lr_active) { tt.b.reserve(g.zones.size()); tt.c.reserve(g.zones.size()); }
      for (SearchZone const& z : g.zones) {
        const IBox lr(z.region.x0 + rox - hkx, z.region.y0 + roy - hky, z.region.x1 + rox + hkx, z.region.y1 + roy + hky);
        const IBox rr(lr.x0 + z.range.x0, lr.y0 + z.range.y0, lr.x1 + z.range.x0 + z.range.dx(), lr.y1 + z.range.y0 + z.range.dy());
        const int zw = z.region.dx(), zh = z.region.dy(), sx = z.range.dx(), sy = z.range.dy();
        if (zw <= 0 || zh <= 0 || sx <= 0 || sy <= 0) continue;
        if (zw > 65535 * 32 || zh > 65535 * 32) { tt.err = 1; return; }
        const long long out_off = (long long)t * (long long)slice_px + (long long)z.region.y0 * dw + z.region.x0;
        if (out_off > INT32_MAX) { tt.err = 2; return; }
        vwgpu_zone_task a{lr.x0, lr.y0, rr.x0, rr.y0, zw, zh, sx, sy, (int)out_off, dw, lr_active ? 0 : z.range.x0, lr_active ? 0 : z.range.y0, t};
        tt.a.push_back(a);
        if (lr_active) {
          const int rlw = rr.dx() - kx + 1, rlh = rr.dy() - ky + 1;
          vwgpu_zone_task b{rr.x0, rr.y0, lr.x0 - sx, lr.y0 - sy, rlw, rlh, sx, sy, (int)tt.rl, rlw, -sx, -sy, t};
          tt.b.push_back(b);
          vwgpu_zone_task cc{(int)tt.rl, 0, rlw, rlh, zw, zh, 0, 0, a.out_off, dw, z.range.x0, z.range.y0, t};
          tt.c.push_back(cc);
          tt.rl += (size_t)rlw * rlh;
          if (tt.rl > (size_t)INT32_MAX / 2) { tt.err = 3; return; }
        }
      }
    };
    {
      size_t nz = 0;
      for (int t = 0; t < n; ++t) nz += T[t].zones.size();
      if (nz >= 1024) pool_for(ctx, n, build_tile);                 // (waking the helpers costs tens of microseconds: not for the coarse levels)
      else for (int t = 0; t < n; ++t) build_tile(t);
    }
    for (int t = 0; t < n; ++t) {
      TileTasks& tt = TT[t];
      if (tt.err == 1) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "pyramid_correlate: zone too large");
      if (tt.err == 2) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "pyramid_correlate: tile group too large for the zone tables");
      if (tt.err == 3 || rl_pixels + tt.rl > (size_t)INT32_MAX / 2) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "pyramid_correlate: tile group too large for the L/R check buffers");
      if (cls[t] < 0) continue;
      const int c = cls[t], base = (int)rl_pixels;
      t1[c].insert(t1[c].end(), tt.a.begin(), tt.a.end());
      for (vwgpu_zone_task& b : tt.b) b.out_off += base;
      for (vwgpu_zone_task& cc : tt.c) cc.ax += base;
      t2[c].insert(t2[c].end(), tt.b.begin(), tt.b.end());
      t3[c].insert(t3[c].end(), tt.c.begin(), tt.c.end());
      rl_pixels += tt.rl;
    }
    stamp("zone tasks built", level);
    if (lr_active && rl_pixels) { if ((rc = vwgpu_arena_reserve(ctx, &ctx->zrl, rl_pixels * 12))) return rc; }
    int32_t* rlbuf = static_cast<int32_t*>(ctx->zrl.base);
    // the "cannot matter" certificate (see vwgpu_pyramid_correlate_impl): margins per level, bounds per tile
    const int edge_k = rox - hkx;
    const int edge_m_lr = P->filter_half_kernel > 0 ? P->filter_half_kernel + (last ? 4 : 8) : 0;
    const int edge_m_rl = (P->consistency_threshold >= 0 && P->consistency_threshold < 1e6) ? (int)std::floor(P->consistency_threshold) + 2 : 0;
    std::vector<int> lr_lo((size_t)n), lr_hi((size_t)n), rl_lo((size_t)n, -edge_m_rl + 1), rl_hi((size_t)n, dw + edge_m_rl - 2), hi_img((size_t)n, INT_MIN);
    for (int t = 0; t < n; ++t) {
      const int rv0 = std::max(0, -T[t].rmb.x0) >> level, rv1 = (std::min(Y.rmw[0], rw - T[t].rmb.x0) + (1 << level) - 1) >> level;
      lr_lo[t] = rv0 - edge_m_lr + 1; lr_hi[t] = rv1 + edge_m_lr - 2;
      hi_img[t] = T[t].cert_hi[level];
    }
    vwgpu_zone_group grp;
    grp.n_img = n; grp.a_stride = slice_f; grp.b_stride = slice_f;
    // certified classes: flag words {any per pass and tile: 2 n} + zone flags of both passes, one block, one fill
    int* d_any = nullptr; int* zflag1 = nullptr; int* zflag2 = nullptr;
    if (!t1[2].empty()) {
      const size_t nflag = 64 + 2 * (size_t)n + t1[2].size() + t2[2].size();
      d_any = GA.take<int>(nflag);
      if (!d_any) return fail_mem();
      VWGPU_HIP(ctx, hipMemsetAsync(d_any, 0, nflag * sizeof(int), st));
      zflag1 = d_any + 64 + 2 * n; zflag2 = zflag1 + t1[2].size();
    }
    for (int c = 0; c < 4; ++c) {
      if (t1[c].empty()) continue;
      if (c == 3) {                                              // the reference's order: tile by tile (zones of one image pair per call)
        for (int t = 0; t < n; ++t) {
          if (cls[t] != 3) continue;
          std::vector<vwgpu_zone_task> za, zb;
          for (auto const& z : t1[3]) if (z.img == t) za.push_back(z);
          for (auto const& z : t2[3]) if (z.img == t) zb.push_back(z);
          const float* At = A0 + (size_t)t * slice_f; const float* Bt = B0 + (size_t)t * slice_f;
          if (!za.empty() && (rc = vwgpu_launch_bm_exact(ctx, P->cost_type, At, aw_, ah_, aw_, Bt, bw_, bh_, bw_, kx, ky, za.data(), (int)za.size(), disp0))) return rc;
          if (!zb.empty() && (rc = vwgpu_launch_bm_exact(ctx, P->cost_type, Bt, bw_, bh_, bw_, At, aw_, ah_, aw_, kx, ky, zb.data(), (int)zb.size(), rlbuf))) return rc;
        }
        continue;
      }
      const bool cert = c == 2;
      grp.cert_hi = cert ? hi_img.data() : nullptr;
      grp.edge_lo = lr_lo.data(); grp.edge_hi = lr_hi.data();
      rc = vwgpu_launch_bm_zones(ctx, P->cost_type, A0, aw_, ah_, B0, bw_, bh_, kx, ky, t1[c].data(), (int)t1[c].size(), disp0, c == 0 ? 1 : 0,
                                 cert ? 0 : INT_MIN, cert ? zflag1 : nullptr, cert ? d_cert_stats : nullptr, cert ? d_any : nullptr, nullptr, nullptr,
                                 cert ? edge_m_lr : 0, edge_k, 0, 0, 0, 0, &grp);
      if (rc) return rc;
      if (lr_active && !t2[c].empty()) {
        const size_t ncells = vwgpu_zone_need_cells(t3[c].data(), (int)t3[c].size());
        int* need = GA.take<int>(8 * t3[c].size() + (ncells + 16 + 3) / 4);
        if (!need) return fail_mem();
        unsigned char* cells = reinterpret_cast<unsigned char*>(need + 8 * t3[c].size());
        if ((rc = vwgpu_launch_zone_need(ctx, t3[c].data(), (int)t3[c].size(), disp0, cert ? zflag1 : nullptr, need, cells, ncells))) return rc;
        std::swap(grp.a_stride, grp.b_stride);                    // (equal anyway: the R->L pass matches the right image against the left)
        grp.edge_lo = rl_lo.data(); grp.edge_hi = rl_hi.data();
        rc = vwgpu_launch_bm_zones(ctx, P->cost_type, B0, bw_, bh_, A0, aw_, ah_, kx, ky, t2[c].data(), (int)t2[c].size(), rlbuf, c == 0 ? 1 : 0,
                                   cert ? 0 : INT_MIN, cert ? zflag2 : nullptr, cert ? d_cert_stats : nullptr, cert ? d_any + n : nullptr, need, cells,
                                   cert ? edge_m_rl : 0, edge_k, 0, 0, 0, 0, &grp);
        if (rc) return rc;
      }
    }
    stamp("matchers queued", level);
    if (d_any) {
      // the flagged zones of the certified passes, per tile: one small read-back for the group, a second one only when some zone was flagged
      std::vector<int> any(2 * (size_t)n, 0);
      VWGPU_HIP(ctx, hipMemcpyAsync(any.data(), d_any, any.size() * sizeof(int), hipMemcpyDeviceToHost, st));
      VWGPU_HIP(ctx, hipStreamSynchronize(st));
      bool some[2] = {false, false};
      for (int t = 0; t < n; ++t) { some[0] = some[0] || any[t]; some[1] = some[1] || any[n + t]; }
      for (int pass = 0; pass < 2; ++pass) {
        if (!some[pass]) continue;
        const std::vector<vwgpu_zone_task>& tz = pass ? t2[2] : t1[2];
        std::vector<int> hflag(tz.size());
        VWGPU_HIP(ctx, hipMemcpyAsync(hflag.data(), pass ? zflag2 : zflag1, hflag.size() * sizeof(int), hipMemcpyDeviceToHost, st));
        VWGPU_HIP(ctx, hipStreamSynchronize(st));
        for (int t = 0; t < n; ++t) {
          if (!any[pass * n + t]) continue;
          std::vector<vwgpu_zone_task> redo;
          for (size_t i = 0; i < tz.size(); ++i) if (hflag[i] && tz[i].img == t) redo.push_back(tz[i]);
          if (redo.empty()) continue;
          const float* At = A0 + (size_t)t * slice_f; const float* Bt = B0 + (size_t)t * slice_f;
          if (pass == 0) rc = vwgpu_launch_bm_exact(ctx, P->cost_type, At, aw_, ah_, aw_, Bt, bw_, bh_, bw_, kx, ky, redo.data(), (int)redo.size(), disp0);
          else rc = vwgpu_launch_bm_exact(ctx, P->cost_type, Bt, bw_, bh_, bw_, At, aw_, ah_, aw_, kx, ky, redo.data(), (int)redo.size(), rlbuf);
          if (rc) return rc;
        }
      }
    }
    if (lr_active)
      for (int c = 0; c < 4; ++c)
        if (!t3[c].empty() && (rc = vwgpu_launch_zone_lr(ctx, t3[c].data(), (int)t3[c].size(), disp0, rlbuf, P->consistency_threshold, nullptr, 0))) return rc;
    // ---- clean-up filters (:702-744) ----
    int32_t* disp2_0 = reinterpret_cast<int32_t*>(at(0, disp2_off));
    int32_t* padded0 = reinterpret_cast<int32_t*>(at(0, Y.padded));
    const bool fused_extents = !last && P->filter_half_kernel > 0;       // second filter pass + mask applied by the zone scheduler's kernel as it reads
    if (P->filter_half_kernel > 0) {
      rc = vwgpu_launch_disparity_filter(ctx, disp0, dw, dh, P->filter_half_kernel, P->filter_half_kernel, 3.0, 0.5, !last, padded0, disp2_0, true, n, slice_i);
      if (rc) return rc;
      if (last) std::swap(disp_off, disp2_off);
    }
    // ---- zone refinement (:754-799): leaf extents of every tile in one launch, one copy, one synchronisation ----
    if (!last) {
      const std::vector<vwgpu::IBox>& leaves = vwgpu::cached_leaves(dw, dh);
      const size_t nleaf = leaves.size();
      int4* d_rects = nullptr;
      for (auto& lr : ctx->leaf_rects)
        if (lr.w == dw && lr.h == dh && lr.n == nleaf) d_rects = static_cast<int4*>(lr.d_rects);
      if (!d_rects) {
        void* p = nullptr;
        VWGPU_HIP(ctx, hipMalloc(&p, std::max<size_t>(nleaf, 1) * sizeof(int4)));
        VWGPU_HIP(ctx, hipMemcpyAsync(p, leaves.data(), nleaf * sizeof(int4), hipMemcpyHostToDevice, st));
        VWGPU_HIP(ctx, hipStreamSynchronize(st));
        if (ctx->leaf_rects.size() >= 64) { (void)hipFree(ctx->leaf_rects.front().d_rects); ctx->leaf_rects.erase(ctx->leaf_rects.begin()); }
        ctx->leaf_rects.push_back({dw, dh, nleaf, p});
        d_rects = static_cast<int4*>(p);
      }
      const size_t ext_bytes = (size_t)n * nleaf * sizeof(vwgpu::LeafExtent);
      if ((rc = vwgpu_arena_reserve(ctx, &ctx->zext, ext_bytes + 256))) return rc;
      int32_t* d_ext = static_cast<int32_t*>(ctx->zext.base);
      {
        vwgpu_prof_scope ps(ctx, "zone_extents");
        if (fused_extents)
          hipLaunchKernelGGL(zone_extent_fused_kernel, dim3((unsigned)nleaf, (unsigned)n), dim3(64), 0, st, padded0, dw, dh, bimg(0, Y.lmp[level]), bimg(0, Y.rmp[level]),
                             Y.rmw[level], Y.rmh[level], d_rects, (int)nleaf, d_ext, slice_i, Y.slice);
        else
          hipLaunchKernelGGL(zone_extent_kernel, dim3((unsigned)nleaf, (unsigned)n), dim3(64), 0, st, reinterpret_cast<const int32_t*>(at(0, disp_off)), dw, dh, d_rects, (int)nleaf, d_ext, slice_i);
      }
      std::vector<vwgpu::LeafExtent> own;
      const vwgpu::LeafExtent* h_ext = static_cast<const vwgpu::LeafExtent*>(vwgpu_host_ring(ctx, ext_bytes));
      if (!h_ext) { own.resize((size_t)n * nleaf); h_ext = own.data(); }
      VWGPU_HIP(ctx, hipMemcpyAsync(const_cast<vwgpu::LeafExtent*>(h_ext), d_ext, ext_bytes, hipMemcpyDeviceToHost, st));
      stamp("level queued, waiting", level);
      VWGPU_HIP(ctx, hipStreamSynchronize(st));
      stamp("leaf extents on the host", level);
      const IBox scale_search(0, 0, Y.rpw[level - 1] - Y.lpw[level - 1], Y.rph[level - 1] - Y.lph[level - 1]);
      const IBox next_size(0, 0, Y.lmw[level - 1], Y.lmh[level - 1]);
      auto refine_tile = [&](int t) {                               // (helper threads: the scheduler keeps its tree of boxes per thread)
        if (!T[t].alive) return;
        std::vector<SearchZone>& zones = T[t].zones;
        zones.clear();
        vwgpu::subdivide_regions_from_leaves(dw, dh, kx, ky, h_ext + (size_t)t * nleaf, nleaf, zones);
        for (SearchZone& z : zones) {
          z.region.scale(2);
          z.region.clip(next_size);
          z.range.scale(2);
          z.range.expand(2);
          z.range.clip(scale_search);
          if (z.range.empty()) z.range = IBox(0, 0, search.width(), search.height());
        }
      };
      if (nleaf * (size_t)n >= 2048) pool_for(ctx, n, refine_tile);
      else for (int t = 0; t < n; ++t) refine_tile(t);
      stamp("zones of the next level ready", level);
    }
  }
  stamp("last level queued", 0);
  if (dw != bw || dh != bh) return vwgpu_fail(ctx, VWGPU_ERR_LOGIC, "PyramidCorrelation: Solved disparity doesn't match requested bbox size.");
  {
    GroupOuts go;
    for (int t = 0; t < VWGPU_MAX_GROUP; ++t) { go.out[t] = nullptr; go.os[t] = 0; go.mode[t] = 0; }
    for (int t = 0; t < n; ++t) { go.out[t] = outs[t]; go.os[t] = oss[t]; go.mode[t] = !T[t].alive ? 3 : (P->filter_half_kernel > 0 ? 2 : 1); }
    hipLaunchKernelGGL(finish_group_kernel, dim3((bw + 63) / 64, (bh + 3) / 4, (unsigned)n), kBlk, 0, st, go, reinterpret_cast<const int32_t*>(at(0, disp_off)), slice_i, bw, bh,
                       bimg(0, Y.lmp[0]), bimg(0, Y.rmp[0]), Y.slice, Y.rmw[0], Y.rmh[0], search.x0, search.y0);
  }
  VWGPU_HIP(ctx, hipGetLastError());
  if (d_cert_stats) {
    unsigned long long got[4] = {0, 0, 0, 0};
    VWGPU_HIP(ctx, hipMemcpyAsync(got, d_cert_stats, sizeof got, hipMemcpyDeviceToHost, st));
    VWGPU_HIP(ctx, hipStreamSynchronize(st));
    ctx->cert_px[0] += got[0]; ctx->cert_px[1] += got[1]; ctx->cert_px[2] += got[2];
  }
  return VWGPU_OK;
}

// ---- extern "C" entry points ------------------------------------------------------------------------------------

extern "C" {

int vwgpu_subdivide_regions(const int32_t* disp, int w, int h, int kx, int ky, int32_t* zones, int cap) {
  if (!disp || w <= 0 || h <= 0 || kx < 1 || ky < 1 || cap < 0 || (cap && !zones)) return VWGPU_ERR_ARGUMENT;
  std::vector<SearchZone> list;
  vwgpu::subdivide_regions(disp, w, h, kx, ky, list);
  int n = 0;
  for (SearchZone const& z : list) {
    if (n < cap) {
      int32_t* o = zones + (size_t)n * 8;
      o[0] = z.region.x0; o[1] = z.region.y0; o[2] = z.region.x1; o[3] = z.region.y1;
      o[4] = z.range.x0; o[5] = z.range.y0; o[6] = z.range.x1; o[7] = z.range.y1;
    }
    ++n;
  }
  return n;
}

int vwgpu_disparity_filter_dev(vwgpu_ctx* ctx, const int32_t* d_src, int w, int h, int hh, int hv,
                               double pthr, double rthr, int cleanup, int32_t* d_dst) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  ctx->err.clear();
  if (!d_src || !d_dst || w <= 0 || h <= 0 || d_src == d_dst) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "rm_outliers_using_thresh: bad image arguments");
  if (hh <= 0 || hv <= 0) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "RmOutliersFunc: half kernel sizes must be non-zero.");
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  int32_t* padded = nullptr;
  if (cleanup) {
    int rc = vwgpu_arena_reserve(ctx, &ctx->pyr, (size_t)(w + 2) * (h + 2) * 12);
    if (rc) return rc;
    padded = static_cast<int32_t*>(ctx->pyr.base);
  }
  return vwgpu_launch_disparity_filter(ctx, d_src, w, h, hh, hv, pthr, rthr, cleanup != 0, padded, d_dst);
}

int vwgpu_disparity_filter(vwgpu_ctx* ctx, const int32_t* src, int w, int h, int hh, int hv,
                           double pthr, double rthr, int cleanup, int32_t* dst) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  if (!src || !dst || w <= 0 || h <= 0) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "rm_outliers_using_thresh: bad image arguments");
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  const size_t nb = vwgpu_align_up((size_t)w * h * 12, 256);
  int rc = vwgpu_arena_reserve(ctx, &ctx->staging, 2 * nb);
  if (rc) return rc;
  int32_t* a = static_cast<int32_t*>(ctx->staging.base);
  int32_t* b = reinterpret_cast<int32_t*>(static_cast<char*>(ctx->staging.base) + nb);
  VWGPU_HIP(ctx, hipMemcpyAsync(a, src, (size_t)w * h * 12, hipMemcpyHostToDevice, ctx->stream));
  rc = vwgpu_disparity_filter_dev(ctx, a, w, h, hh, hv, pthr, rthr, cleanup, b);
  if (rc) return rc;
  VWGPU_HIP(ctx, hipMemcpyAsync(dst, b, (size_t)w * h * 12, hipMemcpyDeviceToHost, ctx->stream));
  VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VWGPU_OK;
}

int vwgpu_disparity_mask_dev(vwgpu_ctx* ctx, int32_t* d_disp, int w, int h, const uint8_t* m1, const uint8_t* m2, int rmw, int rmh) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  ctx->err.clear();
  if (!d_disp || !m1 || !m2 || w <= 0 || h <= 0 || rmw <= 0 || rmh <= 0) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "disparity_mask: bad image arguments");
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  return vwgpu_launch_disparity_mask(ctx, d_disp, w, h, m1, m2, rmw, rmh);
}

int vwgpu_disparity_mask(vwgpu_ctx* ctx, int32_t* disp, int w, int h, const uint8_t* m1, const uint8_t* m2, int rmw, int rmh) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  if (!disp || !m1 || !m2 || w <= 0 || h <= 0 || rmw <= 0 || rmh <= 0) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "disparity_mask: bad image arguments");
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  const size_t db = vwgpu_align_up((size_t)w * h * 12, 256), ab = vwgpu_align_up((size_t)w * h, 256), bb = vwgpu_align_up((size_t)rmw * rmh, 256);
  int rc = vwgpu_arena_reserve(ctx, &ctx->staging, db + ab + bb);
  if (rc) return rc;
  char* base = static_cast<char*>(ctx->staging.base);
  int32_t* d = reinterpret_cast<int32_t*>(base);
  uint8_t* a = reinterpret_cast<uint8_t*>(base + db);
  uint8_t* b = reinterpret_cast<uint8_t*>(base + db + ab);
  VWGPU_HIP(ctx, hipMemcpyAsync(d, disp, (size_t)w * h * 12, hipMemcpyHostToDevice, ctx->stream));
  VWGPU_HIP(ctx, hipMemcpyAsync(a, m1, (size_t)w * h, hipMemcpyHostToDevice, ctx->stream));
  VWGPU_HIP(ctx, hipMemcpyAsync(b, m2, (size_t)rmw * rmh, hipMemcpyHostToDevice, ctx->stream));
  rc = vwgpu_launch_disparity_mask(ctx, d, w, h, a, b, rmw, rmh);
  if (rc) return rc;
  VWGPU_HIP(ctx, hipMemcpyAsync(disp, d, (size_t)w * h * 12, hipMemcpyDeviceToHost, ctx->stream));
  VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VWGPU_OK;
}

int vwgpu_disparity_blob_filter_dev(vwgpu_ctx* ctx, int32_t* d_disp, int w, int h, int max_blob_area) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  ctx->err.clear();
  if (!d_disp || w <= 0 || h <= 0 || (long long)w * h > 0x7fffffffLL) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "disparity_blob_filter: bad image arguments");
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  int rc = vwgpu_arena_reserve(ctx, &ctx->pyr, (size_t)w * h * 8);
  if (rc) return rc;
  return vwgpu_launch_blob_filter(ctx, d_disp, w, h, max_blob_area, static_cast<int*>(ctx->pyr.base));
}

int vwgpu_disparity_blob_filter(vwgpu_ctx* ctx, int32_t* disp, int w, int h, int max_blob_area) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  if (!disp || w <= 0 || h <= 0) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "disparity_blob_filter: bad image arguments");
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  int rc = vwgpu_arena_reserve(ctx, &ctx->staging, (size_t)w * h * 12);
  if (rc) return rc;
  int32_t* d = static_cast<int32_t*>(ctx->staging.base);
  VWGPU_HIP(ctx, hipMemcpyAsync(d, disp, (size_t)w * h * 12, hipMemcpyHostToDevice, ctx->stream));
  rc = vwgpu_disparity_blob_filter_dev(ctx, d, w, h, max_blob_area);
  if (rc) return rc;
  VWGPU_HIP(ctx, hipMemcpyAsync(disp, d, (size_t)w * h * 12, hipMemcpyDeviceToHost, ctx->stream));
  VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VWGPU_OK;
}

static int check_pyramid_args(vwgpu_ctx* ctx, const void* l, int lw, int lh, const void* r, int rw, int rh,
                              const vwgpu_pyramid_params* P, int bw, int bh, const void* out) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  ctx->err.clear();
  if (!l || !r || !out || !P || lw <= 0 || lh <= 0 || rw <= 0 || rh <= 0 || bw <= 0 || bh <= 0)
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "pyramid_correlate: empty image / tile or null pointer");
  if (P->kernel_x < 1 || P->kernel_y < 1 || P->kernel_x % 2 != 1 || P->kernel_y % 2 != 1)
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "pyramid_correlate: Kernel input not sized with odd values.");
  if (P->search_max_x <= P->search_min_x || P->search_max_y <= P->search_min_y)
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "PyramidCorrelationView: Invalid search region: (%d,%d)-(%d,%d); a box of zero height is empty (use height >= 1)",
                      P->search_min_x, P->search_min_y, P->search_max_x, P->search_max_y);
  if (P->algorithm == 0 && (P->cost_type < VWGPU_ABSOLUTE_DIFFERENCE || P->cost_type > VWGPU_CROSS_CORRELATION))
    return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "pyramid_correlate: cost type %d is not a block-matching cost", P->cost_type);
  if (P->algorithm < 0 || P->algorithm > 3)
    return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "pyramid_correlate: algorithm %d is none of VW_CORRELATION_BM / _SGM / _MGM / _FINAL_MGM", P->algorithm);
  if (P->algorithm != 0) {
    if (P->cost_type != VWGPU_CENSUS_TRANSFORM && P->cost_type != VWGPU_TERNARY_CENSUS_TRANSFORM)
      return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "With SGM/MGM, only the census transform cost mode gives good results.");
    if (P->kernel_x != P->kernel_y || (P->kernel_x != 3 && P->kernel_x != 5 && P->kernel_x != 7 && P->kernel_x != 9))
      return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "Census transforms are only available in size 3, 5, 7, and 9.");
  }
  if (P->blob_filter_area < 0) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "pyramid_correlate: negative blob filter area");
  if (P->lr_disp_diff) {          // CorrelationView.cc:277-283
    const ptrdiff_t st = P->lr_disp_diff_stride ? P->lr_disp_diff_stride : P->lr_disp_diff_cols;
    if (P->lr_disp_diff_cols <= 0 || P->lr_disp_diff_rows <= 0 || st < P->lr_disp_diff_cols)
      return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "pyramid_correlate: bad lr_disp_diff geometry");
  }
  if (P->max_pyramid_levels < 0 || P->filter_half_kernel < 0) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "pyramid_correlate: negative level / filter size");
  return VWGPU_OK;
}

int vwgpu_pyramid_correlate_dev(vwgpu_ctx* ctx, const float* d_left, int lw, int lh, ptrdiff_t ls,
                                const float* d_right, int rw, int rh, ptrdiff_t rs,
                                const uint8_t* d_lmask, ptrdiff_t lms, const uint8_t* d_rmask, ptrdiff_t rms,
                                const vwgpu_pyramid_params* P, int bx, int by, int bw, int bh, float* d_out, ptrdiff_t os) {
  int rc = check_pyramid_args(ctx, d_left, lw, lh, d_right, rw, rh, P, bw, bh, d_out);
  if (rc) return rc;
  if (ls == 0) ls = lw;
  if (rs == 0) rs = rw;
  if (lms == 0) lms = lw;
  if (rms == 0) rms = rw;
  if (os == 0) os = bw;
  if (ls < lw || rs < rw || lms < lw || rms < rw || os < bw) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "pyramid_correlate: row stride smaller than row width");
  if (P->lr_disp_diff && (bx < P->region_ul_x || by < P->region_ul_y || bx + bw > P->region_ul_x + P->lr_disp_diff_cols ||
                          by + bh > P->region_ul_y + P->lr_disp_diff_rows))
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "The L-R to R-L difference image domain does not contain the current tile.");
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  return vwgpu_pyramid_correlate_impl(ctx, d_left, lw, lh, ls, d_right, rw, rh, rs, d_lmask, lms, d_rmask, rms, P, bx, by, bw, bh, d_out, os,
                                      P->lr_disp_diff);
}

int vwgpu_pyramid_correlate_batch_dev(vwgpu_ctx* ctx, const float* d_left, int lw, int lh, ptrdiff_t ls,
                                      const float* d_right, int rw, int rh, ptrdiff_t rs,
                                      const uint8_t* d_lmask, ptrdiff_t lms, const uint8_t* d_rmask, ptrdiff_t rms,
                                      const vwgpu_pyramid_params* P, int n_tiles, const int* bx, const int* by, const int* bw, const int* bh,
                                      float* const* d_outs, const ptrdiff_t* os) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  ctx->err.clear();
  if (n_tiles < 0 || (n_tiles > 0 && (!bx || !by || !bw || !bh || !d_outs))) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "pyramid_correlate_batch: null tile table");
  if (ls == 0) ls = lw;
  if (rs == 0) rs = rw;
  if (lms == 0) lms = lw;
  if (rms == 0) rms = rw;
  for (int t = 0; t < n_tiles; ++t) {
    int rc = check_pyramid_args(ctx, d_left, lw, lh, d_right, rw, rh, P, bw[t], bh[t], d_outs[t]);
    if (rc) return rc;
    if (os && os[t] != 0 && os[t] < bw[t]) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "pyramid_correlate: row stride smaller than row width");
  }
  if (n_tiles == 0) return VWGPU_OK;
  if (ls < lw || rs < rw || lms < lw || rms < rw) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "pyramid_correlate: row stride smaller than row width");
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  // groups: runs of consecutive tiles of equal size, at most VWGPU_MAX_GROUP each; what a group cannot take (SGM / MGM, lr_disp_diff, blob
  // filter, time budget, a forced kernel family) and lone tiles go through the single-tile entry — same results either way
  int t0 = 0, cap = VWGPU_MAX_GROUP;                   // cap: halved when a group's arena does not fit the device
  while (t0 < n_tiles) {
    int t1 = t0 + 1;
    while (t1 < n_tiles && t1 - t0 < cap && bw[t1] == bw[t0] && bh[t1] == bh[t0]) ++t1;
    const int m = t1 - t0;
    if (m > 1 && vwgpu_pyramid_group_eligible(ctx, P, m, bw + t0, bh + t0)) {
      ptrdiff_t oss[VWGPU_MAX_GROUP];
      for (int t = 0; t < m; ++t) oss[t] = (os && os[t0 + t]) ? os[t0 + t] : bw[t0 + t];
      int rc = vwgpu_pyramid_group_impl(ctx, d_left, lw, lh, ls, d_right, rw, rh, rs, d_lmask, lms, d_rmask, rms, P, m, bx + t0, by + t0, bw[t0], bh[t0], d_outs + t0, oss);
      if (rc == VWGPU_ERR_NOMEM) {
        // n slices + the group's tables did not fit where the single-tile entry's arena would: the batch entry must never be less robust
        // than a loop over the single-tile entry.  Nothing of the group has been written (the reservation comes first): smaller groups.
        ctx->err.clear();
        cap = m / 2;
        continue;
      }
      if (rc) return rc;
    } else {
      for (int t = t0; t < t1; ++t) {
        if (P->lr_disp_diff && (bx[t] < P->region_ul_x || by[t] < P->region_ul_y || bx[t] + bw[t] > P->region_ul_x + P->lr_disp_diff_cols ||
                                by[t] + bh[t] > P->region_ul_y + P->lr_disp_diff_rows))
          return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "The L-R to R-L difference image domain does not contain the current tile.");
        int rc = vwgpu_pyramid_correlate_impl(ctx, d_left, lw, lh, ls, d_right, rw, rh, rs, d_lmask, lms, d_rmask, rms, P, bx[t], by[t], bw[t], bh[t], d_outs[t],
                                              (os && os[t]) ? os[t] : bw[t], P->lr_disp_diff);
        if (rc) return rc;
      }
    }
    t0 = t1;
  }
  return VWGPU_OK;
}

// One run of tiles of the host batch entry: ONE staged window (the union of what the run's tiles can touch), one device batch call.
static int pyramid_batch_host_run(vwgpu_ctx* ctx, const float* left, int lw, int lh, ptrdiff_t ls, const float* right, int rw, int rh, ptrdiff_t rs,
                                  const uint8_t* lmask, ptrdiff_t lms, const uint8_t* rmask, ptrdiff_t rms, const vwgpu_pyramid_params* P, int n_tiles,
                                  const int* bx, const int* by, const int* bw, const int* bh, float* const* outs, const ptrdiff_t* os) {
  // the window of the sources the tiles can touch (see vwgpu_pyramid_correlate): the union over the run's tiles, one origin for both images and masks
  const int upb = 1 << std::max(0, std::min(P->max_pyramid_levels, 12));
  const int sdx = P->search_max_x - P->search_min_x, sdy = P->search_max_y - P->search_min_y;
  const long long padx = (long long)(P->kernel_x / 2) * upb + 2LL * std::max(sdx, 0) + 8;
  const long long pady = (long long)(P->kernel_y / 2) * upb + 2LL * std::max(sdy, 0) + 8;
  long long ux0 = LLONG_MAX, uy0 = LLONG_MAX, ux1 = LLONG_MIN, uy1 = LLONG_MIN;
  size_t out_bytes = 0;
  for (int t = 0; t < n_tiles; ++t) {
    ux0 = std::min<long long>(ux0, bx[t]); uy0 = std::min<long long>(uy0, by[t]);
    ux1 = std::max<long long>(ux1, (long long)bx[t] + bw[t]); uy1 = std::max<long long>(uy1, (long long)by[t] + bh[t]);
    out_bytes += vwgpu_align_up((size_t)bw[t] * bh[t] * 12, 256);
  }
  const long long wx0 = std::max<long long>(0, ux0 - padx + std::min(P->search_min_x, 0)), wy0 = std::max<long long>(0, uy0 - pady + std::min(P->search_min_y, 0));
  const long long wx1 = ux1 + padx + std::max(P->search_max_x, 0), wy1 = uy1 + pady + std::max(P->search_max_y, 0);
  const int ox = (int)std::min<long long>(wx0, std::min(lw, rw)), oy = (int)std::min<long long>(wy0, std::min(lh, rh));
  const int lww = (int)(std::min<long long>(wx1, lw) - ox), lwh = (int)(std::min<long long>(wy1, lh) - oy);
  const int rww = (int)(std::min<long long>(wx1, rw) - ox), rwh = (int)(std::min<long long>(wy1, rh) - oy);
  if (lww <= 0 || lwh <= 0 || rww <= 0 || rwh <= 0) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "pyramid_correlate: the tiles lie outside the images");
  const size_t lb = vwgpu_align_up((size_t)lww * lwh * 4, 256), rb = vwgpu_align_up((size_t)rww * rwh * 4, 256);
  const size_t lmb = vwgpu_align_up((size_t)lww * lwh, 256), rmb = vwgpu_align_up((size_t)rww * rwh, 256);
  int rc = vwgpu_arena_reserve(ctx, &ctx->staging, lb + rb + lmb + rmb + out_bytes);
  if (rc) return rc;
  char* base = static_cast<char*>(ctx->staging.base);
  float* d_l = reinterpret_cast<float*>(base);
  float* d_r = reinterpret_cast<float*>(base + lb);
  uint8_t* d_lm = reinterpret_cast<uint8_t*>(base + lb + rb);
  uint8_t* d_rm = reinterpret_cast<uint8_t*>(base + lb + rb + lmb);
  VWGPU_HIP(ctx, hipMemcpy2DAsync(d_l, (size_t)lww * 4, left + (ptrdiff_t)oy * ls + ox, (size_t)ls * 4, (size_t)lww * 4, lwh, hipMemcpyHostToDevice, ctx->stream));
  VWGPU_HIP(ctx, hipMemcpy2DAsync(d_r, (size_t)rww * 4, right + (ptrdiff_t)oy * rs + ox, (size_t)rs * 4, (size_t)rww * 4, rwh, hipMemcpyHostToDevice, ctx->stream));
  if (lmask) VWGPU_HIP(ctx, hipMemcpy2DAsync(d_lm, (size_t)lww, lmask + (ptrdiff_t)oy * lms + ox, (size_t)lms, (size_t)lww, lwh, hipMemcpyHostToDevice, ctx->stream));
  if (rmask) VWGPU_HIP(ctx, hipMemcpy2DAsync(d_rm, (size_t)rww, rmask + (ptrdiff_t)oy * rms + ox, (size_t)rms, (size_t)rww, rwh, hipMemcpyHostToDevice, ctx->stream));
  std::vector<int> wbx((size_t)n_tiles), wby((size_t)n_tiles);
  std::vector<float*> d_outs((size_t)n_tiles);
  char* q = base + lb + rb + lmb + rmb;
  for (int t = 0; t < n_tiles; ++t) {
    wbx[t] = bx[t] - ox; wby[t] = by[t] - oy;
    d_outs[t] = reinterpret_cast<float*>(q);
    q += vwgpu_align_up((size_t)bw[t] * bh[t] * 12, 256);
  }
  rc = vwgpu_pyramid_correlate_batch_dev(ctx, d_l, lww, lwh, lww, d_r, rww, rwh, rww, lmask ? d_lm : nullptr, lww, rmask ? d_rm : nullptr, rww, P, n_tiles,
                                         wbx.data(), wby.data(), bw, bh, d_outs.data(), nullptr);
  if (rc) return rc;
  for (int t = 0; t < n_tiles; ++t) {
    const ptrdiff_t o = (os && os[t]) ? os[t] : bw[t];
    VWGPU_HIP(ctx, hipMemcpy2DAsync(outs[t], (size_t)o * 12, d_outs[t], (size_t)bw[t] * 12, (size_t)bw[t] * 12, bh[t], hipMemcpyDeviceToHost, ctx->stream));
  }
  VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VWGPU_OK;
}

int vwgpu_pyramid_correlate_batch(vwgpu_ctx* ctx, const float* left, int lw, int lh, ptrdiff_t ls,
                                  const float* right, int rw, int rh, ptrdiff_t rs,
                                  const uint8_t* lmask, ptrdiff_t lms, const uint8_t* rmask, ptrdiff_t rms,
                                  const vwgpu_pyramid_params* P, int n_tiles, const int* bx, const int* by, const int* bw, const int* bh,
                                  float* const* outs, const ptrdiff_t* os) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  ctx->err.clear();
  if (n_tiles < 0 || (n_tiles > 0 && (!bx || !by || !bw || !bh || !outs))) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "pyramid_correlate_batch: null tile table");
  if (n_tiles == 0) return VWGPU_OK;
  // an lr_disp_diff image is staged per tile by the single-tile entry (concurrent tiles must not write back each other's pixels)
  if (P && P->lr_disp_diff) {
    for (int t = 0; t < n_tiles; ++t) {
      int rc = vwgpu_pyramid_correlate(ctx, left, lw, lh, ls, right, rw, rh, rs, lmask, lms, rmask, rms, P, bx[t], by[t], bw[t], bh[t], outs[t], os ? os[t] : 0);
      if (rc) return rc;
    }
    return VWGPU_OK;
  }
  for (int t = 0; t < n_tiles; ++t) {
    int rc = check_pyramid_args(ctx, left, lw, lh, right, rw, rh, P, bw[t], bh[t], outs[t]);
    if (rc) return rc;
  }
  if (ls == 0) ls = lw;
  if (rs == 0) rs = rw;
  if (lms == 0) lms = lw;
  if (rms == 0) rms = rw;
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  // Staged per RUN of tiles (consecutive tiles of equal size, at most VWGPU_MAX_GROUP: what one device group takes), not per call: tiles
  // scattered over a large pair would otherwise upload nearly the whole pair.  A run whose union window is more than three
  // times the windows of its tiles taken one by one (tiles far apart inside the run) goes through the single-tile entry.
  const int upb_ = 1 << std::max(0, std::min(P->max_pyramid_levels, 12));
  const long long padx_ = (long long)(P->kernel_x / 2) * upb_ + 2LL * std::max(P->search_max_x - P->search_min_x, 0) + 8 + std::max(P->search_max_x, 0) - std::min(P->search_min_x, 0);
  const long long pady_ = (long long)(P->kernel_y / 2) * upb_ + 2LL * std::max(P->search_max_y - P->search_min_y, 0) + 8 + std::max(P->search_max_y, 0) - std::min(P->search_min_y, 0);
  int t0 = 0;
  while (t0 < n_tiles) {
    int t1 = t0 + 1;
    while (t1 < n_tiles && t1 - t0 < VWGPU_MAX_GROUP && bw[t1] == bw[t0] && bh[t1] == bh[t0]) ++t1;
    long long ux0 = LLONG_MAX, uy0 = LLONG_MAX, ux1 = LLONG_MIN, uy1 = LLONG_MIN;
    double each = 0.0;
    for (int t = t0; t < t1; ++t) {
      ux0 = std::min<long long>(ux0, bx[t]); uy0 = std::min<long long>(uy0, by[t]);
      ux1 = std::max<long long>(ux1, (long long)bx[t] + bw[t]); uy1 = std::max<long long>(uy1, (long long)by[t] + bh[t]);
      each += (double)(bw[t] + 2 * padx_) * (double)(bh[t] + 2 * pady_);
    }
    const double uni = (double)(ux1 - ux0 + 2 * padx_) * (double)(uy1 - uy0 + 2 * pady_);
    if (t1 - t0 > 1 && uni <= 3.0 * each) {
      int rc = pyramid_batch_host_run(ctx, left, lw, lh, ls, right, rw, rh, rs, lmask, lms, rmask, rms, P, t1 - t0, bx + t0, by + t0, bw + t0, bh + t0, outs + t0, os ? os + t0 : nullptr);
      if (rc) return rc;
    } else {
      for (int t = t0; t < t1; ++t) {
        int rc = vwgpu_pyramid_correlate(ctx, left, lw, lh, ls, right, rw, rh, rs, lmask, lms, rmask, rms, P, bx[t], by[t], bw[t], bh[t], outs[t], os ? os[t] : 0);
        if (rc) return rc;
      }
    }
    t0 = t1;
  }
  return VWGPU_OK;
}

int vwgpu_pyramid_correlate(vwgpu_ctx* ctx, const float* left, int lw, int lh, ptrdiff_t ls,
                            const float* right, int rw, int rh, ptrdiff_t rs,
                            const uint8_t* lmask, ptrdiff_t lms, const uint8_t* rmask, ptrdiff_t rms,
                            const vwgpu_pyramid_params* P, int bx, int by, int bw, int bh, float* out, ptrdiff_t os) {
  int rc = check_pyramid_args(ctx, left, lw, lh, right, rw, rh, P, bw, bh, out);
  if (rc) return rc;
  if (ls == 0) ls = lw;
  if (rs == 0) rs = rw;
  if (lms == 0) lms = lw;
  if (rms == 0) rms = rw;
  if (os == 0) os = bw;
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  // Only the window of the sources this tile can touch goes to the device — the reference's prerasterize also pulls just
  // its padded ROIs through the views (CorrelationView.cc:89-97) — not the whole images: a 1024^2 tile of a 32768^2 pair
  // stages ~2000^2 pixels per image.  Window = tile grown by half_kernel * 2^max_levels (pyramid padding), twice the
  // search extent (R->L runs of the SGM branch) and the search range itself, cut at the image borders; both images and
  // masks share the window origin, so disparities and the edge extension at true image borders are unchanged.
  const int upb = 1 << std::max(0, std::min(P->max_pyramid_levels, 12));
  const int sdx = P->search_max_x - P->search_min_x, sdy = P->search_max_y - P->search_min_y;
  const long long padx = (long long)(P->kernel_x / 2) * upb + 2LL * std::max(sdx, 0) + 8;
  const long long pady = (long long)(P->kernel_y / 2) * upb + 2LL * std::max(sdy, 0) + 8;
  const long long wx0 = std::max<long long>(0, (long long)bx - padx + std::min(P->search_min_x, 0));
  const long long wy0 = std::max<long long>(0, (long long)by - pady + std::min(P->search_min_y, 0));
  const long long wx1 = (long long)bx + bw + padx + std::max(P->search_max_x, 0);
  const long long wy1 = (long long)by + bh + pady + std::max(P->search_max_y, 0);
  const int ox = (int)std::min<long long>(wx0, std::min(lw, rw)), oy = (int)std::min<long long>(wy0, std::min(lh, rh));
  const int lww = (int)(std::min<long long>(wx1, lw) - ox), lwh = (int)(std::min<long long>(wy1, lh) - oy);
  const int rww = (int)(std::min<long long>(wx1, rw) - ox), rwh = (int)(std::min<long long>(wy1, rh) - oy);
  if (lww <= 0 || lwh <= 0 || rww <= 0 || rwh <= 0)
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "pyramid_correlate: the tile lies outside the images");
  const size_t lb = vwgpu_align_up((size_t)lww * lwh * 4, 256), rb = vwgpu_align_up((size_t)rww * rwh * 4, 256);
  const size_t lmb = vwgpu_align_up((size_t)lww * lwh, 256), rmb = vwgpu_align_up((size_t)rww * rwh, 256), ob = vwgpu_align_up((size_t)bw * bh * 12, 256);
  if (P->lr_disp_diff && (bx < P->region_ul_x || by < P->region_ul_y || bx + bw > P->region_ul_x + P->lr_disp_diff_cols ||
                          by + bh > P->region_ul_y + P->lr_disp_diff_rows))
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "The L-R to R-L difference image domain does not contain the current tile.");
  // The discrepancy image is shared by the tile threads (each writes its own pixels, CorrelationView.cc:683-694): only the
  // tile's rectangle of it is staged, so that concurrent tiles never write back each other's stale pixels.
  const ptrdiff_t hdst = P->lr_disp_diff_stride ? P->lr_disp_diff_stride : P->lr_disp_diff_cols;
  const size_t db = P->lr_disp_diff ? vwgpu_align_up((size_t)bw * bh * 8, 256) : 0;
  float* h_diff = P->lr_disp_diff ? P->lr_disp_diff + ((ptrdiff_t)(by - P->region_ul_y) * hdst + (bx - P->region_ul_x)) * 2 : nullptr;
  rc = vwgpu_arena_reserve(ctx, &ctx->staging, lb + rb + lmb + rmb + ob + db);
  if (rc) return rc;
  char* base = static_cast<char*>(ctx->staging.base);
  float* d_l = reinterpret_cast<float*>(base);
  float* d_r = reinterpret_cast<float*>(base + lb);
  uint8_t* d_lm = reinterpret_cast<uint8_t*>(base + lb + rb);
  uint8_t* d_rm = reinterpret_cast<uint8_t*>(base + lb + rb + lmb);
  float* d_o = reinterpret_cast<float*>(base + lb + rb + lmb + rmb);
  const float* lsrc = left + (ptrdiff_t)oy * ls + ox;
  const float* rsrc = right + (ptrdiff_t)oy * rs + ox;
  VWGPU_HIP(ctx, hipMemcpy2DAsync(d_l, (size_t)lww * 4, lsrc, (size_t)ls * 4, (size_t)lww * 4, lwh, hipMemcpyHostToDevice, ctx->stream));
  VWGPU_HIP(ctx, hipMemcpy2DAsync(d_r, (size_t)rww * 4, rsrc, (size_t)rs * 4, (size_t)rww * 4, rwh, hipMemcpyHostToDevice, ctx->stream));
  if (lmask) VWGPU_HIP(ctx, hipMemcpy2DAsync(d_lm, (size_t)lww, lmask + (ptrdiff_t)oy * lms + ox, (size_t)lms, (size_t)lww, lwh, hipMemcpyHostToDevice, ctx->stream));
  if (rmask) VWGPU_HIP(ctx, hipMemcpy2DAsync(d_rm, (size_t)rww, rmask + (ptrdiff_t)oy * rms + ox, (size_t)rms, (size_t)rww, rwh, hipMemcpyHostToDevice, ctx->stream));
  float* d_d = reinterpret_cast<float*>(base + lb + rb + lmb + rmb + ob);
  vwgpu_pyramid_params Pd = *P;                     // the device-side discrepancy image is the tile's rectangle, dense
  Pd.lr_disp_diff_cols = bw; Pd.lr_disp_diff_rows = bh; Pd.lr_disp_diff_stride = bw;
  Pd.region_ul_x = bx - ox;                         // everything below runs in window coordinates
  Pd.region_ul_y = by - oy;
  if (P->lr_disp_diff)
    VWGPU_HIP(ctx, hipMemcpy2DAsync(d_d, (size_t)bw * 8, h_diff, (size_t)hdst * 8, (size_t)bw * 8, bh, hipMemcpyHostToDevice, ctx->stream));
  rc = vwgpu_pyramid_correlate_impl(ctx, d_l, lww, lwh, lww, d_r, rww, rwh, rww, lmask ? d_lm : nullptr, lww, rmask ? d_rm : nullptr, rww,
                                    &Pd, bx - ox, by - oy, bw, bh, d_o, bw, P->lr_disp_diff ? d_d : nullptr);
  if (rc) return rc;
  if (P->lr_disp_diff)
    VWGPU_HIP(ctx, hipMemcpy2DAsync(h_diff, (size_t)hdst * 8, d_d, (size_t)bw * 8, (size_t)bw * 8, bh, hipMemcpyDeviceToHost, ctx->stream));
  VWGPU_HIP(ctx, hipMemcpy2DAsync(out, (size_t)os * 12, d_o, (size_t)bw * 12, (size_t)bw * 12, bh, hipMemcpyDeviceToHost, ctx->stream));
  VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VWGPU_OK;
}

}  // extern "C"
