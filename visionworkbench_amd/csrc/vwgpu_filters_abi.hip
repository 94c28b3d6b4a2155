// vwgpu_filters_abi.hip — extern "C" entry points of the pyramid / prefilter family (include/vwgpu.h).
#include <algorithm>
#include <cmath>
#include <vector>

#include <climits>

#include "vwgpu_internal.h"

namespace {

int check_image(vwgpu_ctx* ctx, const char* what, const void* src, int w, int h, ptrdiff_t& stride, const void* dst) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  ctx->err.clear();
  if (!src || !dst || w <= 0 || h <= 0) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "%s: empty image or null pointer", what);
  if (stride == 0) stride = w;
  if (stride < w) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "%s: row stride smaller than row width", what);
  return VWGPU_OK;
}

// Host-pointer variants: stage src, run `body(d_src, d_dst)`, copy back.
template <class T, class Body>
int staged(vwgpu_ctx* ctx, const T* src, int w, int h, ptrdiff_t stride, T* dst, int ow, int oh, ptrdiff_t dstride, Body body) {
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  const size_t sb = vwgpu_align_up((size_t)w * h * sizeof(T), 256), db = vwgpu_align_up((size_t)ow * oh * sizeof(T), 256);
  int rc = vwgpu_arena_reserve(ctx, &ctx->staging, sb + db);
  if (rc) return rc;
  T* d_s = reinterpret_cast<T*>(ctx->staging.base);
  T* d_d = reinterpret_cast<T*>(static_cast<char*>(ctx->staging.base) + sb);
  VWGPU_HIP(ctx, hipMemcpy2DAsync(d_s, (size_t)w * sizeof(T), src, (size_t)stride * sizeof(T), (size_t)w * sizeof(T), h,
                                  hipMemcpyHostToDevice, ctx->stream));
  rc = body(d_s, d_d);
  if (rc) return rc;
  VWGPU_HIP(ctx, hipMemcpy2DAsync(dst, (size_t)dstride * sizeof(T), d_d, (size_t)ow * sizeof(T), (size_t)ow * sizeof(T), oh,
                                  hipMemcpyDeviceToHost, ctx->stream));
  VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VWGPU_OK;
}

}  // namespace

extern "C" {

int vwgpu_generate_gaussian_kernel(double sigma, int size, float* taps, int cap) {
  // vw::generate_gaussian_kernel<float>, src/vw/Image/Filter.tcc:37-78 (vw::erf is ::erf on Linux,
  // src/vw/Math/Functions.h:197).
  if (sigma == 0) return 0;
  if (size == 0) {                                 // vw::compute_kernel_size, src/vw/Image/Filter.cc:32-37
    size = (int)(7 * sigma);
    if (size < 3) size = 3;
    else if (size % 2 == 0) size -= 1;
  }
  if (size < 0 || size > cap || !taps) return VWGPU_ERR_ARGUMENT;
  const int center = size / 2;
  double sum = 0.0, tap;
  const double z = 1 / (std::sqrt(2.0) * sigma);
  if (size % 2 == 0) {
    for (int i = 0; i < center; ++i) {
      tap = std::erf((i + 1.0) * z) - std::erf(i * z);
      sum += tap;
      taps[center + i] = taps[center - i - 1] = (float)tap;
    }
    sum *= 2.0;
  } else {
    for (int i = 1; i <= center; ++i) {
      tap = std::erf((i + 0.5) * z) - std::erf((i - 0.5) * z);
      sum += tap;
      taps[center + i] = taps[center - i] = (float)tap;
    }
    sum *= 2.0;
    tap = std::erf(0.5 * z) - std::erf(-0.5 * z);
    sum += tap;
    taps[center] = (float)tap;
  }
  const double norm = 1.0 / sum;
  for (int i = 0; i < size; ++i) taps[i] *= norm;  // float *= double, as the reference does
  return size;
}

int vwgpu_separable_convolution_dev(vwgpu_ctx* ctx, const float* d_src, int w, int h, ptrdiff_t stride,
                                    const float* xk, int nx, int cx, const float* yk, int ny, int cy,
                                    int edge, int subsample, float* d_dst, ptrdiff_t dstride) {
  int rc = check_image(ctx, "separable_convolution_filter", d_src, w, h, stride, d_dst);
  if (rc) return rc;
  if (nx < 0 || ny < 0 || (nx && !xk) || (ny && !yk) || subsample < 1 || (edge != 0 && edge != 1) ||
      (nx && (cx < 0 || cx >= nx)) || (ny && (cy < 0 || cy >= ny)))
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "separable_convolution_filter: bad kernel / origin / edge / subsample argument");
  const int ow = 1 + (w - 1) / subsample;
  if (dstride == 0) dstride = ow;
  if (dstride < ow) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "separable_convolution_filter: destination stride too small");
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  return vwgpu_launch_sepconv(ctx, d_src, w, h, stride, xk, nx, cx, yk, ny, cy, edge, subsample, d_dst, dstride);
}

int vwgpu_separable_convolution(vwgpu_ctx* ctx, const float* src, int w, int h, ptrdiff_t stride,
                                const float* xk, int nx, int cx, const float* yk, int ny, int cy,
                                int edge, int subsample, float* dst, ptrdiff_t dstride) {
  int rc = check_image(ctx, "separable_convolution_filter", src, w, h, stride, dst);
  if (rc) return rc;
  if (subsample < 1) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "separable_convolution_filter: subsample < 1");
  const int ow = 1 + (w - 1) / subsample, oh = 1 + (h - 1) / subsample;
  if (dstride == 0) dstride = ow;
  return staged<float>(ctx, src, w, h, stride, dst, ow, oh, dstride, [&](float* ds, float* dd) {
    return vwgpu_separable_convolution_dev(ctx, ds, w, h, w, xk, nx, cx, yk, ny, cy, edge, subsample, dd, ow);
  });
}

int vwgpu_convolution_2d_dev(vwgpu_ctx* ctx, const float* d_src, int w, int h, ptrdiff_t stride,
                             const float* kernel, int kw, int kh, int ci, int cj, int edge,
                             float* d_dst, ptrdiff_t dstride) {
  int rc = check_image(ctx, "convolution_filter", d_src, w, h, stride, d_dst);
  if (rc) return rc;
  if (!kernel || kw <= 0 || kh <= 0 || ci < 0 || ci >= kw || cj < 0 || cj >= kh || (edge != 0 && edge != 1))
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "convolution_filter: bad kernel / origin / edge argument");
  if (dstride == 0) dstride = w;
  if (dstride < w) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "convolution_filter: destination stride too small");
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  return vwgpu_launch_conv2d(ctx, d_src, w, h, stride, kernel, kw, kh, ci, cj, edge, d_dst, dstride);
}

int vwgpu_convolution_2d(vwgpu_ctx* ctx, const float* src, int w, int h, ptrdiff_t stride,
                         const float* kernel, int kw, int kh, int ci, int cj, int edge, float* dst, ptrdiff_t dstride) {
  int rc = check_image(ctx, "convolution_filter", src, w, h, stride, dst);
  if (rc) return rc;
  if (dstride == 0) dstride = w;
  return staged<float>(ctx, src, w, h, stride, dst, w, h, dstride, [&](float* ds, float* dd) {
    return vwgpu_convolution_2d_dev(ctx, ds, w, h, w, kernel, kw, kh, ci, cj, edge, dd, w);
  });
}

int vwgpu_subsample_mask_by_two_dev(vwgpu_ctx* ctx, const uint8_t* d_src, int w, int h, ptrdiff_t stride,
                                    uint8_t* d_dst, ptrdiff_t dstride) {
  int rc = check_image(ctx, "subsample_mask_by_two", d_src, w, h, stride, d_dst);
  if (rc) return rc;
  const int ow = 1 + (w - 1) / 2;
  if (dstride == 0) dstride = ow;
  if (dstride < ow) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "subsample_mask_by_two: destination stride too small");
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  return vwgpu_launch_mask_by_two(ctx, d_src, w, h, stride, d_dst, dstride);
}

int vwgpu_subsample_mask_by_two(vwgpu_ctx* ctx, const uint8_t* src, int w, int h, ptrdiff_t stride,
                                uint8_t* dst, ptrdiff_t dstride) {
  int rc = check_image(ctx, "subsample_mask_by_two", src, w, h, stride, dst);
  if (rc) return rc;
  const int ow = 1 + (w - 1) / 2, oh = 1 + (h - 1) / 2;
  if (dstride == 0) dstride = ow;
  return staged<uint8_t>(ctx, src, w, h, stride, dst, ow, oh, dstride, [&](uint8_t* ds, uint8_t* dd) {
    return vwgpu_subsample_mask_by_two_dev(ctx, ds, w, h, w, dd, ow);
  });
}

int vwgpu_prefilter_image_dev(vwgpu_ctx* ctx, const float* d_src, int w, int h, ptrdiff_t stride,
                              int mode, float width, float* d_dst, ptrdiff_t dstride) {
  int rc = check_image(ctx, "prefilter_image", d_src, w, h, stride, d_dst);
  if (rc) return rc;
  if (dstride == 0) dstride = w;
  if (dstride < w) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "prefilter_image: destination stride too small");
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  if (mode != VWGPU_PREFILTER_LOG && mode != VWGPU_PREFILTER_MEANSUB) {
    // NullOperation: edge_extend(image) rasterised = a copy (src/vw/Stereo/PreFilter.h:41-47, default :91-93)
    if (d_src != d_dst)
      VWGPU_HIP(ctx, hipMemcpy2DAsync(d_dst, (size_t)dstride * 4, d_src, (size_t)stride * 4, (size_t)w * 4, h,
                                      hipMemcpyDeviceToDevice, ctx->stream));
    return VWGPU_OK;
  }
  float taps[1024];
  const int nt = vwgpu_generate_gaussian_kernel((double)width, 0, taps, 1024);   // gaussian_filter(image, kernel_width)
  if (nt < 0) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "prefilter_image: prefilter width %g too large", (double)width);
  if (nt == 0) {   // sigma == 0: empty kernels, gaussian_filter is the identity
    if (mode == VWGPU_PREFILTER_MEANSUB) return vwgpu_launch_subtract(ctx, d_src, stride, d_src, stride, w, h, d_dst, dstride);
    const float lap0[9] = {0, 1, 0, 1, -4, 1, 0, 1, 0};
    return vwgpu_launch_conv2d(ctx, d_src, w, h, stride, lap0, 3, 3, 1, 1, 0, d_dst, dstride);
  }
  rc = vwgpu_arena_reserve(ctx, &ctx->filt, (size_t)w * h * sizeof(float));
  if (rc) return rc;
  float* g = static_cast<float*>(ctx->filt.base);
  rc = vwgpu_launch_sepconv(ctx, d_src, w, h, stride, taps, nt, (nt - 1) / 2, taps, nt, (nt - 1) / 2, 0, 1, g, w);
  if (rc) return rc;
  if (mode == VWGPU_PREFILTER_MEANSUB)                                            // PreFilter.h:73
    return vwgpu_launch_subtract(ctx, d_src, stride, g, w, w, h, d_dst, dstride);
  const float lap[9] = {0, 1, 0, 1, -4, 1, 0, 1, 0};                               // Filter.h:320-335
  return vwgpu_launch_conv2d(ctx, g, w, h, w, lap, 3, 3, 1, 1, 0, d_dst, dstride);
}

}  // extern "C"

// prefilter_image for several dense images in two launches (the pyramid of a tile: both images of every level, CorrelationView.cc:232-236)
int vwgpu_prefilter_images_dev(vwgpu_ctx* ctx, int n, const float* const* srcs, const int* ws, const int* hs, int mode, float width, float* const* dsts) {
  if (n <= 0) return VWGPU_OK;
  if (mode != VWGPU_PREFILTER_LOG && mode != VWGPU_PREFILTER_MEANSUB) {
    for (int i = 0; i < n; ++i)
      if (srcs[i] != dsts[i]) VWGPU_HIP(ctx, hipMemcpyAsync(dsts[i], srcs[i], (size_t)ws[i] * hs[i] * 4, hipMemcpyDeviceToDevice, ctx->stream));
    return VWGPU_OK;
  }
  float taps[1024];
  const int nt = vwgpu_generate_gaussian_kernel((double)width, 0, taps, 1024);
  if (nt < 0) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "prefilter_image: prefilter width %g too large", (double)width);
  const float lap[9] = {0, 1, 0, 1, -4, 1, 0, 1, 0};                               // Filter.h:320-335
  size_t total = 0;
  for (int i = 0; i < n; ++i) total += vwgpu_align_up((size_t)ws[i] * hs[i] * sizeof(float), 256);
  int rc = nt ? vwgpu_arena_reserve(ctx, &ctx->filt, total) : VWGPU_OK;
  if (rc) return rc;
  for (int i0 = 0; i0 < n; i0 += VWGPU_MAX_IMG_JOBS) {
    const int m = std::min(VWGPU_MAX_IMG_JOBS, n - i0);
    vwgpu_img_job a[VWGPU_MAX_IMG_JOBS], b[VWGPU_MAX_IMG_JOBS];
    char* g = static_cast<char*>(ctx->filt.base);
    for (int i = 0; i < i0; ++i) g += vwgpu_align_up((size_t)ws[i] * hs[i] * sizeof(float), 256);
    for (int k = 0; k < m; ++k) {
      const int i = i0 + k, w = ws[i], h = hs[i];
      float* gi = reinterpret_cast<float*>(g);
      g += vwgpu_align_up((size_t)w * h * sizeof(float), 256);
      if (nt == 0) gi = const_cast<float*>(srcs[i]);                           // sigma == 0: gaussian_filter is the identity
      a[k] = vwgpu_img_job{srcs[i], w, w, h, gi, w, w, h, 0, 0, nullptr, 0};    // gaussian_filter(image, kernel_width)
      if (mode == VWGPU_PREFILTER_MEANSUB) b[k] = vwgpu_img_job{srcs[i], w, w, h, dsts[i], w, w, h, 0, 0, gi, w};     // image - gaussian (PreFilter.h:73)
      else b[k] = vwgpu_img_job{gi, w, w, h, dsts[i], w, w, h, 0, 0, nullptr, 0};                                      // laplacian_filter(gaussian)
    }
    if (nt && (rc = vwgpu_launch_sepconv_jobs(ctx, a, m, taps, nt, (nt - 1) / 2, taps, nt, (nt - 1) / 2, 0, 1))) return rc;
    if (mode == VWGPU_PREFILTER_MEANSUB) rc = vwgpu_launch_subtract_jobs(ctx, b, m);
    else rc = vwgpu_launch_conv2d_jobs(ctx, b, m, lap, 3, 3, 1, 1, 0);
    if (rc) return rc;
  }
  return VWGPU_OK;
}

extern "C" {

int vwgpu_prefilter_image(vwgpu_ctx* ctx, const float* src, int w, int h, ptrdiff_t stride,
                          int mode, float width, float* dst, ptrdiff_t dstride) {
  int rc = check_image(ctx, "prefilter_image", src, w, h, stride, dst);
  if (rc) return rc;
  if (dstride == 0) dstride = w;
  return staged<float>(ctx, src, w, h, stride, dst, w, h, dstride, [&](float* ds, float* dd) {
    return vwgpu_prefilter_image_dev(ctx, ds, w, h, w, mode, width, dd, w);
  });
}


// ---- parabola sub-pixel -------------------------------------------------------------------------------------

int vwgpu_parabola_subpixel_dev(vwgpu_ctx* ctx, const float* d_disp, int w, int h, ptrdiff_t dstride,
                                const float* d_left, ptrdiff_t lstride,
                                const float* d_right, int rw, int rh, ptrdiff_t rstride,
                                int mode, float width, int kx, int ky, float* d_out, ptrdiff_t ostride) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  ctx->err.clear();
  if (!d_disp || !d_left || !d_right || !d_out || w <= 0 || h <= 0 || rw <= 0 || rh <= 0)
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "parabola_subpixel: empty image or null pointer");
  if (kx < 1 || ky < 1 || kx % 2 != 1 || ky % 2 != 1)
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "parabola_subpixel: Kernel input not sized with odd values.");
  if (dstride == 0) dstride = w;
  if (ostride == 0) ostride = w;
  if (lstride == 0) lstride = w;
  if (rstride == 0) rstride = rw;
  if (dstride < w || ostride < w || lstride < w || rstride < rw)
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "parabola_subpixel: row stride smaller than row width");
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));

  // entire_search_range = [min, max+1) expanded by 1 over the whole tile (ParabolaSubpixelView.cc:287-290)
  int rc = vwgpu_arena_reserve(ctx, &ctx->misc, 256);
  if (rc) return rc;
  int* d_range = static_cast<int*>(ctx->misc.base);
  // class of the imagery, measured in the same launch and the same round trip: small integers take the integer SAD form of the kernel
  int* d_grain = d_range + 4;
  int grain[3] = {INT_MAX, INT_MIN, 0};
  rc = vwgpu_launch_parabola_prepass(ctx, d_disp, w, h, dstride, d_range, d_left, w, h, lstride, d_right, rw, rh, rstride,
                                     mode == VWGPU_PREFILTER_NONE ? d_grain : nullptr);
  if (rc) return rc;
  if (mode == VWGPU_PREFILTER_NONE) VWGPU_HIP(ctx, hipMemcpyAsync(grain, d_grain, sizeof grain, hipMemcpyDeviceToHost, ctx->stream));
  int r4[4];
  VWGPU_HIP(ctx, hipMemcpyAsync(r4, d_range, sizeof r4, hipMemcpyDeviceToHost, ctx->stream));
  VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));              // the ROI sizes depend on the data
  // 0: any float (float64 sums); 1: integers of magnitude below 2^21; 2: integers in [0,255]
  int integer_class = 0;
  if (mode == VWGPU_PREFILTER_NONE && !(grain[2] & 1) && (grain[0] == INT_MAX || (grain[0] >= 0 && grain[1] <= 20)))
    integer_class = ((grain[0] == INT_MAX || grain[1] <= 7) && !(grain[2] & 2)) ? 2 : 1;     // bytes need non-negative pixels
  // class 1 sums |a - b| < 2^(hi + 2) over kx * ky pixels in 32-bit unsigned integers: the window must not be able to wrap them
  // (e.g. 15 x 69 pixels of 2^21-sized data would); otherwise the float64 sums
  if (integer_class == 1 && grain[0] != INT_MAX && ((long long)kx * ky << (grain[1] + 2)) >= (1LL << 32)) integer_class = 0;
  const long long rminx = (long long)r4[0] - 1, rminy = (long long)r4[1] - 1;
  const long long rsx = (long long)r4[2] + 1 - r4[0] + 2, rsy = (long long)r4[3] + 1 - r4[1] + 2;
  if (rsx > 8192 || rsy > 8192 || rminx < -(1 << 20) || rminx > (1 << 20) || rminy < -(1 << 20) || rminy > (1 << 20))
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "parabola_subpixel: disparity range [%d,%d]x[%d,%d] is not plausible", r4[0], r4[2], r4[1], r4[3]);
  const int hx = kx / 2, hy = ky / 2;
  // left_region = bbox -/+ half_kernel; right_region = left_region + range.min, max += range.size  (:293-298)
  const int lrw = w + 2 * hx, lrh = h + 2 * hy;
  const int rrw = lrw + (int)rsx, rrh = lrh + (int)rsy;
  const size_t lb = vwgpu_align_up((size_t)lrw * lrh * 4, 256), rb = vwgpu_align_up((size_t)rrw * rrh * 4, 256);
  // integers in [0,255]: the kernel reads byte copies of the two rasters (built in the scratch area once the crops are done)
  const int lp8 = vwgpu_parabola_u8_pitch(lrw), rp8 = vwgpu_parabola_u8_pitch(rrw);
  const size_t l8b = vwgpu_align_up((size_t)lp8 * lrh + 64, 256), r8b = vwgpu_align_up((size_t)rp8 * rrh + 64, 256);
  const size_t sb = std::max(vwgpu_align_up(std::max({(size_t)lrw * lrh, (size_t)rrw * rrh, (size_t)w * h, (size_t)rw * rh}) * 4, 256), l8b + r8b);
  rc = vwgpu_arena_reserve(ctx, &ctx->filt, lb + rb + sb);
  if (rc) return rc;
  char* base = static_cast<char*>(ctx->filt.base);
  float* lras = reinterpret_cast<float*>(base);
  float* rras = reinterpret_cast<float*>(base + lb);
  float* scratch = reinterpret_cast<float*>(base + lb + rb);
  if (integer_class == 2 && kx >= 3 && kx <= 15 && (kx & 1)) {                   // the sizes parabola_kernel<K, 2> is instantiated for
    // integers in [0,255] (PREFILTER_NONE): the byte rasters straight from the images, edge extension included
    uint8_t* l8 = reinterpret_cast<uint8_t*>(scratch);
    uint8_t* r8 = l8 + l8b;
    vwgpu_launch_f32_ext_to_u8_raster(ctx, d_left, lstride, w, h, -hx, -hy, lrw, lrh, l8, lp8);
    vwgpu_launch_f32_ext_to_u8_raster(ctx, d_right, rstride, rw, rh, -hx + (int)rminx, -hy + (int)rminy, rrw, rrh, r8, rp8);
    return vwgpu_launch_parabola(ctx, d_disp, w, h, dstride, reinterpret_cast<const float*>(l8), lp8, reinterpret_cast<const float*>(r8), rp8,
                                 (int)rminx, (int)rminy, kx, ky, d_out, ostride, 2);
  }
  rc = vwgpu_prefilter_region(ctx, d_left, w, h, lstride, mode, width, -hx, -hy, lrw, lrh, lras, scratch);
  if (rc) return rc;
  rc = vwgpu_prefilter_region(ctx, d_right, rw, rh, rstride, mode, width, -hx + (int)rminx, -hy + (int)rminy, rrw, rrh, rras, scratch);
  if (rc) return rc;
  if (integer_class == 2) integer_class = 1;                                      // other kernel sizes: the run-time loop on the float rasters
  return vwgpu_launch_parabola(ctx, d_disp, w, h, dstride, lras, lrw, rras, rrw, (int)rminx, (int)rminy, kx, ky, d_out, ostride, integer_class);
}

int vwgpu_parabola_subpixel(vwgpu_ctx* ctx, const float* disp, int w, int h, ptrdiff_t dstride,
                            const float* left, ptrdiff_t lstride, const float* right, int rw, int rh, ptrdiff_t rstride,
                            int mode, float width, int kx, int ky, float* out, ptrdiff_t ostride) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  if (!disp || !left || !right || !out || w <= 0 || h <= 0 || rw <= 0 || rh <= 0)
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "parabola_subpixel: empty image or null pointer");
  if (dstride == 0) dstride = w;
  if (ostride == 0) ostride = w;
  if (lstride == 0) lstride = w;
  if (rstride == 0) rstride = rw;
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  const size_t db = vwgpu_align_up((size_t)w * h * 12, 256), lb = vwgpu_align_up((size_t)w * h * 4, 256),
               rb = vwgpu_align_up((size_t)rw * rh * 4, 256);
  int rc = vwgpu_arena_reserve(ctx, &ctx->staging, 2 * db + lb + rb);
  if (rc) return rc;
  char* base = static_cast<char*>(ctx->staging.base);
  float* d_d = reinterpret_cast<float*>(base);
  float* d_o = reinterpret_cast<float*>(base + db);
  float* d_l = reinterpret_cast<float*>(base + 2 * db);
  float* d_r = reinterpret_cast<float*>(base + 2 * db + lb);
  VWGPU_HIP(ctx, hipMemcpy2DAsync(d_d, (size_t)w * 12, disp, (size_t)dstride * 12, (size_t)w * 12, h, hipMemcpyHostToDevice, ctx->stream));
  VWGPU_HIP(ctx, hipMemcpy2DAsync(d_l, (size_t)w * 4, left, (size_t)lstride * 4, (size_t)w * 4, h, hipMemcpyHostToDevice, ctx->stream));
  VWGPU_HIP(ctx, hipMemcpy2DAsync(d_r, (size_t)rw * 4, right, (size_t)rstride * 4, (size_t)rw * 4, rh, hipMemcpyHostToDevice, ctx->stream));
  rc = vwgpu_parabola_subpixel_dev(ctx, d_d, w, h, w, d_l, w, d_r, rw, rh, rw, mode, width, kx, ky, d_o, w);
  if (rc) return rc;
  VWGPU_HIP(ctx, hipMemcpy2DAsync(out, (size_t)ostride * 12, d_o, (size_t)w * 12, (size_t)w * 12, h, hipMemcpyDeviceToHost, ctx->stream));
  VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VWGPU_OK;
}

}  // extern "C"
