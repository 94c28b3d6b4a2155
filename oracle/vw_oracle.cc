// vw_oracle.cc — CPU parity oracle: a literal, dependency-free restatement of the Vision Workbench
// reference on the dense block-matching hot path.
//
// TEST INFRASTRUCTURE ONLY (see vw_oracle.h).  Never linked into or loaded by the product.
// Build: g++ -O2 -ffp-contract=off -fno-fast-math (the reference builds -O3 -std=c++14 -msse4.1: no FMA,
// no fast-math; src/vw/CMakeLists.txt:81-88).
//
// Parity pins (re-typed from the reference's own tests, tests/test_oracle_golden.py):
//   fast_box_sum        src/vw/Stereo/tests/TestAlgorithms.cxx:46-174
//   cost functors       src/vw/Stereo/tests/TestCostFunctions.cxx:48-79
//   calc_disparity      src/vw/Stereo/tests/TestCorrelation.cxx:45-214
//   L/R check           src/vw/Stereo/tests/TestCorrelate.cxx:29-55
#include "vw_oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>
#include <thread>
#include <utility>
#include <vector>

namespace {

// Minimal stand-in for vw::ImageView<T>: contiguous row-major, zero-initialised
// (src/vw/Image/ImageView.h:209-239).
template <class T>
struct Img {
  int w = 0, h = 0;
  std::vector<T> d;
  Img() {}
  Img(int w_, int h_) : w(w_), h(h_), d((size_t)w_ * h_, T()) {}
  T& operator()(int x, int y) { return d[(size_t)y * w + x]; }
  const T& operator()(int x, int y) const { return d[(size_t)y * w + x]; }
};

// fast_box_sum<AccumT>, src/vw/Stereo/Algorithms.h:43-129.  `in(x,y)` is any pixel accessor.
// Same statement order as the reference so that rounding (when sums are inexact) matches:
//   col_sum starts 0 and adds the first ky rows top->bottom (:62-75)
//   row_sum = accumulate(col_sum[0..kx)) left->right (:84), slide row_sum += (front - back) (:92)
//   col_sum[x] += in(x, y+ky); col_sum[x] -= in(x, y)  as two statements (:100-103)
template <class InT>
int box_sum(const InT* in, int64_t istride, int w, int h, int kx, int ky, double* out) {
  if (kx % 2 != 1 || ky % 2 != 1) return -1;  // VW_ASSERT, Algorithms.h:45-46
  if (w < kx || h < ky) return -2;
  const int ow = w - kx + 1, oh = h - ky + 1;
  std::vector<double> col_sum((size_t)w, 0.0);
  for (int r = 0; r < ky; ++r)
    for (int x = 0; x < w; ++x) col_sum[x] += in[(int64_t)r * istride + x];

  double* dst = out;
  for (int y = 0; y < oh; ++y) {
    double row_sum = 0;
    row_sum = std::accumulate(&col_sum[0], &col_sum[0] + kx, row_sum);
    int cback = 0, cfront = kx;
    while (cfront != w) {
      *dst++ = row_sum;
      row_sum += col_sum[cfront++] - col_sum[cback++];
    }
    *dst++ = row_sum;
    if (y != oh - 1) {
      const InT* back = in + (int64_t)y * istride;
      const InT* front = in + (int64_t)(y + ky) * istride;
      for (int x = 0; x < w; ++x) {
        col_sum[x] += front[x];
        col_sum[x] -= back[x];
      }
    }
  }
  (void)ow;
  return 0;
}

// Cost functors, src/vw/Stereo/CostFunctions.h:72-141: element computed in FLOAT, widened to double on store.
inline double cost_abs(float a, float b) { return (double)std::fabs(a - b); }        // :79-81
inline double cost_sq(float a, float b)  { float d = a - b; return (double)(d * d); } // :94-101 (float*float)
inline double cost_xc(float a, float b)  { return (double)(a * b); }                 // :120-127

// best_of_search_convolution<CostT, PixelGray<float>>, src/vw/Stereo/Correlation.cc:33-137.
int best_of_search(int cost_type, const float* left, int lw, int lh, int64_t ls,
                   const float* right, int rw, int rh, int64_t rs,
                   int kx, int ky, int sx, int sy, int32_t* out) {
  if (kx % 2 != 1 || ky % 2 != 1) return -1;
  if (lw < kx || lh < ky || sx < 1 || sy < 1) return -2;
  if (rw < lw + sx - 1 || rh < lh + sy - 1) return -3;
  const int ow = lw - kx + 1, oh = lh - ky + 1;
  const size_t on = (size_t)ow * oh;
  const int VALID = std::numeric_limits<int32_t>::max();

  // disparity_map filled with valid (0,0)  (:52-53)
  for (size_t i = 0; i < on; ++i) { out[3*i] = 0; out[3*i+1] = 0; out[3*i+2] = VALID; }
  std::vector<std::pair<double,double>> quality(on);     // first = best, second = worst (:55)
  std::vector<double> cost_metric(on);                   // (:58)
  std::vector<double> cost_applied((size_t)lw * lh);     // (:59)
  std::vector<float>  right_crop((size_t)lw * lh);       // (:60)

  // NCCCost ctor: precision images = 1.0 / fast_box_sum<double>(square(img))  (CostFunctions.h:214-219)
  std::vector<double> lprec, rprec;
  int rpw = 0;
  if (cost_type == VWO_CROSS_CORRELATION) {
    const int rcw = lw + sx - 1, rch = lh + sy - 1;      // right raster as cropped by calc_disparity
    std::vector<float> sq((size_t)rcw * rch);
    lprec.resize(on);
    for (int y = 0; y < lh; ++y) for (int x = 0; x < lw; ++x) { float v = left[y*ls + x]; sq[(size_t)y*lw + x] = v * v; }
    box_sum(sq.data(), lw, lw, lh, kx, ky, lprec.data());
    for (auto& v : lprec) v = 1.0 / v;
    rpw = rcw - kx + 1;
    const int rph = rch - ky + 1;
    rprec.resize((size_t)rpw * rph);
    for (int y = 0; y < rch; ++y) for (int x = 0; x < rcw; ++x) { float v = right[y*rs + x]; sq[(size_t)y*rcw + x] = v * v; }
    box_sum(sq.data(), rcw, rcw, rch, kx, ky, rprec.data());
    for (auto& v : rprec) v = 1.0 / v;
  }

  for (int dy = 0; dy != sy; ++dy) {
    for (int dx = 0; dx != sx; ++dx) {
      // right_raster_crop = crop(right_raster, bbox(left)+disparity)  (:79)
      for (int y = 0; y < lh; ++y)
        std::memcpy(&right_crop[(size_t)y * lw], right + (int64_t)(y + dy) * rs + dx, sizeof(float) * lw);
      // cost_applied = cost_function(left, right_crop)  (:80)
      for (int y = 0; y < lh; ++y) {
        const float* l = left + (int64_t)y * ls;
        const float* r = &right_crop[(size_t)y * lw];
        double* c = &cost_applied[(size_t)y * lw];
        switch (cost_type) {
          case VWO_CROSS_CORRELATION: for (int x = 0; x < lw; ++x) c[x] = cost_xc(l[x], r[x]); break;
          case VWO_SQUARED_DIFFERENCE: for (int x = 0; x < lw; ++x) c[x] = cost_sq(l[x], r[x]); break;
          default: for (int x = 0; x < lw; ++x) c[x] = cost_abs(l[x], r[x]); break;
        }
      }
      // cost_metric = fast_box_sum<double>(cost_applied, kernel)  (:81)
      box_sum(cost_applied.data(), lw, lw, lh, kx, ky, cost_metric.data());
      // cost_function.cost_modification(cost_metric, disparity)  (:82; NCC: CostFunctions.h:227-231)
      if (cost_type == VWO_CROSS_CORRELATION) {
        for (int y = 0; y < oh; ++y)
          for (int x = 0; x < ow; ++x)
            cost_metric[(size_t)y*ow + x] *= std::sqrt(lprec[(size_t)y*ow + x] * rprec[(size_t)(y+dy)*rpw + (x+dx)]);
      }
      // compare loop (:91-117)
      if (dx != 0 || dy != 0) {
        if (cost_type == VWO_CROSS_CORRELATION) {
          for (size_t i = 0; i < on; ++i) {
            const double c = cost_metric[i];
            if (c > quality[i].first) { quality[i].first = c; out[3*i] = dx; out[3*i+1] = dy; }
            else if (!(c > quality[i].second)) quality[i].second = c;
          }
        } else {
          for (size_t i = 0; i < on; ++i) {
            const double c = cost_metric[i];
            if (c < quality[i].first) { quality[i].first = c; out[3*i] = dx; out[3*i+1] = dy; }
            else if (!(c < quality[i].second)) quality[i].second = c;
          }
        }
      } else {
        for (size_t i = 0; i < on; ++i) quality[i].first = quality[i].second = cost_metric[i];
      }
    }
  }
  // validity pass (:121-133)
  for (size_t i = 0; i < on; ++i)
    if (quality[i].first == quality[i].second) out[3*i+2] = 0;
  return 0;
}

}  // namespace

extern "C" {

int vwo_fast_box_sum_f32(const float* in, int w, int h, int kx, int ky, double* out) {
  return box_sum(in, w, w, h, kx, ky, out);
}
int vwo_fast_box_sum_f64(const double* in, int w, int h, int kx, int ky, double* out) {
  return box_sum(in, w, w, h, kx, ky, out);
}

int vwo_cost_image(int cost_type, const float* a, const float* b, int w, int h, double* out) {
  const size_t n = (size_t)w * h;
  for (size_t i = 0; i < n; ++i) {
    switch (cost_type) {
      case VWO_CROSS_CORRELATION: out[i] = cost_xc(a[i], b[i]); break;
      case VWO_SQUARED_DIFFERENCE: out[i] = cost_sq(a[i], b[i]); break;
      case VWO_ABSOLUTE_DIFFERENCE: out[i] = cost_abs(a[i], b[i]); break;
      default: return -1;
    }
  }
  return 0;
}

int vwo_calc_disparity(int cost_type, const float* left, int lw, int lh, int64_t ls,
                       const float* right, int rw, int rh, int64_t rs,
                       int kx, int ky, int sx, int sy, int32_t* out) {
  return best_of_search(cost_type, left, lw, lh, ls, right, rw, rh, rs, kx, ky, sx, sy, out);
}

int vwo_calc_disparity_tiled(int cost_type, const float* left, int lw, int lh,
                             const float* right, int rw, int rh,
                             int kx, int ky, int sx, int sy, int32_t* out,
                             int tile, int threads, int max_tiles, int64_t* pixels_done) {
  if (rw < lw + sx - 1 || rh < lh + sy - 1) return -3;
  const int ow = lw - kx + 1, oh = lh - ky + 1;
  if (ow < 1 || oh < 1 || tile < 1 || threads < 1) return -2;
  const int tx = (ow + tile - 1) / tile, ty = (oh + tile - 1) / tile;
  int ntiles = tx * ty;
  if (max_tiles > 0 && max_tiles < ntiles) ntiles = max_tiles;
  std::atomic<int> next(0);
  std::atomic<int64_t> done(0);
  std::atomic<int> err(0);
  auto worker = [&]() {
    std::vector<int32_t> tmp;
    for (;;) {
      const int t = next.fetch_add(1);
      if (t >= ntiles) break;
      const int x0 = (t % tx) * tile, y0 = (t / tx) * tile;
      const int tw = std::min(tile, ow - x0), th = std::min(tile, oh - y0);
      // padded crops: left (tw+kx-1) x (th+ky-1) at (x0,y0); right grown by s-1 (Correlation.cc:356-359)
      const int clw = tw + kx - 1, clh = th + ky - 1;
      tmp.resize((size_t)tw * th * 3);
      int rc = best_of_search(cost_type, left + (int64_t)y0 * lw + x0, clw, clh, lw,
                              right + (int64_t)y0 * rw + x0, clw + sx - 1, clh + sy - 1, rw,
                              kx, ky, sx, sy, tmp.data());
      if (rc) { err = rc; break; }
      for (int y = 0; y < th; ++y)
        std::memcpy(out + ((int64_t)(y0 + y) * ow + x0) * 3, &tmp[(size_t)y * tw * 3], sizeof(int32_t) * 3 * tw);
      done += (int64_t)tw * th;
    }
  };
  std::vector<std::thread> pool;
  for (int i = 1; i < threads; ++i) pool.emplace_back(worker);
  worker();
  for (auto& t : pool) t.join();
  if (pixels_done) *pixels_done = done;
  return err;
}

int vwo_cross_corr_consistency_check(int32_t* l2r, int lw, int lh,
                                     const int32_t* r2l, int rw, int rh, float thr) {
  // src/vw/Stereo/Correlate.cc:1462-1495
  for (int r = 0; r < lh; ++r) {
    for (int c = 0; c < lw; ++c) {
      int32_t* p = l2r + ((int64_t)r * lw + c) * 3;
      const int x = c + p[0], y = r + p[1];
      if (x < 0 || x >= rw || y < 0 || y >= rh) { p[2] = 0; continue; }
      const int32_t* q = r2l + ((int64_t)y * rw + x) * 3;
      if (!p[2] || !q[2]) { p[2] = 0; continue; }
      float diff = (float)std::max(std::fabs((double)(p[0] + q[0])), std::fabs((double)(p[1] + q[1])));
      if (!(thr >= diff)) p[2] = 0;
    }
  }
  return 0;
}

}  // extern "C"

// ---- image filters on the path ---------------------------------------------------------------------------

namespace {

// Edge extension of src/vw/Image/EdgeExtension.h: Constant = clamp coordinates, Zero = 0 outside.
template <class T>
inline T ext_at(const T* src, int w, int h, int x, int y, int edge) {
  if (edge == VWO_EDGE_ZERO) {
    if (x < 0 || y < 0 || x >= w || y >= h) return T(0);
    return src[(size_t)y * w + x];
  }
  x = x < 0 ? 0 : (x >= w ? w - 1 : x);
  y = y < 0 ? 0 : (y >= h ? h - 1 : y);
  return src[(size_t)y * w + x];
}

// generate_gaussian_kernel<KernelT>, src/vw/Image/Filter.tcc:37-78.
template <class K>
int gaussian_kernel(double sigma, int size, K* out, int cap) {
  if (sigma == 0) return 0;
  if (size == 0) {                                  // compute_kernel_size, Filter.cc:32-37
    size = (int)(7 * sigma);
    if (size < 3) size = 3;
    else if (size % 2 == 0) size -= 1;
  }
  if (size > cap) return -1;
  const int center = size / 2;
  double sum = 0.0, tap;
  const double z = 1 / (std::sqrt(2.0) * sigma);
  if (size % 2 == 0) {
    for (int i = 0; i < center; ++i) {
      tap = std::erf((i + 1.0) * z) - std::erf(i * z);
      sum += tap;
      out[center + i] = out[center - i - 1] = (K)tap;
    }
    sum *= 2.0;
  } else {
    for (int i = 1; i <= center; ++i) {
      tap = std::erf((i + 0.5) * z) - std::erf((i - 0.5) * z);
      sum += tap;
      out[center + i] = out[center - i] = (K)tap;
    }
    sum *= 2.0;
    tap = std::erf(0.5 * z) - std::erf(-0.5 * z);
    sum += tap;
    out[center] = (K)tap;
  }
  const double norm = 1.0 / sum;
  for (int i = 0; i < size; ++i) out[i] *= norm;   // KernelT *= double, as in the reference (:76-77)
  return size;
}

// SeparableConvolutionView::rasterize over the whole image + SubsampleView  (Convolution.h:275-328).
// correlate_1d_at_point(src, kernel.rbegin(), n): result = 0; result += k[n-1-i] * s[i], i = 0..n-1  (:53-65).
template <class T>
int sepconv(const T* src, int w, int h, const T* xk, int nx, int cx, const T* yk, int ny, int cy,
            int edge, int step, T* dst) {
  if (w <= 0 || h <= 0 || step < 1 || nx < 0 || ny < 0) return -1;
  const int x_lo = nx ? nx - cx - 1 : 0, y_lo = ny ? ny - cy - 1 : 0;   // child_bbox.min() -= ...  (:282)
  const int y_hi = ny ? cy : 0;
  const int ch = h + y_lo + y_hi;                                      // child_bbox.height()
  std::vector<T> work((size_t)w * ch);
  for (int yy = 0; yy < ch; ++yy) {
    const int sy = yy - y_lo;
    for (int x = 0; x < w; ++x) {
      if (nx) {
        T result = T(0);
        for (int i = 0; i < nx; ++i) result += xk[nx - 1 - i] * ext_at(src, w, h, x - x_lo + i, sy, edge);
        work[(size_t)yy * w + x] = result;
      } else {
        work[(size_t)yy * w + x] = ext_at(src, w, h, x, sy, edge);
      }
    }
  }
  const int ow = 1 + (w - 1) / step, oh = 1 + (h - 1) / step;            // SubsampleView sizes (Manipulation.h:238-243)
  for (int oy = 0; oy < oh; ++oy) {
    const int y = oy * step;
    for (int ox = 0; ox < ow; ++ox) {
      const int x = ox * step;
      if (ny) {
        T result = T(0);
        for (int j = 0; j < ny; ++j) result += yk[ny - 1 - j] * work[(size_t)(y + j) * w + x];
        dst[(size_t)oy * ow + ox] = result;
      } else {
        dst[(size_t)oy * ow + ox] = work[(size_t)(y + y_lo) * w + x];
      }
    }
  }
  return 0;
}

// ConvolutionView with the kernel rotated by 180 degrees (Convolution.h:105-170), correlate_2d_at_point (:66-88).
template <class T>
int conv2d(const T* src, int w, int h, const T* k, int kw, int kh, int mci, int mcj, int edge, T* dst) {
  if (w <= 0 || h <= 0 || kw <= 0 || kh <= 0) return -1;
  const int ci = kw - 1 - mci, cj = kh - 1 - mcj;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      T result = T(0);
      for (int j = 0; j < kh; ++j)
        for (int i = 0; i < kw; ++i)
          result += k[(size_t)(kh - 1 - j) * kw + (kw - 1 - i)] * ext_at(src, w, h, x - ci + i, y - cj + j, edge);
      dst[(size_t)y * w + x] = result;
    }
  return 0;
}

}  // namespace

extern "C" {

int vwo_generate_gaussian_kernel_f32(double sigma, int size, float* out, int cap) { return gaussian_kernel(sigma, size, out, cap); }
int vwo_generate_gaussian_kernel_f64(double sigma, int size, double* out, int cap) { return gaussian_kernel(sigma, size, out, cap); }

int vwo_separable_convolution_f32(const float* src, int w, int h, const float* xk, int nx, int cx,
                                  const float* yk, int ny, int cy, int edge, int subsample, float* dst) {
  return sepconv(src, w, h, xk, nx, cx, yk, ny, cy, edge, subsample, dst);
}
int vwo_separable_convolution_f64(const double* src, int w, int h, const double* xk, int nx, int cx,
                                  const double* yk, int ny, int cy, int edge, int subsample, double* dst) {
  return sepconv(src, w, h, xk, nx, cx, yk, ny, cy, edge, subsample, dst);
}
int vwo_convolution_2d_f32(const float* src, int w, int h, const float* k, int kw, int kh, int ci, int cj, int edge, float* dst) {
  return conv2d(src, w, h, k, kw, kh, ci, cj, edge, dst);
}
int vwo_convolution_2d_f64(const double* src, int w, int h, const double* k, int kw, int kh, int ci, int cj, int edge, double* dst) {
  return conv2d(src, w, h, k, kw, kh, ci, cj, edge, dst);
}

int vwo_subsample_mask_by_two(const uint8_t* src, int w, int h, uint8_t* dst) {
  // SubsampleMaskByTwoFunc over a ZeroEdgeExtension'd image (Filter.h:133-136), then subsample(...,2)
  if (w <= 0 || h <= 0) return -1;
  const int ow = 1 + (w - 1) / 2, oh = 1 + (h - 1) / 2;
  for (int oy = 0; oy < oh; ++oy)
    for (int ox = 0; ox < ow; ++ox) {
      const int x = 2 * ox, y = 2 * oy;
      int count = 0;
      if (ext_at(src, w, h, x, y, VWO_EDGE_ZERO)) count++;
      if (ext_at(src, w, h, x + 1, y, VWO_EDGE_ZERO)) count++;
      if (ext_at(src, w, h, x, y + 1, VWO_EDGE_ZERO)) count++;
      if (ext_at(src, w, h, x + 1, y + 1, VWO_EDGE_ZERO)) count++;
      dst[(size_t)oy * ow + ox] = count > 1 ? 255 : 0;
    }
  return 0;
}

int vwo_prefilter_image(const float* src, int w, int h, int mode, float width, float* dst) {
  const size_t n = (size_t)w * h;
  if (mode == VWO_PREFILTER_NONE) { std::memcpy(dst, src, n * sizeof(float)); return 0; }   // NullOperation
  float taps[1024];
  const int nt = gaussian_kernel<float>((double)width, 0, taps, 1024);                       // gaussian_filter(image, width)
  if (nt < 0) return -1;
  std::vector<float> g(n);
  int rc = sepconv<float>(src, w, h, taps, nt, (nt - 1) / 2, taps, nt, (nt - 1) / 2, VWO_EDGE_CONSTANT, 1, g.data());
  if (rc) return rc;
  if (mode == VWO_PREFILTER_MEANSUB) {                                                        // I - gaussian (PreFilter.h:73)
    for (size_t i = 0; i < n; ++i) dst[i] = src[i] - g[i];
    return 0;
  }
  const float lap[9] = {0, 1, 0, 1, -4, 1, 0, 1, 0};                                          // laplacian_filter (Filter.h:320-335)
  return conv2d<float>(g.data(), w, h, lap, 3, 3, 1, 1, VWO_EDGE_CONSTANT, dst);
}

}  // extern "C"
