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

// The same with the optional lr_disp_diff image (:1480-1484): PixelMask<float> = {value, valid} per pixel, written for the
// kept pixels only, at (c + ulx, r + uly).
int vwo_cross_corr_consistency_check_diff(int32_t* l2r, int lw, int lh, const int32_t* r2l, int rw, int rh, float thr,
                                          float* diff2, int dcols, int drows, int ulx, int uly) {
  for (int r = 0; r < lh; ++r) {
    for (int c = 0; c < lw; ++c) {
      int32_t* p = l2r + ((int64_t)r * lw + c) * 3;
      const int x = c + p[0], y = r + p[1];
      if (x < 0 || x >= rw || y < 0 || y >= rh) { p[2] = 0; continue; }
      const int32_t* q = r2l + ((int64_t)y * rw + x) * 3;
      if (!p[2] || !q[2]) { p[2] = 0; continue; }
      float diff = (float)std::max(std::fabs((double)(p[0] + q[0])), std::fabs((double)(p[1] + q[1])));
      if (thr >= diff) {
        if (diff2) {
          const int dx = c + ulx, dy = r + uly;
          if (dx < 0 || dy < 0 || dx >= dcols || dy >= drows) return -1;
          diff2[((size_t)dy * dcols + dx) * 2] = diff; diff2[((size_t)dy * dcols + dx) * 2 + 1] = 1.0f;
        }
      } else {
        p[2] = 0;
      }
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

// ---- zone subdivision and parabola sub-pixel refinement ---------------------------------------------------------

namespace {

// vw::BBox2i with the exact semantics of src/vw/Math/BBox.tcc (default box :40-45, grow :82-108, crop :110-121,
// empty/width/height/area :156-197).
struct Box {
  int minx, miny, maxx, maxy;
  Box() { const int big = std::numeric_limits<int32_t>::max() - 1; minx = miny = big; maxx = maxy = -big; }
  Box(int x0, int y0, int x1, int y1) : minx(x0), miny(y0), maxx(x1), maxy(y1) {}
  static Box xywh(int x, int y, int w, int h) { return Box(x, y, x + w, y + h); }
  bool empty() const { return minx >= maxx || miny >= maxy; }
  int width() const { return empty() ? 0 : maxx - minx; }
  int height() const { return empty() ? 0 : maxy - miny; }
  int area() const { return empty() ? 0 : (maxx - minx) * (maxy - miny); }
  int sizex() const { return maxx - minx; }
  int sizey() const { return maxy - miny; }
  void grow_pt(int x, int y) { if (x > maxx) maxx = x; if (x < minx) minx = x; if (y > maxy) maxy = y; if (y < miny) miny = y; }
  void grow(Box const& b) { if (b.empty()) return; grow_pt(b.minx, b.miny); grow_pt(b.maxx, b.maxy); }
  void crop(Box const& b) { if (minx < b.minx) minx = b.minx; if (maxx > b.maxx) maxx = b.maxx; if (miny < b.miny) miny = b.miny; if (maxy > b.maxy) maxy = b.maxy; }
  void expand(int n) { if (empty()) return; minx -= n; miny -= n; maxx += n; maxy += n; }   // BBox::expand (BBox.tcc:228-234)
  bool operator==(Box const& o) const { return minx == o.minx && miny == o.miny && maxx == o.maxx && maxy == o.maxy; }
  bool operator!=(Box const& o) const { return !(*this == o); }
};

struct Zone { Box region, range; };

// PixelAccumulator<EWMinMaxAccumulator<Vector2i>> over crop(disparity, box): element-wise min/max of the VALID pixels.
struct MinMax {
  bool valid = false; int mnx = 0, mny = 0, mxx = 0, mxy = 0;
  void add(int x, int y) {
    if (!valid) { mnx = mxx = x; mny = mxy = y; valid = true; return; }
    if (x < mnx) mnx = x;
    if (x > mxx) mxx = x;
    if (y < mny) mny = y;
    if (y > mxy) mxy = y;
  }
};
MinMax minmax_in(const int32_t* d, int w, Box const& b) {
  MinMax a;
  for (int y = b.miny; y < b.maxy; ++y)
    for (int x = b.minx; x < b.maxx; ++x) {
      const int32_t* p = d + ((size_t)y * w + x) * 3;
      if (p[2]) a.add(p[0], p[1]);
    }
  return a;
}

// subdivide_regions, src/vw/Stereo/Correlation.cc:139-328 (statement by statement).
bool subdivide(const int32_t* d, int w, int h, Box const& cur, std::vector<Zone>& list, int kx, int ky, int fail_count = 0) {
  const int MIN_REGION_SIZE = 16;
  if (cur.sizex() * cur.sizey() <= 200 || cur.width() < MIN_REGION_SIZE || cur.height() < MIN_REGION_SIZE) {   // :149-162
    Box expanded = cur;
    expanded.expand(1);
    expanded.crop(Box(0, 0, w, h));
    MinMax a = minmax_in(d, w, expanded);
    if (!a.valid) return true;
    list.push_back(Zone{cur, Box(a.mnx, a.mny, a.mxx + 1, a.mxy + 1)});
    return true;
  }
  const int sx = cur.sizex() / 2, sy = cur.sizey() / 2;                                                         // :165-171
  Box q1(cur.minx, cur.miny, cur.minx + sx, cur.miny + sy);
  Box q4(cur.minx + sx, cur.miny + sy, cur.maxx, cur.maxy);
  Box q2(cur.minx + sx, cur.miny, cur.maxx, cur.miny + sy);
  Box q3(cur.minx, cur.miny + sy, cur.minx + sx, cur.maxy);
  Box qs[4] = {q1, q2, q3, q4};
  Box search[4];
  int32_t split_search = 0;
  for (int i = 0; i < 4; ++i) {                                                                                 // :179-218
    MinMax a = minmax_in(d, w, qs[i]);
    if (a.valid) {
      search[i] = Box(a.mnx, a.mny, a.mxx + 1, a.mxy + 1);
      split_search += search[i].area() * ((qs[i].sizex() + kx) * (qs[i].sizey() + ky));
    }
  }
  Box cs;                                                                                                      // :225-239
  if (search[0] != Box()) cs = search[0];
  for (int i = 1; i < 4; ++i) {
    if (search[i] != Box() && cs == Box()) cs = search[i];
    else cs.grow(search[i]);
  }
  const int32_t current_search = cs.area() * ((cur.sizex() + kx) * (cur.sizey() + ky));                         // :241
  const double IMPROVEMENT_RATIO = 0.8;
  if (split_search > current_search * IMPROVEMENT_RATIO && fail_count == 0) {                                   // :245
    std::vector<Zone> failed;
    for (int i = 0; i < 4; ++i)
      if (!subdivide(d, w, h, qs[i], list, kx, ky, fail_count + 1)) failed.push_back(Zone{qs[i], search[i]});
    auto adjacent_same = [](Zone const& a, Zone const& b) {
      return (a.region.minx == b.region.minx || a.region.miny == b.region.miny) && a.range == b.range;
    };
    if (failed.size() == 4) {
      list.push_back(Zone{cur, cs});
      return true;
    } else if (failed.size() == 3) {                                                                            // :264-299
      if (adjacent_same(failed[0], failed[1])) {
        Box m = failed[0].region; m.grow(failed[1].region);
        list.push_back(Zone{m, failed[0].range}); list.push_back(failed[2]); return true;
      }
      if (adjacent_same(failed[1], failed[2])) {
        Box m = failed[1].region; m.grow(failed[2].region);
        list.push_back(Zone{m, failed[1].range}); list.push_back(failed[0]); return true;
      }
      if (adjacent_same(failed[0], failed[2])) {
        Box m = failed[0].region; m.grow(failed[2].region);
        list.push_back(Zone{m, failed[0].range}); list.push_back(failed[1]); return true;
      }
      list.insert(list.end(), failed.begin(), failed.end());
    } else if (failed.size() == 2) {                                                                            // :300-313
      if (adjacent_same(failed[0], failed[1])) {
        Box m = failed[0].region; m.grow(failed[1].region);
        list.push_back(Zone{m, failed[0].range});
        return true;
      }
      list.insert(list.end(), failed.begin(), failed.end());
    } else if (failed.size() == 1) {
      list.push_back(failed[0]);
    }
    return true;
  } else if (split_search > current_search * IMPROVEMENT_RATIO && fail_count > 0) {                           // :319
    return false;
  } else {                                                                                                      // :322-327
    for (int i = 0; i < 4; ++i) subdivide(d, w, h, qs[i], list, kx, ky);
  }
  return true;
}

// prefilter.filter(image) rasterised over a region that may leave the image (see vwo_prefilter_region).
int prefilter_region(const float* src, int w, int h, int mode, float width, int x0, int y0, int bw, int bh, float* dst) {
  if (mode != VWO_PREFILTER_LOG && mode != VWO_PREFILTER_MEANSUB) {      // NullOperation: edge_extend(image, Constant)
    for (int y = 0; y < bh; ++y) for (int x = 0; x < bw; ++x) dst[(size_t)y * bw + x] = ext_at(src, w, h, x0 + x, y0 + y, VWO_EDGE_CONSTANT);
    return 0;
  }
  float taps[1024];
  const int nt = gaussian_kernel<float>((double)width, 0, taps, 1024);
  if (nt < 0) return -1;
  const int c = nt ? (nt - 1) / 2 : 0, lo = nt ? nt - c - 1 : 0;
  // gaussian view evaluated at (x,y), any integer coordinates: filter of the constant-extended source
  auto gauss_at = [&](int x, int y) -> float {
    if (nt == 0) return ext_at(src, w, h, x, y, VWO_EDGE_CONSTANT);
    float result = 0.0f;
    for (int j = 0; j < nt; ++j) {
      float hrow = 0.0f;                                                  // the float `work` pixel of the H pass
      for (int i = 0; i < nt; ++i) hrow += taps[nt - 1 - i] * ext_at(src, w, h, x - lo + i, y - lo + j, VWO_EDGE_CONSTANT);
      result += taps[nt - 1 - j] * hrow;
    }
    return result;
  };
  if (mode == VWO_PREFILTER_MEANSUB) {                                     // edge_extend(image) - gaussian_filter(image)
    for (int y = 0; y < bh; ++y) for (int x = 0; x < bw; ++x)
      dst[(size_t)y * bw + x] = ext_at(src, w, h, x0 + x, y0 + y, VWO_EDGE_CONSTANT) - gauss_at(x0 + x, y0 + y);
    return 0;
  }
  // LoG: laplacian_filter(gaussian view) with ConstantEdgeExtension of the gaussian VIEW (its domain is the image)
  std::vector<float> g((size_t)w * h);
  for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) g[(size_t)y * w + x] = gauss_at(x, y);
  const float lap[9] = {0, 1, 0, 1, -4, 1, 0, 1, 0};
  for (int y = 0; y < bh; ++y) for (int x = 0; x < bw; ++x) {
    float result = 0.0f;
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i)
      result += lap[(2 - j) * 3 + (2 - i)] * ext_at(g.data(), w, h, x0 + x - 1 + i, y0 + y - 1 + j, VWO_EDGE_CONSTANT);
    dst[(size_t)y * bw + x] = result;
  }
  return 0;
}

}  // namespace

extern "C" {

int vwo_subdivide_regions(const int32_t* disp3, int w, int h, int kx, int ky, int32_t* zones, int cap) {
  std::vector<Zone> list;
  subdivide(disp3, w, h, Box(0, 0, w, h), list, kx, ky);
  int n = 0;
  for (auto const& z : list) {
    if (n < cap) {
      int32_t* o = zones + (size_t)n * 8;
      o[0] = z.region.minx; o[1] = z.region.miny; o[2] = z.region.maxx; o[3] = z.region.maxy;
      o[4] = z.range.minx; o[5] = z.range.miny; o[6] = z.range.maxx; o[7] = z.range.maxy;
    }
    ++n;
  }
  return n;
}

int vwo_prefilter_region(const float* src, int w, int h, int mode, float width, int x0, int y0, int bw, int bh, float* dst) {
  return prefilter_region(src, w, h, mode, width, x0, y0, bw, bh, dst);
}

int vwo_parabola_subpixel(const float* disp3f, int w, int h, const float* left, const float* right, int rw, int rh,
                          int prefilter_mode, float prefilter_width, int kx, int ky, float* out3f) {
  if (kx % 2 != 1 || ky % 2 != 1 || w <= 0 || h <= 0) return -1;
  const size_t n = (size_t)w * h;
  // prerasterize(bbox = whole image), ParabolaSubpixelView.cc:277-298
  std::vector<int32_t> idisp(n * 3);                                       // crop(m_disparity, bbox) as PixelMask<Vector2i>
  for (size_t i = 0; i < n; ++i) {
    idisp[3*i] = (int32_t)disp3f[3*i]; idisp[3*i+1] = (int32_t)disp3f[3*i+1];  // float -> int32 conversion truncates
    idisp[3*i+2] = disp3f[3*i+2] != 0.0f ? std::numeric_limits<int32_t>::max() : 0;
  }
  // get_disparity_range does NOT skip invalid pixels (src/vw/Stereo/DisparityMap.h:52-66)
  int mnx = idisp[0], mxx = idisp[0], mny = idisp[1], mxy = idisp[1];
  for (size_t i = 0; i < n; ++i) {
    mnx = std::min(mnx, idisp[3*i]); mxx = std::max(mxx, idisp[3*i]);
    mny = std::min(mny, idisp[3*i+1]); mxy = std::max(mxy, idisp[3*i+1]);
  }
  Box range(mnx, mny, mxx + 1, mxy + 1);                                  // entire_search_range, max += (1,1)
  range.expand(1);
  const int hx = kx / 2, hy = ky / 2;
  Box left_region(-hx, -hy, w + hx, h + hy);
  Box right_region(left_region.minx + range.minx, left_region.miny + range.miny,
                   left_region.maxx + range.minx + range.sizex(), left_region.maxy + range.miny + range.sizey());
  const int lrw = left_region.sizex(), lrh = left_region.sizey(), rrw = right_region.sizex(), rrh = right_region.sizey();
  std::vector<float> lras((size_t)lrw * lrh), rras((size_t)rrw * rrh);
  if (prefilter_region(left, w, h, prefilter_mode, prefilter_width, left_region.minx, left_region.miny, lrw, lrh, lras.data())) return -1;
  if (prefilter_region(right, rw, rh, prefilter_mode, prefilter_width, right_region.minx, right_region.miny, rrw, rrh, rras.data())) return -1;

  // evaluate(), ParabolaSubpixelView.cc:31-274
  std::vector<float> patch(n * 9, 0.0f);                                   // ImageView<Vector<float,9>> zero-initialised
  std::vector<Zone> big, zones;
  subdivide(idisp.data(), w, h, Box(0, 0, w, h), big, kx, ky);
  const double ratio = 1.0;
  for (auto const& z : big) {                                              // :76-101
    const double len1 = z.region.area(), len2 = z.range.area();
    if (len2 / len1 < ratio) { zones.push_back(z); continue; }
    for (int dx = z.region.minx; dx < z.region.maxx; ++dx)
      for (int dy = z.region.miny; dy < z.region.maxy; ++dy) {
        const int32_t* p = &idisp[((size_t)dy * w + dx) * 3];
        if (!p[2]) continue;
        zones.push_back(Zone{Box(dx, dy, dx + 1, dy + 1), Box(p[0], p[1], p[0] + 1, p[1] + 1)});
      }
  }
  std::vector<double> cost_applied, cost_metric;
  for (auto zone : zones) {                                                // :104-218
    zone.range.expand(1);
    const int zw = zone.region.width(), zh = zone.region.height();
    const int cw = zw + kx - 1, chh = zh + ky - 1;                          // left_zone = region, max += kernel-1
    cost_applied.resize((size_t)cw * chh);
    cost_metric.resize((size_t)zw * zh);
    for (int dx = 0; dx < zone.range.width(); ++dx)
      for (int dy = 0; dy < zone.range.height(); ++dy) {
        const int ax = dx + zone.range.minx, ay = dy + zone.range.miny;    // disparity_abs
        // crops of left_raster / right_raster; zone coordinates are relative to the rasters' origins
        const int rx0 = zone.region.minx + ax - range.minx, ry0 = zone.region.miny + ay - range.miny;
        for (int y = 0; y < chh; ++y)
          for (int x = 0; x < cw; ++x)
            cost_applied[(size_t)y * cw + x] =
                cost_abs(lras[(size_t)(zone.region.miny + y) * lrw + zone.region.minx + x], rras[(size_t)(ry0 + y) * rrw + rx0 + x]);
        box_sum(cost_applied.data(), cw, cw, chh, kx, ky, cost_metric.data());
        for (int j = 0; j < zh; ++j)
          for (int i = 0; i < zw; ++i) {
            const size_t pi = (size_t)(zone.region.miny + j) * w + zone.region.minx + i;
            const int ddx = ax - idisp[3*pi], ddy = ay - idisp[3*pi+1];
            if (ddx >= -1 && ddx <= 1 && ddy >= -1 && ddy <= 1)
              patch[pi * 9 + (ddy + 1) * 3 + (ddx + 1)] = (float)cost_metric[(size_t)j * zw + i];   // :187-205
          }
      }
  }
  // pinvA rows a..f, ParabolaSubpixelView.h:83-88 (static float array initialised from double literals)
  static const float A[6][9] = {
    { (float)(1.0/6), (float)(-1.0/3), (float)(1.0/6), (float)(1.0/6), (float)(-1.0/3), (float)(1.0/6), (float)(1.0/6), (float)(-1.0/3), (float)(1.0/6) },
    { (float)(1.0/6), (float)(1.0/6), (float)(1.0/6), (float)(-1.0/3), (float)(-1.0/3), (float)(-1.0/3), (float)(1.0/6), (float)(1.0/6), (float)(1.0/6) },
    { (float)(1.0/4), 0.0f, (float)(-1.0/4), 0.0f, 0.0f, 0.0f, (float)(-1.0/4), 0.0f, (float)(1.0/4) },
    { (float)(-1.0/6), 0.0f, (float)(1.0/6), (float)(-1.0/6), 0.0f, (float)(1.0/6), (float)(-1.0/6), 0.0f, (float)(1.0/6) },
    { (float)(-1.0/6), (float)(-1.0/6), (float)(-1.0/6), 0.0f, 0.0f, 0.0f, (float)(1.0/6), (float)(1.0/6), (float)(1.0/6) },
    { (float)(-1.0/9), (float)(2.0/9), (float)(-1.0/9), (float)(2.0/9), (float)(5.0/9), (float)(2.0/9), (float)(-1.0/9), (float)(2.0/9), (float)(-1.0/9) } };
  for (size_t i = 0; i < n; ++i) {                                         // :221-271
    float* o = out3f + 3 * i;
    if (!idisp[3*i+2]) { o[0] = o[1] = o[2] = 0.0f; continue; }
    const float* pc = &patch[i * 9];
    bool all_equal = true;
    for (int c = 1; c < 9; ++c) if (pc[c] != pc[c - 1]) { all_equal = false; break; }
    o[0] = (float)idisp[3*i]; o[1] = (float)idisp[3*i+1]; o[2] = 1.0f;
    if (all_equal) continue;
    float x[6];
    for (int r = 0; r < 6; ++r) {                                          // Matrix<float,6,9> * Vector<float,9>: sequential dot product
      float acc = 0.0f;
      for (int c = 0; c < 9; ++c) acc += A[r][c] * pc[c];
      x[r] = acc;
    }
    const float denom = 4 * x[0] * x[1] - (x[2] * x[2]);
    const float ox = (x[2] * x[4] - 2 * x[1] * x[3]) / denom;
    const float oy = (x[2] * x[3] - 2 * x[0] * x[4]) / denom;
    const float MAX_SUBPIXEL_SHIFT = 5.0;
    // norm_2 = sqrt(norm_2_sqr) in double, norm_2_sqr accumulates float products in double and returns them cast to
    // the vector's value type (src/vw/Math/Vector.h:1591-1604)
    double n2 = 0.0; n2 += ox * ox; n2 += oy * oy;
    if (std::sqrt((double)(float)n2) < MAX_SUBPIXEL_SHIFT) { o[0] = (float)idisp[3*i] + ox; o[1] = (float)idisp[3*i+1] + oy; }
  }
  return 0;
}

}  // extern "C"

// ---- pyramid block matching -------------------------------------------------------------------------------------

namespace {

const int32_t kValid = std::numeric_limits<int32_t>::max();

// RmOutliersUsingThreshFunc evaluated at any integer position of the edge-extended (Constant) disparity image
// (src/vw/Stereo/DisparityMap.h:357-385).
struct DispImg {
  const int32_t* d; int w, h;
  const int32_t* at(int x, int y) const {
    x = x < 0 ? 0 : (x >= w ? w - 1 : x);
    y = y < 0 ? 0 : (y >= h ? h - 1 : y);
    return d + ((size_t)y * w + x) * 3;
  }
};
inline void rm_outliers_at(DispImg const& im, int x, int y, int hh, int hv, double pthr, double rthr, int32_t out[3]) {
  const int32_t* c = im.at(x, y);
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2];
  if (!c[2]) return;
  int matched = 0, total = 0;
  for (int yk = -hv; yk <= hv; ++yk)
    for (int xk = -hh; xk <= hh; ++xk) {
      const int32_t* n = im.at(x + xk, y + yk);
      if (n[2] && std::fabs((double)(c[0] - n[0])) <= pthr && std::fabs((double)(c[1] - n[1])) <= pthr) matched++;
      total++;
    }
  if (((double)matched / (double)total) < rthr) { out[0] = out[1] = out[2] = 0; }   // invalid pixel (PixelMask())
}

void disparity_filter(std::vector<int32_t>& disp, int w, int h, int hh, int hv, double pthr, double rthr, bool cleanup) {
  DispImg im{disp.data(), w, h};
  if (!cleanup) {                                    // rm_outliers_using_thresh (:409-419)
    std::vector<int32_t> out((size_t)w * h * 3);
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) rm_outliers_at(im, x, y, hh, hv, pthr, rthr, &out[((size_t)y * w + x) * 3]);
    disp.swap(out);
    return;
  }
  // disparity_cleanup_using_thresh (:427-441): the outer functor (1,1,3.0,0.20) reads the INNER VIEW, also at
  // positions outside the image (the inner view is defined there through its edge-extended child).
  const int pw = w + 2, ph = h + 2;
  std::vector<int32_t> inner((size_t)pw * ph * 3);
  for (int y = -1; y <= h; ++y) for (int x = -1; x <= w; ++x)
    rm_outliers_at(im, x, y, hh, hv, pthr, rthr, &inner[((size_t)(y + 1) * pw + (x + 1)) * 3]);
  std::vector<int32_t> out((size_t)w * h * 3);
  for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
    const int32_t* c = &inner[((size_t)(y + 1) * pw + (x + 1)) * 3];
    int32_t* o = &out[((size_t)y * w + x) * 3];
    o[0] = c[0]; o[1] = c[1]; o[2] = c[2];
    if (!c[2]) continue;
    int matched = 0, total = 0;
    for (int yk = -1; yk <= 1; ++yk) for (int xk = -1; xk <= 1; ++xk) {
      const int32_t* n = &inner[((size_t)(y + 1 + yk) * pw + (x + 1 + xk)) * 3];
      if (n[2] && std::fabs((double)(c[0] - n[0])) <= 3.0 && std::fabs((double)(c[1] - n[1])) <= 3.0) matched++;
      total++;
    }
    if (((double)matched / (double)total) < 0.20) { o[0] = o[1] = o[2] = 0; }
  }
  disp.swap(out);
}

// DisparityMaskView::operator() (src/vw/Stereo/DisparityMap.h:132-155)
void disparity_mask(std::vector<int32_t>& disp, int w, int h, const uint8_t* m1, const uint8_t* m2, int m2w, int m2h) {
  for (int j = 0; j < h; ++j) for (int i = 0; i < w; ++i) {
    int32_t* p = &disp[((size_t)j * w + i) * 3];
    bool keep = m1[(size_t)j * w + i] != 0 && p[2] != 0;
    if (keep) {
      const int x = i + p[0], y = j + p[1];
      keep = !(x < 0 || x >= m2w || y < 0 || y >= m2h || m2[(size_t)y * m2w + x] == 0);
    }
    if (!keep) { p[0] = p[1] = p[2] = 0; }
  }
}

template <class T>
std::vector<T> crop_ext(const T* src, int w, int h, Box const& b, int edge) {
  std::vector<T> out((size_t)b.sizex() * b.sizey());
  for (int y = 0; y < b.sizey(); ++y) for (int x = 0; x < b.sizex(); ++x)
    out[(size_t)y * b.sizex() + x] = ext_at(src, w, h, b.minx + x, b.miny + y, edge);
  return out;
}

struct FImg { std::vector<float> d; int w = 0, h = 0; };
struct MImg { std::vector<uint8_t> d; int w = 0, h = 0; };

}  // namespace

extern "C" {

int vwo_disparity_filter(int32_t* disp3, int w, int h, int half_h, int half_v, double pixel_thr, double rej_thr, int cleanup) {
  std::vector<int32_t> d(disp3, disp3 + (size_t)w * h * 3);
  disparity_filter(d, w, h, half_h, half_v, pixel_thr, rej_thr, cleanup != 0);
  std::memcpy(disp3, d.data(), d.size() * sizeof(int32_t));
  return 0;
}

int vwo_disparity_mask(int32_t* disp3, int w, int h, const uint8_t* lmask, const uint8_t* rmask, int rmw, int rmh) {
  std::vector<int32_t> d(disp3, disp3 + (size_t)w * h * 3);
  disparity_mask(d, w, h, lmask, rmask, rmw, rmh);
  std::memcpy(disp3, d.data(), d.size() * sizeof(int32_t));
  return 0;
}

// disparity_blob_filter (src/vw/Stereo/CorrelationView.cc:242-271): BlobIndexThreaded over the whole image as ONE tile
// (src/vw/Image/BlobIndex.h:385-455; labelling :135-262 = 8-connected components of the VALID pixels), blobs larger than
// `area` pixels are dropped from the index (:445-453), the rest are erased by ErodeView (src/vw/Image/ErodeView.h:199-222:
// a covered pixel becomes the default, invalid, zero pixel).
int vwo_blob_sizes(const int32_t* disp3, int w, int h, uint32_t* sizes) {
  if (!disp3 || !sizes || w <= 0 || h <= 0) return -1;
  std::vector<int> label((size_t)w * h, -1);
  std::vector<int> stack;
  for (size_t i = 0; i < (size_t)w * h; ++i) sizes[i] = 0;
  for (int y0 = 0; y0 < h; ++y0)
    for (int x0 = 0; x0 < w; ++x0) {
      const size_t s0 = (size_t)y0 * w + x0;
      if (!disp3[s0 * 3 + 2] || label[s0] >= 0) continue;
      std::vector<size_t> members;
      stack.clear(); stack.push_back((int)s0); label[s0] = (int)s0;
      while (!stack.empty()) {
        const int c = stack.back(); stack.pop_back();
        members.push_back((size_t)c);
        const int cy = c / w, cx = c - cy * w;
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx) {
            const int nx = cx + dx, ny = cy + dy;
            if ((dx == 0 && dy == 0) || nx < 0 || ny < 0 || nx >= w || ny >= h) continue;
            const size_t n = (size_t)ny * w + nx;
            if (disp3[n * 3 + 2] && label[n] < 0) { label[n] = (int)s0; stack.push_back((int)n); }
          }
      }
      for (size_t m : members) sizes[m] = (uint32_t)members.size();
    }
  return 0;
}

int vwo_disparity_blob_filter(int32_t* disp3, int w, int h, int area) {
  if (!disp3 || w <= 0 || h <= 0) return -1;
  if (area < 1) return 0;
  std::vector<uint32_t> sizes((size_t)w * h);
  vwo_blob_sizes(disp3, w, h, sizes.data());
  for (size_t i = 0; i < (size_t)w * h; ++i)
    if (sizes[i] && sizes[i] <= (uint32_t)area) { disp3[i * 3] = 0; disp3[i * 3 + 1] = 0; disp3[i * 3 + 2] = 0; }
  return 0;
}

}  // extern "C" (reopened below)

static thread_local int g_blob_filter_area = 0;
static thread_local int g_sgm_algorithm = 1;
// lr_disp_diff of the NEXT pyramid call on this thread (CorrelationView.h:84: m_lr_disp_diff, m_region_ul)
static thread_local float* g_lr_diff = nullptr;
static thread_local int g_lr_cols = 0, g_lr_rows = 0, g_lr_ulx = 0, g_lr_uly = 0;

// algorithm 0 = VW_CORRELATION_BM, 1 = VW_CORRELATION_SGM, 2 = VW_CORRELATION_MGM, 3 = VW_CORRELATION_FINAL_MGM (MGM at level 0 only)
static int pyramid_impl(const float* left, int lw, int lh, const float* right, int rw, int rh,
                        const uint8_t* lmask_in, const uint8_t* rmask_in,
                        int prefilter_mode, float prefilter_width,
                        int sminx, int sminy, int smaxx, int smaxy, int kx, int ky, int cost_type,
                        int corr_timeout, double seconds_per_op, float consistency_threshold,
                        int filter_half_kernel, int max_pyramid_levels_arg,
                        int bx, int by, int bw, int bh, float* out3f,
                        int algorithm, int min_consistency_level, int sgm_subpixel_mode, int sgm_sbx, int sgm_sby,
                        size_t memory_limit_mb, int num_threads, int blob_filter_area) {
  if (kx % 2 != 1 || ky % 2 != 1 || bw <= 0 || bh <= 0) return -1;
  const bool use_sgm = algorithm != 0;
  if (use_sgm) prefilter_mode = VWO_PREFILTER_NONE;                      // CorrelationView.h:96-97
  float* lr_diff = g_lr_diff;
  if (lr_diff) {                                                         // the tile must fit the buffer (CorrelationView.cc:277-283)
    if (bx < g_lr_ulx || by < g_lr_uly || bx + bw > g_lr_ulx + g_lr_cols || by + bh > g_lr_uly + g_lr_rows) return -4;
  }
  std::vector<int32_t> prev_disparity, disparity_rl, prev_disparity_rl;
  int pdw = 0, pdh = 0, rlw_ = 0, rlh_ = 0, prlw = 0, prlh = 0;
  std::vector<float> subpixel_disparity;
  const Box search(sminx, sminy, smaxx, smaxy);
  const Box bbox = Box::xywh(bx, by, bw, bh);
  std::vector<uint8_t> lm_all, rm_all;
  if (!lmask_in) { lm_all.assign((size_t)lw * lh, 255); lmask_in = lm_all.data(); }
  if (!rmask_in) { rm_all.assign((size_t)rw * rh, 255); rmask_in = rm_all.data(); }

  // constructor: m_max_level_by_search (CorrelationView.h:99-105)
  const int largest_search = std::max(search.sizex(), search.sizey());
  int max_level_by_search = (int)(std::floor(std::log(float(largest_search)) / std::log(2.0f)) - 1);
  if (max_level_by_search > max_pyramid_levels_arg) max_level_by_search = max_pyramid_levels_arg;
  if (max_level_by_search < 0) max_level_by_search = 0;
  // 1.0) number of levels (CorrelationView.cc:301-310)
  const int smallest_bbox = std::min(bw, bh), largest_kernel = std::max(kx, ky);
  int L = (int)std::floor(std::log((double)smallest_bbox) / std::log(2.0f) - std::log((double)largest_kernel) / std::log(2.0f));
  if (max_level_by_search < L) L = max_level_by_search;
  if (L < 1) L = 0;
  const int hkx = kx / 2, hky = ky / 2;
  const int max_upscaling = 1 << L;
  for (size_t i = 0; i < (size_t)bw * bh * 3; ++i) out3f[i] = 0.0f;

  // 2.0) build_image_pyramids (:67-239)
  std::vector<FImg> lp(L + 1), rp(L + 1);
  std::vector<MImg> lmp(L + 1), rmp(L + 1);
  {
    Box lg = bbox; lg.minx -= hkx * max_upscaling; lg.maxx += hkx * max_upscaling; lg.miny -= hky * max_upscaling; lg.maxy += hky * max_upscaling;
    Box rg(lg.minx + search.minx, lg.miny + search.miny, lg.maxx + search.minx + search.sizex(), lg.maxy + search.miny + search.sizey());
    lp[0].w = lg.sizex(); lp[0].h = lg.sizey(); lp[0].d = crop_ext(left, lw, lh, lg, VWO_EDGE_CONSTANT);
    rp[0].w = rg.sizex(); rp[0].h = rg.sizey(); rp[0].d = crop_ext(right, rw, rh, rg, VWO_EDGE_CONSTANT);
    std::vector<uint8_t> lmx = crop_ext(lmask_in, lw, lh, lg, VWO_EDGE_CONSTANT), rmx = crop_ext(rmask_in, rw, rh, rg, VWO_EDGE_CONSTANT);
    // mean of the valid pixels of subsample(., 2)  (:137-149); MeanAccumulator sums in double (Math/Functors.h:469-487)
    auto fill_mean = [](FImg& im, std::vector<uint8_t> const& m) -> bool {
      double acc = 0.0, cnt = 0.0;
      for (int y = 0; y < im.h; y += 2) for (int x = 0; x < im.w; x += 2)
        if (m[(size_t)y * im.w + x]) { acc += im.d[(size_t)y * im.w + x]; cnt += 1.0; }
      if (!cnt) return false;
      const float mean = (float)(acc / cnt);
      for (size_t i = 0; i < im.d.size(); ++i) if (!m[i]) im.d[i] = mean;
      return true;
    };
    if (!fill_mean(lp[0], lmx) || !fill_mean(rp[0], rmx)) return 0;      // tile has no data: all invalid (:318-327)
    Box rmb(bbox.minx + search.minx, bbox.miny + search.miny, bbox.maxx + search.minx + search.sizex(), bbox.maxy + search.miny + search.sizey());
    lmp[0].w = bw; lmp[0].h = bh; lmp[0].d = crop_ext(lmask_in, lw, lh, bbox, VWO_EDGE_ZERO);
    rmp[0].w = rmb.sizex(); rmp[0].h = rmb.sizey(); rmp[0].d = crop_ext(rmask_in, rw, rh, rmb, VWO_EDGE_ZERO);
    const float k5[5] = {(float)(1.0 / 16.0), (float)(4.0 / 16.0), (float)(6.0 / 16.0), (float)(4.0 / 16.0), (float)(1.0 / 16.0)};
    for (int i = 1; i <= L; ++i) {
      auto down = [&](FImg const& a, FImg& o) {
        o.w = 1 + (a.w - 1) / 2; o.h = 1 + (a.h - 1) / 2; o.d.resize((size_t)o.w * o.h);
        sepconv<float>(a.d.data(), a.w, a.h, k5, 5, 2, k5, 5, 2, VWO_EDGE_CONSTANT, 2, o.d.data());
      };
      down(lp[i - 1], lp[i]); down(rp[i - 1], rp[i]);
      auto mdown = [&](MImg const& a, MImg& o) {
        o.w = 1 + (a.w - 1) / 2; o.h = 1 + (a.h - 1) / 2; o.d.resize((size_t)o.w * o.h);
        vwo_subsample_mask_by_two(a.d.data(), a.w, a.h, o.d.data());
      };
      mdown(lmp[i - 1], lmp[i]); mdown(rmp[i - 1], rmp[i]);
    }
    for (int i = 0; i <= L; ++i) {                                       // prefilter every level (:232-236)
      std::vector<float> t(lp[i].d.size());
      vwo_prefilter_image(lp[i].d.data(), lp[i].w, lp[i].h, prefilter_mode, prefilter_width, t.data()); lp[i].d.swap(t);
      t.assign(rp[i].d.size(), 0.0f);
      vwo_prefilter_image(rp[i].d.data(), rp[i].w, rp[i].h, prefilter_mode, prefilter_width, t.data()); rp[i].d.swap(t);
    }
  }

  // 3.0) level loop
  std::vector<int32_t> disparity;
  int dw = 0, dh = 0;
  std::vector<Zone> zones;
  zones.push_back(Zone{Box(0, 0, lmp[L].w, lmp[L].h), Box(0, 0, search.width() / max_upscaling + 1, search.height() / max_upscaling + 1)});
  double estim_elapsed = 0.0;
  for (int level = L; level >= 0; --level) {
    const bool on_last_level = (level == 0);
    int scaling = 1 << level;
    dw = lmp[level].w; dh = lmp[level].h;
    disparity.assign((size_t)dw * dh * 3, 0);                            // set_size: default (invalid) pixels
    const int rox = max_upscaling * hkx / scaling, roy = max_upscaling * hky / scaling;   // region_offset (:381)
    std::stable_sort(zones.begin(), zones.end(), [](Zone const& a, Zone const& b) {       // SearchParamLessThan (:606)
      return (double)a.region.width() * a.region.height() * a.range.width() * a.range.height() <
             (double)b.region.width() * b.region.height() * b.range.width() * b.range.height(); });
    FImg const& Lv = lp[level]; FImg const& Rv = rp[level];
    bool check_rl = false;
    std::vector<uint8_t> right_rl_mask, left_rl_mask;
    int rrm_w = 0, rrm_h = 0, lrm_w = 0, lrm_h = 0;
    if (use_sgm) {                                                       // SGM branch (:391-595)
      const int sx = search.width() / scaling, sy = search.height() / scaling;     // zone.disparity_range().size()
      Box lr(0 + rox - hkx, 0 + roy - hky, dw + rox + hkx, dh + roy + hky);        // zone.image_region() + offset, expand(half_kernel)
      Box rr(lr.minx, lr.miny, lr.maxx + sx, lr.maxy + sy);
      std::vector<float> lc = crop_ext(Lv.d.data(), Lv.w, Lv.h, lr, VWO_EDGE_CONSTANT);
      std::vector<float> rc = crop_ext(Rv.d.data(), Rv.w, Rv.h, rr, VWO_EDGE_CONSTANT);
      const bool have_prev = level < L;
      int ow = 0, oh = 0;
      std::vector<int32_t> d((size_t)lr.sizex() * lr.sizey() * 3);
      std::vector<float> sub(on_last_level ? d.size() : 0);
      const int use_mgm = (algorithm == 2 || (algorithm == 3 && level == 0)) ? 1 : 0;      // CorrelationView.cc:365-366
      int rc_ = vwo_calc_disparity_sgm_x(cost_type, use_mgm, lc.data(), lr.sizex(), lr.sizey(), rc.data(), rr.sizex(), rr.sizey(), sx, sy, kx,
                                       sgm_subpixel_mode, sgm_sbx, sgm_sby, memory_limit_mb, num_threads,
                                       lmp[level].d.data(), lmp[level].w, lmp[level].h, rmp[level].d.data(), rmp[level].w, rmp[level].h,
                                       have_prev ? prev_disparity.data() : nullptr, pdw, pdh, 0, 0, d.data(), on_last_level ? sub.data() : nullptr, &ow, &oh);
      if (rc_) return rc_;
      if (ow != dw || oh != dh) return -3;
      std::copy(d.begin(), d.begin() + (size_t)dw * dh * 3, disparity.begin());
      if (on_last_level) subpixel_disparity.assign(sub.begin(), sub.begin() + (size_t)dw * dh * 3);   // before the filters (:444-445)
      if (consistency_threshold >= 0.0f && level >= min_consistency_level) {
        check_rl = true;
        Box rrev = rr;
        Box lrev(lr.minx - sx, lr.miny - sy, lr.maxx - sx + 2 * sx, lr.maxy - sy + 2 * sy);
        Box rmb_(0, 0, rrev.sizex() - 2 * hkx, rrev.sizey() - 2 * hky);
        Box lmb_(0 - sx, 0 - sy, lrev.sizex() - 2 * hkx - sx, lrev.sizey() - 2 * hky - sy);
        right_rl_mask = crop_ext(rmp[level].d.data(), rmp[level].w, rmp[level].h, rmb_, VWO_EDGE_ZERO);
        left_rl_mask = crop_ext(lmp[level].d.data(), lmp[level].w, lmp[level].h, lmb_, VWO_EDGE_ZERO);
        rrm_w = rmb_.sizex(); rrm_h = rmb_.sizey(); lrm_w = lmb_.sizex(); lrm_h = lmb_.sizey();
        std::vector<float> a = crop_ext(Rv.d.data(), Rv.w, Rv.h, rrev, VWO_EDGE_CONSTANT);
        std::vector<float> b = crop_ext(Lv.d.data(), Lv.w, Lv.h, lrev, VWO_EDGE_CONSTANT);
        std::vector<int32_t> rl((size_t)rrev.sizex() * rrev.sizey() * 3);
        int row = 0, roh = 0;
        rc_ = vwo_calc_disparity_sgm_x(cost_type, use_mgm, a.data(), rrev.sizex(), rrev.sizey(), b.data(), lrev.sizex(), lrev.sizey(), sx, sy, kx,
                                     sgm_subpixel_mode, sgm_sbx, sgm_sby, memory_limit_mb, num_threads,
                                     right_rl_mask.data(), rrm_w, rrm_h, left_rl_mask.data(), lrm_w, lrm_h,
                                     have_prev && !prev_disparity_rl.empty() ? prev_disparity_rl.data() : nullptr, prlw, prlh,
                                     0, 0, rl.data(), nullptr, &row, &roh);
        if (rc_) return rc_;
        rl.resize((size_t)row * roh * 3);
        for (size_t i = 0; i < (size_t)row * roh; ++i) { rl[3*i] -= sx; rl[3*i+1] -= sy; }
        if (level == 0 && lr_diff)                                       // ul_corner_offset = zone.min (0,0) + bbox.min - region_ul
          vwo_cross_corr_consistency_check_diff(disparity.data(), dw, dh, rl.data(), row, roh, consistency_threshold,
                                                lr_diff, g_lr_cols, g_lr_rows, bx - g_lr_ulx, by - g_lr_uly);
        else
          vwo_cross_corr_consistency_check(disparity.data(), dw, dh, rl.data(), row, roh, consistency_threshold);
        for (size_t i = 0; i < (size_t)row * roh; ++i) { rl[3*i] += sx; rl[3*i+1] += sy; }
        disparity_rl.swap(rl); rlw_ = row; rlh_ = roh;
      }
    } else
    for (Zone const& zone : zones) {
      Box lr(zone.region.minx + rox - hkx, zone.region.miny + roy - hky, zone.region.maxx + rox + hkx, zone.region.maxy + roy + hky);
      Box rr(lr.minx + zone.range.minx, lr.miny + zone.range.miny, lr.maxx + zone.range.minx + zone.range.sizex(), lr.maxy + zone.range.miny + zone.range.sizey());
      const double next_elapsed = seconds_per_op * ((double)lr.width() * lr.height() * zone.range.width() * zone.range.height());
      if (corr_timeout > 0 && estim_elapsed + next_elapsed > corr_timeout) break;
      estim_elapsed += next_elapsed;
      const int zw = zone.region.sizex(), zh = zone.region.sizey(), sx = zone.range.sizex(), sy = zone.range.sizey();
      if (zw <= 0 || zh <= 0) continue;
      std::vector<float> lc = crop_ext(Lv.d.data(), Lv.w, Lv.h, lr, VWO_EDGE_CONSTANT);    // crops lie inside by construction
      std::vector<float> rc = crop_ext(Rv.d.data(), Rv.w, Rv.h, rr, VWO_EDGE_CONSTANT);
      std::vector<int32_t> zd((size_t)zw * zh * 3);
      if (best_of_search(cost_type, lc.data(), lr.sizex(), lr.sizey(), lr.sizex(), rc.data(), rr.sizex(), rr.sizey(), rr.sizex(),
                         kx, ky, sx, sy, zd.data())) return -1;
      if (consistency_threshold >= 0 && level == 0) {                    // R->L + check (:654-694)
        const double ne2 = seconds_per_op * ((double)rr.width() * rr.height() * zone.range.width() * zone.range.height());
        if (corr_timeout > 0 && estim_elapsed + ne2 > corr_timeout) break;
        estim_elapsed += ne2;
        Box l2(lr.minx - sx, lr.miny - sy, lr.maxx - sx, lr.maxy - sy);   // left_region - range.size()
        // right crop is the "left" image of this call, grown on the max side by s-1 by calc_disparity itself
        Box l2g(l2.minx, l2.miny, l2.minx + rr.sizex() + sx - 1, l2.miny + rr.sizey() + sy - 1);
        std::vector<float> a = crop_ext(Rv.d.data(), Rv.w, Rv.h, rr, VWO_EDGE_CONSTANT);
        std::vector<float> b = crop_ext(Lv.d.data(), Lv.w, Lv.h, l2g, VWO_EDGE_CONSTANT);
        const int rlw = rr.sizex() - kx + 1, rlh = rr.sizey() - ky + 1;
        std::vector<int32_t> rl((size_t)rlw * rlh * 3);
        if (best_of_search(cost_type, a.data(), rr.sizex(), rr.sizey(), rr.sizex(), b.data(), l2g.sizex(), l2g.sizey(), l2g.sizex(),
                           kx, ky, sx, sy, rl.data())) return -1;
        for (size_t i = 0; i < (size_t)rlw * rlh; ++i) { rl[3*i] -= sx; rl[3*i+1] -= sy; }   // - pixel_typeI(range.size())
        if (lr_diff)
          vwo_cross_corr_consistency_check_diff(zd.data(), zw, zh, rl.data(), rlw, rlh, consistency_threshold, lr_diff, g_lr_cols, g_lr_rows,
                                                zone.region.minx + bx - g_lr_ulx, zone.region.miny + by - g_lr_uly);
        else
          vwo_cross_corr_consistency_check(zd.data(), zw, zh, rl.data(), rlw, rlh, consistency_threshold);
      }
      for (int y = 0; y < zh; ++y) for (int x = 0; x < zw; ++x) {         // crop(disparity, region) = ...; += range.min()
        const int32_t* s3 = &zd[((size_t)y * zw + x) * 3];
        int32_t* d3 = &disparity[((size_t)(zone.region.miny + y) * dw + zone.region.minx + x) * 3];
        d3[0] = s3[0] + zone.range.minx; d3[1] = s3[1] + zone.range.miny; d3[2] = s3[2];
      }
    }
    // 3.2a) clean-up filters (:702-744)
    if (filter_half_kernel > 0) {
      disparity_filter(disparity, dw, dh, filter_half_kernel, filter_half_kernel, 3.0, 0.5, !on_last_level);
      disparity_mask(disparity, dw, dh, lmp[level].d.data(), rmp[level].d.data(), rmp[level].w, rmp[level].h);
      if (!on_last_level && check_rl && use_sgm) {                       // the R->L result seeds the next level's R->L run (:722-730)
        disparity_filter(disparity_rl, rlw_, rlh_, filter_half_kernel, filter_half_kernel, 3.0, 0.5, true);
        disparity_mask(disparity_rl, rlw_, rlh_, right_rl_mask.data(), left_rl_mask.data(), lrm_w, lrm_h);
      }
    }
    // the kernel based filtering tends to leave isolated blobs behind (:746-750)
    vwo_disparity_blob_filter(disparity.data(), dw, dh, blob_filter_area / (1 << level));
    if (check_rl && !on_last_level) vwo_disparity_blob_filter(disparity_rl.data(), rlw_, rlh_, blob_filter_area / (1 << level));
    if (use_sgm) {                                                       // prev_disparity = disparity at the top of the next level (:368-371)
      prev_disparity = disparity; pdw = dw; pdh = dh;
      if (check_rl) { prev_disparity_rl = disparity_rl; prlw = rlw_; prlh = rlh_; } else { prev_disparity_rl.clear(); }
    }
    // 3.2b) refine the search estimates (:754-799)
    if (!on_last_level && !use_sgm) {
      zones.clear();
      subdivide(disparity.data(), dw, dh, Box(0, 0, dw, dh), zones, kx, ky);
      scaling >>= 1;
      const Box scale_search(0, 0, rp[level - 1].w - lp[level - 1].w, rp[level - 1].h - lp[level - 1].h);
      const Box next_zone_size(0, 0, lmp[level - 1].w, lmp[level - 1].h);
      const Box default_range(0, 0, search.width(), search.height());
      for (Zone& z : zones) {
        // BBox *= 2 scales both corners; empty boxes are left alone by operator*= (BBox.h) — regions are never empty here
        z.region = Box(z.region.minx * 2, z.region.miny * 2, z.region.maxx * 2, z.region.maxy * 2);
        z.region.crop(next_zone_size);
        if (!z.range.empty()) z.range = Box(z.range.minx * 2, z.range.miny * 2, z.range.maxx * 2, z.range.maxy * 2);
        z.range.expand(2);
        z.range.crop(scale_search);
        if (z.range.empty()) z.range = default_range;
      }
    }
  }
  if (dw != bw || dh != bh) return -2;
  if (lr_diff)                                                           // filtered pixels lose their discrepancy too (:846-855)
    for (int r = 0; r < bh; ++r) for (int c = 0; c < bw; ++c)
      if (!disparity[((size_t)r * bw + c) * 3 + 2])
        lr_diff[((size_t)(r + by - g_lr_uly) * g_lr_cols + (c + bx - g_lr_ulx)) * 2 + 1] = 0.0f;
  if (use_sgm) {                                                         // (:862-875) sub-pixel view, filtered pixels invalidated
    for (size_t i = 0; i < (size_t)bw * bh; ++i) {
      out3f[3*i] = subpixel_disparity[3*i] + (float)search.minx;
      out3f[3*i+1] = subpixel_disparity[3*i+1] + (float)search.miny;
      out3f[3*i+2] = (disparity[3*i+2] && subpixel_disparity[3*i+2] != 0.0f) ? 1.0f : 0.0f;
    }
    return 0;
  }
  // 5.0) + search.min, cast to float (:876-885); invalid pixels keep valid = 0, child gets the offset as well
  for (size_t i = 0; i < (size_t)bw * bh; ++i) {
    out3f[3*i] = (float)(disparity[3*i] + search.minx);
    out3f[3*i+1] = (float)(disparity[3*i+1] + search.miny);
    out3f[3*i+2] = disparity[3*i+2] ? 1.0f : 0.0f;
  }
  return 0;
}


extern "C" {

int vwo_pyramid_correlate(const float* left, int lw, int lh, const float* right, int rw, int rh,
                          const uint8_t* lmask, const uint8_t* rmask, int prefilter_mode, float prefilter_width,
                          int sminx, int sminy, int smaxx, int smaxy, int kx, int ky, int cost_type,
                          int corr_timeout, double seconds_per_op, float consistency_threshold,
                          int filter_half_kernel, int max_pyramid_levels, int bx, int by, int bw, int bh, float* out3f) {
  return pyramid_impl(left, lw, lh, right, rw, rh, lmask, rmask, prefilter_mode, prefilter_width, sminx, sminy, smaxx, smaxy, kx, ky,
                      cost_type, corr_timeout, seconds_per_op, consistency_threshold, filter_half_kernel, max_pyramid_levels,
                      bx, by, bw, bh, out3f, 0, 0, 0, 0, 0, 0, 1, g_blob_filter_area);
}

int vwo_pyramid_correlate_sgm(const float* left, int lw, int lh, const float* right, int rw, int rh,
                              const uint8_t* lmask, const uint8_t* rmask,
                              int sminx, int sminy, int smaxx, int smaxy, int kernel, int cost_type,
                              float consistency_threshold, int min_consistency_level, int filter_half_kernel, int max_pyramid_levels,
                              int sgm_subpixel_mode, int sgm_sbx, int sgm_sby, size_t memory_limit_mb, int num_threads,
                              int bx, int by, int bw, int bh, float* out3f) {
  return pyramid_impl(left, lw, lh, right, rw, rh, lmask, rmask, 0, 0.0f, sminx, sminy, smaxx, smaxy, kernel, kernel,
                      cost_type, 0, 0.0, consistency_threshold, filter_half_kernel, max_pyramid_levels,
                      bx, by, bw, bh, out3f, g_sgm_algorithm, min_consistency_level, sgm_subpixel_mode, sgm_sbx, sgm_sby, memory_limit_mb, num_threads,
                      g_blob_filter_area);
}

// algorithm of the following vwo_pyramid_correlate_sgm calls of this thread: 1 = VW_CORRELATION_SGM (default), 2 = _MGM, 3 = _FINAL_MGM
void vwo_set_sgm_algorithm(int algorithm) { g_sgm_algorithm = algorithm >= 1 && algorithm <= 3 ? algorithm : 1; }

// blob_filter_area of the NEXT vwo_pyramid_correlate / _sgm call on this thread (keeps the long signatures stable)
void vwo_set_blob_filter_area(int area) { g_blob_filter_area = area; }
// lr_disp_diff (cols x rows x {value, valid} float, NULL = off) and region_ul of the following pyramid calls of this thread
void vwo_set_lr_disp_diff(float* buf, int cols, int rows, int ulx, int uly) { g_lr_diff = buf; g_lr_cols = cols; g_lr_rows = rows; g_lr_ulx = ulx; g_lr_uly = uly; }

}  // extern "C"
