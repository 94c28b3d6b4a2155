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
