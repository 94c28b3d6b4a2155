// vw_sgm_oracle.cc — CPU restatement of the reference's semi-global matching path.  TEST INFRASTRUCTURE ONLY: loaded by
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product.
//
// Follows, function by function (all paths relative to /root/reference):
//   u8_convert                          src/vw/Image/ImageThresh.h:275-286 (+ find_image_min_max Statistics.h:114-127,
//                                       ChannelNormalizeFunctor Algorithms.h:105-126)
//   get_census_value_* / ternary        src/vw/Image/CensusTransform.h:64-340
//   hamming_distance                    src/vw/Math/Functions.h:226-260
//   calc_disparity_sgm                  src/vw/Stereo/SGM.cc:167-229
//   SemiGlobalMatcher::set_parameters   SGM.cc:77-165 (p1 / p2 defaults)
//   semi_global_matching_func           SGM.cc:2387-2448 (output extent)
//   populate_adjacent_disp_lookup_table SGM.cc:755-799
//   populate_disp_bound_image           SGM.cc:241-499, constrain_disp_bound_image :502-672, calc_main_buf_size :677-731
//   compute_disparity_costs             SGM.cc:1740-1893 -> get_hamming_distance_costs :40-75
//   accum_sgm_multithread               SGM.cc:2462-2612 (8 directions; PixelPassTask SGMAssist.h:691-832,
//                                       PixelLineIterator src/vw/Image/PixelIterator.h:140-212)
//   evaluate_path (SSE semantics)       SGM.cc:936-984,1013-1150: saturating u16 adds / subtract (x86 builds define VW_ENABLE_SSE)
//   select_best_disparity               SGM.cc:1159-1284, create_disparity_view :1286-1408
//   create_disparity_view_subpixel      SGM.cc:1497-1614, compute_subpixel_offset :1445-1479, fits :1411-1436,
//                                       ParabolaFit2d::find_peak SGMAssist.h:99-135
//   accum_mgm_multithread (use_mgm)     SGM.cc:2619-2700: eight SmoothPathAccumTask passes (SGMAssist.h:835-1239), each pixel the
//                                       mean of TWO evaluate_path results (path predecessor + perpendicular predecessor);
//                                       MultiAccumRowBuffer :236-543 (line buffers, added to the sums line by line);
//                                       get_path_pixel_diff SGM.cc:2715-2721; small-buffer size for MGM :703-713
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "vw_oracle.h"

namespace {

bool g_allow_block_cost = false;         // vwo_set_sgm_allow_block_cost: the MAD block cost the reference keeps behind a throw (SGM.cc:1887-1892)
int g_host_threads = 1;                  // vwo_set_sgm_host_threads: run-time only, results do not depend on it

typedef uint8_t CostType;
typedef uint16_t AccumCostType;

enum { COST_CENSUS = 3, COST_TERNARY = 4 };   // CostFunctions.h:143-149
enum { SUB_NONE = 0, SUB_PARABOLA = 1, SUB_LINEAR = 2, SUB_POLY4 = 3, SUB_COSINE = 4, SUB_LC_BLEND = 5 };

struct U8Img {
  const uint8_t* p; int w, h;
  int operator()(int c, int r) const { return p[(size_t)r * w + c]; }
};

// ---- census (CensusTransform.h) ------------------------------------------------------------------------------------

const int k9cols[32] = {0, 4, 8, 1, 3, 5, 7, 2, 4, 6, 1, 4, 7, 0, 2, 3, 5, 6, 8, 1, 4, 7, 2, 4, 6, 1, 3, 5, 7, 0, 4, 8};
const int k9rows[32] = {0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4, 4, 4, 4, 5, 5, 5, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8};
const int k7cols[32] = {0, 2, 3, 4, 6, 1, 3, 5, 0, 2, 3, 4, 6, 0, 1, 2, 4, 5, 6, 0, 2, 3, 4, 6, 1, 3, 5, 0, 2, 3, 4, 6};
const int k7rows[32] = {0, 0, 0, 0, 0, 1, 1, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 5, 5, 5, 6, 6, 6, 6, 6};

uint64_t census_value(U8Img const& im, int col, int row, int k, bool ternary, int thr) {
  const int center = im(col, row);
  uint64_t out = 0, addend = 1;
  if (!ternary) {
    if (k == 3) {                                    // :64-77 (explicit weights; same order as the generic loop)
      const int wts[8] = {128, 64, 32, 16, 8, 4, 2, 1};
      int n = 0;
      for (int r = row - 1; r <= row + 1; ++r)
        for (int c = col - 1; c <= col + 1; ++c) {
          if (r == row && c == col) continue;
          if (im(c, r) > center) out += wts[n];
          ++n;
        }
      return out;
    }
    if (k == 5 || k == 7) {                          // :78-110
      const int h = k / 2;
      for (int r = row + h; r >= row - h; --r)
        for (int c = col + h; c >= col - h; --c) {
          if (r == row && c == col) continue;
          if (im(c, r) > center) out += addend;
          addend *= 2;
        }
      return out;
    }
    for (int i = 0; i < 32; ++i) {                   // 9x9 sparse pattern :112-160
      if (im(col + k9cols[i] - 4, row + k9rows[i] - 4) > center) out += addend;
      addend *= 2;
    }
    return out;
  }
  const int lo = center - thr, hi = center + thr;
  auto tern = [&](int val) {
    if (val >= lo) { out += addend; if (val > hi) out += addend * 2; }
    addend *= 4;
  };
  if (k == 3 || k == 5) {                            // :163-212
    const int h = k / 2;
    for (int r = row + h; r >= row - h; --r)
      for (int c = col + h; c >= col - h; --c) {
        if (r == row && c == col) continue;
        tern(im(c, r));
      }
    return out;
  }
  if (k == 7) { for (int i = 0; i < 32; ++i) tern(im(col + k7cols[i] - 3, row + k7rows[i] - 3)); return out; }   // :215-270
  for (int i = 0; i < 32; ++i) tern(im(col + k9cols[i] - 4, row + k9rows[i] - 4));                                // :272-337
  return out;
}

// storage width of the census word decides which hamming_distance overload runs (SGM.cc:1740-1871); all of them count
// the set bits of a ^ b, so only the value matters.
inline int hamming(uint64_t a, uint64_t b) { return __builtin_popcountll(a ^ b); }

void census_image(U8Img const& im, int k, bool ternary, int thr, std::vector<uint64_t>& out, int& ow, int& oh) {
  const int h = (k - 1) / 2;
  ow = im.w - 2 * h; oh = im.h - 2 * h;
  if (ow < 0) ow = 0;
  if (oh < 0) oh = 0;
  out.assign((size_t)ow * oh, 0);
  for (int r = 0; r < oh; ++r)
    for (int c = 0; c < ow; ++c) out[(size_t)r * ow + c] = census_value(im, c + h, r + h, k, ternary, thr);
}

// ---- the matcher ---------------------------------------------------------------------------------------------------

struct Bounds { int v[4]; };                       // min_x, min_y, max_x, max_y (inclusive)
inline bool is_zero_area(Bounds const& b) { return b.v[0] == 0 && b.v[1] == 0 && b.v[2] == -1 && b.v[3] == -1; }

struct IBox {                                      // vw::BBox2i as used by the bound logic (grow with points, crop)
  int x0 = 0x7ffffffe, y0 = 0x7ffffffe, x1 = -0x7ffffffe, y1 = -0x7ffffffe;
  bool empty() const { return x0 >= x1 || y0 >= y1; }
  void grow(int x, int y) { if (x > x1) x1 = x; if (x < x0) x0 = x; if (y > y1) y1 = y; if (y < y0) y0 = y; }   // BBox.tcc:82-92
  void crop(IBox const& b) { x0 = std::max(x0, b.x0); y0 = std::max(y0, b.y0); x1 = std::min(x1, b.x1); y1 = std::min(y1, b.y1); }
  void expand(int n) { if (empty()) return; x0 -= n; y0 -= n; x1 += n; y1 += n; }
};

struct Matcher {
  bool use_mgm = false;
  int cost_type, min_dx, min_dy, max_dx, max_dy, kernel, subpixel, sbx, sby, ternary_thr, num_threads;
  size_t memory_limit_mb;
  AccumCostType p1, p2;
  int num_dx, num_dy, num_disp;
  int min_row, max_row, min_col, max_col, ocols, orows;
  std::vector<Bounds> bounds;
  std::vector<size_t> starts;
  std::vector<int> adj;
  std::vector<CostType> cost;
  std::vector<AccumCostType> accum;
  size_t main_buf_size = 0;

  int ndisp(int c, int r) const { Bounds const& b = bounds[(size_t)r * ocols + c]; return (b.v[2] - b.v[0] + 1) * (b.v[3] - b.v[1] + 1); }
  int xy_to_disp(int dx, int dy) const { return (dy - min_dy) * num_dx + (dx - min_dx); }
  AccumCostType bad_val() const { return (AccumCostType)(255 + p2); }

  void set_parameters(uint16_t up1, uint16_t up2) {          // SGM.cc:77-165
    num_dx = max_dx - min_dx + 1; num_dy = max_dy - min_dy + 1; num_disp = num_dx * num_dy;
    if (up1 > 0) p1 = up1;
    else if (cost_type == COST_CENSUS) p1 = kernel == 3 ? 3 : kernel == 5 ? 15 : kernel == 7 ? 30 : kernel == 9 ? 20 : 3;
    else if (cost_type == COST_TERNARY) p1 = kernel == 3 ? 12 : kernel == 5 ? 30 : kernel == 7 ? 40 : kernel == 9 ? 40 : 30;
    else p1 = 3;
    if (up2 > 0) p2 = up2;
    else if (cost_type == COST_CENSUS) p2 = kernel == 3 ? 70 : kernel == 5 ? 750 : kernel == 7 ? 1500 : kernel == 9 ? 1000 : 22;
    else if (cost_type == COST_TERNARY) p2 = kernel == 3 ? 600 : kernel == 5 ? 1500 : kernel == 7 ? 2000 : kernel == 9 ? 2000 : 30;
    else p2 = 250;
  }

  void populate_adjacent() {                                  // SGM.cc:755-799
    adj.resize((size_t)num_disp * 8);
    int d = 0;
    for (int dy = min_dy; dy <= max_dy; ++dy) {
      int yl = dy - 1, ym = dy + 1;
      if (yl < min_dy) yl = dy;
      if (ym > max_dy) ym = dy;
      const int ylo = yl - min_dy, yo = dy - min_dy, ymo = ym - min_dy;
      for (int dx = min_dx; dx <= max_dx; ++dx) {
        int xl = dx - 1, xm = dx + 1;
        if (xl < min_dx) xl = dx;
        if (xm > max_dx) xm = dx;
        const int xlo = xl - min_dx, xo = dx - min_dx, xmo = xm - min_dx;
        int* t = &adj[(size_t)d * 8];
        t[0] = ylo * num_dx + xo;  t[1] = yo * num_dx + xlo;  t[2] = yo * num_dx + xmo;  t[3] = ymo * num_dx + xo;
        t[4] = ylo * num_dx + xlo; t[5] = ylo * num_dx + xmo; t[6] = ymo * num_dx + xlo; t[7] = ymo * num_dx + xmo;
        ++d;
      }
    }
  }

  // calc_main_buf_size (:677-731): fills starts; returns false when over the memory cap
  bool calc_main_buf_size() {
    starts.resize((size_t)ocols * orows);
    size_t n = 0;
    for (int r = 0; r < orows; ++r)
      for (int c = 0; c < ocols; ++c) { starts[(size_t)r * ocols + c] = n; n += ndisp(c, r); }
    if (n < 6) n = 6;
    main_buf_size = n;
    const int line_size = (int)(std::sqrt((double)(ocols * ocols + orows * orows)) + 1);      // OneLineBuffer::one_buf_size
    size_t one_buf = (size_t)line_size * num_disp;
    if (one_buf > main_buf_size) one_buf = main_buf_size;
    size_t small_buf = one_buf * num_threads;
    if (use_mgm) {                                             // :709-713: four vertical and four horizontal one-path line buffers
      size_t vert = (size_t)orows * num_disp, horiz = (size_t)ocols * num_disp;      // MultiAccumRowBuffer::multi_buf_size
      if (vert > main_buf_size) vert = main_buf_size;
      if (horiz > main_buf_size) horiz = main_buf_size;
      small_buf = 4 * vert + 4 * horiz;
    }
    const double MB = 1024.0 * 1024.0;
    const double total = (double)n * (3.0 / MB) + (double)small_buf * (2.0 / MB);
    return !(total > (double)memory_limit_mb);
  }

  bool constrain(std::vector<uint8_t> const& full_search, bool have_prev, double percent_masked, double area, int conserve) {   // :502-672
    const Bounds ZERO{{0, 0, -1, -1}};
    IBox max_range; max_range.x0 = min_dx; max_range.y0 = min_dy; max_range.x1 = max_dx; max_range.y1 = max_dy;
    int RANGE = 10; const int EXPANSION = 2;
    if (conserve == 1) RANGE = 25;
    if (conserve == 2) RANGE = 3;
    if (conserve == 3) RANGE = 0;
    if (have_prev) {
      for (int r = 0; r < orows; ++r) {
        const int r0 = std::max(r - RANGE, 0), r1 = std::min(r + RANGE, orows - 1);
        for (int c = 0; c < ocols; ++c) {
          if (!full_search[(size_t)r * ocols + c]) continue;
          const int c0 = std::max(c - RANGE, 0), c1 = std::min(c + RANGE, ocols - 1);
          IBox nr;
          for (int rs = r0; rs <= r1; ++rs)
            for (int cs = c0; cs <= c1; ++cs) {
              if (full_search[(size_t)rs * ocols + cs]) continue;
              Bounds const& v = bounds[(size_t)rs * ocols + cs];
              if (is_zero_area(v)) continue;
              nr.grow(v.v[0], v.v[1]);
              nr.grow(v.v[2], v.v[3]);
            }
          if (nr.empty()) {
            if (conserve > 0) bounds[(size_t)r * ocols + c] = ZERO;
            continue;
          }
          nr.expand(EXPANSION);
          nr.crop(max_range);
          bounds[(size_t)r * ocols + c] = Bounds{{nr.x0, nr.y0, nr.x1, nr.y1}};
        }
      }
    }
    const double num_pixels = (double)orows * ocols;
    area /= num_pixels;
    if (area <= 0 || percent_masked >= 100) return false;
    return true;
  }

  // populate_disp_bound_image (:241-499)
  bool populate_bounds(const uint8_t* lmask, int lmw, int lmh, const uint8_t* rmask, int rmw, int rmh,
                       const int32_t* prev, int pw, int ph, int* err) {
    const Bounds ZERO{{0, 0, -1, -1}};
    if (lmask && !(lmw == ocols && lmh == orows)) { *err = -5; return false; }        // LogicErr: left mask size
    if (rmask && !(rmw >= ocols + num_dx - 1 && rmh >= orows + num_dy - 1)) { *err = -5; return false; }
    const int SCALE_UP = 2;
    const bool check_x_edge = num_dx >= 10, check_y_edge = num_dy >= 10;
    double area = 0, percent_trusted = 0, percent_masked = 0;
    std::vector<uint8_t> full_search((size_t)ocols * orows, 0);
    int min_valid_right_row = 0, max_valid_right_row = 0;
    if (rmask) {
      min_valid_right_row = rmh - 1;
      for (int c = 0; c < ocols; ++c) {
        for (int i = rmh - 1; i > 0; --i)
          if (rmask[(size_t)i * rmw + c] > 0) { if (i > max_valid_right_row) max_valid_right_row = i; break; }
        for (int i = 0; i < rmh; ++i)
          if (rmask[(size_t)i * rmw + c] > 0) { if (i < min_valid_right_row) min_valid_right_row = i; break; }
      }
    }
    int dx_scaled = 0, dy_scaled = 0;
    for (int r = 0; r < orows; ++r) {
      const int r_in = r / SCALE_UP;
      int min_valid_right_column = -1, max_valid_right_column = -2;
      if (rmask) {
        for (int i = rmw - 1; i > 0; --i) if (rmask[(size_t)r * rmw + i] > 0) { max_valid_right_column = i; break; }
        if (max_valid_right_column > 0)
          for (int i = 0; i < rmw; ++i) if (rmask[(size_t)r * rmw + i] > 0) { min_valid_right_column = i; break; }
      }
      for (int c = 0; c < ocols; ++c) {
        const size_t idx = (size_t)r * ocols + c;
        if (lmask && lmask[idx] == 0) { bounds[idx] = ZERO; ++percent_masked; continue; }
        bool good = false;
        const int c_in = c / SCALE_UP;
        if (prev) {
          if (!(c_in >= pw || r_in >= ph)) {
            const int32_t* d = prev + ((size_t)r_in * pw + c_in) * 3;
            dx_scaled = d[0] * SCALE_UP; dy_scaled = d[1] * SCALE_UP;
            const bool on_edge = (check_x_edge && (dx_scaled <= min_dx || dx_scaled >= max_dx)) ||
                                 (check_y_edge && (dy_scaled <= min_dy || dy_scaled >= max_dy));
            good = d[2] != 0 && !on_edge;
          }
        }
        Bounds b;
        if (good) {
          b.v[0] = dx_scaled - sbx; b.v[2] = dx_scaled + sbx; b.v[1] = dy_scaled - sby; b.v[3] = dy_scaled + sby;
          if (b.v[0] < min_dx) b.v[0] = min_dx;
          if (b.v[1] < min_dy) b.v[1] = min_dy;
          if (b.v[2] > max_dx) b.v[2] = max_dx;
          if (b.v[3] > max_dy) b.v[3] = max_dy;
          percent_trusted += 1.0;
        } else {
          b = Bounds{{min_dx, min_dy, max_dx, max_dy}};
          full_search[idx] = 255;
        }
        if (rmask) {
          // BBox2i valid_region(min corner, max corner) - (c, r), cropped by the pixel's bounds; corners used inclusively
          int vx0 = min_valid_right_column, vy0 = min_valid_right_row, vx1 = max_valid_right_column, vy1 = max_valid_right_row;
          // operator-= on an EMPTY box leaves it untouched (BBox.tcc:268-290)
          if (!(vx0 >= vx1 || vy0 >= vy1)) { vx0 -= c; vx1 -= c; vy0 -= r; vy1 -= r; }
          vx0 = std::max(vx0, b.v[0]); vy0 = std::max(vy0, b.v[1]); vx1 = std::min(vx1, b.v[2]); vy1 = std::min(vy1, b.v[3]);
          if (vx0 > vx1 || vy0 > vy1) { bounds[idx] = ZERO; ++percent_masked; full_search[idx] = 0; continue; }
          b = Bounds{{vx0, vy0, vx1, vy1}};
        }
        bounds[idx] = b;
        area += (double)((b.v[3] - b.v[1] + 1) * (b.v[2] - b.v[0] + 1));
      }
    }
    const double num_pixels = (double)orows * ocols;
    percent_masked /= num_pixels; percent_trusted /= num_pixels;
    bool result = false;
    for (int level = 0; level <= 3; ++level) {
      constrain(full_search, prev != nullptr, percent_masked, area, level);
      if (calc_main_buf_size()) { result = main_buf_size != 0; break; }
      result = false;
    }
    return result;
  }

  // get_cost_block, "Mean of abs differences" branch (SGM.cc:1651-1709): int sum of |L - R| over the kernel, integer division by
  // the pixel count, saturated to 255; the single-pixel kernel returns the plain difference.
  CostType cost_block(U8Img const& L, U8Img const& R, int lx, int ly, int rx, int ry) const {
    if (kernel == 1) return (CostType)std::abs(L(lx, ly) - R(rx, ry));
    const int hk = (kernel - 1) / 2, count = kernel * kernel;
    int sum = 0;
    for (int j = -hk; j <= hk; ++j)
      for (int i = -hk; i <= hk; ++i) sum += std::abs(L(lx + i, ly + j) - R(rx + i, ry + j));
    sum /= count;
    return (CostType)(sum > 255 ? 255 : sum);
  }
  // fill_costs_block (SGM.cc:1711-1738): unreachable upstream (compute_disparity_costs throws first, :1887-1892); run here only
  // after vwo_set_sgm_allow_block_cost(1)
  void compute_costs_block(U8Img const& L, U8Img const& R) {
    for_lines(max_row - min_row + 1, [&](int ri, int) {
      const int r = min_row + ri;
      for (int c = min_col; c <= max_col; ++c) {
        Bounds const& b = bounds[(size_t)(r - min_row) * ocols + (c - min_col)];
        size_t ci = starts[(size_t)(r - min_row) * ocols + (c - min_col)];
        for (int dy = b.v[1]; dy <= b.v[3]; ++dy)
          for (int dx = b.v[0]; dx <= b.v[2]; ++dx) cost[ci++] = cost_block(L, R, c, r, c + dx, r + dy);
      }
    });
  }

  void compute_costs(U8Img const& L, U8Img const& R) {         // :1740-1893, :40-75
    if (cost_type != COST_CENSUS && cost_type != COST_TERNARY) { compute_costs_block(L, R); return; }
    std::vector<uint64_t> lc, rc; int lw, lh, rw, rh;
    const bool tern = cost_type == COST_TERNARY;
    census_image(L, kernel, tern, ternary_thr, lc, lw, lh);
    census_image(R, kernel, tern, ternary_thr, rc, rw, rh);
    const int hk = (kernel - 1) / 2;
    for_lines(max_row - min_row + 1, [&](int ri, int) {
      const int r = min_row + ri, br = r - hk;
      for (int c = min_col; c <= max_col; ++c) {
        const int bc = c - hk;
        Bounds const& b = bounds[(size_t)(r - min_row) * ocols + (c - min_col)];
        size_t ci = starts[(size_t)(r - min_row) * ocols + (c - min_col)];
        for (int dy = b.v[1]; dy <= b.v[3]; ++dy)
          for (int dx = b.v[0]; dx <= b.v[2]; ++dx)
            cost[ci++] = (CostType)hamming(lc[(size_t)br * lw + bc], rc[(size_t)(br + dy) * rw + bc + dx]);
      }
    });
  }

  static AccumCostType adds(AccumCostType a, AccumCostType b) { unsigned s = (unsigned)a + b; return s > 65535u ? 65535 : (AccumCostType)s; }
  static AccumCostType subs(AccumCostType a, AccumCostType b) { return a > b ? (AccumCostType)(a - b) : 0; }

  // evaluate_path, SSE flavour (:1013-1150 + :936-984)
  void evaluate_path(int col, int row, int col_p, int row_p, const AccumCostType* prior, AccumCostType* full_prior,
                     const CostType* local, AccumCostType* output, int gradient) const {
    AccumCostType p2_mod = p2;
    if (gradient > 0) p2_mod /= gradient;
    if (p2_mod < p1) p2_mod = p1;
    Bounds const& b = bounds[(size_t)row * ocols + col];
    Bounds const& bp = bounds[(size_t)row_p * ocols + col_p];
    const AccumCostType BAD = bad_val();
    AccumCostType min_prior = BAD;
    int d = 0;
    for (int dy = bp.v[1]; dy <= bp.v[3]; ++dy) {
      int fi = xy_to_disp(bp.v[0], dy);
      for (int dx = bp.v[0]; dx <= bp.v[2]; ++dx) {
        if (prior[d] < min_prior) min_prior = prior[d];
        full_prior[fi++] = prior[d++];
      }
    }
    const AccumCostType dJ = (AccumCostType)(min_prior + p2_mod);
    int packed = 0;
    for (int dy = b.v[1]; dy <= b.v[3]; ++dy) {
      int fd = xy_to_disp(b.v[0], dy);
      for (int dx = b.v[0]; dx <= b.v[2]; ++dx) {
        const int* t = &adj[(size_t)fd * 8];
        AccumCostType m = full_prior[t[0]];
        for (int i = 1; i < 8; ++i) m = std::min(m, full_prior[t[i]]);
        AccumCostType res = adds(m, p1);
        res = std::min(res, std::min(full_prior[fd], dJ));
        res = adds(res, local[packed]);
        res = subs(res, min_prior);
        output[packed] = res;
        ++packed; ++fd;
      }
    }
    for (int dy = bp.v[1]; dy <= bp.v[3]; ++dy) {
      int fi = xy_to_disp(bp.v[0], dy);
      for (int dx = bp.v[0]; dx <= bp.v[2]; ++dx) full_prior[fi++] = BAD;
    }
  }

  // one PixelPassTask (SGMAssist.h:705-819): a line from (c0, r0) in direction (dc, dr)
  void pass_line(U8Img const& L, int c0, int r0, int dc, int dr, std::vector<AccumCostType>& line, std::vector<AccumCostType>& full_prior) {
    size_t need = 0;
    for (int c = c0, r = r0; c >= 0 && r >= 0 && c < ocols && r < orows; c += dc, r += dr) need += ndisp(c, r);
    line.assign(need, 0);
    std::fill(full_prior.begin(), full_prior.end(), bad_val());
    int last_val = -1, cp = -1, rp = -1;
    AccumCostType* out = line.data();
    const AccumCostType* prior = nullptr;
    for (int c = c0, r = r0; c >= 0 && r >= 0 && c < ocols && r < orows; c += dc, r += dr) {
      const int nd = ndisp(c, r);
      const CostType* local = cost.data() + starts[(size_t)r * ocols + c];
      const int cur = L(c + min_col, r + min_row);
      const int diff = std::abs(cur - last_val);
      if (last_val >= 0) evaluate_path(c, r, cp, rp, prior, full_prior.data(), local, out, diff);
      else for (int d = 0; d < nd; ++d) out[d] = local[d];
      prior = out; out += nd; last_val = cur; cp = c; rp = r;
    }
    const AccumCostType* src = line.data();
    for (int c = c0, r = r0; c >= 0 && r >= 0 && c < ocols && r < orows; c += dc, r += dr) {
      const int nd = ndisp(c, r);
      AccumCostType* dst = accum.data() + starts[(size_t)r * ocols + c];
      for (int i = 0; i < nd; ++i) dst[i] = (AccumCostType)(dst[i] + src[i]);
      src += nd;
    }
  }

  // The lines of ONE direction never share a pixel, so they may run on any number of host threads with identical results (the
  // reference does the same with its PixelPassTask pool, SGM.cc:2462-2612); g_host_threads only shortens the oracle's run time.
  template <class F> static void for_lines(int n, F&& body) {
    const int T = std::max(1, std::min(g_host_threads, n));
    if (T == 1) { for (int i = 0; i < n; ++i) body(i, 0); return; }
    std::atomic<int> next(0);
    std::vector<std::thread> pool;
    for (int t = 0; t < T; ++t) pool.emplace_back([&, t]() { for (int i = next.fetch_add(16); i < n; i = next.fetch_add(16)) for (int j = i; j < std::min(n, i + 16); ++j) body(j, t); });
    for (auto& th : pool) th.join();
  }
  void accumulate(U8Img const& L) {                            // :2462-2612
    const int W = ocols, H = orows, T = std::max(1, g_host_threads);
    std::vector<std::vector<AccumCostType>> line(T), full_prior(T, std::vector<AccumCostType>(num_disp));
#define VWO_LINES(n, c0, r0, dc, dr) for_lines((n), [&](int i, int t) { (void)i; pass_line(L, (c0), (r0), (dc), (dr), line[t], full_prior[t]); })
    VWO_LINES(W, i, 0, 0, 1);                 // B
    VWO_LINES(W, i, H - 1, 0, -1);            // T
    VWO_LINES(H, 0, i, 1, 0);                 // R
    VWO_LINES(H, W - 1, i, -1, 0);            // L
    VWO_LINES(W, i, 0, 1, 1);                 // BR
    VWO_LINES(H - 1, 0, i + 1, 1, 1);
    VWO_LINES(W, i, 0, -1, 1);                // BL
    VWO_LINES(H - 1, W - 1, i + 1, -1, 1);
    VWO_LINES(W, i, H - 1, 1, -1);            // TR
    VWO_LINES(H - 1, 0, i, 1, -1);
    VWO_LINES(W, i, H - 1, -1, -1);           // TL
    VWO_LINES(H - 1, W - 1, i, -1, -1);
#undef VWO_LINES
  }


  // accum_mgm_multithread (:2619-2700).  One SmoothPathAccumTask per direction (SGMAssist.h:911-1236): pixels are visited in
  // raster order of the task's trip; a pixel with both predecessors inside (the task's own border test) gets
  // (evaluate_path(path predecessor) + evaluate_path(perpendicular predecessor)) / 2, every other pixel its local costs.
  // Both evaluations use ONE intensity difference, get_path_pixel_diff(col, row, dir) = |I(col,row) - I(col-dir_x, row-dir_y)|
  // (:2715-2721) — with the task's dir pointing AT the path predecessor that is the pixel on the far side, not the predecessor.
  // The line buffers of MultiAccumRowBuffer only bound the memory: a direction's values are kept here in a full volume `vol`
  // and added to the sums (u16 wrap-around) once per pixel, which is what add_lead_buffer_to_accum does line by line.
  // Pixels without disparities are skipped (get_num_disp == 0 -> continue) but can still serve as (empty) predecessors.
  struct MgmDir { int ax, ay, bx, by; int need_c_lo, need_c_hi, need_r_lo, need_r_hi; int order; };
  void mgm_pass(U8Img const& L, MgmDir const& D, std::vector<AccumCostType>& vol, std::vector<AccumCostType>& full_prior,
                std::vector<AccumCostType>& tmp) {
    const int W = ocols, H = orows, lastc = W - 1, lastr = H - 1;
    auto visit = [&](int col, int row) {
      const int nd = ndisp(col, row);
      if (nd == 0) return;
      const size_t st = starts[(size_t)row * W + col];
      const CostType* local = cost.data() + st;
      AccumCostType* out = vol.data() + st;
      const bool ok = (!D.need_c_lo || col > 0) && (!D.need_c_hi || col < lastc) && (!D.need_r_lo || row > 0) && (!D.need_r_hi || row < lastr);
      if (ok) {
        // the reference indexes the image unchecked; with min_disp = 0 (calc_disparity_sgm) and kernel >= 3 the far-side pixel
        // is always inside the image.  Clamped here so that other callers stay defined.
        const int fc = std::min(std::max(col - D.ax + min_col, 0), L.w - 1), fr = std::min(std::max(row - D.ay + min_row, 0), L.h - 1);
        const int diff = std::abs((int)L(col + min_col, row + min_row) - (int)L(fc, fr));
        const int ca = col + D.ax, ra = row + D.ay, cb = col + D.bx, rb = row + D.by;
        evaluate_path(col, row, ca, ra, vol.data() + starts[(size_t)ra * W + ca], full_prior.data(), local, out, diff);
        evaluate_path(col, row, cb, rb, vol.data() + starts[(size_t)rb * W + cb], full_prior.data(), local, tmp.data(), diff);
        for (int d = 0; d < nd; ++d) out[d] = (AccumCostType)(((int)out[d] + (int)tmp[d]) / 2);
      } else {
        for (int d = 0; d < nd; ++d) out[d] = local[d];
      }
      AccumCostType* dst = accum.data() + st;
      for (int d = 0; d < nd; ++d) dst[d] = (AccumCostType)(dst[d] + out[d]);
    };
    switch (D.order) {
      case 0: for (int r = 0; r < H; ++r) for (int c = 0; c < W; ++c) visit(c, r); break;             // rows down, columns right
      case 1: for (int r = lastr; r >= 0; --r) for (int c = lastc; c >= 0; --c) visit(c, r); break;   // rows up, columns left
      case 2: for (int c = 0; c < W; ++c) for (int r = lastr; r >= 0; --r) visit(c, r); break;        // columns right, rows up
      default: for (int c = lastc; c >= 0; --c) for (int r = 0; r < H; ++r) visit(c, r); break;       // columns left, rows down
    }
  }
  void accumulate_mgm(U8Img const& L) {
    std::vector<AccumCostType> vol(main_buf_size), full_prior(num_disp, bad_val()), tmp(num_disp);
    //                         path pred  perp pred   col>0 col<last row>0 row<last  trip
    const MgmDir dirs[8] = {{-1,  0,  0, -1,   1, 0, 1, 0,  0},      // L   (SGMAssist.h:911-955)
                            { 1,  0,  0,  1,   0, 1, 0, 1,  1},      // R   (:998-1033)
                            {-1, -1,  1, -1,   1, 1, 1, 0,  0},      // TL  (:958-996)
                            { 1,  1, -1,  1,   1, 1, 0, 1,  1},      // BR  (:1035-1071)
                            { 0, -1,  1,  0,   0, 1, 1, 0,  3},      // T   (:1147-1182)
                            { 0,  1, -1,  0,   1, 0, 0, 1,  2},      // B   (:1073-1108)
                            { 1, -1,  1,  1,   0, 1, 1, 1,  3},      // TR  (:1184-1219)
                            {-1,  1, -1, -1,   1, 0, 1, 1,  2}};     // BL  (:1110-1145)
    for (const MgmDir& d : dirs) mgm_pass(L, d, vol, full_prior, tmp);
  }

  // select_best_disparity (:1159-1284); smooths accum_vec in place when several minima tie
  void select_best(AccumCostType* accum_vec, Bounds const& b, int& min_index, std::vector<AccumCostType>& buffer) const {
    const int height = b.v[3] - b.v[1] + 1, width = b.v[2] - b.v[0] + 1, n = height * width;
    buffer.resize(n);
    int index = 0, min_count = 0;
    min_index = 0;
    AccumCostType min_val = 65535, value;
    for (index = 0; index < n; ++index) {
      value = accum_vec[index];
      buffer[index] = value;
      if (value == min_val) ++min_count;
      if (value < min_val) { min_index = index; min_val = value; min_count = 1; }
    }
    AccumCostType* in = accum_vec; AccumCostType* out = buffer.data();
    const double filter[3] = {1.0 / 3.0, 1.0 / 3.0, 1.0 / 3.0};
    int iter = 0;
    while (min_count > 1) {
      std::swap(in, out);
      index = 0; min_count = 0; min_val = 65535; min_index = 0;
      for (int row = 0; row < height; ++row)
        for (int col = 0; col < width; ++col) {
          int mn = -1, mx = 1;
          double result = 0, wtot = 0;
          if (iter < 5) {
            if (mn + col < 0) mn = 0;
            if (mx + col >= width) mx = 0;
            for (int k = mn; k <= mx; ++k) { result += (double)in[index + k] * filter[k + 1]; wtot += filter[k + 1]; }
          } else {
            if (mn + row < 0) mn = 0;
            if (mx + row >= height) mx = 0;
            for (int k = mn; k <= mx; ++k) { result += (double)in[index + k * width] * filter[k + 1]; wtot += filter[k + 1]; }
          }
          value = (AccumCostType)std::round(result / wtot);
          if (value == min_val) ++min_count;
          if (value < min_val) { min_index = index; min_val = value; min_count = 1; }
          out[index] = value;
          ++index;
        }
      ++iter;
      if (iter >= 6) break;
    }
    if (iter > 0 && iter % 2 == 0)
      for (int i = 0; i < index; ++i) in[i] = out[i];
  }

  void create_disparity_view(int32_t* disp) {                  // :1286-1408
    std::vector<AccumCostType> buffer;
    for (int j = 0; j < orows; ++j)
      for (int i = 0; i < ocols; ++i) {
        int32_t* o = disp + ((size_t)j * ocols + i) * 3;
        if (ndisp(i, j) == 0) { o[0] = o[1] = o[2] = 0; continue; }
        Bounds const& b = bounds[(size_t)j * ocols + i];
        int min_index = 0;
        select_best(accum.data() + starts[(size_t)j * ocols + i], b, min_index, buffer);
        const int dw = b.v[2] - b.v[0] + 1;
        int dy = min_index / dw;
        const int dx = min_index - dy * dw + b.v[0];
        dy += b.v[1];
        o[0] = dx; o[1] = dy; o[2] = 0x7fffffff;
      }
  }

  static double linearFit(double x) { return x / 2.0; }
  static double poly4Fit(double x) { return (x * x * x * x + x) / 4.0; }
  static double cosFit(double x) { const double PI = 3.14159265359; return (1 - std::cos(x * PI / 3.0)); }
  static double lcBlendFit(double x) {
    const double PI = 3.14159265359;
    const double factor = 1.195 - std::cos(x * (PI / 2.3));
    return cosFit(x) * factor + linearFit(x) * (1.0 - factor);
  }
  double subpixel_offset(AccumCostType prev, AccumCostType center, AccumCostType next, bool lb, bool rb) const {   // :1445-1479
    const double ld = (int)prev - (int)center, rd = (int)next - (int)center;
    if (rd == 0 && ld == 0) return 0;
    if (lb) return 0.5 * ((double)center / (double)next);
    if (rb) return -1.0 * (0.5 * ((double)center / (double)prev));
    double x = rd / ld, mult = -1.0;
    if (ld < rd) { x = ld / rd; mult = 1.0; }
    double value;
    switch (subpixel) {
      case SUB_POLY4: value = poly4Fit(x); break;
      case SUB_COSINE: value = cosFit(x); break;
      case SUB_LC_BLEND: value = lcBlendFit(x); break;
      default: value = linearFit(x); break;
    }
    return (value - 0.5) * mult;
  }
  static bool parabola_peak(const double z[9], double& dx, double& dy) {      // SGMAssist.h:99-135
    static const double pinvA[54] = {
      1.0 / 6, -1.0 / 3, 1.0 / 6, 1.0 / 6, -1.0 / 3, 1.0 / 6, 1.0 / 6, -1.0 / 3, 1.0 / 6,
      1.0 / 6, 1.0 / 6, 1.0 / 6, -1.0 / 3, -1.0 / 3, -1.0 / 3, 1.0 / 6, 1.0 / 6, 1.0 / 6,
      1.0 / 4, 0.0, -1.0 / 4, 0.0, 0.0, 0.0, -1.0 / 4, 0.0, 1.0 / 4,
      -1.0 / 6, 0.0, 1.0 / 6, -1.0 / 6, 0.0, 1.0 / 6, -1.0 / 6, 0.0, 1.0 / 6,
      -1.0 / 6, -1.0 / 6, -1.0 / 6, 0.0, 0.0, 0.0, 1.0 / 6, 1.0 / 6, 1.0 / 6,
      -1.0 / 9, 2.0 / 9, -1.0 / 9, 2.0 / 9, 5.0 / 9, 2.0 / 9, -1.0 / 9, 2.0 / 9, -1.0 / 9};
    double vals[6];
    for (int i = 0; i < 6; ++i) {                    // Matrix<float,6,9> * Vector<double,9>
      double s = 0;
      for (int j = 0; j < 9; ++j) s += (double)(float)pinvA[i * 9 + j] * z[j];
      vals[i] = s;
    }
    const double denom = 4.0 * vals[0] * vals[1] - (vals[2] * vals[2]);
    if (std::fabs(denom) < 0.01) return false;
    const float ox = (float)((vals[2] * vals[4] - 2.0 * vals[1] * vals[3]) / denom);
    const float oy = (float)((vals[2] * vals[3] - 2.0 * vals[0] * vals[4]) / denom);
    dx = ox; dy = oy;
    const double sX = 0.34574, sY = 0.38944;
    dx = std::erf(dx / (sX * std::sqrt(2.0))) / 2.0;
    dy = std::erf(dy / (sY * std::sqrt(2.0))) / 2.0;
    const double nrm = std::sqrt(dx * dx + dy * dy);
    if (nrm >= 0.5) { const double scale = nrm / 0.5; dx /= scale; dy /= scale; }
    return true;
  }

  void subpixel_view(const int32_t* idisp, float* out) const {                 // :1497-1614
    for (int j = 0; j < orows; ++j)
      for (int i = 0; i < ocols; ++i) {
        const size_t idx = (size_t)j * ocols + i;
        const int32_t* ip = idisp + idx * 3;
        float* o = out + idx * 3;
        if (!ip[2]) { o[0] = (float)ip[0]; o[1] = (float)ip[1]; o[2] = 0.0f; continue; }
        const int dx = ip[0], dy = ip[1];
        if (subpixel == SUB_NONE) { o[0] = (float)dx; o[1] = (float)dy; o[2] = 1.0f; continue; }
        Bounds const& b = bounds[idx];
        const int width = b.v[2] - b.v[0] + 1;
        const int mi = (dy - b.v[1]) * width + (dx - b.v[0]);
        int xl = -1, xr = 1, yu = -width, yd = width;
        bool tb = false, bb = false, lb = false, rb = false;
        if (dx == b.v[0]) { xl = 0; lb = true; }
        if (dx == b.v[2]) { xr = 0; rb = true; }
        if (dy == b.v[1]) { yu = 0; tb = true; }
        if (dy == b.v[3]) { yd = 0; bb = true; }
        const AccumCostType* a = accum.data() + starts[idx];
        double ddx = 0, ddy = 0;
        bool valid = true;
        if (subpixel == SUB_PARABOLA) {
          const double z[9] = {(double)a[mi + xl + yu], (double)a[mi + yu], (double)a[mi + xr + yu], (double)a[mi + xl], (double)a[mi],
                               (double)a[mi + xr], (double)a[mi + xl + yd], (double)a[mi + yd], (double)a[mi + xr + yd]};
          valid = parabola_peak(z, ddx, ddy);
        } else {
          ddx = subpixel_offset(a[mi + xl], a[mi], a[mi + xr], lb, rb);
          ddy = subpixel_offset(a[mi + yu], a[mi], a[mi + yd], tb, bb);
        }
        if (valid) { o[0] = (float)(dx + ddx); o[1] = (float)(dy + ddy); } else { o[0] = (float)dx; o[1] = (float)dy; }
        o[2] = 1.0f;
      }
  }

  // semi_global_matching_func (:2387-2448)
  int run(U8Img const& L, U8Img const& R, const uint8_t* lmask, int lmw, int lmh, const uint8_t* rmask, int rmw, int rmh,
          const int32_t* prev, int pw, int ph) {
    const int hk = (kernel - 1) / 2;
    min_row = hk - min_dy; min_col = hk - min_dx;
    max_row = std::min(L.h - 1 - hk, R.h - 1 - (hk + max_dy));
    max_col = std::min(L.w - 1 - hk, R.w - 1 - (hk + max_dx));
    if (min_row < 0) min_row = 0;
    if (min_col < 0) min_col = 0;
    if (max_row > L.h - 1) max_row = L.h - 1;
    if (max_col > L.w - 1) max_col = L.w - 1;
    ocols = max_col - min_col + 1; orows = max_row - min_row + 1;
    if (ocols <= 0 || orows <= 0) return -1;
    populate_adjacent();
    bounds.assign((size_t)ocols * orows, Bounds{{min_dx, min_dy, max_dx, max_dy}});
    int err = 0;
    if (!populate_bounds(lmask, lmw, lmh, rmask, rmw, rmh, prev, pw, ph, &err)) return err ? err : 1;   // 1 = "no valid data": all-invalid output
    cost.assign(main_buf_size, 0);
    accum.assign(main_buf_size, 0);
    compute_costs(L, R);
    if (use_mgm) accumulate_mgm(L); else accumulate(L);
    return 0;
  }
};

}  // namespace

struct vwo_sgm { Matcher m; };

extern "C" {

int vwo_u8_convert(const float* src, int w, int h, uint8_t* dst) {
  if (!src || !dst || w <= 0 || h <= 0) return -1;
  double min_val = std::numeric_limits<double>::max(), max_val = -std::numeric_limits<double>::max();
  for (size_t i = 0; i < (size_t)w * h; ++i) { const double v = src[i]; if (v < min_val) min_val = v; if (v > max_val) max_val = v; }
  if (max_val == min_val) max_val = min_val + 1.0;
  const float old_min = (float)min_val, old_max = (float)max_val, new_min = 0.0f, new_max = 255.0f;
  const double ratio = (old_max == old_min) ? 0.0 : (new_max - new_min) / (double)(old_max - old_min);
  for (size_t i = 0; i < (size_t)w * h; ++i) {
    float v = src[i];
    if (v > old_max) v = old_max;                    // clamp(input, min, max): no-op up to float rounding of the limits
    if (v < old_min) v = old_min;
    const float n = (float)((v - old_min) * ratio + new_min);
    dst[i] = (uint8_t)n;                             // pixel_cast<uint8>: truncation
  }
  return 0;
}

int vwo_census_transform(const uint8_t* img, int w, int h, int kernel, int ternary, int threshold, uint64_t* out) {
  if (!img || !out || (kernel != 3 && kernel != 5 && kernel != 7 && kernel != 9) || w < kernel || h < kernel) return -1;
  std::vector<uint64_t> v; int ow, oh;
  census_image(U8Img{img, w, h}, kernel, ternary != 0, threshold, v, ow, oh);
  std::memcpy(out, v.data(), v.size() * 8);
  return 0;
}

int vwo_hamming_distance(uint64_t a, uint64_t b) { return hamming(a, b); }

vwo_sgm* vwo_sgm_create(int cost_type, int use_mgm, int min_dx, int min_dy, int max_dx, int max_dy, int kernel, int subpixel,
                        int sbx, int sby, size_t memory_limit_mb, int p1, int p2, int ternary_thr, int num_threads) {
  const bool census = cost_type == COST_CENSUS || cost_type == COST_TERNARY;
  // SGM.cc:1886-1890 (NoImplErr) unless the caller opted in to the block cost; cost type 2 (the NCC flavour of get_cost_block) is not restated
  if (!census && !(g_allow_block_cost && (cost_type == 0 || cost_type == 1))) return nullptr;
  if (census && kernel != 3 && kernel != 5 && kernel != 7 && kernel != 9) return nullptr;    // :1877-1884
  if (!census && (kernel < 1 || kernel % 2 == 0)) return nullptr;
  vwo_sgm* s = new vwo_sgm;
  Matcher& m = s->m;
  m.use_mgm = use_mgm != 0;
  m.cost_type = cost_type; m.min_dx = min_dx; m.min_dy = min_dy; m.max_dx = max_dx; m.max_dy = max_dy; m.kernel = kernel;
  m.subpixel = subpixel; m.sbx = sbx; m.sby = sby; m.memory_limit_mb = memory_limit_mb; m.ternary_thr = ternary_thr;
  m.num_threads = num_threads > 0 ? num_threads : 1;
  m.set_parameters((uint16_t)p1, (uint16_t)p2);
  return s;
}
void vwo_sgm_destroy(vwo_sgm* s) { delete s; }
void vwo_set_sgm_host_threads(int n) { g_host_threads = n > 0 ? n : 1; }
void vwo_set_sgm_allow_block_cost(int on) { g_allow_block_cost = on != 0; }

int vwo_sgm_output_size(vwo_sgm* s, int* ow, int* oh) { *ow = s->m.ocols; *oh = s->m.orows; return 0; }

int vwo_sgm_run(vwo_sgm* s, const uint8_t* left, int lw, int lh, const uint8_t* right, int rw, int rh,
                const uint8_t* lmask, int lmw, int lmh, const uint8_t* rmask, int rmw, int rmh,
                const int32_t* prev, int pw, int ph, int32_t* out_disp, int cap_pixels) {
  Matcher& m = s->m;
  int rc = m.run(U8Img{left, lw, lh}, U8Img{right, rw, rh}, lmask, lmw, lmh, rmask, rmw, rmh, prev, pw, ph);
  if (rc < 0) return rc;
  if ((size_t)m.ocols * m.orows > (size_t)cap_pixels) return -4;
  if (rc == 1) { std::memset(out_disp, 0, (size_t)m.ocols * m.orows * 12); m.accum.clear(); return 0; }
  m.create_disparity_view(out_disp);
  return 0;
}

int vwo_sgm_subpixel(vwo_sgm* s, const int32_t* int_disp, float* out3f) {
  if (s->m.accum.empty()) {
    for (size_t i = 0; i < (size_t)s->m.ocols * s->m.orows; ++i) { out3f[i * 3] = (float)int_disp[i * 3]; out3f[i * 3 + 1] = (float)int_disp[i * 3 + 1]; out3f[i * 3 + 2] = 0; }
    return 0;
  }
  s->m.subpixel_view(int_disp, out3f);
  return 0;
}

// accessors for the tests
size_t vwo_sgm_buffer_size(vwo_sgm* s) { return s->m.accum.size(); }
int vwo_sgm_read(vwo_sgm* s, int32_t* bounds4, uint64_t* starts, uint8_t* cost, uint16_t* accum) {
  Matcher& m = s->m;
  const size_t n = (size_t)m.ocols * m.orows;
  if (bounds4) for (size_t i = 0; i < n; ++i) for (int k = 0; k < 4; ++k) bounds4[i * 4 + k] = m.bounds[i].v[k];
  if (starts) for (size_t i = 0; i < n; ++i) starts[i] = m.starts[i];
  if (cost) std::memcpy(cost, m.cost.data(), m.cost.size());
  if (accum) std::memcpy(accum, m.accum.data(), m.accum.size() * 2);
  return 0;
}
int vwo_sgm_p1p2(vwo_sgm* s, int* p1, int* p2) { *p1 = s->m.p1; *p2 = s->m.p2; return 0; }

// calc_disparity_sgm (SGM.cc:167-229) on already cropped float regions: left lw x lh, right (lw + sx) x (lh + sy).
int vwo_calc_disparity_sgm_x(int cost_type, int use_mgm, const float* left, int lw, int lh, const float* right, int rw, int rh,
                           int sx, int sy, int kernel, int subpixel, int sbx, int sby, size_t memory_limit_mb, int num_threads,
                           const uint8_t* lmask, int lmw, int lmh, const uint8_t* rmask, int rmw, int rmh,
                           const int32_t* prev, int pw, int ph, int p1, int p2, int32_t* out_disp, float* out_subpixel, int* ow, int* oh) {
  if (!left || !right || lw <= 0 || lh <= 0 || rw <= 0 || rh <= 0) return -1;
  std::vector<uint8_t> l8((size_t)lw * lh), r8((size_t)rw * rh);
  vwo_u8_convert(left, lw, lh, l8.data());
  vwo_u8_convert(right, rw, rh, r8.data());
  vwo_sgm* s = vwo_sgm_create(cost_type, use_mgm, 0, 0, sx, sy, kernel, subpixel, sbx, sby, memory_limit_mb, p1, p2, 5, num_threads);      // p1 / p2 = 0: the defaults of SGM.cc:106-160
  if (!s) return -2;
  const int hk = (kernel - 1) / 2;
  const int eow = std::min(lw - 1 - hk, rw - 1 - (hk + sx)) - hk + 1, eoh = std::min(lh - 1 - hk, rh - 1 - (hk + sy)) - hk + 1;
  int rc = vwo_sgm_run(s, l8.data(), lw, lh, r8.data(), rw, rh, lmask, lmw, lmh, rmask, rmw, rmh, prev, pw, ph, out_disp,
                       eow > 0 && eoh > 0 ? eow * eoh : 0);
  if (rc == 0) {
    *ow = s->m.ocols; *oh = s->m.orows;
    if (out_subpixel) vwo_sgm_subpixel(s, out_disp, out_subpixel);
  }
  vwo_sgm_destroy(s);
  return rc;
}
int vwo_calc_disparity_sgm_p(int cost_type, const float* left, int lw, int lh, const float* right, int rw, int rh,
                           int sx, int sy, int kernel, int subpixel, int sbx, int sby, size_t memory_limit_mb, int num_threads,
                           const uint8_t* lmask, int lmw, int lmh, const uint8_t* rmask, int rmw, int rmh,
                           const int32_t* prev, int pw, int ph, int p1, int p2, int32_t* out_disp, float* out_subpixel, int* ow, int* oh) {
  return vwo_calc_disparity_sgm_x(cost_type, 0, left, lw, lh, right, rw, rh, sx, sy, kernel, subpixel, sbx, sby, memory_limit_mb, num_threads,
                                  lmask, lmw, lmh, rmask, rmw, rmh, prev, pw, ph, p1, p2, out_disp, out_subpixel, ow, oh);
}

int vwo_calc_disparity_sgm(int cost_type, const float* left, int lw, int lh, const float* right, int rw, int rh,
                           int sx, int sy, int kernel, int subpixel, int sbx, int sby, size_t memory_limit_mb, int num_threads,
                           const uint8_t* lmask, int lmw, int lmh, const uint8_t* rmask, int rmw, int rmh,
                           const int32_t* prev, int pw, int ph, int32_t* out_disp, float* out_subpixel, int* ow, int* oh) {
  return vwo_calc_disparity_sgm_p(cost_type, left, lw, lh, right, rw, rh, sx, sy, kernel, subpixel, sbx, sby, memory_limit_mb, num_threads,
                                  lmask, lmw, lmh, rmask, rmw, rmh, prev, pw, ph, 0, 0, out_disp, out_subpixel, ow, oh);
}

}  // extern "C"
