// vw/Stereo.h — vw::stereo entry points of the block-matching hot path with the reference's signatures,
// implemented on libvwgpu.so (include/vwgpu.h).  Header-only; link with -lvwgpu.
//
//   calc_disparity               src/vw/Stereo/Correlation.h:50-57   (impl Correlation.cc:330-375)
//   fast_box_sum                 src/vw/Stereo/Algorithms.h:41-129
//   cross_corr_consistency_check src/vw/Stereo/Correlate.h:52-58     (impl Correlate.cc:1441-1502)
//   correlate                    legacy single-level entry, signature recovered from
//                                src/vw/Stereo/tests/TestCorrelationView.cxx:79-82,213-215
//   pyramid_correlate            src/vw/Stereo/CorrelationView.h:195-230 (PyramidCorrelationView :35-190, BM algorithm)
//   parabola_subpixel            src/vw/Stereo/ParabolaSubpixelView.h:112-117
//   prefilter_image              src/vw/Stereo/PreFilter.h:76-95
//   rm_outliers_using_thresh / disparity_cleanup_using_thresh / disparity_mask
//                                src/vw/Stereo/DisparityMap.h:387-441, 236-253
//   SearchParam, subdivide_regions, calc_seconds_per_op   src/vw/Stereo/Correlation.h:66-122
//   SemiGlobalMatcher, calc_disparity_sgm                 src/vw/Stereo/SGM.h:75-157,360-375
// Errors: the C ABI's status codes become the reference's exception types (src/vw/Core/Exception.h:225-253).
// Threading: one engine context per (host thread x GPU), created lazily — the reference calls these functions
// concurrently from its tile threads (src/vw/Image/ImageIO.h:228-251).
#ifndef VWLITE_STEREO_H
#define VWLITE_STEREO_H

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <utility>
#include <vector>

#include "Engine.h"
#include "Filter.h"
#include "Image.h"
#include "vwgpu.h"

namespace vw {
namespace stereo {

enum CostFunctionType {                  // src/vw/Stereo/CostFunctions.h:143-149
  ABSOLUTE_DIFFERENCE, SQUARED_DIFFERENCE, CROSS_CORRELATION, CENSUS_TRANSFORM, TERNARY_CENSUS_TRANSFORM
};
enum PrefilterModeType { PREFILTER_NONE = 0, PREFILTER_MEANSUB = 1, PREFILTER_LOG = 2 };   // PrefilterEnum.h:24-28

enum CorrelationAlgorithm {              // src/vw/Stereo/CorrelationAlgorithms.h:29-35
  VW_CORRELATION_BM = 0, VW_CORRELATION_SGM = 1, VW_CORRELATION_MGM = 2, VW_CORRELATION_FINAL_MGM = 3, VW_CORRELATION_OTHER = 4
};

namespace detail {
using engine::thread_context;
using engine::check;
}  // namespace detail

/// calc_disparity — same signature and semantics as the reference (Correlation.h:50-57).
inline ImageView<PixelMask<Vector2i>>
calc_disparity(CostFunctionType cost_type,
               ImageViewRef<PixelGray<float>> const& left_in,
               ImageViewRef<PixelGray<float>> const& right_in,
               BBox2i const& left_region,     // valid region in the left image
               Vector2i const& search_volume, // max disparity to search in the right image
               Vector2i const& kernel_size) {
  // The reference's sanity checks (Correlation.cc:341-351); the engine repeats the rest behind the ABI.
  VW_ASSERT(left_region.min().x() >= 0 && left_region.min().y() >= 0 &&
            left_region.max().x() <= left_in.cols() && left_region.max().y() <= left_in.rows(),
            ArgumentErr() << "calc_disparity: Region not inside left image.");
  BBox2i right_region = left_region;
  right_region.max() += search_volume - Vector2i(1, 1);                          // Correlation.cc:356-357
  const int32 lw = left_region.width(), lh = left_region.height();
  const int32 rw = right_region.width(), rh = right_region.height();

  // Rasterise the two crops (Correlation.cc:358-359).  A handle that wraps a plain ImageView is used in place.
  ImageView<PixelGray<float>> lbuf, rbuf;
  int32 ls = 0, rs = 0;
  const PixelGray<float>* lp = left_in.plain_data(ls);
  const PixelGray<float>* rp = right_in.plain_data(rs);
  if (lp) lp += (size_t)left_region.min().y() * ls + left_region.min().x();
  else { lbuf = crop(left_in, left_region); lp = lbuf.data(); ls = lw; }
  if (rp && right_region.max().x() <= right_in.cols() && right_region.max().y() <= right_in.rows())
    rp += (size_t)right_region.min().y() * rs + right_region.min().x();
  else { rbuf = crop(right_in, right_region); rp = rbuf.data(); rs = rw; }

  const int32 ow = lw - kernel_size[0] + 1, oh = lh - kernel_size[1] + 1;
  ImageView<PixelMask<Vector2i>> out(ow > 0 ? ow : 0, oh > 0 ? oh : 0);
  vwgpu_ctx* ctx = detail::thread_context();
  detail::check(ctx, vwgpu_calc_disparity(ctx, (int)cost_type,
                                          reinterpret_cast<const float*>(lp), lw, lh, ls,
                                          reinterpret_cast<const float*>(rp), rw, rh, rs,
                                          kernel_size[0], kernel_size[1], search_volume[0], search_volume[1],
                                          reinterpret_cast<int32_t*>(out.data()), 0));
  return out;
}

/// cross_corr_consistency_check — in place on l2r, as the reference (Correlate.h:52-58).
inline void cross_corr_consistency_check(ImageView<PixelMask<Vector2i>> const& l2r,
                                         ImageView<PixelMask<Vector2i>> const& r2l,
                                         float cross_corr_threshold, bool /*verbose*/ = false) {
  vwgpu_ctx* ctx = detail::thread_context();
  detail::check(ctx, vwgpu_cross_corr_consistency_check(ctx, reinterpret_cast<int32_t*>(l2r.data()), l2r.cols(), l2r.rows(), 0,
                                                        reinterpret_cast<const int32_t*>(r2l.data()), r2l.cols(), r2l.rows(), 0,
                                                        cross_corr_threshold));
}

/// fast_box_sum<AccumulatorType>(image, kernel) — Algorithms.h:41-43.  The engine forms the sums in float64 in the
/// reference's running-sum order (the only accumulator the reference's callers use is double, CostFunctions.h:55-57);
/// other accumulator types receive the rounded float64 sums.
template <class AccumulatorType, class ViewT>
ImageView<AccumulatorType> fast_box_sum(ImageViewBase<ViewT> const& image, Vector2i const& kernel) {
  VW_ASSERT(kernel[0] % 2 == 1 && kernel[1] % 2 == 1, ArgumentErr() << "fast_box_sum: Kernel input not sized with odd values.");
  ImageView<PixelGray<float>> in = pixel_cast<PixelGray<float>>(image.impl());
  VW_ASSERT(in.cols() >= kernel[0] && in.rows() >= kernel[1], ArgumentErr() << "fast_box_sum: Image is not big enough for kernel.");
  const int32 ow = in.cols() - kernel[0] + 1, oh = in.rows() - kernel[1] + 1;
  ImageView<double> sums(ow, oh);
  vwgpu_ctx* ctx = detail::thread_context();
  detail::check(ctx, vwgpu_fast_box_sum(ctx, reinterpret_cast<const float*>(in.data()), in.cols(), in.rows(), 0, kernel[0], kernel[1],
                                        sums.data(), 0));
  ImageView<AccumulatorType> out(ow, oh);
  for (int32 r = 0; r < oh; ++r) for (int32 c = 0; c < ow; ++c) out(c, r) = AccumulatorType(sums(c, r));
  return out;
}

/// Legacy correlate(): prefilter -> calc_disparity over the whole left image -> optional R->L run + L/R check.
/// search_volume is a BBox2i of signed disparities, max() INCLUSIVE as the legacy view took it (its own test searches
/// BBox2i(1,1,1,1) and expects valid (1,1) answers, TestCorrelationView.cxx:46,64-66); the result holds signed disparities
/// (offset by search_volume.min()), like the legacy view did.  The overload taking a prefilter object follows below.
inline ImageView<PixelMask<Vector2i>>
correlate(ImageView<PixelGray<float>> const& left, ImageView<PixelGray<float>> const& right,
          BBox2i const& search_volume, Vector2i const& kernel_size,
          CostFunctionType cost_type = ABSOLUTE_DIFFERENCE, float consistency_threshold = -1) {
  const Vector2i half(kernel_size[0] / 2, kernel_size[1] / 2);
  const Vector2i s = search_volume.size() + Vector2i(1, 1);
  // Pad so that every pixel of `left` gets a disparity: windows are centred (edge-extended with zeros like the
  // legacy CorrelationView) and the right crop is shifted by search_volume.min().
  const int32 W = left.cols(), H = left.rows();
  ImageView<PixelGray<float>> lpad = crop(edge_extend(left, ZeroEdgeExtension()), -half[0], -half[1], W + 2 * half[0], H + 2 * half[1]);
  ImageView<PixelGray<float>> rpad = crop(edge_extend(right, ZeroEdgeExtension()),
                                         -half[0] + search_volume.min().x(), -half[1] + search_volume.min().y(),
                                         W + 2 * half[0] + s[0] - 1, H + 2 * half[1] + s[1] - 1);
  ImageView<PixelMask<Vector2i>> l2r = calc_disparity(cost_type, lpad, rpad, bounding_box(lpad), s, kernel_size);
  if (consistency_threshold >= 0) {
    // R->L: search the left image around each right pixel over the mirrored range.
    ImageView<PixelGray<float>> rp2 = crop(edge_extend(right, ZeroEdgeExtension()), -half[0], -half[1], W + 2 * half[0], H + 2 * half[1]);
    ImageView<PixelGray<float>> lp2 = crop(edge_extend(left, ZeroEdgeExtension()),
                                          -half[0] - search_volume.max().x(), -half[1] - search_volume.max().y(),
                                          W + 2 * half[0] + s[0] - 1, H + 2 * half[1] + s[1] - 1);
    ImageView<PixelMask<Vector2i>> r2l = calc_disparity(cost_type, rp2, lp2, bounding_box(rp2), s, kernel_size);
    const Vector2i lmin = search_volume.min(), rmin(-search_volume.max().x(), -search_volume.max().y());
    for (int32 r = 0; r < H; ++r) for (int32 c = 0; c < W; ++c) { l2r(c, r).child() += lmin; r2l(c, r).child() += rmin; }
    cross_corr_consistency_check(l2r, r2l, consistency_threshold);
    return l2r;
  }
  for (int32 r = 0; r < H; ++r) for (int32 c = 0; c < W; ++c) l2r(c, r).child() += search_volume.min();
  return l2r;
}

/// prefilter_image — rasterised (PreFilter.h:76-95).
inline ImageView<PixelGray<float>>
prefilter_image(ImageView<PixelGray<float>> const& image, PrefilterModeType prefilter_mode, float prefilter_width) {
  ImageView<PixelGray<float>> out(image.cols(), image.rows());
  if (image.cols() == 0 || image.rows() == 0) return out;
  vwgpu_ctx* ctx = detail::thread_context();
  detail::check(ctx, vwgpu_prefilter_image(ctx, reinterpret_cast<const float*>(image.data()), image.cols(), image.rows(), 0,
                                           (int)prefilter_mode, prefilter_width, reinterpret_cast<float*>(out.data()), 0));
  return out;
}

/// The prefilter objects of PreFilter.h:40-73; filter() returns the rasterised image (zero outside, like ConstantEdgeExtension
/// views do once rasterised over the image's own box).
struct NullOperation {
  ImageView<PixelGray<float>> filter(ImageView<PixelGray<float>> const& image) const { return prefilter_image(image, PREFILTER_NONE, 0.0f); }
};
struct LaplacianOfGaussian {
  float kernel_width;
  explicit LaplacianOfGaussian(float size) : kernel_width(size) {}
  ImageView<PixelGray<float>> filter(ImageView<PixelGray<float>> const& image) const { return prefilter_image(image, PREFILTER_LOG, kernel_width); }
};
struct SubtractedMean {
  float kernel_width;
  explicit SubtractedMean(float size) : kernel_width(size) {}
  ImageView<PixelGray<float>> filter(ImageView<PixelGray<float>> const& image) const { return prefilter_image(image, PREFILTER_MEANSUB, kernel_width); }
};

/// Legacy correlate() with a prefilter object (TestCorrelationView.cxx:79-84 call shape).
template <class PreFilterT>
inline ImageView<PixelMask<Vector2i>>
correlate(ImageView<PixelGray<float>> const& left, ImageView<PixelGray<float>> const& right, PreFilterT const& prefilter,
          BBox2i const& search_volume, Vector2i const& kernel_size,
          CostFunctionType cost_type = ABSOLUTE_DIFFERENCE, float consistency_threshold = -1) {
  return correlate(prefilter.filter(left), prefilter.filter(right), search_volume, kernel_size, cost_type, consistency_threshold);
}

/// parabola_subpixel — the rasterised ParabolaSubpixelView (ParabolaSubpixelView.h:112-117).
inline ImageView<PixelMask<Vector2f>>
parabola_subpixel(ImageViewRef<PixelMask<Vector2f>> const& disparity,
                  ImageViewRef<PixelGray<float>> const& left_image, ImageViewRef<PixelGray<float>> const& right_image,
                  PrefilterModeType prefilter_mode, float prefilter_width, Vector2i const& kernel_size) {
  VW_ASSERT(disparity.cols() == left_image.cols() && disparity.rows() == left_image.rows(),
            ArgumentErr() << "SubpixelView: Disparity image must match left image.");   // ParabolaSubpixelView.h:67-69
  ImageView<PixelMask<Vector2f>> d = disparity, out(disparity.cols(), disparity.rows());
  ImageView<PixelGray<float>> l = left_image, r = right_image;
  if (d.cols() == 0 || d.rows() == 0) return out;
  vwgpu_ctx* ctx = detail::thread_context();
  detail::check(ctx, vwgpu_parabola_subpixel(ctx, reinterpret_cast<const float*>(d.data()), d.cols(), d.rows(), 0,
                                             reinterpret_cast<const float*>(l.data()), 0,
                                             reinterpret_cast<const float*>(r.data()), r.cols(), r.rows(), 0,
                                             (int)prefilter_mode, prefilter_width, kernel_size[0], kernel_size[1],
                                             reinterpret_cast<float*>(out.data()), 0));
  return out;
}

namespace detail {
inline ImageView<PixelMask<Vector2i>> disparity_filter(ImageView<PixelMask<Vector2i>> const& d, int32 hh, int32 hv,
                                                       double pthr, double rthr, int cleanup) {
  ImageView<PixelMask<Vector2i>> out(d.cols(), d.rows());
  if (d.cols() == 0 || d.rows() == 0) return out;
  vwgpu_ctx* ctx = thread_context();
  check(ctx, vwgpu_disparity_filter(ctx, reinterpret_cast<const int32_t*>(d.data()), d.cols(), d.rows(), hh, hv, pthr, rthr, cleanup,
                                    reinterpret_cast<int32_t*>(out.data())));
  return out;
}
}  // namespace detail

/// rm_outliers_using_thresh — rasterised over the whole image (DisparityMap.h:403-414).
template <class ViewT>
ImageView<PixelMask<Vector2i>> rm_outliers_using_thresh(ImageViewBase<ViewT> const& disparity_map, int32 half_h_kernel, int32 half_v_kernel,
                                                        double pixel_threshold, double rejection_threshold) {
  VW_ASSERT(half_h_kernel > 0 && half_v_kernel > 0, ArgumentErr() << "RmOutliersFunc: half kernel sizes must be non-zero.");
  ImageView<PixelMask<Vector2i>> d = disparity_map.impl();
  return detail::disparity_filter(d, half_h_kernel, half_v_kernel, pixel_threshold, rejection_threshold, 0);
}
/// disparity_cleanup_using_thresh — two passes, the second with (1,1,3.0,0.20) (DisparityMap.h:422-441).
template <class ViewT>
ImageView<PixelMask<Vector2i>> disparity_cleanup_using_thresh(ImageViewBase<ViewT> const& disparity_map, int32 h_half_kernel, int32 v_half_kernel,
                                                              double pixel_threshold, double rejection_threshold) {
  VW_ASSERT(h_half_kernel > 0 && v_half_kernel > 0, ArgumentErr() << "RmOutliersFunc: half kernel sizes must be non-zero.");
  ImageView<PixelMask<Vector2i>> d = disparity_map.impl();
  return detail::disparity_filter(d, h_half_kernel, v_half_kernel, pixel_threshold, rejection_threshold, 1);
}
/// disparity_mask — rasterised (DisparityMap.h:236-253); masks are uint8 images, 0 = no data.
template <class ViewT, class M1, class M2>
ImageView<PixelMask<Vector2i>> disparity_mask(ImageViewBase<ViewT> const& disparity_map, ImageViewBase<M1> const& left_mask,
                                              ImageViewBase<M2> const& right_mask) {
  ImageView<PixelMask<Vector2i>> d = copy(disparity_map.impl());
  ImageView<uint8> m1 = left_mask.impl(), m2 = right_mask.impl();
  VW_ASSERT(d.cols() == m1.cols() && d.rows() == m1.rows(), ArgumentErr() << "disparity_mask: input and left mask are not same dimensions.");
  if (d.cols() == 0 || d.rows() == 0) return d;
  vwgpu_ctx* ctx = detail::thread_context();
  detail::check(ctx, vwgpu_disparity_mask(ctx, reinterpret_cast<int32_t*>(d.data()), d.cols(), d.rows(), m1.data(), m2.data(), m2.cols(), m2.rows()));
  return d;
}

/// SearchParam / subdivide_regions — the zone scheduler (Correlation.h:66-91,118-122).
struct SearchParam : public std::pair<BBox2i, BBox2i> {
  SearchParam(BBox2i const& image_region, BBox2i const& disparity_range) : std::pair<BBox2i, BBox2i>(image_region, disparity_range) {}
  BBox2i& image_region() { return this->first; }            BBox2i const& image_region() const { return this->first; }
  BBox2i& disparity_range() { return this->second; }        BBox2i const& disparity_range() const { return this->second; }
  double search_volume() const {
    return (double)first.width() * (double)first.height() * (double)second.width() * (double)second.height();
  }
};
struct SearchParamLessThan {
  bool operator()(SearchParam const& A, SearchParam const& B) const { return A.search_volume() < B.search_volume(); }
};
inline bool subdivide_regions(ImageView<PixelMask<Vector2i>> const& disparity, BBox2i const& current_bbox,
                              std::vector<SearchParam>& list, Vector2i const& kernel_size, int32 /*fail_count*/ = 0) {
  // the engine scheduler works on a whole image: hand it the crop and shift the zones back
  ImageView<PixelMask<Vector2i>> d = crop(disparity, current_bbox);
  if (d.cols() == 0 || d.rows() == 0) return true;
  std::vector<int32_t> zones(8 * 1024);
  int n = vwgpu_subdivide_regions(reinterpret_cast<const int32_t*>(d.data()), d.cols(), d.rows(), kernel_size[0], kernel_size[1],
                                  zones.data(), (int)zones.size() / 8);
  if (n > (int)zones.size() / 8) {
    zones.resize((size_t)n * 8);
    n = vwgpu_subdivide_regions(reinterpret_cast<const int32_t*>(d.data()), d.cols(), d.rows(), kernel_size[0], kernel_size[1], zones.data(), n);
  }
  VW_ASSERT(n >= 0, ArgumentErr() << "subdivide_regions: bad arguments");
  for (int i = 0; i < n; ++i) {
    const int32_t* z = &zones[(size_t)i * 8];
    list.push_back(SearchParam(BBox2i(Vector2i(z[0], z[1]) + current_bbox.min(), Vector2i(z[2], z[3]) + current_bbox.min()),
                               BBox2i(Vector2i(z[4], z[5]), Vector2i(z[6], z[7]))));
  }
  return true;
}

/// calc_seconds_per_op (Correlation.cc:382-436): time a fake calc_disparity and divide by region x search volume.
/// The reference grows the problem until one call takes a second; on the GPU that would need tens of GB, so the
/// loop stops at 20 ms or 2048^2 — the figure only feeds pyramid_correlate's corr_timeout estimate.
inline double calc_seconds_per_op(CostFunctionType cost_type, Vector2i const& kernel_size) {
  double elapsed = -1.0, seconds_per_op = -1.0;
  int lsize = 100;
  while (elapsed < 0.02 && lsize < 2048) {
    lsize = (int)std::ceil(lsize * 1.2) + std::max(kernel_size[0], kernel_size[1]);
    ImageView<PixelGray<float>> fake_left(lsize, lsize), fake_right(lsize + lsize / 5, lsize + lsize / 5);
    for (int row = 0; row < fake_left.rows(); ++row) for (int col = 0; col < fake_left.cols(); ++col) fake_left(col, row) = float(col % 2 + 2 * (row % 5));
    for (int row = 0; row < fake_right.rows(); ++row) for (int col = 0; col < fake_right.cols(); ++col) fake_right(col, row) = float(3 * (col % 7) + row % 3);
    const Vector2i search(std::max(lsize / 5, 1), std::max(lsize / 5, 1));
    calc_disparity(cost_type, fake_left, fake_right, bounding_box(fake_left), search, kernel_size);   // warm-up (allocations)
    const auto t0 = std::chrono::steady_clock::now();
    calc_disparity(cost_type, fake_left, fake_right, bounding_box(fake_left), search, kernel_size);
    elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    seconds_per_op = elapsed / ((double)lsize * lsize * search[0] * search[1]);
  }
  return seconds_per_op;
}

/// SemiGlobalMatcher — the slice of the reference's class (SGM.h:75-352) that callers of calc_disparity_sgm touch: the
/// sub-pixel mode enum and create_disparity_view_subpixel on the matcher handed back through matcher_ptr.  The engine
/// computes the sub-pixel view in the same pass (the accumulation buffers live in HBM only during the call).
class SemiGlobalMatcher {
public:
  typedef ImageView<PixelMask<Vector2i>> DisparityImage;
  enum SgmSubpixelMode { SUBPIXEL_NONE = 0, SUBPIXEL_PARABOLA = 1, SUBPIXEL_LINEAR = 2, SUBPIXEL_POLY4 = 3,
                         SUBPIXEL_COSINE = 4, SUBPIXEL_LC_BLEND = 5 };
  /// SGM.h:157 — valid for the integer disparity this matcher produced.
  ImageView<PixelMask<Vector2f>> create_disparity_view_subpixel(DisparityImage const& integer_disparity) const {
    VW_ASSERT(integer_disparity.cols() == m_subpixel.cols() && integer_disparity.rows() == m_subpixel.rows(),
              ArgumentErr() << "create_disparity_view_subpixel: not the disparity this matcher produced.");
    ImageView<PixelMask<Vector2f>> out = copy(m_subpixel);
    for (int32 r = 0; r < out.rows(); ++r)
      for (int32 c = 0; c < out.cols(); ++c)
        if (!is_valid(integer_disparity(c, r))) {          // SGM.cc:1523-1527
          out(c, r) = PixelMask<Vector2f>(Vector2f(float(integer_disparity(c, r)[0]), float(integer_disparity(c, r)[1])));
          out(c, r).invalidate();
        }
    return out;
  }
  ImageView<PixelMask<Vector2f>> m_subpixel;   // filled by calc_disparity_sgm
};

/// Opt-in to the code the reference keeps behind a throw: with the flag set, calc_disparity_sgm with ABSOLUTE_DIFFERENCE /
/// SQUARED_DIFFERENCE runs fill_costs_block's mean-abs-difference cost (SGM.cc:1651-1738, p1 = 3, p2 = 250) instead of throwing
/// NoImplErr like compute_disparity_costs (SGM.cc:1887-1892).  Process wide, off by default; not part of the reference's API.
inline bool& sgm_allow_block_cost() { static bool flag = false; return flag; }

/// calc_disparity_sgm — the reference's signature (SGM.h:360-375); std::shared_ptr stands in for boost::shared_ptr.
inline ImageView<PixelMask<Vector2i>>
calc_disparity_sgm(CostFunctionType cost_type,
                   ImageView<PixelGray<float>> const& left_in, ImageView<PixelGray<float>> const& right_in,
                   BBox2i const& left_region, Vector2i const& search_volume, Vector2i const& kernel_size,
                   bool const use_mgm, SemiGlobalMatcher::SgmSubpixelMode const& subpixel_mode,
                   Vector2i const search_buffer, size_t const memory_limit_mb,
                   std::shared_ptr<SemiGlobalMatcher>& matcher_ptr,
                   ImageView<uint8> const* left_mask_ptr = 0, ImageView<uint8> const* right_mask_ptr = 0,
                   SemiGlobalMatcher::DisparityImage const* prev_disparity = 0) {
  VW_ASSERT(kernel_size[0] % 2 == 1 && kernel_size[1] % 2 == 1, ArgumentErr() << "calc_disparity_sgm: Kernel input not sized with odd values.");
  VW_ASSERT(kernel_size[0] <= left_region.width() && kernel_size[1] <= left_region.height(),
            ArgumentErr() << "calc_disparity_sgm: Kernel size too large of active region.");
  VW_ASSERT(left_region.min().x() >= 0 && left_region.min().y() >= 0 && left_region.max().x() <= left_in.cols() &&
            left_region.max().y() <= left_in.rows(), ArgumentErr() << "calc_disparity_sgm: Region not inside left image.");
  BBox2i right_region = left_region;
  right_region.max() += search_volume;                               // inclusive search volume (SGM.cc:186-191)
  ImageView<PixelGray<float>> l = crop(left_in, left_region);
  ImageView<PixelGray<float>> r = crop(edge_extend(right_in, ConstantEdgeExtension()), right_region);
  vwgpu_sgm_params p = vwgpu_sgm_params();
  p.cost_type = (int)cost_type; p.use_mgm = use_mgm ? 1 : 0; p.kernel_size = kernel_size[0]; p.subpixel_mode = (int)subpixel_mode;
  p.search_buffer_x = search_buffer[0]; p.search_buffer_y = search_buffer[1]; p.memory_limit_mb = memory_limit_mb;
  p.p1 = 0; p.p2 = 0; p.ternary_census_threshold = 5; p.num_threads = 1;
  p.allow_block_cost = sgm_allow_block_cost() ? 1 : 0;
  const size_t cap = (size_t)l.cols() * l.rows();
  std::vector<int32_t> disp(cap * 3);
  std::vector<float> sub(cap * 3);
  int ow = 0, oh = 0;
  vwgpu_ctx* ctx = detail::thread_context();
  try {
    detail::check(ctx, vwgpu_calc_disparity_sgm(ctx, &p, reinterpret_cast<const float*>(l.data()), l.cols(), l.rows(), 0,
                                                reinterpret_cast<const float*>(r.data()), r.cols(), r.rows(), 0, search_volume[0], search_volume[1],
                                                left_mask_ptr ? left_mask_ptr->data() : 0, left_mask_ptr ? left_mask_ptr->cols() : 0, left_mask_ptr ? left_mask_ptr->rows() : 0,
                                                right_mask_ptr ? right_mask_ptr->data() : 0, right_mask_ptr ? right_mask_ptr->cols() : 0, right_mask_ptr ? right_mask_ptr->rows() : 0,
                                                prev_disparity ? reinterpret_cast<const int32_t*>(prev_disparity->data()) : 0,
                                                prev_disparity ? prev_disparity->cols() : 0, prev_disparity ? prev_disparity->rows() : 0,
                                                disp.data(), sub.data(), cap, &ow, &oh));
  } catch (NoImplErr const&) {
    throw;                                                           // MAD cost / census size: as the reference's NoImplErr
  } catch (std::exception const& e) {                                // SGM.cc:221-226
    vw_throw(ArgumentErr() << "Failed to compute the correlation. See the online documentation (next_steps.html) for how to "
                           << "handle failures.\nDetailed error message: " << e.what() << "\n");
  }
  ImageView<PixelMask<Vector2i>> out(ow, oh);
  matcher_ptr.reset(new SemiGlobalMatcher());
  matcher_ptr->m_subpixel.set_size(ow, oh);
  std::memcpy(static_cast<void*>(out.data()), disp.data(), (size_t)ow * oh * 12);
  std::memcpy(static_cast<void*>(matcher_ptr->m_subpixel.data()), sub.data(), (size_t)ow * oh * 12);
  return out;
}

/// PyramidCorrelationView — lazy like the reference's (CorrelationView.h:35-190): nothing runs until a tile is
/// requested through prerasterize(bbox) / rasterize, and every tile is independent (CorrelationView.cc:273-886).
class PyramidCorrelationView : public ImageViewBase<PyramidCorrelationView> {
  // Type-erased handles, as in the reference (CorrelationView.h:38-45): nothing is rasterised until a tile is requested,
  // and then only the window that tile can touch (CorrelationView.cc:89-97) — the sources may be file-backed views.
  ImageViewRef<PixelGray<float>> m_left, m_right;
  ImageViewRef<uint8> m_left_mask, m_right_mask;
  vwgpu_pyramid_params m_p;
  int m_collar_size;                                      ///< Expand the size of the image for each tile before correlating
  ImageView<PixelMask<float>>* m_lr_disp_diff;            ///< L-R / R-L discrepancy output (CorrelationView.h:84), or NULL
  Vector2i m_region_ul;
public:
  typedef PixelMask<Vector2f> pixel_type;
  typedef pixel_type result_type;
  typedef CropView<ImageView<pixel_type>> prerasterize_type;

  PyramidCorrelationView(ImageViewRef<PixelGray<float>> const& left, ImageViewRef<PixelGray<float>> const& right,
                         ImageViewRef<uint8> const& left_mask, ImageViewRef<uint8> const& right_mask, vwgpu_pyramid_params const& p,
                         int collar_size = 0, ImageView<PixelMask<float>>* lr_disp_diff = NULL, Vector2i const& region_ul = Vector2i(0, 0))
      : m_left(left), m_right(right), m_left_mask(left_mask), m_right_mask(right_mask), m_p(p), m_collar_size(collar_size),
        m_lr_disp_diff(lr_disp_diff), m_region_ul(region_ul) {
    if (m_lr_disp_diff) {                                 // the tile threads write disjoint pixels of the caller's image
      m_p.lr_disp_diff = reinterpret_cast<float*>(m_lr_disp_diff->data());
      m_p.lr_disp_diff_cols = m_lr_disp_diff->cols(); m_p.lr_disp_diff_rows = m_lr_disp_diff->rows(); m_p.lr_disp_diff_stride = 0;
      m_p.region_ul_x = region_ul[0]; m_p.region_ul_y = region_ul[1];
    }
    VW_ASSERT(left_mask.cols() == left.cols() && left_mask.rows() == left.rows() &&
              right_mask.cols() == right.cols() && right_mask.rows() == right.rows(),
              ArgumentErr() << "PyramidCorrelationView: masks must match their images.");
  }
  int32 cols() const { return m_left.cols(); }
  int32 rows() const { return m_left.rows(); }
  int32 planes() const { return 1; }
  pixel_type operator()(int32 /*i*/, int32 /*j*/, int32 /*p*/ = 0) const {
    vw_throw(NoImplErr() << "PyramidCorrelationView::operator()(....) has not been implemented.");   // CorrelationView.h:166-171
    return pixel_type();
  }
  /// One tile; the returned image is indexed from the tile's own origin (0,0) = bbox.min().
  ImageView<pixel_type> correlate_tile(BBox2i const& bbox) const {
    ImageView<pixel_type> out(bbox.width(), bbox.height());
    if (bbox.empty()) return out;
    vwgpu_ctx* ctx = detail::thread_context();
    int32 ls = 0, rs = 0, lms = 0, rms = 0;
    const PixelGray<float>* lp = m_left.plain_data(ls);
    const PixelGray<float>* rp = m_right.plain_data(rs);
    const uint8* lmp = m_left_mask.plain_data(lms);
    const uint8* rmp = m_right_mask.plain_data(rms);
    if (lp && rp && lmp && rmp) {                  // plain in-memory images: used in place (the engine ships the window)
      detail::check(ctx, vwgpu_pyramid_correlate(ctx, reinterpret_cast<const float*>(lp), m_left.cols(), m_left.rows(), ls,
                                                 reinterpret_cast<const float*>(rp), m_right.cols(), m_right.rows(), rs,
                                                 lmp, lms, rmp, rms, &m_p,
                                                 bbox.min().x(), bbox.min().y(), bbox.width(), bbox.height(),
                                                 reinterpret_cast<float*>(out.data()), 0));
      return out;
    }
    // lazy sources: rasterise the tile's window only.  Window = tile grown by the pyramid padding half_kernel * 2^levels,
    // twice the search extent (the SGM branch's R->L runs) and the search range, cut at the image borders; all four
    // rasters share its origin, so the engine sees a smaller pair in which the tile sits at bbox - origin.
    const int64 up = int64(1) << std::max(0, std::min<int>(m_p.max_pyramid_levels, 12));
    const int64 sdx = std::max(0, m_p.search_max_x - m_p.search_min_x), sdy = std::max(0, m_p.search_max_y - m_p.search_min_y);
    const int64 padx = (m_p.kernel_x / 2) * up + 2 * sdx + 8, pady = (m_p.kernel_y / 2) * up + 2 * sdy + 8;
    const int64 wx0 = std::max<int64>(0, bbox.min().x() - padx + std::min(m_p.search_min_x, 0));
    const int64 wy0 = std::max<int64>(0, bbox.min().y() - pady + std::min(m_p.search_min_y, 0));
    const int64 wx1 = int64(bbox.max().x()) + padx + std::max(m_p.search_max_x, 0);
    const int64 wy1 = int64(bbox.max().y()) + pady + std::max(m_p.search_max_y, 0);
    const int32 ox = (int32)std::min<int64>(wx0, std::min(m_left.cols(), m_right.cols()));
    const int32 oy = (int32)std::min<int64>(wy0, std::min(m_left.rows(), m_right.rows()));
    const BBox2i lwin(ox, oy, (int32)(std::min<int64>(wx1, m_left.cols()) - ox), (int32)(std::min<int64>(wy1, m_left.rows()) - oy));
    const BBox2i rwin(ox, oy, (int32)(std::min<int64>(wx1, m_right.cols()) - ox), (int32)(std::min<int64>(wy1, m_right.rows()) - oy));
    VW_ASSERT(!lwin.empty() && !rwin.empty(), ArgumentErr() << "PyramidCorrelationView: the tile lies outside the images.");
    ImageView<PixelGray<float>> l = m_left.prerasterize(lwin), r = m_right.prerasterize(rwin);
    ImageView<uint8> lm = m_left_mask.prerasterize(lwin), rm = m_right_mask.prerasterize(rwin);
    vwgpu_pyramid_params p = m_p;
    p.region_ul_x -= ox; p.region_ul_y -= oy;
    detail::check(ctx, vwgpu_pyramid_correlate(ctx, reinterpret_cast<const float*>(l.data()), l.cols(), l.rows(), 0,
                                               reinterpret_cast<const float*>(r.data()), r.cols(), r.rows(), 0,
                                               lm.data(), 0, rm.data(), 0, &p,
                                               bbox.min().x() - ox, bbox.min().y() - oy, bbox.width(), bbox.height(),
                                               reinterpret_cast<float*>(out.data()), 0));
    return out;
  }
  /// Several tiles in one engine call (vwgpu_pyramid_correlate_batch, include/vwgpu.h): runs of equal-sized tiles go through the pyramid
  /// level loop together — every launch serves the group, one host round trip per level.  Tile i of the result is what correlate_tile(boxes[i])
  /// returns.  The reference hands tiles to its threads one by one (ImageIO.h:228-251); block_write_image below hands this view a run of
  /// blocks of a block row at a time (group_size()).
  std::vector<ImageView<pixel_type>> correlate_tiles(std::vector<BBox2i> const& boxes) const {
    const int n = (int)boxes.size();
    std::vector<ImageView<pixel_type>> out((size_t)n);
    if (n == 0) return out;
    std::vector<int> bx((size_t)n), by((size_t)n), bw((size_t)n), bh((size_t)n);
    std::vector<float*> ptr((size_t)n);
    BBox2i all;
    for (int i = 0; i < n; ++i) {
      VW_ASSERT(!boxes[i].empty(), ArgumentErr() << "PyramidCorrelationView: empty tile in a group.");
      out[i].set_size(boxes[i].width(), boxes[i].height());
      bx[i] = boxes[i].min().x(); by[i] = boxes[i].min().y(); bw[i] = boxes[i].width(); bh[i] = boxes[i].height();
      ptr[i] = reinterpret_cast<float*>(out[i].data());
      all.grow(boxes[i]);
    }
    vwgpu_ctx* ctx = detail::thread_context();
    int32 ls = 0, rs = 0, lms = 0, rms = 0;
    const PixelGray<float>* lp = m_left.plain_data(ls);
    const PixelGray<float>* rp = m_right.plain_data(rs);
    const uint8* lmp = m_left_mask.plain_data(lms);
    const uint8* rmp = m_right_mask.plain_data(rms);
    if (lp && rp && lmp && rmp) {
      detail::check(ctx, vwgpu_pyramid_correlate_batch(ctx, reinterpret_cast<const float*>(lp), m_left.cols(), m_left.rows(), ls,
                                                       reinterpret_cast<const float*>(rp), m_right.cols(), m_right.rows(), rs, lmp, lms, rmp, rms, &m_p,
                                                       n, bx.data(), by.data(), bw.data(), bh.data(), ptr.data(), NULL));
      return out;
    }
    // lazy sources: one window for the group (the union of the tiles grown as in correlate_tile)
    const int64 up = int64(1) << std::max(0, std::min<int>(m_p.max_pyramid_levels, 12));
    const int64 sdx = std::max(0, m_p.search_max_x - m_p.search_min_x), sdy = std::max(0, m_p.search_max_y - m_p.search_min_y);
    const int64 padx = (m_p.kernel_x / 2) * up + 2 * sdx + 8, pady = (m_p.kernel_y / 2) * up + 2 * sdy + 8;
    const int64 wx0 = std::max<int64>(0, all.min().x() - padx + std::min(m_p.search_min_x, 0));
    const int64 wy0 = std::max<int64>(0, all.min().y() - pady + std::min(m_p.search_min_y, 0));
    const int64 wx1 = int64(all.max().x()) + padx + std::max(m_p.search_max_x, 0);
    const int64 wy1 = int64(all.max().y()) + pady + std::max(m_p.search_max_y, 0);
    const int32 ox = (int32)std::min<int64>(wx0, std::min(m_left.cols(), m_right.cols()));
    const int32 oy = (int32)std::min<int64>(wy0, std::min(m_left.rows(), m_right.rows()));
    const BBox2i lwin(ox, oy, (int32)(std::min<int64>(wx1, m_left.cols()) - ox), (int32)(std::min<int64>(wy1, m_left.rows()) - oy));
    const BBox2i rwin(ox, oy, (int32)(std::min<int64>(wx1, m_right.cols()) - ox), (int32)(std::min<int64>(wy1, m_right.rows()) - oy));
    VW_ASSERT(!lwin.empty() && !rwin.empty(), ArgumentErr() << "PyramidCorrelationView: the tiles lie outside the images.");
    ImageView<PixelGray<float>> l = m_left.prerasterize(lwin), r = m_right.prerasterize(rwin);
    ImageView<uint8> lm = m_left_mask.prerasterize(lwin), rm = m_right_mask.prerasterize(rwin);
    vwgpu_pyramid_params p = m_p;
    p.region_ul_x -= ox; p.region_ul_y -= oy;
    for (int i = 0; i < n; ++i) { bx[i] -= ox; by[i] -= oy; }
    detail::check(ctx, vwgpu_pyramid_correlate_batch(ctx, reinterpret_cast<const float*>(l.data()), l.cols(), l.rows(), 0,
                                                     reinterpret_cast<const float*>(r.data()), r.cols(), r.rows(), 0, lm.data(), 0, rm.data(), 0, &p,
                                                     n, bx.data(), by.data(), bw.data(), bh.data(), ptr.data(), NULL));
    return out;
  }
  /// How many blocks of a block row block_write_image hands over at a time (rasterize_group).
  int32 group_size() const { return 8; }
  /// rasterize() for a run of blocks: collars as in rasterize (CorrelationView.h:123-133), the tiles correlated as one group.
  template <class DestT> void rasterize_group(std::vector<DestT> const& dests, std::vector<BBox2i> const& boxes) const {
    std::vector<BBox2i> proc(boxes);
    if (m_collar_size > 0) for (BBox2i& b : proc) b.expand(m_collar_size);
    std::vector<ImageView<pixel_type>> tiles = correlate_tiles(proc);
    for (size_t i = 0; i < boxes.size(); ++i)
      vw::rasterize(prerasterize_type(tiles[i], -proc[i].min().x(), -proc[i].min().y(), cols(), rows()), dests[i], boxes[i]);
  }
  /// CorrelationView.cc:876-885: the tile wrapped so that GLOBAL pixel coordinates inside bbox address it.
  prerasterize_type prerasterize(BBox2i const& bbox) const {
    return prerasterize_type(correlate_tile(bbox), -bbox.min().x(), -bbox.min().y(), cols(), rows());
  }
  /// CorrelationView.h:123-133: the tile is correlated over the collared box and its centre is kept.
  template <class DestT> void rasterize(DestT const& dest, BBox2i const& bbox) const {
    BBox2i proc_bbox = bbox;
    if (m_collar_size > 0) proc_bbox.expand(m_collar_size);
    vw::rasterize(prerasterize(proc_bbox), dest, bbox);
  }
};

namespace detail {
// A plain ImageView<PixelGray<float>> is wrapped as it is (the engine then reads it in place); anything else goes through a
// lazy pixel_cast.
inline ImageViewRef<PixelGray<float>> gray_float_ref(ImageView<PixelGray<float>> const& v) { return ImageViewRef<PixelGray<float>>(v); }
template <class ViewT> ImageViewRef<PixelGray<float>> gray_float_ref(ImageViewBase<ViewT> const& v) {
  return ImageViewRef<PixelGray<float>>(pixel_cast<PixelGray<float>>(v.impl()));
}
inline ImageViewRef<uint8> mask_ref(ImageView<uint8> const& v) { return ImageViewRef<uint8>(v); }
template <class ViewT> ImageViewRef<uint8> mask_ref(ImageViewBase<ViewT> const& v) { return ImageViewRef<uint8>(pixel_cast<uint8>(v.impl())); }
}  // namespace detail

/// pyramid_correlate — the reference's argument list (CorrelationView.h:195-230), verbatim; the image / mask arguments
/// accept any view (the reference takes ImageViewRef handles, which convert from any view as well).  All four algorithms
/// are implemented (VW_CORRELATION_MGM: use_mgm at every level, _FINAL_MGM: at level 0 only).  write_debug_images is
/// accepted and ignored (the reference's debug dumps are TIFF files written next to the process).
template <class Image1T, class Image2T, class Mask1T, class Mask2T>
PyramidCorrelationView
pyramid_correlate(ImageViewBase<Image1T> const& left, ImageViewBase<Image2T> const& right,
                  ImageViewBase<Mask1T> const& left_mask, ImageViewBase<Mask2T> const& right_mask,
                  PrefilterModeType prefilter_mode, float prefilter_width,
                  BBox2i const& search_region, Vector2i const& kernel_size, CostFunctionType cost_type,
                  int corr_timeout, double seconds_per_op, float consistency_threshold, int min_consistency_level,
                  int filter_half_kernel, int32 max_pyramid_levels,
                  CorrelationAlgorithm algorithm = VW_CORRELATION_BM, int collar_size = 0,
                  SemiGlobalMatcher::SgmSubpixelMode sgm_subpixel_mode = SemiGlobalMatcher::SUBPIXEL_LC_BLEND,
                  Vector2i sgm_search_buffer = Vector2i(2, 2), size_t memory_limit_mb = 6000, int blob_filter_area = 0,
                  ImageView<PixelMask<float>>* lr_disp_diff = NULL, Vector2i const& region_ul = Vector2i(0, 0),
                  bool write_debug_images = false) {
  (void)write_debug_images;
  vwgpu_pyramid_params p = vwgpu_pyramid_params();   // lr_disp_diff is wired by the view's constructor
  p.prefilter_mode = (int)prefilter_mode; p.prefilter_width = prefilter_width;
  p.search_min_x = search_region.min().x(); p.search_min_y = search_region.min().y();
  p.search_max_x = search_region.max().x(); p.search_max_y = search_region.max().y();
  p.kernel_x = kernel_size[0]; p.kernel_y = kernel_size[1];
  p.cost_type = (int)cost_type; p.corr_timeout = corr_timeout; p.seconds_per_op = seconds_per_op;
  p.consistency_threshold = consistency_threshold; p.min_consistency_level = min_consistency_level;
  p.filter_half_kernel = filter_half_kernel; p.max_pyramid_levels = max_pyramid_levels;
  p.algorithm = (int)algorithm; p.blob_filter_area = blob_filter_area;
  p.sgm_subpixel_mode = (int)sgm_subpixel_mode; p.sgm_search_buffer_x = sgm_search_buffer[0]; p.sgm_search_buffer_y = sgm_search_buffer[1];
  p.memory_limit_mb = memory_limit_mb; p.sgm_num_threads = 1;
  return PyramidCorrelationView(detail::gray_float_ref(left.impl()), detail::gray_float_ref(right.impl()),
                                detail::mask_ref(left_mask.impl()), detail::mask_ref(right_mask.impl()), p, collar_size, lr_disp_diff, region_ul);
}

}  // namespace stereo
}  // namespace vw
#endif
