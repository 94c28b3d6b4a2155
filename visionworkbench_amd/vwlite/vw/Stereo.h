// vw/Stereo.h — vw::stereo entry points of the block-matching hot path with the reference's signatures,
// implemented on libvwgpu.so (include/vwgpu.h).  Header-only; link with -lvwgpu.
//
//   calc_disparity               src/vw/Stereo/Correlation.h:50-57   (impl Correlation.cc:330-375)
//   cross_corr_consistency_check src/vw/Stereo/Correlate.h:52-58     (impl Correlate.cc:1441-1502)
//   correlate                    legacy single-level entry, signature recovered from
//                                src/vw/Stereo/tests/TestCorrelationView.cxx:79-82,213-215
// Errors: the C ABI's status codes become the reference's exception types (src/vw/Core/Exception.h:225-253).
// Threading: one engine context per (host thread x GPU), created lazily — the reference calls these functions
// concurrently from its tile threads (src/vw/Image/ImageIO.h:228-251).
#ifndef VWLITE_STEREO_H
#define VWLITE_STEREO_H

#include <cstdlib>

#include "Image.h"
#include "vwgpu.h"

namespace vw {
namespace stereo {

enum CostFunctionType {                  // src/vw/Stereo/CostFunctions.h:143-149
  ABSOLUTE_DIFFERENCE, SQUARED_DIFFERENCE, CROSS_CORRELATION, CENSUS_TRANSFORM, TERNARY_CENSUS_TRANSFORM
};
enum PrefilterModeType { PREFILTER_NONE = 0, PREFILTER_MEANSUB = 1, PREFILTER_LOG = 2 };   // PrefilterEnum.h:24-28

namespace detail {
struct ThreadContext {
  vwgpu_ctx* ctx = nullptr;
  ~ThreadContext() { if (ctx) vwgpu_destroy(ctx); }
};
inline vwgpu_ctx* thread_context() {
  static thread_local ThreadContext tc;
  if (!tc.ctx) {
    const char* dev = std::getenv("VWGPU_DEVICE");
    int rc = vwgpu_create(&tc.ctx, dev ? std::atoi(dev) : 0);
    if (rc != VWGPU_OK)
      vw_throw(LogicErr() << "vwgpu_create failed: " << vwgpu_strerror(rc) << " (no GPU; there is no CPU fallback)");
  }
  return tc.ctx;
}
inline void check(vwgpu_ctx* ctx, int rc) {
  if (rc == VWGPU_OK) return;
  std::string msg = vwgpu_last_error(ctx);
  if (msg.empty()) msg = vwgpu_strerror(rc);
  switch (rc) {
    case VWGPU_ERR_ARGUMENT: vw_throw(ArgumentErr() << msg);
    case VWGPU_ERR_NOIMPL: vw_throw(NoImplErr() << msg);
    default: vw_throw(LogicErr() << msg);
  }
}
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

/// Legacy correlate(): prefilter -> calc_disparity over the whole left image -> optional R->L run + L/R check.
/// search_volume is a BBox2i of signed disparities [min, max); the result holds signed disparities
/// (offset by search_volume.min()), like the legacy view did.  Only the Null prefilter is wired so far.
inline ImageView<PixelMask<Vector2i>>
correlate(ImageView<PixelGray<float>> const& left, ImageView<PixelGray<float>> const& right,
          BBox2i const& search_volume, Vector2i const& kernel_size,
          CostFunctionType cost_type = ABSOLUTE_DIFFERENCE, float consistency_threshold = -1) {
  const Vector2i half(kernel_size[0] / 2, kernel_size[1] / 2);
  const Vector2i s = search_volume.size();
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
                                          -half[0] - (search_volume.max().x() - 1), -half[1] - (search_volume.max().y() - 1),
                                          W + 2 * half[0] + s[0] - 1, H + 2 * half[1] + s[1] - 1);
    ImageView<PixelMask<Vector2i>> r2l = calc_disparity(cost_type, rp2, lp2, bounding_box(rp2), s, kernel_size);
    const Vector2i lmin = search_volume.min(), rmin(-(search_volume.max().x() - 1), -(search_volume.max().y() - 1));
    for (int32 r = 0; r < H; ++r) for (int32 c = 0; c < W; ++c) { l2r(c, r).child() += lmin; r2l(c, r).child() += rmin; }
    cross_corr_consistency_check(l2r, r2l, consistency_threshold);
    return l2r;
  }
  for (int32 r = 0; r < H; ++r) for (int32 c = 0; c < W; ++c) l2r(c, r).child() += search_volume.min();
  return l2r;
}

}  // namespace stereo
}  // namespace vw
#endif
