// vw/Filter.h — the convolution entry points of vw/Image/Filter.h that the stereo pyramid uses, on libvwgpu.so.
//
//   generate_gaussian_kernel          src/vw/Image/Filter.tcc:37-78 (+ compute_kernel_size, Filter.cc:32-37)
//   generate_pyramid_smoothing_kernel src/vw/Image/Filter.h:89-99
//   separable_convolution_filter      src/vw/Image/Filter.h:156-191   (SeparableConvolutionView, Convolution.h:275-328)
//   convolution_filter                src/vw/Image/Filter.h:107-140   (ConvolutionView, Convolution.h:105-170)
//   gaussian_filter                   src/vw/Image/Filter.h:205-258
//   laplacian_filter                  src/vw/Image/Filter.h:320-335
//   subsample                         src/vw/Image/Manipulation.h:214-311
// The reference returns lazy views; these return the rasterised ImageView (what assigning the view yields).
// Only single-channel float pixels (PixelGray<float> / float) are wired — the stereo path's pixel type.
#ifndef VWLITE_FILTER_H
#define VWLITE_FILTER_H

#include <vector>

#include "Engine.h"
#include "Image.h"

namespace vw {

template <class KernelT>
inline void generate_gaussian_kernel(std::vector<KernelT>& kernel, double sigma, int32 size = 0) {
  std::vector<float> taps(size > 0 ? size : 4096);
  const int n = vwgpu_generate_gaussian_kernel(sigma, size, taps.data(), (int)taps.size());
  VW_ASSERT(n >= 0, ArgumentErr() << "generate_gaussian_kernel: bad arguments");
  kernel.assign(taps.begin(), taps.begin() + n);
}

inline std::vector<float> generate_pyramid_smoothing_kernel() {
  return std::vector<float>{1.0f / 16, 4.0f / 16, 6.0f / 16, 4.0f / 16, 1.0f / 16};
}

namespace detail {
template <class P> struct is_float_pixel { static const bool value = false; };
template <> struct is_float_pixel<float> { static const bool value = true; };
template <> struct is_float_pixel<PixelGray<float>> { static const bool value = true; };
inline int edge_code(ConstantEdgeExtension) { return VWGPU_EDGE_CONSTANT; }
inline int edge_code(ZeroEdgeExtension) { return VWGPU_EDGE_ZERO; }

template <class PixelT>
ImageView<PixelT> separable(ImageView<PixelT> const& src, std::vector<float> const& xk, int cx,
                            std::vector<float> const& yk, int cy, int edge, int step) {
  static_assert(is_float_pixel<PixelT>::value, "vwlite filters are wired for single-channel float pixels");
  const int32 w = src.cols(), h = src.rows();
  ImageView<PixelT> out(w > 0 ? 1 + (w - 1) / step : 0, h > 0 ? 1 + (h - 1) / step : 0);
  if (w == 0 || h == 0) return out;
  vwgpu_ctx* ctx = engine::thread_context();
  engine::check(ctx, vwgpu_separable_convolution(ctx, reinterpret_cast<const float*>(src.data()), w, h, 0,
                                                 xk.data(), (int)xk.size(), cx, yk.data(), (int)yk.size(), cy, edge, step,
                                                 reinterpret_cast<float*>(out.data()), 0));
  return out;
}
}  // namespace detail

template <class ViewT, class EdgeT>
ImageView<typename ViewT::pixel_type>
separable_convolution_filter(ImageViewBase<ViewT> const& src, std::vector<float> const& x_kernel, std::vector<float> const& y_kernel,
                             int32 cx, int32 cy, EdgeT edge) {
  ImageView<typename ViewT::pixel_type> in = src.impl();
  return detail::separable(in, x_kernel, cx, y_kernel, cy, detail::edge_code(edge), 1);
}
template <class ViewT, class EdgeT>
ImageView<typename ViewT::pixel_type>
separable_convolution_filter(ImageViewBase<ViewT> const& src, std::vector<float> const& x_kernel, std::vector<float> const& y_kernel, EdgeT edge) {
  // default origins: the kernel centre (Filter.h:174-177)
  return separable_convolution_filter(src, x_kernel, y_kernel, (int32)((x_kernel.size() - 1) / 2) * (x_kernel.empty() ? 0 : 1),
                                      (int32)((y_kernel.size() - 1) / 2) * (y_kernel.empty() ? 0 : 1), edge);
}
template <class ViewT>
ImageView<typename ViewT::pixel_type>
separable_convolution_filter(ImageViewBase<ViewT> const& src, std::vector<float> const& x_kernel, std::vector<float> const& y_kernel) {
  return separable_convolution_filter(src, x_kernel, y_kernel, ConstantEdgeExtension());
}

// kernel: a small ImageView<float> (cols x rows), origin (ci, cj)
template <class ViewT, class EdgeT>
ImageView<typename ViewT::pixel_type>
convolution_filter(ImageViewBase<ViewT> const& src, ImageView<float> const& kernel, int32 ci, int32 cj, EdgeT edge) {
  static_assert(detail::is_float_pixel<typename ViewT::pixel_type>::value, "vwlite filters are wired for single-channel float pixels");
  ImageView<typename ViewT::pixel_type> in = src.impl();
  ImageView<typename ViewT::pixel_type> out(in.cols(), in.rows());
  if (in.cols() == 0 || in.rows() == 0) return out;
  vwgpu_ctx* ctx = engine::thread_context();
  engine::check(ctx, vwgpu_convolution_2d(ctx, reinterpret_cast<const float*>(in.data()), in.cols(), in.rows(), 0,
                                          kernel.data(), kernel.cols(), kernel.rows(), ci, cj, detail::edge_code(edge),
                                          reinterpret_cast<float*>(out.data()), 0));
  return out;
}

template <class ViewT, class EdgeT>
ImageView<typename ViewT::pixel_type>
gaussian_filter(ImageViewBase<ViewT> const& src, double x_sigma, double y_sigma, int32 x_dim, int32 y_dim, EdgeT edge) {
  std::vector<float> xk, yk;
  generate_gaussian_kernel(xk, x_sigma, x_dim);
  generate_gaussian_kernel(yk, y_sigma, y_dim);
  return separable_convolution_filter(src, xk, yk, edge);
}
template <class ViewT>
ImageView<typename ViewT::pixel_type> gaussian_filter(ImageViewBase<ViewT> const& src, double sigma) {
  return gaussian_filter(src, sigma, sigma, 0, 0, ConstantEdgeExtension());
}
template <class ViewT>
ImageView<typename ViewT::pixel_type> gaussian_filter(ImageViewBase<ViewT> const& src, double x_sigma, double y_sigma) {
  return gaussian_filter(src, x_sigma, y_sigma, 0, 0, ConstantEdgeExtension());
}

template <class ViewT, class EdgeT>
ImageView<typename ViewT::pixel_type> laplacian_filter(ImageViewBase<ViewT> const& src, EdgeT edge) {
  ImageView<float> k(3, 3);
  const float taps[9] = {0, 1, 0, 1, -4, 1, 0, 1, 0};
  for (int i = 0; i < 9; ++i) k.data()[i] = taps[i];
  return convolution_filter(src, k, 1, 1, edge);
}
template <class ViewT>
ImageView<typename ViewT::pixel_type> laplacian_filter(ImageViewBase<ViewT> const& src) {
  return laplacian_filter(src, ConstantEdgeExtension());
}

// subsample(view, s): pixel (s*i, s*j); size 1 + (N-1)/s (Manipulation.h:233-240)
template <class ViewT>
ImageView<typename ViewT::pixel_type> subsample(ImageViewBase<ViewT> const& v, int32 s) {
  VW_ASSERT(s >= 1, ArgumentErr() << "SubsampleView: Arguments must be greater than zero.");
  const int32 w = v.impl().cols(), h = v.impl().rows();
  ImageView<typename ViewT::pixel_type> out(w > 0 ? 1 + (w - 1) / s : 0, h > 0 ? 1 + (h - 1) / s : 0);
  for (int32 r = 0; r < out.rows(); ++r)
    for (int32 c = 0; c < out.cols(); ++c) out(c, r) = v.impl()(c * s, r * s);
  return out;
}

}  // namespace vw
#endif
