// vw/Image.h — the operator surface of Vision Workbench's Image module that the stereo hot path uses,
// boost-free and header-only: ImageView<T> storage, the CRTP lazy-view protocol, crop / edge_extend views,
// ImageViewRef<T> type erasure, PixelGray<T> and PixelMask<T>.
//
// Protocol kept from the reference:
//   every view derives from ImageViewBase<Impl> (src/vw/Image/ImageViewBase.h:57-122) and offers cols/rows/planes,
//   operator()(col,row), prerasterize(bbox) and rasterize(dest,bbox); nothing is computed until a view is assigned
//   to an ImageView<T> (src/vw/Image/ImageView.h:113-119,136-141); ImageViewRef<T> erases the view type behind
//   virtual calls (src/vw/Image/ImageViewRef.h:189-267).
// Layouts are the reference's: ImageView is row-major contiguous and zero-initialised (ImageView.h:209-239),
// PixelMask<Vector2i> is {int32,int32,int32 valid in {0,INT32_MAX}} (src/vw/Image/PixelMask.h:48-120).
#ifndef VWLITE_IMAGE_H
#define VWLITE_IMAGE_H

#include <atomic>
#include <cmath>
#include <exception>
#include <limits>
#include <memory>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

#include "Math.h"

namespace vw {

// ---- pixel types -------------------------------------------------------------------------------------------
template <class ChannelT>
class PixelGray {
  ChannelT m_v;
public:
  typedef ChannelT channel_type;
  PixelGray() : m_v() {}
  PixelGray(ChannelT v) : m_v(v) {}
  template <class U> explicit PixelGray(PixelGray<U> const& o) : m_v(ChannelT(o.v())) {}
  ChannelT& v() { return m_v; }  ChannelT const& v() const { return m_v; }
  operator ChannelT() const { return m_v; }
  ChannelT& operator[](int) { return m_v; }  ChannelT const& operator[](int) const { return m_v; }
};

template <class T> struct CompoundChannelType { typedef T type; };
template <class T> struct CompoundChannelType<PixelGray<T>> { typedef T type; };
template <class T, int N> struct CompoundChannelType<Vector<T, N>> { typedef T type; };

// ChannelRange (src/vw/Image/PixelTypeInfo.h:85-118): integer max = numeric max, float max = 1.0.
template <class T> struct ChannelRange {
  static T max() { return std::is_floating_point<T>::value ? T(1) : std::numeric_limits<T>::max(); }
  static T min() { return T(); }
};

template <class ChildT>
class PixelMask {
public:
  typedef typename CompoundChannelType<ChildT>::type channel_type;
private:
  ChildT m_child;
  channel_type m_valid;
public:
  PixelMask() : m_child(), m_valid(ChannelRange<channel_type>::min()) {}          // invalid by default (:58-61)
  PixelMask(ChildT const& c) : m_child(c), m_valid(ChannelRange<channel_type>::max()) {}
  PixelMask(channel_type a0, channel_type a1) : m_valid(ChannelRange<channel_type>::max()) { m_child[0] = a0; m_child[1] = a1; }
  template <class U> PixelMask(PixelMask<U> const& o)
      : m_child(ChildT(o.child())), m_valid(o.valid() ? ChannelRange<channel_type>::max() : ChannelRange<channel_type>::min()) {}
  channel_type valid() const { return m_valid; }
  void invalidate() { m_valid = ChannelRange<channel_type>::min(); }
  void validate() { m_valid = ChannelRange<channel_type>::max(); }
  ChildT& child() { return m_child; }  ChildT const& child() const { return m_child; }
  channel_type& operator[](int i) { return m_child[i]; }
  channel_type const& operator[](int i) const { return m_child[i]; }
};
template <class T> bool is_valid(PixelMask<T> const& p) { return p.valid() != 0; }
template <class T> bool is_valid(T const&) { return true; }
template <class T> void invalidate(PixelMask<T>& p) { p.invalidate(); }
template <class T> void validate(PixelMask<T>& p) { p.validate(); }
static_assert(sizeof(PixelMask<Vector2i>) == 12, "PixelMask<Vector2i> must keep the reference's 12-byte layout");
static_assert(sizeof(PixelMask<Vector2f>) == 12, "PixelMask<Vector2f> must keep the reference's 12-byte layout");
static_assert(sizeof(PixelGray<float>) == 4, "PixelGray<float> is one float");

// ---- view protocol -----------------------------------------------------------------------------------------
template <class ImplT>
struct ImageViewBase {
  ImplT& impl() { return static_cast<ImplT&>(*this); }
  ImplT const& impl() const { return static_cast<ImplT const&>(*this); }
  int32 get_cols() const { return impl().cols(); }
  int32 get_rows() const { return impl().rows(); }
};

template <class ViewT> BBox2i bounding_box(ImageViewBase<ViewT> const& v) {
  return BBox2i(0, 0, v.impl().cols(), v.impl().rows());
}

// Default rasterisation: walk the source pixel by pixel (src/vw/Image/ImageViewBase.h:282-316).
template <class SrcT, class DestT>
void rasterize(SrcT const& src, DestT const& dest, BBox2i const& bbox) {
  for (int32 r = 0; r < bbox.height(); ++r)
    for (int32 c = 0; c < bbox.width(); ++c)
      dest(c, r) = typename DestT::pixel_type(src(c + bbox.min().x(), r + bbox.min().y()));
}

template <class PixelT>
class ImageView : public ImageViewBase<ImageView<PixelT>> {
  std::shared_ptr<std::vector<PixelT>> m_data;   // shared, shallow-copied by value (ImageView.h:72-75,98-103)
  int32 m_cols, m_rows;
public:
  typedef PixelT pixel_type;
  typedef ImageView prerasterize_type;
  ImageView() : m_cols(0), m_rows(0) {}
  ImageView(int32 cols, int32 rows) { set_size(cols, rows); }
  template <class ViewT> ImageView(ImageViewBase<ViewT> const& view) {          // rasterising ctor (:113-119)
    set_size(view.impl().cols(), view.impl().rows());
    view.impl().rasterize(*this, BBox2i(0, 0, m_cols, m_rows));
  }
  template <class ViewT> ImageView& operator=(ImageViewBase<ViewT> const& view) {  // (:136-141)
    set_size(view.impl().cols(), view.impl().rows());
    view.impl().rasterize(*this, BBox2i(0, 0, m_cols, m_rows));
    return *this;
  }
  void set_size(int32 cols, int32 rows) {
    VW_ASSERT(cols >= 0 && rows >= 0, ArgumentErr() << "Cannot allocate image with negative pixel count.");
    m_cols = cols; m_rows = rows;
    m_data = std::make_shared<std::vector<PixelT>>((size_t)cols * rows);          // zero / default initialised
  }
  int32 cols() const { return m_cols; }
  int32 rows() const { return m_rows; }
  int32 planes() const { return 1; }
  PixelT* data() const { return m_data ? m_data->data() : nullptr; }
  PixelT& operator()(int32 c, int32 r) const { return (*m_data)[(size_t)r * m_cols + c]; }
  prerasterize_type prerasterize(BBox2i const&) const { return *this; }
  template <class DestT> void rasterize(DestT const& dest, BBox2i const& bbox) const { vw::rasterize(*this, dest, bbox); }
};

// crop(view, bbox): lazy window; negative / out-of-range origins are legal when the child is edge-extended
// (src/vw/Image/Manipulation.h:82-147).
template <class ChildT>
class CropView : public ImageViewBase<CropView<ChildT>> {
  ChildT m_child; int32 m_x, m_y, m_w, m_h;
public:
  typedef typename ChildT::pixel_type pixel_type;
  typedef ImageView<pixel_type> prerasterize_type;
  CropView(ChildT const& c, int32 x, int32 y, int32 w, int32 h) : m_child(c), m_x(x), m_y(y), m_w(w), m_h(h) {}
  int32 cols() const { return m_w; }  int32 rows() const { return m_h; }  int32 planes() const { return 1; }
  // l-value when the child yields references (crop(image, ...) = ... ; src/vw/Image/Manipulation.h:122-133)
  decltype(auto) operator()(int32 c, int32 r) const { return m_child(c + m_x, r + m_y); }
  prerasterize_type prerasterize(BBox2i const& b) const { ImageView<pixel_type> o(b.width(), b.height()); rasterize(o, b); return o; }
  template <class DestT> void rasterize(DestT const& dest, BBox2i const& bbox) const { vw::rasterize(*this, dest, bbox); }
};
template <class ViewT> CropView<ViewT> crop(ImageViewBase<ViewT> const& v, int32 x, int32 y, int32 w, int32 h) {
  return CropView<ViewT>(v.impl(), x, y, w, h);
}
template <class ViewT> CropView<ViewT> crop(ImageViewBase<ViewT> const& v, BBox2i const& b) {
  return CropView<ViewT>(v.impl(), b.min().x(), b.min().y(), b.width(), b.height());
}

// Edge extension (src/vw/Image/EdgeExtension.h): Constant = clamp to the nearest edge pixel, Zero = 0 outside.
struct ConstantEdgeExtension {
  template <class V> typename V::pixel_type operator()(V const& v, int32 c, int32 r) const {
    c = c < 0 ? 0 : (c >= v.cols() ? v.cols() - 1 : c);
    r = r < 0 ? 0 : (r >= v.rows() ? v.rows() - 1 : r);
    return v(c, r);
  }
};
struct ZeroEdgeExtension {
  template <class V> typename V::pixel_type operator()(V const& v, int32 c, int32 r) const {
    if (c < 0 || r < 0 || c >= v.cols() || r >= v.rows()) return typename V::pixel_type();
    return v(c, r);
  }
};
template <class ChildT, class ExtT>
class EdgeExtensionView : public ImageViewBase<EdgeExtensionView<ChildT, ExtT>> {
  ChildT m_child; ExtT m_ext;
public:
  typedef typename ChildT::pixel_type pixel_type;
  typedef ImageView<pixel_type> prerasterize_type;
  EdgeExtensionView(ChildT const& c, ExtT e = ExtT()) : m_child(c), m_ext(e) {}
  int32 cols() const { return m_child.cols(); }  int32 rows() const { return m_child.rows(); }  int32 planes() const { return 1; }
  pixel_type operator()(int32 c, int32 r) const { return m_ext(m_child, c, r); }
  template <class DestT> void rasterize(DestT const& dest, BBox2i const& bbox) const { vw::rasterize(*this, dest, bbox); }
};
template <class ViewT, class ExtT> EdgeExtensionView<ViewT, ExtT> edge_extend(ImageViewBase<ViewT> const& v, ExtT e) {
  return EdgeExtensionView<ViewT, ExtT>(v.impl(), e);
}
template <class ViewT> EdgeExtensionView<ViewT, ConstantEdgeExtension> edge_extend(ImageViewBase<ViewT> const& v) {
  return EdgeExtensionView<ViewT, ConstantEdgeExtension>(v.impl());
}

// constant_view(value, cols, rows) / constant_view(value, like_view): a view with the same value everywhere
// (src/vw/Image/Algorithms.h ConstantView) — e.g. an all-valid mask without allocating one.
template <class PixelT>
class ConstantView : public ImageViewBase<ConstantView<PixelT>> {
  PixelT m_value;
  int32 m_cols, m_rows;
public:
  typedef PixelT pixel_type;
  typedef PixelT result_type;
  typedef ConstantView prerasterize_type;
  ConstantView(PixelT const& value, int32 cols, int32 rows) : m_value(value), m_cols(cols), m_rows(rows) {}
  int32 cols() const { return m_cols; }  int32 rows() const { return m_rows; }  int32 planes() const { return 1; }
  result_type operator()(int32, int32) const { return m_value; }
  prerasterize_type prerasterize(BBox2i const&) const { return *this; }
  template <class DestT> void rasterize(DestT const& dest, BBox2i const& bbox) const {
    for (int32 r = 0; r < bbox.height(); ++r) for (int32 c = 0; c < bbox.width(); ++c) dest(c, r) = m_value;
  }
};
template <class PixelT> ConstantView<PixelT> constant_view(PixelT const& value, int32 cols, int32 rows) { return ConstantView<PixelT>(value, cols, rows); }
template <class PixelT, class ViewT> ConstantView<PixelT> constant_view(PixelT const& value, ImageViewBase<ViewT> const& like) {
  return ConstantView<PixelT>(value, like.impl().cols(), like.impl().rows());
}

// pixel_cast<DestPixelT>(view): lazy per-pixel conversion.
template <class ChildT, class DestPixelT>
class PixelCastView : public ImageViewBase<PixelCastView<ChildT, DestPixelT>> {
  ChildT m_child;
public:
  typedef DestPixelT pixel_type;
  PixelCastView(ChildT const& c) : m_child(c) {}
  int32 cols() const { return m_child.cols(); }  int32 rows() const { return m_child.rows(); }  int32 planes() const { return 1; }
  pixel_type operator()(int32 c, int32 r) const { return pixel_type(m_child(c, r)); }
  template <class DestT> void rasterize(DestT const& dest, BBox2i const& bbox) const { vw::rasterize(*this, dest, bbox); }
};
template <class DestPixelT, class ViewT> PixelCastView<ViewT, DestPixelT> pixel_cast(ImageViewBase<ViewT> const& v) {
  return PixelCastView<ViewT, DestPixelT>(v.impl());
}

// ImageViewRef<PixelT>: type-erased view handle (src/vw/Image/ImageViewRef.h:189-267).
template <class PixelT>
class ImageViewRef : public ImageViewBase<ImageViewRef<PixelT>> {
  struct Base {
    virtual ~Base() {}
    virtual int32 cols() const = 0;
    virtual int32 rows() const = 0;
    virtual PixelT at(int32 c, int32 r) const = 0;
    virtual void raster(ImageView<PixelT> const& dest, BBox2i const& bbox) const = 0;
    virtual const PixelT* plain(int32& stride) const = 0;   // non-null when the view is a plain ImageView
  };
  template <class ViewT> struct Impl : Base {
    ViewT v;
    Impl(ViewT const& view) : v(view) {}
    int32 cols() const override { return v.cols(); }
    int32 rows() const override { return v.rows(); }
    PixelT at(int32 c, int32 r) const override { return PixelT(v(c, r)); }
    void raster(ImageView<PixelT> const& dest, BBox2i const& bbox) const override { v.rasterize(dest, bbox); }
    const PixelT* plain(int32& stride) const override { return plain_of(v, stride); }
    static const PixelT* plain_of(ImageView<PixelT> const& iv, int32& stride) { stride = iv.cols(); return iv.data(); }
    template <class Other> static const PixelT* plain_of(Other const&, int32&) { return nullptr; }
  };
  std::shared_ptr<Base> m_view;
public:
  typedef PixelT pixel_type;
  typedef ImageView<PixelT> prerasterize_type;
  ImageViewRef() {}
  template <class ViewT> ImageViewRef(ImageViewBase<ViewT> const& view) : m_view(new Impl<ViewT>(view.impl())) {}
  int32 cols() const { return m_view->cols(); }
  int32 rows() const { return m_view->rows(); }
  int32 planes() const { return 1; }
  PixelT operator()(int32 c, int32 r) const { return m_view->at(c, r); }
  prerasterize_type prerasterize(BBox2i const& b) const { ImageView<PixelT> o(b.width(), b.height()); m_view->raster(o, b); return o; }
  template <class DestT> void rasterize(DestT const& dest, BBox2i const& bbox) const {
    ImageView<PixelT> tmp(bbox.width(), bbox.height());
    m_view->raster(tmp, bbox);
    vw::rasterize(tmp, dest, BBox2i(0, 0, bbox.width(), bbox.height()));
  }
  void rasterize(ImageView<PixelT> const& dest, BBox2i const& bbox) const { m_view->raster(dest, bbox); }
  // Engine fast path: direct pointer when the handle wraps a plain ImageView (no copy).
  const PixelT* plain_data(int32& stride) const { return m_view->plain(stride); }
};

// block_rasterize(view, block_size, num_threads): rasterise the child view block by block from a pool of threads
// (src/vw/Image/BlockRasterize.h:43-176 without the cache; block walk of src/vw/Image/BlockProcessor.h:63-176: blocks are
// aligned to multiples of block_size counted from the image origin, cropped to the requested bbox, handed out in raster order).
// Each worker calls child.rasterize(crop(dest, block - offset), block); with the engine's views every worker thread owns
// its own GPU context, so several tiles are in flight on the device at once.
template <class ImageT>
class BlockRasterizeView : public ImageViewBase<BlockRasterizeView<ImageT>> {
  std::shared_ptr<ImageT> m_child;
  Vector2i m_block_size;
  int32 m_num_threads;
  static int32 round_down(int32 val, int32 mod) { return val + ((val >= 0) ? (-(val % mod)) : (((-val - 1) % mod) - mod + 1)); }
public:
  typedef typename ImageT::pixel_type pixel_type;
  typedef pixel_type result_type;
  typedef ImageView<pixel_type> prerasterize_type;
  BlockRasterizeView(ImageT const& image, Vector2i const& block_size, int32 num_threads = 0)
      : m_child(new ImageT(image)), m_block_size(block_size), m_num_threads(num_threads) {
    if (m_block_size.x() <= 0 || m_block_size.y() <= 0) m_block_size = Vector2i(1024, 1024);
    if (m_num_threads <= 0) {
      const unsigned hc = std::thread::hardware_concurrency();
      m_num_threads = (int32)(hc == 0 ? 1 : (hc > 8 ? 8 : hc));
    }
  }
  int32 cols() const { return m_child->cols(); }
  int32 rows() const { return m_child->rows(); }
  int32 planes() const { return 1; }
  ImageT const& child() const { return *m_child; }
  result_type operator()(int32 x, int32 y) const { return (*m_child)(x, y); }
  prerasterize_type prerasterize(BBox2i const& b) const { ImageView<pixel_type> o(b.width(), b.height()); rasterize(o, b); return o; }
  template <class DestT> void rasterize(DestT const& dest, BBox2i const& bbox) const {
    if (bbox.empty()) return;
    const int32 bx0 = round_down(bbox.min().x(), m_block_size.x()), by0 = round_down(bbox.min().y(), m_block_size.y());
    const int32 nbx = (bbox.max().x() - bx0 + m_block_size.x() - 1) / m_block_size.x();
    const int32 nby = (bbox.max().y() - by0 + m_block_size.y() - 1) / m_block_size.y();
    std::atomic<int32> next(0);
    std::exception_ptr error;
    std::mutex error_mutex;
    auto worker = [&](int32 index) {
      engine::thread_worker_index() = index;      // the engine wrappers bind worker w to GPU w % ndev (vw/Engine.h)
      try {
        for (;;) {
          const int32 i = next.fetch_add(1);
          if (i >= nbx * nby) return;
          BBox2i block(bx0 + (i % nbx) * m_block_size.x(), by0 + (i / nbx) * m_block_size.y(), m_block_size.x(), m_block_size.y());
          block.crop(bbox);
          if (block.empty()) continue;
          m_child->rasterize(crop(dest, block - bbox.min()), block);
        }
      } catch (...) {
        std::lock_guard<std::mutex> lock(error_mutex);
        if (!error) error = std::current_exception();
        next.store(nbx * nby);
      }
    };
    const int32 nt = std::min<int32>(m_num_threads, nbx * nby);
    if (nt <= 1) worker(engine::thread_worker_index());
    else {
      std::vector<std::thread> pool;
      for (int32 t = 0; t < nt; ++t) pool.emplace_back(worker, t);
      for (std::thread& t : pool) t.join();
    }
    if (error) std::rethrow_exception(error);
  }
};
template <class ImageT>
BlockRasterizeView<ImageT> block_rasterize(ImageViewBase<ImageT> const& image, Vector2i const& block_size, int32 num_threads = 0) {
  return BlockRasterizeView<ImageT>(image.impl(), block_size, num_threads);
}

template <class ViewT, class ValT> void fill(ImageViewBase<ViewT> const& v, ValT const& val) {
  for (int32 r = 0; r < v.impl().rows(); ++r)
    for (int32 c = 0; c < v.impl().cols(); ++c) v.impl()(c, r) = typename ViewT::pixel_type(val);
}
template <class ViewT> ImageView<typename ViewT::pixel_type> copy(ImageViewBase<ViewT> const& v) {
  ImageView<typename ViewT::pixel_type> o(v.impl().cols(), v.impl().rows());
  v.impl().rasterize(o, BBox2i(0, 0, o.cols(), o.rows()));
  return o;
}

}  // namespace vw
#endif
