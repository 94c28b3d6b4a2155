// vw/FileIO.h — file-backed image views and the block writer that feeds the correlator tile by tile: the pieces of
// src/vw/Image/ImageIO.h:228-314 (block_write_image: rasterise blocks of a lazy view from a pool of threads, write each
// block where it belongs) and src/vw/FileIO/DiskImageView.h (a view whose pixels live on disk and are read per requested
// bbox) that the per-tile loop of tools/correlate.cc:240-270 relies on.  The reference reads TIFF/GDAL formats through
// DiskImageResource plug-ins; none of those libraries exist here, so the on-disk formats are the two header-plus-raster
// ones that need no library: PGM "P5" (8- or 16-bit grey, big-endian, rows top to bottom) and PFM "Pf" / "PF" (1 or 3
// float32 channels, little-endian when the scale line is negative, rows BOTTOM to top).  A 3-channel PFM is exactly a
// PixelMask<Vector2f> disparity image {dx, dy, valid}.
#ifndef VWLITE_FILEIO_H
#define VWLITE_FILEIO_H

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <string>

#include "Image.h"

namespace vw {

namespace fileio {
struct Header {
  int32 cols = 0, rows = 0, channels = 1;
  int32 bytes_per_channel = 1;     // 1 / 2 (PGM) or 4 (PFM)
  bool is_float = false, bottom_up = false, little_endian = false;
  int64 data_offset = 0;
  int64 row_bytes() const { return int64(cols) * channels * bytes_per_channel; }
};

inline Header read_header(std::string const& path) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) vw_throw(IOErr() << "DiskImageView: cannot open " << path);
  Header h;
  char magic[3] = {0, 0, 0};
  auto token = [&](char* buf, size_t n) {            // next whitespace-delimited token, '#' comments skipped
    int c = std::fgetc(f);
    for (;;) {
      while (c == ' ' || c == '\t' || c == '\n' || c == '\r') c = std::fgetc(f);
      if (c == '#') { while (c != '\n' && c != EOF) c = std::fgetc(f); continue; }
      break;
    }
    size_t i = 0;
    while (c != EOF && c != ' ' && c != '\t' && c != '\n' && c != '\r' && i + 1 < n) { buf[i++] = (char)c; c = std::fgetc(f); }
    buf[i] = 0;                                      // exactly one whitespace byte has been consumed after the token
  };
  char buf[64];
  token(magic, sizeof magic);
  token(buf, sizeof buf); h.cols = std::atoi(buf);
  token(buf, sizeof buf); h.rows = std::atoi(buf);
  token(buf, sizeof buf);
  if (!std::strcmp(magic, "P5")) {
    const int maxval = std::atoi(buf);
    h.bytes_per_channel = maxval < 256 ? 1 : 2;
  } else if (!std::strcmp(magic, "Pf") || !std::strcmp(magic, "PF")) {
    h.is_float = true; h.bytes_per_channel = 4; h.bottom_up = true;
    h.channels = magic[1] == 'F' ? 3 : 1;
    h.little_endian = std::atof(buf) < 0;
  } else {
    std::fclose(f);
    vw_throw(IOErr() << "DiskImageView: " << path << " is neither a P5 PGM nor a PFM");
  }
  h.data_offset = std::ftell(f);
  std::fclose(f);
  if (h.cols <= 0 || h.rows <= 0) vw_throw(IOErr() << "DiskImageView: bad size in " << path);
  return h;
}

// pixel <-> channel helpers for the pixel types the stereo path stores
template <class PixelT> struct Channels { static const int n = 1; };
template <class T> struct Channels<PixelMask<Vector<T, 2>>> { static const int n = 3; };
inline void put(float* dst, float v) { dst[0] = v; }
inline void put(float* dst, PixelGray<float> const& v) { dst[0] = v.v(); }
inline void put(float* dst, PixelMask<Vector2f> const& v) { dst[0] = v.child()[0]; dst[1] = v.child()[1]; dst[2] = is_valid(v) ? 1.0f : 0.0f; }
inline void get(const float* src, float& v) { v = src[0]; }
inline void get(const float* src, PixelGray<float>& v) { v = PixelGray<float>(src[0]); }
inline void get(const float* src, uint8& v) { v = (uint8)src[0]; }
inline void get(const float* src, PixelMask<Vector2f>& v) {
  v = PixelMask<Vector2f>(Vector2f(src[0], src[1]));
  if (src[2] == 0.0f) v.invalidate();
}
}  // namespace fileio

/// A view over a PGM / PFM file: nothing is read until a bbox is rasterised, then exactly those rows and columns
/// (pread, so any number of tile threads may pull from the same view).
template <class PixelT>
class DiskImageView : public ImageViewBase<DiskImageView<PixelT>> {
  struct File {
    int fd = -1;
    fileio::Header h;
    ~File() { if (fd >= 0) ::close(fd); }
  };
  std::shared_ptr<File> m_file;
public:
  typedef PixelT pixel_type;
  typedef PixelT result_type;
  typedef ImageView<PixelT> prerasterize_type;
  explicit DiskImageView(std::string const& path) : m_file(new File) {
    m_file->h = fileio::read_header(path);
    VW_ASSERT(m_file->h.channels == fileio::Channels<PixelT>::n || m_file->h.channels == 1,
              IOErr() << "DiskImageView: " << path << " has " << m_file->h.channels << " channels");
    m_file->fd = ::open(path.c_str(), O_RDONLY);
    if (m_file->fd < 0) vw_throw(IOErr() << "DiskImageView: cannot open " << path);
  }
  int32 cols() const { return m_file->h.cols; }
  int32 rows() const { return m_file->h.rows; }
  int32 planes() const { return 1; }
  result_type operator()(int32 c, int32 r) const { return prerasterize(BBox2i(c, r, 1, 1))(0, 0); }
  prerasterize_type prerasterize(BBox2i const& b) const { ImageView<PixelT> o(b.width(), b.height()); rasterize(o, b); return o; }
  template <class DestT> void rasterize(DestT const& dest, BBox2i const& bbox) const {
    fileio::Header const& h = m_file->h;
    VW_ASSERT(bbox.min().x() >= 0 && bbox.min().y() >= 0 && bbox.max().x() <= h.cols && bbox.max().y() <= h.rows,
              ArgumentErr() << "DiskImageView: bbox outside the image");
    const int32 w = bbox.width(), nch = h.channels;
    std::vector<unsigned char> raw((size_t)w * nch * h.bytes_per_channel);
    std::vector<float> rowf((size_t)w * nch);
    for (int32 r = 0; r < bbox.height(); ++r) {
      const int32 y = bbox.min().y() + r, frow = h.bottom_up ? h.rows - 1 - y : y;
      const int64 off = h.data_offset + int64(frow) * h.row_bytes() + int64(bbox.min().x()) * nch * h.bytes_per_channel;
      size_t got = 0;
      while (got < raw.size()) {
        const ssize_t n = ::pread(m_file->fd, raw.data() + got, raw.size() - got, off + (int64)got);
        if (n <= 0) vw_throw(IOErr() << "DiskImageView: short read");
        got += (size_t)n;
      }
      for (int32 i = 0; i < w * nch; ++i) {
        const unsigned char* p = raw.data() + (size_t)i * h.bytes_per_channel;
        if (h.is_float) {
          unsigned char b[4] = {p[0], p[1], p[2], p[3]};
          if (!h.little_endian) { b[0] = p[3]; b[1] = p[2]; b[2] = p[1]; b[3] = p[0]; }
          std::memcpy(&rowf[i], b, 4);
        } else rowf[i] = h.bytes_per_channel == 1 ? (float)p[0] : (float)((p[0] << 8) | p[1]);
      }
      for (int32 c = 0; c < w; ++c) {
        PixelT px;
        fileio::get(&rowf[(size_t)c * nch], px);
        dest(c, r) = px;
      }
    }
  }
};

namespace fileio {
/// Creates the file with its PFM header and full size; rows are then written in place by block.
template <class PixelT>
class PfmWriter {
  int m_fd = -1;
  int32 m_cols, m_rows;
  int64 m_off = 0;
public:
  PfmWriter(std::string const& path, int32 cols, int32 rows) : m_cols(cols), m_rows(rows) {
    char head[64];
    const int n = std::snprintf(head, sizeof head, "%s\n%d %d\n-1.0\n", Channels<PixelT>::n == 3 ? "PF" : "Pf", cols, rows);
    m_fd = ::open(path.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
    if (m_fd < 0) vw_throw(IOErr() << "block_write_image: cannot create " << path);
    if (::pwrite(m_fd, head, n, 0) != n) vw_throw(IOErr() << "block_write_image: header write failed");
    m_off = n;
    if (::ftruncate(m_fd, m_off + int64(cols) * rows * Channels<PixelT>::n * 4) != 0) vw_throw(IOErr() << "block_write_image: cannot size the file");
  }
  ~PfmWriter() { if (m_fd >= 0) ::close(m_fd); }
  void write(ImageView<PixelT> const& block, BBox2i const& bbox) const {      // thread safe (pwrite at disjoint offsets)
    const int nch = Channels<PixelT>::n;
    std::vector<float> row((size_t)bbox.width() * nch);
    for (int32 r = 0; r < bbox.height(); ++r) {
      for (int32 c = 0; c < bbox.width(); ++c) put(&row[(size_t)c * nch], block(c, r));
      const int64 frow = m_rows - 1 - (bbox.min().y() + r);
      const int64 off = m_off + (frow * m_cols + bbox.min().x()) * nch * 4;
      const size_t bytes = row.size() * 4;
      if (::pwrite(m_fd, row.data(), bytes, off) != (ssize_t)bytes) vw_throw(IOErr() << "block_write_image: write failed");
    }
  }
};
}  // namespace fileio

/// write_image: the whole view as one raster (PFM).
template <class ViewT>
void write_image(std::string const& path, ImageViewBase<ViewT> const& view) {
  typedef typename ViewT::pixel_type pixel_type;
  const BBox2i all(0, 0, view.impl().cols(), view.impl().rows());
  ImageView<pixel_type> img(all.width(), all.height());
  view.impl().rasterize(img, all);
  fileio::PfmWriter<pixel_type>(path, all.width(), all.height()).write(img, all);
}

/// block_write_image (ImageIO.h:257-314): the view is rasterised block by block — blocks aligned to multiples of block_size,
/// handed out in raster order to num_threads workers — and every block is written where it belongs as soon as it is done,
/// so reading (lazy sources pull their windows inside rasterize), correlation on the GPU (one engine context per worker
/// thread) and writing of different tiles overlap.  The whole image never exists in memory.
namespace fileio {
// Views that can rasterise a run of blocks in one go (PyramidCorrelationView::rasterize_group: the blocks go through the engine's pyramid
// level loop as one group) say so with group_size(); every other view is rasterised block by block.
template <class V> auto view_group_size(V const& v, int) -> decltype(v.group_size()) { return v.group_size(); }
template <class V> int32 view_group_size(V const&, long) { return 1; }
template <class V, class T> auto rasterize_blocks(V const& v, std::vector<ImageView<T>> const& tiles, std::vector<BBox2i> const& blocks, int)
    -> decltype(v.rasterize_group(tiles, blocks)) { v.rasterize_group(tiles, blocks); }
template <class V, class T> void rasterize_blocks(V const& v, std::vector<ImageView<T>> const& tiles, std::vector<BBox2i> const& blocks, long) {
  for (size_t i = 0; i < blocks.size(); ++i) v.rasterize(tiles[i], blocks[i]);
}
}  // namespace fileio

template <class ViewT>
void block_write_image(std::string const& path, ImageViewBase<ViewT> const& view,
                       Vector2i block_size = Vector2i(1024, 1024), int32 num_threads = 0) {
  typedef typename ViewT::pixel_type pixel_type;
  ViewT const& v = view.impl();
  const int32 W = v.cols(), H = v.rows();
  if (block_size.x() <= 0 || block_size.y() <= 0) block_size = Vector2i(1024, 1024);
  if (num_threads <= 0) {
    const unsigned hc = std::thread::hardware_concurrency();
    num_threads = (int32)(hc == 0 ? 1 : (hc > 8 ? 8 : hc));
  }
  fileio::PfmWriter<pixel_type> writer(path, W, H);
  const int32 nbx = (W + block_size.x() - 1) / block_size.x(), nby = (H + block_size.y() - 1) / block_size.y();
  std::atomic<int32> next(0);
  std::exception_ptr error;
  std::mutex error_mutex;
  // a worker takes a run of up to `group` blocks of one block row at a time (1 for views without rasterize_group)
  const int32 group = std::max<int32>(1, std::min<int32>(fileio::view_group_size(v, 0), std::max<int32>(1, (nbx * nby + num_threads - 1) / num_threads)));
  const int32 runs_per_row = (nbx + group - 1) / group;
  auto worker = [&](int32 index) {
    engine::thread_worker_index() = index;        // worker w -> GPU w % ndev (vw/Engine.h)
    try {
      for (;;) {
        const int32 i = next.fetch_add(1);
        if (i >= runs_per_row * nby) return;
        const int32 row = i / runs_per_row, b0 = (i % runs_per_row) * group, b1 = std::min(nbx, b0 + group);
        std::vector<BBox2i> blocks;
        std::vector<ImageView<pixel_type>> tiles;
        for (int32 b = b0; b < b1; ++b) {
          BBox2i block(b * block_size.x(), row * block_size.y(), block_size.x(), block_size.y());
          block.crop(BBox2i(0, 0, W, H));
          blocks.push_back(block);
          tiles.push_back(ImageView<pixel_type>(block.width(), block.height()));
        }
        fileio::rasterize_blocks(v, tiles, blocks, 0);
        for (size_t k = 0; k < blocks.size(); ++k) writer.write(tiles[k], blocks[k]);
      }
    } catch (...) {
      std::lock_guard<std::mutex> lock(error_mutex);
      if (!error) error = std::current_exception();
      next.store(runs_per_row * nby);
    }
  };
  const int32 nt = std::min<int32>(num_threads, runs_per_row * nby);
  if (nt <= 1) worker(engine::thread_worker_index());
  else {
    std::vector<std::thread> pool;
    for (int32 t = 0; t < nt; ++t) pool.emplace_back(worker, t);
    for (std::thread& t : pool) t.join();
  }
  if (error) std::rethrow_exception(error);
}

}  // namespace vw
#endif
