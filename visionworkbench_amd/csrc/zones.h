// zones.h — host-side zone scheduler of the pyramid correlator: quad-tree split of a disparity image into regions
// with tight disparity ranges.  Replaces vw::stereo::subdivide_regions (src/vw/Stereo/Correlation.cc:139-328) and
// vw::stereo::SearchParam (src/vw/Stereo/Correlation.h:66-91).
#pragma once
#include <cstdint>
#include <limits>
#include <vector>

namespace vwgpu {

// Half-open integer box with vw::BBox2i semantics (src/vw/Math/BBox.tcc): an empty box has zero width/height/area,
// and expand / scale leave an empty box untouched.
struct IBox {
  int x0, y0, x1, y1;
  IBox() { const int big = std::numeric_limits<int32_t>::max() - 1; x0 = y0 = big; x1 = y1 = -big; }
  IBox(int ax0, int ay0, int ax1, int ay1) : x0(ax0), y0(ay0), x1(ax1), y1(ay1) {}
  bool empty() const { return x0 >= x1 || y0 >= y1; }
  int dx() const { return x1 - x0; }
  int dy() const { return y1 - y0; }
  int width() const { return empty() ? 0 : x1 - x0; }
  int height() const { return empty() ? 0 : y1 - y0; }
  int area() const { return empty() ? 0 : (x1 - x0) * (y1 - y0); }
  void include(int x, int y) { if (x > x1) x1 = x; if (x < x0) x0 = x; if (y > y1) y1 = y; if (y < y0) y0 = y; }
  void grow(IBox const& b) { if (b.empty()) return; include(b.x0, b.y0); include(b.x1, b.y1); }
  void clip(IBox const& b) { if (x0 < b.x0) x0 = b.x0; if (x1 > b.x1) x1 = b.x1; if (y0 < b.y0) y0 = b.y0; if (y1 > b.y1) y1 = b.y1; }
  void expand(int n) { if (empty()) return; x0 -= n; y0 -= n; x1 += n; y1 += n; }
  void scale(int s) { if (empty()) return; x0 *= s; y0 *= s; x1 *= s; y1 *= s; }
  bool same(IBox const& o) const { return x0 == o.x0 && y0 == o.y0 && x1 == o.x1 && y1 == o.y1; }
};

struct SearchZone {
  IBox region;   // where in the (level) image
  IBox range;    // disparity range [min, max+1)
  double volume() const { return (double)region.width() * region.height() * (double)range.width() * range.height(); }
};

// disp: w x h x {dx, dy, valid} int32 (host memory).  Appends zones to `out`.
void subdivide_regions(const int32_t* disp, int w, int h, int kx, int ky, std::vector<SearchZone>& out);

// The same scheduler with the pixel work left on the device.  The quad tree of boxes depends on the image size only, so
// its leaves can be listed up front (depth-first, the order the recursion meets them); a kernel measures, per leaf, the
// disparity extent of the box and of its 1-px neighbourhood (zone_extent_kernel, bm_zones.hip), and the recursion then
// runs on that small table instead of the image.
struct LeafExtent {          // 10 ints per leaf, as the kernel writes them
  int32_t any, lo_x, lo_y, hi_x, hi_y;            // valid disparities inside the box
  int32_t any_a, lo_xa, lo_ya, hi_xa, hi_ya;      // ... inside the box grown by 1 px (clipped to the image)
};
void enumerate_leaves(int w, int h, std::vector<IBox>& leaves);
const std::vector<IBox>& cached_leaves(int w, int h);     // the same list, kept per thread and image size
void subdivide_regions_from_leaves(int w, int h, int kx, int ky, const LeafExtent* leaf, size_t nleaf, std::vector<SearchZone>& out);

}  // namespace vwgpu
