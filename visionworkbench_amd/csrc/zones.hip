// zones.hip — see zones.h.  Pure host code (compiled with the rest of the library).
#include "zones.h"

namespace vwgpu {
namespace {

struct Extent {            // element-wise min/max of the VALID disparities inside a box
  bool any = false;
  int lo_x = 0, lo_y = 0, hi_x = 0, hi_y = 0;
  IBox as_range() const { return IBox(lo_x, lo_y, hi_x + 1, hi_y + 1); }
};

Extent measure(const int32_t* disp, int w, IBox const& b) {
  Extent e;
  for (int y = b.y0; y < b.y1; ++y) {
    const int32_t* p = disp + ((size_t)y * w + b.x0) * 3;
    for (int x = b.x0; x < b.x1; ++x, p += 3) {
      if (!p[2]) continue;
      if (!e.any) { e.any = true; e.lo_x = e.hi_x = p[0]; e.lo_y = e.hi_y = p[1]; continue; }
      if (p[0] < e.lo_x) e.lo_x = p[0];
      if (p[0] > e.hi_x) e.hi_x = p[0];
      if (p[1] < e.lo_y) e.lo_y = p[1];
      if (p[1] > e.hi_y) e.hi_y = p[1];
    }
  }
  return e;
}

bool mergeable(SearchZone const& a, SearchZone const& b) {
  return (a.region.x0 == b.region.x0 || a.region.y0 == b.region.y0) && a.range.same(b.range);
}

// Returns false when a second-chance split (depth_after_failure > 0) still does not pay (Correlation.cc:319-321).
bool split(const int32_t* disp, int w, int h, IBox const& box, int kx, int ky, int depth_after_failure,
           std::vector<SearchZone>& out) {
  // 1) too small to split: emit with the range of the 1-px expanded neighbourhood (:149-162)
  if (box.dx() * box.dy() <= 200 || box.width() < 16 || box.height() < 16) {
    IBox around = box;
    around.expand(1);
    around.clip(IBox(0, 0, w, h));
    const Extent e = measure(disp, w, around);
    if (e.any) out.push_back(SearchZone{box, e.as_range()});
    return true;
  }
  // 2) the four quadrants in the reference's order q1, q2, q3, q4 (:165-171)
  const int mx = box.x0 + box.dx() / 2, my = box.y0 + box.dy() / 2;
  const IBox quad[4] = {IBox(box.x0, box.y0, mx, my), IBox(mx, box.y0, box.x1, my),
                        IBox(box.x0, my, mx, box.y1), IBox(mx, my, box.x1, box.y1)};
  IBox qrange[4];
  int32_t split_cost = 0;
  for (int i = 0; i < 4; ++i) {
    const Extent e = measure(disp, w, quad[i]);
    if (!e.any) continue;
    qrange[i] = e.as_range();
    split_cost += qrange[i].area() * ((quad[i].dx() + kx) * (quad[i].dy() + ky));
  }
  // 3) range of the whole box = union of the quadrant ranges, built the way the reference builds it (:225-239)
  const IBox none;
  IBox whole;
  if (!qrange[0].same(none)) whole = qrange[0];
  for (int i = 1; i < 4; ++i) {
    if (!qrange[i].same(none) && whole.same(none)) whole = qrange[i];
    else whole.grow(qrange[i]);
  }
  const int32_t whole_cost = whole.area() * ((box.dx() + kx) * (box.dy() + ky));
  const bool not_worth_it = split_cost > whole_cost * 0.8;
  if (not_worth_it && depth_after_failure > 0) return false;
  if (!not_worth_it) {
    for (int i = 0; i < 4; ++i) split(disp, w, h, quad[i], kx, ky, 0, out);
    return true;
  }
  // first failure: give every quadrant one more chance (:245-318)
  std::vector<SearchZone> stuck;
  for (int i = 0; i < 4; ++i)
    if (!split(disp, w, h, quad[i], kx, ky, depth_after_failure + 1, out)) stuck.push_back(SearchZone{quad[i], qrange[i]});
  auto merged = [](SearchZone const& a, SearchZone const& b) { IBox m = a.region; m.grow(b.region); return SearchZone{m, a.range}; };
  switch (stuck.size()) {
    case 4: out.push_back(SearchZone{box, whole}); break;
    case 3:
      if (mergeable(stuck[0], stuck[1])) { out.push_back(merged(stuck[0], stuck[1])); out.push_back(stuck[2]); }
      else if (mergeable(stuck[1], stuck[2])) { out.push_back(merged(stuck[1], stuck[2])); out.push_back(stuck[0]); }
      else if (mergeable(stuck[0], stuck[2])) { out.push_back(merged(stuck[0], stuck[2])); out.push_back(stuck[1]); }
      else out.insert(out.end(), stuck.begin(), stuck.end());
      break;
    case 2:
      if (mergeable(stuck[0], stuck[1])) out.push_back(merged(stuck[0], stuck[1]));
      else out.insert(out.end(), stuck.begin(), stuck.end());
      break;
    case 1: out.push_back(stuck[0]); break;
    default: break;
  }
  return true;
}

}  // namespace

void subdivide_regions(const int32_t* disp, int w, int h, int kx, int ky, std::vector<SearchZone>& out) {
  split(disp, w, h, IBox(0, 0, w, h), kx, ky, 0, out);
}

}  // namespace vwgpu
