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

Extent unite(Extent a, Extent const& b) {
  if (!b.any) return a;
  if (!a.any) return b;
  if (b.lo_x < a.lo_x) a.lo_x = b.lo_x;
  if (b.hi_x > a.hi_x) a.hi_x = b.hi_x;
  if (b.lo_y < a.lo_y) a.lo_y = b.lo_y;
  if (b.hi_y > a.hi_y) a.hi_y = b.hi_y;
  return a;
}

bool too_small_to_split(IBox const& box) { return box.dx() * box.dy() <= 200 || box.width() < 16 || box.height() < 16; }

// The quad tree of boxes is fixed by the image size alone (midpoint splits down to the "too small" leaves); only which
// splits are ACCEPTED depends on the data.  So the disparity extents are measured once, bottom up — pixels at the leaves,
// unions above — instead of once per tree level as a literal restatement of the recursion would (a 512^2 level: 2.7 ms
// -> 0.4 ms of host time per tile).
struct Node {
  IBox box;
  Extent ext;
  Extent around;                     // leaves: extent of the box grown by 1 px (filled only from a leaf table)
  int child[4] = {-1, -1, -1, -1};   // q1, q2, q3, q4 in the reference's order (Correlation.cc:165-171); -1 on a leaf
};

// Leaf extents come either from the pixels (disp != nullptr) or from a table in leaf order (leaf, *next).
int build(std::vector<Node>& tree, const int32_t* disp, int w, IBox const& box, const LeafExtent* leaf = nullptr, size_t* next = nullptr,
          std::vector<IBox>* list = nullptr) {
  const int idx = (int)tree.size();
  tree.push_back(Node());
  tree[idx].box = box;
  if (too_small_to_split(box)) {
    if (list) list->push_back(box);
    else if (leaf) {
      const LeafExtent& e = leaf[(*next)++];
      Extent in, ar;
      in.any = e.any != 0; in.lo_x = e.lo_x; in.lo_y = e.lo_y; in.hi_x = e.hi_x; in.hi_y = e.hi_y;
      ar.any = e.any_a != 0; ar.lo_x = e.lo_xa; ar.lo_y = e.lo_ya; ar.hi_x = e.hi_xa; ar.hi_y = e.hi_ya;
      tree[idx].ext = in;
      tree[idx].around = ar;
    } else tree[idx].ext = measure(disp, w, box);
    return idx;
  }
  const int mx = box.x0 + box.dx() / 2, my = box.y0 + box.dy() / 2;
  const IBox quad[4] = {IBox(box.x0, box.y0, mx, my), IBox(mx, box.y0, box.x1, my),
                        IBox(box.x0, my, mx, box.y1), IBox(mx, my, box.x1, box.y1)};
  Extent e;
  for (int i = 0; i < 4; ++i) {
    const int c = build(tree, disp, w, quad[i], leaf, next, list);
    tree[idx].child[i] = c;
    e = unite(e, tree[c].ext);
  }
  tree[idx].ext = e;
  return idx;
}

bool mergeable(SearchZone const& a, SearchZone const& b) {
  return (a.region.x0 == b.region.x0 || a.region.y0 == b.region.y0) && a.range.same(b.range);
}

// Returns false when a second-chance split (depth_after_failure > 0) still does not pay (Correlation.cc:319-321).
bool split(std::vector<Node> const& tree, int node, const int32_t* disp, int w, int h, int kx, int ky, int depth_after_failure,
           std::vector<SearchZone>& out) {
  const IBox box = tree[node].box;
  // 1) too small to split: emit with the range of the 1-px expanded neighbourhood (:149-162)
  if (tree[node].child[0] < 0) {
    Extent e = tree[node].around;
    if (disp) {
      IBox around = box;
      around.expand(1);
      around.clip(IBox(0, 0, w, h));
      e = measure(disp, w, around);
    }
    if (e.any) out.push_back(SearchZone{box, e.as_range()});
    return true;
  }
  // 2) the four quadrants in the reference's order q1, q2, q3, q4 (:165-171)
  IBox quad[4], qrange[4];
  int32_t split_cost = 0;
  for (int i = 0; i < 4; ++i) {
    const Node& c = tree[tree[node].child[i]];
    quad[i] = c.box;
    if (!c.ext.any) continue;
    qrange[i] = c.ext.as_range();
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
    for (int i = 0; i < 4; ++i) split(tree, tree[node].child[i], disp, w, h, kx, ky, 0, out);
    return true;
  }
  // first failure: give every quadrant one more chance (:245-318)
  std::vector<SearchZone> stuck;
  for (int i = 0; i < 4; ++i)
    if (!split(tree, tree[node].child[i], disp, w, h, kx, ky, depth_after_failure + 1, out)) stuck.push_back(SearchZone{quad[i], qrange[i]});
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
  std::vector<Node> tree;
  tree.reserve(((size_t)w * h) / 48 + 16);
  build(tree, disp, w, IBox(0, 0, w, h));
  split(tree, 0, disp, w, h, kx, ky, 0, out);
}

// The tree of boxes for an image size, built once per thread (a context lives on one thread and its tiles repeat the same level
// sizes): the per-call work is filling the leaf extents, the bottom-up unions and the accept / retry / merge walk.
namespace {
struct CachedTree {
  int w = -1, h = -1;
  std::vector<Node> nodes;          // depth-first preorder: children follow their parent
  std::vector<int> leaf_nodes;      // node index of every leaf, in leaf order
  std::vector<IBox> leaves;
};
CachedTree& cached_tree(int w, int h) {
  static thread_local std::vector<CachedTree> cache;
  for (CachedTree& t : cache)
    if (t.w == w && t.h == h) return t;
  if (cache.size() >= 32) cache.erase(cache.begin());
  cache.emplace_back();
  CachedTree& t = cache.back();
  t.w = w; t.h = h;
  t.nodes.reserve(((size_t)w * h) / 48 + 16);
  build(t.nodes, nullptr, w, IBox(0, 0, w, h), nullptr, nullptr, &t.leaves);
  for (size_t i = 0; i < t.nodes.size(); ++i)
    if (t.nodes[i].child[0] < 0) t.leaf_nodes.push_back((int)i);
  return t;
}
}  // namespace

const std::vector<IBox>& cached_leaves(int w, int h) { return cached_tree(w, h).leaves; }

void enumerate_leaves(int w, int h, std::vector<IBox>& leaves) { leaves = cached_tree(w, h).leaves; }

void subdivide_regions_from_leaves(int w, int h, int kx, int ky, const LeafExtent* leaf, size_t nleaf, std::vector<SearchZone>& out) {
  CachedTree& t = cached_tree(w, h);
  if (nleaf != t.leaf_nodes.size()) return;
  for (size_t i = 0; i < nleaf; ++i) {
    const LeafExtent& e = leaf[i];
    Node& n = t.nodes[t.leaf_nodes[i]];
    n.ext.any = e.any != 0; n.ext.lo_x = e.lo_x; n.ext.lo_y = e.lo_y; n.ext.hi_x = e.hi_x; n.ext.hi_y = e.hi_y;
    n.around.any = e.any_a != 0; n.around.lo_x = e.lo_xa; n.around.lo_y = e.lo_ya; n.around.hi_x = e.hi_xa; n.around.hi_y = e.hi_ya;
  }
  for (size_t i = t.nodes.size(); i-- > 0;) {                     // unions bottom up: children have larger indices
    Node& n = t.nodes[i];
    if (n.child[0] < 0) continue;
    Extent e;
    for (int c = 0; c < 4; ++c) e = unite(e, t.nodes[n.child[c]].ext);
    n.ext = e;
  }
  split(t.nodes, 0, nullptr, w, h, kx, ky, 0, out);
}

}  // namespace vwgpu
