// mgm_schedule.h — the launch schedule of the MGM passes, shared by the kernels (sgm.hip), the launcher and the host-side inspector
// vwgpu_mgm_front_pixel (tests/test_host_logic.py checks it on the CPU).
//
// accum_mgm_multithread (src/vw/Stereo/SGM.cc:2619-2700) runs eight SmoothPathAccumTask passes (SGMAssist.h:835-1239).  In each, a
// pixel that passes the task's border test takes the mean of two evaluate_path results: from its path predecessor A = (c + ax, r + ay)
// and from a second predecessor B = (c + bx, r + by); every other pixel keeps its local costs.  The raster order of a task only has
// to respect those two dependencies, so the engine walks FRONTS — sets of pixels whose predecessors all lie in the previous front:
//   kind 0  L / R / T / B      anti-diagonals c' + r' (coordinates counted from the side the pass starts on): W + H - 1 fronts
//   kind 1  TL / BR            rows (both predecessors lie in the neighbouring row):                          H fronts
//   kind 2  BL / TR            columns:                                                                       W fronts
#pragma once
#if defined(__HIPCC__)
#define VWGPU_HD __host__ __device__ __forceinline__
#else
#define VWGPU_HD inline
#endif

namespace vwgpu {

struct MgmDir { int ax, ay, bx, by, need, kind, flipx, flipy; };
// need: the task's border test — bit 0 col > 0, bit 1 col < last, bit 2 row > 0, bit 3 row < last.
// Order: directions that share a launch sequence are interleaved so that any aligned group of 1, 2, 4 or 8 mixes both kinds.
//                                A (path)  B (second)  border test       fronts
constexpr MgmDir kMgmDirs[8] = {{-1,  0,  0, -1, 1 | 4,      0, 0, 0},      // L   SGMAssist.h:911-955
                                {-1, -1,  1, -1, 1 | 2 | 4,  1, 0, 0},      // TL  :958-996
                                { 1,  0,  0,  1, 2 | 8,      0, 1, 1},      // R   :998-1033
                                { 1,  1, -1,  1, 1 | 2 | 8,  1, 0, 1},      // BR  :1035-1071
                                { 0, -1,  1,  0, 2 | 4,      0, 1, 0},      // T   :1147-1182
                                {-1,  1, -1, -1, 1 | 4 | 8,  2, 0, 0},      // BL  :1110-1145
                                { 0,  1, -1,  0, 1 | 8,      0, 0, 1},      // B   :1073-1108
                                { 1, -1,  1,  1, 2 | 4 | 8,  2, 1, 0}};     // TR  :1184-1219

VWGPU_HD int mgm_front_count(int kind, int W, int H) { return kind == 0 ? W + H - 1 : kind == 1 ? H : W; }

// pixels of front f (0 when the direction has no such front)
VWGPU_HD int mgm_front_width(int kind, int f, int W, int H) {
  if (f < 0) return 0;
  if (kind == 0) {
    if (f > W + H - 2) return 0;
    int n = f < W + H - 2 - f ? f : W + H - 2 - f;
    const int m = (W < H ? W : H) - 1;
    if (n > m) n = m;
    return n + 1;
  }
  if (kind == 1) return f < H ? W : 0;
  return f < W ? H : 0;
}

// pixel i of front f; false when the front has no such pixel
VWGPU_HD bool mgm_front_pixel(int kind, int flipx, int flipy, int f, int i, int W, int H, int& c, int& r) {
  if (i < 0 || f < 0) return false;
  if (kind == 0) {
    const int lo = f - (H - 1) > 0 ? f - (H - 1) : 0;
    const int cc = lo + i, rr = f - cc;
    if (cc >= W || rr < 0) return false;
    c = flipx ? W - 1 - cc : cc; r = flipy ? H - 1 - rr : rr;
  } else if (kind == 1) {
    if (f >= H || i >= W) return false;
    c = i; r = flipy ? H - 1 - f : f;
  } else {
    if (f >= W || i >= H) return false;
    r = i; c = flipx ? W - 1 - f : f;
  }
  return true;
}

// the task's border test: does the pixel use its predecessors?
VWGPU_HD bool mgm_uses_predecessors(int need, int c, int r, int W, int H) {
  return (!(need & 1) || c > 0) && (!(need & 2) || c < W - 1) && (!(need & 4) || r > 0) && (!(need & 8) || r < H - 1);
}

}  // namespace vwgpu
