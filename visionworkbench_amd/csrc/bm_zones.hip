// bm_zones.hip — all search zones of one pyramid level in ONE launch sequence.
//
// PyramidCorrelationView::prerasterize runs calc_disparity once per SearchParam zone
// (src/vw/Stereo/CorrelationView.cc:596-700); at level 0 of a 1024^2 tile that is ~2000 zones of ~32x32 pixels with
// ~5x5 disparities each — as separate launches they cost ~600 ms of launch latency.  Here a zone is a row of a device
// table and a workgroup serves a WORK ITEM: one 32x32 output tile of one zone and a run of its disparities:
//   * the tile's left patch and (per dy, per chunk of dx) right patch are staged in LDS with coordinates CLAMPED into
//     the level image — exactly the ConstantEdgeExtension crops the reference hands to calc_disparity when a padded
//     zone sticks out of the level image (CorrelationView.cc:607-616,660-668);
//   * per disparity: horizontal kx-sums of the float cost elements (widened to float64, CostFunctions.h:72-141) into
//     an LDS plane, then vertical ky-sums — the box sum of fast_box_sum (Algorithms.h:43-129) in a different but
//     exact order (float data summed in float64: every partial sum is representable unless the window spans > 2^20 in
//     magnitude, see bm_generic.hip);
//   * best / worst / first-wins compare chain and the best == worst validity rule of best_of_search_convolution
//     (Correlation.cc:91-133), dy outer / dx inner like the reference;
//   * NCC: cost *= sqrt(precA * precB) with the 1/box-sum(img^2) images (CostFunctions.h:214-231) precomputed over the
//     (clamped) union of all zone origins of the level.
// Work items (round 4).  Rounds 1-3 gave a tile ALL its disparities and issued the tiles in zone order — the zones arrive sorted by
// ascending search volume, so the few tiles that search 200+ disparities started LAST and the launch waited for them: PMC on the
// level-0 launch of a 1024^2 tile showed 1.7 resident waves per SIMD on average and a launch 2.4x (NCC) to 4x (SAD) longer than its
// work spread evenly (profiles/r04_zones_pmc_*.md).  Now a tile whose search is longer than a cap — a third of the level's work per
// resident workgroup — is cut into runs of disparities in index order, the items are issued longest first, and the runs of a
// tile leave (best, worst, first index) records that zones_merge_kernel folds in index order: the compare chain of
// Correlation.cc:91-117 is a (value, first index) minimum plus an extremum as long as no cost is NaN, and both fold exactly.  A
// tile with a non-finite cost is order dependent: the merge flags it and the tile is redone as ONE item by the launch queued behind
// the merge (every other workgroup of that launch leaves at once).
// Certification (round 4, levels whose box sums could round).  The tile-parallel sums differ from the reference's serial running sums by
// at most eps (derived in vwgpu_launch_bm_zones from the level's largest exponent and the chain lengths of the zone); a pixel whose best
// cost beats its runner-up by more than 2 eps provably has the reference's disparity AND validity.  With `cert` the kernels track the
// runner-up (NCC: also the largest right precision), and a pixel that cannot be certified raises its ZONE's flag: bm_exact.hip then
// redoes exactly the flagged zones in the reference's order.  Default results stay bit-identical to the oracle.
// A second tiny kernel applies the per-zone L/R consistency check (Correlate.cc:1441-1502) and the
// `+= zone.disparity_range().min()` offset (CorrelationView.cc:696-697) for every zone at once.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vwgpu_internal.h"

#ifdef VWGPU_TILE_STAMPS
// Tools build only (make stamps; tools/zones_timeline.py): every workgroup of the zone matcher leaves {start, end (100 MHz wall clock), where it
// ran, its item's evaluations} in a buffer set through vwgpu_debug_set_zone_stamps.  Not in the product library.
__device__ unsigned long long* g_zone_stamps = nullptr;
__device__ unsigned int g_zone_stamp_count = 0;
// knock-out experiments (tools/zones_knockout.py): bits switch parts of the big launches (> 2000 workgroups) off — wrong results, honest timing
__device__ int g_zone_knock = 0;
extern "C" int vwgpu_debug_set_zone_knock(int bits) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_zone_knock), &bits, sizeof bits) == hipSuccess ? 0 : -3;
}
extern "C" int vwgpu_debug_set_zone_stamps(void* d_buf) {
  unsigned long long* p = static_cast<unsigned long long*>(d_buf);
  unsigned int zero = 0;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_zone_stamp_count), &zero, sizeof zero) != hipSuccess) return -3;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_zone_stamps), &p, sizeof p) == hipSuccess ? 0 : -3;
}
#endif

namespace {

constexpr int ZT = 32;            // output tile side of the per-zone L/R kernel
constexpr int ZTHREADS = 256;
// (the matcher kernels take the tile side as a template parameter; only 32 x 32 outputs on 256 threads is instantiated, see the launcher)

template <int COST, typename ACC = double>
__device__ __forceinline__ ACC zcost(float a, float b) {
  if (COST == VWGPU_CROSS_CORRELATION) return (ACC)(a * b);
  if (COST == VWGPU_SQUARED_DIFFERENCE) { float d = a - b; return (ACC)(d * d); }
  return (ACC)fabsf(a - b);
}
template <int COST>
__device__ __forceinline__ bool zbetter(double c, double q) {
  return COST == VWGPU_CROSS_CORRELATION ? (c > q) : (c < q);
}

struct PrecView {           // 1 / box-sum(img^2) over origins [x0, x0+w) x [y0, y0+h); pf: the same image rounded to float32 (fp32 tier), or null
  const double* p; int x0, y0, w, h; const float* pf;
};

// prec(x, y) = 1.0 / sum_{ky x kx} img(clamp)^2 for window origins (x0 + i, y0 + j).  A 64 x 4 output tile: the squares of its
// (64 + kx - 1) x (4 + ky - 1) pixels go to LDS once, then row sums and column sums (the direct form read kx * ky floats per output
// through the L1: 0.3 ms per 1024^2 NCC tile).  Used on data whose box sums are exact in any order (vwgpu_sums_order_free) and, as
// sqrt(1 / S) (root != 0), by the certified pass, whose error bound covers the order of the sums (root == 2, the passes with the "cannot
// matter" certificate, additionally turns the infinite precision of an all-zero window into NaN for EVERY pixel of the pass, see below).
struct ZPrecJob { const float* img; int w, h; ptrdiff_t pitch; double* prec; int x0, y0, pw, ph; float* prec32; };
struct ZPrecJobs { ZPrecJob j[2]; size_t img_tile[2], prec_tile[2]; };      // blockIdx.z & 1: the left / right image of a pass, blockIdx.z >> 1: the image pair of a group
constexpr int ZP_TH = 16;          // output rows per workgroup (round 5: 4 -> 16: a 4-row tile squares (4 + ky - 1) / 4 = 3.5 x the pixels it needs at 11 x 11)
__global__ void __launch_bounds__(256)
zone_precision_kernel(ZPrecJobs jobs, int kx, int ky, int root) {
  extern __shared__ double zp_sm[];
  const ZPrecJob J = jobs.j[blockIdx.z & 1];
  const size_t gi = blockIdx.z >> 1;
  const float* __restrict__ img = J.img + gi * jobs.img_tile[blockIdx.z & 1];
  double* __restrict__ prec = J.prec + gi * jobs.prec_tile[blockIdx.z & 1];
  const int w = J.w, h = J.h, x0 = J.x0, y0 = J.y0, pw = J.pw, ph = J.ph;
  const ptrdiff_t pitch = J.pitch;
  if ((int)blockIdx.x * 64 >= pw || (int)blockIdx.y * ZP_TH >= ph) return;
  const int tw = 64 + kx - 1, th = ZP_TH + ky - 1;
  double* sq = zp_sm;                 // th x tw squares
  double* hs = zp_sm + (size_t)th * tw;   // th x 64 row sums
  const int tid = threadIdx.y * 64 + threadIdx.x;
  const int bx = blockIdx.x * 64, by = blockIdx.y * ZP_TH;
  for (int i = tid; i < tw * th; i += 256) {
    const int r = i / tw, c = i - r * tw;
    int xx = x0 + bx + c; xx = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
    int yy = y0 + by + r; yy = yy < 0 ? 0 : (yy >= h ? h - 1 : yy);
    const float v = img[(ptrdiff_t)yy * pitch + xx];
    sq[i] = (double)(v * v);
  }
  __syncthreads();
  for (int i = tid; i < th * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    double s = 0.0;
    for (int a = 0; a < kx; ++a) s += sq[r * tw + c + a];
    hs[i] = s;
  }
  __syncthreads();
  const int i = bx + threadIdx.x;
  if (i >= pw) return;
  for (int jr = threadIdx.y; jr < ZP_TH; jr += 4) {
    const int j = by + jr;
    if (j >= ph) break;
    double s = 0.0;
    for (int b = 0; b < ky; ++b) s += hs[(jr + b) * 64 + threadIdx.x];
    // root: the certified kernels multiply by sqrt(1 / S), see there; root == 2 (passes with the "cannot matter" certificate): NaN for an
    // all-zero window — such a candidate's cost is NaN in the reference (0 * inf) and never wins unless it is the first one; the chain's
    // min / max instructions skip a NaN the same way, and the first candidate is looked at separately (ZEdge)
    double pr = root ? sqrt(1.0 / s) : 1.0 / s;
    if (root == 2 && !(pr <= 1.7976931348623157e308)) pr = __builtin_nan("");
    prec[(size_t)j * pw + i] = pr;
    if (J.prec32) J.prec32[gi * jobs.prec_tile[blockIdx.z & 1] + (size_t)j * pw + i] = (float)pr;
  }
}

// The same image for square windows K x K, K = 3 ... 13 (every kernel size the matchers are instantiated for), without the load phase of the
// kernel above (7.5 clamped scalar loads per thread, each waited for before its square went to LDS: the launch ran at a fifth of its
// arithmetic).  A WAVEFRONT owns 64 consecutive image columns x ZP_R output rows: a lane requests the ZP_R + K - 1 pixels of its column
// together (rows are 256 contiguous bytes across the lanes), squares them, forms the ZP_R column sums of K rows in registers and leaves them
// in the wavefront's own LDS plane; the first 64 - (K - 1) lanes then add K neighbouring column sums.  No workgroup barrier (a wavefront
// reads only what it wrote itself).  Columns first, rows second — fast_box_sum's own nesting; the kernel above sums rows first: both are K + K
// additions of at most K^2 elements, the case the error bound of the certified pass is derived for (sum_error_units), and on order-free
// data every order returns the same bits.
constexpr int ZP_R = 16;
template <int K>
__global__ void __launch_bounds__(256)
zone_precision_sq_kernel(ZPrecJobs jobs, int root) {
  __shared__ double vs[4][ZP_R][64];
  const ZPrecJob J = jobs.j[blockIdx.z & 1];
  const size_t gi = blockIdx.z >> 1;
  const float* __restrict__ img = J.img + gi * jobs.img_tile[blockIdx.z & 1];
  const size_t poff = gi * jobs.prec_tile[blockIdx.z & 1];
  double* __restrict__ prec = J.prec + poff;
  float* __restrict__ prec32 = J.prec32 ? J.prec32 + poff : nullptr;
  const int w = J.w, h = J.h, pw = J.pw, ph = J.ph;
  const ptrdiff_t pitch = J.pitch;
  constexpr int OW = 64 - (K - 1), NR = ZP_R + K - 1;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), lane = (int)threadIdx.x & 63;
  const int bx = (int)blockIdx.x * OW, by = ((int)blockIdx.y * 4 + wave) * ZP_R;
  if (bx >= pw || by >= ph) return;
  int xx = J.x0 + bx + lane; xx = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
  float px[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    int yy = J.y0 + by + r; yy = yy < 0 ? 0 : (yy >= h ? h - 1 : yy);
    px[r] = img[(ptrdiff_t)yy * pitch + xx];
  }
  double sq[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) sq[r] = (double)(px[r] * px[r]);              // square(): float product (Math/Functors.h:316-321)
  double (*V)[64] = vs[wave];
#pragma unroll
  for (int j = 0; j < ZP_R; ++j) {
    double s = 0.0;
#pragma unroll
    for (int b = 0; b < K; ++b) s += sq[j + b];
    V[j][lane] = s;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wavefront's own LDS writes, then reads by other lanes
  const int i = bx + lane;
  if (lane >= OW || i >= pw) return;
#pragma unroll 4
  for (int j = 0; j < ZP_R; ++j) {
    if (by + j >= ph) break;
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < K; ++a) s += V[j][lane + a];
    double pr = root ? sqrt(1.0 / s) : 1.0 / s;             // (root, NaN: see the kernel above)
    if (root == 2 && !(pr <= 1.7976931348623157e308)) pr = __builtin_nan("");
    prec[(size_t)(by + j) * pw + i] = pr;
    if (prec32) prec32[(size_t)(by + j) * pw + i] = (float)pr;
  }
}

// A work item: disparities [i0, i0 + n) (index = dy * sx + dx, the reference's loop order) of one 32 x 32 tile of one zone.
struct ZItem {
  int zone, txy;        // zone row; tile x | tile y << 16
  int i0, n;
  int slot;             // >= 0: the tile is cut into several items, this one leaves its records in partial slot `slot`; -1: the tile's only item
  int gate;             // >= 0: run only if redo[gate] != 0 (the redo launch behind the merge); -1: always
  int pad0, pad1;
};
struct ZMergeItem { int zone, txy, slot0, nitems, gate, pad0, pad1, pad2; };
// records of the partial slots, one plane of 1024 pixels per slot and field
struct ZPart { double* best; double* worst; int* idx; double* second; double* rpmax; int* bad; int* redo; double* bnf; };
// certification constants of a zone: bounds on |tile-parallel sum - reference running sum| (see vwgpu_launch_bm_zones)
struct ZCert { double eps_s, eps_ll, eps_rr, eps32; int edge_lo, edge_hi, trow, pad1; };      // eps32: bound on |float32 tile sum - exact sum| of the fp32 tier, see vwgpu_launch_bm_zones
struct ZCertArgs { const ZCert* zc; int* zflag; unsigned long long* stats; int* any; const int* need; const unsigned char* cells;
                   int edge_m, edge_k;        // zc == nullptr: no certification; any[image]: "some zone was flagged"; edge_* (the bounds: per zone, in ZCert): see ZEdge
                   int* tflag; };             // tflag != nullptr: the 32-row bands of a zone that hold an unproven pixel, tflag[ZCert::trow + tile y] = 1 + the last such tile's column (see vwgpu_zone_row_flags)
// The "cannot matter" certificate (EDGE kernels; edge_m > 0).  A candidate (pixel, disparity) whose partner lies edge_m or more columns
// outside the other image — partner column = origin of its window in the other image - edge_k, outside [edge_lo, edge_hi] — is FAR
// (edge_m > 0 switches the certificate on; the caller folds the margin into the two bounds).  Far windows are clamped copies of the border column: whole runs of them have bit-identical data, their costs tie exactly
// in any arithmetic that treats them alike and in the reference's running sums only almost — no certificate can order them.  But when
// the best cost is far and leads the best NOT-far cost (`bnf`) by more than 2 eps, the reference's winner is SOME far candidate, and
// every one of them meets the same end: an L->R pixel that points edge_m = filter half kernel + 4 columns outside the right image is
// never within 3 px of a neighbour that points inside (the two clean-up passes count nothing else) and is then erased by
// disparity_mask (DisparityMap.h:132-155); an R->L pixel that points floor(threshold) + 2 columns outside the left image is
// inconsistent with every left pixel of the check (cross_corr_consistency_check), its only reader.  Which far candidate won cannot be seen
// in the output.  pyramid.hip says when the premises hold (a mask pass follows; no lr_disp_diff image requested).
struct ZEdge {
  int lo, hi, base;                                             // partner column of (lane's pixel, dx index i) = base + i
  __device__ __forceinline__ bool notfar(int i) const { return (unsigned)(base + i - lo) <= (unsigned)(hi - lo); }
};
// need != nullptr (the R->L pass of a level with the L/R check): eight ints per zone from zone_need_kernel — the rectangle of the zone the
// check will read, and where the zone's 16 x 16-pixel cell flags start.  The tiles are laid from the rectangle's corner and stop at its
// far side, and a tile none of whose cells holds a position the check reads is skipped; the rest of the zone is not matched (every pixel
// of these kernels is independent of the others, so the pixels that are matched get the same result).
struct ZGeom { int ox, oy, tw, th; };
template <int ZT>
__device__ __forceinline__ bool ztile_geometry(const vwgpu_zone_task& z, int zone, int txy, const int* need, const unsigned char* cells, ZGeom& g) {
  int x0 = 0, y0 = 0, x1 = z.zw, y1 = z.zh;
  int cell0 = -1;
  if (need) {
    const int4 b = reinterpret_cast<const int4*>(need)[2 * zone];   // {zw - x0, zh - y0, x1, y1} as maxima, all 0 = nothing needed
    if (b.z <= 0) return false;
    x0 = z.zw - b.x; y0 = z.zh - b.y; x1 = b.z; y1 = b.w;
    const int4 e = reinterpret_cast<const int4*>(need)[2 * zone + 1];      // {first cell, 1 = every cell, -, -}
    if (e.y == 0) cell0 = e.x;
  }
  g.ox = x0 + (txy & 0xffff) * ZT; g.oy = y0 + (txy >> 16) * ZT;
  if (g.ox >= x1 || g.oy >= y1) return false;
  g.tw = min(ZT, x1 - g.ox); g.th = min(ZT, y1 - g.oy);
  if (cell0 >= 0) {
    const int ncx = (z.zw + 15) >> 4;
    bool any = false;
    for (int cy = g.oy >> 4; cy <= (g.oy + g.th - 1) >> 4; ++cy)
      for (int cx = g.ox >> 4; cx <= (g.ox + g.tw - 1) >> 4; ++cx) any = any || cells[cell0 + cy * ncx + cx] != 0;
    if (!any) return false;
  }
  return true;
}

// "can this pixel's result be proven equal to the reference's?"  best / second / rpmax: what the chain has seen over ALL D disparities of
// the pixel (second = the best cost among the disparities other than the winner; equal costs => second == best).  A certified pixel with
// D >= 2 is valid (best > second >= worst in the reference's arithmetic too), so the chain keeps no `worst`; D == 1 is invalid
// (best == worst) provided the reference's one cost is not a NaN — see the NCC branch.
// T32 (the fp32 tier): the window sums were formed in float32 — the SAME float cost elements as the reference's (CostFunctions.h:72-141
// forms them in float), summed in another order AND another precision.  Two bounds on |float32 sum - exact sum|, u = 2^-24:
//   absolute  zc.eps32 = 2 (2 K + 24) kx ky u E   (E bounds an element, K = max(kx, ky)): the sliding form of zwindow_sums subtracts, so an
//             operation's rounding scales with the running value (<= kx ky E), not with the window it ends in; (2 K + 24) counts the
//             operations on a result's path through both passes (initial K adds + 2 per slide step, the ky inherited row errors folded in);
//   relative  REL32 (the tree form of zwindow_sums, no subtraction: KS >= 9 here): depth <= 22 additions on any element's path, so
//             |error| <= 22 u sum|e|; SAD / SSD elements are >= 0 (sum|e| = the sum itself), NCC: sum|a b| <= sqrt(S_ll S_rr)
//             (Cauchy-Schwarz) = 1 in cost units.  2^-18 = 64 u leaves a factor ~3; NCC adds the float32 roundings of the right factor and
//             of the product (3 u) inside the same constant.
template <int COST, bool T32 = false, bool REL32 = false>
__device__ __forceinline__ bool zcertified(const ZCert& zc, int D, bool bad, double best, double second, double lprec, double rpmax) {
  if (bad) return false;                                     // a non-finite cost: the reference's chain is order dependent there
  if (D == 1 && COST != VWGPU_CROSS_CORRELATION) return true;   // (finite pixels: a finite cost, best == worst)
  double eps;
  if (COST == VWGPU_CROSS_CORRELATION) {
    // cost = S_lr * sqrt((1 / S_ll) * (1 / S_rr)), every S off by at most its eps, 1/x and sqrt propagated to first order with a factor 2.
    // |cost| <= 1 for every disparity (Cauchy-Schwarz: the three sums run over the same window positions), up to the roundings.
    const double dl = 2.0 * zc.eps_ll * lprec, dr = 2.0 * zc.eps_rr * rpmax;
    if (!(lprec > 0.0) || !(rpmax > 0.0) || !(dl <= 0x1p-10) || !(dr <= 0x1p-10)) return false;
    // D == 1 is "invalid" only if the reference's one cost is a NUMBER (best == worst; a NaN compares unequal and the pixel stays valid,
    // Correlation.cc:121-133).  Its running box sums of squares can cancel to zero or below on data of many decades — 1 / S infinite or
    // negative, the cost NaN — where the tile sums here are fine: the two tests above say the reference's S_ll and S_rr lie within 2^-11 of
    // these (positive, finite precisions), so its cost is finite too.  (Found by the round-5 campaign: 12-decade data, 1 x 7 window, one disparity.)
    if (D == 1) return true;
    const double cmax = fmax(1.0 + 0x1p-20, fmax(fabs(best), fabs(second)));
    double s32 = 0.0;                                        // the fp32 tier's own error, in cost units
    if (T32) {
      s32 = zc.eps32 * sqrt(lprec * rpmax) + 0x1p-21 * cmax;   // absolute form (+ 8 u: right factor and product in float32)
      if (REL32) s32 = fmin(s32, 0x1p-18 * cmax);
    }
    eps = 2.0 * (cmax * (dl + dr + 0x1p-49) + zc.eps_s * sqrt(lprec * rpmax) + s32);      // 2^-49: the roundings of either way to the cost
  } else {
    eps = zc.eps_s;
    if (T32) {
      double s32 = zc.eps32;
      if (REL32) s32 = fmin(s32, 0x1p-18 * fmax(fabs(best), fabs(second)));
      eps += s32;
    }
  }
  const double gap = COST == VWGPU_CROSS_CORRELATION ? best - second : second - best;
  return gap > 2.0 * eps;                                    // (false for NaN)
}

// The certified pass scores NCC as S_lr * sr * sl with sl = sqrt(1 / S_ll), sr = sqrt(1 / S_rr) from the precision images (the reference:
// S_lr * sqrt((1 / S_ll) * (1 / S_rr)), CostFunctions.h:207-236 — a square root sequence per evaluation): the same number up to a
// handful of roundings on either side (2^-49 in zcertified).  sl is the same for all disparities of a pixel, so the chain runs on
// S_lr * sr and sl scales its records afterwards.  zcertified takes precisions: the callers square the roots.
// min / max of the chain as the bare instructions: fmin / fmax spell a canonicalisation (v_max_f64 x, x, x) in front of every operand that
// comes round the loop — five of the 21 instructions of an evaluation.  A NaN operand is ignored (IEEE mode: the other one is returned).
__device__ __forceinline__ double zmin_raw(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double zmax_raw(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float zmin_raw(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float zmax_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

typedef float zfloat2 __attribute__((ext_vector_type(2)));

// N window sums of KS elements from KS + N - 1 elements e[]: w[j] = e[j] + ... + e[j + KS - 1].  Not as a slide (w' = w - e_old + e_new: one
// chain of KS + 2 N dependent float64 additions, and four wavefronts per SIMD do not cover the latency of such a chain — knock-out
// timing, tools/zones_knockout.py) but as the elements every window shares, summed as a tree, plus a run of prefix sums on either side:
// three independent chains of depth <= N.  On order-free data any order gives the same bits; with CERT the roundings are part of eps.
template <int n, typename ACC>
__device__ __forceinline__ ACC ztree_sum(const ACC* e) {
  if constexpr (n == 1) return e[0];
  else if constexpr (n == 2) return e[0] + e[1];
  else return ztree_sum<n / 2, ACC>(e) + ztree_sum<n - n / 2, ACC>(e + n / 2);
}
template <int KS, int N, typename ACC>
__device__ __forceinline__ void zwindow_sums(const ACC* e, ACC* w) {
  if constexpr (KS >= N) {
    const ACC core = ztree_sum<KS - N + 1, ACC>(e + N - 1);      // e[N - 1 .. KS - 1]: in every window
    ACC left[N], right[N];                                      // left[j] = e[j] + ... + e[N - 2], right[j] = e[KS] + ... + e[KS + j - 1]
    left[N - 2] = e[N - 2];
#pragma unroll
    for (int j = N - 3; j >= 0; --j) left[j] = e[j] + left[j + 1];
    right[1] = e[KS];
#pragma unroll
    for (int j = 2; j < N; ++j) right[j] = right[j - 1] + e[KS + j - 1];
    w[0] = left[0] + core;
#pragma unroll
    for (int j = 1; j < N - 1; ++j) w[j] = (left[j] + core) + right[j];
    w[N - 1] = core + right[N - 1];
  } else {
    ACC sacc = e[0];
#pragma unroll
    for (int a = 1; a < KS; ++a) sacc += e[a];
    w[0] = sacc;
#pragma unroll
    for (int j = 1; j < N; ++j) { sacc = sacc - e[j - 1] + e[j - 1 + KS]; w[j] = sacc; }
  }
}

#ifndef VWGPU_TILE_STAMPS
#define ZKNOCK(bit) false
#endif

// KS > 0: a square KS x KS window known at compile time — the horizontal and vertical window sums are unrolled (with run-time
// sizes the loop overhead outweighed the sums, as PMC showed for bm_generic).  KS == 0: any kx, ky.
// ACC: the type of the window sums.  float64 is the reference's; float32 is taken when every intermediate value is exactly representable in 24
// bits as well (vwgpu_sums_bits <= 24: byte imagery under SAD) — then both give the same numbers, the LDS planes are half as large and the sums
// full-rate.  The compare chain runs on doubles either way.
// CERT: track the runner-up (and the largest right precision) and certify / flag, see the file header.
// what every item of a launch shares
struct ZLaunch {
  const float* A; int aw, ah, ap; const float* B; int bw, bh, bp;
  int kx, ky, sxc; PrecView pa, pb; int32_t* out; ZPart P; ZCertArgs C;
  size_t a_tile, b_tile, pa_tile, pb_tile;      // groups (vwgpu_zone_group): elements between the images / precision images of consecutive image pairs
};

// One work item.  Returns (CERT, an item that finishes its tile): true when some pixel of the tile could not be certified — the caller
// decides what happens then (the fp32 tier runs the item again in float64; the float64 tier flags the zone).  Every thread of the workgroup
// runs this function; the value is per thread (fold it with __syncthreads_or).
// ACC = float with CERT is the FP32 TIER (T32): window sums, compare chain and NCC right factors in float32 — simple float32
// instructions issue at twice the rate of float64 ones (profiles/r04_ubench_valu.txt, r04_ubench_f64.txt) and the sum planes are half as
// large — certified against BOTH the reference's summation order and its own float32 roundings (zcertified<T32>).
template <int COST, int KS, typename ACC, bool CERT, int ZS, bool EDGE>
__device__ __forceinline__ bool zmatch_item(const ZLaunch& G, const ZItem& it, const vwgpu_zone_task& z, const ZGeom& geom, char* smem) {
  const float* __restrict__ A = G.A + (size_t)z.img * G.a_tile; const float* __restrict__ B = G.B + (size_t)z.img * G.b_tile;
  const int aw = G.aw, ah = G.ah, ap = G.ap, bw = G.bw, bh = G.bh, bp = G.bp, kx = G.kx, ky = G.ky, sxc = G.sxc;
  PrecView pa = G.pa, pb = G.pb;
  if (COST == VWGPU_CROSS_CORRELATION) {
    pa.p += (size_t)z.img * G.pa_tile; pb.p += (size_t)z.img * G.pb_tile;
    if (pb.pf) pb.pf += (size_t)z.img * G.pb_tile;
  }
  int32_t* __restrict__ out = G.out;
  const ZPart& P = G.P; const ZCertArgs& C = G.C;
  // tile side, threads, columns per horizontal item (float64 sums: four — eight need 36 registers for the elements alone and cost a
  // resident workgroup per CU), items per row
  constexpr int ZT = ZS, ZTHREADS = ZS * ZS / 4, HW = 8, QL = ZS / HW;
  constexpr bool T32 = CERT && sizeof(ACC) == 4;
  constexpr int DP = (KS > 0 && T32) ? 2 : 1;                     // disparities per step: two in the fp32 tier, see the disparity loop (the lean float32 chain of order-free SAD / SSD levels gains nothing from it and keeps its small LDS footprint: six workgroups per CU)
  constexpr bool REL32 = T32 && KS >= 9;                          // the tree form of zwindow_sums in both passes (no subtraction), see zcertified
  typedef typename std::conditional<T32, float, double>::type CT; // type of the compare chain and of the NCC right factors
#ifdef VWGPU_TILE_STAMPS
  const int knock = gridDim.x > 2000 ? g_zone_knock : 0;
#define ZKNOCK(bit) (knock & (bit))
  unsigned long long stamp_t0 = 0, stamp_c0 = 0;
  if (g_zone_stamps && threadIdx.x == 0) { stamp_t0 = wall_clock64(); stamp_c0 = clock64(); }
#endif
  // LDS pitches: ODD row pitches for the two float patches and ZT + 1 for the sum planes.  A horizontal item is lane <-> (row, group of
  // HW columns): with the natural pitches (42 floats at 11 x 11, 32 sums) the rows of a half wave fell on the same banks — four-way
  // conflicts on every patch read, eight-way on the plane writes (PMC round 4: 40 % of the LDS-active cycles were conflict cycles).
  // (compile-time for a compile-time window: the row offsets of the vertical pass become immediate offsets of the LDS reads)
  const int PW = KS > 0 ? ((ZT + KS - 1) | 1) : ((ZT + kx - 1) | 1), PH = KS > 0 ? ZT + KS - 1 : ZT + ky - 1, RW = (ZT + kx - 1 + sxc - 1) | 1;
  constexpr int HP = ZT + 1;
  float* Lp = reinterpret_cast<float*>(smem);                    // PH x PW
  float* Rp = Lp + PH * PW;                                      // PH x RW
  ACC* H = reinterpret_cast<ACC*>(smem + (((size_t)(PH * PW + PH * RW) * 4 + 7) & ~size_t(7)));   // 2 x DP x PH x HP (float32: four planes in the room of two float64 ones)

  const int ox = geom.ox, oy = geom.oy, tw = geom.tw, th = geom.th;
  const int pw = tw + kx - 1, ph = th + ky - 1;
  const int t = threadIdx.x;
  // Lane <-> pixels.  Wide tiles: column t % ZT, rows 4 (t / ZT) .. + 3.  NARROW tiles (round 5: at most 16 columns — the 16 x 16 leaves of the
  // quad tree are 1249 of the ~2000 zones of a level-0 tile): column t % 16, rows 4 (t / 16) .. + 3, so that a 16 x 16 tile is ONE
  // full wavefront instead of halves of two, and the wavefronts whose rows lie below the tile skip the vertical pass and the chain
  // altogether (`wave_active`, a scalar branch — not one of the exec-mask regions the loop was cleared of).  The sum planes keep their
  // layout; the horizontal pass of a narrow tile has two items per row instead of four.
  const bool narrow = ZT == 32 && tw <= 16;                       // (workgroup-uniform)
  const int c = narrow ? (t & 15) : t % ZT, y0 = narrow ? (t >> 4) * 4 : (t / ZT) * 4;
  const int wave_row0 = __builtin_amdgcn_readfirstlane(narrow ? (t >> 6) * 16 : (t >> 6) * (256 / ZT));      // first row of the wavefront's lanes
  const bool wave_active = wave_row0 < th;
  const int qsh = narrow ? 1 : (QL == 4 ? 2 : (QL == 2 ? 1 : 0)); // log2 of the horizontal items per row

  // staging: thread <-> (column t % 32, rows t / 32, + ZTHREADS / 32, ...) — no division by the run-time patch width (a flat index
  // cost a 40-instruction division per element: two thirds of the instructions of a 16 x 16 zone with a dozen disparities)
  constexpr int SROWS = ZTHREADS / 32;
  // (compile-time windows: the rows of a column are requested TOGETHER and stored afterwards — the plain nested loop compiled to one load
  // per s_waitcnt, five or six dependent memory round trips per patch: the timeline showed 10 - 20 us of an item's 40 - 50 before its loop)
  constexpr int SRIT = KS > 0 ? (ZT + KS - 1 + SROWS - 1) / SROWS : 1;      // row iterations of a staging thread
  {
    const int q0 = t & 31, r0 = t >> 5;
    for (int q = q0; q < pw; q += 32) {
      int xx = z.ax + ox + q; xx = xx < 0 ? 0 : (xx >= aw ? aw - 1 : xx);
      if (KS > 0) {
        float v[SRIT];
#pragma unroll
        for (int k = 0; k < SRIT; ++k) {
          const int r = r0 + k * SROWS;
          int yy = z.ay + oy + r; yy = yy < 0 ? 0 : (yy >= ah ? ah - 1 : yy);
          v[k] = r < ph ? A[(size_t)yy * ap + xx] : 0.0f;
        }
#pragma unroll
        for (int k = 0; k < SRIT; ++k) {
          const int r = r0 + k * SROWS;
          if (r < ph) Lp[r * PW + q] = v[k];
        }
      } else {
        for (int r = r0; r < ph; r += SROWS) {
          int yy = z.ay + oy + r; yy = yy < 0 ? 0 : (yy >= ah ? ah - 1 : yy);
          Lp[r * PW + q] = A[(size_t)yy * ap + xx];
        }
      }
    }
  }
  CT best[4], worst[4], second[4], rpmax[4];
#ifdef VWGPU_TILE_STAMPS
  unsigned long long stamp_l = 0, stamp_r1 = 0;                  // tools build: after the left patch is staged / after the first right patch is
  if (g_zone_stamps && threadIdx.x == 0) stamp_l = wall_clock64();      // (the loads are in flight here; the first barrier waits for them)
#endif
  double lprec[4];
  int bidx[4];
  bool bad = false;
  constexpr CT kBestInit = COST == VWGPU_CROSS_CORRELATION ? -INFINITY : INFINITY;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    best[m] = worst[m] = 0; bidx[m] = 0; lprec[m] = 0.0; second[m] = 0; rpmax[m] = 0;
    if (CERT) { best[m] = second[m] = kBestInit; worst[m] = -kBestInit; }
    if (COST == VWGPU_CROSS_CORRELATION && c < tw && y0 + m < th)
      lprec[m] = pa.p[(size_t)(z.ay + oy + y0 + m - pa.y0) * pa.w + (z.ax + ox + c - pa.x0)];
  }
  CT s0[4] = {0, 0, 0, 0};                                        // EDGE: the cost of the search's FIRST candidate (a NaN there is the reference's winner: `fnan` below)
  CT bnf[4];                                                      // EDGE: the best cost among the candidates that are not far (ZEdge)
  int elo = 0, ehi = 0;
  if (EDGE) { elo = C.zc[it.zone].edge_lo; ehi = C.zc[it.zone].edge_hi; }
#pragma unroll
  for (int m = 0; m < 4; ++m) bnf[m] = kBestInit;
  ACC bestA[4], worstA[4];                                        // the lean chain of the order-free SAD / SSD levels
#pragma unroll
  for (int m = 0; m < 4; ++m) { bestA[m] = (ACC)INFINITY; worstA[m] = -(ACC)INFINITY; }
  constexpr bool LEAN = !CERT && COST != VWGPU_CROSS_CORRELATION;
  int hb = 0;
  const int iend = it.i0 + it.n;
  // (Tried and dropped, round 4: requesting the right patch of the next run of dx while this run is matched, and four instead of three
  // resident workgroups per CU through a 128-register cap.  tools/zones_timeline.py: with four residents every workgroup advanced at
  // 260 instead of 400 evaluations per microsecond — the launch is bound by the VALU instructions per evaluation, not by latency.)
  for (int i0 = it.i0; i0 < iend;) {
    const int dy = i0 / z.sx, dx0 = i0 - dy * z.sx;
    const int nd = min(min(sxc, z.sx - dx0), iend - i0);          // a run of dx inside one search row
    const int rwid = pw + nd - 1;
    __syncthreads();                                              // everyone done with the previous right patch
    {
      const int q0 = t & 31, r0 = t >> 5;
      for (int q = q0; q < rwid; q += 32) {
        int xx = z.bx + ox + dx0 + q; xx = xx < 0 ? 0 : (xx >= bw ? bw - 1 : xx);
        if (KS > 0) {
          float v[SRIT];
#pragma unroll
          for (int k = 0; k < SRIT; ++k) {
            const int r = r0 + k * SROWS;
            int yy = z.by + oy + dy + r; yy = yy < 0 ? 0 : (yy >= bh ? bh - 1 : yy);
            v[k] = r < ph ? B[(size_t)yy * bp + xx] : 0.0f;
          }
#pragma unroll
          for (int k = 0; k < SRIT; ++k) {
            const int r = r0 + k * SROWS;
            if (r < ph) Rp[r * RW + q] = v[k];
          }
        } else {
          for (int r = r0; r < ph; r += SROWS) {
            int yy = z.by + oy + dy + r; yy = yy < 0 ? 0 : (yy >= bh ? bh - 1 : yy);
            Rp[r * RW + q] = B[(size_t)yy * bp + xx];
          }
        }
      }
    }
    __syncthreads();
#ifdef VWGPU_TILE_STAMPS
    if (g_zone_stamps && threadIdx.x == 0 && stamp_r1 == 0) stamp_r1 = wall_clock64();
#endif
    // NCC: the right precisions of a disparity are requested before its horizontal pass and consumed after it (a load issued where it is
    // used put a memory round trip on the critical path of every disparity)
    // (Nothing inside the disparity loop sits under a per-row or per-lane condition that can be avoided: twelve exec-mask regions — four
    // row tests, four guarded precision loads, the lane tests, a one-trip loop — were a third of the loop's instructions, 13 saveexec /
    // 12 restores / 14 branches, and 14 % of a level-0 launch.  Lanes and rows without a pixel load the first precision of the image,
    // run the chain on whatever their plane rows hold, and are dropped in the epilogue.  Fewer s_waitcnt or address instructions did
    // not pay the same way, see below.)
    const ZEdge edge{elo, ehi, z.bx + ox + c + dx0 - C.edge_k};
    const CT* pbase;
    if constexpr (T32) pbase = pb.pf; else pbase = pb.p;
    const CT* prow[4] = {pbase, pbase, pbase, pbase};
    CT rpn[DP][4];
#pragma unroll
    for (int u = 0; u < DP; ++u)
#pragma unroll
      for (int m = 0; m < 4; ++m) rpn[u][m] = 0;
    if (COST == VWGPU_CROSS_CORRELATION) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const bool in = c < tw && y0 + m < th;
        const size_t off = (size_t)(z.by + oy + y0 + m + dy - pb.y0) * pb.w + (z.bx + ox + c + dx0 - pb.x0);
        prow[m] = pbase + (in ? off : 0);
      }
    }
    // DP disparities per barrier-delimited step (round 5).  A step is a chain of dependent latencies — patch reads, products, window sums,
    // plane writes, barrier, plane reads, window sums, compare chain — and four resident workgroups per CU do not cover it: the timeline
    // (tools/zones_timeline.py) shows 2.5 us per step and workgroup where the VALU work is 1.0.  With float32 sums (half the registers) a
    // step serves TWO disparities: the left values are read once, the right values overlap in all but one, the second disparity's loads and
    // arithmetic fill the first one's waits, and the barriers halve.  The chain still sees the disparities in index order.
    for (int d = 0; d < nd; d += DP) {
      const bool two = DP == 2 && d + 1 < nd;                     // (workgroup-uniform) the step holds a second disparity
      ACC* const Hb = H + hb * (DP * PH * HP);
      if (COST == VWGPU_CROSS_CORRELATION && wave_active) {
#pragma unroll
        for (int u = 0; u < DP; ++u) {
          const int dd = two ? d + u : d;
#pragma unroll
          for (int m = 0; m < 4; ++m) rpn[u][m] = ZKNOCK(32) ? (CT)0 : prow[m][dd];      // (lanes without a pixel read pb.p[dd]: dd < nd <= z.sx <= pb.w, inside the precision image's first row)
        }
      }
      if (KS > 0) {
        // HW adjacent columns per thread: KS + HW - 1 cost elements are formed once and the window slides (s' = s - e[j] + e[j + KS]).
        // With float32 sums HW = 8: 2 (KS + 7) LDS reads and KS + 13 adds for eight sums, and the ph x ZT / 8 items of a disparity are
        // ONE round of the workgroup (with four columns per item a 32 x 32 tile is 1.3 rounds: two of the four waves run twice and the
        // others wait at the barrier).  On order-free data the slide is exact; with CERT its roundings are part of eps.
        // ((ZT + KS - 1) x QL items <= ZTHREADS for every compile-time window: one item per thread, no loop)
        static_assert(KS == 0 || (ZT + KS - 1) * QL <= ZTHREADS, "one horizontal item per thread");
        {
          const int i = t;
          const int r = i >> qsh, q = (i & ((1 << qsh) - 1)) * HW;
          if (i < (ph << qsh) && q < tw) {
            const float* lp = Lp + r * PW + q;
            const float* rp = Rp + r * RW + q + d;
            constexpr int NE = KS > 0 ? KS + HW - 1 : 1;
            // (Tried: every patch read of the item behind ONE wait — an empty asm statement that takes all the values, the compiler then
            // places a single s_waitcnt instead of eight — and the same for the plane reads of the vertical pass: 5 % SLOWER; the early
            // products overlap the later reads inside the wavefront.)
            float lv[NE], rv[NE + DP - 1];                        // (the right values of disparity d + 1 are those of d, one further; the last one is the patch's next column or, in the very last step of a row of dx, a stale word that `two` keeps unused)
#pragma unroll
            for (int a = 0; a < NE; ++a) lv[a] = lp[a];
#pragma unroll
            for (int a = 0; a < NE + DP - 1; ++a) rv[a] = rp[a];
#pragma unroll
            for (int u = 0; u < DP; ++u) {
              if (u == 1 && !two) break;
              ACC e[NE];
              if (COST == VWGPU_CROSS_CORRELATION && NE % 2 == 0) {      // two float products per instruction (v_pk_mul_f32)
#pragma unroll
                for (int a = 0; a < NE; a += 2) {
#ifdef VWGPU_TILE_STAMPS
                  zfloat2 l2, r2;
                  if (ZKNOCK(1)) { l2.x = (float)(i + a); l2.y = (float)(i - a); r2 = l2; asm volatile("" : "+v"(l2.x), "+v"(r2.y)); }
                  else { l2 = zfloat2{lp[a], lp[a + 1]}; r2 = zfloat2{rp[a + u], rp[a + u + 1]}; }
#else
                  const zfloat2 l2 = {lv[a], lv[a + 1]}, r2 = {rv[a + u], rv[a + u + 1]};
#endif
                  const zfloat2 p2 = l2 * r2;
                  e[a] = (ACC)p2.x; e[a + 1] = (ACC)p2.y;
                }
              } else {
#pragma unroll
                for (int a = 0; a < NE; ++a) e[a] = zcost<COST, ACC>(lv[a], rv[a + u]);
              }
              ACC wsum[HW];
              zwindow_sums<KS, HW, ACC>(e, wsum);
              // one address register + immediate offsets (the compiler re-derived base + constant per store); through an address-space-3
              // pointer, so that the laundered address still selects ds_* instructions
              typedef __attribute__((address_space(3))) ACC lds_acc;
              lds_acc* h = (lds_acc*)(Hb + u * (PH * HP) + r * HP + q);
              asm volatile("" : "+v"(h));
              if (ZKNOCK(2)) {
                ACC tot = 0;
#pragma unroll
                for (int j = 0; j < HW; ++j) tot += wsum[j];
                if (tot == (ACC)12345.678) h[0] = tot;
              } else {
#pragma unroll
                for (int j = 0; j < HW; ++j) h[j] = wsum[j];
              }
            }
          }
        }
      } else {
        for (int i = t; i < ph * ZT; i += ZTHREADS) {             // horizontal sums
          const int r = i / ZT, q = i % ZT;
          if (q < tw) {
            const float* lp = Lp + r * PW + q;
            const float* rp = Rp + r * RW + q + d;
            ACC s = 0;
            for (int a = 0; a < kx; ++a) s += zcost<COST, ACC>(lp[a], rp[a]);
            Hb[r * HP + q] = s;
          }
        }
      }
      if (!ZKNOCK(16)) __syncthreads();
      if (wave_active && (KS > 0 || c < tw)) {                    // (compile-time window: every lane and row of a wavefront with pixels runs, see above)
#pragma unroll
       for (int u = 0; u < DP; ++u) {
        if (u == 1 && !two) break;
        ACC* const Hc = Hb + u * (PH * HP);
        const int di = i0 + d + u;
        const bool first = (di == it.i0);
        int div = di;
        if (CERT || COST != VWGPU_CROSS_CORRELATION) asm volatile("" : "+v"(div));      // one copy to a vector register per disparity instead of one per select
        const bool nfar = EDGE ? edge.notfar(d + u) : true;
        ACC vs[4] = {0, 0, 0, 0};
        if (KS > 0) {                                           // the same slide down the rows (rows beyond th hold stale planes: unused)
          ACC h[KS > 0 ? KS + 3 : 1];
#ifdef VWGPU_TILE_STAMPS
#pragma unroll
          for (int b = 0; b < KS + 3; ++b) { if (ZKNOCK(4)) { h[b] = (ACC)(t + b + d); asm volatile("" : "+v"(h[b])); } else h[b] = Hc[(y0 + b) * HP + c]; }
#else
          typedef __attribute__((address_space(3))) ACC lds_acc;
          const lds_acc* hv = (const lds_acc*)(Hc + y0 * HP + c);
          asm volatile("" : "+v"(hv));
#pragma unroll
          for (int b = 0; b < KS + 3; ++b) h[b] = hv[b * HP];                     // y0 + b <= ZT - 4 + KS + 2 = PH - 1
#endif
          zwindow_sums<KS, 4, ACC>(h, vs);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          const int y = y0 + m;
          if (KS > 0 || y < th) {
            ACC sa = vs[m];
            if (KS == 0) {
              for (int b = 0; b < ky; ++b) sa += Hc[(y + b) * HP + c];
            }
            CT s = (CT)sa;
            if (CERT && ZKNOCK(8)) { best[m] += s; }
            else if (CERT) {
              // Certified pass: the chain of Correlation.cc:91-117 reduced to what a certificate needs — the minimum with its first index, the
              // runner-up (= min over the others: min(second, max(s, best)) before best moves; equal costs give second == best) and, NCC,
              // the largest right factor — as min / max instructions instead of compare-and-select pairs: 5 (NCC 7) instructions per
              // evaluation, 20 before.  Non-finite costs: the level holds finite pixels below 2^60 (cert_hi), so S_lr is finite; an
              // infinite right factor shows in rpmax, an infinite cost in best — both end without a certificate.
              if (COST == VWGPU_CROSS_CORRELATION) {
                const CT rp = rpn[u][m];                                   // sqrt(1 / S_rr); the left factor scales the records afterwards
                rpmax[m] = zmax_raw(rpmax[m], rp);
                s *= rp;
              }
              const bool cb = COST == VWGPU_CROSS_CORRELATION ? (s > best[m]) : (s < best[m]);
              // (a select under the workgroup-uniform test "first disparity of the search": ONE instruction per evaluation — `if (di == 0)
              // fnan = !(s == s)` compiled to five: compare, two flag materialisations, and, compare)
              if (EDGE) s0[m] = di == 0 ? s : s0[m];
              if (EDGE) {
                const CT snf = nfar ? s : kBestInit;
                bnf[m] = COST == VWGPU_CROSS_CORRELATION ? zmax_raw(bnf[m], snf) : zmin_raw(bnf[m], snf);
              }
              if (COST == VWGPU_CROSS_CORRELATION) {
                second[m] = zmax_raw(second[m], zmin_raw(s, best[m]));
                best[m] = zmax_raw(best[m], s);
              } else {
                second[m] = zmin_raw(second[m], zmax_raw(s, best[m]));
                best[m] = zmin_raw(best[m], s);
              }
              bidx[m] = cb ? div : bidx[m];
            } else if (COST != VWGPU_CROSS_CORRELATION) {
              // Order-free SAD / SSD level: the costs are finite (the level's pixels are), and on finite costs the chain of
              // Correlation.cc:91-117 IS (minimum, its first index, maximum) — a cost that becomes the new best is below the first cost,
              // which `worst` starts from, so it never is the maximum.  In the type of the sums (float32 when they are exact there).
              const bool cb = sa < bestA[m];
              bestA[m] = zmin_raw(bestA[m], sa);
              worstA[m] = zmax_raw(worstA[m], sa);
              bidx[m] = cb ? div : bidx[m];
            } else if constexpr (!T32) {
            if (COST == VWGPU_CROSS_CORRELATION) {
              const double rp = rpn[u][m];
              s *= sqrt(lprec[m] * rp);
            }
            if (it.slot >= 0) bad = bad || (c < tw && y < th && !(fabs(s) <= 1.7976931348623157e308));
            // Correlation.cc:91-117 as selects (a branch per comparison costs more than the comparisons): the first disparity sets
            // best = worst; a strictly better cost takes best and the index; otherwise a cost that is not better than worst takes worst
            // (a NaN cost compares false both times: it never wins and becomes `worst`, as in the reference)
            const bool cb = zbetter<COST>(s, best[m]), cw = zbetter<COST>(s, worst[m]);
            const bool ub = first || cb, uw = first || (!cb && !cw);
            best[m] = ub ? s : best[m];
            bidx[m] = ub ? di : bidx[m];
            worst[m] = uw ? s : worst[m];
            }
          }
        }
       }
      }
      hb ^= 1;                                                  // next disparity writes the other plane
    }
    i0 += nd;
  }
#ifdef VWGPU_TILE_STAMPS
  if (g_zone_stamps && threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned k = atomicAdd(&g_zone_stamp_count, 1u);
    if (k < (1u << 18)) {
      unsigned long long* rec = g_zone_stamps + (size_t)k * 4;
      rec[0] = stamp_t0; rec[1] = wall_clock64(); rec[2] = ((unsigned long long)xcc << 32) | hw;
      // (phases: ticks of the 100 MHz clock from the start to "left patch requested" and to "first right patch staged", 12 bits each, in the top of the clock word)
      const unsigned long long pl_ = min(stamp_l - stamp_t0, 4095ull), pr_ = min(stamp_r1 - stamp_t0, 4095ull);
      rec[3] = (pl_ << 52) | (pr_ << 40) | (((unsigned long long)(unsigned)(clock64() - stamp_c0) & 0xffffffull) << 32 >> 0) | (unsigned)(it.n * tw * th);
    }
  }
#endif
  if (LEAN) {
#pragma unroll
    for (int m = 0; m < 4; ++m) { best[m] = (CT)bestA[m]; worst[m] = (CT)worstA[m]; }
  }
  bool fnan[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) fnan[m] = EDGE && !(s0[m] == s0[m]);
  if (it.slot >= 0) {                                           // one of several runs of this tile: leave the records to zones_merge_kernel
    const size_t base = (size_t)it.slot * (ZT * ZT);
    if (c < tw) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int y = y0 + m;
        if (y < th) {
          const size_t o = base + (size_t)y * ZT + c;
          P.best[o] = (double)best[m]; P.idx[o] = (EDGE && fnan[m]) ? (bidx[m] | (int)0x80000000) : bidx[m];      // (bit 31: the run starts the search with a NaN)
          if (CERT) {
            P.second[o] = (double)second[m]; if (COST == VWGPU_CROSS_CORRELATION) P.rpmax[o] = (double)rpmax[m];
            if (EDGE) P.bnf[o] = (double)bnf[m];
            bad = bad || !(fabs((double)best[m]) <= 1.7976931348623157e308);
          } else {
            P.worst[o] = (double)worst[m];
          }
        }
      }
    }
    if (__syncthreads_or(bad ? 1 : 0) && t == 0) P.bad[it.slot] = 1;     // (the slot flags are zeroed with the tables)
    return false;
  }
  bool uncert = false;
  if (c < tw) {
    const int D = z.sx * z.sy;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int y = y0 + m;
      if (y < th) {
        int32_t* o = out + ((size_t)z.out_off + (size_t)(oy + y) * z.out_stride + ox + c) * 3;
        const int bsel = (EDGE && fnan[m]) ? 0 : bidx[m];          // (a NaN first candidate is the reference's winner)
        const int by_ = bsel / z.sx, bx_ = bsel - by_ * z.sx;
        o[0] = bx_ + z.addx; o[1] = by_ + z.addy;
        if (CERT) {
          const bool ncc = COST == VWGPU_CROSS_CORRELATION;         // NCC: lprec / rpmax hold square roots of precisions here
          const double sl = ncc ? lprec[m] : 1.0;
          const double bd = (double)best[m], sd = (double)second[m], rq = (double)rpmax[m], nfd = (double)bnf[m];
          o[2] = D == 1 ? 0 : 0x7fffffff;                           // (what a certificate implies; an uncertified zone is matched again)
          const bool badpx = !(fabs(bd) <= 1.7976931348623157e308);
          bool okpx = zcertified<COST, T32, REL32>(C.zc[it.zone], D, badpx, bd * sl, sd * sl, sl * sl, rq * rq);
          if (EDGE) {
            const ZEdge e1{elo, ehi, z.bx + ox + c - C.edge_k};
            if (fnan[m]) okpx = !e1.notfar(0);                    // the reference's winner is candidate 0, whatever follows: fine iff that one is far
            else if (!okpx && !badpx && !e1.notfar(bx_))          // the best is far and ahead of everything that is not
              okpx = bnf[m] == kBestInit || zcertified<COST, T32, REL32>(C.zc[it.zone], D, false, bd * sl, nfd * sl, sl * sl, rq * rq);
          }
          if (!okpx) uncert = true;
        } else {
          o[2] = (best[m] == worst[m]) ? 0 : 0x7fffffff;
        }
      }
    }
  }
  return uncert;
}

// T32: the certified pass as two tiers in ONE launch — an item that finishes its tile runs in float32 first; a tile with a pixel the fp32
// tier cannot prove runs again in float64 (same workgroup, same LDS), and only what THAT cannot prove flags the zone for the reference's
// order.  The runs of a CUT tile stay in float64: their records are certified after the merge, and an unproven tile would have to be matched
// again over ALL its disparities by one workgroup — measured on a LoG + NCC pyramid tile (41 of ~2200 tiles unproven in float32): the redo
// launches took 0.39 ms, four times what the float32 runs had saved.
template <int COST, int KS, typename ACC, bool CERT, int ZS, bool EDGE, bool T32>
__global__ void __launch_bounds__(ZS * ZS / 4, 4)
bm_zones_kernel(ZLaunch G, const vwgpu_zone_task* __restrict__ zones, const ZItem* __restrict__ items) {
  extern __shared__ char smem[];
  static_assert(!T32 || (CERT && sizeof(ACC) == 8), "T32 wraps the float64 certified kernel");
  const ZItem it = items[blockIdx.x];
  if (it.gate >= 0 && G.P.redo[it.gate] == 0) return;             // redo launch: only the tiles the merge flagged
  const vwgpu_zone_task z = zones[it.zone];
  ZGeom geom;
  if (!ztile_geometry<ZS>(z, it.zone, it.txy, G.C.need, G.C.cells, geom)) return;          // (workgroup-uniform)
  bool uncert;
  int tier = 0;
  if constexpr (T32) {
    if (it.slot >= 0) {                                           // a run of a cut tile: float64 records for zones_merge_kernel (see above)
      uncert = zmatch_item<COST, KS, double, true, ZS, EDGE>(G, it, z, geom, smem);
    } else {
      uncert = zmatch_item<COST, KS, float, true, ZS, EDGE>(G, it, z, geom, smem);
      if (__syncthreads_or(uncert ? 1 : 0)) {                     // (__syncthreads_or is also the barrier between the two uses of the LDS)
        tier = 1;
        uncert = zmatch_item<COST, KS, double, true, ZS, EDGE>(G, it, z, geom, smem);
      }
    }
  } else {
    uncert = zmatch_item<COST, KS, ACC, CERT, ZS, EDGE>(G, it, z, geom, smem);
  }
  if (CERT && it.slot < 0) {
    const int any = __syncthreads_or(uncert ? 1 : 0);
    if (threadIdx.x == 0) {
      if (any) {
        G.C.zflag[it.zone] = 1; if (G.C.any) G.C.any[z.img] = 1;
        if (G.C.tflag) atomicMax(&G.C.tflag[G.C.zc[it.zone].trow + (int)((unsigned)it.txy >> 16)], (int)(it.txy & 0xffff) + 1);      // (ty in the upper half: up to 65535 tile rows; the value: 1 + the band's last flagged tile column)
      }
      if (G.C.stats) {
        atomicAdd(&G.C.stats[any ? 1 : 0], (unsigned long long)(geom.tw * geom.th));
        if (tier) atomicAdd(&G.C.stats[2], (unsigned long long)(geom.tw * geom.th));      // pixels of tiles the fp32 tier passed on
      }
    }
  }
}

// Folds the runs of a tile in index order (see the file header): (value, first index) minimum, extremum, runner-up, largest right precision.
template <int COST, bool CERT, int ZS>
__global__ void __launch_bounds__(ZS * ZS / 4)
zones_merge_kernel(const vwgpu_zone_task* __restrict__ zones, const ZMergeItem* __restrict__ items, PrecView pa, size_t pa_tile,
                   int32_t* __restrict__ out, ZPart P, ZCertArgs C) {
  constexpr int ZT = ZS;
  const ZMergeItem it = items[blockIdx.x];
  const vwgpu_zone_task z = zones[it.zone];
  if (CERT && COST == VWGPU_CROSS_CORRELATION) pa.p += (size_t)z.img * pa_tile;
  ZGeom geom;
  if (!ztile_geometry<ZT>(z, it.zone, it.txy, C.need, C.cells, geom)) return;
  const int ox = geom.ox, oy = geom.oy, tw = geom.tw, th = geom.th;
  const int t = threadIdx.x;
  const int c = t % ZT, y0 = (t / ZT) * 4;
  // the slot flags of the runs: one load per lane instead of a chain of dependent scalar loads
  const bool bad = __syncthreads_or(t < it.nitems && P.bad[it.slot0 + t] != 0) != 0;
  if (bad && !CERT) {                                           // order dependent: the redo launch recomputes the tile as one item
    if (t == 0) P.redo[it.gate] = 1;
    return;
  }
  // Every record plane is a full ZT x ZT block, so the loads need no bounds: all four rows of a lane are read for run k + 1 while run k is
  // folded, and the fold is selects (the branchy form waited out a memory round trip per run and row: 25 us for a few hundred tiles).
  struct Rec { double b, x, rp, nf; int idx; };
  const bool edge = CERT && C.edge_m > 0;                        // (workgroup-uniform) the runs carry their best not-far cost, see ZEdge
  const size_t px0 = (size_t)y0 * ZT + c;
  auto load = [&](int k, int m) __attribute__((always_inline)) {
    const size_t o = (size_t)(it.slot0 + k) * (ZT * ZT) + px0 + (size_t)m * ZT;
    Rec r;
    r.b = P.best[o]; r.x = CERT ? P.second[o] : P.worst[o]; r.idx = P.idx[o];
    r.rp = (CERT && COST == VWGPU_CROSS_CORRELATION) ? P.rpmax[o] : 0.0;
    r.nf = edge ? P.bnf[o] : 0.0;
    return r;
  };
  double best[4], other[4], rpmax[4], bnf[4];
  int bi[4];
  bool fn[4] = {false, false, false, false};                     // run 0 started the search with a NaN cost (bit 31 of its index record)
  Rec nxt[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) nxt[m] = load(0, m);
  for (int k = 0; k < it.nitems; ++k) {
    Rec cur[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) cur[m] = nxt[m];
    if (k + 1 < it.nitems) {
#pragma unroll
      for (int m = 0; m < 4; ++m) nxt[m] = load(k + 1, m);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const double b = cur[m].b, x = cur[m].x;
      if (k == 0) { best[m] = b; other[m] = x; bi[m] = cur[m].idx & 0x7fffffff; fn[m] = cur[m].idx < 0; rpmax[m] = cur[m].rp; bnf[m] = cur[m].nf; }
      else {
        const bool better = zbetter<COST>(b, best[m]);          // strictly better: ties stay with the earlier run (first wins)
        if (CERT) {
          // runner-up of the union: the run's own runner-up or the old best when the run wins, else the run's best if it beats the old runner-up
          const double when_better = zbetter<COST>(x, best[m]) ? x : best[m];
          const double when_not = zbetter<COST>(b, other[m]) ? b : other[m];
          other[m] = better ? when_better : when_not;
          rpmax[m] = fmax(rpmax[m], cur[m].rp);
          bnf[m] = zbetter<COST>(cur[m].nf, bnf[m]) ? cur[m].nf : bnf[m];
        } else {
          other[m] = !zbetter<COST>(x, other[m]) ? x : other[m];      // the extremum (a NaN never gets here: bad tiles left above)
        }
        bi[m] = better ? (cur[m].idx & 0x7fffffff) : bi[m];
        best[m] = better ? b : best[m];
      }
    }
  }
  bool uncert = false;
  if (c < tw) {
    const int D = z.sx * z.sy;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int y = y0 + m;
      if (y >= th) continue;
      int32_t* o3 = out + ((size_t)z.out_off + (size_t)(oy + y) * z.out_stride + ox + c) * 3;
      const int bsel = fn[m] ? 0 : bi[m];                        // (a NaN first candidate is the reference's winner)
      const int by_ = bsel / z.sx, bx_ = bsel - by_ * z.sx;
      o3[0] = bx_ + z.addx; o3[1] = by_ + z.addy;
      o3[2] = CERT ? (D == 1 ? 0 : 0x7fffffff) : ((best[m] == other[m]) ? 0 : 0x7fffffff);
      if (CERT) {
        double lprec = 0.0;
        if (COST == VWGPU_CROSS_CORRELATION) lprec = pa.p[(size_t)(z.ay + oy + y - pa.y0) * pa.w + (z.ax + ox + c - pa.x0)];
        const double sl = COST == VWGPU_CROSS_CORRELATION ? lprec : 1.0;       // (square roots of precisions, as in bm_zones_kernel)
        bool okpx = zcertified<COST>(C.zc[it.zone], D, bad, best[m] * sl, other[m] * sl, sl * sl, rpmax[m] * rpmax[m]);
        if (edge) {                                                // the "cannot matter" certificate, as in bm_zones_kernel
          constexpr double kInit = COST == VWGPU_CROSS_CORRELATION ? -INFINITY : INFINITY;
          const ZEdge e1{C.zc[it.zone].edge_lo, C.zc[it.zone].edge_hi, z.bx + ox + c - C.edge_k};
          if (fn[m]) okpx = !e1.notfar(0);
          else if (!okpx && !bad && fabs(best[m]) <= 1.7976931348623157e308 && !e1.notfar(bx_))
            okpx = bnf[m] == kInit || zcertified<COST>(C.zc[it.zone], D, false, best[m] * sl, bnf[m] * sl, sl * sl, rpmax[m] * rpmax[m]);
        }
        if (!okpx) uncert = true;
      }
    }
  }
  if (CERT) {
    const int any = __syncthreads_or(uncert ? 1 : 0);
    if (t == 0) {
      if (any) {
        C.zflag[it.zone] = 1; if (C.any) C.any[z.img] = 1;
        if (C.tflag) atomicMax(&C.tflag[C.zc[it.zone].trow + (int)((unsigned)it.txy >> 16)], (int)(it.txy & 0xffff) + 1);
      }
      if (C.stats) { atomicAdd(&C.stats[any ? 1 : 0], (unsigned long long)(tw * th)); }
    }
  }
}

// Per zone: cross_corr_consistency_check(crop(disparity, zone), rl_zone, thr), then += (addx, addy).
__global__ void zone_lr_kernel(const vwgpu_zone_task* __restrict__ zones, const int2* __restrict__ tiles,
                               int32_t* __restrict__ l2r, const int32_t* __restrict__ r2l, float thr,
                               float* __restrict__ diff2, ptrdiff_t dstride) {
  const int2 tl = tiles[blockIdx.x];
  const vwgpu_zone_task z = zones[tl.x];          // zw/zh/out_* = the L->R zone; bx/by = size of its R->L image;
  const int ox = (tl.y & 0xffff) * ZT, oy = (tl.y >> 16) * ZT;   // ax = element offset of that image in r2l
  const int c = ox + (threadIdx.x & 31);
  for (int r = oy + (threadIdx.x >> 5); r < min(oy + ZT, z.zh); r += ZTHREADS / 32) {
    if (c >= z.zw) continue;
    int32_t* p = l2r + ((size_t)z.out_off + (size_t)r * z.out_stride + c) * 3;
    const int dx = p[0], dy = p[1], v = p[2];
    const int x = c + dx, y = r + dy;
    bool keep = false;
    if (x >= 0 && x < z.bx && y >= 0 && y < z.by) {
      const int32_t* q = r2l + ((size_t)z.ax + (size_t)y * z.bx + x) * 3;
      if (v != 0 && q[2] != 0) {
        const float diff = (float)fmax(fabs((double)(dx + q[0])), fabs((double)(dy + q[1])));
        keep = thr >= diff;
        if (keep && diff2) {                        // lr_disp_diff(c + ul.x, r + ul.y) = PixelMask<float>(disp_diff)
          float* d = diff2 + ((ptrdiff_t)(r + z.sy) * dstride + (c + z.sx)) * 2;
          d[0] = diff; d[1] = 1.0f;
        }
      }
    }
    p[0] = dx + z.addx; p[1] = dy + z.addy;
    if (!keep) p[2] = 0;
  }
}

// Which part of its R->L image will a zone's L/R check read?  zone_lr_kernel looks at (c + dx, r + dy) for every valid pixel (c, r) of the
// L->R result: per zone the bounding rectangle of those positions as four maxima over a zeroed record — {w - x, h - y, x + 1, y + 1}
// (w x h = the R->L image) — and a flag per 16 x 16 cell of the R->L image that holds one (z.ay = the zone's first cell).  A zone whose
// L->R result is not final yet (flagged by the certified pass: the exact-order kernels will match it again) asks for everything.
// Same tasks and tiles as zone_lr_kernel.
// maximum over the 64 lanes of a wavefront with DPP moves only (the pattern of sgm.hip's wave_min_u32)
__device__ __forceinline__ unsigned zwave_max_u32(unsigned v) {
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));   // row_half_mirror
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));   // row_mirror
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xA, 0xF, false));   // row_bcast:15 -> rows 1, 3
  v = max(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xC, 0xF, false));   // row_bcast:31 -> rows 2, 3
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__global__ void __launch_bounds__(ZTHREADS)
zone_need_kernel(const vwgpu_zone_task* __restrict__ zones, const int2* __restrict__ tiles, const int32_t* __restrict__ l2r,
                 const int* __restrict__ zflag, int* __restrict__ need, unsigned char* __restrict__ cells) {
  // the cell flags of the zone are gathered in shared memory first and only the set ones go out, once per workgroup: the stores of
  // the 1024 workgroups of a 512 x 512 zone all land on the same two dozen cache lines (0.1 ms when every pixel stored its own)
  constexpr int LCELLS = 8192;
  __shared__ unsigned lc32[LCELLS / 4];
  unsigned char* lc = reinterpret_cast<unsigned char*>(lc32);
  const int2 tl = tiles[blockIdx.x];
  const vwgpu_zone_task z = zones[tl.x];
  int* rec = need + 8 * tl.x;
  if (zflag && zflag[tl.x]) {
    if (threadIdx.x == 0 && tl.y == 0) { rec[0] = z.bx; rec[1] = z.by; rec[2] = z.bx; rec[3] = z.by; rec[5] = 1; }      // (the zone's first tile; the others do not touch the record)
    return;
  }
  // A zone of many tiles gets the whole rectangle (its cell flags do the work): the wavefronts of a 512 x 512 zone would queue 4096
  // atomic updates on one cache line — 35 ns each, 0.14 ms.
  const bool whole = ((z.zw + ZT - 1) / ZT) * ((z.zh + ZT - 1) / ZT) > 16;
  if (threadIdx.x == 0 && tl.y == 0) {
    rec[4] = z.ay;
    if (whole) { rec[0] = z.bx; rec[1] = z.by; rec[2] = z.bx; rec[3] = z.by; }
  }
  const int ncx = (z.bx + 15) >> 4, ncells = ncx * ((z.by + 15) >> 4);
  const bool local = whole && ncells <= LCELLS;                 // (a small zone's few workgroups store their flags directly)
  if (local) {
    for (int i = threadIdx.x; i < (ncells + 3) / 4; i += ZTHREADS) lc32[i] = 0u;
    __syncthreads();
  }
  const int ox = (tl.y & 0xffff) * ZT, oy = (tl.y >> 16) * ZT;
  const int c = ox + (threadIdx.x & 31);
  int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  if (c < z.zw)
    for (int r = oy + (threadIdx.x >> 5); r < min(oy + ZT, z.zh); r += ZTHREADS / 32) {
      const int32_t* p = l2r + ((size_t)z.out_off + (size_t)r * z.out_stride + c) * 3;
      if (p[2] == 0) continue;
      const int x = c + p[0], y = r + p[1];
      if (x < 0 || x >= z.bx || y < 0 || y >= z.by) continue;
      a0 = max(a0, z.bx - x); a1 = max(a1, z.by - y); a2 = max(a2, x + 1); a3 = max(a3, y + 1);
      const int ci = (y >> 4) * ncx + (x >> 4);
      if (local) lc[ci] = 1; else cells[z.ay + ci] = 1;
    }
  if (!whole) { a0 = (int)zwave_max_u32((unsigned)a0); a1 = (int)zwave_max_u32((unsigned)a1); a2 = (int)zwave_max_u32((unsigned)a2); a3 = (int)zwave_max_u32((unsigned)a3); }
  if (!whole && (threadIdx.x & 63) == 0 && a2 > 0) { atomicMax(rec + 0, a0); atomicMax(rec + 1, a1); atomicMax(rec + 2, a2); atomicMax(rec + 3, a3); }
  if (local) {
    __syncthreads();
    for (int i = threadIdx.x; i < ncells; i += ZTHREADS)
      if (lc[i]) cells[z.ay + i] = 1;
  }
}

// Tables of one launch sequence, side by side in one half of the ztab arena (the previous sequence may still be reading the other half):
// pieces[i] = {host pointer, bytes}; d[i] receives the device address.  One asynchronous copy from the pinned ring when they fit.
int upload_pieces(vwgpu_ctx* ctx, const void* const* src, const size_t* bytes, int n, char** d, size_t zero_tail, char** d_zero) {
  size_t off[8], all = 0;
  for (int i = 0; i < n; ++i) { off[i] = all; all += vwgpu_align_up(bytes[i], 256); }
  const size_t zoff = all;
  all += vwgpu_align_up(zero_tail, 256);
  const size_t half = vwgpu_align_up(all, 4096);
  if (ctx->ztab.cap < 2 * half) {
    int rc = vwgpu_arena_reserve(ctx, &ctx->ztab, 2 * half + (1 << 20));
    if (rc) return rc;
  }
  ctx->ztab_parity ^= 1;
  char* base = static_cast<char*>(ctx->ztab.base) + (ctx->ztab_parity ? ctx->ztab.cap / 2 : 0);
  // a small zeroed tail travels as zeros in the same copy (a fill of its own is one more 5 us launch per launch sequence)
  const bool tail_in_copy = zero_tail > 0 && zero_tail <= 64 * 1024;
  if (char* h = static_cast<char*>(vwgpu_host_ring(ctx, tail_in_copy ? zoff + zero_tail : zoff))) {
    for (int i = 0; i < n; ++i) if (bytes[i]) memcpy(h + off[i], src[i], bytes[i]);
    if (tail_in_copy) memset(h + zoff, 0, zero_tail);
    VWGPU_HIP(ctx, hipMemcpyAsync(base, h, tail_in_copy ? zoff + zero_tail : zoff, hipMemcpyHostToDevice, ctx->stream));
    if (tail_in_copy) zero_tail = 0;
  } else {
    for (int i = 0; i < n; ++i)
      if (bytes[i]) VWGPU_HIP(ctx, hipMemcpyAsync(base + off[i], src[i], bytes[i], hipMemcpyHostToDevice, ctx->stream));
    VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));          // the host vectors go out of scope
  }
  if (zero_tail) VWGPU_HIP(ctx, hipMemsetAsync(base + zoff, 0, zero_tail, ctx->stream));
  for (int i = 0; i < n; ++i) d[i] = base + off[i];
  if (d_zero) *d_zero = base + zoff;
  return VWGPU_OK;
}

int upload_tables(vwgpu_ctx* ctx, const vwgpu_zone_task* zones, int n, std::vector<int2> const& tiles,
                  const vwgpu_zone_task** d_zones, const int2** d_tiles) {
  const void* src[2] = {zones, tiles.data()};
  const size_t bytes[2] = {(size_t)n * sizeof(vwgpu_zone_task), tiles.size() * sizeof(int2)};
  char* d[2];
  int rc = upload_pieces(ctx, src, bytes, 2, d, 0, nullptr);
  if (rc) return rc;
  *d_zones = reinterpret_cast<const vwgpu_zone_task*>(d[0]);
  *d_tiles = reinterpret_cast<const int2*>(d[1]);
  return VWGPU_OK;
}

void build_tiles(const vwgpu_zone_task* zones, int n, std::vector<int2>& tiles) {
  tiles.clear();
  for (int i = 0; i < n; ++i) {
    const int nx = (zones[i].zw + ZT - 1) / ZT, ny = (zones[i].zh + ZT - 1) / ZT;
    for (int ty = 0; ty < ny; ++ty)
      for (int tx = 0; tx < nx; ++tx) tiles.push_back(make_int2(i, tx | (ty << 16)));
  }
}

// bound on |any-order float64 sum - the reference's running sum| of a kx x ky window over a W x H output region, in units of the
// largest element magnitude: the reference's column chain (Algorithms.h:62-75,100-103) makes 2 roundings per row step on values below
// (ky + 1) elements, its row chain (:84-92) 2 per column step on values below kx ky + 2 ky elements, the inherited column errors
// telescope along a row, and the tile-parallel sums (kx + 5 and ky + 5 additions of at most kx ky elements) are a lower-order term:
//   ref <= u [kx ky^2 + 2 kx H (ky + 1) + kx ky (kx - 1) + W ky (kx + 2)],  tile <= u (kx + 2)(ky + 2)(kx + ky + 9)
// both below u (kx + 2)(ky + 2)(2 kx + 2 ky + 2 W + 2 H + 9); the constant 8 leaves a factor > 3 for the second-order terms.
double sum_error_units(int kx, int ky, int W, int H) {
  return 8.0 * (kx + 2.0) * (ky + 2.0) * ((double)kx + ky + W + H + 5.0) * 0x1p-53;
}

}  // namespace

bool vwgpu_bm_zones_supported(int kx, int ky) {
  // LDS: left patch + right patch with at least 8 disparities per chunk + two sum planes within 64 KB
  const size_t PW = (ZT + kx - 1) | 1, PH = ZT + ky - 1;
  return (PH * PW + PH * (PW + 8)) * 4 + 2 * PH * (ZT + 1) * 8 + 16 <= 64 * 1024;
}

// cert_hi: INT_MIN = no certification (the level is order free: any summation order returns the reference's bits — which includes "every
// pixel finite", vwgpu_sums_bits: the SAD / SSD chain relies on finite costs).  Otherwise the
// largest binary exponent of the level's pixels (|pixel| < 2^(cert_hi + 1), all finite): the kernels certify every pixel against the
// error bound above and raise d_zflag[zone] (n ints, zeroed by the CALLER) for zones with a pixel they cannot certify; the caller redoes those
// zones in the reference's order (vwgpu_launch_bm_exact with the same flags as its gate).
namespace {
// the work of one tile size of a launch sequence
struct ZPlan {
  int zs = 32;                                   // tile side
  int sxc = 1;                                   // dx per right patch
  size_t lds = 0;
  std::vector<ZItem> items, redo;
  std::vector<ZMergeItem> merges;
  int nslots = 0;
  ZPart P{};
  const ZItem* d_items = nullptr; const ZItem* d_redo = nullptr; const ZMergeItem* d_merges = nullptr;
};
size_t zones_lds_fixed(int zs, int kx, int ky, size_t accb) { return (size_t)(zs + ky - 1) * ((zs + kx - 1) | 1) * 4 + 2 * (size_t)(zs + ky - 1) * (zs + 1) * accb + 16; }
}  // namespace

int vwgpu_launch_bm_zones(vwgpu_ctx* ctx, int cost_type, const float* A, int aw, int ah, const float* B, int bw, int bh,
                          int kx, int ky, const vwgpu_zone_task* zones, int n, int32_t* out, int f32_sums, int cert_hi, int* d_zflag,
                          unsigned long long* d_stats, int* d_any, const int* d_need, const unsigned char* d_cells,
                          int edge_m, int edge_k, int edge_lo, int edge_hi, ptrdiff_t as, ptrdiff_t bs, const vwgpu_zone_group* grp, int* d_tflag) {
  const int n_img = grp ? std::max(1, grp->n_img) : 1;
  if (grp && grp->cert_hi) {                                      // (a group is certified as a whole or not at all: the caller sorts its image pairs into such groups)
    cert_hi = INT_MIN;
    for (int i = 0; i < n_img; ++i) cert_hi = std::max(cert_hi, grp->cert_hi[i]);
  }
  for (int i = 0; i < n; ++i)
    if (zones[i].img < 0 || zones[i].img >= n_img) return vwgpu_fail(ctx, VWGPU_ERR_LOGIC, "bm_zones: zone %d names image pair %d of %d", i, zones[i].img, n_img);
  const bool cert = cert_hi != INT_MIN;
  // the fp32 tier of the certified pass (VWGPU_OPT_CERT_F32, default on): compile-time square windows only
  const bool t32 = cert && ctx->cert_f32 && kx == ky && kx >= 3 && kx <= 13;
  if (as == 0) as = aw;
  if (bs == 0) bs = bw;
  if (as > INT32_MAX || bs > INT32_MAX) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "bm_zones: row stride too large");
  const int ap = (int)as, bp = (int)bs;
  if (cost_type == VWGPU_CROSS_CORRELATION || cert) f32_sums = 0;        // (NCC sums are scaled in float64 anyway)
  const size_t accb = f32_sums ? 4 : 8;                           // (the fp32 tier keeps four float32 planes — two disparities per step — in the room of the two float64 ones)
  if (n <= 0) return VWGPU_OK;
  if (!vwgpu_bm_zones_supported(kx, ky)) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "bm_zones: kernel %dx%d too large", kx, ky);
  if (cert && !d_zflag) return vwgpu_fail(ctx, VWGPU_ERR_LOGIC, "bm_zones: certification without zone flags");
  const bool ncc = cost_type == VWGPU_CROSS_CORRELATION;
  int max_sx = 1;
  int ax0 = INT32_MAX, ay0 = INT32_MAX, ax1 = INT32_MIN, ay1 = INT32_MIN, bx0 = INT32_MAX, by0 = INT32_MAX, bx1 = INT32_MIN, by1 = INT32_MIN;
  double total = 0.0;                                            // evaluations of the level
  for (int i = 0; i < n; ++i) {
    const vwgpu_zone_task& z = zones[i];
    if (z.zw <= 0 || z.zh <= 0 || z.sx <= 0 || z.sy <= 0) continue;
    if ((long long)z.sx * z.sy > INT32_MAX) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "bm_zones: search volume exceeds the index range");
    max_sx = std::max(max_sx, z.sx);
    total += (double)z.zw * z.zh * z.sx * z.sy;
    ax0 = std::min(ax0, z.ax); ay0 = std::min(ay0, z.ay); ax1 = std::max(ax1, z.ax + z.zw); ay1 = std::max(ay1, z.ay + z.zh);
    bx0 = std::min(bx0, z.bx); by0 = std::min(by0, z.by);
    bx1 = std::max(bx1, z.bx + z.zw + z.sx - 1); by1 = std::max(by1, z.by + z.zh + z.sy - 1);
  }
  if (total == 0.0) return VWGPU_OK;

  ZPlan plan[2];
  plan[0].zs = 32; plan[1].zs = 16;
  for (ZPlan& pl : plan) {
    const size_t PW = pl.zs + kx - 1, PH = pl.zs + ky - 1;
    const size_t fixed = zones_lds_fixed(pl.zs, kx, ky, accb);
    const size_t budget = pl.zs == 32 ? 64 * 1024 : 20 * 1024;     // a wavefront-sized tile keeps its right patch short: more of them fit a CU
    long long room = (long long)((budget - std::min(budget, fixed)) / (PH * 4)) - (long long)PW;            // (the right pitch is rounded up to odd)
    room = std::min<long long>(room, ctx->zone_sxc > 0 ? ctx->zone_sxc : 16);      // (measured: 16 dx per patch keep four workgroups on a CU; longer patches three)
    pl.sxc = (int)std::max<long long>(1, std::min<long long>(max_sx, room));
    const size_t LPW = PW | 1, RPW = (PW + pl.sxc - 1) | 1;
    pl.lds = (((PH * LPW + PH * RPW) * 4 + 7) & ~size_t(7)) + 2 * PH * (pl.zs + 1) * accb;
  }

  // Work items.  A tile is cut when its evaluations exceed `cap` = a third of the level's work per resident workgroup (so that the longest
  // item cannot hold the launch much longer than an even spread would take), never below 16 disparities of a full tile (the left patch,
  // the records and the merge are per item).  Items are issued longest first.
  // Tile size per zone: 16 x 16 tiles when they cover the zone with less padded work than 32 x 32 tiles (a 16-tile costs ~1.4x per pixel:
  // its patches carry more halo) — the 16 x 16 leaves of the quad tree and the thin zones.
  plan[0].items.reserve((size_t)n * 2 + 64);
  const int resident = std::max(1, (int)std::min<size_t>(8, (160 * 1024) / (plan[0].lds + 512))) * ctx->num_cu;
  // (a level that cannot fill the GPU with 16-disparity runs — the coarse levels of a tile: one zone of a few dozen tiles — is cut down to
  // 4-disparity runs: its launch is as long as its longest item)
  const double min_run = std::min(16.0, std::max(4.0, total / (1024.0 * resident)));
  double cap = std::max(min_run * 32 * 32, total / (3.0 * resident));
  // A level whose WHOLE tiles already are a round of the resident workgroups, none much longer than the mean — level 1 of a tile group: four
  // tiles' 512 x 512 zones x 60 disparities = 1024 equal items — is not cut: its runs would leave 44 bytes of records per pixel and run,
  // a merge launch, and the cut tiles stay out of the fp32 tier.
  {
    long long ntiles = 0;
    double longest = 0.0;
    for (int i = 0; i < n; ++i) {
      const vwgpu_zone_task& z = zones[i];
      if (z.zw <= 0 || z.zh <= 0 || z.sx <= 0 || z.sy <= 0) continue;
      ntiles += (long long)((z.zw + 31) / 32) * ((z.zh + 31) / 32);
      longest = std::max(longest, (double)std::min(32, z.zw) * std::min(32, z.zh) * z.sx * z.sy);
    }
    if (ntiles >= resident && longest * (double)ntiles <= 1.5 * total) cap = std::max(cap, longest);
  }
  for (int i = 0; i < n; ++i) {
    const vwgpu_zone_task& z = zones[i];
    if (z.zw <= 0 || z.zh <= 0 || z.sx <= 0 || z.sy <= 0) continue;
    // (Round 4 also built 16 x 16 tiles on one wavefront for the 16 x 16 leaves of the quad tree — on a 32 x 32 tile a quarter of the lanes
    // have pixels there.  Measured on 1024^2 pyramid tiles: every zone on 16-tiles = the same time (their patches carry 2.6x halo and a
    // wavefront per workgroup exposes every LDS round trip); both sizes in one level = two launches with a tail each, slower.  Dropped.)
    // Round 5: with TILE GROUPS a launch holds the zones of several tiles, the tails are shared, and the leaves pay again: zones that fit a
    // 16 x 16 tile go to the one-wavefront kernels when VWGPU_OPT_ZONE_TILE16 says so (1 = always, 0 = in group launches, 2 = never).
    const bool small = z.zw <= 16 && z.zh <= 16 && (ctx->zone_tile16 == 1 || (ctx->zone_tile16 == 0 && n_img > 1));
    ZPlan& pl = small ? plan[1] : plan[0];
    const int ZS = pl.zs;
    const int nx = (z.zw + ZS - 1) / ZS, ny = (z.zh + ZS - 1) / ZS;
    const long long D = (long long)z.sx * z.sy;
    for (int ty = 0; ty < ny; ++ty)
      for (int tx = 0; tx < nx; ++tx) {
        const int tw = std::min(ZS, z.zw - tx * ZS), th = std::min(ZS, z.zh - ty * ZS);
        const double px = (double)tw * th;
        // (Tried: runs of unequal length, 1.35 ... 0.65 or 1.6 ... 0.4 of the mean, so that the short ones — issued last — fill the end of
        // the launch with the same number of records: -1.5 % on the NCC launches, +7 % on SAD (tools/zones_ab.py medians).)
        // (Tried: three tiles in ten cut four times finer, so that short items fill the end of the launch, where a third of the wave slots
        // stand empty for a third of the span — tools/zones_timeline.py.  The matcher launches got 10 % shorter and the merges of the
        // extra records took it back: 28 bytes per pixel and run, written and read.)
        int pieces = (int)std::min<double>((double)D, std::ceil(px * (double)D / cap));
        if (pieces < 1) pieces = 1;
        const int txy = tx | (ty << 16);
        if (pieces == 1) {
          pl.items.push_back(ZItem{i, txy, 0, (int)D, -1, -1, (int)(px * (double)D), 0});
          continue;
        }
        const int per = (int)((D + pieces - 1) / pieces);
        const int gate = (int)pl.merges.size();
        pl.merges.push_back(ZMergeItem{i, txy, pl.nslots, 0, gate, 0, 0, 0});
        for (long long i0 = 0; i0 < D; i0 += per) {
          const int cnt = (int)std::min<long long>(per, D - i0);
          pl.items.push_back(ZItem{i, txy, (int)i0, cnt, pl.nslots++, -1, (int)(px * cnt), 0});
          pl.merges.back().nitems++;
        }
        pl.redo.push_back(ZItem{i, txy, 0, (int)D, -1, gate, 0, 0});
      }
  }
  // longest first (pad0 = the item's evaluations), as a counting sort on a 5-bit logarithm of the length — a level of 2000 leaf zones has
  // 4000 items and a comparison sort of them was a tenth of a millisecond of host time with the device waiting; items of one bin keep
  // their zone order
  for (ZPlan& pl : plan) {
    if (pl.items.size() < 2) continue;
    auto bin = [](const ZItem& it) {
      const unsigned v = (unsigned)std::max(it.pad0, 1);
      const int e = 31 - __builtin_clz(v);
      const int m = e >= 2 ? (int)((v >> (e - 2)) & 3) : 0;
      return 127 - (e * 4 + m);                                 // 0 = the longest
    };
    unsigned count[129] = {0};
    for (const ZItem& it : pl.items) count[bin(it) + 1]++;
    for (int b = 0; b < 128; ++b) count[b + 1] += count[b];
    std::vector<ZItem> sorted(pl.items.size());
    for (const ZItem& it : pl.items) sorted[count[bin(it)]++] = it;
    pl.items.swap(sorted);
  }
  if (plan[0].items.empty() && plan[1].items.empty()) return VWGPU_OK;

  PrecView pa{nullptr, 0, 0, 0, 0, nullptr}, pb{nullptr, 0, 0, 0, 0, nullptr};
  // partial records of the cut tiles behind the precision images in the scratch arena
  // the "cannot matter" certificate (ZEdge) only where a zone of the pass has far candidates at all: its kernels cost 4 instructions per evaluation more
  bool edge = false;
  if (cert && edge_m > 0)
    for (int i = 0; i < n && !edge; ++i) {
      const vwgpu_zone_task& z = zones[i];
      if (z.zw <= 0 || z.zh <= 0 || z.sx <= 0 || z.sy <= 0) continue;
      const long long lo = (long long)z.bx - edge_k, hi = lo + z.zw - 1 + z.sx - 1;        // partner columns of the zone's candidates
      const int elo = grp && grp->edge_lo ? grp->edge_lo[z.img] : edge_lo, ehi = grp && grp->edge_hi ? grp->edge_hi[z.img] : edge_hi;
      edge = lo < (long long)elo || hi > (long long)ehi;
    }
  if (!edge) edge_m = 0;
  const size_t rec_bytes = 8 + 8 + 4 + (cert ? 8 : 0) + (cert && ncc ? 8 : 0) + (edge ? 8 : 0);
  size_t part_bytes = 1024;
  for (ZPlan& pl : plan) part_bytes += vwgpu_align_up((size_t)pl.nslots * pl.zs * pl.zs * rec_bytes + 64, 256);
  size_t na = 0, nb = 0, nb32 = 0;
  if (ncc) {
    pa.x0 = ax0; pa.y0 = ay0; pa.w = ax1 - ax0; pa.h = ay1 - ay0;
    pb.x0 = bx0; pb.y0 = by0; pb.w = bx1 - bx0; pb.h = by1 - by0;
    na = vwgpu_align_up((size_t)pa.w * pa.h * 8, 256); nb = vwgpu_align_up((size_t)pb.w * pb.h * 8, 256);
    if (t32) nb32 = vwgpu_align_up((size_t)pb.w * pb.h * 4, 256);      // the right factors once more in float32 (the chain of the fp32 tier reads them per evaluation)
  }
  // (groups: one precision image pair per image pair, all over the common origin rectangle, at a fixed stride)
  const size_t pa_tile = (size_t)pa.w * pa.h, pb_tile = (size_t)pb.w * pb.h;
  na *= (size_t)n_img; nb *= (size_t)n_img; nb32 *= (size_t)n_img;
  int rc = vwgpu_arena_reserve(ctx, &ctx->scratch, na + nb + nb32 + part_bytes);
  if (rc) return rc;
  char* sbase = static_cast<char*>(ctx->scratch.base);
  if (ncc) {
    double* da = reinterpret_cast<double*>(sbase);
    double* db = reinterpret_cast<double*>(sbase + na);
    float* db32 = t32 ? reinterpret_cast<float*>(sbase + na + nb) : nullptr;
    vwgpu_prof_scope ps(ctx, "zone_precision");
    const size_t zp_lds = ((size_t)(64 + kx - 1) * (ZP_TH + ky - 1) + (size_t)(ZP_TH + ky - 1) * 64) * sizeof(double);
    ZPrecJobs zj;
    zj.j[0] = ZPrecJob{A, aw, ah, ap, da, pa.x0, pa.y0, pa.w, pa.h, nullptr};
    zj.j[1] = ZPrecJob{B, bw, bh, bp, db, pb.x0, pb.y0, pb.w, pb.h, db32};
    zj.img_tile[0] = grp ? grp->a_stride : 0; zj.img_tile[1] = grp ? grp->b_stride : 0;
    zj.prec_tile[0] = pa_tile; zj.prec_tile[1] = pb_tile;
    const int root = cert ? (edge ? 2 : 1) : 0, mw = std::max(pa.w, pb.w), mh = std::max(pa.h, pb.h);
#define VW_ZPSQ(K_) hipLaunchKernelGGL((zone_precision_sq_kernel<K_>), dim3((mw + 64 - K_) / (65 - K_), (mh + 4 * ZP_R - 1) / (4 * ZP_R), 2 * n_img), dim3(256), 0, ctx->stream, zj, root)
    if (kx != ky) hipLaunchKernelGGL(zone_precision_kernel, dim3((mw + 63) / 64, (mh + ZP_TH - 1) / ZP_TH, 2 * n_img), dim3(64, 4), zp_lds, ctx->stream, zj, kx, ky, root);
    else switch (kx) {
      case 3: VW_ZPSQ(3); break;  case 5: VW_ZPSQ(5); break;   case 7: VW_ZPSQ(7); break;
      case 9: VW_ZPSQ(9); break;  case 11: VW_ZPSQ(11); break; case 13: VW_ZPSQ(13); break;
      default: hipLaunchKernelGGL(zone_precision_kernel, dim3((mw + 63) / 64, (mh + ZP_TH - 1) / ZP_TH, 2 * n_img), dim3(64, 4), zp_lds, ctx->stream, zj, kx, ky, root);
    }
#undef VW_ZPSQ
    pa.p = da; pb.p = db; pb.pf = db32;
  }
  {
    char* q = sbase + na + nb + nb32;
    for (ZPlan& pl : plan) {
      const size_t slot_px = (size_t)pl.nslots * pl.zs * pl.zs;
      char* q0 = q;
      pl.P.best = reinterpret_cast<double*>(q); q += slot_px * 8;
      pl.P.worst = reinterpret_cast<double*>(q); q += slot_px * 8;
      if (cert) { pl.P.second = reinterpret_cast<double*>(q); q += slot_px * 8; }
      if (cert && ncc) { pl.P.rpmax = reinterpret_cast<double*>(q); q += slot_px * 8; }
      if (edge) { pl.P.bnf = reinterpret_cast<double*>(q); q += slot_px * 8; }
      pl.P.idx = reinterpret_cast<int*>(q);
      q = q0 + vwgpu_align_up(slot_px * rec_bytes + 64, 256);
    }
  }

  // certification constants per zone
  std::vector<ZCert> zc;
  if (cert) {
    zc.resize((size_t)n);
    int trow = 0;                                                   // the zone's first band flag (vwgpu_zone_row_flags: the same walk)
    double e_el = cost_type == VWGPU_ABSOLUTE_DIFFERENCE ? std::ldexp(1.0, cert_hi + 2)          // |a - b| < 2^(hi + 2)
                        : cost_type == VWGPU_SQUARED_DIFFERENCE ? std::ldexp(1.0, 2 * cert_hi + 4)     // (a - b)^2
                        : std::ldexp(1.0, 2 * cert_hi + 2);                                            // |a b|, a^2
    for (int i = 0; i < n; ++i) {
      const vwgpu_zone_task& z = zones[i];
      if (grp && grp->cert_hi) {                                  // the zone's own image pair bounds its elements
        const int h = grp->cert_hi[z.img];
        e_el = cost_type == VWGPU_ABSOLUTE_DIFFERENCE ? std::ldexp(1.0, h + 2) : cost_type == VWGPU_SQUARED_DIFFERENCE ? std::ldexp(1.0, 2 * h + 4) : std::ldexp(1.0, 2 * h + 2);
      }
      zc[i].edge_lo = grp && grp->edge_lo ? grp->edge_lo[z.img] : edge_lo; zc[i].edge_hi = grp && grp->edge_hi ? grp->edge_hi[z.img] : edge_hi;
      zc[i].trow = trow; zc[i].pad1 = 0;
      trow += (z.zh + 31) / 32;
      zc[i].eps_s = sum_error_units(kx, ky, z.zw, z.zh) * e_el;
      zc[i].eps_ll = zc[i].eps_s;
      zc[i].eps_rr = sum_error_units(kx, ky, z.zw + z.sx - 1, z.zh + z.sy - 1) * e_el;
      zc[i].eps32 = 2.0 * (2.0 * std::max(kx, ky) + 24.0) * kx * ky * 0x1p-24 * e_el;      // the fp32 tier's absolute bound, see zcertified
    }
  }
  const void* src[8] = {zones, zc.data(), plan[0].items.data(), plan[0].merges.data(), plan[0].redo.data(),
                        plan[1].items.data(), plan[1].merges.data(), plan[1].redo.data()};
  const size_t bytes[8] = {(size_t)n * sizeof(vwgpu_zone_task), zc.size() * sizeof(ZCert),
                           plan[0].items.size() * sizeof(ZItem), plan[0].merges.size() * sizeof(ZMergeItem), plan[0].redo.size() * sizeof(ZItem),
                           plan[1].items.size() * sizeof(ZItem), plan[1].merges.size() * sizeof(ZMergeItem), plan[1].redo.size() * sizeof(ZItem)};
  char* d[8]; char* dzero = nullptr;
  // zeroed tail: per tile size the slot flags, then the redo flags of the cut tiles
  const size_t nflags = (size_t)plan[0].nslots + plan[0].merges.size() + plan[1].nslots + plan[1].merges.size();
  rc = upload_pieces(ctx, src, bytes, 8, d, nflags * sizeof(int), &dzero);
  if (rc) return rc;
  const vwgpu_zone_task* dz = reinterpret_cast<const vwgpu_zone_task*>(d[0]);
  {
    int* f = reinterpret_cast<int*>(dzero);
    for (int k = 0; k < 2; ++k) {
      plan[k].d_items = reinterpret_cast<const ZItem*>(d[2 + 3 * k]);
      plan[k].d_merges = reinterpret_cast<const ZMergeItem*>(d[3 + 3 * k]);
      plan[k].d_redo = reinterpret_cast<const ZItem*>(d[4 + 3 * k]);
      plan[k].P.bad = f; f += plan[k].nslots;
      plan[k].P.redo = f; f += plan[k].merges.size();
    }
  }
  if (d_tflag && d_need) return vwgpu_fail(ctx, VWGPU_ERR_LOGIC, "bm_zones: band flags are for passes whose tiles start at the zone's origin");
  ZCertArgs C{cert ? reinterpret_cast<const ZCert*>(d[1]) : nullptr, d_zflag, d_stats, d_any, d_need, d_cells, edge_m, edge_k, d_tflag};
  const size_t a_tile = grp ? grp->a_stride : 0, b_tile = grp ? grp->b_stride : 0;

#define VW_ZN6(C_, K_, A_, T_, S_, E_, T32_) hipLaunchKernelGGL((bm_zones_kernel<C_, K_, A_, T_, S_, E_, T32_>), grd, dim3(S_ * S_ / 4), pl.lds, ctx->stream, \
                                                               ZLaunch{A, aw, ah, ap, B, bw, bh, bp, kx, ky, pl.sxc, pa, pb, out, pl.P, C, a_tile, b_tile, pa_tile, pb_tile}, dz, tab)
#define VW_ZN5(C_, K_, A_, T_, S_, E_) do { if (T_ && (K_) > 0 && use32) VW_ZN6(C_, K_, A_, T_, S_, E_, (T_ && (K_) > 0)); else VW_ZN6(C_, K_, A_, T_, S_, E_, false); } while (0)
#define VW_ZN4(C_, K_, A_, T_, S_) do { if (T_ && edge) VW_ZN5(C_, K_, A_, T_, S_, T_); else VW_ZN5(C_, K_, A_, T_, S_, false); } while (0)
#define VW_ZN3(C_, K_, A_, T_) do { if (pl.zs == 32) VW_ZN4(C_, K_, A_, T_, 32); else VW_ZN4(C_, K_, A_, T_, 16); } while (0)
#define VW_ZN(C_, K_) do { if (cert) VW_ZN3(C_, K_, double, true); else if (f32_sums) VW_ZN3(C_, K_, float, false); else VW_ZN3(C_, K_, double, false); } while (0)
#define VW_ZN_K(C_) do { switch (kx == ky ? kx : 0) { case 3: VW_ZN(C_, 3); break; case 5: VW_ZN(C_, 5); break; case 7: VW_ZN(C_, 7); break; \
                                                     case 9: VW_ZN(C_, 9); break; case 11: VW_ZN(C_, 11); break; case 13: VW_ZN(C_, 13); break; \
                                                     default: VW_ZN(C_, 0); break; } } while (0)
#define VW_ZN_C() do { switch (cost_type) { case VWGPU_CROSS_CORRELATION: VW_ZN_K(VWGPU_CROSS_CORRELATION); break; \
                                            case VWGPU_SQUARED_DIFFERENCE: VW_ZN_K(VWGPU_SQUARED_DIFFERENCE); break; \
                                            default: VW_ZN_K(VWGPU_ABSOLUTE_DIFFERENCE); break; } } while (0)
#define VW_MG2(C_, T_) do { if (pl.zs == 32) hipLaunchKernelGGL((zones_merge_kernel<C_, T_, 32>), mgrd, dim3(256), 0, ctx->stream, dz, pl.d_merges, pa, pa_tile, out, pl.P, C); \
                            else hipLaunchKernelGGL((zones_merge_kernel<C_, T_, 16>), mgrd, dim3(64), 0, ctx->stream, dz, pl.d_merges, pa, pa_tile, out, pl.P, C); } while (0)
#define VW_MG(C_) do { if (cert) VW_MG2(C_, true); else VW_MG2(C_, false); } while (0)
  const bool use32 = t32;
  for (ZPlan& pl : plan) {                                      // the 32-tiles hold the long items: first
    if (pl.items.empty()) continue;
    vwgpu_prof_scope ps(ctx, "bm_zones");
    const dim3 grd((unsigned)pl.items.size());
    const ZItem* tab = pl.d_items;
    VW_ZN_C();
  }
  for (ZPlan& pl : plan) {
    if (pl.merges.empty()) continue;
    {
      vwgpu_prof_scope ps(ctx, "bm_zones_merge");
      const dim3 mgrd((unsigned)pl.merges.size());
      switch (cost_type) {
        case VWGPU_CROSS_CORRELATION: VW_MG(VWGPU_CROSS_CORRELATION); break;
        case VWGPU_SQUARED_DIFFERENCE: VW_MG(VWGPU_SQUARED_DIFFERENCE); break;
        default: VW_MG(VWGPU_ABSOLUTE_DIFFERENCE); break;
      }
    }
    // Non-finite costs (NCC over an all-zero window: 0 * inf) make the chain order dependent: the flagged tiles again, as one item each.
    // Order-free SAD / SSD levels hold finite costs only; with certification a non-finite cost flags the zone instead.
    if (ncc && !cert) {
      vwgpu_prof_scope ps(ctx, "bm_zones_redo");
      const dim3 grd((unsigned)pl.redo.size());
      const ZItem* tab = pl.d_redo;
      VW_ZN_C();
    }
  }
#undef VW_MG
#undef VW_MG2
#undef VW_ZN_C
#undef VW_ZN_K
#undef VW_ZN
#undef VW_ZN3
#undef VW_ZN4
#undef VW_ZN5
#undef VW_ZN6
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}

size_t vwgpu_zone_need_cells(vwgpu_zone_task* zones, int n) {
  size_t cells = 0;
  for (int i = 0; i < n; ++i) {
    zones[i].ay = (int)cells;
    cells += (size_t)((zones[i].bx + 15) >> 4) * ((zones[i].by + 15) >> 4);
  }
  return cells;
}

int vwgpu_launch_zone_need(vwgpu_ctx* ctx, const vwgpu_zone_task* zones, int n, const int32_t* l2r, const int* d_zflag, int* d_need,
                           unsigned char* d_cells, size_t ncells) {
  if (n <= 0) return VWGPU_OK;
  if (d_cells == reinterpret_cast<unsigned char*>(d_need + (size_t)8 * n)) {       // one block: one fill
    VWGPU_HIP(ctx, hipMemsetAsync(d_need, 0, (size_t)n * 8 * sizeof(int) + ncells, ctx->stream));
  } else {
    VWGPU_HIP(ctx, hipMemsetAsync(d_need, 0, (size_t)n * 8 * sizeof(int), ctx->stream));
    VWGPU_HIP(ctx, hipMemsetAsync(d_cells, 0, ncells, ctx->stream));
  }
  std::vector<int2> tiles;
  build_tiles(zones, n, tiles);
  if (tiles.empty()) return VWGPU_OK;
  const vwgpu_zone_task* dz; const int2* dt;
  int rc = upload_tables(ctx, zones, n, tiles, &dz, &dt);
  if (rc) return rc;
  vwgpu_prof_scope ps(ctx, "zone_lr_need");
  hipLaunchKernelGGL(zone_need_kernel, dim3((unsigned)tiles.size()), dim3(ZTHREADS), 0, ctx->stream, dz, dt, l2r, d_zflag, d_need, d_cells);
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}

int vwgpu_launch_zone_lr(vwgpu_ctx* ctx, const vwgpu_zone_task* zones, int n, int32_t* l2r, const int32_t* r2l, float thr,
                         float* diff2, ptrdiff_t dstride) {
  if (n <= 0) return VWGPU_OK;
  std::vector<int2> tiles;
  build_tiles(zones, n, tiles);
  if (tiles.empty()) return VWGPU_OK;
  const vwgpu_zone_task* dz; const int2* dt;
  int rc = upload_tables(ctx, zones, n, tiles, &dz, &dt);
  if (rc) return rc;
  vwgpu_prof_scope ps(ctx, "zone_lr_check");
  hipLaunchKernelGGL(zone_lr_kernel, dim3((unsigned)tiles.size()), dim3(ZTHREADS), 0, ctx->stream, dz, dt, l2r, r2l, thr, diff2, dstride);
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}
