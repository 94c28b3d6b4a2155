// bm_exact.hip — block matching and box sums in the reference's OWN summation order, for inputs on which that
// order matters.
//
// fast_box_sum (src/vw/Stereo/Algorithms.h:43-129) is two families of serial float64 recurrences:
//     col_sum(x, 0)   = ((0 + e(x,0)) + e(x,1)) + ... + e(x,ky-1)                          (:62-75)
//     col_sum(x, y+1) = (col_sum(x, y) + e(x, y+ky)) - e(x, y)                             (:100-103, two statements)
//     row_sum(0, y)   = ((0 + col_sum(0,y)) + col_sum(1,y)) + ... + col_sum(kx-1,y)        (:84, std::accumulate)
//     row_sum(x+1, y) = row_sum(x, y) + (col_sum(x+kx, y) - col_sum(x, y))                 (:92)
// and best_of_search_convolution (src/vw/Stereo/Correlation.cc:64-119) runs them once per disparity over the whole
// left raster.  When every partial sum is exactly representable (integer-valued imagery, most float imagery) the
// order is irrelevant and the tile-parallel kernels (bm_sad_u8 / bm_dot_u8 / bm_generic / bm_zones) return the same
// bits.  Otherwise — LoG / mean-subtracted imagery with values near zero, mean-filled nodata, deep pyramid levels
// with SSD / NCC — the roundings depend on the raster position, and exact cost ties are broken by them.  The chains
// are serial only along ONE axis each: every (column, disparity) pair is an independent chain down the rows and
// every (row, disparity) pair an independent chain along the columns, so the reference order IS reproducible in
// parallel — with the column sums of all disparities held in HBM between the two passes:
//
//   pass 1  bmx_col_kernel   thread <-> (zone, column, disparity), serial in y: writes col_sum(x, y, d) into the
//                            zone's volume [row][column][disparity] (disparity fastest: lanes <-> disparities, so the
//                            right-image reads and the volume writes are coalesced, the left pixel is a broadcast);
//   pass 2a bmx_rowsum_kernel  wave <-> (zone, 64 / lanes rows), lane <-> disparity, serial in x: the row recurrence from
//                            coalesced volume reads, written back in place — nothing but the two additions of a chain step;
//   pass 2b bmx_select_kernel  lane <-> pixel, no chain: NCC scaling and the reference's compare chain verbatim over the
//                            disparities in index order (Correlation.cc:91-117; a NaN cost needs no special case).
//   (Until round 3 pass 2 was ONE kernel that also scaled and reduced across the disparity lanes inside every chain step: it is still
//   here, bmx_row_fused_kernel.  The DEFAULT for zones narrower than 1024 pixels — every zone of a pyramid tile — is a third form,
//   bmx_rowsel_kernel + bmx_merge_kernel: row chains and selection in one kernel, transposed through LDS; 2a / 2b serve whole rasters.)
//
// A "zone" is one calc_disparity problem: a SearchParam zone of a pyramid level (CorrelationView.cc:596-700; crops
// with clamped coordinates = the ConstantEdgeExtension crops the reference hands over) or a whole raster.  NCC side
// cars (CostFunctions.h:214-219: 1.0 / fast_box_sum(square(crop))) go through the same two passes with one
// "disparity" per zone crop; vwgpu_fast_box_sum exports that form (Algorithms.h:41-43).
//
// HBM traffic: 8 B written + 8 B read per (pixel, disparity) for the column sums (+ 1.5 B of group records in the tiled form; the same again
// for the row sums in the split form) — the price of the reference's order; the volume of a whole-raster call is processed in row bands
// (column-chain state carried between bands) so that it stays within a scratch budget.  Roofline: HBM bound by the byte count (16 - 32 B per
// evaluation), in fact bound by the wave-cycles of latency-bound wavefronts; it is the correctness path, not the headline.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vwgpu_internal.h"

namespace {

constexpr int XCOST_BOX = 3;        // out = box sum of A                      (fast_box_sum)
constexpr int XCOST_PREC = 4;       // out = 1.0 / box sum of A * A            (NCCCost side car)
constexpr int XMAX_CHUNKS = 8;      // 64-disparity chunks per lane: up to 512 disparities per zone

struct XZone {
  int ax, ay, bx, by;               // crop origins in A / B (clamped reads)
  int zw, zh;                       // output size; crop = (zw + kx - 1) x (zh + ky - 1)
  int sx, sy;
  int out_off, out_stride, addx, addy;
  int d0, dn;                       // this pass serves the disparities [d0, d0 + dn) of the zone's sx * sy (dn <= 512; D = dn below)
  int carry_mode;                   // bit 0: continue the compare chain from the zone's carry records, bit 1: leave it there (more groups follow)
  long long carry;                  // offset (records) of the zone's carry records, one per output pixel
  long long part;                   // bmx_rowsel_kernel: offset (records) of the zone's group records [group][row][x]
  int img;                          // box-sum zones: 0 = the crop is taken from image A, 1 = from image B (one launch serves both NCC side cars)
  int tiled;                        // volume layout: 0 = [row][column][disparity], 1 = [row][group of XTL disparities][column][XTL] (bmx_rowsel_kernel)
  int lanes_log2;                   // lanes per unit = 1 << lanes_log2  (>= min(D, 64), power of two)
  int nchunk;                       // ceil(D / 64)
  long long vol;                    // offset (doubles) of the column-sum volume [rows][cw][dp]
  long long lprec, rprec;           // NCC: offsets (doubles) of the zone's precision images; box modes: lprec = output offset
  long long gate;                   // 0, or the device address of an int: every work item of the zone leaves at once while it is 0 (the zones a
                                    // certified tile-parallel pass did NOT flag, bm_zones.hip)
};
__device__ __forceinline__ bool xgated_off(const XZone& z) { return z.gate != 0 && *reinterpret_cast<const int*>(z.gate) == 0; }

// The two volume layouts.  Zones that go through bmx_rowsel_kernel keep the XTL disparities of a group next to each other for every column, so
// that a row chain of a group streams through consecutive 128-byte lines (with the disparity-fastest layout a chain step of 16 lanes was one
// 128-byte line out of every dp * 8 bytes: more than half of that kernel's time was waiting for them).
constexpr int XTL = 16;
struct XLayout {
  int xs;                           // doubles between two columns of a (row, disparity)
  size_t rs;                        // doubles between two rows
  int tiled, cw;
  __host__ __device__ size_t off(int d) const { return tiled ? (size_t)(d / XTL) * cw * XTL + (d % XTL) : (size_t)d; }
};
__host__ __device__ inline XLayout xlayout(const XZone& z, int cw, int dp) {
  XLayout l;
  l.tiled = z.tiled; l.cw = cw;
  const int ngrp = (z.dn + XTL - 1) / XTL;
  l.xs = z.tiled ? XTL : dp;
  l.rs = z.tiled ? (size_t)ngrp * XTL * cw : (size_t)cw * dp;
  return l;
}

// State of the reference's compare chain (Correlation.cc:91-117) after the disparities of the groups served so far.
struct XCarry { double best, worst; int idx, pad; };

template <int COST>
__device__ __forceinline__ double xelem(float a, float b) {
  if (COST == VWGPU_CROSS_CORRELATION) return (double)(a * b);                       // CostFunctions.h:120-127
  if (COST == VWGPU_SQUARED_DIFFERENCE) { const float d = a - b; return (double)(d * d); }   // :94-101
  if (COST == XCOST_BOX) return (double)a;
  if (COST == XCOST_PREC) return (double)(a * a);                                    // square(): float product
  return (double)fabsf(a - b);                                                        // :72-81
}
template <int COST>
__device__ __forceinline__ bool xbetter(double c, double q) {
  return COST == VWGPU_CROSS_CORRELATION ? (c > q) : (c < q);
}
__device__ __forceinline__ int xclamp(int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); }

// ---- pass 1: column chains -----------------------------------------------------------------------------------------
// items[i] = {zone, first column, disparity chunk, -}.  Rows [y_begin, y_end) of every zone are produced (clipped to
// the zone); y_begin > 0 resumes the chains from `state` (single-zone band mode), and `state` receives them at the end.
// KY > 0: the window height is a compile-time constant and the chain advances in batches of KY rows — the elements that leave the window
// during a batch are the ones that entered it during the previous batch, so every cost element is formed once (KY == 0, any height: the
// entering and the leaving element of every row are both formed from their pixels).
template <int COST, int KY>
__global__ void __launch_bounds__(256)
bmx_col_kernel(const float* __restrict__ A, int aw, int ah, ptrdiff_t as, const float* __restrict__ B, int bw, int bh, ptrdiff_t bs,
               int kx, int ky, const XZone* __restrict__ zones, const int4* __restrict__ items, double* __restrict__ vol,
               int y_begin, int y_end, double* __restrict__ state) {
  constexpr bool BOX = (COST == XCOST_BOX || COST == XCOST_PREC);
  const int4 it = items[blockIdx.x];
  const XZone z = zones[it.x];
  if (xgated_off(z)) return;
  const int cw = z.zw + kx - 1;
  const int lanes = 1 << z.lanes_log2;
  // (tiled volumes: a wavefront is 4 columns x the XTL disparities of a group = 512 consecutive bytes of a volume row)
  const int col = it.y + (z.tiled ? (int)threadIdx.x / XTL : ((int)threadIdx.x >> z.lanes_log2));
  const int d = z.tiled ? it.z * XTL + (int)threadIdx.x % XTL : it.z * 64 + ((int)threadIdx.x & (lanes - 1));
  const int D = z.dn;
  if (col >= cw || d >= D) return;
  const int dp = z.nchunk == 1 ? lanes : z.nchunk * 64;
  const XLayout lay = xlayout(z, cw, dp);
  const int dy = BOX ? 0 : (z.d0 + d) / z.sx, dx = BOX ? 0 : (z.d0 + d) - dy * z.sx;
  if (BOX && z.img) { A = B; aw = bw; ah = bh; as = bs; }
  const float* ac = A + xclamp(z.ax + col, aw);
  const float* bc = BOX ? nullptr : B + xclamp(z.bx + col + dx, bw);
  auto elem = [&](int y) __attribute__((always_inline)) -> double {
    const float a = ac[(ptrdiff_t)xclamp(z.ay + y, ah) * as];
    const float b = BOX ? 0.0f : bc[(ptrdiff_t)xclamp(z.by + y + dy, bh) * bs];
    return xelem<COST>(a, b);
  };
  const int y0 = y_begin < 0 ? 0 : y_begin, y1 = y_end < z.zh ? y_end : z.zh;
  double cs;
  double prev[KY > 0 ? KY : 1];                          // KY > 0: the elements of rows y .. y + KY - 1 (they leave the window next)
  if (KY > 0) {
#pragma unroll
    for (int i = 0; i < KY; ++i) prev[i] = elem(y0 + i);
  }
  if (y0 == 0) {
    cs = 0.0;                                           // std::valarray<AccumT> col_sum(cols): zero-initialised
    if (KY > 0) {
#pragma unroll
      for (int j = 0; j < KY; ++j) cs += prev[j];       // Algorithms.h:62-75
    } else {
      for (int j = 0; j < ky; ++j) cs += elem(j);
    }
  } else {
    cs = state[(size_t)col * dp + d];
  }
  const bool wr = vol != nullptr;                        // nullptr: only advance the chains (rows nobody will read: partial redo of a raster)
  double* v = vol + z.vol + (size_t)col * lay.xs + lay.off(d);
  const size_t rstride = lay.rs;
  int y = y0;
  if (KY > 0) {
    for (; y + KY <= y1 && y + KY < z.zh; y += KY) {    // KY rows' pixels requested together: the chain itself is serial
      double in[KY > 0 ? KY : 1];
#pragma unroll
      for (int i = 0; i < KY; ++i) in[i] = elem(y + KY + i);
#pragma unroll
      for (int i = 0; i < KY; ++i) {
        if (wr) v[(size_t)(y + i - y0) * rstride] = cs;
        cs += in[i];                                    // Algorithms.h:100-103: two statements, this order
        cs -= prev[i];
      }
#pragma unroll
      for (int i = 0; i < KY; ++i) prev[i] = in[i];
    }
  }
  for (; KY == 0 && y + 8 <= y1 && y + 8 < z.zh; y += 8) {         // eight rows' pixels requested together: the chain itself is serial
    double in[8], out[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { in[i] = elem(y + ky + i); out[i] = elem(y + i); }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (wr) v[(size_t)(y + i - y0) * rstride] = cs;
      cs += in[i];                                      // Algorithms.h:100-103: two statements, this order
      cs -= out[i];
    }
  }
  for (; y < y1; ++y) {
    if (wr) v[(size_t)(y - y0) * rstride] = cs;
    if (y + 1 < z.zh) {
      cs += elem(y + ky);
      cs -= elem(y);
    }
  }
  if (state) state[(size_t)col * dp + d] = cs;
}

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)dpp_i32<CTRL>((int)(unsigned)b), hi = (unsigned)dpp_i32<CTRL>((int)(unsigned)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// ---- pass 2 of the box sums (fast_box_sum, NCC precision images): the row chain of ONE image, the sum itself is the result ----------
// items[i] = {zone, first row}; one wave per item, a lane per row.  (The matchers' pass 2: bmx_rowsum_kernel + bmx_select_kernel.)
template <int COST>
__global__ void __launch_bounds__(256)
bmx_box_row_kernel(int kx, const XZone* __restrict__ zones, const int2* __restrict__ items, const double* __restrict__ vol,
                   int y_begin, int y_end, double* __restrict__ outd) {
  static_assert(COST == XCOST_BOX || COST == XCOST_PREC, "box sums only");
  // The chain of a row is two additions per pixel; the reciprocal of the precision images (a division, ~25 instructions) and the store do
  // not belong into it.  16 steps' sums go to LDS, then the 64 x 16 block is finished by all lanes along the rows: independent divisions,
  // 128-byte stores (a lane per row wrote 8 bytes out of every row).
  __shared__ double R[4][64 * 17];
  const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  const int2 it = items[blockIdx.x * 4 + wave];
  if (it.x < 0) return;
  const XZone z = zones[it.x];                          // lanes_log2 == 0, one "disparity": a lane per row
  if (xgated_off(z)) return;
  const int cw = z.zw + kx - 1;
  const int ylim = y_end < z.zh ? y_end : z.zh;
  const int y = it.y + lane;
  const bool act = y < ylim;                            // idle lanes walk the item's first row
  const double* base = vol + z.vol + (size_t)((act ? y : it.y) - y_begin) * cw;
  double r = 0.0;
  for (int i = 0; i < kx; ++i) r += base[i];            // Algorithms.h:84: accumulate from 0
  double* Rw = R[wave];
  double* o = outd + z.lprec;
  for (int x0 = 0; x0 < z.zw; x0 += 16) {
    // A lane streams its own row: 16 steps' operands — one 128-byte line of each stream — are requested together
    double l[16], t[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { l[i] = base[min(x0 + i + kx, cw - 1)]; t[i] = base[min(x0 + i, cw - 1)]; }
#pragma unroll
    for (int i = 0; i < 16; ++i) { Rw[lane * 17 + i] = r; r += l[i] - t[i]; }       // Algorithms.h:92 (sums past the row's end are not used)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wavefront's own LDS writes, then reads by other lanes
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int i = lane + 64 * k, row = i >> 4, col = i & 15;
      const int yy = it.y + row, xx = x0 + col;
      const double v = Rw[row * 17 + col];
      if (yy < ylim && xx < z.zw) o[(size_t)yy * z.zw + xx] = (COST == XCOST_PREC) ? 1.0 / v : v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next block overwrites
  }
}

// ---- pass 2 of the matchers, fused form (round 2): row chains, NCC scaling and the winner across the disparity lanes in ONE kernel ----
// Kept for zone lists whose longest chain is short (the ~1000 zones of a pyramid level): there the three kernels of the split form move
// the level's volumes three times and cost two more launches per level and direction — the tile loop is launch bound (LoG + NCC tile loop
// 327 -> 291 Mpix/s with the split form everywhere).  A chain step here is ~300 dependent instructions (1.1 us): long chains — whole
// rasters, zones as wide as a whole tile — take bmx_rowsum_kernel + bmx_select_kernel below.
// items[i] = {zone, first row}; one wave per item, 64 / lanes rows per wave.  NCH = 64-disparity chunks a lane can hold (instantiated
// for 1, 3 and XMAX_CHUNKS; the launcher picks by the largest zone).  CARRY: a disparity group of a zone with more than 512 disparities.
template <int COST, int NCH, bool CARRY = false>
__global__ void __launch_bounds__(256)
bmx_row_fused_kernel(int kx, const XZone* __restrict__ zones, const int2* __restrict__ items, const double* __restrict__ vol,
               int y_begin, int y_end, const double* __restrict__ prec, int32_t* __restrict__ out, double* __restrict__ outd,
               XCarry* __restrict__ carry = nullptr, const int* __restrict__ zone_flag = nullptr) {
  constexpr bool BOX = (COST == XCOST_BOX || COST == XCOST_PREC);
  constexpr bool NCC = (COST == VWGPU_CROSS_CORRELATION);
  __shared__ double park[4][NCH][64];           // costs of a NaN pixel, for the verbatim replay
  const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  const int2 it = items[blockIdx.x * 4 + wave];
  if (it.x < 0) return;
  if (zone_flag && !zone_flag[it.x]) return;    // the pass after bmx_merge_kernel: only the zones it flagged
  const XZone z = zones[it.x];
  if (xgated_off(z)) return;
  const int cw = z.zw + kx - 1;
  const int lanes = 1 << z.lanes_log2;
  const int g = lane >> z.lanes_log2, dl = lane & (lanes - 1);
  const int y = it.y + g;
  const int ylim = y_end < z.zh ? y_end : z.zh;
  const bool row_ok = y < ylim;
  const int D = z.dn;
  const int dp = z.nchunk == 1 ? lanes : z.nchunk * 64;
  const int nch = z.nchunk;
  const XLayout lay = xlayout(z, cw, dp);
  const double* base = vol + z.vol + (size_t)(row_ok ? y - y_begin : 0) * lay.rs;

  double r[NCH];
  int dk[NCH];
  const double* rp[NCH];
  bool act[NCH];
  const int rpw = z.zw + z.sx - 1;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    dk[k] = k * 64 + dl;
    act[k] = row_ok && k < nch && dk[k] < D;
    r[k] = 0.0;
    rp[k] = nullptr;
    if (act[k]) {
      for (int i = 0; i < kx; ++i) r[k] += base[(size_t)i * lay.xs + lay.off(dk[k])];       // Algorithms.h:84: accumulate from 0
      if (NCC) {
        const int dy = (z.d0 + dk[k]) / z.sx, dx = (z.d0 + dk[k]) - dy * z.sx;
        rp[k] = prec + z.rprec + (size_t)(y + dy) * rpw + dx;
      }
    }
  }
  const double* lp = NCC ? prec + z.lprec + (size_t)(row_ok ? y : 0) * z.zw : nullptr;
  const double SENT_BEST = NCC ? -INFINITY : INFINITY, SENT_WORST = NCC ? INFINITY : -INFINITY;

  static_assert(!BOX, "box sums: bmx_box_row_kernel");
  // The operands of a step — the two column sums that advance each chain and, for NCC, the precisions — are requested one step
  // ahead, unconditionally (idle lanes / chunks read a valid dummy address, the last step re-reads clamped indices): loaded and
  // consumed in the same step they cost a memory round trip per step; a load under a lane condition makes the compiler drain
  // every outstanding request (s_waitcnt vmcnt(0)) before the next use.
  size_t off[NCH];
  const double* rps[NCH];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    off[k] = act[k] ? lay.off(dk[k]) : 0;
    rps[k] = (NCC && act[k]) ? rp[k] : prec;
  }
  const double* lps = NCC ? (row_ok ? lp : prec) : nullptr;
  // Round 3: the operands of a step are requested PF steps ahead (a ring of PF register sets, the step loop unrolled PF times).  One
  // step ahead left a chain step at the latency of a memory round trip — 1.1 us per step at any chunk count, 0.46 ms for a 256-step
  // chain (tools/time_exact_zone.py) — and the longest chains of a level ARE the time of the launch.  (4 / 3 / 2 sets for 1 / 3 / 8
  // chunks per lane: what the register file takes.)
  constexpr int PF = NCH <= 1 ? 4 : (NCH <= 3 ? 3 : 2);
  double pl[PF][NCH], pt[PF][NCH], crp[NCH], nrp[PF][NCH];
  double clp = 0.0, nlp[PF];
  auto request = [&](int u, int t) __attribute__((always_inline)) {     // the operands that END step t (clamped: requests past the row re-read its end)
    const int tc = min(t, z.zw - 1);
    const size_t li = (size_t)min(tc + kx, cw - 1) * lay.xs, ti = (size_t)tc * lay.xs;
    const int xn = min(tc + 1, z.zw - 1);
#pragma unroll
    for (int k = 0; k < NCH; ++k)
      if (k < nch) {                                    // wave-uniform
        pl[u][k] = base[li + off[k]]; pt[u][k] = base[ti + off[k]];
        if (NCC) nrp[u][k] = rps[k][xn];
      }
    if (NCC) nlp[u] = lps[xn];
  };
#pragma unroll
  for (int k = 0; k < NCH; ++k) { crp[k] = 0.0; if (NCC && k < nch) crp[k] = rps[k][0]; }
#pragma unroll
  for (int u = 0; u < PF; ++u) {
#pragma unroll
    for (int k = 0; k < NCH; ++k) pl[u][k] = pt[u][k] = nrp[u][k] = 0.0;
    nlp[u] = 0.0;
    request(u, u);
  }
  if (NCC) clp = lps[0];

  int res_d = 0, res_v = 0;                             // buffered result of the step x with (x & (lanes-1)) == dl
  for (int x0 = 0; x0 < z.zw; x0 += PF)
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const int x = x0 + u;
    if (x >= z.zw) break;
    // this lane's candidates, in disparity order
    double c[NCH];
    double best = SENT_BEST, worst = SENT_WORST;
    int bd = INT_MAX;
    bool nan = false;
    const double lpx = clp;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      c[k] = 0.0;
      if (k < nch && act[k]) {
        double v = r[k];
        if (NCC) v *= sqrt(lpx * crp[k]);               // CostFunctions.h:227-231
        c[k] = v;
        nan |= (v != v);
        if (xbetter<COST>(v, best) || (v == best && dk[k] < bd)) { best = v; bd = dk[k]; }
        if (xbetter<COST>(worst, v)) worst = v;
      }
    }
    // winner across the disparity lanes of the row's group
    int nanw = nan ? 1 : 0;
    // Winner across the group's lanes in two butterflies: first the extreme VALUES (v_max_f64 / v_min_f64: one instruction per
    // stage), then the smallest disparity among the lanes that hold the best value — "strict compare, first wins" without carrying
    // (value, index) pairs and a lexicographic select through every stage (that form was ~150 of the ~320 VALU instructions of a
    // pixel step).  -0.0 and +0.0 compare equal, as in the reference's chain; NaN costs take the replay below.
    // Partners 32 and 16 lanes away through ds_bpermute, the four nearest stages as DPP moves (quad permutes, row_half_mirror,
    // row_mirror: inside an aligned group of 2 / 4 / 8 / 16 lanes they pair the same halves).
    auto fold_v = [&](double ob, double ow, int on) __attribute__((always_inline)) {
      nanw |= on;
      best = NCC ? fmax(best, ob) : fmin(best, ob);
      worst = NCC ? fmin(worst, ow) : fmax(worst, ow);
    };
    const double lbest = best;                            // this lane's best value; bd = its (smallest) disparity
    if (lanes > 32) fold_v(__shfl_xor(best, 32), __shfl_xor(worst, 32), __shfl_xor(nanw, 32));
    if (lanes > 16) fold_v(__shfl_xor(best, 16), __shfl_xor(worst, 16), __shfl_xor(nanw, 16));
    if (lanes > 8) fold_v(dpp_f64<0x140>(best), dpp_f64<0x140>(worst), dpp_i32<0x140>(nanw));     // row_mirror
    if (lanes > 4) fold_v(dpp_f64<0x141>(best), dpp_f64<0x141>(worst), dpp_i32<0x141>(nanw));     // row_half_mirror
    if (lanes > 2) fold_v(dpp_f64<0x4E>(best), dpp_f64<0x4E>(worst), dpp_i32<0x4E>(nanw));         // quad_perm [2,3,0,1]
    if (lanes > 1) fold_v(dpp_f64<0xB1>(best), dpp_f64<0xB1>(worst), dpp_i32<0xB1>(nanw));         // quad_perm [1,0,3,2]
    bd = (lbest == best) ? bd : INT_MAX;                   // lanes without a candidate hold the sentinel: never equal, or bd = INT_MAX
    if (lanes > 32) bd = min(bd, __shfl_xor(bd, 32));
    if (lanes > 16) bd = min(bd, __shfl_xor(bd, 16));
    if (lanes > 8) bd = min(bd, dpp_i32<0x140>(bd));
    if (lanes > 4) bd = min(bd, dpp_i32<0x141>(bd));
    if (lanes > 2) bd = min(bd, dpp_i32<0x4E>(bd));
    if (lanes > 1) bd = min(bd, dpp_i32<0xB1>(bd));
    // advance the chains (Algorithms.h:92) with the operands requested at the top of the step
    if (x + 1 < z.zw) {
#pragma unroll
      for (int k = 0; k < NCH; ++k)
        if (k < nch && act[k]) r[k] += pl[u][k] - pt[u][k];
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) crp[k] = nrp[u][k];
    clp = nlp[u];
    request(u, x + PF);                                 // this register set again PF steps from now
    // Disparity groups: the chain continues from the state the earlier groups left.  Without NaNs it is a (value, index) minimum
    // and a maximum, so the states merge (an earlier group wins ties: its indices are smaller); with a NaN in this group's costs
    // or in the carried state the chain is order dependent and is replayed verbatim from the carried state.
    XCarry cin{0.0, 0.0, 0, 0};
    bool cont = false;                                  // this pixel's chain has a carried state
    if (CARRY) {
      cont = (z.carry_mode & 1) != 0;
      if (cont && row_ok) {
        cin = carry[z.carry + (size_t)y * z.zw + x];
        if (cin.best != cin.best || cin.worst != cin.worst) nanw = 1;
      }
    }
    if (__any(nanw)) {                                  // wave-uniform: some pixel of this step has a NaN cost
#pragma unroll
      for (int k = 0; k < NCH; ++k)
        if (k < nch) park[wave][k][lane] = c[k];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (nanw && dl == 0 && row_ok) {                  // Correlation.cc:91-117, verbatim
        double b2 = cin.best, w2 = cin.worst;
        int i2 = cin.idx;
        for (int d = 0; d < D; ++d) {
          const double v = nch == 1 ? park[wave][0][g * lanes + d] : park[wave][d >> 6][d & 63];
          if (d == 0 && !cont) { b2 = w2 = v; i2 = 0; }
          else if (xbetter<COST>(v, b2)) { b2 = v; i2 = (CARRY ? z.d0 : 0) + d; }
          else if (!xbetter<COST>(v, w2)) { w2 = v; }
        }
        best = b2; worst = w2; bd = i2 - (CARRY ? z.d0 : 0);
      }
      __builtin_amdgcn_wave_barrier();
      if (lanes > 1) {                                  // hand the replayed result to the lane that buffers this step
        const int src = g * lanes;
        const double b3 = __shfl(best, src), w3 = __shfl(worst, src);
        const int d3 = __shfl(bd, src);
        if (nanw) { best = b3; worst = w3; bd = d3; }
      }
    }
    if (CARRY) {
      int gd = z.d0 + bd;                               // index in the zone's whole search volume
      if (cont && !nanw) {
        if (!xbetter<COST>(best, cin.best)) { best = cin.best; gd = cin.idx; }
        if (xbetter<COST>(worst, cin.worst)) worst = cin.worst;
      }
      if (row_ok && dl == 0) {
        if (z.carry_mode & 2) {
          carry[z.carry + (size_t)y * z.zw + x] = XCarry{best, worst, gd, 0};
        } else {
          const int dy = gd / z.sx, dx = gd - dy * z.sx;
          int32_t* o = out + ((size_t)z.out_off + (size_t)y * z.out_stride + x) * 3;
          o[0] = dx + z.addx; o[1] = dy + z.addy; o[2] = (best == worst) ? 0 : 0x7fffffff;
        }
      }
      continue;
    }
    const int slot = x & (lanes - 1);
    if (dl == slot) { res_d = bd; res_v = (best == worst) ? 0 : 0x7fffffff; }   // Correlation.cc:121-133
    if (slot == lanes - 1 || x == z.zw - 1) {
      const int xb = x - slot;
      if (row_ok && dl <= slot) {
        const int dy = res_d / z.sx, dx = res_d - dy * z.sx;
        int32_t* o = out + ((size_t)z.out_off + (size_t)y * z.out_stride + xb + dl) * 3;
        o[0] = dx + z.addx; o[1] = dy + z.addy; o[2] = res_v;
      }
    }
  }
}


// ---- pass 2 of the matchers (round 3): the row chains alone, then a parallel selection ------------------------------------------
// Until round 3 one kernel ran the row chain AND, inside every chain step, the NCC scaling and the winner across the disparity
// lanes (two f64 butterflies, a NaN replay): ~300 dependent instructions per step, 1.1 us, so a 256-pixel zone row took 0.46 ms and
// the longest chains of a level were the time of its launch (tools/time_exact_zone.py).  The recurrence itself is two additions:
//   bmx_rowsum_kernel   lane <-> disparity, serial in x: row_sum(x+1) = row_sum(x) + (col_sum(x+kx) - col_sum(x))  (Algorithms.h:84,92),
//                       operands requested PF steps ahead, the row sum written IN PLACE over col_sum(x) (read for the last time in
//                       that very step);
//   bmx_select_kernel   lane <-> pixel: the costs of all disparities in index order through the reference's compare chain VERBATIM
//                       (Correlation.cc:91-117; NaN behaviour for free), NCC scaling (CostFunctions.h:227-231) on the way; zones of
//                       more than 512 disparities continue the chain from / leave it in the XCarry records of their pixels.
// items[i] = {zone, first row, 64-disparity chunk, -}: one wave per item (64 / lanes rows of a zone that searches fewer than 64
// disparities, one chunk of 64 otherwise — every chain is a lane of its own, whatever the zone's search volume).
__global__ void __launch_bounds__(256)
bmx_rowsum_kernel(int kx, const XZone* __restrict__ zones, const int4* __restrict__ items, double* __restrict__ vol, int y_begin, int y_end) {
  const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  const int4 it = items[blockIdx.x * 4 + wave];
  if (it.x < 0) return;
  const XZone z = zones[it.x];
  if (xgated_off(z)) return;
  const int cw = z.zw + kx - 1;
  const int lanes = 1 << z.lanes_log2;
  const int g = lane >> z.lanes_log2, dl = lane & (lanes - 1);
  const int y = it.y + g;
  const int ylim = y_end < z.zh ? y_end : z.zh;
  const int dp = z.nchunk == 1 ? lanes : z.nchunk * 64;
  const int dk = it.z * 64 + dl;
  if (y >= ylim || dk >= z.dn) return;
  double* base = vol + z.vol + (size_t)(y - y_begin) * cw * dp + dk;
  double r = 0.0;
  for (int i = 0; i < kx; ++i) r += base[(size_t)i * dp];          // Algorithms.h:84: accumulate from 0
  // requests in flight per chain: the volumes of a level are far larger than the L2, a request is ~2 us away
  constexpr int PF = 12;
  double pl[PF], pt[PF];
  auto request = [&](int u, int t) __attribute__((always_inline)) {     // the operands that END step t (requests past the row re-read its end)
    const int tc = min(t, z.zw - 1);
    pl[u] = base[(size_t)min(tc + kx, cw - 1) * dp];
    pt[u] = base[(size_t)tc * dp];
  };
#pragma unroll
  for (int u = 0; u < PF; ++u) request(u, u);
  for (int x0 = 0; x0 < z.zw; x0 += PF)
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int x = x0 + u;
      if (x < z.zw) {                                  // (wave-uniform; no `break`: the loop must unroll, pl / pt are registers)
        base[(size_t)x * dp] = r;                      // in place: col_sum(x) was requested PF steps ago and is not needed again
        r += pl[u] - pt[u];                            // Algorithms.h:92 (the value after the last pixel is not used)
        // requests of step x + PF read col_sum(x + PF) and col_sum(x + PF + kx): columns this wave has not overwritten yet
        request(u, x + PF);
      }
    }
}

// items[i] = {zone, y0 | x-chunk << 20}: a wave serves 64 pixels — (64 >> xlog) rows x (1 << xlog) columns, xlog = log2 of the zone's
// width rounded up to a power of two (at most 64 columns per chunk).
template <int COST, bool CARRY>
__global__ void __launch_bounds__(256)
bmx_select_kernel(int kx, const XZone* __restrict__ zones, const int2* __restrict__ items, const double* __restrict__ vol, int y_begin, int y_end,
                  const double* __restrict__ prec, int32_t* __restrict__ out, XCarry* __restrict__ carry) {
  constexpr bool NCC = (COST == VWGPU_CROSS_CORRELATION);
  const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  const int2 it = items[blockIdx.x * 4 + wave];
  if (it.x < 0) return;
  const XZone z = zones[it.x];
  if (xgated_off(z)) return;
  const int cw = z.zw + kx - 1;
  const int lanes = 1 << z.lanes_log2;
  const int dp = z.nchunk == 1 ? lanes : z.nchunk * 64;
  int xlog = 0;
  while ((1 << xlog) < z.zw && xlog < 6) ++xlog;
  const int x = (it.y >> 20) * 64 + (lane & ((1 << xlog) - 1));
  const int y = (it.y & 0xfffff) + (lane >> xlog);
  const int ylim = y_end < z.zh ? y_end : z.zh;
  const bool ok = x < z.zw && y < ylim;
  if (!ok) return;
  const double* v = vol + z.vol + ((size_t)(y - y_begin) * cw + x) * dp;
  const int rpw = z.zw + z.sx - 1;
  const double lp = NCC ? prec[z.lprec + (size_t)y * z.zw + x] : 0.0;
  const double* rpb = NCC ? prec + z.rprec + (size_t)y * rpw + x : nullptr;
  double best = 0.0, worst = 0.0;
  int idx = 0;
  bool first = true;                                    // no value seen yet: the chain starts with `best = worst = v` (Correlation.cc:93-96)
  if (CARRY && (z.carry_mode & 1)) {
    const XCarry c = carry[z.carry + (size_t)y * z.zw + x];
    best = c.best; worst = c.worst; idx = c.idx; first = false;
  }
  int dy = z.d0 / z.sx, dx = z.d0 - dy * z.sx;          // the group's first disparity
  const int D = z.dn;
  auto take = [&](double c, int d) __attribute__((always_inline)) {
    if (first) { best = worst = c; idx = d; first = false; }
    else if (xbetter<COST>(c, best)) { best = c; idx = d; }
    else if (!xbetter<COST>(c, worst)) { worst = c; }
  };
  int d = 0;
  // A lane streams its own pixel's vector: consecutive lanes are dp * 8 bytes apart, so one request touches 64 cache lines whatever
  // its width — eight disparities (two 32-byte requests) are asked for together and a line is used up in two visits instead of 16.
  struct __attribute__((packed, aligned(8))) D4 { double v[4]; };
  for (; d + 8 <= D; d += 8) {
    const D4 a = *reinterpret_cast<const D4*>(v + d), b = *reinterpret_cast<const D4*>(v + d + 4);
    double c[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      c[i] = i < 4 ? a.v[i] : b.v[i - 4];
      if (NCC) {
        q[i] = rpb[(size_t)dy * rpw + dx];
        if (++dx == z.sx) { dx = 0; ++dy; }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (NCC) c[i] *= sqrt(lp * q[i]);                 // CostFunctions.h:227-231
      take(c[i], z.d0 + d + i);
    }
  }
  for (; d < D; ++d) {
    double c = v[d];
    if (NCC) {
      c *= sqrt(lp * rpb[(size_t)dy * rpw + dx]);
      if (++dx == z.sx) { dx = 0; ++dy; }
    }
    take(c, z.d0 + d);
  }
  if (CARRY && (z.carry_mode & 2)) {
    carry[z.carry + (size_t)y * z.zw + x] = XCarry{best, worst, idx, 0};
  } else {
    const int qy = idx / z.sx, qx = idx - qy * z.sx;
    int32_t* o = out + ((size_t)z.out_off + (size_t)y * z.out_stride + x) * 3;
    o[0] = qx + z.addx; o[1] = qy + z.addy; o[2] = (best == worst) ? 0 : 0x7fffffff;       // Correlation.cc:121-133
  }
}

// ---- pass 2 of the matchers, tiled form (round 3): row chains and selection in ONE kernel, transposed through LDS ----------------
// The fused kernel above reads the volume once but selects ACROSS the disparity lanes inside every chain step (~300 dependent
// instructions per step: it is instruction bound, 130 G evaluations/s); the split form is cheap per evaluation but moves the volume three
// times.  Here one wavefront owns (zone, 4 rows, a group of 16 disparities) and walks the rows in chunks of 16 pixels:
//   chains   lane <-> (row, disparity): 16 steps of row_sum(x+1) = row_sum(x) + (col_sum(x+kx) - col_sum(x)), every operand of the
//            chunk requested up front (Algorithms.h:84,92); every row sum goes to LDS, T[chain][step]; the chain's state is a register
//   select   lane <-> (row, pixel): the 16 costs of its pixel from LDS (conflict free: consecutive lanes, consecutive words), NCC
//            scaling, the reference's compare chain VERBATIM over the group (Correlation.cc:91-117).
// A zone of up to 16 disparities is finished by that wavefront.  Wider searches: the groups of a pixel run in parallel (a wavefront that
// walked all of them in turn was the critical path of a level: 16 chunks x 7 groups x 9 us for a 512-pixel zone with 216 disparities)
// and leave (best, worst, index, "saw a NaN") per group; bmx_merge_kernel folds them in index order.  Without NaN costs the chain is a
// (value, first index) minimum and a maximum, which fold exactly; a pixel with a NaN cost (NCC over an all-zero window: 0 * inf) is
// order dependent, so its zone is flagged and the fused kernel above recomputes the flagged zones from the same volume.
// 8 B of HBM traffic per evaluation (the column sums, once) + 1.5 B for the group records, ~40 instruction slots instead of ~300.
constexpr int TL_X = XTL, TL_PITCH = TL_X + 1;       // pixels per chunk = disparities per group; 64 / TL_X rows per wavefront
constexpr int TL_R = 64 / TL_X, TL_LOG = 4;
static_assert((1 << TL_LOG) == TL_X, "TL_X is a power of two");
// KX > 0: the window width is a compile-time constant and the column sums of a chunk live in a register window of TL_X + KX values — the
// last KX of a chunk are the first KX of the next one, so every column sum is requested exactly once (KX == 0, any width: lead and trail of
// every step are requested separately; the halo of a chunk is then read twice, 27 lines per 16 steps at 11 x 11).
template <int COST, bool CARRY, int KX>
__global__ void __launch_bounds__(64)
bmx_rowsel_kernel(int kx, const XZone* __restrict__ zones, const int4* __restrict__ items, const double* __restrict__ vol, int y_begin, int y_end,
                  const double* __restrict__ prec, int32_t* __restrict__ out, XCarry* __restrict__ carry, XCarry* __restrict__ part) {
  constexpr bool NCC = (COST == VWGPU_CROSS_CORRELATION);
  __shared__ double T[64 * TL_PITCH];                   // [chain][step]
  const int lane = (int)threadIdx.x;
  const int4 it = items[blockIdx.x];
  const XZone z = zones[it.x];
  if (xgated_off(z)) return;
  const int cw = z.zw + kx - 1;
  const int lanes = 1 << z.lanes_log2;
  const int dp = z.nchunk == 1 ? lanes : z.nchunk * 64;
  const int D = z.dn;
  const int ylim = y_end < z.zh ? y_end : z.zh;
  const int g = lane >> TL_LOG, dl = lane & (TL_X - 1);              // row of the pair; disparity of the group (chains) / pixel of the chunk (select)
  const int y = it.y + g;
  const bool row_ok = y < ylim;
  const int dg = it.z;
  const bool single = D <= TL_X;                          // the whole search in this group: the result is final
  const int d = dg * TL_X + dl;
  const bool act = row_ok && d < D;
  // idle lanes walk a valid chain whose sums nobody reads
  const XLayout lay = xlayout(z, cw, dp);
  const double* base = vol + z.vol + (size_t)((row_ok ? y : it.y) - y_begin) * lay.rs + lay.off(act ? d : dg * TL_X);
  const int xs = XTL;                                   // (z.tiled: the only zones this kernel serves)
  const int jn = min(TL_X, D - dg * TL_X);
  const int rpw = z.zw + z.sx - 1;
  const int dfirst = z.d0 + dg * TL_X;                    // the group's first disparity in the zone's search volume
  const int dy0 = dfirst / z.sx, dx0 = dfirst - dy0 * z.sx;
  const int prows = (y_end < z.zh ? y_end : z.zh) - y_begin;     // rows of this band: the group records are [group][row of the band][x]
  constexpr int NV = KX > 0 ? TL_X + KX : 1;
  double v[NV];                                         // KX > 0: v[i] = col_sum(x0 + i) of this chain
  double r = 0.0;
  if (KX > 0) {
#pragma unroll
    for (int i = 0; i < KX; ++i) v[TL_X + i] = base[(size_t)i * xs];
#pragma unroll
    for (int i = 0; i < KX; ++i) r += v[TL_X + i];     // Algorithms.h:84: accumulate from 0
  } else {
    for (int i = 0; i < kx; ++i) r += base[(size_t)i * xs];
  }
  // The column sums of chunk k + 1 are requested while chunk k is selected (a level's tiled pass is ~9000 wavefronts, 10 per CU on average:
  // there are not enough of them to hide a memory round trip per chunk behind each other).
  double vn[KX > 0 ? TL_X : 1];
  auto request_chunk = [&](int xc) __attribute__((always_inline)) {       // col_sum(xc + KX + u), u < TL_X, into vn
    if (xc + KX + TL_X <= cw) {                         // wave-uniform: constant offsets, no address arithmetic per request
      const double* pb = base + (size_t)(xc + KX) * XTL;
#pragma unroll
      for (int u = 0; u < TL_X; ++u) vn[u] = pb[u * XTL];
    } else {
#pragma unroll
      for (int u = 0; u < TL_X; ++u) vn[u] = base[(size_t)min(xc + KX + u, cw - 1) * XTL];     // (past the row's end: sums nobody uses)
    }
  };
  if (KX > 0) request_chunk(0);
  for (int x0 = 0; x0 < z.zw; x0 += TL_X) {
    // ---- chains: every operand of the chunk's steps is in registers before the first step ----
    // (xs == XTL here: tiled volumes only)
    double pl[TL_X], pt[TL_X];
    if (KX > 0) {
#pragma unroll
      for (int i = 0; i < KX; ++i) v[i] = v[TL_X + i];
#pragma unroll
      for (int u = 0; u < TL_X; ++u) v[KX + u] = vn[u];
      request_chunk(x0 + TL_X);
#pragma unroll
      for (int u = 0; u < TL_X; ++u) { pl[u] = v[KX + u]; pt[u] = v[u]; }
    } else if (x0 + kx + TL_X <= cw) {
      const double* pb = base + (size_t)x0 * XTL;
#pragma unroll
      for (int u = 0; u < TL_X; ++u) { pl[u] = pb[(size_t)(u + kx) * XTL]; pt[u] = pb[u * XTL]; }
    } else {
#pragma unroll
      for (int u = 0; u < TL_X; ++u) {                  // the operands that END step u (clamped at the row's end: those sums are not used)
        const int tc = min(x0 + u, z.zw - 1);
        pl[u] = base[(size_t)min(tc + kx, cw - 1) * XTL];
        pt[u] = base[(size_t)tc * XTL];
      }
    }
    const int x = x0 + dl;
    const bool pok = row_ok && x < z.zw;
    const double lp = (NCC && pok) ? prec[z.lprec + (size_t)y * z.zw + x] : 0.0;
    const double* rpb = NCC ? prec + z.rprec + (pok ? (size_t)y * rpw + x : 0) : nullptr;
    double* trow = T + lane * TL_PITCH;
#pragma unroll
    for (int u = 0; u < TL_X; ++u) {
      trow[u] = r;                                      // (steps past the row's end write sums nobody reads)
      r += pl[u] - pt[u];                               // Algorithms.h:92
    }
    double q[TL_X];                                     // (requested once the chain operands are dead: together they would be 280 registers)
    const bool full = jn == TL_X;                       // wave-uniform
    if (NCC) {
      if (full && dx0 + TL_X <= z.sx) {                 // the group stays on one search row
        const double* qb = rpb + (size_t)dy0 * rpw + dx0;
#pragma unroll
        for (int i = 0; i < TL_X; ++i) q[i] = qb[i];
      } else {
        int dy = dy0, dx = dx0;
#pragma unroll
        for (int i = 0; i < TL_X; ++i) {
          q[i] = rpb[(size_t)dy * rpw + dx];
          if (i + 1 < jn) { if (++dx == z.sx) { dx = 0; ++dy; } }     // never past the group's last disparity (the read stays inside the image)
        }
      }
    }
    __syncthreads();
    // ---- select: lane <-> pixel (row g, x0 + dl); the group's costs in index order ----
    double best = 0.0, worst = 0.0;
    int idx = 0;
    bool first = true;                                  // no value seen yet: the chain starts with `best = worst = v` (Correlation.cc:93-96)
    bool nan = false;
    if (CARRY && single && (z.carry_mode & 1) && pok) {
      const XCarry c = carry[z.carry + (size_t)y * z.zw + x];
      best = c.best; worst = c.worst; idx = c.idx; first = false;
    }
    const double* tcol = T + (g * TL_X) * TL_PITCH + dl;
    double c[TL_X];
#pragma unroll
    for (int i = 0; i < TL_X; ++i) c[i] = tcol[i * TL_PITCH];
    auto take = [&](int i) __attribute__((always_inline)) {
      double v = c[i];
      if (NCC) v *= sqrt(lp * q[i]);                    // CostFunctions.h:227-231
      nan |= (v != v);
      // Correlation.cc:91-117 as selects (a branch per comparison cost more than the comparisons)
      const bool b = first || xbetter<COST>(v, best);
      const bool w = first || (!b && !xbetter<COST>(v, worst));
      best = b ? v : best;
      idx = b ? dfirst + i : idx;
      worst = w ? v : worst;
      first = false;
    };
    if (full) {
#pragma unroll
      for (int i = 0; i < TL_X; ++i) take(i);
    } else {
#pragma unroll
      for (int i = 0; i < TL_X; ++i)
        if (i < jn) take(i);                            // wave-uniform
    }
    __syncthreads();                                    // T is rewritten by the next chunk's chains
    if (pok) {
      if (!single) {
        part[z.part + ((size_t)dg * prows + (y - y_begin)) * z.zw + x] = XCarry{best, worst, idx, nan ? 1 : 0};
      } else if (CARRY && (z.carry_mode & 2)) {
        carry[z.carry + (size_t)y * z.zw + x] = XCarry{best, worst, idx, 0};
      } else {
        const int qy = idx / z.sx, qx = idx - qy * z.sx;
        int32_t* o = out + ((size_t)z.out_off + (size_t)y * z.out_stride + x) * 3;
        o[0] = qx + z.addx; o[1] = qy + z.addy; o[2] = (best == worst) ? 0 : 0x7fffffff;       // Correlation.cc:121-133
      }
    }
  }
}

// Folds the group records of bmx_rowsel_kernel in index order.  items[i] = {zone, y0 | x-chunk << 20} as bmx_select_kernel's (zones of
// more than 16 disparities only).  A NaN in any record (or in the carried state) flags the zone: the fused kernel recomputes it.
template <int COST, bool CARRY>
__global__ void __launch_bounds__(256)
bmx_merge_kernel(const XZone* __restrict__ zones, const int2* __restrict__ items, int y_begin, int y_end, const XCarry* __restrict__ part,
                 int32_t* __restrict__ out, XCarry* __restrict__ carry, int* __restrict__ zone_flag) {
  const int wave = (int)threadIdx.x >> 6, lane = (int)threadIdx.x & 63;
  const int2 it = items[blockIdx.x * 4 + wave];
  if (it.x < 0) return;
  const XZone z = zones[it.x];
  if (xgated_off(z)) return;
  int xlog = 0;
  while ((1 << xlog) < z.zw && xlog < 6) ++xlog;
  const int x = (it.y >> 20) * 64 + (lane & ((1 << xlog) - 1));
  const int y = (it.y & 0xfffff) + (lane >> xlog);
  const int ylim = y_end < z.zh ? y_end : z.zh;
  if (x >= z.zw || y >= ylim) return;
  const int prows = ylim - y_begin;
  const int ngrp = (z.dn + TL_X - 1) / TL_X;
  const XCarry* p = part + z.part + (size_t)(y - y_begin) * z.zw + x;
  const size_t gstride = (size_t)prows * z.zw;
  XCarry s = p[0];
  int nan = s.pad;
  if (CARRY && (z.carry_mode & 1)) {
    const XCarry c = carry[z.carry + (size_t)y * z.zw + x];
    nan |= (c.best != c.best || c.worst != c.worst) ? 1 : 0;
    if (!xbetter<COST>(s.best, c.best)) { s.best = c.best; s.idx = c.idx; }       // the earlier disparities win ties
    if (xbetter<COST>(s.worst, c.worst)) s.worst = c.worst;
  }
  for (int k = 1; k < ngrp; ++k) {
    const XCarry n = p[(size_t)k * gstride];
    nan |= n.pad;
    if (xbetter<COST>(n.best, s.best)) { s.best = n.best; s.idx = n.idx; }
    if (!xbetter<COST>(n.worst, s.worst)) s.worst = n.worst;
  }
  if (nan) { zone_flag[it.x] = 1; return; }             // (every writer stores the same value)
  if (CARRY && (z.carry_mode & 2)) {
    carry[z.carry + (size_t)y * z.zw + x] = XCarry{s.best, s.worst, s.idx, 0};
  } else {
    const int qy = s.idx / z.sx, qx = s.idx - qy * z.sx;
    int32_t* o = out + ((size_t)z.out_off + (size_t)y * z.out_stride + x) * 3;
    o[0] = qx + z.addx; o[1] = qy + z.addy; o[2] = (s.best == s.worst) ? 0 : 0x7fffffff;   // Correlation.cc:121-133
  }
}

// (Rounds 3-4: bmx_zone_lds_kernel — one wavefront running best_of_search_convolution on a zone with everything in LDS — was an
// opt-in here.  Identical results, slower on the tile loop (15 % of a level's evaluations are in zones that fit, and a 16 x 16 zone that
// searches 264 disparities is one wavefront for 0.5 ms): removed from the product in round 4, tools/experiments/ keeps the source.)

// ---- order-freeness of an image: lowest set bit and magnitude of its pixels ------------------------------------------
// cell[0] = min over non-zero pixels of the exponent of the lowest set mantissa bit, cell[1] = max exponent,
// cell[2] bit 0: a non-finite pixel, bit 1: a negative pixel.  (Every pixel is an integer multiple of 2^cell[0] and smaller
// in magnitude than 2^(cell[1]+1).)
// One launch measures up to GRAIN_MAX images (blockIdx.y = image), GRAIN_BLOCKS workgroups each: one set of atomics per
// WORKGROUP and few workgroups, because same-address atomics serialise at the L2.
constexpr int GRAIN_MAX = 32, GRAIN_BLOCKS = 32;
struct GrainJobs {
  const float* img[GRAIN_MAX];
  int w[GRAIN_MAX], h[GRAIN_MAX];
  long long stride[GRAIN_MAX];
  int* cell[GRAIN_MAX];
};
__global__ void __launch_bounds__(256)
float_grain_kernel(GrainJobs jobs) {
  const int j = blockIdx.y;
  const float* __restrict__ img = jobs.img[j];
  const int w = jobs.w[j], h = jobs.h[j];
  const ptrdiff_t stride = (ptrdiff_t)jobs.stride[j];
  int lo = INT_MAX, hi = INT_MIN, bad = 0;
  const int cols_per_row = (w + 255) / 256;                     // column chunks of 256 pixels
  auto take = [&](unsigned u) __attribute__((always_inline)) {
    const int e = (int)((u >> 23) & 0xffu);
    unsigned m = u & 0x7fffffu;
    if (e == 0xff) { bad |= 1; return; }
    if (e == 0 && m == 0) return;                               // +-0
    if (u >> 31) bad |= 2;                                      // a negative pixel
    int base;
    if (e == 0) base = -149;                                    // subnormal: m * 2^-149
    else { m |= 0x800000u; base = e - 127 - 23; }
    lo = min(lo, base + (__ffs((int)m) - 1));
    hi = max(hi, base + (31 - __clz((int)m)));
  };
  // A workgroup walks rows blockIdx.x, blockIdx.x + gridDim.x, ...; eight 256-pixel chunks of a row are in flight per thread (out-of-range
  // chunks read pixel 0 of the row, which is harmless to count twice).  (Flat chunk indices needed a 64-bit division per request: the kernel
  // was bound by those — 84 us for two 4096^2 images.)
  // workgroups of this image: ~16 K pixels each but no more than eight rows (a row is a dependent round trip) — the launch is sized for
  // the largest image of the batch; a 32 x 32 pyramid level gets five workgroups and sets of atomics, not eighty
  const int nb = (int)min((size_t)gridDim.x, max((size_t)(h + 7) / 8, ((size_t)w * (size_t)h) >> 14));
  if ((int)blockIdx.x >= nb) return;
  for (int y = blockIdx.x; y < h; y += nb) {
    const float* row = img + (ptrdiff_t)y * stride;
    for (int xc = 0; xc < cols_per_row; xc += 8) {
      unsigned u[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        int x = (xc + k) * 256 + (int)threadIdx.x;
        if (x >= w) x = 0;
        u[k] = __float_as_uint(row[x]);
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) take(u[k]);
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = min(lo, __shfl_xor(lo, o)); hi = max(hi, __shfl_xor(hi, o)); bad |= __shfl_xor(bad, o);
  }
  __shared__ int part[4][3];
  if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6][0] = lo; part[threadIdx.x >> 6][1] = hi; part[threadIdx.x >> 6][2] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i) { lo = min(lo, part[i][0]); hi = max(hi, part[i][1]); bad |= part[i][2]; }
    // Same-address atomics serialise at the L2 (thousands of workgroups x 3 of them were most of this kernel's time): a workgroup whose
    // values cannot move the cell — it only ever moves one way, so a stale read errs on the side of an atomic — skips them.
    int* cell = jobs.cell[j];
    const int c0 = __hip_atomic_load(&cell[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int c1 = __hip_atomic_load(&cell[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int c2 = __hip_atomic_load(&cell[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lo != INT_MAX) {
      if (lo < c0) atomicMin(&cell[0], lo);
      if (hi > c1) atomicMax(&cell[1], hi);
    }
    if (bad & ~c2) atomicOr(&cell[2], bad);
  }
}

int lanes_log2_for(int D) {
  int l = 0;
  while ((1 << l) < D && l < 6) ++l;
  return l;
}

// doubles per (row, column) of a matcher's volume: enough for either layout (the tiled one rounds the disparities up to groups of XTL)
int dp_alloc(int D) {
  const int nchunk = (D + 63) / 64, dp = nchunk == 1 ? (1 << lanes_log2_for(D)) : nchunk * 64;
  return std::max(dp, (D + XTL - 1) / XTL * XTL);
}

size_t exact_scratch_budget(const vwgpu_ctx* ctx) { return (size_t)ctx->exact_scratch_mb << 20; }     // VWGPU_OPT_EXACT_SCRATCH_MB

struct Tables {
  std::vector<XZone> zones;
  std::vector<int4> col_items;
  std::vector<int2> row_items;
  std::vector<int2> sel_items;      // bmx_select_kernel: {zone, first row | x-chunk << 20}
  std::vector<int4> rs_items;       // bmx_rowsum_kernel: {zone, first row, chunk, -}
  std::vector<int4> tl_items;       // bmx_rowsel_kernel: {zone, first of 4 rows, group of 16 disparities, -}
  std::vector<int2> mg_items;       // bmx_merge_kernel: {zone, first row | x-chunk << 20}, zones of more than 16 disparities
  size_t part_records = 0;          // group records of those zones
  size_t vol_doubles = 0;
};

// appends one zone; rows = the number of volume rows to reserve for it
void add_zone(Tables& t, XZone z, int kx, int rows, bool box) {
  if (box) { z.d0 = 0; z.dn = 1; }
  else if (z.dn == 0) { z.d0 = 0; z.dn = z.sx * z.sy; }
  const int D = z.dn;
  z.lanes_log2 = lanes_log2_for(D);
  z.nchunk = (D + 63) / 64;
  const int lanes = 1 << z.lanes_log2;
  const int dp = z.nchunk == 1 ? lanes : z.nchunk * 64;
  const int cw = z.zw + kx - 1;
  z.vol = (long long)t.vol_doubles;
  t.vol_doubles += (size_t)rows * cw * (box ? dp : dp_alloc(D));
  t.zones.push_back(z);
}

// split_from: zones at least this wide take the split pass 2 (rowsum + select items), narrower ones the fused kernel (row items);
// box tables pass INT_MAX (their pass 2 is bmx_box_row_kernel over the row items).
// (zones that take the tiled form get the offsets of their group records here: the table is uploaded afterwards)
// cols_only: a segment whose rows nobody reads (partial redo): the column chains advance through it, no row / tile / selection / merge item is
// built or uploaded (thousands of rows of tables for launches that return before using them)
void build_items(Tables& t, int kx, int y_begin, int y_end, int split_from = INT_MAX, int tiled_from = INT_MAX, int tiled_to = INT_MAX, bool cols_only = false) {
  t.col_items.clear();
  t.row_items.clear();
  t.sel_items.clear();
  t.rs_items.clear();
  t.tl_items.clear();
  t.mg_items.clear();
  t.part_records = 0;
  for (size_t i = 0; i < t.zones.size(); ++i) {
    XZone& z = t.zones[i];
    const int lanes = 1 << z.lanes_log2, cpw = 256 / lanes, cw = z.zw + kx - 1, rpw = 64 / lanes;
    z.tiled = (z.zw >= tiled_from && z.zw < tiled_to) ? 1 : 0;
    if (z.tiled) {
      for (int c = 0; c < (z.dn + XTL - 1) / XTL; ++c)
        for (int x0 = 0; x0 < cw; x0 += 256 / XTL) t.col_items.push_back(make_int4((int)i, x0, c, 0));
    } else {
      for (int c = 0; c < z.nchunk; ++c)
        for (int x0 = 0; x0 < cw; x0 += cpw) t.col_items.push_back(make_int4((int)i, x0, c, 0));
    }
    const int y1 = cols_only ? std::max(y_begin, 0) : std::min(y_end, z.zh);      // (cols_only: the row loops below are empty)
    if (z.tiled) {
      const int ngrp = (z.dn + TL_X - 1) / TL_X, yb = std::max(y_begin, 0);
      for (int y0 = yb; y0 < y1; y0 += TL_R)
        for (int dg = 0; dg < ngrp; ++dg) t.tl_items.push_back(make_int4((int)i, y0, dg, 0));
      // the fused kernel's items: it recomputes the zones bmx_merge_kernel flags (a NaN cost), and only those
      for (int y0 = yb; y0 < y1; y0 += rpw) t.row_items.push_back(make_int2((int)i, y0));
      if (ngrp > 1) {
        z.part = (long long)t.part_records;
        t.part_records += (size_t)ngrp * (y1 - yb) * z.zw;
        int xlog = 0;
        while ((1 << xlog) < z.zw && xlog < 6) ++xlog;
        const int rps = 64 >> xlog, nxc = (z.zw + 63) / 64;
        for (int y0 = yb; y0 < y1; y0 += rps)
          for (int xc = 0; xc < nxc; ++xc) t.mg_items.push_back(make_int2((int)i, y0 | (xc << 20)));
      }
      continue;
    }
    if (z.zw < split_from) {
      for (int y0 = std::max(y_begin, 0); y0 < y1; y0 += rpw) t.row_items.push_back(make_int2((int)i, y0));
      continue;
    }
    for (int y0 = std::max(y_begin, 0); y0 < y1; y0 += rpw)
      for (int c = 0; c < z.nchunk; ++c) t.rs_items.push_back(make_int4((int)i, y0, c, 0));
    int xlog = 0;
    while ((1 << xlog) < z.zw && xlog < 6) ++xlog;
    const int rps = 64 >> xlog, nxc = (z.zw + 63) / 64;
    for (int y0 = std::max(y_begin, 0); y0 < y1; y0 += rps)
      for (int xc = 0; xc < nxc; ++xc) t.sel_items.push_back(make_int2((int)i, y0 | (xc << 20)));
  }
  // tallest zones first: a column chain is zh serial steps (the zones arrive sorted by ascending search volume: the level's largest zone,
  // 256 rows, would start last and finish alone)
  std::stable_sort(t.col_items.begin(), t.col_items.end(), [&](const int4& a, const int4& b) { return t.zones[a.x].zh > t.zones[b.x].zh; });
  // Longest chains first: a row item is a serial recurrence of zw steps, and the zones arrive sorted by ascending search volume —
  // the 512-wide level-0 zones would start last and finish alone (LoG + NCC tile: 5.57 -> 5.32 ms of bmx_row).
  std::stable_sort(t.row_items.begin(), t.row_items.end(), [&](const int2& a, const int2& b) {
      return (long long)t.zones[a.x].zw * t.zones[a.x].nchunk > (long long)t.zones[b.x].zw * t.zones[b.x].nchunk;
    });
  while (t.row_items.size() % 4) t.row_items.push_back(make_int2(-1, 0));
  while (t.sel_items.size() % 4) t.sel_items.push_back(make_int2(-1, 0));
  // longest chains first, as the row items
  std::stable_sort(t.rs_items.begin(), t.rs_items.end(), [&](const int4& a, const int4& b) { return t.zones[a.x].zw > t.zones[b.x].zw; });
  while (t.rs_items.size() % 4) t.rs_items.push_back(make_int4(-1, 0, 0, 0));
  // longest rows first
  std::stable_sort(t.tl_items.begin(), t.tl_items.end(), [&](const int4& a, const int4& b) { return t.zones[a.x].zw > t.zones[b.x].zw; });
  while (t.mg_items.size() % 4) t.mg_items.push_back(make_int2(-1, 0));
}

struct DevTables { const XZone* zones; const int4* col; const int2* row; const int2* sel; const int4* rs; const int4* tl; const int2* mg; int* flags; };

// the pieces of one table set, in upload order (the last one, the zone flags of bmx_merge_kernel, is a block of zeros)
struct TablePieces {
  const void* src[8];
  size_t bytes[8], off[8], all;
  std::vector<int> zeros;
  explicit TablePieces(const Tables& t) : zeros(t.mg_items.empty() ? 0 : t.zones.size(), 0) {
    const void* p[8] = {t.zones.data(), t.col_items.data(), t.row_items.data(), t.sel_items.data(), t.rs_items.data(), t.tl_items.data(),
                        t.mg_items.data(), zeros.data()};
    const size_t n[8] = {t.zones.size() * sizeof(XZone), t.col_items.size() * sizeof(int4), t.row_items.size() * sizeof(int2),
                         t.sel_items.size() * sizeof(int2), t.rs_items.size() * sizeof(int4), t.tl_items.size() * sizeof(int4),
                         t.mg_items.size() * sizeof(int2), zeros.size() * sizeof(int)};
    all = 0;
    for (int i = 0; i < 8; ++i) { src[i] = p[i]; bytes[i] = n[i]; off[i] = all; all += vwgpu_align_up(n[i], 256); }
  }
};

int upload(vwgpu_ctx* ctx, const Tables& t, char*& cursor, char* end, DevTables* d) {
  const TablePieces tp(t);
  if (cursor + tp.all > end) return vwgpu_fail(ctx, VWGPU_ERR_LOGIC, "bm_exact: table arena too small");
  // The host vectors are rebuilt for the next band / go out of scope while the stream still runs: the tables cross PCIe from a
  // piece of the pinned ring (one asynchronous copy); tables too large for the ring are copied from the vectors and waited for.
  if (char* h = static_cast<char*>(vwgpu_host_ring(ctx, tp.all))) {
    for (int i = 0; i < 8; ++i)
      if (tp.bytes[i]) memcpy(h + tp.off[i], tp.src[i], tp.bytes[i]);
    VWGPU_HIP(ctx, hipMemcpyAsync(cursor, h, tp.all, hipMemcpyHostToDevice, ctx->stream));
  } else {
    for (int i = 0; i < 8; ++i)
      if (tp.bytes[i]) VWGPU_HIP(ctx, hipMemcpyAsync(cursor + tp.off[i], tp.src[i], tp.bytes[i], hipMemcpyHostToDevice, ctx->stream));
    VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  d->zones = reinterpret_cast<const XZone*>(cursor + tp.off[0]);
  d->col = reinterpret_cast<const int4*>(cursor + tp.off[1]);
  d->row = reinterpret_cast<const int2*>(cursor + tp.off[2]);
  d->sel = reinterpret_cast<const int2*>(cursor + tp.off[3]);
  d->rs = reinterpret_cast<const int4*>(cursor + tp.off[4]);
  d->tl = reinterpret_cast<const int4*>(cursor + tp.off[5]);
  d->mg = reinterpret_cast<const int2*>(cursor + tp.off[6]);
  d->flags = reinterpret_cast<int*>(cursor + tp.off[7]);
  cursor += tp.all;
  return VWGPU_OK;
}

size_t table_bytes(const Tables& t) { return TablePieces(t).all; }

template <int COST>
void launch_pair(vwgpu_ctx* ctx, const char* n1, const char* n2, const float* A, int aw, int ah, ptrdiff_t as,
                 const float* B, int bw, int bh, ptrdiff_t bs, int kx, int ky, const Tables& t, const DevTables& d, double* vol,
                 int y_begin, int y_end, double* state, const double* prec, int32_t* out, double* outd, XCarry* carry = nullptr, XCarry* part = nullptr,
                 bool col_only = false) {
  if (col_only) vol = nullptr;                          // the column chains advance over [y_begin, y_end) without leaving their sums
  if (!t.col_items.empty()) {
    vwgpu_prof_scope ps(ctx, n1);
#define VWGPU_CK(K) hipLaunchKernelGGL((bmx_col_kernel<COST, K>), dim3((unsigned)t.col_items.size()), dim3(256), 0, ctx->stream, \
                                      A, aw, ah, as, B, bw, bh, bs, kx, ky, d.zones, d.col, vol, y_begin, y_end, state)
    switch (ky) {
      case 3: VWGPU_CK(3); break;
      case 5: VWGPU_CK(5); break;
      case 7: VWGPU_CK(7); break;
      case 9: VWGPU_CK(9); break;
      case 11: VWGPU_CK(11); break;
      case 13: VWGPU_CK(13); break;
      case 15: VWGPU_CK(15); break;
      default: VWGPU_CK(0); break;
    }
#undef VWGPU_CK
  }
  if (col_only) return;
  if (!t.row_items.empty() || !t.rs_items.empty() || !t.tl_items.empty()) {
    constexpr bool BOX = (COST == XCOST_BOX || COST == XCOST_PREC);
    const dim3 grd((unsigned)(t.row_items.size() / 4)), blk(256);
    if constexpr (BOX) {
      vwgpu_prof_scope ps(ctx, n2);
      hipLaunchKernelGGL((bmx_box_row_kernel<COST>), grd, blk, 0, ctx->stream, kx, d.zones, d.row, vol, y_begin, y_end, outd);
    } else {
      // Long chains (wide zones, whole rasters): the recurrence alone, then a parallel selection.  Short chains — most zones of a
      // pyramid level — keep the fused kernel: one pass over their volumes, and its register budget is set by THEIR chunk counts.
      if (!t.rs_items.empty()) {
        {
          vwgpu_prof_scope ps(ctx, n2);
          hipLaunchKernelGGL(bmx_rowsum_kernel, dim3((unsigned)(t.rs_items.size() / 4)), blk, 0, ctx->stream, kx, d.zones, d.rs, vol, y_begin, y_end);
        }
        vwgpu_prof_scope ps(ctx, "bmx_select");
        const dim3 sgrd((unsigned)(t.sel_items.size() / 4));
        if (carry) hipLaunchKernelGGL((bmx_select_kernel<COST, true>), sgrd, blk, 0, ctx->stream, kx, d.zones, d.sel, vol, y_begin, y_end, prec, out, carry);
        else hipLaunchKernelGGL((bmx_select_kernel<COST, false>), sgrd, blk, 0, ctx->stream, kx, d.zones, d.sel, vol, y_begin, y_end, prec, out, carry);
      }
      const bool tiled = !t.tl_items.empty();           // then the row items are the redo pass over the zones bmx_merge_kernel flags
      if (tiled) {
        {
          vwgpu_prof_scope ps(ctx, "bmx_rowsel");
          const dim3 tgrd((unsigned)t.tl_items.size());
#define VWGPU_RS(K) do { if (carry) hipLaunchKernelGGL((bmx_rowsel_kernel<COST, true, K>), tgrd, dim3(64), 0, ctx->stream, kx, d.zones, d.tl, vol, y_begin, y_end, prec, out, carry, part); \
                         else hipLaunchKernelGGL((bmx_rowsel_kernel<COST, false, K>), tgrd, dim3(64), 0, ctx->stream, kx, d.zones, d.tl, vol, y_begin, y_end, prec, out, carry, part); } while (0)
          switch (kx) {
            case 3: VWGPU_RS(3); break;
            case 5: VWGPU_RS(5); break;
            case 7: VWGPU_RS(7); break;
            case 9: VWGPU_RS(9); break;
            case 11: VWGPU_RS(11); break;
            case 13: VWGPU_RS(13); break;
            case 15: VWGPU_RS(15); break;
            default: VWGPU_RS(0); break;
          }
#undef VWGPU_RS
        }
        if (t.mg_items.empty()) return;                  // no zone searches more than 16 disparities: every result is final
        vwgpu_prof_scope ps(ctx, "bmx_merge");
        const dim3 mgrd((unsigned)(t.mg_items.size() / 4));
        if (carry) hipLaunchKernelGGL((bmx_merge_kernel<COST, true>), mgrd, blk, 0, ctx->stream, d.zones, d.mg, y_begin, y_end, part, out, carry, d.flags);
        else hipLaunchKernelGGL((bmx_merge_kernel<COST, false>), mgrd, blk, 0, ctx->stream, d.zones, d.mg, y_begin, y_end, part, out, carry, d.flags);
      }
      const int* zflag = tiled ? d.flags : nullptr;
      if (!t.row_items.empty()) {
        int nch = 1;
        for (const int2& it : t.row_items) if (it.x >= 0) nch = std::max(nch, t.zones[it.x].nchunk);
        vwgpu_prof_scope ps(ctx, n2);
        if (carry)
          hipLaunchKernelGGL((bmx_row_fused_kernel<COST, XMAX_CHUNKS, true>), grd, blk, 0, ctx->stream, kx, d.zones, d.row, vol, y_begin, y_end, prec, out, outd, carry, zflag);
        else if (nch == 1)
          hipLaunchKernelGGL((bmx_row_fused_kernel<COST, 1>), grd, blk, 0, ctx->stream, kx, d.zones, d.row, vol, y_begin, y_end, prec, out, outd, static_cast<XCarry*>(nullptr), zflag);
        else if (nch <= 3)
          hipLaunchKernelGGL((bmx_row_fused_kernel<COST, 3>), grd, blk, 0, ctx->stream, kx, d.zones, d.row, vol, y_begin, y_end, prec, out, outd, static_cast<XCarry*>(nullptr), zflag);
        else
          hipLaunchKernelGGL((bmx_row_fused_kernel<COST, XMAX_CHUNKS>), grd, blk, 0, ctx->stream, kx, d.zones, d.row, vol, y_begin, y_end, prec, out, outd, static_cast<XCarry*>(nullptr), zflag);
      }
    }
  }
}

}  // namespace

// Any search volume: zones of more than 512 disparities are swept in disparity groups with the compare-chain state carried in HBM.
bool vwgpu_bm_exact_supported(int sx, int sy) { return sx > 0 && sy > 0 && (long long)sx * sy <= INT_MAX; }

// The order-freeness test.  Every pixel of both images is an integer multiple of g = 2^lo and smaller than 2^(hi+1); then
// every cost element is a multiple of g (SAD) or g^2 (SSD / NCC: the float product rounds to a coarser multiple) and
// every intermediate value of the reference's chains — at most 2 * kx * ky elements in magnitude — is exactly
// representable in float64 iff it stays below 2^53 units, in which case ANY summation order returns the same bits.
// mantissa bits the largest intermediate value of the chains can need (INT_MAX: a non-finite pixel; 0: all-zero images)
int vwgpu_sums_bits(int cost_type, int kx, int ky, int lo, int hi, int nonfinite) {
  if (nonfinite & 1) return INT_MAX;
  if (lo == INT_MAX) return 0;
  int lg = 0;
  while ((1LL << lg) < (long long)kx * ky) ++lg;
  const long long E = (long long)hi + 1;                // |pixel| < 2^E
  long long bits;
  if (cost_type == VWGPU_ABSOLUTE_DIFFERENCE) bits = (E + 1) + lg + 1 - lo;              // |a-b| < 2^(E+1)
  else if (cost_type == VWGPU_SQUARED_DIFFERENCE) bits = 2 * (E + 1) + 1 + lg + 1 - 2LL * lo;
  else bits = 2 * E + 1 + lg + 1 - 2LL * lo;
  return (int)std::min<long long>(bits, INT_MAX - 1);
}
bool vwgpu_sums_order_free(int cost_type, int kx, int ky, int lo, int hi, int nonfinite) {
  return vwgpu_sums_bits(cost_type, kx, ky, lo, hi, nonfinite) <= 53;
}

// Measures images[i] into d_cells[i] (3 ints each, initialised by the caller with {INT_MAX, INT_MIN, 0}); several images may
// share a cell.  Asynchronous.
void vwgpu_launch_float_grain(vwgpu_ctx* ctx, int n, const float* const* img, const int* w, const int* h, const ptrdiff_t* stride, int* const* d_cells) {
  for (int i0 = 0; i0 < n; i0 += GRAIN_MAX) {
    GrainJobs jobs;
    int m = 0;
    size_t largest = 0;
    for (int i = i0; i < n && m < GRAIN_MAX; ++i) {
      if (!img[i] || w[i] <= 0 || h[i] <= 0) continue;
      jobs.img[m] = img[i]; jobs.w[m] = w[i]; jobs.h[m] = h[i]; jobs.stride[m] = stride[i]; jobs.cell[m] = d_cells[i];
      largest = std::max(largest, (size_t)w[i] * h[i]);
      ++m;
    }
    if (m == 0) continue;
    // workgroups per image: ~16 K pixels each (8 loads in flight per thread), at least GRAIN_BLOCKS, at most 2048
    const unsigned blocks = (unsigned)std::min<size_t>(2048, std::max<size_t>(GRAIN_BLOCKS, largest >> 14));
    vwgpu_prof_scope ps(ctx, "float_grain");
    hipLaunchKernelGGL(float_grain_kernel, dim3(blocks, (unsigned)m), dim3(256), 0, ctx->stream, jobs);
  }
}

// Measures (lo, hi, nonfinite) over up to two images.  Synchronises the stream.
int vwgpu_float_grain(vwgpu_ctx* ctx, const float* a, int aw, int ah, ptrdiff_t as, const float* b, int bw, int bh, ptrdiff_t bs,
                      int* lo, int* hi, int* nonfinite) {
  int rc = vwgpu_arena_reserve(ctx, &ctx->misc, 256);
  if (rc) return rc;
  int* cell = static_cast<int*>(ctx->misc.base) + 16;
  const int init[3] = {INT_MAX, INT_MIN, 0};
  VWGPU_HIP(ctx, hipMemcpyAsync(cell, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
  const float* imgs[2] = {a, b};
  const int ws[2] = {aw, bw}, hs[2] = {ah, bh};
  const ptrdiff_t ss[2] = {as, bs};
  int* cells[2] = {cell, cell};
  vwgpu_launch_float_grain(ctx, 2, imgs, ws, hs, ss, cells);
  int got[3];
  VWGPU_HIP(ctx, hipMemcpyAsync(got, cell, sizeof got, hipMemcpyDeviceToHost, ctx->stream));
  VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
  *lo = got[0]; *hi = got[1]; *nonfinite = got[2];
  return VWGPU_OK;
}

// One group of zones (its column-sum volumes fit the scratch budget, or it is a single zone swept in row bands).
// dgroup != nullptr: the single zone is served for the disparities [d0, d0 + dn) only, chain state in `carry` (see XCarry).
struct DGroup { int d0, dn, carry_mode; XCarry* carry; };
static int run_group(vwgpu_ctx* ctx, int cost_type, const float* A, int aw, int ah, ptrdiff_t as,
                     const float* B, int bw, int bh, ptrdiff_t bs, int kx, int ky,
                     const vwgpu_zone_task* zones, int n, int32_t* out, const DGroup* dgroup = nullptr, const int* const* gates = nullptr,
                     const int* rows = nullptr, int nranges = 0) {
  const bool ncc = cost_type == VWGPU_CROSS_CORRELATION;
  Tables match, box;                                  // box: the NCC precision images of every zone, left and right crops in one table
  size_t prec_doubles = 0;
  for (int i = 0; i < n; ++i) {
    const vwgpu_zone_task& s = zones[i];
    XZone z{};
    z.ax = s.ax; z.ay = s.ay; z.bx = s.bx; z.by = s.by; z.zw = s.zw; z.zh = s.zh; z.sx = s.sx; z.sy = s.sy;
    z.out_off = s.out_off; z.out_stride = s.out_stride; z.addx = s.addx; z.addy = s.addy;
    z.gate = gates ? (long long)reinterpret_cast<uintptr_t>(gates[i]) : 0;
    if (dgroup) { z.d0 = dgroup->d0; z.dn = dgroup->dn; z.carry_mode = dgroup->carry_mode; z.carry = 0; }
    if (ncc) {
      // NCCCost ctor over the zone's own crops (CostFunctions.h:214-219): box sums restart at the crop origin
      z.lprec = (long long)prec_doubles; prec_doubles += (size_t)s.zw * s.zh;
      z.rprec = (long long)prec_doubles; prec_doubles += (size_t)(s.zw + s.sx - 1) * (s.zh + s.sy - 1);
      XZone a{}; a.ax = s.ax; a.ay = s.ay; a.zw = s.zw; a.zh = s.zh; a.sx = a.sy = 1; a.lprec = z.lprec; a.img = 0; a.gate = z.gate;
      add_zone(box, a, kx, a.zh, true);
      XZone b{}; b.ax = s.bx; b.ay = s.by; b.zw = s.zw + s.sx - 1; b.zh = s.zh + s.sy - 1; b.sx = b.sy = 1; b.lprec = z.rprec; b.img = 1; b.gate = z.gate;
      add_zone(box, b, kx, b.zh, true);
    }
    add_zone(match, z, kx, z.zh, false);
  }
  if (match.zones.empty()) return VWGPU_OK;

  // band mode: a single zone whose volume exceeds the budget is swept in row bands
  const size_t budget = exact_scratch_budget(ctx);
  int band = INT_MAX;
  size_t state_doubles = 0;
  const bool partial = rows && nranges > 0 && match.zones.size() == 1 && !dgroup;      // only some output rows are wanted (a certified raster's flagged tile rows)
  if ((match.vol_doubles * 8 > budget || partial) && match.zones.size() == 1) {
    XZone& z = match.zones[0];
    const int dp = dp_alloc(z.dn), cw = z.zw + kx - 1;
    const size_t row = (size_t)cw * dp * 8;
    band = (int)std::max<size_t>(1, budget / row);
    if (partial) {                                              // bands no taller than the tallest wanted range; the chain state is always carried
      int tallest = 1;
      for (int i = 0; i < nranges; ++i) tallest = std::max(tallest, rows[2 * i + 1] - rows[2 * i]);
      band = std::min(band, tallest);
      match.vol_doubles = (size_t)band * cw * dp; state_doubles = (size_t)cw * dp;
    } else if (band >= z.zh) band = INT_MAX;
    else { match.vol_doubles = (size_t)band * cw * dp; state_doubles = (size_t)cw * dp; }
  }
  // zones at least this wide take the split pass 2, narrower ones the tiled kernel (VWGPU_OPT_EXACT_SPLIT: 0 = 1024 pixels — whole rasters
  // run the recurrence alone + a parallel selection; 1 = split, 2 = fused, 3 = tiled for every zone)
  const int split_from = ctx->exact_split == 1 ? 0 : (ctx->exact_split == 2 || ctx->exact_split == 3 ? INT_MAX : 1024);
  const int tiled_from = ctx->exact_split == 1 || ctx->exact_split == 2 ? INT_MAX : 0;
  const int tiled_to = ctx->exact_split == 3 ? INT_MAX : split_from;      // tiled: tiled_from <= width < tiled_to
  build_items(match, kx, 0, band, split_from, tiled_from, tiled_to);
  build_items(box, kx, 0, INT_MAX);
  const size_t vol_need = std::max(match.vol_doubles, box.vol_doubles);
  const size_t part_doubles = match.part_records * (sizeof(XCarry) / sizeof(double));      // (the first band is the tallest)
  int rc = vwgpu_arena_reserve(ctx, &ctx->xvol, (vol_need + state_doubles + prec_doubles + part_doubles) * 8 + 1024);
  if (rc) return rc;
  double* vol = static_cast<double*>(ctx->xvol.base);
  double* state = state_doubles ? vol + vol_need : nullptr;
  double* prec = vol + vol_need + state_doubles;
  XCarry* part = reinterpret_cast<XCarry*>(prec + prec_doubles);
  // tables of every launch of this call live side by side (uploads are stream ordered)
  size_t tb = table_bytes(match) + table_bytes(box) + 4096;
  if (band != INT_MAX) tb += (size_t)((match.zones[0].zh + band - 1) / band + 2 * (size_t)std::max(nranges, 0) + 2) * (table_bytes(match) + 1024);
  rc = vwgpu_arena_reserve(ctx, &ctx->xtab, tb);
  if (rc) return rc;
  char* cur = static_cast<char*>(ctx->xtab.base);
  char* end = cur + ctx->xtab.cap;
  DevTables d;
  if (ncc) {
    if ((rc = upload(ctx, box, cur, end, &d))) return rc;
    launch_pair<XCOST_PREC>(ctx, "bmx_prec_col", "bmx_prec_row", A, aw, ah, as, B, bw, bh, bs, kx, ky, box, d, vol, 0, INT_MAX, nullptr, nullptr, nullptr, prec);
  }
  const int zh0 = match.zones[0].zh;
  // the row segments of the call: {first row, last row + 1, column chains only}
  struct Seg { int yb, ye; bool col_only; };
  std::vector<Seg> segs;
  if (partial) {
    int y = 0;
    for (int i = 0; i < nranges; ++i) {
      const int a = std::max(rows[2 * i], y), b = std::min(rows[2 * i + 1], zh0);
      if (b <= a) continue;
      if (a > y) segs.push_back(Seg{y, a, true});              // nobody reads these rows: the chains run through them, nothing is stored
      for (int yb = a; yb < b; yb += band) segs.push_back(Seg{yb, std::min(yb + band, b), false});
      y = b;
    }                                                           // (rows below the last wanted range are not visited at all)
  } else {
    const int nbands = band == INT_MAX ? 1 : (zh0 + band - 1) / band;
    for (int b = 0; b < nbands; ++b) segs.push_back(Seg{band == INT_MAX ? 0 : b * band, band == INT_MAX ? INT_MAX : b * band + band, false});
  }
  for (const Seg& sg : segs) {
    const int yb = sg.yb, ye = sg.ye;
    const bool col_only = sg.col_only;
    if (band != INT_MAX) build_items(match, kx, yb, ye, split_from, tiled_from, tiled_to, col_only);
    if ((rc = upload(ctx, match, cur, end, &d))) return rc;
    switch (cost_type) {
      case VWGPU_CROSS_CORRELATION:
        launch_pair<VWGPU_CROSS_CORRELATION>(ctx, "bmx_col", "bmx_row", A, aw, ah, as, B, bw, bh, bs, kx, ky, match, d, vol, yb, ye, state, prec, out, nullptr, dgroup ? dgroup->carry : nullptr, part, col_only); break;
      case VWGPU_SQUARED_DIFFERENCE:
        launch_pair<VWGPU_SQUARED_DIFFERENCE>(ctx, "bmx_col", "bmx_row", A, aw, ah, as, B, bw, bh, bs, kx, ky, match, d, vol, yb, ye, state, prec, out, nullptr, dgroup ? dgroup->carry : nullptr, part, col_only); break;
      default:
        launch_pair<VWGPU_ABSOLUTE_DIFFERENCE>(ctx, "bmx_col", "bmx_row", A, aw, ah, as, B, bw, bh, bs, kx, ky, match, d, vol, yb, ye, state, prec, out, nullptr, dgroup ? dgroup->carry : nullptr, part, col_only); break;
    }
  }
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}

// All zones in the reference's summation order.  A / B dense or strided images (clamped reads).  Zones are processed in
// groups whose column-sum volumes fit the scratch budget (VWGPU_OPT_EXACT_SCRATCH_MB, default 4096).
int vwgpu_launch_bm_exact(vwgpu_ctx* ctx, int cost_type, const float* A, int aw, int ah, ptrdiff_t as,
                          const float* B, int bw, int bh, ptrdiff_t bs, int kx, int ky,
                          const vwgpu_zone_task* zones, int n, int32_t* out, const int* d_gate, const int* rows, int nranges) {
  if (n != 1 || d_gate) { rows = nullptr; nranges = 0; }          // row ranges: single-zone calls (a raster whose certified pass flagged some tile rows)
  const size_t budget = exact_scratch_budget(ctx);
  std::vector<vwgpu_zone_task> group;
  std::vector<const int*> group_gate;                 // d_gate != nullptr: zone i works only if d_gate[i] != 0 (decided on the device)
  auto gates_of = [&]() -> const int* const* { return d_gate ? group_gate.data() : nullptr; };
  size_t bytes = 0;
  for (int i = 0; i < n; ++i) {
    const vwgpu_zone_task& s = zones[i];
    if (s.zw <= 0 || s.zh <= 0 || s.sx <= 0 || s.sy <= 0) continue;
    if (!vwgpu_bm_exact_supported(s.sx, s.sy))
      return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "bm_exact: %d x %d disparities exceed the index range", s.sx, s.sy);
    if ((long long)s.sx * s.sy > 64LL * XMAX_CHUNKS) {
      // more disparities than a lane can hold: groups of 512 in index order, one launch pair each, compare-chain state in HBM
      if (!group.empty()) {
        int rc = run_group(ctx, cost_type, A, aw, ah, as, B, bw, bh, bs, kx, ky, group.data(), (int)group.size(), out, nullptr, gates_of());
        if (rc) return rc;
        group.clear(); group_gate.clear(); bytes = 0;
      }
      int rc = vwgpu_arena_reserve(ctx, &ctx->xcarry, (size_t)s.zw * s.zh * sizeof(XCarry) + 256);
      if (rc) return rc;
      const int D = s.sx * s.sy, G = 64 * XMAX_CHUNKS;
      for (int d0 = 0; d0 < D; d0 += G) {
        DGroup dg{d0, std::min(G, D - d0), (d0 > 0 ? 1 : 0) | (d0 + G < D ? 2 : 0), static_cast<XCarry*>(ctx->xcarry.base)};
        const int* one_gate = d_gate ? d_gate + i : nullptr;
        rc = run_group(ctx, cost_type, A, aw, ah, as, B, bw, bh, bs, kx, ky, &s, 1, out, &dg, d_gate ? &one_gate : nullptr);
        if (rc) return rc;
      }
      continue;
    }
    const size_t need = (size_t)s.zh * (s.zw + kx - 1) * dp_alloc(s.sx * s.sy) * 8;
    if (!group.empty() && bytes + need > budget) {
      int rc = run_group(ctx, cost_type, A, aw, ah, as, B, bw, bh, bs, kx, ky, group.data(), (int)group.size(), out, nullptr, gates_of());
      if (rc) return rc;
      group.clear(); group_gate.clear(); bytes = 0;
    }
    group.push_back(s);
    if (d_gate) group_gate.push_back(d_gate + i);
    bytes += need;
  }
  if (group.empty()) return VWGPU_OK;
  int rcg = run_group(ctx, cost_type, A, aw, ah, as, B, bw, bh, bs, kx, ky, group.data(), (int)group.size(), out, nullptr, gates_of(), rows, nranges);
  if (rcg == VWGPU_ERR_LOGIC && rows && nranges > 0) {            // the partial path ran out of table space: the whole raster in the reference's order
    ctx->err.clear();
    rcg = run_group(ctx, cost_type, A, aw, ah, as, B, bw, bh, bs, kx, ky, group.data(), (int)group.size(), out, nullptr, gates_of(), nullptr, 0);
  }
  return rcg;
}

// fast_box_sum<double>(image, kernel) (Algorithms.h:41-43) in the reference's order.  d_out: (w-kx+1) x (h-ky+1) doubles, dense.
int vwgpu_launch_box_sum_exact(vwgpu_ctx* ctx, const float* img, int w, int h, ptrdiff_t stride, int kx, int ky, double* d_out) {
  Tables t;
  XZone z{};
  z.zw = w - kx + 1; z.zh = h - ky + 1; z.sx = z.sy = 1; z.lprec = 0;
  add_zone(t, z, kx, z.zh, true);
  const size_t budget = exact_scratch_budget(ctx);
  int band = INT_MAX;
  size_t state_doubles = 0;
  if (t.vol_doubles * 8 > budget) {
    band = (int)std::max<size_t>(1, budget / ((size_t)w * 8));
    t.vol_doubles = (size_t)band * w;
    state_doubles = (size_t)w;
  }
  int rc = vwgpu_arena_reserve(ctx, &ctx->xvol, (t.vol_doubles + state_doubles) * 8 + 1024);
  if (rc) return rc;
  double* vol = static_cast<double*>(ctx->xvol.base);
  double* state = state_doubles ? vol + t.vol_doubles : nullptr;
  build_items(t, kx, 0, band);
  const int nb = band == INT_MAX ? 1 : (z.zh + band - 1) / band;
  rc = vwgpu_arena_reserve(ctx, &ctx->xtab, (size_t)nb * (table_bytes(t) + 1024) + 4096);
  if (rc) return rc;
  char* cur = static_cast<char*>(ctx->xtab.base);
  char* end = cur + ctx->xtab.cap;
  for (int b = 0; b < nb; ++b) {
    const int yb = band == INT_MAX ? 0 : b * band, ye = band == INT_MAX ? INT_MAX : yb + band;
    if (band != INT_MAX) build_items(t, kx, yb, ye);
    DevTables d;
    if ((rc = upload(ctx, t, cur, end, &d))) return rc;
    launch_pair<XCOST_BOX>(ctx, "box_sum_col", "box_sum_row", img, w, h, stride, nullptr, 0, 0, 0, kx, ky, t, d, vol, yb, ye, state, nullptr, nullptr, d_out);
  }
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}
