// sgm.hip — semi-global matching (vw::stereo::calc_disparity_sgm, src/vw/Stereo/SGM.cc:167-229) resident in HBM.
//
//   u8_convert                      ImageThresh.h:275-286        minmax_kernel + u8_convert_kernel (float64 scale, truncation)
//   census / ternary census         CensusTransform.h:64-340     census_kernel -> one uint64 per pixel
//   populate_disp_bound_image       SGM.cc:241-499               mask_extent kernels + bounds_kernel (masks, 2x previous level +- buffer)
//   constrain_disp_bound_image      SGM.cc:502-672               constrain_kernel (full-search pixels adopt the box of their neighbours)
//   calc_main_buf_size              SGM.cc:677-731               row_count_kernel + host prefix over rows + row_scan_kernel (ragged starts)
//   compute_disparity_costs         SGM.cc:1740-1893, :40-75     cost_kernel: popcount(left_census ^ right_census) inside each pixel's bounds
//   accum_sgm_multithread           SGM.cc:2462-2612             one wavefront per scan line, path vector in registers, one direction per launch,
//                                                                plain read-modify-write of the sums: path_ring_kernel (round 6: one search row,
//                                                                every pixel the full range, strides <= 160 B; costs and sums prefetched by
//                                                                LDS-DMA into a ring per wave, four neighbouring lines per workgroup),
//                                                                path_uniform_reg_kernel (round 2: prefetch in registers; wider vectors),
//                                                                path_uniform_kernel / path_inplace_kernel / path_kernel (2-D or ragged boxes,
//                                                                all directions in one launch, atomics), path_multi_kernel (round 6: ragged
//                                                                boxes of a large level, two or four scan lines per wavefront)
//   evaluate_path (SSE semantics)   SGM.cc:936-984, 1013-1150    saturating u16 add/sub, 8-neighbour 2-D disparity adjacency with
//                                                                repetition at the global range border, BAD_VAL outside the prior's box
//   select_best_disparity           SGM.cc:1159-1284             wta_kernel: (value << 16 | index) min = first minimum; tie smoothing loop
//   create_disparity_view_subpixel  SGM.cc:1497-1614             subpixel_kernel (linear / poly4 / cosine / LC-blend / 2-D parabola)
//
// Layout in HBM: per output pixel a box [min_x,max_x] x [min_y,max_y] of searched disparities (4 x int32) and a uint64 start
// into two ragged arrays: cost (u8) and accumulated cost (u16) — the reference's m_cost_buffer / m_accum_buffer
// (SGM.cc:733-751).  Algorithmic bytes ("materialised volume" model): 20 + 11 D bytes per pixel
// (D = disparities per pixel): cost written once (D) and read by 8 paths, accum read-modify-written per path — 23 GB for
// 2048^2 x 129: the seven read-modify-write passes run at 4.5-4.9 TB/s of mixed traffic, the first and the last pass at the issue rate of the
// recurrence (round 6: 5.05 ms; the recurrence alone 3.9 ms; profiles/r06_sgm_traffic.md).
#include <algorithm>
#include <cmath>
#include <vector>

#include <cstdlib>
#include "vwgpu_internal.h"
#include "mgm_schedule.h"
#include <type_traits>

namespace {

struct B4 { int x0, y0, x1, y1; };
__device__ __forceinline__ int b4_count(B4 b) { return (b.x1 - b.x0 + 1) * (b.y1 - b.y0 + 1); }

// ---- u8_convert -------------------------------------------------------------------------------------------------------

__device__ __forceinline__ unsigned f2ord(float f) { unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__global__ void minmax_kernel(const float* __restrict__ img, ptrdiff_t stride, int w, int h, unsigned* __restrict__ mm) {
  unsigned lo = 0xffffffffu, hi = 0u;
  for (int y = blockIdx.y; y < h; y += gridDim.y)
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < w; x += gridDim.x * blockDim.x) {
      const unsigned o = f2ord(img[(ptrdiff_t)y * stride + x]);
      lo = min(lo, o); hi = max(hi, o);
    }
  for (int s = 32; s > 0; s >>= 1) { lo = min(lo, (unsigned)__shfl_xor((int)lo, s)); hi = max(hi, (unsigned)__shfl_xor((int)hi, s)); }
  // one pair of atomics per workgroup: same-address atomics serialise at the L2
  __shared__ unsigned part[4][2];
  if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6][0] = lo; part[threadIdx.x >> 6][1] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < (int)(blockDim.x >> 6); ++k) { lo = min(lo, part[k][0]); hi = max(hi, part[k][1]); }
    atomicMin(mm, lo); atomicMax(mm + 1, hi);
  }
}

__global__ void u8_convert_kernel(const float* __restrict__ img, ptrdiff_t stride, int w, int h, const unsigned* __restrict__ mm,
                                  uint8_t* __restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= w) return;
  double min_val = (double)ord2f(mm[0]), max_val = (double)ord2f(mm[1]);
  if (max_val == min_val) max_val = min_val + 1.0;
  const float old_min = (float)min_val, old_max = (float)max_val;
  const double ratio = (old_max == old_min) ? 0.0 : (double)(255.0f - 0.0f) / (double)(old_max - old_min);
  float v = img[(ptrdiff_t)y * stride + x];
  if (v > old_max) v = old_max;
  if (v < old_min) v = old_min;
  const float n = (float)((double)(v - old_min) * ratio + (double)0.0f);
  out[(size_t)y * w + x] = (uint8_t)n;
}

// ---- census -----------------------------------------------------------------------------------------------------------

__constant__ int c9cols[32] = {0, 4, 8, 1, 3, 5, 7, 2, 4, 6, 1, 4, 7, 0, 2, 3, 5, 6, 8, 1, 4, 7, 2, 4, 6, 1, 3, 5, 7, 0, 4, 8};
__constant__ int c9rows[32] = {0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4, 4, 4, 4, 5, 5, 5, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8};
__constant__ int c7cols[32] = {0, 2, 3, 4, 6, 1, 3, 5, 0, 2, 3, 4, 6, 0, 1, 2, 4, 5, 6, 0, 2, 3, 4, 6, 1, 3, 5, 0, 2, 3, 4, 6};
__constant__ int c7rows[32] = {0, 0, 0, 0, 0, 1, 1, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 4, 4, 4, 4, 4, 5, 5, 5, 6, 6, 6, 6, 6};

// out(c, r) = census word of the window centred at (c + hk, r + hk); out is (w - 2hk) x (h - 2hk)
__global__ void census_kernel(const uint8_t* __restrict__ img, int w, int h, int k, int ternary, int thr, uint64_t* __restrict__ out) {
  const int hk = (k - 1) / 2, ow = w - 2 * hk, oh = h - 2 * hk;
  const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y * blockDim.y + threadIdx.y;
  if (c >= ow || r >= oh) return;
  const int col = c + hk, row = r + hk;
  auto at = [&](int x, int y) -> int { return img[(size_t)y * w + x]; };
  const int center = at(col, row);
  uint64_t o = 0, addend = 1;
  if (!ternary) {
    if (k == 9) {
      for (int i = 0; i < 32; ++i) { if (at(col + c9cols[i] - 4, row + c9rows[i] - 4) > center) o += addend; addend *= 2; }
    } else {   // 3x3 (explicit weights 128..1 = the same descending raster order), 5x5, 7x7
      for (int y = row + hk; y >= row - hk; --y)
        for (int x = col + hk; x >= col - hk; --x) {
          if (y == row && x == col) continue;
          if (at(x, y) > center) o += addend;
          addend *= 2;
        }
    }
  } else {
    const int lo = center - thr, hi = center + thr;
    auto tern = [&](int val) { if (val >= lo) { o += addend; if (val > hi) o += addend * 2; } addend *= 4; };
    if (k == 7) { for (int i = 0; i < 32; ++i) tern(at(col + c7cols[i] - 3, row + c7rows[i] - 3)); }
    else if (k == 9) { for (int i = 0; i < 32; ++i) tern(at(col + c9cols[i] - 4, row + c9rows[i] - 4)); }
    else {
      for (int y = row + hk; y >= row - hk; --y)
        for (int x = col + hk; x >= col - hk; --x) { if (y == row && x == col) continue; tern(at(x, y)); }
    }
  }
  out[(size_t)r * ow + c] = o;
}

// ---- disparity bounds -------------------------------------------------------------------------------------------------

// Rows of the right mask that hold a valid pixel in the columns [0, ocols): ext[0] = first such row, ext[1] = last such row >= 1
// (SGM.cc:303-327 scans every column from both ends, the scan from the bottom stopping above row 0 — quirk kept).  Initialised by
// the host to {rmh - 1, 0}.  One wave per row, coalesced.
__global__ void __launch_bounds__(256)
mask_col_extent_kernel(const uint8_t* __restrict__ rmask, int rmw, int rmh, int ocols, int* __restrict__ ext) {
  const int row = blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rmh) return;
  const uint8_t* m = rmask + (size_t)row * rmw;
  bool any = false;
  for (int c = lane; c < ocols && c < rmw; c += 64) any |= m[c] > 0;
  if (__any(any) && lane == 0) {
    atomicMin(ext, row);
    if (row > 0) atomicMax(ext + 1, row);
  }
}
// Per row of the right mask: (first, last) valid column, the last one searched down to column 1 only ((-1, -2) without one).
__global__ void __launch_bounds__(256)
mask_row_extent_kernel(const uint8_t* __restrict__ rmask, int rmw, int orows, int2* __restrict__ rowext) {
  const int r = blockIdx.x * 4 + ((int)threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= orows) return;
  const uint8_t* m = rmask + (size_t)r * rmw;
  int mn = 0x7fffffff, mx = -2;
  for (int i = lane; i < rmw; i += 64)
    if (m[i] > 0) { mn = min(mn, i); if (i > 0) mx = max(mx, i); }
  for (int s = 32; s > 0; s >>= 1) { mn = min(mn, __shfl_xor(mn, s)); mx = max(mx, __shfl_xor(mx, s)); }
  if (lane == 0) rowext[r] = mx > 0 ? make_int2(mn, mx) : make_int2(-1, -2);
}

struct SgmGeom { int min_dx, min_dy, max_dx, max_dy, num_dx, num_dy, sbx, sby, ocols, orows; };

__global__ void bounds_kernel(SgmGeom g, const uint8_t* lmask, const uint8_t* rmask_present,
                              const int* ext, const int2* rowext, const int32_t* prev, int pw, int ph,
                              B4* bounds, uint8_t* full_search) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, r = blockIdx.y;
  if (c >= g.ocols) return;
  const size_t idx = (size_t)r * g.ocols + c;
  B4 b{0, 0, -1, -1};                       // ZERO_SEARCH_AREA
  uint8_t fs = 0;
  if (!(lmask && lmask[idx] == 0)) {
    bool good = false;
    int dxs = 0, dys = 0;
    if (prev) {
      const int c_in = c / 2, r_in = r / 2;
      if (!(c_in >= pw || r_in >= ph)) {
        const int32_t* d = prev + ((size_t)r_in * pw + c_in) * 3;
        dxs = d[0] * 2; dys = d[1] * 2;
        const bool on_edge = (g.num_dx >= 10 && (dxs <= g.min_dx || dxs >= g.max_dx)) || (g.num_dy >= 10 && (dys <= g.min_dy || dys >= g.max_dy));
        good = d[2] != 0 && !on_edge;
      }
    }
    if (good) {
      b.x0 = max(dxs - g.sbx, g.min_dx); b.x1 = min(dxs + g.sbx, g.max_dx);
      b.y0 = max(dys - g.sby, g.min_dy); b.y1 = min(dys + g.sby, g.max_dy);
    } else {
      b.x0 = g.min_dx; b.y0 = g.min_dy; b.x1 = g.max_dx; b.y1 = g.max_dy;
      fs = 255;
    }
    if (rmask_present) {
      int vx0 = rowext[r].x, vx1 = rowext[r].y, vy0 = ext[0], vy1 = ext[1];
      if (!(vx0 >= vx1 || vy0 >= vy1)) { vx0 -= c; vx1 -= c; vy0 -= r; vy1 -= r; }    // BBox -= on an empty box is a no-op
      vx0 = max(vx0, b.x0); vy0 = max(vy0, b.y0); vx1 = min(vx1, b.x1); vy1 = min(vy1, b.y1);
      if (vx0 > vx1 || vy0 > vy1) { b = B4{0, 0, -1, -1}; fs = 0; }
      else b = B4{vx0, vy0, vx1, vy1};
    }
  }
  bounds[idx] = b;
  full_search[idx] = fs;
}

// A wave owns CONSTRAIN_CPW consecutive pixels of a row and visits its full-search pixels one after the other, all lanes scanning that
// pixel's (2 range + 1)^2 neighbourhood together (one lane per pixel walked the window alone: 441 or 2601 dependent 17-byte loads with
// a few lanes of the wave active).  ONE pixel per wave (round 6; 64 before): full-search pixels come in patches, a wave inside one
// visited 64 of them in a row — 450 dependent window trips — while most waves had none: 0.15-0.19 ms per 1024^2 level; with 8 / 2 / 1
// pixels per wave the ten launches of a pyramid tile take 0.69 / 0.53 / 0.46 ms (0.89 before) — a wave without such a pixel is one load.
constexpr int CONSTRAIN_CPW = 1;
__global__ void __launch_bounds__(256)
constrain_kernel(SgmGeom g, const uint8_t* __restrict__ full_search, B4* __restrict__ bounds, int range, int conserve) {
  const int lane = threadIdx.x & 63;
  const int c_base = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * CONSTRAIN_CPW, c_mine = c_base + lane, r = blockIdx.y;
  const bool mine = lane < CONSTRAIN_CPW && c_mine < g.ocols && full_search[(size_t)r * g.ocols + c_mine] != 0;
  unsigned long long todo = __ballot(mine);
  const int r0 = max(r - range, 0), r1 = min(r + range, g.orows - 1);
  while (todo) {                                       // wave-uniform
    const int j = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int c = c_base + j;
    const int c0 = max(c - range, 0), c1 = min(c + range, g.ocols - 1);
    const int ww = c1 - c0 + 1, n = ww * (r1 - r0 + 1);
    const float inv = __builtin_amdgcn_rcpf((float)ww);
    int x0 = 0x7ffffffe, y0 = 0x7ffffffe, x1 = -0x7ffffffe, y1 = -0x7ffffffe;
    for (int i = lane; i < n; i += 64) {
      int q = (int)(((float)i + 0.5f) * inv);            // i / ww for i < 2^16 (the window has at most 51 x 51 cells)
      int rem = i - q * ww;
      if (rem < 0) { rem += ww; --q; } else if (rem >= ww) { rem -= ww; ++q; }
      const size_t idx = (size_t)(r0 + q) * g.ocols + (c0 + rem);
      const uint8_t fs = full_search[idx];                // (both requested before either is looked at)
      const B4 v = bounds[idx];
      if (fs) continue;
      if (v.x0 == 0 && v.y0 == 0 && v.x1 == -1 && v.y1 == -1) continue;
      x0 = min(x0, min(v.x0, v.x1)); x1 = max(x1, max(v.x0, v.x1));       // grow(min corner); grow(max corner)
      y0 = min(y0, min(v.y0, v.y1)); y1 = max(y1, max(v.y0, v.y1));
    }
    for (int s = 32; s > 0; s >>= 1) {
      x0 = min(x0, __shfl_xor(x0, s)); y0 = min(y0, __shfl_xor(y0, s));
      x1 = max(x1, __shfl_xor(x1, s)); y1 = max(y1, __shfl_xor(y1, s));
    }
    if (lane == 0) {
      const size_t idx = (size_t)r * g.ocols + c;
      if (x0 >= x1 || y0 >= y1) {                       // empty(): no estimate
        if (conserve > 0) bounds[idx] = B4{0, 0, -1, -1};
      } else {
        x0 -= 2; y0 -= 2; x1 += 2; y1 += 2;             // expand(NEARBY_DISP_EXPANSION)
        x0 = max(x0, g.min_dx); y0 = max(y0, g.min_dy); x1 = min(x1, g.max_dx); y1 = min(y1, g.max_dy);
        bounds[idx] = B4{x0, y0, x1, y1};
      }
    }
  }
}

// ---- ragged starts ----------------------------------------------------------------------------------------------------

// rowsum[r] = cells of row r; rowsum[gridDim.x + r] = its pixels with boxes of at most 16 cells (low word) and of at most 32 (high word);
// rowsum[2 gridDim.x + r] = the cells of the former: what the lines-per-wavefront of the ragged path kernel is chosen by
__global__ void row_count_kernel(const B4* __restrict__ bounds, int ocols, unsigned long long* __restrict__ rowsum) {
  const int r = blockIdx.x;
  unsigned long long s = 0, small = 0, cells16 = 0;
  for (int c = threadIdx.x; c < ocols; c += blockDim.x) {
    const int n = b4_count(bounds[(size_t)r * ocols + c]);
    s += (unsigned long long)n;
    small += (n <= 16 ? 1ull : 0ull) + (n <= 32 ? 1ull << 32 : 0ull);
    cells16 += n <= 16 ? (unsigned long long)n : 0ull;
  }
  __shared__ unsigned long long sh[256], shm[256], shc[256];
  sh[threadIdx.x] = s; shm[threadIdx.x] = small; shc[threadIdx.x] = cells16;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { sh[threadIdx.x] += sh[threadIdx.x + o]; shm[threadIdx.x] += shm[threadIdx.x + o]; shc[threadIdx.x] += shc[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { rowsum[r] = sh[0]; rowsum[gridDim.x + r] = shm[0]; rowsum[2 * gridDim.x + r] = shc[0]; }
}

__global__ void row_scan_kernel(const B4* __restrict__ bounds, int ocols, const unsigned long long* __restrict__ rowoff,
                                unsigned long long* __restrict__ starts) {
  const int r = blockIdx.x;
  __shared__ unsigned long long sh[256];
  __shared__ unsigned long long carry;
  if (threadIdx.x == 0) carry = rowoff[r];
  __syncthreads();
  for (int base = 0; base < ocols; base += 256) {
    const int c = base + threadIdx.x;
    const unsigned long long v = c < ocols ? (unsigned long long)b4_count(bounds[(size_t)r * ocols + c]) : 0ull;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {               // Hillis-Steele inclusive scan
      unsigned long long t = (int)threadIdx.x >= o ? sh[threadIdx.x - o] : 0ull;
      __syncthreads();
      sh[threadIdx.x] += t;
      __syncthreads();
    }
    if (c < ocols) starts[(size_t)r * ocols + c] = carry + sh[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 255) carry += sh[255];
    __syncthreads();
  }
}

// ---- cost fill --------------------------------------------------------------------------------------------------------

// One wavefront per COST_PPW consecutive output pixels, lanes over a pixel's disparities (get_hamming_distance_costs, SGM.cc:40-75).
// A pixel is three dependent memory round trips (record -> left census word / vector start -> right census words) and a handful
// of instructions: the records of all COST_PPW pixels are requested together, then their first 64 right words, before the first
// popcount (one pixel per wave was pure latency: 0.7 ms per 1024^2 SGM tile).
constexpr int COST_PPW = 4;
__global__ void __launch_bounds__(256)
cost_kernel(const uint64_t* __restrict__ lc, int lcw, const uint64_t* __restrict__ rc, int rcw,
            const B4* __restrict__ bounds, const unsigned long long* __restrict__ starts, int ocols, size_t npix,
            int off_c, int off_r, uint8_t* __restrict__ cost) {
  const size_t p0 = ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * COST_PPW;
  if (p0 >= npix) return;
  const int lane = threadIdx.x & 63;
  B4 bq[COST_PPW];
  unsigned long long sq[COST_PPW];
  int rr[COST_PPW], cc[COST_PPW];
#pragma unroll
  for (int q = 0; q < COST_PPW; ++q) {
    const size_t p = p0 + q < npix ? p0 + q : npix - 1;
    bq[q] = bounds[p]; sq[q] = starts[p];
    rr[q] = (int)(p / ocols); cc[q] = (int)(p - (size_t)rr[q] * ocols);
  }
  uint64_t lv[COST_PPW], r0[COST_PPW];
  int wd[COST_PPW], n[COST_PPW];
#pragma unroll
  for (int q = 0; q < COST_PPW; ++q) {
    const int bc = cc[q] + off_c, br = rr[q] + off_r;       // (min_col - half_kernel, min_row - half_kernel) offsets
    wd[q] = bq[q].x1 - bq[q].x0 + 1; n[q] = wd[q] * (bq[q].y1 - bq[q].y0 + 1);
    lv[q] = lc[(size_t)br * lcw + bc];
    const int i = min(lane, max(n[q] - 1, 0));
    const int qy = n[q] > 0 ? i / wd[q] : 0, qx = n[q] > 0 ? i - qy * wd[q] : 0;
    r0[q] = n[q] > 0 ? rc[(size_t)(br + bq[q].y0 + qy) * rcw + bc + bq[q].x0 + qx] : 0ull;
  }
#pragma unroll
  for (int q = 0; q < COST_PPW; ++q) {
    if (p0 + q >= npix || n[q] <= 0) continue;              // wave-uniform
    uint8_t* o = cost + sq[q];
    if (lane < n[q]) o[lane] = (uint8_t)__popcll(lv[q] ^ r0[q]);
    const int bc = cc[q] + off_c, br = rr[q] + off_r;
    for (int i = lane + 64; i < n[q]; i += 64) {
      const int qy = i / wd[q], qx = i - qy * wd[q];
      o[i] = (uint8_t)__popcll(lv[q] ^ rc[(size_t)(br + bq[q].y0 + qy) * rcw + bc + bq[q].x0 + qx]);
    }
  }
}

// Uniform layout (every pixel searches the full range, vectors `stride` apart, stride a multiple of 16): one thread per 16
// consecutive disparities — 16 census words of the right image in flight, one 16-byte store (4 disparities and a dword store per
// thread ran at 0.7 TB/s: 0.79 ms for the 541 MB of a 2048^2 x 129 volume).
// ONE_ROW: num_dy == 1, the disparity index is the column offset.
template <bool ONE_ROW>
__global__ void cost_uniform16_kernel(const uint64_t* __restrict__ lc, int lcw, const uint64_t* __restrict__ rc, int rcw,
                                      int ocols, int orows, int num_dx, int num_disp, int stride, int off_c, int off_r,
                                      uint32_t* __restrict__ cost32) {
  const int q = stride / 16;
  // thread <-> (pixel column c = blockIdx.x * 16 + threadIdx.y, 16-disparity group w = threadIdx.x): no division at all
  const int w = threadIdx.x, c = blockIdx.x * blockDim.y + threadIdx.y, r = blockIdx.y;
  if (w >= q || c >= ocols) return;
  const int bc = c + off_c, br = r + off_r;
  const uint64_t lv = lc[(size_t)br * lcw + bc];
  uint64_t rv[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int i = 16 * w + e;
    const int ii = i < num_disp ? i : 0;                  // dead slots read disparity 0 and are zeroed below
    const int qy = ONE_ROW ? 0 : ii / num_dx, qx = ii - qy * num_dx;
    rv[e] = rc[(size_t)(br + qy) * rcw + bc + qx];
  }
  uint32_t v[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int e = 0; e < 16; ++e)
    if (16 * w + e < num_disp) v[e >> 2] |= (uint32_t)__popcll(lv ^ rv[e]) << (8 * (e & 3));
  reinterpret_cast<uint4*>(cost32)[((size_t)r * ocols + c) * q + w] = make_uint4(v[0], v[1], v[2], v[3]);
}

// One search row (num_dy == 1), uniform layout: a workgroup owns 256 consecutive pixels of a row, stages the 256 + D - 1 census
// words of the right image they can reach in LDS once (6 KB at most) and every lane walks its pixel's disparities through
// conflict-free 8-byte LDS reads (lane c reads word c + d).  The thread-per-16-disparities kernel above asked the L1 for 9-18
// cache lines per wave load and ran at 0.7 TB/s (0.88 ms for the 606 MB of a 2048^2 x 129 volume).
// Round 6: the 256 vectors of a workgroup are one contiguous run of the volume (256 x stride bytes), but a lane storing 8 bytes of ITS
// vector put 64 addresses `stride` apart into every store instruction (1.9 GB of write requests for a 0.57 GB volume, 0.45 ms at
// 2048^2 x 129).  The costs now go to a tile in LDS first and leave it as consecutive 16-byte pieces: every store instruction of a
// wavefront writes 1 KiB of consecutive bytes.
__global__ void __launch_bounds__(256)
cost_row_kernel(const uint64_t* __restrict__ lc, int lcw, const uint64_t* __restrict__ rc, int rcw, int ocols, int num_disp, int stride,
                int off_c, int off_r, uint8_t* __restrict__ cost) {
  extern __shared__ uint64_t words[];                    // 256 + num_disp - 1 census words, then the tile of 256 x stride cost bytes
  const int tid = threadIdx.x, c0 = blockIdx.x * 256, r = blockIdx.y;
  const int br = r + off_r, bc0 = c0 + off_c;
  const int nwords = (256 + num_disp - 1 + 1) & ~1;      // the tile starts 16-byte aligned
  uint2* const tile = reinterpret_cast<uint2*>(words + nwords);
  const uint64_t* rrow = rc + (size_t)br * rcw;
  for (int i = tid; i < 256 + num_disp - 1; i += 256) words[i] = rrow[min(bc0 + i, rcw - 1)];
  const uint64_t lv = lc[(size_t)br * lcw + min(bc0 + tid, lcw - 1)];
  __syncthreads();
  const int q = stride / 8;                              // stride % 8 == 0: 8 disparities per 8-byte piece
  for (int w = 0; w < q; ++w) {
    unsigned v[2] = {0u, 0u};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int i = 8 * w + e;
      const uint64_t rv = words[tid + min(i, num_disp - 1)];
      const unsigned cst = i < num_disp ? (unsigned)__popcll(lv ^ rv) : 0u;      // dead slots of the stride are zero
      v[e >> 2] |= cst << (8 * (e & 3));
    }
    tile[tid * q + w] = make_uint2(v[0], v[1]);
  }
  __syncthreads();
  const int npx = min(256, ocols - c0);                  // pixels of this workgroup inside the row
  const int n16 = npx * stride / 16, rem8 = (npx * stride) & 8;      // 16-byte pieces (+ one 8-byte piece when npx * stride % 16 == 8)
  uint8_t* const o = cost + ((size_t)r * ocols + c0) * stride;       // 8-byte aligned; 16-byte aligned when the pixel index is even
  const bool a16 = (reinterpret_cast<uintptr_t>(o) & 15) == 0;
  if (a16) {
    const uint4* t4 = reinterpret_cast<const uint4*>(tile);
    uint4* o4 = reinterpret_cast<uint4*>(o);
    for (int i = tid; i < n16; i += 256) o4[i] = t4[i];
    if (rem8 && tid == 0) reinterpret_cast<uint2*>(o)[2 * n16] = tile[2 * n16];
  } else {                                               // (odd first pixel and stride % 16 == 8)
    uint2* o2 = reinterpret_cast<uint2*>(o);
    for (int i = tid; i < npx * q; i += 256) o2[i] = tile[i];
  }
}

// ---- the mean-abs-difference block cost (opt-in) ------------------------------------------------------------------------
// get_cost_block / fill_costs_block (src/vw/Stereo/SGM.cc:1651-1738, "Mean of abs differences" branch): cost = min(255,
// (sum over the k x k window of |L - R|) / (k * k)) on the u8 images, integer division.  The reference keeps this path behind a
// NoImplErr (:1887-1892); vwgpu_sgm_params.allow_block_cost opts in (BASELINE configs[3] names a SAD cost into SGM).
// Exact division of n < 2^16 by the window's pixel count: __umulhi(n, 2^32 / c + 1) (checked exhaustively by the launcher).
//
// General form: one wavefront per pixel, lane = disparity of the pixel's box (any bounds: masks, previous level, 2-D searches).
__global__ void __launch_bounds__(256)
cost_block_kernel(const uint8_t* __restrict__ l8, int lw, const uint8_t* __restrict__ r8, int rw, const B4* __restrict__ bounds,
                  const unsigned long long* __restrict__ starts, int ocols, size_t npix, int min_col, int min_row, int kernel, unsigned magic,
                  uint8_t* __restrict__ cost) {
  const size_t p = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (p >= npix) return;
  const int lane = threadIdx.x & 63;
  const B4 b = bounds[p];
  const int wd = b.x1 - b.x0 + 1, n = wd * (b.y1 - b.y0 + 1);
  if (n <= 0) return;
  const int r = (int)(p / ocols), c = (int)(p - (size_t)r * ocols);
  const int hk = (kernel - 1) / 2;
  const uint8_t* lp = l8 + (size_t)(r + min_row - hk) * lw + (c + min_col - hk);
  uint8_t* o = cost + starts[p];
  for (int i = lane; i < n; i += 64) {
    const int qy = i / wd, qx = i - qy * wd;
    const uint8_t* rp = r8 + (size_t)(r + min_row - hk + b.y0 + qy) * rw + (c + min_col - hk + b.x0 + qx);
    unsigned sum = 0;
    for (int j = 0; j < kernel; ++j)
      for (int k = 0; k < kernel; ++k) sum += (unsigned)abs((int)lp[(size_t)j * lw + k] - (int)rp[(size_t)j * rw + k]);
    const unsigned q = kernel == 1 ? sum : __umulhi(sum, magic);
    o[i] = (uint8_t)min(q, 255u);
  }
}

__host__ __device__ inline int block_row_dwords(int num_disp, int nw) { const int n = (256 + num_disp + 4 * nw + 8) / 4 + 1; return (n + 15) / 32 * 32 + 16; }

// Uniform layout, one search row (config 4's single-level strips): a workgroup owns 256 consecutive pixels of a row.  The K right
// rows its windows can reach are staged in LDS as dwords at EVERY byte offset (four pre-shifted copies), the lane's K x K left
// window lives in registers as NW = ceil(K / 4) dwords per row with the bytes beyond K cleared, and four consecutive disparities
// are one chain of v_qsad_pk_u16_u8 down the rows: slot i = sum over (row, word) of SAD4(R[x + d + i + 4n ..], L[4n ..]).
// The cleared left bytes pick up |0 - R| = R at the window positions K .. 4 NW - 1: that excess is the column sum of the right rows
// there — formed once per workgroup — and is subtracted; then the exact division and one 8-byte store per 8 disparities.
template <int K>
__global__ void __launch_bounds__(256)
cost_block_row_kernel(const uint8_t* __restrict__ l8, int lw, int lh, const uint8_t* __restrict__ r8, int rw, int rh, int ocols, int num_disp, int stride,
                      int min_col, int min_row, unsigned magic, uint8_t* __restrict__ cost) {
  constexpr int NW = (K + 3) / 4, HK = (K - 1) / 2, PAD = 4 * NW - K;
  extern __shared__ unsigned lds_u32[];
  const int tid = threadIdx.x, c0 = blockIdx.x * 256, r = blockIdx.y;
  const int rlw = block_row_dwords(num_disp, NW);                       // dwords per staged row (% 32 == 16: the four copies a wave reads fall on disjoint banks)
  unsigned* shifted = lds_u32;                                           // [K][4][rlw]: dword i of copy t = bytes 4 i + t .. 4 i + t + 3 of the row
  unsigned* colsum = shifted + K * 4 * rlw;                              // [4 * rlw]: sum of the K rows' bytes at each column
  unsigned* base = colsum + 4 * rlw;                                     // [K][rlw + 1]: the rows as aligned dwords
  const int x0 = c0 + min_col - HK;                                      // image column of byte 0 of the staged rows (may be < 0 never: min_col >= HK)
  for (int j = 0; j < K; ++j) {
    const uint8_t* row = r8 + (size_t)min(r + min_row - HK + j, rh - 1) * rw;
    for (int i = tid; i < rlw + 1; i += 256) {
      unsigned w = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) w |= (unsigned)row[min(x0 + 4 * i + b, rw - 1)] << (8 * b);
      base[j * (rlw + 1) + i] = w;
    }
  }
  __syncthreads();
  for (int i = tid; i < K * rlw; i += 256) {
    const int j = i / rlw, w = i - j * rlw;
    const unsigned lo = base[j * (rlw + 1) + w], hi = base[j * (rlw + 1) + w + 1];
    shifted[(j * 4 + 0) * rlw + w] = lo;
    shifted[(j * 4 + 1) * rlw + w] = __builtin_amdgcn_alignbyte(hi, lo, 1);
    shifted[(j * 4 + 2) * rlw + w] = __builtin_amdgcn_alignbyte(hi, lo, 2);
    shifted[(j * 4 + 3) * rlw + w] = __builtin_amdgcn_alignbyte(hi, lo, 3);
  }
  for (int i = tid; i < 4 * rlw; i += 256) {
    unsigned sum = 0;
#pragma unroll
    for (int j = 0; j < K; ++j) sum += (base[j * (rlw + 1) + (i >> 2)] >> (8 * (i & 3))) & 0xffu;
    colsum[i] = sum;
  }
  // the lane's left window
  const int c = c0 + tid;
  unsigned lwin[K][NW];
  {
    const int lx = min(c, ocols - 1) + min_col - HK;
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const uint8_t* row = l8 + (size_t)min(r + min_row - HK + j, lh - 1) * lw;
#pragma unroll
      for (int n = 0; n < NW; ++n) {
        unsigned w = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (4 * n + b < K) w |= (unsigned)row[min(lx + 4 * n + b, lw - 1)] << (8 * b);
        lwin[j][n] = w;
      }
    }
  }
  __syncthreads();
  if (c >= ocols) return;
  const int t = tid & 3, w0 = tid >> 2;
  uint2* o = reinterpret_cast<uint2*>(cost + ((size_t)r * ocols + c) * stride);
  for (int d8 = 0; d8 < stride; d8 += 8) {
    unsigned out[2] = {0u, 0u};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int d = d8 + 4 * h;                                          // disparities d .. d + 3: right bytes from tid + d
      if (d >= num_disp) continue;                                       // dead slots of the stride stay zero
      unsigned long long acc = 0ull;
#pragma unroll
      for (int j = 0; j < K; ++j) {
        const unsigned* sr = shifted + (j * 4 + t) * rlw + w0 + (d >> 2);
        unsigned w[NW + 1];
#pragma unroll
        for (int n = 0; n <= NW; ++n) w[n] = sr[n];
#pragma unroll
        for (int n = 0; n < NW; ++n)
          acc = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)w[n + 1] << 32) | w[n], lwin[j][n], acc);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned sum = (unsigned)(acc >> (16 * i)) & 0xffffu;
#pragma unroll
        for (int q = 0; q < PAD; ++q) sum -= colsum[tid + d + i + K + q];
        const unsigned v = d + i < num_disp ? min(__umulhi(sum, magic), 255u) : 0u;
        out[h] |= v << (8 * i);
      }
    }
    o[d8 >> 3] = make_uint2(out[0], out[1]);
  }
}

// ---- path aggregation -------------------------------------------------------------------------------------------------

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait for the global
// loads that were just issued as a prefetch for the NEXT step of the scan-line recurrence.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// for LDS data a wavefront exchanges with itself only: its own LDS operations complete in order, the compiler must not move them
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ unsigned adds16(unsigned a, unsigned b) { return min(a + b, 65535u); }
__device__ __forceinline__ unsigned subs16(unsigned a, unsigned b) { return max(a, b) - b; }

// Minimum over the 64 lanes of a wavefront with DPP moves only (no LDS traffic): butterflies inside each row of 16 lanes,
// then the gfx9 row broadcasts; every lane of the result holds the minimum after the final readlane.
__device__ __forceinline__ unsigned wave_min_u32(unsigned v) {
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));   // row_half_mirror
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));   // row_mirror
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x142, 0xA, 0xF, false));   // row_bcast:15 -> rows 1, 3
  v = min(v, (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x143, 0xC, 0xF, false));   // row_bcast:31 -> rows 2, 3
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// The scan lines of up to 8 directions in one launch.  Direction q owns blocks [line0[q], line0[q+1]); its lines start on
// the first border (x index, or y index for the purely horizontal passes) and then on the second one, exactly as
// accum_sgm_multithread enumerates them (SGM.cc:2488-2610).  Different directions add into the same accumulator elements,
// so the adds are 32-bit atomics on the u16 pair that holds the element (no carry can cross the halves: the caller only
// groups directions when 8 * (255 + max(P1, P2)) < 65536; otherwise it launches them one by one).
struct DirSet {
  int n, rev_second;
  int dc[8], dr[8], n_first[8], row_border[8], second_skip[8], line0[9];
};
__device__ __forceinline__ void line_start(const DirSet& D, const SgmGeom& g, int block, int& dc, int& dr, int& c, int& r) {
  int q = 0;
  while (q + 1 < D.n && block >= D.line0[q + 1]) ++q;
  dc = D.dc[q]; dr = D.dr[q];
  const int line = block - D.line0[q];
  if (line < D.n_first[q]) {
    if (D.row_border[q]) { c = line; r = dr > 0 ? 0 : g.orows - 1; }
    else { r = line; c = dc > 0 ? 0 : g.ocols - 1; }
  } else {
    r = line - D.n_first[q] + D.second_skip[q];
    c = dc > 0 ? 0 : g.ocols - 1;
  }
}
// Wave minimum with the DPP source fused into v_min_u32 (the compiler emits v_mov_dpp + v_min for the builtin form); a VALU
// result needs two wait states before a DPP read, hence the s_nop 1 in front of every step of the dependent chain.
__device__ __forceinline__ unsigned wave_min_u32_fused(unsigned v) {
  asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
               "s_nop 1"
               : "+v"(v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned wave_shr1(unsigned v, unsigned edge) {      // lane l <- lane l-1, lane 0 <- edge
  return (unsigned)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x138, 0xF, 0xF, false);
}
__device__ __forceinline__ unsigned wave_shl1(unsigned v, unsigned edge) {      // lane l <- lane l+1, lane 63 <- edge
  return (unsigned)__builtin_amdgcn_update_dpp((int)edge, (int)v, 0x130, 0xF, 0xF, false);
}

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ us2 as_us2(unsigned v) { return __builtin_bit_cast(us2, v); }
__device__ __forceinline__ unsigned as_u32(us2 v) { return __builtin_bit_cast(unsigned, v); }
// `plain`: the launch holds ONE direction (DirSet::n == 1 — the caller separates the directions when 8 * (255 + max(P1, P2)) could
// overflow 16 bits), so every pixel lies on exactly one line and the sum is a plain u16 read-modify-write that wraps around like the
// reference's `+=`; the packed 32-bit atomics would carry an overflowing low half into its neighbour.
__device__ __forceinline__ void accum_add_u16(uint16_t* accum, unsigned long long e, unsigned v, bool plain = false) {
  if (plain) { accum[e] = (uint16_t)(accum[e] + v); return; }
  atomicAdd(reinterpret_cast<unsigned*>(accum) + (e >> 1), v << ((e & 1) * 16));
}

// i / d and i % d for 0 <= i < 2^16, 1 <= d < 2^16 through one reciprocal (an integer division is ~25 VALU instructions
// and the recurrence below needs three per pixel).
__device__ __forceinline__ void divmod_f(int i, int d, float inv, int& quo, int& rem) {
  int q = (int)(((float)i + 0.5f) * inv);
  int r = i - q * d;
  if (r < 0) { --q; r += d; }
  if (r >= d) { ++q; r -= d; }
  quo = q; rem = r;
}

// One wavefront per scan line (PixelPassTask, SGMAssist.h:705-819).  line -> start pixel as in accum_sgm_multithread
// (SGM.cc:2488-2610): first `n_first` lines start on the first border (index i), the rest on the second (index i + skip).
__global__ void __launch_bounds__(64)
path_kernel(SgmGeom g, DirSet D,
            const uint8_t* __restrict__ left, int lw, int min_col, int min_row,
            const B4* __restrict__ bounds, const unsigned long long* __restrict__ starts,
            const uint8_t* __restrict__ cost, uint16_t* __restrict__ accum, unsigned p1, unsigned p2) {
  extern __shared__ uint16_t sm[];
  const int num_disp = g.num_dx * g.num_dy;
  uint16_t* full_prior = sm;                 // num_disp
  uint16_t* prev_out = sm + num_disp;        // num_disp (packed vector of the previous pixel)
  const int lane = threadIdx.x;
  int c, r, dc, dr;
  line_start(D, g, blockIdx.x, dc, dr, c, r);
  const unsigned BAD = (255u + p2) & 0xffffu;
  for (int i = lane; i < num_disp; i += 64) full_prior[i] = (uint16_t)BAD;
  lds_barrier();
  int last_val = -1;
  B4 bp{0, 0, -1, -1};
  // Software pipeline over the pixels of the line: the per-pixel records (box, vector start, grey value) of pixel k+1 are
  // requested while pixel k is processed, and its first two cost bytes per lane right after the scatter phase — the
  // dependent chain box/start -> cost -> recurrence would otherwise expose two memory round trips per pixel.
  auto inside = [&](int cc, int rr) { return cc >= 0 && rr >= 0 && cc < g.ocols && rr < g.orows; };
  B4 b{0, 0, -1, -1};
  unsigned long long st = 0;
  int cur = 0;
  unsigned cv0 = 0, cv1 = 0;                    // cost[st + lane], cost[st + lane + 64] of the current pixel
  if (inside(c, r)) {
    const size_t p = (size_t)r * g.ocols + c;
    b = bounds[p]; st = starts[p];
    cur = left[(size_t)(r + min_row) * lw + (c + min_col)];
    const int nd0 = (b.x1 - b.x0 + 1) * (b.y1 - b.y0 + 1);
    if (lane < nd0) cv0 = cost[st + lane];
    if (lane + 64 < nd0) cv1 = cost[st + lane + 64];
  }
  while (inside(c, r)) {
    const int wd = b.x1 - b.x0 + 1, nd = wd * (b.y1 - b.y0 + 1);
    // request the next pixel's records
    const int cn = c + dc, rn = r + dr;
    const bool has_next = inside(cn, rn);
    B4 b_n{0, 0, -1, -1};
    unsigned long long st_n = 0;
    int cur_n = 0;
    if (has_next) {
      const size_t pn = (size_t)rn * g.ocols + cn;
      b_n = bounds[pn]; st_n = starts[pn];
      cur_n = left[(size_t)(rn + min_row) * lw + (cn + min_col)];
    }
    unsigned cn0 = 0, cn1 = 0;
    auto request_next_cost = [&]() {
      if (!has_next) return;
      const int ndn = (b_n.x1 - b_n.x0 + 1) * (b_n.y1 - b_n.y0 + 1);
      if (lane < ndn) cn0 = cost[st_n + lane];
      if (lane + 64 < ndn) cn1 = cost[st_n + lane + 64];
    };
    auto cost_at = [&](int i) -> unsigned { return i < 64 ? cv0 : (i < 128 ? cv1 : (unsigned)cost[st + i]); };
    if (last_val < 0) {
      request_next_cost();
      for (int i = lane; i < nd; i += 64) {
        const unsigned v = cost_at(i);
        prev_out[i] = (uint16_t)v;
        accum_add_u16(accum, st + i, v, D.n == 1);
      }
    } else {
      int grad = cur - last_val; grad = grad < 0 ? -grad : grad;
      unsigned p2_mod = p2;
      if (grad > 0) p2_mod /= (unsigned)grad;
      if (p2_mod < p1) p2_mod = p1;
      // scatter the prior vector into the full-range buffer, reduce its minimum
      const int wp = bp.x1 - bp.x0 + 1, np = wp * (bp.y1 - bp.y0 + 1);
      const float inv_wp = __builtin_amdgcn_rcpf((float)wp), inv_wd = __builtin_amdgcn_rcpf((float)wd);
      unsigned mn = BAD;
      for (int i = lane; i < np; i += 64) {
        int qy, qx;
        divmod_f(i, wp, inv_wp, qy, qx);
        const unsigned v = prev_out[i];
        mn = min(mn, v);
        full_prior[(bp.y0 + qy - g.min_dy) * g.num_dx + (bp.x0 + qx - g.min_dx)] = (uint16_t)v;
      }
      const unsigned min_prior = wave_min_u32(mn);
      const unsigned dJ = (min_prior + p2_mod) & 0xffffu;
      request_next_cost();
      lds_barrier();
      for (int i = lane; i < nd; i += 64) {
        int qy, qx;
        divmod_f(i, wd, inv_wd, qy, qx);
        const int dx = b.x0 + qx, dy = b.y0 + qy;
        const int xo = dx - g.min_dx, yo = dy - g.min_dy;
        const int xl = dx - 1 < g.min_dx ? xo : xo - 1, xm = dx + 1 > g.max_dx ? xo : xo + 1;
        const int yl = (dy - 1 < g.min_dy ? yo : yo - 1) * g.num_dx, ym = (dy + 1 > g.max_dy ? yo : yo + 1) * g.num_dx, yc = yo * g.num_dx;
        unsigned m = full_prior[yl + xo];
        m = min(m, (unsigned)full_prior[yc + xl]); m = min(m, (unsigned)full_prior[yc + xm]); m = min(m, (unsigned)full_prior[ym + xo]);
        m = min(m, (unsigned)full_prior[yl + xl]); m = min(m, (unsigned)full_prior[yl + xm]);
        m = min(m, (unsigned)full_prior[ym + xl]); m = min(m, (unsigned)full_prior[ym + xm]);
        unsigned res = adds16(m, p1);
        res = min(res, min((unsigned)full_prior[yc + xo], dJ));
        res = adds16(res, cost_at(i));
        res = subs16(res, min_prior);
        // prev_out is still being read by nobody (scatter finished at the barrier): reuse it for this pixel's vector
        prev_out[i] = (uint16_t)res;
        accum_add_u16(accum, st + i, res, D.n == 1);
      }
      lds_barrier();
      for (int i = lane; i < np; i += 64) {
        int qy, qx;
        divmod_f(i, wp, inv_wp, qy, qx);
        full_prior[(bp.y0 + qy - g.min_dy) * g.num_dx + (bp.x0 + qx - g.min_dx)] = (uint16_t)BAD;
      }
    }
    lds_barrier();
    bp = b; last_val = cur;
    b = b_n; st = st_n; cur = cur_n; cv0 = cn0; cv1 = cn1;
    c = cn; r = rn;
  }
}

// ---- MGM (accum_mgm_multithread, SGM.cc:2619-2700; SmoothPathAccumTask, SGMAssist.h:835-1239) ---------------------------------
// Every pixel of a pass takes the mean of TWO evaluate_path results — from its path predecessor A and from a second, "perpendicular"
// predecessor B — so a pass is no longer a set of independent scan lines but a 2-D recurrence.  Its parallel structure is the FRONT:
//   L  (A = left,  B = above)       depends on the previous anti-diagonal  c + r              -> W + H - 1 fronts of <= min(W, H) pixels
//   R, T, B likewise on the mirrored anti-diagonals
//   TL (A = (c-1, r-1), B = (c+1, r-1)) depends on the previous ROW       -> H fronts of W pixels;      BR: rows upwards
//   BL (A = (c-1, r+1), B = (c-1, r-1)) depends on the previous COLUMN    -> W fronts of H pixels;      TR: columns leftwards
// One launch per front, one wavefront per pixel of the front and direction (blockIdx.y): the launch boundary is the only
// synchronisation, so nothing here can spin.  A direction's values live in a volume of their own (`vols`, laid out like the
// sums), because both predecessors are read from it; mgm_sum_kernel adds the volumes to the sums afterwards with the u16
// wrap-around of the reference's `+=` (add_lead_buffer_to_accum, SGMAssist.h:357-388, does that line by line).
// Both evaluations of a pixel use ONE intensity difference, get_path_pixel_diff (SGM.cc:2715-2721) = |I(c, r) - I(c - ax, r - ay)|:
// with (ax, ay) pointing at the path predecessor that is the pixel on the FAR side of the path.
struct MgmDirs {
  int n;
  int ax[8], ay[8], bx[8], by[8];
  int need[8];                    // the task's border test: bit 0 col > 0, bit 1 col < last, bit 2 row > 0, bit 3 row < last
  int kind[8];                    // 0 anti-diagonal fronts, 1 row fronts, 2 column fronts
  int flipx[8], flipy[8];         // fronts counted from the right / from the bottom
};
// pixel of (front, index in the front) for direction q; false when the front has no such pixel (mgm_schedule.h)
__device__ __forceinline__ bool mgm_front_pixel(const MgmDirs& D, int q, int front, int i, int W, int H, int& c, int& r) {
  return vwgpu::mgm_front_pixel(D.kind[q], D.flipx[q], D.flipy[q], front, i, W, H, c, r);
}

__global__ void __launch_bounds__(256)
mgm_front_kernel(SgmGeom g, MgmDirs D, int front, const uint8_t* __restrict__ left, int lw, int lh, int min_col, int min_row,
                 const B4* __restrict__ bounds, const unsigned long long* __restrict__ starts, const uint8_t* __restrict__ cost,
                 uint16_t* __restrict__ vols, size_t vol_elems, unsigned p1, unsigned p2) {
  extern __shared__ uint16_t sm[];
  const int num_disp = g.num_dx * g.num_dy;
  // One pixel per WAVEFRONT; a workgroup holds one or four independent wavefronts (the dispatcher starts workgroups, and a front of a
  // 1100-pixel tile in 8 directions is 8800 of them).  A wave only ever reads the LDS cells it wrote itself, so the phases are
  // separated by s_waitcnt alone (wave_lds_sync) — no workgroup barrier, and a wave may return early.
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), wpw = (int)blockDim.x >> 6;
  // the two predecessors' vectors over the whole search range, BAD_VAL elsewhere (evaluate_path's full_prior_buffer, SGM.cc:1013-1150)
  // Round 6: ONE 32-bit cell per disparity — low half = predecessor A's value, high half = predecessor B's — so that the nine reads of an
  // evaluation serve both predecessors and the minima are packed 16-bit instructions (18 u16 reads and two scalar chains before).
  unsigned* fp = reinterpret_cast<unsigned*>(sm) + (size_t)wave * num_disp;
  uint16_t* fp_a = reinterpret_cast<uint16_t*>(fp);              // cell i of A at fp_a[2 i], of B at fp_a[2 i + 1]
  uint16_t* fp_b = fp_a + 1;
  const int lane = threadIdx.x & 63, q = blockIdx.y, W = g.ocols, H = g.orows;
  int c, r;
  if (!mgm_front_pixel(D, q, front, (int)blockIdx.x * wpw + wave, W, H, c, r)) return;
  const int need = D.need[q];
  const bool ok = vwgpu::mgm_uses_predecessors(need, c, r, W, H);
  // every record of the step is addressed by the coordinates alone: one memory round trip for the three boxes, the three vector
  // starts and the two grey values, a second one for the first 128 elements of both predecessor vectors and of the costs
  const size_t p = (size_t)r * W + c;
  const size_t pa = ok ? (size_t)(r + D.ay[q]) * W + (c + D.ax[q]) : p, pb = ok ? (size_t)(r + D.by[q]) * W + (c + D.bx[q]) : p;
  const B4 b = bounds[p], ba = bounds[pa], bb = bounds[pb];
  const unsigned long long st = starts[p], sta = starts[pa], stb = starts[pb];
  // the reference indexes the image unchecked; the far-side pixel is inside it whenever the kernel is >= 3 (calc_disparity_sgm
  // always searches from 0).  Clamped like the oracle so that the read is defined for any geometry.
  const int fc = min(max(c - D.ax[q] + min_col, 0), lw - 1), fr = min(max(r - D.ay[q] + min_row, 0), lh - 1);
  int grad = (int)left[(size_t)(r + min_row) * lw + (c + min_col)] - (int)left[(size_t)fr * lw + fc];
  const int wd = b.x1 - b.x0 + 1, nd = wd * (b.y1 - b.y0 + 1);
  if (nd <= 0) return;                                         // get_num_disp() == 0: skipped (SGMAssist.h:904-909)
  uint16_t* vol = vols + (size_t)q * vol_elems;
  const unsigned c0 = lane < nd ? cost[st + lane] : 0u, c1 = lane + 64 < nd ? cost[st + lane + 64] : 0u;
  if (!ok) {                                                   // "Just init to the local cost"
    for (int i = lane; i < nd; i += 64) vol[st + i] = (uint16_t)(i < 64 ? c0 : i < 128 ? c1 : (unsigned)cost[st + i]);
    return;
  }
  const unsigned BAD = (255u + p2) & 0xffffu;
  const int wa = ba.x1 - ba.x0 + 1, na = wa * (ba.y1 - ba.y0 + 1);              // 0 for a skipped predecessor: an empty vector
  const int wb = bb.x1 - bb.x0 + 1, nb = wb * (bb.y1 - bb.y0 + 1);
  const uint16_t* pra = vol + sta;
  const uint16_t* prb = vol + stb;
  const unsigned a0 = lane < na ? pra[lane] : BAD, a1 = lane + 64 < na ? pra[lane + 64] : BAD;
  const unsigned b0 = lane < nb ? prb[lane] : BAD, b1 = lane + 64 < nb ? prb[lane + 64] : BAD;
  for (int i = lane; i < num_disp; i += 64) fp[i] = BAD | (BAD << 16);
  grad = grad < 0 ? -grad : grad;
  unsigned p2_mod = p2;
  if (grad > 0) p2_mod /= (unsigned)grad;
  if (p2_mod < p1) p2_mod = p1;
  wave_lds_sync();
  auto scatter = [&](uint16_t* fp, const B4& bp, int wp, int np, const uint16_t* prior, unsigned v0, unsigned v1) __attribute__((always_inline)) {
    const float inv_wp = __builtin_amdgcn_rcpf((float)max(wp, 1));
    unsigned mn = BAD;
    for (int i = lane; i < np; i += 64) {
      int qy, qx;
      divmod_f(i, wp, inv_wp, qy, qx);
      const unsigned v = i < 64 ? v0 : i < 128 ? v1 : (unsigned)prior[i];
      mn = min(mn, v);
      fp[2 * ((bp.y0 + qy - g.min_dy) * g.num_dx + (bp.x0 + qx - g.min_dx))] = (uint16_t)v;
    }
    return wave_min_u32(mn);
  };
  const unsigned min_a = scatter(fp_a, ba, wa, na, pra, a0, a1), min_b = scatter(fp_b, bb, wb, nb, prb, b0, b1);
  const unsigned dj_a = (min_a + p2_mod) & 0xffffu, dj_b = (min_b + p2_mod) & 0xffffu;
  const float inv_wd = __builtin_amdgcn_rcpf((float)wd);
  const unsigned p1c = min(p1, 65535u);                          // (adds16 saturates: a larger penalty is 65535)
  const us2 p1p1 = as_us2(p1c | (p1c << 16)), dJ2 = as_us2(dj_a | (dj_b << 16)), mp2 = as_us2((min_a & 0xffffu) | (min_b << 16));
  wave_lds_sync();
  for (int i = lane; i < nd; i += 64) {
    int qy, qx;
    divmod_f(i, wd, inv_wd, qy, qx);
    const int dx = b.x0 + qx, dy = b.y0 + qy;
    const int xo = dx - g.min_dx, yo = dy - g.min_dy;
    const int xl = dx - 1 < g.min_dx ? xo : xo - 1, xm = dx + 1 > g.max_dx ? xo : xo + 1;
    const int yl = (dy - 1 < g.min_dy ? yo : yo - 1) * g.num_dx, ym = (dy + 1 > g.max_dy ? yo : yo + 1) * g.num_dx, yc = yo * g.num_dx;
    const unsigned lc = i < 64 ? c0 : i < 128 ? c1 : (unsigned)cost[st + i];
    // both predecessors at once: packed saturating u16 arithmetic = adds16 / subs16 per half
    us2 m = as_us2(fp[yl + xo]);
    m = __builtin_elementwise_min(m, as_us2(fp[yc + xl])); m = __builtin_elementwise_min(m, as_us2(fp[yc + xm]));
    m = __builtin_elementwise_min(m, as_us2(fp[ym + xo])); m = __builtin_elementwise_min(m, as_us2(fp[yl + xl]));
    m = __builtin_elementwise_min(m, as_us2(fp[yl + xm])); m = __builtin_elementwise_min(m, as_us2(fp[ym + xl]));
    m = __builtin_elementwise_min(m, as_us2(fp[ym + xm]));
    us2 res = __builtin_elementwise_add_sat(m, p1p1);
    res = __builtin_elementwise_min(res, __builtin_elementwise_min(as_us2(fp[yc + xo]), dJ2));
    res = __builtin_elementwise_add_sat(res, as_us2(lc | (lc << 16)));
    res = __builtin_elementwise_sub_sat(res, mp2);
    const unsigned rr = as_u32(res);
    vol[st + i] = (uint16_t)(((rr & 0xffffu) + (rr >> 16)) >> 1);      // "(a + b) / 2" in int (SGMAssist.h:945-946)
  }
}

// The ragged recurrence for num_disp <= 64 * R, in place.  The full-range buffer always holds the PREVIOUS pixel's vector at
// its box cells and BAD_VAL elsewhere, so a step is: read the neighbours of every cell of the current box into registers,
// barrier, write the new values at the current box's cells, put BAD_VAL back into the cells of the previous box that the
// current one does not cover (usually none or a rim: neighbouring pixels descend from the same coarser pixel), barrier.
// No scatter / reset pass over the whole previous vector, no second copy of the vector, two barriers instead of three, and
// the minimum of the new vector is reduced from registers.
template <int R>
__global__ void __launch_bounds__(64)
path_inplace_kernel(SgmGeom g, DirSet D,
                    const uint8_t* __restrict__ left, int lw, int min_col, int min_row,
                    const B4* __restrict__ bounds, const unsigned long long* __restrict__ starts,
                    const uint8_t* __restrict__ cost, uint16_t* __restrict__ accum, unsigned p1, unsigned p2) {
  extern __shared__ uint16_t sm[];
  const int num_disp = g.num_dx * g.num_dy;
  uint16_t* full_prior = sm;                 // num_disp
  uint16_t* p2tab = sm + ((num_disp + 1) & ~1);   // 256: max(P1, P2 / |grey step|), the adaptive P2 of SGM.cc:813-818
  const int lane = threadIdx.x;
  int c, r, dc, dr;
  line_start(D, g, blockIdx.x, dc, dr, c, r);
  // wave-uniform values live in SGPRs: with 8 waves per SIMD the vector ALU is the contended unit, the scalar unit is not
  dc = __builtin_amdgcn_readfirstlane(dc); dr = __builtin_amdgcn_readfirstlane(dr);
  c = __builtin_amdgcn_readfirstlane(c); r = __builtin_amdgcn_readfirstlane(r);
  auto uni = [](B4 v) __attribute__((always_inline)) {
    return B4{__builtin_amdgcn_readfirstlane(v.x0), __builtin_amdgcn_readfirstlane(v.y0), __builtin_amdgcn_readfirstlane(v.x1),
              __builtin_amdgcn_readfirstlane(v.y1)};
  };
  auto uni64 = [](unsigned long long v) __attribute__((always_inline)) {
    return (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v) |
           ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32);
  };
  const unsigned BAD = (255u + p2) & 0xffffu;
  for (int i = lane; i < num_disp; i += 64) full_prior[i] = (uint16_t)BAD;
  for (int q = lane; q < 256; q += 64) {
    unsigned v = p2;
    if (q > 0) v /= (unsigned)q;
    if (v < p1) v = p1;
    p2tab[q] = (uint16_t)v;
  }
  lds_barrier();
  const long long delta = (long long)dr * g.ocols + dc;         // pixel index step along the line
  const long long ldelta = (long long)dr * lw + dc;
  auto inside = [&](int cc, int rr) { return cc >= 0 && rr >= 0 && cc < g.ocols && rr < g.orows; };
  int last_val = -1;
  unsigned min_prior = 0;
  B4 bp{0, 0, -1, -1};
  B4 b{0, 0, -1, -1};
  unsigned long long st = 0;
  int cur = 0;
  unsigned cv0 = 0, cv1 = 0;                    // cost[st + lane], cost[st + lane + 64] of the current pixel (prefetched)
  long long p = (long long)r * g.ocols + c;
  long long lp = (long long)(r + min_row) * lw + (c + min_col);
  // steps of the line inside the image (line_start() puts the first pixel on a border)
  int len = 0;
  if (inside(c, r)) {
    const int len_c = dc > 0 ? g.ocols - c : (dc < 0 ? c + 1 : 0x7fffffff);
    const int len_r = dr > 0 ? g.orows - r : (dr < 0 ? r + 1 : 0x7fffffff);
    len = min(len_c, len_r);
  }
  if (len > 0) {
    b = uni(bounds[p]); st = uni64(starts[p]);
    cur = __builtin_amdgcn_readfirstlane((int)left[lp]);
    const int nd0 = (b.x1 - b.x0 + 1) * (b.y1 - b.y0 + 1);
    cv0 = cost[st + max(min(lane, nd0 - 1), 0)];                    // unconditional, clamped (see the loop)
    cv1 = cost[st + max(min(lane + 64, nd0 - 1), 0)];
  }
  int wd_prev = -1, qy0 = 0, qx0 = 0;             // lane's cell coordinates inside a box of width wd_prev (chunk 0)
  for (int step = 0; step < len; ++step) {
    const int wd = b.x1 - b.x0 + 1, nd = wd * (b.y1 - b.y0 + 1);
    const bool has_next = step + 1 < len;
    B4 b_n{0, 0, -1, -1};
    unsigned long long st_n = 0;
    int cur_n = 0;
    // The next pixel's records, one memory round trip ahead.  No load of the loop sits under a condition (the last pixel of a
    // line re-reads its own records; lanes beyond a box read its last cost byte): a conditional load makes the compiler drain
    // all outstanding requests (s_waitcnt vmcnt(0)) right after issuing them, i.e. no prefetch at all.
    if (has_next) { p += delta; lp += ldelta; }   // wave-uniform
    b_n = bounds[p]; st_n = starts[p];
    cur_n = left[lp];
    const float inv_wd = __builtin_amdgcn_rcpf((float)wd);
    if (wd != wd_prev) {                          // wave-uniform: the box width changes every other pixel at most
      divmod_f(lane, max(wd, 1), inv_wd, qy0, qx0);
      wd_prev = wd;
    }
    unsigned res[R];
    int cell[R];
    int grad = cur - last_val; grad = grad < 0 ? -grad : grad;
    const unsigned dJ = (min_prior + (unsigned)p2tab[last_val >= 0 ? grad : 0]) & 0xffffu;
    // ---- phase 1: every read of the previous vector
    auto phase1 = [&](int k) __attribute__((always_inline)) {
      const int i = lane + 64 * k;
      if (i < nd) {
        int qy = qy0, qx = qx0;
        if (k > 0) divmod_f(i, wd, inv_wd, qy, qx);
        const int dx = b.x0 + qx, dy = b.y0 + qy;
        const int xo = dx - g.min_dx, yo = dy - g.min_dy;
        cell[k] = yo * g.num_dx + xo;
        const unsigned cb = k == 0 ? cv0 : (k == 1 ? cv1 : (unsigned)cost[st + i]);
        if (last_val < 0) res[k] = cb;
        else {
          const int xl = dx - 1 < g.min_dx ? xo : xo - 1, xm = dx + 1 > g.max_dx ? xo : xo + 1;
          const int yl = (dy - 1 < g.min_dy ? yo : yo - 1) * g.num_dx, ym = (dy + 1 > g.max_dy ? yo : yo + 1) * g.num_dx, yc = yo * g.num_dx;
          unsigned m = full_prior[yl + xo];
          m = min(m, (unsigned)full_prior[yc + xl]); m = min(m, (unsigned)full_prior[yc + xm]); m = min(m, (unsigned)full_prior[ym + xo]);
          m = min(m, (unsigned)full_prior[yl + xl]); m = min(m, (unsigned)full_prior[yl + xm]);
          m = min(m, (unsigned)full_prior[ym + xl]); m = min(m, (unsigned)full_prior[ym + xm]);
          unsigned v = adds16(m, p1);
          v = min(v, min((unsigned)full_prior[yc + xo], dJ));
          v = adds16(v, cb);
          res[k] = subs16(v, min_prior);
        }
      }
    };
#pragma unroll
    for (int k = 0; k < R; ++k) { res[k] = 0xffffu; cell[k] = -1; }
    phase1(0);                                    // boxes of up to 64 cells (nearly all): no chunk loop, no per-chunk branches
    if (nd > 64) {
#pragma unroll
      for (int k = 1; k < R; ++k)
        if (64 * k < nd) phase1(k);               // wave-uniform
    }
    // the next pixel's first cost bytes (its vector start has arrived by now)
    unsigned cn0, cn1;
    {
      const int ndn = (b_n.x1 - b_n.x0 + 1) * (b_n.y1 - b_n.y0 + 1);
      cn0 = cost[st_n + max(min(lane, ndn - 1), 0)];
      cn1 = cost[st_n + max(min(lane + 64, ndn - 1), 0)];
    }
    lds_barrier();
    // ---- phase 2: the new vector goes to its cells, cells only the previous box covered go back to BAD_VAL
    unsigned mn = BAD;                           // (the minimum of an empty vector is BAD_VAL, SGMAssist.h)
    auto phase2 = [&](int k) __attribute__((always_inline)) {
      const bool on = cell[k] >= 0;
      if (on) {
        full_prior[cell[k]] = (uint16_t)res[k];
        mn = min(mn, res[k]);
      }
      // The sums are u16 pairs in dwords and device-scope atomics are executed at the memory side, one lane-operation at a time:
      // the lane whose element is the low half of a dword adds its right neighbour's value with it (half the atomics).  Lane 0
      // always adds its own element (it may be a high half), lane 63 only its own (its neighbour is lane 0 of the next chunk).
      const unsigned mine = on ? res[k] : 0u;
      const unsigned next = wave_shl1(mine, 0u);
      const unsigned long long e = st + (unsigned)(lane + 64 * k);
      const bool low = (e & 1ull) == 0;
      if (D.n == 1) {                                   // one direction per launch: plain u16 sums (see accum_add_u16)
        if (on) accum[e] = (uint16_t)(accum[e] + mine);
      } else if (on && (low || lane == 0)) {
        const unsigned v = low ? (mine | (lane < 63 ? next << 16 : 0u)) : (mine << 16);
        atomicAdd(reinterpret_cast<unsigned*>(accum) + (e >> 1), v);
      }
    };
    phase2(0);
    if (nd > 64) {
#pragma unroll
      for (int k = 1; k < R; ++k)
        if (64 * k < nd) phase2(k);
    }
    if (last_val >= 0 && (bp.x0 < b.x0 || bp.x1 > b.x1 || bp.y0 < b.y0 || bp.y1 > b.y1)) {   // wave-uniform
      const int wp = bp.x1 - bp.x0 + 1, np = wp * (bp.y1 - bp.y0 + 1);
      const float inv_wp = __builtin_amdgcn_rcpf((float)wp);
      for (int i = lane; i < np; i += 64) {
        int qy, qx;
        divmod_f(i, wp, inv_wp, qy, qx);
        const int dx = bp.x0 + qx, dy = bp.y0 + qy;
        if (dx < b.x0 || dx > b.x1 || dy < b.y0 || dy > b.y1)
          full_prior[(dy - g.min_dy) * g.num_dx + (dx - g.min_dx)] = (uint16_t)BAD;
      }
    }
    min_prior = wave_min_u32_fused(mn);
    lds_barrier();
    bp = b; last_val = cur;
    b = uni(b_n); st = uni64(st_n); cur = __builtin_amdgcn_readfirstlane(cur_n); cv0 = cn0; cv1 = cn1;
  }
}

// Minimum over each group of SUB = 16 or 32 consecutive lanes, returned to every lane of the group.
template <int SUB>
__device__ __forceinline__ unsigned sub_min_u32_fused(unsigned v, int lane) {
  asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1"
               : "+v"(v));
  if constexpr (SUB == 32) {
    asm volatile("v_min_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1" : "+v"(v));
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)v, 31), hi = (unsigned)__builtin_amdgcn_readlane((int)v, 63);
    return lane >= 32 ? hi : lo;
  }
  return v;
}

// The ragged recurrence of path_inplace_kernel with 64 / SUB scan lines per wavefront, SUB = 16 or 32 lanes each (round 6).  The boxes of a
// pyramid level descend from the coarser level's disparities +- a margin: at level 0 of a 1024^2 tile with a 129 x 3 search 92 % of the
// pixels have 15 cells, at the levels above 5-12 — a quarter of a wavefront's lanes, and the kernel is bound by the vector instructions of
// a step.  A group of SUB lanes owns a line: what was wave-uniform (records of the pixel, vector start, grey step, minimum of the previous
// vector) is per lane and equal within the group, the full-range buffer exists once per group, the minimum is a group reduction, and a
// step ends when every group has finished its pixel.  A box of more than SUB cells (pixels without a trusted coarser disparity search the
// whole range: 1.5 % of the pixels, a quarter of the cells) is served by ALL lanes of the wavefront, one such line after the other, with
// the line's records broadcast and the new values staged in a vector of their own.
// Same arithmetic per line as path_inplace_kernel (evaluate_path, SGMAssist.h:705-819; accum_sgm_multithread, SGM.cc:2488-2610).
template <int SUB>
__global__ void __launch_bounds__(64)
path_multi_kernel(SgmGeom g, DirSet D, int lines, int spread,
                  const uint8_t* __restrict__ left, int lw, int min_col, int min_row,
                  const B4* __restrict__ bounds, const unsigned long long* __restrict__ starts,
                  const uint8_t* __restrict__ cost, uint16_t* __restrict__ accum, unsigned p1, unsigned p2) {
  constexpr int NSUB = 64 / SUB;
  extern __shared__ uint16_t sm[];
  const int num_disp = g.num_dx * g.num_dy;
  const int fpn = (num_disp + 1) & ~1;
  const int lane = threadIdx.x, hl = lane & (SUB - 1), sub = lane / SUB;
  uint16_t* full_prior = sm + sub * fpn;            // num_disp per group
  uint16_t* p2tab = sm + NSUB * fpn;                // 256: max(P1, P2 / |grey step|), the adaptive P2 of SGM.cc:813-818
  uint16_t* stage = p2tab + 256;                    // num_disp: the new vector of a large box
  // A wavefront's lines are `spread` lines apart: pixels without a trusted coarser disparity come in patches, a line inside one is slow for
  // as long as it stays there, and neighbouring lines are slow TOGETHER — a wavefront that owns one of them at a time falls less far behind
  // the others than one that owns four (the launch ends with its slowest wavefront).
  const int line = ((int)blockIdx.x / spread) * (NSUB * spread) + (int)blockIdx.x % spread + sub * spread;
  int c, r, dc, dr;
  line_start(D, g, line < lines ? line : 0, dc, dr, c, r);
  const unsigned BAD = (255u + p2) & 0xffffu;
  for (int i = lane; i < NSUB * fpn; i += 64) sm[i] = (uint16_t)BAD;
  for (int q = lane; q < 256; q += 64) {
    unsigned v = p2;
    if (q > 0) v /= (unsigned)q;
    if (v < p1) v = p1;
    p2tab[q] = (uint16_t)v;
  }
  lds_barrier();
  const long long delta = (long long)dr * g.ocols + dc;         // pixel index step along the line
  const long long ldelta = (long long)dr * lw + dc;
  int len = 0;                                                  // steps of the line inside the image
  if (line < lines && c >= 0 && r >= 0 && c < g.ocols && r < g.orows) {
    const int len_c = dc > 0 ? g.ocols - c : (dc < 0 ? c + 1 : 0x7fffffff);
    const int len_r = dr > 0 ? g.orows - r : (dr < 0 ? r + 1 : 0x7fffffff);
    len = min(len_c, len_r);
  }
  auto group_max = [&](int v) __attribute__((always_inline)) {  // the largest of the groups' (group-uniform) values, wave-uniform
    int m = max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 32));
    if (NSUB == 4) m = max(m, max(__builtin_amdgcn_readlane(v, 16), __builtin_amdgcn_readlane(v, 48)));
    return m;
  };
  // (a group without a line reads the records of pixel 0 and never writes)
  long long p = len > 0 ? (long long)r * g.ocols + c : 0;
  long long lp = len > 0 ? (long long)(r + min_row) * lw + (c + min_col) : (long long)min_row * lw + min_col;
  const int maxlen = group_max(len);
  // The records (box, vector start, grey value) are requested TWO pixels ahead and the cost bytes one: a step of four short lines is over
  // before a memory round trip is, and the sums' atomics of the step before sit in the same in-order queue as the loads.
  B4 b = bounds[p];
  unsigned long long st = starts[p];
  int cur = left[lp];
  if (1 < len) { p += delta; lp += ldelta; }
  B4 b_n = bounds[p];
  unsigned long long st_n = starts[p];
  int cur_n = left[lp];
  unsigned cv0 = cost[st + max(min(hl, b4_count(b) - 1), 0)];     // cost[st + hl] of the current pixel (prefetched, clamped)
  // cost bytes of the large boxes of the coming step: chunk k of group s's vector, element 64 k + lane
  constexpr int KC = 8;
  unsigned cbig[NSUB][KC];
#pragma unroll
  for (int s = 0; s < NSUB; ++s)
#pragma unroll
    for (int k = 0; k < KC; ++k) cbig[s][k] = 0;
  auto request_big = [&](const B4& bb, unsigned long long stt, bool on_line) __attribute__((always_inline)) {
    const int ndl = on_line ? b4_count(bb) : 0;
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(ndl > SUB);
    if (mask == 0) return;
#pragma unroll
    for (int s = 0; s < NSUB; ++s) {
      if (!((mask >> (s * SUB)) & 1ull)) continue;               // wave-uniform
      const int l0 = s * SUB;
      const int nd_s = __builtin_amdgcn_readlane(ndl, l0);
      const unsigned long long st_s = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)stt, l0) |
                                      ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(stt >> 32), l0) << 32);
#pragma unroll
      for (int k = 0; k < KC; ++k)
        if (64 * k < nd_s) cbig[s][k] = cost[st_s + (unsigned)min(64 * k + lane, nd_s - 1)];
    }
  };
  request_big(b, st, len > 0);
  int last_val = -1;
  unsigned min_prior = 0;
  B4 bp{0, 0, -1, -1};
  // value of the box's element i from the previous vector, and its cell in the full-range buffer: value | cell << 16
  auto eval = [&](int i, unsigned cb, const B4& bb, int wd, float inv_wd, const uint16_t* fp, bool has_prior, unsigned dJ, unsigned minp)
      __attribute__((always_inline)) -> unsigned {
    int qy, qx;
    divmod_f(i, wd, inv_wd, qy, qx);
    const int dx = bb.x0 + qx, dy = bb.y0 + qy;
    const int xo = dx - g.min_dx, yo = dy - g.min_dy;
    unsigned v = cb;
    if (has_prior) {
      const int xl = dx - 1 < g.min_dx ? xo : xo - 1, xm = dx + 1 > g.max_dx ? xo : xo + 1;
      const int yl = (dy - 1 < g.min_dy ? yo : yo - 1) * g.num_dx, ym = (dy + 1 > g.max_dy ? yo : yo + 1) * g.num_dx, yc = yo * g.num_dx;
      unsigned m = fp[yl + xo];
      m = min(m, (unsigned)fp[yc + xl]); m = min(m, (unsigned)fp[yc + xm]); m = min(m, (unsigned)fp[ym + xo]);
      m = min(m, (unsigned)fp[yl + xl]); m = min(m, (unsigned)fp[yl + xm]);
      m = min(m, (unsigned)fp[ym + xl]); m = min(m, (unsigned)fp[ym + xm]);
      v = adds16(m, p1);
      v = min(v, min((unsigned)fp[yc + xo], dJ));
      v = adds16(v, cb);
      v = subs16(v, minp);
    }
    return v | ((unsigned)(yo * g.num_dx + xo) << 16);
  };
  // an element goes to its cell and to the sums.  One 32-bit atomic per pair of u16 elements (see path_inplace_kernel); `edge`: the last lane
  // of those that share the vector — it never takes its right neighbour's value; `first`: the first of them always adds its own element.
  auto commit = [&](unsigned rcv, unsigned long long e, uint16_t* fp, unsigned& mn, bool first, bool edge) __attribute__((always_inline)) {
    const unsigned cell = rcv >> 16, val = rcv & 0xffffu;
    const bool on = cell != 0xffffu;
    if (on) {
      fp[cell] = (uint16_t)val;
      mn = min(mn, val);
    }
    const unsigned mine = on ? val : 0u;
    const unsigned next = wave_shl1(mine, 0u);
    const bool low = (e & 1ull) == 0;
    if (D.n == 1) {                                     // one direction per launch: plain u16 sums (see accum_add_u16)
      if (on) accum[e] = (uint16_t)(accum[e] + mine);
    } else if (on && (low || first)) {
      const unsigned v = low ? (mine | (edge ? 0u : next << 16)) : (mine << 16);
      atomicAdd(reinterpret_cast<unsigned*>(accum) + (e >> 1), v);
    }
  };
  for (int step = 0; step < maxlen; ++step) {
    const bool act = step < len;
    const int wd = b.x1 - b.x0 + 1, nd = act ? wd * (b.y1 - b.y0 + 1) : 0;
    const bool big = nd > SUB;
    const unsigned long long bigmask = __builtin_amdgcn_ballot_w64(big);
    // the records of the pixel after the next one; no load under a condition (path_inplace_kernel)
    if (step + 2 < len) { p += delta; lp += ldelta; }
    const B4 b_nn = bounds[p];
    const unsigned long long st_nn = starts[p];
    const int cur_nn = left[lp];
    const float inv_wd = __builtin_amdgcn_rcpf((float)max(wd, 1));
    int grad = cur - last_val; grad = grad < 0 ? -grad : grad;
    const unsigned dJ = (min_prior + (unsigned)p2tab[last_val >= 0 ? grad : 0]) & 0xffffu;
    // ---- the lines whose box fits the group: reads of the previous vector, barrier, the new vector
    unsigned rc = 0xffffffffu;
    if (hl < nd && !big) rc = eval(hl, cv0, b, wd, inv_wd, full_prior, last_val >= 0, dJ, min_prior);
    const unsigned cn0 = cost[st_n + max(min(hl, b4_count(b_n) - 1), 0)];      // the next pixel's first cost bytes
    lds_barrier();
    unsigned mn = BAD;                            // (the minimum of an empty vector is BAD_VAL, SGMAssist.h)
    commit(rc, st + (unsigned)hl, full_prior, mn, hl == 0, hl == SUB - 1);
    unsigned new_min = sub_min_u32_fused<SUB>(mn, lane);
    // ---- the lines with a large box, one after the other, on all 64 lanes
    if (bigmask != 0) {
#pragma unroll
      for (int s = 0; s < NSUB; ++s) {
        if (!((bigmask >> (s * SUB)) & 1ull)) continue;          // wave-uniform
        const int l0 = s * SUB;
        const B4 bs{__builtin_amdgcn_readlane(b.x0, l0), __builtin_amdgcn_readlane(b.y0, l0), __builtin_amdgcn_readlane(b.x1, l0), __builtin_amdgcn_readlane(b.y1, l0)};
        const unsigned long long st_s = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)st, l0) |
                                        ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(st >> 32), l0) << 32);
        const int wd_s = bs.x1 - bs.x0 + 1, nd_s = wd_s * (bs.y1 - bs.y0 + 1);
        const float inv_s = __builtin_amdgcn_rcpf((float)wd_s);
        const bool prior_s = __builtin_amdgcn_readlane(last_val, l0) >= 0;
        const unsigned dJ_s = (unsigned)__builtin_amdgcn_readlane((int)dJ, l0), minp_s = (unsigned)__builtin_amdgcn_readlane((int)min_prior, l0);
        uint16_t* fp_s = sm + s * fpn;
#pragma unroll
        for (int k = 0; k < KC; ++k)
          if (64 * k < nd_s && 64 * k + lane < nd_s) stage[64 * k + lane] = (uint16_t)eval(64 * k + lane, cbig[s][k], bs, wd_s, inv_s, fp_s, prior_s, dJ_s, minp_s);
        for (int i = 64 * KC + lane; i < nd_s; i += 64) stage[i] = (uint16_t)eval(i, (unsigned)cost[st_s + (unsigned)i], bs, wd_s, inv_s, fp_s, prior_s, dJ_s, minp_s);
        lds_barrier();
        unsigned mn_s = BAD;
        for (int i0 = 0; i0 < nd_s; i0 += 64) {                  // (every lane takes part in every trip: the neighbour exchange of commit)
          const int i = i0 + lane;
          unsigned rcv = 0xffffffffu;
          if (i < nd_s) {
            int qy, qx;
            divmod_f(i, wd_s, inv_s, qy, qx);
            rcv = (unsigned)stage[i] | ((unsigned)((bs.y0 + qy - g.min_dy) * g.num_dx + (bs.x0 + qx - g.min_dx)) << 16);
          }
          commit(rcv, st_s + (unsigned)i, fp_s, mn_s, lane == 0, lane == 63);
        }
        mn_s = wave_min_u32_fused(mn_s);
        if (sub == s) new_min = mn_s;
        lds_barrier();                                          // `stage` is free again
      }
    }
    request_big(b_n, st_n, step + 1 < len);                     // (their registers are free now)
    // cells only the previous box covered go back to BAD_VAL
    const bool shrunk = act && last_val >= 0 && (bp.x0 < b.x0 || bp.x1 > b.x1 || bp.y0 < b.y0 || bp.y1 > b.y1);
    if (__builtin_amdgcn_ballot_w64(shrunk) != 0) {
      const int wp = bp.x1 - bp.x0 + 1, np = shrunk ? wp * (bp.y1 - bp.y0 + 1) : 0;
      const int npmax = group_max(np);
      const float inv_wp = __builtin_amdgcn_rcpf((float)max(wp, 1));
      for (int i = hl; i < npmax; i += SUB) {
        if (i < np) {
          int qy, qx;
          divmod_f(i, wp, inv_wp, qy, qx);
          const int dx = bp.x0 + qx, dy = bp.y0 + qy;
          if (dx < b.x0 || dx > b.x1 || dy < b.y0 || dy > b.y1)
            full_prior[(dy - g.min_dy) * g.num_dx + (dx - g.min_dx)] = (uint16_t)BAD;
        }
      }
    }
    min_prior = new_min;
    lds_barrier();
    if (act) { bp = b; last_val = cur; }
    b = b_n; st = st_n; cur = cur_n; cv0 = cn0;
    b_n = b_nn; st_n = st_nn; cur_n = cur_nn;
  }
}

// Same recurrence when EVERY pixel searches the full disparity range (no masks, no previous level — e.g. a single-level
// SGM run): the previous pixel's vector already is the full-range buffer, so there is no scatter / reset phase.
//  * ONE wavefront per scan line (no workgroup barrier on the serial path; the lines of a direction are the parallelism),
//    EPT = ceil(num_disp / 64) disparities per lane; the vector is updated IN PLACE: a step reads all its neighbours into
//    registers before it writes (LDS executes a wave's accesses in order), so every LDS address is fixed for the whole line;
//  * ONE_D (num_dy == 1, the +-N x 1 searches): the clamped 2-D adjacency collapses to {left, right, self};
//  * the line is processed in chunks of K pixels: cost (u8) and accumulator (u16) vectors, stored with a stride of `stride`
//    elements (multiple of 16) per pixel, are bulk-loaded into LDS 16 bytes per lane (many loads in flight while other lines of the
//    CU compute), the K serial steps touch LDS only, and the K updated accumulator vectors are written back as dwords.
template <int EPT, bool ONE_D>
__global__ void __launch_bounds__(64)
path_uniform_kernel(SgmGeom g, DirSet D, int K, int stride,
                    const uint8_t* __restrict__ left, int lw, int min_col, int min_row,
                    const uint8_t* __restrict__ cost, uint16_t* __restrict__ accum, unsigned p1, unsigned p2) {
  extern __shared__ uint16_t sm[];
  const int num_disp = g.num_dx * g.num_dy;
  uint16_t* buf = sm;                                                                // EPT*64 current path costs
  uint16_t* p2tab = sm + EPT * 64;                                                   // max(p1, p2 / gradient), gradient 0..255
  uint16_t* cacc = p2tab + 256;                                                      // K x stride
  uint8_t* ccost = reinterpret_cast<uint8_t*>(cacc + (size_t)K * stride);            // K x stride
  uint8_t* pix = ccost + (size_t)K * stride;                                         // K left-image values
  const int tid = threadIdx.x;
  int c0, r0, dc, dr;
  line_start(D, g, blockIdx.x, dc, dr, c0, r0);
  const int len_c = dc > 0 ? g.ocols - c0 : (dc < 0 ? c0 + 1 : 0x7fffffff);
  const int len_r = dr > 0 ? g.orows - r0 : (dr < 0 ? r0 + 1 : 0x7fffffff);
  const int len = min(len_c, len_r);
  for (int q = tid; q < 256; q += 64) {
    unsigned v = p2;
    if (q > 0) v /= (unsigned)q;
    if (v < p1) v = p1;
    p2tab[q] = (uint16_t)v;
  }
  // LDS addresses of the centre and its neighbours, fixed for the whole line; dead lanes of the last slot read slot 0
  constexpr int NN = ONE_D ? 3 : 9;
  const uint16_t* nptr[EPT][NN];
  bool live[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int i = tid + e * 64;
    live[e] = i < num_disp;
    const int ii = live[e] ? i : 0;
    const int qy = ii / g.num_dx, qx = ii - qy * g.num_dx;
    const int xl = qx - 1 < 0 ? qx : qx - 1, xm = qx + 1 > g.max_dx - g.min_dx ? qx : qx + 1;
    const int yl = (qy - 1 < 0 ? qy : qy - 1) * g.num_dx, ym = (qy + 1 > g.max_dy - g.min_dy ? qy : qy + 1) * g.num_dx, yc = qy * g.num_dx;
    nptr[e][0] = buf + yc + qx;
    nptr[e][1] = buf + yc + xl; nptr[e][2] = buf + yc + xm;
    if (!ONE_D) {
      nptr[e][3] = buf + yl + qx; nptr[e][4] = buf + ym + qx;
      nptr[e][5] = buf + yl + xl; nptr[e][6] = buf + yl + xm; nptr[e][7] = buf + ym + xl; nptr[e][8] = buf + ym + xm;
    }
  }
  // 16-byte quanta per pixel vector (stride is a multiple of 16 elements), exact division j / q for j < 2^16 by a
  // multiply-high with ceil(2^32 / q), and the extra global offset per chunk pixel
  const int q_cost = stride / 16;
  const unsigned m_cost = (unsigned)((0x100000000ull + q_cost - 1) / q_cost);
  const long long delta = (long long)dr * g.ocols + dc;
  const long long d_cost = (delta - 1) * q_cost;
  int last_val = -1;
  unsigned min_prior = 0;
  for (int base = 0; base < len; base += K) {
    const int kk = min(K, len - base);
    {                                                           // bulk load, 16 bytes per lane
      const uint4* gc = reinterpret_cast<const uint4*>(cost);
      uint4* lc = reinterpret_cast<uint4*>(ccost);
      // global index of chunk element j = k * q + w:  (P0 + k * delta) * q + w  =  P0 * q + j + k * (delta - 1) * q
      const long long pbase = (long long)(r0 + base * dr) * g.ocols + (c0 + base * dc);
      for (int j = tid; j < kk * q_cost; j += 64) {
        const int k = q_cost == 1 ? j : (int)__umulhi((unsigned)j, m_cost);     // ceil(2^32 / 1) does not fit 32 bits
        lc[j] = gc[pbase * q_cost + j + (long long)k * d_cost];
      }
    }
    if (tid < kk) pix[tid] = left[(size_t)(r0 + (base + tid) * dr + min_row) * lw + (c0 + (base + tid) * dc + min_col)];
    __builtin_amdgcn_wave_barrier();
    for (int k = 0; k < kk; ++k) {                              // K serial steps, LDS only
      const int vcur = pix[k];
      const uint8_t* cc = ccost + k * stride;
      uint16_t* ac = cacc + k * stride;
      unsigned res[EPT];
      if (last_val < 0) {
#pragma unroll
        for (int e = 0; e < EPT; ++e) res[e] = live[e] ? (unsigned)cc[tid + e * 64] : 0xffffu;
      } else {
        int grad = vcur - last_val; grad = grad < 0 ? -grad : grad;
        const unsigned dJ = (min_prior + (unsigned)p2tab[grad]) & 0xffffu;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          unsigned m = *nptr[e][1];
#pragma unroll
          for (int q = 2; q < NN; ++q) m = min(m, (unsigned)*nptr[e][q]);
          const unsigned ctr = *nptr[e][0];
          if (ONE_D) m = min(m, ctr);                           // the clamped vertical neighbours are the centre itself
          unsigned v = adds16(m, p1);
          v = min(v, min(ctr, dJ));
          v = adds16(v, live[e] ? (unsigned)cc[tid + e * 64] : 0u);
          res[e] = live[e] ? subs16(v, min_prior) : 0xffffu;
        }
      }
      __builtin_amdgcn_wave_barrier();                          // all neighbour reads are issued before the in-place writes
      unsigned mn = 0xffffu;
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * 64;
        if (live[e]) { buf[i] = (uint16_t)res[e]; ac[i] = (uint16_t)res[e]; mn = min(mn, res[e]); }
      }
      min_prior = wave_min_u32(mn);
      __builtin_amdgcn_wave_barrier();
      last_val = vcur;
    }
    {                                                           // bulk add: one 32-bit atomic per pair of path costs
      unsigned* ga = reinterpret_cast<unsigned*>(accum);
      const unsigned* la = reinterpret_cast<const unsigned*>(cacc);
      const int q32 = stride / 2;                               // dwords per pixel vector
      const unsigned m32 = (unsigned)((0x100000000ull + q32 - 1) / q32);
      const long long pbase = (long long)(r0 + base * dr) * g.ocols + (c0 + base * dc);
      const int nlive = (num_disp + 1) / 2;                     // dwords that hold live disparities
      for (int j = tid; j < kk * q32; j += 64) {
        const int k = (int)__umulhi((unsigned)j, m32), w = j - k * q32;
        if (w < nlive) {
          unsigned v = la[j];
          if (2 * w + 1 >= num_disp) v &= 0xffffu;              // the odd tail shares its dword with a dead slot
          unsigned* a = ga + pbase * q32 + j + (long long)k * (delta - 1) * q32;
          if (D.n == 1) *a = as_u32(as_us2(*a) + as_us2(v));    // one direction per launch: u16 wrap-around per element
          else atomicAdd(a, v);
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// +-N x 1 searches on the packed 16-bit ALU: a lane owns PAIRS of neighbouring disparities (d_2j, d_2j+1) in one dword and the
// whole recurrence runs on v_pk_{min,add,sub}_u16 with clamping (= the reference's saturating SSE ops, SGM.cc:936-984).
// The left / right neighbour pairs come from the adjacent dwords through v_alignbit; because the 1-D adjacency already
// contains the centre, the clamped neighbours at both ends of the range are represented by 0xffff guards (they never win).

// ---- register-resident scan lines ---------------------------------------------------------------------------------------
// The kernel above spends ~95 issue slots per pixel step (LDS neighbour exchange with two wave barriers, a staged chunk of costs,
// a staged chunk of results, a bulk atomic pass with index arithmetic) and, with lane <-> pair j and j + 64, runs the whole second
// slot for ONE live pair at 129 disparities.  Here the path vector never leaves the registers:
//   * lane l owns the EPT CONSECUTIVE pairs l * EPT .. l * EPT + EPT - 1 (129 disparities: 33 lanes x 2 pairs), so the d-1 / d+1
//     neighbours are the lane's own registers (`v_alignbit`) except at the two ends, which come from lane l-1 / l+1 by one DPP
//     wave shift each;
//   * the lane's cost bytes of a step are EPT * 2 consecutive bytes: one global load per step, issued a chunk of KC steps ahead
//     into a second register set (no LDS, no barrier);
//   * the lane's results are EPT consecutive dwords of the accumulator: one 64-bit (or 32-bit) atomic per step straight from the
//     registers, addressed by a scalar base that advances by the line's stride;
//   * everything that is uniform over the wave (the adaptive P2 of the step, min_prior, the penalties) lives on the scalar unit:
//     the P2 values of 64 steps are computed at once, lane k for step k, and read back with v_readlane.
// A line without predecessor starts from r = 0, min_prior = 0: min(prev..) = 0 and min(.., ctr = 0, ..) = 0, so the general step
// yields the plain cost, exactly the reference's first pixel (SGM.cc:1013-1150).
template <int EPT>
struct CostWords { static constexpr int N = EPT == 3 ? 3 : (EPT + 1) / 2; };

template <int EPT>
__device__ __forceinline__ void load_cost_words(const uint8_t* p, bool in, unsigned (&w)[CostWords<EPT>::N]) {
  if constexpr (EPT == 1) w[0] = in ? (unsigned)*reinterpret_cast<const uint16_t*>(p) : 0u;
  else if constexpr (EPT == 2) w[0] = in ? *reinterpret_cast<const unsigned*>(p) : 0u;
  else if constexpr (EPT == 3) {
#pragma unroll
    for (int e = 0; e < 3; ++e) w[e] = in ? (unsigned)reinterpret_cast<const uint16_t*>(p)[e] : 0u;
  } else {
    const uint2 v = in ? *reinterpret_cast<const uint2*>(p) : make_uint2(0u, 0u);
    w[0] = v.x; w[1] = v.y;
  }
}
template <int EPT>
__device__ __forceinline__ unsigned cost_pair(const unsigned (&w)[CostWords<EPT>::N], int e) {   // (c_2j, c_2j+1) as two u16
  if constexpr (EPT == 3) return __builtin_amdgcn_perm(0u, w[e], 0x0c010c00u);
  else return __builtin_amdgcn_perm(0u, w[e >> 1], (e & 1) ? 0x0c030c02u : 0x0c010c00u);
}

// dst keeps its value in the lane without a source (lane 0 / lane 63): initialised once to the guard 0xffffffff, never rewritten.
__device__ __forceinline__ void wave_shr1_keep(unsigned& dst, unsigned src) {
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(dst) : "v"(src));
}
__device__ __forceinline__ void wave_shl1_keep(unsigned& dst, unsigned src) {
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(dst) : "v"(src));
}

// ACC: how the path costs reach the accumulated-cost volume.
//   ACC_ATOMIC  64-bit atomics (pairs of dwords) — needed when several directions share a launch.  Measured on 2048^2 x 129, all
//               8 directions in one launch: 8.0 ms, of which the recurrence is 3.05 ms (ACC_NONE) — device-scope atomics are executed at
//               the memory side, one 1.2 GB read-modify-write pass per direction; 32-bit atomics: 13.4 ms;
//   ACC_STORE   plain store: the first direction of a call initialises the volume (no memset);
//   ACC_RMW     load (fetched a chunk ahead, like the costs) + v_pk_add_u16 + store: race-free when the launch holds ONE direction,
//               because then every pixel lies on exactly one line.  u16 wrap-around per element, as the reference's `+=`;
//   ACC_RMW_WTA ACC_RMW for the LAST direction of a call: the pixel's sums are final in the step's registers, so the winner is taken there — a
//               unique minimum writes the disparity, a tie marks the pixel for wta_kernel (the smoothing loop of the reference) — instead of a
//               ninth pass over the volume (wta_uniform_kernel: 0.52 of the 6.9 ms at 2048^2 x 129);
//   ACC_NONE    timing experiments only.
// No load of the step loop sits under a condition: a lane outside the pixel's vector reads lane 0's bytes, and the fetch pointers
// stop advancing at the last pixel of the line — with conditional loads the compiler falls back to s_waitcnt vmcnt(0) in front
// of every chunk, i.e. no prefetch at all (measured: 770 clk per step).
enum { ACC_ATOMIC = 1, ACC_STORE = 2, ACC_NONE = 3, ACC_RMW = 4, ACC_RMW_WTA = 5 };
#ifndef VWGPU_PATH_KC
#define VWGPU_PATH_KC 8
#endif
// CenArgs: the two census rasters at output pixel (0, 0), disparity 0 (path_ring_kernel<.., CEN> forms the Hamming costs itself).
struct CenArgs {
  const uint64_t* lcen; const uint64_t* rcen;
  int lcw, rcw;
};
// WPB lines (wavefronts) per workgroup: the lines of a workgroup are neighbours, so the 128-byte lines that two neighbouring
// pixels' vectors share (a 272-byte vector straddles three) are fetched by ONE compute unit / XCD instead of two.
// (Round 6 also built this kernel with the census costs formed in registers — 32 bytes of census words per lane and step, fetched a
// chunk ahead: 320 registers, one wave per SIMD, 9.6 ms against 5.4 — and removed it; profiles/r06_sgm_traffic.md.)
template <int EPT, int ACC, int KC, int WPB>
__global__ void __launch_bounds__(64 * WPB)
path_uniform_reg_kernel(SgmGeom g, DirSet D, int stride, const uint8_t* __restrict__ left, int lw, int min_col, int min_row,
                        const uint8_t* __restrict__ cost, uint16_t* __restrict__ accum, unsigned p1, unsigned p2,
                        int32_t* __restrict__ disp = nullptr, uint8_t* __restrict__ todo = nullptr, int* __restrict__ any_todo = nullptr) {
  constexpr bool RMW = (ACC == ACC_RMW || ACC == ACC_RMW_WTA);
  constexpr int NW = CostWords<EPT>::N;
  const int num_disp = g.num_dx;                                                     // num_dy == 1
  const int npairs = (num_disp + 1) / 2;
  const int tid = threadIdx.x & 63;
  const int gline = __builtin_amdgcn_readfirstlane((int)blockIdx.x * WPB + (int)(threadIdx.x >> 6));
  if (gline >= D.line0[D.n]) return;                                                 // (the last workgroup of a launch)
  // the tables are indexed dynamically (they land in scratch): readfirstlane tells the compiler the values are wave-uniform
  int dirq = 0;
  while (dirq + 1 < D.n && gline >= D.line0[dirq + 1]) ++dirq;
  dirq = __builtin_amdgcn_readfirstlane(dirq);
  const int dc = __builtin_amdgcn_readfirstlane(D.dc[dirq]), dr = __builtin_amdgcn_readfirstlane(D.dr[dirq]);
  const int ww = g.ocols, wh = g.orows;
  int c0, r0;
  {
    const int line = gline - __builtin_amdgcn_readfirstlane(D.line0[dirq]);
    const int nf = __builtin_amdgcn_readfirstlane(D.n_first[dirq]);
    if (line < nf) {
      if (__builtin_amdgcn_readfirstlane(D.row_border[dirq])) { c0 = line; r0 = dr > 0 ? 0 : wh - 1; }
      else { r0 = line; c0 = dc > 0 ? 0 : ww - 1; }
    } else {
      // The side-border lines of a diagonal direction in the order opposite to the row-border ones (long -> short, then short ->
      // long): all lines of a launch are resident at once, and lines i, i + n, i + 2n ... share a SIMD.
      int i = line - nf;
      if (D.rev_second && (dc > 0) == (dr > 0)) i = wh - 2 - i;
      r0 = i + __builtin_amdgcn_readfirstlane(D.second_skip[dirq]);
      c0 = dc > 0 ? 0 : ww - 1;
    }
  }
  const int len_c = dc > 0 ? ww - c0 : (dc < 0 ? c0 + 1 : 0x7fffffff);
  const int len_r = dr > 0 ? wh - r0 : (dr < 0 ? r0 + 1 : 0x7fffffff);
  const int len = min(len_c, len_r);
  const int q32 = stride / 2;
  unsigned dead[EPT], r[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int j = tid * EPT + e;
    dead[e] = j >= npairs ? 0xffffffffu : ((2 * j + 1 >= num_disp) ? 0xffff0000u : 0u);
    r[e] = dead[e];                                      // no predecessor: r = 0 in the live elements (see above)
  }
  const bool in = tid * EPT + EPT <= q32;               // the lane's dwords / cost bytes lie inside the pixel's vector
  bool live0[EPT], live1[EPT];                          // ACC_RMW_WTA: the halves of the lane's pairs that are disparities of the search
#pragma unroll
  for (int e = 0; e < EPT; ++e) { live0[e] = 2 * (tid * EPT + e) < num_disp; live1[e] = 2 * (tid * EPT + e) + 1 < num_disp; }
  bool flagged = false;
  int last_val = 0;
  unsigned min_prior = 0;
  // uniform (scalar) running pointers + a 32-bit lane offset for the loads; a per-lane pointer with a per-lane stride for the stores
  const long long delta = (long long)dr * g.ocols + dc;
  const long long pbase = (long long)r0 * g.ocols + c0;
  long long pcur = pbase;                               // ACC_RMW_WTA: the pixel of the current step
  const unsigned loff_c = in ? (unsigned)(tid * EPT * 2) : 0u, loff_a = in ? (unsigned)(tid * EPT) : 0u;
  const uint8_t* cfetch = cost + pbase * stride;                    // cost vector of the next step to fetch
  const long long cstep = delta * stride;
  const unsigned* afetch = reinterpret_cast<const unsigned*>(accum) + pbase * q32;
  const long long astep = delta * q32;
  unsigned* astore = reinterpret_cast<unsigned*>(accum) + pbase * q32 + loff_a;      // stores are masked by `in`
  const long long astore_step = astep;
  const us2 p1p1 = as_us2(p1 | (p1 << 16));
  const uint8_t* lp = left + (size_t)(r0 + min_row) * lw + (c0 + min_col);
  const long long lstep = (long long)dr * lw + dc;

  unsigned penv = 0;
  int pvn = (int)lp[min(tid, len - 1) * lstep];                     // grey values of the first 64 steps
  auto refresh = [&](int s0) __attribute__((always_inline)) {      // s0 % 64 == 0: penalties of steps s0 .. s0 + 63
    const int pv = pvn;
    pvn = (int)lp[min(s0 + 64 + tid, len - 1) * lstep];             // next block's, a block ahead
    const int prev = (int)wave_shr1((unsigned)pv, (unsigned)last_val);
    int grad = pv - prev; grad = grad < 0 ? -grad : grad;
    unsigned v = p2 / (unsigned)max(grad, 1);                        // branch-free: p2 itself where the grey value does not change
    if (v < p1) v = p1;
    penv = v & 0xffffu;
    last_val = __builtin_amdgcn_readlane(pv, min(63, len - 1 - s0));
  };
  unsigned pm = 0xffffffffu, pn = 0xffffffffu;                      // lane 0 / lane 63 keep the guard for good
  auto step = [&](const unsigned (&w)[NW], const unsigned (&aw)[EPT], int s) __attribute__((always_inline)) {
    const unsigned pen = (unsigned)__builtin_amdgcn_readlane((int)penv, s & 63);
    const unsigned dj = (min_prior + pen) & 0xffffu;
    const us2 dJ = as_us2(dj | (dj << 16)), mp = as_us2(min_prior | (min_prior << 16));
    wave_shr1_keep(pm, r[EPT - 1]);
    wave_shl1_keep(pn, r[0]);
    unsigned al[EPT + 1];                                           // al[e] = (d_2j-1, d_2j) of pair e; al[e+1] = (d_2j+1, d_2j+2)
    al[0] = __builtin_amdgcn_alignbit(r[0], pm, 16);
#pragma unroll
    for (int e = 1; e < EPT; ++e) al[e] = __builtin_amdgcn_alignbit(r[e], r[e - 1], 16);
    al[EPT] = __builtin_amdgcn_alignbit(pn, r[EPT - 1], 16);
    us2 mn2 = as_us2(0xffffffffu);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const us2 ctr = as_us2(r[e]);
      us2 m = __builtin_elementwise_min(__builtin_elementwise_min(as_us2(al[e]), as_us2(al[e + 1])), ctr);
      us2 v = __builtin_elementwise_add_sat(m, p1p1);
      v = __builtin_elementwise_min(v, __builtin_elementwise_min(ctr, dJ));
      v = __builtin_elementwise_add_sat(v, as_us2(cost_pair<EPT>(w, e)));
      v = __builtin_elementwise_sub_sat(v, mp);
      r[e] = as_u32(v) | dead[e];
      mn2 = e == 0 ? as_us2(r[e]) : __builtin_elementwise_min(mn2, as_us2(r[e]));
    }
    // dead halves put garbage into the padding of the pixel's vector (never read)
    if constexpr (ACC == ACC_ATOMIC) {
      if (in) {
        if constexpr (EPT == 2 || EPT == 4) {
#pragma unroll
          for (int e = 0; e < EPT; e += 2)
            atomicAdd(reinterpret_cast<unsigned long long*>(astore + e), (unsigned long long)r[e] | ((unsigned long long)r[e + 1] << 32));
        } else {
#pragma unroll
          for (int e = 0; e < EPT; ++e) atomicAdd(astore + e, r[e]);
        }
      }
    } else if constexpr (ACC != ACC_NONE) {
      unsigned o[EPT];
#pragma unroll
      for (int e = 0; e < EPT; ++e) o[e] = RMW ? as_u32(as_us2(aw[e]) + as_us2(r[e])) : r[e];
      if constexpr (ACC == ACC_RMW_WTA) {
        // keys (value << 16 | index) of the lane's live elements; the wave minimum is the winner, ties on its value go to wta_kernel
        unsigned k = 0xffffffffu;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
          const unsigned j2 = (unsigned)(tid * EPT + e) * 2u;
          const unsigned k0 = live0[e] ? ((o[e] << 16) | j2) : 0xffffffffu, k1 = live1[e] ? ((o[e] & 0xffff0000u) | (j2 + 1u)) : 0xffffffffu;
          k = min(k, min(k0, k1));
        }
        const unsigned key = wave_min_u32_fused(k), mv = key >> 16;
        int cnt = 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e)
          cnt += __popcll(__ballot(live0[e] && (o[e] & 0xffffu) == mv)) + __popcll(__ballot(live1[e] && (o[e] >> 16) == mv));
        if (tid == 0) {
          if (cnt > 1) { todo[pcur] = 1; }
          else {
            todo[pcur] = 0;
            int32_t* w3 = disp + pcur * 3;
            w3[0] = (int)(key & 0xffffu) + g.min_dx; w3[1] = g.min_dy; w3[2] = 0x7fffffff;
          }
        }
        flagged |= cnt > 1;
        pcur += delta;
      }
      if (in) {                                                      // (a shared dump slot for the other lanes: 8.5 ms instead of 5)
        if constexpr (EPT == 2) *reinterpret_cast<uint2*>(astore) = make_uint2(o[0], o[1]);
        else if constexpr (EPT == 4) *reinterpret_cast<uint4*>(astore) = make_uint4(o[0], o[1], o[2], o[3]);
        else {
#pragma unroll
          for (int e = 0; e < EPT; ++e) astore[e] = o[e];
        }
      }
    }
    astore += astore_step;
    const unsigned mnu = as_u32(mn2);
    min_prior = wave_min_u32_fused(min(mnu & 0xffffu, mnu >> 16));
  };
  auto fetch = [&](unsigned (&buf)[KC][NW], unsigned (&abuf)[KC][EPT]) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      load_cost_words<EPT>(cfetch + loff_c, true, buf[k]);
      if constexpr (RMW) {
        const unsigned* a = afetch + loff_a;
        if constexpr (EPT == 2) { const uint2 v = *reinterpret_cast<const uint2*>(a); abuf[k][0] = v.x; abuf[k][1] = v.y; }
        else if constexpr (EPT == 4) { const uint4 v = *reinterpret_cast<const uint4*>(a); abuf[k][0] = v.x; abuf[k][1] = v.y; abuf[k][2] = v.z; abuf[k][3] = v.w; }
        else {
#pragma unroll
          for (int e = 0; e < EPT; ++e) abuf[k][e] = a[e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < EPT; ++e) abuf[k][e] = 0u;
      }
      cfetch += cstep;                                               // runs up to 2 KC steps past the line: guard zones (host side)
      afetch += astep;
    }
  };
  auto steps = [&](const unsigned (&buf)[KC][NW], const unsigned (&abuf)[KC][EPT], int s0, auto guarded) __attribute__((always_inline)) {
    if ((s0 & 63) == 0 && s0 < len) refresh(s0);
#pragma unroll
    for (int k = 0; k < KC; ++k)
      if (!decltype(guarded)::value || s0 + k < len) step(buf[k], abuf[k], s0 + k);
  };

  unsigned ca[KC][NW], cb[KC][NW], aa[KC][EPT], ab[KC][EPT];
  const int nfull = len / KC;
  fetch(ca, aa);
  int ch = 0;
  for (; ch + 2 <= nfull; ch += 2) {                                // full chunks: no per-step test
    fetch(cb, ab);
    steps(ca, aa, ch * KC, std::false_type());
    fetch(ca, aa);
    steps(cb, ab, (ch + 1) * KC, std::false_type());
  }
  if (ch * KC < len) {                                              // the last one or two chunks
    fetch(cb, ab);
    steps(ca, aa, ch * KC, std::true_type());
    steps(cb, ab, (ch + 1) * KC, std::true_type());
  }
  if constexpr (ACC == ACC_RMW_WTA) {
    if (flagged && tid == 0) atomicOr(any_todo, 1);
  }
}


// ---- round 6: the same recurrence fed through an LDS ring (path_ring_kernel) ---------------------------------------------------
// What bounds path_uniform_reg_kernel is not the HBM rate but the bytes it keeps in flight: a wavefront prefetches two chunks of 8
// steps into registers (6.5 KB; 162 VGPRs, three waves per SIMD at most), a vertical or horizontal pass has 2048 lines = two waves per
// SIMD, and 13 MB in flight at a loaded-memory latency of 2-3 us is the 5 TB/s the counters show.  Here the prefetch costs no register:
// every wave owns a ring of RC chunks of 6 steps in LDS, filled by LDS-DMA (`global_load_lds_dwordx4`: per-lane global address, the
// destination is wave-linear), and the step reads its 8 + 4 bytes per lane back with ds_read.  One DMA instruction moves three
// accumulator vectors (3 x S/8 lanes x 16 B; S = vector stride, <= 160) or six cost vectors (6 x ceil(S/16) lanes).  Completion is
// tracked with counted `s_waitcnt vmcnt(N)`: VMEM operations of a wave complete in issue order, and between the DMAs of chunk c and
// the moment chunk c is needed the wave has issued exactly (RC - 1) x (NG DMAs + 6 stores) further operations (more only in the WTA
// direction, whose per-pixel output stores are extra — a larger count errs on the safe side).  No ordinary global load sits in the
// loop (the compiler would wait vmcnt(0) for it): the grey values of the next 60 steps arrive through a byte-wide DMA as well, and the
// LDS reads are inline asm the compiler cannot see, fenced by hand.  Steps past the end of a line are clamped to its last pixel
// (L2 hits), so every chunk issues the same number of operations and no guard zone is needed.
#ifdef VWGPU_RING_DBG
__constant__ int ring_dbg;     // tools build only (timing experiments, wrong results): 1 no sum stores, 2 no accumulator DMA, 4 no cost DMA, 8 no wave minimum
#define RING_DBG(bit) (dbgv & (bit))
#else
#define RING_DBG(bit) false
#endif
// CEN: no cost volume.  The wave forms the Hamming costs of a chunk itself, right before the chunk's steps: lane l holds the right census
// words of disparities 2 l and 2 l + 1 (one 16-byte load per step, requested two chunks ahead into registers; the words of neighbouring
// lines overlap, so these are L1 / L2 hits), the left word and the right word of disparity 128 arrive through the scalar cache; xor +
// v_bcnt, two cost bytes per lane written to the chunk's cost area in LDS, from where the step reads its four bytes as before (LDS
// operations of a wave complete in order).  That is 9 vector instructions per step on ALL 64 lanes instead of 19 on the 33 the recurrence
// uses, 136 of the 680 bytes a step moved, and no cost_row_kernel launch (0.45 ms and 0.6 GB written at 2048^2 x 129).  Up to 129 disparities.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int EPT, int ACC, int RC, int WPB, bool CEN>
__global__ void __launch_bounds__(64 * WPB)
path_ring_kernel(SgmGeom g, DirSet D, int stride, const uint8_t* __restrict__ left, int lw, int min_col, int min_row,
                 const uint8_t* __restrict__ cost, uint16_t* __restrict__ accum, unsigned p1, unsigned p2,
                 int32_t* __restrict__ disp, uint8_t* __restrict__ todo, int* __restrict__ any_todo, int cluster, CenArgs CA) {
  constexpr bool RMW = (ACC == ACC_RMW || ACC == ACC_RMW_WTA);
  constexpr int NGA = RMW ? 2 : 0;                                 // accumulator DMA instructions per chunk
  constexpr int NG = NGA + (CEN ? 0 : 1);                          // DMA instructions per chunk
  constexpr int CH = 6;                                            // steps per chunk
  constexpr int BLK = 60;                                          // steps per penalty block (a multiple of CH)
  typedef __attribute__((address_space(3))) uint8_t lds_u8;
  typedef __attribute__((address_space(1))) const void* gptr_t;
  extern __shared__ __attribute__((aligned(16))) uint8_t ring_raw[];
#ifdef VWGPU_RING_DBG
  const int dbgv = __builtin_amdgcn_readfirstlane(ring_dbg);
#endif
  const int num_disp = g.num_dx;
  const int npairs = (num_disp + 1) / 2;
  const int tid = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // Workgroups go round the eight XCDs (each with an L2 of its own): `cluster` consecutive groups of lines are given to ONE XCD, so the
  // 128-byte lines that the last vector of a group shares with the first of the next are fetched once as well.
  int grp = (int)blockIdx.x;
  {
    const int span = 8 * cluster, whole = ((int)gridDim.x / span) * span;
    if (cluster > 1 && grp < whole) { const int xcd = grp & 7, slot = grp >> 3; grp = (slot / cluster) * span + xcd * cluster + slot % cluster; }
  }
  const int gline = __builtin_amdgcn_readfirstlane(grp * WPB + wv);
  if (gline >= D.line0[D.n]) return;
  int dirq = 0;
  while (dirq + 1 < D.n && gline >= D.line0[dirq + 1]) ++dirq;
  dirq = __builtin_amdgcn_readfirstlane(dirq);
  const int dc = __builtin_amdgcn_readfirstlane(D.dc[dirq]), dr = __builtin_amdgcn_readfirstlane(D.dr[dirq]);
  const int ww = g.ocols, wh = g.orows;
  int c0, r0;
  {
    const int line = gline - __builtin_amdgcn_readfirstlane(D.line0[dirq]);
    const int nf = __builtin_amdgcn_readfirstlane(D.n_first[dirq]);
    if (line < nf) {
      if (__builtin_amdgcn_readfirstlane(D.row_border[dirq])) { c0 = line; r0 = dr > 0 ? 0 : wh - 1; }
      else { r0 = line; c0 = dc > 0 ? 0 : ww - 1; }
    } else {
      int i = line - nf;
      if (D.rev_second && (dc > 0) == (dr > 0)) i = wh - 2 - i;
      r0 = i + __builtin_amdgcn_readfirstlane(D.second_skip[dirq]);
      c0 = dc > 0 ? 0 : ww - 1;
    }
  }
  const int len_c = dc > 0 ? ww - c0 : (dc < 0 ? c0 + 1 : 0x7fffffff);
  const int len_r = dr > 0 ? wh - r0 : (dr < 0 ? r0 + 1 : 0x7fffffff);
  const int len = min(len_c, len_r);
  const int S = stride, q32 = S / 2;
  const int nA = S >> 3, nC = (S + 15) >> 4, CS = nC * 16;         // lanes per accumulator / cost vector of a DMA, LDS pitch of a cost vector
  // LDS of a wave: RC ring slots (6 accumulator vectors; without CEN the 6 cost vectors of the chunk behind them), CEN: ONE cost area
  // (the wave writes it right before the chunk's steps), then 256 B of grey values (the dword around each byte), a 16-byte dump slot and,
  // CEN, two sets of 6 left + 6 disparity-128 census words (16 bytes apart: the DMA moves 16 bytes per lane)
  const int ACB = 2 * CH * S, CCB = CH * CS;
  const int CHB = ACB + (CEN ? 0 : CCB);
  const int TB = RC * CHB + (CEN ? CCB : 0);
  const int WB = TB + 256 + (CEN ? 16 + 384 : 0);                     // (without CEN, 129 disparities, RC = 4: 10240 B — four workgroups of four waves per CU)
  lds_u8* const wring = (lds_u8*)ring_raw + wv * WB;
  const unsigned wbase = (unsigned)(uintptr_t)wring;
  const unsigned gbase = wbase + (unsigned)TB;

  unsigned dead[EPT], r[EPT];
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const int j = tid * EPT + e;
    dead[e] = j >= npairs ? 0xffffffffu : ((2 * j + 1 >= num_disp) ? 0xffff0000u : 0u);
    r[e] = dead[e];
  }
  const bool in = tid * EPT + EPT <= q32;
  unsigned ork0[EPT], ork1[EPT];                        // ACC_RMW_WTA: the index of each half that is a disparity of the search, all ones otherwise
#pragma unroll
  for (int e = 0; e < EPT; ++e) {
    const unsigned j2 = (unsigned)(tid * EPT + e) * 2u;
    ork0[e] = (int)j2 < num_disp ? j2 : 0xffffffffu; ork1[e] = (int)j2 + 1 < num_disp ? j2 + 1u : 0xffffffffu;
  }
  bool flagged = false;
  int last_val = 0;
  unsigned min_prior = 0;
  const long long delta = (long long)dr * g.ocols + dc;
  const long long pbase = (long long)r0 * g.ocols + c0;
  long long pcur = pbase;
  const unsigned loff_c = in ? (unsigned)(tid * EPT * 2) : 0u, loff_a = in ? (unsigned)(tid * EPT * 4) : 0u;     // bytes
  unsigned* astore = reinterpret_cast<unsigned*>(accum) + pbase * q32 + (in ? tid * EPT : 0);
  const long long astore_step = delta * q32;
  const us2 p1p1 = as_us2(p1 | (p1 << 16));
  const uint8_t* lp = left + (size_t)(r0 + min_row) * lw + (c0 + min_col);
  const int lstep = dr * lw + dc;

  // producer side: per-lane source addresses of the three DMA shapes
  const int ja = min(tid / nA, 2), ia = tid - ja * nA;              // accumulator DMA: lane -> (step of the triple, 16-byte piece)
  const int jc = min(tid / nC, CH - 1), ic = tid - jc * nC;         // cost DMA: lane -> (step of the chunk, 16-byte piece)
  const bool a_on = tid < 3 * nA, c_on = tid < CH * nC;
  const uint8_t* const abase = reinterpret_cast<const uint8_t*>(accum) + pbase * (2 * S) + ia * 16;
  const uint8_t* const cbase = cost + pbase * S + ic * 16;
  const int astepB = (int)(delta * 2 * S), cstepB = (int)(delta * S);      // |delta| <= ocols + 1, S <= 160: fits
  const int last = len - 1;
  auto issue_chunk = [&](int c) __attribute__((always_inline)) {      // chunk c -> ring slot c % RC
    lds_u8* slot = wring + (c % RC) * CHB;
    const int s0 = c * CH;
    if constexpr (RMW) {
      if (a_on && !RING_DBG(2)) {
        const uint8_t* p0 = abase + (long long)min(s0 + ja, last) * astepB;
        __builtin_amdgcn_global_load_lds((gptr_t)p0, slot, 16, 0, 0);
        const uint8_t* p1_ = abase + (long long)min(s0 + 3 + ja, last) * astepB;
        __builtin_amdgcn_global_load_lds((gptr_t)p1_, slot + 6 * S, 16, 0, 0);
      }
    }
    if constexpr (!CEN) {
      if (c_on && !RING_DBG(4)) {
        const uint8_t* pc = cbase + (long long)min(s0 + jc, last) * cstepB;
        __builtin_amdgcn_global_load_lds((gptr_t)pc, slot + ACB, 16, 0, 0);
      }
    }
  };
  // CEN: census words of a chunk -> registers (two chunks in flight; the uniform words -> LDS), costs of a chunk -> the cost area
  u32x4 cen[2][CEN ? CH : 1];
  const uint8_t* const rbase = reinterpret_cast<const uint8_t*>(CA.rcen + ((long long)r0 * CA.rcw + c0)) + tid * 16;
  const int rstepW = dr * CA.rcw + dc, lstepW = dr * CA.lcw + dc;
  // uniform words: lanes 0..5 the left word of step k = lane, lanes 6..11 the right word at disparity 128 of step k = lane - 6
  const int ju = tid < CH ? tid : min(tid - CH, CH - 1);
  const uint64_t* const ubase = tid < CH ? CA.lcen + ((long long)r0 * CA.lcw + c0) : CA.rcen + ((long long)r0 * CA.rcw + c0) + 128;
  const int ustepW = tid < CH ? lstepW : rstepW;
  const unsigned dump = gbase + 256u, ubase_lds = gbase + 272u;
  const unsigned cw_lane = 2 * tid < S ? (unsigned)(2 * tid) : 0xffffffffu;      // the lane's two cost bytes in a cost vector (or the dump slot)
  auto load_census = [&](int c, int b) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < (CEN ? CH : 0); ++k) {
      const int sk = __builtin_amdgcn_readfirstlane(min(c * CH + k, last));
      const uint8_t* pr = rbase + (long long)sk * rstepW * 8;
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(cen[b][k]) : "v"(pr) : "memory");
    }
    if constexpr (CEN) {
      if (tid < 2 * CH) {
        const uint64_t* pu = ubase + (long long)min(c * CH + ju, last) * ustepW;
        __builtin_amdgcn_global_load_lds((gptr_t)pu, wring + TB + 272 + b * 192, 16, 0, 0);
      }
    }
  };
  auto produce_costs = [&](int c, int b) __attribute__((always_inline)) {
    const unsigned cb = wbase + (unsigned)(RC * CHB);
    // the uniform words: LDS -> (the same 12 registers twice) -> scalar registers
    unsigned long long lw_[CEN ? CH : 1], xw_[CEN ? CH : 1];
    {
      unsigned long long t[CEN ? CH : 1];
#pragma unroll
      for (int k = 0; k < (CEN ? CH : 0); ++k) asm volatile("ds_read_b64 %0, %1" : "=v"(t[k]) : "v"(ubase_lds + (unsigned)(b * 192 + k * 16)) : "memory");
      if constexpr (CEN) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]) :: "memory");
#pragma unroll
      for (int k = 0; k < (CEN ? CH : 0); ++k)
        lw_[k] = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)t[k]) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(t[k] >> 32)) << 32);
#pragma unroll
      for (int k = 0; k < (CEN ? CH : 0); ++k) asm volatile("ds_read_b64 %0, %1" : "=v"(t[k]) : "v"(ubase_lds + (unsigned)(b * 192 + 96 + k * 16)) : "memory");
      if constexpr (CEN) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]) :: "memory");
#pragma unroll
      for (int k = 0; k < (CEN ? CH : 0); ++k)
        xw_[k] = (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)t[k]) | ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(t[k] >> 32)) << 32);
    }
#pragma unroll
    for (int k = 0; k < (CEN ? CH : 0); ++k) {
      const unsigned llo = (unsigned)lw_[k], lhi = (unsigned)(lw_[k] >> 32);
      const unsigned c0_ = __builtin_popcount(cen[b][k].x ^ llo) + __builtin_popcount(cen[b][k].y ^ lhi);
      const unsigned c1_ = __builtin_popcount(cen[b][k].z ^ llo) + __builtin_popcount(cen[b][k].w ^ lhi);
      const unsigned pk = c0_ | (c1_ << 8);
      const unsigned cx = (unsigned)__builtin_popcountll(xw_[k] ^ lw_[k]);
      const unsigned a2 = cw_lane == 0xffffffffu ? dump : cb + (unsigned)(k * CS) + cw_lane;
      const unsigned a1 = 128 < S ? cb + (unsigned)(k * CS) + 128u : dump + 8u;
      asm volatile("ds_write_b16 %0, %1" :: "v"(a2), "v"(pk) : "memory");
      asm volatile("ds_write_b8 %0, %1" :: "v"(a1), "v"(cx) : "memory");
    }
  };
  auto issue_grey = [&](int b) __attribute__((always_inline)) {       // grey values of steps 60 b .. 60 b + 63: the aligned dword around each
    const uint8_t* pg = lp + (long long)min(b * BLK + tid, last) * lstep;
    __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<uintptr_t>(pg) & ~(uintptr_t)3), wring + TB, 4, 0, 0);
  };
  unsigned penv = 0;
  auto refresh = [&](int b) __attribute__((always_inline)) {          // penalties of steps 60 b .. 60 b + 59 (the block's DMA completed long ago)
    unsigned pvu;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(pvu) : "v"(gbase + (unsigned)(tid * 4)) : "memory");
    const unsigned bsel = (unsigned)(reinterpret_cast<uintptr_t>(lp + (long long)min(b * BLK + tid, last) * lstep) & 3u);
    issue_grey(b + 1);                                               // (the read above has completed: the one block can be refilled)
    const int pv = (int)((pvu >> (8 * bsel)) & 0xffu);
    const int prev = (int)wave_shr1((unsigned)pv, (unsigned)last_val);
    int grad = pv - prev; grad = grad < 0 ? -grad : grad;
    unsigned v = p2 / (unsigned)max(grad, 1);
    if (v < p1) v = p1;
    penv = v & 0xffffu;
    last_val = __builtin_amdgcn_readlane(pv, min(BLK - 1, last - b * BLK));
  };
  unsigned pm = 0xffffffffu, pn = 0xffffffffu;
  auto step = [&](unsigned cw, const unsigned (&aw)[EPT], int k) __attribute__((always_inline)) {      // k: step within the penalty block
    const unsigned pen = (unsigned)__builtin_amdgcn_readlane((int)penv, k);
    const unsigned dj = (min_prior + pen) & 0xffffu;
    const us2 dJ = as_us2(dj | (dj << 16)), mp = as_us2(min_prior | (min_prior << 16));
    wave_shr1_keep(pm, r[EPT - 1]);
    wave_shl1_keep(pn, r[0]);
    unsigned al[EPT + 1];
    al[0] = __builtin_amdgcn_alignbit(r[0], pm, 16);
#pragma unroll
    for (int e = 1; e < EPT; ++e) al[e] = __builtin_amdgcn_alignbit(r[e], r[e - 1], 16);
    al[EPT] = __builtin_amdgcn_alignbit(pn, r[EPT - 1], 16);
    us2 mn2 = as_us2(0xffffffffu);
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const us2 ctr = as_us2(r[e]);
      us2 m = __builtin_elementwise_min(__builtin_elementwise_min(as_us2(al[e]), as_us2(al[e + 1])), ctr);
      us2 v = __builtin_elementwise_add_sat(m, p1p1);
      v = __builtin_elementwise_min(v, __builtin_elementwise_min(ctr, dJ));
      const unsigned cpair = __builtin_amdgcn_perm(0u, cw, (e & 1) ? 0x0c030c02u : 0x0c010c00u);      // (c_2j, c_2j+1) as two u16
      v = __builtin_elementwise_add_sat(v, as_us2(cpair));
      v = __builtin_elementwise_sub_sat(v, mp);
      r[e] = as_u32(v) | dead[e];
      mn2 = e == 0 ? as_us2(r[e]) : __builtin_elementwise_min(mn2, as_us2(r[e]));
    }
    if constexpr (ACC != ACC_NONE) {
      unsigned o[EPT];
#pragma unroll
      for (int e = 0; e < EPT; ++e) o[e] = RMW ? as_u32(as_us2(aw[e]) + as_us2(r[e])) : r[e];
      if constexpr (ACC == ACC_RMW_WTA) {
        // keys (value << 16 | index); a half that is no disparity of the search has all ones OR-ed in (a per-lane constant: no select)
        unsigned kk = 0xffffffffu;
#pragma unroll
        for (int e = 0; e < EPT; ++e) kk = min(kk, min((o[e] << 16) | ork0[e], (o[e] & 0xffff0000u) | ork1[e]));
        const unsigned key = wave_min_u32_fused(kk), mv = key >> 16;
        // Ties: the count may include a padding half that happens to hold the minimum's value (the sums of the padding are arbitrary) —
        // that sends a pixel with a unique minimum to wta_kernel, which finds the same minimum.
        int cnt = 0;
#pragma unroll
        for (int e = 0; e < EPT; ++e)
          cnt += __popcll(__ballot((o[e] & 0xffffu) == mv)) + __popcll(__ballot((o[e] >> 16) == mv));
        if (tid == 0) {
          if (cnt > 1) { todo[pcur] = 1; }
          else {
            todo[pcur] = 0;
            int32_t* w3 = disp + pcur * 3;
            w3[0] = (int)(key & 0xffffu) + g.min_dx; w3[1] = g.min_dy; w3[2] = 0x7fffffff;
          }
        }
        flagged |= cnt > 1;
        pcur += delta;
      }
      if (in && !RING_DBG(1)) {
        if constexpr (EPT == 2) *reinterpret_cast<uint2*>(astore) = make_uint2(o[0], o[1]);
        else astore[0] = o[0];
      }
    }
    astore += astore_step;
    const unsigned mnu = as_u32(mn2);
    if (!RING_DBG(8)) min_prior = wave_min_u32_fused(min(mnu & 0xffffu, mnu >> 16));
  };
  // one chunk: its 6 steps' bytes from the ring slot (all reads requested together, waited for step by step), then the steps
  auto run_chunk = [&](int c, int kblk, int nsteps) __attribute__((always_inline)) {
    const unsigned sb = wbase + (unsigned)((c % RC) * CHB);
    unsigned cw[CH];
    typedef typename std::conditional<EPT == 2, unsigned long long, unsigned>::type acc_t;
    acc_t a64[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      if constexpr (RMW) {
        const unsigned aa = sb + (unsigned)(k * 2 * S) + loff_a;
        if constexpr (EPT == 2) asm volatile("ds_read_b64 %0, %1" : "=v"(a64[k]) : "v"(aa) : "memory");
        else asm volatile("ds_read_b32 %0, %1" : "=v"(a64[k]) : "v"(aa) : "memory");
      } else a64[k] = 0;
      const unsigned ca = (CEN ? wbase + (unsigned)(RC * CHB) : sb + (unsigned)ACB) + (unsigned)(k * CS) + loff_c;
      if constexpr (EPT == 2) asm volatile("ds_read_b32 %0, %1" : "=v"(cw[k]) : "v"(ca) : "memory");
      else asm volatile("ds_read_u16 %0, %1" : "=v"(cw[k]) : "v"(ca) : "memory");
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      // LDS operations of a wave complete in order: the reads of steps k + 1 .. 5 may still be under way
      if constexpr (RMW) {
        switch (k) {
          case 0: asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(cw[0]), "+v"(a64[0]) :: "memory"); break;
          case 1: asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(cw[1]), "+v"(a64[1]) :: "memory"); break;
          case 2: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(cw[2]), "+v"(a64[2]) :: "memory"); break;
          case 3: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(cw[3]), "+v"(a64[3]) :: "memory"); break;
          case 4: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(cw[4]), "+v"(a64[4]) :: "memory"); break;
          default: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cw[5]), "+v"(a64[5]) :: "memory"); break;
        }
      } else if (k == 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cw[0]), "+v"(cw[1]), "+v"(cw[2]), "+v"(cw[3]), "+v"(cw[4]), "+v"(cw[5]) :: "memory");
      }
      unsigned aw[EPT];
      if constexpr (EPT == 2) { aw[0] = (unsigned)a64[k]; aw[1] = (unsigned)(a64[k] >> 32); } else aw[0] = (unsigned)a64[k];
      if (k < nsteps) step(cw[k], aw, kblk + k);
    }
  };

  // prologue: the first penalty block, (CEN: the census words of two chunks,) then RC chunks in flight
  issue_grey(0);
  if constexpr (CEN) { load_census(0, 0); load_census(1, 1); }
#pragma unroll
  for (int c = 0; c < RC; ++c) issue_chunk(c);
  const int nfull = len / CH;
  int c = 0, kblk = 0, blk = 0;
  // Operations a wave issues per chunk in the steady state, in this order: (CEN: 6 census loads of chunk c + 2,) 6 stores, NG DMAs of chunk
  // c + RC.  Chunk c needs its DMAs (issued RC chunks ago) and, CEN, its census words (issued two chunks ago, BEFORE that chunk's stores):
  // everything but the operations younger than those may still be under way.
  constexpr int STORES = ACC == ACC_NONE ? 0 : CH;
  constexpr int N_STEADY = CEN ? CH + 1 + 2 * (STORES + NG) : (RC - 1) * (NG + STORES);  // CEN: stores + DMAs of c - 2, census (6 + 1) + stores + DMAs of c - 1
  constexpr int N_EARLY = CEN ? 0 : (RC - 1) * NG;                   // the first chunks: fewer stores lie in between — wait as if there were none
  constexpr int C_EARLY = CEN ? RC : RC - 1;
  static_assert(RC >= 3, "the census words of chunk c are requested after the DMAs of chunk c");
  // one chunk; B = its census register set (compile time: the sets must stay registers), MODE = 0 one of the first chunks, 1 steady state,
  // 2 the last, partial chunk
  auto chunk = [&](auto Bc, auto MODEc, int nsteps) __attribute__((always_inline)) {
    constexpr int B = decltype(Bc)::value;
    constexpr int MODE = decltype(MODEc)::value;
    constexpr int N = MODE == 2 ? 0 : MODE == 0 ? N_EARLY : (N_STEADY > 63 ? 63 : N_STEADY);
    if constexpr (CEN) asm volatile("s_waitcnt vmcnt(%6)" : "+v"(cen[B][0]), "+v"(cen[B][1]), "+v"(cen[B][2]), "+v"(cen[B][3]), "+v"(cen[B][4]), "+v"(cen[B][5]) : "n"(N) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
    if (kblk == BLK) { kblk = 0; ++blk; }
    if (kblk == 0) refresh(blk);
    if constexpr (CEN) {
      produce_costs(c, B);
      if constexpr (MODE != 2) load_census(c + 2, B);
    }
    run_chunk(c, kblk, nsteps);
    if constexpr (MODE != 2) issue_chunk(c + RC);
    kblk += CH;
    ++c;
  };
  typedef std::integral_constant<int, 0> I0;
  typedef std::integral_constant<int, 1> I1;
  typedef std::integral_constant<int, 2> I2;
  if constexpr (CEN) {
    const int ce = min(nfull & ~1, (C_EARLY + 1) & ~1);
    while (c < ce) { chunk(I0(), I0(), CH); chunk(I1(), I0(), CH); }
    while (c + 2 <= nfull) { chunk(I0(), I1(), CH); chunk(I1(), I1(), CH); }
    if (c < nfull) {
      chunk(I0(), I0(), CH);
      if (c * CH < len) chunk(I1(), I2(), len - c * CH);
    } else if (c * CH < len) chunk(I0(), I2(), len - c * CH);
  } else {
    const int ce = min(nfull, C_EARLY);
    while (c < ce) chunk(I0(), I0(), CH);
    while (c < nfull) chunk(I0(), I1(), CH);
    if (c * CH < len) chunk(I0(), I2(), len - c * CH);
  }
  if constexpr (CEN) {
    // The census words requested last are never used, and a register the compiler takes for dead is free for anything else — while its load
    // is still under way.  Every register set stays allocated until the loads have landed:
    asm volatile("s_waitcnt vmcnt(0)" :: "v"(cen[0][0]), "v"(cen[0][1]), "v"(cen[0][2]), "v"(cen[0][3]), "v"(cen[0][4]), "v"(cen[0][5]),
                 "v"(cen[1][0]), "v"(cen[1][1]), "v"(cen[1][2]), "v"(cen[1][3]), "v"(cen[1][4]), "v"(cen[1][5]) : "memory");
  }
  if constexpr (ACC == ACC_RMW_WTA) {
    if (flagged && tid == 0) atomicOr(any_todo, 1);
  }
}


// ---- fused raster sweeps (round 3) ------------------------------------------------------------------------------------------
// One direction per launch touches the u16 sums eight times (27.7 GB per 2048^2 x 129 call, measured).  Here the eight directions
// run as TWO sweeps over the image that proceed CONCURRENTLY:
//   forward  (rows top -> bottom, pixels left -> right):  L->R, T->B, TL->BR, TR->BL
//   backward (rows bottom -> top, pixels right -> left):  R->L, B->T, BR->TL, BL->TR
// Every pixel is visited once per sweep; the four path vectors of the visit are summed in registers and stored once, into a
// volume per sweep (the winner-take-all kernels read both): cost read twice (2 x 1 B), sums written twice, read once (3 x 2 B
// per disparity) instead of 8 x (1 + 2 + 2) B.
//
// Mapping: one wavefront per image ROW and sweep, marching along the row.  The four directions of a pixel are the four DPP ROWS
// of the wavefront (16 lanes each: row 0 = along the row, rows 1..3 = from above, from above-left, from above-right), lane j of a
// row owns Q consecutive disparity pairs, so ONE packed-u16 evaluation (the arithmetic of path_uniform_reg_kernel) serves all four
// directions: the d-1 / d+1 neighbours are `row_shr:1` / `row_shl:1` moves (a row's first / last lane keeps its 0xffff guard), the
// four minima are one four-stage row reduction, the thresholds are per-lane values.  A step is ~110 vector instructions, and the
// step time IS the run time: row y must stay two pixels behind row y-1 (it needs the vectors of (x, y-1), (x-1, y-1), (x+1, y-1)),
// so all rows of the image are in flight at once, skewed, and a sweep takes W + 2 H pixel steps.
// The "from above" vectors travel through an LDS ring per row (producer wave = row y-1, consumer wave = row y of the same
// workgroup, progress counters in LDS, no barrier).  A workgroup holds NW consecutive rows; its first row takes the vectors of the
// previous workgroup's last row from HBM: that row stores every 32-bit word as (word, epoch) in one 64-bit coherent store, and a
// FEEDER wave of the consuming workgroup polls the entries with coherent loads (an entry is complete when every word carries this
// launch's epoch — no fence, no cache invalidation) and refills an LDS ring for row wave 0.  Workgroups take their (sweep, block)
// from a ticket counter, so a block's predecessor has always started before it: nothing can wait on work that is not yet scheduled.
// A line's first pixel starts from r = 0, min_prior = 0 (see path_uniform_reg_kernel): the general step then yields the plain cost.
//
// Measured (MI355X, 2048^2 x 129): identical sums, a fifth of the HBM traffic — and 8.0 ms against 5.5 ms for the eight
// per-direction launches.  The pixel step of a row wave takes ~1 us (130 vector + ~100 other instructions, a dependent chain, 13
// co-resident rows of a workgroup sharing 4 SIMDs), and the sweep needs W + 2 H of them in sequence, while the per-direction kernel
// keeps 2048 .. 4101 independent lines in flight and is bandwidth bound.  So the sweeps are an OPTION (VWGPU_OPT_SGM_SWEEP), not
// the default schedule.
#ifndef VWGPU_SWEEP_KC
#define VWGPU_SWEEP_KC 8
#endif
constexpr int SWEEP_RING = 8, SWEEP_FRING = 16, SWEEP_FB = 8, SWEEP_KC = VWGPU_SWEEP_KC;

struct SweepParams {
  SgmGeom g;
  int stride, lw, min_col, min_row, nblk, nw;
  unsigned p1, p2, epoch;
  const uint8_t* left;
  const uint8_t* cost;
  uint16_t* out0;
  uint16_t* out1;
  unsigned* sync;                    // [0] ticket counter
  unsigned long long* bnd;           // boundary rows: [sweep][block][pixel][3 vectors + minima] of (word, epoch)
#ifdef VWGPU_SWEEP_DEBUG
  int dbg;                           // tools build only (timing experiments, wrong results): 1 no boundary stores, 2 feeder trusts every word, 4 rows do not wait, 8 no sum stores
#endif
};

typedef __attribute__((address_space(3))) unsigned lds_uint;
#ifdef VWGPU_SWEEP_DEBUG
__device__ unsigned long long sweep_trace[2 * 2048 * 8];          // tools build: [sweep][block][event] timestamps (s_memrealtime, 100 MHz)
#define SWEEP_STAMP(ev) do { if (tid == 0 && blk < 2048) sweep_trace[((size_t)pass * 2048 + blk) * 8 + (ev)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SWEEP_STAMP(ev) do {} while (0)
#endif

__device__ __forceinline__ unsigned long long ld_coherent(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_coherent(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// lane j of a DPP row <- lane j -+ 1 of the same row; the row's first / last lane keeps dst (bound_ctrl off): its guard
__device__ __forceinline__ void row_shr1_keep(unsigned& dst, unsigned src) {
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(dst) : "v"(src));
}
__device__ __forceinline__ void row_shl1_keep(unsigned& dst, unsigned src) {
  asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(dst) : "v"(src));
}
// minimum over each DPP row of 16 lanes; every lane of a row ends up with its row's minimum
__device__ __forceinline__ unsigned row_min_u32(unsigned v) {
  asm volatile("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
               "s_nop 1"
               : "+v"(v));
  return v;
}
template <int N> struct __attribute__((packed, aligned(4))) SweepWords { unsigned v[N]; };


// The feeder wave of a sweep workgroup: the boundary row of the previous block (W entries of NWORDS tagged words) -> the LDS ring of
// row wave 0.  An entry is complete when every word carries this launch's epoch.
template <int SLOT, int NWORDS>
__device__ __forceinline__ void sweep_feeder(const unsigned long long* __restrict__ src, int W, lds_uint* ring, volatile lds_uint* done, int NW,
                                             unsigned epoch, int tid
#ifdef VWGPU_SWEEP_DEBUG
                                             , int dbg, int pass, int blk
#endif
                                             ) {
  constexpr int FRING = SWEEP_FRING, PER = (NWORDS + 63) / 64;
  constexpr int FB = PER <= 4 ? SWEEP_FB : SWEEP_FB / 2;            // entries requested at once: 32 / 28 loads in flight per lane
  const size_t ENT = SLOT;
  int e = 0;
  unsigned cons = 0;
  while (e < W) {
    const int B = min(FB, W - e);
    const int needc = e + B - FRING + 2;                           // row wave 0 must be done with the slots about to be overwritten
    while ((int)cons < needc) {
#ifdef VWGPU_SWEEP_DEBUG
      if (dbg & 4) break;
#endif
      cons = done[0];
      if ((int)cons < needc) __builtin_amdgcn_s_sleep(2);
    }
    unsigned long long wd[FB][PER];
#pragma unroll
    for (int b = 0; b < FB; ++b) {
      const unsigned long long* ent = src + (size_t)min(e + b, W - 1) * ENT;
#pragma unroll
      for (int k = 0; k < PER; ++k) wd[b][k] = ld_coherent(ent + min(k * 64 + tid, NWORDS - 1));
    }
    int k = 0;                                                     // entries e .. e + k - 1 are complete
#pragma unroll
    for (int b = 0; b < FB; ++b) {
      bool ok = true;
#pragma unroll
      for (int q = 0; q < PER; ++q) ok = ok && (unsigned)(wd[b][q] >> 32) == epoch;
#ifdef VWGPU_SWEEP_DEBUG
      if (dbg & 2) ok = true;
#endif
      if (__all(ok) && b < B && k == b) k = b + 1;
    }
#pragma unroll
    for (int b = 0; b < FB; ++b) {
      if (b < k) {                                                 // wave-uniform
        lds_uint* slot = ring + ((e + b) & (FRING - 1)) * SLOT;
#pragma unroll
        for (int q = 0; q < PER; ++q)
          if (q * 64 + tid < NWORDS) slot[q * 64 + tid] = (unsigned)wd[b][q];
      }
    }
    if (k) {
#ifdef VWGPU_SWEEP_DEBUG
      if (e == 0 && tid == 0 && blk < 2048) sweep_trace[((size_t)pass * 2048 + blk) * 8 + 4] = __builtin_amdgcn_s_memrealtime();
#endif
      asm volatile("" ::: "memory");                               // LDS executes a wave's accesses in order: data, then the counter
      done[NW] = (unsigned)(e + k);
      e += k;
    } else {
      __builtin_amdgcn_s_sleep(8);
    }
  }
#ifdef VWGPU_SWEEP_DEBUG
  if (tid == 0 && blk < 2048) sweep_trace[((size_t)pass * 2048 + blk) * 8 + 5] = __builtin_amdgcn_s_memrealtime();
#endif
}

template <int Q>
__global__ void __launch_bounds__(1024)
sweep_uniform_kernel(SweepParams A) {
  constexpr int RING = SWEEP_RING, FRING = SWEEP_FRING, KC = SWEEP_KC;
  constexpr int VQ = 16 * Q;                                       // dwords of a vector in a slot: every lane of a row addresses its own Q
  constexpr int SLOT = 3 * VQ + 4;                                 // from above, from above-left, from above-right + their three minima
  constexpr int NWORDS = 3 * VQ + 4;                              // (the pad word is tagged too)
  constexpr int NWC = (Q + 1) / 2;                                 // dwords that hold a lane's 2 Q cost bytes (from a dword-aligned address)
  extern __shared__ unsigned lds_u32[];
  lds_uint* const sm = (lds_uint*)lds_u32;                         // every LDS access through address-space-3 pointers: ds_* instructions, lgkmcnt only
  const int tid = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int NW = A.nw;                                             // row waves 0 .. NW-1; wave NW is the feeder
  const int W = A.g.ocols, H = A.g.orows, stride = A.stride, q32 = stride / 2;
  const size_t ENT = SLOT;                                         // 64-bit words of a boundary entry (same order as a slot)
  volatile lds_uint* done = sm + 1;                                // [i]: pixels completed by row wave i; [NW]: entries delivered by the feeder
  lds_uint* rings = sm + 32;
  if (threadIdx.x == 0) sm[0] = atomicAdd(&A.sync[0], 1u);
  if (threadIdx.x >= 1 && threadIdx.x < 32) sm[threadIdx.x] = 0u;
  __syncthreads();
  const unsigned ticket = sm[0];
  const int pass = (int)(ticket & 1u), blk = (int)(ticket >> 1);
  const unsigned epoch = A.epoch;

  if (wv == NW) {                                                  // ---- feeder: boundary row of the previous block -> LDS ring of row wave 0
    if (blk > 0) sweep_feeder<SLOT, NWORDS>(A.bnd + ((size_t)(pass * A.nblk + blk - 1) * W) * ENT, W, rings + NW * RING * SLOT, done, NW, epoch, tid
#ifdef VWGPU_SWEEP_DEBUG
                                            , A.dbg, pass, blk
#endif
                                            );
    return;
  }

  // ---- a row wave
  const int yy = blk * NW + wv;                                    // row in sweep order
  if (yy >= H) return;
  const int sgn = pass ? -1 : 1;
  const int y = pass ? H - 1 - yy : yy;
  const bool has_above = yy > 0;
  const int mode = yy + 1 >= H ? 0 : (wv < NW - 1 ? 1 : 2);      // where this row's vectors go: nowhere / LDS ring / HBM boundary row
  const int aidx = wv > 0 ? wv - 1 : NW;
  const int AMASK = (wv > 0 ? RING : FRING) - 1;
  const lds_uint* aring = rings + aidx * RING * SLOT;
  volatile lds_uint* adone = done + aidx;
  lds_uint* oring = rings + wv * RING * SLOT;
  unsigned long long* bdst = A.bnd + ((size_t)(pass * A.nblk + blk) * W) * ENT;

  const int row = tid >> 4, j = tid & 15;                          // DPP row = direction: 0 along the row, 1 from above, 2 above-left, 3 above-right
  const int dl = row == 2 ? -1 : (row == 3 ? 1 : 0);              // pixel of the row above this direction continues from
  const int voff = (row > 0 ? (row - 1) * VQ : 0) + j * Q;        // the lane's dwords inside a slot / boundary entry
  const int moff = 3 * VQ + (row > 0 ? row - 1 : 0);              // the direction's minimum inside a slot
  const int num_disp = A.g.num_dx, npairs = (num_disp + 1) / 2;
  unsigned dead[Q], r[Q];
#pragma unroll
  for (int e = 0; e < Q; ++e) {
    const int p = j * Q + e;
    dead[e] = p >= npairs ? 0xffffffffu : ((2 * p + 1 >= num_disp) ? 0xffff0000u : 0u);
    r[e] = dead[e];
  }
  unsigned mp = 0;                                                 // min_prior of the lane's direction
  const bool st_full = row == 0 && j * Q + Q <= q32, st_part = row == 0 && j * Q < q32 && !st_full;
  const long long p0 = (long long)y * W + (pass ? W - 1 : 0);     // the row's first pixel in sweep order
  const unsigned cb0 = (unsigned)(2 * Q * j);
  const uint8_t* cfetch = A.cost + p0 * stride + (cb0 & ~3u);
  const unsigned csh = (cb0 & 2u) * 8u;                            // odd Q: every other lane's bytes start in the middle of a dword
  const long long cstep = (long long)sgn * stride;
  unsigned* ostore = reinterpret_cast<unsigned*>(pass ? A.out1 : A.out0) + p0 * q32 + j * Q;
  const long long ostep = (long long)sgn * q32;
  const us2 p1p1 = as_us2(A.p1 | (A.p1 << 16));
  const unsigned p1 = A.p1, p2 = A.p2;
  const uint8_t* rowp = A.left + (size_t)(y + A.min_row) * A.lw + A.min_col;
  const uint8_t* rowa = has_above ? rowp - (long long)sgn * A.lw : rowp;
  const uint8_t* prow = row == 0 ? rowp : rowa;                    // the row of the lane's predecessor pixel

  // grey values of a block of 16 steps (lane j of every row <-> step s0 + j) and of the direction's predecessor, fetched a block ahead
  int gc, gp;
  auto grey = [&](int s0) __attribute__((always_inline)) {
    const int xs = min(s0 + j, W - 1);
    const int x = pass ? W - 1 - xs : xs;
    const int xl = xs > 0 ? x - sgn : x, xr = xs < W - 1 ? x + sgn : x;
    gc = rowp[x];
    gp = prow[row == 1 ? x : (row == 3 ? xr : xl)];
  };
  grey(0);
  unsigned penl = 0;
  auto refresh = [&](int s0) __attribute__((always_inline)) {      // s0 % 16 == 0
    int grad = gc - gp; grad = grad < 0 ? -grad : grad;
    unsigned v = p2 / (unsigned)max(grad, 1);
    if (v < p1) v = p1;
    penl = v & 0xffffu;
    grey(min(s0 + 16, W - 1) & ~15);
  };
  unsigned pm = 0xffffffffu, pn = 0xffffffffu;
  unsigned avail = 0, cavail = 0;
  const int bp_pen = (tid & 48) << 2, bp_x32 = (tid ^ 32) << 2;

  // The sum of a pixel's four directions crosses the DPP rows through the LDS crossbar (ds_swizzle lane ^ 16, ds_bpermute lane ^ 32):
  // two dependent round trips.  They are issued for the PREVIOUS pixel at the start of a step, next to the poll of the row above,
  // and consumed after the evaluation — off every critical path; the pixel's sum is stored one step late (flush after the loop).
  unsigned rs[Q];                                                  // the previous pixel's four vectors (one per DPP row)
  unsigned sw[Q];                                                  // ... and their lane ^ 16 partners, in flight
  bool have_prev = false;
  auto sum_issue = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < Q; ++e) sw[e] = (unsigned)__builtin_amdgcn_ds_swizzle((int)rs[e], 0x401f);
  };
  unsigned sa[Q], sb[Q];                                           // half sums, and their lane ^ 32 partners in flight
  auto sum_mid = [&]() __attribute__((always_inline)) {            // after the step's first LDS wait: the swizzles have arrived
#pragma unroll
    for (int e = 0; e < Q; ++e) {
      sa[e] = as_u32(as_us2(rs[e]) + as_us2(sw[e]));               // u16 wrap-around, the reference's +=
      sb[e] = (unsigned)__builtin_amdgcn_ds_bpermute(bp_x32, (int)sa[e]);
    }
  };
  auto sum_finish = [&]() __attribute__((always_inline)) {
    unsigned o[Q];
#pragma unroll
    for (int e = 0; e < Q; ++e) o[e] = as_u32(as_us2(sa[e]) + as_us2(sb[e]));
#ifdef VWGPU_SWEEP_DEBUG
    if (A.dbg & 8) {} else
#endif
    if (st_full) {
      SweepWords<Q> w;
#pragma unroll
      for (int e = 0; e < Q; ++e) w.v[e] = o[e];
      *reinterpret_cast<SweepWords<Q>*>(ostore) = w;
    } else if (st_part) {                                          // the lane whose pairs straddle the end of the pixel's vector
#pragma unroll
      for (int e = 0; e < Q; ++e)
        if (j * Q + e < q32) ostore[e] = o[e];
    }
    ostore += ostep;
  };

  auto step = [&](const unsigned (&cw)[NWC], int xx) __attribute__((always_inline)) {
    // (0) crossbar requests that need nothing from this step: the previous pixel's partial sums, this pixel's threshold
    if (have_prev) sum_issue();
    const unsigned pen = (unsigned)__builtin_amdgcn_ds_bpermute(bp_pen + ((xx & 15) << 2), (int)penl);
    // (1) the vectors this pixel continues: row 0 keeps its own, rows 1 .. 3 take them from the row above
    if (has_above) {                                               // wave-uniform
      const int need = min(xx + 2, W);
      const int sl = ((xx + dl) & AMASK) * SLOT;
      unsigned t[Q], tm = 0;
      for (;;) {
        // the counter first, then the slots (LDS serves a wave's requests in order): if the counter is high enough, so are the slots
        const unsigned got = avail >= (unsigned)need ? avail : *adone;
        asm volatile("" ::: "memory");
        if (row > 0) {
#pragma unroll
          for (int e = 0; e < Q; ++e) t[e] = aring[sl + voff + e];
          tm = aring[sl + moff];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        avail = (unsigned)__builtin_amdgcn_readfirstlane((int)got);
#ifdef VWGPU_SWEEP_DEBUG
        if (A.dbg & 4) break;
#endif
        if (avail >= (unsigned)need) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (row > 0) {
#pragma unroll
        for (int e = 0; e < Q; ++e) r[e] = t[e];
        mp = tm;
      }
      if (xx == 0 || xx + 1 == W) {                                // the diagonal lines that start at this pixel
        if ((row == 2 && xx == 0) || (row == 3 && xx + 1 == W)) {
#pragma unroll
          for (int e = 0; e < Q; ++e) r[e] = dead[e];
          mp = 0;
        }
      }
    } else if (row > 0) {                                          // first row of the sweep: every "from above" line starts here
#pragma unroll
      for (int e = 0; e < Q; ++e) r[e] = dead[e];
      mp = 0;
    }
    if (have_prev) sum_mid();
    // (2) thresholds of the lane's direction
    const unsigned dj = (mp + pen) & 0xffffu;
    const us2 dJ = as_us2(dj | (dj << 16)), mpp = as_us2(mp | (mp << 16));
    // (3) the cost pairs of the lane
    unsigned cs[NWC];
#pragma unroll
    for (int n = 0; n < NWC; ++n) cs[n] = (Q & 1) ? __builtin_amdgcn_alignbit(n + 1 < NWC ? cw[n + 1] : 0u, cw[n], csh) : cw[n];
    // (4) one evaluation for the four directions (SGM.cc:936-984,1013-1060 on packed u16 pairs)
    row_shr1_keep(pm, r[Q - 1]);
    row_shl1_keep(pn, r[0]);
    unsigned al[Q + 1];
    al[0] = __builtin_amdgcn_alignbit(r[0], pm, 16);
#pragma unroll
    for (int e = 1; e < Q; ++e) al[e] = __builtin_amdgcn_alignbit(r[e], r[e - 1], 16);
    al[Q] = __builtin_amdgcn_alignbit(pn, r[Q - 1], 16);
    us2 mn2 = as_us2(0xffffffffu);
#pragma unroll
    for (int e = 0; e < Q; ++e) {
      const us2 ctr = as_us2(r[e]);
      us2 m = __builtin_elementwise_min(__builtin_elementwise_min(as_us2(al[e]), as_us2(al[e + 1])), ctr);
      us2 v = __builtin_elementwise_add_sat(m, p1p1);
      v = __builtin_elementwise_min(v, __builtin_elementwise_min(ctr, dJ));
      v = __builtin_elementwise_add_sat(v, as_us2(__builtin_amdgcn_perm(0u, cs[e >> 1], (e & 1) ? 0x0c030c02u : 0x0c010c00u)));
      v = __builtin_elementwise_sub_sat(v, mpp);
      r[e] = as_u32(v) | dead[e];
      mn2 = e == 0 ? as_us2(r[e]) : __builtin_elementwise_min(mn2, as_us2(r[e]));
    }
    const unsigned mnu = as_u32(mn2);
    const unsigned nm = row_min_u32(min(mnu & 0xffffu, mnu >> 16));  // every lane: the minimum of its direction
    // (5) hand the three "from above" vectors on — before anything else: the row below is waiting for them
    if (mode == 1) {                                               // to the row below, through this wave's LDS ring
      const int needc = xx - RING + 2;
      while ((int)cavail < needc) {
#ifdef VWGPU_SWEEP_DEBUG
        if (A.dbg & 4) break;
#endif
        cavail = done[wv + 1];
        if ((int)cavail < needc) __builtin_amdgcn_s_sleep(1);
      }
      lds_uint* slot = oring + (xx & (RING - 1)) * SLOT;
      if (row > 0) {
#pragma unroll
        for (int e = 0; e < Q; ++e) slot[voff + e] = r[e];
        if (j == 0) slot[moff] = nm;
      }
    } else if (mode == 2
#ifdef VWGPU_SWEEP_DEBUG
               && !(A.dbg & 1)
#endif
               ) {                                                 // to the next block, through HBM: every word with this launch's epoch
      unsigned long long* ent = bdst + (size_t)xx * ENT;
      const unsigned long long tag = (unsigned long long)epoch << 32;
      if (row > 0) {
#pragma unroll
        for (int e = 0; e < Q; ++e) st_coherent(ent + voff + e, tag | r[e]);
        if (j < 2) st_coherent(ent + moff + (j ? 4 - row : 0), tag | nm);   // (lane 1 of row 1 tags the entry's pad word: the feeder reads word pairs)
      }
    }
    // progress of this row: the row below reads it before it takes the slots, the row above (or the feeder) before it reuses them —
    // every row publishes it, also the ones whose vectors go to HBM or nowhere
    asm volatile("" ::: "memory");                                 // LDS executes a wave's accesses in order: slot data, then the counter
    done[wv] = (unsigned)(xx + 1);
    // (6) the previous pixel's sum leaves; this pixel's vectors wait for the next step
    if (have_prev) sum_finish();
#pragma unroll
    for (int e = 0; e < Q; ++e) rs[e] = r[e];
    have_prev = true;
    if (row == 0) mp = nm;                                         // (rows 1 .. 3 reload theirs)
  };

  // The cost words of the next KC steps wait in registers: the loop is unrolled KC times so that step k of a round uses cq[k] and
  // refills it for the round after.
  unsigned cq[KC][NWC];
  auto load_cost = [&](unsigned (&w)[NWC]) __attribute__((always_inline)) {
    // one dword per load on purpose: a dwordx3 lands in a register triple that the loop-carried cq[k][n] cannot be, and the copies
    // out of the triple sit right behind the load — every round waited for the load it had just issued
#pragma unroll
    for (int n = 0; n < NWC; ++n) {
      unsigned off = 4u * n;
      asm("" : "+v"(off));                                         // (keeps the vectoriser from seeing three adjacent loads)
      w[n] = *reinterpret_cast<const unsigned*>(cfetch + off);
    }
    cfetch += cstep;                                               // runs up to 2 KC steps past the row: guard zones (host side)
  };
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    load_cost(cq[k]);
    // pinned in this order: the waits of the loop count the loads issued AFTER the one a step needs, on every path into the loop
    __builtin_amdgcn_sched_barrier(0);
  }
  if (wv == 0) SWEEP_STAMP(0);
  if (mode != 1) SWEEP_STAMP(2);
  for (int x0 = 0; x0 < W; x0 += KC) {
    if ((x0 & 15) == 0) refresh(x0);
    if (x0 == KC) { if (wv == 0) SWEEP_STAMP(6); if (mode != 1) SWEEP_STAMP(7); }
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      if (x0 + k < W) step(cq[k], x0 + k);
      load_cost(cq[k]);                                            // after the step's last use of cq[k]: the refill lands in the same registers
    }
  }
  sum_issue();                                                     // the row's last pixel
  sum_mid();
  sum_finish();
  if (wv == 0) SWEEP_STAMP(1);
  if (mode != 1) SWEEP_STAMP(3);
}


// ---- MGM as four concurrent sweeps (round 3) ----------------------------------------------------------------------------------------
// accum_mgm_multithread (SGM.cc:2619-2700): eight passes in which a pixel takes the mean of TWO evaluate_path results of the pass's
// own values, from a path predecessor A and a second predecessor B (mgm_schedule.h).  One launch per front made that 2035 launches
// for a 1024^2 image (5.5 us each).  In a suitable frame (u = marching axis, v = line index) every pass is one of two shapes:
//     axis      A = (u-1, v), B = (u, v-1)         L in (x, y), R in (-x, -y), T in (y, -x), B in (-y, x)
//     diagonal  A = (u-1, v-1), B = (u+1, v-1)     TL           BR             TR            BL          (same frames)
// so the eight passes are FOUR sweeps of the sweep_uniform_kernel kind — one wavefront per line v marching along u, two pixels
// behind the line before it, vectors handed on through LDS rings (and through tagged HBM rows between workgroups) — that run
// concurrently, each carrying one axis and one diagonal pass in the four DPP rows of its wavefronts:
//     row 0  axis pass from A (the wavefront's own previous pixel)      row 2  diagonal pass from A (line v-1, pixel u-1)
//     row 1  axis pass from B (line v-1, pixel u)                       row 3  diagonal pass from B (line v-1, pixel u+1)
// One packed evaluation per pixel serves the four, the means are taken between DPP rows (ds_swizzle lane ^ 16), the pass values go
// to the per-direction volumes that mgm_sum_kernel adds to the sums.  Border tests (SGMAssist.h:911-1219) in frame coordinates:
// axis u > 0 and v > 0, diagonal 0 < u < U-1 and v > 0; a pixel that fails it keeps its local costs (both predecessors "none").
struct MgmSweepParams {
  SgmGeom g;
  int stride, lw, lh, min_col, min_row, nw;
  int nblk[4], blk0[5];              // blocks of NW lines per sweep; blk0 = prefix sums (tickets are dealt round robin over the sweeps)
  unsigned p1, p2, epoch;
  const uint8_t* left;
  const uint8_t* cost;
  uint16_t* vols;                    // per-direction volumes, kMgmDirs order: L, TL, R, BR, T, BL, B, TR
  size_t vol_stride;                 // u16 elements between two volumes
  unsigned* sync;
  unsigned long long* bnd;           // boundary rows of all sweeps, block-major: entry = 2 vectors + 2 minima of (word, epoch)
  size_t bnd_off[4];                 // first entry of each sweep's boundary rows
};

template <int Q>
__global__ void __launch_bounds__(1024)
mgm_sweep_kernel(MgmSweepParams A) {
  constexpr int RING = SWEEP_RING, FRING = SWEEP_FRING, KC = SWEEP_KC;
  constexpr int VQ = 16 * Q, SLOT = 2 * VQ + 4, NWORDS = SLOT, NWC = (Q + 1) / 2;
  extern __shared__ unsigned lds_u32[];
  lds_uint* const sm = (lds_uint*)lds_u32;
  const int tid = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int NW = A.nw;
  const int W = A.g.ocols, H = A.g.orows, stride = A.stride, q32 = stride / 2;
  const size_t ENT = SLOT;
  volatile lds_uint* done = sm + 1;
  lds_uint* rings = sm + 32;
  if (threadIdx.x == 0) sm[0] = atomicAdd(&A.sync[0], 1u);
  if (threadIdx.x >= 1 && threadIdx.x < 32) sm[threadIdx.x] = 0u;
  __syncthreads();
  // tickets -> (sweep, block): round robin over the sweeps while they have blocks left, so that every sweep's blocks start in order
  int sweep = 0, blk = 0;
  {
    int t = (int)sm[0];
    const int m = min(min(A.nblk[0], A.nblk[1]), min(A.nblk[2], A.nblk[3]));      // all four sweeps alive: rounds of four tickets
    if (t < 4 * m) { sweep = t & 3; blk = t >> 2; }
    else {                                                          // the two longer sweeps (more lines) share the rest, round robin
      t -= 4 * m;
      int alive[4], n = 0;
      for (int q = 0; q < 4; ++q) if (A.nblk[q] > m) alive[n++] = q;
      sweep = alive[t % n]; blk = m + t / n;                        // (equal line counts among the survivors: same-shape frames come in pairs)
    }
  }
  const unsigned epoch = A.epoch;
  // frames: 0 (u, v) = (x, y); 1 (-x, -y); 2 (y, -x); 3 (-y, x)
  const bool transposed = sweep >= 2;
  const int U = transposed ? H : W, V = transposed ? W : H;
  unsigned long long* bnd = A.bnd + A.bnd_off[sweep] * ENT;
  if (wv == NW) {
    if (blk > 0) sweep_feeder<SLOT, NWORDS>(bnd + (size_t)(blk - 1) * U * ENT, U, rings + NW * RING * SLOT, done, NW, epoch, tid
#ifdef VWGPU_SWEEP_DEBUG
                                            , 0, 0, 4096
#endif
                                            );
    return;
  }
  const int vv = blk * NW + wv;                                    // line in sweep order
  if (vv >= V) return;
  const bool has_above = vv > 0;
  const int mode = vv + 1 >= V ? 0 : (wv < NW - 1 ? 1 : 2);
  const int aidx = wv > 0 ? wv - 1 : NW;
  const int AMASK = (wv > 0 ? RING : FRING) - 1;
  const lds_uint* aring = rings + aidx * RING * SLOT;
  volatile lds_uint* adone = done + aidx;
  lds_uint* oring = rings + wv * RING * SLOT;
  unsigned long long* bdst = bnd + (size_t)blk * U * ENT;

  // image coordinates of (u, v): x = x0 + u * xu, y = y0 + u * yu
  int x0, y0, xu, yu, axis_dir, diag_dir;
  if (sweep == 0) { x0 = 0; y0 = vv; xu = 1; yu = 0; axis_dir = 0; diag_dir = 1; }                    // L, TL
  else if (sweep == 1) { x0 = W - 1; y0 = H - 1 - vv; xu = -1; yu = 0; axis_dir = 2; diag_dir = 3; }  // R, BR
  else if (sweep == 2) { x0 = W - 1 - vv; y0 = 0; xu = 0; yu = 1; axis_dir = 4; diag_dir = 7; }       // T, TR
  else { x0 = vv; y0 = H - 1; xu = 0; yu = -1; axis_dir = 6; diag_dir = 5; }                          // B, BL
  // the far-side pixel of the intensity difference: (col - ax, row - ay) of the pass's path predecessor (SGM.cc:2715-2721)
  const int row = tid >> 4, j = tid & 15;
  const bool diag = row >= 2;
  const vwgpu::MgmDir md = vwgpu::kMgmDirs[diag ? diag_dir : axis_dir];
  const int dl = row == 2 ? -1 : (row == 3 ? 1 : 0);
  const int voff = (diag ? VQ : 0) + j * Q;
  const int moff = 2 * VQ + (diag ? 1 : 0);
  const int num_disp = A.g.num_dx, npairs = (num_disp + 1) / 2;
  unsigned dead[Q], r[Q];
#pragma unroll
  for (int e = 0; e < Q; ++e) {
    const int p = j * Q + e;
    dead[e] = p >= npairs ? 0xffffffffu : ((2 * p + 1 >= num_disp) ? 0xffff0000u : 0u);
    r[e] = dead[e];
  }
  unsigned mp = 0;
  const bool writer = row == 0 || row == 2;                        // the lanes that hold a pass's values for the volumes / the ring
  const bool st_full = writer && j * Q + Q <= q32, st_part = writer && j * Q < q32 && !st_full;
  const long long p0 = (long long)y0 * W + x0;
  const long long pstep = (long long)yu * W + xu;                   // pixels per marching step
  const unsigned cb0 = (unsigned)(2 * Q * j);
  const uint8_t* cfetch = A.cost + p0 * stride + (cb0 & ~3u);
  const unsigned csh = (cb0 & 2u) * 8u;
  const long long cstep = pstep * stride;
  unsigned* ostore = reinterpret_cast<unsigned*>(A.vols + (size_t)(diag ? diag_dir : axis_dir) * A.vol_stride) + p0 * q32 + j * Q;
  const long long ostep = pstep * q32;
  const us2 p1p1 = as_us2(A.p1 | (A.p1 << 16));
  const unsigned p1 = A.p1, p2 = A.p2;

  int gc, gp;
  auto grey = [&](int s0) __attribute__((always_inline)) {
    const int us = min(s0 + j, U - 1);
    const int col = x0 + us * xu, rw_ = y0 + us * yu;
    const int fc = min(max(col - md.ax + A.min_col, 0), A.lw - 1), fr = min(max(rw_ - md.ay + A.min_row, 0), A.lh - 1);
    gc = A.left[(size_t)(rw_ + A.min_row) * A.lw + col + A.min_col];
    gp = A.left[(size_t)fr * A.lw + fc];
  };
  grey(0);
  unsigned penl = 0;
  auto refresh = [&](int s0) __attribute__((always_inline)) {
    int grad = gc - gp; grad = grad < 0 ? -grad : grad;
    unsigned v = p2 / (unsigned)max(grad, 1);
    if (v < p1) v = p1;
    penl = v & 0xffffu;
    grey(min(s0 + 16, U - 1) & ~15);
  };
  unsigned pm = 0xffffffffu, pn = 0xffffffffu;
  unsigned avail = 0, cavail = 0;
  const int bp_pen = (tid & 48) << 2;

  auto step = [&](const unsigned (&cw)[NWC], int uu) __attribute__((always_inline)) {
    const unsigned pen = (unsigned)__builtin_amdgcn_ds_bpermute(bp_pen + ((uu & 15) << 2), (int)penl);
    // (1) the vectors the four evaluations start from
    if (has_above) {
      const int need = min(uu + 2, U);
      const int sl = ((uu + dl) & AMASK) * SLOT;
      unsigned t[Q], tm = 0;
      for (;;) {
        const unsigned got = avail >= (unsigned)need ? avail : *adone;
        asm volatile("" ::: "memory");
        if (row > 0) {
#pragma unroll
          for (int e = 0; e < Q; ++e) t[e] = aring[sl + voff + e];
          tm = aring[sl + moff];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        avail = (unsigned)__builtin_amdgcn_readfirstlane((int)got);
        if (avail >= (unsigned)need) break;
        __builtin_amdgcn_s_sleep(1);
      }
      if (row > 0) {
#pragma unroll
        for (int e = 0; e < Q; ++e) r[e] = t[e];
        mp = tm;
      }
    }
    // the pass's border test: a pixel that fails it keeps its local costs — both evaluations start from "none"
    const bool uses = has_above && (diag ? (uu > 0 && uu + 1 < U) : uu > 0);
    if (!uses) {
#pragma unroll
      for (int e = 0; e < Q; ++e) r[e] = dead[e];
      mp = 0;
    }
    const unsigned dj = (mp + pen) & 0xffffu;
    const us2 dJ = as_us2(dj | (dj << 16)), mpp = as_us2(mp | (mp << 16));
    unsigned cs[NWC];
#pragma unroll
    for (int n = 0; n < NWC; ++n) cs[n] = (Q & 1) ? __builtin_amdgcn_alignbit(n + 1 < NWC ? cw[n + 1] : 0u, cw[n], csh) : cw[n];
    // (2) evaluate_path for the four (pass, predecessor) pairs
    row_shr1_keep(pm, r[Q - 1]);
    row_shl1_keep(pn, r[0]);
    unsigned al[Q + 1];
    al[0] = __builtin_amdgcn_alignbit(r[0], pm, 16);
#pragma unroll
    for (int e = 1; e < Q; ++e) al[e] = __builtin_amdgcn_alignbit(r[e], r[e - 1], 16);
    al[Q] = __builtin_amdgcn_alignbit(pn, r[Q - 1], 16);
    unsigned ev[Q];
#pragma unroll
    for (int e = 0; e < Q; ++e) {
      const us2 ctr = as_us2(r[e]);
      us2 m = __builtin_elementwise_min(__builtin_elementwise_min(as_us2(al[e]), as_us2(al[e + 1])), ctr);
      us2 v = __builtin_elementwise_add_sat(m, p1p1);
      v = __builtin_elementwise_min(v, __builtin_elementwise_min(ctr, dJ));
      v = __builtin_elementwise_add_sat(v, as_us2(__builtin_amdgcn_perm(0u, cs[e >> 1], (e & 1) ? 0x0c030c02u : 0x0c010c00u)));
      v = __builtin_elementwise_sub_sat(v, mpp);
      ev[e] = as_u32(v);
    }
    // (3) the mean of a pass's two evaluations: (a + b) / 2 per u16 without the 17th bit — (a & b) + ((a ^ b) >> 1)
    us2 mn2 = as_us2(0xffffffffu);
#pragma unroll
    for (int e = 0; e < Q; ++e) {
      const unsigned o = (unsigned)__builtin_amdgcn_ds_swizzle((int)ev[e], 0x401f);                   // lane ^ 16: the pass's other evaluation
      const us2 x1 = as_us2(ev[e] ^ o);
      const us2 half = x1 >> (us2)(1);
      r[e] = as_u32(as_us2(ev[e] & o) + half) | dead[e];
      mn2 = e == 0 ? as_us2(r[e]) : __builtin_elementwise_min(mn2, as_us2(r[e]));
    }
    const unsigned mnu = as_u32(mn2);
    const unsigned nm = row_min_u32(min(mnu & 0xffffu, mnu >> 16));
    // (4) hand the two passes' vectors to the next line
    if (mode == 1) {
      const int needc = uu - RING + 2;
      while ((int)cavail < needc) {
        cavail = done[wv + 1];
        if ((int)cavail < needc) __builtin_amdgcn_s_sleep(1);
      }
      lds_uint* slot = oring + (uu & (RING - 1)) * SLOT;
      if (writer) {
#pragma unroll
        for (int e = 0; e < Q; ++e) slot[voff + e] = r[e];
        if (j == 0) slot[moff] = nm;
      }
    } else if (mode == 2) {
      unsigned long long* ent = bdst + (size_t)uu * ENT;
      const unsigned long long tag = (unsigned long long)epoch << 32;
      if (writer) {
#pragma unroll
        for (int e = 0; e < Q; ++e) st_coherent(ent + voff + e, tag | r[e]);
        if (j < 2) st_coherent(ent + moff + (j ? (diag ? 2 : 2) : 0), tag | nm);      // lanes 1 tag the entry's two pad words
      }
    }
    asm volatile("" ::: "memory");
    done[wv] = (unsigned)(uu + 1);
    // (5) the passes' values of this pixel -> their volumes
    if (st_full) {
      SweepWords<Q> w;
#pragma unroll
      for (int e = 0; e < Q; ++e) w.v[e] = r[e];
      *reinterpret_cast<SweepWords<Q>*>(ostore) = w;
    } else if (st_part) {
#pragma unroll
      for (int e = 0; e < Q; ++e)
        if (j * Q + e < q32) ostore[e] = r[e];
    }
    ostore += ostep;
    mp = nm;                                                       // row 0 continues from it; rows 1 .. 3 reload theirs
  };

  unsigned cq[KC][NWC];
  auto load_cost = [&](unsigned (&w)[NWC]) __attribute__((always_inline)) {
#pragma unroll
    for (int n = 0; n < NWC; ++n) {
      unsigned off = 4u * n;
      asm("" : "+v"(off));
      w[n] = *reinterpret_cast<const unsigned*>(cfetch + off);
    }
    cfetch += cstep;
  };
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    load_cost(cq[k]);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int u0 = 0; u0 < U; u0 += KC) {
    if ((u0 & 15) == 0) refresh(u0);
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      if (u0 + k < U) step(cq[k], u0 + k);
      load_cost(cq[k]);
    }
  }
}

// starts of the uniform layout: pixel p's vectors begin at p * stride
__global__ void uniform_starts_kernel(unsigned long long* __restrict__ starts, size_t npix, unsigned long long stride) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < npix) starts[p] = p * stride;
}

// MGM front for full boxes on ONE search row (the layout of path_uniform_reg_kernel: lane l owns the EPT disparity pairs l * EPT ..,
// dead halves / pairs hold 0xffff so that they never win a minimum).  Both predecessor vectors, the costs and the two grey values
// are requested together — one memory round trip per front instead of the chain box -> start -> vector -> LDS of the general
// kernel — and the two evaluations run in registers (neighbours d - 1 / d + 1 through wave shifts, as there).
template <int EPT>
__global__ void __launch_bounds__(256)
mgm_front_uniform_kernel(SgmGeom g, MgmDirs D, int front, int stride, const uint8_t* __restrict__ left, int lw, int lh, int min_col, int min_row,
                         const uint8_t* __restrict__ cost, uint16_t* __restrict__ vols, size_t vol_elems, unsigned p1, unsigned p2) {
  constexpr int NW = CostWords<EPT>::N;
  const int num_disp = g.num_dx, npairs = (num_disp + 1) / 2, q32 = stride / 2;
  // four independent wavefronts (pixels) per workgroup: a front of 2048 pixels x 8 directions is 16 K wavefronts, and the
  // dispatcher starts workgroups, not waves
  const int tid = threadIdx.x & 63, q = blockIdx.y, W = g.ocols, H = g.orows;
  int c, r;
  if (!mgm_front_pixel(D, q, front, (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6), W, H, c, r)) return;
  const int ax = D.ax[q], ay = D.ay[q], bx = D.bx[q], by = D.by[q], need = D.need[q];
  const bool ok = vwgpu::mgm_uses_predecessors(need, c, r, W, H);
  const bool in = tid * EPT + EPT <= q32;
  const size_t p = (size_t)r * W + c;
  unsigned* vol = reinterpret_cast<unsigned*>(vols + (size_t)q * vol_elems);
  unsigned cw[NW];
  load_cost_words<EPT>(cost + p * stride + (in ? tid * EPT * 2 : 0), true, cw);
  unsigned out[EPT];
  if (!ok) {                                                   // "Just init to the local cost"
#pragma unroll
    for (int e = 0; e < EPT; ++e) out[e] = cost_pair<EPT>(cw, e);
  } else {
    const size_t pa = (size_t)(r + ay) * W + (c + ax), pb = (size_t)(r + by) * W + (c + bx);
    const unsigned loff = in ? (unsigned)(tid * EPT) : 0u;
    unsigned va[EPT], vb[EPT], dead[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) { va[e] = vol[pa * q32 + loff + e]; vb[e] = vol[pb * q32 + loff + e]; }
    const int fc = min(max(c - ax + min_col, 0), lw - 1), fr = min(max(r - ay + min_row, 0), lh - 1);
    int grad = (int)left[(size_t)(r + min_row) * lw + (c + min_col)] - (int)left[(size_t)fr * lw + fc];
    grad = grad < 0 ? -grad : grad;
    unsigned p2_mod = p2 / (unsigned)max(grad, 1);
    if (p2_mod < p1) p2_mod = p1;
    const unsigned BAD = (255u + p2) & 0xffffu;
    const us2 p1p1 = as_us2(p1 | (p1 << 16));
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
      const int j = tid * EPT + e;
      dead[e] = j >= npairs ? 0xffffffffu : ((2 * j + 1 >= num_disp) ? 0xffff0000u : 0u);
    }
    auto evaluate = [&](const unsigned (&pv)[EPT], unsigned (&res)[EPT]) __attribute__((always_inline)) {
      unsigned pr[EPT];
      us2 mn2 = as_us2(0xffffffffu);
#pragma unroll
      for (int e = 0; e < EPT; ++e) { pr[e] = pv[e] | dead[e]; mn2 = __builtin_elementwise_min(mn2, as_us2(pr[e])); }
      const unsigned mnu = as_u32(mn2);
      const unsigned min_prior = min(wave_min_u32(min(mnu & 0xffffu, mnu >> 16)), BAD);
      const unsigned dj = (min_prior + p2_mod) & 0xffffu;
      const us2 dJ = as_us2(dj | (dj << 16)), mp = as_us2(min_prior | (min_prior << 16));
      const unsigned pm = wave_shr1(pr[EPT - 1], 0xffffffffu), pn = wave_shl1(pr[0], 0xffffffffu);
      unsigned al[EPT + 1];                                         // al[e] = (d_2j-1, d_2j) of pair e; al[e+1] = (d_2j+1, d_2j+2)
      al[0] = __builtin_amdgcn_alignbit(pr[0], pm, 16);
#pragma unroll
      for (int e = 1; e < EPT; ++e) al[e] = __builtin_amdgcn_alignbit(pr[e], pr[e - 1], 16);
      al[EPT] = __builtin_amdgcn_alignbit(pn, pr[EPT - 1], 16);
#pragma unroll
      for (int e = 0; e < EPT; ++e) {
        const us2 ctr = as_us2(pr[e]);
        us2 m = __builtin_elementwise_min(__builtin_elementwise_min(as_us2(al[e]), as_us2(al[e + 1])), ctr);
        us2 v = __builtin_elementwise_add_sat(m, p1p1);
        v = __builtin_elementwise_min(v, __builtin_elementwise_min(ctr, dJ));
        v = __builtin_elementwise_add_sat(v, as_us2(cost_pair<EPT>(cw, e)));
        v = __builtin_elementwise_sub_sat(v, mp);
        res[e] = as_u32(v);
      }
    };
    unsigned ra[EPT], rb[EPT];
    evaluate(va, ra);
    evaluate(vb, rb);
#pragma unroll
    for (int e = 0; e < EPT; ++e) out[e] = (ra[e] & rb[e]) + (((ra[e] ^ rb[e]) >> 1) & 0x7fff7fffu);      // (a + b) / 2 per u16 half, no carry
  }
  if (in) {
    unsigned* o = vol + p * q32 + tid * EPT;
    if constexpr (EPT == 2) *reinterpret_cast<uint2*>(o) = make_uint2(out[0], out[1]);
    else if constexpr (EPT == 4) *reinterpret_cast<uint4*>(o) = make_uint4(out[0], out[1], out[2], out[3]);
    else {
#pragma unroll
      for (int e = 0; e < EPT; ++e) o[e] = out[e];
    }
  }
}

// sums (u16, wrapping like the reference's `+=`) of n direction volumes: accum = (add ? accum : 0) + vol_0 + .. + vol_{n-1}
__global__ void __launch_bounds__(256)
mgm_sum_kernel(uint4* __restrict__ accum, const uint4* __restrict__ vols, size_t vol_words, size_t words, int n, int add) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) {
    uint4 a = add ? accum[i] : make_uint4(0u, 0u, 0u, 0u);
    for (int k = 0; k < n; ++k) {
      const uint4 v = vols[(size_t)k * vol_words + i];
      a.x = as_u32(as_us2(a.x) + as_us2(v.x)); a.y = as_u32(as_us2(a.y) + as_us2(v.y));
      a.z = as_u32(as_us2(a.z) + as_us2(v.z)); a.w = as_u32(as_us2(a.w) + as_us2(v.w));
    }
    accum[i] = a;
  }
}

// ---- winner take all --------------------------------------------------------------------------------------------------

// One wavefront per pixel, WTA_PPW consecutive pixels per wavefront: their records (box, vector start) and the first 128
// elements of their vectors are requested together before the first one is reduced — a pixel is two dependent memory round
// trips (record -> vector) and a few dozen instructions, so one pixel per wave was pure latency (1.5 ms for 2048^2 x 129).
// Ties on the minimum trigger the reference's smoothing loop on an LDS copy.
constexpr int WTA_PPW = 4;
__global__ void __launch_bounds__(256)
wta_kernel(const B4* __restrict__ bounds, const unsigned long long* __restrict__ starts, size_t npix, int max_nd,
           uint16_t* __restrict__ accum, int32_t* __restrict__ disp, const uint8_t* __restrict__ todo, const int* __restrict__ any_todo) {
  extern __shared__ uint16_t sm[];
  if (any_todo && *any_todo == 0) return;                  // wta_uniform_kernel decided every pixel
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t p0 = ((size_t)blockIdx.x * 4 + wv) * WTA_PPW;
  if (p0 >= npix) return;                                  // whole wave exits together
  if (todo) {                                              // nothing left for this wave: leave before requesting any vector
    bool any = false;
#pragma unroll
    for (int q = 0; q < WTA_PPW; ++q) any |= todo[p0 + q < npix ? p0 + q : npix - 1] != 0;
    if (!any) return;
  }
  uint16_t* A = sm + (size_t)wv * 2 * max_nd;
  uint16_t* Bf = A + max_nd;
  B4 bq[WTA_PPW];
  unsigned long long sq[WTA_PPW];
  unsigned v0[WTA_PPW], v1[WTA_PPW];
#pragma unroll
  for (int q = 0; q < WTA_PPW; ++q) {
    const size_t pq = p0 + q < npix ? p0 + q : npix - 1;
    bq[q] = bounds[pq]; sq[q] = starts[pq];
  }
#pragma unroll
  for (int q = 0; q < WTA_PPW; ++q) {
    const int nq = (bq[q].x1 - bq[q].x0 + 1) * (bq[q].y1 - bq[q].y0 + 1);
    v0[q] = lane < nq ? accum[sq[q] + lane] : 0u;
    v1[q] = lane + 64 < nq ? accum[sq[q] + lane + 64] : 0u;
  }
#pragma unroll
  for (int q = 0; q < WTA_PPW; ++q) {
  const size_t p = p0 + q;
  if (p >= npix) break;                                    // wave-uniform
  if (todo && !todo[p]) continue;                          // already decided by wta_uniform_kernel
  const B4 b = bq[q];
  const int width = b.x1 - b.x0 + 1, height = b.y1 - b.y0 + 1, n = width * height;
  int32_t* o = disp + p * 3;
  if (n <= 0) { if (lane == 0) { o[0] = 0; o[1] = 0; o[2] = 0; } continue; }
  uint16_t* av = accum + sq[q];
  unsigned key = 0xffffffffu;
  __builtin_amdgcn_wave_barrier();                         // the previous pixel's LDS traffic is done
  for (int i = lane; i < n; i += 64) {
    const unsigned v = i < 64 ? v0[q] : (i < 128 ? v1[q] : (unsigned)av[i]);
    A[i] = (uint16_t)v; key = min(key, (v << 16) | (unsigned)i);
  }
  key = wave_min_u32(key);
  unsigned min_val = key >> 16;
  int cnt = 0;
  for (int i0 = 0; i0 < n; i0 += 64) cnt += __popcll(__ballot(i0 + lane < n && A[i0 + lane] == min_val));   // every lane holds the total
  // NB: the reference starts min_val at 65535 and counts `==` before `<`, so an all-65535 vector counts every element
  int iter = 0;
  uint16_t* in = A; uint16_t* out = Bf;
  while (cnt > 1) {
    __builtin_amdgcn_wave_barrier();
    unsigned k2 = 0xffffffffu;
    for (int i = lane; i < n; i += 64) {
      const int row = i / width, col = i - row * width;
      int mn = -1, mx = 1;
      double result = 0, wtot = 0;
      if (iter < 5) {
        if (mn + col < 0) mn = 0;
        if (mx + col >= width) mx = 0;
        for (int k = mn; k <= mx; ++k) { result += (double)in[i + k] * (1.0 / 3.0); wtot += (1.0 / 3.0); }
      } else {
        if (mn + row < 0) mn = 0;
        if (mx + row >= height) mx = 0;
        for (int k = mn; k <= mx; ++k) { result += (double)in[i + k * width] * (1.0 / 3.0); wtot += (1.0 / 3.0); }
      }
      const unsigned v = (unsigned)(uint16_t)round(result / wtot);
      out[i] = (uint16_t)v;
      k2 = min(k2, (v << 16) | (unsigned)i);
    }
    key = wave_min_u32(k2); min_val = key >> 16;
    __builtin_amdgcn_wave_barrier();
    cnt = 0;
    for (int i0 = 0; i0 < n; i0 += 64) cnt += __popcll(__ballot(i0 + lane < n && out[i0 + lane] == min_val));
    uint16_t* t = in; in = out; out = t;
    ++iter;
    if (iter >= 6) break;
  }
  if (iter > 0) for (int i = lane; i < n; i += 64) av[i] = in[i];     // the accumulation vector keeps the smoothed values
  if (lane == 0) {
    const int mi = (int)(key & 0xffffu);
    int dy = mi / width;
    const int dx = mi - dy * width + b.x0;
    dy += b.y1 - height + 1;
    o[0] = dx; o[1] = dy; o[2] = 0x7fffffff;
  }
  }  // pixels of this wave
}

// Uniform layout, one search row: a pixel's vector is `stride` u16 at p * stride and a lane owns the pairs (2 lane, 2 lane + 1)
// (+ the pair 128 + 2 lane for up to 256 disparities): one wave reduces WTAU_PPW pixels whose dwords are all requested up front.
// Pixels whose minimum is not unique need the reference's smoothing loop: they are marked in `todo` and left to wta_kernel.
constexpr int WTAU_PPW = 8;
__global__ void __launch_bounds__(256)
wta_uniform_kernel(size_t npix, int num_disp, int stride, int min_dx, int row_dy, uint16_t* __restrict__ accum, uint16_t* __restrict__ accum2,
                   int32_t* __restrict__ disp, uint8_t* __restrict__ todo, int* __restrict__ any_todo) {
  // accum2 != nullptr: the sums of the backward sweep (sweep_uniform_kernel); a pixel's vector is accum + accum2 (u16 wrap-around)
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t p0 = ((size_t)blockIdx.x * 4 + wv) * WTAU_PPW;
  if (p0 >= npix) return;
  const int npairs = (num_disp + 1) / 2;
  unsigned va[WTAU_PPW], vb[WTAU_PPW];
#pragma unroll
  for (int q = 0; q < WTAU_PPW; ++q) {
    const size_t p = p0 + q < npix ? p0 + q : npix - 1;
    const unsigned* v32 = reinterpret_cast<const unsigned*>(accum + p * (size_t)stride);
    va[q] = lane < npairs ? v32[lane] : 0xffffffffu;
    vb[q] = lane + 64 < npairs ? v32[lane + 64] : 0xffffffffu;
    if (accum2) {
      const unsigned* w32 = reinterpret_cast<const unsigned*>(accum2 + p * (size_t)stride);
      if (lane < npairs) va[q] = as_u32(as_us2(va[q]) + as_us2(w32[lane]));
      if (lane + 64 < npairs) vb[q] = as_u32(as_us2(vb[q]) + as_us2(w32[lane + 64]));
    }
  }
  bool flagged = false;
#pragma unroll
  for (int q = 0; q < WTAU_PPW; ++q) {
    const size_t p = p0 + q;
    if (p >= npix) break;
    // keys (value << 16 | index); the dead half of an odd tail is 0xffff (never the minimum unless everything is)
    auto keys = [&](unsigned v, int j, unsigned& k0, unsigned& k1) __attribute__((always_inline)) {
      const unsigned lo = v & 0xffffu, hi = (2 * j + 1 < num_disp) ? (v >> 16) : 0xffffu;
      k0 = (2 * j < num_disp) ? ((lo << 16) | (unsigned)(2 * j)) : 0xffffffffu;
      k1 = (2 * j + 1 < num_disp) ? ((hi << 16) | (unsigned)(2 * j + 1)) : 0xffffffffu;
    };
    unsigned k0, k1, k2, k3;
    keys(va[q], lane, k0, k1);
    keys(vb[q], lane + 64, k2, k3);
    const unsigned key = wave_min_u32(min(min(k0, k1), min(k2, k3)));
    const unsigned mv = key >> 16;
    const int cnt = __popcll(__ballot((k0 >> 16) == mv && k0 != 0xffffffffu)) + __popcll(__ballot((k1 >> 16) == mv && k1 != 0xffffffffu)) +
                    __popcll(__ballot((k2 >> 16) == mv && k2 != 0xffffffffu)) + __popcll(__ballot((k3 >> 16) == mv && k3 != 0xffffffffu));
    if (cnt > 1) {
      if (lane == 0) todo[p] = 1;
      flagged = true;
      if (accum2) {        // the general kernel (tie smoothing, write-back) and the sub-pixel kernel find the whole vector in accum
        unsigned* v32 = reinterpret_cast<unsigned*>(accum + p * (size_t)stride);
        unsigned* w32 = reinterpret_cast<unsigned*>(accum2 + p * (size_t)stride);
        if (lane < npairs) { v32[lane] = va[q]; w32[lane] = 0u; }
        if (lane + 64 < npairs) { v32[lane + 64] = vb[q]; w32[lane + 64] = 0u; }
      }
    } else if (lane == 0) {
      todo[p] = 0;
      int32_t* o = disp + p * 3;
      o[0] = (int)(key & 0xffffu) + min_dx; o[1] = row_dy; o[2] = 0x7fffffff;
    }
  }
  if (flagged && lane == 0) atomicOr(any_todo, 1);
}

// ---- sub-pixel --------------------------------------------------------------------------------------------------------

__device__ double sp_linear(double x) { return x / 2.0; }
__device__ double sp_poly4(double x) { return (x * x * x * x + x) / 4.0; }
__device__ double sp_cos(double x) { const double PI = 3.14159265359; return (1 - cos(x * PI / 3.0)); }
__device__ double sp_lcblend(double x) {
  const double PI = 3.14159265359;
  const double factor = 1.195 - cos(x * (PI / 2.3));
  return sp_cos(x) * factor + sp_linear(x) * (1.0 - factor);
}
__device__ double sp_offset(int mode, unsigned prev, unsigned center, unsigned next, bool lb, bool rb) {
  const double ld = (double)((int)prev - (int)center), rd = (double)((int)next - (int)center);
  if (rd == 0 && ld == 0) return 0;
  if (lb) return 0.5 * ((double)center / (double)next);
  if (rb) return -1.0 * (0.5 * ((double)center / (double)prev));
  double x = rd / ld, mult = -1.0;
  if (ld < rd) { x = ld / rd; mult = 1.0; }
  double value;
  switch (mode) {
    case 3: value = sp_poly4(x); break;
    case 4: value = sp_cos(x); break;
    case 5: value = sp_lcblend(x); break;
    default: value = sp_linear(x); break;
  }
  return (value - 0.5) * mult;
}
__device__ bool sp_parabola(const double* z, double& dx, double& dy) {
  const double pinvA[54] = {
    1.0 / 6, -1.0 / 3, 1.0 / 6, 1.0 / 6, -1.0 / 3, 1.0 / 6, 1.0 / 6, -1.0 / 3, 1.0 / 6,
    1.0 / 6, 1.0 / 6, 1.0 / 6, -1.0 / 3, -1.0 / 3, -1.0 / 3, 1.0 / 6, 1.0 / 6, 1.0 / 6,
    1.0 / 4, 0.0, -1.0 / 4, 0.0, 0.0, 0.0, -1.0 / 4, 0.0, 1.0 / 4,
    -1.0 / 6, 0.0, 1.0 / 6, -1.0 / 6, 0.0, 1.0 / 6, -1.0 / 6, 0.0, 1.0 / 6,
    -1.0 / 6, -1.0 / 6, -1.0 / 6, 0.0, 0.0, 0.0, 1.0 / 6, 1.0 / 6, 1.0 / 6,
    -1.0 / 9, 2.0 / 9, -1.0 / 9, 2.0 / 9, 5.0 / 9, 2.0 / 9, -1.0 / 9, 2.0 / 9, -1.0 / 9};
  double vals[6];
  for (int i = 0; i < 6; ++i) { double s = 0; for (int j = 0; j < 9; ++j) s += (double)(float)pinvA[i * 9 + j] * z[j]; vals[i] = s; }
  const double denom = 4.0 * vals[0] * vals[1] - (vals[2] * vals[2]);
  if (fabs(denom) < 0.01) return false;
  dx = (double)(float)((vals[2] * vals[4] - 2.0 * vals[1] * vals[3]) / denom);
  dy = (double)(float)((vals[2] * vals[3] - 2.0 * vals[0] * vals[4]) / denom);
  dx = erf(dx / (0.34574 * sqrt(2.0))) / 2.0;
  dy = erf(dy / (0.38944 * sqrt(2.0))) / 2.0;
  const double nrm = sqrt(dx * dx + dy * dy);
  if (nrm >= 0.5) { const double scale = nrm / 0.5; dx /= scale; dy /= scale; }
  return true;
}

__global__ void subpixel_kernel(int mode, const B4* __restrict__ bounds, const unsigned long long* __restrict__ starts, size_t npix,
                                const uint16_t* __restrict__ accum, const uint16_t* __restrict__ accum2, const int32_t* __restrict__ idisp,
                                float* __restrict__ out) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  const int32_t* ip = idisp + p * 3;
  float* o = out + p * 3;
  if (!ip[2]) { o[0] = (float)ip[0]; o[1] = (float)ip[1]; o[2] = 0.0f; return; }
  const int dx = ip[0], dy = ip[1];
  if (mode == 0) { o[0] = (float)dx; o[1] = (float)dy; o[2] = 1.0f; return; }
  const B4 b = bounds[p];
  const int width = b.x1 - b.x0 + 1;
  const int mi = (dy - b.y0) * width + (dx - b.x0);
  int xl = -1, xr = 1, yu = -width, yd = width;
  bool tb = false, bb = false, lb = false, rb = false;
  if (dx == b.x0) { xl = 0; lb = true; }
  if (dx == b.x1) { xr = 0; rb = true; }
  if (dy == b.y0) { yu = 0; tb = true; }
  if (dy == b.y1) { yd = 0; bb = true; }
  const uint16_t* a1 = accum + starts[p];
  const uint16_t* a2 = accum2 ? accum2 + starts[p] : nullptr;     // the backward sweep's sums (sweep_uniform_kernel)
  auto a = [&](int i) __attribute__((always_inline)) -> unsigned { return a2 ? (unsigned)(uint16_t)(a1[i] + a2[i]) : (unsigned)a1[i]; };
  double ddx = 0, ddy = 0;
  bool valid = true;
  if (mode == 1) {
    const double z[9] = {(double)a(mi + xl + yu), (double)a(mi + yu), (double)a(mi + xr + yu), (double)a(mi + xl), (double)a(mi),
                         (double)a(mi + xr), (double)a(mi + xl + yd), (double)a(mi + yd), (double)a(mi + xr + yd)};
    valid = sp_parabola(z, ddx, ddy);
  } else {
    ddx = sp_offset(mode, a(mi + xl), a(mi), a(mi + xr), lb, rb);
    ddy = sp_offset(mode, a(mi + yu), a(mi), a(mi + yd), tb, bb);
  }
  if (valid) { o[0] = (float)((double)dx + ddx); o[1] = (float)((double)dy + ddy); } else { o[0] = (float)dx; o[1] = (float)dy; }
  o[2] = 1.0f;
}

struct Bump {
  char* base; size_t cap, off = 0;
  template <class T> T* take(size_t n) {
    off = vwgpu_align_up(off, 256);
    T* p = reinterpret_cast<T*>(base + off);
    off += n * sizeof(T);
    return off <= cap ? p : nullptr;
  }
};

void default_p1p2(int cost_type, int kernel, int& p1, int& p2) {     // SGM.cc:105-160
  if (cost_type != VWGPU_CENSUS_TRANSFORM && cost_type != VWGPU_TERNARY_CENSUS_TRANSFORM) {     // "default: // MAD" (:128-131, :156-158)
    if (p1 <= 0) p1 = 3;
    if (p2 <= 0) p2 = 250;
    return;
  }
  const bool tern = cost_type == VWGPU_TERNARY_CENSUS_TRANSFORM;
  if (p1 <= 0) p1 = tern ? (kernel == 3 ? 12 : kernel == 5 ? 30 : kernel == 7 ? 40 : kernel == 9 ? 40 : 30)
                         : (kernel == 3 ? 3 : kernel == 5 ? 15 : kernel == 7 ? 30 : kernel == 9 ? 20 : 3);
  if (p2 <= 0) p2 = tern ? (kernel == 3 ? 600 : kernel == 5 ? 1500 : kernel == 7 ? 2000 : kernel == 9 ? 2000 : 30)
                         : (kernel == 3 ? 70 : kernel == 5 ? 750 : kernel == 7 ? 1500 : kernel == 9 ? 1000 : 22);
}

}  // namespace

// left: lw x lh float (device), right: rw x rh float (device).  Masks / prev may be null.  out_disp: ocols x orows x 3 int32,
// out_sub: optional ocols x orows x 3 float.  Returns the output size through ow / oh.
int vwgpu_sgm_impl(vwgpu_ctx* ctx, const vwgpu_sgm_params* P, const float* left, int lw, int lh, ptrdiff_t ls,
                   const float* right, int rw, int rh, ptrdiff_t rs, int sx, int sy,
                   const uint8_t* lmask, int lmw, int lmh, const uint8_t* rmask, int rmw, int rmh,
                   const int32_t* prev, int pw, int ph, int32_t* out_disp, float* out_sub, size_t out_cap_pixels, int* ow, int* oh) {
  hipStream_t st = ctx->stream;
  const int kernel = P->kernel_size;
  const int hk = (kernel - 1) / 2;
  SgmGeom g;
  g.min_dx = 0; g.min_dy = 0; g.max_dx = sx; g.max_dy = sy; g.num_dx = sx + 1; g.num_dy = sy + 1;
  g.sbx = P->search_buffer_x; g.sby = P->search_buffer_y;
  // output extent (semi_global_matching_func, SGM.cc:2397-2420)
  int min_row = hk - g.min_dy, min_col = hk - g.min_dx;
  int max_row = std::min(lh - 1 - hk, rh - 1 - (hk + g.max_dy)), max_col = std::min(lw - 1 - hk, rw - 1 - (hk + g.max_dx));
  if (min_row < 0) min_row = 0;
  if (min_col < 0) min_col = 0;
  if (max_row > lh - 1) max_row = lh - 1;
  if (max_col > lw - 1) max_col = lw - 1;
  g.ocols = max_col - min_col + 1; g.orows = max_row - min_row + 1;
  if (g.ocols <= 0 || g.orows <= 0) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "calc_disparity_sgm: Kernel size too large of active region.");
  *ow = g.ocols; *oh = g.orows;
  const size_t npix = (size_t)g.ocols * g.orows;
  if (npix > out_cap_pixels) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "calc_disparity_sgm: output buffer too small (%d x %d needed)", g.ocols, g.orows);
  const long long num_disp = (long long)g.num_dx * g.num_dy;
  if (num_disp > 16000) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "calc_disparity_sgm: %lld disparities per pixel exceed the LDS path buffers", num_disp);
  if (lmask && !(lmw == g.ocols && lmh == g.orows)) return vwgpu_fail(ctx, VWGPU_ERR_LOGIC, "Left mask size does not match the output size.");
  if (rmask && !(rmw >= g.ocols + g.num_dx - 1 && rmh >= g.orows + g.num_dy - 1))
    return vwgpu_fail(ctx, VWGPU_ERR_LOGIC, "Right mask size is not large enough to support search range.");
  int p1 = P->p1, p2 = P->p2;
  default_p1p2(P->cost_type, kernel, p1, p2);

  // fixed-size part of the arena
  const int lcw = lw - 2 * hk, lch = lh - 2 * hk, rcw = rw - 2 * hk, rch = rh - 2 * hk;
  // (guard zones around the census rasters: the register-resident path kernel requests census words up to 2 KC steps past a line's end)
  const size_t cen_guard = (size_t)2 * VWGPU_PATH_KC * ((size_t)std::max(lcw, rcw) + 1) + 64;
  const size_t fixed = (size_t)lw * lh + (size_t)rw * rh + 8 * ((size_t)lcw * lch + (size_t)rcw * rch + 3 * cen_guard) + npix * (16 + 8 + 1) +
                       (size_t)g.orows * (24 + 8 + 8) + (1 << 16);
  int rc = vwgpu_arena_reserve(ctx, &ctx->sgm, fixed);
  if (rc) return rc;
  Bump A{static_cast<char*>(ctx->sgm.base), ctx->sgm.cap};
  uint8_t* l8 = A.take<uint8_t>((size_t)lw * lh);
  uint8_t* r8 = A.take<uint8_t>((size_t)rw * rh);
  A.take<uint64_t>(cen_guard);
  uint64_t* lc = A.take<uint64_t>((size_t)lcw * lch + cen_guard);
  uint64_t* rcen = A.take<uint64_t>((size_t)rcw * rch + cen_guard);
  B4* bounds = A.take<B4>(npix);
  unsigned long long* starts = A.take<unsigned long long>(npix);
  uint8_t* full_search = A.take<uint8_t>(npix);
  unsigned long long* rowsum = A.take<unsigned long long>((size_t)3 * g.orows);      // sums, then the rows' counts of small boxes and their cells
  unsigned long long* rowoff = A.take<unsigned long long>(g.orows);
  int2* rowext = A.take<int2>(g.orows);
  unsigned* mm = A.take<unsigned>(8);
  int* ext = reinterpret_cast<int*>(mm + 4);
  if (!l8 || !r8 || !lc || !rcen || !bounds || !starts || !full_search || !rowsum || !rowoff || !rowext || !mm)
    return vwgpu_fail(ctx, VWGPU_ERR_LOGIC, "calc_disparity_sgm: internal arena too small");

  // u8_convert both images (SGM.cc:210-212)
  {
    vwgpu_prof_scope ps(ctx, "sgm_u8_convert");
    const unsigned init[8] = {0xffffffffu, 0u, 0xffffffffu, 0u, (unsigned)(rmh - 1), 0u, 0u, 0u};
    VWGPU_HIP(ctx, hipMemcpyAsync(mm, init, sizeof init, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(minmax_kernel, dim3(std::min((lw + 255) / 256, 8), std::min(lh, 128)), dim3(256), 0, st, left, ls, lw, lh, mm);
    hipLaunchKernelGGL(minmax_kernel, dim3(std::min((rw + 255) / 256, 8), std::min(rh, 128)), dim3(256), 0, st, right, rs, rw, rh, mm + 2);
    hipLaunchKernelGGL(u8_convert_kernel, dim3((lw + 255) / 256, lh), dim3(256), 0, st, left, ls, lw, lh, mm, l8);
    hipLaunchKernelGGL(u8_convert_kernel, dim3((rw + 255) / 256, rh), dim3(256), 0, st, right, rs, rw, rh, mm + 2, r8);
  }
  const bool block_cost = P->cost_type != VWGPU_CENSUS_TRANSFORM && P->cost_type != VWGPU_TERNARY_CENSUS_TRANSFORM;   // opt-in checked by the ABI layer
  if (!block_cost) {
    vwgpu_prof_scope ps(ctx, "sgm_census");
    const int tern = P->cost_type == VWGPU_TERNARY_CENSUS_TRANSFORM;
    hipLaunchKernelGGL(census_kernel, dim3((lcw + 63) / 64, (lch + 3) / 4), dim3(64, 4), 0, st, l8, lw, lh, kernel, tern, P->ternary_census_threshold, lc);
    hipLaunchKernelGGL(census_kernel, dim3((rcw + 63) / 64, (rch + 3) / 4), dim3(64, 4), 0, st, r8, rw, rh, kernel, tern, P->ternary_census_threshold, rcen);
  }
  // per-pixel disparity bounds
  {
    vwgpu_prof_scope ps(ctx, "sgm_bounds");
    if (rmask) {
      hipLaunchKernelGGL(mask_col_extent_kernel, dim3((rmh + 3) / 4), dim3(256), 0, st, rmask, rmw, rmh, g.ocols, ext);
      hipLaunchKernelGGL(mask_row_extent_kernel, dim3((g.orows + 3) / 4), dim3(256), 0, st, rmask, rmw, g.orows, rowext);
    }
    hipLaunchKernelGGL(bounds_kernel, dim3((g.ocols + 255) / 256, g.orows), dim3(256), 0, st, g, lmask, rmask, ext, rowext, prev, pw, ph, bounds, full_search);
  }
  // memory-cap loop over the conservation levels (SGM.cc:468-491) + ragged starts
  std::vector<unsigned long long> h_rows((size_t)3 * g.orows);
  unsigned long long main_buf = 0, n_total = 0, small16 = 0, small32 = 0, cells16 = 0;
  bool ok = false;
  const int threads = P->num_threads > 0 ? P->num_threads : 1;
  for (int level = 0; level <= 3; ++level) {
    if (prev) {
      const int range = level == 0 ? 10 : level == 1 ? 25 : level == 2 ? 3 : 0;
      vwgpu_prof_scope ps(ctx, "sgm_constrain");
      hipLaunchKernelGGL(constrain_kernel, dim3((g.ocols + 4 * CONSTRAIN_CPW - 1) / (4 * CONSTRAIN_CPW), g.orows), dim3(256), 0, st, g, full_search, bounds, range, level);
    }
    hipLaunchKernelGGL(row_count_kernel, dim3(g.orows), dim3(256), 0, st, bounds, g.ocols, rowsum);
    VWGPU_HIP(ctx, hipMemcpyAsync(h_rows.data(), rowsum, (size_t)g.orows * 24, hipMemcpyDeviceToHost, st));
    VWGPU_HIP(ctx, hipStreamSynchronize(st));
    unsigned long long n = 0;
    small16 = small32 = cells16 = 0;
    for (int r = 0; r < g.orows; ++r) {
      const unsigned long long v = h_rows[r], sm_ = h_rows[(size_t)g.orows + r];
      h_rows[r] = n; n += v; small16 += sm_ & 0xffffffffull; small32 += sm_ >> 32; cells16 += h_rows[(size_t)2 * g.orows + r];
    }
    n_total = n;
    if (n < 6) n = 6;
    main_buf = n;
    const int line_size = (int)(std::sqrt((double)(g.ocols * g.ocols + g.orows * g.orows)) + 1);
    unsigned long long one_buf = (unsigned long long)line_size * (unsigned long long)num_disp;
    if (one_buf > main_buf) one_buf = main_buf;
    const double MB = 1024.0 * 1024.0;
    unsigned long long small_buf = one_buf * threads;
    if (P->use_mgm) {                                        // four vertical + four horizontal one-path line buffers (SGM.cc:707-713)
      const unsigned long long vert = std::min((unsigned long long)g.orows * (unsigned long long)num_disp, main_buf);
      const unsigned long long horiz = std::min((unsigned long long)g.ocols * (unsigned long long)num_disp, main_buf);
      small_buf = 4 * vert + 4 * horiz;
    }
    const double total = (double)n * (3.0 / MB) + (double)small_buf * (2.0 / MB);
    if (!(total > (double)P->memory_limit_mb)) { ok = true; break; }
  }
  if (!ok) {   // "Unable to compute valid search ranges for SGM input": an all-invalid disparity (SGM.cc:2428-2434)
    VWGPU_HIP(ctx, hipMemsetAsync(out_disp, 0, npix * 12, st));
    if (out_sub) VWGPU_HIP(ctx, hipMemsetAsync(out_sub, 0, npix * 12, st));
    return VWGPU_OK;
  }
  // every pixel searches the whole range: no masks, no previous level (bounds_kernel then wrote the full box everywhere)
  const bool uniform = (unsigned long long)npix * (unsigned long long)num_disp == n_total && num_disp <= 64 * 8;
  // per-pixel vector stride of the uniform layout: one search row -> a multiple of 8 (the register-resident path kernel moves 8 or
  // 16 bytes per lane; every byte of padding is carried through 8 read-modify-write passes), 2-D searches -> 16 (16-byte LDS chunks)
  const int ustride = g.num_dy == 1 ? (int)((num_disp + 7) / 8 * 8) : (int)((num_disp + 15) / 16 * 16);
  if (uniform) {
    main_buf = (unsigned long long)npix * ustride;
    hipLaunchKernelGGL(uniform_starts_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, starts, npix, (unsigned long long)ustride);
  } else {
    VWGPU_HIP(ctx, hipMemcpyAsync(rowoff, h_rows.data(), (size_t)g.orows * 8, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(row_scan_kernel, dim3(g.orows), dim3(256), 0, st, bounds, g.ocols, rowoff, starts);
  }

  // the two ragged buffers (separate arena: reserving may reallocate, the fixed part above must stay put)
  // The register-resident path kernel fetches two chunks of steps ahead without looking at the end of its line: up to 16 steps
  // past either end, i.e. 16 rows + 16 pixels before / behind a volume.  Guard zones keep those (unused) reads inside the arena.
  const size_t guard = uniform ? vwgpu_align_up((size_t)2 * VWGPU_PATH_KC * ((size_t)g.ocols + 1) * ustride * 2, 256) : 0;
  // MGM keeps the values of the directions in flight in volumes of their own (mgm_front_kernel): all eight, unless that takes
  // more than 24 GB (then four, two or one at a time: more launches, same result)
  const size_t vol_bytes = vwgpu_align_up((size_t)main_buf * 2, 256);
  int mgm_vols = 0;
  if (P->use_mgm) for (mgm_vols = 8; mgm_vols > 1 && (size_t)mgm_vols * vol_bytes > ((size_t)24 << 30); mgm_vols /= 2) {}
  // fused raster sweeps (sweep_uniform_kernel): full boxes on one search row, up to 256 disparities (the packed winner-take-all kernel
  // adds the two sweeps' volumes), enough LDS for at least two rows per workgroup
  int sweep_nw = 0, sweep_q = 0;
  if (uniform && g.num_dy == 1 && !P->use_mgm && num_disp <= 256 && ctx->sgm_sweep >= 1) {      // opt-in: see the kernel's header for the trade
    sweep_q = (int)(((num_disp + 1) / 2 + 15) / 16);               // disparity pairs per lane of a 16-lane row: 1 .. 8
    if (sweep_q == 7) sweep_q = 8;
    const size_t slot = (size_t)(3 * 16 * sweep_q + 4) * 4;
    // rows per workgroup: as many as the LDS takes (13 at 129 disparities) unless VWGPU_OPT_SGM_SWEEP pins the count (2 .. 15):
    // every workgroup boundary costs a hand-off through HBM (5 .. 8 us measured), every row of a workgroup shares its CU
    const size_t budget = 120 * 1024 - 128 - SWEEP_FRING * slot;
    sweep_nw = (int)std::min<size_t>(ctx->sgm_sweep >= 2 ? ctx->sgm_sweep : 15, budget / (SWEEP_RING * slot));
    if (sweep_nw < 2 || g.ocols < 4) sweep_nw = 0;
  }
  const size_t accum2_bytes = sweep_nw ? vol_bytes + guard : 0;
  rc = vwgpu_arena_reserve(ctx, &ctx->sgm_main, (size_t)main_buf * 3 + 1024 + 3 * guard + (size_t)mgm_vols * vol_bytes + accum2_bytes + 512);
  if (rc) return rc;
  uint8_t* cost = static_cast<uint8_t*>(ctx->sgm_main.base) + guard;
  uint16_t* accum = reinterpret_cast<uint16_t*>(static_cast<char*>(ctx->sgm_main.base) + 2 * guard + vwgpu_align_up((size_t)main_buf, 256));
  uint16_t* mgm_vol = reinterpret_cast<uint16_t*>(static_cast<char*>(ctx->sgm_main.base) + 3 * guard + vwgpu_align_up((size_t)main_buf, 256) + vol_bytes);
  uint16_t* accum2 = sweep_nw ? reinterpret_cast<uint16_t*>(static_cast<char*>(ctx->sgm_main.base) + 3 * guard + vwgpu_align_up((size_t)main_buf, 256) + vol_bytes) : nullptr;
  // one direction per launch, plain store / read-modify-write
  const bool dir_paths = uniform && g.num_dy == 1 && !P->use_mgm;      // the first direction initialises the volume
  // the last direction of that schedule also takes the winners (ACC_RMW_WTA)
  int* const wta_flag = reinterpret_cast<int*>(mm + 6);
  const bool wta_in_paths = dir_paths && num_disp <= 256;
  bool wta_done = false;
  if (!dir_paths && !P->use_mgm) VWGPU_HIP(ctx, hipMemsetAsync(accum, 0, (size_t)main_buf * 2, st));      // (MGM: mgm_sum_kernel stores)
  // Census costs on that schedule are formed inside the path kernel from the census rasters (CEN): no u8 volume at all
  const int cen_oc = min_col - hk, cen_or = min_row - hk;
  const bool cen_ok = dir_paths && !block_cost && cen_oc >= 0 && cen_or >= 0 && cen_oc + g.ocols <= lcw && cen_or + g.orows <= std::min(lch, rch) &&
                      cen_oc + g.ocols - 1 + (int)num_disp <= rcw;
  const bool ring_paths = dir_paths && !(ctx->sgm_path_mode & 32) && ustride <= 160;            // path_ring_kernel
  const bool ring_cen = ring_paths && cen_ok && num_disp <= 129 && (ctx->sgm_path_mode & 2048) && (ctx->sgm_path_mode & 15) < 8;   // ... forming the census costs itself (opt-in: measured slower)
  if (ring_cen) {
  } else if (block_cost) {
    // fill_costs_block (SGM.cc:1711-1738).  Exact n / count for every n the sums can reach: multiply-high by 2^32 / count + 1.
    vwgpu_prof_scope ps(ctx, "sgm_cost");
    const unsigned count = (unsigned)(kernel * kernel), magic = (unsigned)((1ull << 32) / count + 1);
    for (unsigned n = 0; count > 1 && n <= 255u * count; ++n)         // (a single-pixel kernel returns the plain difference, SGM.cc:1658-1662)
      if ((unsigned)(((unsigned long long)n * magic) >> 32) != n / count) return vwgpu_fail(ctx, VWGPU_ERR_LOGIC, "calc_disparity_sgm: division constant");
    const bool fast = uniform && g.num_dy == 1 && (kernel == 3 || kernel == 5 || kernel == 7 || kernel == 9 || kernel == 11);
    if (fast) {
      const int nw = (kernel + 3) / 4, rlw = block_row_dwords((int)num_disp, nw);
      const size_t lds = ((size_t)kernel * 4 * rlw + 4 * rlw + (size_t)kernel * (rlw + 1)) * sizeof(unsigned);
      const dim3 grd((g.ocols + 255) / 256, g.orows);
#define VWGPU_BLOCK_ROW(KK) hipLaunchKernelGGL(cost_block_row_kernel<KK>, grd, dim3(256), lds, st, l8, lw, lh, r8, rw, rh, g.ocols, (int)num_disp, ustride, \
                                               min_col, min_row, magic, cost)
      switch (kernel) { case 3: VWGPU_BLOCK_ROW(3); break; case 5: VWGPU_BLOCK_ROW(5); break; case 7: VWGPU_BLOCK_ROW(7); break;
                        case 9: VWGPU_BLOCK_ROW(9); break; default: VWGPU_BLOCK_ROW(11); break; }
#undef VWGPU_BLOCK_ROW
    } else {
      if (uniform) VWGPU_HIP(ctx, hipMemsetAsync(cost, 0, (size_t)main_buf, st));       // the dead slots of the stride
      hipLaunchKernelGGL(cost_block_kernel, dim3((unsigned)((npix + 3) / 4)), dim3(256), 0, st, l8, lw, r8, rw, bounds, starts, g.ocols, npix, min_col, min_row,
                         kernel, magic, cost);
    }
  } else {
    vwgpu_prof_scope ps(ctx, "sgm_cost");
    if (uniform)
    {
      if (g.num_dy == 1) {
        const size_t crl = (size_t)(256 + num_disp + 1) * 8 + (size_t)256 * ustride;      // <= 6 KB + 128 KB (512 disparities)
        VWGPU_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(cost_row_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)crl));
        hipLaunchKernelGGL(cost_row_kernel, dim3((g.ocols + 255) / 256, g.orows), dim3(256), crl, st, lc, lcw, rcen, rcw,
                           g.ocols, (int)num_disp, ustride, min_col - hk, min_row - hk, cost);
      } else {
        const int qd = ustride / 16;                        // 16-disparity groups per pixel (2-D searches: stride % 16 == 0, 1 .. 32 groups)
        const dim3 blk(qd, std::max(1, 256 / qd)), grd((g.ocols + blk.y - 1) / blk.y, g.orows);
        hipLaunchKernelGGL(cost_uniform16_kernel<false>, grd, blk, 0, st, lc, lcw, rcen, rcw, g.ocols, g.orows, g.num_dx, (int)num_disp, ustride,
                           min_col - hk, min_row - hk, reinterpret_cast<uint32_t*>(cost));
      }
    }
    else
      hipLaunchKernelGGL(cost_kernel, dim3((unsigned)((npix + 4 * COST_PPW - 1) / (4 * COST_PPW))), dim3(256), 0, st, lc, lcw, rcen, rcw, bounds, starts, g.ocols, npix,
                         min_col - hk, min_row - hk, cost);
  }
  // 8 directions, in the reference's order (SGM.cc:2488-2610)
  {
    const size_t lds = (size_t)num_disp * 2 * sizeof(uint16_t);
    const int W = g.ocols, H = g.orows;
    struct Dir { int dc, dr, n_first, first_is_row_border, n_second, second_skip; };
    const Dir dirs[8] = {
      {0, 1, W, 1, 0, 0},   {0, -1, W, 1, 0, 0}, {1, 0, H, 0, 0, 0},       {-1, 0, H, 0, 0, 0},
      {1, 1, W, 1, H - 1, 1}, {-1, 1, W, 1, H - 1, 1}, {1, -1, W, 1, H - 1, 0}, {-1, -1, W, 1, H - 1, 0}};
    // all 8 directions in one launch when the packed 16-bit atomics cannot carry; one launch per direction otherwise
    const bool together = 8 * (255 + std::max(p1, p2)) < 65536;
    const int ept = (int)((num_disp + 63) / 64);
    // pixels per staged chunk: small chunks keep the LDS footprint of a one-wave workgroup near 4 KB, so ~30 of them fit
    // a CU — the per-pixel recurrence is a serial LDS / DPP chain that only occupancy hides (2048^2, D = 129: 10.7 ms
    // with 28-pixel chunks, 8.8 ms with 8..12)
    int K = 4096 / (3 * ustride);
    K = std::max(4, std::min(K, 32));
    const size_t ulds = ((size_t)ept * 64 + 256) * sizeof(uint16_t) + (size_t)K * ustride * 3 + K + 16;
    const bool one_d = g.num_dy == 1;
    int mgm_q = 0, mgm_nw = 0;
    if (P->use_mgm && uniform && one_d && num_disp <= 256 && mgm_vols == 8 && ctx->mgm_sweep != 1 && g.ocols >= 4 && g.orows >= 4) {
      mgm_q = (int)(((num_disp + 1) / 2 + 15) / 16);
      if (mgm_q == 7) mgm_q = 8;
      const size_t slot = (size_t)(2 * 16 * mgm_q + 4) * 4;
      mgm_nw = (int)std::min<size_t>(ctx->mgm_sweep >= 2 ? ctx->mgm_sweep : 15, (120 * 1024 - 128 - SWEEP_FRING * slot) / (SWEEP_RING * slot));
      if (mgm_nw < 2) mgm_nw = 0;
    }
    if (mgm_nw) {
      // the eight passes as four concurrent sweeps (mgm_sweep_kernel), then one pass adds the volumes to the sums
      vwgpu_prof_scope ps(ctx, "sgm_mgm_paths");
      MgmSweepParams MA;
      MA.g = g; MA.stride = ustride; MA.lw = lw; MA.lh = lh; MA.min_col = min_col; MA.min_row = min_row; MA.nw = mgm_nw;
      const size_t slot_dwords = (size_t)2 * 16 * mgm_q + 4;
      size_t entries = 0;
      int total = 0;
      for (int q = 0; q < 4; ++q) {
        const int lines = q < 2 ? H : W, len = q < 2 ? W : H;
        MA.nblk[q] = (lines + mgm_nw - 1) / mgm_nw;
        MA.blk0[q] = total; total += MA.nblk[q];
        MA.bnd_off[q] = entries; entries += (size_t)MA.nblk[q] * len;
      }
      MA.blk0[4] = total;
      const size_t bnd_bytes = entries * slot_dwords * 8 + 256;
      if (ctx->sgm_bnd.cap < bnd_bytes || ctx->sgm_epoch == 0xffffffffu) {
        rc = vwgpu_arena_reserve(ctx, &ctx->sgm_bnd, bnd_bytes);
        if (rc) return rc;
        VWGPU_HIP(ctx, hipMemsetAsync(ctx->sgm_bnd.base, 0, ctx->sgm_bnd.cap, st));
        ctx->sgm_epoch = 0;
      }
      MA.p1 = (unsigned)p1; MA.p2 = (unsigned)p2; MA.epoch = ++ctx->sgm_epoch;
      MA.left = l8; MA.cost = cost; MA.vols = mgm_vol; MA.vol_stride = vol_bytes / 2;
      MA.sync = reinterpret_cast<unsigned*>(static_cast<char*>(ctx->sgm_bnd.base));
      MA.bnd = reinterpret_cast<unsigned long long*>(static_cast<char*>(ctx->sgm_bnd.base) + 256);
      VWGPU_HIP(ctx, hipMemsetAsync(MA.sync, 0, 256, st));
      const size_t mlds = (32 + ((size_t)mgm_nw * SWEEP_RING + SWEEP_FRING) * slot_dwords) * 4;
      const dim3 grd((unsigned)total), blk((unsigned)((mgm_nw + 1) * 64));
#define VWGPU_MSWEEP(QQ) do { VWGPU_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(mgm_sweep_kernel<QQ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds)); \
                              hipLaunchKernelGGL(mgm_sweep_kernel<QQ>, grd, blk, mlds, st, MA); } while (0)
      switch (mgm_q) {
        case 1: VWGPU_MSWEEP(1); break; case 2: VWGPU_MSWEEP(2); break; case 3: VWGPU_MSWEEP(3); break; case 4: VWGPU_MSWEEP(4); break;
        case 5: VWGPU_MSWEEP(5); break; case 6: VWGPU_MSWEEP(6); break; default: VWGPU_MSWEEP(8); break;
      }
#undef VWGPU_MSWEEP
      const size_t words = (vol_bytes + 15) / 16;
      hipLaunchKernelGGL(mgm_sum_kernel, dim3((unsigned)std::min<size_t>((words + 255) / 256, 8192)), dim3(256), 0, st, reinterpret_cast<uint4*>(accum),
                         reinterpret_cast<const uint4*>(mgm_vol), vol_bytes / 16, words, 8, 0);
    } else if (P->use_mgm) {
      // accum_mgm_multithread: one launch per front (see mgm_front_kernel) for all directions whose volumes fit: the axis
      // directions have W + H - 1 anti-diagonal fronts, the diagonal ones max(W, H) row / column fronts, so eight directions
      // together take W + H - 1 launches; then one pass adds the volumes to the sums.
      vwgpu_prof_scope ps(ctx, "sgm_mgm_paths");
      typedef vwgpu::MgmDir MD;
      const MD* md = vwgpu::kMgmDirs;                               // L, TL, R, BR, T, BL, B, TR (mgm_schedule.h)
      const int per = mgm_vols;                                     // directions per launch
      int pe = (int)(((num_disp + 1) / 2 + 63) / 64);
      if (pe == 3) pe = 4;
      const bool reg_fronts = uniform && one_d && pe <= 4;
      const size_t words = (vol_bytes + 15) / 16;
      for (int first = 0; first < 8; first += per) {
        MgmDirs M;
        M.n = per;
        int fronts = 0;
        for (int q = 0; q < per; ++q) {
          const MD& d = md[first + q];
          M.ax[q] = d.ax; M.ay[q] = d.ay; M.bx[q] = d.bx; M.by[q] = d.by; M.need[q] = d.need; M.kind[q] = d.kind; M.flipx[q] = d.flipx; M.flipy[q] = d.flipy;
          fronts = std::max(fronts, vwgpu::mgm_front_count(d.kind, W, H));
        }
        for (int f = 0; f < fronts; ++f) {
          int fw = 0;                                                // pixels of this front in the longest direction of the launch
          for (int q = 0; q < per; ++q) fw = std::max(fw, vwgpu::mgm_front_width(md[first + q].kind, f, W, H));
          if (fw <= 0) continue;
#define VWGPU_MGM_U(E) hipLaunchKernelGGL((mgm_front_uniform_kernel<E>), dim3((fw + 3) / 4, per), dim3(256), 0, st, g, M, f, ustride, l8, lw, lh, min_col, min_row, \
                                          cost, mgm_vol, vol_bytes / 2, (unsigned)p1, (unsigned)p2)
          if (reg_fronts) { switch (pe) { case 1: VWGPU_MGM_U(1); break; case 2: VWGPU_MGM_U(2); break; default: VWGPU_MGM_U(4); break; } }
          else
          {
            const int wpw = 4 * lds <= 48 * 1024 ? 4 : 1;              // wavefronts (pixels) per workgroup
            hipLaunchKernelGGL(mgm_front_kernel, dim3((fw + wpw - 1) / wpw, per), dim3(64 * wpw), lds * wpw, st, g, M, f, l8, lw, lh, min_col, min_row,
                               bounds, starts, cost, mgm_vol, vol_bytes / 2, (unsigned)p1, (unsigned)p2);
          }
#undef VWGPU_MGM_U
        }
        hipLaunchKernelGGL(mgm_sum_kernel, dim3((unsigned)std::min<size_t>((words + 255) / 256, 8192)), dim3(256), 0, st, reinterpret_cast<uint4*>(accum),
                           reinterpret_cast<const uint4*>(mgm_vol), vol_bytes / 16, words, per, first > 0 ? 1 : 0);
      }
    } else if (sweep_nw) {
      // two concurrent raster sweeps, four directions each: the sums are written once per sweep (sweep_uniform_kernel)
      vwgpu_prof_scope ps(ctx, "sgm_paths");
      const int nblk = (g.orows + sweep_nw - 1) / sweep_nw;
      const size_t slot_dwords = (size_t)3 * 16 * sweep_q + 4;
      const size_t bnd_bytes = (size_t)2 * nblk * g.ocols * slot_dwords * 8 + 256;
      if (ctx->sgm_bnd.cap < bnd_bytes || ctx->sgm_epoch == 0xffffffffu) {       // fresh memory (or the epochs are used up): no word may carry a future epoch
        rc = vwgpu_arena_reserve(ctx, &ctx->sgm_bnd, bnd_bytes);
        if (rc) return rc;
        VWGPU_HIP(ctx, hipMemsetAsync(ctx->sgm_bnd.base, 0, ctx->sgm_bnd.cap, st));
        ctx->sgm_epoch = 0;
      }
      SweepParams SA;
      SA.g = g; SA.stride = ustride; SA.lw = lw; SA.min_col = min_col; SA.min_row = min_row; SA.nblk = nblk; SA.nw = sweep_nw;
      SA.p1 = (unsigned)p1; SA.p2 = (unsigned)p2; SA.epoch = ++ctx->sgm_epoch;
      SA.left = l8; SA.cost = cost; SA.out0 = accum; SA.out1 = accum2;
#ifdef VWGPU_SWEEP_DEBUG
      SA.dbg = getenv("VWGPU_SWEEP_DBG") ? atoi(getenv("VWGPU_SWEEP_DBG")) : 0;
#endif
      SA.sync = reinterpret_cast<unsigned*>(static_cast<char*>(ctx->sgm_bnd.base));
      SA.bnd = reinterpret_cast<unsigned long long*>(static_cast<char*>(ctx->sgm_bnd.base) + 256);
      VWGPU_HIP(ctx, hipMemsetAsync(SA.sync, 0, 256, st));
      const size_t lds = (32 + ((size_t)sweep_nw * SWEEP_RING + SWEEP_FRING) * slot_dwords) * 4;
      const dim3 grd((unsigned)(2 * nblk)), blk((unsigned)((sweep_nw + 1) * 64));
#define VWGPU_SWEEP(QQ) do { VWGPU_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_uniform_kernel<QQ>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                             hipLaunchKernelGGL(sweep_uniform_kernel<QQ>, grd, blk, lds, st, SA); } while (0)
      switch (sweep_q) {
        case 1: VWGPU_SWEEP(1); break; case 2: VWGPU_SWEEP(2); break; case 3: VWGPU_SWEEP(3); break; case 4: VWGPU_SWEEP(4); break;
        case 5: VWGPU_SWEEP(5); break; case 6: VWGPU_SWEEP(6); break; default: VWGPU_SWEEP(8); break;
      }
#undef VWGPU_SWEEP
#ifdef VWGPU_SWEEP_DEBUG
      if (getenv("VWGPU_SWEEP_TRACE")) {
        std::vector<unsigned long long> tr((size_t)2 * 2048 * 8);
        VWGPU_HIP(ctx, hipStreamSynchronize(st));
        VWGPU_HIP(ctx, hipMemcpyFromSymbol(tr.data(), HIP_SYMBOL(sweep_trace), tr.size() * 8));
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < nblk; ++b) for (int q = 0; q < 2; ++q) if (tr[((size_t)q * 2048 + b) * 8]) t0 = std::min(t0, tr[((size_t)q * 2048 + b) * 8]);
        for (int b = 0; b < std::min(nblk, 2048); b += std::max(1, nblk / 24))
          fprintf(stderr, "blk %4d fwd: w0 start %8.2f step8 %8.2f end %8.2f | last row start %8.2f step8 %8.2f end %8.2f | feeder first %8.2f last %8.2f us\n", b,
                  (tr[b * 8 + 0] - t0) * 0.01, (tr[b * 8 + 6] - t0) * 0.01, (tr[b * 8 + 1] - t0) * 0.01, (tr[b * 8 + 2] - t0) * 0.01, (tr[b * 8 + 7] - t0) * 0.01,
                  (tr[b * 8 + 3] - t0) * 0.01, (tr[b * 8 + 4] - t0) * 0.01, (tr[b * 8 + 5] - t0) * 0.01);
      }
#endif
    } else if (dir_paths) {
      // One direction per launch: every pixel lies on exactly one line of a launch, so the path costs are accumulated with plain
      // loads and stores, and the first direction stores (no memset).  (Tried: bands of rows with the directions pipelined over
      // them, so that a band's vectors are revisited while they sit in the memory-side cache — lines of a few hundred steps pay
      // their start-up latencies too often: 7.7 ms with 231-row bands, 15 ms with 74-row bands, against 5.9 ms unbanded.)
      vwgpu_prof_scope ps(ctx, "sgm_paths");
      int pe = (int)(((num_disp + 1) / 2 + 63) / 64);
      if (pe == 3) pe = 4;                                            // a lane's pairs must not straddle the end of the vector (stride % 8 == 0: 4 divides stride / 2)
      // dirs[]: 0 T->B, 1 B->T, 2 L->R, 3 R->L, 4 TL->BR, 5 TR->BL, 6 BL->TR, 7 BR->TL
      // (Round 4: orders that start every pass where the previous one ended — 4 7 5 6, so that it meets the last ~300 columns or rows of
      // sums in the 256 MB memory-side cache — measured 5.5 … 6.0 ms against 5.65 ms: no lever.  Bands of rows for L->R / R->L alone keep
      // the lines long but leave a band's worth of lines in flight: a line is one wavefront, 2048 lines are what hides its step latency.)
      const int order[8] = {2, 3, 0, 1, 4, 5, 6, 7};
      if (wta_in_paths) VWGPU_HIP(ctx, hipMemsetAsync(wta_flag, 0, sizeof(int), st));
      for (int q = 0; q < 8; ++q) {
        const Dir& d = dirs[order[q]];
        DirSet S;
        S.n = 1; S.rev_second = 1;
        S.dc[0] = d.dc; S.dr[0] = d.dr; S.n_first[0] = d.n_first; S.row_border[0] = d.first_is_row_border; S.second_skip[0] = d.second_skip;
        S.line0[0] = 0;
        const int nlines = d.n_first + d.n_second;
        S.line0[1] = nlines;
        if (nlines <= 0) continue;
        // the last direction takes the winners on the way (every pixel lies on exactly one of its lines)
        const int acc = q == 0 ? ACC_STORE : ((q == 7 && wta_in_paths) ? ACC_RMW_WTA : ACC_RMW);
        // lines per workgroup: neighbouring lines of the six directions that cross the rows share 128-byte lines of both volumes
        int wpb = ctx->sgm_path_mode & 15;
        if (wpb == 0) wpb = d.dr != 0 ? 4 : 1;
        CenArgs CA;
        CA.lcen = lc + (size_t)cen_or * lcw + cen_oc; CA.rcen = rcen + (size_t)cen_or * rcw + cen_oc; CA.lcw = lcw; CA.rcw = rcw;
        // round 6: the same recurrence fed through an LDS ring (path_ring_kernel) — vector strides up to 160 bytes
        if (ring_paths) {
#ifdef VWGPU_RING_DBG
          { const int dv = getenv("VWGPU_RING_DBG") ? atoi(getenv("VWGPU_RING_DBG")) : 0; VWGPU_HIP(ctx, hipMemcpyToSymbolAsync(HIP_SYMBOL(ring_dbg), &dv, sizeof dv, 0, hipMemcpyHostToDevice, st)); }
#endif
          const int rwpb = (ctx->sgm_path_mode & 15) >= 8 ? 8 : 4, cluster = 1 << ((ctx->sgm_path_mode >> 8) & 7);
#define VWGPU_RING5(E, A, RCC, WP, C) do { const size_t acb = (size_t)12 * ustride, ccb = (size_t)6 * 16 * ((ustride + 15) / 16); \
          const size_t rl = (size_t)WP * (C ? RCC * acb + ccb + 272 + 384 : RCC * (acb + ccb) + 256); \
          VWGPU_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(path_ring_kernel<E, A, RCC, WP, C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rl)); \
          hipLaunchKernelGGL((path_ring_kernel<E, A, RCC, WP, C>), dim3((nlines + WP - 1) / WP), dim3(64 * WP), rl, st, g, S, ustride, l8, lw, min_col, min_row, cost, accum, \
                             (unsigned)p1, (unsigned)p2, out_disp, full_search, wta_flag, cluster, CA); } while (0)
#define VWGPU_RING2(E, A) do { if (rwpb == 8) VWGPU_RING5(E, A, 3, 8, false); else if (ring_cen) VWGPU_RING5(E, A, 4, 4, true); else VWGPU_RING5(E, A, 4, 4, false); } while (0)
#define VWGPU_RING(E) do { if (acc == ACC_STORE) VWGPU_RING2(E, ACC_STORE); else if (acc == ACC_RMW_WTA) VWGPU_RING2(E, ACC_RMW_WTA); else VWGPU_RING2(E, ACC_RMW); } while (0)
          if (pe == 1) VWGPU_RING(1); else VWGPU_RING(2);
#undef VWGPU_RING
#undef VWGPU_RING2
#undef VWGPU_RING5
          if (acc == ACC_RMW_WTA) wta_done = true;
          continue;
        }
#define VWGPU_PATH_DIR3(E, A, WP) hipLaunchKernelGGL((path_uniform_reg_kernel<E, A, VWGPU_PATH_KC, WP>), dim3((nlines + WP - 1) / WP), dim3(64 * WP), 0, st, \
                                 g, S, ustride, l8, lw, min_col, min_row, cost, accum, (unsigned)p1, (unsigned)p2, out_disp, full_search, wta_flag)
#define VWGPU_PATH_DIR2(E, A) do { if (wpb >= 4) VWGPU_PATH_DIR3(E, A, 4); else VWGPU_PATH_DIR3(E, A, 1); } while (0)
#define VWGPU_PATH_DIR(E) do { if (acc == ACC_STORE) VWGPU_PATH_DIR2(E, ACC_STORE); else if (acc == ACC_RMW_WTA) VWGPU_PATH_DIR2(E, ACC_RMW_WTA); \
                               else VWGPU_PATH_DIR2(E, ACC_RMW); } while (0)
        switch (pe) { case 1: VWGPU_PATH_DIR(1); break; case 2: VWGPU_PATH_DIR(2); break; default: VWGPU_PATH_DIR(4); break; }
        if (acc == ACC_RMW_WTA) wta_done = true;
#undef VWGPU_PATH_DIR
#undef VWGPU_PATH_DIR2
#undef VWGPU_PATH_DIR3
      }
    } else
    for (int first = 0; first < 8; first += together ? 8 : 1) {
      DirSet D;
      D.rev_second = 0;
      D.n = together ? 8 : 1;
      int lines = 0;
      for (int q = 0; q < D.n; ++q) {
        const Dir& d = dirs[first + q];
        D.dc[q] = d.dc; D.dr[q] = d.dr; D.n_first[q] = d.n_first; D.row_border[q] = d.first_is_row_border; D.second_skip[q] = d.second_skip;
        D.line0[q] = lines;
        lines += d.n_first + d.n_second;
      }
      D.line0[D.n] = lines;
      if (lines <= 0) continue;
      vwgpu_prof_scope ps(ctx, together ? "sgm_paths" : "sgm_path");
      if (uniform) {
        // (2-D searches only: one search row is served by path_uniform_reg_kernel above)
#define VWGPU_PATH_U(E) hipLaunchKernelGGL((path_uniform_kernel<E, false>), dim3(lines), dim3(64), ulds, st, g, D, K, ustride, \
                               l8, lw, min_col, min_row, cost, accum, (unsigned)p1, (unsigned)p2)
        switch (ept) {
          case 1: VWGPU_PATH_U(1); break; case 2: VWGPU_PATH_U(2); break; case 3: VWGPU_PATH_U(3); break; case 4: VWGPU_PATH_U(4); break;
          case 5: VWGPU_PATH_U(5); break; case 6: VWGPU_PATH_U(6); break; case 7: VWGPU_PATH_U(7); break; default: VWGPU_PATH_U(8); break;
        }
#undef VWGPU_PATH_U
      } else {
#define VWGPU_PATH_IP(RR) hipLaunchKernelGGL(path_inplace_kernel<RR>, dim3(lines), dim3(64), ((size_t)num_disp + 2 + 256) * sizeof(uint16_t), st, g, D, l8, lw, \
                                             min_col, min_row, bounds, starts, cost, accum, (unsigned)p1, (unsigned)p2)
        // Several lines per wavefront (path_multi_kernel) on the large levels whose boxes nearly all fit a quarter or a half of one.  Four
        // lines when the large boxes hold little of the volume as well (a line inside a patch of them occupies the whole wavefront);
        // levels of fewer than 6 000 lines keep one line per wavefront.  SGM_PATH_MODE: bit 6 off, bit 4 / bit 7 force
        // four / two lines.
        constexpr int spread = 64;
#define VWGPU_PATH_MULTI(SUBL) hipLaunchKernelGGL(path_multi_kernel<SUBL>, dim3((unsigned)(((size_t)lines + (size_t)(64 / SUBL) * spread - 1) / ((size_t)(64 / SUBL) * spread) * spread)), dim3(64), \
                                                  ((size_t)(64 / SUBL + 1) * ((num_disp + 1) & ~1) + 256) * sizeof(uint16_t), st, \
                                                  g, D, lines, spread, l8, lw, min_col, min_row, bounds, starts, cost, accum, (unsigned)p1, (unsigned)p2)
        const int pm = ctx->sgm_path_mode;
        const bool multi_ok = num_disp <= 4096 && !(pm & 64);           // (five full-range vectors of u16 in LDS; the cell index shares a register with the value)
        const bool large = lines >= 6000;                               // (a 512^2 level; below, the launch is short of wavefronts either way: no gain measured)
        if (multi_ok && ((pm & 16) || (!(pm & 128) && large && small16 * 5 >= npix * 4 && (n_total - cells16) * 100 <= n_total * 15))) VWGPU_PATH_MULTI(16);
        else if (multi_ok && ((pm & 128) || (large && small32 * 5 >= npix * 4))) VWGPU_PATH_MULTI(32);
        else
#undef VWGPU_PATH_MULTI
        if (num_disp <= 64) VWGPU_PATH_IP(1);
        else if (num_disp <= 128) VWGPU_PATH_IP(2);
        else if (num_disp <= 256) VWGPU_PATH_IP(4);
        else if (num_disp <= 512) VWGPU_PATH_IP(8);
        else if (num_disp <= 1024) VWGPU_PATH_IP(16);
        else                                        // very large 2-D searches: the three-phase kernel, any size
          hipLaunchKernelGGL(path_kernel, dim3(lines), dim3(64), lds, st, g, D, l8, lw, min_col, min_row, bounds, starts, cost, accum,
                             (unsigned)p1, (unsigned)p2);
#undef VWGPU_PATH_IP
      }
    }
  }
  {
    vwgpu_prof_scope ps(ctx, "sgm_wta");
    const size_t lds = (size_t)4 * 2 * num_disp * sizeof(uint16_t);
    const uint8_t* todo = nullptr;
    const int* any_todo = nullptr;
    if (wta_done) {                                         // the last direction took the unique minima; ties go to the general kernel
      todo = full_search; any_todo = wta_flag;
    } else if (uniform && g.num_dy == 1 && num_disp <= 256) {      // unique minima straight from the packed vectors; ties go to the general kernel
      int* flag = reinterpret_cast<int*>(mm + 6);
      VWGPU_HIP(ctx, hipMemsetAsync(flag, 0, sizeof(int), st));
      hipLaunchKernelGGL(wta_uniform_kernel, dim3((unsigned)((npix + 4 * WTAU_PPW - 1) / (4 * WTAU_PPW))), dim3(256), 0, st, npix, (int)num_disp, ustride,
                         g.min_dx, g.min_dy, accum, accum2, out_disp, full_search, flag);
      todo = full_search; any_todo = flag;
    }
    hipLaunchKernelGGL(wta_kernel, dim3((unsigned)((npix + 4 * WTA_PPW - 1) / (4 * WTA_PPW))), dim3(256), lds, st, bounds, starts, npix, (int)num_disp, accum, out_disp,
                       todo, any_todo);
  }
  if (out_sub) {
    vwgpu_prof_scope ps(ctx, "sgm_subpixel");
    hipLaunchKernelGGL(subpixel_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, P->subpixel_mode, bounds, starts, npix, accum, accum2, out_disp, out_sub);
  }
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}
