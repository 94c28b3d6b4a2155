// bm_zones.hip — all search zones of one pyramid level in ONE launch.
//
// PyramidCorrelationView::prerasterize runs calc_disparity once per SearchParam zone
// (src/vw/Stereo/CorrelationView.cc:596-700); at level 0 of a 1024^2 tile that is ~2000 zones of ~32x32 pixels with
// ~5x5 disparities each — as separate launches they cost ~600 ms of launch latency.  Here a zone is a row of a device
// table and a workgroup serves one 32x32 output tile of one zone:
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
// A second tiny kernel applies the per-zone L/R consistency check (Correlate.cc:1441-1502) and the
// `+= zone.disparity_range().min()` offset (CorrelationView.cc:696-697) for every zone at once.
//
// Roofline: LDS-bandwidth bound (~2*kx 4-byte + ky 8-byte LDS reads per pixel*disparity); the point of this kernel is
// launch count, the level-0 work of a refined pyramid is only ~25 disparities per pixel.
#include <algorithm>
#include <cstring>
#include <vector>

#include "vwgpu_internal.h"

namespace {

constexpr int ZT = 32;            // output tile side
constexpr int ZTHREADS = 256;

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

struct PrecView {           // 1 / box-sum(img^2) over origins [x0, x0+w) x [y0, y0+h)
  const double* p; int x0, y0, w, h;
};

// prec(x, y) = 1.0 / sum_{ky x kx} img(clamp)^2 for window origins (x0 + i, y0 + j).  A 64 x 4 output tile: the squares of its
// (64 + kx - 1) x (4 + ky - 1) pixels go to LDS once, then row sums and column sums (the direct form read kx * ky floats per output
// through the L1: 0.3 ms per 1024^2 NCC tile).  Only used on data whose box sums are exact in any order (vwgpu_sums_order_free).
__global__ void __launch_bounds__(256)
zone_precision_kernel(const float* __restrict__ img, int w, int h, int kx, int ky,
                      double* __restrict__ prec, int x0, int y0, int pw, int ph) {
  extern __shared__ double zp_sm[];
  const int tw = 64 + kx - 1, th = 4 + ky - 1;
  double* sq = zp_sm;                 // th x tw squares
  double* hs = zp_sm + (size_t)th * tw;   // th x 64 row sums
  const int tid = threadIdx.y * 64 + threadIdx.x;
  const int bx = blockIdx.x * 64, by = blockIdx.y * 4;
  for (int i = tid; i < tw * th; i += 256) {
    const int r = i / tw, c = i - r * tw;
    int xx = x0 + bx + c; xx = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
    int yy = y0 + by + r; yy = yy < 0 ? 0 : (yy >= h ? h - 1 : yy);
    const float v = img[(size_t)yy * w + xx];
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
  const int i = bx + threadIdx.x, j = by + threadIdx.y;
  if (i >= pw || j >= ph) return;
  double s = 0.0;
  for (int b = 0; b < ky; ++b) s += hs[(threadIdx.y + b) * 64 + threadIdx.x];
  prec[(size_t)j * pw + i] = 1.0 / s;
}

// KS > 0: a square KS x KS window known at compile time — the horizontal and vertical window sums are unrolled (with run-time
// sizes the loop overhead outweighed the sums, as PMC showed for bm_generic).  KS == 0: any kx, ky.
// ACC: the type of the window sums.  float64 is the reference's; float32 is taken when every intermediate value is exactly representable in 24
// bits as well (vwgpu_sums_bits <= 24: byte imagery under SAD) — then both give the same numbers, the LDS planes are half as large and the sums
// full-rate.  The compare chain runs on doubles either way.
template <int COST, int KS, typename ACC>
__global__ void __launch_bounds__(ZTHREADS)
bm_zones_kernel(const float* __restrict__ A, int aw, int ah, const float* __restrict__ B, int bw, int bh,
                int kx, int ky, const vwgpu_zone_task* __restrict__ zones, const int2* __restrict__ tiles,
                int sxc, PrecView pa, PrecView pb, int32_t* __restrict__ out) {
  extern __shared__ char smem[];
  const int PW = ZT + kx - 1, PH = ZT + ky - 1, RW = PW + sxc - 1;
  float* Lp = reinterpret_cast<float*>(smem);                    // PH x PW
  float* Rp = Lp + PH * PW;                                      // PH x RW
  ACC* H = reinterpret_cast<ACC*>(smem + (((size_t)(PH * PW + PH * RW) * 4 + 7) & ~size_t(7)));   // 2 x PH x ZT

  const int2 tl = tiles[blockIdx.x];
  const vwgpu_zone_task z = zones[tl.x];
  const int ox = (tl.y & 0xffff) * ZT, oy = (tl.y >> 16) * ZT;
  const int tw = min(ZT, z.zw - ox), th = min(ZT, z.zh - oy);
  const int pw = tw + kx - 1, ph = th + ky - 1;
  const int t = threadIdx.x;
  const int c = t & 31, y0 = (t >> 5) * 4;

  for (int i = t; i < ph * pw; i += ZTHREADS) {
    const int r = i / pw, q = i - r * pw;
    int xx = z.ax + ox + q; xx = xx < 0 ? 0 : (xx >= aw ? aw - 1 : xx);
    int yy = z.ay + oy + r; yy = yy < 0 ? 0 : (yy >= ah ? ah - 1 : yy);
    Lp[r * PW + q] = A[(size_t)yy * aw + xx];
  }
  double best[4], worst[4], lprec[4];
  int bdx[4], bdy[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    best[m] = worst[m] = 0.0; bdx[m] = bdy[m] = 0; lprec[m] = 0.0;
    if (COST == VWGPU_CROSS_CORRELATION && c < tw && y0 + m < th)
      lprec[m] = pa.p[(size_t)(z.ay + oy + y0 + m - pa.y0) * pa.w + (z.ax + ox + c - pa.x0)];
  }
  int hb = 0;
  for (int dy = 0; dy < z.sy; ++dy) {
    for (int dx0 = 0; dx0 < z.sx; dx0 += sxc) {
      const int nd = min(sxc, z.sx - dx0);
      const int rwid = pw + nd - 1;
      __syncthreads();                                            // everyone done with the previous right patch
      for (int i = t; i < ph * rwid; i += ZTHREADS) {
        const int r = i / rwid, q = i - r * rwid;
        int xx = z.bx + ox + dx0 + q; xx = xx < 0 ? 0 : (xx >= bw ? bw - 1 : xx);
        int yy = z.by + oy + dy + r; yy = yy < 0 ? 0 : (yy >= bh ? bh - 1 : yy);
        Rp[r * RW + q] = B[(size_t)yy * bw + xx];
      }
      __syncthreads();
      for (int d = 0; d < nd; ++d) {
        ACC* Hc = H + hb * (PH * ZT);
        if (KS > 0) {
          // Four adjacent columns per thread: KS + 3 cost elements are formed once and the window slides (s' = s - e[j] + e[j + KS]),
          // 2 (KS + 3) LDS reads and 3 KS + ... adds for four sums instead of 8 KS reads and 4 KS adds.  The slide is exact here:
          // this kernel only runs on data whose running sums are exactly representable (vwgpu_sums_order_free).
          for (int i = t; i < ph * (ZT / 4); i += ZTHREADS) {
            const int r = i >> 3, q = (i & 7) * 4;
            if (q < tw) {
              const float* lp = Lp + r * PW + q;
              const float* rp = Rp + r * RW + q + d;
              ACC e[KS > 0 ? KS + 3 : 1];
#pragma unroll
              for (int a = 0; a < KS + 3; ++a) e[a] = zcost<COST, ACC>(lp[a], rp[a]);
              ACC s0 = 0;
#pragma unroll
              for (int a = 0; a < KS; ++a) s0 += e[a];
              const ACC s1 = s0 - e[0] + e[KS], s2 = s1 - e[1] + e[KS + 1], s3 = s2 - e[2] + e[KS + 2];
              ACC* h = Hc + r * ZT + q;
              h[0] = s0; h[1] = s1; h[2] = s2; h[3] = s3;
            }
          }
        } else {
        for (int i = t; i < ph * ZT; i += ZTHREADS) {             // horizontal sums
          const int r = i >> 5, q = i & 31;
          if (q < tw) {
            const float* lp = Lp + r * PW + q;
            const float* rp = Rp + r * RW + q + d;
            ACC s = 0;
            for (int a = 0; a < kx; ++a) s += zcost<COST, ACC>(lp[a], rp[a]);
            Hc[r * ZT + q] = s;
          }
        }
        }
        __syncthreads();
        if (c < tw) {
          const int dx = dx0 + d;
          const bool first = (dx == 0 && dy == 0);
          ACC vs[4] = {0, 0, 0, 0};
          if (KS > 0) {                                           // the same slide down the rows (rows beyond th hold stale planes: unused)
            ACC h[KS > 0 ? KS + 3 : 1];
#pragma unroll
            for (int b = 0; b < KS + 3; ++b) h[b] = Hc[min(y0 + b, PH - 1) * ZT + c];
#pragma unroll
            for (int b = 0; b < KS; ++b) vs[0] += h[b];
            vs[1] = vs[0] - h[0] + h[KS]; vs[2] = vs[1] - h[1] + h[KS + 1]; vs[3] = vs[2] - h[2] + h[KS + 2];
          }
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const int y = y0 + m;
            if (y < th) {
              ACC sa = vs[m];
              if (KS == 0) {
                for (int b = 0; b < ky; ++b) sa += Hc[(y + b) * ZT + c];
              }
              double s = (double)sa;
              if (COST == VWGPU_CROSS_CORRELATION)
                s *= sqrt(lprec[m] * pb.p[(size_t)(z.by + oy + y + dy - pb.y0) * pb.w + (z.bx + ox + c + dx - pb.x0)]);
              if (first) { best[m] = worst[m] = s; }
              else if (zbetter<COST>(s, best[m])) { best[m] = s; bdx[m] = dx; bdy[m] = dy; }
              else if (!zbetter<COST>(s, worst[m])) { worst[m] = s; }
            }
          }
        }
        hb ^= 1;                                                  // next disparity writes the other plane
      }
    }
  }
  if (c < tw) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int y = y0 + m;
      if (y < th) {
        int32_t* o = out + ((size_t)z.out_off + (size_t)(oy + y) * z.out_stride + ox + c) * 3;
        o[0] = bdx[m] + z.addx; o[1] = bdy[m] + z.addy;
        o[2] = (best[m] == worst[m]) ? 0 : 0x7fffffff;
      }
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

int upload_tables(vwgpu_ctx* ctx, const vwgpu_zone_task* zones, int n, std::vector<int2> const& tiles,
                  const vwgpu_zone_task** d_zones, const int2** d_tiles) {
  const size_t zb = vwgpu_align_up((size_t)n * sizeof(vwgpu_zone_task), 256), tb = tiles.size() * sizeof(int2);
  // the previous launch may still be reading the table: alternate between two halves of the arena
  const size_t half = vwgpu_align_up(zb + tb, 4096);
  if (ctx->ztab.cap < 2 * half) {
    int rc = vwgpu_arena_reserve(ctx, &ctx->ztab, 2 * half + (1 << 20));
    if (rc) return rc;
  }
  ctx->ztab_parity ^= 1;
  char* base = static_cast<char*>(ctx->ztab.base) + (ctx->ztab_parity ? ctx->ztab.cap / 2 : 0);
  if (char* h = static_cast<char*>(vwgpu_host_ring(ctx, zb + tb))) {   // one asynchronous copy from pinned memory
    memcpy(h, zones, (size_t)n * sizeof(vwgpu_zone_task));
    memcpy(h + zb, tiles.data(), tb);
    VWGPU_HIP(ctx, hipMemcpyAsync(base, h, zb + tb, hipMemcpyHostToDevice, ctx->stream));
  } else {
    VWGPU_HIP(ctx, hipMemcpyAsync(base, zones, (size_t)n * sizeof(vwgpu_zone_task), hipMemcpyHostToDevice, ctx->stream));
    VWGPU_HIP(ctx, hipMemcpyAsync(base + zb, tiles.data(), tb, hipMemcpyHostToDevice, ctx->stream));
  }
  *d_zones = reinterpret_cast<const vwgpu_zone_task*>(base);
  *d_tiles = reinterpret_cast<const int2*>(base + zb);
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

}  // namespace

bool vwgpu_bm_zones_supported(int kx, int ky) {
  // LDS: left patch + right patch with at least 8 disparities per chunk + two sum planes within 64 KB
  const size_t PW = ZT + kx - 1, PH = ZT + ky - 1;
  return (PH * PW + PH * (PW + 7)) * 4 + 2 * PH * ZT * 8 + 16 <= 64 * 1024;
}

int vwgpu_launch_bm_zones(vwgpu_ctx* ctx, int cost_type, const float* A, int aw, int ah, const float* B, int bw, int bh,
                          int kx, int ky, const vwgpu_zone_task* zones, int n, int32_t* out, int f32_sums) {
  if (cost_type == VWGPU_CROSS_CORRELATION) f32_sums = 0;        // (its sums are scaled in float64 anyway)
  const size_t accb = f32_sums ? 4 : 8;
  if (n <= 0) return VWGPU_OK;
  if (!vwgpu_bm_zones_supported(kx, ky)) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "bm_zones: kernel %dx%d too large", kx, ky);
  std::vector<int2> tiles;
  build_tiles(zones, n, tiles);
  if (tiles.empty()) return VWGPU_OK;
  int max_sx = 1;
  int ax0 = INT32_MAX, ay0 = INT32_MAX, ax1 = INT32_MIN, ay1 = INT32_MIN, bx0 = INT32_MAX, by0 = INT32_MAX, bx1 = INT32_MIN, by1 = INT32_MIN;
  for (int i = 0; i < n; ++i) {
    const vwgpu_zone_task& z = zones[i];
    if (z.zw <= 0 || z.zh <= 0) continue;
    max_sx = std::max(max_sx, z.sx);
    ax0 = std::min(ax0, z.ax); ay0 = std::min(ay0, z.ay); ax1 = std::max(ax1, z.ax + z.zw); ay1 = std::max(ay1, z.ay + z.zh);
    bx0 = std::min(bx0, z.bx); by0 = std::min(by0, z.by);
    bx1 = std::max(bx1, z.bx + z.zw + z.sx - 1); by1 = std::max(by1, z.by + z.zh + z.sy - 1);
  }
  const size_t PW = ZT + kx - 1, PH = ZT + ky - 1;
  const size_t fixed = PH * PW * 4 + 2 * PH * ZT * accb + 16;
  int sxc = (int)std::min<size_t>((size_t)max_sx, ((64 * 1024 - fixed) / (PH * 4)) - PW + 1);
  if (sxc < 1) sxc = 1;
  const size_t lds = (((PH * PW + PH * (PW + sxc - 1)) * 4 + 7) & ~size_t(7)) + 2 * PH * ZT * accb;

  PrecView pa{nullptr, 0, 0, 0, 0}, pb{nullptr, 0, 0, 0, 0};
  if (cost_type == VWGPU_CROSS_CORRELATION) {
    pa.x0 = ax0; pa.y0 = ay0; pa.w = ax1 - ax0; pa.h = ay1 - ay0;
    pb.x0 = bx0; pb.y0 = by0; pb.w = bx1 - bx0; pb.h = by1 - by0;
    const size_t na = vwgpu_align_up((size_t)pa.w * pa.h * 8, 256), nb = vwgpu_align_up((size_t)pb.w * pb.h * 8, 256);
    int rc = vwgpu_arena_reserve(ctx, &ctx->scratch, na + nb);
    if (rc) return rc;
    double* da = static_cast<double*>(ctx->scratch.base);
    double* db = reinterpret_cast<double*>(static_cast<char*>(ctx->scratch.base) + na);
    vwgpu_prof_scope ps(ctx, "zone_precision");
    const size_t zp_lds = ((size_t)(64 + kx - 1) * (4 + ky - 1) + (size_t)(4 + ky - 1) * 64) * sizeof(double);
    hipLaunchKernelGGL(zone_precision_kernel, dim3((pa.w + 63) / 64, (pa.h + 3) / 4), dim3(64, 4), zp_lds, ctx->stream, A, aw, ah, kx, ky, da, pa.x0, pa.y0, pa.w, pa.h);
    hipLaunchKernelGGL(zone_precision_kernel, dim3((pb.w + 63) / 64, (pb.h + 3) / 4), dim3(64, 4), zp_lds, ctx->stream, B, bw, bh, kx, ky, db, pb.x0, pb.y0, pb.w, pb.h);
    pa.p = da; pb.p = db;
  }
  const vwgpu_zone_task* dz; const int2* dt;
  int rc = upload_tables(ctx, zones, n, tiles, &dz, &dt);
  if (rc) return rc;
  vwgpu_prof_scope ps(ctx, "bm_zones");
  const dim3 grd((unsigned)tiles.size()), blk(ZTHREADS);
#define VW_ZN(C, K) do { if (f32_sums) hipLaunchKernelGGL((bm_zones_kernel<C, K, float>), grd, blk, lds, ctx->stream, A, aw, ah, B, bw, bh, kx, ky, dz, dt, sxc, pa, pb, out); \
                         else hipLaunchKernelGGL((bm_zones_kernel<C, K, double>), grd, blk, lds, ctx->stream, A, aw, ah, B, bw, bh, kx, ky, dz, dt, sxc, pa, pb, out); } while (0)
#define VW_ZN_K(C) do { switch (kx == ky ? kx : 0) { case 3: VW_ZN(C, 3); break; case 5: VW_ZN(C, 5); break; case 7: VW_ZN(C, 7); break; \
                                                     case 9: VW_ZN(C, 9); break; case 11: VW_ZN(C, 11); break; case 13: VW_ZN(C, 13); break; \
                                                     default: VW_ZN(C, 0); break; } } while (0)
  switch (cost_type) {
    case VWGPU_CROSS_CORRELATION: VW_ZN_K(VWGPU_CROSS_CORRELATION); break;
    case VWGPU_SQUARED_DIFFERENCE: VW_ZN_K(VWGPU_SQUARED_DIFFERENCE); break;
    default: VW_ZN_K(VWGPU_ABSOLUTE_DIFFERENCE); break;
  }
#undef VW_ZN_K
#undef VW_ZN
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
