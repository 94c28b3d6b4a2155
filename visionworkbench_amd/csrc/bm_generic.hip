// bm_generic.hip — reference-arithmetic block matching for any float input (SAD / SSD / NCC).
//
// Replaces best_of_search_convolution (src/vw/Stereo/Correlation.cc:33-137) + fast_box_sum
// (src/vw/Stereo/Algorithms.h:43-129) + the cost functors (src/vw/Stereo/CostFunctions.h:72-236) with one
// kernel that never materialises a cost image:
//   * a workgroup owns an output tile of (256-kx+1) columns x TY rows; thread c owns tile column c;
//   * for each disparity (dy outer, dx inner — the reference's raster order) the thread keeps the running
//     float64 column sum of its column exactly like fast_box_sum does (`+= front; -= back`), publishes it
//     in a double-buffered LDS row (one barrier per output row) and the first 256-kx+1 threads add kx
//     neighbouring column sums to get the window cost;
//   * best / worst / disparity live in registers for the TY rows of the thread's column and follow the
//     reference's compare chain literally (strict compare, first wins, `else if` for worst), so NaN costs
//     (NCC over an all-zero window) behave the same way;
//   * cost elements are computed in FLOAT and widened (CostFunctions.h:79-81,94-101,120-127); the build
//     uses -ffp-contract=off so no FMA is formed.
// Sums are order-independent (hence bit-exact against the reference) whenever every partial sum is exactly
// representable — in particular for integer-valued pixels.  For arbitrary floats the
// reference's own serial running sums are position dependent and no parallel algorithm reproduces them.
//
// Roofline note: this is the fallback family; it is LDS/barrier bound (~10 LDS ops per pixel*disparity).
// The headline configuration is served by bm_sad_u8.hip.
#include "vwgpu_internal.h"

namespace {

constexpr int GEN_THREADS = 256;
constexpr int GEN_TY = 16;

template <int COST>
__device__ __forceinline__ double cost_elem(float a, float b) {
  if (COST == VWGPU_CROSS_CORRELATION) return (double)(a * b);
  if (COST == VWGPU_SQUARED_DIFFERENCE) { float d = a - b; return (double)(d * d); }
  return (double)fabsf(a - b);
}

template <int COST>
__device__ __forceinline__ bool better(double c, double q) {
  return COST == VWGPU_CROSS_CORRELATION ? (c > q) : (c < q);
}

// NCC side-car: prec(x,y) = 1.0 / sum_{ky x kx}(img^2) in float64 (NCCCost ctor, CostFunctions.h:214-219).
__global__ void ncc_precision_kernel(const float* __restrict__ img, ptrdiff_t stride, int w, int h,
                                     int kx, int ky, double* __restrict__ prec, const int* __restrict__ run_flag) {
  if (run_flag && *run_flag == 0) return;  // the fast path already produced the result
  const int ow = w - kx + 1, oh = h - ky + 1;
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= ow || y >= oh) return;
  double s = 0.0;
  for (int j = 0; j < ky; ++j) {
    const float* row = img + (ptrdiff_t)(y + j) * stride + x;
    for (int i = 0; i < kx; ++i) { float v = row[i]; s += (double)(v * v); }
  }
  prec[(size_t)y * ow + x] = 1.0 / s;
}

// KX > 0: the window width is a compile-time constant, so the horizontal sum of each evaluation is kx unrolled LDS reads and
// adds; with a run-time width the loop overhead was ~100 of the ~137 VALU instructions per evaluation (PMC).  KX == 0: any width.
template <int COST, int KX>
__global__ void __launch_bounds__(GEN_THREADS)
bm_generic_kernel(const float* __restrict__ left, ptrdiff_t ls, int lw, int lh,
                  const float* __restrict__ right, ptrdiff_t rs,
                  int kx, int ky, int sx, int sy,
                  const double* __restrict__ lprec, const double* __restrict__ rprec, int rpw,
                  int32_t* __restrict__ out, ptrdiff_t os, int ow, int oh,
                  const int* __restrict__ run_flag) {
  if (run_flag && *run_flag == 0) return;  // the fast path already produced the result
  // Column sums of all GEN_TY output rows of one disparity go to an LDS plane at once: two barriers per disparity instead of
  // one per (disparity, row) with a global-load round trip inside each.  One 32 KB plane (not two alternating ones): four
  // workgroups per CU hide the LDS / global latency better than a saved barrier does.
  __shared__ double plane[1][GEN_TY][GEN_THREADS];

  const int c = threadIdx.x;
  const int tile_w = GEN_THREADS - kx + 1;
  const int x0 = blockIdx.x * tile_w;
  const int y0 = blockIdx.y * GEN_TY;
  const int col = x0 + c;
  const bool col_ok = col < lw;
  const bool out_col = (c < tile_w) && (col < ow);

  double best[GEN_TY], worst[GEN_TY];
  int bdx[GEN_TY], bdy[GEN_TY];
#pragma unroll
  for (int y = 0; y < GEN_TY; ++y) { best[y] = 0; worst[y] = 0; bdx[y] = 0; bdy[y] = 0; }

  int p = 0;
  for (int dy = 0; dy < sy; ++dy) {
    for (int dx = 0; dx < sx; ++dx) {
      // running row pointers (no 64-bit multiply per access): *_b = the row that leaves the window, *_f = the row that enters
      const float* lb = left + (ptrdiff_t)y0 * ls + col;
      const float* rb = right + (ptrdiff_t)(y0 + dy) * rs + col + dx;
      const float* lf = lb;
      const float* rf_ = rb;
      // column sum over the first ky rows of the tile (Algorithms.h:62-75), then slid down the tile's rows
      // (Algorithms.h:100-103: two statements)
      double cs = 0.0;
      for (int j = 0; j < ky; ++j) {
        if (col_ok && y0 + j < lh) cs += cost_elem<COST>(*lf, *rf_);
        lf += ls; rf_ += rs;
      }
#pragma unroll
      for (int y = 0; y < GEN_TY; ++y) {
        plane[p][y][c] = cs;
        if (col_ok && y0 + y + ky < lh) {
          cs += cost_elem<COST>(*lf, *rf_);
          cs -= cost_elem<COST>(*lb, *rb);
        }
        lf += ls; rf_ += rs; lb += ls; rb += rs;
      }
      __syncthreads();                                    // plane complete
      const bool first = (dx == 0 && dy == 0);
      if (out_col) {
#pragma unroll
        for (int y = 0; y < GEN_TY; ++y) {
          const int oy = y0 + y;
          if (oy < oh) {
            double s = 0.0;
            if (KX > 0) {
#pragma unroll
              for (int i = 0; i < KX; ++i) s += plane[p][y][c + i];
            } else {
              for (int i = 0; i < kx; ++i) s += plane[p][y][c + i];
            }
            if (COST == VWGPU_CROSS_CORRELATION) {
              // cost_metric *= sqrt(left_precision * crop(right_precision, bbox + disparity))  (:227-231)
              s *= sqrt(lprec[(size_t)oy * ow + col] * rprec[(size_t)(oy + dy) * rpw + col + dx]);
            }
            if (first) {
              best[y] = worst[y] = s;                       // Correlation.cc:110-117
            } else if (better<COST>(s, best[y])) {          // :95-98
              best[y] = s; bdx[y] = dx; bdy[y] = dy;
            } else if (!better<COST>(s, worst[y])) {        // :99-102
              worst[y] = s;
            }
          }
        }
      }
      __syncthreads();                                    // plane consumed
    }
  }

  if (out_col) {
#pragma unroll
    for (int y = 0; y < GEN_TY; ++y) {
      const int oy = y0 + y;
      if (oy < oh) {
        int32_t* o = out + ((ptrdiff_t)oy * os + col) * 3;
        o[0] = bdx[y];
        o[1] = bdy[y];
        o[2] = (best[y] == worst[y]) ? 0 : 0x7fffffff;    // validity pass, Correlation.cc:121-133
      }
    }
  }
}

}  // namespace

int vwgpu_launch_bm_generic_flag(vwgpu_ctx* ctx, int cost_type,
                                 const float* left, int lw, int lh, ptrdiff_t ls,
                                 const float* right, int rw, int rh, ptrdiff_t rs,
                                 int kx, int ky, int sx, int sy, int32_t* out, ptrdiff_t os,
                                 const int* run_flag) {
  (void)rw; (void)rh;
  const int ow = lw - kx + 1, oh = lh - ky + 1;
  if (kx > GEN_THREADS / 2) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "kernel width %d > %d", kx, GEN_THREADS / 2);

  double* lprec = nullptr;
  double* rprec = nullptr;
  int rpw = 0;
  if (cost_type == VWGPU_CROSS_CORRELATION) {
    const int rcw = lw + sx - 1, rch = lh + sy - 1;
    rpw = rcw - kx + 1;
    const int rph = rch - ky + 1;
    const size_t lbytes = vwgpu_align_up((size_t)ow * oh * sizeof(double), 256);
    const size_t rbytes = vwgpu_align_up((size_t)rpw * rph * sizeof(double), 256);
    int rc = vwgpu_arena_reserve(ctx, &ctx->scratch, lbytes + rbytes);
    if (rc) return rc;
    lprec = reinterpret_cast<double*>(ctx->scratch.base);
    rprec = reinterpret_cast<double*>(reinterpret_cast<char*>(lprec) + lbytes);
    dim3 blk(64, 4);
    {
      vwgpu_prof_scope ps(ctx, "ncc_precision_left");
      dim3 grd((ow + 63) / 64, (oh + 3) / 4);
      hipLaunchKernelGGL(ncc_precision_kernel, grd, blk, 0, ctx->stream, left, ls, lw, lh, kx, ky, lprec, run_flag);
    }
    {
      vwgpu_prof_scope ps(ctx, "ncc_precision_right");
      dim3 grd((rpw + 63) / 64, (rph + 3) / 4);
      hipLaunchKernelGGL(ncc_precision_kernel, grd, blk, 0, ctx->stream, right, rs, rcw, rch, kx, ky, rprec, run_flag);
    }
  }

  const int tile_w = GEN_THREADS - kx + 1;
  dim3 grd((ow + tile_w - 1) / tile_w, (oh + GEN_TY - 1) / GEN_TY);
  dim3 blk(GEN_THREADS);
  vwgpu_prof_scope ps(ctx, "bm_generic");
#define VW_GEN(C, K) hipLaunchKernelGGL((bm_generic_kernel<C, K>), grd, blk, 0, ctx->stream, \
                                       left, ls, lw, lh, right, rs, kx, ky, sx, sy, lprec, rprec, rpw, out, os, ow, oh, run_flag)
#define VW_GEN_K(C) do { switch (kx) { case 3: VW_GEN(C, 3); break; case 5: VW_GEN(C, 5); break; case 7: VW_GEN(C, 7); break; \
                                      case 9: VW_GEN(C, 9); break; case 11: VW_GEN(C, 11); break; case 13: VW_GEN(C, 13); break; \
                                      case 15: VW_GEN(C, 15); break; default: VW_GEN(C, 0); break; } } while (0)
  switch (cost_type) {
    case VWGPU_CROSS_CORRELATION: VW_GEN_K(VWGPU_CROSS_CORRELATION); break;
    case VWGPU_SQUARED_DIFFERENCE: VW_GEN_K(VWGPU_SQUARED_DIFFERENCE); break;
    default: VW_GEN_K(VWGPU_ABSOLUTE_DIFFERENCE); break;
  }
#undef VW_GEN_K
#undef VW_GEN
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}

int vwgpu_launch_bm_generic(vwgpu_ctx* ctx, int cost_type,
                            const float* left, int lw, int lh, ptrdiff_t ls,
                            const float* right, int rw, int rh, ptrdiff_t rs,
                            int kx, int ky, int sx, int sy, int32_t* out, ptrdiff_t os) {
  return vwgpu_launch_bm_generic_flag(ctx, cost_type, left, lw, lh, ls, right, rw, rh, rs,
                                      kx, ky, sx, sy, out, os, nullptr);
}
