// lr_check.hip — left/right consistency check on PixelMask<Vector2i> disparity images.
// Replaces vw::stereo::cross_corr_consistency_check, src/vw/Stereo/Correlate.cc:1441-1502.
// One gather per pixel; HBM-bound (reads 12 B + gathers 12 B, writes 4 B per pixel).
#include "vwgpu_internal.h"

namespace {

// diff (optional): PixelMask<float> image {value, valid}; kept pixels store their discrepancy at (c + ulx, r + uly) (:1480-1484)
__global__ void lr_check_kernel(int32_t* __restrict__ l2r, int lw, int lh, ptrdiff_t ls,
                                const int32_t* __restrict__ r2l, int rw, int rh, ptrdiff_t rs, float thr,
                                float* __restrict__ diff2, ptrdiff_t dstride, int ulx, int uly) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y * blockDim.y + threadIdx.y;
  if (c >= lw || r >= lh) return;
  int32_t* p = l2r + ((ptrdiff_t)r * ls + c) * 3;
  const int dx = p[0], dy = p[1], v = p[2];
  const int x = c + dx, y = r + dy;                       // :1466-1467
  bool keep = false;
  if (x >= 0 && x < rw && y >= 0 && y < rh) {             // :1469-1471
    const int32_t* q = r2l + ((ptrdiff_t)y * rs + x) * 3;
    if (v != 0 && q[2] != 0) {                            // :1472-1474
      // :1476-1478 — fabs on int sums evaluated in double, max assigned to float
      const float diff = (float)fmax(fabs((double)(dx + q[0])), fabs((double)(dy + q[1])));
      keep = thr >= diff;                                 // :1479
      if (keep && diff2) {
        float* d = diff2 + ((ptrdiff_t)(r + uly) * dstride + (c + ulx)) * 2;
        d[0] = diff; d[1] = 1.0f;
      }
    }
  }
  if (!keep) p[2] = 0;
}

}  // namespace

int vwgpu_launch_lr_check(vwgpu_ctx* ctx, int32_t* l2r, int lw, int lh, ptrdiff_t ls,
                          const int32_t* r2l, int rw, int rh, ptrdiff_t rs, float thr) {
  return vwgpu_launch_lr_check_diff(ctx, l2r, lw, lh, ls, r2l, rw, rh, rs, thr, nullptr, 0, 0, 0);
}

int vwgpu_launch_lr_check_diff(vwgpu_ctx* ctx, int32_t* l2r, int lw, int lh, ptrdiff_t ls,
                               const int32_t* r2l, int rw, int rh, ptrdiff_t rs, float thr,
                               float* diff2, ptrdiff_t dstride, int ulx, int uly) {
  dim3 blk(64, 4), grd((lw + 63) / 64, (lh + 3) / 4);
  vwgpu_prof_scope ps(ctx, "lr_check");
  hipLaunchKernelGGL(lr_check_kernel, grd, blk, 0, ctx->stream, l2r, lw, lh, ls, r2l, rw, rh, rs, thr, diff2, dstride, ulx, uly);
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}
