// filters.hip — the image filters on the stereo hot path: separable convolution (+ decimation) for the Gaussian
// pyramid and the Gaussian prefilters, the 3x3 Laplacian, mask decimation, image difference.
//
// Reference semantics kept exactly (float images, float kernels, the reference is built without FMA contraction and
// so is this file):
//   SeparableConvolutionView::rasterize / convolve_1d   src/vw/Image/Convolution.h:275-328
//   correlate_1d_at_point                               src/vw/Image/Convolution.h:53-65
//       result = 0; for i in 0..n-1: result += kernel[n-1-i] * src[i]      (float multiply, then float add)
//   horizontal pass into a float work image, then the vertical pass; edge extension applied to the SOURCE only
//   SubsampleView picks (s*i, s*j), size 1+(N-1)/s       src/vw/Image/Manipulation.h:214-293
//   ConvolutionView with the kernel rotated by 180 deg   src/vw/Image/Convolution.h:105-170, :66-88
//   subsample_mask_by_two                               src/vw/Stereo/CorrelationView.cc:38-63
// Because the accumulation order is fixed (not a running sum), results are bit-identical to the reference for ANY
// float input.
//
// Roofline: all of these are HBM bound.  Pyramid level: reads 4 B per source pixel once (tile + small halo through
// LDS), writes 1 B per source pixel (4 B per output) -> 5 B/source pixel; prefilter: 4 B in + 4 B out per pixel.
#include <algorithm>

#include "vwgpu_internal.h"

namespace {

constexpr int MAXT = 160;          // taps per axis (sigma up to ~22)
constexpr int TW = 64, TH = 16;    // outputs per workgroup

struct Taps {
  int nx, ny, cx, cy;
  float x[MAXT];
  float y[MAXT];
};

template <int EDGE>
__device__ __forceinline__ float ext_load(const float* __restrict__ src, ptrdiff_t stride, int w, int h, int x, int y) {
  if (EDGE == 1) {   // ZeroEdgeExtension
    if (x < 0 || y < 0 || x >= w || y >= h) return 0.0f;
    return src[(ptrdiff_t)y * stride + x];
  }
  x = x < 0 ? 0 : (x >= w ? w - 1 : x);   // ConstantEdgeExtension
  y = y < 0 ? 0 : (y >= h ? h - 1 : y);
  return src[(ptrdiff_t)y * stride + x];
}

struct ImgJobs { vwgpu_img_job j[VWGPU_MAX_IMG_JOBS]; };

template <int EDGE>
__global__ void __launch_bounds__(256)
sepconv_kernel(ImgJobs jobs, Taps t, int step, int sw, int sh) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const vwgpu_img_job J = jobs.j[blockIdx.z];
  const float* __restrict__ src = static_cast<const float*>(J.src);
  float* __restrict__ dst = static_cast<float*>(J.dst);
  const ptrdiff_t stride = J.stride, dstride = J.dstride;
  const int w = J.w, h = J.h, ow = J.ow, oh = J.oh, offx = J.offx, offy = J.offy;
  if ((int)blockIdx.x * TW >= ow || (int)blockIdx.y * TH >= oh) return;       // (the grid is the largest job's)
  if (J.bs == VWGPU_JOB_MASK_BY_TWO) {
    // a mask job riding in the launch of a pyramid level's images (subsample_mask_by_two, CorrelationView.cc:38-63; mask_by_two_kernel):
    // a launch of its own was 6 us of pure latency per level
    const uint8_t* __restrict__ msrc = static_cast<const uint8_t*>(J.src);
    uint8_t* __restrict__ mdst = static_cast<uint8_t*>(J.dst);
    for (int i = threadIdx.x; i < TH * TW; i += 256) {
      const int oy = blockIdx.y * TH + i / TW, ox = blockIdx.x * TW + (i % TW);
      if (ox >= ow || oy >= oh) continue;
      const int x = 2 * ox, y = 2 * oy;
      auto at = [&](int xx, int yy) -> int { return (xx < w && yy < h) ? (msrc[(ptrdiff_t)yy * stride + xx] != 0) : 0; };
      const int count = at(x, y) + at(x + 1, y) + at(x, y + 1) + at(x + 1, y + 1);
      mdst[(ptrdiff_t)oy * dstride + ox] = count > 1 ? 255 : 0;
    }
    return;
  }
  float* tile = smem;                       // [sh][sw]   edge-extended source
  float* work = smem + (size_t)sh * sw;     // [sh][TW]   horizontal pass (float, like the reference's `work`)
  const int tid = threadIdx.x;
  const int ox0 = blockIdx.x * TW, oy0 = blockIdx.y * TH;
  const int x_lo = t.nx ? t.nx - t.cx - 1 : 0, y_lo = t.ny ? t.ny - t.cy - 1 : 0;
  // output (ox, oy) sits at source position (ox*step + offx, oy*step + offy); the offsets let a caller rasterise the
  // view over a region that leaves the image (filter of the edge-extended source, as the reference's lazy views do)
  const int X0 = ox0 * step + offx - x_lo, Y0 = oy0 * step + offy - y_lo;

  // The source tile, LB rows of a column strip per batch: all loads of a batch are requested before the first one is stored (the flat
  // `for i: tile[i] = load(i)` form compiled to two loads per s_waitcnt vmcnt(0) — nine dependent memory round trips per workgroup, and a
  // 40-instruction division by the tile width per element: a pyramid level of a tile was 19 us of latency for 3 us of work).
  {
    constexpr int LB = 12;
    const int tx = tid & 63, ty = tid >> 6;
    for (int yy0 = ty; yy0 < sh; yy0 += 4 * LB)
      for (int xx = tx; xx < sw; xx += 64) {
        float v[LB];
#pragma unroll
        for (int b = 0; b < LB; ++b) {
          const int yy = yy0 + 4 * b;
          v[b] = yy < sh ? ext_load<EDGE>(src, stride, w, h, X0 + xx, Y0 + yy) : 0.0f;
        }
#pragma unroll
        for (int b = 0; b < LB; ++b) {
          const int yy = yy0 + 4 * b;
          if (yy < sh) tile[yy * sw + xx] = v[b];
        }
      }
  }
  __syncthreads();
  for (int i = tid; i < sh * TW; i += 256) {
    const int yy = i / TW, ox = i - yy * TW;
    const float* s = tile + (size_t)yy * sw + ox * step;
    float result;
    if (t.nx) {
      result = 0.0f;
      for (int k = 0; k < t.nx; ++k) result += t.x[t.nx - 1 - k] * s[k];
    } else {
      result = s[0];
    }
    work[i] = result;
  }
  __syncthreads();
  for (int i = tid; i < TH * TW; i += 256) {
    const int oyl = i / TW, ox = i - oyl * TW;
    if (ox0 + ox >= ow || oy0 + oyl >= oh) continue;
    const float* s = work + (size_t)(oyl * step) * TW + ox;
    float result;
    if (t.ny) {
      result = 0.0f;
      for (int k = 0; k < t.ny; ++k) result += t.y[t.ny - 1 - k] * s[(size_t)k * TW];
    } else {
      result = s[0];
    }
    dst[(ptrdiff_t)(oy0 + oyl) * dstride + ox0 + ox] = result;
  }
}

struct Kernel2D {
  int kw, kh, ci, cj;
  float k[49];
};

template <int EDGE>
__global__ void conv2d_kernel(ImgJobs jobs, Kernel2D kk) {
  const vwgpu_img_job J = jobs.j[blockIdx.z];
  const float* __restrict__ src = static_cast<const float*>(J.src);
  float* __restrict__ dst = static_cast<float*>(J.dst);
  const ptrdiff_t stride = J.stride, dstride = J.dstride;
  const int w = J.w, h = J.h, ow = J.ow, oh = J.oh, offx = J.offx, offy = J.offy;
  const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y * blockDim.y + threadIdx.y;
  if (ox >= ow || oy >= oh) return;
  const int x = ox + offx, y = oy + offy;          // source position of this output (may be outside the image)
  const int ci = kk.kw - 1 - kk.ci, cj = kk.kh - 1 - kk.cj;
  float result = 0.0f;
  if (kk.kw == 3 && kk.kh == 3) {                   // the Laplacian of the LoG prefilter: nine loads in flight, the same sum in the same order
    float v[9];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int i = 0; i < 3; ++i) v[j * 3 + i] = ext_load<EDGE>(src, stride, w, h, x - ci + i, y - cj + j);
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int i = 0; i < 3; ++i) result += kk.k[(2 - j) * 3 + (2 - i)] * v[j * 3 + i];
  } else {
    for (int j = 0; j < kk.kh; ++j)
      for (int i = 0; i < kk.kw; ++i)
        result += kk.k[(kk.kh - 1 - j) * kk.kw + (kk.kw - 1 - i)] * ext_load<EDGE>(src, stride, w, h, x - ci + i, y - cj + j);
  }
  dst[(ptrdiff_t)oy * dstride + ox] = result;
}

// dst(ox,oy) = a(clamp(ox+offx), clamp(oy+offy)) - (b ? b(ox,oy) : 0): NullOperation / SubtractedMean over a region
__global__ void ext_sub_kernel(const float* __restrict__ a, ptrdiff_t as, int w, int h, const float* __restrict__ b, ptrdiff_t bs,
                               float* __restrict__ dst, ptrdiff_t ds, int ow, int oh, int offx, int offy) {
  const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y * blockDim.y + threadIdx.y;
  if (ox >= ow || oy >= oh) return;
  const float v = ext_load<0>(a, as, w, h, ox + offx, oy + offy);
  dst[(ptrdiff_t)oy * ds + ox] = b ? v - b[(ptrdiff_t)oy * bs + ox] : v;
}

__global__ void mask_by_two_kernel(ImgJobs jobs) {
  const vwgpu_img_job J = jobs.j[blockIdx.z];
  const uint8_t* __restrict__ src = static_cast<const uint8_t*>(J.src);
  uint8_t* __restrict__ dst = static_cast<uint8_t*>(J.dst);
  const ptrdiff_t stride = J.stride, dstride = J.dstride;
  const int w = J.w, h = J.h, ow = J.ow, oh = J.oh;
  const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y * blockDim.y + threadIdx.y;
  if (ox >= ow || oy >= oh) return;
  const int x = 2 * ox, y = 2 * oy;
  auto at = [&](int xx, int yy) -> int { return (xx < w && yy < h) ? (src[(ptrdiff_t)yy * stride + xx] != 0) : 0; };
  const int count = at(x, y) + at(x + 1, y) + at(x, y + 1) + at(x + 1, y + 1);
  dst[(ptrdiff_t)oy * dstride + ox] = count > 1 ? 255 : 0;
}

__global__ void subtract_kernel(ImgJobs jobs) {
  const vwgpu_img_job J = jobs.j[blockIdx.z];
  const float* __restrict__ a = static_cast<const float*>(J.src);
  const float* __restrict__ b = J.b;
  float* __restrict__ dst = static_cast<float*>(J.dst);
  const ptrdiff_t as = J.stride, bs = J.bs, ds = J.dstride;
  const int w = J.w, h = J.h;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  dst[(ptrdiff_t)y * ds + x] = a[(ptrdiff_t)y * as + x] - b[(ptrdiff_t)y * bs + x];
}

}  // namespace

int vwgpu_launch_sepconv_region(vwgpu_ctx* ctx, const float* src, int w, int h, ptrdiff_t stride,
                                const float* xk, int nx, int cx, const float* yk, int ny, int cy,
                                int edge, int step, float* dst, ptrdiff_t dstride, int ow, int oh, int offx, int offy);

int vwgpu_launch_sepconv(vwgpu_ctx* ctx, const float* src, int w, int h, ptrdiff_t stride,
                         const float* xk, int nx, int cx, const float* yk, int ny, int cy,
                         int edge, int step, float* dst, ptrdiff_t dstride) {
  return vwgpu_launch_sepconv_region(ctx, src, w, h, stride, xk, nx, cx, yk, ny, cy, edge, step, dst, dstride,
                                     1 + (w - 1) / step, 1 + (h - 1) / step, 0, 0);
}

namespace {
int pack_jobs(vwgpu_ctx* ctx, const vwgpu_img_job* jobs, int n, ImgJobs& out, int& mw, int& mh) {
  if (n < 1 || n > VWGPU_MAX_IMG_JOBS) return vwgpu_fail(ctx, VWGPU_ERR_LOGIC, "filter launch: %d image jobs", n);
  mw = mh = 0;
  for (int i = 0; i < n; ++i) { out.j[i] = jobs[i]; mw = std::max(mw, jobs[i].ow); mh = std::max(mh, jobs[i].oh); }
  for (int i = n; i < VWGPU_MAX_IMG_JOBS; ++i) out.j[i] = jobs[0];
  return VWGPU_OK;
}
}  // namespace

int vwgpu_launch_sepconv_jobs(vwgpu_ctx* ctx, const vwgpu_img_job* jobs, int n, const float* xk, int nx, int cx, const float* yk, int ny, int cy,
                              int edge, int step) {
  if (nx > MAXT || ny > MAXT) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "separable convolution: more than %d taps", MAXT);
  Taps t;
  t.nx = nx; t.ny = ny; t.cx = cx; t.cy = cy;
  for (int i = 0; i < nx; ++i) t.x[i] = xk[i];
  for (int i = 0; i < ny; ++i) t.y[i] = yk[i];
  const int sw = (TW - 1) * step + 1 + (nx ? nx - 1 : 0), sh = (TH - 1) * step + 1 + (ny ? ny - 1 : 0);
  const size_t shmem = ((size_t)sh * sw + (size_t)sh * TW) * sizeof(float);
  if (shmem > 64 * 1024) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "separable convolution: kernel %dx%d step %d needs %zu B of LDS", nx, ny, step, shmem);
  ImgJobs J; int mw, mh;
  int rc = pack_jobs(ctx, jobs, n, J, mw, mh);
  if (rc) return rc;
  dim3 grd((mw + TW - 1) / TW, (mh + TH - 1) / TH, n), blk(256);
  vwgpu_prof_scope ps(ctx, step > 1 ? "sepconv_decimate" : "sepconv");
  if (edge == 1) hipLaunchKernelGGL(sepconv_kernel<1>, grd, blk, shmem, ctx->stream, J, t, step, sw, sh);
  else hipLaunchKernelGGL(sepconv_kernel<0>, grd, blk, shmem, ctx->stream, J, t, step, sw, sh);
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}

int vwgpu_launch_sepconv_region(vwgpu_ctx* ctx, const float* src, int w, int h, ptrdiff_t stride,
                                const float* xk, int nx, int cx, const float* yk, int ny, int cy,
                                int edge, int step, float* dst, ptrdiff_t dstride, int ow, int oh, int offx, int offy) {
  const vwgpu_img_job j{src, stride, w, h, dst, dstride, ow, oh, offx, offy, nullptr, 0};
  return vwgpu_launch_sepconv_jobs(ctx, &j, 1, xk, nx, cx, yk, ny, cy, edge, step);
}

int vwgpu_launch_conv2d(vwgpu_ctx* ctx, const float* src, int w, int h, ptrdiff_t stride,
                        const float* k, int kw, int kh, int ci, int cj, int edge, float* dst, ptrdiff_t dstride) {
  return vwgpu_launch_conv2d_region(ctx, src, w, h, stride, k, kw, kh, ci, cj, edge, dst, dstride, w, h, 0, 0);
}

int vwgpu_launch_conv2d_jobs(vwgpu_ctx* ctx, const vwgpu_img_job* jobs, int n, const float* k, int kw, int kh, int ci, int cj, int edge) {
  if (kw * kh > 49) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "2-D convolution: kernel %dx%d larger than 49 taps", kw, kh);
  Kernel2D kk;
  kk.kw = kw; kk.kh = kh; kk.ci = ci; kk.cj = cj;
  for (int i = 0; i < kw * kh; ++i) kk.k[i] = k[i];
  ImgJobs J; int mw, mh;
  int rc = pack_jobs(ctx, jobs, n, J, mw, mh);
  if (rc) return rc;
  dim3 blk(64, 4), grd((mw + 63) / 64, (mh + 3) / 4, n);
  vwgpu_prof_scope ps(ctx, "conv2d");
  if (edge == 1) hipLaunchKernelGGL(conv2d_kernel<1>, grd, blk, 0, ctx->stream, J, kk);
  else hipLaunchKernelGGL(conv2d_kernel<0>, grd, blk, 0, ctx->stream, J, kk);
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}

int vwgpu_launch_conv2d_region(vwgpu_ctx* ctx, const float* src, int w, int h, ptrdiff_t stride,
                               const float* k, int kw, int kh, int ci, int cj, int edge, float* dst, ptrdiff_t dstride,
                               int ow, int oh, int offx, int offy) {
  const vwgpu_img_job j{src, stride, w, h, dst, dstride, ow, oh, offx, offy, nullptr, 0};
  return vwgpu_launch_conv2d_jobs(ctx, &j, 1, k, kw, kh, ci, cj, edge);
}

int vwgpu_launch_mask_by_two_jobs(vwgpu_ctx* ctx, const vwgpu_img_job* jobs, int n) {
  ImgJobs J; int mw, mh;
  int rc = pack_jobs(ctx, jobs, n, J, mw, mh);
  if (rc) return rc;
  dim3 blk(64, 4), grd((mw + 63) / 64, (mh + 3) / 4, n);
  vwgpu_prof_scope ps(ctx, "mask_by_two");
  hipLaunchKernelGGL(mask_by_two_kernel, grd, blk, 0, ctx->stream, J);
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}

int vwgpu_launch_mask_by_two(vwgpu_ctx* ctx, const uint8_t* src, int w, int h, ptrdiff_t stride,
                             uint8_t* dst, ptrdiff_t dstride) {
  const vwgpu_img_job j{src, stride, w, h, dst, dstride, 1 + (w - 1) / 2, 1 + (h - 1) / 2, 0, 0, nullptr, 0};
  return vwgpu_launch_mask_by_two_jobs(ctx, &j, 1);
}

int vwgpu_launch_subtract_jobs(vwgpu_ctx* ctx, const vwgpu_img_job* jobs, int n) {
  ImgJobs J; int mw, mh;
  int rc = pack_jobs(ctx, jobs, n, J, mw, mh);
  if (rc) return rc;
  dim3 blk(64, 4), grd((mw + 63) / 64, (mh + 3) / 4, n);
  vwgpu_prof_scope ps(ctx, "subtract");
  hipLaunchKernelGGL(subtract_kernel, grd, blk, 0, ctx->stream, J);
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}

int vwgpu_launch_subtract(vwgpu_ctx* ctx, const float* a, ptrdiff_t as, const float* b, ptrdiff_t bs, int w, int h,
                          float* dst, ptrdiff_t ds) {
  const vwgpu_img_job j{a, as, w, h, dst, ds, w, h, 0, 0, b, bs};
  return vwgpu_launch_subtract_jobs(ctx, &j, 1);
}

int vwgpu_launch_ext_sub(vwgpu_ctx* ctx, const float* a, ptrdiff_t as, int w, int h, const float* b, ptrdiff_t bs,
                         float* dst, ptrdiff_t ds, int ow, int oh, int offx, int offy) {
  dim3 blk(64, 4), grd((ow + 63) / 64, (oh + 3) / 4);
  vwgpu_prof_scope ps(ctx, "edge_extend_sub");
  hipLaunchKernelGGL(ext_sub_kernel, grd, blk, 0, ctx->stream, a, as, w, h, b, bs, dst, ds, ow, oh, offx, offy);
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}

// prefilter.filter(image) rasterised over the region [x0,x0+bw) x [y0,y0+bh), which may leave the image
// (ParabolaSubpixelView::prerasterize crops the lazy prefilter views like this, ParabolaSubpixelView.cc:302-327).
// scratch must hold max(w*h, bw*bh) floats.
int vwgpu_prefilter_region(vwgpu_ctx* ctx, const float* src, int w, int h, ptrdiff_t stride, int mode, float width,
                           int x0, int y0, int bw, int bh, float* dst, float* scratch) {
  if (mode != VWGPU_PREFILTER_LOG && mode != VWGPU_PREFILTER_MEANSUB)
    return vwgpu_launch_ext_sub(ctx, src, stride, w, h, nullptr, 0, dst, bw, bw, bh, x0, y0);
  float taps[1024];
  const int nt = vwgpu_generate_gaussian_kernel((double)width, 0, taps, 1024);
  if (nt < 0) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "prefilter width %g too large", (double)width);
  const int c = nt ? (nt - 1) / 2 : 0;
  if (mode == VWGPU_PREFILTER_MEANSUB) {
    int rc = vwgpu_launch_sepconv_region(ctx, src, w, h, stride, taps, nt, c, taps, nt, c, 0, 1, scratch, bw, bw, bh, x0, y0);
    if (rc) return rc;
    return vwgpu_launch_ext_sub(ctx, src, stride, w, h, scratch, bw, dst, bw, bw, bh, x0, y0);
  }
  int rc = vwgpu_launch_sepconv(ctx, src, w, h, stride, taps, nt, c, taps, nt, c, 0, 1, scratch, w);   // gaussian on the image domain
  if (rc) return rc;
  const float lap[9] = {0, 1, 0, 1, -4, 1, 0, 1, 0};
  return vwgpu_launch_conv2d_region(ctx, scratch, w, h, w, lap, 3, 3, 1, 1, 0, dst, bw, bw, bh, x0, y0);
}
