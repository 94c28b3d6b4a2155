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

template <int EDGE>
__global__ void __launch_bounds__(256)
sepconv_kernel(const float* __restrict__ src, ptrdiff_t stride, int w, int h, Taps t, int step,
               float* __restrict__ dst, ptrdiff_t dstride, int ow, int oh, int sw, int sh) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tile = smem;                       // [sh][sw]   edge-extended source
  float* work = smem + (size_t)sh * sw;     // [sh][TW]   horizontal pass (float, like the reference's `work`)
  const int tid = threadIdx.x;
  const int ox0 = blockIdx.x * TW, oy0 = blockIdx.y * TH;
  const int x_lo = t.nx ? t.nx - t.cx - 1 : 0, y_lo = t.ny ? t.ny - t.cy - 1 : 0;
  const int X0 = ox0 * step - x_lo, Y0 = oy0 * step - y_lo;

  for (int i = tid; i < sh * sw; i += 256) {
    const int yy = i / sw, xx = i - yy * sw;
    tile[i] = ext_load<EDGE>(src, stride, w, h, X0 + xx, Y0 + yy);
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
__global__ void conv2d_kernel(const float* __restrict__ src, ptrdiff_t stride, int w, int h, Kernel2D kk,
                              float* __restrict__ dst, ptrdiff_t dstride) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  const int ci = kk.kw - 1 - kk.ci, cj = kk.kh - 1 - kk.cj;
  float result = 0.0f;
  for (int j = 0; j < kk.kh; ++j)
    for (int i = 0; i < kk.kw; ++i)
      result += kk.k[(kk.kh - 1 - j) * kk.kw + (kk.kw - 1 - i)] * ext_load<EDGE>(src, stride, w, h, x - ci + i, y - cj + j);
  dst[(ptrdiff_t)y * dstride + x] = result;
}

__global__ void mask_by_two_kernel(const uint8_t* __restrict__ src, ptrdiff_t stride, int w, int h,
                                   uint8_t* __restrict__ dst, ptrdiff_t dstride, int ow, int oh) {
  const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y * blockDim.y + threadIdx.y;
  if (ox >= ow || oy >= oh) return;
  const int x = 2 * ox, y = 2 * oy;
  auto at = [&](int xx, int yy) -> int { return (xx < w && yy < h) ? (src[(ptrdiff_t)yy * stride + xx] != 0) : 0; };
  const int count = at(x, y) + at(x + 1, y) + at(x, y + 1) + at(x + 1, y + 1);
  dst[(ptrdiff_t)oy * dstride + ox] = count > 1 ? 255 : 0;
}

__global__ void subtract_kernel(const float* __restrict__ a, ptrdiff_t as, const float* __restrict__ b, ptrdiff_t bs,
                                int w, int h, float* __restrict__ dst, ptrdiff_t ds) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= w || y >= h) return;
  dst[(ptrdiff_t)y * ds + x] = a[(ptrdiff_t)y * as + x] - b[(ptrdiff_t)y * bs + x];
}

}  // namespace

int vwgpu_launch_sepconv(vwgpu_ctx* ctx, const float* src, int w, int h, ptrdiff_t stride,
                         const float* xk, int nx, int cx, const float* yk, int ny, int cy,
                         int edge, int step, float* dst, ptrdiff_t dstride) {
  if (nx > MAXT || ny > MAXT) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "separable convolution: more than %d taps", MAXT);
  Taps t;
  t.nx = nx; t.ny = ny; t.cx = cx; t.cy = cy;
  for (int i = 0; i < nx; ++i) t.x[i] = xk[i];
  for (int i = 0; i < ny; ++i) t.y[i] = yk[i];
  const int ow = 1 + (w - 1) / step, oh = 1 + (h - 1) / step;
  const int sw = (TW - 1) * step + 1 + (nx ? nx - 1 : 0), sh = (TH - 1) * step + 1 + (ny ? ny - 1 : 0);
  const size_t shmem = ((size_t)sh * sw + (size_t)sh * TW) * sizeof(float);
  if (shmem > 64 * 1024) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "separable convolution: kernel %dx%d step %d needs %zu B of LDS", nx, ny, step, shmem);
  dim3 grd((ow + TW - 1) / TW, (oh + TH - 1) / TH), blk(256);
  vwgpu_prof_scope ps(ctx, step > 1 ? "sepconv_decimate" : "sepconv");
  if (edge == 1)
    hipLaunchKernelGGL(sepconv_kernel<1>, grd, blk, shmem, ctx->stream, src, stride, w, h, t, step, dst, dstride, ow, oh, sw, sh);
  else
    hipLaunchKernelGGL(sepconv_kernel<0>, grd, blk, shmem, ctx->stream, src, stride, w, h, t, step, dst, dstride, ow, oh, sw, sh);
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}

int vwgpu_launch_conv2d(vwgpu_ctx* ctx, const float* src, int w, int h, ptrdiff_t stride,
                        const float* k, int kw, int kh, int ci, int cj, int edge, float* dst, ptrdiff_t dstride) {
  if (kw * kh > 49) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "2-D convolution: kernel %dx%d larger than 49 taps", kw, kh);
  Kernel2D kk;
  kk.kw = kw; kk.kh = kh; kk.ci = ci; kk.cj = cj;
  for (int i = 0; i < kw * kh; ++i) kk.k[i] = k[i];
  dim3 blk(64, 4), grd((w + 63) / 64, (h + 3) / 4);
  vwgpu_prof_scope ps(ctx, "conv2d");
  if (edge == 1) hipLaunchKernelGGL(conv2d_kernel<1>, grd, blk, 0, ctx->stream, src, stride, w, h, kk, dst, dstride);
  else hipLaunchKernelGGL(conv2d_kernel<0>, grd, blk, 0, ctx->stream, src, stride, w, h, kk, dst, dstride);
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}

int vwgpu_launch_mask_by_two(vwgpu_ctx* ctx, const uint8_t* src, int w, int h, ptrdiff_t stride,
                             uint8_t* dst, ptrdiff_t dstride) {
  const int ow = 1 + (w - 1) / 2, oh = 1 + (h - 1) / 2;
  dim3 blk(64, 4), grd((ow + 63) / 64, (oh + 3) / 4);
  vwgpu_prof_scope ps(ctx, "mask_by_two");
  hipLaunchKernelGGL(mask_by_two_kernel, grd, blk, 0, ctx->stream, src, stride, w, h, dst, dstride, ow, oh);
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}

int vwgpu_launch_subtract(vwgpu_ctx* ctx, const float* a, ptrdiff_t as, const float* b, ptrdiff_t bs, int w, int h,
                          float* dst, ptrdiff_t ds) {
  dim3 blk(64, 4), grd((w + 63) / 64, (h + 3) / 4);
  vwgpu_prof_scope ps(ctx, "subtract");
  hipLaunchKernelGGL(subtract_kernel, grd, blk, 0, ctx->stream, a, as, b, bs, w, h, dst, ds);
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}
