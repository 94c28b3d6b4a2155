// bm_sad_u16.hip — SAD block matching + winner-take-all for integer-valued imagery in [0,65535] (16-bit sensors; the
// reference's own TestCorrelation fixture runs at scale 32767, src/vw/Stereo/tests/TestCorrelation.cxx:45-214).
//
// Same domain argument as the packed-u8 path (bm_sad_u8.hip): on integer-valued pixels |a-b| in float, the float64 box
// sums of fast_box_sum (src/vw/Stereo/Algorithms.h:43-129) and the strict `<` of best_of_search_convolution
// (src/vw/Stereo/Correlation.cc:91-117) are exact, so the result is the smallest (cost, dy, dx) — tracked here as one
// 32-bit key (cost << 8 | d; cost <= 225 * 65535 < 2^24) — and a pixel is invalid iff minimum == maximum (:121-133).
//
//   mapping   lane <-> one output column x TY rows, workgroup = 4 waves = 256 columns (the layout of bm_corr_u8.hip)
//   abs-diffs v_sad_u16 on pixel PAIRS: the lane's LEFT pairs (kx pixels per row, all TY+ky-1 rows) live in registers, the
//             RIGHT pairs come from an LDS array holding the pair that starts at EVERY pixel; the accumulator of a chain
//             down the rows is the vertical prefix sum, the ky-row window sum is P[r] - P[r-ky]
//   quads     disparities {d0, d0+2, d0+4, d0+6} share right pairs (pair n of d is pair n-1 of d+2): kx/2 + 3 full pairs and
//             four half pairs per row instead of four windows
//   odd kx    the last pair of a window holds one live pixel; its RIGHT half is cleared (v_and), so the dead LEFT pixel adds
//             the same value to every disparity of the pixel — argmin and the equality test do not see it (the trick of the
//             u8 kernel's zero-padded last word)
// Inputs outside the domain raise the device flag; the caller then runs the float64 kernel.  One search row (sy == 1).
//
// Roofline: HBM bound by the task's definition (20 B per output pixel); VALU-issue bound in fact: (TY+ky-1)/TY * (ceil(kx/2)
// + 1) + 4 instruction slots per (pixel, disparity).
#include <algorithm>

#include "vwgpu_internal.h"

namespace {

typedef uint32_t u32;

constexpr int STW = 256;          // output columns per workgroup
constexpr int STHREADS = 256;

__device__ __forceinline__ u32 umin3(u32 a, u32 b, u32 c) { u32 r; asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ u32 umax3(u32 a, u32 b, u32 c) { u32 r; asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ u32 sad_u16(u32 a, u32 b, u32 c) { u32 r; asm("v_sad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// float -> u16 with the exactness test of the path: integer-valued and inside [0,65535]
__device__ __forceinline__ u32 to_u16(float v, bool& bad) {
  const float r = rintf(v);
  bad |= !(r == v && v >= 0.0f && v <= 65535.0f);
  return (u32)(int)fminf(fmaxf(r, 0.0f), 65535.0f);
}

// dst[r][g] = pixels (x0 + 2g, x0 + 2g + 1) of row y0 + r as a u16 pair, zero outside the image
template <int NROWS>
__device__ __forceinline__ void stage_u16_rows(const float* __restrict__ img, ptrdiff_t stride, int w, int h, int x0, int y0,
                                               int npairs, u32* __restrict__ dst, int tid, bool& bad) {
  for (int i = tid; i < NROWS * npairs; i += STHREADS) {
    const int r = i / npairs, gq = i - r * npairs;
    const int x = x0 + 2 * gq, y = y0 + r;
    u32 p = 0;
    if (y < h) {
      const float* row = img + (ptrdiff_t)y * stride;
      if (x < w) p = to_u16(row[x], bad);
      if (x + 1 < w) p |= to_u16(row[x + 1], bad) << 16;
    }
    dst[i] = p;
  }
}

template <int KX, int KY, int TY>
__global__ void __launch_bounds__(STHREADS, 2)
bm_sad_u16_kernel(const float* __restrict__ L, ptrdiff_t ls, int lw, int lh,
                  const float* __restrict__ R, ptrdiff_t rs, int rcw, int rch, int sx, int urp, int rpd,
                  int32_t* __restrict__ out, ptrdiff_t os, int ow, int oh,
                  int* __restrict__ flag_set, int* __restrict__ flag_clear) {
  constexpr int NWF = KX / 2;                                    // full pairs of a window row; one more half pair (kx is odd)
  constexpr int NR = TY + KY - 1;
  constexpr int LPD = STW / 2 + NWF + 2;                         // aligned pairs per staged left row
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  u32* UR = lds;                                                 // [NR][urp]  the pair starting at every pixel of the right rows
  u32* XR = UR + (size_t)NR * urp;                               // [NR][rpd]  aligned right pairs (staging)
  u32* LW = XR + (size_t)NR * rpd;                               // [NR][LPD]  aligned left pairs
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * STW, y0 = blockIdx.y * TY;
  const int x = x0 + tid;
  if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) *flag_clear = 0;     // the NEXT call's flag

  bool bad = false;
  stage_u16_rows<NR>(L, ls, lw, lh, x0, y0, LPD, LW, tid, bad);
  stage_u16_rows<NR>(R, rs, rcw, rch, x0, y0, rpd, XR, tid, bad);
  __syncthreads();
  // LEFT pairs of this lane's column: pixels x + 2n, x + 2n + 1
  u32 lwn[NR][NWF + 1];
  {
    const int w0 = tid >> 1;
    const u32 sh = (u32)(tid & 1) * 16u;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      u32 a[NWF + 2];
#pragma unroll
      for (int n = 0; n <= NWF + 1; ++n) a[n] = LW[r * LPD + w0 + n];
#pragma unroll
      for (int n = 0; n <= NWF; ++n) lwn[r][n] = __builtin_amdgcn_alignbit(a[n + 1], a[n], sh);
    }
  }
  // the pair starting at every pixel of the right rows
  for (int i = tid; i < NR * urp; i += STHREADS) {
    const int r = i / urp, b = i - r * urp, w = b >> 1;
    UR[i] = __builtin_amdgcn_alignbit(XR[r * rpd + w + 1], XR[r * rpd + w], (u32)(b & 1) * 16u);
  }
  __syncthreads();

  u32 K[TY], Wk[TY];
#pragma unroll
  for (int y = 0; y < TY; ++y) { K[y] = 0xffffffffu; Wk[y] = 0u; }
  const u32* ur0 = UR + tid;
  constexpr int Q = 4, NWQ = Q + NWF - 1, PF = 3;
  // One quad of disparities d0 + {0, 2, 4, 6}; LDS reads issued PF rows ahead and pinned there (see bm_corr_u8.hip).
  auto quad = [&](int d0, int nv, auto masked) __attribute__((always_inline)) {
    constexpr bool MASKED = decltype(masked)::value;
    u32 Wd[NR][NWQ], Hd[NR][Q], P[Q][NR], acc[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = 0;
    auto fetch = [&](int r) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < NWQ; ++j) Wd[r][j] = ur0[r * urp + d0 + 2 * j];
#pragma unroll
      for (int q = 0; q < Q; ++q) Hd[r][q] = ur0[r * urp + d0 + 2 * q + 2 * NWF];
    };
#pragma unroll
    for (int r = 0; r < PF && r < NR; ++r) fetch(r);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if (r + PF < NR) fetch(r + PF);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < NWF; ++n)
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[q] = sad_u16(lwn[r][n], Wd[r][q + n], acc[q]);
#pragma unroll
      for (int q = 0; q < Q; ++q) acc[q] = sad_u16(lwn[r][NWF], Hd[r][q] & 0xffffu, acc[q]);   // last pixel; the dead one adds L only
#pragma unroll
      for (int q = 0; q < Q; ++q) P[q][r] = acc[q];
      if (r >= KY - 1) {
        const int y = r - (KY - 1);
        u32 k[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const u32 s = r >= KY ? P[q][r] - P[q][r - KY] : P[q][r];
          k[q] = (s << 8) | (u32)(d0 + 2 * q);
        }
        if (!MASKED) {
          K[y] = umin3(umin3(K[y], k[0], k[1]), k[2], k[3]);
          Wk[y] = umax3(umax3(Wk[y], k[0], k[1]), k[2], k[3]);
        } else {
#pragma unroll
          for (int q = 0; q < Q; ++q)
            if (q < nv) { K[y] = K[y] < k[q] ? K[y] : k[q]; Wk[y] = Wk[y] > k[q] ? Wk[y] : k[q]; }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  for (int t = 0; t < 2; ++t) {                                  // phase t = d mod 2, then steps of 8
    const int nt = (sx - t + 1) >> 1;
    int a0 = 0;
    for (; a0 + Q <= nt; a0 += Q) quad(2 * a0 + t, Q, std::false_type{});
    if (a0 < nt) quad(2 * a0 + t, nt - a0, std::true_type{});
  }
  if (x < ow) {
#pragma unroll
    for (int y = 0; y < TY; ++y) {
      if (y0 + y < oh) {
        int32_t* o = out + ((ptrdiff_t)(y0 + y) * os + x) * 3;
        o[0] = (int32_t)(K[y] & 0xffu); o[1] = 0;
        o[2] = ((K[y] >> 8) == (Wk[y] >> 8)) ? 0 : 0x7fffffff;     // best == worst (Correlation.cc:121-133)
      }
    }
  }
  if (__syncthreads_or(bad) && tid == 0) atomicOr(flag_set, 1);
}

typedef void (*Sad16Fn)(const float*, ptrdiff_t, int, int, const float*, ptrdiff_t, int, int, int, int, int, int32_t*, ptrdiff_t, int, int, int*, int*);
struct Sad16Launch { int kx, ky, ty; Sad16Fn fn; };
#define VW_S16(KX, KY, TY) Sad16Launch{KX, KY, TY, bm_sad_u16_kernel<KX, KY, TY>}
const Sad16Launch kSad16[] = {VW_S16(3, 3, 16), VW_S16(5, 5, 16), VW_S16(7, 7, 16), VW_S16(7, 5, 16), VW_S16(9, 9, 16), VW_S16(11, 11, 16)};
#undef VW_S16

const Sad16Launch* find_sad16(int kx, int ky) {
  for (const Sad16Launch& l : kSad16)
    if (l.kx == kx && l.ky == ky) return &l;
  return nullptr;
}
void sad16_geom(int kx, int sx, int* urp, int* rpd) {
  *urp = STW + sx - 1 + kx + 1 + 16;                             // every pixel a window of a (clamped tail) quad can start a pair at
  *rpd = (*urp + 1) / 2 + 2;
}
size_t sad16_lds(const Sad16Launch& l, int sx) {
  int urp, rpd;
  sad16_geom(l.kx, sx, &urp, &rpd);
  const int nr = l.ty + l.ky - 1;
  return ((size_t)nr * urp + (size_t)nr * rpd + (size_t)nr * (STW / 2 + l.kx / 2 + 2)) * sizeof(u32);
}

}  // namespace

bool vwgpu_bm_sad_u16_supported(int cost_type, int kx, int ky, int sx, int sy) {
  if (cost_type != VWGPU_ABSOLUTE_DIFFERENCE || sy != 1 || sx > 256) return false;
  const Sad16Launch* l = find_sad16(kx, ky);
  return l && sad16_lds(*l, sx) <= 80 * 1024;
}

int vwgpu_launch_bm_sad_u16(vwgpu_ctx* ctx, const float* left, int lw, int lh, ptrdiff_t ls,
                            const float* right, int rw, int rh, ptrdiff_t rs, int kx, int ky, int sx, int sy,
                            int32_t* out, ptrdiff_t os, int** d_fallback_flag) {
  (void)rw; (void)rh; (void)sy;
  const Sad16Launch* l = find_sad16(kx, ky);
  if (!l) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "no packed-u16 SAD kernel for %dx%d", kx, ky);
  const int ow = lw - kx + 1, oh = lh - ky + 1;
  int urp, rpd;
  sad16_geom(kx, sx, &urp, &rpd);
  int* flag_set = nullptr; int* flag_clear = nullptr;
  int rc = vwgpu_next_flags(ctx, 0, &flag_set, &flag_clear, nullptr);
  if (rc) return rc;
  *d_fallback_flag = flag_set;
  const size_t shmem = sad16_lds(*l, sx);
  if (shmem > 64 * 1024)
    VWGPU_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(l->fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  vwgpu_prof_scope ps(ctx, "bm_sad_u16");
  hipLaunchKernelGGL(l->fn, dim3((ow + STW - 1) / STW, (oh + l->ty - 1) / l->ty), dim3(STHREADS), shmem, ctx->stream,
                     left, ls, lw, lh, right, rs, lw + sx - 1, lh, sx, urp, rpd, out, os, ow, oh, flag_set, flag_clear);
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}
