// bm_sad_u8.hip — packed-u8 SAD block matching + winner-take-all for integer-valued imagery.
//
// The headline kernel (BASELINE.json config 2: 4096^2, 7x7 SAD, 129x1 search).  It replaces the whole
// per-disparity pipeline of best_of_search_convolution (src/vw/Stereo/Correlation.cc:64-133: crop copy ->
// AbsoluteCost image -> fast_box_sum -> compare loop -> validity pass) for inputs on which the reference's
// float/double arithmetic is exact: pixel values that are integers in [0,255].  On that
// domain |a-b| in float, the float64 box sums and the strict `<` compares are all exact, so the result only
// depends on (cost, dy, dx) lexicographic order ("strict compare, first wins" == smallest (cost, dy, dx)),
// which is what this kernel tracks.  Anything else raises a device flag and the generic float64 kernel
// (bm_generic.hip) recomputes the image.
//
// Why not the float formulation: 4096^2 x 129 disparities = 2.16 G (pixel, disparity) evaluations against
// 337 MB of compulsory HBM traffic (42 us) — the kernel is VALU-issue bound, not HBM bound.
// v_qsad_pk_u16_u8 does 16 abs-diffs + 4 accumulates per lane per issue, so the design is built around it:
//
//   mapping   lane <-> 4 consecutive output pixels (q..q+3) x TY output rows; wave <-> 256 x TY pixels;
//             workgroup = 4 (or 2) waves side by side.
//   qsad      src0 (64 bit) = 8 LEFT bytes L[q+4n .. q+4n+7] held in registers for all TY+ky-1 rows,
//             src1 (32 bit) = 4 RIGHT bytes R[q+j+4n .. +3] read from LDS, j = "step" = 4a+t.
//             Result slot i (u16) = sum_b |L[q+i+4n+b] - R[q+j+4n+b]|  -> pixel q+i at disparity d = j-i.
//             Putting RIGHT in the 32-bit operand makes a partial last word (kx % 4 != 0) harmless: its
//             missing bytes are zeroed in LDS, which adds sum L[...] to every disparity of that pixel —
//             a per-pixel constant that changes neither the argmin nor the best==worst validity test.
//   vertical  one accumulator chain per step runs down the rows (the adds are free inside qsad); the
//             ky-row window cost is P[r] - P[r-ky] (v_pk_sub_u16, modulo 2^16, exact because the true
//             window sum is < 2^16).
//   LDS       RIGHT words at byte phase t = j mod 4 are unaligned in memory, so the step loop runs t-outer:
//             for each t the workgroup rebuilds one LDS array of pre-shifted, pre-masked word groups
//             (v_alignbyte once per word), then walks a = 0..A with one 8/16-byte LDS read per row.
//   WTA       key = cost << 16 | (dy*sx + dx); K = v_min3_u32(K, key_a, key_a+1).  Order independent, so
//             the t-outer loop order is fine.  Worst cost is tracked with v_pk_max_u16 for the validity test.
//
// Limits (else VWGPU_ERR_NOIMPL -> generic path): SAD only; kx <= 15; 4*ceil(kx/4)*ky*255 < 65536;
// sx*sy <= 65535; LDS footprint <= 64 KiB.
#include <type_traits>

#include "vwgpu_internal.h"

namespace {

typedef uint32_t u32;
typedef uint64_t u64;
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32 pk_sub_u16(u32 a, u32 b) {
  return __builtin_bit_cast(u32, (u16x2)(__builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ u32 pk_max_u16(u32 a, u32 b) {
  return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ u32 umin3(u32 a, u32 b, u32 c) { return min(min(a, b), c); }

// ---- float -> u8 planes --------------------------------------------------------------------------------
// Writes the whole padded plane (pad = 0) and raises *flag if any pixel is not an integer in [0,255].
__global__ void pack_u8_kernel(const float* __restrict__ src, ptrdiff_t stride, int w, int h,
                               u32* __restrict__ dst, int pitch_dw, int rows, int* __restrict__ flag) {
  const int xd = blockIdx.x * blockDim.x + threadIdx.x;   // dword column
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (xd >= pitch_dw || y >= rows) return;
  u32 packed = 0;
  bool bad = false;
  if (y < h) {
    const float* p = src + (ptrdiff_t)y * stride;
    const int x = xd * 4;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if (x + b < w) {
        const float v = p[x + b];
        const float r = rintf(v);
        bad |= !(r == v && v >= 0.0f && v <= 255.0f);
        packed |= ((u32)(int)fminf(fmaxf(r, 0.0f), 255.0f)) << (8 * b);
      }
    }
  }
  dst[(size_t)y * pitch_dw + xd] = packed;
  if (bad) atomicOr(flag, 1);
}

// ---- the matcher -----------------------------------------------------------------------------------------
template <int KX, int KY, int TY>
struct Cfg {
  static constexpr int NW = (KX + 3) / 4;          // qsads per row
  static constexpr int EW = NW <= 2 ? 2 : 4;       // dwords per LDS entry
  static constexpr int WAVES = NW <= 2 ? 4 : 2;    // waves side by side
  static constexpr int THREADS = WAVES * 64;
  static constexpr int TWB = WAVES * 256;          // output columns per workgroup
  static constexpr int NR = TY + KY - 1;           // input rows per tile
  static constexpr u32 LAST_MASK = (KX % 4 == 0) ? 0xffffffffu : ((1u << (8 * (KX % 4))) - 1u);
};

template <int KX, int KY, int TY>
__global__ void __launch_bounds__((KX <= 8 ? 256 : 128), 2)
bm_sad_u8_kernel(const uint8_t* __restrict__ L8, int pitch_l, const uint8_t* __restrict__ R8, int pitch_r,
                 int sx, int sy, int ne, int32_t* __restrict__ out, ptrdiff_t os, int ow, int oh) {
  typedef Cfg<KX, KY, TY> C;
  constexpr int NW = C::NW, EW = C::EW, NR = C::NR;
  extern __shared__ __attribute__((aligned(16))) u32 lds[];   // [NR][ne][EW]

  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * C::TWB;
  const int y0 = blockIdx.y * TY;
  const int lqb = tid;                    // entry index of this lane's pixel group inside the tile
  const int q = x0 + 4 * tid;             // first of the lane's 4 output pixels

  // LEFT windows for all rows: win[r][n] = bytes L[q+4n .. q+4n+7]
  u64 win[NR][NW];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const u32* lp = reinterpret_cast<const u32*>(L8 + (size_t)(y0 + r) * pitch_l + q);
    u32 a[NW + 1];
#pragma unroll
    for (int n = 0; n <= NW; ++n) a[n] = lp[n];
#pragma unroll
    for (int n = 0; n < NW; ++n) win[r][n] = (u64)a[n] | ((u64)a[n + 1] << 32);
  }

  u32 K[TY][4];       // best key per pixel: cost << 16 | disparity index
  u32 MX[TY][2];      // worst cost, packed u16 x 2
#pragma unroll
  for (int y = 0; y < TY; ++y) {
    K[y][0] = K[y][1] = K[y][2] = K[y][3] = 0xffffffffu;
    MX[y][0] = MX[y][1] = 0u;
  }
  const u32 HI = 0xffff0000u;

  for (int dy = 0; dy < sy; ++dy) {
    for (int t = 0; t < 4; ++t) {
      const int a_last = (sx + 2 - t) >> 2;            // last step with any valid slot (floor; sx+2-t >= 0)
      // ---- rebuild the LDS array of RIGHT word groups at byte phase t ----
      __syncthreads();                                   // previous pass done reading
      for (int r = 0; r < NR; ++r) {
        const uint8_t* rrow = R8 + (size_t)(y0 + r + dy) * pitch_r + x0;
        for (int m = tid; m < ne; m += C::THREADS) {
          const u32* bp = reinterpret_cast<const u32*>(rrow + 4 * m);
          u32 b[NW + 1];
#pragma unroll
          for (int n = 0; n <= NW; ++n) b[n] = bp[n];
          u32 w[EW];
#pragma unroll
          for (int n = 0; n < EW; ++n) w[n] = 0;
#pragma unroll
          for (int n = 0; n < NW; ++n) w[n] = __builtin_amdgcn_alignbyte(b[n + 1], b[n], t);
          w[NW - 1] &= C::LAST_MASK;
          u32* e = lds + ((size_t)r * ne + m) * EW;
          if (EW == 2) *reinterpret_cast<uint2*>(e) = make_uint2(w[0], w[1]);
          else *reinterpret_cast<uint4*>(e) = make_uint4(w[0], w[1], w[2 % EW], w[3 % EW]);
        }
      }
      __syncthreads();

      // One step = one accumulator chain down the NR rows.  MASKED handles range ends (some slots outside
      // [0,sx)); PAIR processes steps a and a+1 together so the WTA is one v_min3_u32 per pixel.
      auto step = [&](int a, auto masked_tag, auto pair_tag) __attribute__((always_inline)) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        constexpr bool PAIR = decltype(pair_tag)::value;
        const int jbase = 4 * a + t;                     // d for slot i is jbase - i
        const int ibase = dy * sx + jbase;
        u32 idxA[4], idxB[4];
        u32 orA[2] = {0u, 0u};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          idxA[i] = (u32)(ibase - i) & 0xffffu;
          idxB[i] = (u32)(ibase + 4 - i) & 0xffffu;
          if (MASKED) {
            const int d = jbase - i;
            if (d < 0 || d >= sx) orA[i >> 1] |= (i & 1) ? 0xffff0000u : 0x0000ffffu;
          }
        }
        const u32* ep = lds + (size_t)(lqb + a) * EW;
        u64 accA = 0, accB = 0;
        u64 PA[NR], PB[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          const u32* e = ep + (size_t)r * ne * EW;
          u32 wa[EW], wb[EW];
          if (EW == 2) {
            const uint2 v = *reinterpret_cast<const uint2*>(e);
            wa[0] = v.x; wa[1] = v.y;
            if (PAIR) { const uint2 u = *reinterpret_cast<const uint2*>(e + EW); wb[0] = u.x; wb[1] = u.y; }
          } else {
            const uint4 v = *reinterpret_cast<const uint4*>(e);
            wa[0] = v.x; wa[1] = v.y; wa[2 % EW] = v.z; wa[3 % EW] = v.w;
            if (PAIR) { const uint4 u = *reinterpret_cast<const uint4*>(e + EW); wb[0] = u.x; wb[1] = u.y; wb[2 % EW] = u.z; wb[3 % EW] = u.w; }
          }
#pragma unroll
          for (int n = 0; n < NW; ++n) {
            accA = __builtin_amdgcn_qsad_pk_u16_u8(win[r][n], wa[n], accA);
            if (PAIR) accB = __builtin_amdgcn_qsad_pk_u16_u8(win[r][n], wb[n], accB);
          }
          PA[r] = accA;
          if (PAIR) PB[r] = accB;
          if (r >= KY - 1) {
            const int y = r - (KY - 1);
            u32 sA0 = (u32)PA[r], sA1 = (u32)(PA[r] >> 32);
            if (r >= KY) { sA0 = pk_sub_u16(sA0, (u32)PA[r - KY]); sA1 = pk_sub_u16(sA1, (u32)(PA[r - KY] >> 32)); }
            if (MASKED) {
              MX[y][0] = pk_max_u16(MX[y][0], sA0 & ~orA[0]);
              MX[y][1] = pk_max_u16(MX[y][1], sA1 & ~orA[1]);
              sA0 |= orA[0]; sA1 |= orA[1];
            } else {
              MX[y][0] = pk_max_u16(MX[y][0], sA0);
              MX[y][1] = pk_max_u16(MX[y][1], sA1);
            }
            const u32 kA0 = (sA0 << 16) | idxA[0], kA1 = (sA0 & HI) | idxA[1];
            const u32 kA2 = (sA1 << 16) | idxA[2], kA3 = (sA1 & HI) | idxA[3];
            if (PAIR) {
              u32 sB0 = (u32)PB[r], sB1 = (u32)(PB[r] >> 32);
              if (r >= KY) { sB0 = pk_sub_u16(sB0, (u32)PB[r - KY]); sB1 = pk_sub_u16(sB1, (u32)(PB[r - KY] >> 32)); }
              MX[y][0] = pk_max_u16(MX[y][0], sB0);
              MX[y][1] = pk_max_u16(MX[y][1], sB1);
              const u32 kB0 = (sB0 << 16) | idxB[0], kB1 = (sB0 & HI) | idxB[1];
              const u32 kB2 = (sB1 << 16) | idxB[2], kB3 = (sB1 & HI) | idxB[3];
              K[y][0] = umin3(K[y][0], kA0, kB0);
              K[y][1] = umin3(K[y][1], kA1, kB1);
              K[y][2] = umin3(K[y][2], kA2, kB2);
              K[y][3] = umin3(K[y][3], kA3, kB3);
            } else {
              K[y][0] = min(K[y][0], kA0);
              K[y][1] = min(K[y][1], kA1);
              K[y][2] = min(K[y][2], kA2);
              K[y][3] = min(K[y][3], kA3);
            }
          }
        }
      };
      typedef std::true_type T;
      typedef std::false_type F;

      // clean range: every slot valid  <=>  jbase-3 >= 0 and jbase <= sx-1
      int a_lo = (t >= 3) ? 0 : 1;
      int a_hi = (sx - 1 - t >= 0) ? ((sx - 1 - t) >> 2) + 1 : 0;   // exclusive
      if (a_hi > a_last + 1) a_hi = a_last + 1;
      if (a_lo > a_hi) a_lo = a_hi;
      int a = 0;
      for (; a < a_lo && a <= a_last; ++a) step(a, T{}, F{});
      for (; a + 1 < a_hi; a += 2) step(a, F{}, T{});
      for (; a <= a_last; ++a) step(a, T{}, F{});
    }
  }

  // ---- epilogue: decode keys, validity, store in the PixelMask<Vector2i> layout --------------------------
#pragma unroll
  for (int y = 0; y < TY; ++y) {
    const int oy = y0 + y;
    if (oy >= oh) continue;
    int32_t* orow = out + ((ptrdiff_t)oy * os + q) * 3;
    u32 mx[4] = {MX[y][0] & 0xffffu, MX[y][0] >> 16, MX[y][1] & 0xffffu, MX[y][1] >> 16};
    int32_t v[12];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const u32 k = K[y][s];
      const u32 di = k & 0xffffu;
      int dx, dy;
      if (sy == 1) { dx = (int)di; dy = 0; } else { dy = (int)(di / (u32)sx); dx = (int)(di - (u32)dy * (u32)sx); }
      v[3 * s] = dx;
      v[3 * s + 1] = dy;
      v[3 * s + 2] = ((k >> 16) == mx[s]) ? 0 : 0x7fffffff;
    }
    if (q + 3 < ow) {
#pragma unroll
      for (int i = 0; i < 12; ++i) orow[i] = v[i];
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (q + s < ow) { orow[3 * s] = v[3 * s]; orow[3 * s + 1] = v[3 * s + 1]; orow[3 * s + 2] = v[3 * s + 2]; }
    }
  }
}

struct Launch {
  int kx, ky, ty;
  int threads, twb, nr, ew;
  void (*fn)(const uint8_t*, int, const uint8_t*, int, int, int, int, int32_t*, ptrdiff_t, int, int);
};

template <int KX, int KY, int TY>
constexpr Launch make_launch() {
  typedef Cfg<KX, KY, TY> C;
  return Launch{KX, KY, TY, C::THREADS, C::TWB, C::NR, C::EW, bm_sad_u8_kernel<KX, KY, TY>};
}

// Instantiated kernel sizes.  Others fall back to the generic path.
const Launch kLaunch[] = {
    make_launch<3, 3, 16>(), make_launch<5, 5, 16>(), make_launch<7, 7, 16>(), make_launch<7, 5, 16>(),
    make_launch<9, 9, 12>(), make_launch<11, 11, 8>(),
};

const Launch* find_launch(int kx, int ky) {
  for (const Launch& l : kLaunch)
    if (l.kx == kx && l.ky == ky) return &l;
  return nullptr;
}

constexpr size_t kMaxLds = 64 * 1024;

size_t lds_bytes(const Launch& l, int sx) {
  const int ne = l.twb / 4 + ((sx + 2) >> 2) + 1;
  return (size_t)l.nr * ne * l.ew * sizeof(u32);
}

}  // namespace

bool vwgpu_bm_sad_u8_supported(int cost_type, int kx, int ky, int sx, int sy) {
  if (cost_type != VWGPU_ABSOLUTE_DIFFERENCE) return false;
  const Launch* l = find_launch(kx, ky);
  if (!l) return false;
  if ((long long)sx * sy > 65535) return false;
  return lds_bytes(*l, sx) <= kMaxLds;
}

int vwgpu_launch_bm_sad_u8(vwgpu_ctx* ctx,
                           const float* left, int lw, int lh, ptrdiff_t ls,
                           const float* right, int rw, int rh, ptrdiff_t rs,
                           int kx, int ky, int sx, int sy, int32_t* out, ptrdiff_t os,
                           int** d_fallback_flag) {
  (void)rw; (void)rh;
  const Launch* l = find_launch(kx, ky);
  if (!l) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "no packed-u8 kernel for %dx%d", kx, ky);
  const int ow = lw - kx + 1, oh = lh - ky + 1;
  const int rcw = lw + sx - 1, rch = lh + sy - 1;

  const int gx = (ow + l->twb - 1) / l->twb, gy = (oh + l->ty - 1) / l->ty;
  const int ne = l->twb / 4 + ((sx + 2) >> 2) + 1;
  // Padded planes so that every tile read stays inside the allocation (pad bytes are zero).
  const int pitch_l = (int)vwgpu_align_up((size_t)gx * l->twb + 64, 64);
  const int pitch_r = (int)vwgpu_align_up((size_t)(gx - 1) * l->twb + 4 * (size_t)(ne + 4) + 64, 64);
  const int rows_l = gy * l->ty + ky;
  const int rows_r = gy * l->ty + ky + sy;
  const size_t lbytes = vwgpu_align_up((size_t)pitch_l * rows_l, 256);
  const size_t rbytes = vwgpu_align_up((size_t)pitch_r * rows_r, 256);
  int rc = vwgpu_arena_reserve(ctx, &ctx->scratch, 256 + lbytes + rbytes);
  if (rc) return rc;
  char* base = static_cast<char*>(ctx->scratch.base);
  int* flag = reinterpret_cast<int*>(base);
  uint8_t* l8 = reinterpret_cast<uint8_t*>(base + 256);
  uint8_t* r8 = l8 + lbytes;
  *d_fallback_flag = flag;

  VWGPU_HIP(ctx, hipMemsetAsync(flag, 0, 256, ctx->stream));
  {
    vwgpu_prof_scope ps(ctx, "pack_u8_left");
    dim3 blk(64, 4), grd((pitch_l / 4 + 63) / 64, (rows_l + 3) / 4);
    hipLaunchKernelGGL(pack_u8_kernel, grd, blk, 0, ctx->stream, left, ls, lw, lh,
                       reinterpret_cast<u32*>(l8), pitch_l / 4, rows_l, flag);
  }
  {
    vwgpu_prof_scope ps(ctx, "pack_u8_right");
    dim3 blk(64, 4), grd((pitch_r / 4 + 63) / 64, (rows_r + 3) / 4);
    hipLaunchKernelGGL(pack_u8_kernel, grd, blk, 0, ctx->stream, right, rs, rcw, rch,
                       reinterpret_cast<u32*>(r8), pitch_r / 4, rows_r, flag);
  }
  {
    vwgpu_prof_scope ps(ctx, "bm_sad_u8");
    const size_t shmem = lds_bytes(*l, sx);
    hipLaunchKernelGGL(l->fn, dim3(gx, gy), dim3(l->threads), shmem, ctx->stream,
                       l8, pitch_l, r8, pitch_r, sx, sy, ne, out, os, ow, oh);
  }
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}
