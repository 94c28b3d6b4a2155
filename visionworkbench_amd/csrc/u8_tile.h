// u8_tile.h — staging of float image tiles as packed u8 dwords in LDS, shared by the packed-u8 matchers
// (bm_sad_u8.hip, bm_corr_u8.hip).  Every pixel is checked on the way: the packed kernels are exact only for
// integer-valued pixels in [0,255]; anything else ORs a non-zero value into `acc`.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace vwgpu_u8 {

typedef uint32_t u32;

// float -> u8 with the exactness test of the fast path: integer-valued and inside [0,255].
__device__ __forceinline__ u32 to_u8(float v, bool& bad) {
  const float r = rintf(v);
  bad |= !(r == v && v >= 0.0f && v <= 255.0f);
  return (u32)(int)fminf(fmaxf(r, 0.0f), 255.0f);
}

// Loads a row-segment of a float image as packed u8 dwords into LDS: dst[g] = bytes (x0+4g .. x0+4g+3),
// zero outside [0,w) x [0,h).  `vec4` (workgroup-uniform) = every in-range group of 4 floats is 16-byte aligned.
__device__ __forceinline__ u32 pack4(float v0, float v1, float v2, float v3, bool& bad) {
  return to_u8(v0, bad) | (to_u8(v1, bad) << 8) | (to_u8(v2, bad) << 16) | (to_u8(v3, bad) << 24);
}

// float4 -> 4 packed u8 (v_cvt_pk_u8_f32 saturates) and the exactness test of the fast path: a pixel is
// representable iff converting the byte back gives the same float; the differences are OR-ed into `acc`
// (non-zero bits => some pixel was not an integer in [0,255]; NaN/Inf/-0.0 also land there).
__device__ __forceinline__ u32 pack4_check(float4 v, u32& acc) {
  u32 p = 0;
  p = __builtin_amdgcn_cvt_pk_u8_f32(v.x, 0, p);
  p = __builtin_amdgcn_cvt_pk_u8_f32(v.y, 1, p);
  p = __builtin_amdgcn_cvt_pk_u8_f32(v.z, 2, p);
  p = __builtin_amdgcn_cvt_pk_u8_f32(v.w, 3, p);
  acc |= __builtin_bit_cast(u32, v.x - (float)(p & 0xffu));
  acc |= __builtin_bit_cast(u32, v.y - (float)((p >> 8) & 0xffu));
  acc |= __builtin_bit_cast(u32, v.z - (float)((p >> 16) & 0xffu));
  acc |= __builtin_bit_cast(u32, v.w - (float)(p >> 24));
  return p;
}

// Groups straddling the right image edge (x < w <= x+3, at most one per row): the vector paths wrote 0 there.
template <int NROWS>
__device__ __forceinline__ void patch_right_edge(const float* __restrict__ img, ptrdiff_t stride, int w, int h,
                                                 int x0, int y0, int ndw, int dst_pitch_dw,
                                                 u32* __restrict__ dst, int tid, int nthreads, u32& acc) {
  const int ge = (w - x0) >> 2;
  if (ge >= 0 && ge < ndw && ((w - x0) & 3) != 0) {
    for (int r = tid; r < NROWS; r += nthreads) {
      const int y = y0 + r;
      u32 p = 0;
      if (y < h) {
        const float* row = img + (ptrdiff_t)y * stride;
        const int x = x0 + 4 * ge;
        bool bad = false;
#pragma unroll
        for (int b = 0; b < 4; ++b)
          if (x + b < w) p |= to_u8(row[x + b], bad) << (8 * b);
        if (bad) acc |= 1u;
      }
      dst[r * dst_pitch_dw + ge] = p;
    }
  }
}

// Loads NROWS rows of a float image as packed u8 dwords into LDS: dst[r][g] = bytes (x0+4g .. x0+4g+3), zero
// outside [0,w) x [0,h).  Row base pointers are workgroup-uniform (scalar), the per-lane part of the address is one
// offset shared by all rows, and all NROWS loads of a thread are unconditional (out-of-range groups read offset 0 of
// a valid row and are zeroed afterwards), so they are in flight together: staging a tile costs about one memory
// latency.  Columns beyond the first `nthreads` groups (the search margin) are spread evenly over the workgroup.
// Groups straddling the right image edge (at most one per row) are patched by a scalar tail loop.
template <int NROWS>
struct U8MainLoads { float4 v[NROWS]; };

// main part of a tile: group g = tid of every row.  `issue` requests the NROWS loads, `finish` converts and stores them.
template <int NROWS>
__device__ __forceinline__ void stage_u8_main_issue(const float* __restrict__ img, ptrdiff_t stride, int w, int h, int x0, int y0, int ndw,
                                                    int tid, U8MainLoads<NROWS>& ld) {
  const bool vec4 = ((reinterpret_cast<uintptr_t>(img) & 15) == 0) && ((stride & 3) == 0);
  if (tid < ndw) {
    const int x = x0 + 4 * tid;
    const bool colin = x + 3 < w;
    const int off = colin ? x : 0;
#pragma unroll
    for (int r = 0; r < NROWS; ++r) {
      const float* rowp = (y0 + r < h) ? img + (ptrdiff_t)(y0 + r) * stride : img;   // uniform
      if (vec4) ld.v[r] = *reinterpret_cast<const float4*>(rowp + off);
      else ld.v[r] = make_float4(rowp[off], rowp[off + (colin ? 1 : 0)], rowp[off + (colin ? 2 : 0)], rowp[off + (colin ? 3 : 0)]);
    }
  }
}
template <int NROWS>
__device__ __forceinline__ void stage_u8_main_finish(int w, int h, int x0, int y0, int ndw, int dst_pitch_dw, u32* __restrict__ dst, int tid,
                                                     const U8MainLoads<NROWS>& ld, u32& acc) {
  if (tid < ndw) {
    const int g = tid, x = x0 + 4 * g;
    const bool colin = x + 3 < w;
#pragma unroll
    for (int r = 0; r < NROWS; ++r) {
      u32 a = 0;
      const u32 p = pack4_check(ld.v[r], a);
      const bool inb = colin && (y0 + r < h);
      if (inb) acc |= a;
      if (colin || x >= w) dst[r * dst_pitch_dw + g] = inb ? p : 0u;   // a straddling group belongs to patch_right_edge
    }
  }
}

// remainder columns [nthreads, ndw): NROWS * rem items spread over all threads, 4 in flight per thread; then the straddling groups
template <int NROWS>
__device__ __forceinline__ void stage_u8_rest(const float* __restrict__ img, ptrdiff_t stride, int w, int h,
                                              int x0, int y0, int ndw, int dst_pitch_dw,
                                              u32* __restrict__ dst, int tid, int nthreads, u32& acc) {
  const bool vec4 = ((reinterpret_cast<uintptr_t>(img) & 15) == 0) && ((stride & 3) == 0);
  auto load4 = [&](const float* rowp, int off, bool full) __attribute__((always_inline)) -> float4 {
    if (vec4) return *reinterpret_cast<const float4*>(rowp + off);
    return make_float4(rowp[off], rowp[off + (full ? 1 : 0)], rowp[off + (full ? 2 : 0)], rowp[off + (full ? 3 : 0)]);
  };
  const int rem = ndw - nthreads;
  if (rem > 0) {
    const int total = NROWS * rem;
    const float inv = 1.0f / (float)rem;
    for (int i0 = 0; i0 < total; i0 += 4 * nthreads) {
      float4 v[4]; int di[4]; bool inb[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int idx = i0 + k * nthreads + tid;
        int r = (int)(((float)(idx < total ? idx : 0) + 0.5f) * inv);
        int c = (idx < total ? idx : 0) - r * rem;
        if (c < 0) { --r; c += rem; }
        if (c >= rem) { ++r; c -= rem; }
        const int g = nthreads + c, x = x0 + 4 * g;
        inb[k] = (idx < total) && (x + 3 < w) && (y0 + r < h);
        di[k] = (idx < total && (x + 3 < w || x >= w)) ? r * dst_pitch_dw + g : -1;   // not the straddling group
        const float* src = inb[k] ? img + (ptrdiff_t)(y0 + r) * stride + x : img;
        v[k] = load4(src, 0, inb[k]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        u32 a = 0;
        const u32 p = pack4_check(v[k], a);
        if (inb[k]) acc |= a;
        if (di[k] >= 0) dst[di[k]] = inb[k] ? p : 0u;
      }
    }
  }
  patch_right_edge<NROWS>(img, stride, w, h, x0, y0, ndw, dst_pitch_dw, dst, tid, nthreads, acc);
}

template <int NROWS>
__device__ __forceinline__ void stage_u8_rows(const float* __restrict__ img, ptrdiff_t stride, int w, int h,
                                              int x0, int y0, int ndw, int dst_pitch_dw,
                                              u32* __restrict__ dst, int tid, int nthreads, u32& acc) {
  U8MainLoads<NROWS> ld;
  stage_u8_main_issue<NROWS>(img, stride, w, h, x0, y0, ndw, tid, ld);
  stage_u8_main_finish<NROWS>(w, h, x0, y0, ndw, dst_pitch_dw, dst, tid, ld, acc);
  stage_u8_rest<NROWS>(img, stride, w, h, x0, y0, ndw, dst_pitch_dw, dst, tid, nthreads, acc);
}

}  // namespace vwgpu_u8
