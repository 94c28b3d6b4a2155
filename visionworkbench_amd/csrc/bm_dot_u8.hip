// bm_dot_u8.hip — SSD and NCC block matching for integer-valued inputs in [0,255] on the packed dot-product unit.
//
// Replaces best_of_search_convolution + fast_box_sum + SquaredCost / NCCCost (src/vw/Stereo/Correlation.cc:33-137,
// src/vw/Stereo/Algorithms.h:43-129, src/vw/Stereo/CostFunctions.h:94-141,179-236) for the same inputs the packed SAD path
// takes.  On such data every quantity the reference accumulates in float64 is an exactly representable integer:
//     S(x,y,d)  = sum_window L * R            (v_dot4_u32_u8 over byte-aligned word pairs, running sum down the rows)
//     A2(x,y)   = sum_window L^2,  B2(x,y) = sum_window R^2
//     SSD cost  = A2(x,y) + B2(x+d,y) - 2 S                                  -> compared as integers (no float at all)
//     NCC cost  = double(S) * sqrt( (1.0 / A2(x,y)) * (1.0 / B2(x+d,y)) )    -> the reference's float64 sequence
//                 (NCCCost ctor :214-219, cost_modification :227-231), maximised.
// Work split: workgroup = 64 output columns x 32 output rows; wave g of 4 owns a contiguous quarter of the disparities, lane
// = column.  Rows are walked serially: S_d += h(enter row) - h(leave row) with h = NW dot4 per row; the byte alignment of
// the two operands comes from four byte-phase copies of each staged row in LDS (word w of phase p = bytes [4w+p, 4w+p+3]).
// Per row the four waves exchange (best, index, worst) through LDS; validity = best != worst = "not all costs equal"
// (Correlation.cc:91-133 reduces to that when no cost is NaN).  Tiles containing an all-zero window (NCC: 1/0) or inputs
// that are not integers in [0,255] raise the device flag and the float64 kernel recomputes the image (same protocol as
// bm_sad_u8.hip).  sy must be 1 (one search row); other shapes stay on the generic path.
//
// Roofline: LDS-bandwidth / VALU bound, not HBM: ~2*NW LDS word reads + 2*NW dot4 per (pixel, disparity), plus ~35 float64
// instructions per evaluation for NCC.  Algorithmic HBM bytes are the same 20 B per output pixel as the SAD path.
#include "vwgpu_internal.h"

namespace {

constexpr int TX = 64;           // output columns per workgroup
constexpr int TYR = 32;          // output rows per workgroup
constexpr int NTHREADS = 256;

struct DotGeom {
  int kx, ky, sx;
  int nr;        // staged rows = TYR + ky - 1
  int lws, rws;  // words per staged row (left / right)
  int nb;        // right window origins per row = TX + sx - 1
};

template <int COST, int NW, int DCH>
__global__ void __launch_bounds__(NTHREADS)
bm_dot_u8_kernel(const float* __restrict__ left, ptrdiff_t ls, int lw, int lh,
                 const float* __restrict__ right, ptrdiff_t rs, int rcw, int rch,
                 DotGeom g, int32_t* __restrict__ out, ptrdiff_t os, int ow, int oh,
                 int* __restrict__ flag_set, int* __restrict__ flag_clear) {
  extern __shared__ uint32_t smem[];
  const int tid = threadIdx.x, xl = tid & 63;
  const int dg = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index: scalar, so the per-disparity guards below are scalar branches
  const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TYR;
  if (tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) *flag_clear = 0;

  const int nr = g.nr, lws = g.lws, rws = g.rws;
  // Byte-indexed word arrays: UL[r][b] = the 32-bit word made of bytes [b, b+3] of staged row r, for EVERY byte offset b.
  // Window word n of a window that starts at byte c is U[r][c + 4n]: lane = column makes consecutive lanes read
  // consecutive dwords (no bank conflicts), and the unrolled disparity index is an immediate offset (no address math).
  const int LL = 4 * lws, RL = 4 * rws;                   // dwords per row of UL / UR
  uint32_t* UL = smem;                                    // [nr][LL]
  uint32_t* UR = UL + nr * LL;                            // [nr][RL]
  uint32_t* Lph = UR + nr * RL;                           // [nr][lws]   the staged bytes (aligned words)
  uint32_t* Rph = Lph + nr * lws;                         // [nr][rws]
  uint32_t* A2row = Rph + nr * rws;                       // [TX]
  uint32_t* B2row = A2row + TX;                           // [256]
  double* precL = reinterpret_cast<double*>(smem + (((B2row + 256) - smem + 1) & ~(ptrdiff_t)1)); // [TX]  (NCC), 8-byte aligned
  double* precR = precL + TX;                             // [256]     (NCC)
  double* mbest = precR + 256;                            // [3][TX]
  double* mworst = mbest + 3 * TX;                        // [3][TX]
  int* midx = reinterpret_cast<int*>(mworst + 3 * TX);    // [3][TX]

  // ---- stage both tiles as bytes (phase 0), checking that every pixel is an integer in [0,255] ----
  bool bad = false;
  {
    uint8_t* Lb = reinterpret_cast<uint8_t*>(Lph);
    uint8_t* Rb = reinterpret_cast<uint8_t*>(Rph);
    const int lbytes = lws * 4, rbytes = rws * 4;
    for (int i = tid; i < nr * lbytes; i += NTHREADS) {
      const int r = i / lbytes, c = i - r * lbytes;
      const int gx = x0 + c, gy = y0 + r;
      uint32_t v = 0;
      if (gx < lw && gy < lh) {
        const float f = left[(ptrdiff_t)gy * ls + gx];
        const int iv = (int)f;
        if (!((float)iv == f && iv >= 0 && iv <= 255)) bad = true;
        v = (uint32_t)iv & 255u;
      }
      Lb[i] = (uint8_t)v;
    }
    for (int i = tid; i < nr * rbytes; i += NTHREADS) {
      const int r = i / rbytes, c = i - r * rbytes;
      const int gx = x0 + c, gy = y0 + r;
      uint32_t v = 0;
      if (gx < rcw && gy < rch) {
        const float f = right[(ptrdiff_t)gy * rs + gx];
        const int iv = (int)f;
        if (!((float)iv == f && iv >= 0 && iv <= 255)) bad = true;
        v = (uint32_t)iv & 255u;
      }
      Rb[i] = (uint8_t)v;
    }
  }
  __syncthreads();
  // byte-indexed words from the aligned ones (zero beyond the row)
  for (int i = tid; i < nr * LL; i += NTHREADS) {
    const int r = i / LL, b = i - r * LL, w = b >> 2;
    const uint32_t a = Lph[r * lws + w], hi = (w + 1 < lws) ? Lph[r * lws + w + 1] : 0u;
    UL[i] = __builtin_amdgcn_alignbyte(hi, a, b & 3);
  }
  for (int i = tid; i < nr * RL; i += NTHREADS) {
    const int r = i / RL, b = i - r * RL, w = b >> 2;
    const uint32_t a = Rph[r * rws + w], hi = (w + 1 < rws) ? Rph[r * rws + w + 1] : 0u;
    UR[i] = __builtin_amdgcn_alignbyte(hi, a, b & 3);
  }
  __syncthreads();

  const int kx = g.kx, ky = g.ky, sx = g.sx;
  const uint32_t kmask = (kx & 3) ? ((1u << (8 * (kx & 3))) - 1u) : 0xffffffffu;     // live bytes of the last window word
  // this wave's disparities
  const int dpt = (sx + 3) / 4;
  const int dbeg = dg * dpt, dcnt = max(0, min(dpt, sx - dbeg));
  const uint32_t* Lrow = UL + xl;                                         // + r * LL + 4 * n
  // running window sums
  uint32_t S[DCH];
#pragma unroll
  for (int i = 0; i < DCH; ++i) S[i] = 0;
  uint32_t a2 = 0, b2 = 0;                                                // thread xl < TX (wave 0): A2 column; thread tid < nb: B2 column
  const uint32_t* Bself = UR + tid;                                       // window origin x' = tid
  const bool ownB = tid < g.nb, ownA = dg == 0;
  const int dlast = dcnt - 1;                                             // last slot of this wave's share
  const bool full = dcnt == DCH;                                          // wave-uniform

  auto self_dot = [&](const uint32_t* row) -> uint32_t {
    uint32_t s = 0;
#pragma unroll
    for (int n = 0; n < NW; ++n) {
      const uint32_t v = row[4 * n];
      s = __builtin_amdgcn_udot4(n == NW - 1 ? (v & kmask) : v, v, s, false);
    }
    return s;
  };

  bool zero_window = false;
  for (int y = 0; y < TYR; ++y) {
    const int gy = y0 + y;
    if (gy >= oh) break;                                                   // uniform
    // ---- update the running sums to the window rows [y, y + ky) ----
    // The slot loops carry no per-slot branch (a guard per disparity serialises every LDS round trip): slots past this
    // wave's share re-read its last disparity — same addresses, a duplicate cost that can neither win (strict compare,
    // larger index) nor change the worst value.
    auto add_row = [&](int r, bool subtract) __attribute__((always_inline)) {
      uint32_t lw_[NW];
#pragma unroll
      for (int n = 0; n < NW; ++n) { lw_[n] = Lrow[r * LL + 4 * n]; if (n == NW - 1) lw_[n] &= kmask; }
      const uint32_t* rb = UR + r * RL + xl + dbeg;
      if (full) {                                                           // every slot live: the slot index is an immediate offset
#pragma unroll
        for (int i = 0; i < DCH; ++i) {
          uint32_t sdot = 0;
#pragma unroll
          for (int n = 0; n < NW; ++n) sdot = __builtin_amdgcn_udot4(lw_[n], rb[i + 4 * n], sdot, false);
          S[i] = subtract ? S[i] - sdot : S[i] + sdot;
        }
      } else {
#pragma unroll
        for (int i = 0; i < DCH; ++i) {
          const int xi = i < dlast ? i : dlast;                             // scalar
          uint32_t sdot = 0;
#pragma unroll
          for (int n = 0; n < NW; ++n) sdot = __builtin_amdgcn_udot4(lw_[n], rb[xi + 4 * n], sdot, false);
          S[i] = subtract ? S[i] - sdot : S[i] + sdot;
        }
      }
    };
    if (y == 0) {
      for (int r = 0; r < ky; ++r) {
        if (ownA) a2 += self_dot(Lrow + r * LL);
        if (ownB) b2 += self_dot(Bself + r * RL);
        if (dcnt > 0) add_row(r, false);
      }
    } else {
      const int re = y + ky - 1, rl = y - 1;
      if (ownA) a2 += self_dot(Lrow + re * LL) - self_dot(Lrow + rl * LL);
      if (ownB) b2 += self_dot(Bself + re * RL) - self_dot(Bself + rl * RL);
      if (dcnt > 0) { add_row(re, false); add_row(rl, true); }
    }
    if (ownA) {
      A2row[xl] = a2;
      if (COST == VWGPU_CROSS_CORRELATION) { precL[xl] = 1.0 / (double)a2; if (a2 == 0 && x0 + xl < ow) zero_window = true; }
    }
    if (ownB) {
      B2row[tid] = b2;
      if (COST == VWGPU_CROSS_CORRELATION) { precR[tid] = 1.0 / (double)b2; if (b2 == 0 && x0 + tid < rcw - kx + 1) zero_window = true; }
    }
    __syncthreads();
    // ---- costs of this wave's disparities, winner and loser ----
    double best = 0.0, worst = 0.0;
    int bidx = 0;
    if (COST == VWGPU_CROSS_CORRELATION) {
      const double pl = precL[xl];
      if (dcnt > 0) {
#pragma unroll
        for (int i = 0; i < DCH; ++i) {
          const int xi = i < dlast ? i : dlast;
          const double c = (double)S[i] * sqrt(pl * precR[xl + dbeg + xi]);
          if (i == 0) { best = worst = c; bidx = dbeg; }
          else { if (c > best) { best = c; bidx = dbeg + xi; } if (c < worst) worst = c; }
        }
      }
    } else {
      // cost = A2 + B2 - 2 S < 2^25 (kx, ky <= 16, 31): cost << 6 | slot is one v_min_u32 per slot, and the slot order inside
      // the key keeps "first wins" (DCH <= 64 slots)
      const uint32_t al = A2row[xl];
      uint32_t kb = 0xffffffffu, uw = 0;
      if (dcnt > 0) {
        const uint32_t* b2p = B2row + xl + dbeg;
#pragma unroll
        for (int i = 0; i < DCH; ++i) {
          const int xi = full ? i : (i < dlast ? i : dlast);
          const uint32_t c = al + b2p[xi] - 2u * S[i];
          const uint32_t key = (c << 6) | (uint32_t)xi;
          kb = key < kb ? key : kb;
          uw = c > uw ? c : uw;
        }
        bidx = dbeg + (int)(kb & 63u);
      }
      best = (double)(kb >> 6); worst = (double)uw;
    }
    if (dg > 0 && dcnt > 0) { mbest[(dg - 1) * TX + xl] = best; mworst[(dg - 1) * TX + xl] = worst; midx[(dg - 1) * TX + xl] = bidx; }
    __syncthreads();
    if (dg == 0) {
      for (int gq = 1; gq < 4; ++gq) {
        if (gq * dpt >= sx) break;                                          // that wave had no disparities
        const double b = mbest[(gq - 1) * TX + xl], w = mworst[(gq - 1) * TX + xl];
        if (COST == VWGPU_CROSS_CORRELATION) { if (b > best) { best = b; bidx = midx[(gq - 1) * TX + xl]; } if (w < worst) worst = w; }
        else { if (b < best) { best = b; bidx = midx[(gq - 1) * TX + xl]; } if (w > worst) worst = w; }
      }
      const int gx = x0 + xl;
      if (gx < ow) {
        int32_t* o = out + ((ptrdiff_t)gy * os + gx) * 3;
        o[0] = bidx; o[1] = 0; o[2] = (best == worst) ? 0 : 0x7fffffff;
      }
    }
    __syncthreads();
  }
  if (bad || zero_window) atomicOr(flag_set, 1);
}

size_t dot_lds_bytes(const DotGeom& g) {
  return (size_t)(5 * g.nr * (g.lws + g.rws) + TX + 256 + 2) * 4 + (size_t)(TX + 256 + 6 * TX) * 8 + 3 * TX * 4;
}

DotGeom make_geom(int kx, int ky, int sx) {
  DotGeom g;
  g.kx = kx; g.ky = ky; g.sx = sx;
  g.nr = TYR + ky - 1;
  const int nw = (kx + 3) / 4;
  g.lws = (TX + 3) / 4 + nw + 1;
  g.nb = TX + sx - 1;
  g.rws = (g.nb + 3) / 4 + nw + 1;
  return g;
}

typedef void (*DotFn)(const float*, ptrdiff_t, int, int, const float*, ptrdiff_t, int, int, DotGeom, int32_t*, ptrdiff_t, int, int, int*, int*);

template <int COST>
DotFn pick(int nw, int dpt) {
// slots per lane: 33 = ceil(129 / 4), the +-64 px search, so that three of its four waves run the branch-free full path
#define VW_DOT(N) (dpt <= 12 ? (DotFn)bm_dot_u8_kernel<COST, N, 12> : dpt <= 24 ? (DotFn)bm_dot_u8_kernel<COST, N, 24> : \
                   dpt <= 33 ? (DotFn)bm_dot_u8_kernel<COST, N, 33> : (DotFn)bm_dot_u8_kernel<COST, N, 40>)
  switch (nw) {
    case 1: return VW_DOT(1);
    case 2: return VW_DOT(2);
    case 3: return VW_DOT(3);
    default: return VW_DOT(4);
  }
#undef VW_DOT
}

}  // namespace

bool vwgpu_bm_dot_u8_supported(int cost_type, int kx, int ky, int sx, int sy) {
  if (cost_type != VWGPU_SQUARED_DIFFERENCE && cost_type != VWGPU_CROSS_CORRELATION) return false;
  if (sy != 1 || kx > 16 || ky > 31 || sx > 160 || TX + sx - 1 > 256) return false;
  return dot_lds_bytes(make_geom(kx, ky, sx)) <= 120 * 1024;
}

int vwgpu_launch_bm_dot_u8(vwgpu_ctx* ctx, int cost_type, const float* left, int lw, int lh, ptrdiff_t ls,
                           const float* right, int rw, int rh, ptrdiff_t rs, int kx, int ky, int sx, int sy,
                           int32_t* out, ptrdiff_t os, int** d_fallback_flag) {
  (void)rw; (void)rh; (void)sy;
  const int ow = lw - kx + 1, oh = lh - ky + 1;
  const int rcw = lw + sx - 1, rch = lh;
  const DotGeom g = make_geom(kx, ky, sx);
  int* flag_set = nullptr; int* flag_clear = nullptr;
  int rc = vwgpu_next_flags(ctx, 0, &flag_set, &flag_clear, nullptr);
  if (rc) return rc;
  *d_fallback_flag = flag_set;
  const size_t shmem = dot_lds_bytes(g);
  const int nw = (kx + 3) / 4, dpt = (sx + 3) / 4;
  DotFn fn = cost_type == VWGPU_CROSS_CORRELATION ? pick<VWGPU_CROSS_CORRELATION>(nw, dpt) : pick<VWGPU_SQUARED_DIFFERENCE>(nw, dpt);
  if (shmem > 64 * 1024)
    VWGPU_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  vwgpu_prof_scope ps(ctx, "bm_dot_u8");
  hipLaunchKernelGGL(fn, dim3((ow + TX - 1) / TX, (oh + TYR - 1) / TYR), dim3(NTHREADS), shmem, ctx->stream,
                     left, ls, lw, lh, right, rs, rcw, rch, g, out, os, ow, oh, flag_set, flag_clear);
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}
