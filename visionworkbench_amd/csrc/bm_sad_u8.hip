// bm_sad_u8.hip — packed-u8 SAD block matching + winner-take-all for integer-valued imagery.
//
// The headline kernel (BASELINE.json config 2: 4096^2, 7x7 SAD, 129x1 search).  ONE launch replaces the whole
// per-disparity pipeline of best_of_search_convolution (src/vw/Stereo/Correlation.cc:64-133: crop copy ->
// AbsoluteCost image -> fast_box_sum -> compare loop -> validity pass) for inputs on which the reference's
// float/double arithmetic is exact: pixel values that are integers in [0,255].  On that
// domain |a-b| in float, the float64 box sums and the strict `<` compares are all exact, so the result only
// depends on (cost, dy, dx) lexicographic order ("strict compare, first wins" == smallest (cost, dy, dx)),
// which is what this kernel tracks.  It reads the float32 images directly (4 B/px, once per tile), checks
// the domain on the fly and raises a device flag otherwise; the generic float64 kernel (bm_generic.hip)
// then recomputes the image.
//
// Issue-bound, not HBM-bound: 4096^2 x 129 disparities = 2.16 G (pixel, disparity) evaluations against
// 337 MB of compulsory HBM traffic (42 us).  Measured on MI355X (profiles/r01_ubench_valu.txt): an ordinary
// VALU op issues 16 lanes/clk/SIMD (64 lane-ops/clk/CU), v_qsad_pk_u16_u8 is quarter rate in isolation (16 abs-diffs
// in 16 clk/wave = the same 4 abs-diffs per lane-slot as v_sad_u8), so the floor of this formulation per 4
// evaluations is 2 qsad x 22/16 halo rows + 4 key + 2 min3.  Measured: a pair step (8 disparities x 4 pixels x 16
// rows per lane) takes 0.92 us of a SIMD shared by two waves — the step loops run at that issue limit; what is left
// are the staging / output phases (DESIGN.md 4.1 has the in-kernel timeline).
//
//   mapping   lane <-> 4 consecutive output pixels (q..q+3) x TY output rows; wave <-> 256 x TY pixels;
//             workgroup = 4 (or 2) waves side by side.
//   qsad      src0 (64 bit) = 8 LEFT bytes L[q+4n .. q+4n+7], in registers for all TY+ky-1 rows,
//             src1 (32 bit) = 4 RIGHT bytes R[q+j+4n .. +3] read from LDS, j = "step" = 4a+t.
//             Result slot i (u16) = sum_b |L[q+i+4n+b] - R[q+j+4n+b]|  -> pixel q+i at disparity d = j-i.
//             RIGHT sits in the 32-bit operand so that a partial last word (kx % 4 != 0) is harmless: its
//             missing bytes are zeroed in LDS, which adds sum L[...] to EVERY disparity of that pixel —
//             a per-pixel constant that changes neither the argmin nor the best==worst validity test.
//   vertical  one accumulator chain per step runs down the rows (the adds are free inside qsad); the
//             ky-row window cost is P[r] - P[r-ky] modulo 2^16 (exact: the true window sum is < 2^16).
//   WTA       key = cost << 16 | (dy*sx + dx).  v_sub_u16_sdwa writes P[r]-P[r-ky] straight into the high
//             half of a register whose low half holds the disparity index, so one op does the vertical
//             difference AND builds the key; K = v_min3_u32(K, key_a, key_a+1).  Order independent.
//   LDS       RIGHT words at byte phase t = j mod 4 are unaligned, so the step loop runs t-outer: the
//             float tile is converted to a u8 base tile in LDS once, and for each t the workgroup derives
//             one array of pre-shifted, pre-masked word groups (v_alignbyte_b32), then walks a = 0..A with
//             one 8/16-byte LDS read per row.
//   validity  invalid <=> all sx*sy costs equal.  Instead of tracking the worst cost everywhere, four cost pairs are
//             compared during the first sweep (one bit per lane and row: "some pixel of the row had equal costs in every
//             probe"); only a workgroup with a surviving row (never on textured data, always on flat data) runs the
//             second sweep — in the same launch, after its epilogue — which recomputes packed best / worst costs and
//             invalidates.  (Rounds 1-3 ran that sweep as a second launch behind a per-tile flag: 6.4 us per call for
//             starting and retiring a grid that had nothing to do.)
//   output    a lane's 4 pixels are 12 consecutive dwords, i.e. 48-byte strided stores; each wave transposes its row
//             through LDS and writes three coalesced 1 KiB pieces instead.
//   grids     1-D grid, tile = XCD-banded raster order; grids too small for two waves per SIMD run the two-wave-group
//             variant (SPLIT); tile height and variant are picked by pick_launch's cost model.
//
// Limits (else the generic path): SAD only; kernel sizes in kLaunch; 4*ceil(kx/4)*ky*255 < 65536;
// sx*sy <= 65535; LDS footprint <= 80 KiB.
#include <cstdlib>
#include <type_traits>

#include "vwgpu_internal.h"
#include "u8_tile.h"

#ifdef VWGPU_TILE_STAMPS
// Tools build only (make stamps -> tools/build/libvwgpu_stamps.so; tools/sad_timeline.py): every workgroup leaves a record of 16 u64 —
// wall clock (100 MHz) and shader clock at its start and end, wall clock after staging / after each byte phase / after the epilogue,
// and where it ran (XCC, SE, CU).  The product library does not contain any of this.
__device__ unsigned long long* g_sad_stamps = nullptr;
extern "C" int vwgpu_debug_set_sad_stamps(void* d_buf) {
  unsigned long long* p = static_cast<unsigned long long*>(d_buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(g_sad_stamps), &p, sizeof p) == hipSuccess ? 0 : -3;
}
#define VWGPU_STAMP(k) do { if (g_sad_stamps && tid == 0) g_sad_stamps[(size_t)wg * 16 + (k)] = wall_clock64(); } while (0)
#else
#define VWGPU_STAMP(k) do { } while (0)
#endif

namespace {

using namespace vwgpu_u8;
typedef uint32_t u32;
typedef uint64_t u64;
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ u32 pk_sub_u16(u32 a, u32 b) {
  return __builtin_bit_cast(u32, (u16x2)(__builtin_bit_cast(u16x2, a) - __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ u32 pk_max_u16(u32 a, u32 b) {
  return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
__device__ __forceinline__ u32 pk_min_u16(u32 a, u32 b) {
  return __builtin_bit_cast(u32, __builtin_elementwise_min(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}
// K = min(K, a, b).  `volatile` keeps the SDWA key writes and the min3 reads in program order: the key registers
// are rewritten in place every row, and letting the scheduler overlap consecutive rows costs a v_mov per key.
__device__ __forceinline__ void umin3_acc(u32& k, u32 a, u32 b) {
  asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(k) : "v"(a), "v"(b));
}

// key.hi16 = a.WORD_w - b.WORD_w (mod 2^16), IN PLACE; key.lo16 (the disparity index) is preserved, so the
// same register serves every row of a step without being re-initialised.
template <int W>
__device__ __forceinline__ void key_sub(u32& key, u32 a, u32 b) {
  if (W == 0)
    asm volatile("v_sub_u16_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0"
                 : "+v"(key) : "v"(a), "v"(b));
  else
    asm volatile("v_sub_u16_sdwa %0, %1, %2 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_1"
                 : "+v"(key) : "v"(a), "v"(b));
}

// WV: wavefronts side by side (0 = 4 for kernels up to 8 columns, 2 for wider ones).  Round 6: WV = 2 with two wave groups is the
// 512-column tile for small grids — a row strip of an 8-GPU run is one tile per CU either way, and what a CU then pays is its tile's
// halo: 22 input rows for 16 output rows on 512 columns instead of 14 for 8 on 1024 (tools/sad_timeline.py: staging 9.0 -> 7.x us,
// the four byte phases 53 -> 4x us per tile).
template <int KX, int KY, int TY, int WV = 0>
struct Cfg {
  static constexpr int NW = (KX + 3) / 4;          // qsads per row
  static constexpr int EW = NW <= 2 ? 2 : 4;       // dwords per LDS entry
  static constexpr int WAVES = WV ? WV : (NW <= 2 ? 4 : 2);    // waves side by side
  static constexpr int THREADS = WAVES * 64;
  static constexpr int TWB = WAVES * 256;          // output columns per workgroup
  static constexpr int NR = TY + KY - 1;           // input rows per tile
  static constexpr int LBW = TWB / 4 + NW + 1;     // dwords per row of the LEFT u8 tile
  static constexpr u32 LAST_MASK = (KX % 4 == 0) ? 0xffffffffu : ((1u << (8 * (KX % 4))) - 1u);
};

// idx / d and idx % d for 0 <= idx < 2^20 without an integer division (d is workgroup-uniform, inv = 1.0f / d).
__device__ __forceinline__ void divmod_small(int idx, int d, float inv, int& quo, int& rem) {
  int qq = (int)(((float)idx + 0.5f) * inv);
  int rr = idx - qq * d;
  if (rr < 0) { --qq; rr += d; }
  if (rr >= d) { ++qq; rr -= d; }
  quo = qq; rem = rr;
}

// dwords of the entry array: the word groups of a byte phase, or what borrows the array after the sweep (see the kernel)
__host__ __device__ constexpr size_t ent_words(int nr, int ne, int ew, int gr, int ty, int pt) {
  const size_t ent = (size_t)nr * ne * ew;
  const size_t merge = gr > 1 ? (size_t)(gr - 1) * (ty / gr) * 4 * pt + (size_t)gr * pt : 0;
  const size_t ob = (size_t)gr * (pt / 64) * 768;
  return ent > merge ? (ent > ob ? ent : ob) : (merge > ob ? merge : ob);
}

// One launch: the matcher sweep (keys + equality probe), the epilogue, and — only in workgroups where some pixel may be
// invalid — the validity sweep: packed best / worst costs recomputed, pixels with best == worst invalidated
// (Correlation.cc:121-133).
// SPLIT: for grids too small to give every SIMD two waves (a 1/8 row strip of the 4096^2 case is 256 eight-row tiles):
//        the workgroup has TWO wave groups on the same tile.  Both hold the LEFT windows; wave w of group 0 and wave w
//        of group 1 cover the same pixels and sit on the same SIMD, draw the step items of a phase from a shared LDS
//        counter (the arbiter favours the older wave, so a fixed split would leave one group waiting), and the partial
//        keys are merged through LDS at the end of the tile — a key minimum is order independent.  (The validity sweep
//        is not split: group 0 walks its items alone, group 1 only helps to build the word groups.)
// GR = number of wave groups on the tile (1, 2, 4); SPLIT = GR > 1.  GR = 4 (round 6) exists for the 512-column tile: two wave columns x
// four groups = the same eight wavefronts per CU as the two-group 1024-column tile.
template <int KX, int KY, int TY, int GR, int WV = 0>
__global__ void __launch_bounds__((GR * (Cfg<KX, KY, TY, WV>::THREADS)), (GR > 1 ? 1 : (TY <= 8 && KX <= 8 ? 3 : 2)))
bm_sad_u8_kernel(const float* __restrict__ L, ptrdiff_t ls, int lw, int lh,
                 const float* __restrict__ R, ptrdiff_t rs, int rcw, int rch,
                 int sx, int sy, int ne, int32_t* __restrict__ out, ptrdiff_t os, int ow, int oh,
                 int* __restrict__ flag_set, int* __restrict__ flag_clear,
                 int gxt, int ntiles) {
  typedef Cfg<KX, KY, TY, WV> C;
  constexpr bool SPLIT = GR > 1;
  constexpr int NW = C::NW, EW = C::EW, NR = C::NR;
  extern __shared__ __attribute__((aligned(16))) u32 lds[];
  u32* ent = lds;                                   // [NR][ne][EW]   word groups of the current byte phase
  const int bpitch = ne + NW + 1;                   // dwords per row of the RIGHT u8 base tile
  // The entry array is borrowed twice after the sweep: by the merge of the wave groups ((GR - 1) x TY / GR x 4 x PT keys + GR x PT equality
  // words) and by the epilogue's per-wave transposition buffers (768 dwords each).  With a narrow search both can exceed NR x ne x EW — and
  // the RIGHT base tile behind it is still needed by the validity sweep (found by the round-6 campaign: constant images, search 2 .. 8 on the
  // 512-column four-group tile).  ent_words() is the size both sides use.
  u32* base = lds + ent_words(NR, ne, EW, GR, TY, C::THREADS);           // [NR][bpitch]
  u32* item_ctr = base + (size_t)NR * bpitch;       // SPLIT: one work item counter per wave pair

  constexpr int PT = C::THREADS;                    // threads that map to pixels
  constexpr int NT = GR * PT;                       // threads of the workgroup
  const int tid = threadIdx.x;
  const int grp = SPLIT ? __builtin_amdgcn_readfirstlane(tid / PT) : 0;    // wave-uniform
  const int ltid = SPLIT ? tid % PT : tid;
  const int pair_id = __builtin_amdgcn_readfirstlane(ltid >> 6);
  // XCD-aware tile order: workgroup i runs on XCD i % 8 (round-robin dispatch), so XCD x takes the contiguous band of
  // tiles [x * per, (x + 1) * per) — whole tile rows in raster order.  Tiles that share halo rows (ky-1 of TY+ky-1) or
  // output cache lines are then staged at the same time behind the same L2 instead of being fetched once per XCD
  // (FETCH_SIZE 203 -> 148 MB per launch on the 4096^2 case; the kernel time does not move: staging is latency bound).
  const int per_xcd = gridDim.x >> 3;
  const int wg = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);   // tile index, raster order
  if (wg >= ntiles) return;                         // grid rounded up to a multiple of 8
  const int x0 = (wg % gxt) * C::TWB;
  const int y0 = (wg / gxt) * TY;
  const int q = x0 + 4 * ltid;                      // first of the lane's 4 output pixels
  u32 bad_acc = 0;                                  // non-zero: some input pixel is not an integer in [0,255]
  if (wg == 0 && tid == 0) *flag_clear = 0;         // the NEXT call's flag
#ifdef VWGPU_TILE_STAMPS
  if (g_sad_stamps && tid == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* rec = g_sad_stamps + (size_t)wg * 16;
    rec[0] = wall_clock64(); rec[1] = clock64(); rec[2] = ((unsigned long long)xcc << 32) | hw;
  }
#endif

  // ---- LEFT: float tile -> u8 in LDS (borrowing the entry array) -> per-lane register windows ----
  // Both tiles are staged before anything else is live in registers (the LEFT tile borrows the entry array).
  // (both images' main loads in flight before the first conversion — one memory latency instead of two — was tried for the
  // latency-bound 1/8 strips in round 4: nothing there, 2 us slower at 4096^2; the tiles are staged one after the other)
  stage_u8_rows<NR>(L, ls, lw, lh, x0, y0, C::LBW, C::LBW, ent, tid, NT, bad_acc);
  stage_u8_rows<NR>(R, rs, rcw, rch, x0, y0, bpitch, bpitch, base, tid, NT, bad_acc);
  __syncthreads();
  u64 win[NR][NW];                                  // win[r][n] = bytes L[q+4n .. q+4n+7]
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    u32 a[NW + 1];
#pragma unroll
    for (int n = 0; n <= NW; ++n) a[n] = ent[r * C::LBW + ltid + n];
#pragma unroll
    for (int n = 0; n < NW; ++n) win[r][n] = (u64)a[n] | ((u64)a[n + 1] << 32);
  }

  VWGPU_STAMP(3);
  u32 K[TY][4];                                     // best key per pixel: cost << 16 | disparity index
  u32 MN[TY][2], MX[TY][2];                         // packed best / worst cost (validity sweep only)
#pragma unroll
  for (int y = 0; y < TY; ++y) K[y][0] = K[y][1] = K[y][2] = K[y][3] = 0xffffffffu;
  // bit y: "in every probed step pair, some in-image pixel of this lane's row y had two equal costs".  A superset of
  // the rows that hold an invalid pixel (all costs equal), kept per row rather than per pixel: one min tree and one
  // bit per row is all the bookkeeping a probe costs.  On noise one pixel pair is equal with p ~ 7e-4, so a row of 4
  // survives NPROBE = 4 probes with ~6e-11; a flat (truly invalid) pixel always passes.
  u32 eq_rows = (1u << TY) - 1u;
  int eq_checks = 0;
  constexpr int NPROBE = 4;
  const u32 zero = 0;

  typedef std::true_type T;
  typedef std::false_type F;

  // MAXSWEEP = false: the matcher sweep (keys K, equality probe).  MAXSWEEP = true: the validity sweep (packed MN / MX).
  auto sweep = [&](auto max_tag) __attribute__((always_inline)) {
    constexpr bool MAXSWEEP = decltype(max_tag)::value;
    for (int dy = 0; dy < sy; ++dy) {
      __syncthreads();                              // everyone is done with the previous base tile / windows
      // (the validity sweep finds the base tile of the last dy: with more than one search row it stages row 0 again)
      if (dy > 0 || (MAXSWEEP && sy > 1)) stage_u8_rows<NR>(R, rs, rcw, rch, x0, y0 + dy, bpitch, bpitch, base, tid, NT, bad_acc);
      for (int t = 0; t < 4; ++t) {
        const int a_last = (sx + 2 - t) >> 2;       // last step with any valid slot
        __syncthreads();                            // base staged / previous phase's readers done
        {
          auto build = [&](int r, int m) __attribute__((always_inline)) {
            const u32* bp = base + (size_t)r * bpitch + m;
            u32 b[NW + 1];
#pragma unroll
            for (int n = 0; n <= NW; ++n) b[n] = bp[n];
            u32 w[EW];
#pragma unroll
            for (int n = 0; n < EW; ++n) w[n] = 0;
#pragma unroll
            for (int n = 0; n < NW; ++n) w[n] = __builtin_amdgcn_alignbyte(b[n + 1], b[n], t);
            w[NW - 1] &= C::LAST_MASK;
            u32* e = ent + ((size_t)r * ne + m) * EW;
            if (EW == 2) *reinterpret_cast<uint2*>(e) = make_uint2(w[0], w[1]);
            else *reinterpret_cast<uint4*>(e) = make_uint4(w[0], w[1], w[2 % EW], w[3 % EW]);
          };
          // entries 0..PT-1 of every row: column = ltid, no index arithmetic (ne > PT always); SPLIT: rows alternate
          // between the two wave groups
#pragma unroll 4
          for (int r = grp; r < NR; r += NT / PT) build(r, ltid);
          // the search margin: NR * (ne - PT) entries spread over the workgroup
          const int rem = ne - PT, total = NR * rem;
          const float inv = 1.0f / (float)rem;
          for (int idx = tid; idx < total; idx += NT) {
            int r, m;
            divmod_small(idx, rem, inv, r, m);
            build(r, PT + m);
          }
          if (SPLIT && tid < PT / 64) item_ctr[tid] = 0u;
        }
        __syncthreads();

        // One step = one accumulator chain down the NR rows.  MASKED handles the range ends (some slots outside
        // [0,sx)); PAIR processes steps a and a+1 together so the WTA is one v_min3_u32 per pixel; PROBE adds the
        // equality probe of the validity pre-filter (two iterations per workgroup).
        auto step = [&](int a, auto masked_tag, auto pair_tag, auto probe_tag) __attribute__((always_inline)) {
          constexpr bool MASKED = decltype(masked_tag)::value;
          constexpr bool PAIR = decltype(pair_tag)::value;
          constexpr bool PROBE = decltype(probe_tag)::value;
          const int jbase = 4 * a + t;              // d for slot i is jbase - i
          const int ibase = dy * sx + jbase;
          u32 kinA[4], kinB[4];                     // key registers: low half = disparity index
          u32 orA[2] = {0u, 0u};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            kinA[i] = (u32)(ibase - i) & 0xffffu;
            kinB[i] = (u32)(ibase + 4 - i) & 0xffffu;
            if (MASKED) {
              const int d = jbase - i;
              if (d < 0 || d >= sx) orA[i >> 1] |= (i & 1) ? 0xffff0000u : 0x0000ffffu;
            }
          }
          const u32* ep = ent + (size_t)(ltid + a) * EW;
          u64 accA = 0, accB = 0;
          u64 PA[NR], PB[NR];
          // LDS reads are issued PF rows ahead by hand: the volatile asm statements below are scheduling barriers
          // for memory operations, so the compiler will not hoist them itself.
          constexpr int PF = 4;
          u32 wa[NR][EW], wb[NR][EW];
          auto fetch = [&](int r) __attribute__((always_inline)) {
            const u32* e = ep + (size_t)r * ne * EW;
            if (EW == 2) {
              const uint2 v = *reinterpret_cast<const uint2*>(e);
              wa[r][0] = v.x; wa[r][1] = v.y;
              if (PAIR) { const uint2 u = *reinterpret_cast<const uint2*>(e + EW); wb[r][0] = u.x; wb[r][1] = u.y; }
            } else {
              const uint4 v = *reinterpret_cast<const uint4*>(e);
              wa[r][0] = v.x; wa[r][1] = v.y; wa[r][2 % EW] = v.z; wa[r][3 % EW] = v.w;
              if (PAIR) { const uint4 u = *reinterpret_cast<const uint4*>(e + EW); wb[r][0] = u.x; wb[r][1] = u.y; wb[r][2 % EW] = u.z; wb[r][3 % EW] = u.w; }
            }
          };
#pragma unroll
          for (int r = 0; r < PF && r < NR; ++r) fetch(r);
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            if (r + PF < NR) fetch(r + PF);
#pragma unroll
            for (int n = 0; n < NW; ++n) {
              accA = __builtin_amdgcn_qsad_pk_u16_u8(win[r][n], wa[r][n], accA);
              if (PAIR) accB = __builtin_amdgcn_qsad_pk_u16_u8(win[r][n], wb[r][n], accB);
            }
            PA[r] = accA;
            if (PAIR) PB[r] = accB;
            if (r >= KY - 1) {
              const int y = r - (KY - 1);
              const int ro = (r >= KY) ? r - KY : 0;
              const u32 nA0 = (u32)PA[r], nA1 = (u32)(PA[r] >> 32);
              const u32 oA0 = (r >= KY) ? (u32)PA[ro] : zero;
              const u32 oA1 = (r >= KY) ? (u32)(PA[ro] >> 32) : zero;
              if (MAXSWEEP) {
                const u32 sA0 = pk_sub_u16(nA0, oA0), sA1 = pk_sub_u16(nA1, oA1);
                MN[y][0] = pk_min_u16(MN[y][0], MASKED ? (sA0 | orA[0]) : sA0);
                MN[y][1] = pk_min_u16(MN[y][1], MASKED ? (sA1 | orA[1]) : sA1);
                MX[y][0] = pk_max_u16(MX[y][0], MASKED ? (sA0 & ~orA[0]) : sA0);
                MX[y][1] = pk_max_u16(MX[y][1], MASKED ? (sA1 & ~orA[1]) : sA1);
                if (PAIR) {
                  const u32 nB0 = (u32)PB[r], nB1 = (u32)(PB[r] >> 32);
                  const u32 oB0 = (r >= KY) ? (u32)PB[ro] : zero;
                  const u32 oB1 = (r >= KY) ? (u32)(PB[ro] >> 32) : zero;
                  const u32 sB0 = pk_sub_u16(nB0, oB0), sB1 = pk_sub_u16(nB1, oB1);
                  MN[y][0] = pk_min_u16(MN[y][0], sB0);
                  MN[y][1] = pk_min_u16(MN[y][1], sB1);
                  MX[y][0] = pk_max_u16(MX[y][0], sB0);
                  MX[y][1] = pk_max_u16(MX[y][1], sB1);
                }
              } else {
                key_sub<0>(kinA[0], nA0, oA0); key_sub<1>(kinA[1], nA0, oA0);
                key_sub<0>(kinA[2], nA1, oA1); key_sub<1>(kinA[3], nA1, oA1);
                if (PAIR) {
                  const u32 nB0 = (u32)PB[r], nB1 = (u32)(PB[r] >> 32);
                  const u32 oB0 = (r >= KY) ? (u32)PB[ro] : zero;
                  const u32 oB1 = (r >= KY) ? (u32)(PB[ro] >> 32) : zero;
                  key_sub<0>(kinB[0], nB0, oB0); key_sub<1>(kinB[1], nB0, oB0);
                  key_sub<0>(kinB[2], nB1, oB1); key_sub<1>(kinB[3], nB1, oB1);
                  if (PROBE) {
                    // cost fields differ <=> xor >= 2^16; pixels right of the image (zero padding: every cost equal)
                    // must not count.  (volatile: the keys are rewritten in place by the SDWA subtractions, so these
                    // reads must stay in program order or the compiler copies every key and spills the windows)
                    u32 x[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                      asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x[i]) : "v"(kinA[i]), "v"(kinB[i]));
                      x[i] = (q + i < ow) ? x[i] : 0xffffffffu;
                    }
                    const u32 m01 = x[0] < x[1] ? x[0] : x[1], m23 = x[2] < x[3] ? x[2] : x[3];
                    if ((m01 < m23 ? m01 : m23) >> 16) eq_rows &= ~(1u << y);
                  }
                  umin3_acc(K[y][0], kinA[0], kinB[0]);
                  umin3_acc(K[y][1], kinA[1], kinB[1]);
                  umin3_acc(K[y][2], kinA[2], kinB[2]);
                  umin3_acc(K[y][3], kinA[3], kinB[3]);
                } else {
                  // range ends: slots outside [0,sx) must never win
                  const u32 kA0 = (MASKED && (orA[0] & 0x0000ffffu)) ? 0xffffffffu : kinA[0];
                  const u32 kA1 = (MASKED && (orA[0] & 0xffff0000u)) ? 0xffffffffu : kinA[1];
                  const u32 kA2 = (MASKED && (orA[1] & 0x0000ffffu)) ? 0xffffffffu : kinA[2];
                  const u32 kA3 = (MASKED && (orA[1] & 0xffff0000u)) ? 0xffffffffu : kinA[3];
                  umin3_acc(K[y][0], kA0, kA0);
                  umin3_acc(K[y][1], kA1, kA1);
                  umin3_acc(K[y][2], kA2, kA2);
                  umin3_acc(K[y][3], kA3, kA3);
                }
              }
            }
          }
        };

        // clean range: every slot valid  <=>  jbase-3 >= 0 and jbase <= sx-1
        int a_lo = (t >= 3) ? 0 : 1;
        int a_hi = (sx - 1 - t >= 0) ? ((sx - 1 - t) >> 2) + 1 : 0;   // exclusive
        if (a_hi > a_last + 1) a_hi = a_last + 1;
        if (a_lo > a_hi) a_lo = a_hi;
        // The phase as a list of work items: [n1 masked singles][npair pairs, the first nprobe probed][n4 masked singles].
        const int n1 = a_lo;
        const int npair = (a_hi - n1) > 0 ? (a_hi - n1) >> 1 : 0;
        int nprobe = 0;
        if (!MAXSWEEP && dy == 0 && t == 0) nprobe = npair < NPROBE - eq_checks ? npair : NPROBE - eq_checks;
        const int a4 = n1 + 2 * npair;
        const int n4 = a_last - a4 + 1 > 0 ? a_last - a4 + 1 : 0;
        const int nitems = n1 + npair + n4;
        int seq = 0;
        auto next_item = [&]() __attribute__((always_inline)) -> int {
          if (MAXSWEEP) return grp == 0 ? seq++ : 0x7fffffff;      // not split: group 1 takes nothing
          if (!SPLIT) return seq++;
          int lane, v = 0;
          asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
          if (lane == 0) v = (int)atomicAdd(&item_ctr[pair_id], 1u);
          return __builtin_amdgcn_readfirstlane(v);
        };
        // item indices only grow, so a wave walks the four loops in order (one loop per step flavour keeps the register
        // allocation of the hot pair loop)
        int i = next_item();
        for (; i < n1; i = next_item()) step(i, T{}, F{}, F{});
        for (; i < n1 + nprobe; i = next_item()) step(n1 + 2 * (i - n1), F{}, T{}, T{});
        for (; i < n1 + npair; i = next_item()) step(n1 + 2 * (i - n1), F{}, T{}, F{});
        for (; i < nitems; i = next_item()) step(a4 + (i - n1 - npair), T{}, F{}, F{});
        eq_checks += nprobe;
        if (!MAXSWEEP && dy == 0) VWGPU_STAMP(4 + t);
      }
    }
  };
  sweep(F{});

  if (bad_acc != 0u) atomicOr(flag_set, 1);
  // ---- SPLIT: merge the two wave groups.  Group 0 finishes rows [0,TY/2), group 1 rows [TY/2,TY): each hands the
  // other its partial keys of the other's rows (and its equality bits) through the entry array, free by now.
  // Group g finishes rows [g TY / GR, (g + 1) TY / GR): round d hands group d the partial keys of ITS rows from every other group (and
  // everybody's equality bits) through the entry array, free by now.
  constexpr int RG = TY / GR;                       // rows a group finishes
  static_assert(TY % GR == 0, "rows per group");
  auto row_mine = [&](int y) __attribute__((always_inline)) -> bool { return !SPLIT || (y / RG == grp); };
  if (SPLIT) {
    u32* xk = ent;                                  // [GR - 1][RG][4][PT] keys, then [GR][PT] equality words
    u32* xe = ent + (size_t)(GR - 1) * RG * 4 * PT;
    u32 eq_all = eq_rows;
#pragma unroll
    for (int d = 0; d < GR; ++d) {                  // round d: everybody -> group d
      __syncthreads();                              // steps done / previous round read
      if (grp != d) {
        const int slot = grp < d ? grp : grp - 1;
#pragma unroll
        for (int y = 0; y < RG; ++y)
#pragma unroll
          for (int s2 = 0; s2 < 4; ++s2) xk[((slot * RG + y) * 4 + s2) * PT + ltid] = K[d * RG + y][s2];
      }
      if (d == 0) xe[grp * PT + ltid] = eq_rows;
      __syncthreads();
      if (grp == d) {
#pragma unroll
        for (int slot = 0; slot < GR - 1; ++slot)
#pragma unroll
          for (int y = 0; y < RG; ++y)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
              const u32 o = xk[((slot * RG + y) * 4 + s2) * PT + ltid];
              K[d * RG + y][s2] = o < K[d * RG + y][s2] ? o : K[d * RG + y][s2];
            }
      }
      if (d == 0) {
#pragma unroll
        for (int g2 = 0; g2 < GR; ++g2) eq_all &= xe[g2 * PT + ltid];      // (xe lies behind the key slots: not overwritten by later rounds)
      }
    }
    eq_rows = eq_all;                               // every group agrees
  }
  // ---- validity: only rows in which some pixel's probed costs were all equal can hold an invalid pixel -----------
  u32 cand = eq_checks < NPROBE ? (1u << TY) - 1u : eq_rows;   // small search range: nothing is known -> every row
  // rows outside the output image (zero padding: every cost equal) must not trigger the second sweep
#pragma unroll
  for (int y = 0; y < TY; ++y)
    if (y0 + y >= oh || q >= ow || !row_mine(y)) cand &= ~(1u << y);
  const int any = __syncthreads_or(cand != 0u);     // never set on textured imagery; also: the entry array is free now

  // ---- epilogue: decode keys, store {dx, dy, VALID} in the PixelMask<Vector2i> layout ----------------------------
  // A lane owns 4 pixels = 12 consecutive dwords, so direct stores would be 48-byte strided (measured: 80 us of the
  // 400 on the 4096^2 case).  Each wave transposes its 256-pixel row (768 dwords) through LDS and writes it as
  // three fully coalesced 1 KiB stores.
  {
    const int wv = ltid >> 6, lane = ltid & 63;
    u32* ob = ent + (size_t)(grp * (PT / 64) + wv) * 768;
    const int xw = x0 + 256 * wv;
    int nd = (ow - xw) * 3;                         // dwords of this wave's row segment inside the image
    nd = nd < 0 ? 0 : (nd > 768 ? 768 : nd);
    // di / sx for di < 2^16 as a multiply-high by ceil(2^32 / sx) (exact: di * (magic * sx - 2^32) < 2^32); sx == 1 apart
    const u32 sx_magic = sx > 1 ? (u32)(0xffffffffu / (u32)sx) + 1u : 0u;
    struct __attribute__((packed, aligned(4))) U4 { u32 x, y, z, w; };
#pragma unroll
    for (int y = 0; y < TY; ++y) {
      const int oy = y0 + y;
      if (oy >= oh || !row_mine(y)) continue;
      u32 v[12];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const u32 di = K[y][s] & 0xffffu;
        u32 dx, dy;
        if (sy == 1) { dx = di; dy = 0; }
        else { dy = sx == 1 ? di : __umulhi(di, sx_magic); dx = di - dy * (u32)sx; }
        v[3 * s] = dx;
        v[3 * s + 1] = dy;
        v[3 * s + 2] = 0x7fffffffu;
      }
      uint4* ow4 = reinterpret_cast<uint4*>(ob + lane * 12);
      ow4[0] = make_uint4(v[0], v[1], v[2], v[3]);
      ow4[1] = make_uint4(v[4], v[5], v[6], v[7]);
      ow4[2] = make_uint4(v[8], v[9], v[10], v[11]);
      int32_t* orow = out + ((ptrdiff_t)oy * os + xw) * 3;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int idx = k * 256 + lane * 4;
        const uint4 t = *reinterpret_cast<const uint4*>(ob + idx);
        if (idx + 3 < nd) {
          U4 u; u.x = t.x; u.y = t.y; u.z = t.z; u.w = t.w;
          *reinterpret_cast<U4*>(orow + idx) = u;
        } else {
          if (idx < nd) orow[idx] = (int32_t)t.x;
          if (idx + 1 < nd) orow[idx + 1] = (int32_t)t.y;
          if (idx + 2 < nd) orow[idx + 2] = (int32_t)t.z;
        }
      }
    }
  }

#ifdef VWGPU_TILE_STAMPS
  if (g_sad_stamps && tid == 0) { g_sad_stamps[(size_t)wg * 16 + 8] = wall_clock64(); g_sad_stamps[(size_t)wg * 16 + 9] = clock64(); }
#endif
  // ---- validity sweep (workgroup-uniform; never taken on textured imagery): the same steps with packed best / worst
  // costs instead of keys, then best == worst => invalid (Correlation.cc:121-133).  The epilogue's stores above are
  // complete before the barrier at the head of the sweep, the zeros below land after them.
  if (__builtin_expect(!any, 1)) return;
#pragma unroll
  for (int y = 0; y < TY; ++y) {
    MN[y][0] = MN[y][1] = 0xffffffffu;
    MX[y][0] = MX[y][1] = 0u;
  }
  sweep(T{});
  if (grp == 0) {
#pragma unroll
    for (int y = 0; y < TY; ++y) {
      const u32 same0 = MN[y][0] ^ MX[y][0], same1 = MN[y][1] ^ MX[y][1];
      const bool inv[4] = {(same0 & 0xffffu) == 0, (same0 >> 16) == 0, (same1 & 0xffffu) == 0, (same1 >> 16) == 0};
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (inv[s] && y0 + y < oh && q + s < ow)
          out[((ptrdiff_t)(y0 + y) * os + q + s) * 3 + 2] = 0;
    }
  }
}

typedef void (*KernelFn)(const float*, ptrdiff_t, int, int, const float*, ptrdiff_t, int, int, int, int, int,
                         int32_t*, ptrdiff_t, int, int, int*, int*, int, int);
struct Launch {
  int kx, ky, ty;
  int threads, twb, nr, ew, nw;
  int split_groups;          // wave groups of split_fn
  KernelFn fn;
  KernelFn split_fn;         // the two-wave-group matcher for small grids (nullptr: not instantiated for this size)
};

template <int KX, int KY, int TY, bool WITH_SPLIT = false, int WV = 0>
constexpr Launch make_launch() {
  typedef Cfg<KX, KY, TY, WV> C;
  return Launch{KX, KY, TY, C::THREADS, C::TWB, C::NR, C::EW, C::NW, WV ? 4 : 2,
                WV ? nullptr : bm_sad_u8_kernel<KX, KY, TY, 1, WV>,              // (the narrow tile exists as the four-group matcher only)
                WITH_SPLIT ? bm_sad_u8_kernel<KX, KY, TY, (WV ? 4 : 2), WV> : nullptr};
}

// Instantiated kernel sizes.  Others fall back to the generic path.
// Two tile heights for the headline size.  A workgroup is 4 waves x 256 columns x TY rows; the 4096^2 image is 4 x 256 = 1024
// of them with TY = 16 — 4 per CU, one round — and the kernel time is linear in workgroups per CU (107 us for one, 407 us for
// four: issue bound from the first wave per SIMD).  A 1/8 row strip is only 128 such workgroups (half the CUs idle), so
// small images / multi-GPU strips use 8-row tiles when 16-row tiles would leave fewer than 2 workgroups per CU
// (tools/time_strips.py: 1/4 strip 137 -> 128 us, 1/8 strip 128 -> 84 us; 4-row tiles never win — 10/4 halo rows).
const Launch kLaunch[] = {
    make_launch<3, 3, 16>(), make_launch<5, 5, 16>(), make_launch<7, 7, 16, true>(), make_launch<7, 7, 8, true>(), make_launch<7, 7, 16, true, 2>(),
    make_launch<7, 5, 16>(), make_launch<9, 9, 12>(), make_launch<11, 11, 8>(),
};

// first entry of the size = the preferred (tallest) tile
const Launch* find_launch(int kx, int ky) {
  for (const Launch& l : kLaunch)
    if (l.kx == kx && l.ky == ky) return &l;
  return nullptr;
}

// Picks tile height and one- or two-group matcher by a small cost model of the busiest CU (units = tile rows of step
// work on a saturated SIMD pair; constants from tools/time_strips.py and the in-kernel timeline of DESIGN.md 4.1):
//   n        tiles on the busiest CU = ceil(workgroups / CUs)
//   steps    one group : rows * 0.5 * n when >= 2 tiles share the SIMDs, rows * 0.74 for a lone 4-wave workgroup (one wave
//                        per SIMD reaches ~2/3 of the issue rate);  two groups: rows * 0.55 * n (8 waves on one tile)
//   rounds   sequential staging / output phases: ceil(n / resident) for one group, n for two; ~4.5 row-units each
// 4096^2: 16-row tiles, one group (4 per CU).  1/4 strip: 16-row tiles, two groups (exactly one per CU; 8-row tiles would
// put a third tile on a few CUs).  1/8 strip: 512-column tiles of 16 rows, four groups (round 6; 8-row tiles with two groups before).
const Launch* pick_launch(int kx, int ky, int ow, int oh, int num_cu, int groups, bool* split) {
  const Launch* best = nullptr;
  double best_cost = 0.0;
  *split = false;
  if (groups == 3) {                                 // sizes without the narrow flavour keep the launcher's choice
    bool has = false;
    for (const Launch& l : kLaunch) has |= l.kx == kx && l.ky == ky && l.split_fn && !l.fn;
    if (!has) groups = 0;
  }
  for (const Launch& l : kLaunch) {
    if (l.kx != kx || l.ky != ky) continue;
    const long long wgs = (long long)((ow + l.twb - 1) / l.twb) * ((oh + l.ty - 1) / l.ty);
    const double n = (double)((wgs + num_cu - 1) / num_cu);
    const int resident = (l.ty <= 8 && l.kx <= 8) ? 3 : 2;
    const bool narrow = l.twb * 2 <= 4 * 256 && l.split_fn && !l.fn;         // the 512-column two-group tile
    const double wfrac = narrow ? 0.5 : 1.0;
    for (int sp = 0; sp < 2; ++sp) {
      if (sp && !l.split_fn) continue;
      if (!sp && !l.fn) continue;
      if (groups == 3 && !narrow) continue;                                    // VWGPU_OPT_SAD_GROUPS pins the flavour (same results)
      if (groups && groups != 3 && (narrow || (l.split_fn && (groups == 2) != (sp == 1)))) continue;
      const double steps = (sp ? l.nr * 0.55 * n : (n < 2.0 ? l.nr * 0.74 : l.nr * 0.5 * n)) * wfrac;
      const double rounds = sp ? n : (double)(((long long)n + resident - 1) / resident);
      const double cost = steps + (narrow ? 5.7 : 4.5) * rounds;               // (four groups: four merge rounds; tools/time_strip_step.py: 59.7 vs 61.7 us for a 1/8 strip)
      if (!best || cost < best_cost) { best = &l; best_cost = cost; *split = sp != 0; }
    }
  }
  return best;
}

constexpr size_t kMaxLds = 80 * 1024;   // two workgroups per CU

int entries_per_row(const Launch& l, int sx) { return l.twb / 4 + ((sx + 2) >> 2) + 1; }

size_t lds_bytes(const Launch& l, int sx, int groups) {
  const int ne = entries_per_row(l, sx);
  const size_t ent = ent_words(l.nr, ne, l.ew, groups, l.ty, l.threads);
  const size_t left = (size_t)l.nr * (l.twb / 4 + l.nw + 1);        // borrowed from the entry array
  const size_t base = (size_t)l.nr * (ne + l.nw + 1);
  return ((ent > left ? ent : left) + base) * sizeof(u32);
}

}  // namespace

bool vwgpu_bm_sad_u8_supported(int cost_type, int kx, int ky, int sx, int sy) {
  if (cost_type != VWGPU_ABSOLUTE_DIFFERENCE) return false;
  const Launch* l = find_launch(kx, ky);
  if (!l) return false;
  if ((long long)sx * sy > 65535) return false;
  return lds_bytes(*l, sx, l->split_fn ? l->split_groups : 1) <= kMaxLds;
}

int vwgpu_launch_bm_sad_u8(vwgpu_ctx* ctx,
                           const float* left, int lw, int lh, ptrdiff_t ls,
                           const float* right, int rw, int rh, ptrdiff_t rs,
                           int kx, int ky, int sx, int sy, int32_t* out, ptrdiff_t os,
                           int** d_fallback_flag) {
  (void)rw; (void)rh;
  const int ow = lw - kx + 1, oh = lh - ky + 1;
  bool split = false;
  const Launch* l = pick_launch(kx, ky, ow, oh, ctx->num_cu, ctx->sad_groups, &split);
  if (!l) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "no packed-u8 kernel for %dx%d", kx, ky);
  const int rcw = lw + sx - 1, rch = lh + sy - 1;
  const int gx = (ow + l->twb - 1) / l->twb, gy = (oh + l->ty - 1) / l->ty;
  const int ne = entries_per_row(*l, sx);

  // Two alternating device flags: call n raises flags[n&1] on unsuitable input and clears flags[(n+1)&1]
  // for the next call, so no memset launch is needed (stream order makes this race free).
  int *flag_set = nullptr, *flag_clear = nullptr;
  int rc = vwgpu_next_flags(ctx, 0, &flag_set, &flag_clear, nullptr);
  if (rc) return rc;
  *d_fallback_flag = flag_set;
  const size_t shmem = lds_bytes(*l, sx, split ? l->split_groups : 1) + (split ? 64 : 0);   // + the item counters of the split variant
  const KernelFn main_fn = split ? l->split_fn : l->fn;
  const unsigned grid1 = (unsigned)((gx * gy + 7) / 8 * 8);   // one tile per workgroup, see the XCD note in the kernel
  if (shmem > 64 * 1024)
    VWGPU_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(main_fn),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  {
    vwgpu_prof_scope ps(ctx, "bm_sad_u8");
    hipLaunchKernelGGL(main_fn, dim3(grid1), dim3(split ? l->split_groups * l->threads : l->threads), shmem, ctx->stream,
                       left, ls, lw, lh, right, rs, rcw, rch, sx, sy, ne, out, os, ow, oh,
                       flag_set, flag_clear, gx, gx * gy);
  }
  VWGPU_HIP(ctx, hipGetLastError());
  return VWGPU_OK;
}
