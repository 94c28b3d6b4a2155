// vwgpu_internal.h — context object and launch helpers shared by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "vwgpu.h"

struct vwgpu_prof_rec {
  const char* name;
  hipEvent_t a, b;
};

// Grow-only device scratch arena: one allocation reused across calls so that steady-state calls do no
// hipMalloc/hipFree (they would serialise the stream).
struct vwgpu_arena {
  void* base = nullptr;
  size_t cap = 0;
};

struct vwgpu_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  bool stream_is_own = true;
  std::string err;
  int forced_path = VWGPU_PATH_NONE;
  int last_path = VWGPU_PATH_NONE;
  bool profiling = false;
  std::vector<vwgpu_prof_rec> prof;
  std::vector<hipEvent_t> event_pool;
  std::vector<void*> graveyard;   // outgrown arena blocks (vwgpu_arena_reserve): released by vwgpu_synchronize / vwgpu_trim / with the context
  size_t graveyard_bytes = 0;
  vwgpu_arena scratch;   // kernel scratch (NCC precision images)
  vwgpu_arena flags;     // two alternating "input not representable" flags of the packed-u8 path
  bool flags_init = false;
  void* flags_base_seen = nullptr;
  int flag_parity = 0;
  int* last_flag = nullptr;
  vwgpu_arena staging;   // device copies of host images for the host-pointer entry points
  vwgpu_arena filt;      // intermediate image of composite filters (prefilter_image)
  vwgpu_arena misc;      // small device words (disparity range of parabola_subpixel)
  vwgpu_arena pyr;       // pyramids, masks and per-level disparities of one pyramid_correlate tile
  vwgpu_arena ztab;      // zone / tile tables of the batched zone kernels (two halves, alternating)
  int ztab_parity = 0;
  vwgpu_arena zrl;       // right-to-left disparity images of all zones of pyramid level 0
  vwgpu_arena zext;      // zone scheduler: leaf boxes of the quad tree and their measured disparity extents
  vwgpu_arena sgm;       // SGM: u8 images, census words, disparity bounds, ragged starts
  vwgpu_arena sgm_main;  // SGM: ragged cost (u8) + accumulated cost (u16) buffers
  vwgpu_arena sgm_bnd;   // SGM fused sweeps: ticket counter + boundary rows handed from block to block (zero-initialised, words tagged with sgm_epoch)
  unsigned sgm_epoch = 0;
  int sgm_sweep = 0;     // VWGPU_OPT_SGM_SWEEP
  int mgm_sweep = 0;     // VWGPU_OPT_MGM_SWEEP
  int sgm_path_mode = 0; // VWGPU_OPT_SGM_PATH_MODE
  vwgpu_arena xvol;      // exact-order path: column-sum volumes, band state, per-zone NCC precision images (bm_exact.hip)
  vwgpu_arena xtab;      // exact-order path: zone / work-item tables of one call
  vwgpu_arena xcarry;    // exact-order path: compare-chain state per pixel between the disparity groups of a zone with > 512 disparities
  // Pinned host memory for the small tables that cross PCIe inside a call (zone / tile tables up, leaf extents down): a copy from or
  // to pageable memory is staged by the runtime and blocks the calling thread.  A ring of two halves: vwgpu_host_ring() hands out the
  // next piece and waits, before it re-enters a half, for the event that marks the last copy queued from that half.
  char* host_ring = nullptr;
  size_t host_cap = 0, host_pos = 0;
  int host_half = 0;
  hipEvent_t ring_event[2] = {nullptr, nullptr};
  bool ring_event_set[2] = {false, false};
  int host_ring_kb = 16384;   // VWGPU_OPT_HOST_RING_KB
  unsigned long long ring_wraps = 0;   // half crossings so far (VWGPU_OPT_HOST_RING_WRAPS, read only: tests)
  struct LeafRects { int w, h; size_t n; void* d_rects; };
  std::vector<LeafRects> leaf_rects;   // zone scheduler: device copies of the leaf boxes of the level sizes seen so far
  bool measure_first = false; // the previous calc_disparity was refused by the packed-u8 kernels: measure the input class first
  bool defer_exact = false;   // VWGPU_OPT_DEFER_EXACTNESS: calc_disparity_dev never waits for the input-class flags
  int sad_groups = 0;         // VWGPU_OPT_SAD_GROUPS
  int exact_scratch_mb = 4096;// VWGPU_OPT_EXACT_SCRATCH_MB
  int exact_split = 0;        // VWGPU_OPT_EXACT_SPLIT: 0 by the longest chain, 1 always the split pass 2, 2 always the fused one
  int trace = 0;              // VWGPU_OPT_TRACE
  int certify = 1;            // VWGPU_OPT_CERTIFY
  int zone_tile16 = 2;        // VWGPU_OPT_ZONE_TILE16: zones that fit 16 x 16 on the one-wavefront tile kernels: 0 = in tile groups, 1 = always, 2 = never
  int cert_f32 = 1;           // VWGPU_OPT_CERT_F32: the certified pass runs its fp32 tier first (bm_zones.hip)
  int zone_sxc = 0;           // VWGPU_OPT_ZONE_SXC: 0 = 16 dx per right patch, else at most this many
  unsigned long long cert_px[3] = {0, 0, 0};   // with VWGPU_OPT_TRACE bit 2: pixels in certified tiles / in flagged tiles / in tiles the fp32 tier passed on to float64, so far
  int num_cu = 256;
  void* host_pool = nullptr;  // helper threads for the per-tile host work of tile groups (pyramid.hip), created on first use
};
// fn(i) for i in [0, n) on the context's helper threads and the calling thread; returns when all are done.  fn must not throw.
void vwgpu_pool_run(vwgpu_ctx* ctx, int n, void (*fn)(void*, int), void* arg);
void vwgpu_pool_destroy(vwgpu_ctx* ctx);

int vwgpu_fail(vwgpu_ctx* ctx, int status, const char* fmt, ...);
int vwgpu_arena_reserve(vwgpu_ctx* ctx, vwgpu_arena* a, size_t bytes);
void* vwgpu_host_ring(vwgpu_ctx* ctx, size_t bytes);     // nullptr: the request does not fit (the caller copies from its own memory)

#define VWGPU_HIP(ctx, call)                                                                     \
  do {                                                                                           \
    hipError_t e_ = (call);                                                                      \
    if (e_ != hipSuccess)                                                                        \
      return vwgpu_fail((ctx), e_ == hipErrorOutOfMemory ? VWGPU_ERR_NOMEM : VWGPU_ERR_HIP,      \
                        "%s failed: %s", #call, hipGetErrorString(e_));                          \
  } while (0)

// RAII event bracket around one kernel launch when profiling is on.
struct vwgpu_prof_scope {
  vwgpu_ctx* ctx;
  hipEvent_t b = nullptr;
  vwgpu_prof_scope(vwgpu_ctx* c, const char* name);
  ~vwgpu_prof_scope();
};

static inline size_t vwgpu_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- kernel-family launchers (each in its own .hip) ------------------------------------------------
// All pointers are device pointers; every launcher enqueues on ctx->stream and returns a vwgpu_status.

int vwgpu_launch_bm_generic(vwgpu_ctx* ctx, int cost_type,
                            const float* left, int lw, int lh, ptrdiff_t ls,
                            const float* right, int rw, int rh, ptrdiff_t rs,
                            int kx, int ky, int sx, int sy, int32_t* out, ptrdiff_t os);

// Same, but the kernel returns immediately when run_flag != NULL and *run_flag == 0 (device memory).
int vwgpu_launch_bm_generic_flag(vwgpu_ctx* ctx, int cost_type,
                                 const float* left, int lw, int lh, ptrdiff_t ls,
                                 const float* right, int rw, int rh, ptrdiff_t rs,
                                 int kx, int ky, int sx, int sy, int32_t* out, ptrdiff_t os,
                                 const int* run_flag);

// Returns VWGPU_ERR_NOIMPL (without touching `out`) if this configuration has no u8 fast path;
// otherwise enqueues pack + match.  *d_fallback_flag (device int) is nonzero afterwards if the inputs were
// not integer-valued in [0,255] — the caller must then run the generic path.
bool vwgpu_bm_sad_u8_supported(int cost_type, int kx, int ky, int sx, int sy);
int vwgpu_launch_bm_sad_u8(vwgpu_ctx* ctx,
                           const float* left, int lw, int lh, ptrdiff_t ls,
                           const float* right, int rw, int rh, ptrdiff_t rs,
                           int kx, int ky, int sx, int sy, int32_t* out, ptrdiff_t os,
                           int** d_fallback_flag);

// bm_dot_u8.hip: SSD / NCC on integer-valued [0,255] inputs (v_dot4_u32_u8); same fallback-flag protocol as the SAD path
bool vwgpu_bm_dot_u8_supported(int cost_type, int kx, int ky, int sx, int sy);
int vwgpu_launch_bm_dot_u8(vwgpu_ctx* ctx, int cost_type, const float* left, int lw, int lh, ptrdiff_t ls,
                           const float* right, int rw, int rh, ptrdiff_t rs, int kx, int ky, int sx, int sy,
                           int32_t* out, ptrdiff_t os, int** d_fallback_flag);
int vwgpu_next_flags(vwgpu_ctx* ctx, size_t extra_ints, int** flag_set, int** flag_clear, int** extra);
// bm_sad_u16.hip: SAD for integer-valued imagery in [0,65535] (v_sad_u16 on pixel pairs); same fallback-flag protocol
bool vwgpu_bm_sad_u16_supported(int cost_type, int kx, int ky, int sx, int sy);
int vwgpu_launch_bm_sad_u16(vwgpu_ctx* ctx, const float* left, int lw, int lh, ptrdiff_t ls,
                            const float* right, int rw, int rh, ptrdiff_t rs, int kx, int ky, int sx, int sy,
                            int32_t* out, ptrdiff_t os, int** d_fallback_flag);
// bm_corr_u8.hip: register-blocked SSD / NCC for the same inputs (preferred over bm_dot_u8 where instantiated)
bool vwgpu_bm_corr_u8_supported(int cost_type, int kx, int ky, int sx, int sy);
int vwgpu_launch_bm_corr_u8(vwgpu_ctx* ctx, int cost_type, const float* left, int lw, int lh, ptrdiff_t ls,
                            const float* right, int rw, int rh, ptrdiff_t rs, int kx, int ky, int sx, int sy,
                            int32_t* out, ptrdiff_t os, int** d_fallback_flag);
int vwgpu_launch_ncc_full(vwgpu_ctx* ctx, const float* left, ptrdiff_t ls, const float* right, ptrdiff_t rs, int kx, int ky, int sx,
                          const uint32_t* a2, const uint32_t* b2, int b2w, int32_t* out, ptrdiff_t os, int ow, int* flag,
                          const uint32_t* full_list, const uint32_t* full_count, uint32_t cap);
// bm_corr_u16.hip: SSD / NCC for integer-valued imagery in [0,4095] (v_dot2_u32_u16 on pixel pairs); same fallback-flag protocol
bool vwgpu_bm_corr_u16_supported(int cost_type, int kx, int ky, int sx, int sy);
int vwgpu_launch_bm_corr_u16(vwgpu_ctx* ctx, int cost_type, const float* left, int lw, int lh, ptrdiff_t ls,
                             const float* right, int rw, int rh, ptrdiff_t rs, int kx, int ky, int sx, int sy,
                             int32_t* out, ptrdiff_t os, int** d_fallback_flag);

// bm_exact.hip: the reference's own summation order (serial column / row chains of fast_box_sum) for inputs whose
// partial sums are not exactly representable; see the file header.
struct vwgpu_zone_task;
bool vwgpu_bm_exact_supported(int sx, int sy);
bool vwgpu_sums_order_free(int cost_type, int kx, int ky, int lo, int hi, int nonfinite);
int vwgpu_sums_bits(int cost_type, int kx, int ky, int lo, int hi, int nonfinite);      // <= 53: any order in float64; <= 24: in float32 too
int vwgpu_float_grain(vwgpu_ctx* ctx, const float* a, int aw, int ah, ptrdiff_t as, const float* b, int bw, int bh, ptrdiff_t bs,
                      int* lo, int* hi, int* nonfinite);
void vwgpu_launch_float_grain(vwgpu_ctx* ctx, int n, const float* const* img, const int* w, const int* h, const ptrdiff_t* stride, int* const* d_cells);
int vwgpu_launch_bm_exact(vwgpu_ctx* ctx, int cost_type, const float* A, int aw, int ah, ptrdiff_t as,
                          const float* B, int bw, int bh, ptrdiff_t bs, int kx, int ky,
                          const vwgpu_zone_task* zones, int n, int32_t* out, const int* d_gate = nullptr,   // d_gate[i] == 0 (device): zone i is skipped
                          const int* rows = nullptr, int nranges = 0);   // one zone only: produce just the output rows [rows[2i], rows[2i+1]) (sorted, disjoint); the column chains still run from row 0
int vwgpu_launch_box_sum_exact(vwgpu_ctx* ctx, const float* img, int w, int h, ptrdiff_t stride, int kx, int ky, double* d_out);

int vwgpu_launch_lr_check(vwgpu_ctx* ctx, int32_t* l2r, int lw, int lh, ptrdiff_t ls,
                          const int32_t* r2l, int rw, int rh, ptrdiff_t rs, float thr);
int vwgpu_launch_lr_check_diff(vwgpu_ctx* ctx, int32_t* l2r, int lw, int lh, ptrdiff_t ls,
                               const int32_t* r2l, int rw, int rh, ptrdiff_t rs, float thr,
                               float* diff2, ptrdiff_t dstride, int ulx, int uly);

// filters.hip
// One launch serves up to VWGPU_MAX_IMG_JOBS images (blockIdx.z = job): the pyramid of a tile is ~60 filter launches of 10-20 us on small
// images, launch bound — the left and right image of a level, or all twelve level images of a prefilter, go through ONE launch.
constexpr int VWGPU_MAX_IMG_JOBS = 12;
struct vwgpu_img_job {
  const void* src; ptrdiff_t stride; int w, h;        // source image
  void* dst; ptrdiff_t dstride; int ow, oh;           // destination and output size
  int offx, offy;                                     // source position of output (0, 0)
  const float* b; ptrdiff_t bs;                       // second operand (subtract / edge_extend_sub), or nullptr
};
// a job of vwgpu_launch_sepconv_jobs with bs == VWGPU_JOB_MASK_BY_TWO is a uint8 mask halved by subsample_mask_by_two instead of an image
constexpr ptrdiff_t VWGPU_JOB_MASK_BY_TWO = -7;
int vwgpu_launch_sepconv_jobs(vwgpu_ctx* ctx, const vwgpu_img_job* jobs, int n, const float* xk, int nx, int cx, const float* yk, int ny, int cy,
                              int edge, int step);
int vwgpu_launch_conv2d_jobs(vwgpu_ctx* ctx, const vwgpu_img_job* jobs, int n, const float* k, int kw, int kh, int ci, int cj, int edge);
int vwgpu_launch_mask_by_two_jobs(vwgpu_ctx* ctx, const vwgpu_img_job* jobs, int n);
int vwgpu_launch_subtract_jobs(vwgpu_ctx* ctx, const vwgpu_img_job* jobs, int n);
int vwgpu_prefilter_images_dev(vwgpu_ctx* ctx, int n, const float* const* srcs, const int* ws, const int* hs, int mode, float width, float* const* dsts);
int vwgpu_launch_sepconv(vwgpu_ctx* ctx, const float* src, int w, int h, ptrdiff_t stride,
                         const float* xk, int nx, int cx, const float* yk, int ny, int cy,
                         int edge, int step, float* dst, ptrdiff_t dstride);
int vwgpu_launch_conv2d(vwgpu_ctx* ctx, const float* src, int w, int h, ptrdiff_t stride,
                        const float* k, int kw, int kh, int ci, int cj, int edge, float* dst, ptrdiff_t dstride);
int vwgpu_launch_mask_by_two(vwgpu_ctx* ctx, const uint8_t* src, int w, int h, ptrdiff_t stride,
                             uint8_t* dst, ptrdiff_t dstride);
int vwgpu_launch_subtract(vwgpu_ctx* ctx, const float* a, ptrdiff_t as, const float* b, ptrdiff_t bs, int w, int h,
                          float* dst, ptrdiff_t ds);
int vwgpu_launch_sepconv_region(vwgpu_ctx* ctx, const float* src, int w, int h, ptrdiff_t stride,
                                const float* xk, int nx, int cx, const float* yk, int ny, int cy,
                                int edge, int step, float* dst, ptrdiff_t dstride, int ow, int oh, int offx, int offy);
int vwgpu_launch_conv2d_region(vwgpu_ctx* ctx, const float* src, int w, int h, ptrdiff_t stride,
                               const float* k, int kw, int kh, int ci, int cj, int edge, float* dst, ptrdiff_t dstride,
                               int ow, int oh, int offx, int offy);
int vwgpu_launch_ext_sub(vwgpu_ctx* ctx, const float* a, ptrdiff_t as, int w, int h, const float* b, ptrdiff_t bs,
                         float* dst, ptrdiff_t ds, int ow, int oh, int offx, int offy);
int vwgpu_prefilter_region(vwgpu_ctx* ctx, const float* src, int w, int h, ptrdiff_t stride, int mode, float width,
                           int x0, int y0, int bw, int bh, float* dst, float* scratch);
extern "C" int vwgpu_generate_gaussian_kernel(double sigma, int size, float* taps, int cap);

// subpixel.hip
int vwgpu_launch_parabola_prepass(vwgpu_ctx* ctx, const float* disp3f, int w, int h, ptrdiff_t stride_px, int* d_out4,
                                  const float* L, int lw, int lh, ptrdiff_t ls, const float* R, int rw, int rh, ptrdiff_t rs, int* d_cell);
void vwgpu_launch_f32_ext_to_u8_raster(vwgpu_ctx* ctx, const float* src, ptrdiff_t stride, int w, int h, int x0, int y0, int bw, int bh,
                                       uint8_t* dst, int pitch);
int vwgpu_parabola_u8_pitch(int w);
int vwgpu_launch_parabola(vwgpu_ctx* ctx, const float* disp3f, int w, int h, ptrdiff_t dstride_px,
                          const float* lras, int lrw, const float* rras, int rrw, int range_minx, int range_miny,
                          int kx, int ky, float* out3f, ptrdiff_t ostride_px, int integer_class = 0);

// bm_zones.hip — one row per search zone (or per R->L zone image); see the kernels for the field meaning
struct vwgpu_zone_task {
  int ax, ay;          // origin of the zone's input crop in image A (may lie outside: coordinates are clamped)
  int bx, by;          // origin in image B of the window for disparity (0,0)
  int zw, zh;          // output size
  int sx, sy;          // search volume
  int out_off;         // pixel offset of the zone's first output pixel in the output buffer
  int out_stride;      // pixels
  int addx, addy;      // added to the winning disparity index of every pixel
  int img = 0;         // zone-matcher groups (vwgpu_zone_group): which image pair of the group the zone belongs to
};
// A launch sequence of the zone matcher over several image pairs of EQUAL geometry laid out at a fixed stride (the tiles of a
// pyramid_correlate group, pyramid.hip): image pair i = (A + i * a_stride, B + i * b_stride); zones name theirs in vwgpu_zone_task::img.
struct vwgpu_zone_group {
  int n_img = 1;
  size_t a_stride = 0, b_stride = 0;       // floats
  const int* cert_hi = nullptr;            // n_img largest exponents (certified passes; overrides the cert_hi argument)
  const int* edge_lo = nullptr;            // n_img bounds of the "cannot matter" certificate (override edge_lo / edge_hi)
  const int* edge_hi = nullptr;
};
bool vwgpu_bm_zones_supported(int kx, int ky);
// f32_sums: vwgpu_sums_bits <= 24 for BOTH images.  cert_hi != INT_MIN: certify against the reference's summation order (bm_zones.hip),
// d_zflag[n] (zeroed by the caller) receives the zones that need vwgpu_launch_bm_exact; d_stats (optional): {pixels in certified tiles, in flagged tiles}
int vwgpu_launch_bm_zones(vwgpu_ctx* ctx, int cost_type, const float* A, int aw, int ah, const float* B, int bw, int bh,
                          int kx, int ky, const vwgpu_zone_task* zones, int n, int32_t* out, int f32_sums = 0,
                          int cert_hi = (-2147483647 - 1), int* d_zflag = nullptr, unsigned long long* d_stats = nullptr,
                          int* d_any = nullptr,       // d_any (optional, zeroed by the caller): set to 1 when some zone was flagged
                          const int* d_need = nullptr, const unsigned char* d_cells = nullptr,      // (optional) what to match of every zone, from vwgpu_launch_zone_need
                          int edge_m = 0, int edge_k = 0, int edge_lo = 0, int edge_hi = 0,      // certified passes: edge_m > 0 turns the "cannot matter" certificate on (bm_zones.hip, ZEdge)
                          ptrdiff_t as = 0, ptrdiff_t bs = 0,                                      // row strides of A / B in floats (0: the widths)
                          const vwgpu_zone_group* grp = nullptr,                                   // several image pairs in one launch sequence; d_any then has n_img words
                          int* d_tflag = nullptr);                                                 // certified passes without d_need: one int per 32-row band of every zone (vwgpu_zone_row_flags of them, zeroed by the caller; zone i's begin where the bands of zones 0 .. i-1 end), set for bands with an unproven pixel
inline size_t vwgpu_zone_row_flags(const vwgpu_zone_task* zones, int n) {
  size_t r = 0;
  for (int i = 0; i < n; ++i) r += (size_t)(zones[i].zh + 31) / 32;
  return r;
}
// The flagged bands of one zone as row ranges {begin, end}, ... for vwgpu_launch_bm_exact; returns the number of flagged rows.
inline int vwgpu_zone_flagged_rows(const int* bands, int zh, std::vector<int>* rows) {
  int flagged = 0;
  rows->clear();
  for (int ty = 0; ty * 32 < zh; ++ty) {
    if (!bands[ty]) continue;
    const int a = ty * 32, b = std::min(zh, a + 32);
    flagged += b - a;
    if (!rows->empty() && rows->back() == a) rows->back() = b; else { rows->push_back(a); rows->push_back(b); }
  }
  return flagged;
}
// The part of every zone's R->L image that its L/R check (vwgpu_launch_zone_lr, same tasks) will read, from the finished L->R result:
// a rectangle per zone (d_need, 8 ints per zone) and a flag per 16 x 16 cell (d_cells; vwgpu_zone_need_cells numbers the cells into the
// tasks' `ay` slot and returns their count).  d_zflag (optional): L->R zones still to be matched again ask for the whole image.
size_t vwgpu_zone_need_cells(vwgpu_zone_task* zones, int n);
int vwgpu_launch_zone_need(vwgpu_ctx* ctx, const vwgpu_zone_task* zones, int n, const int32_t* l2r, const int* d_zflag, int* d_need,
                           unsigned char* d_cells, size_t ncells);
// lr tasks: ax = pixel offset of the zone's R->L image, (bx, by) = its size, (sx, sy) = the zone's origin in the diff image
int vwgpu_launch_zone_lr(vwgpu_ctx* ctx, const vwgpu_zone_task* zones, int n, int32_t* l2r, const int32_t* r2l, float thr,
                         float* diff2, ptrdiff_t dstride);

// pyramid.hip
int vwgpu_launch_disparity_filter(vwgpu_ctx* ctx, const int32_t* src, int w, int h, int hh, int hv, double pthr, double rthr,
                                  bool cleanup, int32_t* tmp_padded, int32_t* dst, bool inner_only = false,      // inner_only: stop after the first pass (tmp_padded)
                                  int tiles = 1, size_t tile_ints = 0);                                        // tile groups: `tiles` images at a stride of tile_ints ints
int vwgpu_launch_blob_filter(vwgpu_ctx* ctx, int32_t* d, int w, int h, int area, int* scratch);
int vwgpu_launch_disparity_mask(vwgpu_ctx* ctx, int32_t* d, int w, int h, const uint8_t* m1, const uint8_t* m2, int m2w, int m2h);
int vwgpu_pyramid_correlate_impl(vwgpu_ctx* ctx, const float* left, int lw, int lh, ptrdiff_t ls,
                                 const float* right, int rw, int rh, ptrdiff_t rs,
                                 const uint8_t* lmask, ptrdiff_t lms, const uint8_t* rmask, ptrdiff_t rms,
                                 const vwgpu_pyramid_params* P, int bx, int by, int bw, int bh,
                                 float* out, ptrdiff_t os, float* lr_diff);

// tile groups (pyramid.hip): several equal-sized tiles of one image pair through the level loop together
bool vwgpu_pyramid_group_eligible(const vwgpu_ctx* ctx, const vwgpu_pyramid_params* P, int n, const int* bw, const int* bh);
int vwgpu_pyramid_group_impl(vwgpu_ctx* ctx, const float* left, int lw, int lh, ptrdiff_t ls, const float* right, int rw, int rh, ptrdiff_t rs,
                             const uint8_t* lmask, ptrdiff_t lms, const uint8_t* rmask, ptrdiff_t rms, const vwgpu_pyramid_params* P,
                             int n, const int* bxs, const int* bys, int bw, int bh, float* const* outs, const ptrdiff_t* oss);

// sgm.hip
int vwgpu_sgm_impl(vwgpu_ctx* ctx, const vwgpu_sgm_params* P, const float* left, int lw, int lh, ptrdiff_t ls,
                   const float* right, int rw, int rh, ptrdiff_t rs, int sx, int sy,
                   const uint8_t* lmask, int lmw, int lmh, const uint8_t* rmask, int rmw, int rmh,
                   const int32_t* prev, int pw, int ph, int32_t* out_disp, float* out_sub, size_t out_cap_pixels, int* ow, int* oh);
