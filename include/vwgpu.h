/* vwgpu.h — C ABI of the MI355X-native stereo-correlation engine (libvwgpu.so).
 *
 * This is the drop-in boundary for Vision Workbench's dense block-matching hot path.  The reference has no
 * FFI/plugin interface (the path is ordinary C++ linked into libVwStereo.so), so each entry
 * point below is what the body of the cited reference function would call once its inputs are rasterised.
 * The C++ surface that keeps the reference's own signatures (vw::stereo::calc_disparity, ...) lives in
 * visionworkbench_amd/vwlite/ and is a thin wrapper over these calls; INTEGRATION.md shows the binding a
 * Vision Workbench maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types.  Images are row-major, `stride` in ELEMENTS per row.
 *   - `*_dev` entry points take DEVICE pointers and are asynchronous on the context's stream;
 *     the un-suffixed ones take HOST pointers, stage through HBM and return when the result is in host memory.
 *   - every call returns VWGPU_OK (0) or a negative vwgpu_status; nothing throws.  One context per
 *     (host thread x GPU), mirroring the reference's one-thread-per-tile model
 *     (src/vw/Image/ImageIO.h:228-251); a context is not thread-safe, different contexts are independent.
 *   - disparity images use the vw::PixelMask<Vector2i> memory layout: 3 x int32 per pixel {dx, dy, valid},
 *     valid = INT32_MAX or 0 (src/vw/Image/PixelMask.h:48-120, src/vw/Image/PixelTypeInfo.h:95-102);
 *     float disparities use vw::PixelMask<Vector2f>: 3 x float {dx, dy, valid in {0.f, 1.f}}.
 */
#ifndef VWGPU_H
#define VWGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): struct vwgpu_sgm_params carries allow_block_cost (appended in round 3 without a bump: a host compiled against
 *     version 1 passes a struct that is 8 bytes shorter), VWGPU_PATH_REFUSED replaces the silent float64 fallback of
 *     VWGPU_OPT_DEFER_EXACTNESS, vwgpu_trim, vwgpu_halo_headers_agree, the host-ring and certification options; options 7
 *     (VWGPU_OPT_EXACT_LDS) and 10 (VWGPU_OPT_CORR_MFMA) are gone with the two slower kernel variants they selected.
 * 3 (round 5): vwgpu_pyramid_correlate_batch[_dev] (tile groups); vwgpu_last_path() may answer VWGPU_PATH_CERTIFIED (single-level calls on
 *     float rasters proven equal to the reference's summation order); options VWGPU_OPT_CERT_F32, _CERT_F64_PERMILLE, _ZONE_TILE16.  No struct
 *     of version 2 changed, but observable behaviour did (VWGPU_PATH_CERTIFIED where a version-2 host saw VWGPU_PATH_EXACT_ORDER), and hosts
 *     compare versions for EQUALITY: a host built against version 2 refuses this library and is rebuilt against this header.
 *     Round 6 added option values only (VWGPU_OPT_SGM_PATH_MODE, VWGPU_OPT_SAD_GROUPS = 3): same version.
 * A host checks vwgpu_abi_version() == VWGPU_ABI_VERSION once after loading the library (vw::engine does, vw/Engine.h). */
#define VWGPU_ABI_VERSION 3

typedef struct vwgpu_ctx vwgpu_ctx;

/* Status codes; the C++ wrappers map them to the reference's exception types
 * (src/vw/Core/Exception.h:225-253): ARGUMENT -> vw::ArgumentErr, NOIMPL -> vw::NoImplErr,
 * LOGIC/HIP/NOMEM -> vw::LogicErr. */
typedef enum vwgpu_status {
  VWGPU_OK = 0,
  VWGPU_ERR_ARGUMENT = -1,
  VWGPU_ERR_NOIMPL = -2,
  VWGPU_ERR_HIP = -3,
  VWGPU_ERR_NOMEM = -4,
  VWGPU_ERR_LOGIC = -5
} vwgpu_status;

/* vw::stereo::CostFunctionType, src/vw/Stereo/CostFunctions.h:143-149 (same numeric values). */
typedef enum vwgpu_cost_type {
  VWGPU_ABSOLUTE_DIFFERENCE = 0,
  VWGPU_SQUARED_DIFFERENCE = 1,
  VWGPU_CROSS_CORRELATION = 2,
  VWGPU_CENSUS_TRANSFORM = 3,          /* SGM only (src/vw/Stereo/SGM.cc:1874-1893) */
  VWGPU_TERNARY_CENSUS_TRANSFORM = 4   /* SGM only */
} vwgpu_cost_type;

/* Which kernel family served the last calc_disparity call (vwgpu_last_path). */
typedef enum vwgpu_path {
  VWGPU_PATH_NONE = 0,
  VWGPU_PATH_GENERIC_F64 = 1,  /* any float input, float64 accumulators (reference arithmetic)          */
  VWGPU_PATH_SAD_U8 = 2,       /* integer-valued inputs in [0,255]: packed u8 SAD (v_qsad_pk_u16_u8)      */
  VWGPU_PATH_DOT_U8 = 3,       /* integer-valued inputs in [0,255]: SSD / NCC on v_dot4_u32_u8            */
  VWGPU_PATH_EXACT_ORDER = 4,  /* inputs whose box sums round: the reference's serial summation order     */
  VWGPU_PATH_SAD_U16 = 5,      /* integer-valued inputs in [0,65535]: SAD on v_sad_u16 pixel pairs        */
  VWGPU_PATH_DOT_U16 = 6,      /* integer-valued inputs in [0,4095]: SSD / NCC on v_dot2_u32_u16          */
  VWGPU_PATH_REFUSED = 7,      /* vwgpu_last_path only: a packed kernel queued WITHOUT waiting for its verdict
                                  (vwgpu_force_path, VWGPU_OPT_DEFER_EXACTNESS) met input outside its domain; that
                                  call produced NO result — repeat it with automatic dispatch                   */
  VWGPU_PATH_CERTIFIED = 8     /* vwgpu_last_path only: inputs whose box sums round, matched by the tile-parallel
                                  kernels with every pixel PROVEN equal to the reference's serial summation order
                                  (winner ahead of the runner-up by more than twice a bound on the difference of the
                                  two orders); a call with a pixel that cannot be proven is redone in the reference's
                                  order and reports VWGPU_PATH_EXACT_ORDER.  VWGPU_OPT_CERTIFY = 0 turns it off      */
} vwgpu_path;

/* ---- context ------------------------------------------------------------------------------------- */

int vwgpu_abi_version(void);

/* Creates a context on HIP device `device` with its own stream. Fails with VWGPU_ERR_HIP if no GPU. */
int vwgpu_create(vwgpu_ctx** ctx, int device);
void vwgpu_destroy(vwgpu_ctx* ctx);

/* Enqueue on an externally owned hipStream_t (passed as void*).  NULL means the legacy default stream
 * (that is what torch's default stream is), NOT the context's own stream — see vwgpu_reset_stream. */
int vwgpu_set_stream(vwgpu_ctx* ctx, void* hip_stream);
/* Back to the context's own (non-blocking) stream. */
int vwgpu_reset_stream(vwgpu_ctx* ctx);
/* Blocks until everything queued on the context's stream is done (and releases the scratch blocks the context has outgrown). */
int vwgpu_synchronize(vwgpu_ctx* ctx);
/* Memory policy: a context keeps one grow-only scratch arena per purpose (pyramids, SGM volumes, exact-order column sums, ...), sized by
 * the largest call it has served — steady-state calls do no hipMalloc / hipFree, which would stall every stream of the device.  A long-lived
 * tile thread therefore holds the high-water mark of every arena until vwgpu_destroy.  vwgpu_trim gives the memory back in between:
 * it waits for the context's stream, then frees every arena (and the outgrown blocks); the next call allocates what it needs again.
 * Call it when a thread changes workload (e.g. after the one config-4-sized SGM strip of a session) or before another context needs
 * the memory; *freed_bytes (optional) receives the amount released. */
int vwgpu_trim(vwgpu_ctx* ctx, size_t* freed_bytes);

const char* vwgpu_strerror(int status);
/* Detail text of the last failing call on this context ("" if none). */
const char* vwgpu_last_error(const vwgpu_ctx* ctx);

/* Per-context options.
 *   VWGPU_OPT_DEFER_EXACTNESS  calc_disparity_dev picks its kernel family from the DATA: integer-valued pixels in [0,255] take
 *       the packed kernels, other inputs the float64 kernel when every box-sum partial is exactly representable (then any
 *       summation order returns the reference's bits), and the reference's own serial summation order otherwise
 *       (VWGPU_PATH_EXACT_ORDER).  The classes are measured on the device; by default (0) the call waits for them, so that
 *       the result is bit-exact for any input.  1 = pipelined callers that queue many calls on byte imagery without a host
 *       round trip: where a packed-u8 kernel exists for the configuration the call is that ONE launch and never waits.  The
 *       option cannot change a result, it can only withhold one: input outside the kernel's domain raises a device flag,
 *       nothing else is computed, vwgpu_last_path() answers VWGPU_PATH_REFUSED for that call (its output buffer is
 *       undefined) and the caller repeats it with the option off.  bench.py asserts VWGPU_PATH_SAD_U8 after its timed steps.
 *       Configurations without a packed-u8 kernel behave as with 0.
 *   VWGPU_OPT_DEVICE_COUNT (read only): HIP devices visible to the process.
 *   The remaining options never change a result (they choose between kernels / schedules that return identical bits, and are
 *   what tests and tools use to reach every variant); values outside the stated range are rejected with VWGPU_ERR_ARGUMENT:
 *   VWGPU_OPT_SAD_GROUPS       packed-u8 SAD matcher flavour: 0 = chosen by the launcher's cost model (default), 1 = one wave
 *       group per tile, 2 = two wave groups per tile, 3 = four wave groups on 512-column tiles of 16 rows (sizes without that flavour
 *       keep the launcher's choice).  Same results in every flavour.
 *   VWGPU_OPT_EXACT_SCRATCH_MB scratch budget of the exact-order path in MiB (16 .. 65536, default 4096): column-sum volumes
 *       beyond it are swept in row bands / zone groups / disparity groups.
 *   VWGPU_OPT_TRACE            bit 0: host-side timeline of a pyramid tile on stderr, bit 1: the zone shapes of a level, bit 2: certification
 *       statistics of every tile (one host round trip per tile; VWGPU_OPT_CERT_PERMILLE reads the running total).
 *   VWGPU_OPT_CERTIFY          pyramid levels whose box sums could round (prefiltered imagery, float imagery, deep levels under SSD / NCC): 1
 *       (default) = the tile-parallel kernels match every zone first and CERTIFY each pixel — best cost ahead of the runner-up by more
 *       than twice a rigorous bound on the difference between any-order float64 sums and fast_box_sum's serial running sums
 *       (src/vw/Stereo/Algorithms.h:43-129) — and only zones that hold an uncertified pixel are redone by the exact-order kernels.
 *       A second certificate covers candidates whose partner lies far outside the other image (tiles at the side of a pair: NaN or exactly
 *       tied costs in the margin that no bound can order): when only such candidates can win, the pixel is erased by the mask pass (L->R)
 *       or fails the consistency check (R->L) whichever of them the reference picks, so the tile is the reference's without knowing which.
 *       0 = every zone of such a level goes to the exact-order kernels (the round-3 schedule).  Same results either way.
 *   VWGPU_OPT_ZONE_SXC         horizontal disparities per staged right patch: 0 (default) = 16, n = at most n (1 .. 4096; the LDS budget caps it).
 *   VWGPU_OPT_CERT_PERMILLE    (read only) per mille of the pixels in certified tiles since VWGPU_OPT_TRACE was last set with bit 2; -1 = none.
 *   VWGPU_OPT_CERT_F32         1 (default): the certified pass is two tiers in one launch — float32 window sums and compare chain first,
 *       proven against the reference's order AND their own float32 roundings; a 32 x 32 tile with a pixel that tier cannot prove runs again
 *       in float64, and only what float64 cannot prove goes to the exact-order kernels.  0: float64 only (the round-4 schedule).  Same results.
 *   VWGPU_OPT_ZONE_TILE16      zone matcher tiles for 16 x 16 leaf zones: 0 (default) = 32 x 32 workgroup tiles for every zone, 1 = one-wavefront
 *       16 x 16 tiles for zones of at most 16 x 16 pixels, 2 = those tiles for every zone (measurements).  Same results.
 *   VWGPU_OPT_CERT_F64_PERMILLE (read only) per mille of the counted pixels that lay in tiles the fp32 tier passed on to float64; -1 = none.
 *   VWGPU_OPT_SGM_SWEEP        SGM path aggregation of full-range one-row searches (<= 256 disparities): 0 = one direction per launch
 *       (default: eight passes over the u16 sums, bandwidth bound), 1 = two concurrent fused raster sweeps of four directions each
 *       (a fifth of the HBM traffic, the same sums; slower on MI355X because a sweep is W + 2 H dependent pixel steps), 2 .. 15 = the
 *       sweeps with that many rows per workgroup (tuning).
 *   VWGPU_OPT_SGM_PATH_MODE    the one-direction-per-launch schedule of full-range one-row searches (tuning / measurements; the same sums in every
 *       mode): bits 0-3 = scan lines per workgroup (0 = default: 4 neighbouring lines; 8 = eight lines with a ring of three chunks; the register
 *       kernel: < 4 = one line), bit 5 = the round-2 kernel that prefetches into registers (path_uniform_reg_kernel) instead of the LDS ring
 *       (path_ring_kernel, vector strides up to 160 bytes), bits 8-10 = log2 of the consecutive workgroups given to one XCD, bit 11 = the ring
 *       kernel forms the census costs itself from the census rasters (up to 129 disparities; no u8 cost volume; measured slower).
 *       Ragged boxes (a previous level's disparities given): bit 4 / bit 7 = four / two scan lines per wavefront whatever the level's boxes
 *       (path_multi_kernel; default: chosen per level from the share of small boxes), bit 6 = always one line per wavefront.
 *   VWGPU_OPT_EXACT_SPLIT      pass 2 of the exact-order matchers: 0 = chosen by the width of a zone (the recurrence alone + a parallel
 *       selection for zones of 1024 pixels and more (whole rasters), the tiled form — row sums transposed through LDS — for narrower
 *       ones), 1 = always the split form, 2 = always the fused form (selection across the disparity lanes inside the chain), 3 = always
 *       the tiled form.  Same results in every form.
 *   VWGPU_OPT_MGM_SWEEP        use_mgm on full-range one-row searches (<= 256 disparities): 0 = the eight passes as four concurrent
 *       sweeps (default), 1 = one launch per front (the round-2 schedule), 2 .. 15 = the sweeps with that many lines per workgroup.
 *   VWGPU_OPT_HOST_RING_KB     size in KiB (16 .. 1048576, default 16384) of the pinned host ring through which zone / work-item tables
 *       reach the device; pieces larger than a quarter of it are copied from pageable memory and waited for.  A small ring makes the
 *       library wait for the device more often, nothing else (tests use it to wrap the ring on small rasters).
 *   VWGPU_OPT_HOST_RING_WRAPS  (read only) how often the ring cursor has changed halves so far. */
typedef enum vwgpu_option {
  VWGPU_OPT_DEFER_EXACTNESS = 1, VWGPU_OPT_DEVICE_COUNT = 2, VWGPU_OPT_SAD_GROUPS = 3, VWGPU_OPT_EXACT_SCRATCH_MB = 4,
  VWGPU_OPT_TRACE = 5, VWGPU_OPT_SGM_SWEEP = 6, /* 7: removed in ABI 2 */ VWGPU_OPT_MGM_SWEEP = 8, VWGPU_OPT_EXACT_SPLIT = 9,
  /* 10: removed in ABI 2 */ VWGPU_OPT_HOST_RING_KB = 11, VWGPU_OPT_HOST_RING_WRAPS = 12, VWGPU_OPT_CERTIFY = 13,
  VWGPU_OPT_CERT_PERMILLE = 14, VWGPU_OPT_ZONE_SXC = 15, VWGPU_OPT_CERT_F32 = 16, VWGPU_OPT_CERT_F64_PERMILLE = 17, VWGPU_OPT_ZONE_TILE16 = 18,
  VWGPU_OPT_SGM_PATH_MODE = 19
} vwgpu_option;
int vwgpu_set_option(vwgpu_ctx* ctx, int option, int value);
int vwgpu_get_option(const vwgpu_ctx* ctx, int option, int* value);

/* Forces a kernel family (testing / benchmarking): VWGPU_PATH_NONE = automatic dispatch (default). */
int vwgpu_force_path(vwgpu_ctx* ctx, int path);
int vwgpu_last_path(const vwgpu_ctx* ctx);

/* Per-kernel timing with HIP events on the context's stream (bench.py roofline leg).
 * enable=1 records an event pair around every kernel launch of subsequent calls; vwgpu_profile_read
 * synchronises and returns up to `cap` (name, milliseconds) records since the last reset. */
int vwgpu_profile_enable(vwgpu_ctx* ctx, int enable);
int vwgpu_profile_reset(vwgpu_ctx* ctx);
int vwgpu_profile_read(vwgpu_ctx* ctx, const char** names, float* ms, int cap);

/* ---- block matching: calc_disparity -------------------------------------------------------------- */

/* Replaces vw::stereo::calc_disparity (decl src/vw/Stereo/Correlation.h:50-57, impl
 * src/vw/Stereo/Correlation.cc:330-375) from the point where it has rasterised its two crops (:356-359)
 * and calls best_of_search_convolution (:33-137) with fast_box_sum (src/vw/Stereo/Algorithms.h:43-129)
 * and the cost functors of src/vw/Stereo/CostFunctions.h:153-236.
 *
 *   left   lw x lh  : crop(left_in, left_region)
 *   right  rw x rh  : crop(right_in, left_region grown by search_volume-1 on the max side);
 *                     rw >= lw+sx-1 and rh >= lh+sy-1 (larger is allowed, the excess is ignored)
 *   kx,ky  kernel_size (odd; <= lw, lh),  sx,sy search_volume (>= 1)
 *   out    (lw-kx+1) x (lh-ky+1) pixels x {dx,dy,valid}; ostride in PIXELS (0 = dense).
 *          dx in [0,sx), dy in [0,sy): offsets into the right crop, exactly as the reference returns them.
 *
 * Semantics kept from the reference: raster order dy-outer/dx-inner, strict comparison, first wins;
 * a pixel is invalid iff its best and worst cost are equal (so search_volume (1,1) is all-invalid).
 * Bit-exact against the reference for any float input and any search volume: inputs whose box sums could round take the
 * kernels that follow fast_box_sum's serial summation order (VWGPU_PATH_EXACT_ORDER; more than 512 disparities are swept in
 * groups of 512 with the compare-chain state carried from group to group).  See VWGPU_OPT_DEFER_EXACTNESS for the
 * asynchronous variant (which either returns the same bits or reports that it returned none). */
int vwgpu_calc_disparity_dev(vwgpu_ctx* ctx, int cost_type,
                             const float* d_left, int lw, int lh, ptrdiff_t lstride,
                             const float* d_right, int rw, int rh, ptrdiff_t rstride,
                             int kx, int ky, int sx, int sy,
                             int32_t* d_out, ptrdiff_t ostride);

int vwgpu_calc_disparity(vwgpu_ctx* ctx, int cost_type,
                         const float* left, int lw, int lh, ptrdiff_t lstride,
                         const float* right, int rw, int rh, ptrdiff_t rstride,
                         int kx, int ky, int sx, int sy,
                         int32_t* out, ptrdiff_t ostride);

/* ---- box sums ------------------------------------------------------------------------------------------ */

/* Replaces vw::stereo::fast_box_sum<double>(image, kernel) (src/vw/Stereo/Algorithms.h:41-129): out(x, y) = sum of the
 * kx x ky window whose top-left pixel is (x, y), float64, (w-kx+1) x (h-ky+1) pixels, ostride in ELEMENTS (0 = dense).
 * The sums are formed in the reference's order (running column sums down the rows, running row sums along the columns:
 * :62-75, :81-110), so the result is bit-identical for ANY float input, not just exactly summable ones.
 * kx, ky must be odd (the reference's always-on VW_ASSERT, :45-46) and fit the image. */
int vwgpu_fast_box_sum_dev(vwgpu_ctx* ctx, const float* d_img, int w, int h, ptrdiff_t stride, int kx, int ky,
                           double* d_out, ptrdiff_t ostride);
int vwgpu_fast_box_sum(vwgpu_ctx* ctx, const float* img, int w, int h, ptrdiff_t stride, int kx, int ky,
                       double* out, ptrdiff_t ostride);

/* ---- left/right consistency check --------------------------------------------------------------- */

/* Replaces vw::stereo::cross_corr_consistency_check (src/vw/Stereo/Correlate.cc:1441-1502; decl
 * src/vw/Stereo/Correlate.h:52-58): invalidates l2r(c,r) when the r2l pixel at (c+dx, r+dy) is out of
 * bounds or invalid, or when max(|dx+dx'|, |dy+dy'|) > threshold.  In place on l2r. Strides in pixels. */
int vwgpu_cross_corr_consistency_check_dev(vwgpu_ctx* ctx,
                                           int32_t* d_l2r, int lw, int lh, ptrdiff_t lstride,
                                           const int32_t* d_r2l, int rw, int rh, ptrdiff_t rstride,
                                           float threshold);
int vwgpu_cross_corr_consistency_check(vwgpu_ctx* ctx,
                                       int32_t* l2r, int lw, int lh, ptrdiff_t lstride,
                                       const int32_t* r2l, int rw, int rh, ptrdiff_t rstride,
                                       float threshold);

/* The same with the reference's optional lr_disp_diff output (Correlate.cc:1441-1502): diff is a PixelMask<float> image
 * {value, valid} of dcols x drows (stride in pixels, 0 = dcols); every KEPT pixel (c, r) writes {max(|dx+dx'|, |dy+dy'|), 1}
 * at (c + ulx, r + uly), which must lie inside the image. */
int vwgpu_cross_corr_consistency_check_diff_dev(vwgpu_ctx* ctx, int32_t* d_l2r, int lw, int lh, ptrdiff_t lstride,
                                                const int32_t* d_r2l, int rw, int rh, ptrdiff_t rstride, float cross_corr_threshold,
                                                float* d_diff, int dcols, int drows, ptrdiff_t dstride, int ulx, int uly);
int vwgpu_cross_corr_consistency_check_diff(vwgpu_ctx* ctx, int32_t* l2r, int lw, int lh, ptrdiff_t lstride,
                                            const int32_t* r2l, int rw, int rh, ptrdiff_t rstride, float cross_corr_threshold,
                                            float* diff, int dcols, int drows, ptrdiff_t dstride, int ulx, int uly);

/* ---- image filters on the path: Gaussian pyramid and prefilters ------------------------------------- */

/* vw::ConstantEdgeExtension / vw::ZeroEdgeExtension (src/vw/Image/EdgeExtension.h). */
typedef enum vwgpu_edge { VWGPU_EDGE_CONSTANT = 0, VWGPU_EDGE_ZERO = 1 } vwgpu_edge;
/* vw::stereo::PrefilterModeType, src/vw/Stereo/PrefilterEnum.h:24-28 (same numeric values). */
typedef enum vwgpu_prefilter { VWGPU_PREFILTER_NONE = 0, VWGPU_PREFILTER_MEANSUB = 1, VWGPU_PREFILTER_LOG = 2 } vwgpu_prefilter;

/* Replaces vw::generate_gaussian_kernel<float> (src/vw/Image/Filter.tcc:37-78; default size
 * vw::compute_kernel_size, src/vw/Image/Filter.cc:32-37).  Host-side math (erf in double), no context needed.
 * size == 0 selects the default size.  Returns the number of taps written (0 for sigma == 0) or a negative
 * vwgpu_status if cap is too small. */
int vwgpu_generate_gaussian_kernel(double sigma, int size, float* taps, int cap);

/* Replaces rasterising vw::separable_convolution_filter(src, x_kernel, y_kernel, cx, cy, edge)
 * (src/vw/Image/Filter.h:156-191 -> SeparableConvolutionView::rasterize, src/vw/Image/Convolution.h:275-328) over the
 * whole image, optionally followed by vw::subsample(., subsample) (src/vw/Image/Manipulation.h:214-311) — the
 * pyramid-level operation of build_image_pyramids (src/vw/Stereo/CorrelationView.cc:210-214).
 *   nx / ny may be 0 (axis not filtered); cx, cy = kernel origins ((n-1)/2 is the reference's default).
 *   dst is (1+(w-1)/subsample) x (1+(h-1)/subsample).  x_kernel / y_kernel are HOST pointers in both variants.
 * Accumulation order and float arithmetic are the reference's; results are bit-identical for any float input. */
int vwgpu_separable_convolution_dev(vwgpu_ctx* ctx, const float* d_src, int w, int h, ptrdiff_t stride,
                                    const float* x_kernel, int nx, int cx, const float* y_kernel, int ny, int cy,
                                    int edge, int subsample, float* d_dst, ptrdiff_t dstride);
int vwgpu_separable_convolution(vwgpu_ctx* ctx, const float* src, int w, int h, ptrdiff_t stride,
                                const float* x_kernel, int nx, int cx, const float* y_kernel, int ny, int cy,
                                int edge, int subsample, float* dst, ptrdiff_t dstride);

/* Replaces rasterising vw::convolution_filter(src, kernel, ci, cj, edge) for small 2-D kernels (<= 49 taps)
 * (ConvolutionView, src/vw/Image/Convolution.h:105-170); kernel is row-major kw x kh, HOST pointer.
 * vw::laplacian_filter (src/vw/Image/Filter.h:320-335) is this call with {0,1,0,1,-4,1,0,1,0}, origin (1,1). */
int vwgpu_convolution_2d_dev(vwgpu_ctx* ctx, const float* d_src, int w, int h, ptrdiff_t stride,
                             const float* kernel, int kw, int kh, int ci, int cj, int edge,
                             float* d_dst, ptrdiff_t dstride);
int vwgpu_convolution_2d(vwgpu_ctx* ctx, const float* src, int w, int h, ptrdiff_t stride,
                         const float* kernel, int kw, int kh, int ci, int cj, int edge,
                         float* dst, ptrdiff_t dstride);

/* Replaces vw::stereo::subsample_mask_by_two (src/vw/Stereo/CorrelationView.cc:38-63): a 2x2 block with at least
 * two non-zero pixels gives 255, else 0; dst is (1+(w-1)/2) x (1+(h-1)/2). */
int vwgpu_subsample_mask_by_two_dev(vwgpu_ctx* ctx, const uint8_t* d_src, int w, int h, ptrdiff_t stride,
                                    uint8_t* d_dst, ptrdiff_t dstride);
int vwgpu_subsample_mask_by_two(vwgpu_ctx* ctx, const uint8_t* src, int w, int h, ptrdiff_t stride,
                                uint8_t* dst, ptrdiff_t dstride);

/* Replaces vw::stereo::prefilter_image (src/vw/Stereo/PreFilter.h:76-95): NONE = copy, MEANSUB = image -
 * gaussian_filter(image, width), LOG = laplacian_filter(gaussian_filter(image, width)); constant edge extension. */
int vwgpu_prefilter_image_dev(vwgpu_ctx* ctx, const float* d_src, int w, int h, ptrdiff_t stride,
                              int mode, float width, float* d_dst, ptrdiff_t dstride);
int vwgpu_prefilter_image(vwgpu_ctx* ctx, const float* src, int w, int h, ptrdiff_t stride,
                          int mode, float width, float* dst, ptrdiff_t dstride);

/* ---- parabola sub-pixel refinement ------------------------------------------------------------------- */

/* Replaces rasterising vw::stereo::parabola_subpixel(disparity, left, right, prefilter_mode, prefilter_width,
 * kernel_size) (src/vw/Stereo/ParabolaSubpixelView.h:112-117; ParabolaSubpixelView::prerasterize + evaluate,
 * src/vw/Stereo/ParabolaSubpixelView.cc:31-330) over the whole image.
 *   disp  w x h x {dx, dy, valid in {0.f,1.f}} float (PixelMask<Vector2f>; truncated to int like the reference),
 *         same size as the left image (the reference VW_ASSERTs this); strides of disp / out in PIXELS.
 *   left  w x h float, right rw x rh float; both are prefiltered internally (mode / width as in prefilter_image),
 *         with the reference's constant edge extension wherever a window leaves an image.
 *   out   w x h x {dx, dy, valid}: integer disparity + parabola offset (if |offset| < 5 and the nine costs differ),
 *         invalid pixels -> {0,0,0}.
 * Bit-exact on integer-valued imagery with PREFILTER_NONE; otherwise to float rounding. */
int vwgpu_parabola_subpixel_dev(vwgpu_ctx* ctx, const float* d_disp, int w, int h, ptrdiff_t dstride,
                                const float* d_left, ptrdiff_t lstride,
                                const float* d_right, int rw, int rh, ptrdiff_t rstride,
                                int prefilter_mode, float prefilter_width, int kx, int ky,
                                float* d_out, ptrdiff_t ostride);
int vwgpu_parabola_subpixel(vwgpu_ctx* ctx, const float* disp, int w, int h, ptrdiff_t dstride,
                            const float* left, ptrdiff_t lstride,
                            const float* right, int rw, int rh, ptrdiff_t rstride,
                            int prefilter_mode, float prefilter_width, int kx, int ky,
                            float* out, ptrdiff_t ostride);

/* ---- disparity clean-up filters and the zone scheduler ------------------------------------------------- */

/* Replaces rasterising vw::stereo::rm_outliers_using_thresh (cleanup == 0) or
 * vw::stereo::disparity_cleanup_using_thresh (cleanup != 0: a second pass with the hard-coded (1,1,3.0,0.20))
 * over a whole PixelMask<Vector2i> image (src/vw/Stereo/DisparityMap.h:318-441).  src and dst must not alias;
 * strides are dense (w pixels). */
int vwgpu_disparity_filter_dev(vwgpu_ctx* ctx, const int32_t* d_src, int w, int h, int half_h_kernel, int half_v_kernel,
                               double pixel_threshold, double rejection_threshold, int cleanup, int32_t* d_dst);
int vwgpu_disparity_filter(vwgpu_ctx* ctx, const int32_t* src, int w, int h, int half_h_kernel, int half_v_kernel,
                           double pixel_threshold, double rejection_threshold, int cleanup, int32_t* dst);
/* Replaces rasterising vw::stereo::disparity_mask(disparity, left_mask, right_mask)
 * (src/vw/Stereo/DisparityMap.h:97-253), in place; the left mask has the disparity's size. */
int vwgpu_disparity_mask_dev(vwgpu_ctx* ctx, int32_t* d_disp, int w, int h, const uint8_t* d_left_mask,
                             const uint8_t* d_right_mask, int rmw, int rmh);
int vwgpu_disparity_mask(vwgpu_ctx* ctx, int32_t* disp, int w, int h, const uint8_t* left_mask,
                         const uint8_t* right_mask, int rmw, int rmh);
/* Replaces PyramidCorrelationView::disparity_blob_filter at one level (src/vw/Stereo/CorrelationView.cc:242-271:
 * BlobIndexThreaded over the whole image as one tile + ErodeView): every 8-connected component of VALID pixels with at
 * most max_blob_area pixels is erased (pixels become {0,0,0}); max_blob_area < 1 is a no-op.  In place. */
int vwgpu_disparity_blob_filter_dev(vwgpu_ctx* ctx, int32_t* d_disp, int w, int h, int max_blob_area);
int vwgpu_disparity_blob_filter(vwgpu_ctx* ctx, int32_t* disp, int w, int h, int max_blob_area);
/* Replaces vw::stereo::subdivide_regions(disparity, bounding_box(disparity), list, kernel_size)
 * (src/vw/Stereo/Correlation.cc:139-328).  Host logic on a HOST disparity image; each zone is 8 ints
 * {region.min.x, region.min.y, region.max.x, region.max.y, range.min.x, range.min.y, range.max.x, range.max.y}.
 * Returns the number of zones found (only `cap` are written) or a negative vwgpu_status. */
int vwgpu_subdivide_regions(const int32_t* disp, int w, int h, int kx, int ky, int32_t* zones, int cap);

/* ---- pyramid block matching -------------------------------------------------------------------------- */

/* Arguments of vw::stereo::pyramid_correlate (src/vw/Stereo/CorrelationView.h:195-230) that steer one tile. */
typedef struct vwgpu_pyramid_params {
  int prefilter_mode;            /* vwgpu_prefilter */
  float prefilter_width;
  int search_min_x, search_min_y, search_max_x, search_max_y;   /* BBox2i search_region, half open */
  int kernel_x, kernel_y;
  int cost_type;                 /* vwgpu_cost_type */
  int corr_timeout;              /* seconds; 0 = none.  Uses the reference's estimate seconds_per_op * search volume */
  double seconds_per_op;
  float consistency_threshold;   /* < 0: no L/R check */
  int min_consistency_level;     /* accepted for signature parity; block matching checks at level 0 only */
  int filter_half_kernel;        /* 0: no clean-up filtering */
  int max_pyramid_levels;
  int algorithm;                 /* 0 = VW_CORRELATION_BM, 1 = _SGM, 2 = _MGM (use_mgm at every level), 3 = _FINAL_MGM (at level 0 only; CorrelationView.cc:365-366) */
  int blob_filter_area;          /* 0 = off; level i erases blobs of <= area / 2^i valid pixels */
  /* SGM only (CorrelationView.h:211-214): */
  int sgm_subpixel_mode;         /* vwgpu_sgm_subpixel; the reference's default is LC_BLEND */
  int sgm_search_buffer_x, sgm_search_buffer_y;   /* default (2,2) */
  size_t memory_limit_mb;        /* default 6000 */
  int sgm_num_threads;           /* enters the memory-cap formula only; 0 = 1 */
  /* Optional L-R / R-L discrepancy output of the level-0 consistency check (m_lr_disp_diff, m_region_ul,
   * CorrelationView.h:84; cross_corr_consistency_check, Correlate.cc:1441-1502): PixelMask<float> = {value, valid} per pixel,
   * lr_disp_diff_cols x _rows, covering image pixels starting at (region_ul_x, region_ul_y); stride in pixels (0 = cols).
   * Device pointer for the _dev entry point, host pointer for the host one; NULL = off.  Only kept pixels are written. */
  float* lr_disp_diff;
  int lr_disp_diff_cols, lr_disp_diff_rows;
  ptrdiff_t lr_disp_diff_stride;
  int region_ul_x, region_ul_y;
} vwgpu_pyramid_params;

/* Replaces PyramidCorrelationView::prerasterize(bbox) for VW_CORRELATION_BM (src/vw/Stereo/CorrelationView.cc:273-886):
 * one output tile [bx,bx+bw) x [by,by+bh) of pyramid_correlate(left, right, left_mask, right_mask, ...).
 * Masks may be NULL (everything valid); out is bw x bh x {dx, dy, valid} float (PixelMask<Vector2f>), ostride in pixels. */
int vwgpu_pyramid_correlate_dev(vwgpu_ctx* ctx, const float* d_left, int lw, int lh, ptrdiff_t lstride,
                                const float* d_right, int rw, int rh, ptrdiff_t rstride,
                                const uint8_t* d_left_mask, ptrdiff_t lmstride,
                                const uint8_t* d_right_mask, ptrdiff_t rmstride,
                                const vwgpu_pyramid_params* params, int bx, int by, int bw, int bh,
                                float* d_out, ptrdiff_t ostride);
int vwgpu_pyramid_correlate(vwgpu_ctx* ctx, const float* left, int lw, int lh, ptrdiff_t lstride,
                            const float* right, int rw, int rh, ptrdiff_t rstride,
                            const uint8_t* left_mask, ptrdiff_t lmstride,
                            const uint8_t* right_mask, ptrdiff_t rmstride,
                            const vwgpu_pyramid_params* params, int bx, int by, int bw, int bh,
                            float* out, ptrdiff_t ostride);

/* Several output tiles of the SAME image pair and parameters in one call (round 5) — what the reference's block rasteriser hands to its tile
 * threads one by one (src/vw/Image/ImageIO.h:228-251, BlockProcessor.h:52-176; tools/correlate.cc:266 cuts 1024^2 tiles).  Tile t is
 * [bx[t], bx[t] + bw[t]) x [by[t], by[t] + bh[t]) and goes to outs[t] (bw[t] x bh[t] x 3 floats, row stride ostride[t] pixels; ostride NULL
 * or 0 = dense).  Runs of consecutive tiles of EQUAL size (at most 16) go through the pyramid level loop TOGETHER: every launch serves the
 * whole group and one host round trip per level brings back the zone scheduler's tables of all its tiles (a lone tile is ~50 dependent
 * launches of which the coarse levels are pure latency).  Each tile's result is identical to vwgpu_pyramid_correlate[_dev] on that tile.
 * Grouping applies to VW_CORRELATION_BM without lr_disp_diff, blob filter and corr_timeout; everything else runs tile by tile inside the
 * call.  One context = one group at a time; callers that want more in flight use one context per host thread, as for single tiles. */
int vwgpu_pyramid_correlate_batch_dev(vwgpu_ctx* ctx, const float* d_left, int lw, int lh, ptrdiff_t lstride,
                                      const float* d_right, int rw, int rh, ptrdiff_t rstride,
                                      const uint8_t* d_left_mask, ptrdiff_t lmstride,
                                      const uint8_t* d_right_mask, ptrdiff_t rmstride,
                                      const vwgpu_pyramid_params* params, int n_tiles, const int* bx, const int* by, const int* bw, const int* bh,
                                      float* const* d_outs, const ptrdiff_t* ostride);
int vwgpu_pyramid_correlate_batch(vwgpu_ctx* ctx, const float* left, int lw, int lh, ptrdiff_t lstride,
                                  const float* right, int rw, int rh, ptrdiff_t rstride,
                                  const uint8_t* left_mask, ptrdiff_t lmstride,
                                  const uint8_t* right_mask, ptrdiff_t rmstride,
                                  const vwgpu_pyramid_params* params, int n_tiles, const int* bx, const int* by, const int* bw, const int* bh,
                                  float* const* outs, const ptrdiff_t* ostride);

/* ---- semi-global matching ---------------------------------------------------------------------------- */

typedef enum vwgpu_sgm_subpixel {      /* SemiGlobalMatcher::SgmSubpixelMode, src/vw/Stereo/SGM.h:93-99 */
  VWGPU_SUBPIXEL_NONE = 0, VWGPU_SUBPIXEL_PARABOLA = 1, VWGPU_SUBPIXEL_LINEAR = 2, VWGPU_SUBPIXEL_POLY4 = 3,
  VWGPU_SUBPIXEL_COSINE = 4, VWGPU_SUBPIXEL_LC_BLEND = 5
} vwgpu_sgm_subpixel;

/* The arguments of vw::stereo::calc_disparity_sgm / SemiGlobalMatcher::set_parameters that are not images
 * (src/vw/Stereo/SGM.h:108-147, 360-375). */
typedef struct vwgpu_sgm_params {
  int cost_type;                 /* VWGPU_CENSUS_TRANSFORM or VWGPU_TERNARY_CENSUS_TRANSFORM (others: NOIMPL, like the reference; see allow_block_cost) */
  int use_mgm;                   /* != 0: accum_mgm_multithread (SGM.cc:2619-2700) instead of the eight independent path sweeps */
  int kernel_size;               /* 3, 5, 7 or 9 */
  int subpixel_mode;             /* vwgpu_sgm_subpixel */
  int search_buffer_x, search_buffer_y;
  size_t memory_limit_mb;        /* cap on the cost + accumulation buffers, as in calc_main_buf_size (SGM.cc:677-731) */
  int p1, p2;                    /* 0 = the reference's defaults for the cost type / kernel size */
  int ternary_census_threshold;  /* the reference's default is 5 */
  int num_threads;               /* only enters the memory-cap formula (line buffers per thread); >= 1 */
  int allow_block_cost;          /* 0 (default): costs other than census return VWGPU_ERR_NOIMPL, as compute_disparity_costs throws
                                  * (SGM.cc:1887-1892).  1: VWGPU_ABSOLUTE_DIFFERENCE / VWGPU_SQUARED_DIFFERENCE run the code behind that
                                  * throw — fill_costs_block's mean-abs-difference block cost (:1651-1738, p1 = 3, p2 = 250 by default),
                                  * odd kernel sizes 1 .. 15.  A reference code path that is unreachable upstream: explicit opt-in only. */
} vwgpu_sgm_params;

/* Replaces vw::stereo::calc_disparity_sgm (src/vw/Stereo/SGM.cc:167-229) on already cropped regions:
 *   left  lw x lh float, right rw x rh float with rw >= lw + sx, rh >= lh + sy (the reference crops the right image to
 *   left_region grown by search_volume, :190-191); search_volume (sx, sy) is INCLUSIVE here: (sx+1) x (sy+1) disparities.
 *   left_mask (optional)  : exactly the output size; right_mask (optional): at least output size + (sx, sy);
 *   prev_disparity (optional): half-resolution PixelMask<Vector2i> of the previous pyramid level.
 *   out_disp : ow x oh x {dx, dy, valid} int32 with ow = lw - kernel + 1 (when the right image is large enough);
 *   out_subpixel (optional): the matcher's create_disparity_view_subpixel (SGM.cc:1497-1614) of that result.
 * cap_pixels = capacity of the output buffers in pixels; *ow / *oh receive the output size. */
int vwgpu_calc_disparity_sgm_dev(vwgpu_ctx* ctx, const vwgpu_sgm_params* params,
                                 const float* d_left, int lw, int lh, ptrdiff_t lstride,
                                 const float* d_right, int rw, int rh, ptrdiff_t rstride, int sx, int sy,
                                 const uint8_t* d_left_mask, int lmw, int lmh, const uint8_t* d_right_mask, int rmw, int rmh,
                                 const int32_t* d_prev_disparity, int pw, int ph,
                                 int32_t* d_out_disp, float* d_out_subpixel, size_t cap_pixels, int* ow, int* oh);
int vwgpu_calc_disparity_sgm(vwgpu_ctx* ctx, const vwgpu_sgm_params* params,
                             const float* left, int lw, int lh, ptrdiff_t lstride,
                             const float* right, int rw, int rh, ptrdiff_t rstride, int sx, int sy,
                             const uint8_t* left_mask, int lmw, int lmh, const uint8_t* right_mask, int rmw, int rmh,
                             const int32_t* prev_disparity, int pw, int ph,
                             int32_t* out_disp, float* out_subpixel, size_t cap_pixels, int* ow, int* oh);

/* Host-side view of the launch schedule of the MGM passes (use_mgm; accum_mgm_multithread, src/vw/Stereo/SGM.cc:2619-2700): the
 * raster loops of the eight SmoothPathAccumTask passes (src/vw/Stereo/SGMAssist.h:911-1236) are walked as FRONTS, sets of pixels whose
 * two predecessors lie in the previous front (csrc/mgm_schedule.h).  Direction 0 L, 1 TL, 2 R, 3 BR, 4 T, 5 BL, 6 B, 7 TR.
 * vwgpu_mgm_front_count: number of fronts of a direction on a cols x rows output (< 0: bad arguments).
 * vwgpu_mgm_front_pixel: pixel `index` of front `front`: returns 1 and fills c_r = {c, r}, preds = {c + ax, r + ay, c + bx, r + by}
 *   (path predecessor, second predecessor) and *uses_preds (the task's border test: 0 = the pixel keeps its local costs);
 *   0 when the front has no such pixel, < 0 on bad arguments.  No context, no device work: this is what the launcher enumerates. */
int vwgpu_mgm_front_count(int cols, int rows, int direction);
int vwgpu_mgm_front_pixel(int cols, int rows, int direction, int front, int index, int* c_r, int* preds, int* uses_preds);

/* ---- multi-GPU: halo rows of a row-sharded source (csrc/halo.hip) ---------------------------------------------------------
 * The reference has no distributed mode; its tiles are independent (CorrelationView.cc:89-97, CorrelationView.h:123-133), so one
 * process per GPU can own a strip of tile rows.  When the SOURCE image is sharded the same way, rank g holds rows
 * [g*rows/world, (g+1)*rows/world) and its tiles additionally read `halo_above` / `halo_below` rows of the neighbours
 * (half_kernel * 2^levels + search + collar).  These calls fetch them with RCCL point-to-point transfers (librccl.so is opened
 * at run time; VWGPU_ERR_NOIMPL when it is absent).  The unique id is produced on one rank and handed to the others by the host
 * application (file, MPI, torch store ...), exactly as ncclGetUniqueId / ncclCommInitRank expect. */
typedef struct vwgpu_comm vwgpu_comm;
#define VWGPU_COMM_ID_BYTES 128
int vwgpu_comm_unique_id(void* id128);
int vwgpu_comm_create(vwgpu_ctx* ctx, const void* id128, int rank, int world, vwgpu_comm** comm);
int vwgpu_comm_destroy(vwgpu_comm* comm);
/* rows owned by `rank` and the rows its window spans (clipped to the image); pure host arithmetic, no context */
int vwgpu_halo_plan(int rank, int world, int rows_total, int halo_above, int halo_below, int* owned_a, int* owned_b, int* need_a,
                    int* need_b);
/* The verdict of the request check below from a table of gathered headers {rows_total, halo_above, halo_below, bytes per row} x world:
 * 1 = every rank asked for the same image and halos, 0 = not (rank_a / rank_b: the first differing pair).  Pure host arithmetic. */
int vwgpu_halo_headers_agree(const long long* headers, int world, int* rank_a, int* rank_b);
/* d_owned: the rank's rows (owned_b - owned_a) x cols, contiguous; d_window: (need_b - need_a) x cols, receives own rows + halos;
 * every rank of the communicator must make the call.  The requests of all ranks are compared first — one 32-byte all-gather over
 * the communicator and one host round trip — and either EVERY rank enters the data exchange or every rank returns
 * VWGPU_ERR_ARGUMENT; the exchange itself (one send / recv per neighbour that owns needed rows) is queued on the context's stream
 * and the call returns without waiting for it.  *first_row = need_a. */
int vwgpu_fetch_strip_window_dev(vwgpu_ctx* ctx, vwgpu_comm* comm, const void* d_owned, int cols, int elem_bytes, int rows_total,
                                 int halo_above, int halo_below, void* d_window, int* first_row);

#ifdef __cplusplus
}
#endif
#endif /* VWGPU_H */
