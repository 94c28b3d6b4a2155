// vwgpu_sgm_abi.hip — extern "C" entry points of the SGM family (include/vwgpu.h); argument checks mirror
// calc_disparity_sgm's asserts (src/vw/Stereo/SGM.cc:183-193) and compute_disparity_costs' NoImplErr cases (:1874-1893).
#include "vwgpu_internal.h"
#include "mgm_schedule.h"

namespace {
int check_sgm(vwgpu_ctx* ctx, const vwgpu_sgm_params* P, const void* l, int lw, int lh, const void* r, int rw, int rh, int sx, int sy,
              const void* out, int* ow, int* oh) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  ctx->err.clear();
  if (!P || !l || !r || !out || !ow || !oh || lw <= 0 || lh <= 0 || rw <= 0 || rh <= 0 || sx < 0 || sy < 0)
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "calc_disparity_sgm: null pointer, empty image or negative search volume");
  const bool census = P->cost_type == VWGPU_CENSUS_TRANSFORM || P->cost_type == VWGPU_TERNARY_CENSUS_TRANSFORM;
  // The reference throws for every other cost (SGM.cc:1887-1892) and so does this, unless the caller opts in to the code behind
  // that throw: fill_costs_block's mean-abs-difference cost (:1651-1738) for ABSOLUTE_DIFFERENCE / SQUARED_DIFFERENCE (both take
  // the "Mean of abs differences" branch; the NCC flavour of get_cost_block, cost type 2, is not restated).
  if (!census && !(P->allow_block_cost == 1 && (P->cost_type == VWGPU_ABSOLUTE_DIFFERENCE || P->cost_type == VWGPU_SQUARED_DIFFERENCE)))
    return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "With SGM/MGM, only the census transform cost mode gives good results.");
  if (census && P->kernel_size != 3 && P->kernel_size != 5 && P->kernel_size != 7 && P->kernel_size != 9)
    return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "Census transforms are only available in size 3, 5, 7, and 9.");
  if (!census && (P->kernel_size < 1 || P->kernel_size % 2 != 1 || P->kernel_size > 15))
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "calc_disparity_sgm: the block cost takes odd kernel sizes 1 .. 15");
  if (P->kernel_size > lw || P->kernel_size > lh)
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "calc_disparity_sgm: Kernel size too large of active region.");
  if (P->subpixel_mode < VWGPU_SUBPIXEL_NONE || P->subpixel_mode > VWGPU_SUBPIXEL_LC_BLEND)
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "calc_disparity_sgm: unknown sub-pixel mode %d", P->subpixel_mode);
  if (P->search_buffer_x < 0 || P->search_buffer_y < 0 || P->p1 < 0 || P->p2 < 0 || P->p1 > 65535 || P->p2 > 65535)
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "calc_disparity_sgm: bad search buffer / penalties");
  return VWGPU_OK;
}
}  // namespace

extern "C" {

int vwgpu_mgm_front_count(int cols, int rows, int direction) {
  if (cols <= 0 || rows <= 0 || direction < 0 || direction > 7) return VWGPU_ERR_ARGUMENT;
  return vwgpu::mgm_front_count(vwgpu::kMgmDirs[direction].kind, cols, rows);
}

int vwgpu_mgm_front_pixel(int cols, int rows, int direction, int front, int index, int* c_r, int* preds, int* uses_preds) {
  if (cols <= 0 || rows <= 0 || direction < 0 || direction > 7 || !c_r || !preds || !uses_preds) return VWGPU_ERR_ARGUMENT;
  const vwgpu::MgmDir& d = vwgpu::kMgmDirs[direction];
  int c = 0, r = 0;
  if (index >= vwgpu::mgm_front_width(d.kind, front, cols, rows)) return 0;          // what the launcher sizes the grid with
  if (!vwgpu::mgm_front_pixel(d.kind, d.flipx, d.flipy, front, index, cols, rows, c, r)) return 0;
  c_r[0] = c; c_r[1] = r;
  preds[0] = c + d.ax; preds[1] = r + d.ay; preds[2] = c + d.bx; preds[3] = r + d.by;
  *uses_preds = vwgpu::mgm_uses_predecessors(d.need, c, r, cols, rows) ? 1 : 0;
  return 1;
}

int vwgpu_calc_disparity_sgm_dev(vwgpu_ctx* ctx, const vwgpu_sgm_params* P, const float* d_left, int lw, int lh, ptrdiff_t ls,
                                 const float* d_right, int rw, int rh, ptrdiff_t rs, int sx, int sy,
                                 const uint8_t* d_lmask, int lmw, int lmh, const uint8_t* d_rmask, int rmw, int rmh,
                                 const int32_t* d_prev, int pw, int ph, int32_t* d_out, float* d_sub, size_t cap, int* ow, int* oh) {
  int rc = check_sgm(ctx, P, d_left, lw, lh, d_right, rw, rh, sx, sy, d_out, ow, oh);
  if (rc) return rc;
  if (ls == 0) ls = lw;
  if (rs == 0) rs = rw;
  if (ls < lw || rs < rw) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "calc_disparity_sgm: row stride smaller than row width");
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  return vwgpu_sgm_impl(ctx, P, d_left, lw, lh, ls, d_right, rw, rh, rs, sx, sy, d_lmask, lmw, lmh, d_rmask, rmw, rmh, d_prev, pw, ph,
                        d_out, d_sub, cap, ow, oh);
}

int vwgpu_calc_disparity_sgm(vwgpu_ctx* ctx, const vwgpu_sgm_params* P, const float* left, int lw, int lh, ptrdiff_t ls,
                             const float* right, int rw, int rh, ptrdiff_t rs, int sx, int sy,
                             const uint8_t* lmask, int lmw, int lmh, const uint8_t* rmask, int rmw, int rmh,
                             const int32_t* prev, int pw, int ph, int32_t* out, float* sub, size_t cap, int* ow, int* oh) {
  int rc = check_sgm(ctx, P, left, lw, lh, right, rw, rh, sx, sy, out, ow, oh);
  if (rc) return rc;
  if (ls == 0) ls = lw;
  if (rs == 0) rs = rw;
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  const size_t lb = vwgpu_align_up((size_t)lw * lh * 4, 256), rb = vwgpu_align_up((size_t)rw * rh * 4, 256);
  const size_t lmb = lmask ? vwgpu_align_up((size_t)lmw * lmh, 256) : 0, rmb = rmask ? vwgpu_align_up((size_t)rmw * rmh, 256) : 0;
  const size_t pb = prev ? vwgpu_align_up((size_t)pw * ph * 12, 256) : 0;
  const size_t ob = vwgpu_align_up((size_t)lw * lh * 12, 256);
  rc = vwgpu_arena_reserve(ctx, &ctx->staging, lb + rb + lmb + rmb + pb + 2 * ob);
  if (rc) return rc;
  char* q = static_cast<char*>(ctx->staging.base);
  float* d_l = reinterpret_cast<float*>(q); q += lb;
  float* d_r = reinterpret_cast<float*>(q); q += rb;
  uint8_t* d_lm = reinterpret_cast<uint8_t*>(q); q += lmb;
  uint8_t* d_rm = reinterpret_cast<uint8_t*>(q); q += rmb;
  int32_t* d_p = reinterpret_cast<int32_t*>(q); q += pb;
  int32_t* d_o = reinterpret_cast<int32_t*>(q); q += ob;
  float* d_s = reinterpret_cast<float*>(q);
  hipStream_t st = ctx->stream;
  VWGPU_HIP(ctx, hipMemcpy2DAsync(d_l, (size_t)lw * 4, left, (size_t)ls * 4, (size_t)lw * 4, lh, hipMemcpyHostToDevice, st));
  VWGPU_HIP(ctx, hipMemcpy2DAsync(d_r, (size_t)rw * 4, right, (size_t)rs * 4, (size_t)rw * 4, rh, hipMemcpyHostToDevice, st));
  if (lmask) VWGPU_HIP(ctx, hipMemcpyAsync(d_lm, lmask, (size_t)lmw * lmh, hipMemcpyHostToDevice, st));
  if (rmask) VWGPU_HIP(ctx, hipMemcpyAsync(d_rm, rmask, (size_t)rmw * rmh, hipMemcpyHostToDevice, st));
  if (prev) VWGPU_HIP(ctx, hipMemcpyAsync(d_p, prev, (size_t)pw * ph * 12, hipMemcpyHostToDevice, st));
  rc = vwgpu_sgm_impl(ctx, P, d_l, lw, lh, lw, d_r, rw, rh, rw, sx, sy, lmask ? d_lm : nullptr, lmw, lmh, rmask ? d_rm : nullptr, rmw, rmh,
                      prev ? d_p : nullptr, pw, ph, d_o, sub ? d_s : nullptr, (size_t)lw * lh, ow, oh);
  if (rc) return rc;
  const size_t n = (size_t)(*ow) * (*oh);
  if (n > cap) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "calc_disparity_sgm: output buffer too small (%d x %d needed)", *ow, *oh);
  VWGPU_HIP(ctx, hipMemcpyAsync(out, d_o, n * 12, hipMemcpyDeviceToHost, st));
  if (sub) VWGPU_HIP(ctx, hipMemcpyAsync(sub, d_s, n * 12, hipMemcpyDeviceToHost, st));
  VWGPU_HIP(ctx, hipStreamSynchronize(st));
  return VWGPU_OK;
}

}  // extern "C"
