// vwgpu_abi.hip — extern "C" entry points of libvwgpu.so (declared in include/vwgpu.h).
// Argument validation, path dispatch, host staging, error text.  No kernels here.
#include <climits>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vwgpu_internal.h"

// ---- helpers ----------------------------------------------------------------------------------------

int vwgpu_fail(vwgpu_ctx* ctx, int status, const char* fmt, ...) {
  if (ctx) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    ctx->err = buf;
  }
  return status;
}

int vwgpu_arena_reserve(vwgpu_ctx* ctx, vwgpu_arena* a, size_t bytes) {
  if (bytes <= a->cap) return VWGPU_OK;
  // Growth is geometric and the outgrown block is NOT freed here: hipFree waits for the whole device — every stream of every tile
  // thread — and a tile loop that meets its tiles once (the reference's block_write_image) grows its arenas all through the first
  // dozens of tiles (LoG + NCC tiles on 4 threads: 7.0 ms per tile with the free-and-reallocate of round 2, measured in round 3).
  // Kernels already queued keep using the old block; it goes to the context's graveyard and is released with the context (or when
  // an allocation fails).  The garbage of a geometric series is at most the size of the live block.
  const size_t want = vwgpu_align_up(std::max(bytes + bytes / 4, std::min(2 * a->cap, a->cap + ((size_t)4 << 30))), 1 << 20);   // doubling, by at most 4 GB
  void* fresh = nullptr;
  hipError_t e = hipMalloc(&fresh, want);
  if (e != hipSuccess) {                                 // out of memory: give the garbage back and try once more
    (void)hipGetLastError();
    (void)hipStreamSynchronize(ctx->stream);
    for (void* g : ctx->graveyard) (void)hipFree(g);
    ctx->graveyard.clear(); ctx->graveyard_bytes = 0;
    if (a->base) { (void)hipFree(a->base); a->base = nullptr; a->cap = 0; }
    VWGPU_HIP(ctx, hipMalloc(&fresh, vwgpu_align_up(bytes + bytes / 4, 1 << 20)));
    a->base = fresh;
    a->cap = vwgpu_align_up(bytes + bytes / 4, 1 << 20);
    return VWGPU_OK;
  }
  if (a->base) { ctx->graveyard.push_back(a->base); ctx->graveyard_bytes += a->cap; }
  a->base = fresh;
  a->cap = want;
  return VWGPU_OK;
}

// The pinned ring is two halves.  Pieces are handed out in address order; when the cursor leaves a half, an event recorded on the
// stream marks "every copy that reads this half has been queued before here", and the cursor enters the other half only after the
// event recorded when IT was last left has completed.  So a piece is never rewritten while a copy that reads it is still pending,
// however far the host runs ahead of the device (the disparity-group loop of bm_exact.hip queues hundreds of table uploads
// without a synchronisation; before this the ring relied on callers synchronising "at least once per pyramid level").
// CONTRACT: the copy that reads a piece must be queued on ctx->stream BEFORE the next vwgpu_host_ring call — the event that
// protects a half is recorded when the cursor leaves it, i.e. inside a later call, and covers only what was queued by then.  Every caller
// fills the piece and queues its hipMemcpyAsync at once (upload_pieces, the leaf-extent read-back, bm_exact's table uploads).
void* vwgpu_host_ring(vwgpu_ctx* ctx, size_t bytes) {
  const size_t CAP = (size_t)ctx->host_ring_kb << 10;
  if (!ctx->host_ring) {
    void* p = nullptr;
    if (hipHostMalloc(&p, CAP, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    ctx->host_ring = static_cast<char*>(p); ctx->host_cap = CAP; ctx->host_pos = 0; ctx->host_half = 0;
    ctx->ring_event_set[0] = ctx->ring_event_set[1] = false;
    for (int i = 0; i < 2; ++i)
      if (!ctx->ring_event[i] && hipEventCreateWithFlags(&ctx->ring_event[i], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  }
  bytes = vwgpu_align_up(bytes, 256);
  if (bytes == 0 || bytes > ctx->host_cap / 4) return nullptr;
  const size_t half = ctx->host_cap / 2;
  size_t pos = ctx->host_pos;
  if (pos < half && pos + bytes > half) pos = half;           // a piece never straddles the halves
  if (pos + bytes > ctx->host_cap) pos = 0;
  const int h = pos >= half ? 1 : 0;
  if (h != ctx->host_half) {
    if (hipEventRecord(ctx->ring_event[ctx->host_half], ctx->stream) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    ctx->ring_event_set[ctx->host_half] = true;
    if (ctx->ring_event_set[h] && hipEventSynchronize(ctx->ring_event[h]) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    ctx->host_half = h;
    ++ctx->ring_wraps;
  }
  ctx->host_pos = pos + bytes;
  return ctx->host_ring + pos;
}

vwgpu_prof_scope::vwgpu_prof_scope(vwgpu_ctx* c, const char* name) : ctx(c) {
  if (!ctx->profiling) return;
  hipEvent_t a = nullptr;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { b = nullptr; return; }
  (void)hipEventRecord(a, ctx->stream);
  ctx->prof.push_back({name, a, b});
}
vwgpu_prof_scope::~vwgpu_prof_scope() {
  if (b) (void)hipEventRecord(b, ctx->stream);
}

static void prof_clear(vwgpu_ctx* ctx) {
  for (auto& r : ctx->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  ctx->prof.clear();
}

// ---- context ------------------------------------------------------------------------------------------

// Two alternating device flags: call n raises flags[n&1] on unsuitable input and clears flags[(n+1)&1] for the next call, so
// no memset launch is needed (stream order makes this race free).  `extra` (optional) = extra_ints ints after the flags.
int vwgpu_next_flags(vwgpu_ctx* ctx, size_t extra_ints, int** flag_set, int** flag_clear, int** extra) {
  int rc = vwgpu_arena_reserve(ctx, &ctx->flags, 256 + extra_ints * sizeof(int));
  if (rc) return rc;
  int* flags = static_cast<int*>(ctx->flags.base);
  if (ctx->flags_base_seen != ctx->flags.base) { ctx->flags_init = false; ctx->flags_base_seen = ctx->flags.base; ctx->last_flag = nullptr; }
  if (!ctx->flags_init) {
    VWGPU_HIP(ctx, hipMemsetAsync(flags, 0, 256, ctx->stream));
    ctx->flags_init = true;
  }
  *flag_set = flags + (ctx->flag_parity & 1);
  *flag_clear = flags + ((ctx->flag_parity + 1) & 1);
  ctx->flag_parity ^= 1;
  if (extra) *extra = flags + 64;
  return VWGPU_OK;
}

extern "C" {

int vwgpu_abi_version(void) { return VWGPU_ABI_VERSION; }

const char* vwgpu_strerror(int status) {
  switch (status) {
    case VWGPU_OK: return "ok";
    case VWGPU_ERR_ARGUMENT: return "invalid argument";
    case VWGPU_ERR_NOIMPL: return "not implemented for this configuration";
    case VWGPU_ERR_HIP: return "HIP runtime error";
    case VWGPU_ERR_NOMEM: return "out of device memory";
    case VWGPU_ERR_LOGIC: return "internal logic error";
    default: return "unknown status";
  }
}

int vwgpu_create(vwgpu_ctx** out, int device) {
  if (!out) return VWGPU_ERR_ARGUMENT;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return VWGPU_ERR_HIP;   // no silent CPU fallback
  if (device < 0 || device >= n) return VWGPU_ERR_ARGUMENT;
  if (hipSetDevice(device) != hipSuccess) return VWGPU_ERR_HIP;
  vwgpu_ctx* ctx = new vwgpu_ctx();
  ctx->device = device;
  if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return VWGPU_ERR_HIP; }
  ctx->stream = ctx->own_stream;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->num_cu = prop.multiProcessorCount;
  *out = ctx;
  return VWGPU_OK;
}

// Every device arena of a context.  The stream must be idle: nothing queued may still use the blocks.
static size_t free_arenas(vwgpu_ctx* ctx) {
  size_t freed = ctx->graveyard_bytes;
  vwgpu_arena* all[] = {&ctx->scratch, &ctx->flags, &ctx->filt, &ctx->misc, &ctx->pyr, &ctx->ztab, &ctx->zrl, &ctx->zext, &ctx->sgm,
                        &ctx->sgm_main, &ctx->sgm_bnd, &ctx->xvol, &ctx->xtab, &ctx->xcarry, &ctx->staging};
  for (vwgpu_arena* a : all) {
    if (a->base) { (void)hipFree(a->base); freed += a->cap; }
    a->base = nullptr; a->cap = 0;
  }
  for (auto& lr : ctx->leaf_rects) if (lr.d_rects) (void)hipFree(lr.d_rects);
  ctx->leaf_rects.clear();
  for (void* g : ctx->graveyard) (void)hipFree(g);
  ctx->graveyard.clear(); ctx->graveyard_bytes = 0;
  // state that pointed into the arenas
  ctx->flags_init = false; ctx->flags_base_seen = nullptr; ctx->last_flag = nullptr;
  ctx->sgm_epoch = 0;                               // (sgm_bnd is zero-initialised on its next reservation)
  return freed;
}

void vwgpu_destroy(vwgpu_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  prof_clear(ctx);
  vwgpu_pool_destroy(ctx);
  (void)free_arenas(ctx);
  if (ctx->host_ring) (void)hipHostFree(ctx->host_ring);
  for (int i = 0; i < 2; ++i) if (ctx->ring_event[i]) (void)hipEventDestroy(ctx->ring_event[i]);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
}

int vwgpu_trim(vwgpu_ctx* ctx, size_t* freed_bytes) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const size_t freed = free_arenas(ctx);
  if (freed_bytes) *freed_bytes = freed;
  return VWGPU_OK;
}

int vwgpu_set_stream(vwgpu_ctx* ctx, void* hip_stream) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  hipStream_t s = static_cast<hipStream_t>(hip_stream);
  if (s == ctx->stream && !ctx->stream_is_own) return VWGPU_OK;
  // The scratch arenas are shared by consecutive calls: work queued on the old stream must finish first.
  (void)hipStreamSynchronize(ctx->stream);
  ctx->stream = s;
  ctx->stream_is_own = false;
  ctx->measure_first = false;       // the launch order of the next call must not depend on what another stream's caller fed last
  return VWGPU_OK;
}

int vwgpu_reset_stream(vwgpu_ctx* ctx) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  if (ctx->stream_is_own) return VWGPU_OK;
  (void)hipStreamSynchronize(ctx->stream);
  ctx->stream = ctx->own_stream;
  ctx->stream_is_own = true;
  return VWGPU_OK;
}

int vwgpu_synchronize(vwgpu_ctx* ctx) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
  // a safe point: nothing queued can still use the blocks the arenas have outgrown (they used to stay until vwgpu_destroy)
  if (!ctx->graveyard.empty()) {
    for (void* g : ctx->graveyard) (void)hipFree(g);
    ctx->graveyard.clear(); ctx->graveyard_bytes = 0;
  }
  return VWGPU_OK;
}

const char* vwgpu_last_error(const vwgpu_ctx* ctx) { return ctx ? ctx->err.c_str() : ""; }

int vwgpu_force_path(vwgpu_ctx* ctx, int path) {
  if (!ctx || path < VWGPU_PATH_NONE || path > VWGPU_PATH_DOT_U16) return VWGPU_ERR_ARGUMENT;
  ctx->forced_path = path;
  ctx->measure_first = false;
  return VWGPU_OK;
}

int vwgpu_set_option(vwgpu_ctx* ctx, int option, int value) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  if (option == VWGPU_OPT_DEFER_EXACTNESS) { ctx->defer_exact = value != 0; return VWGPU_OK; }
  // the options below select between variants that return identical results; out-of-range values are refused
  if (option == VWGPU_OPT_SAD_GROUPS && value >= 0 && value <= 3) { ctx->sad_groups = value; return VWGPU_OK; }
  if (option == VWGPU_OPT_EXACT_SCRATCH_MB && value >= 16 && value <= 65536) { ctx->exact_scratch_mb = value; return VWGPU_OK; }
  if (option == VWGPU_OPT_TRACE && value >= 0 && value <= 7) { ctx->trace = value; ctx->cert_px[0] = ctx->cert_px[1] = ctx->cert_px[2] = 0; return VWGPU_OK; }
  if (option == VWGPU_OPT_CERTIFY && (value == 0 || value == 1)) { ctx->certify = value; return VWGPU_OK; }
  if (option == VWGPU_OPT_CERT_F32 && (value == 0 || value == 1)) { ctx->cert_f32 = value; return VWGPU_OK; }
  if (option == VWGPU_OPT_ZONE_TILE16 && value >= 0 && value <= 2) { ctx->zone_tile16 = value; return VWGPU_OK; }
  if (option == VWGPU_OPT_ZONE_SXC && value >= 0 && value <= 4096) { ctx->zone_sxc = value; return VWGPU_OK; }
  if (option == VWGPU_OPT_SGM_SWEEP && value >= 0 && value <= 15) { ctx->sgm_sweep = value; return VWGPU_OK; }
  if (option == VWGPU_OPT_MGM_SWEEP && value >= 0 && value <= 15) { ctx->mgm_sweep = value; return VWGPU_OK; }
  if (option == VWGPU_OPT_SGM_PATH_MODE && value >= 0 && value <= 4095) { ctx->sgm_path_mode = value; return VWGPU_OK; }
  if (option == VWGPU_OPT_EXACT_SPLIT && value >= 0 && value <= 3) { ctx->exact_split = value; return VWGPU_OK; }
  if (option == VWGPU_OPT_HOST_RING_KB && value >= 16 && value <= (1 << 20)) {
    if (value != ctx->host_ring_kb && ctx->host_ring) {            // pending copies read the old ring
      VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
      (void)hipHostFree(ctx->host_ring);
      ctx->host_ring = nullptr; ctx->host_cap = ctx->host_pos = 0;
    }
    ctx->host_ring_kb = value;
    return VWGPU_OK;
  }
  return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "vwgpu_set_option: unknown or read-only option %d, or value %d out of range", option, value);
}

int vwgpu_get_option(const vwgpu_ctx* ctx, int option, int* value) {
  if (!value) return VWGPU_ERR_ARGUMENT;
  if (option == VWGPU_OPT_DEVICE_COUNT) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return VWGPU_ERR_HIP;
    *value = n;
    return VWGPU_OK;
  }
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  if (option == VWGPU_OPT_DEFER_EXACTNESS) { *value = ctx->defer_exact ? 1 : 0; return VWGPU_OK; }
  if (option == VWGPU_OPT_SAD_GROUPS) { *value = ctx->sad_groups; return VWGPU_OK; }
  if (option == VWGPU_OPT_EXACT_SCRATCH_MB) { *value = ctx->exact_scratch_mb; return VWGPU_OK; }
  if (option == VWGPU_OPT_TRACE) { *value = ctx->trace; return VWGPU_OK; }
  if (option == VWGPU_OPT_SGM_SWEEP) { *value = ctx->sgm_sweep; return VWGPU_OK; }
  if (option == VWGPU_OPT_MGM_SWEEP) { *value = ctx->mgm_sweep; return VWGPU_OK; }
  if (option == VWGPU_OPT_SGM_PATH_MODE) { *value = ctx->sgm_path_mode; return VWGPU_OK; }
  if (option == VWGPU_OPT_EXACT_SPLIT) { *value = ctx->exact_split; return VWGPU_OK; }
  if (option == VWGPU_OPT_HOST_RING_KB) { *value = ctx->host_ring_kb; return VWGPU_OK; }
  if (option == VWGPU_OPT_CERTIFY) { *value = ctx->certify; return VWGPU_OK; }
  if (option == VWGPU_OPT_ZONE_SXC) { *value = ctx->zone_sxc; return VWGPU_OK; }
  if (option == VWGPU_OPT_CERT_PERMILLE) {          // share of the pixels (per mille) that were certified since VWGPU_OPT_TRACE was last set; -1: none counted
    const unsigned long long all = ctx->cert_px[0] + ctx->cert_px[1];
    *value = all ? (int)((ctx->cert_px[0] * 1000ull) / all) : -1;
    return VWGPU_OK;
  }
  if (option == VWGPU_OPT_CERT_F32) { *value = ctx->cert_f32; return VWGPU_OK; }
  if (option == VWGPU_OPT_ZONE_TILE16) { *value = ctx->zone_tile16; return VWGPU_OK; }
  if (option == VWGPU_OPT_CERT_F64_PERMILLE) {
    const unsigned long long all = ctx->cert_px[0] + ctx->cert_px[1];
    *value = all ? (int)((ctx->cert_px[2] * 1000ull) / all) : -1;
    return VWGPU_OK;
  }
  if (option == VWGPU_OPT_HOST_RING_WRAPS) { *value = (int)(ctx->ring_wraps & 0x7fffffff); return VWGPU_OK; }
  return VWGPU_ERR_ARGUMENT;
}

int vwgpu_last_path(const vwgpu_ctx* cctx) {
  vwgpu_ctx* ctx = const_cast<vwgpu_ctx*>(cctx);
  if (!ctx) return VWGPU_PATH_NONE;
  if ((ctx->last_path == VWGPU_PATH_SAD_U8 || ctx->last_path == VWGPU_PATH_DOT_U8 || ctx->last_path == VWGPU_PATH_SAD_U16 || ctx->last_path == VWGPU_PATH_DOT_U16) && ctx->last_flag) {
    // A packed kernel was queued without waiting for its verdict (a forced path, or VWGPU_OPT_DEFER_EXACTNESS): it reports input
    // outside its domain through a device flag, and NOTHING recomputed the image then — the call has no result.
    int flag = 0;
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) return VWGPU_PATH_NONE;
    if (hipMemcpy(&flag, ctx->last_flag, sizeof flag, hipMemcpyDeviceToHost) != hipSuccess) return VWGPU_PATH_NONE;
    return flag ? VWGPU_PATH_REFUSED : ctx->last_path;
  }
  return ctx->last_path;
}

int vwgpu_profile_enable(vwgpu_ctx* ctx, int enable) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  ctx->profiling = enable != 0;
  return VWGPU_OK;
}

int vwgpu_profile_reset(vwgpu_ctx* ctx) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  (void)hipStreamSynchronize(ctx->stream);
  prof_clear(ctx);
  return VWGPU_OK;
}

int vwgpu_profile_read(vwgpu_ctx* ctx, const char** names, float* ms, int cap) {
  if (!ctx || cap < 0) return VWGPU_ERR_ARGUMENT;
  VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
  int n = 0;
  for (auto& r : ctx->prof) {
    if (n >= cap) break;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) t = -1.f;
    if (names) names[n] = r.name;
    if (ms) ms[n] = t;
    ++n;
  }
  return n;
}

// ---- calc_disparity -----------------------------------------------------------------------------------

static int check_bm_args(vwgpu_ctx* ctx, int cost_type, const void* l, int lw, int lh, ptrdiff_t ls,
                         const void* r, int rw, int rh, ptrdiff_t rs, int kx, int ky, int sx, int sy,
                         const void* out, ptrdiff_t os) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  ctx->err.clear();
  if (!l || !r || !out) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "calc_disparity: null image pointer");
  if (cost_type < VWGPU_ABSOLUTE_DIFFERENCE || cost_type > VWGPU_CROSS_CORRELATION)
    return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "calc_disparity: cost type %d is not a block-matching cost", cost_type);
  // The reference's checks (src/vw/Stereo/Correlation.cc:341-351, Algorithms.h:45-46), always on here.
  if (kx < 1 || ky < 1 || kx % 2 != 1 || ky % 2 != 1)
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "calc_disparity: Kernel input not sized with odd values.");
  if (kx > lw || ky > lh)
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "calc_disparity: Kernel size too large of active region.");
  if (sx < 1 || sy < 1)
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "calc_disparity: Search volume must be greater than 0.");
  if (rw < lw + sx - 1 || rh < lh + sy - 1)
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "calc_disparity: right raster %dx%d smaller than %dx%d", rw, rh,
                      lw + sx - 1, lh + sy - 1);
  if (ls < lw || rs < rw || (os != 0 && os < lw - kx + 1))
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "calc_disparity: row stride smaller than row width");
  if ((long long)sx * sy > INT32_MAX)        // (the reference's disparity index is an int as well, Correlation.cc:64-66)
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "calc_disparity: search volume %dx%d exceeds the index range", sx, sy);
  return VWGPU_OK;
}

// Inputs outside the packed kernels' domain: float64 kernel with tile-local sums when every partial sum is exactly
// representable (any order returns the reference's bits), the reference's serial summation order otherwise.
struct Grain { bool known = false; int lo = 0, hi = 0, nonfinite = 0; };

static int calc_disparity_classified(vwgpu_ctx* ctx, int cost_type, const float* d_left, int lw, int lh, ptrdiff_t ls,
                                     const float* d_right, int rw, int rh, ptrdiff_t rs, int kx, int ky, int sx, int sy,
                                     int32_t* d_out, ptrdiff_t os, Grain gr = Grain()) {
  bool exact = ctx->forced_path == VWGPU_PATH_EXACT_ORDER;
  ctx->last_flag = nullptr;
  int g_lo = INT_MAX, g_hi = INT_MIN, g_nonfinite = 1;
  if (ctx->forced_path == VWGPU_PATH_NONE) {
    int lo = gr.lo, hi = gr.hi, nonfinite = gr.nonfinite;
    int rc = gr.known ? VWGPU_OK : vwgpu_float_grain(ctx, d_left, lw, lh, ls, d_right, lw + sx - 1, lh + sy - 1, rs, &lo, &hi, &nonfinite);
    if (rc) return rc;
    g_lo = lo; g_hi = hi; g_nonfinite = nonfinite;
    exact = !vwgpu_sums_order_free(cost_type, kx, ky, lo, hi, nonfinite);      // never the tile-local sums on data whose roundings depend on the order
    // integers below 2^16 (16-bit imagery): the packed-u16 SAD kernel; it checks the sign itself and raises its flag
    if (!exact && !nonfinite && lo != INT_MAX && lo >= 0 && hi <= 15 && vwgpu_bm_sad_u16_supported(cost_type, kx, ky, sx, sy)) {   // (nonfinite bit 1 = negative pixels)
      int* d_flag = nullptr;
      rc = vwgpu_launch_bm_sad_u16(ctx, d_left, lw, lh, ls, d_right, rw, rh, rs, kx, ky, sx, sy, d_out, os, &d_flag);
      if (rc) return rc;
      int flag = 0;
      VWGPU_HIP(ctx, hipMemcpyAsync(&flag, d_flag, sizeof flag, hipMemcpyDeviceToHost, ctx->stream));
      VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (!flag) { ctx->last_path = VWGPU_PATH_SAD_U16; return VWGPU_OK; }
    }
    // integers below 2^12 (11- / 12-bit imagery): float32 products and squared differences are exact, SSD / NCC on v_dot2_u32_u16
    if (!exact && !nonfinite && lo != INT_MAX && lo >= 0 && hi <= 11 && vwgpu_bm_corr_u16_supported(cost_type, kx, ky, sx, sy)) {
      int* d_flag = nullptr;
      rc = vwgpu_launch_bm_corr_u16(ctx, cost_type, d_left, lw, lh, ls, d_right, rw, rh, rs, kx, ky, sx, sy, d_out, os, &d_flag);
      if (rc) return rc;
      int flag = 0;
      VWGPU_HIP(ctx, hipMemcpyAsync(&flag, d_flag, sizeof flag, hipMemcpyDeviceToHost, ctx->stream));
      VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (!flag) { ctx->last_path = VWGPU_PATH_DOT_U16; return VWGPU_OK; }
    }
  } else if (ctx->forced_path == VWGPU_PATH_DOT_U16) {
    if (!vwgpu_bm_corr_u16_supported(cost_type, kx, ky, sx, sy))
      return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "calc_disparity: no packed-u16 SSD / NCC path for cost %d kernel %dx%d search %dx%d", cost_type, kx, ky, sx, sy);
    int* d_flag = nullptr;
    int rc = vwgpu_launch_bm_corr_u16(ctx, cost_type, d_left, lw, lh, ls, d_right, rw, rh, rs, kx, ky, sx, sy, d_out, os, &d_flag);
    ctx->last_path = VWGPU_PATH_DOT_U16;
    ctx->last_flag = d_flag;
    return rc;
  } else if (ctx->forced_path == VWGPU_PATH_SAD_U16) {
    if (!vwgpu_bm_sad_u16_supported(cost_type, kx, ky, sx, sy))
      return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "calc_disparity: no packed-u16 path for cost %d kernel %dx%d search %dx%d", cost_type, kx, ky, sx, sy);
    int* d_flag = nullptr;
    int rc = vwgpu_launch_bm_sad_u16(ctx, d_left, lw, lh, ls, d_right, rw, rh, rs, kx, ky, sx, sy, d_out, os, &d_flag);
    ctx->last_path = VWGPU_PATH_SAD_U16;
    ctx->last_flag = d_flag;
    return rc;
  }
  // Float imagery outside the packed classes, automatic dispatch: the tile-parallel zone matcher (bm_zones.hip) with the whole raster as its
  // one zone — the kernel family the pyramid levels run on, several times faster than bm_generic's column-sum tiles and than the
  // exact-order passes.  Order-free data: any summation order returns the reference's bits.  Data whose sums round (finite, |pixel| <
  // 2^60): the CERTIFIED pass — every pixel's winner must lead its runner-up by more than twice the bound on |tile-parallel sum -
  // reference running sum| (chain lengths W + H of the whole raster); a single unproven pixel sends the call to the reference's order.
  if (ctx->forced_path == VWGPU_PATH_NONE && vwgpu_bm_zones_supported(kx, ky) && lw - kx + 1 <= 65535 * 32 && lh - ky + 1 <= 65535 * 32 &&
      os <= INT32_MAX && (long long)sx * sy <= INT32_MAX) {
    vwgpu_zone_task z{0, 0, 0, 0, lw - kx + 1, lh - ky + 1, sx, sy, 0, (int)os, 0, 0};
    const int rcw = lw + sx - 1, rch = lh + sy - 1;               // the part of the right raster the search can reach (no clamped reads)
    if (!exact) {                                                  // (order free implies finite)
      ctx->last_path = VWGPU_PATH_GENERIC_F64;
      return vwgpu_launch_bm_zones(ctx, cost_type, d_left, lw, lh, d_right, rcw, rch, kx, ky, &z, 1, d_out,
                                   vwgpu_sums_bits(cost_type, kx, ky, g_lo, g_hi, g_nonfinite) <= 24 ? 1 : 0, INT_MIN, nullptr, nullptr, nullptr, nullptr, nullptr,
                                   0, 0, 0, 0, ls, rs);
    }
    if (exact && ctx->certify && (g_nonfinite & 1) == 0 && g_lo != INT_MAX && g_hi < 60 && g_hi > -60 && vwgpu_bm_exact_supported(sx, sy)) {
      const int tny = (z.zh + 31) / 32;
      int rc = vwgpu_arena_reserve(ctx, &ctx->misc, 512 + (size_t)tny * sizeof(int));
      if (rc) return rc;
      int* d_word = static_cast<int*>(ctx->misc.base);            // [0] any zone flagged, [1] the zone's flag, [128 ...] one flag per band of 32 rows
      int* d_tflag = d_word + 128;
      VWGPU_HIP(ctx, hipMemsetAsync(d_word, 0, 512 + (size_t)tny * sizeof(int), ctx->stream));
      unsigned long long* d_stats = nullptr;
      if (ctx->trace & 4) {
        d_stats = reinterpret_cast<unsigned long long*>(d_word + 32);
      }
      rc = vwgpu_launch_bm_zones(ctx, cost_type, d_left, lw, lh, d_right, rcw, rch, kx, ky, &z, 1, d_out, 0, g_hi, d_word + 1, d_stats, d_word, nullptr, nullptr,
                                 0, 0, 0, 0, ls, rs, nullptr, d_tflag);
      if (rc) return rc;
      int any = 0;
      VWGPU_HIP(ctx, hipMemcpyAsync(&any, d_word, sizeof any, hipMemcpyDeviceToHost, ctx->stream));
      unsigned long long got[3] = {0, 0, 0};
      if (d_stats) VWGPU_HIP(ctx, hipMemcpyAsync(got, d_stats, sizeof got, hipMemcpyDeviceToHost, ctx->stream));
      VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (d_stats) { ctx->cert_px[0] += got[0]; ctx->cert_px[1] += got[1]; ctx->cert_px[2] += got[2]; }
      if (!any) { ctx->last_path = VWGPU_PATH_CERTIFIED; return VWGPU_OK; }
      // Some pixel could not be proven.  The reference's order is needed only for the BANDS of 32 rows that hold such a pixel — every other
      // pixel of the image already has the reference's value — but its chains start at the raster's first row and column: the column chains
      // run from row 0 down to the last flagged band (without storing anything for the rows nobody reads), the row chains and the selection
      // only over the flagged bands (bm_exact.hip, row ranges).  The rows are rewritten whole: certified pixels get the same values again.
      if ((long long)sx * sy <= 512) {
        std::vector<int> tf((size_t)tny);
        VWGPU_HIP(ctx, hipMemcpyAsync(tf.data(), d_tflag, tf.size() * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::vector<int> rows;
        const int flagged_rows = vwgpu_zone_flagged_rows(tf.data(), z.zh, &rows);
        if (!rows.empty() && flagged_rows * 2 < z.zh) {             // (most of the image flagged: the whole raster in one go is cheaper)
          ctx->last_path = VWGPU_PATH_EXACT_ORDER;
          // ... and only for the columns up to the last flagged tile (round 6): the reference's row chains run from column 0 rightwards, a sum at
          // column x has seen columns <= x + kx - 1 only, so the narrower raster [0, wneed) reproduces them bit for bit (a band's flag holds 1 + its
          // last flagged tile column; tiles are at most 32 pixels wide).  The columns right of it keep their certified values.
          int tmax = 0;
          for (int v : tf) tmax = std::max(tmax, v);
          vwgpu_zone_task z2 = z;
          z2.zw = (int)std::min<long long>(z.zw, (long long)tmax * 32);
          return vwgpu_launch_bm_exact(ctx, cost_type, d_left, lw, lh, ls, d_right, rw, rh, rs, kx, ky, &z2, 1, d_out, nullptr, rows.data(), (int)rows.size() / 2);
        }
      }
    }
  }
  if (exact) {
    if (!vwgpu_bm_exact_supported(sx, sy))
      return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "calc_disparity: %d x %d disparities exceed the index range of the exact-order path", sx, sy);
    ctx->last_path = VWGPU_PATH_EXACT_ORDER;
    vwgpu_zone_task z{0, 0, 0, 0, lw - kx + 1, lh - ky + 1, sx, sy, 0, (int)os, 0, 0};
    return vwgpu_launch_bm_exact(ctx, cost_type, d_left, lw, lh, ls, d_right, rw, rh, rs, kx, ky, &z, 1, d_out);
  }
  ctx->last_path = VWGPU_PATH_GENERIC_F64;
  return vwgpu_launch_bm_generic(ctx, cost_type, d_left, lw, lh, ls, d_right, rw, rh, rs, kx, ky, sx, sy, d_out, os);
}

int vwgpu_calc_disparity_dev(vwgpu_ctx* ctx, int cost_type,
                             const float* d_left, int lw, int lh, ptrdiff_t ls,
                             const float* d_right, int rw, int rh, ptrdiff_t rs,
                             int kx, int ky, int sx, int sy, int32_t* d_out, ptrdiff_t os) {
  int rc = check_bm_args(ctx, cost_type, d_left, lw, lh, ls, d_right, rw, rh, rs, kx, ky, sx, sy, d_out, os);
  if (rc) return rc;
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  if (os == 0) os = lw - kx + 1;
  if (os > INT32_MAX) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "calc_disparity: output stride too large");

  const bool sad_ok = vwgpu_bm_sad_u8_supported(cost_type, kx, ky, sx, sy);
  const bool corr_ok = !sad_ok && vwgpu_bm_corr_u8_supported(cost_type, kx, ky, sx, sy);
  const bool dot_ok = !sad_ok && (corr_ok || vwgpu_bm_dot_u8_supported(cost_type, kx, ky, sx, sy));
  if (ctx->forced_path == VWGPU_PATH_SAD_U8 && !sad_ok)
    return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "calc_disparity: no packed-u8 path for cost %d kernel %dx%d search %dx%d",
                      cost_type, kx, ky, sx, sy);
  if (ctx->forced_path == VWGPU_PATH_DOT_U8 && !dot_ok)
    return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "calc_disparity: no dot-product path for cost %d kernel %dx%d search %dx%d",
                      cost_type, kx, ky, sx, sy);
  const bool try_packed = (sad_ok && (ctx->forced_path == VWGPU_PATH_NONE || ctx->forced_path == VWGPU_PATH_SAD_U8)) ||
                          (dot_ok && (ctx->forced_path == VWGPU_PATH_NONE || ctx->forced_path == VWGPU_PATH_DOT_U8));
  if (!try_packed)
    return calc_disparity_classified(ctx, cost_type, d_left, lw, lh, ls, d_right, rw, rh, rs, kx, ky, sx, sy, d_out, os);

  // A context whose previous call was not byte imagery measures the class first (0.1 ms at 4096^2) instead of spending a packed-u8
  // launch on data that will be refused again; the order of the attempts is all this changes.
  if (ctx->forced_path == VWGPU_PATH_NONE && !ctx->defer_exact && ctx->measure_first) {
    Grain gr;
    rc = vwgpu_float_grain(ctx, d_left, lw, lh, ls, d_right, lw + sx - 1, lh + sy - 1, rs, &gr.lo, &gr.hi, &gr.nonfinite);
    if (rc) return rc;
    gr.known = true;
    const bool bytes = !gr.nonfinite && (gr.lo == INT_MAX || (gr.lo >= 0 && gr.hi <= 7));
    if (!bytes) return calc_disparity_classified(ctx, cost_type, d_left, lw, lh, ls, d_right, rw, rh, rs, kx, ky, sx, sy, d_out, os, gr);
  }
  // Integer-valued inputs in [0,255]: the packed kernels; they check the domain while converting and raise a device flag.
  int* d_flag = nullptr;
  if (sad_ok) rc = vwgpu_launch_bm_sad_u8(ctx, d_left, lw, lh, ls, d_right, rw, rh, rs, kx, ky, sx, sy, d_out, os, &d_flag);
  else if (corr_ok) rc = vwgpu_launch_bm_corr_u8(ctx, cost_type, d_left, lw, lh, ls, d_right, rw, rh, rs, kx, ky, sx, sy, d_out, os, &d_flag);
  else rc = vwgpu_launch_bm_dot_u8(ctx, cost_type, d_left, lw, lh, ls, d_right, rw, rh, rs, kx, ky, sx, sy, d_out, os, &d_flag);
  if (rc) return rc;
  ctx->last_path = sad_ok ? VWGPU_PATH_SAD_U8 : VWGPU_PATH_DOT_U8;
  ctx->last_flag = d_flag;
  if (ctx->forced_path != VWGPU_PATH_NONE) return VWGPU_OK;       // caller inspects vwgpu_last_path()
  // Pipelined callers: ONE launch and no host round trip.  Nothing runs behind the flag (rounds 1-3 queued the float64 tile kernel
  // there, which is not the reference's arithmetic on data whose sums round): input outside the packed kernel's domain leaves
  // the call without a result, vwgpu_last_path() says VWGPU_PATH_REFUSED, and the caller repeats it with the option off.
  if (ctx->defer_exact) return VWGPU_OK;
  int flag = 0;
  VWGPU_HIP(ctx, hipMemcpyAsync(&flag, d_flag, sizeof flag, hipMemcpyDeviceToHost, ctx->stream));
  VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->last_flag = nullptr;
  ctx->measure_first = flag != 0;
  if (!flag) return VWGPU_OK;
  return calc_disparity_classified(ctx, cost_type, d_left, lw, lh, ls, d_right, rw, rh, rs, kx, ky, sx, sy, d_out, os);
}

int vwgpu_calc_disparity(vwgpu_ctx* ctx, int cost_type,
                         const float* left, int lw, int lh, ptrdiff_t ls,
                         const float* right, int rw, int rh, ptrdiff_t rs,
                         int kx, int ky, int sx, int sy, int32_t* out, ptrdiff_t os) {
  int rc = check_bm_args(ctx, cost_type, left, lw, lh, ls, right, rw, rh, rs, kx, ky, sx, sy, out, os);
  if (rc) return rc;
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  const int ow = lw - kx + 1, oh = lh - ky + 1;
  if (os == 0) os = ow;
  // Only the part of the right raster the search can reach is staged (Correlation.cc:356-359).
  const int rcw = lw + sx - 1, rch = lh + sy - 1;
  const size_t lb = vwgpu_align_up((size_t)lw * lh * sizeof(float), 256);
  const size_t rb = vwgpu_align_up((size_t)rcw * rch * sizeof(float), 256);
  const size_t ob = vwgpu_align_up((size_t)ow * oh * 3 * sizeof(int32_t), 256);
  rc = vwgpu_arena_reserve(ctx, &ctx->staging, lb + rb + ob);
  if (rc) return rc;
  char* base = static_cast<char*>(ctx->staging.base);
  float* d_l = reinterpret_cast<float*>(base);
  float* d_r = reinterpret_cast<float*>(base + lb);
  int32_t* d_o = reinterpret_cast<int32_t*>(base + lb + rb);
  VWGPU_HIP(ctx, hipMemcpy2DAsync(d_l, (size_t)lw * 4, left, (size_t)ls * 4, (size_t)lw * 4, lh, hipMemcpyHostToDevice, ctx->stream));
  VWGPU_HIP(ctx, hipMemcpy2DAsync(d_r, (size_t)rcw * 4, right, (size_t)rs * 4, (size_t)rcw * 4, rch, hipMemcpyHostToDevice, ctx->stream));
  rc = vwgpu_calc_disparity_dev(ctx, cost_type, d_l, lw, lh, lw, d_r, rcw, rch, rcw, kx, ky, sx, sy, d_o, ow);
  if (rc) return rc;
  VWGPU_HIP(ctx, hipMemcpy2DAsync(out, (size_t)os * 12, d_o, (size_t)ow * 12, (size_t)ow * 12, oh, hipMemcpyDeviceToHost, ctx->stream));
  VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VWGPU_OK;
}

// ---- fast_box_sum -------------------------------------------------------------------------------------------

static int check_box_args(vwgpu_ctx* ctx, const void* img, int w, int h, ptrdiff_t stride, int kx, int ky, const void* out, ptrdiff_t os) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  ctx->err.clear();
  if (!img || !out) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "fast_box_sum: null image pointer");
  if (kx < 1 || ky < 1 || kx % 2 != 1 || ky % 2 != 1)       // Algorithms.h:45-46, an always-on VW_ASSERT
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "fast_box_sum: Kernel input not sized with odd values.");
  if (w < kx || h < ky) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "fast_box_sum: Image is not big enough for kernel.");
  if (stride < w || (os != 0 && os < w - kx + 1)) return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "fast_box_sum: row stride smaller than row width");
  return VWGPU_OK;
}

int vwgpu_fast_box_sum_dev(vwgpu_ctx* ctx, const float* d_img, int w, int h, ptrdiff_t stride, int kx, int ky,
                           double* d_out, ptrdiff_t os) {
  if (stride == 0) stride = w;
  int rc = check_box_args(ctx, d_img, w, h, stride, kx, ky, d_out, os);
  if (rc) return rc;
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  const int ow = w - kx + 1, oh = h - ky + 1;
  if (os == 0 || os == ow) return vwgpu_launch_box_sum_exact(ctx, d_img, w, h, stride, kx, ky, d_out);
  // strided destination: through a dense scratch image
  rc = vwgpu_arena_reserve(ctx, &ctx->filt, (size_t)ow * oh * sizeof(double));
  if (rc) return rc;
  double* tmp = static_cast<double*>(ctx->filt.base);
  rc = vwgpu_launch_box_sum_exact(ctx, d_img, w, h, stride, kx, ky, tmp);
  if (rc) return rc;
  VWGPU_HIP(ctx, hipMemcpy2DAsync(d_out, (size_t)os * 8, tmp, (size_t)ow * 8, (size_t)ow * 8, oh, hipMemcpyDeviceToDevice, ctx->stream));
  return VWGPU_OK;
}

int vwgpu_fast_box_sum(vwgpu_ctx* ctx, const float* img, int w, int h, ptrdiff_t stride, int kx, int ky, double* out, ptrdiff_t os) {
  if (stride == 0) stride = w;
  int rc = check_box_args(ctx, img, w, h, stride, kx, ky, out, os);
  if (rc) return rc;
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  const int ow = w - kx + 1, oh = h - ky + 1;
  if (os == 0) os = ow;
  const size_t ib = vwgpu_align_up((size_t)w * h * sizeof(float), 256), ob = (size_t)ow * oh * sizeof(double);
  rc = vwgpu_arena_reserve(ctx, &ctx->staging, ib + ob);
  if (rc) return rc;
  float* d_i = static_cast<float*>(ctx->staging.base);
  double* d_o = reinterpret_cast<double*>(static_cast<char*>(ctx->staging.base) + ib);
  VWGPU_HIP(ctx, hipMemcpy2DAsync(d_i, (size_t)w * 4, img, (size_t)stride * 4, (size_t)w * 4, h, hipMemcpyHostToDevice, ctx->stream));
  rc = vwgpu_launch_box_sum_exact(ctx, d_i, w, h, w, kx, ky, d_o);
  if (rc) return rc;
  VWGPU_HIP(ctx, hipMemcpy2DAsync(out, (size_t)os * 8, d_o, (size_t)ow * 8, (size_t)ow * 8, oh, hipMemcpyDeviceToHost, ctx->stream));
  VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VWGPU_OK;
}

// ---- left/right consistency check -----------------------------------------------------------------------

int vwgpu_cross_corr_consistency_check_dev(vwgpu_ctx* ctx, int32_t* d_l2r, int lw, int lh, ptrdiff_t ls,
                                           const int32_t* d_r2l, int rw, int rh, ptrdiff_t rs, float thr) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  ctx->err.clear();
  if (!d_l2r || !d_r2l || lw < 0 || lh < 0 || rw < 0 || rh < 0)
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "cross_corr_consistency_check: bad image");
  // VW_DEBUG_ASSERT in the reference (Correlate.cc:1457-1460); always on here.
  if (!(thr >= 0.0f))
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "cross_corr_consistency_check: the threshold is less than 0.");
  if (ls == 0) ls = lw;
  if (rs == 0) rs = rw;
  if (lw == 0 || lh == 0) return VWGPU_OK;
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  return vwgpu_launch_lr_check(ctx, d_l2r, lw, lh, ls, d_r2l, rw, rh, rs, thr);
}

int vwgpu_cross_corr_consistency_check(vwgpu_ctx* ctx, int32_t* l2r, int lw, int lh, ptrdiff_t ls,
                                       const int32_t* r2l, int rw, int rh, ptrdiff_t rs, float thr) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  if (!l2r || !r2l || lw < 0 || lh < 0 || rw < 0 || rh < 0)
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "cross_corr_consistency_check: bad image");
  if (ls == 0) ls = lw;
  if (rs == 0) rs = rw;
  if (lw == 0 || lh == 0) return VWGPU_OK;
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  const size_t lb = vwgpu_align_up((size_t)lw * lh * 12, 256), rb = vwgpu_align_up((size_t)rw * rh * 12, 256);
  int rc = vwgpu_arena_reserve(ctx, &ctx->staging, lb + rb);
  if (rc) return rc;
  char* base = static_cast<char*>(ctx->staging.base);
  int32_t* d_l = reinterpret_cast<int32_t*>(base);
  int32_t* d_r = reinterpret_cast<int32_t*>(base + lb);
  VWGPU_HIP(ctx, hipMemcpy2DAsync(d_l, (size_t)lw * 12, l2r, (size_t)ls * 12, (size_t)lw * 12, lh, hipMemcpyHostToDevice, ctx->stream));
  if (rw > 0 && rh > 0)
    VWGPU_HIP(ctx, hipMemcpy2DAsync(d_r, (size_t)rw * 12, r2l, (size_t)rs * 12, (size_t)rw * 12, rh, hipMemcpyHostToDevice, ctx->stream));
  rc = vwgpu_cross_corr_consistency_check_dev(ctx, d_l, lw, lh, lw, d_r, rw, rh, rw, thr);
  if (rc) return rc;
  VWGPU_HIP(ctx, hipMemcpy2DAsync(l2r, (size_t)ls * 12, d_l, (size_t)lw * 12, (size_t)lw * 12, lh, hipMemcpyDeviceToHost, ctx->stream));
  VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VWGPU_OK;
}

int vwgpu_cross_corr_consistency_check_diff_dev(vwgpu_ctx* ctx, int32_t* d_l2r, int lw, int lh, ptrdiff_t ls,
                                                const int32_t* d_r2l, int rw, int rh, ptrdiff_t rs, float thr,
                                                float* d_diff, int dcols, int drows, ptrdiff_t dstride, int ulx, int uly) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  ctx->err.clear();
  if (!d_l2r || !d_r2l || lw < 0 || lh < 0 || rw < 0 || rh < 0)
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "cross_corr_consistency_check: bad image");
  if (!(thr >= 0.0f))
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "cross_corr_consistency_check: the threshold is less than 0.");
  if (dstride == 0) dstride = dcols;
  if (d_diff && (ulx < 0 || uly < 0 || ulx + lw > dcols || uly + lh > drows || dstride < dcols))
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "cross_corr_consistency_check: lr_disp_diff does not contain the checked region");
  if (ls == 0) ls = lw;
  if (rs == 0) rs = rw;
  if (lw == 0 || lh == 0) return VWGPU_OK;
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  return vwgpu_launch_lr_check_diff(ctx, d_l2r, lw, lh, ls, d_r2l, rw, rh, rs, thr, d_diff, dstride, ulx, uly);
}

int vwgpu_cross_corr_consistency_check_diff(vwgpu_ctx* ctx, int32_t* l2r, int lw, int lh, ptrdiff_t ls,
                                            const int32_t* r2l, int rw, int rh, ptrdiff_t rs, float thr,
                                            float* diff, int dcols, int drows, ptrdiff_t dstride, int ulx, int uly) {
  if (!ctx) return VWGPU_ERR_ARGUMENT;
  if (!l2r || !r2l || lw < 0 || lh < 0 || rw < 0 || rh < 0)
    return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "cross_corr_consistency_check: bad image");
  if (ls == 0) ls = lw;
  if (rs == 0) rs = rw;
  if (dstride == 0) dstride = dcols;
  if (lw == 0 || lh == 0) return VWGPU_OK;
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  const size_t lb = vwgpu_align_up((size_t)lw * lh * 12, 256), rb = vwgpu_align_up((size_t)rw * rh * 12, 256);
  const size_t db = diff ? vwgpu_align_up((size_t)dcols * drows * 8, 256) : 0;
  int rc = vwgpu_arena_reserve(ctx, &ctx->staging, lb + rb + db);
  if (rc) return rc;
  char* base = static_cast<char*>(ctx->staging.base);
  int32_t* d_l = reinterpret_cast<int32_t*>(base);
  int32_t* d_r = reinterpret_cast<int32_t*>(base + lb);
  float* d_d = reinterpret_cast<float*>(base + lb + rb);
  VWGPU_HIP(ctx, hipMemcpy2DAsync(d_l, (size_t)lw * 12, l2r, (size_t)ls * 12, (size_t)lw * 12, lh, hipMemcpyHostToDevice, ctx->stream));
  if (rw > 0 && rh > 0)
    VWGPU_HIP(ctx, hipMemcpy2DAsync(d_r, (size_t)rw * 12, r2l, (size_t)rs * 12, (size_t)rw * 12, rh, hipMemcpyHostToDevice, ctx->stream));
  if (diff) VWGPU_HIP(ctx, hipMemcpy2DAsync(d_d, (size_t)dcols * 8, diff, (size_t)dstride * 8, (size_t)dcols * 8, drows, hipMemcpyHostToDevice, ctx->stream));
  rc = vwgpu_cross_corr_consistency_check_diff_dev(ctx, d_l, lw, lh, lw, d_r, rw, rh, rw, thr, diff ? d_d : nullptr, dcols, drows, dcols, ulx, uly);
  if (rc) return rc;
  VWGPU_HIP(ctx, hipMemcpy2DAsync(l2r, (size_t)ls * 12, d_l, (size_t)lw * 12, (size_t)lw * 12, lh, hipMemcpyDeviceToHost, ctx->stream));
  if (diff) VWGPU_HIP(ctx, hipMemcpy2DAsync(diff, (size_t)dstride * 8, d_d, (size_t)dcols * 8, (size_t)dcols * 8, drows, hipMemcpyDeviceToHost, ctx->stream));
  VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return VWGPU_OK;
}

}  // extern "C"
