// halo.hip — halo rows of a row-sharded source image, fetched from the neighbouring ranks with RCCL send / recv over xGMI.
//
// Output tiles are independent units in the reference (each PyramidCorrelationView::prerasterize(bbox) works from its own padded
// crop, src/vw/Stereo/CorrelationView.cc:89-97; SGM tiles overlap by collar_size, CorrelationView.h:123-133), so the path shards
// with no collective on the data path.  The one real exchange: when the SOURCE pair itself is sharded by rows over the HBMs of a
// node (one process per GPU), a rank's tiles need the rows of its strip plus half_kernel * 2^levels + search (+ collar) rows above
// and below, which live on the neighbouring ranks.  One point-to-point transfer per neighbour that owns needed rows, all of a
// call inside one ncclGroupStart / ncclGroupEnd on the context's stream — xGMI is point-to-point, a ring or tree collective
// would only add hops.  bench.py --gpus N drives THIS exchange (partition.EngineComm; the unique id travels over the process group
// that launched the ranks), C++ hosts through vw::engine::StripComm; the torch.distributed mirror in visionworkbench_amd/partition.py
// has the same plan, runs on gloo in the CPU tests and is bench.py's stated fallback.
//
// librccl.so is opened at run time (dlopen): a single-GPU installation does not need it, and libvwgpu.so does not link it.
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <vector>

#include "vwgpu_internal.h"

namespace {

struct NcclId { char internal[128]; };                 // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* NcclComm;
struct Rccl {
  void* so = nullptr;
  int (*GetUniqueId)(NcclId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*Send)(const void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {getenv("VWGPU_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      r.so = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.so) break;
    }
    if (!r.so) return;
#define VW_SYM(field, name) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.so, name))
    VW_SYM(GetUniqueId, "ncclGetUniqueId"); VW_SYM(CommInitRank, "ncclCommInitRank"); VW_SYM(CommDestroy, "ncclCommDestroy");
    VW_SYM(Send, "ncclSend"); VW_SYM(Recv, "ncclRecv"); VW_SYM(GroupStart, "ncclGroupStart"); VW_SYM(GroupEnd, "ncclGroupEnd");
    VW_SYM(GetErrorString, "ncclGetErrorString"); VW_SYM(AllGather, "ncclAllGather");
#undef VW_SYM
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.Send && r.Recv && r.GroupStart && r.GroupEnd && r.AllGather;
  });
  return r;
}

void row_strip(int rank, int world, int rows, int* a, int* b) {      // partition.row_strip
  *a = (int)((long long)rank * rows / world);
  *b = (int)((long long)(rank + 1) * rows / world);
}

}  // namespace

struct vwgpu_comm {
  NcclComm comm = nullptr;
  int rank = 0, world = 1;
};

extern "C" {

int vwgpu_halo_plan(int rank, int world, int rows_total, int halo_above, int halo_below, int* owned_a, int* owned_b, int* need_a,
                    int* need_b) {
  if (world < 1 || rank < 0 || rank >= world || rows_total < 0 || halo_above < 0 || halo_below < 0) return VWGPU_ERR_ARGUMENT;
  int a, b;
  row_strip(rank, world, rows_total, &a, &b);
  if (owned_a) *owned_a = a;
  if (owned_b) *owned_b = b;
  if (need_a) *need_a = std::max(0, a - halo_above);
  if (need_b) *need_b = std::min(rows_total, b + halo_below);
  return VWGPU_OK;
}

/* The verdict every rank reaches from the SAME table of gathered request headers (4 values per rank: rows_total, halo_above,
 * halo_below, bytes per row): 1 = all ranks asked for the same image and halos; 0 = they did not, and *rank_a / *rank_b name the
 * first pair that differs.  Pure host arithmetic (tests/test_host_logic.py drives it with three ranks). */
int vwgpu_halo_headers_agree(const long long* headers, int world, int* rank_a, int* rank_b) {
  if (!headers || world < 1) return 0;
  for (int p = 1; p < world; ++p)
    if (memcmp(headers + 4 * (size_t)p, headers, 4 * sizeof(long long)) != 0) {
      if (rank_a) *rank_a = 0;
      if (rank_b) *rank_b = p;
      return 0;
    }
  return 1;
}

int vwgpu_comm_unique_id(void* id128) {
  if (!id128) return VWGPU_ERR_ARGUMENT;
  Rccl& r = rccl();
  if (!r.ok) return VWGPU_ERR_NOIMPL;                                  // librccl.so not found
  NcclId id;
  if (r.GetUniqueId(&id) != 0) return VWGPU_ERR_HIP;
  memcpy(id128, id.internal, sizeof id.internal);
  return VWGPU_OK;
}

int vwgpu_comm_create(vwgpu_ctx* ctx, const void* id128, int rank, int world, vwgpu_comm** out) {
  if (!ctx || !id128 || !out || world < 1 || rank < 0 || rank >= world) return VWGPU_ERR_ARGUMENT;
  Rccl& r = rccl();
  if (!r.ok) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "vwgpu_comm_create: librccl.so could not be opened (set VWGPU_RCCL_LIBRARY)");
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  NcclId id;
  memcpy(id.internal, id128, sizeof id.internal);
  vwgpu_comm* c = new vwgpu_comm;
  c->rank = rank; c->world = world;
  const int rc = r.CommInitRank(&c->comm, world, id, rank);
  if (rc != 0) {
    delete c;
    return vwgpu_fail(ctx, VWGPU_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, r.GetErrorString ? r.GetErrorString(rc) : "?");
  }
  *out = c;
  return VWGPU_OK;
}

int vwgpu_comm_destroy(vwgpu_comm* comm) {
  if (!comm) return VWGPU_OK;
  Rccl& r = rccl();
  if (r.ok && comm->comm) r.CommDestroy(comm->comm);
  delete comm;
  return VWGPU_OK;
}

int vwgpu_fetch_strip_window_dev(vwgpu_ctx* ctx, vwgpu_comm* comm, const void* d_owned, int cols, int elem_bytes, int rows_total,
                                 int halo_above, int halo_below, void* d_window, int* first_row) {
  if (!ctx || !comm || !d_owned || !d_window || cols <= 0 || elem_bytes <= 0) return VWGPU_ERR_ARGUMENT;
  Rccl& r = rccl();
  if (!r.ok) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "vwgpu_fetch_strip_window_dev: librccl.so could not be opened");
  const int rank = comm->rank, world = comm->world;
  int a, b, na, nb;
  int rc = vwgpu_halo_plan(rank, world, rows_total, halo_above, halo_below, &a, &b, &na, &nb);
  if (rc) return vwgpu_fail(ctx, rc, "vwgpu_fetch_strip_window_dev: bad strip request");
  const size_t rowb = (size_t)cols * elem_bytes;
  char* win = static_cast<char*>(d_window);
  const char* own = static_cast<const char*>(d_owned);
  VWGPU_HIP(ctx, hipSetDevice(ctx->device));
  // Every rank must describe the same image and halos, or the byte counts of a send and its receive differ and a rank waits
  // for ever.  The decision is COLLECTIVE and does not depend on the arguments it checks: every rank contributes its 32-byte
  // request to one all-gather over the whole communicator (a peer set computed from a rank's own halos differs between ranks that
  // disagree — rank A posts nothing while rank B waits for it; and a pairwise check lets a third rank, whose header matches B's,
  // walk into the data exchange with a B that has already returned: round-3 findings).  All ranks then hold the same table, reach
  // the same verdict (vwgpu_halo_headers_agree) and either all enter the data exchange or all return the error.
  // (One small host round trip per fetch: the call is synchronous up to here; the data exchange below is queued.)
  if (world > 1) {
    rc = vwgpu_arena_reserve(ctx, &ctx->misc, 4096);
    if (rc) return rc;
    long long* hdr = reinterpret_cast<long long*>(static_cast<char*>(ctx->misc.base) + 1024);     // [0..3] mine, [4 + 4 p ..] rank p's
    if ((size_t)(4 + 4 * world) * sizeof(long long) > 3072) return vwgpu_fail(ctx, VWGPU_ERR_NOIMPL, "vwgpu_fetch_strip_window_dev: more than 95 ranks");
    const long long mine[4] = {rows_total, halo_above, halo_below, (long long)cols * elem_bytes};
    VWGPU_HIP(ctx, hipMemcpyAsync(hdr, mine, sizeof mine, hipMemcpyHostToDevice, ctx->stream));
    const int hrc = r.AllGather(hdr, hdr + 4, sizeof mine, /*ncclChar*/ 0, comm->comm, ctx->stream);
    if (hrc != 0) return vwgpu_fail(ctx, VWGPU_ERR_HIP, "RCCL halo header all-gather failed: %s", r.GetErrorString ? r.GetErrorString(hrc) : "?");
    std::vector<long long> got((size_t)4 * world);
    VWGPU_HIP(ctx, hipMemcpyAsync(got.data(), hdr + 4, got.size() * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
    VWGPU_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int ra = 0, rb = 0;
    if (!vwgpu_halo_headers_agree(got.data(), world, &ra, &rb))
      return vwgpu_fail(ctx, VWGPU_ERR_ARGUMENT, "vwgpu_fetch_strip_window_dev: rank %d asked for (rows %lld, halos %lld / %lld, %lld bytes per row), rank %d for "
                        "(%lld, %lld / %lld, %lld): every rank must pass the same image and halos (this is rank %d; no rank exchanged data)",
                        rb, got[(size_t)4 * rb], got[(size_t)4 * rb + 1], got[(size_t)4 * rb + 2], got[(size_t)4 * rb + 3],
                        ra, got[(size_t)4 * ra], got[(size_t)4 * ra + 1], got[(size_t)4 * ra + 2], got[(size_t)4 * ra + 3], rank);
  }
  // my own rows
  if (b > a) VWGPU_HIP(ctx, hipMemcpyAsync(win + (size_t)(a - na) * rowb, own, (size_t)(b - a) * rowb, hipMemcpyDeviceToDevice, ctx->stream));
  int nrc = r.GroupStart();
  for (int p = 0; p < world && nrc == 0; ++p) {
    if (p == rank) continue;
    int pa, pb, pna, pnb;
    vwgpu_halo_plan(p, world, rows_total, halo_above, halo_below, &pa, &pb, &pna, &pnb);
    const int lo = std::max(na, pa), hi = std::min(nb, pb);            // rows of rank p that I need
    if (lo < hi) nrc = r.Recv(win + (size_t)(lo - na) * rowb, (size_t)(hi - lo) * rowb, /*ncclChar*/ 0, p, comm->comm, ctx->stream);
    const int slo = std::max(pna, a), shi = std::min(pnb, b);          // rows of mine that rank p needs
    if (nrc == 0 && slo < shi) nrc = r.Send(own + (size_t)(slo - a) * rowb, (size_t)(shi - slo) * rowb, 0, p, comm->comm, ctx->stream);
  }
  const int erc = r.GroupEnd();
  if (nrc == 0) nrc = erc;
  if (nrc != 0) return vwgpu_fail(ctx, VWGPU_ERR_HIP, "RCCL halo exchange failed: %s", r.GetErrorString ? r.GetErrorString(nrc) : "?");
  if (first_row) *first_row = na;
  return VWGPU_OK;
}

}  // extern "C"
