// vw/Halo.h — halo rows of a row-sharded source image for one-process-per-GPU hosts (RCCL point-to-point over xGMI through
// libvwgpu.so's vwgpu_comm_* / vwgpu_fetch_strip_window_dev, csrc/halo.hip).
//
// The reference has no distributed mode: its tiles are independent units pulled by a thread pool (src/vw/Image/ImageIO.h:228-251),
// each padded by half_kernel * 2^levels (+ search) (src/vw/Stereo/CorrelationView.cc:89-97) and, for SGM, by the collar
// (CorrelationView.h:123-133).  A node-wide run therefore gives every rank a strip of tile rows; when the source pair is sharded
// the same way, the rows a rank's tiles touch beyond its strip are fetched once, up front, with this class.
#ifndef VWLITE_HALO_H
#define VWLITE_HALO_H

#include <cstring>
#include <vector>

#include "Engine.h"

namespace vw {
namespace engine {

/// Rows [owned_a, owned_b) live on `rank`; its tiles read [need_a, need_b).
struct StripPlan { int owned_a, owned_b, need_a, need_b; };
inline StripPlan strip_plan(int rank, int world, int rows_total, int halo_above, int halo_below) {
  StripPlan p{};
  if (vwgpu_halo_plan(rank, world, rows_total, halo_above, halo_below, &p.owned_a, &p.owned_b, &p.need_a, &p.need_b) != VWGPU_OK)
    vw_throw(ArgumentErr() << "engine::strip_plan: bad strip request");
  return p;
}
/// Rows above / below a tile that PyramidCorrelationView::prerasterize can touch (the window vwgpu_pyramid_correlate stages).
inline void pyramid_halo_rows(int kernel_y, int max_pyramid_levels, int search_min_y, int search_max_y, int collar, int& above, int& below) {
  const int lv = max_pyramid_levels < 0 ? 0 : (max_pyramid_levels > 12 ? 12 : max_pyramid_levels);
  const int sdy = search_max_y > search_min_y ? search_max_y - search_min_y : 0;
  const int pad = (kernel_y / 2) * (1 << lv) + 2 * sdy + 8 + collar;
  above = pad - (search_min_y < 0 ? search_min_y : 0);
  below = pad + (search_max_y > 0 ? search_max_y : 0);
}

/// One RCCL communicator per process (= per GPU).  `unique_id()` is called on one rank; the 128 bytes reach the others through
/// whatever the host application uses to start its ranks (a file, MPI, a key-value store).
class StripComm {
  vwgpu_comm* m_comm = nullptr;
  vwgpu_ctx* m_ctx = nullptr;
  int m_rank = 0, m_world = 1;
public:
  static std::vector<char> unique_id() {
    std::vector<char> id(VWGPU_COMM_ID_BYTES);
    if (vwgpu_comm_unique_id(id.data()) != VWGPU_OK) vw_throw(LogicErr() << "engine::StripComm: librccl.so is not available");
    return id;
  }
  /// device < 0: devices()[rank % ndev] — one process per GPU, rank r on the r-th device of the list (NOT the calling thread's
  /// default device, which is the first one on every rank's main thread: RCCL refuses two ranks on one GPU).
  StripComm(std::vector<char> const& id, int rank, int world, int device = -1) : m_rank(rank), m_world(world) {
    VW_ASSERT((int)id.size() == VWGPU_COMM_ID_BYTES, ArgumentErr() << "engine::StripComm: the unique id has 128 bytes");
    VW_ASSERT(world >= 1 && rank >= 0 && rank < world, ArgumentErr() << "engine::StripComm: rank " << rank << " of " << world);
    if (device < 0) { std::vector<int> d = devices(); device = d[(size_t)rank % d.size()]; }
    m_ctx = thread_context(device);
    check(m_ctx, vwgpu_comm_create(m_ctx, id.data(), rank, world, &m_comm));
  }
  ~StripComm() { if (m_comm) vwgpu_comm_destroy(m_comm); }
  StripComm(StripComm const&) = delete;
  StripComm& operator=(StripComm const&) = delete;
  int rank() const { return m_rank; }
  int world() const { return m_world; }
  /// d_owned: this rank's rows, contiguous, in device memory; d_window: room for (need_b - need_a) x cols elements.
  /// Every rank calls this with the SAME (cols, elem_bytes, rows_total, halo_above, halo_below); the engine gathers the requests of
  /// ALL ranks first (one 32-byte all-gather and one host round trip) and throws ArgumentErr on every rank if any two differ — no
  /// rank moves rows then.  The row exchange itself is queued on the calling thread's context stream; returns the first row of the window.
  int fetch_strip_window(const void* d_owned, int cols, int elem_bytes, int rows_total, int halo_above, int halo_below, void* d_window) {
    int first = 0;
    check(m_ctx, vwgpu_fetch_strip_window_dev(m_ctx, m_comm, d_owned, cols, elem_bytes, rows_total, halo_above, halo_below, d_window, &first));
    return first;
  }
};

}  // namespace engine
}  // namespace vw
#endif
