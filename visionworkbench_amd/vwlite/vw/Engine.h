// vw/Engine.h — libvwgpu.so contexts for the vwlite wrappers: one per (host thread x GPU), and the status -> exception mapping.
// The reference calls its entry points concurrently from tile threads (src/vw/Image/ImageIO.h:228-251); its errors are
// exceptions derived from vw::Exception (src/vw/Core/Exception.h:225-253).
//
// Multi-GPU: output tiles are independent units, so the tile threads are spread over the GPUs of the node —
// worker w of a block rasteriser runs on devices()[w % ndev], tiles are pulled dynamically.  The device list is
//   set_devices({...}) / set_device(d)   explicit,
//   VWGPU_DEVICES="0,2,5" | "all"        environment (VWGPU_DEVICE=<d> is the single-device form),
//   every visible HIP device             otherwise.
#ifndef VWLITE_ENGINE_H
#define VWLITE_ENGINE_H

#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "Core.h"
#include "vwgpu.h"

namespace vw {
namespace engine {
struct ThreadContexts {
  std::map<int, vwgpu_ctx*> by_device;
  ~ThreadContexts() { for (auto& kv : by_device) if (kv.second) vwgpu_destroy(kv.second); }
};
namespace detail {
inline std::mutex& device_mutex() { static std::mutex m; return m; }
inline std::vector<int>& device_list() { static std::vector<int> d; return d; }
inline void resolve_devices_locked() {
  std::vector<int>& d = device_list();
  if (!d.empty()) return;
  const char* list = std::getenv("VWGPU_DEVICES");
  const char* one = std::getenv("VWGPU_DEVICE");
  if (list && std::string(list) != "all") {
    std::string s(list);
    size_t pos = 0;
    while (pos < s.size()) {
      size_t next = s.find(',', pos);
      if (next == std::string::npos) next = s.size();
      if (next > pos) d.push_back(std::atoi(s.substr(pos, next - pos).c_str()));
      pos = next + 1;
    }
  } else if (!list && one) {
    d.push_back(std::atoi(one));
  } else {
    int n = 0;
    if (vwgpu_get_option(nullptr, VWGPU_OPT_DEVICE_COUNT, &n) != VWGPU_OK || n <= 0)
      vw_throw(LogicErr() << "no HIP device is visible (there is no CPU fallback)");
    for (int i = 0; i < n; ++i) d.push_back(i);
  }
  if (d.empty()) d.push_back(0);
}
}  // namespace detail

/// The GPUs the tile threads are spread over.
inline std::vector<int> devices() {
  std::lock_guard<std::mutex> lock(detail::device_mutex());
  detail::resolve_devices_locked();
  return detail::device_list();
}
inline void set_devices(std::vector<int> const& d) {
  VW_ASSERT(!d.empty(), ArgumentErr() << "engine::set_devices: empty device list");
  std::lock_guard<std::mutex> lock(detail::device_mutex());
  detail::device_list() = d;
}
inline void set_device(int device) { set_devices(std::vector<int>(1, device)); }
/// The GPU of the calling thread: devices()[thread_worker_index() % ndev].
inline int thread_device() {
  std::lock_guard<std::mutex> lock(detail::device_mutex());
  detail::resolve_devices_locked();
  std::vector<int> const& d = detail::device_list();
  const int w = thread_worker_index();
  return d[(size_t)(w < 0 ? 0 : w) % d.size()];
}
/// The calling thread's context on `device` (default: thread_device()), created on first use.
inline vwgpu_ctx* thread_context(int device = -1) {
  static thread_local ThreadContexts tc;
  if (device < 0) device = thread_device();
  vwgpu_ctx*& ctx = tc.by_device[device];
  if (!ctx) {
    if (vwgpu_abi_version() != VWGPU_ABI_VERSION)     // e.g. vwgpu_sgm_params grew between versions 1 and 2: never pass structs across a mismatch
      vw_throw(LogicErr() << "libvwgpu.so speaks ABI version " << vwgpu_abi_version() << ", these headers were written for " << VWGPU_ABI_VERSION);
    int rc = vwgpu_create(&ctx, device);
    if (rc != VWGPU_OK) {
      ctx = nullptr;
      vw_throw(LogicErr() << "vwgpu_create(device " << device << ") failed: " << vwgpu_strerror(rc) << " (no GPU; there is no CPU fallback)");
    }
  }
  return ctx;
}
inline void check(vwgpu_ctx* ctx, int rc) {
  if (rc == VWGPU_OK) return;
  std::string msg = vwgpu_last_error(ctx);
  if (msg.empty()) msg = vwgpu_strerror(rc);
  switch (rc) {
    case VWGPU_ERR_ARGUMENT: vw_throw(ArgumentErr() << msg);
    case VWGPU_ERR_NOIMPL: vw_throw(NoImplErr() << msg);
    default: vw_throw(LogicErr() << msg);
  }
}
}  // namespace engine
}  // namespace vw
#endif
