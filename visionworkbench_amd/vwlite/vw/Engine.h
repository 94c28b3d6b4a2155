// vw/Engine.h — one libvwgpu.so context per (host thread x GPU) and the status -> exception mapping shared by the
// vwlite wrappers.  The reference calls its entry points concurrently from tile threads
// (src/vw/Image/ImageIO.h:228-251); its errors are exceptions derived from vw::Exception
// (src/vw/Core/Exception.h:225-253).
#ifndef VWLITE_ENGINE_H
#define VWLITE_ENGINE_H

#include <cstdlib>
#include <string>

#include "Core.h"
#include "vwgpu.h"

namespace vw {
namespace engine {
struct ThreadContext {
  vwgpu_ctx* ctx = nullptr;
  ~ThreadContext() { if (ctx) vwgpu_destroy(ctx); }
};
inline vwgpu_ctx* thread_context() {
  static thread_local ThreadContext tc;
  if (!tc.ctx) {
    const char* dev = std::getenv("VWGPU_DEVICE");
    int rc = vwgpu_create(&tc.ctx, dev ? std::atoi(dev) : 0);
    if (rc != VWGPU_OK)
      vw_throw(LogicErr() << "vwgpu_create failed: " << vwgpu_strerror(rc) << " (no GPU; there is no CPU fallback)");
  }
  return tc.ctx;
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
