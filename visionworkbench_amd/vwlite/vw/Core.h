// vw/Core.h — the slice of Vision Workbench's Core module that the stereo hot path touches, boost-free.
// Exceptions mirror src/vw/Core/Exception.h:225-253 (vw::Exception hierarchy, vw_throw, VW_ASSERT :275-289).
#ifndef VWLITE_CORE_H
#define VWLITE_CORE_H

#include <cstdint>
#include <sstream>
#include <stdexcept>
#include <string>

namespace vw {

typedef int8_t int8;   typedef uint8_t uint8;
typedef int16_t int16; typedef uint16_t uint16;
typedef int32_t int32; typedef uint32_t uint32;
typedef int64_t int64; typedef uint64_t uint64;
typedef float float32; typedef double float64;

// vw::Exception: streamable message, what() returns it (Exception.h:111-160).
class Exception : public std::exception {
  std::string m_desc;
public:
  Exception() {}
  explicit Exception(std::string const& s) : m_desc(s) {}
  virtual ~Exception() throw() {}
  const char* what() const throw() override { return m_desc.c_str(); }
  std::string desc() const { return m_desc; }
  template <class T> Exception& operator<<(T const& t) { std::ostringstream o; o << t; m_desc += o.str(); return *this; }
  virtual std::string name() const { return "Exception"; }
};

#define VWLITE_EXCEPTION(Name, Parent)                                                     \
  class Name : public Parent {                                                             \
  public:                                                                                  \
    Name() {}                                                                              \
    template <class T> Name& operator<<(T const& t) { Parent::operator<<(t); return *this; } \
    std::string name() const override { return #Name; }                                    \
  }

VWLITE_EXCEPTION(ArgumentErr, Exception);
VWLITE_EXCEPTION(LogicErr, Exception);
VWLITE_EXCEPTION(InputErr, Exception);
VWLITE_EXCEPTION(IOErr, Exception);
VWLITE_EXCEPTION(MathErr, Exception);
VWLITE_EXCEPTION(NoImplErr, Exception);

template <class ExcT> inline void vw_throw(ExcT const& e) { throw e; }

namespace engine {
// Worker threads of the block rasterisers (block_rasterize, block_write_image) announce their pool index here; the engine
// wrappers (vw/Engine.h) bind worker w to GPU devices[w % ndev] with one context per (thread x GPU) — the reference's tile
// threads (src/vw/Image/ImageIO.h:228-251) spread over the GPUs of the node.
inline int& thread_worker_index() { static thread_local int idx = 0; return idx; }
}  // namespace engine

#define VW_ASSERT(cond, excep) do { if (!(cond)) vw::vw_throw(excep); } while (0)
// VW_DEBUG_ASSERT is compiled out in the reference's release builds (Exception.h:286-289); the engine
// re-checks the same conditions behind the C ABI, so they are always on here.
#define VW_DEBUG_ASSERT(cond, excep) VW_ASSERT(cond, excep)

}  // namespace vw
#endif
