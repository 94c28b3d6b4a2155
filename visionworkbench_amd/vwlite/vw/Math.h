// vw/Math.h — Vector2i/Vector2f and the half-open integer box BBox2i with the semantics of
// src/vw/Math/BBox.tcc:82-197,268-290.
#ifndef VWLITE_MATH_H
#define VWLITE_MATH_H

#include <algorithm>
#include <ostream>

#include "Core.h"

namespace vw {

template <class T, int N>
class Vector {
  T m[N];
public:
  Vector() { for (int i = 0; i < N; ++i) m[i] = T(); }
  Vector(T a, T b) { static_assert(N == 2, "2-vector ctor"); m[0] = a; m[1] = b; }
  template <class U> Vector(Vector<U, N> const& o) { for (int i = 0; i < N; ++i) m[i] = T(o[i]); }
  T& operator[](int i) { return m[i]; }
  T const& operator[](int i) const { return m[i]; }
  T& x() { return m[0]; }  T const& x() const { return m[0]; }
  T& y() { return m[1]; }  T const& y() const { return m[1]; }
  Vector& operator+=(Vector const& o) { for (int i = 0; i < N; ++i) m[i] += o.m[i]; return *this; }
  Vector& operator-=(Vector const& o) { for (int i = 0; i < N; ++i) m[i] -= o.m[i]; return *this; }
  bool operator==(Vector const& o) const { for (int i = 0; i < N; ++i) if (m[i] != o.m[i]) return false; return true; }
  bool operator!=(Vector const& o) const { return !(*this == o); }
};
template <class T, int N> Vector<T, N> operator+(Vector<T, N> a, Vector<T, N> const& b) { return a += b; }
template <class T, int N> Vector<T, N> operator-(Vector<T, N> a, Vector<T, N> const& b) { return a -= b; }
template <class T, int N> Vector<T, N> operator*(Vector<T, N> a, T s) { for (int i = 0; i < N; ++i) a[i] *= s; return a; }
template <class T, int N> Vector<T, N> operator/(Vector<T, N> a, T s) { for (int i = 0; i < N; ++i) a[i] /= s; return a; }
template <class T, int N> T prod(Vector<T, N> const& v) { T p = 1; for (int i = 0; i < N; ++i) p *= v[i]; return p; }
template <class T, int N> std::ostream& operator<<(std::ostream& o, Vector<T, N> const& v) {
  o << "Vector" << N << "(";
  for (int i = 0; i < N; ++i) o << (i ? "," : "") << v[i];
  return o << ")";
}
typedef Vector<int32, 2> Vector2i;
typedef Vector<float, 2> Vector2f;
typedef Vector<double, 2> Vector2;

// Half-open box [min, max).  An empty box reports zero width/height/area (BBox.tcc:156-174).
class BBox2i {
  Vector2i m_min, m_max;
public:
  // default = empty, with the reference's +-(INT32_MAX - 1) corners (src/vw/Math/BBox.tcc:37-45)
  BBox2i() : m_min(0x7ffffffe, 0x7ffffffe), m_max(-0x7ffffffe, -0x7ffffffe) {}
  BBox2i(Vector2i const& mn, Vector2i const& mx) : m_min(mn), m_max(mx) {}
  BBox2i(int32 x, int32 y, int32 w, int32 h) : m_min(x, y), m_max(x + w, y + h) {}
  Vector2i& min() { return m_min; }  Vector2i const& min() const { return m_min; }
  Vector2i& max() { return m_max; }  Vector2i const& max() const { return m_max; }
  bool empty() const { return m_min[0] >= m_max[0] || m_min[1] >= m_max[1]; }
  int32 width() const { return empty() ? 0 : m_max[0] - m_min[0]; }
  int32 height() const { return empty() ? 0 : m_max[1] - m_min[1]; }
  int64 area() const { return (int64)width() * height(); }
  Vector2i size() const { return m_max - m_min; }
  void grow(Vector2i const& p) {            // include a point: max becomes p itself (BBox.tcc:82-92)
    for (int i = 0; i < 2; ++i) { if (p[i] > m_max[i]) m_max[i] = p[i]; if (p[i] < m_min[i]) m_min[i] = p[i]; }
  }
  void grow(BBox2i const& b) { if (!b.empty()) { grow(b.min()); grow(b.max()); } }
  void crop(BBox2i const& b) {              // intersect
    for (int i = 0; i < 2; ++i) { m_min[i] = std::max(m_min[i], b.m_min[i]); m_max[i] = std::min(m_max[i], b.m_max[i]); }
  }
  void expand(int32 n) { if (empty()) return; m_min -= Vector2i(n, n); m_max += Vector2i(n, n); }
  void contract(int32 n) { expand(-n); }
  bool contains(Vector2i const& p) const { return p[0] >= m_min[0] && p[0] < m_max[0] && p[1] >= m_min[1] && p[1] < m_max[1]; }
  bool contains(BBox2i const& b) const {
    return b.m_min[0] >= m_min[0] && b.m_min[1] >= m_min[1] && b.m_max[0] <= m_max[0] && b.m_max[1] <= m_max[1];
  }
  BBox2i& operator+=(Vector2i const& v) { if (!empty()) { m_min += v; m_max += v; } return *this; }
  BBox2i& operator-=(Vector2i const& v) { if (!empty()) { m_min -= v; m_max -= v; } return *this; }
  BBox2i& operator*=(int32 s) { if (!empty()) { m_min = m_min * s; m_max = m_max * s; } return *this; }
  BBox2i& operator/=(int32 s) { if (!empty()) { m_min = m_min / s; m_max = m_max / s; } return *this; }
  bool operator==(BBox2i const& o) const { return m_min == o.m_min && m_max == o.m_max; }
};
inline BBox2i operator+(BBox2i b, Vector2i const& v) { return b += v; }
inline BBox2i operator-(BBox2i b, Vector2i const& v) { return b -= v; }
inline BBox2i operator*(BBox2i b, int32 s) { return b *= s; }
inline BBox2i operator/(BBox2i b, int32 s) { return b /= s; }
inline std::ostream& operator<<(std::ostream& o, BBox2i const& b) {
  return o << "(" << b.min() << "-" << b.max() << ")";
}

}  // namespace vw
#endif
