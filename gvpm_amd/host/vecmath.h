// Minimal 3-vector helpers for the host side (double precision; everything is
// rounded to fp32 when it crosses the C ABI).
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GVPM_HD __host__ __device__
#else
#define GVPM_HD
#endif

namespace gvpm {

struct V3 {
  double x, y, z;
  GVPM_HD V3() : x(0), y(0), z(0) {}
  GVPM_HD V3(double a, double b, double c) : x(a), y(b), z(c) {}
  GVPM_HD explicit V3(double a) : x(a), y(a), z(a) {}
  GVPM_HD double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
GVPM_HD inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
GVPM_HD inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
GVPM_HD inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
GVPM_HD inline V3 operator*(V3 a, double s) { return V3(a.x * s, a.y * s, a.z * s); }
GVPM_HD inline V3 operator*(double s, V3 a) { return a * s; }
GVPM_HD inline V3 operator*(V3 a, V3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
GVPM_HD inline V3 operator/(V3 a, double s) { return V3(a.x / s, a.y / s, a.z / s); }
GVPM_HD inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
GVPM_HD inline V3 cross(V3 a, V3 b) {
  return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
GVPM_HD inline double length(V3 a) { return std::sqrt(dot(a, a)); }
GVPM_HD inline V3 normalize(V3 a) { return a / length(a); }
GVPM_HD inline double maxc(V3 a) { return std::fmax(a.x, std::fmax(a.y, a.z)); }

// Orthonormal basis around n (Mitsuba coordinateSystem, src/libcore/util.cpp:487-505)
GVPM_HD inline void coordinateSystem(const V3 &a, V3 &b, V3 &c) {
  if (std::fabs(a.x) > std::fabs(a.y)) {
    double invLen = 1.0 / std::sqrt(a.x * a.x + a.z * a.z);
    c = V3(a.z * invLen, 0.0, -a.x * invLen);
  } else {
    double invLen = 1.0 / std::sqrt(a.y * a.y + a.z * a.z);
    c = V3(0.0, a.z * invLen, -a.y * invLen);
  }
  b = cross(c, a);
}

}  // namespace gvpm
