// fp64 helpers shared by the kernels that transcribe the reference literally (G-Beams, G-Planes).
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"
#include "vec.h"

namespace gvpm {

struct RayD {
  d3 o, d;
  double mint, maxt;
};
__device__ __forceinline__ d3 at(const RayD &r, double t) { return r.o + r.d * t; }
__device__ __forceinline__ d3 crossd(d3 a, d3 b) {
  return mkd(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ d3 operator*(double s, d3 a) { return a * s; }

// HomogeneousMedium::eval over [0, dist], homogeneous.cpp:432-513 (equal sigma_t per channel)
struct MRecD {
  double tr, pdfSuccess, pdfFailure;
};
__device__ __forceinline__ MRecD mediumEvalD(const MediumDev &m, double dist) {
  MRecD r;
  const double st = (double)m.sigmaT[0], msw = (double)m.msw;
  double e = exp(-st * dist);
  r.pdfSuccess = st * e * msw;
  r.pdfFailure = e * msw + (1.0 - msw);
  if (e < 1e-20) e = 0.0;
  r.tr = e;
  return r;
}

__device__ __forceinline__ double phaseD(double g, d3 wi, d3 wo) {
  const double INV_FOURPI = 0.07957747154594766788;
  if (g == 0.0) return INV_FOURPI;
  const double temp = 1.0 + g * g + 2.0 * g * dot(wi, wo);
  return INV_FOURPI * (1.0 - g * g) / (temp * sqrt(temp));
}

}  // namespace gvpm
