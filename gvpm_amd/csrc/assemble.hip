// Per-pixel assembly kernels (HBM-bound, elementwise / 5-point stencil).
//
//   finalize_iteration  normalisation by the emitted path count + APA running mean,
//                       gvpm/gvpm.cpp:1055-1069 (BRE), :868-875 (planes), :964-975 (beams)
//   film_kernel         throughput (gvpm.cpp:480-500), reusePrimal (:503-532) and
//                       computeGradient (:1205-1306) for an APA volume estimator
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace gvpm {

__global__ __launch_bounds__(256) void finalize_iteration_kernel(float *__restrict__ accum,
                                                                 const float *__restrict__ iter, size_t n, float it,
                                                                 float invPaths) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = iter[i] * invPaths;
  accum[i] = (accum[i] * (it - 1.f) + v) / it;
}

// accumulator slots: 0 mediumFlux, 1+i shifted[i], 5+i weighted[i]  (i = L,R,T,B)
__device__ __forceinline__ float A(const float *acc, int w, int x, int y, int k, int c) {
  return acc[((size_t)y * w + x) * 27 + k * 3 + c];
}

__global__ __launch_bounds__(256) void film_kernel(const float *__restrict__ acc, const float *__restrict__ emission,
                                                   int w, int h, float it, int reusePrimal, float invDiv,
                                                   float *thr, float *dx, float *dy) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)w * h * 3) return;
  const int c = (int)(i % 3);
  const int x = (int)((i / 3) % w), y = (int)(i / 3 / w);
  // invDiv = 1 for APA estimators, 1/m_totalEmittedVolume otherwise (gvpm.cpp:489-492,526-528)
  float v = A(acc, w, x, y, 0, c) * invDiv + (emission ? emission[i] / it : 0.f);
  if (reusePrimal) {
    float T = 0.f;
    if (x != w - 1) T += A(acc, w, x + 1, y, 1 + GVPM_LEFT, c);
    if (x != 0) T += A(acc, w, x - 1, y, 1 + GVPM_RIGHT, c);
    if (y != h - 1) T += A(acc, w, x, y + 1, 1 + GVPM_BOTTOM, c);
    if (y != 0) T += A(acc, w, x, y - 1, 1 + GVPM_TOP, c);
    T += A(acc, w, x, y, 5 + GVPM_BOTTOM, c) + A(acc, w, x, y, 5 + GVPM_TOP, c) + A(acc, w, x, y, 5 + GVPM_RIGHT, c) +
         A(acc, w, x, y, 5 + GVPM_LEFT, c);
    v = (T / 4.0f) * invDiv;
  }
  thr[i] = v;
  float gx = A(acc, w, x, y, 1 + GVPM_RIGHT, c) - A(acc, w, x, y, 5 + GVPM_RIGHT, c);
  if (x != w - 1) gx += A(acc, w, x + 1, y, 5 + GVPM_LEFT, c) - A(acc, w, x + 1, y, 1 + GVPM_LEFT, c);
  float gy = A(acc, w, x, y, 1 + GVPM_TOP, c) - A(acc, w, x, y, 5 + GVPM_TOP, c);
  if (y != h - 1) gy += A(acc, w, x, y + 1, 5 + GVPM_BOTTOM, c) - A(acc, w, x, y + 1, 1 + GVPM_BOTTOM, c);
  dx[i] = gx * invDiv;
  dy[i] = gy * invDiv;
}

// out = in * scale (the G-BRE accumulators hold the running SUM over iterations; the APA mean is sum / it)
__global__ __launch_bounds__(256) void scale_kernel(const float *in, float *out, size_t n, float scale) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] * scale;
}
void launch_scale(const float *in, float *out, size_t n, float scale, hipStream_t s) {
  hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n, scale);
}

void launch_finalize(float *accum, const float *iter, size_t n, int it, uint64_t nbPaths, hipStream_t s) {
  hipLaunchKernelGGL(finalize_iteration_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, accum, iter, n,
                     (float)it, 1.0f / (float)nbPaths);
}

void launch_film(const float *acc, const float *emission, int w, int h, int it, int reusePrimal, float invDiv,
                 float *thr, float *dx, float *dy, hipStream_t s) {
  const size_t n = (size_t)w * h * 3;
  hipLaunchKernelGGL(film_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, acc, emission, w, h, (float)it,
                     reusePrimal, invDiv, thr, dx, dy);
}

}  // namespace gvpm
