// Small device vector helpers (fp32 radiometry, fp64 geometry).
#pragma once
#include <hip/hip_runtime.h>

namespace gvpm {

struct f3 {
  float x, y, z;
};
struct d3 {
  double x, y, z;
};

__device__ __forceinline__ f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
__device__ __forceinline__ f3 mk3(float s) { return f3{s, s, s}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ f3 operator*(float s, f3 a) { return f3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
  return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ float maxc(f3 a) { return fmaxf(a.x, fmaxf(a.y, a.z)); }
__device__ __forceinline__ float comp(f3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }

__device__ __forceinline__ d3 mkd(double x, double y, double z) { return d3{x, y, z}; }
__device__ __forceinline__ d3 tod(f3 a) { return d3{(double)a.x, (double)a.y, (double)a.z}; }
__device__ __forceinline__ f3 tof(d3 a) { return f3{(float)a.x, (float)a.y, (float)a.z}; }
__device__ __forceinline__ d3 operator+(d3 a, d3 b) { return d3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return d3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ d3 operator-(d3 a) { return d3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ d3 operator*(d3 a, double s) { return d3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ d3 operator/(d3 a, double s) { return d3{a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double len2(d3 a) { return dot(a, a); }

// 64-lane reductions (wave = 64 on gfx950)
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// single-instruction reciprocal / square root (1 ulp; the parity bar is set against the fp64 oracle, so
// the ~10-instruction IEEE division and the denormal-safe sqrt expansions buy nothing here)
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float frsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }

// inclusive prefix sum over the wave
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t n = __shfl_up(v, o, 64);
    if (lane >= o) v += n;
  }
  return v;
}

}  // namespace gvpm
