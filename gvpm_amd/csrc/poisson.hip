// Screened-Poisson reconstruction for gfx950 (SURVEY section 8 "next" row f1), hand-written HIP.
//
// Replaces poisson::Solver as gvpm.cpp:610-690 uses it (presets "L2D" and "L1D"):
//   Solver::setupBackend / solveIndirect / exportImagesMTS    poisson_solver/Solver.cpp:297-341, 376-497, 560-581
//   Backend::calc_Px, calc_axpy, calc_w2, calc_PTW2x, calc_Ax_xAx, calc_r_rz, calc_x_p
//                                                             poisson_solver/Backend.cpp:154-384
// x = argmin L1 or L2 (b - P x), b = (alpha * throughput, dx, dy), by iteratively reweighted least squares
// around a conjugate-gradient solve of the normal equations (P' W^2 P) x = P' W^2 b.
//
// Layout: the reference's own (RGB triplets, pixel-major; b / e / w2 as three stacked images).  All vectors
// are image sized (9.4 MB at 512^2 for the largest), i.e. L2 / Infinity-Cache resident: the solve is bound
// by launch latency and by the two global reductions per CG iteration, not by HBM.  One CG iteration is five
// small kernels with no host involvement; a whole CG run (cgIterMax iterations) is captured once into a HIP
// graph and replayed per IRLS iteration.  Every thread owns one float (pixel, channel): all accesses are
// coalesced; the two dot products are reduced per block in fp64 and finished by a one-block kernel in a fixed
// order, so the result is deterministic (the reference's OpenMP backend is not).
#include <float.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "../../include/gvpm_hip.h"

namespace gvpm {

constexpr int PB = 256;      // threads per block
constexpr int PGRID = 512;   // blocks of the grid-stride kernels (= partial sums per reduction)

struct PoissonBufs {
  float *b, *e, *w2, *x, *r, *p, *Ap;  // b, e: 9n floats; w2: 3n; the rest 3n floats (n pixels)
  double *part;                        // PGRID * 3 partial sums
  float *pAp, *rz, *rz2, *w2coef;      // 3 floats each (w2coef: 1)
  int W, H;
  float alpha;
};

__device__ __forceinline__ double blockSum(double v, double *sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < PB / 64; ++w) t += sh[w];
  return t;
}

// b = (alpha * throughput, dx, dy), x = throughput (Solver.cpp:325-340)
__global__ __launch_bounds__(PB) void poisson_setup_kernel(PoissonBufs q, const float *dx, const float *dy,
                                                           const float *tp) {
  const size_t n3 = (size_t)q.W * q.H * 3;
  for (size_t i = blockIdx.x * (size_t)PB + threadIdx.x; i < n3; i += (size_t)gridDim.x * PB) {
    const float t = tp ? tp[i] : 0.f;
    q.b[i] = t * q.alpha;
    q.b[n3 + i] = dx[i];
    q.b[2 * n3 + i] = dy[i];
    q.x[i] = t;
  }
}

// e = b - P x (calc_Px + calc_axpy with a = -1)
__global__ __launch_bounds__(PB) void poisson_residual_kernel(PoissonBufs q) {
  const size_t n3 = (size_t)q.W * q.H * 3;
  const size_t rowF = (size_t)q.W * 3;
  for (size_t i = blockIdx.x * (size_t)PB + threadIdx.x; i < n3; i += (size_t)gridDim.x * PB) {
    const size_t pix = i / 3;
    const int xx = (int)(pix % q.W), yy = (int)(pix / q.W);
    const float xi = q.x[i];
    const float p0 = xi * q.alpha;
    const float p1 = xx != q.W - 1 ? q.x[i + 3] - xi : 0.f;
    const float p2 = yy != q.H - 1 ? q.x[i + rowF] - xi : 0.f;
    q.e[i] = -1.f * p0 + q.b[i];
    q.e[n3 + i] = -1.f * p1 + q.b[n3 + i];
    q.e[2 * n3 + i] = -1.f * p2 + q.b[2 * n3 + i];
  }
}

__global__ __launch_bounds__(PB) void poisson_fill_kernel(float *v, size_t n, float a) {
  for (size_t i = blockIdx.x * (size_t)PB + threadIdx.x; i < n; i += (size_t)gridDim.x * PB) v[i] = a;
}

// w2 = 1 / (|e| + reg) per RGB triplet of the stacked residual, and its sum (calc_w2, first loop)
__global__ __launch_bounds__(PB) void poisson_w2_kernel(PoissonBufs q, float reg) {
  __shared__ double sh[PB / 64];
  const size_t m = (size_t)q.W * q.H * 3;  // triplets in e
  double acc = 0.0;
  for (size_t i = blockIdx.x * (size_t)PB + threadIdx.x; i < m; i += (size_t)gridDim.x * PB) {
    const float ex = q.e[3 * i], ey = q.e[3 * i + 1], ez = q.e[3 * i + 2];
    const float w = 1.0f / (sqrtf(ex * ex + ey * ey + ez * ez) + reg);
    q.w2[i] = w;
    acc += (double)w;
  }
  const double t = blockSum(acc, sh);
  if (threadIdx.x == 0) q.part[blockIdx.x * 3] = t;
}
__global__ __launch_bounds__(64) void poisson_w2coef_kernel(PoissonBufs q, int nparts) {
  const int lane = threadIdx.x;
  double s = 0.0;
  for (int k = lane; k < nparts; k += 64) s += q.part[k * 3];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  const size_t m = (size_t)q.W * q.H * 3;
  if (lane == 0) *q.w2coef = (float)m / (float)s;  // normalise so that average(w2) = 1
}
__global__ __launch_bounds__(PB) void poisson_w2scale_kernel(PoissonBufs q) {
  const size_t m = (size_t)q.W * q.H * 3;
  const float c = *q.w2coef;
  for (size_t i = blockIdx.x * (size_t)PB + threadIdx.x; i < m; i += (size_t)gridDim.x * PB) q.w2[i] *= c;
}

// r = P' diag(w2) e, p = r (calc_PTW2x + copy)
__global__ __launch_bounds__(PB) void poisson_rhs_kernel(PoissonBufs q) {
  const size_t n = (size_t)q.W * q.H, n3 = n * 3;
  const size_t rowF = (size_t)q.W * 3;
  for (size_t i = blockIdx.x * (size_t)PB + threadIdx.x; i < n3; i += (size_t)gridDim.x * PB) {
    const size_t pix = i / 3;
    const int xx = (int)(pix % q.W), yy = (int)(pix / q.W);
    float t = q.w2[pix] * q.e[i] * q.alpha;
    if (xx != 0) t += q.w2[n + pix - 1] * q.e[n3 + i - 3];
    if (xx != q.W - 1) t -= q.w2[n + pix] * q.e[n3 + i];
    if (yy != 0) t += q.w2[2 * n + pix - q.W] * q.e[2 * n3 + i - rowF];
    if (yy != q.H - 1) t -= q.w2[2 * n + pix] * q.e[2 * n3 + i];
    q.r[i] = t;
    q.p[i] = t;
  }
}

// per-channel dot product partials of u'v over the image (grid-stride over pixels)
__global__ __launch_bounds__(PB) void poisson_dot_kernel(PoissonBufs q, const float *u, const float *v) {
  __shared__ double sh[PB / 64];
  const size_t n = (size_t)q.W * q.H;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  for (size_t pix = blockIdx.x * (size_t)PB + threadIdx.x; pix < n; pix += (size_t)gridDim.x * PB) {
    a0 += (double)(u[3 * pix] * v[3 * pix]);
    a1 += (double)(u[3 * pix + 1] * v[3 * pix + 1]);
    a2 += (double)(u[3 * pix + 2] * v[3 * pix + 2]);
  }
  const double t0 = blockSum(a0, sh), t1 = blockSum(a1, sh), t2 = blockSum(a2, sh);
  if (threadIdx.x == 0) {
    q.part[blockIdx.x * 3] = t0;
    q.part[blockIdx.x * 3 + 1] = t1;
    q.part[blockIdx.x * 3 + 2] = t2;
  }
}

// finish a reduction: dst = sum of the partials (fixed order: deterministic); optionally copy rz -> rz2 first
// (the `swap(rz, rz2)` of Solver.cpp:456).  One block of 192 threads: one wave per channel.
__global__ __launch_bounds__(192) void poisson_finish_kernel(PoissonBufs q, float *dst, int nparts, int saveRz) {
  const int c = threadIdx.x / 64, lane = threadIdx.x % 64;
  if (saveRz && lane == 0) q.rz2[c] = q.rz[c];
  double s = 0.0;
  for (int k = lane; k < nparts; k += 64) s += q.part[k * 3 + c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) dst[c] = (float)s;
}

// Ap = A p with A = P' diag(w2) P, and the partials of p'Ap (calc_Ax_xAx)
__global__ __launch_bounds__(PB) void poisson_Ap_kernel(PoissonBufs q) {
  __shared__ double sh[PB / 64];
  const size_t n = (size_t)q.W * q.H;
  const float alphaSqr = q.alpha * q.alpha;
  double a0 = 0.0, a1 = 0.0, a2 = 0.0;
  for (size_t pix = blockIdx.x * (size_t)PB + threadIdx.x; pix < n; pix += (size_t)gridDim.x * PB) {
    const int xx = (int)(pix % q.W), yy = (int)(pix / q.W);
    const float wl = xx != 0 ? q.w2[n + pix - 1] : 0.f, wr = xx != q.W - 1 ? q.w2[n + pix] : 0.f;
    const float wu = yy != 0 ? q.w2[2 * n + pix - q.W] : 0.f, wd = yy != q.H - 1 ? q.w2[2 * n + pix] : 0.f;
    double acc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const size_t i = 3 * pix + c;
      const float xi = q.p[i];
      float a = q.w2[pix] * xi * alphaSqr;
      if (xx != 0) a += wl * (xi - q.p[i - 3]);
      if (xx != q.W - 1) a += wr * (xi - q.p[i + 3]);
      if (yy != 0) a += wu * (xi - q.p[i - (size_t)q.W * 3]);
      if (yy != q.H - 1) a += wd * (xi - q.p[i + (size_t)q.W * 3]);
      q.Ap[i] = a;
      acc[c] = (double)(xi * a);
    }
    a0 += acc[0];
    a1 += acc[1];
    a2 += acc[2];
  }
  const double t0 = blockSum(a0, sh), t1 = blockSum(a1, sh), t2 = blockSum(a2, sh);
  if (threadIdx.x == 0) {
    q.part[blockIdx.x * 3] = t0;
    q.part[blockIdx.x * 3 + 1] = t1;
    q.part[blockIdx.x * 3 + 2] = t2;
  }
}

// r -= Ap (rz2 / pAp) and the partials of r'r (calc_r_rz)
__global__ __launch_bounds__(PB) void poisson_r_kernel(PoissonBufs q) {
  __shared__ double sh[PB / 64];
  const size_t n = (size_t)q.W * q.H;
  float a[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) a[c] = q.rz2[c] / fmaxf(q.pAp[c], FLT_MIN);
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (size_t pix = blockIdx.x * (size_t)PB + threadIdx.x; pix < n; pix += (size_t)gridDim.x * PB) {
    float ri[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const size_t i = 3 * pix + c;
      ri[c] = q.r[i] - q.Ap[i] * a[c];
      q.r[i] = ri[c];
    }
    s0 += (double)(ri[0] * ri[0]);
    s1 += (double)(ri[1] * ri[1]);
    s2 += (double)(ri[2] * ri[2]);
  }
  const double t0 = blockSum(s0, sh), t1 = blockSum(s1, sh), t2 = blockSum(s2, sh);
  if (threadIdx.x == 0) {
    q.part[blockIdx.x * 3] = t0;
    q.part[blockIdx.x * 3 + 1] = t1;
    q.part[blockIdx.x * 3 + 2] = t2;
  }
}

// x += p (rz2 / pAp), p = r + p (rz / rz2) (calc_x_p)
__global__ __launch_bounds__(PB) void poisson_xp_kernel(PoissonBufs q) {
  const size_t n3 = (size_t)q.W * q.H * 3;
  float a[3], b[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    a[c] = q.rz2[c] / fmaxf(q.pAp[c], FLT_MIN);
    b[c] = q.rz[c] / fmaxf(q.rz2[c], FLT_MIN);
  }
  const size_t stride = (size_t)gridDim.x * PB * 3;
  for (size_t i0 = (blockIdx.x * (size_t)PB + threadIdx.x) * 3; i0 < n3; i0 += stride)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float pi = q.p[i0 + c];
      q.x[i0 + c] += pi * a[c];
      q.p[i0 + c] = q.r[i0 + c] + pi * b[c];
    }
}

// out = x (+ direct) (exportImagesMTS "final")
__global__ __launch_bounds__(PB) void poisson_export_kernel(PoissonBufs q, const float *direct, float *out) {
  const size_t n3 = (size_t)q.W * q.H * 3;
  for (size_t i = blockIdx.x * (size_t)PB + threadIdx.x; i < n3; i += (size_t)gridDim.x * PB)
    out[i] = direct ? 1.0f * direct[i] + q.x[i] : q.x[i];
}

#define PT(expr)                          \
  do {                                    \
    hipError_t _e = (expr);               \
    if (_e != hipSuccess) return _e;      \
  } while (0)

static void cgIteration(const PoissonBufs &q, int grid, hipStream_t s) {
  hipLaunchKernelGGL(poisson_Ap_kernel, dim3(grid), dim3(PB), 0, s, q);
  hipLaunchKernelGGL(poisson_finish_kernel, dim3(1), dim3(192), 0, s, q, q.pAp, grid, 1);  // pAp; rz2 = rz
  hipLaunchKernelGGL(poisson_r_kernel, dim3(grid), dim3(PB), 0, s, q);
  hipLaunchKernelGGL(poisson_finish_kernel, dim3(1), dim3(192), 0, s, q, q.rz, grid, 0);  // rz = r'r
  hipLaunchKernelGGL(poisson_xp_kernel, dim3(grid), dim3(PB), 0, s, q);
}

// dx, dy, throughput (nullable), direct (nullable), out: device pointers, W*H*3 floats.  scratch: device
// memory of poisson_scratch_bytes(W, H).  Solver::solveIndirect, Solver.cpp:376-497 (cgPrecond = false).
// the captured CG run of the last solve, kept by the caller: valid while scratch, size, alpha and cgIterMax stay
struct PoissonGraphCache {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  void *scratch = nullptr;
  int W = 0, H = 0, cgMax = 0;
  float alpha = 0.f;
};
void poisson_graph_release(PoissonGraphCache &c) {
  if (c.exec) (void)hipGraphExecDestroy(c.exec);
  if (c.graph) (void)hipGraphDestroy(c.graph);
  c = PoissonGraphCache();
}

hipError_t poisson_solve_device(const gvpm_poisson_params &prm, int W, int H, const float *dx, const float *dy,
                                const float *tp, const float *direct, float *out, void *scratch,
                                PoissonGraphCache &cache, hipStream_t s) {
  const size_t n = (size_t)W * H, n3 = 3 * n;
  PoissonBufs q;
  q.part = reinterpret_cast<double *>(scratch);
  float *f = reinterpret_cast<float *>(q.part + (size_t)PGRID * 3);
  q.b = f; f += 3 * n3;
  q.e = f; f += 3 * n3;
  q.w2 = f; f += n3;
  q.x = f; f += n3;
  q.r = f; f += n3;
  q.p = f; f += n3;
  q.Ap = f; f += n3;
  q.pAp = f; f += 4;
  q.rz = f; f += 4;
  q.rz2 = f; f += 4;
  q.w2coef = f; f += 4;
  q.W = W;
  q.H = H;
  q.alpha = tp ? fmaxf(prm.alpha, 0.f) : 0.f;
  const int grid = (int)std::min<size_t>(PGRID, (n + PB - 1) / PB);
  const int irlsMax = std::max(prm.irls_iter_max, 1), cgMax = std::max(prm.cg_iter_max, 1);
  const int cgCheck = std::max(prm.cg_iter_check, 1);
  const float tol = fmaxf(prm.cg_tolerance, 0.f);

  hipLaunchKernelGGL(poisson_setup_kernel, dim3(grid), dim3(PB), 0, s, q, dx, dy, tp);
  // one CG run without host checks = cgMax iterations: capture once, replay per IRLS iteration
  const bool hostChecks = tol > 0.f;
  hipGraphExec_t exec = nullptr;
  if (!hostChecks && cgMax > 1) {
    if (!(cache.exec && cache.scratch == scratch && cache.W == W && cache.H == H && cache.cgMax == cgMax &&
          cache.alpha == q.alpha)) {
      poisson_graph_release(cache);
      PT(hipStreamSynchronize(s));
      PT(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int k = 0; k < cgMax; ++k) cgIteration(q, grid, s);
      PT(hipStreamEndCapture(s, &cache.graph));
      PT(hipGraphInstantiate(&cache.exec, cache.graph, nullptr, nullptr, 0));
      cache.scratch = scratch; cache.W = W; cache.H = H; cache.cgMax = cgMax; cache.alpha = q.alpha;
    }
    exec = cache.exec;
  }
  hipError_t rc = hipSuccess;
  for (int irls = 0; irls < irlsMax && rc == hipSuccess; ++irls) {
    hipLaunchKernelGGL(poisson_residual_kernel, dim3(grid), dim3(PB), 0, s, q);
    if (irls == 0) {
      hipLaunchKernelGGL(poisson_fill_kernel, dim3(grid), dim3(PB), 0, s, q.w2, n3, 1.0f);
    } else {
      const float reg = fmaxf(prm.irls_reg_init, 0.f) * powf(fmaxf(prm.irls_reg_iter, 0.f), (float)(irls - 1));
      hipLaunchKernelGGL(poisson_w2_kernel, dim3(grid), dim3(PB), 0, s, q, reg);
      hipLaunchKernelGGL(poisson_w2coef_kernel, dim3(1), dim3(64), 0, s, q, grid);
      hipLaunchKernelGGL(poisson_w2scale_kernel, dim3(grid), dim3(PB), 0, s, q);
    }
    hipLaunchKernelGGL(poisson_rhs_kernel, dim3(grid), dim3(PB), 0, s, q);
    hipLaunchKernelGGL(poisson_dot_kernel, dim3(grid), dim3(PB), 0, s, q, q.r, q.r);
    hipLaunchKernelGGL(poisson_finish_kernel, dim3(1), dim3(192), 0, s, q, q.rz, grid, 0);
    if (exec) {
      rc = hipGraphLaunch(exec, s);
    } else {
      for (int cg = 0;; ++cg) {
        if (cg % cgCheck == 0 || cg == cgMax) {
          if (cg == cgMax) break;
          if (hostChecks) {
            float rz[3];
            rc = hipMemcpyAsync(rz, q.rz, sizeof(rz), hipMemcpyDeviceToHost, s);
            if (rc == hipSuccess) rc = hipStreamSynchronize(s);
            if (rc != hipSuccess || rz[0] + rz[1] + rz[2] <= tol) break;
          }
        }
        cgIteration(q, grid, s);
      }
    }
  }
  if (rc == hipSuccess) {
    hipLaunchKernelGGL(poisson_export_kernel, dim3(grid), dim3(PB), 0, s, q, direct, out);
    rc = hipGetLastError();
  }
  return rc;
}

size_t poisson_scratch_bytes(int W, int H) {
  const size_t n3 = (size_t)W * H * 3;
  return (3 * n3 * 2 + 5 * n3 + 16) * sizeof(float) + (size_t)PGRID * 3 * sizeof(double) + 64;
}

}  // namespace gvpm
