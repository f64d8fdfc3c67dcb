// ORACLE -- TEST INFRASTRUCTURE ONLY (see gvpm_oracle.hpp).
//
// Screened-Poisson reconstruction (next row f1 of SURVEY section 8): CPU restatement of the
// reference's solver with its naive backend,
//   poisson::Solver::Params::setConfigPreset     poisson_solver/Solver.cpp:91-164
//   Solver::setupBackend (b, x0, P)              Solver.cpp:297-341
//   Solver::solveIndirect (IRLS around CG)       Solver.cpp:376-497
//   Solver::exportImagesMTS ("final" = x + direct) Solver.cpp:560-581
//   Backend::calc_Px / calc_PTW2x / calc_Ax_xAx / calc_axpy / calc_xdoty / calc_r_rz / calc_x_p /
//   calc_w2                                      poisson_solver/Backend.cpp:154-384
// Sequential float sums in the reference's loop order, so that it can be compared bit for bit with the
// reference itself (oracle/_ref, built from the reference sources by Makefile.ref) -- this part of the
// oracle IS pinned: tests/test_oracle_poisson.py checks it against oracle/_ref when present and against
// the committed vectors that oracle/_ref produced (tests/golden/poisson_*.npz).
// The preconditioned branch (cgPrecond, calc_MIx) is not restated: no preset enables it (Solver.cpp:99).
#pragma once
#include <float.h>
#include <math.h>
#include <string.h>

#include <vector>

namespace oracle {

struct PoissonParams {
  float alpha;
  int irlsIterMax;
  float irlsRegInit, irlsRegIter;
  int cgIterMax, cgIterCheck;
  float cgTolerance;
};

// Solver::Params::setConfigPreset, Solver.cpp:91-164 (+ sanitize :168-181)
inline bool poissonPreset(const char *preset, PoissonParams &p) {
  p.irlsIterMax = 1; p.irlsRegInit = 0.f; p.irlsRegIter = 0.f; p.cgIterMax = 1; p.cgIterCheck = 100; p.cgTolerance = 0.f;
  if (!strcmp(preset, "L1D")) { p.irlsIterMax = 20; p.irlsRegInit = 0.05f; p.irlsRegIter = 0.5f; p.cgIterMax = 50; return true; }
  if (!strcmp(preset, "L1Q")) { p.irlsIterMax = 64; p.irlsRegInit = 1.0f; p.irlsRegIter = 0.7f; p.cgIterMax = 1000; return true; }
  if (!strcmp(preset, "L1L")) { p.irlsIterMax = 7; p.irlsRegInit = 1.0e-4f; p.irlsRegIter = 1.0e-1f; p.cgIterMax = 20000; p.cgTolerance = 1.0e-20f; return true; }
  if (!strcmp(preset, "L2D")) { p.cgIterMax = 50; return true; }
  if (!strcmp(preset, "L2Q")) { p.cgIterMax = 500; return true; }
  return false;
}

struct V3 {
  float x, y, z;
};
inline V3 v3(float a) { return V3{a, a, a}; }
inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator/(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }
inline V3 operator*(V3 a, float b) { return a * v3(b); }
inline V3 operator*(float a, V3 b) { return v3(a) * b; }
inline V3 vmax(V3 a, V3 b) { return V3{a.x > b.x ? a.x : b.x, a.y > b.y ? a.y : b.y, a.z > b.z ? a.z : b.z}; }
inline float vlength(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }

// out = x + direct (direct may be null); dx, dy, throughput: H*W*3 floats
inline void poissonSolve(const PoissonParams &prm, int W, int H, const float *dx_, const float *dy_, const float *tp_,
                         const float *direct_, float *out) {
  const int n = W * H;
  const V3 *dx = reinterpret_cast<const V3 *>(dx_), *dy = reinterpret_cast<const V3 *>(dy_);
  const V3 *tp = reinterpret_cast<const V3 *>(tp_);
  const float alpha = tp ? fmaxf(prm.alpha, 0.f) : 0.f;  // m_P.alpha, Solver.cpp:323
  std::vector<V3> b(3 * (size_t)n), e(3 * (size_t)n), x(n), r(n), p(n), Ap(n);
  std::vector<float> w2(3 * (size_t)n);
  // setupBackend, Solver.cpp:325-340
  for (int i = 0; i < n; ++i) {
    b[i] = tp ? tp[i] * alpha : v3(0.f);
    b[n + i] = dx[i];
    b[2 * (size_t)n + i] = dy[i];
    x[i] = tp ? tp[i] : v3(0.f);
  }
  const float alphaSqr = alpha * alpha;
  V3 rzA = v3(0.f), rzB = v3(0.f), pAp = v3(0.f);
  for (int irls = 0; irls < prm.irlsIterMax; ++irls) {
    // e = b - P*x (calc_Px + calc_axpy, Backend.cpp:154-176, 239-256)
    for (int yy = 0, i = 0; yy < H; ++yy)
      for (int xx = 0; xx < W; ++xx, ++i) {
        const V3 xi = x[i];
        const V3 p0 = xi * alpha;
        const V3 p1 = xx != W - 1 ? x[i + 1] - xi : v3(0.f);
        const V3 p2 = yy != H - 1 ? x[i + W] - xi : v3(0.f);
        e[i] = v3(-1.f) * p0 + b[i];
        e[n + i] = v3(-1.f) * p1 + b[n + i];
        e[2 * (size_t)n + i] = v3(-1.f) * p2 + b[2 * (size_t)n + i];
      }
    // weights (calc_w2, Backend.cpp:362-384)
    if (irls == 0) {
      for (size_t i = 0; i < 3 * (size_t)n; ++i) w2[i] = 1.f;
    } else {
      const float reg = prm.irlsRegInit * powf(prm.irlsRegIter, (float)(irls - 1));
      float w2sum = 0.f;
      for (size_t i = 0; i < 3 * (size_t)n; ++i) {
        const float w = 1.0f / (vlength(e[i]) + reg);
        w2[i] = w;
        w2sum += w;
      }
      const float coef = (float)(3 * (size_t)n) / w2sum;
      for (size_t i = 0; i < 3 * (size_t)n; ++i) w2[i] *= coef;
    }
    // r = P' diag(w2) e (calc_PTW2x, Backend.cpp:180-206); rz = r'r; p = r
    V3 *rz = &rzA, *rz2 = &rzB;
    for (int yy = 0, i = 0; yy < H; ++yy)
      for (int xx = 0; xx < W; ++xx, ++i) {
        V3 t = w2[i] * e[i] * alpha;
        if (xx != 0) t = t + w2[n + i - 1] * e[n + i - 1];
        if (xx != W - 1) t = t - w2[n + i] * e[n + i];
        if (yy != 0) t = t + w2[2 * (size_t)n + i - W] * e[2 * (size_t)n + i - W];
        if (yy != H - 1) t = t - w2[2 * (size_t)n + i] * e[2 * (size_t)n + i];
        r[i] = t;
      }
    *rz = v3(0.f);
    for (int i = 0; i < n; ++i) *rz = *rz + r[i] * r[i];
    for (int i = 0; i < n; ++i) p[i] = r[i];
    for (int cg = 0;; ++cg) {
      if (cg % prm.cgIterCheck == 0 || cg == prm.cgIterMax) {
        const float errL2W = rz->x + rz->y + rz->z;
        if (cg == prm.cgIterMax || errL2W <= prm.cgTolerance) break;
      }
      { V3 *t = rz; rz = rz2; rz2 = t; }
      // Ap = A p, pAp = p'Ap (calc_Ax_xAx, Backend.cpp:210-235)
      pAp = v3(0.f);
      for (int yy = 0, i = 0; yy < H; ++yy)
        for (int xx = 0; xx < W; ++xx, ++i) {
          const V3 xi = p[i];
          V3 a = w2[i] * xi * alphaSqr;
          if (xx != 0) a = a + w2[n + i - 1] * (xi - p[i - 1]);
          if (xx != W - 1) a = a + w2[n + i] * (xi - p[i + 1]);
          if (yy != 0) a = a + w2[2 * (size_t)n + i - W] * (xi - p[i - W]);
          if (yy != H - 1) a = a + w2[2 * (size_t)n + i] * (xi - p[i + W]);
          Ap[i] = a;
          pAp = pAp + xi * a;
        }
      // r -= Ap (rz2/pAp), rz = r'r (calc_r_rz, Backend.cpp:278-304)
      const V3 aa = *rz2 / vmax(pAp, v3(FLT_MIN));
      *rz = v3(0.f);
      for (int i = 0; i < n; ++i) {
        const V3 ri = r[i] - Ap[i] * aa;
        r[i] = ri;
        *rz = *rz + ri * ri;
      }
      // x += p (rz2/pAp), p = r + p (rz/rz2) (calc_x_p, Backend.cpp:308-338)
      const V3 bb = *rz / vmax(*rz2, v3(FLT_MIN));
      for (int i = 0; i < n; ++i) {
        const V3 pi = p[i];
        x[i] = x[i] + pi * aa;
        p[i] = r[i] + pi * bb;
      }
    }
  }
  // exportImagesMTS "final", Solver.cpp:560-581: r = direct, r = 1*r + x
  const V3 *direct = reinterpret_cast<const V3 *>(direct_);
  V3 *o = reinterpret_cast<V3 *>(out);
  for (int i = 0; i < n; ++i) o[i] = direct ? v3(1.f) * direct[i] + x[i] : x[i];
}

}  // namespace oracle
