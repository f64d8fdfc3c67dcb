// Device helpers shared by the gather kernels: the closed-form pieces of the shift
// (phase / medium / occluder / reconnection / MIS) -- see gather_bre.hip for the citations.
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"
#include "vec.h"

namespace gvpm {

#define INV_PI_F 0.31830988618379067154f
#define INV_FOURPI_F 0.07957747154594766788f


struct RayReg {
  f3 o, d, eye;
  float len, pdf, jac, gop;
  bool valid;
};

// g == 0: the isotropic plugin's constant (src/phase/isotropic.cpp:76-78) -- the formula returns 1/4pi exactly there too,
// but g is wave-uniform wherever it is the medium's: a scalar branch instead of a dot product, a square root and a
// reciprocal per call (six calls an evaluation; C2: -1 % on the step)
__device__ __forceinline__ float phaseEval(float g, f3 wi, f3 wo) {
  if (g == 0.f) return INV_FOURPI_F;
  const float temp = 1.0f + g * g + 2.0f * g * dot(wi, wo);
  return INV_FOURPI_F * (1.f - g * g) * frcp(temp * fsqrt(temp));
}

// HomogeneousMedium::eval over a distance (balance strategy)
// (sigma_t is equal across channels -- homogeneous.cpp:196-200, enforced by gvpm_upload_medium --
// so the three channel exponentials are one)
__device__ __forceinline__ void mediumEval(const MediumDev &m, float dist, f3 &tr, float &pdfSuccess) {
  float e = __expf(-m.sigmaT[0] * dist);
  pdfSuccess = m.sigmaT[0] * e * m.msw;
  if (e < 1e-20f) e = 0.f;
  tr = mk3(e);
}

// Moeller-Trumbore, triangle.h:109-145 + interval test skdtree.h:318-320, in THREE states (round 5).
//
// The reference decides  det != 0, 0 <= u <= 1, v >= 0, u + v <= 1, mint <= t <= maxt  with u = A / C, v = B / C, t = T / C,
// C = e1 . (d x e2), A = tvec . (d x e2), B = d . (tvec x e1), T = e2 . (tvec x e1), tvec = o - v0.  Here the four
// barycentric comparisons are taken division-free on sg A, sg B, |C| (sg = sign C) and the interval test on the plane
// distances of the segment's ends (see triHit3), each with a RIGOROUS fp32 error margin: with eps = 2^-24,
// S = |o|_1 + |v0|_1, L1 = |e1|_1, L2 = |e2|_1 the rounding of the sums above (and the ~1e-7 the device's fp32 direction
// is off the oracle's) is bounded by  errC <= 5 eps L1 L2,  errA <= 8 eps S L2,  errB <= 18 eps S L1;
// the margins take 1e-6 ~ 17 eps.  Outside every margin the decision is the one exact arithmetic
// on the same fp32 data takes -- the fp64 oracle's, and a double-precision reference's.  Inside one:
//   TRI_AMB -- fp32 cannot tell.  The caller DEFERS the shift to the exact pass (exact_shift.hip: the reference's
//   statement in uncontracted fp64), or, where no exact pass exists, takes bit 0: the plain fp32 decision.
// The systematic case is a segment that STARTS within rounding of the triangle's plane -- a parent that fp32 left behind
// the wall it sits on (grid_build.hip, ownWall) -- along a grazing direction: the plane distance of the start is pure
// rounding residue and t >= mint is decided by it.  The generic near-threshold cases (a hit within 1e-6 of an edge) go the same way.
#define GVPM_TRI_MISS 0
#define GVPM_TRI_HIT 1
#define GVPM_TRI_AMB 2
// s0 = n . (o - v0), sd = n . d with the stored unit normal (callers have them for the plane-side early-out).  The interval
// test  mint <= t <= maxt  is the statement "the segment's ends lie on different sides of the triangle's plane":
// e0 = s0 + sd mint and e1 = s0 + sd maxt, each good to mE ~ 5e-7 (|o|_1 + |v0|_1 + maxt) -- eight times tighter than the
// same decision through T = e2 . (tvec x e1), whose rounding carries the triangle's extent.
__device__ __forceinline__ int triHit3(f3 v0, f3 e1, f3 e2, f3 o, f3 d, float mint, float maxt, float oAbs1, float s0, float sd) {
  const f3 pvec = cross(d, e2);
  const float C = dot(e1, pvec);
  const f3 tvec = o - v0;
  const float A = dot(tvec, pvec);
  const f3 qvec = cross(tvec, e1);
  const float Bq = dot(d, qvec);
  const float L1 = fabsf(e1.x) + fabsf(e1.y) + fabsf(e1.z), L2 = fabsf(e2.x) + fabsf(e2.y) + fabsf(e2.z);
  const float S = oAbs1 + fabsf(v0.x) + fabsf(v0.y) + fabsf(v0.z);
  const float k = 1e-6f;
  const float mC = k * L1 * L2, mA = k * S * L2, mB = k * S * L1, mE = 5e-7f * (S + maxt);
  const float aC = fabsf(C);
  const float sA = C < 0.f ? -A : A, sB = C < 0.f ? -Bq : Bq;
  const float s2 = aC - sA, s4 = s2 - sB;
  const float m2 = mA + mC, m4 = m2 + mB;
  const float e0 = s0 + sd * mint, e1p = s0 + sd * maxt;
  const float lo = fminf(e0, e1p), hi = fmaxf(e0, e1p);
  const bool noCross = lo > mE || hi < -mE, cross = lo < -mE && hi > mE;
  // (an EMPTY interval, mint > maxt -- the as-written visibility of a reconnection shorter than Epsilon / ShadowEpsilon: the
  // reference's mint <= t <= maxt holds for no t.  The ends' sides are symmetric in the two: a plane certainly crossed between
  // them is then a certain miss -- found by tests/stress_vpm.py, a medium parent 1e-4 from a wall)
  const bool empty = mint > maxt;
  const bool fail = noCross || (cross && empty) || sA < -mA || s2 < -m2 || sB < -mB || s4 < -m4;
  const bool pass = cross && !empty && aC > mC && sA > mA && s2 > m2 && sB > mB && s4 > m4;
  return fail ? GVPM_TRI_MISS : (pass ? GVPM_TRI_HIT : GVPM_TRI_AMB);
}
__device__ __forceinline__ int triHit3(const float4 t0, const float4 t1, const float4 t2, f3 o, f3 d, float mint, float maxt, float oAbs1) {
  const f3 v0 = mk3(t0.x, t0.y, t0.z), nrm = mk3(t0.w, t1.w, t2.w);
  return triHit3(v0, mk3(t1.x, t1.y, t1.z), mk3(t2.x, t2.y, t2.z), o, d, mint, maxt, oAbs1, dot(nrm, o - v0), dot(nrm, d));
}
// A second opinion on a triangle triHit3 left undecided (the G-Beams shadow segments, gather_beams.hip).  The division-free
// comparisons above bound the errors of A, B and C independently -- each carries |o - v0| |e|, the distance to the triangle's
// FAR corner -- although an error of the direction moves A / C only by the lever from the origin to the crossing point.
// Here the crossing point itself is formed, P = (o - v0) + d t with t = -s0 / sd, and tested against the edges in the
// triangle's plane: its error is ~4u (|o - v0| + t) of rounding, dirErr t of the direction (the device's fp32 direction
// against the reference's: dirErr ~ 1e-6) and the plane distance's own error over |sd| -- 1e-6 of the scene where the
// margins above are 1e-5 |e| / sin(crossing angle): the plate of S-laser, whose thin triangles' edge LINES run through the
// aperture, went from 2.8 % undecided shadow segments to ~0.03 %.  The normal is recomputed (N = e1 x e2: a point exactly in
// an axis plane gets a plane distance of exactly zero error).  endErr: absolute error of the segment's end point.
__device__ __forceinline__ int triHitFine(f3 v0, f3 e1, f3 e2, f3 o, f3 d, float mint, float maxt, float dirErr, float endErr) {
  const f3 tv = o - v0;  // (one rounding per component)
  const f3 N = cross(e1, e2);
  const float n1 = fabsf(N.x) + fabsf(N.y) + fabsf(N.z);
  const float s0 = dot(N, tv), sd = dot(N, d);
  const float a0 = fabsf(N.x * tv.x) + fabsf(N.y * tv.y) + fabsf(N.z * tv.z);
  const float t1 = fabsf(tv.x) + fabsf(tv.y) + fabsf(tv.z);
  const float eS = s0 + sd * mint, eE = s0 + sd * maxt;
  const float mS = 5e-7f * (a0 + fabsf(sd) * mint) + n1 * dirErr * mint;
  const float mE = 5e-7f * (a0 + fabsf(sd) * maxt) + n1 * endErr;
  const bool sP = eS > mS, sN = eS < -mS, eP = eE > mE, eN = eE < -mE;
  if ((sP && eP) || (sN && eN)) return GVPM_TRI_MISS;
  if (!((sP && eN) || (sN && eP))) return GVPM_TRI_AMB;
  if (mint > maxt) return GVPM_TRI_MISS;  // (an empty interval whose ends certainly straddle the plane: see triHit3)
  const float isd = frcp(sd);
  const float t = -s0 * isd;
  const f3 P = tv + d * t;
  const float p1n = fabsf(P.x) + fabsf(P.y) + fabsf(P.z);
  // position error of P: rounding of tv + d t, the direction's error over t, the plane distance's error over |sd|
  const float pe = 3e-7f * (t1 + t + p1n) + dirErr * t + 5e-7f * a0 * fabsf(isd);
  const float l1 = fabsf(e1.x) + fabsf(e1.y) + fabsf(e1.z), l2 = fabsf(e2.x) + fabsf(e2.y) + fabsf(e2.z);
  const float NN = dot(N, N);
  // one edge function per edge, each with the margin of ITS edge (u + v <= 1 taken as 1 - u - v would add the margins of two
  // nearly parallel edges of a thin triangle: forty times the third edge's own)
  const f3 e3 = e2 - e1;
  const float l3 = fabsf(e3.x) + fabsf(e3.y) + fabsf(e3.z);
  const float uN = dot(cross(P, e2), N), vN = dot(cross(e1, P), N), wN = dot(cross(e3, P - e1), N);
  const float mu = pe * l2 * n1, mv = pe * l1 * n1, mw = (pe + 2e-7f * l1) * l3 * n1;
  if (!(NN > 0.f)) return GVPM_TRI_AMB;
  if (uN < -mu || vN < -mv || wN < -mw) return GVPM_TRI_MISS;
  if (uN > mu && vN > mv && wN > mw) return GVPM_TRI_HIT;
  return GVPM_TRI_AMB;
}
// any-hit over a list: a certain hit settles it; else an undecidable triangle makes the whole answer undecidable
__device__ __forceinline__ int triCombine(int acc, int t) {
  if (acc == GVPM_TRI_HIT || t == GVPM_TRI_HIT) return GVPM_TRI_HIT;
  return (acc | t) & GVPM_TRI_AMB;
}
// the plain fp32 test (branch-free: the tests of the reference are and-ed; a zero determinant gives inf / NaN, which fail
// the comparisons like the early return): what the literal fp64 cross-check of G-Beams (GVPM_BEAMS_FP64) walks the scene with
__device__ __forceinline__ bool triHit(f3 v0, f3 e1, f3 e2, f3 o, f3 d, float mint, float maxt) {
  const f3 pvec = cross(d, e2);
  const float det = dot(e1, pvec);
  const float inv = frcp(det);
  const f3 tvec = o - v0;
  const float u = dot(tvec, pvec) * inv;
  const f3 qvec = cross(tvec, e1);
  const float v = dot(d, qvec) * inv;
  const float t = dot(e2, qvec) * inv;
  return det != 0.f && u >= 0.f && u <= 1.f && v >= 0.f && u + v <= 1.f && t >= mint && t <= maxt;
}
// The same test in uncontracted fp64, in the operation order of the oracle's (and the reference's) statement.
__device__ __forceinline__ bool triHitExact(f3 v0f, f3 e1f, f3 e2f, f3 of, d3 dd, double mint, double maxt) {
#pragma clang fp contract(off)
  const double e1x = e1f.x, e1y = e1f.y, e1z = e1f.z, e2x = e2f.x, e2y = e2f.y, e2z = e2f.z;
  const double dx = dd.x, dy = dd.y, dz = dd.z;
  const double px = dy * e2z - dz * e2y, py = dz * e2x - dx * e2z, pz = dx * e2y - dy * e2x;
  const double det = e1x * px + e1y * py + e1z * pz;
  if (det == 0.0) return false;
  const double inv = 1.0 / det;
  const double tx = (double)of.x - (double)v0f.x, ty = (double)of.y - (double)v0f.y, tz = (double)of.z - (double)v0f.z;
  const double u = (tx * px + ty * py + tz * pz) * inv;
  if (u < 0.0 || u > 1.0) return false;
  const double qx = ty * e1z - tz * e1y, qy = tz * e1x - tx * e1z, qz = tx * e1y - ty * e1x;
  const double v = (dx * qx + dy * qy + dz * qz) * inv;
  if (!(v >= 0.0 && u + v <= 1.0)) return false;
  const double t = (e2x * qx + e2y * qy + e2z * qz) * inv;
  return t >= mint && t <= maxt;
}

// scene->rayIntersect(ray), any-hit, exactly: the occluder BVH's boxes are padded (scene_bvh.cpp), the slab test runs in
// fp64 on them -- conservative -- and every triangle of a reached leaf takes the reference's test in fp64.
static __device__ bool anyHitExact(const GatherArgs &a, f3 o, d3 d, double mint, double maxt) {
#pragma clang fp contract(off)
  if (a.ntri == 0u) return false;
  const double ox = o.x, oy = o.y, oz = o.z;
  const double ix = 1.0 / d.x, iy = 1.0 / d.y, iz = 1.0 / d.z;
  uint32_t stack[32];
  int sp = 0;
  uint32_t cur = 0;
  for (;;) {
    const float4 lo = a.bvh[2 * (size_t)cur], hi = a.bvh[2 * (size_t)cur + 1];
    const double tx0 = ((double)lo.x - ox) * ix, tx1 = ((double)hi.x - ox) * ix;
    const double ty0 = ((double)lo.y - oy) * iy, ty1 = ((double)hi.y - oy) * iy;
    const double tz0 = ((double)lo.z - oz) * iz, tz1 = ((double)hi.z - oz) * iz;
    // (fmin / fmax drop the NaNs of 0 * inf; a box is entered when in doubt: slack of 1e-9 on the interval)
    const double tn = fmax(fmax(fmin(tx0, tx1), fmin(ty0, ty1)), fmax(fmin(tz0, tz1), mint)) - 1e-9;
    const double tf = fmin(fmin(fmax(tx0, tx1), fmax(ty0, ty1)), fmin(fmax(tz0, tz1), maxt)) + 1e-9;
    bool descend = false;
    if (tn <= tf) {
      const uint32_t first = __float_as_uint(lo.w), count = __float_as_uint(hi.w);
      if (count == 0u) {
        if (sp < 32) stack[sp++] = first + 1u;
        cur = first;
        descend = true;
      } else {
        for (uint32_t i = first; i < first + count; ++i) {
          const float4 t0 = a.tri4[3 * (size_t)i], t1 = a.tri4[3 * (size_t)i + 1], t2 = a.tri4[3 * (size_t)i + 2];
          if (triHitExact(mk3(t0.x, t0.y, t0.z), mk3(t1.x, t1.y, t1.z), mk3(t2.x, t2.y, t2.z), o, d, mint, maxt)) return true;
        }
      }
    }
    if (!descend) {
      if (sp == 0) return false;
      cur = stack[--sp];
    }
  }
}

// scene->rayIntersect(ray), any-hit: stack walk of the occluder BVH (scene_bvh.h), triangles as
// {v0,n.x} {e1,n.y} {e2,n.z} in leaf order.  Deliberately not inlined: it is the rare path (the
// as-written shadow segment is served by the per-photon near-occluder list below) and inlining
// it cost the evaluation kernels ~160 VGPRs.  Returns a GVPM_TRI_* state.
template <bool PLAIN = false>
static __device__ __noinline__ int anyHitScene(const float4 *bvh, const float4 *tri4, uint32_t ntri, f3 o, f3 d, float mint,
                                        float maxt) {
  if (ntri == 0u) return GVPM_TRI_MISS;
  const f3 inv = mk3(1.f / d.x, 1.f / d.y, 1.f / d.z);
  const float oAbs1 = fabsf(o.x) + fabsf(o.y) + fabsf(o.z);
  uint32_t stack[32];
  int sp = 0;
  uint32_t cur = 0;
  int res = GVPM_TRI_MISS;
  for (;;) {
    const float4 lo = bvh[2 * (size_t)cur], hi = bvh[2 * (size_t)cur + 1];
    // slab test; fminf/fmaxf drop the NaNs of 0 * inf
    const float tx0 = (lo.x - o.x) * inv.x, tx1 = (hi.x - o.x) * inv.x;
    const float ty0 = (lo.y - o.y) * inv.y, ty1 = (hi.y - o.y) * inv.y;
    const float tz0 = (lo.z - o.z) * inv.z, tz1 = (hi.z - o.z) * inv.z;
    const float tn = fmaxf(fmaxf(fminf(tx0, tx1), fminf(ty0, ty1)), fmaxf(fminf(tz0, tz1), mint));
    const float tf = fminf(fminf(fmaxf(tx0, tx1), fmaxf(ty0, ty1)), fminf(fmaxf(tz0, tz1), maxt));
    bool descend = false;
    if (tn <= tf) {
      const uint32_t first = __float_as_uint(lo.w), count = __float_as_uint(hi.w);
      if (count == 0u) {
        if (sp < 32) stack[sp++] = first + 1u;
        cur = first;
        descend = true;
      } else {
        for (uint32_t i = first; i < first + count; ++i) {
          if (PLAIN) {
            // (PLAIN: the fp64 cross-check of G-Beams -- its shadow segment's direction is the double one, rounded once)
            const float4 t0 = tri4[3 * (size_t)i], t1 = tri4[3 * (size_t)i + 1], t2 = tri4[3 * (size_t)i + 2];
            if (triHit(mk3(t0.x, t0.y, t0.z), mk3(t1.x, t1.y, t1.z), mk3(t2.x, t2.y, t2.z), o, d, mint, maxt)) return GVPM_TRI_HIT;
          } else {
            res = triCombine(res, triHit3(tri4[3 * (size_t)i], tri4[3 * (size_t)i + 1], tri4[3 * (size_t)i + 2], o, d, mint, maxt, oAbs1));
            if (res == GVPM_TRI_HIT) return res;
          }
        }
      }
    }
    if (!descend) {
      if (sp == 0) return res;
      cur = stack[--sp];
    }
  }
}

// As written (shift_volume_photon.cpp:396) the shadow segment is [Epsilon, lProj*ShadowEpsilon]
// from the photon's parent: only occluders within that distance of the parent can be hit.  The
// grid build lists them per photon (reorder_kernel: up to 12 byte indices in the three spare
// words of the record), so the loop touches 0-12 triangles.  FULLVIS kernels (intended visibility,
// more than 254 occluders, or a photon whose list overflowed) walk the BVH instead; the fast
// kernels carry no call, which is worth ~30 VGPRs.
// Margin of the plane-side early-out in front of a triangle test: the signed distances s0 + sd * t of the segment's
// two ends to the triangle's plane are fp32 sums of products of O(|o|_1 + |v0|_1) and O(maxt) operands, so their
// rounding error is a few ulps of that magnitude.  A triangle is skipped only when BOTH ends lie on one side by MORE
// than this margin; anything closer goes to triHit3, which decides as the reference's rayIntersect does -- or says it cannot.
__device__ __forceinline__ float planeSideMargin(float triAbs1, f3 o, float maxt) {
  return 2e-6f * (fabsf(o.x) + fabsf(o.y) + fabsf(o.z) + triAbs1 + maxt);
}
__device__ __forceinline__ bool planeSideMiss(float s0, float sd, float mint, float maxt, float margin) {
  const float e0 = s0 + sd * mint, e1 = s0 + sd * maxt;
  return fminf(e0, e1) > margin || fmaxf(e0, e1) < -margin;
}

template <bool PLAIN = false>
__device__ __forceinline__ int nearListHit(const float4 *tri, uint32_t nl0, uint32_t nl1, uint32_t nl2, f3 o, f3 d,
                                           float mint, float maxt, float margin) {
  int res = GVPM_TRI_MISS;
  const float oAbs1 = fabsf(o.x) + fabsf(o.y) + fabsf(o.z);
  uint32_t l = nl0;
#pragma unroll 1
  for (int k = 0; k < 12; ++k) {
    const uint32_t i = l & 0xFFu;
    if (i == 0xFFu) break;
    l = k == 3 ? nl1 : (k == 7 ? nl2 : (l >> 8) | 0xFF000000u);
    const float4 t0 = tri[3 * i], t1 = tri[3 * i + 1], t2 = tri[3 * i + 2];
    // both ends of the segment strictly on one side of the triangle's plane (the stored unit normal; zero for a
    // degenerate triangle, which then goes to the full test): nothing to intersect -- a quarter of the work of the test
    // it spares, and at C3 a third of a beam's listed occluders (ceiling and floor under and above a vertical beam)
    const f3 v0 = mk3(t0.x, t0.y, t0.z), nrm = mk3(t0.w, t1.w, t2.w);
    const float s0 = dot(nrm, o - v0), sd = dot(nrm, d);
    if (planeSideMiss(s0, sd, mint, maxt, margin)) continue;
    if (PLAIN) res |= triHit(v0, mk3(t1.x, t1.y, t1.z), mk3(t2.x, t2.y, t2.z), o, d, mint, maxt) ? GVPM_TRI_HIT : GVPM_TRI_MISS;
    else res = triCombine(res, triHit3(v0, mk3(t1.x, t1.y, t1.z), mk3(t2.x, t2.y, t2.z), o, d, mint, maxt, oAbs1, s0, sd));
  }
  return res;
}
// the same for scenes of more than 254 occluders: six 16-bit indices (grid_build.hip, nearOccluders)
__device__ __forceinline__ int nearListHitWide(const float4 *tri, uint32_t nl0, uint32_t nl1, uint32_t nl2, f3 o, f3 d,
                                               float mint, float maxt) {
  int res = GVPM_TRI_MISS;
  const float oAbs1 = fabsf(o.x) + fabsf(o.y) + fabsf(o.z);
  uint32_t l = nl0;
#pragma unroll 1
  for (int k = 0; k < 6; ++k) {
    const uint32_t i = l & 0xFFFFu;
    if (i == 0xFFFFu) break;
    l = k == 1 ? nl1 : (k == 3 ? nl2 : (l >> 16) | 0xFFFF0000u);
    res = triCombine(res, triHit3(tri[3 * (size_t)i], tri[3 * (size_t)i + 1], tri[3 * (size_t)i + 2], o, d, mint, maxt, oAbs1));
  }
  return res;
}
// extension list (lists longer than the inline slots; every list of a scene beyond 16-bit indices)
__device__ __forceinline__ int nearListHitExt(const float4 *tri, const uint32_t *ext, uint32_t off, f3 o, f3 d,
                                              float mint, float maxt) {
  int res = GVPM_TRI_MISS;
  const float oAbs1 = fabsf(o.x) + fabsf(o.y) + fabsf(o.z);
  const uint32_t n = ext[off];
#pragma unroll 1
  for (uint32_t k = 0; k < n; ++k) {
    const uint32_t i = ext[off + 1u + k];
    res = triCombine(res, triHit3(tri[3 * (size_t)i], tri[3 * (size_t)i + 1], tri[3 * (size_t)i + 2], o, d, mint, maxt, oAbs1));
  }
  return res;
}
// ldsTri: the occluders staged in LDS by the kernel (small scenes), or null.  Returns a GVPM_TRI_* state.
template <bool FULLVIS>
__device__ __forceinline__ int shadowBlocked(const GatherArgs &a, const float4 *ldsTri, uint32_t nl0, uint32_t nl1,
                                             uint32_t nl2, f3 o, f3 d, float mint, float maxt) {
#ifdef GVPM_PROBE_PLAINVIS
  constexpr bool PL = true;  // probe builds only: what the three-state test costs
#else
  constexpr bool PL = false;
#endif
  if (FULLVIS) return anyHitScene<PL>(a.bvh, a.tri4, a.ntri, o, d, mint, maxt);
  if (a.ntri > GVPM_NEAR_NARROW_MAX) {
    if ((nl0 >> 24) == 0xFDu) return nearListHitExt(a.tri4, a.nearExt, nl1, o, d, mint, maxt);
    return nearListHitWide(a.tri4, nl0, nl1, nl2, o, d, mint, maxt);
  }
  if ((nl0 >> 24) == 0xFDu) return nearListHitExt(a.tri4, a.nearExt, nl1, o, d, mint, maxt);
  const float margin = planeSideMargin(a.triAbs1, o, maxt);
  return ldsTri ? nearListHit<PL>(ldsTri, nl0, nl1, nl2, o, d, mint, maxt, margin)
                : nearListHit<PL>(a.tri4, nl0, nl1, nl2, o, d, mint, maxt, margin);
}

// ---- deferral to the exact pass (device_types.h, ExEntry) ----
// the hot loops' side: one global atomic and one 16-byte store per deferred shift (rare)
__device__ __forceinline__ void deferNote(const GatherArgs &a, uint32_t kind, uint32_t set, uint32_t recIdx, uint32_t shift, uint32_t cause) {
  const uint32_t slot = atomicAdd(a.exOvfCount, 1u);
  if (slot < a.exOvfCap) a.exOvf[slot] = make_uint4(set, recIdx, kind | (shift << 8) | (cause << 16), 0u);
}
// one quad of an entry: 0 header, 1..8 the record, 9..28 the five rays, 29 per technique, 30..31 zero
__device__ __forceinline__ float4 exQuad(const GatherArgs &a, uint32_t set, uint32_t recIdx, uint32_t meta, uint32_t part) {
  float outScale = a.iterScale, radius = a.radius;
  float4 extra = make_float4(0.f, 0.f, 0.f, 0.f);
  if ((meta & 0xFFu) == GVPM_EX_KIND_VPM) {
    // G-VPM notes name the camera SAMPLE: its beam set, its random number and selection pdf, the pixel's own radius
    // (querySize = R * POURCENTAGE_BS * gp.scaleVol, gvpm.cpp:1082,1132 -- read before this iteration's update)
    const gvpm_vpm_sample sm = a.samples[set];
    set = sm.set;
    const uint32_t pix = a.rays[(size_t)set * 5].pixel;
    const float scaleVol = a.scaleVol[(size_t)(pix >> 16) * a.cfg.width + (pix & 0xFFFFu)];
    radius = (a.cfg.bsphere_radius * 0.01f) * scaleVol;
    outScale = 1.f / (float)a.cfg.nb_camera_samples;
    extra = make_float4(sm.rand, sm.pdf_sel, scaleVol, 0.f);  // (the pixel's scale itself: the pass forms the radius in double)
  }
  if (part == 0u) return make_float4(__uint_as_float(meta), 0.f, outScale, radius);
  if (part <= GVPM_REC_QUADS) return a.cold[(size_t)recIdx * GVPM_REC_QUADS + (part - 1u)];
  if (part <= GVPM_REC_QUADS + 20u) return reinterpret_cast<const float4 *>(a.rays + (size_t)set * 5)[part - 1u - GVPM_REC_QUADS];
  if (part == GVPM_REC_QUADS + 21u) return extra;
  return make_float4(0.f, 0.f, 0.f, 0.f);
}

// GatherPoint::sensorMIS, gvpm_struct.h:608-631 (sDist == bDist for BRE: same t')
__device__ __forceinline__ float sensorMIS(const RayReg &s, const RayReg &b, uint32_t edge) {
  float jacobian = s.jac;
  float ratio = fdiv(s.pdf, b.pdf);
  if (edge != 1u) {
    jacobian *= fdiv(s.gop, b.gop);
    ratio *= fdiv(b.gop, s.gop);
  }
  return ratio * jacobian;
}

struct PhotonCold {
  f3 pos, wi, flux, parentPos, parentN, prefixW, parentScat, parentWi;
  float parentPdf, edgePdf, parentRR, parentG;
  uint32_t bits;
  uint32_t nl0, nl1, nl2;  // up to 12 occluder indices near the parent (0xFF = none); top byte of nl0 0xFE: overflow
};

// the front of the record: all the base contribution and the null shifts need
struct PhotonFront {
  f3 pos, wi, flux;
  uint32_t bits;
};
__device__ __forceinline__ PhotonFront loadFront(const GatherArgs &a, uint32_t idx) {
  PhotonFront c;
  const float4 *rec = a.cold + (size_t)idx * GVPM_REC_QUADS;
  const float4 c0 = rec[0], c1 = rec[1], c2 = rec[2];
  c.pos = mk3(c0.x, c0.y, c0.z); c.bits = __float_as_uint(c0.w);
  c.wi = mk3(c1.x, c1.y, c1.z);
  c.flux = mk3(c2.x, c2.y, c2.z);
  return c;
}

// the photon's 128-byte record (one cache line)
__device__ __forceinline__ PhotonCold loadCold(const GatherArgs &a, uint32_t idx) {
  PhotonCold c;
  const float4 *rec = a.cold + (size_t)idx * GVPM_REC_QUADS;
  const float4 c0 = rec[0], c1 = rec[1], c2 = rec[2], c3 = rec[3], c4 = rec[4], c5 = rec[5], c6 = rec[6], c7 = rec[7];
  c.pos = mk3(c0.x, c0.y, c0.z); c.bits = __float_as_uint(c0.w);
  c.wi = mk3(c1.x, c1.y, c1.z); c.parentPdf = c1.w;
  c.flux = mk3(c2.x, c2.y, c2.z); c.edgePdf = c2.w;
  c.parentPos = mk3(c3.x, c3.y, c3.z); c.parentRR = c3.w;
  c.parentN = mk3(c4.x, c4.y, c4.z); c.parentG = c4.w;
  c.prefixW = mk3(c5.x, c5.y, c5.z);
  c.nl0 = __float_as_uint(c5.w);
  c.parentScat = mk3(c6.x, c6.y, c6.z);
  c.nl1 = __float_as_uint(c6.w);
  c.parentWi = mk3(c7.x, c7.y, c7.z);
  c.nl2 = __float_as_uint(c7.w);
  return c;
}

// MicrofacetDistribution, isotropic (src/bsdfs/microfacet.h): D of a half vector with cosine cH to the normal (:191-232) and
// Smith's G1 of a direction with cosine cV to the normal and vDotH to the half vector (:477-518).
__device__ __forceinline__ float microfacetD(int ggx, float alpha, float cH) {
  if (cH <= 0.f) return 0.f;
  const float c2 = cH * cH;
  const float e = fdiv(fmaxf(1.f - c2, 0.f), alpha * alpha * c2);  // tan^2 / alpha^2
  float r;
  if (ggx) {
    const float root = (1.f + e) * c2;
    r = frcp(3.14159265358979323846f * alpha * alpha * root * root);
  } else {
    r = fdiv(__expf(-e), 3.14159265358979323846f * alpha * alpha * c2 * c2);
  }
  return r * cH < 1e-20f ? 0.f : r;
}
__device__ __forceinline__ float microfacetG1(int ggx, float alpha, float cV, float vDotH) {
  if (vDotH * cV <= 0.f) return 0.f;
  const float t2 = 1.f - cV * cV;
  if (t2 <= 0.f) return 1.f;  // perpendicular incidence
  const float tanT = fabsf(fdiv(fsqrt(t2), cV));
  if (ggx) {
    const float root = alpha * tanT;
    return fdiv(2.f, 1.f + fsqrt(1.f + root * root));
  }
  const float a = frcp(alpha * tanT);
  if (a >= 1.6f) return 1.f;
  const float a2 = a * a;
  return fdiv(3.535f * a + 2.181f * a2, 1.f + 2.276f * a + 2.577f * a2);
}
// fresnelConductorExact, one channel (src/libcore/util.cpp:747-769)
__device__ __forceinline__ float fresnelConductor(float cI, float eta, float k) {
  const float c2 = cI * cI, s2 = 1.f - c2, s4 = s2 * s2;
  const float t1 = eta * eta - k * k - s2;
  const float a2pb2 = fsqrt(fmaxf(t1 * t1 + k * k * eta * eta * 4.f, 0.f));
  const float aa = fsqrt(fmaxf((a2pb2 + t1) * 0.5f, 0.f));
  const float term1 = a2pb2 + c2, term2 = aa * (2.f * cI);
  const float Rs2 = fdiv(term1 - term2, term1 + term2);
  const float term3 = a2pb2 * c2 + s4, term4 = term2 * s2;
  const float Rp2 = Rs2 * fdiv(term3 - term4, term3 + term4);
  return 0.5f * (Rp2 + Rs2);
}

// A glossy surface parent (GVPM_PARENT_SURFACE_BSDF): BSDF::eval and BSDF::pdf * pdfComponent of the table entry the
// record names, towards the new direction `wo` (shift_diffuse.cpp:25-41 with bRec.component = -1).  Phong, src/bsdfs/
// phong.cpp:121-186: eval = (ks (e + 2) / 2pi alpha^e + kd / pi) cos_o, pdf = w alpha^e (e + 1) / 2pi + (1 - w) cos_o / pi,
// alpha = wo . reflect(wi).  Rough conductor, src/bsdfs/roughconductor.cpp:257-319: eval = F D G / (4 cos_i), pdf = D G1(wi)
// / (4 cos_i) or D cos_H / (4 |wo . H|) (include/gvpm_hip.h).  cosWi, cosWo > 0 is the caller's test.  False: no such
// entry (a failed shift).
// state (optional out, round 5): bit 0 -- the pdf is POSITIVE in double precision although it underflowed here (the
// specular component of a Phong wall alone, exponent ~1000: alpha^e leaves fp32 below alpha ~ 0.94 and fp64 only below ~0.6;
// with pdf == 0 the reference fails the shift, with a positive one -- however small -- it succeeds, with weight 1 and a
// flux that rounds to zero: only the counter tells them apart); bit 1 -- within rounding of the double's own underflow:
// the exact pass decides, with the lobe in fp64 (phongEvalD).
__device__ __forceinline__ bool glossyParentEval(const GatherArgs &a, float index, f3 kd, f3 n, f3 wi, f3 wo, float cosWi,
                                                 float cosWo, f3 &f, float &pdf, uint32_t *state = nullptr) {
  const uint32_t bi = (uint32_t)index;
  f = mk3(0.f);
  pdf = 0.f;
  if (state) *state = 0u;
  if (!(index >= 0.f) || bi >= a.nbsdfs) return false;
  const float4 b0 = a.bsdfs[4 * bi], b1 = a.bsdfs[4 * bi + 1];
  const int kind = __float_as_int(b0.x);
  if (kind == GVPM_BSDF_PHONG) {
    const float e = b1.x, w = b1.y;
    const f3 refl = n * (2.f * cosWi) - wi;
    const float alpha = dot(wo, refl);
    const float l2 = alpha > 0.f ? e * __builtin_log2f(alpha) : -INFINITY;
    float lobe = alpha > 0.f ? __builtin_exp2f(l2) : 0.f;  // std::pow(alpha, exponent)
    const float INV_TWOPI_F = 0.15915494309189533577f;
    // (the entry's component: 0 both, 1 the specular lobe alone, 2 the diffuse one alone -- bRec.component + 1; a component's
    // pdf times its pdfComponent IS its term of the mixture, phong.cpp:157-186,331-342)
    const int comp = __float_as_int(b1.z);
    const float dOn = comp == 1 ? 0.f : 1.f;
    if (comp == 2) lobe = 0.f;
    f = (mk3(b0.y, b0.z, b0.w) * ((e + 2.f) * INV_TWOPI_F * lobe) + kd * (INV_PI_F * dOn)) * cosWo;
    pdf = w * (lobe * (e + 1.f) * INV_TWOPI_F) + (1.f - w) * (INV_PI_F * cosWo * dOn);
    if (state && comp == 1 && pdf == 0.f && w > 0.f) {
      // the double's lobe is zero below 2^-1074; the factors beside it (w (e + 1) / 2 pi, 1 / l^2, the medium's pdf) move the
      // product's own underflow by a few tens of binades: a band of +-64 around it, and |alpha| within rounding of zero
      *state = (l2 > -1010.f ? 1u : 0u) | ((l2 > -1138.f && l2 <= -1010.f) || fabsf(alpha) <= 1e-6f ? 2u : 0u);
    }
    return true;
  }
  if (kind == GVPM_BSDF_WARD) {
    // src/bsdfs/ward.cpp:178-266, isotropic (alphaU == alphaV = b1.x), both components (roughness >= 0.05); H NOT normalised in
    // eval, as the reference has it; the variant rides in the field the rough conductor uses for its pdf's form
    const float al = b1.x, w = b1.y, ia2 = frcp(al * al);
    const int variant = __float_as_int(b1.w);
    const f3 H = wi + wo;
    const float HH = dot(H, H), Hz = cosWi + cosWo;
    const float E = __expf(-(HH - Hz * Hz) * frcp(Hz * Hz) * ia2);
    const float INV_FOURPI = 0.07957747154594766788f;
    float factor1;
    if (variant == GVPM_WARD_WARD) factor1 = INV_FOURPI * ia2 * frsq(cosWi * cosWo);
    else if (variant == GVPM_WARD_DUER) factor1 = INV_FOURPI * ia2 * frcp(cosWi * cosWo);
    else factor1 = HH * INV_PI_F * ia2 * frcp(Hz * Hz * Hz * Hz);
    const float specRef = factor1 * E;
    f = (mk3(b0.y, b0.z, b0.w) * (specRef > 1e-10f ? specRef : 0.f) + kd * INV_PI_F) * cosWo;
    // pdf: the normalised half vector; Hn . wi = (1 + wi . wo) / |H|, cos(theta_Hn) = Hz / |H|
    const float iH = frsq(HH), cH = Hz * iH, wiH = dot(wi, H) * iH;
    pdf = w * (INV_FOURPI * ia2 * E * frcp(wiH * cH * cH * cH)) + (1.f - w) * (INV_PI_F * cosWo);
    return true;
  }
  if (kind == GVPM_BSDF_ROUGHCONDUCTOR) {
    const float4 b2 = a.bsdfs[4 * bi + 2], b3 = a.bsdfs[4 * bi + 3];
    const float alpha = b1.x;
    const int ggx = __float_as_int(b1.z) == GVPM_MICROFACET_GGX, vis = __float_as_int(b1.w) != 0;
    f3 H = wi + wo;
    H = H * frsq(dot(H, H));
    const float cH = dot(H, n), wiH = dot(wi, H), woH = dot(wo, H);
    const float D = microfacetD(ggx, alpha, cH);
    if (D == 0.f) return true;  // eval and pdf both zero (pdfAll = D cos_H, pdfVisible = D G1 ...)
    const float G1i = microfacetG1(ggx, alpha, cosWi, wiH), G1o = microfacetG1(ggx, alpha, cosWo, woH);
    const float model = fdiv(D * G1i * G1o, 4.f * cosWi);
    f = mk3(fresnelConductor(wiH, b2.x, b2.w) * b0.y, fresnelConductor(wiH, b2.y, b3.x) * b0.z,
            fresnelConductor(wiH, b2.z, b3.y) * b0.w) * model;
    pdf = vis ? fdiv(D * G1i, 4.f * cosWi) : fdiv(D * cH, 4.f * fabsf(woH));
    return true;
  }
  return false;
}

// Phong::eval (x cos) and Phong::pdf x pdfComponent of a table entry in fp64 (phong.cpp:121-186,331-342): what the exact
// passes and the fp64 transcription of G-Beams evaluate a Phong parent with -- a lobe of exponent ~1000 lives where fp32 has
// no numbers.  False: not a Phong entry.
__device__ __forceinline__ bool phongEvalD(const GatherArgs &a, float index, d3 kd, d3 n, d3 wi, d3 wo, double cosWi, double cosWo,
                                           d3 &f, double &pdf) {
  const uint32_t bi = (uint32_t)index;
  if (!(index >= 0.f) || bi >= a.nbsdfs) return false;
  const float4 b0 = a.bsdfs[4 * bi], b1 = a.bsdfs[4 * bi + 1];
  if (__float_as_int(b0.x) != GVPM_BSDF_PHONG) return false;
  const double INV_PI = 0.31830988618379067154, INV_TWOPI = 0.15915494309189533577;
  const double e = b1.x, w = b1.y;
  const int comp = __float_as_int(b1.z);  // 0 both, 1 specular only, 2 diffuse only (gvpm_hip.h, gvpm_bsdf)
  const d3 refl = n * (2.0 * cosWi) - wi;
  const double alpha = dot(wo, refl);
  const double lobe = (alpha > 0 && comp != 2) ? pow(alpha, e) : 0.0, dOn = comp == 1 ? 0.0 : 1.0;
  f = (mkd(b0.y, b0.z, b0.w) * ((e + 2.0) * INV_TWOPI * lobe) + kd * (INV_PI * dOn)) * cosWo;
  pdf = w * (lobe * (e + 1.0) * INV_TWOPI) + (1.0 - w) * (INV_PI * cosWo * dOn);
  return true;
}

// shiftPhotonDiffuse + diffuseReconnection.  Returns the MIS weight, writes the shifted flux.
// Written branch-free apart from the shadow-ray loop: every early `return false` of the reference
// (shift_volume_photon.cpp:398-412, shift_diffuse.cpp:43-47, 100-104, :463-470) clears `good`, the
// arithmetic runs for all lanes and the result is selected at the end.
template <bool FULLVIS>
__device__ __forceinline__ float shiftDiffuse(const GatherArgs &a, const PhotonCold &ph,
                                              uint32_t bits, f3 dProjU, const RayReg &sh, const RayReg &base,
                                              uint32_t edge, f3 trShift, float pdfBaseRay, float pdfShiftRay,
                                              f3 &shiftedFlux, bool &ok, const float4 *ldsTri = nullptr,
                                              float sensorMisPre = -1.f, uint32_t *amb = nullptr) {
  const uint32_t ptype = GVPM_PF_PARENT_TYPE(bits);
  const float l2Proj = dot(dProjU, dProjU);
  const float lProj = fsqrt(l2Proj);
  const f3 dProj = dProjU * frcp(lProj);
  const float eps = a.cfg.epsilon, seps = a.cfg.shadow_epsilon;
  const float vmax = a.cfg.visibility_as_written ? lProj * seps : lProj * (1.f - seps);
  // (amb: the caller defers undecidable shifts to the exact pass; without one the plain fp32 decision stands)
  // bit 15 of the record's flags: the parent lies behind a wall it sits on, word 2 of its list is that wall's REACH
  // cstar = |delta| / Epsilon instead of entries (grid_build.hip, ownWall): the wall itself is not listed
  const bool behind = (bits >> 15) & 1u;
  const float cosWo = dot(ph.parentN, dProj);
  // (a segment that leaves within cstar of grazing meets the own wall's plane at t >= Epsilon: the exact pass decides; 1.2e-3:
  // the parent's normal against the triangle's, ownWall's parallel test.  Decided BEFORE the visibility loop: the reach is
  // not carried across it)
  const bool ownAmb = behind && cosWo > 0.f && cosWo <= __uint_as_float(ph.nl2) + 1.2e-3f;
  const int vis = shadowBlocked<FULLVIS>(a, ldsTri, ph.nl0, ph.nl1, behind ? 0xFFFFFFFFu : ph.nl2, ph.parentPos, dProj, eps, vmax);
  bool good = vis == GVPM_TRI_MISS;
#ifdef GVPM_DBG_SHIFT2  // (probe builds, on a single pair: scripts/probes_py/vpm_bisect.py)
  printf("shiftDiffuse vis %d ownAmb %d behind %d cosWo %g lProj %.9g vmax %.9g lists %08x %08x %08x parent %.9g %.9g %.9g dir %.9g %.9g %.9g\n", vis,
         (int)ownAmb, (int)behind, cosWo, lProj, vmax, ph.nl0, ph.nl1, ph.nl2, ph.parentPos.x, ph.parentPos.y, ph.parentPos.z, dProj.x, dProj.y,
         dProj.z);
#endif
  // (the sign / cosine tests below flip within fp32 rounding of a grazing direction)
  if (amb)
    *amb = (((vis & GVPM_TRI_AMB) || ownAmb) ? 16u : 0u) |
           ((GVPM_PF_PARENT_TYPE(bits) != GVPM_PARENT_MEDIUM && fabsf(cosWo) <= 2e-6f) ? 32u : 0u);
  // surface / emitter parents: the offset direction must leave on the side the photon left (sign of
  // dot(n, dProj) / dot(n, -wi))
  const bool isMedium = ptype == GVPM_PARENT_MEDIUM, isGlossy = ptype == GVPM_PARENT_SURFACE_BSDF;
  const bool isSurface = ptype == GVPM_PARENT_SURFACE || isGlossy;
  good = good && (isMedium || cosWo * dot(ph.parentN, -ph.wi) >= 0.f);
  // eval / pdf of the parent towards the offset position (diffuse.cpp:110-127, phase eval, area.cpp:132-150)
  const float cosWi = dot(ph.parentN, ph.parentWi);
  good = good && (!isSurface || (cosWi > 0.f && cosWo > 0.f));  // eval/pdf = 0 or the shading-normal reject
  const float lam = INV_PI_F * fmaxf(cosWo, 0.f);
  const float pMed = phaseEval(ph.parentG, ph.parentWi, dProj);
  float pdfValue = isMedium ? pMed : lam;
  f3 thr = isSurface ? ph.parentScat * lam : (isMedium ? ph.parentScat * pMed : mk3(lam));
  bool pdfTiny = false;  // (glossyParentEval: the pdf underflowed HERE, not in the reference's double)
  if (isGlossy) {
    // (a branch of its own: scenes without glossy walls pay one wave-uniform test)
    uint32_t gst = 0u;
    if (!glossyParentEval(a, ph.parentG, ph.parentScat, ph.parentN, ph.parentWi, dProj, cosWi, cosWo, thr, pdfValue, &gst)) good = false;
    pdfTiny = (gst & 1u) != 0u;
    if (amb && (gst & 2u)) *amb |= 32u;
  }
  const float gop = frcp(l2Proj);
  float sPdf = pdfValue * gop;
  good = good && ph.parentPdf != 0.f;
  thr = thr * (gop * ph.parentRR * frcp(ph.parentPdf));
  if (GVPM_PF_EDGE_IN_MEDIUM(bits)) {
    f3 tr;
    float pdfSuccess;
    mediumEval(a.med, lProj, tr, pdfSuccess);
    sPdf *= pdfSuccess;
    thr = thr * tr * frcp(ph.edgePdf);
  }
  good = good && (sPdf != 0.f || pdfTiny);
  const f3 photonWeight = ph.prefixW * thr;
  const f3 sigS = mk3(a.med.sigmaS[0], a.med.sigmaS[1], a.med.sigmaS[2]);
  const f3 contrib = sigS * photonWeight * phaseEval(a.med.g, -dProj, -sh.d);
  float w = 0.5f;
  bool misOk = true;
  if (a.cfg.use_mis) {
    const float basePdf = pdfBaseRay * ph.parentPdf * ph.edgePdf;
    const float offsetPdf = sPdf * pdfShiftRay;
    misOk = !((offsetPdf == 0.f && !(pdfTiny && pdfShiftRay != 0.f)) || basePdf == 0.f);
    // (sensorMisPre: the caller's per-(shift, beam) value, when it keeps one)
    const float v = (sensorMisPre >= 0.f ? sensorMisPre : sensorMIS(sh, base, edge)) * fdiv(offsetPdf, basePdf);
    w = a.cfg.power_heuristic ? frcp(1.f + v * v) : frcp(1.f + v);
  }
  // a failed MIS keeps the flux it computed and takes weight 1 (shift_volume_photon.cpp:463-470)
  shiftedFlux = good ? trShift * contrib * sh.eye : mk3(0.f);  // jacobian == 1
  ok = good && misOk;
  return ok ? w : 1.f;
}

// A shift that needs the manifold walk (shiftPhotonManifold, shift_volume_photon.cpp:160-295): what the walk reads goes
// to the host's request list, what the device needs to finish the shift once the host has answered (:217-279) stays beside
// it.  Rare and register hungry: not inlined -- and handed the list BY VALUE: a `const GatherArgs &` here makes the kernel
// keep its whole argument block in scratch (a 600-byte frame, every a.field a scratch load).  False: the list is full --
// a failed shift.
struct ReqSink {
  gvpm_shift_request *host;
  float4 *ctx;
  uint32_t *count;
  uint32_t cap;
  const uint32_t *origIdx;
};
__device__ __forceinline__ ReqSink reqSink(const GatherArgs &a) { return ReqSink{a.reqHost, a.reqCtx, a.reqCount, a.reqCap, a.origIdx}; }

static __device__ __noinline__ bool recordShiftRequest(ReqSink a, float radius, uint32_t pidx, uint32_t set, int i, f3 offsetPos, f3 basePt,
                                                       f3 shiftPt, float tPrime, float tr, float pdfCam, float pdfShiftPos, float sMIS,
                                                       float scale, f3 bc, f3 shD, f3 eye, uint32_t pix) {
  const uint32_t slot = atomicAdd(a.count, 1u);
  if (slot >= a.cap) return false;
  gvpm_shift_request rq;
  rq.photon = a.origIdx[pidx];
  rq.set = set;
  rq.shift = (uint32_t)i;
  rq.reserved = 0u;
  rq.offset_pos[0] = offsetPos.x; rq.offset_pos[1] = offsetPos.y; rq.offset_pos[2] = offsetPos.z;
  rq.radius = radius;  // (G-VPM: the pixel's own radius)
  rq.base_point[0] = basePt.x; rq.base_point[1] = basePt.y; rq.base_point[2] = basePt.z;
  rq.t = tPrime;
  rq.shift_point[0] = shiftPt.x; rq.shift_point[1] = shiftPt.y; rq.shift_point[2] = shiftPt.z;
  rq.reserved2 = 0.f;
  a.host[slot] = rq;
  float4 *c = a.ctx + 4 * (size_t)slot;
  c[0] = make_float4(tr, pdfCam, pdfShiftPos, sMIS);
  c[1] = make_float4(scale, bc.x, bc.y, bc.z);
  c[2] = make_float4(shD.x, shD.y, shD.z, __uint_as_float(pix));
  c[3] = make_float4(eye.x, eye.y, eye.z, __uint_as_float((uint32_t)i));
  return true;
}

// computeVolumeContribution (gvpm/shift/shift_utilities.h:231-253) and the debugShift filter
// (shift_volume_photon.cpp:680-687) depend only on the photon and the configuration: fold
// them into bit 6 of the hot record.
__device__ __forceinline__ bool photonContributes(uint32_t flags, const gvpm_params &cfg) {
  const int mode = cfg.lighting_interaction_mode;
  const uint32_t ptype = GVPM_PF_PARENT_TYPE(flags);
  if (!((mode & GVPM_SURF2MEDIA) && (mode & GVPM_MEDIA2MEDIA))) {
    if (ptype == GVPM_PARENT_MEDIUM && !(mode & GVPM_MEDIA2MEDIA)) return false;
    if (ptype != GVPM_PARENT_MEDIUM && !(mode & GVPM_SURF2MEDIA)) return false;
  }
  const int compo = (int)GVPM_PF_PREV_COMPONENT(flags);
  if (cfg.bsdf_interaction_mode != GVPM_BSDF_ALL && compo > 0 && !(compo & cfg.bsdf_interaction_mode)) return false;
  if (cfg.debug_shift != GVPM_SHIFT_ALL && cfg.debug_shift != GVPM_SHIFT_NULL) {
    int st;
    switch (GVPM_PF_SHIFT_TYPE(flags)) {
      case 1: st = GVPM_SHIFT_DIFFUSE; break;
      case 2: st = GVPM_SHIFT_MEDIUM; break;
      case 3: st = GVPM_SHIFT_MANIFOLD; break;
      default: st = GVPM_SHIFT_INVALID; break;
    }
    if (cfg.debug_shift != st) return false;
  }
  return true;
}


}  // namespace gvpm
