// fp32 evaluation of one (camera ray, sub-beam) pair for the G-Beams kernels: the same functions as the fp64
// transcription in gather_beams.hip (BeamKernelRecord, shiftNull3D, shiftBeamDiffuse, kernelPDF, getShiftPos*),
// re-derived in a LOCAL frame so that single precision is enough.
//
// Why a local frame.  The reference works on absolute coordinates (scene extent ~10^3, kernel radius ~1): the
// cylinder quadratic  C = ox^2 + oy^2 - r^2  then cancels six digits and fp32 would leave ~1 % noise on the chord
// ends.  Here every point is expressed relative to the sub-beam's centre Cb, which the traversal already placed
// within radius + half a sub-beam of the camera ray, and every ray through its foot point A = o + sC*d with
// sC = (Cb - o).d.  The only large-operand reductions -- Cb - o, sC and the perpendicular offset D0 = Cb - A, per
// ray -- are done in fp64 (a dozen FMAs) and rounded once; everything after that has operands of the order of the
// kernel radius.  Decisions that determine WHICH sub-beam evaluates a pair (ownership) are flagged when they fall
// inside the fp32 error band and are then settled by the fp64 transcription (beamOwnerExact), so the evaluated set
// is the reference's; the contribution itself agrees with the fp64 path to ~1e-6 relative.
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"
#include "shift_device.h"
#include "vec.h"

namespace gvpm {

struct BeamF {
  f3 p1, p2, bd;
  float len;
  f3 flux, prefixW, parentScat, parentN, parentWi, endN;
  float parentPdf, parentRR, parentG;
  uint32_t flags;
  uint32_t nl0, nl1, nl2;  // occluders near the beam (beam_near_kernel): 12 byte indices, 0xFF = none, top byte 0xFE: all
  bool endOnSurface;
};

__device__ __forceinline__ BeamF loadBeamF(const GatherArgs &a, uint32_t idx) {
  // one 128-byte record (grid_build.hip, beam_cold_kernel)
  const float4 *rec = a.cold + (size_t)idx * GVPM_REC_QUADS;
  const float4 c1 = rec[0], c2 = rec[1], c3 = rec[2], c4 = rec[3], c5 = rec[4], c6 = rec[5], c7 = rec[6], c8 = rec[7];
  BeamF b;
  b.parentPdf = c1.w;
  b.flux = mk3(c1.x, c1.y, c1.z);
  b.p1 = mk3(c2.x, c2.y, c2.z); b.parentRR = c2.w;
  b.parentN = mk3(c3.x, c3.y, c3.z); b.parentG = c3.w;
  b.prefixW = mk3(c4.x, c4.y, c4.z); b.nl0 = __float_as_uint(c4.w);
  b.parentScat = mk3(c5.x, c5.y, c5.z); b.nl1 = __float_as_uint(c5.w);
  b.parentWi = mk3(c6.x, c6.y, c6.z); b.nl2 = __float_as_uint(c6.w);
  b.p2 = mk3(c7.x, c7.y, c7.z); b.flags = __float_as_uint(c7.w);
  b.endN = mk3(c8.x, c8.y, c8.z);
  b.endOnSurface = !(c8.x == 0.f && c8.y == 0.f && c8.z == 0.f);
  // PhotonBeam::setEndPoint, pm/beams_struct.h:73-81: the length (hence the sub-beam count) is the fp64 path's,
  // rounded once at the build (beam_cold_kernel)
  b.len = c8.w;
  b.bd = tof(tod(b.p2) - tod(b.p1)) * frcp(b.len);
  return b;
}

// A ray seen from the local origin Cb: foot point A = o + s0*d, D0 = Cb - A (perpendicular to d)
struct LocalRay {
  f3 d, D0;
  double s0;     // parameter of the foot point
  float s0f;
  float mint, maxt;
};
__device__ __forceinline__ LocalRay localRay(f3 o, f3 d, float mint, float maxt, f3 Cb) {
  LocalRay r;
  const d3 c = tod(Cb) - tod(o), dd = tod(d);
  r.s0 = dot(c, dd);
  r.D0 = tof(c - dd * r.s0);
  r.d = d;
  r.s0f = (float)r.s0;
  r.mint = mint;
  r.maxt = maxt;
  return r;
}
// point of the ray at absolute parameter (s0 + sigma), relative to Cb
__device__ __forceinline__ f3 atLocal(const LocalRay &r, float sigma) { return r.d * sigma - r.D0; }

// cylinderIntersection (pm/beams_3d_intersections.h:77-140) for a view line V + t*dv against the cylinder of
// radius r around A + z*da, z in [z0, z1]; rel = V - A.  tLo / tHi: the view ray's 0 and maxt, measured from V.
//
// amb (round 5, optional): set when one of the function's DECISIONS -- the discriminant's sign, the roots against
// [tLo, tHi], the ends' heights against [z0, z1] -- lies within the fp32 error of its operands (the caller then hands the
// shift to the exact pass -- gather_beams.hip, exact_beams_kernel -- and the return value is not used).  The error model,
// with u = 2^-24 and a safety factor of ~20 on every term: the dot products are good to ~4u of the products' magnitudes, so
// disc = Bh^2 - A C carries ~4e-7 (Bh^2 + R2), R2 = |rel|^2 + r^2 (A <= 1; its own rounding times |C| <= R2 included);
// sqrt turns that into eq = err(disc) / (2 sqrt(disc)) on q; x0 = q / A adds A's rounding over A; x1 = C / q adds C's.
__device__ __forceinline__ bool cylLocal(f3 rel, f3 dv, f3 da, float z0, float z1, float r, float tLo, float tHi,
                                         float &tN, float &tF, bool *amb = nullptr) {
  const float dd = dot(dv, da);
  const float rz = dot(rel, da);
  const float A = 1.f - dd * dd;
  const float Bh = dot(rel, dv) - rz * dd;
  const float R2 = dot(rel, rel) + r * r;
  const float C = dot(rel, rel) - rz * rz - r * r;
  const float disc = Bh * Bh - A * C;
  const float S = Bh * Bh + R2;
  if (amb && (A < 1e-6f || fabsf(disc) <= 2e-5f * S)) {
    *amb = true;
    return false;
  }
  if (!(A > 0.f) || !(disc > 0.f)) return false;  // lines farther apart than r (or parallel)
  const float sq = fsqrt(disc);
  const float q = Bh < 0.f ? (sq - Bh) : -(Bh + sq);
  float x0 = fdiv(q, A), x1 = fdiv(C, q);
  float tE = 0.f;
  if (amb) {
    const float eq = fdiv(2e-6f * S, sq), iA = frcp(A), iq = frcp(fabsf(q));
    tE = fmaxf((eq + 4e-6f * fabsf(x0)) * iA, (2e-6f * R2 + fabsf(x1) * eq) * iq) + 2e-6f * (fabsf(x0) + fabsf(x1));
  }
  if (x0 > x1) { const float t = x0; x0 = x1; x1 = t; }
  tN = x0;
  tF = x1;
  // (strict comparisons: an infinite bound -- tHi of a new beam, z1 of the last sub-beam -- is near nothing, inf < inf is false)
  if (amb && (fabsf(tN - tHi) < tE + 1e-6f * fabsf(tHi) || fabsf(tF - tLo) < tE + 1e-6f * fabsf(tLo))) *amb = true;
  if (tN > tHi || tF < tLo) return false;
  const float zN = rz + dd * tN, zF = rz + dd * tF;
  if (amb) {
    const float zE = tE + 2e-6f * (fabsf(rz) + fabsf(tN) + fabsf(tF));
    const float e0 = zE + 1e-6f * fabsf(z0), e1 = zE + 1e-6f * fabsf(z1);
    if (fabsf(zN - z0) < e0 || fabsf(zF - z0) < e0 || fabsf(zN - z1) < e1 || fabsf(zF - z1) < e1) *amb = true;
  }
  if (zN < z0) {
    if (zF < z0) return false;
    tN = tN + (tF - tN) * fdiv(zN - z0, zN - zF);
    return true;
  } else if (zN < z1) {
    return true;
  } else if (zN > z1) {
    if (zF > z1) return false;
    tN = tN + (tF - tN) * fdiv(zN - z1, zN - zF);
    return true;
  }
  return false;
}

// Scale of the pair's LOCAL frame: the kernel, the sub-beam, the base ray's foot all lie within it of the local origin
__device__ __forceinline__ float beamLocalScale(const GatherArgs &a) { return 4.f * a.kernelRadius + 2.f * a.subLen; }
// "x < y" between squared lengths of LOCAL vectors cannot be trusted: m2 = the sum of the squared magnitudes of the operands
// the compared vector was summed from.  A vector summed from operands of total magnitude M is good to ~4u M, its square to
// 2 |v| 4u M <= 2.4e-7 (|v|^2 + M^2) and M^2 <= 3 m2 for three operands: 1e-5 leaves a factor of ten.
__device__ __forceinline__ bool nearSq(float x, float y, float m2) { return fabsf(x - y) <= 1e-5f * (x + y + m2); }

struct MRecF {
  float tr, pdfFailure;
};
__device__ __forceinline__ MRecF mediumEvalF(const MediumDev &m, float dist) {
  MRecF r;
  float e = __expf(-m.sigmaT[0] * dist);
  r.pdfFailure = e * m.msw + (1.f - m.msw);
  if (e < 1e-20f) e = 0.f;
  r.tr = e;
  return r;
}

// coordinateSystemCoherent, util.cpp:592-599
__device__ __forceinline__ void coordSysCoherentF(f3 n, f3 &b1, f3 &b2) {
  const float sign = copysignf(1.0f, n.z);
  const float aa = -frcp(sign + n.z);
  const float bb = n.x * n.y * aa;
  b1 = mk3(1.0f + sign * n.x * n.x * aa, sign * bb, -sign * n.x);
  b2 = mk3(bb, sign + n.y * n.y * aa, -n.y);
}

struct KRecF {
  float tauV;     // v - tc
  float v, w;     // absolute parameters on the beam / on the camera ray
  float sigmaW;   // w - (camera foot parameter)
  float pdfKernel, pdfEdgeFailure, u, weightKernel;
  float sc;       // contrib = flux * sigma_s * sc
  f3 contrib;
};

// shift(), shift_volume_beams.cpp:47-79: the point at distance u from the line r (at parameter w) in the plane
// through the line and `a`, on a's side (phi = pi/2 - asin(u/|ly|): cos phi = u/|ly|, sin phi = +-sqrt(1 - cos^2)).
// aRel = a - (foot point of r); returns the point relative to that foot point.
__device__ __forceinline__ f3 shiftPointLocal(f3 dr, f3 aRel, float u, float sigma, bool flip) {
  const f3 av = aRel - dr * dot(aRel, dr);
  const float ly = fsqrt(dot(av, av));
  const f3 sv = av * frcp(ly);
  const f3 tv = cross(dr, sv);
  const float x = fminf(1.f, fmaxf(-1.f, fdiv(u, ly)));
  float sn = fsqrt(fmaxf(0.f, 1.f - x * x));
  if (flip) sn = -sn;
  return dr * sigma + sv * (u * x) + tv * (u * sn);
}

// What shiftPointLocal's sine can be off by when the distance ly of `a` to the line is good to dly: sn = sqrt(1 - x^2), x = u / ly,
// has slope x / sn -- unbounded at x = 1, where the clamp takes over (both sides of it must agree with the reference's double:
// a clamp that is certain costs nothing).
__device__ __forceinline__ float shiftSinErr(f3 dr, f3 aRel, float u, float dly) {
  const f3 av = aRel - dr * dot(aRel, dr);
  const float ly = fsqrt(dot(av, av));
  const float xr = fdiv(u, ly), dx = fdiv(dly, ly) * fminf(xr, 2.f);
  if (xr > 1.f + dx) return 0.f;
  const float x = fminf(1.f, xr), sn2 = fmaxf(0.f, 1.f - x * x);
  return sn2 > 4.f * dx ? x * dx * frsq(sn2) : fsqrt(6.f * dx);
}

}  // namespace gvpm
