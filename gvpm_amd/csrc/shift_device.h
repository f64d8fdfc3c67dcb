// Device helpers shared by the gather kernels: the closed-form pieces of the shift
// (phase / medium / occluder / reconnection / MIS) -- see gather_bre.hip for the citations.
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"
#include "vec.h"

namespace gvpm {

#define INV_PI_F 0.31830988618379067154f
#define INV_FOURPI_F 0.07957747154594766788f

constexpr int MAXTRI_LDS = 32;

struct RayReg {
  f3 o, d, eye;
  float len, pdf, jac, gop;
  bool valid;
};

__device__ __forceinline__ float phaseEval(float g, f3 wi, f3 wo) {
  if (g == 0.f) return INV_FOURPI_F;
  const float temp = 1.0f + g * g + 2.0f * g * dot(wi, wo);
  return INV_FOURPI_F * (1.f - g * g) / (temp * sqrtf(temp));
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

// Moeller-Trumbore, triangle.h:109-145 + interval test skdtree.h:318-320
__device__ __forceinline__ bool triHit(f3 v0, f3 e1, f3 e2, f3 o, f3 d, float mint, float maxt) {
  const f3 pvec = cross(d, e2);
  const float det = dot(e1, pvec);
  if (det == 0.f) return false;
  const float inv = 1.0f / det;
  const f3 tvec = o - v0;
  const float u = dot(tvec, pvec) * inv;
  if (u < 0.f || u > 1.f) return false;
  const f3 qvec = cross(tvec, e1);
  const float v = dot(d, qvec) * inv;
  if (v >= 0.f && u + v <= 1.f) {
    const float t = dot(e2, qvec) * inv;
    return t >= mint && t <= maxt;
  }
  return false;
}

// scene->rayIntersect(ray), any-hit over the occluder list.  Triangles sit in LDS (broadcast
// reads); a plane-distance early-out skips triangles the segment [mint,maxt] cannot reach.
__device__ __forceinline__ bool anyHit(const GatherArgs &a, const float4 (*tri)[3], f3 o, f3 d, float mint, float maxt) {
  bool hit = false;
  const uint32_t nl = min(a.ntri, (uint32_t)MAXTRI_LDS);
  for (uint32_t i = 0; i < nl; ++i) {
    const float4 t0 = tri[i][0], t1 = tri[i][1], t2 = tri[i][2];
    const f3 v0 = mk3(t0.x, t0.y, t0.z), n = mk3(t0.w, t1.w, t2.w);
    const float dist0 = dot(n, o - v0);
    const float dn = dot(n, d);
    // signed plane distances at both segment ends; no sign change (with margin) => no hit
    const float da = dist0 + mint * dn, db = dist0 + maxt * dn;
    const float margin = 1e-4f * (fabsf(dist0) + maxt) + 1e-7f;
    if ((da > margin && db > margin) || (da < -margin && db < -margin)) continue;
    if (triHit(v0, mk3(t1.x, t1.y, t1.z), mk3(t2.x, t2.y, t2.z), o, d, mint, maxt)) hit = true;
  }
  for (uint32_t i = nl; i < a.ntri; ++i) {
    const f3 v0 = mk3(a.triV0[3 * i], a.triV0[3 * i + 1], a.triV0[3 * i + 2]);
    const f3 e1 = mk3(a.triE1[3 * i], a.triE1[3 * i + 1], a.triE1[3 * i + 2]);
    const f3 e2 = mk3(a.triE2[3 * i], a.triE2[3 * i + 1], a.triE2[3 * i + 2]);
    if (triHit(v0, e1, e2, o, d, mint, maxt)) hit = true;
  }
  return hit;
}

// As written (shift_volume_photon.cpp:396) the shadow segment is [Epsilon, lProj*ShadowEpsilon]
// from the photon's parent: only occluders within that distance of the parent can be hit.  The
// grid build lists them per photon (reorder_kernel), so the any-hit loop touches 0-4 triangles.
__device__ __forceinline__ bool shadowBlocked(const GatherArgs &a, const float4 (*tri)[3], uint32_t nearList, f3 o,
                                              f3 d, float mint, float maxt) {
  if (!a.cfg.visibility_as_written || (nearList >> 24) == 0xFEu) return anyHit(a, tri, o, d, mint, maxt);
  bool hit = false;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t i = (nearList >> (8 * k)) & 0xFFu;
    if (i == 0xFFu) continue;
    f3 v0, e1, e2;
    if (i < (uint32_t)MAXTRI_LDS) {
      const float4 t0 = tri[i][0], t1 = tri[i][1], t2 = tri[i][2];
      v0 = mk3(t0.x, t0.y, t0.z); e1 = mk3(t1.x, t1.y, t1.z); e2 = mk3(t2.x, t2.y, t2.z);
    } else {
      v0 = mk3(a.triV0[3 * i], a.triV0[3 * i + 1], a.triV0[3 * i + 2]);
      e1 = mk3(a.triE1[3 * i], a.triE1[3 * i + 1], a.triE1[3 * i + 2]);
      e2 = mk3(a.triE2[3 * i], a.triE2[3 * i + 1], a.triE2[3 * i + 2]);
    }
    if (triHit(v0, e1, e2, o, d, mint, maxt)) hit = true;
  }
  return hit;
}

// GatherPoint::sensorMIS, gvpm_struct.h:608-631 (sDist == bDist for BRE: same t')
__device__ __forceinline__ float sensorMIS(const RayReg &s, const RayReg &b, uint32_t edge) {
  float jacobian = s.jac;
  float ratio = s.pdf / b.pdf;
  if (edge != 1u) {
    jacobian *= s.gop / b.gop;
    ratio *= b.gop / s.gop;
  }
  return ratio * jacobian;
}

struct PhotonCold {
  f3 wi, flux, parentPos, parentN, prefixW, parentScat, parentWi;
  float parentPdf, edgePdf, parentRR, parentG;
  uint32_t nearList;  // up to 4 occluder indices near the parent (0xFF = none); top byte 0xFE: overflow
};

__device__ __forceinline__ PhotonCold loadCold(const GatherArgs &a, uint32_t idx) {
  PhotonCold c;
  const float4 c0 = a.cold[0 * (size_t)a.nph + idx];
  const float4 c1 = a.cold[1 * (size_t)a.nph + idx];
  const float4 c2 = a.cold[2 * (size_t)a.nph + idx];
  const float4 c3 = a.cold[3 * (size_t)a.nph + idx];
  const float4 c4 = a.cold[4 * (size_t)a.nph + idx];
  const float4 c5 = a.cold[5 * (size_t)a.nph + idx];
  const float4 c6 = a.cold[6 * (size_t)a.nph + idx];
  c.wi = mk3(c0.x, c0.y, c0.z); c.parentPdf = c0.w;
  c.flux = mk3(c1.x, c1.y, c1.z); c.edgePdf = c1.w;
  c.parentPos = mk3(c2.x, c2.y, c2.z); c.parentRR = c2.w;
  c.parentN = mk3(c3.x, c3.y, c3.z); c.parentG = c3.w;
  c.prefixW = mk3(c4.x, c4.y, c4.z);
  c.nearList = __float_as_uint(c4.w);
  c.parentScat = mk3(c5.x, c5.y, c5.z);
  c.parentWi = mk3(c6.x, c6.y, c6.z);
  return c;
}

// shiftPhotonDiffuse + diffuseReconnection.  Returns the MIS weight, writes the shifted flux.
__device__ __forceinline__ float shiftDiffuse(const GatherArgs &a, const float4 (*tri)[3], const PhotonCold &ph,
                                              uint32_t bits, f3 dProjU, const RayReg &sh, const RayReg &base,
                                              uint32_t edge, f3 trShift, float pdfBaseRay, float pdfShiftRay,
                                              f3 &shiftedFlux, bool &ok) {
  ok = false;
  shiftedFlux = mk3(0.f);
  const uint32_t ptype = GVPM_PF_PARENT_TYPE(bits);
  const float lProj = sqrtf(dot(dProjU, dProjU));
  const f3 dProj = dProjU * (1.f / lProj);
  const float eps = a.cfg.epsilon, seps = a.cfg.shadow_epsilon;
  const float vmax = a.cfg.visibility_as_written ? lProj * seps : lProj * (1.f - seps);
  if (shadowBlocked(a, tri, ph.nearList, ph.parentPos, dProj, eps, vmax)) return 1.f;
  if (ptype != GVPM_PARENT_MEDIUM) {
    const float signDot = dot(ph.parentN, dProj) / dot(ph.parentN, -ph.wi);
    if (signDot < 0.f) return 1.f;
  }
  f3 thr;
  float pdfValue;
  if (ptype == GVPM_PARENT_SURFACE) {
    const float cosWo = dot(ph.parentN, dProj), cosWi = dot(ph.parentN, ph.parentWi);
    if (cosWi <= 0.f || cosWo <= 0.f) return 1.f;  // eval/pdf = 0 or the shading-normal reject: sRec.pdf == 0
    thr = ph.parentScat * (INV_PI_F * cosWo);
    pdfValue = INV_PI_F * cosWo;
  } else if (ptype == GVPM_PARENT_MEDIUM) {
    const float p = phaseEval(ph.parentG, ph.parentWi, dProj);
    thr = ph.parentScat * p;
    pdfValue = p;
  } else {
    float dp = dot(dProj, ph.parentN);
    if (dp < 0.f) dp = 0.f;
    thr = mk3(INV_PI_F * dp);
    pdfValue = INV_PI_F * dp;
  }
  const float gop = 1.f / (lProj * lProj);
  float sPdf = pdfValue * gop;
  thr = thr * gop;
  if (ph.parentPdf == 0.f) return 1.f;
  thr = thr * (ph.parentRR / ph.parentPdf);
  if (GVPM_PF_EDGE_IN_MEDIUM(bits)) {
    f3 tr;
    float pdfSuccess;
    mediumEval(a.med, lProj, tr, pdfSuccess);
    sPdf *= pdfSuccess;
    thr = thr * tr * (1.f / ph.edgePdf);
  }
  if (sPdf == 0.f) return 1.f;
  const f3 photonWeight = ph.prefixW * thr;
  const f3 sigS = mk3(a.med.sigmaS[0], a.med.sigmaS[1], a.med.sigmaS[2]);
  const f3 contrib = sigS * photonWeight * phaseEval(a.med.g, -dProj, -sh.d);
  shiftedFlux = trShift * contrib * sh.eye;  // jacobian == 1
  ok = true;
  float w = 0.5f;
  if (a.cfg.use_mis) {
    const float basePdf = pdfBaseRay * ph.parentPdf * ph.edgePdf;
    const float offsetPdf = sPdf * pdfShiftRay;
    if (offsetPdf == 0.f || basePdf == 0.f) {
      ok = false;
      return 1.f;
    }
    const float v = sensorMIS(sh, base, edge) * (offsetPdf / basePdf);
    w = a.cfg.power_heuristic ? 1.f / (1.f + v * v) : 1.f / (1.f + v);
  }
  return w;
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
