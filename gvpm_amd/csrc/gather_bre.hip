// G-BRE gather + gradient-domain shift for gfx950 (CDNA4), hand-written HIP.
//
// Replaces, for one SPPM iteration, the body of
//   GPMIntegrator::computeVolumeGradientPhotonBRE   gvpm/gvpm.cpp:988-1079
//   GradientBeamRadianceEstimator::query            gvpm/gvpm_accel.h:268-312
//   VolumeGradientBREQuery::operator()              gvpm/shift/shift_volume_photon.cpp:658-856
//   shiftNull / shiftPhoton / shiftPhotonDiffuse    shift_volume_photon.cpp:49-158,382-486
//   diffuseReconnection                             gvpm/shift/operation/shift_diffuse.cpp:11-134
//   getShiftPos                                     shift_volume_photon.cpp:858-896
//   GatherPoint::sensorMIS                          gvpm/gvpm_struct.h:608-631
//   HomogeneousMedium::eval, phase eval             src/medium/homogeneous.cpp:432-513, src/phase/*.cpp
//
// Execution model (one 64-lane wave per workgroup, no MFMA: gather / divergent math):
//   * a TILE is a bundle of B camera-beam sets (B = 16/32/64, 64/B lanes per beam) that the
//     tile sort made spatially coherent (8x8 / 8x4 / 4x4 pixel tiles);
//   * the photon map is a uniform grid (all photons share one radius, gvpm.cpp:989) sorted by
//     cell with x fastest; a tile walks the grid in thick slabs along its major axis,
//     wave-reduces the fattened footprint of its beams into a cell box and turns the box into
//     x-contiguous photon ranges;
//   * plan_kernel walks every tile once WITHOUT touching photons (cellStart differences only) and
//     cuts it into work items of roughly equal candidate count -- the load balancer that replaces
//     BlockScheduler's dynamic image blocks (photonmapper/utilities/block_sched.h:87-113);
//   * gather_bre_kernel is persistent: waves pull items from an atomic queue; for each slab
//     step the 16-byte hot photon records of the ranges are copied coalesced into an LDS stage
//     (photons staged into LDS tiles) and every lane tests them against its own beam (LDS
//     broadcast reads): fp32 conservative pre-test, then the reference predicate in fp64
//     without contraction, so the hit set equals the fp64 oracle's bit for bit;
//   * hits are compacted with __ballot / popcount prefix sums into an LDS ring of
//     (photon, beam) pairs; whenever 64 are pending every lane evaluates one of them: base
//     contribution + 4 shifts (null shift, or offset-path reconnection with shadow ray,
//     Jacobian and MIS weight) and adds 27 partial sums to the beam's LDS accumulators.
#include <hip/hip_runtime.h>

#include "device_types.h"
#include "vec.h"

namespace gvpm {

#define INV_PI_F 0.31830988618379067154f
#define INV_FOURPI_F 0.07957747154594766788f

constexpr int RAYF = 13;  // o3 d3 len(<0: invalid) pdf eye3 jac gop
constexpr int STAGE = 256;
constexpr int QCAP = 128;
constexpr int MAXTRI_LDS = 32;
constexpr int PLAN_MAX_ITEMS = 64;  // per tile chunk

template <int B> struct TileLds {
  float ray[5][RAYF][B];
  float acc[27][B];
  float4 stage[STAGE];
  uint32_t stageIdx[STAGE];
  uint2 queue[QCAP];
  float4 tri[MAXTRI_LDS][3];  // {v0,n.x} {e1,n.y} {e2,n.z}
  float rnd[B];
  uint32_t pix[B];
  uint32_t edge[B];
};

struct RayReg {
  f3 o, d, eye;
  float len, pdf, jac, gop;
  bool valid;
};

template <int B> __device__ __forceinline__ RayReg loadRay(const TileLds<B> &s, int k, int b) {
  RayReg r;
  r.o = mk3(s.ray[k][0][b], s.ray[k][1][b], s.ray[k][2][b]);
  r.d = mk3(s.ray[k][3][b], s.ray[k][4][b], s.ray[k][5][b]);
  const float l = s.ray[k][6][b];
  r.len = fabsf(l);
  r.valid = l >= 0.f;
  r.pdf = s.ray[k][7][b];
  r.eye = mk3(s.ray[k][8][b], s.ray[k][9][b], s.ray[k][10][b]);
  r.jac = s.ray[k][11][b];
  r.gop = s.ray[k][12][b];
  return r;
}

// ---- exact predicate (fp64, no contraction): gvpm_accel.h:279-301 + aabb.h:310-340 ----
struct HitGeom {
  double disk, distSqr;
};

// diskDistance / distSqr of gvpm_accel.h:296-299 in the reference's operation order
__device__ __forceinline__ HitGeom hitGeom(f3 pf, f3 of, f3 df) {
#pragma clang fp contract(off)
  const double px = pf.x, py = pf.y, pz = pf.z;
  const double ox = of.x, oy = of.y, oz = of.z;
  const double dx = df.x, dy = df.y, dz = df.z;
  const double cx = px - ox, cy = py - oy, cz = pz - oz;
  HitGeom g;
  g.disk = cx * dx + cy * dy + cz * dz;
  const double qx = ox + dx * g.disk, qy = oy + dy * g.disk, qz = oz + dz * g.disk;
  const double vx = qx - px, vy = qy - py, vz = qz - pz;
  g.distSqr = vx * vx + vy * vy + vz * vz;
  return g;
}

// own sphere box vs ray segment: what every ancestor AABB of the reference BVH implies
__device__ __forceinline__ bool ownBoxHit(f3 pf, f3 of, f3 df, d3 rcp, double mint, double maxt, double radius) {
#pragma clang fp contract(off)
  double nearT = -INFINITY, farT = INFINITY;
  const double o3[3] = {of.x, of.y, of.z}, dd[3] = {df.x, df.y, df.z}, c3[3] = {pf.x, pf.y, pf.z};
  const double r3[3] = {rcp.x, rcp.y, rcp.z};
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const double minVal = c3[i] - radius, maxVal = c3[i] + radius;
    if (dd[i] == 0.0) {
      if (o3[i] < minVal || o3[i] > maxVal) return false;
    } else {
      double t1 = (minVal - o3[i]) * r3[i];
      double t2 = (maxVal - o3[i]) * r3[i];
      if (t1 > t2) { double t = t1; t1 = t2; t2 = t; }
      nearT = fmax(t1, nearT);
      farT = fmin(t2, farT);
      if (!(nearT <= farT)) return false;
    }
  }
  return !(farT < mint || nearT > maxt);
}

// 3D-kernel resample of the camera distance, shift_volume_photon.cpp:707-726
__device__ __forceinline__ bool resample3D(const HitGeom &g, double radius, double rnd, double mint, double edgeLen,
                                           double &tPrime, double &deltaT) {
#pragma clang fp contract(off)
  deltaT = sqrt(fmax(0.0, radius * radius - g.distSqr));
  const double tminKernel = g.disk - deltaT;
  tPrime = tminKernel + (deltaT * 2) * rnd;
  return !(tPrime < mint || tPrime > edgeLen);
}

__device__ __forceinline__ float phaseEval(float g, f3 wi, f3 wo) {
  if (g == 0.f) return INV_FOURPI_F;
  const float temp = 1.0f + g * g + 2.0f * g * dot(wi, wo);
  return INV_FOURPI_F * (1.f - g * g) / (temp * sqrtf(temp));
}

// HomogeneousMedium::eval over a distance (balance strategy)
__device__ __forceinline__ void mediumEval(const MediumDev &m, float dist, f3 &tr, float &pdfSuccess) {
  tr.x = __expf(-m.sigmaT[0] * dist);
  tr.y = __expf(-m.sigmaT[1] * dist);
  tr.z = __expf(-m.sigmaT[2] * dist);
  pdfSuccess = (m.sigmaT[0] * tr.x + m.sigmaT[1] * tr.y + m.sigmaT[2] * tr.z) * (1.f / 3.f) * m.msw;
  if (maxc(tr) < 1e-20f) tr = mk3(0.f);
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
template <int B>
__device__ __forceinline__ bool anyHit(const GatherArgs &a, const TileLds<B> &s, f3 o, f3 d, float mint, float maxt) {
  bool hit = false;
  const uint32_t nl = min(a.ntri, (uint32_t)MAXTRI_LDS);
  for (uint32_t i = 0; i < nl; ++i) {
    const float4 t0 = s.tri[i][0], t1 = s.tri[i][1], t2 = s.tri[i][2];
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
  c.parentScat = mk3(c5.x, c5.y, c5.z);
  c.parentWi = mk3(c6.x, c6.y, c6.z);
  return c;
}

// shiftPhotonDiffuse + diffuseReconnection.  Returns the MIS weight, writes the shifted flux.
template <int B>
__device__ __forceinline__ float shiftDiffuse(const GatherArgs &a, const TileLds<B> &s, const PhotonCold &ph,
                                              uint32_t bits, d3 offsetPos, const RayReg &sh, const RayReg &base,
                                              uint32_t edge, f3 trShift, float pdfBaseRay, float pdfShiftRay,
                                              f3 &shiftedFlux, bool &ok) {
  ok = false;
  shiftedFlux = mk3(0.f);
  const uint32_t ptype = GVPM_PF_PARENT_TYPE(bits);
  const d3 dProjD = offsetPos - tod(ph.parentPos);
  const double lProjD = sqrt(len2(dProjD));
  const f3 dProj = tof(dProjD / lProjD);
  const float lProj = (float)lProjD;
  const float eps = a.cfg.epsilon, seps = a.cfg.shadow_epsilon;
  const float vmax = a.cfg.visibility_as_written ? lProj * seps : lProj * (1.f - seps);
  if (anyHit<B>(a, s, ph.parentPos, dProj, eps, vmax)) return 1.f;
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

// coordinateSystemCoherent (float intermediates), src/libcore/util.cpp:592-599
__device__ __forceinline__ void coherentFrame(d3 n, d3 &b1, d3 &b2) {
  const float sign = copysignf(1.0f, (float)n.z);
  const float aa = (float)(-1.0 / ((double)sign + n.z));
  const float bb = (float)(n.x * n.y * (double)aa);
  b1 = mkd(1.0 + (double)sign * n.x * n.x * (double)aa, (double)sign * (double)bb, -(double)sign * n.x);
  b2 = mkd((double)bb, (double)sign + n.y * n.y * (double)aa, -n.y);
}

// One evaluation: VolumeGradientBREQuery::operator() after the filters.
template <int B>
__device__ __forceinline__ void evaluate(const GatherArgs &a, TileLds<B> &s, uint32_t pidx, uint32_t b,
                                         uint32_t &nNull, uint32_t &nDiff, uint32_t &nFail) {
  const float4 hot = a.hot[pidx];
  const uint32_t bits = __float_as_uint(hot.w);
  const f3 pos = mk3(hot.x, hot.y, hot.z);
  const PhotonCold ph = loadCold(a, pidx);
  const RayReg base = loadRay(s, 0, b);
  const uint32_t edge = s.edge[b];
  const uint32_t pix = s.pix[b];
  const int px = (int)(pix & 0xFFFFu), py = (int)(pix >> 16);
  const bool use3D = a.cfg.vol_technique == GVPM_VOL_BRE3D;
  const double radius = (double)a.radius;
  const double mint = (double)a.cfg.epsilon;

  // geometry of the hit, recomputed exactly as in the test phase
  const HitGeom g = hitGeom(pos, base.o, base.d);
  double tPrime = g.disk, deltaT = 0.0;
  double kernelVolD = 3.14159265358979323846 * radius * radius;
  float pdfCam = 1.f;
  if (use3D) {
    resample3D(g, radius, (double)s.rnd[b], mint, (double)base.len, tPrime, deltaT);
    kernelVolD = (4.0 / 3.0) * 3.14159265358979323846 * radius * radius * radius;
    pdfCam = (float)(1.0 / fmax(deltaT * 2.0, 0.0001));
  }
  const float rr = a.cfg.path_set ? 2.f : 1.f;
  const float scale = rr / ((float)kernelVolD * pdfCam);

  const f3 sigS = mk3(a.med.sigmaS[0], a.med.sigmaS[1], a.med.sigmaS[2]);
  // base contribution, shift_volume_photon.cpp:735-751; the base and the four shifted rays
  // all carry mint = Epsilon and maxt = t' (:769-770), hence one transmittance
  f3 trT;
  float dummy;
  mediumEval(a.med, (float)(tPrime - mint), trT, dummy);
  const f3 photonIn = sigS * ph.flux;
  const f3 baseContrib = trT * (photonIn * phaseEval(a.med.g, ph.wi, -base.d)) * base.eye;
  atomicAdd(&s.acc[0][b], baseContrib.x * scale);
  atomicAdd(&s.acc[1][b], baseContrib.y * scale);
  atomicAdd(&s.acc[2][b], baseContrib.z * scale);

  const d3 pD = tod(pos);
  const d3 basePt = tod(base.o) + tod(base.d) * tPrime;  // baseRay(t')
  const d3 rel = pD - basePt;
#pragma unroll 1
  for (int i = 0; i < 4; ++i) {
    const RayReg sh = loadRay(s, 1 + i, b);
    float w = 1.f;
    f3 sflux = mk3(0.f);
    if (sh.valid) {
      const d3 shO = tod(sh.o), shD = tod(sh.d);
      const d3 zP = shO + shD * tPrime;  // shiftRay(t')
      bool alreadyShift = false;
      if (a.cfg.use_shift_null) {
        const double ZPtoY = len2(zP - pD);
        if (ZPtoY < radius * radius && tPrime < (double)sh.len) {
          // shiftNull, shift_volume_photon.cpp:119-158 with the kernel pdfs of :782-801
          const double diskS = dot(pD - shO, shD);
          const double distSqrS = len2((shO + shD * diskS) - pD);
          const double deltaS = sqrt(fmax(0.0, radius * radius - distSqrS));
          const float pdfShiftPos = (float)(1.0 / fmax(2.0 * deltaS, 0.0001));
          sflux = trT * (photonIn * phaseEval(a.med.g, ph.wi, -sh.d)) * sh.eye;
          w = 0.5f;
          if (a.cfg.use_mis) {
            if (pdfShiftPos == 0.f || pdfCam == 0.f) w = 1.f;
            else w = 1.f / (1.f + sensorMIS(sh, base, edge) * pdfShiftPos / pdfCam);
          }
          alreadyShift = true;
          nNull++;
        }
      }
      if (!alreadyShift && (double)sh.len >= tPrime) {
        // getShiftPos, shift_volume_photon.cpp:858-896
        d3 offsetPos = zP + rel;
        if (!use3D) {
          d3 bn = tod(base.d), bs, bt, nn = shD, ns, nt;
          coherentFrame(bn, bs, bt);
          coherentFrame(nn, ns, nt);
          const double lx = dot(rel, bs), ly = dot(rel, bt), lz = dot(rel, bn);
          offsetPos = zP + (ns * lx + nt * ly + nn * lz);
        }
        if (a.cfg.use_shift_null) {
          const double offDistSqr = len2(basePt - offsetPos);
          if (offDistSqr < radius * radius) {
            d3 dShift = zP - basePt;
            dShift = dShift / sqrt(len2(dShift));
            const double cosD = dot(dShift, -(offsetPos - zP));
            offsetPos = offsetPos + dShift * (cosD * 2.0);
          }
        }
        float pdfShiftPos = 1.f;
        if (use3D) {
          const double diskO = dot(offsetPos - shO, shD);
          const double distSqrO = len2((shO + shD * diskO) - offsetPos);
          const double deltaO = sqrt(fmax(0.0, radius * radius - distSqrO));
          pdfShiftPos = (float)(1.0 / fmax(2.0 * deltaO, 0.0001));
        }
        if (a.cfg.debug_shift != GVPM_SHIFT_NULL) {
          // shiftPhoton dispatch, shift_volume_photon.cpp:49-117
          const uint32_t st = GVPM_PF_SHIFT_TYPE(bits);
          bool ok = false;
          if (st == 1u || st == 2u) {
            w = shiftDiffuse<B>(a, s, ph, bits, offsetPos, sh, base, edge, trT, pdfCam, pdfShiftPos, sflux, ok);
          }
          if (ok) nDiff++; else nFail++;
        }
      }
    }
    // no reverse shift at the right and top borders, shift_volume_photon.cpp:843-846
    if ((i == GVPM_RIGHT && px == a.cfg.width - 1) || (i == GVPM_TOP && py == a.cfg.height - 1)) w = 1.f;
    const float ws = w * scale;
    if (sflux.x != 0.f || sflux.y != 0.f || sflux.z != 0.f) {
      atomicAdd(&s.acc[3 + 3 * i + 0][b], sflux.x * ws);
      atomicAdd(&s.acc[3 + 3 * i + 1][b], sflux.y * ws);
      atomicAdd(&s.acc[3 + 3 * i + 2][b], sflux.z * ws);
    }
    atomicAdd(&s.acc[15 + 3 * i + 0][b], baseContrib.x * ws);
    atomicAdd(&s.acc[15 + 3 * i + 1][b], baseContrib.y * ws);
    atomicAdd(&s.acc[15 + 3 * i + 2][b], baseContrib.z * ws);
  }
}

// ------------------------------------------------------------------------------------------
// Tile traversal state shared by the planner and the gather kernel
// ------------------------------------------------------------------------------------------
struct TileWalk {
  // per lane
  RayReg base;
  bool beamValid;
  float oA, dA, oU, dU, oV, dV, t0, t1;
  // wave-uniform
  int A, cA0, cA1, K;
  float orgA, orgU, orgV, pad;
  int dimU, dimV;
  bool any;
};

template <int B>
__device__ __forceinline__ void loadTileRays(const GatherArgs &a, TileLds<B> &s, uint32_t setBase, uint32_t nb,
                                             int lane) {
  for (int idx = lane; idx < B * 20; idx += 64) {
    const int b = idx / 20, k = (idx % 20) / 4, q = idx % 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((uint32_t)b < nb) {
      const uint32_t set = a.setPerm[setBase + b];
      v = reinterpret_cast<const float4 *>(a.rays + (size_t)set * 5 + k)[q];
    }
    if (q == 0) { s.ray[k][0][b] = v.x; s.ray[k][1][b] = v.y; s.ray[k][2][b] = v.z; s.ray[k][6][b] = v.w; }
    else if (q == 1) { s.ray[k][3][b] = v.x; s.ray[k][4][b] = v.y; s.ray[k][5][b] = v.z; s.ray[k][7][b] = v.w; }
    else if (q == 2) { s.ray[k][8][b] = v.x; s.ray[k][9][b] = v.y; s.ray[k][10][b] = v.z; s.ray[k][11][b] = v.w; }
    else {
      s.ray[k][12][b] = v.x;
      const uint32_t info = __float_as_uint(v.y);
      if (k == 0) {
        s.rnd[b] = v.z;
        s.pix[b] = __float_as_uint(v.w);
        s.edge[b] = GVPM_RAY_EDGE(info);
      }
    }
  }
  __syncthreads();
  // fold the valid bit into the sign of len (written by the q == 0 lanes above)
  for (int idx = lane; idx < B * 5; idx += 64) {
    const int b = idx / 5, k = idx % 5;
    bool valid = false;
    if ((uint32_t)b < nb) {
      const uint32_t set = a.setPerm[setBase + b];
      valid = GVPM_RAY_VALID(a.rays[(size_t)set * 5 + k].info) != 0;
    }
    const float l = fabsf(s.ray[k][6][b]);
    s.ray[k][6][b] = valid ? l : -fmaxf(l, 1e-30f);
  }
  __syncthreads();
}

template <int B>
__device__ __forceinline__ void tileSetup(const GatherArgs &a, const TileLds<B> &s, uint32_t nb, int lane,
                                          TileWalk &w) {
  const int b = lane % B;
  w.base = loadRay(s, 0, b);
  w.beamValid = (uint32_t)b < nb && w.base.valid;
  const float r = a.radius, eps = a.cfg.epsilon;
  const float mint = eps, maxt = w.base.len - eps;
  int A;
  {
    const float ax = fabsf(w.base.d.x), ay = fabsf(w.base.d.y), az = fabsf(w.base.d.z);
    const int my = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
    const int n0 = __popcll(__ballot(w.beamValid && my == 0));
    const int n1 = __popcll(__ballot(w.beamValid && my == 1));
    const int n2 = __popcll(__ballot(w.beamValid && my == 2));
    A = (n0 >= n1 && n0 >= n2) ? 0 : (n1 >= n2 ? 1 : 2);
  }
  w.A = A;
  const int U = (A + 1) % 3, V = (A + 2) % 3;
  w.oA = comp(w.base.o, A); w.dA = comp(w.base.d, A);
  w.oU = comp(w.base.o, U); w.dU = comp(w.base.d, U);
  w.oV = comp(w.base.o, V); w.dV = comp(w.base.d, V);
  const f3 org = mk3(a.grid.org[0], a.grid.org[1], a.grid.org[2]);
  w.orgA = comp(org, A); w.orgU = comp(org, U); w.orgV = comp(org, V);
  const int dimA = A == 0 ? a.grid.dim[0] : (A == 1 ? a.grid.dim[1] : a.grid.dim[2]);
  w.dimU = U == 0 ? a.grid.dim[0] : (U == 1 ? a.grid.dim[1] : a.grid.dim[2]);
  w.dimV = V == 0 ? a.grid.dim[0] : (V == 1 ? a.grid.dim[1] : a.grid.dim[2]);
  w.pad = r * 1.01f + 1e-6f;
  // fattened parameter range: photons may hit with diskDistance up to ~maxt + sqrt(3) r
  w.t0 = mint - 2.f * r;
  w.t1 = maxt + 2.f * r;
  float aLo = INFINITY, aHi = -INFINITY;
  if (w.beamValid) {
    const float e0 = w.oA + w.dA * w.t0, e1 = w.oA + w.dA * w.t1;
    aLo = fminf(e0, e1) - w.pad;
    aHi = fmaxf(e0, e1) + w.pad;
  }
  aLo = wave_min(aLo);
  aHi = wave_max(aHi);
  w.any = aLo <= aHi && a.nph > 0;
  w.cA0 = 1;
  w.cA1 = 0;
  if (w.any) {
    w.cA0 = max(0, (int)floorf((aLo - w.orgA) * a.grid.invCell));
    w.cA1 = min(dimA - 1, (int)floorf((aHi - w.orgA) * a.grid.invCell));
  }
  // layers per step: thicker slabs when the contiguous (x) axis is the slab axis
  w.K = (A == 0) ? 8 : 4;
}

struct CellBox {
  int bx0, bx1, by0, by1, bz0, bz1;
};

// cell box of the slab layers [cA, cAe]; false when no beam of the tile reaches the slab
__device__ __forceinline__ bool slabBox(const GatherArgs &a, const TileWalk &w, int cA, int cAe, CellBox &bx) {
  const float lo = w.orgA + cA * a.grid.cell - w.pad, hi = w.orgA + (cAe + 1) * a.grid.cell + w.pad;
  float uLo = INFINITY, uHi = -INFINITY, vLo = INFINITY, vHi = -INFINITY;
  if (w.beamValid) {
    float ta = w.t0, tb = w.t1;
    bool act = true;
    if (fabsf(w.dA) > 1e-12f) {
      const float inv = 1.f / w.dA;
      const float s0 = (lo - w.oA) * inv, s1 = (hi - w.oA) * inv;
      ta = fmaxf(ta, fminf(s0, s1));
      tb = fminf(tb, fmaxf(s0, s1));
      act = ta <= tb;
    } else {
      act = w.oA >= lo && w.oA <= hi;
    }
    if (act) {
      const float u0 = w.oU + w.dU * ta, u1 = w.oU + w.dU * tb, v0 = w.oV + w.dV * ta, v1 = w.oV + w.dV * tb;
      uLo = fminf(u0, u1) - w.pad; uHi = fmaxf(u0, u1) + w.pad;
      vLo = fminf(v0, v1) - w.pad; vHi = fmaxf(v0, v1) + w.pad;
    }
  }
  uLo = wave_min(uLo); uHi = wave_max(uHi);
  vLo = wave_min(vLo); vHi = wave_max(vHi);
  if (!(uLo <= uHi)) return false;
  const int cU0 = max(0, (int)floorf((uLo - w.orgU) * a.grid.invCell));
  const int cU1 = min(w.dimU - 1, (int)floorf((uHi - w.orgU) * a.grid.invCell));
  const int cV0 = max(0, (int)floorf((vLo - w.orgV) * a.grid.invCell));
  const int cV1 = min(w.dimV - 1, (int)floorf((vHi - w.orgV) * a.grid.invCell));
  if (cU0 > cU1 || cV0 > cV1) return false;
  // (A,U,V) -> (x,y,z): A=0: x=A y=U z=V; A=1: x=V y=A z=U; A=2: x=U y=V z=A
  const int A = w.A;
  bx.bx0 = A == 0 ? cA : (A == 1 ? cV0 : cU0); bx.bx1 = A == 0 ? cAe : (A == 1 ? cV1 : cU1);
  bx.by0 = A == 0 ? cU0 : (A == 1 ? cA : cV0); bx.by1 = A == 0 ? cU1 : (A == 1 ? cAe : cV1);
  bx.bz0 = A == 0 ? cV0 : (A == 1 ? cU0 : cA); bx.bz1 = A == 0 ? cV1 : (A == 1 ? cU1 : cAe);
  return true;
}

// x-contiguous photon range of this lane for range index ri of the box
__device__ __forceinline__ void boxRange(const GatherArgs &a, const CellBox &bx, int ri, int nranges, uint32_t &start,
                                         uint32_t &count) {
  start = 0;
  count = 0;
  if (ri < nranges) {
    const int nyr = bx.by1 - bx.by0 + 1;
    const int y = bx.by0 + ri % nyr, z = bx.bz0 + ri / nyr;
    const uint32_t row = ((uint32_t)z * a.grid.dim[1] + y) * a.grid.dim[0];
    start = a.cellStart[row + bx.bx0];
    count = a.cellStart[row + bx.bx1 + 1] - start;
  }
}

// number of photons staged for a slab step (sum over the box's ranges), wave-uniform
__device__ __forceinline__ uint32_t boxCount(const GatherArgs &a, const CellBox &bx, int lane) {
  const int nranges = (bx.by1 - bx.by0 + 1) * (bx.bz1 - bx.bz0 + 1);
  uint32_t c = 0;
  for (int rbase = 0; rbase < nranges; rbase += 64) {
    uint32_t st, cnt;
    boxRange(a, bx, rbase + lane, nranges, st, cnt);
    c += cnt;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  return c;
}

// ------------------------------------------------------------------------------------------
// plan: cut every tile chunk into work items of ~equal candidate count
// item = {setBase, nb, firstLayer, lastLayer}
// ------------------------------------------------------------------------------------------
template <int B>
__global__ __launch_bounds__(64) void plan_kernel(GatherArgs a, uint32_t ntiles, uint32_t target, uint4 *items,
                                                  uint32_t *itemCount) {
  __shared__ TileLds<B> s;
  const int lane = threadIdx.x;
  const uint32_t tile = blockIdx.x;
  if (tile >= ntiles) return;
  const uint32_t tileBeg = a.tileStart[tile], tileEnd = a.tileStart[tile + 1];
  for (uint32_t setBase = tileBeg; setBase < tileEnd; setBase += B) {
    const uint32_t nb = min((uint32_t)B, tileEnd - setBase);
    loadTileRays<B>(a, s, setBase, nb, lane);
    TileWalk w;
    tileSetup<B>(a, s, nb, lane, w);
    if (!w.any) continue;
    // pass A: total candidates of the chunk
    uint32_t total = 0;
    for (int cA = w.cA0; cA <= w.cA1; cA += w.K) {
      CellBox bx;
      if (slabBox(a, w, cA, min(cA + w.K - 1, w.cA1), bx)) total += boxCount(a, bx, lane);
    }
    if (total == 0) continue;
    const uint32_t nItems = min((uint32_t)PLAN_MAX_ITEMS, (total + target - 1) / target);
    const uint32_t per = (total + nItems - 1) / nItems;
    uint32_t slot = 0;
    if (lane == 0) slot = atomicAdd(itemCount, nItems);
    slot = __shfl(slot, 0, 64);
    // pass B: emit items at the crossings of k * per
    uint32_t run = 0, emitted = 0;
    int first = w.cA0;
    for (int cA = w.cA0; cA <= w.cA1; cA += w.K) {
      const int cAe = min(cA + w.K - 1, w.cA1);
      CellBox bx;
      if (slabBox(a, w, cA, cAe, bx)) run += boxCount(a, bx, lane);
      const bool last = cAe == w.cA1;
      if ((run >= per && emitted + 1 < nItems) || last) {
        if (lane == 0) items[slot + emitted] = make_uint4(setBase, nb, (uint32_t)first, (uint32_t)cAe);
        emitted++;
        run = 0;
        first = cAe + 1;
      }
    }
    // unused reserved slots (possible when the crossings come late): mark empty
    for (uint32_t e = emitted + lane; e < nItems; e += 64) items[slot + e] = make_uint4(setBase, 0u, 1u, 0u);
  }
}

// ------------------------------------------------------------------------------------------
// gather: persistent waves pulling work items
// ------------------------------------------------------------------------------------------
template <int B>
__global__ __launch_bounds__(64, 2) void gather_bre_kernel(GatherArgs a, const uint4 *__restrict__ items,
                                                           const uint32_t *__restrict__ itemCount,
                                                           uint32_t *queueHead) {
  constexpr int LPB = 64 / B;
  __shared__ TileLds<B> s;
  const int lane = threadIdx.x;
  const uint32_t nItems = *itemCount;
  const int b = lane % B, sub = lane / B;
  const float r = a.radius;
  const float eps = a.cfg.epsilon;
  const bool use3D = a.cfg.vol_technique == GVPM_VOL_BRE3D;

  // occluders -> LDS, with the unit normal for the plane early-out
  for (uint32_t i = lane; i < min(a.ntri, (uint32_t)MAXTRI_LDS); i += 64) {
    const f3 v0 = mk3(a.triV0[3 * i], a.triV0[3 * i + 1], a.triV0[3 * i + 2]);
    const f3 e1 = mk3(a.triE1[3 * i], a.triE1[3 * i + 1], a.triE1[3 * i + 2]);
    const f3 e2 = mk3(a.triE2[3 * i], a.triE2[3 * i + 1], a.triE2[3 * i + 2]);
    f3 n = cross(e1, e2);
    const float l = sqrtf(dot(n, n));
    n = l > 0.f ? n * (1.f / l) : mk3(0.f);
    s.tri[i][0] = make_float4(v0.x, v0.y, v0.z, n.x);
    s.tri[i][1] = make_float4(e1.x, e1.y, e1.z, n.y);
    s.tri[i][2] = make_float4(e2.x, e2.y, e2.z, n.z);
  }

  uint32_t nEval = 0, nNull = 0, nDiff = 0, nFail = 0;
  unsigned long long nCand = 0;

  for (;;) {
    uint32_t it = 0;
    if (lane == 0) it = atomicAdd(queueHead, 1u);
    it = __shfl(it, 0, 64);
    if (it >= nItems) break;
    const uint4 item = items[it];
    const uint32_t setBase = item.x, nb = item.y;
    if (nb == 0) continue;
    __syncthreads();
    loadTileRays<B>(a, s, setBase, nb, lane);
    for (int idx = lane; idx < 27 * B; idx += 64) (&s.acc[0][0])[idx] = 0.f;
    __syncthreads();
    TileWalk w;
    tileSetup<B>(a, s, nb, lane, w);
    const RayReg base = w.base;
    const bool beamValid = w.beamValid;
    const float mint = eps, maxt = base.len - eps;
    const double mintD = (double)eps, maxtD = (double)base.len - (double)eps;
    const d3 rcpD = mkd(1.0 / (double)base.d.x, 1.0 / (double)base.d.y, 1.0 / (double)base.d.z);
    const float rnd = s.rnd[b];
    const uint32_t edge = s.edge[b];
    const uint32_t pixv = s.pix[b];
    const uint32_t pixParity = ((pixv & 0xFFFFu) + (pixv >> 16)) & 1u;
    uint32_t qHead = 0, qCount = 0;  // wave-uniform ring state
    uint32_t staged = 0;

    const int cBeg = max((int)item.z, w.cA0), cEnd = min((int)item.w, w.cA1);
    for (int cA = cBeg; cA <= cEnd; cA += w.K) {
      const int cAe = min(cA + w.K - 1, cEnd);
      CellBox bx;
      if (!slabBox(a, w, cA, cAe, bx)) continue;
      const int nranges = (bx.by1 - bx.by0 + 1) * (bx.bz1 - bx.bz0 + 1);
      for (int rbase = 0; rbase < nranges; rbase += 64) {
        uint32_t start, count;
        boxRange(a, bx, rbase + lane, nranges, start, count);
        const uint32_t incl = wave_scan_incl(count, lane);
        const uint32_t excl = incl - count;
        const uint32_t total = __shfl(incl, 63, 64);
        for (uint32_t win = 0; win < total; win += STAGE) {
          // stage [win, win + STAGE) of the concatenated ranges
          {
            const uint32_t lo_i = max(excl, win), hi_i = min(excl + count, win + STAGE);
            for (uint32_t i = lo_i; i < hi_i; ++i) {
              const uint32_t gi = start + (i - excl);
              s.stage[i - win] = a.hot[gi];
              s.stageIdx[i - win] = gi;
            }
          }
          __syncthreads();
          const uint32_t nst = min((uint32_t)STAGE, total - win);
          staged += nst;
          const uint32_t iters = (nst + LPB - 1) / LPB;
          for (uint32_t jj = 0; jj < iters; ++jj) {
            const uint32_t j = jj * LPB + sub;
            bool hit = false;
            uint32_t gi = 0;
            if (beamValid && j < nst) {
              const float4 hp = s.stage[j];
              const f3 p = mk3(hp.x, hp.y, hp.z);
              const f3 wv = p - base.o;
              const float disk = dot(wv, base.d);
              const f3 v = wv - base.d * disk;
              const float d2 = dot(v, v);
              // conservative fp32 pre-test
              if (d2 < r * r * 1.002f + 1e-12f && disk > mint - 1e-3f && disk < maxt + 2.f * r) {
                const HitGeom g = hitGeom(p, base.o, base.d);
                if (g.disk > mintD && g.distSqr < (double)r * (double)r &&
                    ownBoxHit(p, base.o, base.d, rcpD, mintD, maxtD, (double)r)) {
                  const uint32_t bits = __float_as_uint(hp.w);
                  // filters, shift_volume_photon.cpp:670-697
                  const int depth = (int)GVPM_PF_DEPTH(bits) + (int)edge;
                  bool keep = true;
                  if (a.cfg.max_depth > 0 && depth > a.cfg.max_depth) keep = false;
                  if (a.cfg.min_depth != 0 && depth < a.cfg.min_depth) keep = false;
                  if (!((bits >> 6) & 1u)) keep = false;  // computeVolumeContribution + debugShift (grid_build)
                  if (a.cfg.path_set && ((bits >> GVPM_HOT_PARITY_BIT) & 1u) != pixParity) keep = false;
                  if (keep && use3D) {
                    double tp, dt;
                    keep = resample3D(g, (double)r, (double)rnd, mintD, (double)base.len, tp, dt);
                  }
                  hit = keep;
                  gi = s.stageIdx[j];
                }
              }
            }
            const unsigned long long m = __ballot(hit);
            if (m) {
              if (hit) {
                const uint32_t off = __popcll(m & ((1ull << lane) - 1ull));
                s.queue[(qHead + qCount + off) % QCAP] = make_uint2(gi, (uint32_t)b);
              }
              qCount += __popcll(m);
              if (qCount >= 64u) {
                __syncthreads();
                const uint2 e = s.queue[(qHead + lane) % QCAP];
                evaluate<B>(a, s, e.x, e.y, nNull, nDiff, nFail);
                nEval++;
                qHead = (qHead + 64u) % QCAP;
                qCount -= 64u;
                __syncthreads();
              }
            }
          }
          __syncthreads();
        }
      }
    }
    // ---- flush the partial batch ----
    __syncthreads();
    if ((uint32_t)lane < qCount) {
      const uint2 e = s.queue[(qHead + lane) % QCAP];
      evaluate<B>(a, s, e.x, e.y, nNull, nDiff, nFail);
      nEval++;
    }
    __syncthreads();
    // ---- write out: 27 partial sums per beam set into the iteration buffer ----
    for (int idx = lane; idx < 27 * B; idx += 64) {
      const int k = idx / B, bb = idx % B;
      if ((uint32_t)bb < nb) {
        const float v = s.acc[k][bb];
        if (v != 0.f) {
          const uint32_t pv = s.pix[bb];
          const size_t p = (size_t)(pv >> 16) * a.cfg.width + (pv & 0xFFFFu);
          atomicAdd(&a.iter[p * 27 + k], v);
        }
      }
    }
    nCand += (unsigned long long)staged * nb;
  }
  // ---- statistics ----
  {
    unsigned long long ev = nEval, nu = nNull, di = nDiff, fa = nFail;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      ev += __shfl_xor(ev, o, 64);
      nu += __shfl_xor(nu, o, 64);
      di += __shfl_xor(di, o, 64);
      fa += __shfl_xor(fa, o, 64);
    }
    if (lane == 0 && (ev | nCand)) {
      atomicAdd(&a.stats[0], ev);
      atomicAdd(&a.stats[1], nCand);
      atomicAdd(&a.stats[2], nu);
      atomicAdd(&a.stats[3], di);
      atomicAdd(&a.stats[4], fa);
    }
  }
}

template <int B>
static void launchT(const GatherArgs &a, uint32_t ntiles, uint32_t target, uint4 *items, uint32_t *itemCount,
                    uint32_t *queueHead, uint32_t nwaves, hipStream_t stream) {
  hipLaunchKernelGGL(plan_kernel<B>, dim3(ntiles), dim3(64), 0, stream, a, ntiles, target, items, itemCount);
  hipLaunchKernelGGL(gather_bre_kernel<B>, dim3(nwaves), dim3(64), 0, stream, a, (const uint4 *)items,
                     (const uint32_t *)itemCount, queueHead);
}

// itemCount / queueHead must be zero on entry (memset on the same stream)
void launch_plan_bre(const GatherArgs &a, int beamsPerWave, uint32_t ntiles, uint32_t target, uint4 *items,
                     uint32_t *itemCount, hipStream_t stream) {
  if (a.nsets == 0 || ntiles == 0) return;
  switch (beamsPerWave) {
    case 64: hipLaunchKernelGGL(plan_kernel<64>, dim3(ntiles), dim3(64), 0, stream, a, ntiles, target, items, itemCount); break;
    case 16: hipLaunchKernelGGL(plan_kernel<16>, dim3(ntiles), dim3(64), 0, stream, a, ntiles, target, items, itemCount); break;
    default: hipLaunchKernelGGL(plan_kernel<32>, dim3(ntiles), dim3(64), 0, stream, a, ntiles, target, items, itemCount); break;
  }
}

void launch_gather_bre(const GatherArgs &a, int beamsPerWave, const uint4 *items, const uint32_t *itemCount,
                       uint32_t *queueHead, uint32_t nwaves, hipStream_t stream) {
  if (a.nsets == 0) return;
  switch (beamsPerWave) {
    case 64: hipLaunchKernelGGL(gather_bre_kernel<64>, dim3(nwaves), dim3(64), 0, stream, a, items, itemCount, queueHead); break;
    case 16: hipLaunchKernelGGL(gather_bre_kernel<16>, dim3(nwaves), dim3(64), 0, stream, a, items, itemCount, queueHead); break;
    default: hipLaunchKernelGGL(gather_bre_kernel<32>, dim3(nwaves), dim3(64), 0, stream, a, items, itemCount, queueHead); break;
  }
}

uint32_t plan_items_capacity(uint32_t nsets, uint32_t ntiles, int beamsPerWave) {
  return (ntiles + nsets / (uint32_t)beamsPerWave + 1u) * (uint32_t)PLAN_MAX_ITEMS;
}

}  // namespace gvpm
