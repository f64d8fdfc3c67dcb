// The EXACT PASS (round 5): the shifts the fp32 evaluation kernels could not decide as the reference does.
//
// The gather kernels take every decision of a shift -- null shift or reconnection (|z' - y|^2 < r^2, t' against the shifted
// edge), the mirror step of getShiftPos, the shadow segment's triangle tests, the sign / cosine tests of the reconnection --
// in fp32 WITH a rigorous error margin.  Outside the margins the fp32 decision is the one exact arithmetic on the same
// fp32 inputs takes, i.e. the fp64 oracle's and a double-precision reference's.  Inside one, the kernel adds nothing and
// counts nothing for that shift and appends a self-contained entry to a list (ExEntry, device_types.h: ~1e-5 of
// the shifts; for a parent that position rounding left BEHIND the wall it sits on, every reconnection within a few
// degrees of grazing).  The kernels below evaluate the listed shifts by the reference's own statement in uncontracted
// fp64, operation for operation as the oracle restates it, and add the two terms the fast kernel left out:
//     weighted[i] += rr w base / (kernelVol pdfCameraPos),   shifted[i] += rr w shiftedFlux / (kernelVol pdfCameraPos).
// With it the shift counters of a gather equal the fp64 oracle's EXACTLY on every scene (tests/test_rotated_gpu.py and the
// parity files), not to a tolerance.
//
//   VolumeGradientBREQuery::operator()      gvpm/shift/shift_volume_photon.cpp:658-856
//   VolumeGradientPositionQuery::operator() shift_volume_photon.cpp:489-655
//   shiftNull / shiftPhoton / shiftPhotonDiffuse   :49-158, 382-486;  getShiftPos :858-896
//   diffuseReconnection                     gvpm/shift/operation/shift_diffuse.cpp:11-134
//   Triangle::rayIntersect                  include/mitsuba/core/triangle.h:109-145
#include <hip/hip_runtime.h>

#include "context.h"
#include "device_types.h"
#include "dmath.h"
#include "shift_device.h"
#include "vec.h"

namespace gvpm {

namespace {

__device__ __forceinline__ double len2d(d3 a) { return dot(a, a); }

struct RayIn {
  d3 o, d, eye;
  double len, pdf, jac, gop;
  bool valid;
  uint32_t edge;
};
__device__ __forceinline__ RayIn loadRayIn(const gvpm_camera_ray &r) {
  RayIn q;
  q.o = mkd(r.o[0], r.o[1], r.o[2]);
  q.d = mkd(r.d[0], r.d[1], r.d[2]);
  q.eye = mkd(r.eye[0], r.eye[1], r.eye[2]);
  q.len = r.len; q.pdf = r.pdf; q.jac = r.jacobian; q.gop = r.gop;
  q.valid = GVPM_RAY_VALID(r.info) != 0;
  q.edge = GVPM_RAY_EDGE(r.info);
  return q;
}

// GatherPoint::sensorMIS, gvpm_struct.h:608-631, literally (the cancelling G / distance factors included)
__device__ __forceinline__ double sensorMISD(const RayIn &s, const RayIn &b, uint32_t edge, double sDist, double bDist) {
#pragma clang fp contract(off)
  double jacobian = s.jac;
  double ratio = s.pdf / b.pdf;
  if (edge != 1u) {
    jacobian *= s.gop / b.gop;
    ratio *= b.gop / s.gop;
    jacobian *= (sDist / bDist) * (sDist / bDist);
    ratio *= (bDist / sDist) * (bDist / sDist);
  }
  return ratio * jacobian;
}

struct ShiftOut {
  d3 flux;       // result.shiftedFlux
  double weight; // result.weight
  int kind;      // 0 nothing counted, 1 null shift, 2 reconnection, 3 failed shift
};

// coordinateSystemCoherent, src/libcore/util.cpp:592-599 (float intermediates, as the reference and the oracle)
__device__ __forceinline__ void coherentFrameD(d3 n, d3 &b1, d3 &b2) {
#pragma clang fp contract(off)
  const float sign = copysignf(1.0f, (float)n.z);
  const float aa = (float)(-1.0f / (sign + n.z));
  const float bb = (float)(n.x * n.y * aa);
  b1 = mkd(1.0f + sign * n.x * n.x * aa, sign * bb, -sign * n.x);
  b2 = mkd(bb, sign + n.y * n.y * aa, -n.y);
}

// diffuseReconnection + shiftPhotonDiffuse (shift_diffuse.cpp:11-134, shift_volume_photon.cpp:382-486) towards offsetPos.
// trShift: transmittance of the shifted camera segment; returns the kind (2 / 3) and fills flux / weight.
__device__ void reconnectExact(const GatherArgs &a, const PhotonCold &ph, d3 offsetPos, const RayIn &sh, const RayIn &base, double shMaxt,
                               double baseMaxt, double trShift, double pdfBaseRay, double pdfShiftRay, ShiftOut &out) {
#pragma clang fp contract(off)
  const double INV_PI = 0.31830988618379067154;
  out.flux = mkd(0, 0, 0);
  out.weight = 1.0;
  out.kind = 3;
  const uint32_t st = GVPM_PF_SHIFT_TYPE(ph.bits);
  // shiftPhoton dispatch (:49-117): invalid -> failed; manifold without the host's walk -> failed
  if (!(st == 1u || st == 2u)) return;
  const d3 parent = tod(ph.parentPos);
  d3 dProj = offsetPos - parent;
  const double lProj = sqrt(len2d(dProj));
  dProj = dProj / lProj;
  const double eps = (double)a.cfg.epsilon, seps = (double)a.cfg.shadow_epsilon;
  const double maxt = a.cfg.visibility_as_written ? lProj * seps : lProj * (1.0 - seps);
  if (anyHitExact(a, ph.parentPos, dProj, eps, maxt)) return;
  const uint32_t ptype = GVPM_PF_PARENT_TYPE(ph.bits);
  const d3 pn = tod(ph.parentN), pwi = tod(ph.parentWi), scat = tod(ph.parentScat);
  if (ptype != GVPM_PARENT_MEDIUM) {
    const double signDot = dot(pn, dProj) / dot(pn, -tod(ph.wi));
    if (signDot < 0.0) return;
  }
  d3 thr = mkd(1, 1, 1);
  double pdfValue = 0.0;
  if (ptype == GVPM_PARENT_SURFACE) {
    const double cosWo = dot(pn, dProj), cosWi = dot(pn, pwi);
    if (cosWi <= 0 || cosWo <= 0) return;  // (eval = pdf = 0 and the normal-consistency reject, :43-47)
    thr = scat * (INV_PI * cosWo);
    pdfValue = INV_PI * cosWo;
  } else if (ptype == GVPM_PARENT_SURFACE_BSDF) {
    const double cosWo = dot(pn, dProj), cosWi = dot(pn, pwi);
    f3 f;
    float pdfF;
    if (cosWi <= 0 || cosWo <= 0) return;
    // Phong: in fp64 (a lobe of exponent ~1000 underflows fp32 where the reference's double is still positive, and pdf == 0
    // is a decision); the rough conductor's values in fp32 -- the table's closed forms, shift_device.h -- the DECISIONS
    // around them are taken here
    if (!phongEvalD(a, ph.parentG, scat, pn, pwi, dProj, cosWi, cosWo, thr, pdfValue)) {
      if (!glossyParentEval(a, ph.parentG, ph.parentScat, ph.parentN, ph.parentWi, tof(dProj), (float)cosWi, (float)cosWo, f, pdfF)) return;
      thr = tod(f);
      pdfValue = pdfF;
    }
  } else if (ptype == GVPM_PARENT_MEDIUM) {
    const double p = phaseD((double)ph.parentG, pwi, dProj);
    thr = scat * p;
    pdfValue = p;
  } else {
    double dp = dot(dProj, pn);
    if (dp < 0) dp = 0.0;
    thr = mkd(INV_PI * dp, INV_PI * dp, INV_PI * dp);
    pdfValue = INV_PI * dp;
  }
  const double GOp = 1.0 / (lProj * lProj);
  double sPdf = pdfValue * GOp;
  thr = thr * GOp;
  if (ph.parentPdf == 0.f) return;
  thr = thr / (double)ph.parentPdf;
  thr = thr * (double)ph.parentRR;
  if (GVPM_PF_EDGE_IN_MEDIUM(ph.bits)) {
    const MRecD m = mediumEvalD(a.med, lProj);
    sPdf *= m.pdfSuccess;
    thr = thr * (m.tr / (double)ph.edgePdf);
  }
  if (sPdf == 0.0) return;  // result.weight = 1
  const d3 pw = tod(ph.prefixW);
  const d3 photonWeight = mkd(pw.x * thr.x, pw.y * thr.y, pw.z * thr.z);
  const double phs = phaseD((double)a.med.g, -dProj, -sh.d);
  const d3 sigS = mkd(a.med.sigmaS[0], a.med.sigmaS[1], a.med.sigmaS[2]);
  const d3 contrib = mkd(sigS.x * photonWeight.x * phs, sigS.y * photonWeight.y * phs, sigS.z * photonWeight.z * phs);
  out.flux = mkd(trShift * contrib.x * sh.eye.x, trShift * contrib.y * sh.eye.y, trShift * contrib.z * sh.eye.z);
  out.weight = 0.5;
  if (a.cfg.use_mis) {
    double basePdf = pdfBaseRay;
    basePdf *= (double)ph.parentPdf;
    basePdf *= (double)ph.edgePdf;
    const double offsetPdf = sPdf * pdfShiftRay;
    if (offsetPdf == 0.0 || basePdf == 0.0) {
      out.weight = 1.0;  // (the flux it computed stays, shift_volume_photon.cpp:463-470)
      return;
    }
    const double sensorPart = sensorMISD(sh, base, base.edge, shMaxt, baseMaxt);
    if (a.cfg.power_heuristic) {
      const double v = sensorPart * (offsetPdf / basePdf);
      out.weight = 1.0 / (1.0 + v * v);
    } else {
      out.weight = 1.0 / (1.0 + sensorPart * (offsetPdf / basePdf));
    }
  }
  out.kind = 2;
}

__device__ __forceinline__ void addShift(const GatherArgs &a, uint32_t pix, int i, d3 flux, d3 baseContrib, double w, double scale,
                                         float outScale, float *dstBase) {
  const int px = (int)(pix & 0xFFFFu), py = (int)(pix >> 16);
  if ((i == GVPM_RIGHT && px == a.cfg.width - 1) || (i == GVPM_TOP && py == a.cfg.height - 1)) w = 1.0;
  float *dst = dstBase + ((size_t)py * a.cfg.width + px) * 27;
  const double ws = w * scale * (double)outScale;
  if (flux.x != 0.0) atomicAdd(&dst[3 + 3 * i + 0], (float)(flux.x * ws));
  if (flux.y != 0.0) atomicAdd(&dst[3 + 3 * i + 1], (float)(flux.y * ws));
  if (flux.z != 0.0) atomicAdd(&dst[3 + 3 * i + 2], (float)(flux.z * ws));
  atomicAdd(&dst[15 + 3 * i + 0], (float)(baseContrib.x * ws));
  atomicAdd(&dst[15 + 3 * i + 1], (float)(baseContrib.y * ws));
  atomicAdd(&dst[15 + 3 * i + 2], (float)(baseContrib.z * ws));
}

// the pass's bookkeeping: counters of the processed shifts, then the LAST workgroup resets the list for the next gather
__device__ __forceinline__ void finishPass(const GatherArgs &a, uint32_t nEval, uint32_t nNull, uint32_t nDiff, uint32_t nFail, uint32_t n,
                                           uint32_t total, unsigned long long *totals) {
  unsigned long long *row = a.stats + 8 * (size_t)(blockIdx.x % GVPM_STAT_ROWS);
  if (nEval) atomicAdd(&row[0], (unsigned long long)nEval);
  if (nNull) atomicAdd(&row[2], (unsigned long long)nNull);
  if (nDiff) atomicAdd(&row[3], (unsigned long long)nDiff);
  if (nFail) atomicAdd(&row[4], (unsigned long long)nFail);
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const uint32_t ticket = atomicAdd(a.exPayCount + 2, 1u);
    if (ticket == gridDim.x - 1u) {
      const unsigned long long lost = (unsigned long long)(total - n) + (unsigned long long)a.exPayCount[1];
      if (totals) {
        atomicAdd(&totals[0], (unsigned long long)n);
        atomicMax(&totals[2], (unsigned long long)total);
        if (lost) atomicAdd(&totals[1], lost);
      }
      if (lost) atomicAdd(&a.stats[7], lost);  // dropped (gvpm_stats::dropped_pairs): gvpm_get_stats fails
      a.exPayCount[0] = 0u;
      a.exPayCount[1] = 0u;
      a.exPayCount[2] = 0u;
      __threadfence();
    }
  }
}

}  // namespace

// ---- G-BRE -----------------------------------------------------------------------------------------------------------------
// the reference's hit decision for one (photon, beam) candidate (gvpm_accel.h:279-301, aabb.h:310-340, the 3D resample of
// shift_volume_photon.cpp:707-726): what gather_bre.hip's exactHit() states, here for the pairs the band left undecided
__device__ bool exactHitBRE(f3 pf, const gvpm_camera_ray &rb, float rf, float epsf, bool use3D) {
#pragma clang fp contract(off)
  const double px = pf.x, py = pf.y, pz = pf.z, ox = rb.o[0], oy = rb.o[1], oz = rb.o[2], dx = rb.d[0], dy = rb.d[1], dz = rb.d[2];
  const double mint = (double)epsf, maxt = (double)rb.len - (double)epsf, radius = (double)rf;
  const double cx = px - ox, cy = py - oy, cz = pz - oz;
  const double disk = cx * dx + cy * dy + cz * dz;
  const double qx = ox + dx * disk, qy = oy + dy * disk, qz = oz + dz * disk;
  const double vx = qx - px, vy = qy - py, vz = qz - pz;
  const double distSqr = vx * vx + vy * vy + vz * vz;
  if (!(disk > mint && distSqr < radius * radius)) return false;
  // the photon's own sphere box against the ray segment: what every ancestor box of the reference's BVH implies
  double nearT = -INFINITY, farT = INFINITY;
  const double o3[3] = {ox, oy, oz}, d3v[3] = {dx, dy, dz}, c3[3] = {px, py, pz};
  for (int k = 0; k < 3; ++k) {
    const double minVal = c3[k] - radius, maxVal = c3[k] + radius;
    if (d3v[k] == 0.0) {
      if (o3[k] < minVal || o3[k] > maxVal) return false;
    } else {
      const double rcp = 1.0 / d3v[k];
      double t1 = (minVal - o3[k]) * rcp, t2 = (maxVal - o3[k]) * rcp;
      if (t1 > t2) { const double t = t1; t1 = t2; t2 = t; }
      nearT = fmax(t1, nearT);
      farT = fmin(t2, farT);
      if (!(nearT <= farT)) return false;
    }
  }
  if (farT < mint || nearT > maxt) return false;
  if (!use3D) return true;
  const double deltaT = sqrt(fmax(0.0, radius * radius - distSqr));
  const double tPrime = (disk - deltaT) + (deltaT * 2) * (double)rb.rand;
  return !(tPrime < mint || tPrime > (double)rb.len);
}

__device__ __forceinline__ PhotonCold coldOf(const ExEntry &e) {
  PhotonCold c;
  const float4 c0 = e.rec[0], c1 = e.rec[1], c2 = e.rec[2], c3 = e.rec[3], c4 = e.rec[4], c5 = e.rec[5], c6 = e.rec[6], c7 = e.rec[7];
  c.pos = mk3(c0.x, c0.y, c0.z); c.bits = __float_as_uint(c0.w);
  c.wi = mk3(c1.x, c1.y, c1.z); c.parentPdf = c1.w;
  c.flux = mk3(c2.x, c2.y, c2.z); c.edgePdf = c2.w;
  c.parentPos = mk3(c3.x, c3.y, c3.z); c.parentRR = c3.w;
  c.parentN = mk3(c4.x, c4.y, c4.z); c.parentG = c4.w;
  c.prefixW = mk3(c5.x, c5.y, c5.z);
  c.parentScat = mk3(c6.x, c6.y, c6.z);
  c.parentWi = mk3(c7.x, c7.y, c7.z);
  c.nl0 = c.nl1 = c.nl2 = 0u;  // (the exact pass walks the whole scene)
  return c;
}

// one entry: shift i of a pair, or (pair: iOnly < 0) the hit decision, the base term and all four shifts
__device__ void exactBRE(const GatherArgs &a, const ExEntry &e, int iOnly, uint32_t &nEval, uint32_t &nNull, uint32_t &nDiff,
                         uint32_t &nFail) {
#pragma clang fp contract(off)
  const double M_PI_D = 3.14159265358979323846;
  const gvpm_camera_ray rb = e.rays[0];
  const PhotonCold ph = coldOf(e);
  const bool use3D = a.cfg.vol_technique == GVPM_VOL_BRE3D;
  if (iOnly < 0 && !exactHitBRE(ph.pos, rb, e.radius, a.cfg.epsilon, use3D)) return;
  const RayIn base = loadRayIn(rb);
  const double radius = (double)e.radius, r2 = radius * radius, eps = (double)a.cfg.epsilon;
  for (int i = iOnly < 0 ? 0 : iOnly; i < (iOnly < 0 ? 4 : iOnly + 1); ++i) {
    const RayIn sh = loadRayIn(e.rays[1 + i]);
    const double rr = a.cfg.path_set ? 2.0 : 1.0;
    const d3 pos = tod(ph.pos);
    // the query's hit geometry (gvpm_accel.h:296-301) and the functor's base term, shift_volume_photon.cpp:701-751
    const d3 otc = pos - base.o;
    const double diskDistance = dot(otc, base.d);
    double baseMaxt = diskDistance;
    double kernelVol = M_PI_D * (radius * radius);
    double pdfCameraPos = 1.0;
    if (use3D) {
      kernelVol = (4.0 / 3.0) * M_PI_D * (radius * radius * radius);
      const double distSqr = len2d((base.o + base.d * baseMaxt) - pos);
      const double deltaT = sqrt(fmax(0.0, r2 - distSqr));
      const double tminKernel = baseMaxt - deltaT;
      baseMaxt = tminKernel + (deltaT * 2) * (double)rb.rand;
      pdfCameraPos = 1.0 / fmax(deltaT * 2.0, 0.0001);
    }
    const MRecD mBase = mediumEvalD(a.med, baseMaxt - eps);
    const d3 sigS = mkd(a.med.sigmaS[0], a.med.sigmaS[1], a.med.sigmaS[2]);
    const d3 flux = tod(ph.flux), wi = tod(ph.wi);
    const double phB = phaseD((double)a.med.g, wi, -base.d);
    const d3 baseContrib = mkd(mBase.tr * (sigS.x * flux.x * phB) * base.eye.x, mBase.tr * (sigS.y * flux.y * phB) * base.eye.y,
                               mBase.tr * (sigS.z * flux.z * phB) * base.eye.z);
    const double scale = rr / (kernelVol * pdfCameraPos);

    ShiftOut out;
    out.flux = mkd(0, 0, 0);
    out.weight = 1.0;
    out.kind = 0;
    if (sh.valid) {
      const double shMaxt = baseMaxt;  // Ray(shiftGather.o, shiftDir, Epsilon, baseRay.maxt)
      const d3 zP = sh.o + sh.d * shMaxt;
      bool alreadyShift = false;
      if (a.cfg.use_shift_null) {
        const double ZPtoY = len2d(zP - pos);
        if (ZPtoY < r2 && shMaxt < sh.len) {
          const d3 o2c = pos - sh.o;
          const double dd = dot(o2c, sh.d);
          const double distSqr = len2d((sh.o + sh.d * dd) - pos);
          const double deltaT = sqrt(fmax(0.0, r2 - distSqr));
          const double pdfShiftPos = 1.0 / fmax(2.0 * deltaT, 0.0001);
          const MRecD mS = mediumEvalD(a.med, shMaxt - eps);
          // shiftNull, :119-158
          out.kind = 1;
          const double phs = phaseD((double)a.med.g, wi, -sh.d);
          out.flux = mkd(mS.tr * (sigS.x * flux.x * phs) * sh.eye.x, mS.tr * (sigS.y * flux.y * phs) * sh.eye.y,
                         mS.tr * (sigS.z * flux.z * phs) * sh.eye.z);
          out.weight = 0.5;
          if (a.cfg.use_mis) {
            if (pdfShiftPos == 0.0 || pdfCameraPos == 0.0) {
              out.weight = 1.0;
            } else {
              const double sensorPart = sensorMISD(sh, base, base.edge, shMaxt, baseMaxt);
              out.weight = 1.0 / (1.0 + sensorPart * pdfShiftPos / pdfCameraPos);
            }
          }
          alreadyShift = true;
        }
      }
      if (!alreadyShift && sh.len >= shMaxt) {
        // getShiftPos, :858-896
        const d3 bP = base.o + base.d * baseMaxt;
        d3 offsetPos = zP + (pos - bP);
        if (!use3D) {
          d3 bs, bt, ns, nt;
          coherentFrameD(base.d, bs, bt);
          coherentFrameD(sh.d, ns, nt);
          const d3 v = pos - bP;
          const double lx = dot(v, bs), ly = dot(v, bt), lz = dot(v, base.d);
          offsetPos = zP + (ns * lx + nt * ly + sh.d * lz);
        }
        if (a.cfg.use_shift_null) {
          const double offDistSqr = len2d(bP - offsetPos);
          if (offDistSqr < r2) {
            d3 dShift = zP - bP;
            dShift = dShift / sqrt(len2d(dShift));
            const double cosD = dot(dShift, -(offsetPos - zP));
            offsetPos = offsetPos + dShift * cosD * 2.0;
          }
        }
        double pdfShiftPos = 1.0;
        if (use3D) {
          const d3 o2c = offsetPos - sh.o;
          const double dd = dot(o2c, sh.d);
          const double distSqr = len2d((sh.o + sh.d * dd) - offsetPos);
          const double deltaT = sqrt(fmax(0.0, r2 - distSqr));
          pdfShiftPos = 1.0 / fmax(2.0 * deltaT, 0.0001);
        }
        if (a.cfg.debug_shift != GVPM_SHIFT_NULL) {
          const MRecD mS = mediumEvalD(a.med, shMaxt - eps);
          reconnectExact(a, ph, offsetPos, sh, base, shMaxt, baseMaxt, mS.tr, pdfCameraPos, pdfShiftPos, out);
        }
      }
    }
    nNull += out.kind == 1 ? 1u : 0u;
    nDiff += out.kind == 2 ? 1u : 0u;
    nFail += out.kind == 3 ? 1u : 0u;
    // (G-BRE keeps the running SUM of the normalised per-iteration estimates: iterScale = 1 / nb_paths)
    addShift(a, rb.pixel, i, out.flux, baseContrib, out.weight, scale, e.outScale, a.iter);
    if (iOnly < 0 && i == 0) {
      // the pair's base term, shift_volume_photon.cpp:748
      nEval++;
      const int px = (int)(rb.pixel & 0xFFFFu), py = (int)(rb.pixel >> 16);
      float *dst = a.iter + ((size_t)py * a.cfg.width + px) * 27;
      const double ws = scale * (double)e.outScale;
      atomicAdd(&dst[0], (float)(baseContrib.x * ws));
      atomicAdd(&dst[1], (float)(baseContrib.y * ws));
      atomicAdd(&dst[2], (float)(baseContrib.z * ws));
    }
  }
}

// ---- G-VPM: VolumeGradientPositionQuery::operator(), shift_volume_photon.cpp:489-655, one shift of one (photon, sample) pair ----
__device__ void exactVPM(const GatherArgs &a, const ExEntry &e, uint32_t &nNull, uint32_t &nDiff, uint32_t &nFail) {
#pragma clang fp contract(off)
  const double M_PI_D = 3.14159265358979323846;
  const int i = (int)((e.meta >> 8) & 0xFFu);
  const gvpm_camera_ray rb = e.rays[0];
  const RayIn base = loadRayIn(rb), sh = loadRayIn(e.rays[1 + i]);
  const PhotonCold ph = coldOf(e);
  // querySize = BBPourcentageCONST * gp.scaleVol (gvpm.cpp:1082,1132) as a double product of the parameters -- not the fast
  // kernel's fp32 product, which is 6e-8 off it: a null-shift test |y|^2 < r^2 that close to equality is exactly what is sent
  // here (tests/stress_vpm.py found one in 2 10^8 shifts that the rounded radius decided the other way)
  const double r = ((double)a.cfg.bsphere_radius * 0.01) * (double)e.extra[0].z, r2 = r * r, eps = (double)a.cfg.epsilon;
  const double rnd = (double)e.extra[0].x, pdfSel = (double)e.extra[0].y;
  const double sigT = (double)a.med.sigmaT[1];
  // sampleDistance(Ray(o, d, Epsilon, len), EDistanceAlwaysValid, rand), homogeneous.cpp:293-430 (currentMediumSampling = 1)
  const double mint = eps, maxt = base.len;
  const double maxDist = fmax((maxt - mint) - eps, 0.0);
  const double normalization = 1.0 - exp(-sigT * maxDist);
  const double sampled = -log(1.0 - rnd * normalization) / sigT;
  const double t = sampled + mint;  // baseRay.maxt
  const double nrm2 = 1.0 - exp(-sigT * (maxt - mint));
  const double tmpB = exp(-sigT * sampled);
  const double pdfBaseRay = (sigT / nrm2) * tmpB * pdfSel;  // baseDistPDF * pdfSelSection
  const double trBase = tmpB < 1e-20 ? 0.0 : tmpB;
  const d3 sigS = mkd(a.med.sigmaS[0], a.med.sigmaS[1], a.med.sigmaS[2]);
  const d3 flux = tod(ph.flux), wi = tod(ph.wi), pos = tod(ph.pos);
  const double phB = phaseD((double)a.med.g, wi, -base.d);
  const d3 baseContrib = mkd(base.eye.x * trBase * (sigS.x * flux.x * phB), base.eye.y * trBase * (sigS.y * flux.y * phB),
                             base.eye.z * trBase * (sigS.z * flux.z * phB));
  const double kernelVol = (4.0 / 3.0) * M_PI_D * (r * r * r);
  const double scale = 1.0 / (kernelVol * pdfBaseRay);
  ShiftOut out;
  out.flux = mkd(0, 0, 0);
  out.weight = 1.0;
  out.kind = 0;
  if (sh.valid && sh.len >= t) {
    // shiftMRec: Medium::eval(Ray(o, d, Epsilon, shiftDistMax), EDistanceAlwaysValid) with mRec.t = baseRay.maxt
    const double nrmS = 1.0 - exp(-sigT * (sh.len - eps));
    const double tmpS = exp(-sigT * t);
    const double pdfShiftRay = (sigT / nrmS) * tmpS * pdfSel;
    const double trS = tmpS < 1e-20 ? 0.0 : tmpS;
    const d3 zP = sh.o + sh.d * t, bP = base.o + base.d * t;
    bool alreadyShifted = false;
    if (a.cfg.use_shift_null) {
      const double distSqr = len2d(pos - zP);
      if (distSqr < r2) {
        alreadyShifted = true;
        out.kind = 1;
        const double phs = phaseD((double)a.med.g, wi, -sh.d);
        out.flux = mkd(trS * (sigS.x * flux.x * phs) * sh.eye.x, trS * (sigS.y * flux.y * phs) * sh.eye.y, trS * (sigS.z * flux.z * phs) * sh.eye.z);
        out.weight = 0.5;
        if (a.cfg.use_mis) {
          if (pdfShiftRay == 0.0 || pdfBaseRay == 0.0) out.weight = 1.0;
          else out.weight = 1.0 / (1.0 + sensorMISD(sh, base, base.edge, t, t) * pdfShiftRay / pdfBaseRay);
        }
      }
    }
    if (!alreadyShifted) {
      // getShiftPos (coherent = false), :858-896
      d3 offsetPos = zP + (pos - bP);
      if (a.cfg.use_shift_null) {
        const double offDistSqr = len2d(bP - offsetPos);
        if (offDistSqr < r2) {
          d3 dShift = zP - bP;
          dShift = dShift / sqrt(len2d(dShift));
          const double cosD = dot(dShift, -(offsetPos - zP));
          offsetPos = offsetPos + dShift * cosD * 2.0;
        }
      }
      if (a.cfg.debug_shift != GVPM_SHIFT_NULL) reconnectExact(a, ph, offsetPos, sh, base, t, t, trS, pdfBaseRay, pdfShiftRay, out);
    }
  }
  nNull += out.kind == 1 ? 1u : 0u;
  nDiff += out.kind == 2 ? 1u : 0u;
  nFail += out.kind == 3 ? 1u : 0u;
  // (G-VPM's accumulators are plain sums over the samples, each weighted 1 / nbCameraSamples: outScale)
  addShift(a, rb.pixel, i, out.flux, baseContrib, out.weight, scale, e.outScale, a.iter);
}

// Behind a gather's kernels: its notes become entries (the gather's record and ray buffers are recycled three gathers on).
// Half a wave per note, a quad per lane; 16 registers.  ONE wave per workgroup: it starts beside the other streams' persistent
// kernels, and a four-wave workgroup waits there until four slots fall free on one CU at the same moment -- as 64 x 256 threads
// this kernel took 90-140 us for a few hundred notes between every two evaluations of the C2 pipeline, as 256 x 64: 20-35 us
// (rocprofv3 kernel trace; the step itself does not move: the other streams fill the gap either way)
__global__ __launch_bounds__(64) void capture_notes_kernel(GatherArgs a) {
  const uint32_t total = a.exOvfCount[0], n = total < a.exOvfCap ? total : a.exOvfCap;
  const uint32_t part = threadIdx.x & 31u, group = (blockIdx.x * blockDim.x + threadIdx.x) >> 5, ngroups = (gridDim.x * blockDim.x) >> 5;
  for (uint32_t e = group; e < n; e += ngroups) {
    uint32_t slot = 0;
    if (part == 0u) slot = atomicAdd(a.exPayCount, 1u);
    slot = __shfl(slot, 0, 32);
    const uint4 nt = a.exOvf[e];
    if (slot < a.exPayCap) reinterpret_cast<float4 *>(a.exPay + slot)[part] = exQuad(a, nt.x, nt.y, nt.z, part);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(a.exOvfCount + 1, 1u) == gridDim.x - 1u) {
      if (total > n) atomicAdd(a.exPayCount + 1, total - n);  // notes that did not fit: lost
      a.exOvfCount[0] = 0u;
      a.exOvfCount[1] = 0u;
    }
  }
}
void launch_capture_notes(const GatherArgs &a, hipStream_t s, uint32_t nblocks) {
  hipLaunchKernelGGL(capture_notes_kernel, dim3(nblocks), dim3(64), 0, s, a);
}

// One pass over the handle's list
// (no register cap: a cap puts its spills in scratch memory, and a kernel with scratch had the runtime re-fit the queue's
// scratch space around it -- 4.8 ms once per process, inside the timed region)
__global__ __launch_bounds__(64) void exact_pass_kernel(GatherArgs a, unsigned long long *totals, uint32_t *hostOut) {
  const uint32_t total = a.exPayCount[0], n = total < a.exPayCap ? total : a.exPayCap;
  uint32_t nEval = 0, nNull = 0, nDiff = 0, nFail = 0;
  for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
    const ExEntry &e = a.exPay[k];
    const uint32_t kind = e.meta & 0xFFu, cause = e.meta >> 16;
    if (totals)
      for (uint32_t c = 0; c < 6u; ++c)
        if ((cause >> c) & 1u) atomicAdd(&totals[4 + c], 1ull);  // by cause: pair, branch, -, mirror, visibility, cosine
    if (kind == GVPM_EX_KIND_BRE) exactBRE(a, e, (int)((e.meta >> 8) & 0xFFu), nEval, nNull, nDiff, nFail);
    else if (kind == GVPM_EX_KIND_BRE_PAIR) exactBRE(a, e, -1, nEval, nNull, nDiff, nFail);
    else if (kind == GVPM_EX_KIND_VPM) exactVPM(a, e, nNull, nDiff, nFail);
  }
  if (hostOut && blockIdx.x == 0 && threadIdx.x == 0) hostOut[0] = total;  // (pinned: the host paces the passes by it)
  finishPass(a, nEval, nNull, nDiff, nFail, n, total, totals);
}

void launch_exact_pass(const GatherArgs &a, unsigned long long *totals, uint32_t *hostOut, hipStream_t s) {
  hipLaunchKernelGGL(exact_pass_kernel, dim3(256), dim3(64), 0, s, a, totals, hostOut);
}

}  // namespace gvpm
