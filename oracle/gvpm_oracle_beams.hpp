// ORACLE -- TEST INFRASTRUCTURE ONLY (see gvpm_oracle.hpp).
//
// G-Beams (beam x beam, 1D kernel and 3D "optimized" kernel): CPU restatement of
//   GPMIntegrator::computeVolumeGradientBeams      gvpm/gvpm.cpp:880-986
//   LTPhotonBeam / LTBeamMap::tryAppendLT          gvpm/gvpm_beams.h:18-84 (flattened by the host)
//   BeamMap::query, ENoAccel loop                  pm/beams.h:286-300
//   BeamKernelRecord                               gvpm/shift/shift_volume_beams.h:24-338
//   PhotonBeam::rayIntersectInternal1D, getContrib pm/beams_struct.h:250-311, 136-185
//   cylinderIntersection                           pm/beams_3d_intersections.h:77-140
//   BeamGradRadianceQuery::operator() and shifts   gvpm/shift/shift_volume_beams.cpp:37-137,139-539,748-786
//   diffuseReconnectionPhotonBeam                  gvpm/shift/operation/shift_diffuse.cpp:136-268
// (pm/ = src/integrators/photonmapper/).
//
// Random numbers: the reference draws sampler->next1D() twice per accepted 3D-kernel hit, in
// BVH traversal order, from a per-image-block SFMT stream (shift_volume_beams.h:224,245) --
// not reproducible on a GPU.  Oracle and device instead derive the two numbers from
// Philox4x32-10 with key = {bits(base ray rand), 0x6265616d} and counter = {beam index, 0, 0, 0}
// (beam index = position of the beam in the uploaded arrays): u_v = out[0], u_w = out[1],
// each mapped to [0,1) as (x >> 8) * 2^-24.
#pragma once

#include "gvpm_oracle.hpp"
#include "gvpm_oracle_accel.hpp"

namespace oracle {

inline void philox4x32_10(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                          uint32_t out[4]) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

template <typename F> inline void beamRandoms(float setRand, uint32_t beamIndex, F &uv, F &uw) {
  uint32_t key;
  std::memcpy(&key, &setRand, 4);
  uint32_t o[4];
  philox4x32_10(key, 0x6265616du, beamIndex, 0, 0, 0, o);
  uv = (F)((float)(o[0] >> 8) * (1.0f / 16777216.0f));
  uw = (F)((float)(o[1] >> 8) * (1.0f / 16777216.0f));
}

// LTPhotonBeam as flattened by the host: the Photon<F> record re-read as
//   pos = vertex(i+1) (beam end), parentPos = vertex(i) (beam origin p1), flux = LTPhotonBeam::flux,
//   depth bits = edgeID i, shift type = getTypeShift(path, i+1), parent* = vertex(i),
// plus the geometric normal of the end vertex (zero when it is a medium interaction).
template <typename F> struct Beam {
  Photon<F> ph;
  Vec3<F> endN;
  bool endOnSurface;
  Vec3<F> dir;
  F length;
  uint32_t index;
  Vec3<F> getPos(F v) const { return ph.parentPos + dir * v; }
};

// solveQuadraticDouble, src/libcore/util.cpp:487-525
inline bool solveQuadraticDouble(double a, double b, double c, double &x0, double &x1) {
  if (a == 0) {
    if (b != 0) {
      x0 = x1 = -c / b;
      return true;
    }
    return false;
  }
  double discrim = b * b - 4.0f * a * c;
  if (discrim < 0) return false;
  double temp, sqrtDiscrim = std::sqrt(discrim);
  if (b < 0) temp = -0.5f * (b - sqrtDiscrim);
  else temp = -0.5f * (b + sqrtDiscrim);
  x0 = temp / a;
  x1 = c / temp;
  if (x0 > x1) std::swap(x0, x1);
  return true;
}

// cylinderIntersection, pm/beams_3d_intersections.h:77-140 (note the float intermediates)
template <typename F>
inline bool cylinderIntersection(const Ray<F> &rCylinder, const Ray<F> &view, F radius, double &tNear, double &tFar) {
  typedef Vec3<F> V;
  const V d1d2c = cross(view.d, rCylinder.d);
  const float sinThetaSqr = (float)dot(d1d2c, d1d2c);
  const float ad = (float)dot((rCylinder.o - view.o), d1d2c);
  if (ad * ad >= (radius * radius) * sinThetaSqr) return false;
  // worldToObject = (translate(o) * fromFrame(Frame(d)))^-1 : local = Frame^T (p - o)
  V s, t, n = rCylinder.d;
  coordinateSystem(n, s, t);
  const F lMax = rCylinder.maxt;
  const V rel = view.o - rCylinder.o;
  const V ro(dot(s, rel), dot(t, rel), dot(n, rel));
  const V rd(dot(s, view.d), dot(t, view.d), dot(n, view.d));
  const double ox = ro.x, oy = ro.y, dx = rd.x, dy = rd.y;
  const double A = dx * dx + dy * dy;
  const double B = 2 * (dx * ox + dy * oy);
  const double C = ox * ox + oy * oy - radius * radius;
  if (!solveQuadraticDouble(A, B, C, tNear, tFar)) return false;
  if (tNear > view.maxt || tFar < 0) return false;
  const double zPosNear = ro.z + rd.z * tNear;
  const double zPosFar = ro.z + rd.z * tFar;
  if (zPosNear < 0) {
    if (zPosFar < 0) return false;
    float th = (float)(tNear + (tFar - tNear) * (zPosNear) / (zPosNear - zPosFar));
    tNear = th;
    return true;
  } else if (zPosNear >= 0 && zPosNear < lMax) {
    return true;
  } else if (zPosNear > lMax) {
    if (zPosFar > lMax) return false;
    float th = (float)(tNear + (tFar - tNear) * (zPosNear - lMax) / (zPosNear - zPosFar));
    tNear = th;
    return true;
  }
  return false;
}

// BeamKernelRecord, shift_volume_beams.h:24-338
template <typename F> struct BeamKernelRecord {
  typedef Vec3<F> V;
  F radius = 0, v = 0, w = 0, pdfKernel = 0, pdfEdgeFailure = 0, u = 0;
  int volTechnique = GVPM_BEAM_BEAM_1D;
  V beamTrans, contrib;
  F weightKernel = 0;

  bool isValid() const { return !(contrib.x == 0 && contrib.y == 0 && contrib.z == 0); }
  F pdf() const { return pdfEdgeFailure * pdfKernel; }

  // PhotonBeam::rayIntersectInternal1D, pm/beams_struct.h:250-311
  static bool rayIntersect1D(const Beam<F> &beam, F radius, const Ray<F> &_ray, F tminBeam, F tmaxBeam, F &u, F &v,
                             F &w, F &sinTheta) {
    const V d1d2c = cross(_ray.d, beam.dir);
    const float sinThetaSqr = (float)dot(d1d2c, d1d2c);
    const float ad = (float)dot((beam.ph.parentPos - _ray.o), d1d2c);
    if (ad * ad >= (radius * radius) * sinThetaSqr) return false;
    const float d1d2 = (float)dot(_ray.d, beam.dir);
    const float d1d2Sqr = d1d2 * d1d2;
    const float d1d2SqrMinus1 = d1d2Sqr - 1.0f;
    if (d1d2SqrMinus1 < 1e-5f && d1d2SqrMinus1 > -1e-5f) return false;
    const float d1O1 = (float)dot(_ray.d, _ray.o);
    const float d1O2 = (float)dot(_ray.d, beam.ph.parentPos);
    w = (d1O1 - d1O2 - d1d2 * (dot(beam.dir, _ray.o) - dot(beam.dir, beam.ph.parentPos))) / d1d2SqrMinus1;
    if (w <= _ray.mint || w >= _ray.maxt) return false;
    v = (w + d1O1 - d1O2) / d1d2;
    if (v <= 0.0 || v >= beam.length || std::isnan(v)) return false;
    if (tminBeam >= v || tmaxBeam < v) return false;
    const float sinThetaConst = std::sqrt(sinThetaSqr);
    u = std::abs(ad) / sinThetaConst;
    sinTheta = sinThetaConst;
    return true;
  }

  // PhotonBeam::getContrib (short beams), pm/beams_struct.h:136-185
  static V getContrib(const Medium<F> &med, const Beam<F> &beam, F v, const MRec<F> &mRecCamera, const V &d,
                      V &beamTransmittance, F &pdfFailure) {
    Ray<F> rayTrans(beam.ph.parentPos, beam.dir, (F)0, v);
    MRec<F> mRec;
    med.eval(rayTrans, mRec);
    beamTransmittance = mRec.transmittance;
    F phaseTerm = med.phase(-beam.dir, -d);
    V beamContrib = mRec.transmittance * mRecCamera.transmittance * mRec.sigmaS * beam.ph.flux * phaseTerm;
    if (mRec.pdfFailure == 0 && mRec.transmittance.max() != 0) {
      pdfFailure = mRec.pdfFailure;
      return V((F)0);
    }
    beamContrib /= mRec.pdfFailure;
    pdfFailure = mRec.pdfFailure;
    return beamContrib;
  }

  // eval(), shift_volume_beams.h:157-290
  void eval(const GatherContext<F> &ctx, const Beam<F> &beam, const Ray<F> &cameraRay, F tmin, F tmax, F uv, F uw) {
    const double M_PI_D = 3.14159265358979323846;
    if (tmax > beam.length) tmax = beam.length;
    if (volTechnique == GVPM_BEAM_BEAM_1D) {
      if (!rayIntersect1D(beam, radius, cameraRay, tmin, tmax, u, v, w, pdfKernel)) return;
      Ray<F> cameraRayEval(cameraRay.o, cameraRay.d, (F)0, w);
      MRec<F> mRecCamera;
      ctx.medium.eval(cameraRayEval, mRecCamera);
      weightKernel = 0.5f / radius;
      pdfEdgeFailure = 0.f;
      contrib = getContrib(ctx.medium, beam, v, mRecCamera, cameraRayEval.d, beamTrans, pdfEdgeFailure);
      if (isValid()) contrib /= pdfKernel;
    } else if (volTechnique == GVPM_BEAM_BEAM_3D_OPTIMIZED) {
      Ray<F> _cam(cameraRay(cameraRay.mint), cameraRay.d, (F)0, cameraRay.maxt - cameraRay.mint);
      Ray<F> _beam(beam.ph.parentPos, beam.dir, (F)0, beam.length);
      double tNearBeam, tFarBeam;
      if (!cylinderIntersection(_cam, _beam, radius, tNearBeam, tFarBeam)) return;
      if (tNearBeam < 0 && tmin <= ctx.Epsilon) {
      } else if (tNearBeam > tmin && tNearBeam < tmax) {
      } else {
        return;
      }
      v = (F)(tNearBeam + (tFarBeam - tNearBeam) * uv);
      pdfKernel = (F)(1.0 / std::max(tFarBeam - tNearBeam, 0.0001));
      if (v < 0 || v > beam.length) return;
      V kernelCentroid = beam.getPos(v);
      F distToProj = dot(kernelCentroid - cameraRay.o, cameraRay.d);
      F distSqr = (cameraRay(distToProj) - kernelCentroid).lengthSquared();
      F radSqr = radius * radius;
      if (distSqr >= radSqr) return;
      F deltaT = safe_sqrt(radSqr - distSqr);
      w = distToProj - deltaT + 2 * deltaT * uw;
      pdfKernel *= (F)(1.0 / std::max(2.0 * (double)deltaT, 0.0001));
      if (w < cameraRay.mint || w > cameraRay.maxt) return;
      Ray<F> rayTrans(beam.ph.parentPos, beam.dir, (F)0, v);
      MRec<F> mRecBeam;
      ctx.medium.eval(rayTrans, mRecBeam);
      Ray<F> cameraRayEval(cameraRay.o, cameraRay.d, (F)0, w);
      MRec<F> mRecCamera;
      ctx.medium.eval(cameraRayEval, mRecCamera);
      F phaseTerm = ctx.medium.phase(-beam.dir, -cameraRay.d);
      F kernelVol = (F)((4.0 / 3.0) * M_PI_D * std::pow((double)radius, 3));
      contrib = beam.ph.flux * mRecBeam.transmittance * mRecCamera.sigmaS * mRecCamera.transmittance * phaseTerm / pdfKernel;
      weightKernel = (F)(1.0 / kernelVol);
      beamTrans = mRecBeam.transmittance;
      contrib /= mRecBeam.pdfFailure;
      pdfEdgeFailure = mRecBeam.pdfFailure;
    }
  }

  // copy-shift constructor (null shift of the 3D kernel), shift_volume_beams.h:40-144
  static BeamKernelRecord shifted(const BeamKernelRecord &ori, const GatherContext<F> &ctx, const Beam<F> &beam,
                                  const Ray<F> &cameraRay) {
    BeamKernelRecord k;
    k.radius = ori.radius;
    k.u = 0.f;
    k.volTechnique = ori.volTechnique;
    k.contrib = V((F)0);
    if (ori.volTechnique == GVPM_BEAM_BEAM_1D) {
      k.eval(ctx, beam, cameraRay, (F)0, beam.length, (F)0, (F)0);
    } else if (ori.volTechnique == GVPM_BEAM_BEAM_3D_OPTIMIZED) {
      Ray<F> _cam(cameraRay(cameraRay.mint), cameraRay.d, (F)0, cameraRay.maxt - cameraRay.mint);
      Ray<F> _beam(beam.ph.parentPos, beam.dir, (F)0, beam.length);
      double tNearBeam, tFarBeam;
      if (!cylinderIntersection(_cam, _beam, k.radius, tNearBeam, tFarBeam)) return k;
      k.v = ori.v;
      k.pdfKernel = (F)(1.0 / std::max(tFarBeam - tNearBeam, 0.0001));
      if (k.v < 0 || k.v > beam.length) return k;
      V kernelCentroid = beam.getPos(k.v);
      F distToProj = dot(kernelCentroid - cameraRay.o, cameraRay.d);
      F distSqr = (cameraRay(distToProj) - kernelCentroid).lengthSquared();
      F radSqr = k.radius * k.radius;
      if (distSqr >= radSqr) return k;  // SLog(EError, "Instersection problem") in the reference
      F deltaT = safe_sqrt(radSqr - distSqr);
      k.w = ori.w;
      k.pdfKernel *= (F)(1.0 / std::max(2.0 * (double)deltaT, 0.0001));
      if (k.w < cameraRay.mint || k.w > cameraRay.maxt) return k;  // SLog(EError, ...) in the reference
      k.contrib = ori.contrib * (ori.pdfKernel / k.pdfKernel);
      k.weightKernel = ori.weightKernel;
      k.beamTrans = ori.beamTrans;
      k.pdfEdgeFailure = ori.pdfEdgeFailure;
    }
    return k;
  }

  // kernelPDF(), shift_volume_beams.h:300-336
  F kernelPDF(const Ray<F> &cameraRay, const V &orgBeam, const V &dBeam, const F newDLength) const {
    if (volTechnique == GVPM_BEAM_BEAM_1D) {
      return std::sqrt(cross(cameraRay.d, dBeam).lengthSquared());
    } else if (volTechnique == GVPM_BEAM_BEAM_3D_OPTIMIZED) {
      // Ray photonRay(orgBeam, dBeam, 0.f): the 3-argument constructor leaves maxt = +inf
      F _pdfKernel = 0;
      Ray<F> _beam(orgBeam, dBeam, (F)0, std::numeric_limits<F>::infinity());
      Ray<F> _cam(cameraRay.o, cameraRay.d, (F)0, cameraRay.maxt);
      double tNearBeam, tFarBeam;
      if (cylinderIntersection(_cam, _beam, radius, tNearBeam, tFarBeam)) {
        _pdfKernel = (F)(1.0 / std::max(tFarBeam - tNearBeam, 0.0001));
        V kernelCentroid = orgBeam + dBeam * newDLength;
        F distToProj = dot(kernelCentroid - cameraRay.o, cameraRay.d);
        F distSqr = (cameraRay(distToProj) - kernelCentroid).lengthSquared();
        F radSqr = radius * radius;
        if (distSqr < radSqr) {
          F deltaT = safe_sqrt(radSqr - distSqr);
          _pdfKernel *= (F)(1.0 / std::max(2.0 * (double)deltaT, 0.0001));
          return _pdfKernel;
        } else {
          return 0.f;
        }
      }
      return 0.f;
    }
    return 0.f;
  }
};

// Frame(n) with toLocal / toWorld, include/mitsuba/core/frame.h
template <typename F> struct FrameT {
  Vec3<F> s, t, n;
  Vec3<F> toLocal(const Vec3<F> &v) const { return Vec3<F>(dot(v, s), dot(v, t), dot(v, n)); }
  Vec3<F> toWorld(const Vec3<F> &v) const { return s * v.x + t * v.y + n * v.z; }
};

// BeamGradRadianceQuery, shift_volume_beams.{h,cpp}
template <typename F> struct BeamGradRadianceQuery {
  typedef Vec3<F> V;
  const GatherContext<F> &ctx;
  const CamRay<F> *baseGather;
  const CamRay<F> *shiftGPs;
  Ray<F> baseCameraRay;
  int currCameraEdge;
  Counters cnt;
  V mediumFlux, shiftedMediumFlux[4], weightedMediumFlux[4];

  BeamGradRadianceQuery(const GatherContext<F> &c, const CamRay<F> *base, const CamRay<F> *shifts, const Ray<F> &ray)
      : ctx(c), baseGather(base), shiftGPs(shifts), baseCameraRay(ray), currCameraEdge(base->edge) {
    for (int i = 0; i < 4; ++i) shiftedMediumFlux[i] = weightedMediumFlux[i] = V((F)0);
    mediumFlux = V((F)0);
  }

  // localMatrix + shift(), shift_volume_beams.cpp:37-79 (Frame(n, s, t) = {r.d, s, t} read as
  // Frame(s = r.d, t = s, n = t): the brace initialiser fills Frame's members s, t, n in order)
  static V shiftPoint(const Ray<F> &r, const V &a, F u, F w, bool flip) {
    const F M_PI_2_ = (F)1.57079632679489661923;
    F d = dot(a - r.o, r.d);
    V sv = normalize(a - r(d));
    V tv = cross(r.d, sv);
    FrameT<F> lT{r.d, sv, tv};
    V tD = r(d);
    V localA = lT.toLocal(a - tD);
    F x = u / std::fabs(localA.y);
    F as = std::asin(std::min((F)1, std::max((F)-1, x)));  // math::safe_asin
    F phi = M_PI_2_ - as;
    if (flip) phi = -phi;
    V localW((F)0, u * std::cos(phi), u * std::sin(phi));
    V worldU = lT.toWorld(localW);
    return r(w) + worldU;
  }

  // getShiftPos1D, shift_volume_beams.cpp:81-96
  static V getShiftPos1D(const Ray<F> &bRay, const Ray<F> &sRay, const V &a, const V &bBeamDir, F w, F u) {
    V baseShitedBack = normalize(shiftPoint(bRay, a, u, w, false) - a);
    bool flipAngle = false;
    if ((baseShitedBack - bBeamDir).lengthSquared() > 0.001) flipAngle = true;
    return shiftPoint(sRay, a, u, w, flipAngle);
  }

  // getShiftPos (coherent = true), shift_volume_beams.cpp:98-137
  V getShiftPos(const Ray<F> &bRay, const Ray<F> &sRay, F w, const V &u, F radius, F newW) const {
    V newPos;
    {
      V bn = bRay.d, bs, bt, nn = sRay.d, ns, nt;
      coordinateSystemCoherent(bn, bs, bt);
      coordinateSystemCoherent(nn, ns, nt);
      const V local(dot(u, bs), dot(u, bt), dot(u, bn));
      newPos = sRay(newW) + (ns * local.x + nt * local.y + nn * local.z);
    }
    if (ctx.cfg.use_shift_null) {
      const V bCamW = bRay(w);
      F offDistSqr = (bCamW - newPos).lengthSquared();
      if (offDistSqr < radius * radius) {
        V dShift = sRay(newW) - bCamW;
        dShift /= dShift.length();
        const F cosD = dot(dShift, -(newPos - sRay(newW)));
        newPos += dShift * cosD * (F)2;
      }
    }
    return newPos;
  }

  // diffuseReconnectionPhotonBeam (short beams), shift_diffuse.cpp:136-268
  bool diffuseReconnectionPhotonBeam(ShiftRecord<F> &sRec, const V &basePos, const V &newD, const F newDLength,
                                     const Beam<F> &beam, F pdfEdgeAndKernel) const {
    const F INV_PI = (F)0.31830988618379067154;
    const Photon<F> &ph = beam.ph;
    F pdfValueSA;
    const unsigned ptype = GVPM_PF_PARENT_TYPE(ph.flags);
    if (ptype == GVPM_PARENT_SURFACE) {
      F cosWo = dot(ph.parentN, newD), cosWi = dot(ph.parentN, ph.parentWi);
      if (cosWi <= 0 || cosWo <= 0) {
        sRec.throughtput *= V((F)0);
        pdfValueSA = 0;
      } else {
        sRec.throughtput *= ph.parentScat * (INV_PI * cosWo);
        pdfValueSA = INV_PI * cosWo;
      }
      if (cosWi * cosWi <= 0 || cosWo * cosWo <= 0) return false;
    } else if (ptype == GVPM_PARENT_SURFACE_BSDF) {
      // a glossy parent (gvpm_upload_bsdfs): shift_diffuse.cpp:150-176 evaluates the parent's BSDF as :25-41 does
      const std::vector<gvpm_bsdf> &tab = bsdfTable();
      const size_t bi = (size_t)ph.parentG;
      V f;
      if (!(ph.parentG >= 0) || bi >= tab.size() || !glossyEvalPdf<F>(tab[bi], ph.parentScat, ph.parentN, ph.parentWi, newD, f, pdfValueSA)) {
        sRec.throughtput *= V((F)0);
        sRec.pdf = 0;
        return false;
      }
      sRec.throughtput *= f;
      F cosWo = dot(ph.parentN, newD), cosWi = dot(ph.parentN, ph.parentWi);
      if (cosWi * cosWi <= 0 || cosWo * cosWo <= 0) return false;
    } else if (ptype == GVPM_PARENT_MEDIUM) {
      F p = Medium<F>::phaseEval(ph.parentG, ph.parentWi, newD);
      sRec.throughtput *= ph.parentScat * p;
      pdfValueSA = p;
    } else {
      F dp = dot(newD, ph.parentN);
      if (dp < 0) dp = 0.0f;
      sRec.throughtput *= V(INV_PI * dp);
      pdfValueSA = INV_PI * dp;
    }
    F GOpNew = 1 / (newDLength * newDLength);
    sRec.pdf = pdfValueSA * GOpNew;
    sRec.throughtput *= GOpNew;
    sRec.jacobian = 1.0f;
    // parentVertex->pdf[EImportance] * |parent - baseVertex|^2 [/ absDot(n_end, edge.d)] * 1/|parent - basePos|^2
    F pdfBasePos = ph.parentPdf * (ph.parentPos - ph.pos).lengthSquared();
    if (beam.endOnSurface) pdfBasePos /= std::abs(dot(beam.endN, beam.dir));
    F GOpBase = 1.0f / (ph.parentPos - basePos).lengthSquared();
    pdfBasePos *= GOpBase;
    if (pdfBasePos == 0.f) {
      sRec.pdf = 0.f;
      return false;
    }
    sRec.throughtput /= pdfBasePos;
    sRec.throughtput *= ph.parentRR;
    if (GVPM_PF_EDGE_IN_MEDIUM(ph.flags)) {
      MRec<F> mRecShift;
      Ray<F> mRay(ph.parentPos, newD, (F)0, newDLength);
      ctx.medium.eval(mRay, mRecShift);
      sRec.pdf *= mRecShift.pdfFailure;
      sRec.throughtput *= mRecShift.transmittance / pdfEdgeAndKernel;
    }
    return true;
  }

  // shiftBeamDiffuse, shift_volume_beams.cpp:410-539
  bool shiftBeamDiffuse(const Beam<F> &beam, const CamRay<F> &shiftGP, const Ray<F> &shiftRay, F shiftW,
                        GradientSamplingResult<F> &result, const BeamKernelRecord<F> &kRec, const V &newPos) {
    const Photon<F> &ph = beam.ph;
    V newPBDir = (newPos - ph.parentPos);
    F newPBDist = newPBDir.length();
    newPBDir /= newPBDist;
    Ray<F> newRayPB(ph.parentPos, newPBDir, ctx.Epsilon, newPBDist);
    if (ctx.scene.rayIntersect(newRayPB)) {
      result.weight = 1.0f;
      return false;
    }
    V basePos = beam.getPos(kRec.v);
    V shiftPhotonWeight = ph.prefixW;
    F pdfKernelAndDist = kRec.pdf();
    ShiftRecord<F> sRec;
    {
      diffuseReconnectionPhotonBeam(sRec, basePos, newPBDir, newPBDist, beam, pdfKernelAndDist);
      if (sRec.pdf == (F)0) {
        result.weight = 1.0f;
        return false;
      }
    }
    result.jacobian *= sRec.jacobian;
    F shiftKernelPDF = kRec.kernelPDF(shiftRay, ph.parentPos, newPBDir, newPBDist);
    if (shiftKernelPDF == 0) {
      result.weight = 1.0f;
      return false;
    }
    shiftPhotonWeight *= sRec.throughtput;
    V eyeShiftContrib = shiftGP.eye;
    MRec<F> mRecShift;
    Ray<F> shiftRayEval(shiftRay.o, shiftRay.d, (F)0, shiftW);
    ctx.medium.eval(shiftRayEval, mRecShift);
    F phaseTerm = ctx.medium.phase(-newPBDir, -shiftRay.d);
    shiftPhotonWeight *= mRecShift.transmittance * mRecShift.sigmaS * phaseTerm;
    result.shiftedFlux = shiftPhotonWeight * eyeShiftContrib * result.jacobian;
    result.weight = 0.5f;
    if (ctx.cfg.use_mis) {
      F basePdf = ph.parentPdf;
      basePdf *= (ph.parentPos - ph.pos).lengthSquared();
      if (beam.endOnSurface) basePdf /= std::abs(dot(beam.endN, beam.dir));
      basePdf /= (ph.parentPos - basePos).lengthSquared();
      basePdf *= pdfKernelAndDist;
      F offsetPdf = shiftKernelPDF;
      offsetPdf *= sRec.pdf;
      if (offsetPdf == (F)0 || basePdf == (F)0) {
        result.weight = 1.0f;
        return false;
      }
      const F sensorPart = sensorMIS(shiftGP, *baseGather, currCameraEdge, shiftW, kRec.w);
      if (ctx.cfg.power_heuristic) {
        F x = sensorPart * offsetPdf * result.jacobian / basePdf;
        result.weight = 1.0f / (1.0f + x * x);
      } else {
        result.weight = 1.0f / (1.0f + sensorPart * offsetPdf * result.jacobian / basePdf);
      }
    }
    return true;
  }

  // shiftBeam dispatch, shift_volume_beams.cpp:355-408
  bool shiftBeam(const V &newPos, const Beam<F> &beam, const CamRay<F> &shiftGP, const Ray<F> &shiftRay, F shiftW,
                 GradientSamplingResult<F> &result, const BeamKernelRecord<F> &kRec) {
    if (ctx.cfg.debug_shift == GVPM_SHIFT_NULL) {
      result.weight = 1.0f;
      return false;
    }
    if (shiftW > shiftRay.maxt) {
      result.weight = 1.0f;
      return false;
    }
    const unsigned st = GVPM_PF_SHIFT_TYPE(beam.ph.flags);
    bool ok = false;
    if (st == 1 || st == 2) ok = shiftBeamDiffuse(beam, shiftGP, shiftRay, shiftW, result, kRec, newPos);
    // EManifoldShift: without useManifold the reference returns false (shift_volume_beams.cpp:398-404); with it, shiftBeamME
    else if (st == 3 && ctx.cfg.use_manifold) ok = shiftBeamME(beam, shiftGP, shiftRay, shiftW, result, kRec, newPos);
    // invalid: return false, result untouched
    if (ok) cnt.diffuseShifts++; else cnt.failedShifts++;
    return ok;
  }

  // The manifold walk (generateShiftPathME + ShiftME + SpecularManifold::det on the functor's cached source path,
  // shift_volume_beams.cpp:541-599,601-640) is Mitsuba's and NOT restated: as for the photons (gvpm_oracle.hpp
  // standinManifoldWalk) a STAND-IN answers in the tests of the host-shift round trip -- a smooth closed-form function of the
  // request (the offset position) and the beam's origin, shaped like a reconnection to it.  `wi` is the proposal's last edge
  // as a vector FROM the new vertex TO its predecessor: direction and length (kernelPDF needs both).
  struct HostShiftBeam {
    bool ok;
    V throughput, wi;
    F pdf, detRatio, basePdf;
  };
  static HostShiftBeam standinBeamWalk(const V &newPos, const Beam<F> &beam) {
    HostShiftBeam r;
    const Photon<F> &ph = beam.ph;
    r.wi = ph.parentPos - newPos;
    const F len = r.wi.length(), lenB = beam.length;
    r.ok = len > (F)0 && len < (F)3 * lenB;
    // (softened by a tenth of the beam's length: an offset position next to the beam's origin would otherwise make the
    // answer a steep function of the request's last bits, and the test would measure that instead of the device)
    const F lenE = std::sqrt(len * len + (F)0.01 * lenB * lenB);
    const F q = (lenB * lenB) / (lenE * lenE);
    r.throughput = ph.prefixW * (lenB / lenE);
    r.pdf = ph.parentPdf * q;
    r.detRatio = q;
    r.basePdf = ph.parentPdf * ph.edgePdf;
    return r;
  }

  // shiftBeamME, shift_volume_beams.cpp:601-746, from the walk's results on
  bool shiftBeamME(const Beam<F> &beam, const CamRay<F> &shiftGP, const Ray<F> &shiftRay, F shiftW,
                   GradientSamplingResult<F> &result, const BeamKernelRecord<F> &kRec, const V &newPos) {
    const HostShiftBeam hs = standinBeamWalk(newPos, beam);
    if (!hs.ok) {  // generateShiftPathME / ShiftME failed (:624-646)
      result.weight = 1.0f;
      return false;
    }
    const F newLen = hs.wi.length();
    const V edgeD = -hs.wi / newLen;  // proposal.edge(c - 1)->d: from vertex c - 1 to the new vertex
    const V orgBeam = newPos + hs.wi;  // proposal.vertex(c - 1)->getPosition()
    F shiftKernelPDF = kRec.kernelPDF(shiftRay, orgBeam, edgeD, newLen);  // (:653-656)
    if (shiftKernelPDF == 0) {
      result.weight = 1.0f;
      return false;
    }
    result.jacobian *= (F)1;          // sRecME.jacobian (shift_utilities.h:30)
    result.jacobian *= hs.detRatio;   // detProposed / cacheDetSource (:674-679)
    if (result.jacobian <= 0.0 || !std::isfinite(result.jacobian)) {
      result.weight = 1.0f;
      return false;
    }
    V shiftPhotonWeight = hs.throughput;
    MRec<F> mRecShift;
    Ray<F> shiftRayEval(shiftRay.o, shiftRay.d, (F)0, shiftW);
    ctx.medium.eval(shiftRayEval, mRecShift);
    F phaseTerm = ctx.medium.phase(-edgeD, -shiftRay.d);
    shiftPhotonWeight *= mRecShift.transmittance * mRecShift.sigmaS * phaseTerm;
    V eyeShiftContrib = shiftGP.eye;
    result.shiftedFlux = shiftPhotonWeight * eyeShiftContrib * result.jacobian;
    result.weight = 0.5f;
    if (ctx.cfg.use_mis) {
      F offsetPdf = hs.pdf * shiftKernelPDF;
      F basePdf = hs.basePdf;
      if (basePdf == (F)0) {
        result.weight = 0.0f;
      } else if (offsetPdf == (F)0) {
        result.weight = 1.0f;
      } else {
        const F sensorPart = sensorMIS(shiftGP, *baseGather, currCameraEdge, shiftW, kRec.w);
        if (ctx.cfg.power_heuristic) {
          F x = sensorPart * result.jacobian * (offsetPdf / basePdf);
          result.weight = 1.0f / (1.0f + x * x);
        } else {
          result.weight = 1.0f / (1.0f + sensorPart * result.jacobian * (offsetPdf / basePdf));
        }
      }
    }
    return true;
  }

  // shiftNull3D, shift_volume_beams.cpp:748-786
  bool shiftNull3D(const CamRay<F> &shiftGP, GradientSamplingResult<F> &result, const BeamKernelRecord<F> &kRec,
                   BeamKernelRecord<F> &kRecShift) {
    if (!kRecShift.isValid()) {
      result.weight = 1.0f;
      return false;
    }
    cnt.nullShifts++;
    V eyeShiftContrib = shiftGP.eye;
    kRecShift.contrib *= kRecShift.pdf() / kRec.pdf();
    result.jacobian = 1.0;
    result.shiftedFlux = kRecShift.contrib * eyeShiftContrib * result.jacobian;
    result.weight = 0.5f;
    if (ctx.cfg.use_mis) {
      F basePdf = kRec.pdf();
      F offsetPdf = kRecShift.pdf();
      if (offsetPdf == (F)0 || basePdf == (F)0) {
        result.weight = 1.0f;
        return false;
      }
      const F sensorPart = sensorMIS(shiftGP, *baseGather, currCameraEdge, kRecShift.w, kRec.w);
      if (ctx.cfg.power_heuristic) {
        F x = sensorPart * result.jacobian * (offsetPdf / basePdf);
        result.weight = 1.0f / (1.0f + x * x);
      } else {
        result.weight = 1.0f / (1.0f + sensorPart * result.jacobian * (offsetPdf / basePdf));
      }
    }
    return true;
  }

  // operator(), shift_volume_beams.cpp:139-353 (excludeLater is dead code: `if (false)`)
  bool operator()(const Beam<F> &beam, F tmin, F tmax) {
    const gvpm_params &config = ctx.cfg;
    cnt.candidates++;
    F rrGlobalWeight = 1;
    int pathLength = currCameraEdge + (int)GVPM_PF_DEPTH(beam.ph.flags);
    if ((config.max_depth > 0 && pathLength > config.max_depth)) return false;
    {
      VolumeGradientRecord<F> helper(ctx, baseGather, shiftGPs);
      if (!helper.computeVolumeContribution(beam.ph)) return false;
    }
    if (config.path_set) {
      unsigned currentGroup = (unsigned)((baseGather->px + baseGather->py) % 2);
      if (beam.ph.pathID % 2 != currentGroup) return false;
      rrGlobalWeight = 2;
    }
    const F radius = ctx_radius;
    F uv, uw;
    beamRandoms<F>((float)baseGather->rand, beam.index, uv, uw);
    BeamKernelRecord<F> kRec;
    kRec.radius = radius;
    kRec.volTechnique = config.vol_technique;
    kRec.contrib = V((F)0);
    kRec.eval(ctx, beam, baseCameraRay, tmin, tmax, uv, uw);
    if (!kRec.isValid()) return false;
    V eyeContrib = baseGather->eye;
    V baseContrib = eyeContrib * kRec.contrib * kRec.weightKernel;
    mediumFlux += baseContrib * rrGlobalWeight;
    {
      int currShift = VolumeGradientRecord<F>::shiftTypeEnum(beam.ph);
      if (config.debug_shift != GVPM_SHIFT_ALL && config.debug_shift != GVPM_SHIFT_NULL &&
          config.debug_shift != currShift)
        return false;
    }
    cnt.evaluations++;
    for (int i = 0; i < 4; ++i) {
      GradientSamplingResult<F> result;
      if (shiftGPs[i].valid) {
        F shiftDistMAX = shiftGPs[i].len;
        Ray<F> shiftRay(shiftGPs[i].o, shiftGPs[i].d, ctx.Epsilon, shiftDistMAX);
        F shiftW = kRec.w;
        bool alreadyShift = false;
        if (config.use_shift_null) {
          V kernelPos = beam.getPos(kRec.v);
          const F ZPtoY = (shiftRay(shiftW) - kernelPos).lengthSquared();
          if (config.vol_technique == GVPM_BEAM_BEAM_1D) {
          } else {
            if (ZPtoY < radius * radius && kRec.w <= shiftDistMAX) {
              BeamKernelRecord<F> kRecShift = BeamKernelRecord<F>::shifted(kRec, ctx, beam, shiftRay);
              if (kRecShift.isValid()) {
                shiftNull3D(shiftGPs[i], result, kRec, kRecShift);
                alreadyShift = true;
              }
            }
          }
        }
        if (!alreadyShift && kRec.w <= shiftDistMAX) {
          const bool newShiftBeam = config.vol_technique == GVPM_BEAM_BEAM_1D;  // gvpm.cpp:96-98
          if (!newShiftBeam) {
            F minDistSqr = (beam.ph.parentPos - shiftRay(dot(beam.ph.parentPos - shiftRay.o, shiftRay.d))).lengthSquared();
            if (minDistSqr > kRec.u * kRec.u) {
              V offsetPos = getShiftPos(baseCameraRay, shiftRay, kRec.w, beam.getPos(kRec.v) - baseCameraRay(kRec.w),
                                        radius, shiftW);
              shiftBeam(offsetPos, beam, shiftGPs[i], shiftRay, shiftW, result, kRec);
            } else {
              result.weight = 1.f;
            }
          } else {
            V offsetPos = getShiftPos1D(baseCameraRay, shiftRay, beam.ph.parentPos, beam.dir, kRec.w, kRec.u);
            shiftBeam(offsetPos, beam, shiftGPs[i], shiftRay, shiftW, result, kRec);
          }
        }
      } else {
        result.weight = 1.f;
      }
      result.shiftedFlux *= kRec.weightKernel;
      if ((i == GVPM_RIGHT && baseGather->px == config.width - 1) ||
          (i == GVPM_TOP && baseGather->py == config.height - 1)) {
        result.weight = 1.0f;
      }
      shiftedMediumFlux[i] += result.shiftedFlux * result.weight * rrGlobalWeight;
      weightedMediumFlux[i] += baseContrib * result.weight * rrGlobalWeight;
    }
    return true;
  }

  F ctx_radius = 0;
};

template <typename F> struct BeamMapO {
  std::vector<Beam<F>> beams;
  void load(const gvpm_photon_soa &s, const float *endN) {
    PhotonMap<F> tmp;
    tmp.load(s);
    beams.resize(s.n);
    for (uint64_t i = 0; i < s.n; ++i) {
      Beam<F> &b = beams[i];
      b.ph = tmp.photons[i];
      b.endN = Vec3<F>(endN + 3 * i);
      b.endOnSurface = !(b.endN.x == 0 && b.endN.y == 0 && b.endN.z == 0);
      // PhotonBeam::setEndPoint, pm/beams_struct.h:73-81
      b.dir = b.ph.pos - b.ph.parentPos;
      b.length = b.dir.length();
      b.dir /= b.length;
      b.index = (uint32_t)i;
    }
  }
};

// One beam set of computeVolumeGradientBeams' inner loop, gvpm.cpp:918-946.  accel != null: BeamMap::query through the
// reference's SubBeamBVH (EBVHAccel, what gvpm.cpp:880-986 builds; gvpm_oracle_accel.hpp).  Otherwise BeamMap::query's
// ENoAccel loop (pm/beams.h:289-294); subBeamSize > 0 additionally cuts every beam into sub-beams
// of that length and calls the functor per sub-beam (the ownership rule of the reference's SubBeamBVH).
template <typename F>
inline void gatherSetBeams(const GatherContext<F> &ctx, const BeamMapO<F> &map, F radius, const gvpm_camera_ray *set,
                           F subBeamSize, F *iter, Counters &cnt, const SubBeamBVHO<F, Beam<F>> *accel = nullptr) {
  CamRay<F> base(set[0]);
  CamRay<F> shifts[4] = {CamRay<F>(set[1]), CamRay<F>(set[2]), CamRay<F>(set[3]), CamRay<F>(set[4])};
  // Ray ray(vertex(idEdge), d, Epsilon, distTotal - Epsilon), gvpm.cpp:932
  Ray<F> ray(base.o, base.d, ctx.Epsilon, base.len - ctx.Epsilon);
  BeamGradRadianceQuery<F> q(ctx, &base, shifts, ray);
  q.ctx_radius = radius;
  if (accel) {
    accel->query(map.beams, ray, q);
  } else for (const Beam<F> &b : map.beams) {
    if (subBeamSize > 0) {
      int nb = (int)std::ceil(b.length / subBeamSize);
      F ls = b.length / nb;
      for (int i = 0; i < nb; ++i) q(b, ls * i, ls * (i + 1));
    } else {
      q(b, (F)0, std::numeric_limits<F>::infinity());
    }
  }
  for (int c = 0; c < 3; ++c) {
    iter[c] += q.mediumFlux[c];
    for (int k = 0; k < 4; ++k) {
      iter[3 + 3 * k + c] += q.shiftedMediumFlux[k][c];
      iter[15 + 3 * k + c] += q.weightedMediumFlux[k][c];
    }
  }
  cnt.add(q.cnt);
}

}  // namespace oracle
