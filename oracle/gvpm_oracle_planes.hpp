// ORACLE -- TEST INFRASTRUCTURE ONLY (see gvpm_oracle.hpp).
//
// G-Planes (0D kernel, experimental in the reference: no visibility, README.md:29-31): CPU restatement of
//   GPMIntegrator::computeVolumeGradientPlanes      gvpm/gvpm.cpp:782-878
//   LTPhotonPlane (flux as LTPhotonBeam)            gvpm/gvpm_plane.h:18-47
//   PhotonPlane::intersectPlane0D / getContrib0D    pm/plane_struct.h:104-192, invJacobian :199
//   PlaneGradRadianceQuery::operator()              gvpm/shift/shift_volume_planes.h:57-101
//   specularShift / intersection                    shift_volume_planes.h:263-416, 426-453
// The plane BVH (pm/plane_accel.h:85-207) is restated in gvpm_oracle_accel.hpp (PhotonPlaneBVHO); it only prunes: the
// functor performs the complete intersection test itself, so a loop over all planes visits the same hits
// (tests/test_oracle_accel.py).
// LTPhotonPlane::transformBeam (gvpm_plane.h:53-73; second distance + phase direction drawn from
// block 0's sampler, gvpm.cpp:793-797) is host-side: the planes arrive with w1 / length1.
#pragma once

#include "gvpm_oracle.hpp"
#include "gvpm_oracle_accel.hpp"

namespace oracle {

template <typename F> struct Plane {
  Vec3<F> ori, w0, w1, flux;
  F length0, length1;
  int edgeID;
};

template <typename F> struct PlaneMapO {
  std::vector<Plane<F>> planes;
  // beams: photon SoA re-read as photon beams (see gvpm_upload_beams); w1: 3 floats, len1: 1 float per plane
  void load(const gvpm_photon_soa &s, const float *w1, const float *len1) {
    planes.resize(s.n);
    for (uint64_t i = 0; i < s.n; ++i) {
      Plane<F> &p = planes[i];
      p.ori = Vec3<F>(s.parent_pos + 3 * i);
      // p->edge(i)->d and p->edge(i)->length, gvpm_plane.h:30-31
      Vec3<F> d = Vec3<F>(s.pos + 3 * i) - p.ori;
      p.length0 = d.length();
      p.w0 = d / p.length0;
      p.w1 = Vec3<F>(w1 + 3 * i);
      p.length1 = (F)len1[i];
      p.flux = Vec3<F>(s.flux + 3 * i);
      p.edgeID = (int)GVPM_PF_DEPTH(s.flags[i]);
    }
  }
};

template <typename F> struct PlaneIts {
  F tCam, t0, t1, invDet;
};

// PhotonPlane::intersectPlane0D, pm/plane_struct.h:104-135 (float det as written)
template <typename F> inline bool intersectPlane0D(const Plane<F> &pl, const Ray<F> &ray_, PlaneIts<F> &r) {
  typedef Vec3<F> V;
  V e0 = pl.w0 * pl.length0;
  V e1 = pl.w1 * pl.length1;
  V P = cross(ray_.d, e1);
  float det = (float)dot(e0, P);
  if (std::abs(det) < 1e-5f) return false;
  r.invDet = 1.0f / det;
  V T = ray_.o - pl.ori;
  r.t0 = dot(T, P) * r.invDet;
  if (r.t0 < 0.0f || r.t0 > 1.0f) return false;
  V Q = cross(T, e0);
  r.t1 = dot(ray_.d, Q) * r.invDet;
  if (r.t1 < 0.0f || r.t1 > 1.0f) return false;
  r.tCam = dot(e1, Q) * r.invDet;
  if (r.tCam <= ray_.mint || r.tCam >= ray_.maxt) return false;
  r.t1 *= pl.length1;
  r.t0 *= pl.length0;
  return true;
}

// intersection(), shift_volume_planes.h:426-453
template <typename F>
inline bool intersectionUnit(const Ray<F> &ray_, const Vec3<F> &ori, const Vec3<F> &w0, const Vec3<F> &w1, F &tCam, F &t0,
                             F &t1, F &invDet) {
  typedef Vec3<F> V;
  V P = cross(ray_.d, w1);
  F det = dot(w0, P);
  if (std::abs(det) < 1e-8f) return false;
  invDet = 1.0f / det;
  V T = ray_.o - ori;
  t0 = dot(T, P) * invDet;
  if (t0 < 0.0f) return false;
  V Q = cross(T, w0);
  t1 = dot(ray_.d, Q) * invDet;
  if (t1 < 0.0f) return false;
  tCam = dot(w1, Q) * invDet;
  return !(tCam <= ray_.mint || tCam >= ray_.maxt);
}

template <typename F> struct PlaneGradRadianceQuery {
  typedef Vec3<F> V;
  const GatherContext<F> &ctx;
  const CamRay<F> *baseGather;
  const CamRay<F> *shiftGPs;
  Ray<F> baseCameraRay;
  int currCameraEdge;
  Counters cnt;
  V mediumFlux, shiftedMediumFlux[4], weightedMediumFlux[4];

  PlaneGradRadianceQuery(const GatherContext<F> &c, const CamRay<F> *base, const CamRay<F> *shifts, const Ray<F> &ray)
      : ctx(c), baseGather(base), shiftGPs(shifts), baseCameraRay(ray), currCameraEdge(base->edge) {
    for (int i = 0; i < 4; ++i) shiftedMediumFlux[i] = weightedMediumFlux[i] = V((F)0);
    mediumFlux = V((F)0);
  }

  static F invJacobian(const Plane<F> &pl, const V &k) { return (F)(1.0 / std::abs(dot(pl.w0, cross(pl.w1, k)))); }

  // getContrib0D, pm/plane_struct.h:150-192
  V getContrib0D(const Plane<F> &pl, const PlaneIts<F> &rec, const MRec<F> &mRecCamera, const V &d) const {
    F phaseTerm = ctx.medium.phase(-pl.w1, -d);
    MRec<F> mRec0, mRec1;
    ctx.medium.eval(Ray<F>(pl.ori, pl.w0, (F)0, rec.t0), mRec0);
    ctx.medium.eval(Ray<F>(pl.ori, pl.w1, (F)0, rec.t1), mRec1);
    V contrib = mRecCamera.transmittance * (mRecCamera.sigmaS) * (mRec0.sigmaS) * pl.flux * phaseTerm;
    contrib *= mRec1.transmittance * mRec0.transmittance;
    contrib /= mRec0.pdfFailure;
    contrib /= mRec1.pdfFailure;
    contrib *= invJacobian(pl, d);
    return contrib;
  }

  // specularShift (BETTERSHIFT 0), shift_volume_planes.h:263-416
  bool specularShift(const CamRay<F> &shiftGP, const PlaneIts<F> &bRec, const Plane<F> &pl, const V &baseContrib,
                     GradientSamplingResult<F> &result) {
    Ray<F> shiftRay(shiftGP.o, shiftGP.d, ctx.Epsilon, shiftGP.len);
    V newIntersection = shiftRay(bRec.tCam);
    V orthNewW1 = newIntersection - (pl.ori + pl.w0 * dot(newIntersection - pl.ori, pl.w0));
    orthNewW1 /= orthNewW1.length();
    F w0Dot = dot(pl.w0, pl.w1);
    V newW1 = orthNewW1 * std::sqrt(1 - (w0Dot * w0Dot)) + pl.w0 * w0Dot;
    F t0New, t1New, tCamNew, invDetNew;
    if (!intersectionUnit(shiftRay, pl.ori, pl.w0, newW1, tCamNew, t0New, t1New, invDetNew)) {
      result.weight = 1.0;
      cnt.failedShifts++;
      return false;
    }
    MRec<F> mRec1, mRec0, mRec1Shift, mRec0Shift;
    ctx.medium.eval(Ray<F>(pl.ori, pl.w1, (F)0, bRec.t1), mRec1);
    ctx.medium.eval(Ray<F>(pl.ori, pl.w0, (F)0, bRec.t0), mRec0);
    ctx.medium.eval(Ray<F>(pl.ori, newW1, (F)0, t1New), mRec1Shift);
    ctx.medium.eval(Ray<F>(pl.ori, pl.w0, (F)0, t0New), mRec0Shift);
    V throughputShift = baseContrib;
    throughputShift *= mRec0Shift.transmittance * V((F)1 / mRec0.transmittance.x, (F)1 / mRec0.transmittance.y, (F)1 / mRec0.transmittance.z);
    throughputShift *= mRec1Shift.transmittance * V((F)1 / mRec1.transmittance.x, (F)1 / mRec1.transmittance.y, (F)1 / mRec1.transmittance.z);
    throughputShift /= invJacobian(pl, baseCameraRay.d);
    throughputShift *= (F)(1.0 / std::abs(dot(pl.w0, cross(newW1, shiftRay.d))));
    result.jacobian = invJacobian(pl, baseCameraRay.d);
    result.jacobian *= std::abs(dot(pl.w0, cross(newW1, shiftRay.d)));
    result.jacobian /= t1New / bRec.t1;
    if (pl.edgeID != 1) result.jacobian /= t0New / bRec.t0;
    const F pBase = ctx.medium.phase(-pl.w1, -baseCameraRay.d);
    const F pNew = ctx.medium.phase(-newW1, -shiftRay.d);
    throughputShift *= pNew;
    throughputShift /= pBase;
    result.weight = 0.5f;
    result.shiftedFlux = throughputShift * result.jacobian;
    cnt.diffuseShifts++;
    if (ctx.cfg.use_mis) {
      F basePdf = mRec0.pdfSuccess;
      basePdf *= mRec1.pdfSuccess;
      basePdf *= pBase;  // PhaseFunction::pdf == eval
      F offsetPdf = mRec0Shift.pdfSuccess;
      offsetPdf *= mRec1Shift.pdfSuccess;
      offsetPdf *= pNew;
      if (offsetPdf == (F)0 || basePdf == (F)0) {
        result.weight = 1.0f;
        return false;
      }
      const F sensorPart = sensorMIS(shiftGP, *baseGather, currCameraEdge, bRec.tCam, bRec.tCam);
      result.weight = 1.0f / (1.0f + sensorPart * result.jacobian * offsetPdf / basePdf);
    }
    return true;
  }

  // operator(), shift_volume_planes.h:57-101 -- note: no depth / path-set / interaction filters and
  // no border rule in the plane functor
  bool operator()(const Plane<F> &pl) {
    cnt.candidates++;
    PlaneIts<F> bRec;
    if (!intersectPlane0D(pl, baseCameraRay, bRec)) return false;
    MRec<F> mRecCam;
    ctx.medium.eval(Ray<F>(baseCameraRay.o, baseCameraRay.d, (F)0, bRec.tCam), mRecCam);
    V baseContrib = getContrib0D(pl, bRec, mRecCam, baseCameraRay.d);
    mediumFlux += baseContrib;
    cnt.evaluations++;
    for (int i = 0; i < 4; ++i) {
      GradientSamplingResult<F> result;
      if (shiftGPs[i].valid) specularShift(shiftGPs[i], bRec, pl, baseContrib, result);
      shiftedMediumFlux[i] += result.shiftedFlux * result.weight;
      weightedMediumFlux[i] += baseContrib * result.weight;
    }
    return false;
  }
};

// one beam set of computeVolumeGradientPlanes' inner loop, gvpm.cpp:821-846
template <typename F>
inline void gatherSetPlanes(const GatherContext<F> &ctx, const PlaneMapO<F> &map, const gvpm_camera_ray *set, F *iter,
                            Counters &cnt, const PhotonPlaneBVHO<F, Plane<F>> *accel = nullptr) {
  CamRay<F> base(set[0]);
  CamRay<F> shifts[4] = {CamRay<F>(set[1]), CamRay<F>(set[2]), CamRay<F>(set[3]), CamRay<F>(set[4])};
  Ray<F> ray(base.o, base.d, ctx.Epsilon, base.len - ctx.Epsilon);
  PlaneGradRadianceQuery<F> q(ctx, &base, shifts, ray);
  if (accel) accel->query(map.planes, ray, q);  // m_planesAccel->query(query), gvpm.cpp:836
  else for (const Plane<F> &p : map.planes) q(p);
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
