// ORACLE -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the reference's photon-gather + gradient-domain shift
// path (gradientpm/gvpm, Mitsuba 0.5 fork), templated on the reference's
// `Float` (float = SCons SINGLE_PRECISION build, double = CMake default).
// Every function cites the reference file:line it follows; paths are relative
// to the reference tree, gvpm/ = src/integrators/photonmapper/gvpm/.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// build, link or call this code.  The product (gvpm_amd/, libgvpm_hip.so)
// never includes or links it.
//
// PARITY UNPINNED: the reference holds no golden vector, known-answer test or
// fixture for this path (`grep -rl "gvpm\|photon" src/tests
// data/tests` is empty) and the Mitsuba-based reference cannot be compiled in
// this image (needs Boost, Eigen, Xerces, OpenEXR).  The oracle is anchored
// instead on (a) the literal statement-by-statement correspondence cited
// below and (b) the analytic invariants of SURVEY 8c (tests/test_oracle_*.py).
//
// The reference works on `Path*`; the oracle works on the flattened records
// of include/gvpm_hip.h (the host-side flattening is documented there).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "../include/gvpm_hip.h"

namespace oracle {

template <typename F> struct Vec3 {
  F x, y, z;
  Vec3() : x(0), y(0), z(0) {}
  Vec3(F a, F b, F c) : x(a), y(b), z(c) {}
  explicit Vec3(F a) : x(a), y(a), z(a) {}
  explicit Vec3(const float *p) : x((F)p[0]), y((F)p[1]), z((F)p[2]) {}
  F operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
  F &at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
  Vec3 operator+(const Vec3 &b) const { return Vec3(x + b.x, y + b.y, z + b.z); }
  Vec3 operator-(const Vec3 &b) const { return Vec3(x - b.x, y - b.y, z - b.z); }
  Vec3 operator-() const { return Vec3(-x, -y, -z); }
  Vec3 operator*(F s) const { return Vec3(x * s, y * s, z * s); }
  Vec3 operator*(const Vec3 &b) const { return Vec3(x * b.x, y * b.y, z * b.z); }
  Vec3 operator/(F s) const { return Vec3(x / s, y / s, z / s); }
  Vec3 &operator+=(const Vec3 &b) { x += b.x; y += b.y; z += b.z; return *this; }
  Vec3 &operator*=(const Vec3 &b) { x *= b.x; y *= b.y; z *= b.z; return *this; }
  Vec3 &operator*=(F s) { x *= s; y *= s; z *= s; return *this; }
  Vec3 &operator/=(F s) { x /= s; y /= s; z /= s; return *this; }
  F lengthSquared() const { return x * x + y * y + z * z; }
  F length() const { return std::sqrt(lengthSquared()); }
  F max() const { return std::max(x, std::max(y, z)); }
};
template <typename F> inline F dot(const Vec3<F> &a, const Vec3<F> &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename F> inline Vec3<F> cross(const Vec3<F> &a, const Vec3<F> &b) {
  return Vec3<F>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
template <typename F> inline Vec3<F> normalize(const Vec3<F> &a) { return a / a.length(); }

// math::safe_sqrt, include/mitsuba/core/math.h
template <typename F> inline F safe_sqrt(F v) { return std::sqrt(std::max((F)0, v)); }

// Ray, include/mitsuba/core/ray.h (o, d, dRcp, mint, maxt)
template <typename F> struct Ray {
  Vec3<F> o, d, dRcp;
  F mint, maxt;
  Ray() : mint(0), maxt(0) {}
  Ray(const Vec3<F> &o_, const Vec3<F> &d_, F mint_, F maxt_) : o(o_), d(d_), mint(mint_), maxt(maxt_) {
    dRcp = Vec3<F>((F)1 / d.x, (F)1 / d.y, (F)1 / d.z);
  }
  Vec3<F> operator()(F t) const { return o + d * t; }
};

// AABB::rayIntersect, include/mitsuba/core/aabb.h:310-340
template <typename F> struct AABB {
  Vec3<F> min, max;
  AABB()
      : min(std::numeric_limits<F>::infinity()), max(-std::numeric_limits<F>::infinity()) {}
  AABB(const Vec3<F> &a, const Vec3<F> &b) : min(a), max(b) {}
  void expandBy(const AABB &o) {
    for (int i = 0; i < 3; ++i) {
      min.at(i) = std::min(min[i], o.min[i]);
      max.at(i) = std::max(max[i], o.max[i]);
    }
  }
  void expandBy(const Vec3<F> &p) {
    for (int i = 0; i < 3; ++i) {
      min.at(i) = std::min(min[i], p[i]);
      max.at(i) = std::max(max[i], p[i]);
    }
  }
  int getLargestAxis() const {  // aabb.h getLargestAxis
    Vec3<F> d = max - min;
    int largest = 0;
    for (int i = 1; i < 3; ++i)
      if (d[i] > d[largest]) largest = i;
    return largest;
  }
  bool rayIntersect(const Ray<F> &ray, F &nearT, F &farT) const {
    nearT = -std::numeric_limits<F>::infinity();
    farT = std::numeric_limits<F>::infinity();
    for (int i = 0; i < 3; i++) {
      const F origin = ray.o[i];
      const F minVal = min[i], maxVal = max[i];
      if (ray.d[i] == 0) {
        if (origin < minVal || origin > maxVal) return false;
      } else {
        F t1 = (minVal - origin) * ray.dRcp[i];
        F t2 = (maxVal - origin) * ray.dRcp[i];
        if (t1 > t2) std::swap(t1, t2);
        nearT = std::max(t1, nearT);
        farT = std::min(t2, farT);
        if (!(nearT <= farT)) return false;
      }
    }
    return true;
  }
};

// MediumSamplingRecord subset
template <typename F> struct MRec {
  Vec3<F> transmittance, sigmaS;
  F pdfSuccess, pdfFailure, t;
};

// HomogeneousMedium (balance strategy), src/medium/homogeneous.cpp
template <typename F> struct Medium {
  Vec3<F> sigmaA, sigmaS, sigmaT;
  F g, mediumSamplingWeight;

  // eval(), homogeneous.cpp:432-513.  alwaysValid = EDistanceAlwaysValid (uses mRec.t)
  void eval(const Ray<F> &ray, MRec<F> &mRec, bool alwaysValid = false) const {
    F currentMediumSampling = mediumSamplingWeight;
    if (alwaysValid && mediumSamplingWeight != (F)1) currentMediumSampling = (F)1;
    F distance = ray.maxt - ray.mint;
    mRec.pdfSuccess = 0;
    mRec.pdfFailure = 0;
    if (alwaysValid) {
      const F maxDist = ray.maxt - ray.mint;
      distance = mRec.t;
      for (int i = 0; i < 3; ++i) {
        const F normalization = 1 - std::exp(-sigmaT[i] * maxDist);
        F tmp = std::exp(-sigmaT[i] * distance);
        mRec.pdfFailure = 0;
        mRec.pdfSuccess += (sigmaT[i] / normalization) * tmp;
      }
    } else {
      for (int i = 0; i < 3; ++i) {
        F tmp = std::exp(-sigmaT[i] * distance);
        mRec.pdfFailure += tmp;
        mRec.pdfSuccess += sigmaT[i] * tmp;
      }
    }
    mRec.pdfSuccess /= 3;
    mRec.pdfFailure /= 3;
    mRec.transmittance = Vec3<F>(std::exp(sigmaT.x * (-distance)), std::exp(sigmaT.y * (-distance)),
                                 std::exp(sigmaT.z * (-distance)));
    mRec.pdfSuccess = mRec.pdfSuccess * currentMediumSampling;
    mRec.pdfFailure = mRec.pdfFailure * currentMediumSampling + (1 - currentMediumSampling);
    mRec.sigmaS = sigmaS;
    if (mRec.transmittance.max() < (F)1e-20) mRec.transmittance = Vec3<F>((F)0);
  }

  // sampleDistance(ray, mRec, sampler, EDistanceAlwaysValid, rand), homogeneous.cpp:293-430
  bool sampleDistanceAlwaysValid(const Ray<F> &ray, MRec<F> &mRec, F rand, F Epsilon) const {
    F samplingDensity = sigmaT[1];  // EBalance: channel min(int(0.5*3), 2) = 1
    F currentMediumSampling = (F)1;
    F sampledDistance;
    if (rand < currentMediumSampling) {
      rand /= currentMediumSampling;
      const F maxDist = std::max((ray.maxt - ray.mint) - Epsilon, (F)0.0);
      const F normalization = 1 - std::exp(-samplingDensity * maxDist);
      sampledDistance = -std::log(1 - rand * normalization) / samplingDensity;
    } else {
      sampledDistance = std::numeric_limits<F>::infinity();
    }
    F distSurf = ray.maxt - ray.mint;
    bool success = true;
    if (sampledDistance < distSurf) {
      mRec.t = sampledDistance + ray.mint;
      Vec3<F> p = ray(mRec.t);
      if (p.x == ray.o.x && p.y == ray.o.y && p.z == ray.o.z) success = false;
    } else {
      sampledDistance = distSurf;
      success = false;
    }
    mRec.pdfFailure = 0;
    mRec.pdfSuccess = 0;
    {
      const F maxDist = ray.maxt - ray.mint;
      for (int i = 0; i < 3; ++i) {
        const F normalization = 1 - std::exp(-sigmaT[i] * maxDist);
        F tmp = std::exp(-sigmaT[i] * sampledDistance);
        mRec.pdfFailure = 0;
        mRec.pdfSuccess += (sigmaT[i] / normalization) * tmp;
      }
    }
    mRec.pdfFailure /= 3;
    mRec.pdfSuccess /= 3;
    mRec.transmittance = Vec3<F>(std::exp(sigmaT.x * (-sampledDistance)), std::exp(sigmaT.y * (-sampledDistance)),
                                 std::exp(sigmaT.z * (-sampledDistance)));
    mRec.pdfSuccess = mRec.pdfSuccess * currentMediumSampling;
    mRec.pdfFailure = currentMediumSampling * mRec.pdfFailure + (1 - currentMediumSampling);
    mRec.sigmaS = sigmaS;
    if (mRec.transmittance.max() < (F)1e-20) mRec.transmittance = Vec3<F>((F)0);
    return success;
  }

  // IsotropicPhaseFunction::eval (src/phase/isotropic.cpp:76-78) /
  // HGPhaseFunction::eval (src/phase/hg.cpp:107-110); g == 0 -> isotropic plugin
  static F phaseEval(F g_, const Vec3<F> &wi, const Vec3<F> &wo) {
    const F INV_FOURPI = (F)0.07957747154594766788;
    if (g_ == (F)0) return INV_FOURPI;
    F temp = (F)1 + g_ * g_ + (F)2 * g_ * dot(wi, wo);
    return INV_FOURPI * (1 - g_ * g_) / (temp * std::sqrt(temp));
  }
  F phase(const Vec3<F> &wi, const Vec3<F> &wo) const { return phaseEval(g, wi, wo); }
};

// SVertexPDF cache + edge geometry of one camera edge (gvpm_struct.h:361-370)
template <typename F> struct CamRay {
  Vec3<F> o, d, eye;
  F len, pdf, jacobian, gop;
  bool valid;
  int edge;
  F rand;
  int px, py;
  explicit CamRay(const gvpm_camera_ray &r)
      : o(r.o), d(r.d), eye(r.eye), len((F)r.len), pdf((F)r.pdf), jacobian((F)r.jacobian), gop((F)r.gop),
        valid(GVPM_RAY_VALID(r.info) != 0), edge((int)GVPM_RAY_EDGE(r.info)), rand((F)r.rand),
        px((int)(r.pixel & 0xFFFFu)), py((int)(r.pixel >> 16)) {}
  CamRay() : len(0), pdf(0), jacobian(0), gop(0), valid(false), edge(0), rand(0), px(0), py(0) {}
};

// GatherPoint::sensorMIS, gvpm/gvpm_struct.h:608-631 (literal, incl. the
// cancelling G/distance factors)
template <typename F>
inline F sensorMIS(const CamRay<F> &shift, const CamRay<F> &base, int idVertex, F sDist, F bDist) {
  F jacobian = shift.jacobian;
  F ratio = shift.pdf / base.pdf;
  if (idVertex != 1) {
    F baseG = base.gop;
    F currG = shift.gop;
    jacobian *= currG / baseG;
    ratio *= baseG / currG;
    jacobian *= (sDist / bDist) * (sDist / bDist);
    ratio *= (bDist / sDist) * (bDist / sDist);
  }
  return ratio * jacobian;
}

template <typename F> struct Photon {
  Vec3<F> pos, wi, flux, parentPos, parentN, prefixW, parentScat, parentWi;
  F parentPdf, edgePdf, parentRR, parentG;
  uint32_t flags, pathID;
};

// GradientSamplingResult / ShiftRecord, gvpm/shift/shift_utilities.h:16-33
template <typename F> struct GradientSamplingResult {
  Vec3<F> shiftedFlux;
  F weight, jacobian;
  GradientSamplingResult() : shiftedFlux((F)0), weight((F)1), jacobian((F)1) {}
};
template <typename F> struct ShiftRecord {
  Vec3<F> throughtput;
  F pdf, jacobian;
  ShiftRecord() : throughtput((F)1), pdf((F)0), jacobian((F)1) {}
};

struct Counters {
  uint64_t evaluations = 0, candidates = 0, nullShifts = 0, diffuseShifts = 0, failedShifts = 0;
  void add(const Counters &o) {
    evaluations += o.evaluations; candidates += o.candidates; nullShifts += o.nullShifts;
    diffuseShifts += o.diffuseShifts; failedShifts += o.failedShifts;
  }
};

template <typename F> struct Scene {
  std::vector<Vec3<F>> v0, e1, e2;
  // scene->rayIntersect(ray): any-hit; Triangle::rayIntersect
  // (include/mitsuba/core/triangle.h:109-145) + interval test
  // (include/mitsuba/render/skdtree.h:318-320)
  bool rayIntersect(const Ray<F> &ray) const {
    for (size_t i = 0; i < v0.size(); ++i) {
      Vec3<F> pvec = cross(ray.d, e2[i]);
      F det = dot(e1[i], pvec);
      if (det == 0) continue;
      F inv_det = (F)1.0 / det;
      Vec3<F> tvec = ray.o - v0[i];
      F u = dot(tvec, pvec) * inv_det;
      if (u < 0.0 || u > 1.0) continue;
      Vec3<F> qvec = cross(tvec, e1[i]);
      F v = dot(ray.d, qvec) * inv_det;
      if (v >= 0.0 && u + v <= 1.0) {
        F t = dot(e2[i], qvec) * inv_det;
        if (t >= ray.mint && t <= ray.maxt) return true;
      }
    }
    return false;
  }
};

// coordinateSystem / coordinateSystemCoherent, src/libcore/util.cpp:592-609 (note the
// float intermediates of the coherent variant)
template <typename F> inline void coordinateSystem(const Vec3<F> &a, Vec3<F> &b, Vec3<F> &c) {
  if (std::abs(a.x) > std::abs(a.y)) {
    F invLen = (F)1.0 / std::sqrt(a.x * a.x + a.z * a.z);
    c = Vec3<F>(a.z * invLen, 0.0f, -a.x * invLen);
  } else {
    F invLen = (F)1.0 / std::sqrt(a.y * a.y + a.z * a.z);
    c = Vec3<F>(0.0f, a.z * invLen, -a.y * invLen);
  }
  b = cross(c, a);
}
template <typename F> inline void coordinateSystemCoherent(const Vec3<F> &n, Vec3<F> &b1, Vec3<F> &b2) {
  float sign = copysignf(1.0f, (float)n.z);
  const float a = (float)(-1.0f / (sign + n.z));
  const float b = (float)(n.x * n.y * a);
  b1 = Vec3<F>(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
  b2 = Vec3<F>(b, sign + n.y * n.y * a, -n.y);
}

// ---------------------------------------------------------------------------
// The per-photon functors.  AbstractVolumeGradientRecord + VolumeGradientBREQuery
// (gvpm/shift/shift_volume_photon.{h,cpp}).
// ---------------------------------------------------------------------------
// The BSDF table of the scene's glossy surfaces (include/gvpm_hip.h, gvpm_upload_bsdfs), set by oracle_set_bsdfs before a
// gather (test infrastructure: one table per process).
inline std::vector<gvpm_bsdf> &bsdfTable() {
  static std::vector<gvpm_bsdf> t;
  return t;
}

// Phong::eval / Phong::pdf * pdfComponent, src/bsdfs/phong.cpp:121-186,331-342, for bRec.component = -1 (both lobes,
// pdfComponent = 1) or one sampled component (round 5), in the LOCAL frame of the intersection as the reference evaluates them: Frame(n) with
// coordinateSystem(n) (frame.h:37-80; the value does not depend on the tangents).  wi, wo: world-space unit vectors.
template <typename F>
inline bool phongEvalPdf(const gvpm_bsdf &b, const Vec3<F> &kd, const Vec3<F> &n, const Vec3<F> &wiW, const Vec3<F> &woW,
                         Vec3<F> &f, F &pdf) {
  typedef Vec3<F> V;
  const F INV_PI = (F)0.31830988618379067154, INV_TWOPI = (F)0.15915494309189533577, M_PI_F = (F)3.14159265358979323846;
  V s, t;
  coordinateSystem(n, s, t);
  const V wi(dot(wiW, s), dot(wiW, t), dot(wiW, n)), wo(dot(woW, s), dot(woW, t), dot(woW, n));  // its.toLocal
  f = V((F)0);
  pdf = 0;
  if (wi.z <= 0 || wo.z <= 0) return false;   // Frame::cosTheta(bRec.wi) <= 0 || Frame::cosTheta(bRec.wo) <= 0
  const V refl(-wi.x, -wi.y, wi.z);            // reflect(wi), :117-119
  const F alpha = dot(wo, refl), exponent = (F)b.exponent;
  // bRec.component = the entry's component (b.distribution - 1: -1 both, 0 specular, 1 diffuse; :131-134,161-164)
  const int component = b.distribution - 1;
  const bool hasSpecular = component == -1 || component == 0, hasDiffuse = component == -1 || component == 1;
  V result((F)0);
  F specProb = 0, diffuseProb = 0;
  if (hasSpecular && alpha > 0) {
    result += V((F)b.specular[0], (F)b.specular[1], (F)b.specular[2]) * ((exponent + 2) * INV_TWOPI * std::pow(alpha, exponent));
    specProb = std::pow(alpha, exponent) * (exponent + (F)1) / ((F)2 * M_PI_F);
  }
  if (hasDiffuse) {
    result += kd * INV_PI;
    diffuseProb = INV_PI * wo.z;               // warp::squareToCosineHemispherePdf
  }
  f = result * wo.z;
  const F w = (F)b.specular_sampling_weight;
  if (hasDiffuse && hasSpecular) pdf = w * specProb + (1 - w) * diffuseProb;
  else if (hasDiffuse) pdf = diffuseProb * (1 - w);   // Phong::pdf * pdfComponent (shift_diffuse.cpp:43-44, phong.cpp:331-342)
  else pdf = specProb * w;
  return true;
}

// Ward::eval / Ward::pdf, src/bsdfs/ward.cpp:178-266, both components (bRec.component = -1: roughness >= 0.05, :370-389),
// alphaU == alphaV = b.exponent, m_modelVariant = b.sample_visible (EWard 0, EWardDuer 1, EBalanced 2), in the LOCAL frame of
// the intersection as the reference evaluates them (the values do not depend on the tangents for an isotropic alpha).
template <typename F>
inline bool wardEvalPdf(const gvpm_bsdf &b, const Vec3<F> &kd, const Vec3<F> &n, const Vec3<F> &wiW, const Vec3<F> &woW,
                        Vec3<F> &f, F &pdf) {
  typedef Vec3<F> V;
  const F INV_PI = (F)0.31830988618379067154, M_PI_F = (F)3.14159265358979323846;
  V s, t;
  coordinateSystem(n, s, t);
  const V wi(dot(wiW, s), dot(wiW, t), dot(wiW, n)), wo(dot(woW, s), dot(woW, t), dot(woW, n));
  f = V((F)0);
  pdf = 0;
  if (wi.z <= 0 || wo.z <= 0) return false;
  const F alphaU = (F)b.exponent, alphaV = (F)b.exponent;
  V result((F)0);
  {
    const V H = wi + wo;
    F factor1 = 0;
    switch (b.sample_visible) {
      case GVPM_WARD_WARD: factor1 = (F)1 / ((F)4 * M_PI_F * alphaU * alphaV * std::sqrt(wi.z * wo.z)); break;
      case GVPM_WARD_DUER: factor1 = (F)1 / ((F)4 * M_PI_F * alphaU * alphaV * wi.z * wo.z); break;
      default: factor1 = dot(H, H) / (M_PI_F * alphaU * alphaV * std::pow(H.z, (F)4)); break;
    }
    const F factor2 = H.x / alphaU, factor3 = H.y / alphaV;
    const F exponent = -(factor2 * factor2 + factor3 * factor3) / (H.z * H.z);
    const F specRef = factor1 * std::exp(exponent);
    if (specRef > (F)1e-10) result += V((F)b.specular[0], (F)b.specular[1], (F)b.specular[2]) * specRef;
  }
  result += kd * INV_PI;
  f = result * wo.z;
  F specProb;
  {
    V H = wi + wo;
    H = H / H.length();
    const F factor1 = (F)1 / ((F)4 * M_PI_F * alphaU * alphaV * dot(H, wi) * std::pow(H.z, (F)3));
    const F factor2 = H.x / alphaU, factor3 = H.y / alphaV;
    const F exponent = -(factor2 * factor2 + factor3 * factor3) / (H.z * H.z);
    specProb = factor1 * std::exp(exponent);
  }
  const F diffuseProb = INV_PI * wo.z;
  const F w = (F)b.specular_sampling_weight;
  pdf = w * specProb + (1 - w) * diffuseProb;
  return true;
}

// RoughConductor::eval / pdf, src/bsdfs/roughconductor.cpp:257-319 (one component: bRec.component -1 or 0), with
// MicrofacetDistribution (src/bsdfs/microfacet.h: eval :191-232, pdf :270-275 -> pdfAll = eval * cosTheta / pdfVisible,
// smithG1 :477-518, G :520-522, projectRoughness = alphaU for an isotropic distribution :541-546) and fresnelConductorExact
// (src/libcore/util.cpp:747-769), in the local frame of the intersection.  Beckmann or GGX, alphaU == alphaV = b.exponent.
template <typename F> inline F microfacetEval(const gvpm_bsdf &b, const Vec3<F> &m) {
  const F M_PI_F = (F)3.14159265358979323846, alpha = (F)b.exponent;
  if (m.z <= 0) return 0;
  const F cosTheta2 = m.z * m.z;
  const F beckmannExponent = ((m.x * m.x) / (alpha * alpha) + (m.y * m.y) / (alpha * alpha)) / cosTheta2;
  F result;
  if (b.distribution == GVPM_MICROFACET_GGX) {
    const F root = ((F)1 + beckmannExponent) * cosTheta2;
    result = (F)1 / (M_PI_F * alpha * alpha * root * root);
  } else {
    result = std::exp(-beckmannExponent) / (M_PI_F * alpha * alpha * cosTheta2 * cosTheta2);  // math::fastexp
  }
  if (result * m.z < (F)1e-20) result = 0;
  return result;
}
template <typename F> inline F microfacetSmithG1(const gvpm_bsdf &b, const Vec3<F> &v, const Vec3<F> &m) {
  if (dot(v, m) * v.z <= 0) return 0;
  const F temp = 1 - v.z * v.z;                       // Frame::tanTheta(v), frame.h
  const F tanTheta = temp <= 0 ? (F)0 : std::abs(std::sqrt(temp) / v.z);
  if (tanTheta == 0) return 1;
  const F alpha = (F)b.exponent;                      // projectRoughness, isotropic
  if (b.distribution == GVPM_MICROFACET_GGX) {
    const F root = alpha * tanTheta;
    return (F)2 / ((F)1 + std::sqrt((F)1 + root * root));  // math::hypot2(1, root)
  }
  const F a = (F)1 / (alpha * tanTheta);
  if (a >= (F)1.6) return 1;
  const F aSqr = a * a;
  return ((F)3.535 * a + (F)2.181 * aSqr) / ((F)1 + (F)2.276 * a + (F)2.577 * aSqr);
}
template <typename F> inline F fresnelConductorExact1(F cosThetaI, F eta, F k) {
  const F cosThetaI2 = cosThetaI * cosThetaI, sinThetaI2 = 1 - cosThetaI2, sinThetaI4 = sinThetaI2 * sinThetaI2;
  const F temp1 = eta * eta - k * k - sinThetaI2;
  const F a2pb2 = std::sqrt(std::max((F)0, temp1 * temp1 + k * k * eta * eta * 4));  // safe_sqrt
  const F a = std::sqrt(std::max((F)0, (a2pb2 + temp1) * (F)0.5));
  const F term1 = a2pb2 + cosThetaI2, term2 = a * (2 * cosThetaI);
  const F Rs2 = (term1 - term2) / (term1 + term2);
  const F term3 = a2pb2 * cosThetaI2 + sinThetaI4, term4 = term2 * sinThetaI2;
  const F Rp2 = Rs2 * (term3 - term4) / (term3 + term4);
  return (F)0.5 * (Rp2 + Rs2);
}
template <typename F>
inline bool roughConductorEvalPdf(const gvpm_bsdf &b, const Vec3<F> &n, const Vec3<F> &wiW, const Vec3<F> &woW, Vec3<F> &f, F &pdf) {
  typedef Vec3<F> V;
  V s, t;
  coordinateSystem(n, s, t);
  const V wi(dot(wiW, s), dot(wiW, t), dot(wiW, n)), wo(dot(woW, s), dot(woW, t), dot(woW, n));
  f = V((F)0);
  pdf = 0;
  if (wi.z <= 0 || wo.z <= 0) return false;
  const V H = normalize(wo + wi);
  const F D = microfacetEval<F>(b, H);
  if (D != 0) {
    const F wiH = dot(wi, H);
    const V Fr(fresnelConductorExact1<F>(wiH, (F)b.eta[0], (F)b.k[0]) * (F)b.specular[0],
               fresnelConductorExact1<F>(wiH, (F)b.eta[1], (F)b.k[1]) * (F)b.specular[1],
               fresnelConductorExact1<F>(wiH, (F)b.eta[2], (F)b.k[2]) * (F)b.specular[2]);
    const F G = microfacetSmithG1<F>(b, wi, H) * microfacetSmithG1<F>(b, wo, H);
    f = Fr * (D * G / ((F)4 * wi.z));
  }
  if (b.sample_visible) pdf = D * microfacetSmithG1<F>(b, wi, H) / ((F)4 * wi.z);
  else pdf = D * H.z / ((F)4 * std::abs(dot(wo, H)));  // pdfAll(H) / (4 absDot(wo, H))
  return true;
}
// the table entry's eval and pdf (false: a kind outside the closed set)
template <typename F>
inline bool glossyEvalPdf(const gvpm_bsdf &b, const Vec3<F> &kd, const Vec3<F> &n, const Vec3<F> &wiW, const Vec3<F> &woW, Vec3<F> &f,
                          F &pdf) {
  if (b.kind == GVPM_BSDF_PHONG) {
    phongEvalPdf<F>(b, kd, n, wiW, woW, f, pdf);
    return true;
  }
  if (b.kind == GVPM_BSDF_WARD) {
    wardEvalPdf<F>(b, kd, n, wiW, woW, f, pdf);
    return true;
  }
  if (b.kind == GVPM_BSDF_ROUGHCONDUCTOR) {
    roughConductorEvalPdf<F>(b, n, wiW, woW, f, pdf);
    return true;
  }
  return false;
}

template <typename F> struct GatherContext {
  gvpm_params cfg;
  Medium<F> medium;
  Scene<F> scene;
  F Epsilon, ShadowEpsilon;
};

template <typename F> struct VolumeGradientRecord {
  typedef Vec3<F> V;
  const GatherContext<F> &ctx;
  const CamRay<F> *baseGather;    // base ray of the set
  const CamRay<F> *shiftGPs;      // 4 shifted rays
  Ray<F> baseRay;
  int currEdge;
  Counters cnt;
  // results
  V mediumFlux, shiftedMediumFlux[4], weightedMediumFlux[4];

  VolumeGradientRecord(const GatherContext<F> &c, const CamRay<F> *base, const CamRay<F> *shifts)
      : ctx(c), baseGather(base), shiftGPs(shifts), currEdge(base->edge) {
    clear();
  }
  void clear() {
    for (int i = 0; i < 4; ++i) shiftedMediumFlux[i] = weightedMediumFlux[i] = V((F)0);
    mediumFlux = V((F)0);
  }

  // getVolumePhotonContrib, shift_volume_photon.h:78-85
  V getVolumePhotonContrib(const V &flux, const MRec<F> &mRec, const V &wi, const V &wo) const {
    return mRec.sigmaS * flux * ctx.medium.phase(wi, wo);
  }

  // computeVolumeContribution, shift_utilities.h:231-253
  bool computeVolumeContribution(const Photon<F> &ph) const {
    const int mode = ctx.cfg.lighting_interaction_mode;
    const unsigned ptype = GVPM_PF_PARENT_TYPE(ph.flags);
    if ((mode & GVPM_SURF2MEDIA) && (mode & GVPM_MEDIA2MEDIA)) {
    } else {
      if (ptype == GVPM_PARENT_MEDIUM && !(mode & GVPM_MEDIA2MEDIA)) return false;
      if ((ptype == GVPM_PARENT_SURFACE || ptype == GVPM_PARENT_EMITTER) && !(mode & GVPM_SURF2MEDIA)) return false;
    }
    const int comp = (int)GVPM_PF_PREV_COMPONENT(ph.flags);
    return !(ctx.cfg.bsdf_interaction_mode != GVPM_BSDF_ALL && comp > 0 && !(comp & ctx.cfg.bsdf_interaction_mode));
  }

  static int shiftTypeEnum(const Photon<F> &ph) {  // 3-bit code -> ELightShiftType
    switch (GVPM_PF_SHIFT_TYPE(ph.flags)) {
      case 1: return GVPM_SHIFT_DIFFUSE;
      case 2: return GVPM_SHIFT_MEDIUM;
      case 3: return GVPM_SHIFT_MANIFOLD;
      default: return GVPM_SHIFT_INVALID;
    }
  }

  // diffuseReconnection(..., isVolumeBase = true), gvpm/shift/operation/shift_diffuse.cpp:11-134
  bool diffuseReconnection(ShiftRecord<F> &sRec, const V &newD, const F newDLength, const Photon<F> &ph) const {
    const F INV_PI = (F)0.31830988618379067154;
    F pdfValue;
    const unsigned ptype = GVPM_PF_PARENT_TYPE(ph.flags);
    if (ptype == GVPM_PARENT_SURFACE) {
      // closed set: one-lobe Lambertian, shading frame == geometric frame
      F cosWo = dot(ph.parentN, newD);       // Frame::cosTheta(pWo)
      F cosWi = dot(ph.parentN, ph.parentWi); // Frame::cosTheta(pWi)
      // diffuse.cpp:110-127 eval / pdf
      if (cosWi <= 0 || cosWo <= 0) {
        sRec.throughtput *= V((F)0);
        pdfValue = 0;
      } else {
        sRec.throughtput *= ph.parentScat * (INV_PI * cosWo);
        pdfValue = INV_PI * cosWo;  // squareToCosineHemispherePdf
      }
      pdfValue *= (F)1;  // pdfComponent
      F wiDotGeoN = cosWi, woDotGeoN = cosWo;
      if (wiDotGeoN * cosWi <= 0 || woDotGeoN * cosWo <= 0) return false;
    } else if (ptype == GVPM_PARENT_SURFACE_BSDF) {
      // a glossy parent: BSDF::eval, BSDF::pdf * pdfComponent of the table's entry (shift_diffuse.cpp:25-41)
      const std::vector<gvpm_bsdf> &tab = bsdfTable();
      const size_t bi = (size_t)ph.parentG;
      V f;
      if (!(ph.parentG >= 0) || bi >= tab.size() || !glossyEvalPdf<F>(tab[bi], ph.parentScat, ph.parentN, ph.parentWi, newD, f, pdfValue)) {
        sRec.throughtput *= V((F)0);  // outside the closed set: a failed shift, as the device makes of it
        sRec.pdf = 0;
        return false;
      }
      sRec.throughtput *= f;
      pdfValue *= (F)1;  // pdfComponent, bRec.component == -1
      F cosWo = dot(ph.parentN, newD), cosWi = dot(ph.parentN, ph.parentWi);
      if (cosWi * cosWi <= 0 || cosWo * cosWo <= 0) return false;  // shading frame == geometric frame
    } else if (ptype == GVPM_PARENT_MEDIUM) {
      V pWo = newD;
      V pWi = ph.parentWi;  // normalize(predPos - pMRec.p)
      F ph_ = Medium<F>::phaseEval(ph.parentG, pWi, pWo);
      sRec.throughtput *= ph.parentScat * ph_;
      pdfValue = ph_;
    } else {
      // AreaLight::evalDirection / pdfDirection, src/emitters/area.cpp:132-150
      F dp = dot(newD, ph.parentN);
      if (dp < 0) dp = 0.0f;
      sRec.throughtput *= V(INV_PI * dp);
      pdfValue = INV_PI * dp;
    }
    {
      F GOp = 1 / (newDLength * newDLength);
      sRec.pdf = pdfValue * GOp;
      sRec.throughtput *= GOp;
    }
    if (ph.parentPdf == (F)0) {
      sRec.pdf = 0.f;
      return false;
    }
    sRec.jacobian = 1.0f;
    sRec.throughtput /= ph.parentPdf;
    sRec.throughtput *= ph.parentRR;
    if (GVPM_PF_EDGE_IN_MEDIUM(ph.flags)) {
      MRec<F> mRecShift;
      Ray<F> mRay(ph.parentPos, newD, (F)0, newDLength);
      ctx.medium.eval(mRay, mRecShift);
      sRec.pdf *= mRecShift.pdfSuccess;
      sRec.throughtput *= mRecShift.transmittance / ph.edgePdf;
    }
    return true;
  }

  // shiftPhotonDiffuse, shift_volume_photon.cpp:382-486
  bool shiftPhotonDiffuse(const V &offsetPos, const Photon<F> &ph, const CamRay<F> &shiftGP, const Ray<F> &shiftRay,
                          const MRec<F> &shiftMRec, GradientSamplingResult<F> &result, F pdfBaseRay, F pdfShiftRay,
                          F additionalJacobian) {
    V dProj = offsetPos - ph.parentPos;
    F lProj = dProj.length();
    dProj /= lProj;
    F maxt = ctx.cfg.visibility_as_written ? lProj * ctx.ShadowEpsilon : lProj * ((F)1 - ctx.ShadowEpsilon);
    Ray<F> projRay(ph.parentPos, dProj, ctx.Epsilon, maxt);
    if (ctx.scene.rayIntersect(projRay)) return false;
    if (GVPM_PF_PARENT_TYPE(ph.flags) != GVPM_PARENT_MEDIUM) {
      // source->edge(currVertex-1)->d == -photon.wi
      const F signDot = dot(ph.parentN, dProj) / dot(ph.parentN, -ph.wi);
      if (signDot < 0.f) return false;
    }
    V photonWeight = ph.prefixW;
    ShiftRecord<F> sRec;
    {
      diffuseReconnection(sRec, dProj, lProj, ph);
      if (sRec.pdf == (F)0) {
        result.weight = 1.0f;
        return false;
      }
    }
    result.jacobian *= sRec.jacobian * additionalJacobian;
    photonWeight *= sRec.throughtput;
    V contrib = getVolumePhotonContrib(photonWeight, shiftMRec, -dProj, -shiftRay.d);
    V eyeShiftContrib = shiftGP.eye;
    result.shiftedFlux = shiftMRec.transmittance * contrib * eyeShiftContrib * result.jacobian;
    result.weight = 0.5f;
    if (ctx.cfg.use_mis) {
      F basePdf = pdfBaseRay;
      basePdf *= ph.parentPdf;
      basePdf *= ph.edgePdf;
      F offsetPdf = sRec.pdf * pdfShiftRay;
      if (offsetPdf == (F)0 || basePdf == (F)0) {
        result.weight = 1.0f;
        return false;
      }
      const F sensorPart = sensorMIS(shiftGP, *baseGather, currEdge, shiftRay.maxt, baseRay.maxt);
      if (ctx.cfg.power_heuristic) {
        F v = sensorPart * result.jacobian * (offsetPdf / basePdf);
        result.weight = 1.0f / (1.0f + v * v);
      } else {
        result.weight = 1.0f / (1.0f + sensorPart * result.jacobian * (offsetPdf / basePdf));
      }
    }
    return true;
  }

  // The manifold walk itself (generateShiftPathME + ShiftME + SpecularManifold::det; shift_ME.cpp:13-142,
  // mut_manifold.cpp:1310-1410) runs over Mitsuba's Path / BSDF objects and is NOT restated: this oracle, like the device,
  // takes its results as given.  STAND-IN used by the tests of the host-shift round trip (include/gvpm_hip.h
  // gvpm_upload_host_shifts): a smooth closed-form function of the request and of the photon's parent vertex, shaped like
  // a reconnection to the parent -- NOT Mitsuba's walk.  What the tests verify with it is the content of the requests and
  // the application of shift_volume_photon.cpp:205-279 below.
  struct HostShift {
    bool ok;
    V throughput, wi;
    F pdf, detRatio, basePdf;
  };
  static HostShift standinManifoldWalk(const V &offsetPos, const V &photonPos, const V &parentPos, const V &prefixW,
                                       F parentPdf, F edgePdf) {
    HostShift r;
    V d = parentPos - offsetPos;
    const F len = d.length(), lenB = (parentPos - photonPos).length();
    r.ok = len > (F)0 && len < (F)3 * lenB;
    r.wi = len > (F)0 ? d / len : V((F)0);
    const F q = len > (F)0 ? (lenB * lenB) / (len * len) : (F)0;
    r.throughput = prefixW * (len > (F)0 ? lenB / len : (F)0);
    r.pdf = parentPdf * q;
    r.detRatio = q;
    r.basePdf = parentPdf * edgePdf;
    return r;
  }

  // A second stand-in (round 5), this one a REAL specular walk through one planar mirror -- the photon's parent, with its
  // normal -- so that a test can answer the device from a genuinely different statement of the same walk (a Newton solve on
  // the half-vector constraint, tests/test_host_shifts_gpu.py) than the one stated here (the image construction).  The
  // boundary record does not carry vertex c - 2: the walk's fixed end is the point at the photon's own distance from the
  // mirror point along the recorded direction to it, a = m + parentWi |m - x|.  The new mirror point m' is where the segment
  // from a to the mirror image of the offset position crosses the plane; lengths unfold: ratio = (|a m| + |m x|) / (|a m'| +
  // |m' x'|), throughput = prefix * ratio, determinant ratio = ratio^2 (the geometric term of the unfolded path), pdf =
  // parentPdf * ratio^2, base pdf = parentPdf * edgePdf; fails when an end lies behind the mirror or the path grows 3-fold.
  static HostShift mirrorManifoldWalk(const V &offsetPos, const V &photonPos, const V &mirrorPos, const V &mirrorN,
                                      const V &dirToSource, const V &prefixW, F parentPdf, F edgePdf) {
    HostShift r;
    const F d2 = (mirrorPos - photonPos).length();
    const V a = mirrorPos + dirToSource * d2;
    const F da = dot(mirrorN, a - mirrorPos), dx = dot(mirrorN, offsetPos - mirrorPos);
    r.ok = da > (F)0 && dx > (F)0;
    r.throughput = r.wi = V((F)0);
    r.pdf = r.detRatio = (F)0;
    r.basePdf = parentPdf * edgePdf;
    if (!r.ok) return r;
    const V image = offsetPos - mirrorN * ((F)2 * dx);
    const V mNew = a + (image - a) * (da / (da + dx));
    const F d1 = (a - mirrorPos).length(), d1n = (a - mNew).length(), d2n = (mNew - offsetPos).length();
    const F ratio = (d1 + d2) / (d1n + d2n);
    r.ok = d1n + d2n < (F)3 * (d1 + d2) && d2n > (F)0;
    r.wi = d2n > (F)0 ? (mNew - offsetPos) / d2n : V((F)0);
    r.throughput = prefixW * ratio;
    r.detRatio = ratio * ratio;
    r.pdf = parentPdf * ratio * ratio;
    return r;
  }
  // which stand-in answers (test infrastructure, one setting per process): 0 the smooth closed form, 1 the planar mirror
  static int &manifoldWalkKind() {
    static int kind = 0;
    return kind;
  }

  // shiftPhotonManifold, shift_volume_photon.cpp:160-295, from the walk's results on
  bool shiftPhotonManifold(const V &offsetPos, const Photon<F> &ph, const CamRay<F> &shiftGP, const Ray<F> &shiftRay,
                           const MRec<F> &shiftMRec, GradientSamplingResult<F> &result, F pdfBaseRay, F pdfShiftRay,
                           F additionalJacobian) {
    const HostShift hs = manifoldWalkKind() == 1
                             ? mirrorManifoldWalk(offsetPos, ph.pos, ph.parentPos, ph.parentN, ph.parentWi, ph.prefixW, ph.parentPdf, ph.edgePdf)
                             : standinManifoldWalk(offsetPos, ph.pos, ph.parentPos, ph.prefixW, ph.parentPdf, ph.edgePdf);
    if (!hs.ok) {  // generateShiftPathME / ShiftME failed (:186-203)
      result.weight = 1.0f;
      return false;
    }
    result.jacobian *= (F)1 * additionalJacobian;  // sRecME.jacobian stays 1 (shift_utilities.h:30)
    result.jacobian *= hs.detRatio;                // detProposed / detSource (:212-214)
    if (result.jacobian <= 0.0 || !std::isfinite(result.jacobian)) {
      result.weight = 1.0f;
      return false;
    }
    V contrib = getVolumePhotonContrib(hs.throughput, shiftMRec, hs.wi, -shiftRay.d);
    V eyeShiftContrib = shiftGP.eye;
    result.weight = 0.5f;
    result.shiftedFlux = shiftMRec.transmittance * contrib * eyeShiftContrib * result.jacobian;
    F offsetPdf = hs.pdf * pdfShiftRay;
    if (offsetPdf == (F)0) {
      result.weight = 1.0f;
      result.shiftedFlux = V((F)0);
    }
    if (ctx.cfg.use_mis) {
      F basePdf = pdfBaseRay * hs.basePdf;
      if (basePdf == (F)0) {
        result.weight = 0.0f;
      } else {
        const F sensorPart = sensorMIS(shiftGP, *baseGather, currEdge, shiftRay.maxt, baseRay.maxt);
        if (ctx.cfg.power_heuristic) {
          F v = sensorPart * result.jacobian * (offsetPdf / basePdf);
          result.weight = 1.0f / (1.0f + v * v);
        } else {
          result.weight = 1.0f / (1.0f + sensorPart * offsetPdf * result.jacobian / basePdf);
        }
      }
    }
    return true;
  }

  // shiftPhoton dispatch, shift_volume_photon.cpp:49-117 (shiftedEnoughRough == true)
  bool shiftPhoton(const V &offsetPos, const Photon<F> &ph, const CamRay<F> &shiftGP, const Ray<F> &shiftRay,
                   const MRec<F> &shiftMRec, GradientSamplingResult<F> &result, F pdfBaseRay, F pdfShiftRay,
                   F additionalJacobian = (F)1) {
    int type = shiftTypeEnum(ph);
    bool ok;
    if (type == GVPM_SHIFT_INVALID) {
      ok = false;
    } else if (type == GVPM_SHIFT_DIFFUSE) {
      ok = shiftPhotonDiffuse(offsetPos, ph, shiftGP, shiftRay, shiftMRec, result, pdfBaseRay, pdfShiftRay, additionalJacobian);
    } else if (type == GVPM_SHIFT_MEDIUM) {
      // noMediumShift (default true): diffuse reconnection; shiftPhotonMedium is SAssert(false)
      ok = shiftPhotonDiffuse(offsetPos, ph, shiftGP, shiftRay, shiftMRec, result, pdfBaseRay, pdfShiftRay, additionalJacobian);
    } else {
      // EManifoldShift: without useManifold the reference returns false (:101-104)
      ok = ctx.cfg.use_manifold ? shiftPhotonManifold(offsetPos, ph, shiftGP, shiftRay, shiftMRec, result, pdfBaseRay, pdfShiftRay,
                                                      additionalJacobian)
                                : false;
    }
    if (ok) cnt.diffuseShifts++; else cnt.failedShifts++;
    return ok;
  }

  // shiftNull, shift_volume_photon.cpp:119-158
  bool shiftNull(const V &photonFlux, const V &photonWi, const CamRay<F> &shiftGP, const Ray<F> &shiftRay,
                 const MRec<F> &shiftMRec, GradientSamplingResult<F> &resultNull, F pdfBaseRay, F pdfShiftRay,
                 F additionalJacobian = (F)1) {
    cnt.nullShifts++;
    V contrib = getVolumePhotonContrib(photonFlux, shiftMRec, photonWi, -shiftRay.d);
    V eyeShiftContrib = shiftGP.eye;
    resultNull.jacobian *= additionalJacobian;
    resultNull.shiftedFlux = shiftMRec.transmittance * contrib * eyeShiftContrib * resultNull.jacobian;
    resultNull.weight = 0.5f;
    if (ctx.cfg.use_mis) {
      F basePdf = pdfBaseRay;
      F offsetPdf = pdfShiftRay;
      if (offsetPdf == (F)0 || basePdf == (F)0) {
        resultNull.weight = 1.0f;
        return false;
      }
      const F sensorPart = sensorMIS(shiftGP, *baseGather, currEdge, shiftRay.maxt, baseRay.maxt);
      resultNull.weight = 1.0f / (1.0f + sensorPart * offsetPdf * resultNull.jacobian / basePdf);
    }
    return true;
  }

  // getShiftPos, shift_volume_photon.cpp:858-896
  V getShiftPos(const Ray<F> &shiftRay, F radius, const V &basePhotonPos, bool coherent = false) const {
    F baseW = baseRay.maxt;
    F shiftW = shiftRay.maxt;
    V offsetPos = shiftRay(shiftW) + (basePhotonPos - baseRay(baseW));
    if (coherent) {
      V bn = baseRay.d, bs, bt, nn = shiftRay.d, ns, nt;
      coordinateSystemCoherent(bn, bs, bt);
      coordinateSystemCoherent(nn, ns, nt);
      const V v = basePhotonPos - baseRay(baseW);
      const V localD(dot(v, bs), dot(v, bt), dot(v, bn));
      offsetPos = shiftRay(shiftW) + (ns * localD.x + nt * localD.y + nn * localD.z);
    }
    if (ctx.cfg.use_shift_null) {
      F offDistSqr = (baseRay(baseW) - offsetPos).lengthSquared();
      if (offDistSqr < radius * radius) {
        V dShift = shiftRay(shiftRay.maxt) - baseRay(baseRay.maxt);
        dShift /= dShift.length();
        const F cosD = dot(dShift, -(offsetPos - shiftRay(shiftRay.maxt)));
        offsetPos += dShift * cosD * (F)2;
      }
    }
    return offsetPos;
  }

  // VolumeGradientBREQuery::operator(), shift_volume_photon.cpp:658-856
  void breFunctor(const Photon<F> &ph, F photonRadius, F randValue) {
    const double M_PI_D = 3.14159265358979323846;
    const int currDepth = (int)GVPM_PF_DEPTH(ph.flags);
    const V &currFlux = ph.flux;
    const V &currWi = ph.wi;
    const gvpm_params &config = ctx.cfg;
    const bool use3D = config.vol_technique == GVPM_VOL_BRE3D;

    if (config.max_depth > 0 && int(currDepth + currEdge) > config.max_depth) return;
    if (config.min_depth != 0 && int(currDepth + currEdge) < config.min_depth) return;
    if (!computeVolumeContribution(ph)) return;
    if (config.debug_shift != GVPM_SHIFT_ALL && config.debug_shift != GVPM_SHIFT_NULL) {
      if (config.debug_shift != shiftTypeEnum(ph)) return;
    }
    F rrGlobalWeight = 1;
    if (config.path_set) {
      unsigned currentGroup = (unsigned)((baseGather->px + baseGather->py) % 2);
      if (ph.pathID % 2 != currentGroup) return;
      rrGlobalWeight = 2;
    }
    // `M_PI * std::pow(photonRadius, 2)`: pow(Float, int) and M_PI are double
    F kernelVol = (F)(M_PI_D * std::pow((double)photonRadius, 2));
    F pdfCameraPos = 1.f;
    bool validBaseDistance = true;
    if (use3D) {
      kernelVol = (F)((4.0 / 3.0) * M_PI_D * std::pow((double)photonRadius, 3));
      F distSqr = (baseRay(baseRay.maxt) - ph.pos).lengthSquared();
      F deltaT = safe_sqrt(photonRadius * photonRadius - distSqr);
      F tminKernel = baseRay.maxt - deltaT;
      F diskDistanceRand = tminKernel + (deltaT * 2) * randValue;
      if (diskDistanceRand < baseRay.mint || diskDistanceRand > baseGather->len) validBaseDistance = false;
      baseRay.maxt = diskDistanceRand;
      pdfCameraPos = (F)(1.f / std::max((double)deltaT * 2.0, 0.0001));
    }
    V baseContrib((F)0);
    if (validBaseDistance) {
      MRec<F> mRecBase;
      ctx.medium.eval(baseRay, mRecBase);
      V contrib = getVolumePhotonContrib(currFlux, mRecBase, currWi, -baseRay.d);
      V eyeContrib = baseGather->eye;
      baseContrib = mRecBase.transmittance * contrib * eyeContrib;
      mediumFlux += (baseContrib / (kernelVol * pdfCameraPos)) * rrGlobalWeight;
    } else {
      return;
    }
    cnt.evaluations++;

    for (int i = 0; i < 4; ++i) {
      GradientSamplingResult<F> result;
      if (shiftGPs[i].valid) {  // validVolumeEdge(currEdge, currMed)
        const CamRay<F> &shiftGather = shiftGPs[i];
        const V &shiftDir = shiftGather.d;
        const F shiftDistTotal = shiftGather.len;
        Ray<F> shiftRay(shiftGather.o, shiftDir, ctx.Epsilon, baseRay.maxt);
        bool alreadyShift = false;
        if (config.use_shift_null) {
          const F ZPtoY = (shiftRay(shiftRay.maxt) - ph.pos).lengthSquared();
          if (ZPtoY < photonRadius * photonRadius && shiftRay.maxt < shiftDistTotal) {
            V originToCenter = ph.pos - shiftRay.o;
            F diskDistance = dot(originToCenter, shiftRay.d);
            const F distSqr = (shiftRay(diskDistance) - ph.pos).lengthSquared();
            const F deltaT = safe_sqrt(photonRadius * photonRadius - distSqr);
            const F pdfShiftPos = (F)(1.f / std::max(2.0 * (double)deltaT, 0.0001));
            MRec<F> mRecShift;
            ctx.medium.eval(shiftRay, mRecShift);
            shiftNull(currFlux, currWi, shiftGather, shiftRay, mRecShift, result, pdfCameraPos, pdfShiftPos);
            alreadyShift = true;
          }
        }
        if (!alreadyShift) {
          if (shiftDistTotal >= shiftRay.maxt) {
            V offsetPos = getShiftPos(shiftRay, photonRadius, ph.pos, !use3D);
            F pdfShiftPos = 1.f;
            if (use3D) {
              V originToCenter = offsetPos - shiftRay.o;
              F diskDistance = dot(originToCenter, shiftRay.d);
              F distSqr = (shiftRay(diskDistance) - offsetPos).lengthSquared();
              F deltaT = safe_sqrt(photonRadius * photonRadius - distSqr);
              pdfShiftPos = (F)(1.f / std::max(2.0 * (double)deltaT, 0.0001));
            }
            if (config.debug_shift != GVPM_SHIFT_NULL) {
              MRec<F> mRecShift;
              ctx.medium.eval(shiftRay, mRecShift);
              shiftPhoton(offsetPos, ph, shiftGather, shiftRay, mRecShift, result, pdfCameraPos, pdfShiftPos);
            }
          }
        }
      }
      if ((i == GVPM_RIGHT && baseGather->px == config.width - 1) ||
          (i == GVPM_TOP && baseGather->py == config.height - 1)) {
        result.weight = 1.0f;
      }
      weightedMediumFlux[i] += baseContrib * (rrGlobalWeight * result.weight) / (kernelVol * pdfCameraPos);
      shiftedMediumFlux[i] += result.shiftedFlux * (rrGlobalWeight * result.weight) / (kernelVol * pdfCameraPos);
    }
  }

  // ---- G-VPM: VolumeGradientPositionQuery / VolumeGradientDistanceQuery state,
  // shift_volume_photon.h:124-197
  F searchRadius = 0;
  MRec<F> baseMRec;           // from sampleDistance(EDistanceAlwaysValid)
  F baseDistPDF = 0, pdfSelSection = 1;
  MRec<F> shiftMRec[4];
  bool shiftMRecIntialized = false;
  bool validShiftDist[4] = {false, false, false, false};
  F shiftDistCamera[4] = {-1, -1, -1, -1};

  void resetMediumRecCache() {
    for (int i = 0; i < 4; ++i) {
      validShiftDist[i] = false;
      shiftDistCamera[i] = -1.f;
    }
    shiftMRecIntialized = false;
  }
  F pdfBaseRay() const { return baseDistPDF * pdfSelSection; }
  F pdfShiftRay(int id) const { return shiftMRec[id].pdfSuccess * pdfSelSection; }

  // VolumeGradientPositionQuery::operator(), shift_volume_photon.cpp:489-655
  void vpmFunctor(const Photon<F> &ph) {
    const double M_PI_D = 3.14159265358979323846;
    const gvpm_params &config = ctx.cfg;
    V pos = baseRay(baseRay.maxt);
    F lengthSqr = (pos - ph.pos).lengthSquared();
    if ((searchRadius * searchRadius - lengthSqr) < 0) return;
    const int depth = (int)GVPM_PF_DEPTH(ph.flags);
    size_t pathLength = (size_t)(currEdge + depth);
    if ((config.max_depth > 0 && pathLength > (size_t)config.max_depth)) return;
    if (!computeVolumeContribution(ph)) return;
    if (config.debug_shift != GVPM_SHIFT_ALL && config.debug_shift != GVPM_SHIFT_NULL) {
      if (config.debug_shift != shiftTypeEnum(ph)) return;
    }
    V photonContrib = getVolumePhotonContrib(ph.flux, baseMRec, ph.wi, -baseRay.d);
    V eyeContrib = baseGather->eye;
    V baseContrib = eyeContrib * baseMRec.transmittance * photonContrib;
    F kernelVol = (F)((4.0 / 3.0) * M_PI_D * std::pow((double)searchRadius, 3));
    mediumFlux += baseContrib / (kernelVol * pdfBaseRay());
    cnt.evaluations++;

    for (int i = 0; i < 4; ++i) {
      GradientSamplingResult<F> result;
      F additionalJacobian = 1.f;
      if (shiftGPs[i].valid && !shiftMRecIntialized) {
        const CamRay<F> &shiftGather = shiftGPs[i];
        V shiftDir = shiftGather.d;
        F shiftDistMax = shiftGather.len;
        if (shiftDistMax >= baseRay.maxt) {
          validShiftDist[i] = true;
          shiftDistCamera[i] = baseRay.maxt;
          Ray<F> shiftRay(shiftGather.o, shiftDir, ctx.Epsilon, shiftDistMax);
          shiftMRec[i].t = shiftDistCamera[i];
          ctx.medium.eval(shiftRay, shiftMRec[i], true);
        }
      }
      if (validShiftDist[i]) {
        const CamRay<F> &shiftGather = shiftGPs[i];
        Ray<F> shiftRay(shiftGather.o, shiftGather.d, ctx.Epsilon, shiftDistCamera[i]);
        bool alreadyShifted = false;
        if (config.use_shift_null) {
          F distSqr = (ph.pos - shiftRay(shiftRay.maxt)).lengthSquared();
          if (distSqr < searchRadius * searchRadius) {
            alreadyShifted = true;
            shiftNull(ph.flux, ph.wi, shiftGather, shiftRay, shiftMRec[i], result, pdfBaseRay(), pdfShiftRay(i),
                      additionalJacobian);
          }
        }
        if (!alreadyShifted) {
          V offsetPos = getShiftPos(shiftRay, searchRadius, ph.pos);
          if (config.debug_shift != GVPM_SHIFT_NULL) {
            shiftPhoton(offsetPos, ph, shiftGather, shiftRay, shiftMRec[i], result, pdfBaseRay(), pdfShiftRay(i),
                        additionalJacobian);
          }
        }
      } else {
        result.weight = 1.f;
      }
      if ((i == GVPM_RIGHT && baseGather->px == config.width - 1) ||
          (i == GVPM_TOP && baseGather->py == config.height - 1)) {
        result.weight = 1.0f;
      }
      shiftedMediumFlux[i] += result.shiftedFlux * result.weight / (kernelVol * pdfBaseRay());
      weightedMediumFlux[i] += baseContrib * result.weight / (kernelVol * pdfBaseRay());
    }
    shiftMRecIntialized = true;
  }
};

// ---------------------------------------------------------------------------
// Photon map acceleration: PointKDTree (sliding midpoint) + BRE hierarchy
// ---------------------------------------------------------------------------
template <typename F> struct PhotonMap {
  typedef Vec3<F> V;
  std::vector<Photon<F>> photons;  // permuted into kd-tree order by build()
  // kd node info (SimpleKDNode, include/mitsuba/core/kdtree.h:44-112)
  std::vector<uint32_t> right;
  std::vector<uint8_t> nodeFlags;  // bit4 leaf, low nibble axis
  AABB<F> aabb;
  size_t depth = 0;
  // BRE nodes (GBRENode, gvpm/gvpm_accel.h:346-350)
  std::vector<AABB<F>> nodeAABB;
  F radius = 0;
  bool built = false;

  void load(const gvpm_photon_soa &s) {
    photons.resize(s.n);
    aabb = AABB<F>();
    for (uint64_t i = 0; i < s.n; ++i) {
      Photon<F> &p = photons[i];
      p.pos = V(s.pos + 3 * i); p.wi = V(s.wi + 3 * i); p.flux = V(s.flux + 3 * i);
      p.parentPos = V(s.parent_pos + 3 * i); p.parentN = V(s.parent_n + 3 * i);
      p.prefixW = V(s.prefix_w + 3 * i); p.parentScat = V(s.parent_scat + 3 * i);
      p.parentWi = V(s.parent_wi + 3 * i);
      p.parentPdf = (F)s.parent_pdf[i]; p.edgePdf = (F)s.edge_pdf[i];
      p.parentRR = (F)s.parent_rr[i]; p.parentG = (F)s.parent_g[i];
      p.flags = s.flags[i]; p.pathID = s.path_id[i];
      aabb.expandBy(p.pos);  // PointKDTree::push_back, kdtree.h:305-308
    }
    built = false;
  }

  // PointKDTree::build (ESlidingMidpoint), kdtree.h:326-395 and 925-1037
  void buildKD() {
    const size_t n = photons.size();
    right.assign(n, 0);
    nodeFlags.assign(n, 0);
    depth = 0;
    if (n == 0) { built = true; return; }
    std::vector<uint32_t> indirection(n);
    for (size_t i = 0; i < n; ++i) indirection[i] = (uint32_t)i;
    std::vector<uint32_t> rightTmp(n, 0);
    std::vector<uint8_t> flagsTmp(n, 0);
    AABB<F> box = aabb;
    buildRec(1, indirection, 0, n, box, rightTmp, flagsTmp);
    // permute_inplace(&m_nodes[0], indirection): node at slot i becomes old node indirection[i]
    std::vector<Photon<F>> tmp(n);
    for (size_t i = 0; i < n; ++i) {
      tmp[i] = photons[indirection[i]];
      right[i] = rightTmp[indirection[i]];
      nodeFlags[i] = flagsTmp[indirection[i]];
    }
    photons.swap(tmp);
    built = true;
  }

  void buildRec(size_t d, std::vector<uint32_t> &ind, size_t rangeStart, size_t rangeEnd, AABB<F> &box,
                std::vector<uint32_t> &rightTmp, std::vector<uint8_t> &flagsTmp) {
    depth = std::max(d, depth);
    size_t count = rangeEnd - rangeStart;
    if (count == 1) {
      flagsTmp[ind[rangeStart]] |= 0x10;
      return;
    }
    int axis = box.getLargestAxis();
    F midpoint = (F)0.5f * (box.max[axis] + box.min[axis]);
    size_t nLT = 0;
    for (size_t i = rangeStart; i < rangeEnd; ++i)
      if (photons[ind[i]].pos[axis] <= midpoint) nLT++;
    size_t split = rangeStart + nLT;
    if (split == rangeStart) ++split;
    else if (split == rangeEnd) --split;
    std::nth_element(ind.begin() + rangeStart, ind.begin() + split, ind.begin() + rangeEnd,
                     [&](uint32_t a, uint32_t b) { return photons[a].pos[axis] < photons[b].pos[axis]; });
    uint32_t splitNode = ind[split];
    flagsTmp[splitNode] = (uint8_t)((flagsTmp[splitNode] & ~0x0F) | axis);
    flagsTmp[splitNode] &= (uint8_t)~0x10;
    if (split + 1 != rangeEnd) rightTmp[splitNode] = (uint32_t)(split + 1);
    else rightTmp[splitNode] = 0;
    std::swap(ind[rangeStart], ind[split]);
    F temp = box.max[axis], splitPos = photons[splitNode].pos[axis];
    box.max.at(axis) = splitPos;
    buildRec(d + 1, ind, rangeStart + 1, split + 1, box, rightTmp, flagsTmp);
    box.max.at(axis) = temp;
    if (split + 1 != rangeEnd) {
      temp = box.min[axis];
      box.min.at(axis) = splitPos;
      buildRec(d + 1, ind, split + 1, rangeEnd, box, rightTmp, flagsTmp);
      box.min.at(axis) = temp;
    }
  }

  bool isLeaf(uint32_t i) const { return nodeFlags[i] & 0x10; }

  // GradientBeamRadianceEstimator ctor + buildHierarchy, gvpm/gvpm_accel.cpp:10-54
  void buildBRE(F scaleVol) {
    if (!built) buildKD();
    radius = scaleVol;
    nodeAABB.assign(photons.size(), AABB<F>());
    if (!photons.empty()) buildHierarchy(0);
  }
  AABB<F> buildHierarchy(uint32_t index) {
    V center = photons[index].pos;
    AABB<F> box(center - V(radius, radius, radius), center + V(radius, radius, radius));
    if (!isLeaf(index)) {
      uint32_t left = index + 1;
      uint32_t r = right[index];
      if (left) box.expandBy(buildHierarchy(left));
      if (r) box.expandBy(buildHierarchy(r));
    }
    nodeAABB[index] = box;
    return box;
  }

  // GradientBeamRadianceEstimator::query, gvpm/gvpm_accel.h:268-312
  template <typename Q> void queryBRE(const Ray<F> &ray, Q &queryRequest, F randValue) const {
    if (photons.empty()) return;
    // (one stack per thread, grown on demand: an allocation per query serialised 256 threads in malloc -- 6x on 256)
    static thread_local std::vector<uint32_t> stackStorage;
    if (stackStorage.size() < (size_t)depth + 2) stackStorage.resize((size_t)depth + 2);
    uint32_t *stack = stackStorage.data();
    uint32_t index = 0, stackPos = 1;
    while (stackPos > 0) {
      F mint, maxt;
      if (!nodeAABB[index].rayIntersect(ray, mint, maxt) || maxt < ray.mint || mint > ray.maxt) {
        index = stack[--stackPos];
        continue;
      }
      const uint32_t cur = index;
      if (!isLeaf(cur)) {
        if (right[cur] != 0) stack[stackPos++] = right[cur];
        index = cur + 1;
      } else {
        index = stack[--stackPos];
      }
      testPhoton(ray, cur, queryRequest, randValue);
    }
  }

  template <typename Q> inline void testPhoton(const Ray<F> &ray, uint32_t cur, Q &queryRequest, F randValue) const {
    const Photon<F> &ph = photons[cur];
    queryRequest.cnt.candidates++;
    V originToCenter = ph.pos - ray.o;
    F diskDistance = dot(originToCenter, ray.d), radSqr = radius * radius;
    F distSqr = (ray(diskDistance) - ph.pos).lengthSquared();
    if (diskDistance > ray.mint && distSqr < radSqr) {
      Ray<F> baseRay(ray);
      baseRay.maxt = diskDistance;
      queryRequest.baseRay = baseRay;  // newRayBase
      queryRequest.breFunctor(ph, radius, randValue);
    }
  }

  // PointKDTree::executeQuery(p, searchRadius, functor), include/mitsuba/core/kdtree.h:675-731.
  // Returns the number of functor invocations (MVol).
  template <typename Q> size_t executeQuery(const V &p, F searchRadius, Q &functor) const {
    if (photons.empty()) return 0;
    // (one stack per thread, grown on demand: an allocation per query serialised 256 threads in malloc -- 6x on 256)
    static thread_local std::vector<uint32_t> stackStorage;
    if (stackStorage.size() < (size_t)depth + 2) stackStorage.resize((size_t)depth + 2);
    uint32_t *stack = stackStorage.data();
    uint32_t index = 0, stackPos = 1, found = 0;
    F distSquared = searchRadius * searchRadius;
    stack[0] = 0;
    while (stackPos > 0) {
      const uint32_t cur = index;
      uint32_t nextIndex;
      if (!isLeaf(cur)) {
        const int axis = nodeFlags[cur] & 0x0F;
        F distToPlane = p[axis] - photons[cur].pos[axis];
        bool searchBoth = distToPlane * distToPlane <= distSquared;
        const bool hasRight = right[cur] != 0;
        if (distToPlane > 0) {
          if (hasRight) {
            if (searchBoth) stack[stackPos++] = cur + 1;
            nextIndex = right[cur];
          } else if (searchBoth) {
            nextIndex = cur + 1;
          } else {
            nextIndex = stack[--stackPos];
          }
        } else {
          if (searchBoth && hasRight) stack[stackPos++] = right[cur];
          nextIndex = cur + 1;
        }
      } else {
        nextIndex = stack[--stackPos];
      }
      const F pointDistSquared = (photons[cur].pos - p).lengthSquared();
      functor.cnt.candidates++;
      if (pointDistSquared < distSquared) {
        ++found;
        functor.vpmFunctor(photons[cur]);
      }
      index = nextIndex;
    }
    return (size_t)found;
  }
  template <typename Q> size_t executeQueryBrute(const V &p, F searchRadius, Q &functor) const {
    size_t found = 0;
    F distSquared = searchRadius * searchRadius;
    for (uint32_t i = 0; i < photons.size(); ++i) {
      functor.cnt.candidates++;
      if ((photons[i].pos - p).lengthSquared() < distSquared) {
        ++found;
        functor.vpmFunctor(photons[i]);
      }
    }
    return found;
  }

  // Accel-free hit set: a photon is a candidate iff its OWN sphere box passes the slab test
  // against [ray.mint, ray.maxt].  For every photon whose projection falls inside the beam
  // (mint < diskDistance <= maxt) this is exactly what the reference BVH visits (the closest
  // ray point lies in the photon's own box, and every ancestor box contains it).  The two
  // differ only BEYOND the beam end: the reference tests an inner-node photon whenever the box
  // of its whole SUBTREE is hit (gvpm_accel.h:279-294), so photons with diskDistance > maxt are
  // accepted or not depending on where the kd-tree build happened to place them.  The 3D functor
  // rejects them again (t' > edge length, shift_volume_photon.cpp:719-721; residual measure
  // ~Epsilon/r); the 2D functor's far check is an empty block (:726-731), so BRE-2D inherits
  // the tree-dependent extras (the authors' comment there: "Not possible").  The device uses
  // this accel-independent definition; tests compare bit-exact against it and bound the
  // difference to the BVH walk.
  template <typename Q> void queryBrute(const Ray<F> &ray, Q &queryRequest, F randValue) const {
    for (uint32_t i = 0; i < photons.size(); ++i) {
      V c = photons[i].pos;
      AABB<F> box(c - V(radius, radius, radius), c + V(radius, radius, radius));
      F mint, maxt;
      if (!box.rayIntersect(ray, mint, maxt) || maxt < ray.mint || mint > ray.maxt) continue;
      testPhoton(ray, i, queryRequest, randValue);
    }
  }
};

// ---------------------------------------------------------------------------
// Drivers
// ---------------------------------------------------------------------------
template <typename F> struct Gatherer {
  typedef Vec3<F> V;
  GatherContext<F> ctx;
  PhotonMap<F> map;

  void setup(const gvpm_params &p, const gvpm_medium &m, const gvpm_triangles &t) {
    ctx.cfg = p;
    ctx.medium.sigmaA = V(m.sigma_a); ctx.medium.sigmaS = V(m.sigma_s); ctx.medium.sigmaT = V(m.sigma_t);
    ctx.medium.g = (F)m.g; ctx.medium.mediumSamplingWeight = (F)m.medium_sampling_weight;
    ctx.Epsilon = (F)p.epsilon; ctx.ShadowEpsilon = (F)p.shadow_epsilon;
    ctx.scene.v0.clear(); ctx.scene.e1.clear(); ctx.scene.e2.clear();
    for (uint32_t i = 0; i < t.n; ++i) {
      ctx.scene.v0.push_back(V(t.v0 + 3 * i));
      ctx.scene.e1.push_back(V(t.e1 + 3 * i));
      ctx.scene.e2.push_back(V(t.e2 + 3 * i));
    }
  }

  // One beam set of computeVolumeGradientPhotonBRE's inner loop, gvpm.cpp:1018-1052.
  // iter: 27 F values of the set's pixel (fluxVolIter, shifted..Iter[4], weighted..Iter[4])
  void gatherSetBRE(const gvpm_camera_ray *set, bool useAccel, F *iter, Counters &cnt) const {
    CamRay<F> base(set[0]);
    CamRay<F> shifts[4] = {CamRay<F>(set[1]), CamRay<F>(set[2]), CamRay<F>(set[3]), CamRay<F>(set[4])};
    // Ray ray(oBeam, dBeam, Epsilon, beamDist - Epsilon), gvpm.cpp:1032-1038
    Ray<F> ray(base.o, base.d, ctx.Epsilon, base.len - ctx.Epsilon);
    VolumeGradientRecord<F> gRec(ctx, &base, shifts);
    gRec.baseRay = ray;
    gRec.clear();
    if (useAccel) map.queryBRE(ray, gRec, base.rand);
    else map.queryBrute(ray, gRec, base.rand);
    for (int c = 0; c < 3; ++c) {
      iter[c] += gRec.mediumFlux[c];
      for (int k = 0; k < 4; ++k) {
        iter[3 + 3 * k + c] += gRec.shiftedMediumFlux[k][c];
        iter[15 + 3 * k + c] += gRec.weightedMediumFlux[k][c];
      }
    }
    cnt.add(gRec.cnt);
  }
};

// One camera sample of computeVolumeGradientPhoton's inner loop, gvpm.cpp:1143-1180.
// iter: 27 F values of the pixel (already multiplied by `normalization`); returns MVol of the sample.
template <typename F>
inline size_t gatherSampleVPM(const Gatherer<F> &g, const gvpm_camera_ray *set, F rand, F pdfSel, F querySize,
                              F normalization, bool useAccel, F *iter, Counters &cnt) {
  CamRay<F> base(set[0]);
  CamRay<F> shifts[4] = {CamRay<F>(set[1]), CamRay<F>(set[2]), CamRay<F>(set[3]), CamRay<F>(set[4])};
  // Ray ray(oBeam, dBeam, Epsilon, beamDist), gvpm.cpp:1165
  Ray<F> ray(base.o, base.d, g.ctx.Epsilon, base.len);
  MRec<F> mRec;
  mRec.t = 0;
  if (!g.ctx.medium.sampleDistanceAlwaysValid(ray, mRec, rand, g.ctx.Epsilon)) return 0;
  ray.maxt = mRec.t;
  VolumeGradientRecord<F> gRec(g.ctx, &base, shifts);
  // changeEdge + newRayBase(ray, mRec, querySize, mRec.pdfSuccess), shift_volume_photon.h:134-176
  gRec.pdfSelSection = pdfSel;
  gRec.baseRay = ray;
  gRec.resetMediumRecCache();
  gRec.baseMRec = mRec;
  gRec.searchRadius = querySize;
  gRec.baseDistPDF = mRec.pdfSuccess;
  gRec.clear();
  const Vec3<F> p = ray.o + ray.d * mRec.t;
  size_t found = useAccel ? g.map.executeQuery(p, querySize, gRec) : g.map.executeQueryBrute(p, querySize, gRec);
  for (int c = 0; c < 3; ++c) {
    iter[c] += gRec.mediumFlux[c] * normalization;
    for (int k = 0; k < 4; ++k) {
      iter[3 + 3 * k + c] += gRec.shiftedMediumFlux[k][c] * normalization;
      iter[15 + 3 * k + c] += gRec.weightedMediumFlux[k][c] * normalization;
    }
  }
  cnt.add(gRec.cnt);
  return found;
}

// scaleVolumeAPA, gvpm.cpp:181-215 (m_independentScale == false, forceAPA empty)
inline double scaleVolumeAPA(double globalScaleVolume, int it, double alpha, int technique) {
  it -= 1;
  double ratioVolAPA = (it + alpha) / (it + 1);
  bool use3D = technique == GVPM_DISTANCE || technique == GVPM_VOL_BRE3D || technique == GVPM_BEAM_BEAM_3D_NAIVE ||
               technique == GVPM_BEAM_BEAM_3D_EGSR || technique == GVPM_BEAM_BEAM_3D_OPTIMIZED;
  if (use3D) return globalScaleVolume * std::cbrt(ratioVolAPA);
  if (technique == GVPM_VOL_BRE2D) return globalScaleVolume * std::sqrt(ratioVolAPA);
  return globalScaleVolume * ratioVolAPA;
}

}  // namespace oracle
