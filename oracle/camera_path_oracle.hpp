// TEST INFRASTRUCTURE (see gvpm_oracle.hpp): CPU restatement of the camera-path shift pieces the synthetic hosts stand
// in for -- only tests/ touch this.
//
//   halfVectorShift             gvpm/shift/shift_utilities.h:42-110
//   reflect / refract           src/libcore/util.cpp:771-800
//   GatherPoint::sensorMIS      gvpm/gvpm_struct.h:608-631 (literal: the host-side check of the device's cancelled form)
#pragma once
#include <cmath>

namespace gvpm_oracle {

struct HV3 {
  double x, y, z;
};
static inline HV3 hv(double x, double y, double z) { return HV3{x, y, z}; }
static inline HV3 operator+(HV3 a, HV3 b) { return hv(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline HV3 operator-(HV3 a, HV3 b) { return hv(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline HV3 operator-(HV3 a) { return hv(-a.x, -a.y, -a.z); }
static inline HV3 operator*(HV3 a, double s) { return hv(a.x * s, a.y * s, a.z * s); }
static inline double hdot(HV3 a, HV3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline HV3 hnormalize(HV3 a) { return a * (1.0 / std::sqrt(hdot(a, a))); }

// util.cpp:771-773
static inline HV3 reflectRef(HV3 wi, HV3 n) { return n * (2 * hdot(wi, n)) - wi; }
// util.cpp:782-800
static inline HV3 refractRef(HV3 wi, HV3 n, double eta) {
  if (eta == 1) return -wi;
  const double cosThetaI = hdot(wi, n);
  if (cosThetaI > 0) eta = 1 / eta;
  const double cosThetaTSqr = 1 - (1 - cosThetaI * cosThetaI) * (eta * eta);
  if (cosThetaTSqr <= 0.0) return hv(0, 0, 0);
  const double sg = cosThetaI > 0 ? 1.0 : (cosThetaI < 0 ? -1.0 : 0.0);  // math::signum
  return n * (cosThetaI * eta - sg * std::sqrt(cosThetaTSqr)) - wi * eta;
}

struct HalfVectorShiftResult {
  bool success;
  double jacobian;
  HV3 wo;
};

// shift_utilities.h:42-110, statement for statement (tangent space: z = cosTheta); D_EPSILON = 1e-14 (:13)
static inline HalfVectorShiftResult halfVectorShift(HV3 mainWi, HV3 mainWo, HV3 shiftedWi, double mainEta, double shiftedEta) {
  const double D_EPSILON = 1e-14;
  HalfVectorShiftResult result;
  result.success = false;
  result.jacobian = 0;
  result.wo = hv(0, 0, 0);
  if (mainWi.z * mainWo.z < 0) {
    // Refraction
    if (mainEta == 1 || shiftedEta == 1) return result;
    HV3 hMain;
    if (mainWi.z < 0) hMain = -(mainWi * mainEta + mainWo);
    else hMain = -(mainWi + mainWo * mainEta);
    const HV3 h = hnormalize(hMain);
    const HV3 shiftedWo = refractRef(shiftedWi, h, shiftedEta);
    if (shiftedWo.x == 0 && shiftedWo.y == 0 && shiftedWo.z == 0) return result;
    HV3 hShift;
    if (shiftedWi.z < 0) hShift = -(shiftedWi * shiftedEta + shiftedWo);
    else hShift = -(shiftedWi + shiftedWo * shiftedEta);
    const double hLengthSquared = hdot(hShift, hShift) / (D_EPSILON + hdot(hMain, hMain));
    const double WoDotH = std::fabs(hdot(mainWo, h)) / (D_EPSILON + std::fabs(hdot(shiftedWo, h)));
    result.success = true;
    result.wo = shiftedWo;
    result.jacobian = hLengthSquared * WoDotH;
  } else {
    // Reflection
    const HV3 h = hnormalize(mainWi + mainWo);
    const HV3 shiftedWo = reflectRef(shiftedWi, h);
    const double WoDotH = hdot(shiftedWo, h) / hdot(mainWo, h);
    result.success = true;
    result.wo = shiftedWo;
    result.jacobian = std::fabs(WoDotH);
  }
  return result;
}

// GatherPoint::sensorMIS, gvpm_struct.h:608-631, as written (pdf, jacobian and GOp of the shifted and the base path)
static inline double sensorMISRef(unsigned idVertex, double sPdf, double sJac, double sG, double bPdf, double bG, double sDist,
                                  double bDist) {
  double jacobian = sJac;
  double ratio = sPdf / bPdf;
  if (idVertex != 1) {
    jacobian *= sG / bG;
    ratio *= bG / sG;
    jacobian *= (sDist / bDist) * (sDist / bDist);
    ratio *= (bDist / sDist) * (bDist / sDist);
  }
  return ratio * jacobian;
}

}  // namespace gvpm_oracle
