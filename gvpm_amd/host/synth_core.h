// The synthetic generators' per-path / per-pixel logic, written once for the host (g++, synth.cpp) and the
// device (hipcc, csrc/synth_device.hip): SURVEY 8f row f3, device-side photon shooting and camera-beam generation
// for closed-form scenes.  Everything here is a pure function of (scene, iteration, index): the Philox streams are
// keyed by the path / pixel index, so the device can run every path in its own lane and still produce the
// sequential host loop's output (up to the last bit of libm's exp / log / sin / cos).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

#include "../../include/gvpm_hip.h"
#include "philox.h"
#include "vecmath.h"

namespace gvpm {

// MAT_MIRROR: perfect specular reflector (a Dirac BSDF, reflectance = albedo): what puts a SECOND medium edge on a
// camera path (randomWalkFromPixelToFirstDiffuse walks on past smooth vertices) and a non-diffuse vertex on light paths
// MAT_PHONG: the modified Phong BSDF of src/bsdfs/phong.cpp, both components sampled together (roughness >= 0.05, so
// that PathVertex::sampleNext keeps sampledComponentIndex = -1, vertex.cpp:160-180): a GLOSSY wall that still classifies
// as diffuse for the reconnection (gvpm_struct.h:66-100) -- the parent type GVPM_PARENT_SURFACE_BSDF of the ABI
// MAT_ROUGHCONDUCTOR: src/bsdfs/roughconductor.cpp, isotropic Beckmann / GGX, sampled WITHOUT visible normals
// (sampleVisible = false: MicrofacetDistribution::sampleAll) -- the table's second kind
// MAT_WARD (round 5): src/bsdfs/ward.cpp, isotropic (alphaU == alphaV = `exponent`), roughness >= 0.05: both components sampled
// together; `distribution` holds the model variant (GVPM_WARD_*)
// The glossy kinds are sampled by the HOST generators only: the device generator's closed set is Lambertian / index-matched /
// mirror (gvpm_devgen_create refuses the others), and their fp64 pow / atan / log chains cost the device walk a third of its
// time in registers alone when they were merely compiled in.
#ifdef __HIP_DEVICE_COMPILE__
#define GVPM_SYNTH_GLOSSY 0
#else
#define GVPM_SYNTH_GLOSSY 1
#endif
enum MatKind { MAT_LAMBERT = 0, MAT_NULL = 1, MAT_MIRROR = 2, MAT_PHONG = 3, MAT_ROUGHCONDUCTOR = 4, MAT_WARD = 5 };
// table entries of a glossy material: PathVertex::sampleNext picks ONE component of a Phong surface below roughness 0.05
// (vertex.cpp:160-165, Phong::getRoughness = sqrt(2 / (2 + exponent)), phong.cpp:293-300): an entry per component then
GVPM_HD inline bool phongOneComponent(double exponent) { return sqrt(2.0 / (2.0 + exponent)) < 0.05; }
GVPM_HD inline int bsdfEntries(int kind, double exponent) {
  return kind == MAT_PHONG ? (phongOneComponent(exponent) ? 2 : 1) : ((kind == MAT_ROUGHCONDUCTOR || kind == MAT_WARD) ? 1 : 0);
}

struct SynthTri {
  V3 v0, e1, e2, n;  // n: geometric normal (front side)
  int mat;
};
struct SynthMat {
  int kind;
  V3 albedo;          // Lambertian / mirror reflectance; Phong: the diffuse reflectance
  V3 spec;            // Phong: specular reflectance
  double exponent;    // Phong exponent
  double specWeight;  // m_specularSamplingWeight = lum(spec) / (lum(diffuse) + lum(spec)), phong.cpp:93-97
  int bsdf;           // index in the table of gvpm_upload_bsdfs (-1: none); a Phong below roughness 0.05 has TWO entries:
                      // bsdf = met through its specular component, bsdf + 1 = through its diffuse one (phongEntries)
  // rough conductor: spec = specular reflectance, exponent = alpha
  V3 eta = V3(0.0), k = V3(0.0);
  int distribution = 0;  // GVPM_MICROFACET_*
};

// what the generators read of a scene (SynthScene::view(); the device gets the arrays in HBM)
struct SceneView {
  const SynthTri *tris;
  int ntri;
  const SynthMat *mats;
  int nmats;
  V3 lightC, lightU, lightV, lightN, radiance;
  double lightArea;
  gvpm_medium medium;
  V3 camPos;
  V3 camX, camY, camZ;  // camera -> world columns; the sensor looks along -camZ (identity for the axis-aligned scenes)
  double tanHalfFovX;
  int width, height;
  uint32_t seed;
  bool cameraInside;
  int maxDepth, rrDepth, minDepth;
  double cameraSphere;
};

constexpr double kPi = 3.14159265358979323846;
constexpr double kInvPi = 1.0 / kPi;
constexpr double kInvFourPi = 1.0 / (4.0 * kPi);
constexpr double kEpsilon = 1e-4;  // Epsilon, single-precision build (constants.h:24-31)

// isotropic microfacet terms of the rough conductor (microfacet.h:191-232, 477-518; util.cpp:747-769), by cosines
GVPM_HD inline double conductorD(int ggx, double alpha, double cH) {
  if (cH <= 0) return 0;
  const double c2 = cH * cH, e = (1 - c2) / (alpha * alpha * c2);
  double r;
  if (ggx) {
    const double root = (1 + e) * c2;
    r = 1.0 / (kPi * alpha * alpha * root * root);
  } else {
    r = std::exp(-e) / (kPi * alpha * alpha * c2 * c2);
  }
  return r * cH < 1e-20 ? 0.0 : r;
}
GVPM_HD inline double conductorG1(int ggx, double alpha, double cV, double vDotH) {
  if (vDotH * cV <= 0) return 0;
  const double t2 = 1 - cV * cV;
  if (t2 <= 0) return 1;
  const double tanT = std::fabs(std::sqrt(t2) / cV);
  if (ggx) {
    const double root = alpha * tanT;
    return 2.0 / (1.0 + std::sqrt(1.0 + root * root));
  }
  const double a = 1.0 / (alpha * tanT);
  if (a >= 1.6) return 1;
  return (3.535 * a + 2.181 * a * a) / (1.0 + 2.276 * a + 2.577 * a * a);
}
GVPM_HD inline double conductorFresnel(double cI, double eta, double k) {
  const double c2 = cI * cI, s2 = 1 - c2, s4 = s2 * s2;
  const double t1 = eta * eta - k * k - s2;
  const double a2pb2 = std::sqrt(std::fmax(0.0, t1 * t1 + k * k * eta * eta * 4));
  const double aa = std::sqrt(std::fmax(0.0, (a2pb2 + t1) * 0.5));
  const double term1 = a2pb2 + c2, term2 = aa * (2 * cI);
  const double Rs2 = (term1 - term2) / (term1 + term2);
  const double term3 = a2pb2 * c2 + s4, term4 = term2 * s2;
  return 0.5 * (Rs2 * (term3 - term4) / (term3 + term4) + Rs2);
}

// ------------------------------------------------------------------ scenes --
struct Hit {
  double t;
  int tri;
};

// closest intersection with any triangle, t in (mint, inf); two-sided like
// Mitsuba's triangle kd-tree
GVPM_HD inline bool closestHit(const SceneView &sc, V3 o, V3 d, double mint, Hit &hit) {
  hit.t = std::numeric_limits<double>::infinity();
  hit.tri = -1;
#ifdef __HIP_DEVICE_COMPILE__
  // (device: the ray's directions and origin once in fp32 for the cull below)
  const float of[3] = {(float)o.x, (float)o.y, (float)o.z}, df[3] = {(float)d.x, (float)d.y, (float)d.z};
#endif
  for (size_t i = 0; i < (size_t)sc.ntri; ++i) {
    const SynthTri &tr = sc.tris[i];
#ifdef __HIP_DEVICE_COMPILE__
    {
      // A conservative fp32 statement of the two barycentric tests, without the division: a triangle it rejects fails the
      // fp64 test below by a wide margin (E bounds the fp32 error of the three sums by their operands' magnitudes, x 30),
      // so the triangles that reach the fp64 test -- one or two of a room's thirty -- decide exactly as the loop over all of
      // them does on the host: same hit, same order of ties.
      const float e1[3] = {(float)tr.e1.x, (float)tr.e1.y, (float)tr.e1.z}, e2[3] = {(float)tr.e2.x, (float)tr.e2.y, (float)tr.e2.z};
      const float tv[3] = {of[0] - (float)tr.v0.x, of[1] - (float)tr.v0.y, of[2] - (float)tr.v0.z};
      const float px = df[1] * e2[2] - df[2] * e2[1], py = df[2] * e2[0] - df[0] * e2[2], pz = df[0] * e2[1] - df[1] * e2[0];
      const float qx = tv[1] * e1[2] - tv[2] * e1[1], qy = tv[2] * e1[0] - tv[0] * e1[2], qz = tv[0] * e1[1] - tv[1] * e1[0];
      const float C = e1[0] * px + e1[1] * py + e1[2] * pz;
      const float A = tv[0] * px + tv[1] * py + tv[2] * pz;
      const float Bq = df[0] * qx + df[1] * qy + df[2] * qz;
      const float l1 = fabsf(px) + fabsf(py) + fabsf(pz), lq = fabsf(qx) + fabsf(qy) + fabsf(qz);
      const float E = 4e-6f * ((fabsf(tv[0]) + fabsf(tv[1]) + fabsf(tv[2]) + fabsf(e1[0]) + fabsf(e1[1]) + fabsf(e1[2])) * l1 +
                               (fabsf(df[0]) + fabsf(df[1]) + fabsf(df[2])) * lq) + 1e-30f;
      const float sg = C < 0.f ? -1.f : 1.f, aC = fabsf(C);
      if (sg * A < -E || sg * A > aC + E || sg * Bq < -E || sg * (A + Bq) > aC + 2.f * E) continue;
    }
#endif
    V3 p = cross(d, tr.e2);
    double det = dot(tr.e1, p);
    if (det == 0.0) continue;
    double inv = 1.0 / det;
    V3 tv = o - tr.v0;
    double u = dot(tv, p) * inv;
    if (u < 0.0 || u > 1.0) continue;
    V3 q = cross(tv, tr.e1);
    double v = dot(d, q) * inv;
    if (v < 0.0 || u + v > 1.0) continue;
    double t = dot(tr.e2, q) * inv;
    if (t > mint && t < hit.t) {
      hit.t = t;
      hit.tri = (int)i;
    }
  }
  return hit.tri >= 0;
}


// ------------------------------------------------------------ light paths --
enum { VT_SUPERNODE = 0, VT_EMITTER = 1, VT_SURFACE = 2, VT_MEDIUM = 3 };

struct LVertex {
  int type;
  V3 pos, n;
  V3 weight;      // weight[EImportance]
  double rr;      // rrWeight
  double pdf;     // pdf[EImportance], area measure once the successor exists
  V3 eWeight;     // edge(i) = edge from vertex i to i+1
  double ePdf;
  bool eMedium;
  V3 albedo;
  int matKind;
  int mat;          // material index of a surface vertex (-1 otherwise)
  uint32_t comp;    // componentType of a sampled surface vertex (BSDF::EBSDFType of the sampled lobe)
  int compSel;      // sampledComponentIndex: -1 all components, else the one sampleComponent picked (vertex.cpp:160-180)
};

// a light path: at most maxDepth + 1 vertices
constexpr int GVPM_SYNTH_MAXV = 16;
struct LPath {
  LVertex v[GVPM_SYNTH_MAXV];
  int n;
  GVPM_HD LPath() : n(0) {}
  GVPM_HD size_t size() const { return (size_t)n; }
  GVPM_HD LVertex &operator[](size_t i) { return v[i]; }
  GVPM_HD const LVertex &operator[](size_t i) const { return v[i]; }
  GVPM_HD void clear() { n = 0; }
  GVPM_HD void push_back(const LVertex &x) {
    if (n < GVPM_SYNTH_MAXV) v[n++] = x;
  }
};


GVPM_HD inline V3 cosineHemisphere(double u1, double u2) {
  // warp::squareToCosineHemisphere (concentric disk not required for parity:
  // the synthetic host only has to be a valid sampler with the stated pdf)
  double r = std::sqrt(u1), phi = 2.0 * kPi * u2;
  double x = r * std::cos(phi), y = r * std::sin(phi);
  return V3(x, y, std::sqrt(std::fmax(0.0, 1.0 - u1)));
}

GVPM_HD inline V3 uniformSphere(double u1, double u2) {
  double z = 1.0 - 2.0 * u1;
  double r = std::sqrt(std::fmax(0.0, 1.0 - z * z)), phi = 2.0 * kPi * u2;
  return V3(r * std::cos(phi), r * std::sin(phi), z);
}

GVPM_HD inline V3 toWorld(V3 n, V3 local) {
  V3 s, t;
  coordinateSystem(n, s, t);
  return s * local.x + t * local.y + n * local.z;
}

GVPM_HD inline double hgEval(double g, double cosWiWo) {
  // HGPhaseFunction::eval, src/phase/hg.cpp:107-110 (wi points away from the vertex)
  double temp = 1.0 + g * g + 2.0 * g * cosWiWo;
  return kInvFourPi * (1 - g * g) / (temp * std::sqrt(temp));
}

// One light path; mirrors Path::randomWalk(scene, sampler, maxDepth, rrDepth, EImportance)
// The walk in two pieces, so that a GPU lane whose path has ended can begin the next one while its neighbours walk on
// (synth_device.hip): walkBegin appends the supernode and the emitter vertex, walkStep(i) samples vertex i's direction and
// appends vertex i + 1 (false: the path ends at vertex i).  randomWalk = walkBegin + walkStep for i = 1 .. maxDepth - 1.
template <class PATH> GVPM_HD inline void walkBegin(const SceneView &sc, Philox &rng, PATH &path, V3 &throughput) {
  path.clear();

  LVertex v0;
  v0.type = VT_SUPERNODE;
  v0.weight = sc.radiance * (kPi * sc.lightArea);  // AreaLight::samplePosition -> m_power
  v0.pdf = 1.0 / sc.lightArea;
  v0.rr = 1.0;
  v0.eWeight = V3(1.0);
  v0.ePdf = 1.0;
  v0.eMedium = true;
  v0.matKind = -1;
  v0.mat = -1;
  v0.comp = GVPM_BSDF_DIFFUSE_REFLECTION;
  v0.compSel = -1;
  double u1 = rng.next1D(), u2 = rng.next1D();
  LVertex v1;
  v1.type = VT_EMITTER;
  v1.pos = sc.lightC + sc.lightU * (u1 - 0.5) + sc.lightV * (u2 - 0.5);
  v1.n = sc.lightN;
  v1.matKind = -1;
  v1.mat = -1;
  v1.comp = GVPM_BSDF_DIFFUSE_REFLECTION;
  v1.compSel = -1;
  path.push_back(v0);
  path.push_back(v1);

  throughput = V3(1.0);  // the supernode case returns before `throughput *= weight`
}

template <class PATH> GVPM_HD inline bool walkStep(const SceneView &sc, Philox &rng, PATH &path, V3 &throughput, int i) {
  const double sigT = sc.medium.sigma_t[1];
  const double msw = sc.medium.medium_sampling_weight;
  const V3 sigS(sc.medium.sigma_s[0], sc.medium.sigma_s[1], sc.medium.sigma_s[2]);
  const double g = sc.medium.g;
  {
    LVertex &cur = path[i];
    V3 wo;
    double mint = kEpsilon;
    bool solidAngle = true;
    if (cur.type == VT_EMITTER) {
      // (second argument drawn first: the order g++ evaluated `cosineHemisphere(rng.next1D(), rng.next1D())` in when
      // the committed fixtures were generated; written out so that host and device compilers agree)
      const double e2 = rng.next1D(), e1 = rng.next1D();
      V3 local = cosineHemisphere(e1, e2);
      wo = toWorld(cur.n, local);
      cur.weight = V3(1.0);
      cur.pdf = local.z * kInvPi;
      if (cur.pdf <= 0) return false;
    } else if (cur.type == VT_SURFACE) {
      V3 wi = normalize(path[i - 1].pos - cur.pos);
      double a = rng.next1D(), b = rng.next1D();
      if (cur.matKind == MAT_NULL) {
        // index-matched medium boundary: passes straight through and leaves the fog
        return false;
      }
      if (dot(cur.n, wi) <= 0) return false;  // one-sided BSDFs
      if (cur.matKind == MAT_MIRROR) {
        // Dirac reflection: weight = reflectance, pdf = 1 in the discrete measure (not converted to area below)
        wo = cur.n * (2.0 * dot(cur.n, wi)) - wi;
        cur.weight = cur.albedo;
        cur.pdf = 1.0;
        solidAngle = false;
        if (maxc(cur.weight) <= 0) return false;
#if GVPM_SYNTH_GLOSSY
      } else if (cur.matKind == MAT_PHONG) {
        // PathVertex::sampleNext (vertex.cpp:160-173): below roughness 0.05 Phong::sampleComponent (phong.cpp:308-329) picks
        // ONE component first -- and rescales the sample as written there (the specular branch MULTIPLIES by the weight) --,
        // then Phong::sample (:188-247) with bRec.component: with both components the sample picks the lobe and weight / pdf
        // are the whole BSDF's; with one, that lobe's alone, then weight /= pdfComponent, pdf *= pdfComponent.
        const SynthMat &pm = sc.mats[cur.mat];
        const double sw = pm.specWeight, e = pm.exponent, cosWi = dot(cur.n, wi);
        const V3 refl = cur.n * (2.0 * cosWi) - wi;  // reflect(wi) in world space
        double sx = a;
        int compSel = -1;
        double pdfComp = 1.0;
        if (phongOneComponent(e)) {  // Phong::getRoughness (:293-300) against sampleNext's constant
          if (sx < sw) {
            compSel = 0;
            pdfComp = sw;
            sx *= sw;
          } else {
            compSel = 1;
            pdfComp = 1 - sw;
            sx = (sx - sw) / (1 - sw);
          }
        }
        const bool hasSpecular = compSel != 1, hasDiffuse = compSel != 0;
        bool choseSpecular = hasSpecular;
        if (hasSpecular && hasDiffuse) {
          if (sx <= sw) {
            sx /= sw;
          } else {
            sx = (sx - sw) / (1 - sw);
            choseSpecular = false;
          }
        }
        if (choseSpecular) {
          const double sinAlpha = std::sqrt(std::fmax(0.0, 1 - std::pow(b, 2 / (e + 1))));
          const double cosAlpha = std::pow(b, 1 / (e + 1));
          const double phi = 2.0 * kPi * sx;
          wo = toWorld(refl, V3(sinAlpha * std::cos(phi), sinAlpha * std::sin(phi), cosAlpha));
          cur.comp = 0x00008u;  // EGlossyReflection
        } else {
          wo = toWorld(cur.n, cosineHemisphere(sx, b));
          cur.comp = GVPM_BSDF_DIFFUSE_REFLECTION;
        }
        cur.compSel = compSel;
        const double cosWo = dot(cur.n, wo);
        if (cosWo <= 0) return false;
        const double alpha = dot(wo, refl);
        const double lobe = (hasSpecular && alpha > 0) ? std::pow(alpha, e) : 0.0;
        const double specProb = lobe * (e + 1) / (2.0 * kPi), diffProb = hasDiffuse ? cosWo * kInvPi : 0.0;
        double pdfW = (hasSpecular && hasDiffuse) ? sw * specProb + (1 - sw) * diffProb : (hasDiffuse ? diffProb : specProb);
        if (pdfW == 0) return false;
        const V3 f = (pm.spec * ((e + 2) / (2.0 * kPi) * lobe) + pm.albedo * (hasDiffuse ? kInvPi : 0.0)) * cosWo;
        cur.weight = f * (1.0 / pdfW);
        if (compSel != -1) {
          cur.weight = cur.weight * (1.0 / pdfComp);
          pdfW *= pdfComp;
        }
        cur.pdf = pdfW;
        if (maxc(cur.weight) <= 0) return false;
      } else if (cur.matKind == MAT_WARD) {
        // Ward::sample with bRec.component = -1 (ward.cpp:268-327; sampleComponent returns -1 for alpha >= 0.05, :370-376): the
        // sample picks the lobe; specular: half vector H ~ the Ward lobe (phiH, thetaH as written there), wo = reflect(wi, H);
        // weight = eval / pdf of the WHOLE BSDF, pdf = the mixture's (:230-266)
        const SynthMat &pm = sc.mats[cur.mat];
        const double sw = pm.specWeight, al = pm.exponent, cosWi = dot(cur.n, wi);
        double sx = a;
        bool choseSpecular = true;
        if (sx <= sw) {
          sx /= sw;
        } else {
          sx = (sx - sw) / (1 - sw);
          choseSpecular = false;
        }
        if (choseSpecular) {
          double phiH = std::atan(std::tan(2.0 * kPi * b));  // (alphaV / alphaU = 1)
          if (b > 0.5) phiH += kPi;
          const double cosPhiH = std::cos(phiH), sinPhiH = std::sqrt(std::fmax(0.0, 1.0 - cosPhiH * cosPhiH));
          const double thetaH = std::atan(std::sqrt(std::fmax(0.0, -std::log(sx) / ((cosPhiH * cosPhiH + sinPhiH * sinPhiH) / (al * al)))));
          const V3 H = toWorld(cur.n, V3(std::sin(thetaH) * std::cos(phiH), std::sin(thetaH) * std::sin(phiH), std::cos(thetaH)));
          wo = H * (2.0 * dot(wi, H)) - wi;
          cur.comp = 0x00008u;  // EGlossyReflection
        } else {
          wo = toWorld(cur.n, cosineHemisphere(sx, b));
          cur.comp = GVPM_BSDF_DIFFUSE_REFLECTION;
        }
        const double cosWo = dot(cur.n, wo);
        if (cosWo <= 0) return false;
        // eval (H not normalised, as written) and pdf (normalised H)
        const V3 Hs = wi + wo;
        const double HH = dot(Hs, Hs), Hz = cosWi + cosWo;
        const double E = std::exp(-((HH - Hz * Hz) / (al * al)) / (Hz * Hz));
        double factor1;
        if (pm.distribution == GVPM_WARD_WARD) factor1 = 1.0 / (4.0 * kPi * al * al * std::sqrt(cosWi * cosWo));
        else if (pm.distribution == GVPM_WARD_DUER) factor1 = 1.0 / (4.0 * kPi * al * al * cosWi * cosWo);
        else factor1 = HH / (kPi * al * al * Hz * Hz * Hz * Hz);
        const double specRef = factor1 * E;
        const double lenH = std::sqrt(HH), cH = Hz / lenH, wiH = dot(wi, Hs) / lenH;
        const double specProb = E / (4.0 * kPi * al * al * wiH * cH * cH * cH);
        const double pdfW = sw * specProb + (1 - sw) * cosWo * kInvPi;
        if (pdfW == 0) return false;
        const V3 f = (pm.spec * (specRef > 1e-10 ? specRef : 0.0) + pm.albedo * kInvPi) * cosWo;
        cur.weight = f * (1.0 / pdfW);
        cur.pdf = pdfW;
        if (maxc(cur.weight) <= 0) return false;
      } else if (cur.matKind == MAT_ROUGHCONDUCTOR) {
        // RoughConductor::sample, sampleVisible = false (roughconductor.cpp:321-389 with MicrofacetDistribution::sampleAll,
        // microfacet.h:287-347): half vector m ~ D cos, wo = reflect(wi, m), weight = F D G (wi . m) / (pdf_m cos_i),
        // pdf = pdf_m / (4 |wo . m|)
        const SynthMat &pm = sc.mats[cur.mat];
        const double alpha = pm.exponent, alphaSqr = alpha * alpha, cosWi = dot(cur.n, wi);
        double tanThetaMSqr, pdfM, cosThetaM;
        if (pm.distribution == GVPM_MICROFACET_GGX) {
          tanThetaMSqr = alphaSqr * a / (1.0 - a);
          cosThetaM = 1.0 / std::sqrt(1.0 + tanThetaMSqr);
          const double temp = 1 + tanThetaMSqr / alphaSqr;
          pdfM = kInvPi / (alphaSqr * cosThetaM * cosThetaM * cosThetaM * temp * temp);
        } else {
          tanThetaMSqr = alphaSqr * -std::log(1.0 - a);
          cosThetaM = 1.0 / std::sqrt(1.0 + tanThetaMSqr);
          pdfM = (1.0 - a) / (kPi * alphaSqr * cosThetaM * cosThetaM * cosThetaM);
        }
        if (!(pdfM >= 1e-20)) return false;
        const double sinThetaM = std::sqrt(std::fmax(0.0, 1 - cosThetaM * cosThetaM)), phi = 2.0 * kPi * b;
        const V3 m = toWorld(cur.n, V3(sinThetaM * std::cos(phi), sinThetaM * std::sin(phi), cosThetaM));
        const double wiM = dot(wi, m);
        wo = m * (2.0 * wiM) - wi;
        const double cosWo = dot(cur.n, wo);
        if (cosWo <= 0) return false;
        const double woM = dot(wo, m);
        const double D = conductorD(pm.distribution, alpha, cosThetaM);
        const double G = conductorG1(pm.distribution, alpha, cosWi, wiM) * conductorG1(pm.distribution, alpha, cosWo, woM);
        const double wgt = D * G * wiM / (pdfM * cosWi);
        cur.weight = V3(conductorFresnel(wiM, pm.eta.x, pm.k.x) * pm.spec.x, conductorFresnel(wiM, pm.eta.y, pm.k.y) * pm.spec.y,
                        conductorFresnel(wiM, pm.eta.z, pm.k.z) * pm.spec.z) * wgt;
        cur.pdf = pdfM / (4.0 * std::fabs(woM));
        cur.comp = 0x00008u;  // EGlossyReflection
        if (maxc(cur.weight) <= 0 || !(cur.pdf > 0)) return false;
#endif
      } else {
        V3 local = cosineHemisphere(a, b);
        wo = toWorld(cur.n, local);
        cur.weight = cur.albedo;
        cur.pdf = local.z * kInvPi;
        if (local.z <= 0 || maxc(cur.weight) <= 0) return false;
      }
    } else {  // medium
      V3 wi = normalize(path[i - 1].pos - cur.pos);
      double a = rng.next1D(), b = rng.next1D();
      if (std::fabs(g) < kEpsilon) {
        wo = uniformSphere(a, b);
        cur.pdf = kInvFourPi;
      } else {
        double sqrTerm = (1 - g * g) / (1 - g + 2 * g * a);
        double cosTheta = (1 + g * g - sqrTerm * sqrTerm) / (2 * g);
        double sinTheta = std::sqrt(std::fmax(0.0, 1.0 - cosTheta * cosTheta));
        double phi = 2 * kPi * b;
        wo = toWorld(-wi, V3(sinTheta * std::cos(phi), sinTheta * std::sin(phi), cosTheta));
        cur.pdf = hgEval(g, dot(wi, wo));
      }
      cur.weight = sigS;  // sigmaS * phase->sample() (= 1)
      mint = 0.0;
    }
    throughput = throughput * cur.weight;
    cur.rr = 1.0;
    if (sc.rrDepth != -1 && i >= sc.rrDepth) {
      double q = std::fmin(maxc(throughput), 0.95);
      if (rng.next1D() > q) return false;
      cur.rr = 1.0 / q;
      throughput = throughput * cur.rr;
    }
    // PathEdge::sampleNext, src/libbidir/edge.cpp:27-84 (short beams)
    Hit hit;
    bool surface = closestHit(sc, cur.pos, wo, mint, hit);
    double distSurf = surface ? hit.t : std::numeric_limits<double>::infinity();
    double rand = rng.next1D();
    double sampled = (rand < msw) ? -std::log(1.0 - rand / msw) / sigT
                                  : std::numeric_limits<double>::infinity();
    LVertex succ;
    double len, tr, pdfSuccess, pdfFailure;
    if (sampled < distSurf) {
      succ.type = VT_MEDIUM;
      len = sampled;
      succ.pos = cur.pos + wo * len;
      succ.n = V3(0, 0, 0);
      succ.matKind = -1;
      succ.mat = -1;
      succ.albedo = V3(0.0);
    } else if (surface) {
      succ.type = VT_SURFACE;
      len = hit.t;
      succ.pos = cur.pos + wo * len;
      const SynthTri &tri = sc.tris[hit.tri];
      succ.n = tri.n;
      succ.matKind = sc.mats[tri.mat].kind;
      succ.mat = tri.mat;
      succ.albedo = sc.mats[tri.mat].albedo;
    } else {
      return false;
    }
    succ.comp = succ.matKind == MAT_MIRROR ? 0x00008u : GVPM_BSDF_DIFFUSE_REFLECTION;
    succ.compSel = -1;
    if (len == 0) return false;
    tr = std::exp(-sigT * len);
    pdfSuccess = sigT * tr * msw;
    pdfFailure = tr * msw + (1 - msw);
    if (tr < 1e-20) return false;
    cur.eMedium = true;
    cur.ePdf = succ.type == VT_MEDIUM ? pdfSuccess : pdfFailure;
    cur.eWeight = V3(tr / cur.ePdf);
    throughput = throughput * cur.eWeight;
    if (solidAngle) {
      cur.pdf /= len * len;
      if (succ.type == VT_SURFACE) cur.pdf *= std::fabs(dot(wo, succ.n));
    }
    succ.weight = V3(0.0);
    succ.rr = 1.0;
    succ.pdf = 0.0;
    succ.eWeight = V3(1.0);
    succ.ePdf = 1.0;
    succ.eMedium = false;
    path.push_back(succ);
  }
  return true;
}

// (PATH: LPath, or any container with clear() / push_back(const LVertex &) / operator[] over the LAST FOUR vertices --
// the walk reads path[i - 1], modifies path[i] and appends path[i + 1]: StreamPath below flattens as it goes)
template <class PATH> GVPM_HD inline void randomWalk(const SceneView &sc, Philox &rng, PATH &path) {
  V3 throughput;
  walkBegin(sc, rng, path, throughput);
  for (int i = 1; i < sc.maxDepth; ++i)
    if (!walkStep(sc, rng, path, throughput, i)) break;
}


// isIntersectedPoint, src/integrators/volume_utils.h:154-169
GVPM_HD inline bool cameraHit(const SceneView &sc, V3 org, V3 dest) {
  if (sc.cameraSphere == 0.0) return false;
  V3 beam = dest - org;
  double l2 = dot(beam, beam);
  if (l2 == 0) return false;
  double t = std::fmin(1.0, std::fmax(0.0, dot(sc.camPos - org, beam) / l2));
  V3 v = (org + beam * t) - sc.camPos;
  return sc.cameraSphere * sc.cameraSphere > dot(v, v);
}

// VertexClassifier::type, gvpm/gvpm_struct.h:66-79 (bounceRoughness < inf for Lambertian)
GVPM_HD inline bool vertexIsDiffuse(const SceneView &sc, const LVertex &v) {
  switch (v.type) {
    case VT_EMITTER: return true;
    // (Phong: the Beckmann-equivalent roughness sqrt(2 / (2 + exponent)) of its glossy lobe, phong.cpp:293-300, is far
    // above bounceRoughness = 0.001 for every exponent the both-components branch admits)
    case VT_SURFACE: return v.matKind == MAT_LAMBERT || v.matKind == MAT_PHONG || v.matKind == MAT_ROUGHCONDUCTOR || v.matKind == MAT_WARD;
    case VT_MEDIUM: return !(sc.medium.g > 0.5);
    default: return false;
  }
}

// getTypeShift, gvpm/shift/shift_utilities.h:112-136; returns the 3-bit code of GVPM_PF_SHIFT_TYPE
GVPM_HD inline int typeShift(const SceneView &sc, const LPath &p, size_t c) {
  int b = -1;
  for (size_t i = c - 1; i > 0 && b == -1; --i) b = vertexIsDiffuse(sc, p[i]) ? (int)i : -1;
  if (b == -1) return 0;
  if ((size_t)b + 1 == c) return 1;
  if (p[c - 1].type == VT_MEDIUM) return 2;
  return 3;
}

struct PhotonRec {
  V3 pos, wi, flux, parentPos, parentN, prefixW, parentScat, parentWi, endN;
  float parentPdf, edgePdf, parentRR, parentG;
  uint32_t flags;
};

struct RecList {
  PhotonRec r[GVPM_SYNTH_MAXV];
  int n;
  GVPM_HD RecList() : n(0) {}
  GVPM_HD void clear() { n = 0; }
  GVPM_HD bool empty() const { return n == 0; }
  GVPM_HD void push_back(const PhotonRec &x) {
    if (n < GVPM_SYNTH_MAXV) r[n++] = x;
  }
};


template <class PATH> GVPM_HD inline void fillParent(const SceneView &sc, const PATH &path, size_t ip, PhotonRec &r,
                       uint32_t &ptype) {
  const LVertex &par = path[ip];
  r.parentPos = par.pos;
  r.parentN = par.n;
  r.parentScat = V3(0.0);
  r.parentWi = V3(1.0, 0.0, 0.0);
  ptype = GVPM_PARENT_EMITTER;
  r.parentG = sc.medium.g;
  if (par.type == VT_SURFACE) {
    ptype = GVPM_PARENT_SURFACE;
    r.parentScat = par.albedo;
    r.parentWi = normalize(path[ip - 1].pos - par.pos);
    if (par.matKind == MAT_PHONG || par.matKind == MAT_ROUGHCONDUCTOR || par.matKind == MAT_WARD) {
      ptype = GVPM_PARENT_SURFACE_BSDF;
      r.parentG = (float)(sc.mats[par.mat].bsdf + (par.compSel == 1 ? 1 : 0));  // (the entry of the component the vertex was sampled through)
    }
  } else if (par.type == VT_MEDIUM) {
    ptype = GVPM_PARENT_MEDIUM;
    r.parentScat = V3(sc.medium.sigma_s[0], sc.medium.sigma_s[1], sc.medium.sigma_s[2]);
    r.parentWi = normalize(path[ip - 1].pos - par.pos);
  }
  r.parentPdf = (float)par.pdf;
  r.edgePdf = (float)par.ePdf;
  r.parentRR = (float)par.rr;
}

// LTBeamMap::tryAppendLT + LTPhotonBeam (gvpm/gvpm_beams.h:18-84) without capacity / pathID bookkeeping.
// Returns false when the path had medium edges but all of them were culled by the camera sphere
// (tryAppendLT returns -1: the path is then not counted as shot, gvpm_proc.cpp:330-336).
// (RL: RecList, or any sink with clear() / push_back(const PhotonRec &) / empty() -- the device generator counts or writes
// straight to its output arrays instead of parking up to 16 records of 236 bytes per lane, synth_device.hip)
template <class RL> GVPM_HD inline bool flattenBeams(const SceneView &sc, const LPath &path, RL &recs) {
  recs.clear();
  for (size_t i = 1; i + 1 < path.size(); ++i)
    if (path[i].pdf == 0.0) return true;
  const size_t first = (size_t)(sc.minDepth > 1 ? sc.minDepth : 1);
  V3 w(1.0);
  for (size_t k = 0; k < first && k < path.size(); ++k) w = w * path[k].weight * path[k].rr * path[k].eWeight;
  bool any = false;
  for (size_t i = first; i + 1 < path.size(); ++i) {
    const V3 prefix = w;  // prod_{k<i}
    w = w * path[i].weight * path[i].rr * path[i].eWeight;
    if (!path[i].eMedium) continue;
    any = true;
    if (cameraHit(sc, path[i].pos, path[i + 1].pos)) continue;
    PhotonRec r;
    uint32_t ptype;
    fillParent(sc, path, i, r, ptype);
    r.pos = path[i + 1].pos;
    r.wi = normalize(path[i].pos - path[i + 1].pos);
    r.flux = prefix * path[i].weight * path[i].rr;  // without the transmittance of edge i
    r.prefixW = prefix;
    r.endN = path[i + 1].type == VT_SURFACE ? path[i + 1].n : V3(0.0);
    r.flags = GVPM_PF_MAKE(ptype, typeShift(sc, path, i + 1), 1, i, GVPM_BSDF_DIFFUSE_REFLECTION);
    recs.push_back(r);
  }
  return !(any && recs.empty());
}

// GPhotonMap::tryAppend (gvpm/gvpm_accel.h:119-199) without the capacity / pathID bookkeeping
template <class RL> GVPM_HD inline void flattenPath(const SceneView &sc, const LPath &path, RL &recs) {
  recs.clear();
  const size_t startIndex = (size_t)(sc.minDepth + 1 > 2 ? sc.minDepth + 1 : 2);
  // generatePath(): reject paths with a zero interior pdf (gvpm_proc.cpp:138-143)
  for (size_t i = 1; i + 1 < path.size(); ++i)
    if (path[i].pdf == 0.0) return;
  if (path.size() <= startIndex) return;
  V3 w(1.0);
  for (size_t i = 0; i < startIndex - 1; ++i) w = w * path[i].weight * path[i].rr * path[i].eWeight;
  for (size_t i = startIndex; i < path.size(); ++i) {
    V3 prefix = w;
    w = w * path[i - 1].weight * path[i - 1].rr * path[i - 1].eWeight;
    if (path[i].type != VT_MEDIUM) continue;
    if (cameraHit(sc, path[i - 1].pos, path[i].pos)) continue;
    const LVertex &par = path[i - 1];
    PhotonRec r;
    r.pos = path[i].pos;
    r.wi = normalize(par.pos - path[i].pos);
    r.flux = w;
    r.parentPos = par.pos;
    r.parentN = par.n;
    r.prefixW = prefix;
    r.parentScat = V3(0.0);
    r.parentWi = V3(1.0, 0.0, 0.0);
    uint32_t ptype = GVPM_PARENT_EMITTER, comp = GVPM_BSDF_DIFFUSE_REFLECTION;
    r.parentG = sc.medium.g;
    if (par.type == VT_SURFACE) {
      if (par.matKind == MAT_MIRROR) comp = 0x00008u;  // BSDF::EDeltaReflection
      ptype = GVPM_PARENT_SURFACE;
      r.parentScat = par.albedo;
      r.parentWi = normalize(path[i - 2].pos - par.pos);
      if (par.matKind == MAT_PHONG || par.matKind == MAT_ROUGHCONDUCTOR || par.matKind == MAT_WARD) {
        ptype = GVPM_PARENT_SURFACE_BSDF;
        r.parentG = (float)(sc.mats[par.mat].bsdf + (par.compSel == 1 ? 1 : 0));  // (the entry of the component the vertex was sampled through)
        comp = par.comp;  // the sampled lobe's type: EGlossyReflection or EDiffuseReflection (vertex.cpp:178-179)
      }
    } else if (par.type == VT_MEDIUM) {
      ptype = GVPM_PARENT_MEDIUM;
      r.parentScat = V3(sc.medium.sigma_s[0], sc.medium.sigma_s[1], sc.medium.sigma_s[2]);
      r.parentWi = normalize(path[i - 2].pos - par.pos);
    }
    r.parentPdf = (float)par.pdf;
    r.edgePdf = (float)par.ePdf;
    r.parentRR = (float)par.rr;
    r.flags = GVPM_PF_MAKE(ptype, typeShift(sc, path, i), par.eMedium ? 1 : 0, i - 1, comp);
    recs.push_back(r);
  }
}


// The same two flattenings, done WHILE the path is walked (the device generator, synth_device.hip): vertex idx is flattened
// when it is appended -- everything flattenPath / flattenBeams read of vertices idx - 1 and idx - 2 is final by then -- so the
// walk keeps a ring of four vertices instead of the path's sixteen (2.7 KB of scratch per lane, written and read back by
// every light path).  Statement for statement the arithmetic of the functions above: the running product `w` takes its
// factors in the same order, typeShift's backward scan becomes the index of the last diffuse vertex seen, and the
// whole-path rejection (a zero interior pdf) clears the records at the end (finish()).
template <class RL, bool BEAMS> struct StreamPath {
  LVertex v[4];
  int n;
  const SceneView &sc;
  RL &recs;
  V3 w;
  bool bad, any;
  int lastDiffuse;  // largest index in [1, n - 2] whose vertex is diffuse (-1: none): typeShift's b for c = n - 1
  GVPM_HD StreamPath(const SceneView &s_, RL &r_) : n(0), sc(s_), recs(r_), w(1.0), bad(false), any(false), lastDiffuse(-1) {}
  GVPM_HD size_t size() const { return (size_t)n; }
  GVPM_HD LVertex &operator[](size_t i) { return v[i & 3]; }
  GVPM_HD const LVertex &operator[](size_t i) const { return v[i & 3]; }
  GVPM_HD void clear() {
    n = 0;
    w = V3(1.0);
    bad = any = false;
    lastDiffuse = -1;
    recs.clear();
  }
  GVPM_HD int shiftCode(size_t c) const {  // typeShift(sc, path, c)
    const int b = lastDiffuse;
    if (b == -1) return 0;
    if ((size_t)b + 1 == c) return 1;
    if ((*this)[c - 1].type == VT_MEDIUM) return 2;
    return 3;
  }
  GVPM_HD void push_back(const LVertex &x) {
    if (n >= GVPM_SYNTH_MAXV) return;
    const size_t idx = (size_t)n;
    v[idx & 3] = x;
    ++n;
    if (idx == 0) return;
    const StreamPath &path = *this;
    // an interior vertex (it has a successor now) with a zero pdf rejects the whole path
    if (idx >= 2 && path[idx - 1].pdf == 0.0) bad = true;
    const V3 prefix = w;  // prod_{k < idx - 1}
    w = w * path[idx - 1].weight * path[idx - 1].rr * path[idx - 1].eWeight;
    if (BEAMS) {
      // edge i = idx - 1 (flattenBeams)
      const size_t i = idx - 1, first = (size_t)(sc.minDepth > 1 ? sc.minDepth : 1);
      if (i >= first && path[i].eMedium) {
        any = true;
        if (!cameraHit(sc, path[i].pos, path[i + 1].pos)) {
          PhotonRec r;
          uint32_t ptype;
          fillParent(sc, path, i, r, ptype);
          r.pos = path[i + 1].pos;
          r.wi = normalize(path[i].pos - path[i + 1].pos);
          r.flux = prefix * path[i].weight * path[i].rr;  // without the transmittance of edge i
          r.prefixW = prefix;
          r.endN = path[i + 1].type == VT_SURFACE ? path[i + 1].n : V3(0.0);
          r.flags = GVPM_PF_MAKE(ptype, shiftCode(i + 1), 1, i, GVPM_BSDF_DIFFUSE_REFLECTION);
          recs.push_back(r);
        }
      }
    } else {
      // vertex i = idx (flattenPath)
      const size_t i = idx, startIndex = (size_t)(sc.minDepth + 1 > 2 ? sc.minDepth + 1 : 2);
      if (i >= startIndex && path[i].type == VT_MEDIUM && !cameraHit(sc, path[i - 1].pos, path[i].pos)) {
        const LVertex &par = path[i - 1];
        PhotonRec r;
        r.pos = path[i].pos;
        r.wi = normalize(par.pos - path[i].pos);
        r.flux = w;
        r.parentPos = par.pos;
        r.parentN = par.n;
        r.prefixW = prefix;
        r.parentScat = V3(0.0);
        r.parentWi = V3(1.0, 0.0, 0.0);
        uint32_t ptype = GVPM_PARENT_EMITTER, comp = GVPM_BSDF_DIFFUSE_REFLECTION;
        r.parentG = sc.medium.g;
        if (par.type == VT_SURFACE) {
          if (par.matKind == MAT_MIRROR) comp = 0x00008u;  // BSDF::EDeltaReflection
          ptype = GVPM_PARENT_SURFACE;
          r.parentScat = par.albedo;
          r.parentWi = normalize(path[i - 2].pos - par.pos);
          if (par.matKind == MAT_PHONG || par.matKind == MAT_ROUGHCONDUCTOR || par.matKind == MAT_WARD) {
            ptype = GVPM_PARENT_SURFACE_BSDF;
            r.parentG = (float)(sc.mats[par.mat].bsdf + (par.compSel == 1 ? 1 : 0));  // (the entry of the component the vertex was sampled through)
            comp = par.comp;
          }
        } else if (par.type == VT_MEDIUM) {
          ptype = GVPM_PARENT_MEDIUM;
          r.parentScat = V3(sc.medium.sigma_s[0], sc.medium.sigma_s[1], sc.medium.sigma_s[2]);
          r.parentWi = normalize(path[i - 2].pos - par.pos);
        }
        r.parentPdf = (float)par.pdf;
        r.edgePdf = (float)par.ePdf;
        r.parentRR = (float)par.rr;
        r.flags = GVPM_PF_MAKE(ptype, shiftCode(i), par.eMedium ? 1 : 0, i - 1, comp);
        recs.push_back(r);
      }
    }
    if (vertexIsDiffuse(sc, path[idx])) lastDiffuse = (int)idx;
  }
  // after the walk: the whole-path rejection; returns what flattenBeams returns (photons: true)
  GVPM_HD bool finish() {
    if (bad) {
      recs.clear();
      return true;
    }
    return BEAMS ? !(any && recs.empty()) : true;
  }
};

// A camera path of the long-beam walk to the first diffuse vertex (randomWalkFromPixelToFirstDiffuse,
// gvpm_gatherpoint.h:22-170) reduced to what the gather reads: its medium edges (at most two here: a second one behind
// a mirror) with the geometry of their end points.
struct CamEdge {
  V3 o, d, nEnd;   // start point, unit direction (away from the camera), geometric normal at the end point
  double len;
  int matEnd;      // material kind at the end point
  V3 albedoEnd;    // its reflectance
};
struct CamPath {
  bool hasBeam;
  int nEdges;        // medium edges
  int firstEdge;     // path index of the first medium edge: 1 (sensor inside the medium) or 2 (behind the index-matched boundary)
  CamEdge e[2];
  V3 n2;             // normal at vertex 2 (the primary hit), len1 = |v2 - v1|: the sensor's area-measure conversion
  double len1;
  V3 d;              // primary direction
  double pdfDir;     // solid-angle pdf of the primary direction (0 outside the film)
  double rr2;        // rrWeight of the mirror vertex (1 when there is none)
  V3 rho;            // its reflectance
};

GVPM_HD inline double importance(const SceneView &sc, double sx, double sy, V3 d) {
  // PerspectiveCamera::importance, src/sensors/perspective.cpp:191-250
  if (sx < 0 || sy < 0 || sx >= sc.width || sy >= sc.height) return 0.0;
  double tx = sc.tanHalfFovX, ty = tx * sc.height / sc.width;
  double area = (2 * tx) * (2 * ty);
  double cosTheta = -dot(d, sc.camZ);
  if (cosTheta <= 0) return 0.0;
  return 1.0 / (area * cosTheta * cosTheta * cosTheta);
}

// Traces the path through film position (sx, sy).  A base path (`base` == nullptr) walks on behind a mirror (its
// Russian roulette is the caller's: cameraBeamSets); a shifted path copies the base path's decisions and half-vectors
// (ShiftGatherPoint::trace, shift_cameraPath.h:146-413) and stops where the base path stops.
GVPM_HD inline void traceCamera(const SceneView &sc, double sx, double sy, CamPath &cp, const CamPath *base = nullptr) {
  double tx = sc.tanHalfFovX, ty = tx * sc.height / sc.width;
  // (camera space looks along -z; with the identity frame the three sums below are exact)
  const double cx = (2 * sx / sc.width - 1) * tx, cy = (2 * sy / sc.height - 1) * ty;
  V3 d = normalize(sc.camX * cx + sc.camY * cy - sc.camZ);
  cp.d = d;
  cp.pdfDir = importance(sc, sx, sy, d);
  cp.hasBeam = false;
  cp.nEdges = 0;
  cp.firstEdge = sc.cameraInside ? 1 : 2;
  cp.rr2 = base ? base->rr2 : 1.0;  // currInfo->weight *= baseVertex->rrWeight (shift_cameraPath.h:353)
  cp.rho = V3(1.0);
  Hit h;
  if (!closestHit(sc, sc.camPos, d, kEpsilon, h)) return;
  const SynthTri &t2 = sc.tris[h.tri];
  cp.n2 = t2.n;
  cp.len1 = h.t;
  if (sc.cameraInside) {
    // sensor inside the medium: edge 1 (sensor sample -> first surface) is the first medium edge
    if (sc.mats[t2.mat].kind == MAT_NULL) return;
    cp.e[0].o = sc.camPos;
    cp.e[0].d = d;
    cp.e[0].len = h.t;
    cp.e[0].nEnd = t2.n;
    cp.e[0].matEnd = sc.mats[t2.mat].kind;
    cp.e[0].albedoEnd = sc.mats[t2.mat].albedo;
  } else {
    if (sc.mats[t2.mat].kind != MAT_NULL) return;  // did not enter through the medium boundary
    // vertex 2 is a null interaction (an index-matched boundary): the direction is kept (shift_cameraPath.h:292-296)
    const V3 v2 = sc.camPos + d * h.t;
    Hit h3;
    if (!closestHit(sc, v2, d, kEpsilon, h3)) return;
    const SynthTri &t3 = sc.tris[h3.tri];
    if (sc.mats[t3.mat].kind == MAT_NULL) return;
    cp.e[0].o = v2;
    cp.e[0].d = d;
    cp.e[0].len = h3.t;
    cp.e[0].nEnd = t3.n;
    cp.e[0].matEnd = sc.mats[t3.mat].kind;
    cp.e[0].albedoEnd = sc.mats[t3.mat].albedo;
  }
  cp.nEdges = 1;
  cp.hasBeam = true;
  // a mirror at the end of the first medium edge: the walk goes on (one bounce: the scenes hold one mirror wall)
  if (cp.e[0].matEnd != MAT_MIRROR || (base && base->nEdges < 2)) return;
  const CamEdge &e0 = cp.e[0];
  if (dot(e0.nEnd, e0.d) >= 0) return;  // one-sided: seen from behind
  cp.rho = e0.albedoEnd;
  // half-vector copy at a Dirac vertex = the mirror direction (halfVectorShift, shift_utilities.h:94-107: the base
  // half-vector of a mirror is its normal; the Jacobian is forced to 1, shift_cameraPath.h:317-318)
  const V3 wi = -e0.d;
  const V3 wo = e0.nEnd * (2.0 * dot(e0.nEnd, wi)) - wi;
  const V3 v3o = e0.o + e0.d * e0.len;
  Hit h2;
  if (!closestHit(sc, v3o, wo, kEpsilon, h2)) return;
  const SynthTri &tn = sc.tris[h2.tri];
  cp.e[1].o = v3o;
  cp.e[1].d = wo;
  cp.e[1].len = h2.t;
  cp.e[1].nEnd = tn.n;
  cp.e[1].matEnd = sc.mats[tn.mat].kind;
  cp.e[1].albedoEnd = sc.mats[tn.mat].albedo;
  cp.nEdges = 2;
}

// The SVertexPDF cache entries of medium edge k (0 or 1) of a path, as the functors read them (gvpm_struct.h:361-370,
// 523-631): base path = generateVertexInfo, shifted path = ShiftGatherPoint::trace + generate.
GVPM_HD inline void fillRay(const SceneView &sc, gvpm_camera_ray &r, const CamPath &cp, int k, double pdfSensorArea, double jac,
                            bool valid) {
  std::memset(&r, 0, sizeof(r));
  const int edge = cp.firstEdge + k;
  if (!valid) {
    r.info = GVPM_RAY_INFO(0, edge);
    return;
  }
  const CamEdge &e = cp.e[k];
  r.o[0] = (float)e.o.x; r.o[1] = (float)e.o.y; r.o[2] = (float)e.o.z;
  r.d[0] = (float)e.d.x; r.d[1] = (float)e.d.y; r.d[2] = (float)e.d.z;
  r.len = (float)e.len;
  // eyeContrib = getWeightBeam(edge - 1) * getWeightVertex(edge): 1 on the first medium edge (perspective sensor with
  // importance sampling; a null boundary has weight 1); behind the mirror the transmittance of the edge before it
  // (long beams: the edge weight) times the mirror's vertex weight rho * rrWeight
  V3 eye(1.0);
  if (k == 1) eye = cp.rho * (std::exp(-(double)sc.medium.sigma_t[1] * cp.e[0].len) * cp.rr2);
  r.eye[0] = (float)eye.x; r.eye[1] = (float)eye.y; r.eye[2] = (float)eye.z;
  // info.pdf: the sensor's pdf in area measure at vertex 2; a Dirac vertex multiplies it by 1 (discrete measure)
  r.pdf = (float)pdfSensorArea;
  r.jacobian = (float)jac;
  // GOp(edge) = geometryOpposingTerm(path, edge, edge + 1), gvpm/gvpm_geoOps.h:17-26
  r.gop = (float)(std::fabs(dot(e.nEnd, e.d)) / (e.len * e.len));
  r.info = GVPM_RAY_INFO(1, edge);
}

// The beam sets (base + L R T B, EPixel order) of pixel (px, py) at `iteration`: one per medium edge of its camera
// path, in path order; 0 when the path has no medium edge or is invalid.  selW[k]: the weight of edge k in the G-VPM
// edge selection (weightBeam.max(), gvpm.cpp:1117-1129).
GVPM_HD inline int cameraBeamSets(const SceneView &sc, int iteration, int px, int py, gvpm_camera_ray out[2][5], float selW[2]) {
  const int offX[4] = {-1, 1, 0, 0}, offY[4] = {0, 0, 1, -1};  // L R T B
  Philox rng(sc.seed, 0xca3eu, (uint32_t)iteration, (uint32_t)(py * sc.width + px));
  double jx = rng.next1D(), jy = rng.next1D();
  float randValue[2];
  randValue[0] = rng.next1D();
  randValue[1] = 0.f;
  CamPath base;
  traceCamera(sc, px + jx, py + jy, base);
  if (!base.hasBeam) return 0;
  if (base.e[0].matEnd == MAT_MIRROR) {
    // (two more draws, only on paths that meet the mirror: the others keep the streams of the mirror-less scenes)
    randValue[1] = rng.next1D();
    const double rr = rng.next1D();
    if (base.nEdges < 2) return 0;  // the walk could not go on: invalid gather point (gvpm_gatherpoint.h:121-124)
    // sampleNext(..., russianRoulette = true, &throughput): the throughput at the mirror is v1's weight (1) * the
    // edge weight (long beam: the transmittance) * the mirror's weight; q = min(max, 0.95)
    const double thr = std::exp(-(double)sc.medium.sigma_t[1] * base.e[0].len) * maxc(base.rho);
    const double q = std::fmin(thr, 0.95);
    if (rr > q) return 0;
    base.rr2 = 1.0 / q;
  }
  // base SVertexPDF (generateVertexInfo): pdf = pdfDir converted to area at vertex 2 (vertex.cpp:403-408); jacobian = 1
  const double gopBase12 = std::fabs(dot(base.n2, base.d)) / (base.len1 * base.len1);
  for (int k = 0; k < base.nEdges; ++k) {
    fillRay(sc, out[k][0], base, k, base.pdfDir * gopBase12, 1.0, true);
    out[k][0].rand = randValue[k];
    out[k][0].pixel = (uint32_t)px | ((uint32_t)py << 16);
  }
  selW[0] = 1.f;
  // weightBeam *= vertex(2).weight[EImportance] * edge(1).weight[EImportance]: no rrWeight (the FIXME at gvpm.cpp:1127)
  selW[1] = base.nEdges == 2 ? (float)(maxc(base.rho) * std::exp(-(double)sc.medium.sigma_t[1] * base.e[0].len)) : 0.f;
  for (int i = 0; i < 4; ++i) {
    CamPath sh;
    traceCamera(sc, px + offX[i] + jx, py + offY[i] + jy, sh, &base);
    double pdf = 0, jac = 0;
    if (sh.hasBeam) {
      // ShiftGatherPoint::trace/generate, shift_cameraPath.h:76-116,191-242
      double pdf1 = base.pdfDir, pdf2 = sh.pdfDir;
      double gopNew12 = std::fabs(dot(sh.n2, sh.d)) / (sh.len1 * sh.len1);
      pdf = (pdf2 == 0.0 ? pdf1 : pdf2) * gopNew12;
      jac = (pdf2 == 0.0 ? 1.0 : pdf1 / pdf2) * (gopBase12 / gopNew12);
    }
    for (int k = 0; k < base.nEdges; ++k) {
      gvpm_camera_ray &r = out[k][1 + i];
      // validVolumeEdge(edge): the shifted path reached this edge (k = 1: its vertex 2 is a mirror too and the
      // reflected ray hit something; a Dirac vertex leaves pdf and Jacobian as they are, shift_cameraPath.h:60-116)
      fillRay(sc, r, sh, k, pdf, jac, sh.hasBeam && k < sh.nEdges);
      r.pixel = 0;  // base ray only (fillRay zeroes rand too)
    }
  }
  return base.nEdges;
}

// single-edge form (scenes without a mirror)
GVPM_HD inline bool cameraBeamSet(const SceneView &sc, int iteration, int px, int py, gvpm_camera_ray out[5]) {
  gvpm_camera_ray sets[2][5];
  float selW[2];
  const int n = cameraBeamSets(sc, iteration, px, py, sets, selW);
  if (n < 1) return false;
  for (int k = 0; k < 5; ++k) out[k] = sets[0][k];
  return true;
}

}  // namespace gvpm
