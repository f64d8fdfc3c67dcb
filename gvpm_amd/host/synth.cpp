#include "synth.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>

#include "philox.h"

namespace gvpm {

static const double kPi = 3.14159265358979323846;
static const double kInvPi = 1.0 / kPi;
static const double kInvFourPi = 1.0 / (4.0 * kPi);
static const double kEpsilon = 1e-4;  // Epsilon, single-precision build (constants.h:24-31)

// ------------------------------------------------------------------ scenes --
double SynthScene::bsphereRadius() const {
  // AABB::getBSphere(): centre = box centre, radius = |max - centre|
  V3 c = (bmin + bmax) * 0.5;
  return length(bmax - c);
}

void SynthScene::addQuad(V3 a, V3 b, V3 c, V3 d, int mat) {
  SynthTri t1, t2;
  t1.v0 = a; t1.e1 = b - a; t1.e2 = c - a; t1.n = normalize(cross(t1.e1, t1.e2)); t1.mat = mat;
  t2.v0 = a; t2.e1 = c - a; t2.e2 = d - a; t2.n = normalize(cross(t2.e1, t2.e2)); t2.mat = mat;
  tris.push_back(t1);
  tris.push_back(t2);
}

static void addBoxRoom(SynthScene &s, int matFloor, int matCeil, int matBack, int matLeft,
                       int matRight, int matFront) {
  // inward-facing quads of the room [-1,1]^3
  s.addQuad(V3(-1, -1, -1), V3(-1, -1, 1), V3(1, -1, 1), V3(1, -1, -1), matFloor);   // n=(0,1,0)
  s.addQuad(V3(-1, 1, -1), V3(1, 1, -1), V3(1, 1, 1), V3(-1, 1, 1), matCeil);        // n=(0,-1,0)
  s.addQuad(V3(-1, -1, -1), V3(1, -1, -1), V3(1, 1, -1), V3(-1, 1, -1), matBack);    // n=(0,0,1)
  s.addQuad(V3(-1, -1, -1), V3(-1, 1, -1), V3(-1, 1, 1), V3(-1, -1, 1), matLeft);    // n=(1,0,0)
  s.addQuad(V3(1, -1, -1), V3(1, -1, 1), V3(1, 1, 1), V3(1, 1, -1), matRight);       // n=(-1,0,0)
  s.addQuad(V3(-1, -1, 1), V3(-1, 1, 1), V3(1, 1, 1), V3(1, -1, 1), matFront);       // n=(0,0,-1)
}

static void setLight(SynthScene &s, V3 c, double sx, double sz, V3 radiance, int mat) {
  s.lightC = c;
  s.lightU = V3(sx, 0, 0);
  s.lightV = V3(0, 0, sz);
  s.lightN = V3(0, -1, 0);
  s.radiance = radiance;
  s.lightArea = sx * sz;
  V3 a = c - s.lightU * 0.5 - s.lightV * 0.5;
  V3 b = c + s.lightU * 0.5 - s.lightV * 0.5;
  V3 cc = c + s.lightU * 0.5 + s.lightV * 0.5;
  V3 d = c - s.lightU * 0.5 + s.lightV * 0.5;
  s.addQuad(a, b, cc, d, mat);  // normal (0,-1,0)
}

static void setMedium(SynthScene &s, double sigmaS, double sigmaA, double g) {
  for (int i = 0; i < 3; ++i) {
    s.medium.sigma_s[i] = (float)sigmaS;
    s.medium.sigma_a[i] = (float)sigmaA;
    s.medium.sigma_t[i] = (float)(sigmaS + sigmaA);
  }
  s.medium.g = (float)g;
  s.medium.medium_sampling_weight = 1.f;  // computeOnlyVolumeInteraction(), gvpm.cpp:135-142
  std::memset(s.medium.reserved, 0, sizeof(s.medium.reserved));
}

static void addInnerBox(SynthScene &s, V3 lo, V3 hi, int mat) {
  // outward-facing quads of a solid box
  s.addQuad(V3(lo.x, lo.y, lo.z), V3(hi.x, lo.y, lo.z), V3(hi.x, lo.y, hi.z), V3(lo.x, lo.y, hi.z), mat);  // -y
  s.addQuad(V3(lo.x, hi.y, lo.z), V3(lo.x, hi.y, hi.z), V3(hi.x, hi.y, hi.z), V3(hi.x, hi.y, lo.z), mat);  // +y
  s.addQuad(V3(lo.x, lo.y, lo.z), V3(lo.x, hi.y, lo.z), V3(hi.x, hi.y, lo.z), V3(hi.x, lo.y, lo.z), mat);  // -z
  s.addQuad(V3(lo.x, lo.y, hi.z), V3(hi.x, lo.y, hi.z), V3(hi.x, hi.y, hi.z), V3(lo.x, hi.y, hi.z), mat);  // +z
  s.addQuad(V3(lo.x, lo.y, lo.z), V3(lo.x, lo.y, hi.z), V3(lo.x, hi.y, hi.z), V3(lo.x, hi.y, lo.z), mat);  // -x
  s.addQuad(V3(hi.x, lo.y, lo.z), V3(hi.x, hi.y, lo.z), V3(hi.x, hi.y, hi.z), V3(hi.x, lo.y, hi.z), mat);  // +x
}

bool makeScene(const std::string &name, int width, int height, uint32_t seed, SynthScene &s) {
  s = SynthScene();
  s.name = name;
  s.width = width;
  s.height = height;
  s.seed = seed;
  s.bmin = V3(-1, -1, -1);
  s.bmax = V3(1, 1, 1);
  s.camPos = V3(0, 0, 3.9);
  s.tanHalfFovX = std::tan(0.5 * 39.0 * kPi / 180.0);
  s.cameraInside = false;
  s.maxDepth = 12;  // scripts/scene/generatorGVPM.py:39-85 paper settings
  s.rrDepth = 1;
  s.minDepth = 0;
  s.mats.push_back({MAT_LAMBERT, V3(0.5, 0.5, 0.5)});       // 0 white
  s.mats.push_back({MAT_LAMBERT, V3(0.63, 0.065, 0.05)});    // 1 red (left)
  s.mats.push_back({MAT_LAMBERT, V3(0.14, 0.45, 0.091)});    // 2 green (right)
  s.mats.push_back({MAT_NULL, V3(0, 0, 0)});                 // 3 medium boundary (front)
  if (name == "cbox") {
    addBoxRoom(s, 0, 0, 0, 1, 2, 3);
    setLight(s, V3(0, 0.998, 0), 0.5, 0.5, V3(15, 15, 15), 0);
    setMedium(s, 0.5, 0.5, 0.0);
  } else if (name == "cbox_hg") {
    // same room, forward-scattering fog: exercises the EMediumShift -> diffuse
    // reconnection branch (g > 0.5 makes medium vertices "glossy",
    // gvpm_struct.h:73-76)
    addBoxRoom(s, 0, 0, 0, 1, 2, 3);
    setLight(s, V3(0, 0.998, 0), 0.5, 0.5, V3(15, 15, 15), 0);
    setMedium(s, 0.5, 0.5, 0.7);
  } else if (name == "cbox_in") {
    // closed room, sensor inside the fog (what the plane estimator requires, gvpm.cpp:784-788)
    addBoxRoom(s, 0, 0, 0, 1, 2, 0);
    setLight(s, V3(0, 0.998, 0), 0.5, 0.5, V3(15, 15, 15), 0);
    setMedium(s, 0.5, 0.5, 0.0);
    s.camPos = V3(0, 0, 0.95);
    s.tanHalfFovX = std::tan(0.5 * 60.0 * kPi / 180.0);
    s.cameraInside = true;
  } else if (name == "laser") {
    // S-laser: sigma_t = 0.5, small emitter behind an aperture plate
    addBoxRoom(s, 0, 0, 0, 1, 2, 3);
    setLight(s, V3(0, 0.998, 0), 0.02, 0.02, V3(15000, 15000, 15000), 0);
    // aperture plate at y = 0.9 with a 0.05 x 0.05 hole: four quads facing up and down
    double h = 0.025, y = 0.9;
    V3 lo[4] = {V3(-1, y, -1), V3(-1, y, h), V3(-1, y, -h), V3(h, y, -h)};
    V3 hi[4] = {V3(1, y, -h), V3(1, y, 1), V3(-h, y, h), V3(1, y, h)};
    for (int k = 0; k < 4; ++k) {
      // facing down (visible from the room)
      s.addQuad(V3(lo[k].x, y, lo[k].z), V3(hi[k].x, y, lo[k].z), V3(hi[k].x, y, hi[k].z),
                V3(lo[k].x, y, hi[k].z), 0);
    }
    setMedium(s, 0.25, 0.25, 0.0);
  } else if (name == "fogroom") {
    addBoxRoom(s, 0, 0, 0, 1, 2, 3);
    setLight(s, V3(0, 0.998, 0), 0.5, 0.5, V3(15, 15, 15), 0);
    Philox rng(seed, 0x0b0c5u, 0, 0);
    for (int k = 0; k < 64; ++k) {
      double cx = -0.85 + 1.7 * rng.next1D(), cz = -0.85 + 1.7 * rng.next1D();
      double hx = 0.04 + 0.06 * rng.next1D(), hz = 0.04 + 0.06 * rng.next1D();
      double hy = 0.1 + 0.5 * rng.next1D();
      addInnerBox(s, V3(cx - hx, -1.0, cz - hz), V3(cx + hx, -1.0 + hy, cz + hz), 0);
    }
    setMedium(s, 0.5, 0.5, 0.0);
  } else {
    return false;
  }
  // gvpm.cpp:162 m_config.cameraSphere = R * cameraSphere(=1) * POURCENTAGE_BS
  s.cameraSphere = s.bsphereRadius() * 1.0 * 0.01;
  return true;
}

void defaultParams(const SynthScene &sc, gvpm_params &p) {
  std::memset(&p, 0, sizeof(p));
  p.abi_version = GVPM_ABI_VERSION;
  p.width = sc.width;
  p.height = sc.height;
  p.vol_technique = GVPM_VOL_BRE3D;
  p.max_depth = sc.maxDepth;
  p.min_depth = sc.minDepth;
  p.use_mis = 1;
  p.use_shift_null = 1;
  p.path_set = 1;
  p.power_heuristic = 0;
  p.no_medium_shift = 1;
  p.use_manifold = 0;
  p.debug_shift = GVPM_SHIFT_ALL;
  p.lighting_interaction_mode = GVPM_SURF2MEDIA | GVPM_MEDIA2MEDIA;  // all2media
  p.bsdf_interaction_mode = GVPM_BSDF_ALL;
  p.nb_camera_samples = 40;
  p.visibility_as_written = 1;
  p.alpha = 0.7f;
  p.initial_scale_volume = 1.0f;
  p.bsphere_radius = (float)sc.bsphereRadius();
  p.epsilon = 1e-4f;
  p.shadow_epsilon = 1e-3f;
}

// ------------------------------------------------------------ ray casting --
struct Hit {
  double t;
  int tri;
};

// closest intersection with any triangle, t in (mint, inf); two-sided like
// Mitsuba's triangle kd-tree
static bool closestHit(const SynthScene &sc, V3 o, V3 d, double mint, Hit &hit) {
  hit.t = std::numeric_limits<double>::infinity();
  hit.tri = -1;
  for (size_t i = 0; i < sc.tris.size(); ++i) {
    const SynthTri &tr = sc.tris[i];
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
};

static V3 cosineHemisphere(double u1, double u2) {
  // warp::squareToCosineHemisphere (concentric disk not required for parity:
  // the synthetic host only has to be a valid sampler with the stated pdf)
  double r = std::sqrt(u1), phi = 2.0 * kPi * u2;
  double x = r * std::cos(phi), y = r * std::sin(phi);
  return V3(x, y, std::sqrt(std::fmax(0.0, 1.0 - u1)));
}

static V3 uniformSphere(double u1, double u2) {
  double z = 1.0 - 2.0 * u1;
  double r = std::sqrt(std::fmax(0.0, 1.0 - z * z)), phi = 2.0 * kPi * u2;
  return V3(r * std::cos(phi), r * std::sin(phi), z);
}

static V3 toWorld(V3 n, V3 local) {
  V3 s, t;
  coordinateSystem(n, s, t);
  return s * local.x + t * local.y + n * local.z;
}

static double hgEval(double g, double cosWiWo) {
  // HGPhaseFunction::eval, src/phase/hg.cpp:107-110 (wi points away from the vertex)
  double temp = 1.0 + g * g + 2.0 * g * cosWiWo;
  return kInvFourPi * (1 - g * g) / (temp * std::sqrt(temp));
}

// One light path; mirrors Path::randomWalk(scene, sampler, maxDepth, rrDepth, EImportance)
static void randomWalk(const SynthScene &sc, Philox &rng, std::vector<LVertex> &path) {
  path.clear();
  const double sigT = sc.medium.sigma_t[1];
  const double msw = sc.medium.medium_sampling_weight;
  const V3 sigS(sc.medium.sigma_s[0], sc.medium.sigma_s[1], sc.medium.sigma_s[2]);
  const double g = sc.medium.g;

  LVertex v0;
  v0.type = VT_SUPERNODE;
  v0.weight = sc.radiance * (kPi * sc.lightArea);  // AreaLight::samplePosition -> m_power
  v0.pdf = 1.0 / sc.lightArea;
  v0.rr = 1.0;
  v0.eWeight = V3(1.0);
  v0.ePdf = 1.0;
  v0.eMedium = true;
  v0.matKind = -1;
  double u1 = rng.next1D(), u2 = rng.next1D();
  LVertex v1;
  v1.type = VT_EMITTER;
  v1.pos = sc.lightC + sc.lightU * (u1 - 0.5) + sc.lightV * (u2 - 0.5);
  v1.n = sc.lightN;
  v1.matKind = -1;
  path.push_back(v0);
  path.push_back(v1);

  V3 throughput(1.0);  // the supernode case returns before `throughput *= weight`
  for (int i = 1; i < sc.maxDepth; ++i) {
    LVertex &cur = path[i];
    V3 wo;
    double mint = kEpsilon;
    bool solidAngle = true;
    if (cur.type == VT_EMITTER) {
      V3 local = cosineHemisphere(rng.next1D(), rng.next1D());
      wo = toWorld(cur.n, local);
      cur.weight = V3(1.0);
      cur.pdf = local.z * kInvPi;
      if (cur.pdf <= 0) break;
    } else if (cur.type == VT_SURFACE) {
      V3 wi = normalize(path[i - 1].pos - cur.pos);
      double a = rng.next1D(), b = rng.next1D();
      if (cur.matKind == MAT_NULL) {
        // index-matched medium boundary: passes straight through and leaves the fog
        break;
      }
      if (dot(cur.n, wi) <= 0) break;  // one-sided diffuse BSDF
      V3 local = cosineHemisphere(a, b);
      wo = toWorld(cur.n, local);
      cur.weight = cur.albedo;
      cur.pdf = local.z * kInvPi;
      if (local.z <= 0 || maxc(cur.weight) <= 0) break;
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
      if (rng.next1D() > q) break;
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
      succ.albedo = V3(0.0);
    } else if (surface) {
      succ.type = VT_SURFACE;
      len = hit.t;
      succ.pos = cur.pos + wo * len;
      const SynthTri &tri = sc.tris[hit.tri];
      succ.n = tri.n;
      succ.matKind = sc.mats[tri.mat].kind;
      succ.albedo = sc.mats[tri.mat].albedo;
    } else {
      break;
    }
    if (len == 0) break;
    tr = std::exp(-sigT * len);
    pdfSuccess = sigT * tr * msw;
    pdfFailure = tr * msw + (1 - msw);
    if (tr < 1e-20) break;
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
}

void PhotonBuffers::clear() {
  pos.clear(); wi.clear(); flux.clear(); parent_pos.clear(); parent_n.clear();
  prefix_w.clear(); parent_scat.clear(); parent_wi.clear();
  parent_pdf.clear(); edge_pdf.clear(); parent_rr.clear(); parent_g.clear();
  flags.clear(); path_id.clear();
  n = 0;
}

void PhotonBuffers::view(gvpm_photon_soa &o) const {
  o.pos = pos.data(); o.wi = wi.data(); o.flux = flux.data();
  o.parent_pos = parent_pos.data(); o.parent_n = parent_n.data();
  o.prefix_w = prefix_w.data(); o.parent_scat = parent_scat.data();
  o.parent_wi = parent_wi.data(); o.parent_pdf = parent_pdf.data();
  o.edge_pdf = edge_pdf.data(); o.parent_rr = parent_rr.data();
  o.parent_g = parent_g.data(); o.flags = flags.data(); o.path_id = path_id.data();
  o.n = n;
}

static inline void push3(std::vector<float> &v, V3 a) {
  v.push_back((float)a.x);
  v.push_back((float)a.y);
  v.push_back((float)a.z);
}

// isIntersectedPoint, src/integrators/volume_utils.h:154-169
static bool cameraHit(const SynthScene &sc, V3 org, V3 dest) {
  if (sc.cameraSphere == 0.0) return false;
  V3 beam = dest - org;
  double l2 = dot(beam, beam);
  if (l2 == 0) return false;
  double t = std::fmin(1.0, std::fmax(0.0, dot(sc.camPos - org, beam) / l2));
  V3 v = (org + beam * t) - sc.camPos;
  return sc.cameraSphere * sc.cameraSphere > dot(v, v);
}

// VertexClassifier::type, gvpm/gvpm_struct.h:66-79 (bounceRoughness < inf for Lambertian)
static bool vertexIsDiffuse(const SynthScene &sc, const LVertex &v) {
  switch (v.type) {
    case VT_EMITTER: return true;
    case VT_SURFACE: return v.matKind == MAT_LAMBERT;
    case VT_MEDIUM: return !(sc.medium.g > 0.5);
    default: return false;
  }
}

// getTypeShift, gvpm/shift/shift_utilities.h:112-136; returns the 3-bit code of GVPM_PF_SHIFT_TYPE
static int typeShift(const SynthScene &sc, const std::vector<LVertex> &p, size_t c) {
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

static void fillParent(const SynthScene &sc, const std::vector<LVertex> &path, size_t ip, PhotonRec &r,
                       uint32_t &ptype) {
  const LVertex &par = path[ip];
  r.parentPos = par.pos;
  r.parentN = par.n;
  r.parentScat = V3(0.0);
  r.parentWi = V3(1.0, 0.0, 0.0);
  ptype = GVPM_PARENT_EMITTER;
  if (par.type == VT_SURFACE) {
    ptype = GVPM_PARENT_SURFACE;
    r.parentScat = par.albedo;
    r.parentWi = normalize(path[ip - 1].pos - par.pos);
  } else if (par.type == VT_MEDIUM) {
    ptype = GVPM_PARENT_MEDIUM;
    r.parentScat = V3(sc.medium.sigma_s[0], sc.medium.sigma_s[1], sc.medium.sigma_s[2]);
    r.parentWi = normalize(path[ip - 1].pos - par.pos);
  }
  r.parentPdf = (float)par.pdf;
  r.edgePdf = (float)par.ePdf;
  r.parentRR = (float)par.rr;
  r.parentG = sc.medium.g;
}

// LTBeamMap::tryAppendLT + LTPhotonBeam (gvpm/gvpm_beams.h:18-84) without capacity / pathID bookkeeping.
// Returns false when the path had medium edges but all of them were culled by the camera sphere
// (tryAppendLT returns -1: the path is then not counted as shot, gvpm_proc.cpp:330-336).
static bool flattenBeams(const SynthScene &sc, const std::vector<LVertex> &path, std::vector<PhotonRec> &recs) {
  recs.clear();
  for (size_t i = 1; i + 1 < path.size(); ++i)
    if (path[i].pdf == 0.0) return true;
  const size_t first = (size_t)std::max(sc.minDepth, 1);
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
static void flattenPath(const SynthScene &sc, const std::vector<LVertex> &path, std::vector<PhotonRec> &recs) {
  recs.clear();
  const size_t startIndex = (size_t)std::max(2, sc.minDepth + 1);
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
    if (par.type == VT_SURFACE) {
      ptype = GVPM_PARENT_SURFACE;
      r.parentScat = par.albedo;
      r.parentWi = normalize(path[i - 2].pos - par.pos);
    } else if (par.type == VT_MEDIUM) {
      ptype = GVPM_PARENT_MEDIUM;
      r.parentScat = V3(sc.medium.sigma_s[0], sc.medium.sigma_s[1], sc.medium.sigma_s[2]);
      r.parentWi = normalize(path[i - 2].pos - par.pos);
    }
    r.parentPdf = (float)par.pdf;
    r.edgePdf = (float)par.ePdf;
    r.parentRR = (float)par.rr;
    r.parentG = sc.medium.g;
    r.flags = GVPM_PF_MAKE(ptype, typeShift(sc, path, i), par.eMedium ? 1 : 0, i - 1, comp);
    recs.push_back(r);
  }
}

static uint64_t shootCommon(const SynthScene &sc, int iteration, uint64_t capacity, PhotonBuffers &out, bool beams,
                            std::vector<float> *endN);

uint64_t shootPhotons(const SynthScene &sc, int iteration, uint64_t capacity, PhotonBuffers &out) {
  return shootCommon(sc, iteration, capacity, out, false, nullptr);
}

uint64_t shootBeams(const SynthScene &sc, int iteration, uint64_t capacity, PhotonBuffers &out,
                    std::vector<float> &endN) {
  return shootCommon(sc, iteration, capacity, out, true, &endN);
}

static uint64_t shootCommon(const SynthScene &sc, int iteration, uint64_t capacity, PhotonBuffers &out, bool beams,
                            std::vector<float> *endN) {
  out.clear();
  if (endN) endN->clear();
  uint64_t nbPaths = 0;
  uint32_t nbLightPathAdded = 0;
  // Paths are keyed by their index, so chunks of them can be generated by worker threads and
  // appended in index order: the result is identical to the sequential loop
  // (`deterministic` mode of the reference, gvpm.cpp:399-409).
  const uint64_t CH = 8192;
  unsigned nthreads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  std::vector<std::vector<PhotonRec>> chunk(CH);
  std::vector<uint8_t> counted(CH, 1);
  uint64_t nextIndex = 0;  // index of the next light path (keys its random stream)
  while (out.n < capacity) {
    const uint64_t base = nextIndex;
    auto worker = [&](unsigned tid) {
      std::vector<LVertex> path;
      for (uint64_t k = tid; k < CH; k += nthreads) {
        counted[k] = 1;
        const uint64_t idx = base + k;
        Philox rng(sc.seed, 0x11ffu, (uint32_t)iteration, (uint32_t)idx, (uint32_t)(idx >> 32));
        randomWalk(sc, rng, path);
        if (beams) counted[k] = flattenBeams(sc, path, chunk[k]) ? 1 : 0;
        else flattenPath(sc, path, chunk[k]);
      }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nthreads; ++t) th.emplace_back(worker, t);
    worker(0);
    for (auto &t : th) t.join();
    for (uint64_t k = 0; k < CH && out.n < capacity; ++k) {
      // paths that store nothing still count as shot (pushVolumeLT(nullptr), gvpm_proc.cpp:302-307)
      nextIndex++;
      if (counted[k]) nbPaths++;
      int nbAppend = 0;
      for (const PhotonRec &r : chunk[k]) {
        if (out.n >= capacity) continue;
        push3(out.pos, r.pos); push3(out.wi, r.wi); push3(out.flux, r.flux);
        push3(out.parent_pos, r.parentPos); push3(out.parent_n, r.parentN);
        push3(out.prefix_w, r.prefixW); push3(out.parent_scat, r.parentScat);
        push3(out.parent_wi, r.parentWi);
        out.parent_pdf.push_back(r.parentPdf);
        out.edge_pdf.push_back(r.edgePdf);
        out.parent_rr.push_back(r.parentRR);
        out.parent_g.push_back(r.parentG);
        out.flags.push_back(r.flags);
        out.path_id.push_back(nbLightPathAdded);
        if (endN) push3(*endN, r.endN);
        out.n++;
        nbAppend++;
      }
      if (nbAppend != 0) nbLightPathAdded++;
    }
  }
  return nbPaths;
}

// ----------------------------------------------------------- camera beams --
struct CamPath {
  bool hasBeam;
  V3 v2, v3, d, n2, n3;
  double len1, len2;
  double pdfDir;  // solid-angle pdf of the primary direction (0 outside the film)
};

static double importance(const SynthScene &sc, double sx, double sy, V3 d) {
  // PerspectiveCamera::importance, src/sensors/perspective.cpp:191-250
  if (sx < 0 || sy < 0 || sx >= sc.width || sy >= sc.height) return 0.0;
  double tx = sc.tanHalfFovX, ty = tx * sc.height / sc.width;
  double area = (2 * tx) * (2 * ty);
  double cosTheta = -d.z;
  if (cosTheta <= 0) return 0.0;
  return 1.0 / (area * cosTheta * cosTheta * cosTheta);
}

static void traceCamera(const SynthScene &sc, double sx, double sy, CamPath &cp) {
  double tx = sc.tanHalfFovX, ty = tx * sc.height / sc.width;
  V3 d = normalize(V3((2 * sx / sc.width - 1) * tx, (2 * sy / sc.height - 1) * ty, -1.0));
  cp.d = d;
  cp.pdfDir = importance(sc, sx, sy, d);
  cp.hasBeam = false;
  Hit h;
  if (!closestHit(sc, sc.camPos, d, kEpsilon, h)) return;
  const SynthTri &t2 = sc.tris[h.tri];
  if (sc.cameraInside) {
    // sensor inside the medium: edge 1 (sensor sample -> first surface) is the medium edge
    if (sc.mats[t2.mat].kind == MAT_NULL) return;
    cp.v2 = sc.camPos;
    cp.v3 = sc.camPos + d * h.t;
    cp.n2 = cp.n3 = t2.n;
    cp.len1 = cp.len2 = h.t;
    cp.hasBeam = true;
    return;
  }
  if (sc.mats[t2.mat].kind != MAT_NULL) return;  // did not enter through the medium boundary
  cp.v2 = sc.camPos + d * h.t;
  cp.n2 = t2.n;
  cp.len1 = h.t;
  Hit h3;
  if (!closestHit(sc, cp.v2, d, kEpsilon, h3)) return;
  const SynthTri &t3 = sc.tris[h3.tri];
  if (sc.mats[t3.mat].kind == MAT_NULL) return;
  cp.v3 = cp.v2 + d * h3.t;
  cp.n3 = t3.n;
  cp.len2 = h3.t;
  cp.hasBeam = true;
}

static void fillRay(gvpm_camera_ray &r, const CamPath &cp, double pdf, double jac, bool valid, int edge) {
  std::memset(&r, 0, sizeof(r));
  if (!valid) {
    r.info = GVPM_RAY_INFO(0, edge);
    return;
  }
  r.o[0] = (float)cp.v2.x; r.o[1] = (float)cp.v2.y; r.o[2] = (float)cp.v2.z;
  r.d[0] = (float)cp.d.x; r.d[1] = (float)cp.d.y; r.d[2] = (float)cp.d.z;
  r.len = (float)cp.len2;
  r.eye[0] = r.eye[1] = r.eye[2] = 1.f;
  r.pdf = (float)pdf;
  r.jacobian = (float)jac;
  // GOp(e) = geometryOpposingTerm(path, 2, 3), gvpm/gvpm_geoOps.h:17-26
  r.gop = (float)(std::fabs(dot(cp.n3, cp.d)) / (cp.len2 * cp.len2));
  r.info = GVPM_RAY_INFO(1, edge);
}

void cameraBeams(const SynthScene &sc, int iteration, int x0, int y0, int x1, int y1,
                 std::vector<gvpm_camera_ray> &out, int tileMod, int tileRem) {
  out.clear();
  static const int offX[4] = {-1, 1, 0, 0}, offY[4] = {0, 0, 1, -1};  // L R T B
  const int tilesX = (sc.width + 3) / 4;
  for (int py = y0; py < y1; ++py) {
    for (int px = x0; px < x1; ++px) {
      // image-sharded hosts: 4x4-pixel tiles dealt round-robin to the ranks (an even split of the work)
      if (tileMod > 1 && ((py / 4) * tilesX + px / 4) % tileMod != tileRem) continue;
      Philox rng(sc.seed, 0xca3eu, (uint32_t)iteration, (uint32_t)(py * sc.width + px));
      double jx = rng.next1D(), jy = rng.next1D();
      float randValue = rng.next1D();
      CamPath base;
      traceCamera(sc, px + jx, py + jy, base);
      if (!base.hasBeam) continue;
      // base SVertexPDF (generateVertexInfo): pdf = pdfDir converted to area at
      // vertex 2 (vertex.cpp:403-408); jacobian = 1
      double gopBase12 = std::fabs(dot(base.n2, base.d)) / (base.len1 * base.len1);
      gvpm_camera_ray r;
      const int edge = sc.cameraInside ? 1 : 2;
      fillRay(r, base, base.pdfDir * gopBase12, 1.0, true, edge);
      r.rand = randValue;
      r.pixel = (uint32_t)px | ((uint32_t)py << 16);
      out.push_back(r);
      for (int k = 0; k < 4; ++k) {
        CamPath sh;
        traceCamera(sc, px + offX[k] + jx, py + offY[k] + jy, sh);
        if (!sh.hasBeam) {
          fillRay(r, sh, 0, 0, false, edge);
        } else {
          // ShiftGatherPoint::trace/generate, shift_cameraPath.h:76-116,191-242
          double pdf1 = base.pdfDir, pdf2 = sh.pdfDir;
          double gopNew12 = std::fabs(dot(sh.n2, sh.d)) / (sh.len1 * sh.len1);
          double pdf = (pdf2 == 0.0 ? pdf1 : pdf2) * gopNew12;
          double jac = (pdf2 == 0.0 ? 1.0 : pdf1 / pdf2) * (gopBase12 / gopNew12);
          fillRay(r, sh, pdf, jac, true, edge);
        }
        r.pixel = 0;  // base ray only
        out.push_back(r);
      }
    }
  }
}

void cameraSamplesVPM(const SynthScene &sc, int iteration, const std::vector<gvpm_camera_ray> &rays,
                      int nbCameraSamples, std::vector<gvpm_vpm_sample> &out) {
  out.clear();
  const size_t nsets = rays.size() / 5;
  for (size_t s = 0; s < nsets; ++s) {
    const uint32_t pix = rays[5 * s].pixel;
    const uint32_t idx = (pix >> 16) * (uint32_t)sc.width + (pix & 0xFFFFu);
    Philox rng(sc.seed, 0x5a3fu, (uint32_t)iteration, idx);
    for (int k = 0; k < nbCameraSamples; ++k) {
      gvpm_vpm_sample sm;
      sm.set = (uint32_t)s;
      sm.rand = rng.next1D();  // sampleReuse over a single-entry CDF leaves the sample unchanged
      sm.pdf_sel = 1.f;
      sm.reserved = 0;
      out.push_back(sm);
    }
  }
}

// LTPhotonPlane::transformBeam, gvpm/gvpm_plane.h:53-73: a second distance along the beam's medium and a
// phase-sampled direction turn every photon beam into a photon plane (host side, gvpm.cpp:793-797).
void planesFromBeams(const SynthScene &sc, int iteration, const PhotonBuffers &beams, std::vector<float> &w1,
                     std::vector<float> &len1) {
  w1.clear();
  len1.clear();
  const double sigT = sc.medium.sigma_t[1], g = sc.medium.g;
  for (uint64_t i = 0; i < beams.n; ++i) {
    Philox rng(sc.seed, 0x91a7eu, (uint32_t)iteration, (uint32_t)i);
    // sampleDistance(Ray(o, d, 0.f) -> mint = Epsilon, maxt = inf), homogeneous.cpp:293-360
    const double t = -std::log(1.0 - rng.next1D()) / sigT + kEpsilon;
    V3 d = normalize(V3(beams.pos[3 * i] - beams.parent_pos[3 * i], beams.pos[3 * i + 1] - beams.parent_pos[3 * i + 1],
                        beams.pos[3 * i + 2] - beams.parent_pos[3 * i + 2]));
    V3 wo;
    do {
      const double a = rng.next1D(), b = rng.next1D();
      if (std::fabs(g) < kEpsilon) {
        wo = uniformSphere(a, b);
      } else {
        double sqrTerm = (1 - g * g) / (1 - g + 2 * g * a);
        double cosTheta = (1 + g * g - sqrTerm * sqrTerm) / (2 * g);
        double sinTheta = std::sqrt(std::fmax(0.0, 1.0 - cosTheta * cosTheta));
        wo = toWorld(d, V3(sinTheta * std::cos(2 * kPi * b), sinTheta * std::sin(2 * kPi * b), cosTheta));
      }
    } while (std::fabs(dot(d, wo)) == 1.0);
    push3(w1, wo);
    len1.push_back((float)t);
  }
}

}  // namespace gvpm
