#include "synth.h"
#include "synth_core.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <thread>

#include "philox.h"

namespace gvpm {

SceneView SynthScene::view() const {
  SceneView v;
  v.tris = tris.data();
  v.ntri = (int)tris.size();
  v.mats = mats.data();
  v.nmats = (int)mats.size();
  v.lightC = lightC; v.lightU = lightU; v.lightV = lightV; v.lightN = lightN; v.radiance = radiance;
  v.lightArea = lightArea;
  v.medium = medium;
  v.camPos = camPos;
  v.camX = camX; v.camY = camY; v.camZ = camZ;
  v.tanHalfFovX = tanHalfFovX;
  v.width = width; v.height = height;
  v.seed = seed;
  v.cameraInside = cameraInside;
  v.maxDepth = maxDepth; v.rrDepth = rrDepth; v.minDepth = minDepth;
  v.cameraSphere = cameraSphere;
  return v;
}

double SynthScene::bsphereRadius() const {
  // AABB::getBSphere(): centre = box centre, radius = |max - centre|
  V3 c = (bmin + bmax) * 0.5;
  return length(bmax - c);
}

V3 SynthScene::toWorld(V3 p) const {
  if (!rotated) return p;  // (the axis-aligned scenes keep their coordinates bit for bit)
  return V3(dot(rot[0], p), dot(rot[1], p), dot(rot[2], p));
}

void SynthScene::addQuad(V3 a, V3 b, V3 c, V3 d, int mat) {
  a = toWorld(a); b = toWorld(b); c = toWorld(c); d = toWorld(d);
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
  const V3 u(sx, 0, 0), v(0, 0, sz);
  s.radiance = radiance;
  s.lightArea = sx * sz;
  V3 a = c - u * 0.5 - v * 0.5;
  V3 b = c + u * 0.5 - v * 0.5;
  V3 cc = c + u * 0.5 + v * 0.5;
  V3 d = c - u * 0.5 + v * 0.5;
  s.addQuad(a, b, cc, d, mat);  // normal (0,-1,0)
  s.lightC = s.toWorld(c);
  s.lightU = s.toWorld(u);
  s.lightV = s.toWorld(v);
  s.lightN = s.toWorld(V3(0, -1, 0));
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

// A solid box in GENERAL position: half extents h around `pivot + up * h.y` in its own frame, that frame turned about the
// pivot by yaw (about y), then leanX (about x), then leanZ (about z) -- no face normal is an axis, no vertex coordinate a
// round number.  `sink`: how far the box reaches below its pivot (a leaning box on a floor has its base under the floor).
static void addTiltedBox(SynthScene &s, V3 pivot, V3 h, double yaw, double leanX, double leanZ, double sink, int mat) {
  const double cy = std::cos(yaw), sy = std::sin(yaw), cx = std::cos(leanX), sx = std::sin(leanX), cz = std::cos(leanZ),
               sz = std::sin(leanZ);
  auto turn = [&](V3 p) {
    p = V3(cy * p.x + sy * p.z, p.y, -sy * p.x + cy * p.z);     // yaw
    p = V3(p.x, cx * p.y - sx * p.z, sx * p.y + cx * p.z);      // lean about x
    p = V3(cz * p.x - sz * p.y, sz * p.x + cz * p.y, p.z);      // lean about z
    return pivot + p;
  };
  const V3 lo(-h.x, -sink, -h.z), hi(h.x, 2.0 * h.y, h.z);
  auto P = [&](double x, double y, double z) { return turn(V3(x, y, z)); };
  s.addQuad(P(lo.x, lo.y, lo.z), P(hi.x, lo.y, lo.z), P(hi.x, lo.y, hi.z), P(lo.x, lo.y, hi.z), mat);  // -y
  s.addQuad(P(lo.x, hi.y, lo.z), P(lo.x, hi.y, hi.z), P(hi.x, hi.y, hi.z), P(hi.x, hi.y, lo.z), mat);  // +y
  s.addQuad(P(lo.x, lo.y, lo.z), P(lo.x, hi.y, lo.z), P(hi.x, hi.y, lo.z), P(hi.x, lo.y, lo.z), mat);  // -z
  s.addQuad(P(lo.x, lo.y, hi.z), P(hi.x, lo.y, hi.z), P(hi.x, hi.y, hi.z), P(lo.x, hi.y, hi.z), mat);  // +z
  s.addQuad(P(lo.x, lo.y, lo.z), P(lo.x, lo.y, hi.z), P(lo.x, hi.y, hi.z), P(lo.x, hi.y, lo.z), mat);  // -x
  s.addQuad(P(hi.x, lo.y, lo.z), P(hi.x, hi.y, lo.z), P(hi.x, hi.y, hi.z), P(hi.x, lo.y, hi.z), mat);  // +x
}

// the two blocks of a Cornell box, tilted (the `_rot` variants of the cbox scenes)
static void addCornellBlocks(SynthScene &s, int matShort, int matTall) {
  const double deg = kPi / 180.0;
  addTiltedBox(s, V3(0.36, -1.0, 0.31), V3(0.27, 0.29, 0.27), -17.3 * deg, 4.1 * deg, -2.7 * deg, 0.1, matShort);
  addTiltedBox(s, V3(-0.37, -1.0, -0.33), V3(0.28, 0.61, 0.28), 19.6 * deg, -3.3 * deg, 5.2 * deg, 0.1, matTall);
}

bool makeScene(const std::string &fullName, int width, int height, uint32_t seed, SynthScene &s) {
  s = SynthScene();
  s.name = fullName;
  // `<scene>_rot`: the scene in GENERAL POSITION -- everything (room, light, sensor) under one fixed rotation about the
  // origin, Euler angles 17 / 31 / 47 degrees (R = Rz Ry Rx), and its inner boxes individually tilted (cbox*: the two
  // Cornell blocks added; fogroom: each of the 64 boxes with its own yaw and lean).  No surface normal is an axis, no
  // wall coordinate is exactly representable: what the axis-aligned scenes cannot tell apart from exact arithmetic
  // (octahedral normals, the own-wall test of the near lists, slab tests with zero direction components, the branches
  // of coordinateSystem) is exercised here.
  const bool rot = fullName.size() > 4 && fullName.compare(fullName.size() - 4, 4, "_rot") == 0;
  const std::string name = rot ? fullName.substr(0, fullName.size() - 4) : fullName;
  if (rot) {
    const double ax = 17.0 * kPi / 180.0, ay = 31.0 * kPi / 180.0, az = 47.0 * kPi / 180.0;
    const double cx = std::cos(ax), sx = std::sin(ax), cy = std::cos(ay), sy = std::sin(ay), cz = std::cos(az), sz = std::sin(az);
    s.rot[0] = V3(cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx);
    s.rot[1] = V3(sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx);
    s.rot[2] = V3(-sy, cy * sx, cy * cx);
    s.rotated = true;
  }
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
  s.mats.push_back({MAT_LAMBERT, V3(0.5, 0.5, 0.5), V3(0.0), 0.0, 0.0, -1});       // 0 white
  s.mats.push_back({MAT_LAMBERT, V3(0.63, 0.065, 0.05), V3(0.0), 0.0, 0.0, -1});    // 1 red (left)
  s.mats.push_back({MAT_LAMBERT, V3(0.14, 0.45, 0.091), V3(0.0), 0.0, 0.0, -1});    // 2 green (right)
  s.mats.push_back({MAT_NULL, V3(0, 0, 0), V3(0.0), 0.0, 0.0, -1});                 // 3 medium boundary (front)
  if (name == "cbox_phong" || name == "cbox_phong_hg" || name == "cbox_phong1") {
    // S-cbox with GLOSSY walls: floor and back wall are Phong surfaces (a polished floor, exponent 40;
    // a satin wall, exponent 12), so that a large part of the photons are re-connected through a non-Lambertian parent
    auto phong = [&](V3 kd, V3 ks, double e) {
      auto lum = [](V3 c) { return 0.212671 * c.x + 0.715160 * c.y + 0.072169 * c.z; };  // Spectrum::getLuminance, RGB
      SynthMat m{MAT_PHONG, kd, ks, e, lum(ks) / (lum(kd) + lum(ks)), 0};
      m.bsdf = 0;
      for (const auto &q : s.mats) m.bsdf += bsdfEntries(q.kind, q.exponent);
      s.mats.push_back(m);
      return (int)s.mats.size() - 1;
    };
    // `cbox_phong1` (round 5): exponents 1500 and 900, i.e. roughness 0.037 and 0.047 < 0.05 -- sampleNext picks ONE component
    // per bounce (Phong::sampleComponent) and the reconnection evaluates that component alone
    const bool one = name == "cbox_phong1";
    const int mFloor = phong(V3(0.3, 0.3, 0.3), V3(0.5, 0.5, 0.45), one ? 1500.0 : 40.0);
    const int mBack = phong(V3(0.2, 0.25, 0.4), V3(0.3, 0.3, 0.3), one ? 900.0 : 12.0);
    addBoxRoom(s, mFloor, 0, mBack, 1, 2, 3);
    if (rot) addCornellBlocks(s, 0, mBack);
    setLight(s, V3(0, 0.998, 0), 0.5, 0.5, V3(15, 15, 15), 0);
    setMedium(s, 0.5, 0.5, name == "cbox_phong_hg" ? 0.7 : 0.0);
  } else if (name == "cbox_ward" || name == "cbox_ward_duer") {
    // S-cbox with WARD walls (row f4, round 5; src/bsdfs/ward.cpp -- a material of the bathroom scene BASELINE configs[3] is named
    // after): the floor a balanced Ward lacquer (alpha 0.15, the plugin's default variant), the back wall the original Ward model
    // (alpha 0.3) -- `_duer`: both with Duer's correction; isotropic, roughness >= 0.05: both components sampled together
    auto ward = [&](V3 kd, V3 ks, double alpha, int variant) {
      auto lum = [](V3 c) { return 0.212671 * c.x + 0.715160 * c.y + 0.072169 * c.z; };  // Spectrum::getLuminance, RGB
      SynthMat m{MAT_WARD, kd, ks, alpha, lum(ks) / (lum(kd) + lum(ks)), 0};
      m.distribution = variant;
      m.bsdf = 0;
      for (const auto &q : s.mats) m.bsdf += bsdfEntries(q.kind, q.exponent);
      s.mats.push_back(m);
      return (int)s.mats.size() - 1;
    };
    const bool duer = name == "cbox_ward_duer";
    const int mFloor = ward(V3(0.3, 0.3, 0.3), V3(0.5, 0.5, 0.45), 0.15, duer ? GVPM_WARD_DUER : GVPM_WARD_BALANCED);
    const int mBack = ward(V3(0.2, 0.25, 0.4), V3(0.3, 0.3, 0.3), 0.3, duer ? GVPM_WARD_DUER : GVPM_WARD_WARD);
    addBoxRoom(s, mFloor, 0, mBack, 1, 2, 3);
    if (rot) addCornellBlocks(s, 0, mBack);
    setLight(s, V3(0, 0.998, 0), 0.5, 0.5, V3(15, 15, 15), 0);
    setMedium(s, 0.5, 0.5, 0.0);
  } else if (name == "cbox_conductor") {
    // S-cbox with METAL walls (row f4): the floor a rough copper plate (Beckmann, alpha 0.3), the back wall brushed aluminium
    // (GGX, alpha 0.2) -- src/bsdfs/roughconductor.cpp, the table's second kind; eta / k: Mitsuba's Cu and Al at RGB
    auto conductor = [&](V3 eta, V3 k, double alpha, int distribution) {
      SynthMat m{MAT_ROUGHCONDUCTOR, V3(0.0), V3(1.0), alpha, 0.0, 0};
      m.eta = eta;
      m.k = k;
      m.distribution = distribution;
      m.bsdf = 0;
      for (const auto &q : s.mats) m.bsdf += bsdfEntries(q.kind, q.exponent);
      s.mats.push_back(m);
      return (int)s.mats.size() - 1;
    };
    const int mFloor = conductor(V3(0.2004, 0.9240, 1.1022), V3(3.9129, 2.4528, 2.1421), 0.3, GVPM_MICROFACET_BECKMANN);
    const int mBack = conductor(V3(1.6574, 0.8803, 0.5212), V3(9.2238, 6.2695, 4.8370), 0.2, GVPM_MICROFACET_GGX);
    addBoxRoom(s, mFloor, 0, mBack, 1, 2, 3);
    if (rot) addCornellBlocks(s, 0, mBack);
    setLight(s, V3(0, 0.998, 0), 0.5, 0.5, V3(15, 15, 15), 0);
    setMedium(s, 0.5, 0.5, 0.0);
  } else if (name == "cbox") {
    addBoxRoom(s, 0, 0, 0, 1, 2, 3);
    if (rot) addCornellBlocks(s, 0, 0);
    setLight(s, V3(0, 0.998, 0), 0.5, 0.5, V3(15, 15, 15), 0);
    setMedium(s, 0.5, 0.5, 0.0);
  } else if (name == "cbox_hg") {
    // same room, forward-scattering fog: exercises the EMediumShift -> diffuse
    // reconnection branch (g > 0.5 makes medium vertices "glossy",
    // gvpm_struct.h:73-76)
    addBoxRoom(s, 0, 0, 0, 1, 2, 3);
    if (rot) addCornellBlocks(s, 0, 0);
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
  } else if (name == "laser" || name == "laser_hg" || name == "laser_in" || name == "laser_in_hg") {
    // S-laser: sigma_t = 0.5, small emitter behind an aperture plate; `_hg`: HG g = 0.7;
    // `_in`: closed room with the sensor inside the fog, what the plane estimator requires (gvpm.cpp:784-788)
    const bool inside = name == "laser_in" || name == "laser_in_hg";
    const bool hg = name == "laser_hg" || name == "laser_in_hg";
    addBoxRoom(s, 0, 0, 0, 1, 2, inside ? 0 : 3);
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
    setMedium(s, 0.25, 0.25, hg ? 0.7 : 0.0);
    if (inside) {
      s.camPos = V3(0, 0, 0.95);
      s.tanHalfFovX = std::tan(0.5 * 60.0 * kPi / 180.0);
      s.cameraInside = true;
    }
  } else if (name == "cbox_mirror" || name == "cbox_mirror_side") {
    // closed room, sensor inside the fog, one wall a perfect mirror: the camera paths that meet it have TWO medium
    // edges (sensor -> mirror -> a diffuse wall), light paths get Dirac vertices (manifold-type shifts).  Back wall:
    // perpendicular to the optical axis (the sensor's area pdf is constant over it); `_side`: the left wall, met at
    // grazing angles (pdf, Jacobian and GOp of the shifted paths all differ from the base path's)
    s.mats.push_back({MAT_MIRROR, V3(0.9, 0.85, 0.8), V3(0.0), 0.0, 0.0, -1});  // 4
    if (name == "cbox_mirror") addBoxRoom(s, 0, 0, 4, 1, 2, 0);
    else addBoxRoom(s, 0, 0, 0, 4, 2, 0);
    setLight(s, V3(0, 0.998, 0), 0.5, 0.5, V3(15, 15, 15), 0);
    setMedium(s, 0.5, 0.5, 0.0);
    s.camPos = V3(0.1, 0.05, 0.95);
    s.tanHalfFovX = std::tan(0.5 * 75.0 * kPi / 180.0);
    s.cameraInside = true;
  } else if (name == "fogroom") {
    addBoxRoom(s, 0, 0, 0, 1, 2, 3);
    setLight(s, V3(0, 0.998, 0), 0.5, 0.5, V3(15, 15, 15), 0);
    Philox rng(seed, 0x0b0c5u, 0, 0);
    for (int k = 0; k < 64; ++k) {
      double cx = -0.85 + 1.7 * rng.next1D(), cz = -0.85 + 1.7 * rng.next1D();
      double hx = 0.04 + 0.06 * rng.next1D(), hz = 0.04 + 0.06 * rng.next1D();
      double hy = 0.1 + 0.5 * rng.next1D();
      if (rot) {
        const double yaw = 2.0 * kPi * rng.next1D(), lx = (rng.next1D() - 0.5) * 0.4, lz = (rng.next1D() - 0.5) * 0.4;
        addTiltedBox(s, V3(cx, -1.0, cz), V3(hx, 0.5 * hy, hz), yaw, lx, lz, 0.05, 0);
      } else {
        addInnerBox(s, V3(cx - hx, -1.0, cz - hz), V3(cx + hx, -1.0 + hy, cz + hz), 0);
      }
    }
    setMedium(s, 0.5, 0.5, 0.0);
  } else {
    return false;
  }
  if (rot) {
    // the sensor turns with the room; the scene's AABB (Scene::getAABB: all shapes) is that of the turned geometry
    s.camPos = s.toWorld(s.camPos);
    s.camX = s.toWorld(V3(1, 0, 0)); s.camY = s.toWorld(V3(0, 1, 0)); s.camZ = s.toWorld(V3(0, 0, 1));
    const double inf = std::numeric_limits<double>::infinity();
    s.bmin = V3(inf, inf, inf);
    s.bmax = V3(-inf, -inf, -inf);
    for (const auto &t : s.tris) {
      const V3 v[3] = {t.v0, t.v0 + t.e1, t.v0 + t.e2};
      for (const V3 &q : v) {
        s.bmin = V3(std::fmin(s.bmin.x, q.x), std::fmin(s.bmin.y, q.y), std::fmin(s.bmin.z, q.z));
        s.bmax = V3(std::fmax(s.bmax.x, q.x), std::fmax(s.bmax.y, q.y), std::fmax(s.bmax.z, q.z));
      }
    }
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

static uint64_t shootCommon(const SynthScene &sc, int iteration, uint64_t capacity, PhotonBuffers &out, bool beams,
                            std::vector<float> *endN);

uint64_t shootPhotons(const SynthScene &sc, int iteration, uint64_t capacity, PhotonBuffers &out) {
  return shootCommon(sc, iteration, capacity, out, false, nullptr);
}

uint64_t shootBeams(const SynthScene &sc, int iteration, uint64_t capacity, PhotonBuffers &out,
                    std::vector<float> &endN) {
  return shootCommon(sc, iteration, capacity, out, true, &endN);
}

static uint64_t shootCommon(const SynthScene &scene, int iteration, uint64_t capacity, PhotonBuffers &out, bool beams,
                            std::vector<float> *endN) {
  const SceneView sc = scene.view();
  out.clear();
  if (endN) endN->clear();
  uint64_t nbPaths = 0;
  uint32_t nbLightPathAdded = 0;
  // Paths are keyed by their index, so chunks of them can be generated by worker threads and
  // appended in index order: the result is identical to the sequential loop
  // (`deterministic` mode of the reference, gvpm.cpp:399-409).
  const uint64_t CH = 8192;
  unsigned nthreads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  std::vector<RecList> chunk(CH);
  std::vector<uint8_t> counted(CH, 1);
  uint64_t nextIndex = 0;  // index of the next light path (keys its random stream)
  while (out.n < capacity) {
    const uint64_t base = nextIndex;
    auto worker = [&](unsigned tid) {
      LPath path;
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
      for (int q = 0; q < chunk[k].n; ++q) {
        const PhotonRec &r = chunk[k].r[q];
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
void cameraBeams(const SynthScene &scene, int iteration, int x0, int y0, int x1, int y1,
                 std::vector<gvpm_camera_ray> &out, int tileMod, int tileRem, std::vector<float> *selW) {
  out.clear();
  if (selW) selW->clear();
  const SceneView sc = scene.view();
  const int tilesX = (sc.width + 3) / 4;
  // pixels are keyed by their index: rows are generated by worker threads and concatenated in row order,
  // which is the sequential loop's output
  const int nrows = y1 > y0 ? y1 - y0 : 0;
  std::vector<std::vector<gvpm_camera_ray>> rows((size_t)nrows);
  std::vector<std::vector<float>> rowW((size_t)nrows);
  const unsigned nthreads = std::max(1u, std::min({16u, std::thread::hardware_concurrency(), (unsigned)std::max(1, nrows)}));
  auto worker = [&](unsigned tid) {
    for (int r = (int)tid; r < nrows; r += (int)nthreads) {
      const int py = y0 + r;
      for (int px = x0; px < x1; ++px) {
        // image-sharded hosts: 4x4-pixel tiles dealt round-robin to the ranks (an even split of the work)
        if (tileMod > 1 && ((py / 4) * tilesX + px / 4) % tileMod != tileRem) continue;
        gvpm_camera_ray sets[2][5];
        float w[2];
        const int n = cameraBeamSets(sc, iteration, px, py, sets, w);  // one set per medium edge of the pixel's path
        for (int k = 0; k < n; ++k) {
          rows[(size_t)r].insert(rows[(size_t)r].end(), sets[k], sets[k] + 5);
          rowW[(size_t)r].push_back(w[k]);
        }
      }
    }
  };
  std::vector<std::thread> th;
  for (unsigned t = 1; t < nthreads; ++t) th.emplace_back(worker, t);
  worker(0);
  for (auto &t : th) t.join();
  size_t total = 0;
  for (const auto &r : rows) total += r.size();
  out.reserve(total);
  for (const auto &r : rows) out.insert(out.end(), r.begin(), r.end());
  if (selW)
    for (const auto &r : rowW) selW->insert(selW->end(), r.begin(), r.end());
}

// computeVolumeGradientPhoton's per-pixel part, gvpm.cpp:1117-1172: the CDF over the medium edges of the pixel's camera
// path (weights weightBeam.max()), one sampler->next1D() per camera sample, sampleReuse (the selected edge and the
// re-stretched sample), and the selection probability the functor divides by.  The sets of a pixel are consecutive.
void cameraSamplesVPM(const SynthScene &sc, int iteration, const std::vector<gvpm_camera_ray> &rays,
                      const std::vector<float> &selW, int nbCameraSamples, std::vector<gvpm_vpm_sample> &out) {
  out.clear();
  const size_t nsets = rays.size() / 5;
  for (size_t s = 0; s < nsets;) {
    const uint32_t pix = rays[5 * s].pixel;
    size_t e = s + 1;
    while (e < nsets && rays[5 * e].pixel == pix) ++e;
    const size_t ne = e - s;  // medium edges of this pixel
    // DiscreteDistribution::normalize (include/mitsuba/core/pmf.h): cdf[i + 1] = cdf[i] + w[i], divided by the sum
    float cdf[3] = {0.f, 0.f, 0.f}, w[2] = {1.f, 0.f};
    for (size_t k = 0; k < ne && k < 2; ++k) {
      w[k] = selW.size() == nsets ? selW[s + k] : 1.f;
      cdf[k + 1] = cdf[k] + w[k];
    }
    const float sum = cdf[ne < 2 ? ne : 2];
    for (size_t k = 1; k <= ne && k <= 2; ++k) cdf[k] = sum > 0.f ? cdf[k] / sum : 0.f;
    cdf[ne < 2 ? ne : 2] = 1.f;
    const uint32_t idx = (pix >> 16) * (uint32_t)sc.width + (pix & 0xFFFFu);
    Philox rng(sc.seed, 0x5a3fu, (uint32_t)iteration, idx);
    for (int k = 0; k < nbCameraSamples; ++k) {
      float u = rng.next1D();
      // sampleReuse: index = the entry whose cdf interval holds u; u re-stretched to [0, 1) inside it
      size_t sel = (ne >= 2 && u >= cdf[1]) ? 1 : 0;
      u = (u - cdf[sel]) / (cdf[sel + 1] - cdf[sel]);
      gvpm_vpm_sample sm;
      sm.set = (uint32_t)(s + sel);
      sm.rand = u;
      sm.pdf_sel = cdf[sel + 1] - cdf[sel];
      sm.reserved = 0;
      out.push_back(sm);
    }
    s = e;
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
