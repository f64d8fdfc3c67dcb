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
// Execution model (64-lane waves that share nothing but staged occluders, no MFMA: gather / divergent math):
//   * a TILE is a bundle of B camera-beam sets (B = 16/32/64, 64/B lanes per beam) that the
//     tile sort made spatially coherent (8x8 / 8x4 / 4x4 pixel tiles);
//   * the photon map is a uniform grid (all photons share one radius, gvpm.cpp:989) sorted by
//     cell with x fastest; a tile walks the grid in thick slabs along its major axis,
//     wave-reduces the fattened footprint of its beams into a cell box and turns the box into
//     x-contiguous photon ranges;
//   * plan_kernel walks every tile once WITHOUT touching photons (cellStart differences only) and
//     cuts it into work items of roughly equal candidate count -- the load balancer that replaces
//     BlockScheduler's dynamic image blocks (photonmapper/utilities/block_sched.h:87-113);
//   * traverse_bre_kernel (one workgroup per item, or persistent waves pulling items from an atomic
//     queue: GVPM_PERSISTENT) holds no evaluation state, so it runs at high occupancy: for each slab step the 16-byte hot photon
//     records of the ranges are copied coalesced into an LDS stage and every lane tests them
//     against its own beam (LDS broadcast reads) with a CONSERVATIVE fp32 test (error band on the
//     safe side) plus the exact integer filters (depth, interaction mode, checkerboard parity).
//     Survivors are compacted with __ballot / popcount and appended to the list of their beam in
//     the item's region of the pair buffer (sized by the planner's upper bound; 4 bytes per pair);
//   * evaluate_bre_kernel (persistent, four waves per workgroup) cuts the concatenated per-beam lists
//     of an item into 64 equal chunks, one per lane, walked in segments of 16 steps.  Every pair is first DECIDED: the fp32 test with a rigorous error
//     band, the reference predicate in uncontracted fp64 only when the band could change the
//     decision (~1e-5 of the pairs), so the evaluated set equals the fp64 oracle's bit for bit.
//     Then phase 1 (base contribution + null shifts, 27 sums in registers) and, in a loop of its
//     own, phase 2 (offset-path reconnections with shadow ray, Jacobian, MIS weight); the sums go
//     to per-beam LDS accumulators and are flushed with global atomics at the end of the item.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "device_types.h"
#include "shift_device.h"
#include "tile_walk.h"
#include "vec.h"

namespace gvpm {

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

// The reference's hit decision for one (photon, beam) candidate, in its own operation order.  Rare
// (only candidates inside the fp32 error band get here) and register hungry (fp64 divisions and
// square root), hence not inlined into the traversal loop.
static __device__ __noinline__ bool exactHit(f3 p, f3 o, f3 d, float len, float r, float rnd, float eps, bool use3D) {
  const double mintD = (double)eps, maxtD = (double)len - (double)eps;
  const d3 rcpD = mkd(1.0 / (double)d.x, 1.0 / (double)d.y, 1.0 / (double)d.z);
  const HitGeom g = hitGeom(p, o, d);
  if (!(g.disk > mintD && g.distSqr < (double)r * (double)r && ownBoxHit(p, o, d, rcpD, mintD, maxtD, (double)r)))
    return false;
  if (!use3D) return true;
  double tp, dt;
  return resample3D(g, (double)r, (double)rnd, mintD, (double)len, tp, dt);
}

// coordinateSystemCoherent, src/libcore/util.cpp:592-599 (its intermediates are float)
__device__ __forceinline__ void coherentFrame(f3 n, f3 &b1, f3 &b2) {
  const float sign = copysignf(1.0f, n.z);
  const float aa = -frcp(sign + n.z);
  const float bb = n.x * n.y * aa;
  b1 = mk3(1.0f + sign * n.x * n.x * aa, sign * bb, -sign * n.x);
  b2 = mk3(bb, sign + n.y * n.y * aa, -n.y);
}

// One evaluation = VolumeGradientBREQuery::operator() after the filters, in two phases so that
// the expensive, divergent part runs on full waves:
//   phase 1 (one lane per (photon, beam) pair): base contribution, then for each of the four
//     shifted rays the null shift if it applies (cheap); a shift that needs the offset-path
//     reconnection is only QUEUED (2 bytes: step, lane, shift, beam);
//   phase 2 (one lane per queued shift): the diffuse reconnection with its shadow ray, Jacobian
//     and MIS weight, in a dense loop of its own (see the kernel).
// At C2 ~70 % of the shifts are null shifts: running both branches on every lane of a mixed
// wave cost ~1.5x the VALU work of this arrangement.
//
// Accumulation: LDS float atomics (ds_add_f32) retire about one lane per clock on CDNA4 -- 24 of
// them per evaluation were half of this kernel's time.  The traversal therefore writes one photon
// list PER BEAM, the evaluation wave cuts the concatenated lists of an item into 64 equal chunks
// (perfect balance) and every lane sums its 27 outputs in REGISTERS; a lane touches the LDS
// accumulators (double: ds_add_f64 runs ~25x the rate of ds_add_f32, scripts/probes/lds_atomics_bench.hip)
// only when its chunk crosses into the next beam and at the end of a segment.
//
// Numerics: every quantity that the reference obtains by subtracting O(1) positions to get an
// O(radius) vector (photon - ray point, shifted ray point - base ray point) is formed in fp64 and
// then carried as a small fp32 vector; everything downstream of those differences (kernel chord
// lengths sqrt(r^2 - d^2), pdfs, BSDF / phase / transmittance products, MIS weights) is fp32.
constexpr uint32_t EVAL_LDS_TRIS = 64;  // occluders staged in the evaluation kernel's LDS when the scene has no more

// the fp32 error band of the hit test: E bounds |disk - disk_exact|, band |d2 - d2_exact| for the
// difference vector wv = photon - ray origin and disk = wv . d
__device__ __forceinline__ void hitBand(f3 wv, float disk, float r, float r2f, float &E, float &band) {
  E = 6e-7f * (fabsf(wv.x) + fabsf(wv.y) + fabsf(wv.z) + fabsf(disk));
  band = 4.f * r * E + r2f * 2e-6f;
}

// The hit decision of one (photon, beam) candidate: gvpm_accel.h:279-301 (disk, distSqr, own box) and, for the
// 3D kernel, the validity of the resampled t' (shift_volume_photon.cpp:707-726).  Decided in fp32 when the error
// band cannot change the reference's decision, otherwise by the reference predicate itself (fp64, uncontracted).
// Returns 1 (hit), 0 (no hit) or 2: the band cannot decide -- about one candidate in 10^5 at C2, plus the photons
// that project beyond the beam's end (the own-box test decides those) -- and exactHit() has to.
// The photon's own sphere box [p - r, p + r] against the ray segment [mint, maxt] (ownBoxHit: what every ancestor box of the
// reference's BVH implies) with fp32 error bands: 1 the slab test surely passes, 0 surely fails, 2 undecidable.  Only asked
// for photons that project at or beyond the beam's END (about one candidate in a thousand): between the ends the point of
// closest approach lies in the sphere, hence in the box, and the test passes by construction.
__device__ __forceinline__ int ownBoxBand(f3 p, f3 o, f3 d, float r, float mint, float maxt) {
  float nearT = -INFINITY, farT = INFINITY, eN = 0.f, eF = 0.f;
  bool open = false;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float pk = comp(p, k), ok = comp(o, k), dk = comp(d, k);
    if (dk == 0.f) open = true;  // (the reference branches on an exactly zero component: the exact pass decides)
    const float rc = frcp(dk);
    float t1 = ((pk - r) - ok) * rc, t2 = ((pk + r) - ok) * rc;
    const float lo = fminf(t1, t2), hi = fmaxf(t1, t2);
    // numerators good to ~3 eps (|p| + |o| + r), the reciprocal to 1 ulp, the product to half an ulp
    const float e = 4e-7f * (fabsf(pk) + fabsf(ok) + r) * fabsf(rc) + 2e-7f * (fabsf(lo) + fabsf(hi));
    if (lo > nearT) { nearT = lo; eN = e; } else if (lo + e > nearT - eN) { eN = fmaxf(eN, e + (nearT - lo)); }
    if (hi < farT) { farT = hi; eF = e; } else if (hi - e < farT + eF) { eF = fmaxf(eF, e + (hi - farT)); }
  }
  if (open || !(nearT == nearT) || !(farT == farT)) return 2;
  const float mE = 2e-7f * (maxt + mint);
  const bool fail = nearT - eN > farT + eF || farT + eF < mint - mE || nearT - eN > maxt + mE;
  const bool pass = nearT + eN < farT - eF && farT - eF > mint + mE && nearT + eN < maxt - mE;
  return pass ? 1 : (fail ? 0 : 2);
}

__device__ __forceinline__ int decidePair(f3 pos, const RayReg &base, float rnd, float r, float eps, bool use3D) {
  const float r2f = r * r, mint = eps, maxt = base.len - eps;
  const f3 wv = pos - base.o;
  const float disk = dot(wv, base.d);
  const f3 v = wv - base.d * disk;
  const float d2 = dot(v, v);
  float E, band;
  hitBand(wv, disk, r, r2f, E, band);
  bool in0 = d2 < r2f - band && disk > mint + E && disk < maxt - E;
  // surely outside: beyond the band of the disk test (the own-box test can only remove more)
  bool outside = d2 > r2f + band || disk < mint - E;
  if (!outside && disk >= maxt - E && d2 < r2f - band) {
    // at or beyond the beam's end: the own-box test decides (round 5: with bands, here, instead of in fp64 later)
    const int bx = ownBoxBand(pos, base.o, base.d, r, mint, maxt);
    in0 = bx == 1;
    outside = bx == 0;
  }
  bool in = in0;
  if (use3D) {
    // t' = (disk - deltaT) + 2 deltaT rnd must lie in [mint, len]: bracket it with deltaT in [dTlo, dTup]
    const float q = r2f - d2;
    const float dTup = fsqrt(fmaxf(q + band, 0.f)) * 1.000001f, dTlo = fsqrt(fmaxf(q - band, 0.f)) * 0.999999f;
    const float slop = 2.f * E + 4e-7f * (fabsf(disk) + dTup);
    const float tLo = (disk - dTup) + 2.f * dTlo * rnd - slop;
    const float tHi = (disk - dTlo) + 2.f * dTup * rnd + slop;
    in = in0 && tLo > mint && tHi < base.len;
    outside = outside || (in0 && (tHi < mint || tLo > base.len));
  }
  return in ? 1 : (outside ? 0 : 2);
}
// The same decision for a candidate the bands above left open -- almost all of them at the RIM of the kernel: the band of d2
// is 4 r E with E the rounding of O(1) coordinate differences, 1e-3 of r^2 -- with the difference vector and the projection
// formed in fp64, as baseTerms forms them, and the perpendicular offset as a small fp32 vector: d2 to ~4e-7 relative, disk
// and t' to ~1e-15.  What is still inside THAT band (~1e-6 of the candidates) goes to the exact pass.
__device__ __forceinline__ int decidePairFine(f3 pos, const RayReg &base, float rnd, float r, float eps, bool use3D) {
  const d3 wD = tod(pos) - tod(base.o), bdD = tod(base.d);
  const double disk = dot(wD, bdD);
  const f3 perp = tof(wD - bdD * disk);
  const float d2 = dot(perp, perp), r2f = r * r;
  const float band = 2e-6f * r2f;
  const double mint = (double)eps, maxt = (double)base.len - (double)eps, tE = 1e-12 * (1.0 + fabs(disk));
  bool in0 = d2 < r2f - band && disk > mint + tE && disk < maxt - tE;
  bool outside = d2 > r2f + band || disk < mint - tE;
  if (!outside && disk >= maxt - tE && d2 < r2f - band) {
    const int bx = ownBoxBand(pos, base.o, base.d, r, eps, base.len - eps);
    in0 = bx == 1;
    outside = bx == 0;
  }
  bool in = in0;
  if (use3D) {
    const float q = r2f - d2;
    const double dTup = (double)(fsqrt(fmaxf(q + band, 0.f)) * 1.000001f), dTlo = (double)(fsqrt(fmaxf(q - band, 0.f)) * 0.999999f);
    const double tLo = (disk - dTup) + 2.0 * dTlo * (double)rnd - tE, tHi = (disk - dTlo) + 2.0 * dTup * (double)rnd + tE;
    in = in0 && tLo > mint && tHi < (double)base.len;
    outside = outside || (in0 && (tHi < mint || tLo > (double)base.len));
  }
  return in ? 1 : (outside ? 0 : 2);
}

// the 27 per-beam outputs of one lane, in registers
struct Acc27 {
  float v[27];
};

__device__ __forceinline__ void borderRule(const GatherArgs &a, uint32_t pix, int i, float &w) {
  // no reverse shift at the right and top borders, shift_volume_photon.cpp:843-846
  const int px = (int)(pix & 0xFFFFu), py = (int)(pix >> 16);
  if ((i == GVPM_RIGHT && px == a.cfg.width - 1) || (i == GVPM_TOP && py == a.cfg.height - 1)) w = 1.f;
}

struct BaseTerms {
  f3 rel;  // photon - baseRay(t')
  double tPrime;
  float pdfCam, scale, tr;  // tr: transmittance over [Epsilon, t'] (equal in the three channels)
  f3 bc;                    // base contribution * scale
};

// hit geometry + base contribution of a (photon, beam) pair, shift_volume_photon.cpp:701-751
template <int B, typename LDS>
__device__ __forceinline__ BaseTerms baseTerms(const GatherArgs &a, const LDS &s, f3 pos, f3 wi, f3 flux,
                                               const RayReg &base, uint32_t b) {
  BaseTerms t;
  const bool use3D = a.cfg.vol_technique == GVPM_VOL_BRE3D;
  const float r = a.radius, r2 = r * r;
  // gvpm_accel.h:296-299: disk in fp64, the perpendicular offset as a small vector
  const d3 wD = tod(pos) - tod(base.o), bdD = tod(base.d);
  const double disk = dot(wD, bdD);
  const f3 perp = tof(wD - bdD * disk);
  const float distSqr = dot(perp, perp);
  t.tPrime = disk;
  float kernelVol = 3.14159265358979323846f * r2;
  t.pdfCam = 1.f;
  if (use3D) {
    // shift_volume_photon.cpp:707-726
    const float deltaT = fsqrt(fmaxf(0.f, r2 - distSqr));
    t.tPrime = (disk - (double)deltaT) + (double)(2.f * deltaT * s.rnd[b]);
    kernelVol = (4.0f / 3.0f) * 3.14159265358979323846f * r2 * r;
    t.pdfCam = frcp(fmaxf(deltaT * 2.f, 0.0001f));
  }
  t.rel = perp + base.d * (float)(disk - t.tPrime);
  const float rr = a.cfg.path_set ? 2.f : 1.f;
  t.scale = rr * frcp(kernelVol * t.pdfCam);
  // the base and the four shifted rays all carry mint = Epsilon and maxt = t' (:769-770): one transmittance
  f3 trT;
  float dummy;
  mediumEval(a.med, (float)t.tPrime - a.cfg.epsilon, trT, dummy);
  t.tr = trT.x;
  const f3 sigS = mk3(a.med.sigmaS[0], a.med.sigmaS[1], a.med.sigmaS[2]);
  t.bc = (sigS * flux) * (t.tr * phaseEval(a.med.g, wi, -base.d) * t.scale) * base.eye;
  return t;
}



// phase 1: base contribution + the four shift attempts of one pair; reconnections are returned in qMask
// (HS: manifold-typed shifts are recorded for the host, gvpm_enable_host_shifts -- its own instantiation of the kernel)
// A shift whose branch fp32 cannot decide as the reference does (round 5) -- the null-shift test |z' - y|^2 < r^2 or t'
// against the shifted edge's length within the error band of the fp32 quantities (branchAmbiguous) -- adds nothing and
// counts nothing here: it is QUEUED like a reconnection, and phase 2, which re-derives the test from the same numbers,
// hands it to the exact pass (one deferral site in the kernel: its copy loops cost registers).
__device__ __forceinline__ bool branchAmbiguous(const GatherArgs &a, float y2, float r2, float tPf, float shLen) {
  // y is good to ~2e-7 |y| (the two differences are formed in fp64 / as exact fp32 differences), t' to half an ulp
  // (cfg.reserved[4]: GVPM_EXACT_ALL -- the band widened to everything, so that the exact pass evaluates EVERY shift: a test
  // of the pass against the oracle on whole frames, tests/test_exact_pass_gpu.py)
  if (a.cfg.reserved[4]) return true;
  return (a.cfg.use_shift_null && fabsf(y2 - r2) <= 4e-6f * r2) || fabsf(tPf - shLen) <= 4e-7f * (tPf + shLen);
}
template <int B, bool HS, typename LDS>
__device__ __forceinline__ void evalPhase1(const GatherArgs &a, LDS &s, const PhotonFront &ph, const RayReg &base,
                                           uint32_t b, Acc27 &acc, uint32_t &nNull, uint32_t &nFail, uint32_t &qMask) {
  const uint32_t pix = s.pix[b];
  const float r2 = a.radius * a.radius;
  const BaseTerms bt = baseTerms<B>(a, s, ph.pos, ph.wi, ph.flux, base, b);
  const f3 bc = bt.bc;
  acc.v[0] += bc.x;
  acc.v[1] += bc.y;
  acc.v[2] += bc.z;
  qMask = 0u;

  const f3 photonIn = mk3(a.med.sigmaS[0], a.med.sigmaS[1], a.med.sigmaS[2]) * ph.flux;
  const float tPf = (float)bt.tPrime;
  const uint32_t st = GVPM_PF_SHIFT_TYPE(ph.bits);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // branch-free: the null-shift arithmetic is cheap and some lane of the wave needs it anyway; a wave
    // spends more on exec-mask bookkeeping and taken branches than on the arithmetic they would skip
    // the shifted ray RELATIVE to the base ray (relToBase, tile_walk.h): shiftRay(t') - baseRay(t') = relO + relD t' is a
    // small fp32 vector (pixel spacing at depth t'), accurate to ~1e-10; the sensorMIS factor rides along
    const ShiftRel sh = loadShiftRel(s, i, b, base.d);
    // photon relative to shiftRay(t') = (photon - baseRay(t')) - (shiftRay(t') - baseRay(t'))
    const f3 y = bt.rel - (sh.ro + sh.rd * tPf);
    // shiftNull, shift_volume_photon.cpp:119-158 with the kernel pdfs of :782-801
    const float y2 = dot(y, y);
    const bool ambBranch = sh.valid && !(HS && st == 3u) && branchAmbiguous(a, y2, r2, tPf, sh.len);
    const bool isNull = !ambBranch && sh.valid && a.cfg.use_shift_null && y2 < r2 && tPf < sh.len;
    const f3 yp = y - sh.d * dot(y, sh.d);
    const float deltaS = fsqrt(fmaxf(0.f, r2 - dot(yp, yp)));
    const float pdfShiftPos = frcp(fmaxf(2.f * deltaS, 0.0001f));
    float wNull = 0.5f;
    if (a.cfg.use_mis)
      wNull = (pdfShiftPos == 0.f || bt.pdfCam == 0.f)
                  ? 1.f
                  : frcp(1.f + sh.sMIS * pdfShiftPos * frcp(bt.pdfCam));
    const f3 nullFlux = photonIn * (bt.tr * phaseEval(a.med.g, ph.wi, -sh.d)) * sh.eye;
    // shiftPhoton dispatch, shift_volume_photon.cpp:49-117: reconnections go to phase 2
    const bool wantsShift = !ambBranch && sh.valid && !isNull && sh.len >= tPf && a.cfg.debug_shift != GVPM_SHIFT_NULL;
    // (a manifold-typed photon goes to phase 2 as well when the host answers such shifts: it records the request there)
    const bool queued = wantsShift && (st == 1u || st == 2u || (HS && st == 3u));
    nNull += isNull ? 1u : 0u;
    nFail += (wantsShift && !queued) ? 1u : 0u;
    qMask |= (queued || ambBranch) ? (1u << i) : 0u;
    float w = isNull ? wNull : 1.f;
    borderRule(a, pix, i, w);
    const float keep = (queued || ambBranch) ? 0.f : 1.f;  // a queued / deferred shift adds nothing here
    const float ws = isNull ? w * bt.scale : 0.f;  // only the null shift has a shifted flux in phase 1
    acc.v[3 + 3 * i + 0] += nullFlux.x * ws;
    acc.v[3 + 3 * i + 1] += nullFlux.y * ws;
    acc.v[3 + 3 * i + 2] += nullFlux.z * ws;
    acc.v[15 + 3 * i + 0] += bc.x * (w * keep);
    acc.v[15 + 3 * i + 1] += bc.y * (w * keep);
    acc.v[15 + 3 * i + 2] += bc.z * (w * keep);
  }
}

// The rest of shiftPhotonManifold for the recorded requests, once the host has run the walks (results == nullptr: it has
// not -- every request is a failed shift): shift_volume_photon.cpp:205-279, then the accumulation of :843-853.
__global__ __launch_bounds__(256) void apply_host_shifts_kernel(GatherArgs a, const gvpm_host_shift *__restrict__ results, uint32_t n) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t ok = 0, bad = 0;
  if (k < n) {
    const float4 c0 = a.reqCtx[4 * (size_t)k], c1 = a.reqCtx[4 * (size_t)k + 1], c2 = a.reqCtx[4 * (size_t)k + 2],
                 c3 = a.reqCtx[4 * (size_t)k + 3];
    const float tr = c0.x, pdfCam = c0.y, pdfShiftPos = c0.z, sMIS = c0.w, scale = c1.x;
    const f3 bc = mk3(c1.y, c1.z, c1.w), shD = mk3(c2.x, c2.y, c2.z), eye = mk3(c3.x, c3.y, c3.z);
    const uint32_t pix = __float_as_uint(c2.w);
    const int i = (int)__float_as_uint(c3.w);
    float w = 1.f;
    f3 sflux = mk3(0.f);
    bool good = false;
    if (results && results[k].ok) {
      const gvpm_host_shift r = results[k];
      // result.jacobian *= sRecME.jacobian (1) * additionalJacobian (1); *= detProposed / detSource
      const float jac = r.det_ratio;
      if (jac > 0.f && isfinite(jac)) {
        good = true;
        const f3 photonWeight = mk3(r.throughput[0], r.throughput[1], r.throughput[2]);
        const f3 wi = mk3(r.wi[0], r.wi[1], r.wi[2]);
        const f3 sigS = mk3(a.med.sigmaS[0], a.med.sigmaS[1], a.med.sigmaS[2]);
        const f3 contrib = (sigS * photonWeight) * phaseEval(a.med.g, wi, -shD);
        sflux = contrib * eye * (tr * jac);
        w = 0.5f;
        const float offsetPdf = r.pdf * pdfShiftPos;
        if (offsetPdf == 0.f) {
          w = 1.f;
          sflux = mk3(0.f);
        }
        if (a.cfg.use_mis) {
          const float basePdf = pdfCam * r.base_pdf;
          if (basePdf == 0.f) {
            w = 0.f;
          } else if (a.cfg.power_heuristic) {
            const float v = sMIS * jac * (offsetPdf / basePdf);
            w = 1.f / (1.f + v * v);
          } else {
            w = 1.f / (1.f + sMIS * offsetPdf * jac / basePdf);
          }
        }
      }
    }
    borderRule(a, pix, i, w);
    const size_t p = (size_t)(pix >> 16) * a.cfg.width + (pix & 0xFFFFu);
    const float ws = w * scale * a.iterScale, wbS = w * a.iterScale;
    float *dst = a.iter + p * 27;
    if (ws != 0.f) {
      atomicAdd(&dst[3 + 3 * i + 0], sflux.x * ws);
      atomicAdd(&dst[3 + 3 * i + 1], sflux.y * ws);
      atomicAdd(&dst[3 + 3 * i + 2], sflux.z * ws);
    }
    atomicAdd(&dst[15 + 3 * i + 0], bc.x * wbS);
    atomicAdd(&dst[15 + 3 * i + 1], bc.y * wbS);
    atomicAdd(&dst[15 + 3 * i + 2], bc.z * wbS);
    ok = good ? 1u : 0u;
    bad = good ? 0u : 1u;
  }
  // statistics: a manifold shift that succeeded counts with the reconnections, the others are failed shifts
  const uint32_t nOk = (uint32_t)__popcll(__ballot(ok != 0u)), nBad = (uint32_t)__popcll(__ballot(bad != 0u));
  if ((threadIdx.x & 63) == 0 && (nOk | nBad)) {
    unsigned long long *row = a.stats + 8 * (size_t)(blockIdx.x % GVPM_STAT_ROWS);
    atomicAdd(&row[3], (unsigned long long)nOk);
    atomicAdd(&row[4], (unsigned long long)nBad);
  }
}
void launch_apply_host_shifts(const GatherArgs &a, const gvpm_host_shift *results, uint32_t n, hipStream_t s) {
  if (n) hipLaunchKernelGGL(apply_host_shifts_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, results, n);
}

// phase 2: one queued reconnection shift (shiftPhotonDiffuse through getShiftPos); the result goes to
// the lane's registers when it belongs to the lane's current beam, else straight to the LDS accumulators
template <int B, bool FULLVIS, bool HS, typename LDS>
__device__ __forceinline__ void evalPhase2Core(const GatherArgs &a, LDS &s, uint32_t pidx, uint32_t b, int i, f3 &sf,
                                               f3 &wb, uint32_t &nDiff, uint32_t &nFail, const float4 *ldsTri) {
  const PhotonCold ph = loadCold(a, pidx);
  const RayReg base = loadRay(s, 0, b);
  const ShiftRel sr = loadShiftRel(s, i, b, base.d);
  RayReg sh;  // what the reconnection reads of the shifted ray: direction, eye contribution
  sh.d = sr.d;
  sh.eye = sr.eye;
  sh.len = sr.len;
  sh.valid = sr.valid;
  const bool use3D = a.cfg.vol_technique == GVPM_VOL_BRE3D;
  const float r = a.radius, r2 = r * r;
  // t', pdfCameraPos and the photon relative to baseRay(t') exactly as phase 1 derived them (baseTerms)
  double tPrime;
  float pdfCam = 1.f;
  f3 rel;
  {
    const d3 wD = tod(ph.pos) - tod(base.o), bdD = tod(base.d);
    const double disk = dot(wD, bdD);
    const f3 perp = tof(wD - bdD * disk);
    tPrime = disk;
    if (use3D) {
      const float deltaT = fsqrt(fmaxf(0.f, r2 - dot(perp, perp)));
      tPrime = (disk - (double)deltaT) + (double)(2.f * deltaT * s.rnd[b]);
      pdfCam = frcp(fmaxf(deltaT * 2.f, 0.0001f));
    }
    rel = perp + base.d * (float)(disk - tPrime);
  }
  const float rr = a.cfg.path_set ? 2.f : 1.f;
  const float kernelVol = use3D ? (4.0f / 3.0f) * 3.14159265358979323846f * r2 * r : 3.14159265358979323846f * r2;
  const float scale = rr * frcp(kernelVol * pdfCam);
  f3 trT;
  float dummy;
  mediumEval(a.med, (float)tPrime - a.cfg.epsilon, trT, dummy);
  const f3 sigS = mk3(a.med.sigmaS[0], a.med.sigmaS[1], a.med.sigmaS[2]);
  const f3 bc = (sigS * ph.flux) * (trT.x * phaseEval(a.med.g, ph.wi, -base.d) * scale) * base.eye;

  const f3 basePt = base.o + base.d * (float)tPrime;  // baseRay(t'), absolute (for the segment to the parent)
  const f3 dS = sr.ro + sr.rd * (float)tPrime;  // shiftRay(t') - baseRay(t')
  // phase 1's branch test from phase 1's numbers (evalPhase1): an undecidable branch was queued to be deferred here
  uint32_t amb = 0u;
  if (!(HS && GVPM_PF_SHIFT_TYPE(ph.bits) == 3u)) {
    const f3 y = rel - dS;
    const float y2 = dot(y, y), tPf = (float)tPrime;
    amb = branchAmbiguous(a, y2, r2, tPf, sh.len) ? 2u : 0u;
    //  Phase 1 queues a shift for two reasons -- it IS a reconnection, or its branch is undecidable -- and
    // tells phase 2 neither: the test above is re-derived from the same numbers.  Should the two sites ever round differently,
    // an entry that is neither here -- not ambiguous, yet a null shift, a photon type without a reconnection, or a shifted edge
    // too short -- was queued as undecidable there: it goes to the exact pass, it is not evaluated as a reconnection.
    const uint32_t stq = GVPM_PF_SHIFT_TYPE(ph.bits);
    const bool nullNow = a.cfg.use_shift_null && y2 < r2 && tPf < sh.len;
    const bool reconnects = sh.valid && !nullNow && sh.len >= tPf && a.cfg.debug_shift != GVPM_SHIFT_NULL && (stq == 1u || stq == 2u);
    if (!amb && !reconnects) amb = 2u;
  }
  // getShiftPos, shift_volume_photon.cpp:858-896: offsetPos = shiftRay(t') + offRel
  f3 offRel = rel;
  if (!use3D) {
    f3 bs, bt, ns, nt;
    coherentFrame(base.d, bs, bt);
    coherentFrame(sh.d, ns, nt);
    offRel = ns * dot(rel, bs) + nt * dot(rel, bt) + sh.d * dot(rel, base.d);
  }
  if (a.cfg.use_shift_null) {
    const f3 bo = dS + offRel;       // offsetPos - baseRay(t')
    const float bo2 = dot(bo, bo);
    // (the mirror decision of getShiftPos moves the offset position by up to 2 r: not a counter, but a different shift)
    amb |= fabsf(bo2 - r2) <= (use3D ? 4e-6f : 2e-5f) * r2 ? 8u : 0u;
    const float cosD2 = bo2 < r2 ? -2.f * dot(dS, offRel) * frcp(dot(dS, dS)) : 0.f;
    offRel = offRel + dS * cosD2;
  }
  float pdfShiftPos = 1.f;
  if (use3D) {
    const f3 op = offRel - sh.d * dot(offRel, sh.d);
    const float deltaO = fsqrt(fmaxf(0.f, r2 - dot(op, op)));
    pdfShiftPos = frcp(fmaxf(2.f * deltaO, 0.0001f));
  }
  if (HS && GVPM_PF_SHIFT_TYPE(ph.bits) == 3u) {
    // EManifoldShift: the walk is the host's (recordShiftRequest); nothing is added now
    const f3 shiftPt = basePt + dS;
    if (recordShiftRequest(reqSink(a), a.radius, pidx, a.setPerm[s.setBase + b], i, shiftPt + offRel, basePt, shiftPt, (float)tPrime, trT.x, pdfCam,
                           pdfShiftPos, sr.sMIS, scale, bc, sh.d, sh.eye, s.pix[b])) {
      sf = wb = mk3(0.f);
    } else {
      nFail++;
      sf = mk3(0.f);
      wb = bc;  // weight 1
    }
    return;
  }
  bool ok = false;
  f3 sflux;
  const f3 dProjU = ((basePt + dS) - ph.parentPos) + offRel;  // offsetPos - parent
  uint32_t ambVis = 0u;
  float w = shiftDiffuse<FULLVIS>(a, ph, ph.bits, dProjU, sh, base, s.edge[b], trT, pdfCam, pdfShiftPos, sflux, ok, ldsTri,
                                 sr.sMIS, &ambVis);
  if (amb | ambVis) {
    // fp32 cannot decide this shift as the reference does: the exact pass evaluates it (nothing added, nothing counted)
    deferNote(a, GVPM_EX_KIND_BRE, a.setPerm[s.setBase + b], pidx, (uint32_t)i, amb | ambVis);
    sf = wb = mk3(0.f);
    return;
  }
  if (ok) nDiff++; else nFail++;
  borderRule(a, s.pix[b], i, w);
  const float ws = w * scale;
  sf = sflux * ws;
  wb = bc * w;
}

// a lane's register sums -> the LDS accumulators of `beam`
template <int B, typename LDS> __device__ __forceinline__ void flushAcc(LDS &s, Acc27 &acc, uint32_t beam) {
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    atomicAdd(&s.acc[k][beam], (double)(acc.v[k]));
    acc.v[k] = 0.f;
  }
}

// ------------------------------------------------------------------------------------------
// traversal: persistent waves pulling work items, (photon, beam) pairs out
// ------------------------------------------------------------------------------------------
// LDS accesses of ONE wave are executed in order; what has to be stopped is the compiler moving them
__device__ __forceinline__ void waveLdsSync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef GVPM_TRAV_WPB
#define GVPM_TRAV_WPB 1
#endif
// waves per traversal workgroup: they share nothing (a slice of LDS each, wave-local synchronisation); one
// workgroup of 4 waves costs the dispatcher what a workgroup of one wave does, and 22 000 single-wave workgroups per
// launch were dispatch-bound (2.4 resident waves per SIMD on average where registers and LDS allow 5)
constexpr int TRAV_WPB = GVPM_TRAV_WPB;
#ifndef GVPM_TRAV_MINW
#define GVPM_TRAV_MINW 4  // (128 VGPRs; 3 -- 147 -- is 2 % slower on the pipelined C2 step, 5 -- 96, 60 spilled -- equal)
#endif
// Work units of the evaluation (round 6).  The planner balances the TRAVERSAL (items of equal staged photons); the pairs an
// item yields vary by more than ten (C2: mean 450, the items in front of the light 4 500+), and the evaluation took one item per
// wave at a time in index order: with ~7 items a wave the last one decided when a wave ended -- the waves finished anywhere
// between 0.47 and 1.19 ms of a 1.19 ms kernel (probe build), mean 0.75: a third of the wave-time was the tail.  The
// traversal now files every item's pairs as parts of at most EVAL_UNIT pairs in two lists -- parts above EVAL_UNIT_SMALL
// pairs, and the small ones -- and the evaluation's queue serves the large list first: the units that end the kernel are small.
#ifndef GVPM_EVAL_UNIT
#define GVPM_EVAL_UNIT 1280
#endif
#ifndef GVPM_EVAL_UNIT_SMALL
#define GVPM_EVAL_UNIT_SMALL 256
#endif
constexpr uint32_t EVAL_UNIT = GVPM_EVAL_UNIT, EVAL_UNIT_SMALL = GVPM_EVAL_UNIT_SMALL;
// the staged photons, one array per component: a lane tests FOUR consecutive photons against its beam, read with four
// ds_read_b128 issued together, two photons per packed-fp32 instruction
struct alignas(16) TravLds {
  float x[STAGE], y[STAGE], z[STAGE];
  uint32_t bits[STAGE];
  uint32_t stageIdx[STAGE];
  float4 cyl[3];  // the tile's cylinder: {axis point, R^2} {axis direction, s0} {s1}
};
typedef float v2f __attribute__((ext_vector_type(2)));

template <int B>
__global__ __launch_bounds__(64 * TRAV_WPB) __attribute__((amdgpu_waves_per_eu(GVPM_TRAV_MINW))) void traverse_bre_kernel(GatherArgs a, const uint4 *__restrict__ items,
                                                          const uint2 *__restrict__ itemOff,
                                                          const uint32_t *__restrict__ itemCount, uint32_t *queueHead,
                                                          uint32_t *__restrict__ pairs, uint32_t *__restrict__ pairCnt,
                                                          uint32_t persistent, uint2 *__restrict__ units, uint32_t *unitCtl,
                                                          uint32_t unitCap) {
  constexpr int LPB = 64 / B;
  __shared__ TravLds sAll[TRAV_WPB];
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (the compiler must see it is wave-uniform)
  TravLds &s = sAll[wv];
  const int lane = threadIdx.x & 63;
  if (itemCount[7] != 0u) return;  // an optimistic step whose buffers were too small (grid_build.hip, TailArgs): queued again
  const uint32_t nItems = *itemCount;
  const int b = lane % B, sub = lane / B;
  const float r = a.radius;
  const float r2f = r * r;
  const float eps = a.cfg.epsilon;
  const uint32_t coalesceAt = a.cfg.reserved[3] ? (uint32_t)a.cfg.reserved[3] : 512u;  // photons in a box row set
  unsigned long long nCand = 0, nOver = 0;

  // One item per wave (the grid is the item count: the hardware dispatcher balances the load and, at every workgroup
  // boundary, lets the other streams' kernels in by priority -- persistent waves hold their registers until the whole
  // kernel ends, which stretched the next step's build, a chain of twenty small kernels, from 0.55 to 1.3 ms).
  // `persistent` (GVPM_PERSISTENT=1): waves loop over items; the first is the wave's own index, a shared counter
  // (one address: ~11 ns per atomic whatever the number of waves) serves the rest.
  bool firstItem = true;
  for (;;) {
    uint32_t it = blockIdx.x * TRAV_WPB + (uint32_t)wv;
    if (!firstItem) {
      // (a grid that covers the items -- the usual case of an optimistic step, whose grid is a guess -- asks the queue nothing)
      if (!persistent || gridDim.x * TRAV_WPB >= nItems) break;
      if (lane == 0) it = gridDim.x * TRAV_WPB + atomicAdd(queueHead, 1u);
      it = __shfl(it, 0, 64);
    }
    firstItem = false;
    it = (uint32_t)__builtin_amdgcn_readfirstlane((int)it);  // (wave-uniform: the item's record and region in scalar registers)
    if (it >= nItems) break;
    const uint4 item = items[it];
    const uint32_t setBase = item.x, nb = item.y & 0xFFu, chunk = item.y >> 8;
    if (nb == 0) continue;
    // the item's region: one list of up to `cap` photon indices per beam of the tile
    const uint2 reg = itemOff[it];
    const uint32_t cap = reg.y;
    uint32_t *out = pairs + (size_t)reg.x * 64u + (size_t)b * cap;
    uint32_t mine = 0;  // hits of this lane's beam so far (equal in the LPB lanes of the beam)
    BaseInfo bi;
    const RayReg base = loadBaseDirect<B>(a, setBase, nb, lane, bi);
    TileWalk w;
    tileSetupFrom(a, base, base.valid, w);
    const bool beamValid = w.beamValid;
    const float mint = eps, maxt = base.len - eps;
    // depth + edge within [minDepth, maxDepth] (shift_volume_photon.cpp:670-673) as a window on the photon's depth
    const int dmax = a.cfg.max_depth > 0 ? a.cfg.max_depth - (int)bi.edge : 0x7FFFFFFF;
    const int dmin = a.cfg.min_depth != 0 ? a.cfg.min_depth - (int)bi.edge : -0x7FFFFFFF;
    const uint32_t pixParity = ((bi.pix & 0xFFFFu) + (bi.pix >> 16)) & 1u;
    const uint32_t fmask = 0x40u | (a.cfg.path_set ? (1u << GVPM_HOT_PARITY_BIT) : 0u);
    const uint32_t fwant = 0x40u | (a.cfg.path_set ? (pixParity << GVPM_HOT_PARITY_BIT) : 0u);
    const bool depthWindow = a.cfg.max_depth > 0 || a.cfg.min_depth != 0;  // wave-uniform
    int dlo = max(dmin, 0), dhi = min(dmax, 255);
    if (dhi < dlo) dlo = dhi = 256;  // nothing passes
    const uint32_t dspan = (uint32_t)(dhi - dlo);
    // thresholds of the conservative test.  The fp32 error of disk is below 6e-7 (|wv|_1 + |disk|) (hitBand) with
    // wv = photon - origin and |disk| <= |wv|_1.  Only pairs the reference accepts must pass, and for those the photon
    // lies within 3 r of the beam's segment: |wv|_1 <= sqrt(3) (len + 3 r), whatever the grid or the other photons are
    const float wv1 = 1.7321f * (fmaxf(base.len, 0.f) + 3.f * r);
    const float Emax = 1.25e-6f * wv1;
    const float thrD2 = beamValid ? r2f + (4.f * r * Emax + r2f * 2e-6f) : -1.f;
    const float thrLo = mint - Emax, thrHi = maxt + 2.f * r;
    // The box of a slab step holds ~3 x the photons of the tile's bounding cylinder (tile_walk.h tileCylinder): a staged
    // window is compacted to the cylinder's photons before the 16-beam pass tests it (GVPM_TRAV_PREFILTER=0: as staged).
    const bool prefilter = !(a.cfg.reserved[0] & 128);
    bool cylOk = false;
    if (prefilter) {
      // foot points of accepted photons have ray parameters in [thrLo, thrHi]; an accepted photon is within
      // sqrt(thrD2) <= r (1 + 1e-6) + 2 Emax of its foot point
      const TileCyl c = tileCylinder(base, beamValid, fminf(thrLo, 0.f) - r, thrHi + r, r, 2.f * Emax);
      cylOk = c.ok;
      // (nine wave-uniform numbers used once per staged window: parked in LDS rather than held in registers beside the
      // test loop -- the kernel's register budget is what lets the build's kernels run beside it)
      waveLdsSync();
      if (lane == 0) {
        s.cyl[0] = make_float4(c.o.x, c.o.y, c.o.z, c.R2);
        s.cyl[1] = make_float4(c.d.x, c.d.y, c.d.z, c.s0);
        s.cyl[2] = make_float4(c.s1, 0.f, 0.f, 0.f);
      }
      waveLdsSync();
    }
    const bool haveCyl = prefilter && __builtin_amdgcn_readfirstlane((int)cylOk);
    unsigned long long tested = 0;  // wave-uniform
    auto putStage = [&](uint32_t k, float4 v, uint32_t gi) {
      s.x[k] = v.x;
      s.y[k] = v.y;
      s.z[k] = v.z;
      s.bits[k] = __float_as_uint(v.w);
      s.stageIdx[k] = gi;
    };

    // (bundle cells: a heavy item is split into parts that take its staging windows round-robin, tile_walk.h)
    const bool bundle = a.grid.mode == 1;
    const uint32_t part = bundle ? (item.z >> 8) & 0xFFFu : 0u, parts = bundle ? max(item.z >> 20, 1u) : 1u;
    uint32_t winNo = 0;
    const int cBeg = max((int)(bundle ? item.z & 0xFFu : item.z), w.cA0), cEnd = min((int)item.w, w.cA1);
    for (int cA = cBeg; cA <= cEnd; cA += w.K) {
      const int cAe = min(cA + w.K - 1, cEnd);
      CellBox bx;
      // the slab's cell box: the planner's (it reduced the tile's footprints to count the box's photons), one load
      // instead of a footprint per lane, four wave reductions and the cell arithmetic per slab step
      const uint32_t stepIdx = (uint32_t)(cA - w.cA0) / (uint32_t)w.K;
      if (a.planBoxes && stepIdx < a.planBoxStride) {
        if (!unpackCellBox(a.planBoxes[(size_t)chunk * a.planBoxStride + stepIdx], bx)) continue;
      } else if (!slabBox(a, w, cA, cAe, bx)) {
        continue;
      }
      const int nranges = (bx.by1 - bx.by0 + 1) * (bx.bz1 - bx.bz0 + 1);
      for (int rbase = 0; rbase < nranges; rbase += 64) {
        uint32_t start, count;
        boxRange(a, bx, rbase + lane, nranges, start, count);
        const uint32_t incl = wave_scan_incl(count, lane);
        const uint32_t excl = incl - count;
        const uint32_t total = __shfl(incl, 63, 64);
        for (uint32_t win = 0; win < total; win += STAGE) {
          if (parts > 1u && (winNo++ % parts) != part) continue;
          // stage [win, win + STAGE) of the concatenated ranges
          waveLdsSync();
          const uint32_t nst = min((uint32_t)STAGE, total - win);
          if (total >= coalesceAt) {
            // dense boxes (C4: 4 M photons): consecutive LANES take consecutive entries of the window (the range an
            // entry falls in: a 6-step search over the exclusive scan, through ds_bpermute), so a load instruction
            // reads a few contiguous runs of records instead of 64 separate ones
#pragma unroll
            for (uint32_t k = (uint32_t)lane; k < (uint32_t)STAGE; k += 64u) {
              const uint32_t e = win + k;
              uint32_t rr = 0;
#pragma unroll
              for (uint32_t step = 32; step; step >>= 1) {
                const uint32_t cand = rr + step;
                const uint32_t v = (uint32_t)__shfl((int)excl, (int)(cand & 63u), 64);
                if (v <= e) rr = cand;
              }
              const uint32_t rStart = (uint32_t)__shfl((int)start, (int)rr, 64), rExcl = (uint32_t)__shfl((int)excl, (int)rr, 64);
              if (k < nst) {
                const uint32_t gi = rStart + (e - rExcl);
                putStage(k, a.hot[gi], gi);
              }
            }
          } else
          {
            const uint32_t lo_i = max(excl, win), hi_i = min(excl + count, win + STAGE);
            uint32_t i = lo_i;
            // (one record per lane and trip: four loads in flight per lane, as this loop had them, cost 44 VGPRs -- a
            // wave per SIMD -- and the kernel a fifth of its speed)
            for (; i < hi_i; ++i) {
              const uint32_t gi = start + (i - excl);
              putStage(i - win, a.hot[gi], gi);
            }
          }
          waveLdsSync();
          uint32_t nkeep = nst;
          if (haveCyl) {
            // compaction in place, 64 entries a round: a round's survivors go to slots at or below the round's own (every
            // lane holds its entry in registers before anything is written; later rounds' slots are not touched)
            nkeep = 0;
            const float4 c0 = s.cyl[0], c1 = s.cyl[1];
            const float cS1 = s.cyl[2].x;
            const f3 axO = mk3(c0.x, c0.y, c0.z), axD = mk3(c1.x, c1.y, c1.z);
            const float cylR2 = c0.w, cylS0 = c1.w, cylS1 = cS1;
            for (uint32_t k0 = 0; k0 < nst; k0 += 64u) {
              const uint32_t k = k0 + (uint32_t)lane;
              const bool live = k < nst;
              const float px = live ? s.x[k] : 0.f, py = live ? s.y[k] : 0.f, pz = live ? s.z[k] : 0.f;
              const uint32_t pb = live ? s.bits[k] : 0u, pi = live ? s.stageIdx[k] : 0u;
              const f3 wv = mk3(px, py, pz) - axO;
              const float sq = dot(wv, axD);
              const float d2 = dot(wv, wv) - sq * sq;
              const bool keep = live && d2 < cylR2 && sq > cylS0 && sq < cylS1 && (pb & 0x40u);
              const unsigned long long m = __ballot(keep);
              waveLdsSync();
              if (keep) {
                const uint32_t dst = nkeep + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                s.x[dst] = px; s.y[dst] = py; s.z[dst] = pz; s.bits[dst] = pb; s.stageIdx[dst] = pi;
              }
              nkeep += (uint32_t)__popcll(m);
            }
            waveLdsSync();
          }
          // the slots between the last photon and the next multiple of 16 hold photons no beam can meet
          if (lane < 16 && nkeep + (uint32_t)lane < ((nkeep + 15u) & ~15u)) s.x[nkeep + lane] = 3.0e38f;
          waveLdsSync();
          tested += nkeep;
          // groups of G * LPB staged photons: sub-lane `sub` of a beam takes G consecutive ones.  A branch-free pass
          // marks the candidates, then the wave appends them one per lane and round.
          constexpr uint32_t G = 4;
          static_assert(STAGE % (G * LPB) == 0 && G == 4, "a lane reads four consecutive photons with one b128 per component");
          for (uint32_t jb = 0; jb < nkeep; jb += G * LPB) {  // wave-uniform trip count: the append below is collective
            const uint32_t j0 = jb + (uint32_t)sub * G;
            const float4 X = *reinterpret_cast<const float4 *>(&s.x[j0]);
            const float4 Y = *reinterpret_cast<const float4 *>(&s.y[j0]);
            const float4 Z = *reinterpret_cast<const float4 *>(&s.z[j0]);
            const uint4 Bt = *reinterpret_cast<const uint4 *>(&s.bits[j0]);
            uint32_t cm = 0;
            {
              const v2f ox = {base.o.x, base.o.x}, oy = {base.o.y, base.o.y}, oz = {base.o.z, base.o.z};
              const v2f dx = {base.d.x, base.d.x}, dy = {base.d.y, base.d.y}, dz = {base.d.z, base.d.z};
              const float xs[4] = {X.x, X.y, X.z, X.w}, ys[4] = {Y.x, Y.y, Y.z, Y.w}, zs[4] = {Z.x, Z.y, Z.z, Z.w};
              const uint32_t bs[4] = {Bt.x, Bt.y, Bt.z, Bt.w};
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                const v2f wx = (v2f){xs[2 * h], xs[2 * h + 1]} - ox, wy = (v2f){ys[2 * h], ys[2 * h + 1]} - oy,
                          wz = (v2f){zs[2 * h], zs[2 * h + 1]} - oz;
                const v2f disk = wx * dx + (wy * dy + wz * dz);
                const v2f vx = wx - dx * disk, vy = wy - dy * disk, vz = wz - dz * disk;
                const v2f d2 = vx * vx + (vy * vy + vz * vz);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                  const int u = 2 * h + e;
                  // conservative: every pair the reference accepts passes (its disk test within the band -- folded into
                  // the thresholds, from the largest |photon - origin| the grid allows; beyond the beam end the
                  // own-box test admits diskDistance up to maxt + sqrt(3) r)
                  uint32_t ok = (uint32_t)(d2[e] < thrD2) & (uint32_t)(disk[e] > thrLo) & (uint32_t)(disk[e] < thrHi);
                  // the exact filters, shift_volume_photon.cpp:670-697: computeVolumeContribution + debugShift (bit 6,
                  // folded by grid_build), checkerboard parity (bit 7), depth window
                  ok &= (uint32_t)((bs[u] & fmask) == fwant);
                  if (depthWindow) ok &= (uint32_t)((uint32_t)((int)GVPM_PF_DEPTH(bs[u]) - dlo) <= dspan);
                  cm |= ok << u;
                }
              }
            }
            if (__ballot(cm != 0u)) {
              // append: the LPB sub-lanes of a beam (lanes b, b + B, ...) write their candidates behind one another.
              // (consecutive photons are neighbours in space, so one lane often holds several of a beam's hits: a
              // collective round per hit would cost the whole wave a round for each)
              const uint32_t c = (uint32_t)__popc(cm);
              uint32_t before = 0, total = c;
              if (B <= 32) {
                const uint32_t t = (uint32_t)__shfl_xor((int)total, 32, 64);
                before += (lane & 32) ? t : 0u;
                total += t;
              }
              if (B == 16) {
                const uint32_t t = (uint32_t)__shfl_xor((int)c, 16, 64);
                // lanes with bit 4 set come after their partner; the pair with bit 5 set after the pair without
                const uint32_t pairSum = c + t;
                const uint32_t t2 = (uint32_t)__shfl_xor((int)pairSum, 32, 64);
                before = ((lane & 16) ? t : 0u) + ((lane & 32) ? t2 : 0u);
                total = pairSum + t2;
              }
              uint32_t off = mine + before;
              while (cm) {
                const uint32_t u = (uint32_t)__ffs(cm) - 1u;
                cm &= cm - 1u;
                if (off < cap) out[off] = s.stageIdx[j0 + u];
                else nOver++;
                ++off;
              }
              mine += total;
            }
          }
        }
      }
    }
    if (sub == 0) pairCnt[(size_t)it * B + b] = (uint32_t)b < nb ? min(mine, cap) : 0u;
    if (units) {
      // the evaluation's WORK UNITS (round 6): this item's pairs, cut into parts of at most EVAL_UNIT -- see evaluate_bre_kernel
      uint32_t tot = (sub == 0 && (uint32_t)b < nb) ? min(mine, cap) : 0u;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) tot += (uint32_t)__shfl_xor((int)tot, o, 64);
      tot = (uint32_t)__builtin_amdgcn_readfirstlane((int)tot);
      if (tot) {
        const uint32_t parts = (tot + EVAL_UNIT - 1u) / EVAL_UNIT;
        const uint32_t cls = (tot + parts - 1u) / parts > EVAL_UNIT_SMALL ? 0u : 1u;
        uint32_t slot = 0;
        if (lane == 0) slot = atomicAdd(&unitCtl[cls], parts);
        slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
        for (uint32_t pp = (uint32_t)lane; pp < parts; pp += 64u) {
          if (slot + pp < unitCap) units[(size_t)cls * unitCap + slot + pp] = make_uint2(it, pp | (parts << 16));
          else nOver++;
        }
      }
    }
    nCand += tested * nb;  // (pairs the 16-beam pass tested)
  }
  {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nOver += __shfl_xor(nOver, o, 64);
    if (lane == 0 && (nCand | nOver)) {
      unsigned long long *row = statRow(a);
      atomicAdd(&row[1], nCand);
      if (nOver) atomicAdd(&row[7], nOver);  // must stay 0: the planner's bound is exact
    }
  }
}

// ------------------------------------------------------------------------------------------
// evaluation: persistent waves, one item at a time, the two phases in SEPARATE loops.
//
// A wave walks a SEGMENT of its item's pairs through the decision + phase 1 only (<= SEG_STEPS steps, or until
// the queue could overflow), appending the reconnections it meets to ONE compact queue (ballot + popcount, 2 bytes
// per entry: step, lane, shift, beam), folds its register sums into the LDS accumulators, and then runs the queue
// through phase 2 in a dense loop of its own: every lane takes an equal, contiguous share of the queue, whose
// entries come in runs of one (beam, shift), so a lane keeps 6 sums in registers and touches the LDS accumulators
// once per run.  (Round 1 interleaved the reconnections with the phase-1 walk: the 27 register sums of phase 1
// stayed live across phase 2, the allocation was the union of both -- 187 VGPRs, two waves per SIMD -- and each
// wave held per-lane queues of 4 KB.)  Here the allocation is the larger of the two loops, the queue is a quarter
// of the size, and the 4 waves of a workgroup share the staged occluders: 12 KB of LDS per wave, 3 waves per SIMD.
// Measured at C2 (MI355X, isolated): 1.09 -> 0.78 ms.
// ------------------------------------------------------------------------------------------
#ifndef GVPM_EVAL_MINW
#define GVPM_EVAL_MINW 3
#endif
#ifndef GVPM_EVAL_ATTR
#define GVPM_EVAL_ATTR
#endif
constexpr int SEG_STEPS = 16;   // steps per segment (4 bits of a queue entry)
constexpr int SEG_QCAP = 1024;  // queue entries per wave; a step appends at most 4 * 64
template <int B> struct SegCfg {
  // waves per workgroup: they share nothing but the staged occluders
  static constexpr int WPB = B == 16 ? 4 : (B == 32 ? 2 : 1);
  using Entry = typename std::conditional<B == 16, uint16_t, uint32_t>::type;
  static constexpr int BEAM_BITS = B == 16 ? 4 : 6;
};
template <int B> struct SegLds : RayTile<B> {
  double acc[27][B];
  uint32_t boff[B + 1];
  typename SegCfg<B>::Entry q[SEG_QCAP];  // step | lane | shift | beam
  uint32_t setBase;                       // of the item (host-shift requests name the beam set)
  // (the shifted rays are kept RELATIVE to their base ray in the ray tile's own slots, with sensorMIS: relToBase)
};


#ifdef GVPM_EVAL_TIMING
// probe builds only: shader-clock ticks per part of the evaluation kernel, summed over waves
// [0] wave lifetime [1] item header + LDS setup [2] decision + phase 1 [3] phase 2 [4] late pass bookkeeping [5] write-out [6] items
__device__ unsigned long long gvpmEvalTiming[16];  // [12] longest wave [13] longest item [14] waves
// wall clock (100 MHz, the same on every XCD) at the start and the end of every wave of the LAST launch, and its units
__device__ unsigned long long gvpmEvalWaveLog[16 * 16384];
extern "C" int gvpm_debug_eval_timing(unsigned long long *out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(gvpmEvalTiming), sizeof(gvpmEvalTiming)) != hipSuccess) return -1;
  if (out && hipMemcpyFromSymbol(out + 16, HIP_SYMBOL(gvpmEvalWaveLog), sizeof(gvpmEvalWaveLog)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(gvpmEvalTiming), z, sizeof(z)) != hipSuccess) return -1;
  }
  return 0;
}
__device__ __forceinline__ unsigned long long tickNow() {
  unsigned long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  return t;
}
#define TICK() tickNow()
__device__ __forceinline__ unsigned long long tickLight() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  return t;
}
#define LTICK() tickLight()
#else
#define TICK() 0ull
#define LTICK() 0ull
#endif

template <int B, bool FULLVIS, bool PF, bool HS = false>
__global__ __launch_bounds__(64 * SegCfg<B>::WPB, GVPM_EVAL_MINW * SegCfg<B>::WPB / 4 > 0 ? GVPM_EVAL_MINW * SegCfg<B>::WPB / 4 : 1) GVPM_EVAL_ATTR
void evaluate_bre_kernel(GatherArgs a, const uint4 *__restrict__ items, const uint2 *__restrict__ itemOff,
                             const uint32_t *__restrict__ itemCount, uint32_t *queueHead,
                             const uint32_t *__restrict__ pairs, const uint32_t *__restrict__ pairCnt,
                             uint32_t persistent, const uint2 *__restrict__ units, const uint32_t *__restrict__ unitCtl,
                             uint32_t unitCap) {
  constexpr int WPB = SegCfg<B>::WPB;
  constexpr int BB = SegCfg<B>::BEAM_BITS;
  using Entry = typename SegCfg<B>::Entry;
  __shared__ SegLds<B> sAll[WPB];
  extern __shared__ float4 sceneTri[];  // the occluders of a small scene (48 bytes each), shared by the waves
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (itemCount[7] != 0u) return;  // (see traverse_bre_kernel)
  SegLds<B> &s = sAll[wv];
  const float4 *ldsTri = nullptr;
  if (!FULLVIS && a.ntri <= EVAL_LDS_TRIS && !(a.cfg.reserved[0] & 16)) {
    for (uint32_t i = threadIdx.x; i < 3u * a.ntri; i += 64u * WPB) sceneTri[i] = a.tri4[i];
    ldsTri = sceneTri;
  }
  __syncthreads();  // the only workgroup barrier: from here on the waves run independently
  const uint32_t nItems = *itemCount;
  const uint32_t waveId = blockIdx.x * WPB + wv, nWaves = gridDim.x * WPB;
  uint32_t nEval = 0, nNull = 0, nDiff = 0, nFail = 0;
  [[maybe_unused]] unsigned long long tk[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  [[maybe_unused]] const unsigned long long tStart = TICK();
  [[maybe_unused]] unsigned long long tItemMax = 0;
#ifdef GVPM_EVAL_TIMING
  const unsigned long long wStart = wall_clock64();
  unsigned long long wLastStart = 0, lastN = 0, lastUk = 0, wLastEnd = 0, tkLast[6] = {0, 0, 0, 0, 0, 0}, suLast[4] = {0, 0, 0, 0};
#endif

  // (Measured in round 6 and dropped, as round 3 had for whole items: asking the queue for the NEXT entry while the current one
  // is evaluated -- the returning atomic is a quarter of a unit's set-up -- leaves every wave sitting on a reserved unit when
  // the queue runs dry: single stream 0.593 -> 0.636 ms, the waves' end times 0.55-1.09 ms instead of 0.58-0.99.)
  bool firstItem = true;
  for (;;) {
    [[maybe_unused]] const unsigned long long tItem = TICK();
    uint32_t it = waveId;
    if (!firstItem) {
      if (!persistent) break;  // one item per wave (see traverse_bre_kernel)
      if (lane == 0) it = nWaves + atomicAdd(queueHead, 1u);
      it = __shfl(it, 0, 64);
    }
    firstItem = false;
    // (wave-uniform, and SAID so: the item's record, its pair region and everything derived from them then live in scalar
    // registers -- as vector registers they were ten of the values this kernel spills around its hot loop, round 5)
    it = (uint32_t)__builtin_amdgcn_readfirstlane((int)it);
    uint32_t part = 0u, parts = 1u;
    if (units) {
      // entry `it` of the unit queue: the list of large parts, then the small ones (see EVAL_UNIT)
      const uint32_t c0 = min(unitCtl[0], unitCap), c1 = min(unitCtl[1], unitCap);
      if (it >= c0 + c1) break;
      const uint2 u = it < c0 ? units[it] : units[(size_t)unitCap + (it - c0)];
      it = u.x;
      part = u.y & 0xFFFFu;
      parts = u.y >> 16;
    }
    if (it >= nItems) break;
    const uint4 item = items[it];
    const uint32_t setBase = item.x, nb = item.y & 0xFFu;
    if (nb == 0) continue;
    const uint32_t cntb = (uint32_t)lane < nb ? pairCnt[(size_t)it * B + lane] : 0u;
    const uint32_t incl = wave_scan_incl(cntb, lane);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (total == 0) continue;
    // the wave's share [r0, r0 + n) of the item's concatenated lists
    const uint32_t per = (total + parts - 1u) / parts;
    const uint32_t r0 = part * per;
    if (r0 >= total) continue;
    const uint32_t n = min(per, total - r0);
#ifdef GVPM_EVAL_TIMING
    wLastStart = wall_clock64();
    lastN = n;
    lastUk = it;
    for (int k = 1; k <= 5; ++k) tkLast[k] = tk[k];  // ([1]: this unit's setup is added after this point)
    const unsigned long long su0 = TICK();
#endif
    const uint2 reg = itemOff[it];
    const uint32_t *lists = pairs + (size_t)reg.x * 64u;
    const uint32_t cap = reg.y;
    waveLdsSync();
    if (lane < B) s.boff[lane + 1] = incl;
    if (lane == 0) s.boff[0] = 0u;
    if (lane == 0) s.setBase = setBase;
#ifdef GVPM_EVAL_TIMING
    const unsigned long long su1 = TICK();
#endif
    loadTileRaysNoSync<B>(a, s, setBase, nb, lane);
#ifdef GVPM_EVAL_TIMING
    const unsigned long long su2 = TICK();
#endif
    for (int idx = lane; idx < 27 * B; idx += 64) (&s.acc[0][0])[idx] = 0.0;
    waveLdsSync();
    relToBase<B>(s, lane);
    waveLdsSync();
#ifdef GVPM_EVAL_TIMING
    suLast[0] = su0 - tItem; suLast[1] = su1 - su0; suLast[2] = su2 - su1; suLast[3] = TICK() - su2;
#endif
    [[maybe_unused]] unsigned long long tMark = TICK();
    tk[1] += tMark - tItem;
    tk[6] += 1;

    const bool use3D = a.cfg.vol_technique == GVPM_VOL_BRE3D;
    // my chunk [g0, g1) of the concatenated per-beam lists (lane l of ANY wave position: g0 = min(total, l * chunk))
    const uint32_t chunk = (n + 63u) / 64u;
    const uint32_t g0 = r0 + min(n, (uint32_t)lane * chunk), g1 = min(r0 + n, g0 + chunk);
    uint32_t cur = 0;  // current beam
    if (g0 < g1)
      while (s.boff[cur + 1] <= g0) cur++;

    for (uint32_t tSeg = 0; tSeg < chunk;) {
      // One pass per segment (round 5): a pair whose HIT the fp32 bands cannot decide (decidePair: ~1e-5 of the candidates)
      // goes to the exact pass behind this kernel (exact_shift.hip, GVPM_EX_KIND_BRE_PAIR: the reference predicate, the base
      // term and the four shifts in fp64) instead of through an fp64 predicate and a second round of both loops in here.
      {
        // ---- decision + phase 1 ----
        uint32_t qn = 0;  // wave-uniform
        Acc27 acc;
#pragma unroll
        for (int k = 0; k < 27; ++k) acc.v[k] = 0.f;
        bool dirty = false;  // acc holds sums of beam `cur`
        const uint32_t tLim = chunk;
        uint32_t t = tSeg;
        // the pair of a lane at step tt: entry g of the concatenated lists, its beam, its photon index.
        // The index of step t + 1 is fetched while step t computes: one dependent global load less per step.
        auto locate = [&](uint32_t tt, uint32_t from, bool &hv, uint32_t &gg, uint32_t &bb, uint32_t &pi) {
          gg = g0 + tt;
          hv = gg < g1 && tt < tLim;
          bb = from;  // (a lane's own pairs come in ascending beam order)
          pi = 0u;
          if (hv) {
            while (s.boff[bb + 1] <= gg) bb++;
            pi = lists[(size_t)bb * cap + (gg - s.boff[bb])];
          }
        };
        // PF (maps beyond the Infinity Cache): the photon INDEX of step t + 2 and the RECORD (its first 48 bytes) of step
        // t + 1 are in flight while step t computes; otherwise the index of step t + 1 only
        bool haveN = false, haveB = false;
        uint32_t gN = 0, bN = 0, pidxN = 0, gB = 0, bB = 0, pidxB = 0;
        locate(t, cur, haveN, gN, bN, pidxN);
        PhotonFront phN = {};
        if (PF) {
          locate(t + 1u, bN, haveB, gB, bB, pidxB);
          phN = loadFront(a, pidxN);  // (index 0 when the lane has no pair: a valid record, not used)
        }
        for (; t < tLim && t - tSeg < (uint32_t)SEG_STEPS && qn + 256u <= (uint32_t)SEG_QCAP; ++t) {
          [[maybe_unused]] const unsigned long long l0 = LTICK();
          const bool have = haveN;
          const uint32_t b = bN, pidx = pidxN;
          const PhotonFront phCur = phN;
          if (PF) {
            haveN = haveB; gN = gB; bN = bB; pidxN = pidxB;
            phN = loadFront(a, pidxN);
            locate(t + 2u, bN, haveB, gB, bB, pidxB);
          } else {
            locate(t + 1u, b, haveN, gN, bN, pidxN);
          }
          [[maybe_unused]] const unsigned long long l1 = LTICK();
          tk[7] += l1 - l0;
          uint32_t qMask = 0;
          [[maybe_unused]] unsigned long long l2 = l1, l3 = l1;
          if (have) {
            if (b != cur) {
              if (dirty) flushAcc<B>(s, acc, cur);
              dirty = false;
              cur = b;
            }
            l2 = LTICK();
            const PhotonFront ph = PF ? phCur : loadFront(a, pidx);
            const RayReg base = loadRay(s, 0, cur);
            int dec = decidePair(ph.pos, base, s.rnd[cur], a.radius, a.cfg.epsilon, use3D);
            if (dec == 2) dec = decidePairFine(ph.pos, base, s.rnd[cur], a.radius, a.cfg.epsilon, use3D);
#ifdef GVPM_EVAL_TIMING
            asm volatile("" :: "v"(dec));
            l3 = LTICK();
#endif
            if (dec == 1) {
              evalPhase1<B, HS>(a, s, ph, base, cur, acc, nNull, nFail, qMask);
              dirty = true;
              nEval++;
            } else if (dec == 2) {
              // the bands cannot decide this pair: the exact pass takes it whole (predicate, base term, four shifts)
              deferNote(a, GVPM_EX_KIND_BRE_PAIR, a.setPerm[s.setBase + cur], pidx, 0u, 1u);
            }
          }
          const uint32_t ent = ((t - tSeg) << (8 + BB)) | ((uint32_t)lane << (2 + BB)) | cur;
#pragma unroll
          for (uint32_t i = 0; i < 4u; ++i) {
            const bool qd = (qMask >> i) & 1u;
            const unsigned long long m = __ballot(qd);
            if (qd) s.q[qn + __popcll(m & ((1ull << lane) - 1ull))] = (Entry)(ent | (i << BB));
            qn += (uint32_t)__popcll(m);
          }
#ifdef GVPM_EVAL_TIMING
          { const unsigned long long l5 = LTICK(); tk[8] += l2 - l1; tk[9] += l3 - l2; tk[10] += l5 - l3; tk[11] += 1; }
#endif
        }
        if (dirty) flushAcc<B>(s, acc, cur);
        const uint32_t tEnd = t;  // first step of the next segment
        waveLdsSync();
        { [[maybe_unused]] const unsigned long long tn = TICK(); tk[2] += tn - tMark; tMark = tn; }
        // ---- phase 2 over the queue: lane l takes entries [l * cq, (l + 1) * cq) ----
        const uint32_t cq = (qn + 63u) / 64u;
        const uint32_t e0 = min(qn, (uint32_t)lane * cq), e1 = min(qn, e0 + cq);
        uint32_t key = 0xFFFFFFFFu;
        f3 rs = mk3(0.f), rw = mk3(0.f);
        // (the photon index of entry j + 1 is fetched while entry j computes)
        auto entry = [&](uint32_t j, bool &hv, uint32_t &k2o, uint32_t &pi, uint32_t &bo, uint32_t &io) {
          hv = j < cq && e0 + j < e1;
          k2o = 0xFFFFFFFFu;
          pi = bo = io = 0u;
          if (hv) {
            const uint32_t e = s.q[e0 + j];
            bo = e & ((1u << BB) - 1u);
            io = (e >> BB) & 3u;
            const uint32_t ln = (e >> (2 + BB)) & 63u, ts = e >> (8 + BB);
            const uint32_t g = r0 + min(n, ln * chunk) + tSeg + ts;
            pi = lists[(size_t)bo * cap + (g - s.boff[bo])];
            k2o = (bo << 2) | io;
          }
        };
        bool haveE = false, haveF = false;
        uint32_t k2E = 0, pidxE = 0, bE = 0, iE = 0, k2F = 0, pidxF = 0, bF = 0, iF = 0;
        entry(0u, haveE, k2E, pidxE, bE, iE);
        if (PF) entry(1u, haveF, k2F, pidxF, bF, iF);
        for (uint32_t j = 0; j <= cq; ++j) {
          const bool have = haveE;
          const uint32_t k2 = k2E, pidx = pidxE, b = bE, i = iE;
          [[maybe_unused]] float warm = 0.f;
          if (PF) {
            // the index of entry j + 2 in flight, and one word of the record of entry j + 1 touched: the line (the whole
            // 128-byte record) is on its way from HBM while entry j computes
            haveE = haveF; k2E = k2F; pidxE = pidxF; bE = bF; iE = iF;
            warm = reinterpret_cast<const float *>(a.cold + (size_t)pidxE * GVPM_REC_QUADS)[0];
            entry(j + 2u, haveF, k2F, pidxF, bF, iF);
          } else {
            entry(j + 1u, haveE, k2E, pidxE, bE, iE);
          }
          if (k2 != key && key != 0xFFFFFFFFu) {
            // the run of one (beam, shift) ended: its 6 sums go to the LDS accumulators
            const uint32_t kb = key >> 2, ki = key & 3u;
            atomicAdd(&s.acc[3 + 3 * ki + 0][kb], (double)rs.x);
            atomicAdd(&s.acc[3 + 3 * ki + 1][kb], (double)rs.y);
            atomicAdd(&s.acc[3 + 3 * ki + 2][kb], (double)rs.z);
            atomicAdd(&s.acc[15 + 3 * ki + 0][kb], (double)rw.x);
            atomicAdd(&s.acc[15 + 3 * ki + 1][kb], (double)rw.y);
            atomicAdd(&s.acc[15 + 3 * ki + 2][kb], (double)rw.z);
            rs = rw = mk3(0.f);
          }
          key = k2;
          if (have) {
            f3 sf, wb;
            evalPhase2Core<B, FULLVIS, HS>(a, s, pidx, b, (int)i, sf, wb, nDiff, nFail, ldsTri);
            rs = rs + sf;
            rw = rw + wb;
          }
          if (PF) asm volatile("" ::"v"(warm));
        }
        waveLdsSync();
        { [[maybe_unused]] const unsigned long long tn = TICK(); tk[3] += tn - tMark; tMark = tn; }
        tSeg = tEnd;
      }
    }
    // ---- write out: 27 partial sums per beam set into the running sum ----
    for (int idx = lane; idx < 27 * B; idx += 64) {
      const int k = idx / B, bb = idx % B;
      if ((uint32_t)bb < nb) {
        const float v = (float)s.acc[k][bb];
        if (v != 0.f) {
          const uint32_t pv = s.pix[bb];
          const size_t p = (size_t)(pv >> 16) * a.cfg.width + (pv & 0xFFFFu);
          atomicAdd(&a.iter[p * 27 + k], v * a.iterScale);
        }
      }
    }
    { [[maybe_unused]] const unsigned long long tn = TICK(); tk[5] += tn - tMark; if (tn - tItem > tItemMax) tItemMax = tn - tItem; }
#ifdef GVPM_EVAL_TIMING
    wLastEnd = wall_clock64();
#endif
  }
#ifdef GVPM_EVAL_TIMING
  tk[0] = TICK() - tStart;
  if (lane == 0)
  {
#if GVPM_EVAL_TIMING > 1  // (same-address atomics of thousands of ending waves stall the loads of the waves still running)
    for (int k = 0; k < 12; ++k) atomicAdd(&gvpmEvalTiming[k], tk[k]);
    atomicMax(&gvpmEvalTiming[12], tk[0]);
    atomicMax(&gvpmEvalTiming[13], tItemMax);
    atomicAdd(&gvpmEvalTiming[14], 1ull);
#endif
    if (waveId < 16384u) {
      unsigned long long *lg = gvpmEvalWaveLog + 16 * waveId;
      for (int k = 0; k < 4; ++k) lg[12 + k] = suLast[k];
      lg[0] = wStart;
      lg[1] = wall_clock64();
      lg[2] = tk[6];
      lg[3] = wLastStart;
      lg[4] = lastN;
      lg[5] = lastUk;
      for (int k = 1; k <= 5; ++k) lg[5 + k] = tk[k] - tkLast[k];
      lg[11] = wLastEnd;
    }
  }
#endif
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
    if (lane == 0 && ev) {
      unsigned long long *row = a.stats + 8 * (size_t)(waveId % GVPM_STAT_ROWS);
      atomicAdd(&row[0], ev);
      atomicAdd(&row[2], nu);
      atomicAdd(&row[3], di);
      atomicAdd(&row[4], fa);
    }
  }
}

// ------------------------------------------------------------------------------------------
// The PRIMAL beam radiance estimate (second half): BeamRadianceEstimator::query,
// src/integrators/photonmapper/bre.cpp:166-254, as SPPMIntegrator::volumePhotonPassBRE drives it (sppm.cpp:882-1000,
// cameraHeuristic = true: the gradient pass's uniform radius).  A strict subset of the gradient gather: same grid, planner,
// traversal and pair lists; this kernel walks an item's lists like evaluate_bre_kernel but evaluates the primal query's
// term only -- re-based ray (:168), stored power without sigma_s, one random number PER HIT for the 3D kernel (Philox keyed
// by the beam's `rand` and the photon's position bits: the reference's comes from a stateful sampler in traversal order),
// a far check for the 2D kernel (:240-242), no checkerboard, no shifts.  The hit decision is the reference's own, in
// uncontracted fp64 in its operation order (this is not the headline kernel: no banded fp32 fast path), so the evaluated
// set equals the fp64 oracle's (oracle/gvpm_oracle_primal.hpp); radiometry fp32.
// ------------------------------------------------------------------------------------------
template <int B> struct PrimalLds : RayTile<B> {
  double acc[3][B];
  uint32_t boff[B + 1];
};

__device__ __forceinline__ float primalHitRandom(float setRand, f3 pos) {
  uint32_t k0 = __float_as_uint(setRand), k1 = 0x70726d6cu;
  uint32_t c0 = __float_as_uint(pos.x), c1 = __float_as_uint(pos.y), c2 = __float_as_uint(pos.z), c3 = 0u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return (float)(c0 >> 8) * (1.0f / 16777216.0f);
}

// bre.cpp:168-223 / :240-242 for one (photon, beam) pair: accepted?  tR: the distance the transmittance is taken over
// (on the re-based ray), deltaT: half the kernel's chord.
__device__ __forceinline__ bool primalDecide(f3 pf, f3 of, f3 df, float len, float r, float eps, float setRand, bool use3D,
                                             double &tR, double &deltaT) {
#pragma clang fp contract(off)
  const double mint = (double)eps, maxtRay = (double)len - (double)eps;
  const double dx = df.x, dy = df.y, dz = df.z;
  // ray = Ray(r(r.mint), r.d, 0, r.maxt - r.mint)
  const double ox = (double)of.x + dx * mint, oy = (double)of.y + dy * mint, oz = (double)of.z + dz * mint;
  const double maxt = maxtRay - mint;
  const double px = pf.x, py = pf.y, pz = pf.z;
  const double cx = px - ox, cy = py - oy, cz = pz - oz;
  const double disk = cx * dx + cy * dy + cz * dz;
  const double qx = ox + dx * disk, qy = oy + dy * disk, qz = oz + dz * disk;
  const double vx = qx - px, vy = qy - py, vz = qz - pz;
  const double distSqr = vx * vx + vy * vy + vz * vz;
  const double radius = (double)r, radSqr = radius * radius;
  tR = disk;
  deltaT = 0.0;
  if (!(disk > 0 && distSqr < radSqr)) return false;
  if (use3D) {
    if (disk - (radius * 2) > maxt) return false;
    deltaT = sqrt(radSqr - distSqr);
    const double tminKernel = disk - deltaT;
    tR = tminKernel + 2 * deltaT * (double)primalHitRandom(setRand, pf);
    return !(tR < 0 || tR > maxt);
  }
  return !(disk > maxt);
}

template <int B>
__global__ __launch_bounds__(64) void evaluate_primal_kernel(GatherArgs a, const uint4 *__restrict__ items,
                                                             const uint2 *__restrict__ itemOff,
                                                             const uint32_t *__restrict__ itemCount, uint32_t *queueHead,
                                                             const uint32_t *__restrict__ pairs,
                                                             const uint32_t *__restrict__ pairCnt) {
  __shared__ PrimalLds<B> s;
  const int lane = threadIdx.x;
  const uint32_t nItems = *itemCount;
  const bool use3D = a.cfg.vol_technique == GVPM_VOL_BRE3D;
  const float r = a.radius;
  const float kernelVol = use3D ? (4.0f / 3.0f) * 3.14159265358979323846f * r * r * r : 3.14159265358979323846f * r * r;
  const float weight = 1.f / kernelVol;
  unsigned long long nEval = 0;
  bool firstItem = true;
  for (;;) {
    uint32_t it = blockIdx.x;
    if (!firstItem) {
      if (lane == 0) it = gridDim.x + atomicAdd(queueHead, 1u);
      it = __shfl(it, 0, 64);
    }
    firstItem = false;
    if (it >= nItems) break;
    const uint4 item = items[it];
    const uint32_t setBase = item.x, nb = item.y & 0xFFu;
    if (nb == 0) continue;
    const uint32_t cntb = (uint32_t)lane < nb ? pairCnt[(size_t)it * B + lane] : 0u;
    const uint32_t incl = wave_scan_incl(cntb, lane);
    const uint32_t total = __shfl(incl, 63, 64);
    if (total == 0) continue;
    const uint2 reg = itemOff[it];
    const uint32_t *lists = pairs + (size_t)reg.x * 64u;
    const uint32_t cap = reg.y;
    __syncthreads();
    if (lane < B) s.boff[lane + 1] = incl;
    if (lane == 0) s.boff[0] = 0u;
    loadTileRaysNoSync<B>(a, s, setBase, nb, lane);
    for (int idx = lane; idx < 3 * B; idx += 64) (&s.acc[0][0])[idx] = 0.0;
    __syncthreads();
    const uint32_t chunk = (total + 63u) / 64u;
    const uint32_t g0 = min(total, (uint32_t)lane * chunk), g1 = min(total, g0 + chunk);
    uint32_t cur = 0;
    if (g0 < g1)
      while (s.boff[cur + 1] <= g0) cur++;
    f3 sum = mk3(0.f);
    bool dirty = false;
    for (uint32_t g = g0; g < g1; ++g) {
      uint32_t b = cur;
      while (s.boff[b + 1] <= g) b++;
      if (b != cur) {
        if (dirty) {
          atomicAdd(&s.acc[0][cur], (double)sum.x);
          atomicAdd(&s.acc[1][cur], (double)sum.y);
          atomicAdd(&s.acc[2][cur], (double)sum.z);
        }
        sum = mk3(0.f);
        dirty = false;
        cur = b;
      }
      const uint32_t pidx = lists[(size_t)b * cap + (g - s.boff[b])];
      const PhotonFront ph = loadFront(a, pidx);
      const RayReg base = loadRay(s, 0, cur);
      double tR, deltaT;
      if (!primalDecide(ph.pos, base.o, base.d, base.len, r, a.cfg.epsilon, s.rnd[cur], use3D, tR, deltaT)) continue;
      // result += Tr * power * phase(wi, -d) * (weight * scaleFactor) [* max(2 deltaT, 1e-4)]; * beam.weight (sppm.cpp:976-981)
      f3 trT;
      float dummy;
      mediumEval(a.med, (float)tR, trT, dummy);
      const float inv = use3D ? fmaxf(2.f * (float)deltaT, 0.0001f) : 1.f;
      sum = sum + ph.flux * base.eye * (trT.x * phaseEval(a.med.g, ph.wi, -base.d) * weight * inv);
      dirty = true;
      nEval++;
    }
    if (dirty) {
      atomicAdd(&s.acc[0][cur], (double)sum.x);
      atomicAdd(&s.acc[1][cur], (double)sum.y);
      atomicAdd(&s.acc[2][cur], (double)sum.z);
    }
    __syncthreads();
    for (int idx = lane; idx < 3 * B; idx += 64) {
      const int k = idx / B, bb = idx % B;
      if ((uint32_t)bb < nb) {
        const float v = (float)s.acc[k][bb];
        if (v != 0.f) {
          const uint32_t pv = s.pix[bb];
          atomicAdd(&a.iter[((size_t)(pv >> 16) * a.cfg.width + (pv & 0xFFFFu)) * 27 + k], v * a.iterScale);
        }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nEval += __shfl_xor(nEval, o, 64);
  if (lane == 0 && nEval) atomicAdd(&statRow(a)[0], nEval);
}

void launch_evaluate_primal(const GatherArgs &a, int beamsPerWave, const uint4 *items, const uint2 *itemOff,
                            const uint32_t *itemCount, uint32_t *queueHead, const uint32_t *pairs, const uint32_t *pairCnt,
                            uint32_t nwaves, hipStream_t stream) {
  if (a.nsets == 0 || nwaves == 0) return;
  switch (beamsPerWave) {
    case 64: hipLaunchKernelGGL(evaluate_primal_kernel<64>, dim3(nwaves), dim3(64), 0, stream, a, items, itemOff, itemCount, queueHead, pairs, pairCnt); break;
    case 32: hipLaunchKernelGGL(evaluate_primal_kernel<32>, dim3(nwaves), dim3(64), 0, stream, a, items, itemOff, itemCount, queueHead, pairs, pairCnt); break;
    default: hipLaunchKernelGGL(evaluate_primal_kernel<16>, dim3(nwaves), dim3(64), 0, stream, a, items, itemOff, itemCount, queueHead, pairs, pairCnt); break;
  }
}

// itemCount / blockTotal must be zero on entry (memset on the same stream)
void launch_plan_bre(const GatherArgs &a, int beamsPerWave, uint32_t ntiles, uint32_t target, uint4 *items,
                     uint32_t *itemCount, uint2 *itemOff, uint32_t *blockTotal, uint32_t itemCap, hipStream_t stream) {
  if (a.nsets == 0 || ntiles == 0) return;
  // waves stride over the tiles; the stride is prime because image-sharded input owns every N-th tile, and a
  // stride that is a multiple of N would leave all the work to 1/N of the blocks
  const uint32_t nwg = ntiles < 4093u ? ntiles : 4093u;
  switch (beamsPerWave) {
    case 64: hipLaunchKernelGGL(plan_kernel<64>, dim3(nwg), dim3(64), 0, stream, a, ntiles, target, items, itemCount, itemOff, blockTotal, itemCap); break;
    case 32: hipLaunchKernelGGL(plan_kernel<32>, dim3(nwg), dim3(64), 0, stream, a, ntiles, target, items, itemCount, itemOff, blockTotal, itemCap); break;
    default: hipLaunchKernelGGL(plan_kernel<16>, dim3(nwg), dim3(64), 0, stream, a, ntiles, target, items, itemCount, itemOff, blockTotal, itemCap); break;
  }
}

// queueHead must be zero on entry
uint32_t eval_unit_pairs() { return EVAL_UNIT; }
void launch_traverse_bre(const GatherArgs &a, int beamsPerWave, const uint4 *items, const uint2 *itemOff,
                         const uint32_t *itemCount, uint32_t *queueHead, uint32_t *pairs, uint32_t *pairCnt,
                         uint32_t nwaves, bool persistent, hipStream_t stream, uint2 *units, uint32_t *unitCtl, uint32_t unitCap) {
  if (a.nsets == 0 || nwaves == 0) return;
  const uint32_t persist = persistent ? 1u : 0u;
  switch (beamsPerWave) {
    case 64: hipLaunchKernelGGL(traverse_bre_kernel<64>, dim3((nwaves + TRAV_WPB - 1) / TRAV_WPB), dim3(64 * TRAV_WPB), 0, stream, a, items, itemOff, itemCount, queueHead, pairs, pairCnt, persist, units, unitCtl, unitCap); break;
    case 32: hipLaunchKernelGGL(traverse_bre_kernel<32>, dim3((nwaves + TRAV_WPB - 1) / TRAV_WPB), dim3(64 * TRAV_WPB), 0, stream, a, items, itemOff, itemCount, queueHead, pairs, pairCnt, persist, units, unitCtl, unitCap); break;
    default: hipLaunchKernelGGL(traverse_bre_kernel<16>, dim3((nwaves + TRAV_WPB - 1) / TRAV_WPB), dim3(64 * TRAV_WPB), 0, stream, a, items, itemOff, itemCount, queueHead, pairs, pairCnt, persist, units, unitCtl, unitCap); break;
  }
}

template <bool FULLVIS, bool PF, bool HS = false>
static void launchEvaluate(const GatherArgs &a, int beamsPerWave, const uint4 *items, const uint2 *itemOff,
                           const uint32_t *itemCount, uint32_t *queueHead, const uint32_t *pairs, const uint32_t *pairCnt,
                           uint32_t nwaves, bool persistent, hipStream_t stream, const uint2 *units, const uint32_t *unitCtl,
                           uint32_t unitCap) {
  const uint32_t persist = persistent ? 1u : 0u;
  const size_t dyn = (!FULLVIS && a.ntri <= EVAL_LDS_TRIS && !(a.cfg.reserved[0] & 16)) ? (size_t)a.ntri * 48u : 0u;
  switch (beamsPerWave) {
    case 64: hipLaunchKernelGGL((evaluate_bre_kernel<64, FULLVIS, PF, HS>), dim3((nwaves + SegCfg<64>::WPB - 1) / SegCfg<64>::WPB), dim3(64 * SegCfg<64>::WPB), dyn, stream, a, items, itemOff, itemCount, queueHead, pairs, pairCnt, persist, units, unitCtl, unitCap); break;
    case 32: hipLaunchKernelGGL((evaluate_bre_kernel<32, FULLVIS, PF, HS>), dim3((nwaves + SegCfg<32>::WPB - 1) / SegCfg<32>::WPB), dim3(64 * SegCfg<32>::WPB), dyn, stream, a, items, itemOff, itemCount, queueHead, pairs, pairCnt, persist, units, unitCtl, unitCap); break;
    default: hipLaunchKernelGGL((evaluate_bre_kernel<16, FULLVIS, PF, HS>), dim3((nwaves + SegCfg<16>::WPB - 1) / SegCfg<16>::WPB), dim3(64 * SegCfg<16>::WPB), dyn, stream, a, items, itemOff, itemCount, queueHead, pairs, pairCnt, persist, units, unitCtl, unitCap); break;
  }
}

// fullVis: shadow rays walk the occluder BVH (intended visibility, > 254 occluders, near-list overflow)
void launch_evaluate_bre(const GatherArgs &a, int beamsPerWave, bool fullVis, const uint4 *items, const uint2 *itemOff,
                         const uint32_t *itemCount, uint32_t *queueHead, const uint32_t *pairs, const uint32_t *pairCnt,
                         uint32_t nwaves, bool persistent, hipStream_t stream, const uint2 *units, const uint32_t *unitCtl,
                         uint32_t unitCap) {
  if (!persistent) units = nullptr;  // (one item per wave: the grid is the item count)
  if (a.nsets == 0 || nwaves == 0) return;
  // the record prefetch of phase 1 pays when the records (128 bytes each) no longer fit the 256 MB Infinity Cache: measured
  // +5 % on a rank's step at C4 (4 M photons), -1 % at C2 (1 M).  GVPM_RECORD_PREFETCH=0/1 (cfg.reserved[0] bits 5, 6) forces it.
  const bool pf = (a.cfg.reserved[0] & 32) ? false : ((a.cfg.reserved[0] & 64) ? true : (size_t)a.nph * 128u > ((size_t)256 << 20));
#define GVPM_LAUNCH_EVAL(FV, P) \
  launchEvaluate<FV, P>(a, beamsPerWave, items, itemOff, itemCount, queueHead, pairs, pairCnt, nwaves, persistent, stream, units, unitCtl, unitCap)
  if (a.reqHost) {
    // manifold-typed shifts go to the host's request list (no record prefetch variant of these)
    if (fullVis) launchEvaluate<true, false, true>(a, beamsPerWave, items, itemOff, itemCount, queueHead, pairs, pairCnt, nwaves, persistent, stream, units, unitCtl, unitCap);
    else launchEvaluate<false, false, true>(a, beamsPerWave, items, itemOff, itemCount, queueHead, pairs, pairCnt, nwaves, persistent, stream, units, unitCtl, unitCap);
  } else if (fullVis) {
    if (pf) GVPM_LAUNCH_EVAL(true, true);
    else GVPM_LAUNCH_EVAL(true, false);
  } else {
    if (pf) GVPM_LAUNCH_EVAL(false, true);
    else GVPM_LAUNCH_EVAL(false, false);
  }
#undef GVPM_LAUNCH_EVAL
}

uint32_t plan_items_capacity(uint32_t nsets, uint32_t ntiles, int beamsPerWave) {
  return (ntiles + nsets / (uint32_t)beamsPerWave + 1u) * (uint32_t)PLAN_MAX_ITEMS;
}

}  // namespace gvpm
