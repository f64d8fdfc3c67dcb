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
//   * traverse_bre_kernel is persistent (waves pull items from an atomic queue) and holds no
//     evaluation state, so it runs at high occupancy: for each slab step the 16-byte hot photon
//     records of the ranges are copied coalesced into an LDS stage and every lane tests them
//     against its own beam (LDS broadcast reads): fp32 test with a rigorous error band, the
//     reference predicate in uncontracted fp64 only when the band could change the decision, so
//     the hit set equals the fp64 oracle's bit for bit.  Hits are compacted with __ballot /
//     popcount and appended as (photon, beam) pairs to the item's region of the pair buffer
//     (the region is sized by the planner's upper bound; ~8 bytes written per hit);
//   * evaluate_bre_kernel, also persistent, takes an item's pairs 64 at a time: every lane
//     evaluates one (photon record = one 128-byte line): base contribution + 4 shifts (null
//     shift, or offset-path reconnection with shadow ray, Jacobian and MIS weight), adds 27
//     partial sums to the beam's LDS accumulators and flushes them with global atomics at the
//     end of the item.  Splitting the two phases gives each its own register budget (the fused
//     kernel sat at 256 VGPRs with spills and 2 waves/SIMD).
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
//     reconnection is only QUEUED (per lane, in LDS) as (photon, beam, shift, t', pdf);
//   phase 2 (one lane per queued shift): the diffuse reconnection with its shadow ray, Jacobian
//     and MIS weight; it runs whenever most lanes have an entry pending.
// At C2 ~70 % of the shifts are null shifts: running both branches on every lane of a mixed
// wave cost ~1.5x the VALU work of this arrangement.
//
// Accumulation: LDS float atomics (ds_add_f32) retire about one lane per clock on CDNA4 -- 24 of
// them per evaluation were half of this kernel's time.  The traversal therefore writes one photon
// list PER BEAM, the evaluation wave cuts the concatenated lists of an item into 64 equal chunks
// (perfect balance) and every lane sums its 27 outputs in REGISTERS; a lane touches the LDS
// accumulators only when its chunk crosses into the next beam and at the end of the item.
//
// Numerics: every quantity that the reference obtains by subtracting O(1) positions to get an
// O(radius) vector (photon - ray point, shifted ray point - base ray point) is formed in fp64 and
// then carried as a small fp32 vector; everything downstream of those differences (kernel chord
// lengths sqrt(r^2 - d^2), pdfs, BSDF / phase / transmittance products, MIS weights) is fp32.
#ifndef GVPM_QD
#define GVPM_QD 8
#endif
constexpr int QD = GVPM_QD;  // per-lane reconnection queue depth (a step adds at most 4)
constexpr uint32_t EVAL_LDS_TRIS = 64;  // occluders staged in the evaluation kernel's LDS when the scene has no more

template <int B> struct EvalLds : RayTile<B> {
  double acc[27][B];  // double: ds_add_f64 runs ~25x the rate of ds_add_f32 on gfx950 (scripts/probes/lds_atomics_bench.hip)
  uint32_t boff[B + 1];      // prefix offsets of the item's per-beam lists
  // per-lane queues, [slot][lane]: photon and beam | shift << 8.  t' and pdfCam are recomputed by the reconnection
  // (a dozen fp64 operations) rather than queued: 12 bytes less per entry is 6 KB of LDS per wave, the difference
  // between 8 and 12 resident waves per CU
  uint32_t qPh[QD][64];
  uint32_t qMeta[QD][64];
  // per (shift, beam), derived once per item: the shifted ray RELATIVE to the base ray {o_s - o_b, sensorMIS} and
  // {d_s - d_b, -}.  shiftRay(t') - baseRay(t') = dO + dD t' then is a small fp32 vector (pixel spacing at depth t'),
  // accurate to ~1e-10: the per-evaluation fp64 evaluation of shiftRay(t') (3 cvt + 3 fma + 3 add fp64 per shift)
  // goes away, and so do the three divisions of sensorMIS
  float4 relO[4][B], relD[4][B];
};

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
template <int B, typename LDS>
__device__ __forceinline__ void evalPhase1(const GatherArgs &a, LDS &s, uint32_t pidx, uint32_t b, Acc27 &acc,
                                           uint32_t &nNull, uint32_t &nFail, uint32_t &qMask, double &tPrimeOut,
                                           float &pdfCamOut) {
  const PhotonFront ph = loadFront(a, pidx);
  const RayReg base = loadRay(s, 0, b);
  const uint32_t pix = s.pix[b];
  const float r2 = a.radius * a.radius;
  const BaseTerms bt = baseTerms<B>(a, s, ph.pos, ph.wi, ph.flux, base, b);
  const f3 bc = bt.bc;
  acc.v[0] += bc.x;
  acc.v[1] += bc.y;
  acc.v[2] += bc.z;
  tPrimeOut = bt.tPrime;
  pdfCamOut = bt.pdfCam;
  qMask = 0u;

  const f3 photonIn = mk3(a.med.sigmaS[0], a.med.sigmaS[1], a.med.sigmaS[2]) * ph.flux;
  const float tPf = (float)bt.tPrime;
  const uint32_t st = GVPM_PF_SHIFT_TYPE(ph.bits);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // branch-free: the null-shift arithmetic is cheap and some lane of the wave needs it anyway; a wave
    // spends more on exec-mask bookkeeping and taken branches than on the arithmetic they would skip
    const RayReg sh = loadRay(s, 1 + i, b);
    const float4 ro = s.relO[i][b], rd = s.relD[i][b];
    // photon relative to shiftRay(t') = (photon - baseRay(t')) - (shiftRay(t') - baseRay(t'))
    const f3 y = bt.rel - (mk3(ro.x, ro.y, ro.z) + mk3(rd.x, rd.y, rd.z) * tPf);
    // shiftNull, shift_volume_photon.cpp:119-158 with the kernel pdfs of :782-801
    const bool isNull = sh.valid && a.cfg.use_shift_null && dot(y, y) < r2 && tPf < sh.len;
    const f3 yp = y - sh.d * dot(y, sh.d);
    const float deltaS = fsqrt(fmaxf(0.f, r2 - dot(yp, yp)));
    const float pdfShiftPos = frcp(fmaxf(2.f * deltaS, 0.0001f));
    float wNull = 0.5f;
    if (a.cfg.use_mis)
      wNull = (pdfShiftPos == 0.f || bt.pdfCam == 0.f)
                  ? 1.f
                  : frcp(1.f + ro.w * pdfShiftPos * frcp(bt.pdfCam));
    const f3 nullFlux = photonIn * (bt.tr * phaseEval(a.med.g, ph.wi, -sh.d)) * sh.eye;
    // shiftPhoton dispatch, shift_volume_photon.cpp:49-117: reconnections go to phase 2
    const bool wantsShift = sh.valid && !isNull && sh.len >= tPf && a.cfg.debug_shift != GVPM_SHIFT_NULL;
    const bool queued = wantsShift && (st == 1u || st == 2u);
    nNull += isNull ? 1u : 0u;
    nFail += (wantsShift && !queued) ? 1u : 0u;
    qMask |= queued ? (1u << i) : 0u;
    float w = isNull ? wNull : 1.f;
    borderRule(a, pix, i, w);
    const float keep = queued ? 0.f : 1.f;         // a queued shift adds nothing here
    const float ws = isNull ? w * bt.scale : 0.f;  // only the null shift has a shifted flux in phase 1
    acc.v[3 + 3 * i + 0] += nullFlux.x * ws;
    acc.v[3 + 3 * i + 1] += nullFlux.y * ws;
    acc.v[3 + 3 * i + 2] += nullFlux.z * ws;
    acc.v[15 + 3 * i + 0] += bc.x * (w * keep);
    acc.v[15 + 3 * i + 1] += bc.y * (w * keep);
    acc.v[15 + 3 * i + 2] += bc.z * (w * keep);
  }
}

// phase 2: one queued reconnection shift (shiftPhotonDiffuse through getShiftPos); the result goes to
// the lane's registers when it belongs to the lane's current beam, else straight to the LDS accumulators
template <int B, bool FULLVIS, typename LDS>
__device__ __forceinline__ void evalPhase2Core(const GatherArgs &a, LDS &s, uint32_t pidx, uint32_t b, int i, f3 &sf,
                                               f3 &wb, uint32_t &nDiff, uint32_t &nFail, const float4 *ldsTri) {
  const PhotonCold ph = loadCold(a, pidx);
  const RayReg base = loadRay(s, 0, b);
  const RayReg sh = loadRay(s, 1 + i, b);
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
  const float4 ro = s.relO[i][b], rd = s.relD[i][b];
  const f3 dS = mk3(ro.x, ro.y, ro.z) + mk3(rd.x, rd.y, rd.z) * (float)tPrime;  // shiftRay(t') - baseRay(t')
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
    const float cosD2 = dot(bo, bo) < r2 ? -2.f * dot(dS, offRel) * frcp(dot(dS, dS)) : 0.f;
    offRel = offRel + dS * cosD2;
  }
  float pdfShiftPos = 1.f;
  if (use3D) {
    const f3 op = offRel - sh.d * dot(offRel, sh.d);
    const float deltaO = fsqrt(fmaxf(0.f, r2 - dot(op, op)));
    pdfShiftPos = frcp(fmaxf(2.f * deltaO, 0.0001f));
  }
  bool ok = false;
  f3 sflux;
  const f3 dProjU = ((basePt + dS) - ph.parentPos) + offRel;  // offsetPos - parent
  float w = shiftDiffuse<FULLVIS>(a, ph, ph.bits, dProjU, sh, base, s.edge[b], trT, pdfCam, pdfShiftPos, sflux, ok, ldsTri,
                                 ro.w);
  if (ok) nDiff++; else nFail++;
  borderRule(a, s.pix[b], i, w);
  const float ws = w * scale;
  sf = sflux * ws;
  wb = bc * w;
}

template <int B, bool FULLVIS>
__device__ __forceinline__ void evalPhase2(const GatherArgs &a, EvalLds<B> &s, uint32_t pidx, uint32_t meta,
                                           uint32_t curBeam, Acc27 &acc, uint32_t &nDiff, uint32_t &nFail,
                                           const float4 *ldsTri) {
  const uint32_t b = meta & 0xFFu;
  const int i = (int)(meta >> 8);
  f3 sf, wb;
  evalPhase2Core<B, FULLVIS>(a, s, pidx, b, i, sf, wb, nDiff, nFail, ldsTri);
  if (b == curBeam) {
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const float m = ii == i ? 1.f : 0.f;
      acc.v[3 + 3 * ii + 0] += m * sf.x;
      acc.v[3 + 3 * ii + 1] += m * sf.y;
      acc.v[3 + 3 * ii + 2] += m * sf.z;
      acc.v[15 + 3 * ii + 0] += m * wb.x;
      acc.v[15 + 3 * ii + 1] += m * wb.y;
      acc.v[15 + 3 * ii + 2] += m * wb.z;
    }
  } else {
    atomicAdd(&s.acc[3 + 3 * i + 0][b], (double)(sf.x));
    atomicAdd(&s.acc[3 + 3 * i + 1][b], (double)(sf.y));
    atomicAdd(&s.acc[3 + 3 * i + 2][b], (double)(sf.z));
    atomicAdd(&s.acc[15 + 3 * i + 0][b], (double)(wb.x));
    atomicAdd(&s.acc[15 + 3 * i + 1][b], (double)(wb.y));
    atomicAdd(&s.acc[15 + 3 * i + 2][b], (double)(wb.z));
  }
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
struct TravLds {
  float4 stage[STAGE];
  uint32_t stageIdx[STAGE];
};

template <int B>
__global__ __launch_bounds__(64) void traverse_bre_kernel(GatherArgs a, const uint4 *__restrict__ items,
                                                          const uint2 *__restrict__ itemOff,
                                                          const uint32_t *__restrict__ itemCount, uint32_t *queueHead,
                                                          uint32_t *__restrict__ pairs, uint32_t *__restrict__ pairCnt) {
  constexpr int LPB = 64 / B;
  __shared__ TravLds s;
  const int lane = threadIdx.x;
  const uint32_t nItems = *itemCount;
  const int b = lane % B, sub = lane / B;
  const float r = a.radius;
  const float r2f = r * r;
  const float eps = a.cfg.epsilon;
  const bool use3D = a.cfg.vol_technique == GVPM_VOL_BRE3D;
  const uint32_t coalesceAt = a.cfg.reserved[3] ? (uint32_t)a.cfg.reserved[3] : 512u;  // photons in a box row set
  unsigned long long nCand = 0, nOver = 0;

  // the first item of every wave is its own index; the shared counter (one address: ~11 ns per atomic whatever the
  // number of waves) serves the rest
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
    const uint32_t setBase = item.x, nb = item.y;
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
    const uint32_t edge = bi.edge;
    const uint32_t pixParity = ((bi.pix & 0xFFFFu) + (bi.pix >> 16)) & 1u;
    uint32_t staged = 0;  // wave-uniform

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
          __syncthreads();
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
                s.stage[k] = a.hot[gi];
                s.stageIdx[k] = gi;
              }
            }
          } else
          {
            const uint32_t lo_i = max(excl, win), hi_i = min(excl + count, win + STAGE);
            uint32_t i = lo_i;
            // four independent 16-byte loads in flight per lane before the LDS writes
            for (; i + 4 <= hi_i; i += 4) {
              const uint32_t gi = start + (i - excl);
              const float4 v0 = a.hot[gi], v1 = a.hot[gi + 1], v2 = a.hot[gi + 2], v3 = a.hot[gi + 3];
              s.stage[i - win] = v0; s.stage[i - win + 1] = v1; s.stage[i - win + 2] = v2; s.stage[i - win + 3] = v3;
              s.stageIdx[i - win] = gi; s.stageIdx[i - win + 1] = gi + 1;
              s.stageIdx[i - win + 2] = gi + 2; s.stageIdx[i - win + 3] = gi + 3;
            }
            for (; i < hi_i; ++i) {
              const uint32_t gi = start + (i - excl);
              s.stage[i - win] = a.hot[gi];
              s.stageIdx[i - win] = gi;
            }
          }
          __syncthreads();
          staged += nst;
          const uint32_t iters = (nst + LPB - 1) / LPB;
          // groups of G staged photons per lane: a branch-free coarse pass marks the candidates (the G LDS
          // reads overlap), then the wave resolves candidates one per lane and round (~1.5 rounds per
          // group instead of G passes through the divergent code)
          constexpr uint32_t G = 4;
          for (uint32_t jj = 0; jj < iters; jj += G) {
            uint32_t cm = 0;
            if (beamValid) {
#pragma unroll
              for (uint32_t u = 0; u < G; ++u) {
                const uint32_t j = (jj + u) * LPB + sub;
                const float4 hp = s.stage[min(j, (uint32_t)STAGE - 1u)];
                const f3 wv = mk3(hp.x, hp.y, hp.z) - base.o;
                const float disk = dot(wv, base.d);
                const f3 v = wv - base.d * disk;
                const float d2 = dot(v, v);
                const float E = 6e-7f * (fabsf(wv.x) + fabsf(wv.y) + fabsf(wv.z) + fabsf(disk));
                const float band = 4.f * r * E + r2f * 2e-6f;
                if (j < nst && d2 < r2f + band && disk > mint - E && disk < maxt + 2.f * r) cm |= 1u << u;
              }
            }
            while (__ballot(cm != 0u)) {
              // straight-line, predicated (a lane without a candidate computes on entry 0 and discards)
              const bool active = cm != 0u;
              const uint32_t j = min((jj + (active ? (uint32_t)__ffs(cm) - 1u : 0u)) * LPB + sub, (uint32_t)STAGE - 1u);
              cm &= cm - 1u;
              const float4 hp = s.stage[j];
              const f3 wv = mk3(hp.x, hp.y, hp.z) - base.o;
              const float disk = dot(wv, base.d);
              const f3 v = wv - base.d * disk;
              const float d2 = dot(v, v);
              // fp32 with a rigorous error band: E bounds |disk - disk_exact|, band |d2 - d2_exact|
              const float E = 6e-7f * (fabsf(wv.x) + fabsf(wv.y) + fabsf(wv.z) + fabsf(disk));
              const float band = 4.f * r * E + r2f * 2e-6f;
              const uint32_t bits = __float_as_uint(hp.w);
              // filters, shift_volume_photon.cpp:670-697
              const int depth = (int)GVPM_PF_DEPTH(bits) + (int)edge;
              const bool keep = active && !(a.cfg.max_depth > 0 && depth > a.cfg.max_depth) &&
                                !(a.cfg.min_depth != 0 && depth < a.cfg.min_depth) &&
                                ((bits >> 6) & 1u) &&  // computeVolumeContribution + debugShift (grid_build)
                                !(a.cfg.path_set && ((bits >> GVPM_HOT_PARITY_BIT) & 1u) != pixParity);
              // decided in fp32 when the error band cannot change the reference's decision ...
              const bool in0 = d2 < r2f - band && disk > mint + E && disk < maxt - E;
              bool in = in0, outside = false;
              if (use3D) {
                // t' = (disk - deltaT) + 2 deltaT rnd must lie in [mint, len] (shift_volume_photon.cpp:707-726):
                // bracket it with deltaT in [dTlo, dTup]
                const float q = r2f - d2;
                const float dTup = fsqrt(fmaxf(q + band, 0.f)) * 1.000001f, dTlo = fsqrt(fmaxf(q - band, 0.f)) * 0.999999f;
                const float slop = 2.f * E + 4e-7f * (fabsf(disk) + dTup);
                const float tLo = (disk - dTup) + 2.f * dTlo * bi.rnd - slop;
                const float tHi = (disk - dTlo) + 2.f * dTup * bi.rnd + slop;
                in = in0 && tLo > mint && tHi < base.len;
                outside = in0 && (tHi < mint || tLo > base.len);
              }
              // ... otherwise the pair is passed on flagged: the evaluation kernel runs the reference
              // predicate itself (fp64, uncontracted) on it -- about one candidate in 10^5
              const bool hit = keep && !outside;
              const bool amb = keep && !in && !outside;
              const unsigned long long m = __ballot(hit);
              if (m) {
                // the hits of my beam in this round sit in lanes b, b + B, ...: append in lane order
                constexpr unsigned long long GROUP = B == 16 ? 0x0001000100010001ull : (B == 32 ? 0x0000000100000001ull : 1ull);
                const unsigned long long g = (m >> b) & GROUP;
                if (hit) {
                  const uint32_t off = mine + __popcll(g & ((1ull << (sub * B)) - 1ull));
                  if (off < cap) out[off] = s.stageIdx[j] | (amb ? 0x80000000u : 0u);
                  else nOver++;
                }
                mine += __popcll(g);
              }
            }
          }
        }
      }
    }
    if (sub == 0) pairCnt[(size_t)it * B + b] = (uint32_t)b < nb ? min(mine, cap) : 0u;
    nCand += (unsigned long long)staged * nb;
  }
  {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nOver += __shfl_xor(nOver, o, 64);
    if (lane == 0 && (nCand | nOver)) {
      unsigned long long *row = a.stats + 8 * (size_t)blockIdx.x;  // this wave's own row
      row[1] += nCand;
      if (nOver) row[7] += nOver;  // must stay 0: the planner's bound is exact
    }
  }
}

// ------------------------------------------------------------------------------------------
// evaluation: persistent waves, one item's pairs at a time, 64 evaluations per step
// ------------------------------------------------------------------------------------------
template <int B, bool FULLVIS>
__global__ __launch_bounds__(64) void evaluate_bre_kernel(GatherArgs a, const uint4 *__restrict__ items,
                                                          const uint2 *__restrict__ itemOff,
                                                          const uint32_t *__restrict__ itemCount, uint32_t *queueHead,
                                                          const uint32_t *__restrict__ pairs,
                                                          const uint32_t *__restrict__ pairCnt) {
  __shared__ EvalLds<B> s;
  // the occluders of a small scene live in LDS: the near-occluder loop of the reconnection then reads LDS instead
  // of (L1/L2-resident) global memory, whose latency two waves per SIMD cannot hide
  // (dynamic shared memory, sized by the launcher: 48 bytes per occluder, nothing for larger scenes)
  extern __shared__ float4 sceneTri[];
  const int lane = threadIdx.x;
  const float4 *ldsTri = nullptr;
  if (!FULLVIS && a.ntri <= EVAL_LDS_TRIS && !(a.cfg.reserved[0] & 16)) {
    for (uint32_t i = lane; i < 3u * a.ntri; i += 64u) sceneTri[i] = a.tri4[i];
    ldsTri = sceneTri;
    __syncthreads();
  }
  const uint32_t nItems = *itemCount;
  const bool skip = (a.cfg.reserved[0] & 1) != 0;  // development switch: count, do not evaluate
  uint32_t nEval = 0, nNull = 0, nDiff = 0, nFail = 0;

  // the first item of every wave is its own index; the shared counter (one address: ~11 ns per atomic whatever the
  // number of waves) serves the rest
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
    const uint32_t setBase = item.x, nb = item.y;
    if (nb == 0) continue;
    // prefix offsets of the per-beam lists
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
    loadTileRays<B>(a, s, setBase, nb, lane);
    for (int idx = lane; idx < 27 * B; idx += 64) (&s.acc[0][0])[idx] = 0.0;
    for (int idx = lane; idx < 4 * B; idx += 64) {
      const int i = idx / B, bb = idx % B;
      const RayReg br = loadRay(s, 0, bb), sr = loadRay(s, 1 + i, bb);
      const f3 dO = tof(tod(sr.o) - tod(br.o)), dD = tof(tod(sr.d) - tod(br.d));
      s.relO[i][bb] = make_float4(dO.x, dO.y, dO.z, sensorMIS(sr, br, s.edge[bb]));
      s.relD[i][bb] = make_float4(dD.x, dD.y, dD.z, 0.f);
    }
    __syncthreads();

    // my chunk [g0, g1) of the concatenated lists
    const uint32_t chunk = (total + 63u) / 64u;
    const uint32_t g0 = min(total, (uint32_t)lane * chunk), g1 = min(total, g0 + chunk);
    uint32_t cur = 0;  // current beam
    if (g0 < g1)
      while (s.boff[cur + 1] <= g0) cur++;
    Acc27 acc;
#pragma unroll
    for (int k = 0; k < 27; ++k) acc.v[k] = 0.f;
    uint32_t qHead = 0, qCount = 0;  // this lane's reconnection queue

    for (uint32_t t = 0; t <= chunk; ++t) {
      const bool last = t == chunk;
      if (!last) {
        const uint32_t g = g0 + t;
        uint32_t qMask = 0;
        double tP = 0.0;
        float pdfCam = 0.f;
        uint32_t pidx = 0;
        if (g < g1) {
          if (g >= s.boff[cur + 1]) {
            flushAcc<B>(s, acc, cur);
            do cur++; while (s.boff[cur + 1] <= g);
          }
          pidx = lists[(size_t)cur * cap + (g - s.boff[cur])];
          bool ok = true;
          if (pidx & 0x80000000u) {
            // the traversal could not decide this pair in fp32: the reference predicate, fp64, uncontracted
            pidx &= 0x7FFFFFFFu;
            const float4 c0 = a.cold[(size_t)pidx * GVPM_REC_QUADS];
            const RayReg br = loadRay(s, 0, cur);
            ok = exactHit(mk3(c0.x, c0.y, c0.z), br.o, br.d, br.len, a.radius, s.rnd[cur], a.cfg.epsilon,
                          a.cfg.vol_technique == GVPM_VOL_BRE3D);
          }
          if (ok) {
            if (!skip) evalPhase1<B>(a, s, pidx, cur, acc, nNull, nFail, qMask, tP, pdfCam);
            nEval++;
          }
        }
        for (uint32_t m = qMask; m; m &= m - 1u) {
          const uint32_t sh = (uint32_t)__ffs(m) - 1u;
          const uint32_t q = (qHead + qCount) % QD;
          s.qPh[q][lane] = pidx;
          s.qMeta[q][lane] = cur | (sh << 8);
          qCount++;
        }
      }
      // reconnections: run while most lanes have one pending, or a queue could overflow next step;
      // after the last step drain everything
      for (;;) {
        const unsigned long long pending = __ballot(qCount > 0u);
        if (!pending) break;
        if (!last && __popcll(pending) < 48 && !__ballot(qCount > (uint32_t)(QD - 4))) break;
        if (qCount > 0u) {
          const uint32_t q = qHead;
          evalPhase2<B, FULLVIS>(a, s, s.qPh[q][lane], s.qMeta[q][lane], cur, acc, nDiff, nFail, ldsTri);
          qHead = (qHead + 1u) % QD;
          qCount--;
        }
      }
    }
    if (g0 < g1) flushAcc<B>(s, acc, cur);
    __syncthreads();
    // ---- write out: 27 partial sums per beam set into the iteration buffer ----
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
    if (lane == 0 && ev) {
      unsigned long long *row = a.stats + 8 * (size_t)blockIdx.x;  // this wave's own row
      row[0] += ev;
      row[2] += nu;
      row[3] += di;
      row[4] += fa;
    }
  }
}

// ------------------------------------------------------------------------------------------
// evaluation, segmented: the same work with the two phases in SEPARATE loops.
//
// The kernel above interleaves the reconnections with the phase-1 walk, so the 27 register sums of phase 1 stay
// live across phase 2 and the register allocation is the union of both (187 VGPRs, two waves per SIMD); each
// wave also holds per-lane reconnection queues of 4 KB.  Here a wave walks a SEGMENT of its item's pairs through
// phase 1 only (<= SEG_STEPS steps, or until the queue could overflow), appending the reconnections it meets to ONE
// compact queue (ballot + popcount, 2 bytes per entry: step, lane, shift, beam), folds its register sums into the
// LDS accumulators, and then runs the queue through phase 2 in a dense loop of its own: every lane takes an equal,
// contiguous share of the queue, whose entries come in runs of one (beam, shift), so a lane keeps 6 sums in
// registers and touches the LDS accumulators once per run.  The allocation is the larger of the two loops, not
// their union, the queue is a quarter of the size, and the waves of a workgroup share the staged occluders:
// three waves per SIMD instead of two.
// ------------------------------------------------------------------------------------------
#ifndef GVPM_EVAL_MINW
#define GVPM_EVAL_MINW 3
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
  float4 relO[4][B], relD[4][B];          // as in EvalLds
};

// LDS accesses of ONE wave are executed in order; what has to be stopped is the compiler moving them
__device__ __forceinline__ void waveLdsSync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int B, bool FULLVIS>
__global__ __launch_bounds__(64 * SegCfg<B>::WPB, GVPM_EVAL_MINW * SegCfg<B>::WPB / 4 > 0 ? GVPM_EVAL_MINW * SegCfg<B>::WPB / 4 : 1)
void evaluate_bre_seg_kernel(GatherArgs a, const uint4 *__restrict__ items, const uint2 *__restrict__ itemOff,
                             const uint32_t *__restrict__ itemCount, uint32_t *queueHead,
                             const uint32_t *__restrict__ pairs, const uint32_t *__restrict__ pairCnt) {
  constexpr int WPB = SegCfg<B>::WPB;
  constexpr int BB = SegCfg<B>::BEAM_BITS;
  using Entry = typename SegCfg<B>::Entry;
  __shared__ SegLds<B> sAll[WPB];
  extern __shared__ float4 sceneTri[];  // the occluders of a small scene (48 bytes each), shared by the waves
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
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

  bool firstItem = true;
  for (;;) {
    uint32_t it = waveId;
    if (!firstItem) {
      if (lane == 0) it = nWaves + atomicAdd(queueHead, 1u);
      it = __shfl(it, 0, 64);
    }
    firstItem = false;
    if (it >= nItems) break;
    const uint4 item = items[it];
    const uint32_t setBase = item.x, nb = item.y;
    if (nb == 0) continue;
    const uint32_t cntb = (uint32_t)lane < nb ? pairCnt[(size_t)it * B + lane] : 0u;
    const uint32_t incl = wave_scan_incl(cntb, lane);
    const uint32_t total = __shfl(incl, 63, 64);
    if (total == 0) continue;
    const uint2 reg = itemOff[it];
    const uint32_t *lists = pairs + (size_t)reg.x * 64u;
    const uint32_t cap = reg.y;
    waveLdsSync();
    if (lane < B) s.boff[lane + 1] = incl;
    if (lane == 0) s.boff[0] = 0u;
    loadTileRaysNoSync<B>(a, s, setBase, nb, lane);
    for (int idx = lane; idx < 27 * B; idx += 64) (&s.acc[0][0])[idx] = 0.0;
    waveLdsSync();
    for (int idx = lane; idx < 4 * B; idx += 64) {
      const int i = idx / B, bb = idx % B;
      const RayReg br = loadRay(s, 0, bb), sr = loadRay(s, 1 + i, bb);
      const f3 dO = tof(tod(sr.o) - tod(br.o)), dD = tof(tod(sr.d) - tod(br.d));
      s.relO[i][bb] = make_float4(dO.x, dO.y, dO.z, sensorMIS(sr, br, s.edge[bb]));
      s.relD[i][bb] = make_float4(dD.x, dD.y, dD.z, 0.f);
    }
    waveLdsSync();

    // my chunk [g0, g1) of the concatenated per-beam lists (lane l of ANY wave position: g0 = min(total, l * chunk))
    const uint32_t chunk = (total + 63u) / 64u;
    const uint32_t g0 = min(total, (uint32_t)lane * chunk), g1 = min(total, g0 + chunk);
    uint32_t cur = 0;  // current beam
    if (g0 < g1)
      while (s.boff[cur + 1] <= g0) cur++;

    for (uint32_t tSeg = 0; tSeg < chunk;) {
      // ---- phase 1 over a segment of steps ----
      uint32_t qn = 0;  // wave-uniform
      Acc27 acc;
#pragma unroll
      for (int k = 0; k < 27; ++k) acc.v[k] = 0.f;
      uint32_t t = tSeg;
      for (; t < chunk && t - tSeg < (uint32_t)SEG_STEPS && qn + 256u <= (uint32_t)SEG_QCAP; ++t) {
        const uint32_t g = g0 + t;
        uint32_t qMask = 0;
        if (g < g1) {
          if (g >= s.boff[cur + 1]) {
            flushAcc<B>(s, acc, cur);
            do cur++; while (s.boff[cur + 1] <= g);
          }
          uint32_t pidx = lists[(size_t)cur * cap + (g - s.boff[cur])];
          bool ok = true;
          if (pidx & 0x80000000u) {
            // the traversal could not decide this pair in fp32: the reference predicate, fp64, uncontracted
            pidx &= 0x7FFFFFFFu;
            const float4 c0 = a.cold[(size_t)pidx * GVPM_REC_QUADS];
            const RayReg br = loadRay(s, 0, cur);
            ok = exactHit(mk3(c0.x, c0.y, c0.z), br.o, br.d, br.len, a.radius, s.rnd[cur], a.cfg.epsilon,
                          a.cfg.vol_technique == GVPM_VOL_BRE3D);
          }
          if (ok) {
            double tP;
            float pdfCam;
            evalPhase1<B>(a, s, pidx, cur, acc, nNull, nFail, qMask, tP, pdfCam);
            nEval++;
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
      }
      if (g0 < g1) flushAcc<B>(s, acc, cur);
      waveLdsSync();
      // ---- phase 2 over the segment's queue: lane l takes entries [l * cq, (l + 1) * cq) ----
      const uint32_t cq = (qn + 63u) / 64u;
      const uint32_t e0 = min(qn, (uint32_t)lane * cq), e1 = min(qn, e0 + cq);
      uint32_t key = 0xFFFFFFFFu;
      f3 rs = mk3(0.f), rw = mk3(0.f);
      for (uint32_t j = 0; j < cq; ++j) {
        if (e0 + j < e1) {
          const uint32_t e = s.q[e0 + j];
          const uint32_t b = e & ((1u << BB) - 1u), i = (e >> BB) & 3u, ln = (e >> (2 + BB)) & 63u, ts = e >> (8 + BB);
          const uint32_t g = min(total, ln * chunk) + tSeg + ts;
          const uint32_t pidx = lists[(size_t)b * cap + (g - s.boff[b])] & 0x7FFFFFFFu;
          const uint32_t k2 = (b << 2) | i;
          if (k2 != key) {
            if (key != 0xFFFFFFFFu) {
              const uint32_t kb = key >> 2, ki = key & 3u;
              atomicAdd(&s.acc[3 + 3 * ki + 0][kb], (double)rs.x);
              atomicAdd(&s.acc[3 + 3 * ki + 1][kb], (double)rs.y);
              atomicAdd(&s.acc[3 + 3 * ki + 2][kb], (double)rs.z);
              atomicAdd(&s.acc[15 + 3 * ki + 0][kb], (double)rw.x);
              atomicAdd(&s.acc[15 + 3 * ki + 1][kb], (double)rw.y);
              atomicAdd(&s.acc[15 + 3 * ki + 2][kb], (double)rw.z);
            }
            key = k2;
            rs = rw = mk3(0.f);
          }
          f3 sf, wb;
          evalPhase2Core<B, FULLVIS>(a, s, pidx, b, (int)i, sf, wb, nDiff, nFail, ldsTri);
          rs = rs + sf;
          rw = rw + wb;
        }
      }
      if (key != 0xFFFFFFFFu) {
        const uint32_t kb = key >> 2, ki = key & 3u;
        atomicAdd(&s.acc[3 + 3 * ki + 0][kb], (double)rs.x);
        atomicAdd(&s.acc[3 + 3 * ki + 1][kb], (double)rs.y);
        atomicAdd(&s.acc[3 + 3 * ki + 2][kb], (double)rs.z);
        atomicAdd(&s.acc[15 + 3 * ki + 0][kb], (double)rw.x);
        atomicAdd(&s.acc[15 + 3 * ki + 1][kb], (double)rw.y);
        atomicAdd(&s.acc[15 + 3 * ki + 2][kb], (double)rw.z);
      }
      waveLdsSync();
      tSeg = t;
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
    if (lane == 0 && ev) {
      unsigned long long *row = a.stats + 8 * (size_t)(waveId % GVPM_STAT_ROWS);  // this wave's own row
      row[0] += ev;
      row[2] += nu;
      row[3] += di;
      row[4] += fa;
    }
  }
}

// itemCount / blockTotal must be zero on entry (memset on the same stream)
void launch_plan_bre(const GatherArgs &a, int beamsPerWave, uint32_t ntiles, uint32_t target, uint4 *items,
                     uint32_t *itemCount, uint2 *itemOff, uint32_t *blockTotal, hipStream_t stream) {
  if (a.nsets == 0 || ntiles == 0) return;
  // waves stride over the tiles; the stride is prime because image-sharded input owns every N-th tile, and a
  // stride that is a multiple of N would leave all the work to 1/N of the blocks
  const uint32_t nwg = ntiles < 4093u ? ntiles : 4093u;
  switch (beamsPerWave) {
    case 64: hipLaunchKernelGGL(plan_kernel<64>, dim3(nwg), dim3(64), 0, stream, a, ntiles, target, items, itemCount, itemOff, blockTotal); break;
    case 32: hipLaunchKernelGGL(plan_kernel<32>, dim3(nwg), dim3(64), 0, stream, a, ntiles, target, items, itemCount, itemOff, blockTotal); break;
    default: hipLaunchKernelGGL(plan_kernel<16>, dim3(nwg), dim3(64), 0, stream, a, ntiles, target, items, itemCount, itemOff, blockTotal); break;
  }
}

// queueHead must be zero on entry
void launch_traverse_bre(const GatherArgs &a, int beamsPerWave, const uint4 *items, const uint2 *itemOff,
                         const uint32_t *itemCount, uint32_t *queueHead, uint32_t *pairs, uint32_t *pairCnt,
                         uint32_t nwaves, hipStream_t stream) {
  if (a.nsets == 0) return;
  switch (beamsPerWave) {
    case 64: hipLaunchKernelGGL(traverse_bre_kernel<64>, dim3(nwaves), dim3(64), 0, stream, a, items, itemOff, itemCount, queueHead, pairs, pairCnt); break;
    case 32: hipLaunchKernelGGL(traverse_bre_kernel<32>, dim3(nwaves), dim3(64), 0, stream, a, items, itemOff, itemCount, queueHead, pairs, pairCnt); break;
    default: hipLaunchKernelGGL(traverse_bre_kernel<16>, dim3(nwaves), dim3(64), 0, stream, a, items, itemOff, itemCount, queueHead, pairs, pairCnt); break;
  }
}

template <bool FULLVIS>
static void launchEvaluate(const GatherArgs &a, int beamsPerWave, const uint4 *items, const uint2 *itemOff,
                           const uint32_t *itemCount, uint32_t *queueHead, const uint32_t *pairs, const uint32_t *pairCnt,
                           uint32_t nwaves, hipStream_t stream) {
  const size_t dyn = (!FULLVIS && a.ntri <= EVAL_LDS_TRIS && !(a.cfg.reserved[0] & 16)) ? (size_t)a.ntri * 48u : 0u;
  static const bool seg = !(getenv("GVPM_EVAL_SEG") && atoi(getenv("GVPM_EVAL_SEG")) == 0);
  if (seg) {
    switch (beamsPerWave) {
      case 64: hipLaunchKernelGGL((evaluate_bre_seg_kernel<64, FULLVIS>), dim3(nwaves / SegCfg<64>::WPB), dim3(64 * SegCfg<64>::WPB), dyn, stream, a, items, itemOff, itemCount, queueHead, pairs, pairCnt); break;
      case 32: hipLaunchKernelGGL((evaluate_bre_seg_kernel<32, FULLVIS>), dim3(nwaves / SegCfg<32>::WPB), dim3(64 * SegCfg<32>::WPB), dyn, stream, a, items, itemOff, itemCount, queueHead, pairs, pairCnt); break;
      default: hipLaunchKernelGGL((evaluate_bre_seg_kernel<16, FULLVIS>), dim3(nwaves / SegCfg<16>::WPB), dim3(64 * SegCfg<16>::WPB), dyn, stream, a, items, itemOff, itemCount, queueHead, pairs, pairCnt); break;
    }
    return;
  }
  switch (beamsPerWave) {
    case 64: hipLaunchKernelGGL((evaluate_bre_kernel<64, FULLVIS>), dim3(nwaves), dim3(64), dyn, stream, a, items, itemOff, itemCount, queueHead, pairs, pairCnt); break;
    case 32: hipLaunchKernelGGL((evaluate_bre_kernel<32, FULLVIS>), dim3(nwaves), dim3(64), dyn, stream, a, items, itemOff, itemCount, queueHead, pairs, pairCnt); break;
    default: hipLaunchKernelGGL((evaluate_bre_kernel<16, FULLVIS>), dim3(nwaves), dim3(64), dyn, stream, a, items, itemOff, itemCount, queueHead, pairs, pairCnt); break;
  }
}

// fullVis: shadow rays walk the occluder BVH (intended visibility, > 254 occluders, near-list overflow)
void launch_evaluate_bre(const GatherArgs &a, int beamsPerWave, bool fullVis, const uint4 *items, const uint2 *itemOff,
                         const uint32_t *itemCount, uint32_t *queueHead, const uint32_t *pairs, const uint32_t *pairCnt,
                         uint32_t nwaves, hipStream_t stream) {
  if (a.nsets == 0) return;
  if (fullVis) launchEvaluate<true>(a, beamsPerWave, items, itemOff, itemCount, queueHead, pairs, pairCnt, nwaves, stream);
  else launchEvaluate<false>(a, beamsPerWave, items, itemOff, itemCount, queueHead, pairs, pairCnt, nwaves, stream);
}

uint32_t plan_items_capacity(uint32_t nsets, uint32_t ntiles, int beamsPerWave) {
  return (ntiles + nsets / (uint32_t)beamsPerWave + 1u) * (uint32_t)PLAN_MAX_ITEMS;
}

}  // namespace gvpm
