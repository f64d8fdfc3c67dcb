// G-VPM (3D point kernel) gather + gradient-domain shift for gfx950, hand-written HIP.
//
// Replaces, for one SPPM iteration, the body of
//   GPMIntegrator::computeVolumeGradientPhoton        gvpm/gvpm.cpp:1081-1203
//   PointKDTree::executeQuery (radius query)          include/mitsuba/core/kdtree.h:675-731
//   VolumeGradientPositionQuery::operator()           gvpm/shift/shift_volume_photon.cpp:489-655
//   VolumeGradientDistanceQuery pdfs                  gvpm/shift/shift_volume_photon.h:162-197
//   HomogeneousMedium::sampleDistance(EDistanceAlwaysValid) / eval(EDistanceAlwaysValid)
//                                                     src/medium/homogeneous.cpp:293-430,432-513
// and shares shiftNull / shiftPhotonDiffuse / diffuseReconnection / getShiftPos / sensorMIS with the
// BRE kernel (shift_device.h).
//
// One wave handles 64 consecutive camera samples (the host emits the nbCameraSamples samples of a
// pixel consecutively, so a wave covers 1-2 pixels and its five-ray beam sets stay L1 hot).
// The wave walks the <= 3x3x3 grid cells of its 64 query spheres together (the lanes' photon ranges
// laid end to end, one candidate per lane and trip); hits are compacted (ballot + popcount) into an
// LDS ring and evaluated 64 at a time, exactly like the BRE kernel.  The radius is per pixel
// (gp.scaleVol), the grid cell is the largest radius of the iteration.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_types.h"
#include "shift_device.h"
#include "vec.h"

namespace gvpm {

constexpr int VQ = 128;  // hit ring capacity
constexpr int VRQ = 128; // reconnection ring: 63 left over + the 64 one shift of a batch can add
// A/B probes (scripts/vpm_probe.py): GVPM_VPM_PROBE 1 = no evaluation, 2 = plain LDS adds instead of atomics
#ifndef GVPM_VPM_PROBE
#define GVPM_VPM_PROBE 0
#endif
// The 27 sums of a PIXEL RUN (the consecutive samples of one pixel: 40 at C1, so a wave holds two or three runs), not of
// a sample: 27 x 64 doubles were 13.8 KB of the kernel's 21 KB of LDS and held it at 7 waves per CU.  A wave with more
// than VPM_RUNS runs (a host that does not group a pixel's samples) adds the surplus straight to the film.
#ifndef GVPM_VPM_RUNS
#define GVPM_VPM_RUNS 4
#endif
#ifndef GVPM_VPM_SUB
#define GVPM_VPM_SUB 4  // (round 4: 8 -> 4 copies, 13 KB of LDS a wave: 12 waves per CU at 168 VGPRs instead of 9; C1 0.587 -> 0.562 ms)
#endif
// waves per SIMD the register allocation aims at (3: 168 VGPRs)
#ifndef GVPM_VPM_MINW
#define GVPM_VPM_MINW 3
#endif
constexpr int VPM_RUNS = GVPM_VPM_RUNS;
constexpr int VPM_ROWS = 9;  // rows of a sample's cell box walked per pass (a box is 3 x 3 x 3 cells but for rounding)
// ... and VPM_SUB copies of a run's sums, picked by the adding lane: all 64 lanes on the two or three addresses of one
// copy serialise in the LDS atomic unit (measured at 4 x the C1 radius, 16 M evaluations: 3.2 -> 3.9 ms with one copy)
constexpr int VPM_SUB = GVPM_VPM_SUB;

// The rays are NOT staged in LDS: the 64 camera samples of a workgroup belong to one or two pixels (40 samples per
// pixel at C1), i.e. to one or two beam sets, so the 5 x 64 bytes of a set are L1-resident broadcast reads -- and 17 KB
// of LDS a wave held the kernel at 6 waves per CU.
struct VpmLds {
  uint32_t set[64];    // beam set of the sample (0xFFFFFFFF: none)
  double acc[27][VPM_RUNS * VPM_SUB];  // double: ds_add_f64 runs ~25x the rate of ds_add_f32 on gfx950 (scripts/probes/lds_atomics_bench.hip)
  uint32_t run[64];    // pixel run of the sample
  uint32_t runPix[VPM_RUNS];  // pixel index of the run
  uint2 queue[VQ];
  double t[64];        // sampled camera distance (mRec.t)
  float pdfBase[64];   // pdfBaseRay() = mRec.pdfSuccess * pdfSel
  float pdfSel[64];
  float trBase[64];    // exp(-sigma_t (t - mint))
  float radius[64];
  uint32_t pix[64];
  uint32_t edge[64];
  float4 qr[64];       // query point and radius of the sample
  // the candidate lists of the 64 samples laid end to end: a sample's first entry, and per row of its cell box (VPM_ROWS
  // rows a pass: the 3 x 3 rows of a box are one pass) the row's first photon and its first entry within the sample's list
  uint32_t segOff[64 + 1];
  uint32_t rowStart[VPM_ROWS][64], rowOff[VPM_ROWS][64];
  uint32_t found[64];  // photons inside the query sphere (M of the SPPM update)
  uint2 rq[VRQ];       // queued reconnections {photon, sample | shift << 8}
  static constexpr uint32_t RUN_LIMIT = (uint32_t)VPM_RUNS;  // runs below it have LDS sums (vpmAdd)
  static constexpr bool RAYS_AHEAD = false;
  static constexpr uint32_t ROW_LIMIT = 0xFFFFFFFFu;  // candidates one sample may list in a pass
  __device__ __forceinline__ void initSample(int lane, uint32_t e) { edge[lane] = e; found[lane] = 0u; }
  __device__ __forceinline__ uint32_t edgeOf(uint32_t b) const { return edge[b]; }
  __device__ __forceinline__ uint32_t foundOf(uint32_t b) const { return found[b]; }
  __device__ __forceinline__ void setRowOff(int k, int lane, uint32_t v) { rowOff[k][lane] = v; }
  __device__ __forceinline__ uint32_t getRowOff(uint32_t k, uint32_t b) const { return rowOff[k][b]; }
  __device__ __forceinline__ uint32_t accSlot(uint32_t r) const { return r * VPM_SUB + (threadIdx.x & (VPM_SUB - 1)); }  // (VPM_SUB divides 64)
  __device__ __forceinline__ uint32_t sampleIndex(uint32_t sBase, uint32_t b) const { return sBase + b; }
};

template <typename LDS>
__device__ __forceinline__ RayReg loadRayV(const GatherArgs &a, const LDS &s, int k, int b) {
  RayReg r;
  const uint32_t set = s.set[b];
  if (set == 0xFFFFFFFFu) {
    r.o = r.eye = mk3(0.f);
    r.d = mk3(0.f, 0.f, 1.f);
    r.len = 1e-30f;
    r.pdf = r.jac = r.gop = 0.f;
    r.valid = false;
    return r;
  }
  const float4 *rp = reinterpret_cast<const float4 *>(a.rays + (size_t)set * 5 + k);
  const float4 q0 = rp[0], q1 = rp[1], q2 = rp[2], q3 = rp[3];
  r.o = mk3(q0.x, q0.y, q0.z);
  r.len = fabsf(q0.w);
  r.valid = GVPM_RAY_VALID(__float_as_uint(q3.y)) != 0;
  r.d = mk3(q1.x, q1.y, q1.z);
  r.pdf = q1.w;
  r.eye = mk3(q2.x, q2.y, q2.z);
  r.jac = q2.w;
  r.gop = q3.x;
  return r;
}

// 1 - exp(-x), x >= 0, without the cancellation of the difference in a thin medium (sigma_t d ~ 1e-3 loses four digits, below
// 6e-8 all of them: ADVICE round 4): the series below 1/16, the difference above (relative error < 2e-6 there).  (expm1f is
// exact to an ulp but costs the C1 kernel 9 %: it sits in every null shift.)
__device__ __forceinline__ float oneMinusExpNeg(float x) {
  const float ser = x * (1.f - x * (0.5f - x * (0.16666667f - x * (0.041666668f - x * 0.0083333333f))));
  return x < 0.0625f ? ser : 1.f - __expf(-x);
}

// The accumulator column of sample b (its pixel run's, one of the run's copies by the adding lane), or none: decided ONCE per
// pair -- decided per value, every one of a pair's 27 adds carried a read of s.run, a compare, a branch and its address (a third
// of phase 1's instructions, round 6)
constexpr uint32_t VPM_NO_COL = 0xFFFFFFFFu;
template <typename LDS>
__device__ __forceinline__ uint32_t vpmColumn(const LDS &s, uint32_t b) {
  const uint32_t r = s.run[b];
  return r < LDS::RUN_LIMIT ? s.accSlot(r) : VPM_NO_COL;
}
// values k, k + 1, k + 2 of the pair's pixel
template <typename LDS>
__device__ __forceinline__ void vpmAdd3(const GatherArgs &a, LDS &s, uint32_t col, int k, uint32_t b, float x, float y, float z) {
  if (col != VPM_NO_COL) {
    atomicAdd(&s.acc[k][col], (double)x);
    atomicAdd(&s.acc[k + 1][col], (double)y);
    atomicAdd(&s.acc[k + 2][col], (double)z);
  } else {
    const uint32_t pv = s.pix[b];
    float *p = &a.iter[((size_t)(pv >> 16) * a.cfg.width + (pv & 0xFFFFu)) * 27 + k];
    atomicAdd(p, x);
    atomicAdd(p + 1, y);
    atomicAdd(p + 2, z);
  }
}

// What both phases of an evaluation need of the (photon, sample) pair.
struct VpmPair {
  PhotonCold ph;
  RayReg base;
  f3 photonIn, baseContrib, rel;
  double t;
  float tf, r2, pdfBase, pdfSel, scale, trS;
  uint32_t edge, col;
  int px, py;
};

template <typename LDS>
__device__ __forceinline__ VpmPair vpmPair(const GatherArgs &a, const LDS &s, uint32_t pidx, uint32_t b, float norm) {
  VpmPair v;
  v.ph = loadCold(a, pidx);
  v.base = loadRayV(a, s, 0, b);
  v.edge = s.edge[b];
  v.col = vpmColumn(s, b);
  const uint32_t pix = s.pix[b];
  v.px = (int)(pix & 0xFFFFu);
  v.py = (int)(pix >> 16);
  const float r = s.radius[b];
  v.r2 = r * r;
  v.t = s.t[b];
  v.tf = (float)v.t;
  v.pdfBase = s.pdfBase[b];
  v.pdfSel = s.pdfSel[b];
  const float sigT = a.med.sigmaT[0];
  const f3 sigS = mk3(a.med.sigmaS[0], a.med.sigmaS[1], a.med.sigmaS[2]);
  const float kernelVol = (4.0f / 3.0f) * 3.14159265358979323846f * v.r2 * r;
  v.scale = norm / (kernelVol * v.pdfBase);
  v.photonIn = sigS * v.ph.flux;
  v.baseContrib = v.base.eye * (v.photonIn * phaseEval(a.med.g, v.ph.wi, -v.base.d)) * s.trBase[b];
  // photon relative to baseRay(maxt): the difference formed in fp64, carried as a small fp32 vector (round 5: the two fp64
  // points themselves are no longer kept -- twelve registers of a kernel that has none to spare)
  v.rel = tof(tod(v.ph.pos) - (tod(v.base.o) + tod(v.base.d) * v.t));
  // shiftMRec: Medium::eval(shiftRay, EDistanceAlwaysValid) with mRec.t = t: Tr = exp(-sigma_t t)
  v.trS = __expf(-sigT * v.tf);
  if (v.trS < 1e-20f) v.trS = 0.f;
  return v;
}

template <typename LDS>
__device__ __forceinline__ void vpmAddShift(LDS &s, uint32_t b, uint32_t col, int i, const f3 &sflux, const f3 &baseContrib, float w,
                                            float scale, int px, int py, const GatherArgs &a) {
  if ((i == GVPM_RIGHT && px == a.cfg.width - 1) || (i == GVPM_TOP && py == a.cfg.height - 1)) w = 1.f;
  const float ws = w * scale;
  if (sflux.x != 0.f || sflux.y != 0.f || sflux.z != 0.f) vpmAdd3(a, s, col, 3 + 3 * i, b, sflux.x * ws, sflux.y * ws, sflux.z * ws);
  vpmAdd3(a, s, col, 15 + 3 * i, b, baseContrib.x * ws, baseContrib.y * ws, baseContrib.z * ws);
}

// Phase 1 of one evaluation (VolumeGradientPositionQuery::operator() after the filters): the base contribution, and
// of each of the four shifts everything but the reconnection -- the null shift, the failed ones.  Returns the mask of
// the shifts that need shiftPhotonDiffuse; those are queued and run densely in phase 2 (87 % of the shifts at C1 are
// null shifts, but a batch that evaluates in one pass pays for the reconnection code of the few lanes that take it).
// The PRIMAL point estimate (sppm.cpp:1087-1112 + RadianceQueryVolume, src/librender/photonmap.cpp:277-308): the term of one
// (photon, sample) pair -- power * phase (no sigma_s) * beam.weight * Tr / (pdfSuccess * sel) * MCNorm / kernelVol
template <typename LDS>
__device__ __forceinline__ void vpmPrimalTerm(const GatherArgs &a, LDS &s, uint32_t pidx, uint32_t b, float norm) {
  const PhotonFront ph = loadFront(a, pidx);
  const RayReg base = loadRayV(a, s, 0, b);
  const float r = s.radius[b];
  const float kernelVol = (4.0f / 3.0f) * 3.14159265358979323846f * r * r * r;
  const float scale = norm / (kernelVol * s.pdfBase[b]);
  const f3 c = base.eye * ph.flux * (phaseEval(a.med.g, ph.wi, -base.d) * s.trBase[b] * scale);
  vpmAdd3(a, s, vpmColumn(s, b), 0, b, c.x, c.y, c.z);
}

template <bool HS, typename LDS>
__device__ __forceinline__ uint32_t vpmPhase1(const GatherArgs &a, LDS &s, uint32_t pidx, uint32_t b, float norm,
                                              uint32_t &nNull, uint32_t &nFail) {
#if GVPM_VPM_PROBE == 1
  return 0u;
#endif
  const VpmPair v = vpmPair(a, s, pidx, b, norm);
  vpmAdd3(a, s, v.col, 0, b, v.baseContrib.x * v.scale, v.baseContrib.y * v.scale, v.baseContrib.z * v.scale);
  const float sigT = a.med.sigmaT[0];
  uint32_t qMask = 0u;
  // (the evaluation kernel: shift i + 1's ray is requested while shift i is computed -- four round trips one after the other
  // were 60 % of its phase 1; the fused kernel has no registers for it)
  [[maybe_unused]] RayReg shAhead;
  if constexpr (LDS::RAYS_AHEAD) shAhead = loadRayV(a, s, 1, b);
#pragma unroll 1
  for (int i = 0; i < 4; ++i) {
    RayReg sh;
    if constexpr (LDS::RAYS_AHEAD) {
      sh = shAhead;
      if (i < 3) shAhead = loadRayV(a, s, 2 + i, b);
    } else {
      sh = loadRayV(a, s, 1 + i, b);
    }
    float w = 1.f;
    f3 sflux = mk3(0.f);
    // validShiftDist: valid edge and shiftDistMax >= baseRay.maxt (shift_volume_photon.cpp:546-566)
    // (round 5: a shift whose branch fp32 cannot decide -- the distance against the shifted edge's length, the null-shift
    // test against r^2, within the error band of the fp32 numbers -- is queued like a reconnection; phase 2 re-derives the
    // test and hands the shift to the exact pass, exact_shift.hip)
    // (cfg.reserved[4]: GVPM_EXACT_ALL -- every shift to the exact pass, tests/test_exact_pass_gpu.py)
    if (sh.valid && !(HS && GVPM_PF_SHIFT_TYPE(v.ph.bits) == 3u) && (a.cfg.reserved[4] || fabsf(v.tf - sh.len) <= 4e-7f * (v.tf + sh.len))) {
      qMask |= 0x11u << i;  // (bit 4 + i: "undecidable", carried to phase 2 in the queue entry)
      continue;
    }
    if (sh.valid && sh.len >= v.tf) {
      // photon relative to shiftRay(t) = (photon - baseRay(t)) - (shiftRay(t) - baseRay(t)), the second difference in fp64
      const f3 y = v.rel - tof((tod(sh.o) - tod(v.base.o)) + (tod(sh.d) - tod(v.base.d)) * v.t);
      const float y2 = dot(y, y);
#ifdef GVPM_DBG_SHIFT2
      printf("vpm phase 1: shift %d tf %.9g sh.len %.9g y2 %.9g r2 %.9g |rel| %.9g t %.12g\n", i, v.tf, sh.len, y2, v.r2, sqrtf(dot(v.rel, v.rel)), v.t);
#endif
      if (a.cfg.use_shift_null && !(HS && GVPM_PF_SHIFT_TYPE(v.ph.bits) == 3u) && fabsf(y2 - v.r2) <= 4e-6f * v.r2) {
        qMask |= 0x11u << i;
        continue;
      }
      if (a.cfg.use_shift_null && y2 < v.r2) {
        // shiftNull, shift_volume_photon.cpp:119-158
        // pdfShiftRay = shiftMRec.pdfSuccess * pdfSel, normalised over [Epsilon, shiftDistMax]
        const float normS = oneMinusExpNeg(sigT * (sh.len - a.cfg.epsilon));
        const float pdfShift = (sigT / normS) * __expf(-sigT * v.tf) * v.pdfSel;
        sflux = mk3(v.trS) * (v.photonIn * phaseEval(a.med.g, v.ph.wi, -sh.d)) * sh.eye;
        w = 0.5f;
        if (a.cfg.use_mis) {
          if (pdfShift == 0.f || v.pdfBase == 0.f) w = 1.f;
          else w = 1.f / (1.f + sensorMIS(sh, v.base, v.edge) * pdfShift / v.pdfBase);
        }
        nNull++;
      } else if (a.cfg.debug_shift != GVPM_SHIFT_NULL) {
        const uint32_t st = GVPM_PF_SHIFT_TYPE(v.ph.bits);
        // (HS: a manifold-typed photon goes to phase 2 too -- it records the host's request there, gvpm_enable_host_shifts)
        if (st == 1u || st == 2u || (HS && st == 3u)) {
          qMask |= 1u << i;
          continue;  // phase 2 adds this shift's terms
        }
        nFail++;
      }
    }
    vpmAddShift(s, b, v.col, i, sflux, v.baseContrib, w, v.scale, v.px, v.py, a);
  }
  return qMask;
}

// Phase 2: the reconnection of shift i (getShiftPos with coherent = false, shift_volume_photon.cpp:858-896, then
// shiftPhotonDiffuse) for one queued (photon, sample, shift).
template <bool FULLVIS, bool HS, typename LDS>
__device__ __forceinline__ void vpmPhase2(const GatherArgs &a, LDS &s, uint32_t pidx, uint32_t meta, float norm,
                                          uint32_t &nDiff, uint32_t &nFail, uint32_t sBase) {
  const uint32_t b = meta & 0xFFu;
  const int i = (int)((meta >> 8) & 0xFFu);
  // (bit 16: phase 1 could not decide this shift's branch and queued it to be deferred here)
  uint32_t amb = (meta >> 16) & 1u ? 2u : 0u;
  const VpmPair v = vpmPair(a, s, pidx, b, norm);
  const RayReg sh = loadRayV(a, s, 1 + i, b);
  const float sigT = a.med.sigmaT[0];
  const float normS = oneMinusExpNeg(sigT * (sh.len - a.cfg.epsilon));
  const float pdfShift = (sigT / normS) * __expf(-sigT * v.tf) * v.pdfSel;
  const d3 zP = tod(sh.o) + tod(sh.d) * v.t;
  f3 offRel = v.rel;
  if (a.cfg.use_shift_null) {
    const f3 dS = tof((tod(sh.o) - tod(v.base.o)) + (tod(sh.d) - tod(v.base.d)) * v.t);  // shiftRay(t) - baseRay(t)
    const f3 bo = dS + offRel;
    const float bo2 = dot(bo, bo);
    // (the mirror decision of getShiftPos moves the offset position by up to 2 r: not a counter, but a different shift)
    amb |= fabsf(bo2 - v.r2) <= 4e-6f * v.r2 ? 8u : 0u;
    if (bo2 < v.r2) offRel = offRel + dS * (-2.f * dot(dS, offRel) / dot(dS, dS));
  }
  if (HS && GVPM_PF_SHIFT_TYPE(v.ph.bits) == 3u) {
    // EManifoldShift (shiftPhoton -> shiftPhotonManifold, shift_volume_photon.cpp:49-117,160-295): the walk is the host's.
    // Recorded with what the device needs to finish the shift (apply_host_shifts_kernel): nothing is added now.  The base
    // contribution rides along already scaled, as G-BRE's does.
    const f3 zPf = tof(zP), basePtF = tof(tod(v.base.o) + tod(v.base.d) * v.t);
    if (!recordShiftRequest(reqSink(a), s.radius[b], pidx, s.set[b], i, zPf + offRel, basePtF, zPf, v.tf, v.trS, v.pdfBase, pdfShift,
                            sensorMIS(sh, v.base, v.edge), v.scale, v.baseContrib * v.scale, sh.d, sh.eye, s.pix[b])) {
      nFail++;  // the list is full: a failed shift (weight 1)
      vpmAddShift(s, b, v.col, i, mk3(0.f), v.baseContrib, 1.f, v.scale, v.px, v.py, a);
    }
    return;
  }
  const f3 dProjU = (tof(zP) - v.ph.parentPos) + offRel;
  bool ok = false;
  f3 sflux = mk3(0.f);
  uint32_t ambVis = 0u;
  const float w = shiftDiffuse<FULLVIS>(a, v.ph, v.ph.bits, dProjU, sh, v.base, v.edge, mk3(v.trS), v.pdfBase, pdfShift, sflux, ok, nullptr,
                                        -1.f, &ambVis);
#ifdef GVPM_DBG_SHIFT2
  printf("vpm phase 2: shift %d ok %d amb %u ambVis %u w %g tf %.9g sh.len %.9g\n", i, (int)ok, amb, ambVis, w, v.tf, sh.len);
#endif
  if (amb | ambVis) {
    // fp32 cannot decide this shift as the reference does: the exact pass evaluates it (nothing added, nothing counted)
    deferNote(a, GVPM_EX_KIND_VPM, s.sampleIndex(sBase, b), pidx, (uint32_t)i, amb | ambVis);
    return;
  }
  if (ok) nDiff++; else nFail++;
  vpmAddShift(s, b, v.col, i, sflux, v.baseContrib, w, v.scale, v.px, v.py, a);
}

// Waves per workgroup: they share nothing (an LDS slice each, wave-local synchronisation).  One wave per workgroup made the
// C1 launch -- 39 k workgroups that live ~6 us each -- DISPATCH-bound: 0.44 resident waves per SIMD on average where registers
// and LDS allow 3 (rocprofv3: SQ_WAVE_CYCLES / (SIMDs x kernel cycles)).
#ifndef GVPM_VPM_WPB
#define GVPM_VPM_WPB 1
#endif
constexpr int VPM_WPB = GVPM_VPM_WPB;
#ifndef GVPM_VPM_SPW
#define GVPM_VPM_SPW 64
#endif
constexpr uint32_t VPM_SPW = GVPM_VPM_SPW;  // camera samples per wave (<= 64: one per lane in the set-up phase)
// LDS accesses of ONE wave are executed in order; what has to be stopped is the compiler moving them
__device__ __forceinline__ void vpmWaveSync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifdef GVPM_VPM_TIMING
// probe builds only: per wave of the last launch {start, end (wall clock, 100 MHz), candidates, evaluations}
__device__ unsigned long long gvpmVpmLog[4 * 65536];
extern "C" int gvpm_debug_vpm_timing(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(gvpmVpmLog), sizeof(gvpmVpmLog)) == hipSuccess ? 0 : -1;
}
#endif
// inclusive prefix sum over the wave: row_shr 1 / 2 / 4 / 8 inside the rows of 16 lanes, then row_bcast 15 (rows 1 and 3 take
// the last lane of the row before) and row_bcast 31 (rows 2 and 3 take lane 31): GFX9 DPP, no address registers, no LDS traffic
__device__ __forceinline__ uint32_t waveScanInclDpp(uint32_t v) {
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
  return (uint32_t)x;
}
// One batch of camera samples [sBase, sBase + ns), ns <= 64, through one wave.
// MODE 0: walk and evaluation (the fused kernel; the parts of a heavy batch in vpm_redo_kernel).
// MODE 1: the walk alone (vpm_find_kernel): the pairs that pass the filters go to chunks of VpmSplit::pairs, 64 at a time, the
// samples that found photons leave their state for vpm_eval_kernel; a batch that finds the pool exhausted is named in the redo
// list and leaves nothing else behind.
template <bool FULLVIS, bool HS, bool PRIMAL, int MODE, typename LDS>
__device__ __forceinline__ void vpmBatch(const GatherArgs &a, LDS &s, const int lane, const uint32_t batch, const uint32_t sBase,
                                         const uint32_t ns, const bool wholeBatch, const VpmSplit &sp) {
#ifdef GVPM_VPM_TIMING
  const unsigned long long tw0 = wall_clock64();
#endif
  // (wave-uniform, but a VALU quotient: handed to the scalar file, or it is carried -- and spilled -- as a vector register)
  const float norm = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(1.f / (float)a.cfg.nb_camera_samples)));
  const float eps = a.cfg.epsilon;

  if constexpr (MODE == 0)
    for (int idx = lane; idx < 27 * VPM_RUNS * VPM_SUB; idx += 64) (&s.acc[0][0])[idx] = 0.0;
  uint32_t nRuns = 0;  // wave-uniform: runs that have LDS accumulators

  // ---- this lane's sample: rays -> LDS, distance sampling ----
  bool active = (uint32_t)lane < ns;
  uint32_t set = 0;
  float rnd = 0.f, pdfSel = 1.f;
  if (active) {
    const gvpm_vpm_sample sm = a.samples[sBase + lane];
    set = sm.set;
    rnd = sm.rand;
    pdfSel = sm.pdf_sel;
    if (set >= a.nsets) active = false;
  }
  s.set[lane] = active ? set : 0xFFFFFFFFu;
  // (the base ray's four quads in one round trip: the pixel and the edge are in its last one)
  RayReg base;
  base.o = base.eye = mk3(0.f);
  base.d = mk3(0.f, 0.f, 1.f);
  base.len = 1e-30f;
  base.pdf = base.jac = base.gop = 0.f;
  base.valid = false;
  {
    float4 q3 = make_float4(0, 0, 0, 0);
    if (active) {
      const float4 *rp = reinterpret_cast<const float4 *>(a.rays + (size_t)set * 5);
      const float4 q0 = rp[0], q1 = rp[1], q2 = rp[2];
      q3 = rp[3];
      base.o = mk3(q0.x, q0.y, q0.z);
      base.len = fabsf(q0.w);
      base.valid = GVPM_RAY_VALID(__float_as_uint(q3.y)) != 0;
      base.d = mk3(q1.x, q1.y, q1.z);
      base.pdf = q1.w;
      base.eye = mk3(q2.x, q2.y, q2.z);
      base.jac = q2.w;
      base.gop = q3.x;
    }
    s.pix[lane] = __float_as_uint(q3.w);
    s.initSample(lane, GVPM_RAY_EDGE(__float_as_uint(q3.y)));  // (the edge; the photon count M at zero)
    // pixel runs: consecutive lanes of one pixel (the C ABI does not promise that a pixel's samples are adjacent: a
    // pixel that comes back later in the wave is another run)
    if constexpr (MODE == 0) {
      const uint32_t pv = __float_as_uint(q3.w);
      const uint32_t prev = __shfl_up(pv, 1u, 64);
      const bool head = lane == 0 || prev != pv;
      const unsigned long long heads = __ballot(head);
      const uint32_t run = (uint32_t)__popcll(heads & ((2ull << lane) - 1ull)) - 1u;
      s.run[lane] = run;
      if (head && run < (uint32_t)VPM_RUNS) s.runPix[run] = pv;
      if (lane == 0) nRuns = min((uint32_t)__popcll(heads), (uint32_t)VPM_RUNS);
    }
  }
  if constexpr (MODE == 0) nRuns = __shfl(nRuns, 0, 64);
  vpmWaveSync();
  active = active && base.valid;
  // HomogeneousMedium::sampleDistance(Ray(o, d, Epsilon, beamDist), EDistanceAlwaysValid, rand),
  // homogeneous.cpp:293-430 (balance strategy, currentMediumSampling = 1)
  const double sigT = (double)a.med.sigmaT[1];
  double t = 0.0;
  float pdfBase = 0.f, trBase = 0.f;
  const uint32_t pixv = s.pix[lane];
  const uint32_t pixIdx = (pixv >> 16) * (uint32_t)a.cfg.width + (pixv & 0xFFFFu);
  float radius = 0.f;
  if (active) {
    const double mint = (double)eps, maxt = (double)base.len;
    const double maxDist = fmax((maxt - mint) - (double)eps, 0.0);
    const double normalization = 1.0 - exp(-sigT * maxDist);
    const double sampled = -log(1.0 - (double)rnd * normalization) / sigT;
    const double distSurf = maxt - mint;
    if (sampled < distSurf) {
      t = sampled + mint;
      // exp(-sigma_t * sampled) IS the argument of the logarithm above (to a rounding of the double, far below the float
      // the two results are stored as); the normalisation over the whole edge feeds a float too: one fp64 exponential and
      // one logarithm per sample instead of three and one
      const double e = 1.0 - (double)rnd * normalization;
      const float nrm2 = oneMinusExpNeg((float)sigT * (float)distSurf);
      pdfBase = ((float)sigT / nrm2) * (float)e * pdfSel;  // mRec.pdfSuccess * pdfSel
      trBase = (float)e;
      if (trBase < 1e-20f) trBase = 0.f;
      // querySize = R * POURCENTAGE_BS * gp.scaleVol, gvpm.cpp:1082,1132
      radius = (a.cfg.bsphere_radius * 0.01f) * a.scaleVol[pixIdx];
    } else {
      active = false;  // "Failed to sample the distance" (cannot happen for rand < 1)
    }
  }
  s.t[lane] = t;
  if constexpr (MODE == 0) {
    s.pdfBase[lane] = pdfBase;
    s.pdfSel[lane] = pdfSel;
    s.trBase[lane] = trBase;
    s.radius[lane] = radius;
  }
  vpmWaveSync();

  // ---- cell box of the query sphere ----
  const d3 qD = tod(base.o) + tod(base.d) * t;
  const f3 q = tof(qD);
  const Grid gr = a.grid;
  int bx0 = 0, bx1 = -1, by0 = 0, by1 = -1, bz0 = 0, bz1 = -1;
  if (active && a.nph > 0) {
    const float pad = radius * 1.0001f + 1e-6f;
    bx0 = max(0, (int)floorf((q.x - pad - gr.org[0]) * gr.invCell));
    bx1 = min(gr.dim[0] - 1, (int)floorf((q.x + pad - gr.org[0]) * gr.invCell));
    by0 = max(0, (int)floorf((q.y - pad - gr.org[1]) * gr.invCell));
    by1 = min(gr.dim[1] - 1, (int)floorf((q.y + pad - gr.org[1]) * gr.invCell));
    bz0 = max(0, (int)floorf((q.z - pad - gr.org[2]) * gr.invCell));
    bz1 = min(gr.dim[2] - 1, (int)floorf((q.z + pad - gr.org[2]) * gr.invCell));
  }
  const int nyr0 = by1 - by0 + 1, nzr = bz1 - bz0 + 1;
  const int nrows = (bx1 >= bx0 && nyr0 > 0 && nzr > 0) ? nyr0 * nzr : 0;
  // What the passes below need of this lane's box, in TWO registers (the grid has at most 385 cells an axis: 10 bits each);
  // the query point and radius come back from s.qr[lane].  Ten registers carried across the whole walk -- with phase 2's
  // 168 in the way the compiler spilled them, and scratch costs this kernel's 39 k short waves at launch (NB r5).
  const uint32_t boxA = nrows ? ((uint32_t)bx0 | ((uint32_t)bx1 << 10) | ((uint32_t)by0 << 20)) : 0u;
  const uint32_t boxB = nrows ? ((uint32_t)bz0 | ((uint32_t)nyr0 << 10) | ((uint32_t)nrows << 14)) : 0u;
  // The wave walks the cells together.  A lane-per-sample walk runs as long as its busiest lane (the candidate counts
  // of the samples of a ray differ by orders of magnitude) and reads 64 scattered lines per trip.  Instead the photon
  // ranges of ALL rows of ALL 64 boxes are laid end to end (round 4: one wave prefix sum over the samples' totals and a
  // per-sample prefix over its <= 9 rows, where round 3 made one list -- scan, two barriers, a mostly empty last trip --
  // per row: 9 of each per wave for ~14 candidates a sample at C1), and lane l of a trip takes candidate j0 + l of the
  // list: it finds the owning sample by a search over the samples' offsets, the row by a search over that sample's row
  // offsets, and tests the photon against that sample's sphere.  Consecutive lanes read consecutive photons within a
  // row and every trip but the last is full.
  int maxRows = nrows;
  {
    // (its own copy of the lane index: the six permute addresses (lane ^ o) * 4 would otherwise be SHARED with the counters'
    // reduction at the kernel's end -- and kept, spilled, across everything in between)
    int laneR = lane;
    asm volatile("" : "+v"(laneR));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) maxRows = max(maxRows, __shfl(maxRows, laneR ^ o, 64));
  }
  s.qr[lane] = make_float4(q.x, q.y, q.z, radius);
  // A row of the box is a run of cells along x at one (y, z): only the part of it the sphere can reach is listed -- the
  // cells of a 3 x 3 x 3 box hold ~6 times the sphere's volume, the trimmed rows ~3 times (C1: 14.6 -> ~8 candidates a
  // sample).  Conservative: the reach carries 1e-4 r + 1e-6 + 2e-4 cells of slack -- a cell's bounds are rebuilt here as
  // org + index * cell, off the build's floor((p - org) * invCell) by up to ~1e-7 * index cells.
  // (no branch around the two loads: a row that lists nothing reads cell 0 and drops it -- the eighteen loads of a pass are
  // then issued back to back instead of one round trip after the other behind their conditions)
  // (row r of the box is (yy, zz) = (r % nyr, r / nyr): the caller steps them -- a quotient by a per-lane divisor is twenty
  // instructions, nine times a pass; the row's base cell through 24-bit multiplies: a grid has at most 385 cells an axis)
  auto rowRange = [&](int r, int yy, int zz, uint32_t &c, uint32_t &e) __attribute__((always_inline)) {
    const int nrowsL = (int)(boxB >> 14);
    const int bx0L = (int)(boxA & 1023u), bx1L = (int)((boxA >> 10) & 1023u), by0L = (int)(boxA >> 20);
    const int bz0L = (int)(boxB & 1023u);
    const float4 qr = s.qr[lane];
    const float padW = qr.w * 1.0001f + 1e-6f + 2e-4f * gr.cell, pad2 = padW * padW;
    const int y = by0L + yy, z = bz0L + zz;
    const float ylo = gr.org[1] + (float)y * gr.cell, zlo = gr.org[2] + (float)z * gr.cell;
    const float dy = fmaxf(0.f, fmaxf(ylo - qr.y, qr.y - (ylo + gr.cell))), dz = fmaxf(0.f, fmaxf(zlo - qr.z, qr.z - (zlo + gr.cell)));
    const float h2 = pad2 - (dy * dy + dz * dz);
    const float hx = sqrtf(fmaxf(h2, 0.f)) + 1e-6f;
    const int x0 = max(bx0L, (int)floorf((qr.x - hx - gr.org[0]) * gr.invCell));
    const int x1 = min(bx1L, (int)floorf((qr.x + hx - gr.org[0]) * gr.invCell));
    const bool ok = r < nrowsL && h2 > 0.f && x1 >= x0;
    const uint32_t rb = __umul24(__umul24((uint32_t)z, (uint32_t)gr.dim[1]) + (uint32_t)y, (uint32_t)gr.dim[0]);
    const uint32_t cc = a.cellStart[ok ? rb + x0 : 0u], ee = a.cellStart[ok ? rb + x1 + 1 : 0u];
    c = ok ? cc : 0u;
    e = ok ? ee : 0u;
  };
  uint32_t qHead = 0, qCount = 0;
  [[maybe_unused]] uint32_t rqHead = 0, rqCount = 0;
  // MODE 1: the pool has no chunk left (wave-uniform); the chunk reserved for the next 64 pairs (lane 0)
  [[maybe_unused]] bool slotFull = false;
  [[maybe_unused]] uint32_t chunkAhead = 0;
  if constexpr (MODE == 1)
    if (lane == 0) chunkAhead = atomicAdd(&sp.ctl[(blockIdx.x % VPM_SHARDS) * 32u], 1u);
  [[maybe_unused]] unsigned long long candAtFull = 0;
  uint32_t nNull = 0, nDiff = 0, nFail = 0;
  // candidates and evaluations are counted per WAVE, in the scalar file (a trip's candidates: its width; a batch's evaluations:
  // its ballot): two vector registers less across the walk
  unsigned long long nCandW = 0, nEvalW = 0;
  auto drain = [&](uint32_t n) __attribute__((always_inline)) {  // phase 2 for the first n <= 64 queued reconnections
    if constexpr (MODE == 0) {
      vpmWaveSync();
      if ((uint32_t)lane < n) {
        const uint2 e = s.rq[(rqHead + lane) % VRQ];
        vpmPhase2<FULLVIS, HS>(a, s, e.x, e.y, norm, nDiff, nFail, sBase);
      }
      rqHead = (rqHead + n) % VRQ;
      rqCount -= n;
      vpmWaveSync();
    }
  };
  auto evalBatch = [&](bool valid, uint2 e) __attribute__((always_inline)) {  // phase 1 for one (photon, sample) pair per lane
    if constexpr (MODE == 0) {
      uint32_t qMask = 0u;
      if (valid) {
        if (PRIMAL) vpmPrimalTerm(a, s, e.x, e.y, norm);
        else qMask = vpmPhase1<HS>(a, s, e.x, e.y, norm, nNull, nFail);
      }
      nEvalW += (unsigned long long)__popcll(__ballot(valid));
#pragma unroll 1
      for (uint32_t i = 0; i < 4u; ++i) {
        const bool want = (qMask >> i) & 1u;
        const unsigned long long m = __ballot(want);
        if (m) {
          if (want) {
            const uint32_t off = __popcll(m & ((1ull << lane) - 1ull));
            s.rq[(rqHead + rqCount + off) % VRQ] = make_uint2(e.x, e.y | (i << 8) | (((qMask >> (4u + i)) & 1u) << 16));
          }
          rqCount += __popcll(m);
          if (rqCount >= 64u) drain(64u);
        }
      }
    } else {
      // the walk alone: the first n = popcount(valid) queued pairs (the valid lanes are the low ones) to a chunk of the pool
      // (the chunk was reserved ahead -- at the wave's start, or when the chunk before it was filled -- so that nobody waits
      // for the cursor's round trip here; the one reserved last and not needed is closed empty at the end)
      const uint32_t n = (uint32_t)__popcll(__ballot(valid));
      const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)chunkAhead);
      if (c >= sp.shardChunks) {
        slotFull = true;
      } else {
        const uint32_t chunk = (blockIdx.x % VPM_SHARDS) * sp.shardChunks + c;
        if (valid) sp.pairs[(size_t)chunk * 64u + (uint32_t)lane] = make_uint2(e.x, sBase + e.y);
        if (lane == 0) {
          sp.chunkMeta[chunk] = make_uint2(n, batch);
          chunkAhead = atomicAdd(&sp.ctl[(blockIdx.x % VPM_SHARDS) * 32u], 1u);
        }
      }
    }
  };
#ifdef GVPM_VPM_TIMING
  [[maybe_unused]] const unsigned long long twSetup = wall_clock64();
  [[maybe_unused]] unsigned long long twRows = 0;
#endif
  for (int r0 = 0; r0 < maxRows && !slotFull; r0 += VPM_ROWS) {
    // this lane's rows of the pass: all ranges in flight together, then the lane's own prefix over them
    uint32_t rcs[VPM_ROWS], res[VPM_ROWS];
    {
      const int nyr = max(1, (int)((boxB >> 10) & 15u));
      int yy = r0 ? r0 % nyr : 0, zz = r0 ? r0 / nyr : 0;
#pragma unroll
      for (int k = 0; k < VPM_ROWS; ++k) {
        rowRange(r0 + k, yy, zz, rcs[k], res[k]);
        if (++yy == nyr) {
          yy = 0;
          ++zz;
        }
      }
    }
    vpmWaveSync();  // (the previous pass has read its offsets)
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < VPM_ROWS; ++k) {
      s.rowStart[k][lane] = rcs[k];
      s.setRowOff(k, lane, cnt);
      cnt += res[k] - rcs[k];
    }
    // (the scan through DPP row shifts and broadcasts -- the sequence LLVM's own atomic optimizer emits for wave64 -- instead of
    // six ds_bpermute: their six address registers were carried, spilled, across the whole walk and reloaded every pass)
    if constexpr (MODE == 1)
      if (__ballot(cnt > LDS::ROW_LIMIT)) {  // (the walk kernel's 16-bit row offsets: the fused code takes the batch)
        slotFull = true;
        break;
      }
    const uint32_t inc = waveScanInclDpp(cnt);
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    s.segOff[lane] = inc - cnt;
    if (lane == 63) s.segOff[64] = total;
    vpmWaveSync();
#ifdef GVPM_VPM_TIMING
    if (r0 == 0) twRows = wall_clock64();
#endif
    candAtFull = nCandW + total;  // (what the cost key of a batch that stops here says: the pass's whole list)
    auto locate = [&](uint32_t j, uint32_t &gi, uint32_t &owner) __attribute__((always_inline)) {
      // the last sample whose offset is <= j (empty lists share the offset of the list after them) ...
      owner = 0;
#pragma unroll
      for (int st = 32; st > 0; st >>= 1)
        if (s.segOff[owner + st] <= j) owner += st;
      // ... and the last of its rows whose offset is <= the entry's place in the sample's list
      const uint32_t kk = j - s.segOff[owner];
      uint32_t row = 0;
      if (s.getRowOff(8, owner) <= kk) {
        row = 8;
      } else {
#pragma unroll
        for (int st = 4; st > 0; st >>= 1)
          if (s.getRowOff(row + st, owner) <= kk) row += st;
      }
      gi = s.rowStart[row][owner] + (kk - s.getRowOff(row, owner));
    };
    // (the walk kernel: the next trip's photon is located and its load issued before this trip's is tested -- a round trip per
    // trip was most of a wave's life; the fused kernel has no registers for it)
    [[maybe_unused]] bool haveN = false;
    [[maybe_unused]] uint32_t giN = 0, ownerN = 0;
    [[maybe_unused]] float4 hpN = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (MODE == 1) {
      haveN = (uint32_t)lane < total;
      if (haveN) {
        locate((uint32_t)lane, giN, ownerN);
        hpN = a.hot[giN];
      }
    }
    for (uint32_t j0 = 0; j0 < total && !slotFull; j0 += 64u) {
      nCandW += min(64u, total - j0);
      const uint32_t j = j0 + (uint32_t)lane;
      bool have = j < total;
      bool hit = false;
      uint32_t gi = 0, owner = 0;
      float4 hp = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (MODE == 1) {
        have = haveN;
        gi = giN;
        owner = ownerN;
        hp = hpN;
        haveN = j + 64u < total;
        if (haveN) {
          locate(j + 64u, giN, ownerN);
          hpN = a.hot[giN];
        }
      } else {
        if (have) {
          locate(j, gi, owner);
          hp = a.hot[gi];
        }
      }
      if (have) {
        const float4 qr = s.qr[owner];
        const f3 p = mk3(hp.x, hp.y, hp.z);
        const f3 qo = mk3(qr.x, qr.y, qr.z);
        const float rad = qr.w, r2f = rad * rad;
        const f3 dv = p - qo;
        const float d2 = dot(dv, dv);
        // pointDistSquared < distSquared (kdtree.h:722) decided in fp32 unless within the error band
        const float E = 3e-7f * (fabsf(p.x) + fabsf(p.y) + fabsf(p.z) + fabsf(qo.x) + fabsf(qo.y) + fabsf(qo.z));
        const float band = 4.f * rad * E + r2f * 2e-6f;
        bool inside = d2 < r2f - band;
        if (!inside && d2 < r2f + band) {
#pragma clang fp contract(off)
          const RayReg bo = loadRayV(a, s, 0, (int)owner);
          const d3 qd = tod(bo.o) + tod(bo.d) * s.t[owner];
          const double dx = (double)p.x - qd.x, dy = (double)p.y - qd.y, dz = (double)p.z - qd.z;
          // (against the DOUBLE product R * 0.01 * scaleVol, gvpm.cpp:1082,1132 -- `rad` is its fp32 rounding, 6e-8 off: at a
          // distance that close to the radius it decided one pair in 3 10^7 the other way, tests/stress_vpm.py)
          const uint32_t pw = s.pix[owner];
          const double rD = ((double)a.cfg.bsphere_radius * 0.01) * (double)a.scaleVol[(size_t)(pw >> 16) * a.cfg.width + (pw & 0xFFFFu)];
          inside = dx * dx + dy * dy + dz * dz < rD * rD;
        }
        if (inside) {
          const uint32_t bits = __float_as_uint(hp.w);
          hit = true;
          if (PRIMAL) {
            // RadianceQueryVolume (photonmap.cpp:287-297): radius, then depth against maxDepth = m_maxDepth - beam.depth --
            // `maxDepth > 0 &&` as written: a bound of zero or less filters nothing -- and M counts what passes both
            const int md = a.cfg.max_depth > 0 ? a.cfg.max_depth - (int)s.edgeOf(owner) : 0x7FFFFFFF;
            if (md > 0 && (int)GVPM_PF_DEPTH(bits) > md) hit = false;
            if (hit) atomicAdd(&s.found[owner], 1u);
          } else {
            atomicAdd(&s.found[owner], 1u);
            // filters, shift_volume_photon.cpp:503-521 (maxDepth only; no path-set in G-VPM)
            const int depth = (int)GVPM_PF_DEPTH(bits) + (int)s.edgeOf(owner);
            if (a.cfg.max_depth > 0 && depth > a.cfg.max_depth) hit = false;
            if (!((bits >> 6) & 1u)) hit = false;
          }
        }
      }
      const unsigned long long m = __ballot(hit);
      if (m) {
        if (hit) {
          const uint32_t off = __popcll(m & ((1ull << lane) - 1ull));
          s.queue[(qHead + qCount + off) % VQ] = make_uint2(gi, owner);
        }
        qCount += __popcll(m);
        if (qCount >= 64u) {
          vpmWaveSync();
          evalBatch(true, s.queue[(qHead + lane) % VQ]);
          qHead = (qHead + 64u) % VQ;
          qCount -= 64u;
          vpmWaveSync();
        }
      }
    }
  }
  vpmWaveSync();
#ifdef GVPM_VPM_TIMING
  [[maybe_unused]] const unsigned long long twTrips = wall_clock64() - tw0;
#endif
  if (qCount && !slotFull) evalBatch((uint32_t)lane < qCount, s.queue[(qHead + lane) % VQ]);
  if constexpr (MODE == 0) {
    if (rqCount) drain(rqCount);
  } else {
    if (slotFull) {
      // the pool is exhausted: the whole batch is the fused kernel's (vpm_redo_kernel); nothing of this walk counts, the chunks
      // it has written are skipped
      if (lane == 0) {
        if (chunkAhead < sp.shardChunks) sp.chunkMeta[(blockIdx.x % VPM_SHARDS) * sp.shardChunks + chunkAhead] = make_uint2(0u, batch);
        sp.status[batch] = 1u;
        sp.redo[atomicAdd(&sp.ctl[VPM_CTL_REDO], 1u)] = batch;
        if (a.vpmCostKey) {
          a.vpmCostKey[batch] = 0xFFFFFu - (uint32_t)min(candAtFull, 0xFFFFFull);
          a.vpmCostVal[batch] = batch;
        }
      }
      return;
    }
    if (lane == 0) {
      sp.status[batch] = 0u;
      if (chunkAhead < sp.shardChunks) sp.chunkMeta[(blockIdx.x % VPM_SHARDS) * sp.shardChunks + chunkAhead] = make_uint2(0u, batch);
    }
    vpmWaveSync();
    if (s.foundOf(lane) != 0u) {
      VpmSampleState st;
      st.t = t;
      st.pdfBase = pdfBase;
      st.trBase = trBase;
      st.radius = radius;
      st.pdfSel = pdfSel;
      st.set = set;
      st.pix = s.pix[lane];
      st.edge = s.edgeOf(lane);
      st.pad[0] = st.pad[1] = st.pad[2] = 0u;
      sp.state[sBase + (uint32_t)lane] = st;
    }
  }
  vpmWaveSync();
  // ---- write out ----
  // one global atomic per (run, value): per-sample atomics would put up to 64 operations on one address, and those
  // serialise in L2 (a run that overflowed VPM_RUNS has added to the film directly)
  if constexpr (MODE == 0) {
    for (uint32_t idx = (uint32_t)lane; idx < 27u * nRuns; idx += 64u) {
      const uint32_t k = idx / nRuns, rr = idx % nRuns;
      double vd = 0.0;
#pragma unroll
      for (int c = 0; c < VPM_SUB; ++c) vd += s.acc[k][rr * VPM_SUB + c];
      const float v = (float)vd;
      if (v != 0.f) {
        const uint32_t pv = s.runPix[rr];
        atomicAdd(&a.iter[((size_t)(pv >> 16) * a.cfg.width + (pv & 0xFFFFu)) * 27 + k], v);
      }
    }
  }
  {
    // the photon counts M of the samples of a run, combined in the wave (segmented suffix sum keyed by the pixel)
    const uint32_t pixE = s.pix[lane];  // (read again: not carried across the walk)
    const uint32_t pixIdxE = (pixE >> 16) * (uint32_t)a.cfg.width + (pixE & 0xFFFFu);
    const uint32_t prev = __shfl_up(pixE, 1u, 64);
    const bool head = lane == 0 || prev != pixE;
    const unsigned long long heads = __ballot(head);
    uint32_t same = 0;  // bit j: no run starts in lanes lane+1 .. lane+2^j, i.e. lane + 2^j is in this lane's run
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int o = 1 << j;
      if (lane + o < 64 && (((heads >> (lane + 1)) & ((1ull << o) - 1ull)) == 0ull)) same |= 1u << j;
    }
    float fv = (float)s.foundOf(lane);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const float w = __shfl_down(fv, 1u << j, 64);
      if ((same >> j) & 1u) fv += w;
    }
    if (head && fv != 0.f) atomicAdd(&a.mvol[pixIdxE], fv);
  }
  {
    unsigned long long nu = nNull, di = nDiff, fa = nFail;
    const unsigned long long ev = nEvalW, ca = nCandW;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      nu += __shfl_xor(nu, o, 64);
      di += __shfl_xor(di, o, 64);
      fa += __shfl_xor(fa, o, 64);
    }
#ifdef GVPM_VPM_TIMING
    {
      const uint32_t wid = batch;
      if (lane == 0 && wid < 65536u) {
        gvpmVpmLog[4 * wid] = tw0;
        gvpmVpmLog[4 * wid + 1] = wall_clock64();
#ifdef GVPM_VPM_TIMING2
        gvpmVpmLog[4 * wid + 2] = twSetup;
        gvpmVpmLog[4 * wid + 3] = ((twRows ? twRows - tw0 : 0ull) & 0xFFFFFFFFull) | ((unsigned long long)twTrips << 32);
#else
        gvpmVpmLog[4 * wid + 2] = ca;
        gvpmVpmLog[4 * wid + 3] = ev;
#endif
      }
    }
#endif
    if (lane == 0 && a.vpmCostKey && wholeBatch) {  // (a part of a heavy batch: the walk that sent it to the redo list has said)
      a.vpmCostKey[batch] = 0xFFFFFu - (uint32_t)min(ca, 0xFFFFFull);
      a.vpmCostVal[batch] = batch;
    }
    if (GVPM_VPM_PROBE != 3 && lane == 0 && (ev | ca)) {
      atomicAdd(&statRow(a)[0], ev);
      atomicAdd(&statRow(a)[1], ca);
      atomicAdd(&statRow(a)[2], nu);
      atomicAdd(&statRow(a)[3], di);
      atomicAdd(&statRow(a)[4], fa);
    }
  }
}

template <bool FULLVIS, bool HS, bool PRIMAL = false>
__global__ __launch_bounds__(64 * VPM_WPB) __attribute__((amdgpu_waves_per_eu(GVPM_VPM_MINW))) void gather_vpm_kernel(GatherArgs a) {
  __shared__ VpmLds sAll[VPM_WPB];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t slot = blockIdx.x * (uint32_t)VPM_WPB + (uint32_t)wv;
  // heaviest first (gatherVPM): slots below vpmOrderN take the permutation's batch, the others their own
  const uint32_t batch = slot < a.vpmOrderN ? a.vpmOrder[slot] : slot;
  const uint32_t sBase = batch * VPM_SPW;
  if (sBase >= a.nsamples) return;  // (no workgroup barrier anywhere: the waves run independently)
  vpmBatch<FULLVIS, HS, PRIMAL, 0>(a, sAll[wv], lane, batch, sBase, min(VPM_SPW, a.nsamples - sBase), true, VpmSplit{});
}

// ---- G-VPM as three kernels (the default; GVPM_VPM_SPLIT=0: the fused kernel above) ---------------------------------------
// The fused kernel's waves live ~25 us on a chain of dependent loads and evaluate what their 64 samples found -- 43 pairs at
// C1, in one phase-1 batch two thirds full and one phase-2 batch one third full -- at three waves per SIMD (168 VGPRs: the
// evaluation's); and the pixels that look at the light (a thousand pairs a batch) are evaluated by ONE wave each, 64 pairs after
// 64.  Split, as G-BRE's traversal and evaluation are:
//   vpm_find_kernel   the walk alone (the fused kernel's own code, MODE 1): 62 registers, so five waves per SIMD; 64 pairs at
//                     a time go to a chunk of a pool (a cursor per shard of the launch: one atomic per chunk), the state of
//                     every sample that found photons to one 48-byte record;
//   vpm_eval_kernel   a wave takes VPM_EVAL_CHUNKS consecutive chunks of a shard, lays their pairs end to end and evaluates
//                     them 64 at a time: at most that many batches a wave, all but its last full -- the heavy pixels'
//                     pairs spread over as many waves as they fill chunks;
//   vpm_redo_kernel   the fallback: batches that found the pool exhausted go through the fused code (VPM_REDO_PARTS waves a
//                     batch, sixteen samples each).  Empty unless a step finds over four times as many pairs as it has
//                     samples (the pool's size, gatherVPM).
// The walk decides pairs exactly as the fused kernel does (one source); the evaluation adds the same terms.
// (6.9 KB: 22 waves per CU where the fused kernel's layout allowed 18 -- the walk is a chain of round trips, its rate is the
// number of waves in flight: a row's offset within its sample's list in 16 bits (a sample that lists 65 536 candidates or more
// in a pass sends its batch to the redo list), the edge in the top byte of the photon count)
struct VpmFindLds {
  uint32_t set[64];
  uint2 queue[VQ];
  double t[64];
  uint32_t pix[64];
  float4 qr[64];
  uint32_t segOff[64 + 1];
  uint32_t rowStart[VPM_ROWS][64];
  uint16_t rowOff16[VPM_ROWS - 1][64];  // rows 1 ..: row 0 starts its sample's list
  uint32_t found[64];                   // photons inside the query sphere | edge << 24
  static constexpr uint32_t ROW_LIMIT = 0xFFFFu;
  __device__ __forceinline__ void initSample(int lane, uint32_t e) { found[lane] = e << 24; }
  __device__ __forceinline__ uint32_t edgeOf(uint32_t b) const { return found[b] >> 24; }
  __device__ __forceinline__ uint32_t foundOf(uint32_t b) const { return found[b] & 0xFFFFFFu; }
  __device__ __forceinline__ void setRowOff(int k, int lane, uint32_t v) {
    if (k > 0) rowOff16[k - 1][lane] = (uint16_t)v;
  }
  __device__ __forceinline__ uint32_t getRowOff(uint32_t k, uint32_t b) const { return k ? (uint32_t)rowOff16[k - 1][b] : 0u; }
};
#ifndef GVPM_VPM_FIND_MINW
#define GVPM_VPM_FIND_MINW 6
#endif
#ifndef GVPM_VPM_REDO_PARTS
#define GVPM_VPM_REDO_PARTS 4
#endif
constexpr uint32_t VPM_REDO_PARTS = GVPM_VPM_REDO_PARTS;
#ifndef GVPM_VPM_EVAL_CHUNKS
#define GVPM_VPM_EVAL_CHUNKS 4
#endif
constexpr int VPM_EVAL_CHUNKS = GVPM_VPM_EVAL_CHUNKS;

// waves per workgroup of the walk (they share nothing: an LDS slice each, no workgroup barrier)
#ifndef GVPM_VPM_FIND_WPB
#define GVPM_VPM_FIND_WPB 1
#endif
constexpr uint32_t VPM_FIND_WPB = GVPM_VPM_FIND_WPB;
__global__ __launch_bounds__(64 * GVPM_VPM_FIND_WPB) __attribute__((amdgpu_waves_per_eu(GVPM_VPM_FIND_MINW))) void vpm_find_kernel(GatherArgs a, VpmSplit sp) {
  __shared__ VpmFindLds sAll[VPM_FIND_WPB];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t slot = blockIdx.x * VPM_FIND_WPB + (uint32_t)wv;
  if (slot == 0 && lane == 0 && sp.zeroWord) *sp.zeroWord = 0u;
  const uint32_t batch = slot < a.vpmOrderN ? a.vpmOrder[slot] : slot;
  const uint32_t sBase = batch * VPM_SPW;
  if (sBase >= a.nsamples) return;
  vpmBatch<false, false, false, 1>(a, sAll[wv], lane, batch, sBase, min(VPM_SPW, a.nsamples - sBase), true, sp);
}

template <bool FULLVIS, bool HS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GVPM_VPM_MINW))) void vpm_redo_kernel(GatherArgs a, VpmSplit sp) {
  __shared__ VpmLds s;
  const int lane = threadIdx.x;
  const uint32_t nUnits = __hip_atomic_load(&sp.ctl[VPM_CTL_REDO], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * VPM_REDO_PARTS;
  constexpr uint32_t PART = VPM_SPW / VPM_REDO_PARTS;
  for (uint32_t u = blockIdx.x; u < nUnits; u += gridDim.x) {
    const uint32_t batch = sp.redo[u / VPM_REDO_PARTS];
    const uint32_t sBase = batch * VPM_SPW + (u % VPM_REDO_PARTS) * PART;
    // (inlined: a call would take the argument block by address, i.e. through scratch)
    if (sBase < a.nsamples) vpmBatch<FULLVIS, HS, false, 0>(a, s, lane, batch, sBase, min(PART, a.nsamples - sBase), false, sp);
    vpmWaveSync();
  }
}

// the evaluation's LDS: the sample state of the 64 PAIRS of a batch (lane l's pair, not sample l), 32 accumulator columns
// shared by the batch's pixel runs (one run: 32 copies of its sums, 16 runs: two each; the 33rd run adds to the film)
struct VpmEvalLds {
  uint32_t set[64];
  double acc[27][32];
  uint32_t run[64];      // first accumulator column of the pair's run (>= RUN_LIMIT: none)
  uint32_t runPix[32];
  double t[64];
  float pdfBase[64];
  float pdfSel[64];
  float trBase[64];
  float radius[64];
  uint32_t pix[64];
  uint32_t edge[64];
  uint32_t gs[64];       // the pair's camera sample (the exact pass's notes name it)
  uint32_t segOff[VPM_EVAL_CHUNKS + 1];
  uint32_t subMask;      // copies per run - 1
  uint2 rq[VRQ];
  static constexpr uint32_t RUN_LIMIT = 32u;
  static constexpr bool RAYS_AHEAD = true;
  __device__ __forceinline__ uint32_t accSlot(uint32_t r) const { return r + (threadIdx.x & subMask); }
  __device__ __forceinline__ uint32_t sampleIndex(uint32_t, uint32_t b) const { return gs[b]; }
};

template <bool FULLVIS, bool HS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GVPM_VPM_MINW))) void vpm_eval_kernel(GatherArgs a, VpmSplit sp) {
  __shared__ VpmEvalLds s;
  const int lane = threadIdx.x;
  // persistent: the launch is a few waves per SIMD (a wave per group of chunks was 39 k waves at C1, two thirds of them
  // empty: the dispatch alone); wave w of shard k takes the shard's groups w, w + stride, ...
  const uint32_t shard = blockIdx.x % VPM_SHARDS, gStride = (gridDim.x / VPM_SHARDS) * (uint32_t)VPM_EVAL_CHUNKS;
  const uint32_t used = min(sp.ctl[shard * 32u], sp.shardChunks);
  const float norm = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(1.f / (float)a.cfg.nb_camera_samples)));
  uint32_t rqHead = 0, rqCount = 0;
  uint32_t nNull = 0, nDiff = 0, nFail = 0;
  unsigned long long nEvalW = 0;
  auto drain = [&](uint32_t n) __attribute__((always_inline)) {
    vpmWaveSync();
    if ((uint32_t)lane < n) {
      const uint2 e = s.rq[(rqHead + lane) % VRQ];
      vpmPhase2<FULLVIS, HS>(a, s, e.x, e.y, norm, nDiff, nFail, 0u);
    }
    rqHead = (rqHead + n) % VRQ;
    rqCount -= n;
    vpmWaveSync();
  };
#ifdef GVPM_VPM_TIMING2
  unsigned long long pt[5] = {0, 0, 0, 0, 0};  // load, phase 1, phase 2, flush, batches
  const unsigned long long ptStart = wall_clock64();
#define PTICK() wall_clock64()
#else
#define PTICK() 0ull
#endif
  for (uint32_t c0 = (blockIdx.x / VPM_SHARDS) * (uint32_t)VPM_EVAL_CHUNKS; c0 < used; c0 += gStride) {
  const uint32_t chunk0 = shard * sp.shardChunks + c0;
  uint32_t cnt = 0;
  if (lane < VPM_EVAL_CHUNKS && c0 + (uint32_t)lane < used) {
    const uint2 m = sp.chunkMeta[chunk0 + (uint32_t)lane];
    cnt = sp.status[m.y] ? 0u : m.x;  // (a batch that went to the redo list: its chunks are void)
  }
  const uint32_t inc = waveScanInclDpp(cnt);
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
  if (total == 0u) continue;
  vpmWaveSync();  // (the group before has read its offsets)
  if (lane <= VPM_EVAL_CHUNKS) s.segOff[lane] = inc - cnt;  // (lane VPM_EVAL_CHUNKS: the total)
  for (uint32_t j0 = 0; j0 < total; j0 += 64u) {
    vpmWaveSync();  // (the batch before has read its state and flushed its sums)
    [[maybe_unused]] const unsigned long long p0 = PTICK();
    const uint32_t j = j0 + (uint32_t)lane;
    const bool valid = j < total;
    uint2 pr = make_uint2(0u, 0u);
    VpmSampleState st = {};
    if (valid) {
      uint32_t sl = 0;
#pragma unroll
      for (int stp = VPM_EVAL_CHUNKS / 2; stp > 0; stp >>= 1)
        if (s.segOff[sl + stp] <= j) sl += stp;
      pr = sp.pairs[(size_t)(chunk0 + sl) * 64u + (j - s.segOff[sl])];
      st = sp.state[pr.y];
    }
    const uint32_t pv = valid ? st.pix : 0xFFFFFFFFu;
    s.set[lane] = valid ? st.set : 0xFFFFFFFFu;
    s.t[lane] = st.t;
    s.pdfBase[lane] = st.pdfBase;
    s.pdfSel[lane] = st.pdfSel;
    s.trBase[lane] = st.trBase;
    s.radius[lane] = st.radius;
    s.pix[lane] = pv;
    s.edge[lane] = st.edge;
    s.gs[lane] = pr.y;
    // pixel runs of the batch (the pairs come in sample order within a slot: a pixel's pairs are adjacent); the 32 columns are
    // dealt out evenly: the largest power of two of copies that gives every run its own
    const uint32_t prev = __shfl_up(pv, 1u, 64);
    const bool head = valid && (lane == 0 || prev != pv);
    const unsigned long long heads = __ballot(head);
    const uint32_t nRuns = min((uint32_t)__popcll(heads), 32u);
    uint32_t cp = 32u;
    while (cp * nRuns > 32u) cp >>= 1;  // (wave-uniform: scalar)
    const uint32_t run = (uint32_t)__popcll(heads & ((2ull << lane) - 1ull)) - 1u;
    s.run[lane] = valid && run < 32u ? run * cp : 0xFFFFFFFFu;
    if (head && run < 32u) s.runPix[run] = pv;
    if (lane == 0) s.subMask = cp - 1u;
    for (int idx = lane; idx < 27 * 32; idx += 64) (&s.acc[0][0])[idx] = 0.0;
    vpmWaveSync();
    [[maybe_unused]] const unsigned long long p1 = PTICK();
    // phase 1, its reconnections queued; all of them drained before the batch's sums are flushed
    {
      uint32_t qMask = 0u;
      if (valid) qMask = vpmPhase1<HS>(a, s, pr.x, (uint32_t)lane, norm, nNull, nFail);
      nEvalW += (unsigned long long)__popcll(__ballot(valid));
#pragma unroll 1
      for (uint32_t i = 0; i < 4u; ++i) {
        const bool want = (qMask >> i) & 1u;
        const unsigned long long m = __ballot(want);
        if (m) {
          if (want) {
            const uint32_t off = __popcll(m & ((1ull << lane) - 1ull));
            s.rq[(rqHead + rqCount + off) % VRQ] = make_uint2(pr.x, (uint32_t)lane | (i << 8) | (((qMask >> (4u + i)) & 1u) << 16));
          }
          rqCount += __popcll(m);
          if (rqCount >= 64u) drain(64u);
        }
      }
    }
    [[maybe_unused]] const unsigned long long p2 = PTICK();
    if (rqCount) drain(rqCount);
    vpmWaveSync();
    [[maybe_unused]] const unsigned long long p3 = PTICK();
    for (uint32_t idx = (uint32_t)lane; idx < 27u * nRuns; idx += 64u) {
      const uint32_t k = idx / nRuns, rr = idx % nRuns;
      double vd = 0.0;
      for (uint32_t c = 0; c < cp; ++c) vd += s.acc[k][rr * cp + c];
      const float v = (float)vd;
      if (v != 0.f) {
        const uint32_t pw = s.runPix[rr];
        atomicAdd(&a.iter[((size_t)(pw >> 16) * a.cfg.width + (pw & 0xFFFFu)) * 27 + k], v);
      }
    }
#ifdef GVPM_VPM_TIMING2
    pt[0] += p1 - p0;
    pt[1] += p2 - p1;
    pt[2] += p3 - p2;
    pt[3] += wall_clock64() - p3;
    pt[4] += 1;
#endif
  }
  }
#ifdef GVPM_VPM_TIMING2
  if (lane == 0 && blockIdx.x < 32768u) {
    unsigned long long *row = gvpmVpmLog + 4 * (32768u + blockIdx.x);  // (the upper half of the log: the walk's rows are below)
    row[0] = ptStart;
    row[1] = wall_clock64();
    row[2] = (pt[0] & 0xFFFFFFFFull) | (pt[1] << 32);
    row[3] = (pt[2] & 0xFFFFFull) | ((pt[3] & 0xFFFFFull) << 20) | (pt[4] << 40);
  }
#endif
  {
    unsigned long long nu = nNull, di = nDiff, fa = nFail;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      nu += __shfl_xor(nu, o, 64);
      di += __shfl_xor(di, o, 64);
      fa += __shfl_xor(fa, o, 64);
    }
    if (lane == 0 && nEvalW) {
      atomicAdd(&statRow(a)[0], nEvalW);
      atomicAdd(&statRow(a)[2], nu);
      atomicAdd(&statRow(a)[3], di);
      atomicAdd(&statRow(a)[4], fa);
    }
  }
}

// The end of a G-VPM iteration in ONE launch (two until round 6: 6 + 14 us for 65 k pixels -- the second one a thousand waves'
// atomicMax on one word, ~11 ns apart):
//   accum += iter, and iter handed back zeroed (every thread: one of the P * 27 sums);
//   SPPM statistics, gvpm.cpp:1191-1195 (the threads below P: their pixel) + the largest scale for the next grid, reduced in the
//   workgroup first; mvol is handed back zeroed: the next gather adds into it without a memset before it.
// maxScaleBits is zero when the kernel starts: the gather that read it (through the host) lies before, gatherVPM clears it.
__global__ __launch_bounds__(256) void vpm_finish_kernel(float *__restrict__ accum, float *__restrict__ iter, size_t n27,
                                                         float *scaleVol, float *nVol, float *mvol, size_t n, float alpha,
                                                         uint32_t *maxScaleBits) {
  __shared__ float wmax[4];
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n27) {
    accum[i] += iter[i];
    iter[i] = 0.f;
  }
  if ((size_t)blockIdx.x * blockDim.x >= n) return;  // (workgroup-uniform: no pixel below this workgroup)
  float sc = 0.f;
  if (i < n) {
    sc = scaleVol[i];
    const float M = mvol[i], N = nVol[i];
    mvol[i] = 0.f;
    if (M + N != 0.f) {
      const float ratio = (N + alpha * M) / (N + M);
      sc = sc * cbrtf(ratio);
      scaleVol[i] = sc;
      nVol[i] = N + alpha * M;
    }
  }
  sc = wave_max(sc);
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = sc;
  __syncthreads();
  if (threadIdx.x == 0) {
    sc = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    // positive floats order as uints; a workgroup first looks whether it would raise the maximum
    if (__float_as_uint(sc) > __hip_atomic_load(maxScaleBits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxScaleBits, __float_as_uint(sc));
  }
}

// accum += iter, and iter handed back zeroed; zeroWord: cleared with it (the other techniques' fold)
__global__ __launch_bounds__(256) void accumulate_kernel(float *__restrict__ accum, float *__restrict__ iter, size_t n,
                                                         uint32_t *zeroWord) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    accum[i] += iter[i];
    iter[i] = 0.f;
  }
  if (i == 0 && zeroWord) *zeroWord = 0u;
}

void launch_gather_vpm(const GatherArgs &a, bool fullVis, bool primal, hipStream_t stream) {
  if (a.nsamples == 0) return;
  const uint32_t nwaves = max((a.nsamples + VPM_SPW - 1u) / VPM_SPW, a.vpmOrderN);
  const dim3 grid((nwaves + (uint32_t)VPM_WPB - 1u) / (uint32_t)VPM_WPB);
  if (primal) {
    // the sppm integrator's point estimate: the walk and the rings of the gradient kernel, the primal term only
    hipLaunchKernelGGL((gather_vpm_kernel<false, false, true>), grid, dim3(64 * VPM_WPB), 0, stream, a);
  } else if (a.reqHost) {
    // manifold-typed shifts go to the host's request list (an instantiation of its own: the default one keeps its registers)
    if (fullVis) hipLaunchKernelGGL((gather_vpm_kernel<true, true>), grid, dim3(64 * VPM_WPB), 0, stream, a);
    else hipLaunchKernelGGL((gather_vpm_kernel<false, true>), grid, dim3(64 * VPM_WPB), 0, stream, a);
  } else if (fullVis) {
    hipLaunchKernelGGL((gather_vpm_kernel<true, false>), grid, dim3(64 * VPM_WPB), 0, stream, a);
  } else {
    hipLaunchKernelGGL((gather_vpm_kernel<false, false>), grid, dim3(64 * VPM_WPB), 0, stream, a);
  }
}

void launch_vpm_find(const GatherArgs &a, const VpmSplit &sp, hipStream_t stream) {
  if (a.nsamples == 0) return;
  const uint32_t nwaves = max((a.nsamples + VPM_SPW - 1u) / VPM_SPW, a.vpmOrderN);
  hipLaunchKernelGGL(vpm_find_kernel, dim3((nwaves + VPM_FIND_WPB - 1u) / VPM_FIND_WPB), dim3(64 * VPM_FIND_WPB), 0, stream, a, sp);
}
void launch_vpm_redo(const GatherArgs &a, const VpmSplit &sp, bool fullVis, uint32_t nwaves, hipStream_t stream) {
  if (a.nsamples == 0) return;
  if (a.reqHost) {
    if (fullVis) hipLaunchKernelGGL((vpm_redo_kernel<true, true>), dim3(nwaves), dim3(64), 0, stream, a, sp);
    else hipLaunchKernelGGL((vpm_redo_kernel<false, true>), dim3(nwaves), dim3(64), 0, stream, a, sp);
  } else if (fullVis) {
    hipLaunchKernelGGL((vpm_redo_kernel<true, false>), dim3(nwaves), dim3(64), 0, stream, a, sp);
  } else {
    hipLaunchKernelGGL((vpm_redo_kernel<false, false>), dim3(nwaves), dim3(64), 0, stream, a, sp);
  }
}
void launch_vpm_eval(const GatherArgs &a, const VpmSplit &sp, bool fullVis, uint32_t wavesPerShard, hipStream_t stream) {
  if (a.nsamples == 0) return;
  const uint32_t nwaves = VPM_SHARDS * std::min(wavesPerShard, (sp.shardChunks + (uint32_t)VPM_EVAL_CHUNKS - 1u) / (uint32_t)VPM_EVAL_CHUNKS);
  if (a.reqHost) {
    if (fullVis) hipLaunchKernelGGL((vpm_eval_kernel<true, true>), dim3(nwaves), dim3(64), 0, stream, a, sp);
    else hipLaunchKernelGGL((vpm_eval_kernel<false, true>), dim3(nwaves), dim3(64), 0, stream, a, sp);
  } else if (fullVis) {
    hipLaunchKernelGGL((vpm_eval_kernel<true, false>), dim3(nwaves), dim3(64), 0, stream, a, sp);
  } else {
    hipLaunchKernelGGL((vpm_eval_kernel<false, false>), dim3(nwaves), dim3(64), 0, stream, a, sp);
  }
}

void launch_vpm_finish(float *accum, float *iter, float *scaleVol, float *nVol, float *mvol, size_t n, float alpha,
                       uint32_t *maxScaleBits, hipStream_t stream) {
  const size_t n27 = n * 27;
  hipLaunchKernelGGL(vpm_finish_kernel, dim3((unsigned)((n27 + 255) / 256)), dim3(256), 0, stream, accum, iter, n27, scaleVol, nVol, mvol,
                     n, alpha, maxScaleBits);
}

void launch_accumulate(float *accum, float *iter, size_t n, uint32_t *zeroWord, hipStream_t stream) {
  hipLaunchKernelGGL(accumulate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, accum, iter, n, zeroWord);
}

}  // namespace gvpm
