// G-Planes (photon planes, 0D kernel) gather + gradient-domain shift for gfx950, hand-written HIP.
//
// Replaces, for one SPPM iteration, the body of
//   GPMIntegrator::computeVolumeGradientPlanes      gvpm/gvpm.cpp:782-878
//   PhotonPlaneBVH (build + query)                  pm/plane_accel.h:85-207
//   PhotonPlane::intersectPlane0D / getContrib0D    pm/plane_struct.h:104-192
//   PlaneGradRadianceQuery::operator()              gvpm/shift/shift_volume_planes.h:57-101
//   specularShift / intersection                    shift_volume_planes.h:263-416, 426-453
// (pm/ = src/integrators/photonmapper/).
//
// A photon plane is a parallelogram {ori, e0 = w0 len0, e1 = w1 len1}; the camera ray collects a
// contribution wherever it pierces one (no kernel radius, no blur).  One 64-lane wave owns a tile
// of 64 camera-beam sets (8x8 pixels): every lane keeps ITS ray in registers and the wave streams
// the plane records through the scalar unit (wave-uniform s_load of a 48-byte test record), so the
// pierce test is pure VALU work with no LDS or vector-memory traffic.  Hits are compacted with
// __ballot into an LDS queue and evaluated 64 at a time.
//
// Precision: the fp32 pierce test is conservative (error band from the operand magnitudes); the
// lane that evaluates a queued pair repeats the reference's test in uncontracted fp64 (with its
// float det / float reciprocal) and drops the pair if that fails, so the set of evaluated pairs
// is the fp64 oracle's.  The evaluation itself is a literal fp64 transcription of the reference.
#include <hip/hip_runtime.h>

#include "device_types.h"
#include "dmath.h"
#include "shift_device.h"
#include "tile_walk.h"
#include "vec.h"

namespace gvpm {

// raw plane inputs: the photon-beam SoA of gvpm_upload_beams + the second edge of every plane
struct PlaneArgs {
  const float4 *test;        // 3 float4 per plane: {ori, |e0|} {e0, |e1|} {e1, -}
  const float *ori, *end;    // 3 floats per plane (beam origin / end vertex)
  const float *flux;         // 3 floats
  const float *w1, *len1;    // 3 floats / 1 float
  const uint32_t *flags;     // GVPM_PF_* (depth = edgeID)
  uint32_t nplanes;
  uint32_t planesPerItem;
};

struct PlaneD {
  d3 ori, w0, w1, flux;
  double len0, len1;
  int edgeID;
};

struct PlaneIts {
  double tCam, t0, t1, invDet;
};

// the tile's rays + the per-lane hit rings: 19 KB a wave (8 waves per CU, what the registers allow; the generic tile structure with its stage
// and accumulators was 30 KB)
template <int B> struct PlaneLds : RayTile<B> {
  uint32_t hitQ[8][64];
  uint32_t cull[64];  // planes of the current batch of 64 that the tile's frustum can reach
};

__device__ __forceinline__ PlaneD loadPlane(const PlaneArgs &pa, uint32_t i) {
  PlaneD p;
  p.ori = mkd(pa.ori[3 * (size_t)i], pa.ori[3 * (size_t)i + 1], pa.ori[3 * (size_t)i + 2]);
  const d3 e = mkd(pa.end[3 * (size_t)i], pa.end[3 * (size_t)i + 1], pa.end[3 * (size_t)i + 2]) - p.ori;
  p.len0 = sqrt(dot(e, e));
  p.w0 = e * (1.0 / p.len0);
  {
    // Vec3 / scalar in the reference multiplies by the reciprocal
  }
  p.w1 = mkd(pa.w1[3 * (size_t)i], pa.w1[3 * (size_t)i + 1], pa.w1[3 * (size_t)i + 2]);
  p.len1 = (double)pa.len1[i];
  p.flux = mkd(pa.flux[3 * (size_t)i], pa.flux[3 * (size_t)i + 1], pa.flux[3 * (size_t)i + 2]);
  p.edgeID = (int)GVPM_PF_DEPTH(pa.flags[i]);
  return p;
}

// PhotonPlane::intersectPlane0D, pm/plane_struct.h:104-135 -- float det and float reciprocal as written
__device__ __forceinline__ bool intersectPlane0D(const PlaneD &pl, const RayD &ray, PlaneIts &r) {
#pragma clang fp contract(off)
  const d3 e0 = pl.w0 * pl.len0;
  const d3 e1 = pl.w1 * pl.len1;
  const d3 P = crossd(ray.d, e1);
  const float det = (float)dot(e0, P);
  if (fabsf(det) < 1e-5f) return false;
  r.invDet = (double)(float)(1.0 / (double)det);
  const d3 T = ray.o - pl.ori;
  r.t0 = dot(T, P) * r.invDet;
  if (r.t0 < 0.0 || r.t0 > 1.0) return false;
  const d3 Q = crossd(T, e0);
  r.t1 = dot(ray.d, Q) * r.invDet;
  if (r.t1 < 0.0 || r.t1 > 1.0) return false;
  r.tCam = dot(e1, Q) * r.invDet;
  if (r.tCam <= ray.mint || r.tCam >= ray.maxt) return false;
  r.t1 *= pl.len1;
  r.t0 *= pl.len0;
  return true;
}

// intersection(), shift_volume_planes.h:426-453
__device__ __forceinline__ bool intersectionUnit(const RayD &ray, d3 ori, d3 w0, d3 w1, double &tCam, double &t0,
                                                 double &t1) {
#pragma clang fp contract(off)
  const d3 P = crossd(ray.d, w1);
  const double det = dot(w0, P);
  if (fabs(det) < (double)1e-8f) return false;
  const double invDet = 1.0 / det;
  const d3 T = ray.o - ori;
  t0 = dot(T, P) * invDet;
  if (t0 < 0.0) return false;
  const d3 Q = crossd(T, w0);
  t1 = dot(ray.d, Q) * invDet;
  if (t1 < 0.0) return false;
  tCam = dot(w1, Q) * invDet;
  return !(tCam <= ray.mint || tCam >= ray.maxt);
}

__device__ __forceinline__ double invJacobian(const PlaneD &pl, d3 k) { return 1.0 / fabs(dot(pl.w0, crossd(pl.w1, k))); }

// One queued (plane, camera ray) pair: PlaneGradRadianceQuery::operator().  True when the pair
// produced a contribution (an evaluation).
struct MRecP {
  float tr, pdfSuccess, pdfFailure;
};
__device__ __forceinline__ MRecP mediumEvalP(const MediumDev &m, float dist) {
  MRecP r;
  float e = __expf(-m.sigmaT[0] * dist);
  r.pdfSuccess = m.sigmaT[0] * e * m.msw;
  r.pdfFailure = e * m.msw + (1.f - m.msw);
  if (e < 1e-20f) e = 0.f;
  r.tr = e;
  return r;
}

// The lane evaluates hits of ITS OWN ray (bIdx = lane): the 27 sums stay in registers (acc) -- no LDS atomics, whose
// ~1 lane per clock and CU had been a third of this kernel -- and the ray columns it reads are its own.
template <int B>
__device__ __forceinline__ bool evaluatePlane(const GatherArgs &a, const PlaneArgs &pa, PlaneLds<B> &s, uint32_t planeIdx,
                                              uint32_t bIdx, float (&acc)[27], uint32_t &nDiff, uint32_t &nFail) {
  const PlaneD pl = loadPlane(pa, planeIdx);
  const RayReg base = loadRay(s, 0, bIdx);
  const uint32_t edge = s.edge[bIdx];
  const double eps = (double)a.cfg.epsilon;
  const RayD cam{tod(base.o), tod(base.d), eps, (double)base.len - eps};
  PlaneIts bRec;
  if (!intersectPlane0D(pl, cam, bRec)) return false;
  // From here on the RADIOMETRY is fp32 (transmittances, phase functions, pdf and Jacobian ratios, MIS weights: ~56
  // divisions and square roots per pair in the fp64 transcription, ~28 instructions each): operands are O(1) and the
  // results go to fp32 accumulators anyway.  The geometry feeding it -- the base pierce point, the rotated plane, the
  // shifted pierce point and its acceptance tests, the Jacobian determinants -- stays fp64.
  const float g = a.med.g;
  const float tCamF = (float)bRec.tCam, t0F = (float)bRec.t0, t1F = (float)bRec.t1;
  // getContrib0D, pm/plane_struct.h:150-192
  const MRecP mCam = mediumEvalP(a.med, tCamF), m0 = mediumEvalP(a.med, t0F), m1 = mediumEvalP(a.med, t1F);
  const f3 w1F = tof(pl.w1);
  const float pBase = phaseEval(g, -w1F, -base.d);
  const float jBase = (float)fabs(dot(pl.w0, crossd(pl.w1, cam.d)));
  const float invJBase = frcp(jBase);
  const f3 sig2 = mk3(a.med.sigmaS[0] * a.med.sigmaS[0], a.med.sigmaS[1] * a.med.sigmaS[1], a.med.sigmaS[2] * a.med.sigmaS[2]);
  const f3 baseContrib = sig2 * tof(pl.flux) * (fdiv(mCam.tr * pBase * m1.tr * m0.tr, m0.pdfFailure * m1.pdfFailure) * invJBase);
  acc[0] += baseContrib.x;
  acc[1] += baseContrib.y;
  acc[2] += baseContrib.z;
  const double w0Dot = dot(pl.w0, pl.w1);
  const double sinW = sqrt(1.0 - w0Dot * w0Dot);
  const float invM0 = frcp(m0.tr), invM1 = frcp(m1.tr);
#pragma unroll 1
  for (int i = 0; i < 4; ++i) {
    const RayReg sh = loadRay(s, 1 + i, bIdx);
    float w = 1.f;
    f3 sflux = mk3(0.f);
    if (sh.valid) {
      // specularShift (BETTERSHIFT 0), shift_volume_planes.h:263-416
      const RayD shiftRay{tod(sh.o), tod(sh.d), eps, (double)sh.len};
      const d3 newIts = at(shiftRay, bRec.tCam);
      d3 orth = newIts - (pl.ori + pl.w0 * dot(newIts - pl.ori, pl.w0));
      orth = orth * (1.0 / sqrt(dot(orth, orth)));
      const d3 newW1 = orth * sinW + pl.w0 * w0Dot;
      double tCamNew, t0New, t1New;
      if (!intersectionUnit(shiftRay, pl.ori, pl.w0, newW1, tCamNew, t0New, t1New)) {
        nFail++;
      } else {
        const float t0N = (float)t0New, t1N = (float)t1New;
        const MRecP m1s = mediumEvalP(a.med, t1N), m0s = mediumEvalP(a.med, t0N);
        const float jShift = (float)fabs(dot(pl.w0, crossd(newW1, shiftRay.d)));
        float f = (m0s.tr * invM0) * (m1s.tr * invM1);
        f = fdiv(f * jBase, jShift);
        float jac = invJBase * jShift;
        jac = fdiv(jac * t1F, t1N);
        if (pl.edgeID != 1) jac = fdiv(jac * t0F, t0N);
        const float pNew = phaseEval(g, -tof(newW1), -sh.d);
        f = fdiv(f * pNew, pBase);
        w = 0.5f;
        sflux = baseContrib * (f * jac);
        nDiff++;
        if (a.cfg.use_mis) {
          const float basePdf = m0.pdfSuccess * m1.pdfSuccess * pBase;
          const float offsetPdf = m0s.pdfSuccess * m1s.pdfSuccess * pNew;
          if (offsetPdf == 0.f || basePdf == 0.f) w = 1.f;
          else w = frcp(1.f + sensorMIS(sh, base, edge) * jac * fdiv(offsetPdf, basePdf));
        }
      }
    }
    // (register arrays need constant indices: the shift's slots are selected by a mask)
    const f3 sw = sflux * w, bw = baseContrib * w;
#pragma unroll
    for (int ii = 0; ii < 4; ++ii) {
      const float m = ii == i ? 1.f : 0.f;
      acc[3 + 3 * ii + 0] += m * sw.x;
      acc[3 + 3 * ii + 1] += m * sw.y;
      acc[3 + 3 * ii + 2] += m * sw.z;
      acc[15 + 3 * ii + 0] += m * bw.x;
      acc[15 + 3 * ii + 1] += m * bw.y;
      acc[15 + 3 * ii + 2] += m * bw.z;
    }
  }
  return true;
}

// grid: x = image tile, y = plane chunk
__global__ __launch_bounds__(64, 2) void gather_planes_kernel(GatherArgs a, PlaneArgs pa) {
  constexpr int B = 64;
  __shared__ PlaneLds<B> s;
  const int lane = threadIdx.x;
  const uint32_t tile = blockIdx.x;
  const uint32_t p0 = blockIdx.y * pa.planesPerItem;
  const uint32_t p1 = min(pa.nplanes, p0 + pa.planesPerItem);
  const uint32_t tileBeg = a.tileStart[tile], tileEnd = a.tileStart[tile + 1];
  const float eps = a.cfg.epsilon;
  uint32_t nEval = 0, nDiff = 0, nFail = 0;
  unsigned long long nCand = 0;

  for (uint32_t setBase = tileBeg; setBase < tileEnd; setBase += B) {
    const uint32_t nb = min((uint32_t)B, tileEnd - setBase);
    __syncthreads();
    loadTileRays<B>(a, s, setBase, nb, lane);
    __syncthreads();
    const RayReg base = loadRay(s, 0, lane);
    const bool rayValid = (uint32_t)lane < nb && base.valid;
    if (!__ballot(rayValid)) continue;
    const float mint = eps, maxt = base.len - eps;
    const f3 o = base.o, d = base.d;
    // per-lane hit queues (ring of QD plane indices in this lane's LDS column): a lane only evaluates hits of its
    // own ray; the wave evaluates when most lanes have one pending, or a queue is about to fill
    constexpr uint32_t QD = 8;
    uint32_t *hitQ = &s.hitQ[0][0];  // [QD][64]
    uint32_t qHead = 0, qCount = 0;
    float acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = 0.f;
    // ---- tile-frustum cull ----------------------------------------------------------------------------------
    // With a common ray origin O (a pinhole sensor's first edge) the quantities of the pierce test are linear in the
    // ray direction:  det = d.(e1 x e0),  u det = d.(e1 x T),  v det = d.(T x e0),  t det = e1.(T x e0),  T = O - ori.
    // Over the componentwise bounds [dlo, dhi] of the tile's 64 directions each has an interval; a plane whose
    // intervals put u, v or t outside its acceptance range for EVERY direction of the box (with the test's own error
    // band as slack) cannot be pierced by any ray of the tile.  One lane culls one plane, 64 planes a batch; the
    // survivors (a few per cent: a ray pierces ~5 % of the planes) go through the per-ray test below.
    const unsigned long long validMask = __ballot(rayValid);
    const int firstValid = __ffsll((long long)validMask) - 1;
    const f3 o0 = mk3(__shfl(o.x, firstValid, 64), __shfl(o.y, firstValid, 64), __shfl(o.z, firstValid, 64));
    const bool cullOk = !__ballot(rayValid && (o.x != o0.x || o.y != o0.y || o.z != o0.z));
    const f3 dlo = mk3(wave_min(rayValid ? d.x : INFINITY), wave_min(rayValid ? d.y : INFINITY), wave_min(rayValid ? d.z : INFINITY));
    const f3 dhi = mk3(wave_max(rayValid ? d.x : -INFINITY), wave_max(rayValid ? d.y : -INFINITY), wave_max(rayValid ? d.z : -INFINITY));
    const float maxtAll = wave_max(rayValid ? maxt : -INFINITY);
    auto range = [&](f3 A, float &lo, float &hi) {
      lo = fminf(A.x * dlo.x, A.x * dhi.x) + fminf(A.y * dlo.y, A.y * dhi.y) + fminf(A.z * dlo.z, A.z * dhi.z);
      hi = fmaxf(A.x * dlo.x, A.x * dhi.x) + fmaxf(A.y * dlo.y, A.y * dhi.y) + fmaxf(A.z * dlo.z, A.z * dhi.z);
    };
    constexpr uint32_t G = 4;
    for (uint32_t pb0 = p0; pb0 < p1; pb0 += 64u) {
      const uint32_t pmine = pb0 + (uint32_t)lane;
      bool keep = pmine < p1;
      if (keep && cullOk) {
        const float4 r0 = pa.test[3 * (size_t)pmine + 0], r1 = pa.test[3 * (size_t)pmine + 1], r2 = pa.test[3 * (size_t)pmine + 2];
        const f3 e0 = mk3(r1.x, r1.y, r1.z), e1 = mk3(r2.x, r2.y, r2.z);
        const float n0 = r0.w, n1 = r1.w;
        const f3 T = o0 - mk3(r0.x, r0.y, r0.z);
        const float nT = fabsf(T.x) + fabsf(T.y) + fabsf(T.z);
        const float K = 8e-6f;  // twice the per-ray test's band
        const float eD = K * n0 * n1, eU = K * nT * n1 + eD, eV = K * nT * n0 + eD, eC = K * nT * n0 * n1;
        const f3 AD = cross(e1, e0), AU = cross(e1, T), AV = cross(T, e0);
        const float Ct = dot(e1, AV);
        float Dlo, Dhi, Ulo, Uhi, Vlo, Vhi, UDlo, UDhi, VDlo, VDhi;
        range(AD, Dlo, Dhi);
        range(AU, Ulo, Uhi);
        range(AV, Vlo, Vhi);
        range(AU - AD, UDlo, UDhi);
        range(AV - AD, VDlo, VDhi);
        if (Dlo - eD > 0.f) {
          // det > 0 for every ray of the tile
          if (Uhi + eU < 0.f || UDlo - eU - eD > 0.f || Vhi + eV < 0.f || VDlo - eV - eD > 0.f) keep = false;
          if (Ct + eC < mint * (Dlo - eD) || Ct - eC > maxtAll * (Dhi + eD)) keep = false;
        } else if (Dhi + eD < 0.f) {
          if (Ulo - eU > 0.f || UDhi + eU + eD < 0.f || Vlo - eV > 0.f || VDhi + eV + eD < 0.f) keep = false;
          if (-Ct + eC < mint * (-Dhi - eD) || -Ct - eC > maxtAll * (-Dlo + eD)) keep = false;
        }
      }
      const unsigned long long km = __ballot(keep);
      const uint32_t nk = (uint32_t)__popcll(km);
      __syncthreads();
      if (keep) s.cull[__popcll(km & ((1ull << lane) - 1ull))] = pmine;
      __syncthreads();
      nCand += (unsigned long long)__popcll(validMask) * nk;  // per-ray pierce tests actually made
      // ---- per-ray pierce test of the survivors, four planes per step (twelve scalar loads and four test chains
      // in flight)
      for (uint32_t kb = 0; kb < nk; kb += G) {
        uint32_t hm = 0;
        uint32_t pidx[G];
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) {
          const uint32_t p = (uint32_t)__builtin_amdgcn_readfirstlane((int)s.cull[min(kb + g, nk - 1u)]);
          pidx[g] = p;
          // wave-uniform record: scalar loads
          const float4 r0 = pa.test[3 * (size_t)p + 0];
          const float4 r1 = pa.test[3 * (size_t)p + 1];
          const float4 r2 = pa.test[3 * (size_t)p + 2];
          const f3 e0 = mk3(r1.x, r1.y, r1.z), e1 = mk3(r2.x, r2.y, r2.z);
          const float n0 = r0.w, n1 = r1.w;
          const f3 T = o - mk3(r0.x, r0.y, r0.z);
          const f3 P = cross(d, e1);
          const float det = dot(e0, P);
          const float u = dot(T, P);
          const f3 Q = cross(T, e0);
          const float v = dot(d, Q);
          const float c = dot(e1, Q);
          const float nT = fabsf(T.x) + fabsf(T.y) + fabsf(T.z);  // >= |T|: the band only has to be conservative
          // conservative acceptance: every quantity carries a relative error <= K of its magnitude bound
          const float K = 4e-6f;
          const float D = fabsf(det);
          const float sgn = det < 0.f ? -1.f : 1.f;
          const float eD = K * n0 * n1;
          const float us = u * sgn, vs = v * sgn, cs = c * sgn;
          const float eU = K * nT * n1 + eD, eV = K * nT * n0 + eD, eC = K * nT * n0 * n1;
          bool hit = rayValid && kb + g < nk && D + eD >= 0.99999e-5f;
          hit = hit && us >= -eU && us <= D + eU && vs >= -eV && vs <= D + eV;
          hit = hit && cs > mint * D - eC - mint * eD && cs < maxt * D + eC + maxt * eD;
          hm |= hit ? (1u << g) : 0u;
        }
        // push this lane's hits (0 - 4), then evaluate while most lanes have one pending or a ring could fill
#pragma unroll
        for (uint32_t g = 0; g < G; ++g) {
          if ((hm >> g) & 1u) {
            hitQ[((qHead + qCount) % QD) * 64u + (uint32_t)lane] = pidx[g];
            qCount++;
          }
        }
        for (;;) {
          const unsigned long long pending = __ballot(qCount > 0u);
          if (!(__popcll(pending) >= 48 || __ballot(qCount > QD - G))) break;
          if (qCount > 0u) {
            if (evaluatePlane<B>(a, pa, s, hitQ[qHead * 64u + (uint32_t)lane], (uint32_t)lane, acc, nDiff, nFail)) nEval++;
            qHead = (qHead + 1u) % QD;
            qCount--;
          }
        }
      }
    }
    while (__ballot(qCount > 0u)) {
      if (qCount > 0u) {
        if (evaluatePlane<B>(a, pa, s, hitQ[qHead * 64u + (uint32_t)lane], (uint32_t)lane, acc, nDiff, nFail)) nEval++;
        qHead = (qHead + 1u) % QD;
        qCount--;
      }
    }
    // this lane's 27 sums -> the film (other plane chunks add to the same pixel)
    if ((uint32_t)lane < nb) {
      const uint32_t pv = s.pix[lane];
      const size_t pix = (size_t)(pv >> 16) * a.cfg.width + (pv & 0xFFFFu);
#pragma unroll
      for (int k = 0; k < 27; ++k)
        if (acc[k] != 0.f) atomicAdd(&a.iter[pix * 27 + k], acc[k]);
    }
  }
  {
    unsigned long long ev = nEval, di = nDiff, fa = nFail;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      ev += __shfl_xor(ev, off, 64);
      di += __shfl_xor(di, off, 64);
      fa += __shfl_xor(fa, off, 64);
    }
    if (lane == 0 && (ev | nCand)) {
      atomicAdd(&statRow(a)[0], ev);
      atomicAdd(&statRow(a)[1], nCand);
      atomicAdd(&statRow(a)[3], di);
      atomicAdd(&statRow(a)[4], fa);
    }
  }
}

// test records {ori, |e0|} {e0, |e1|} {e1, -} of every plane
__global__ __launch_bounds__(256) void plane_records_kernel(PlaneArgs pa, float4 *out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= pa.nplanes) return;
  const f3 ori = mk3(pa.ori[3 * (size_t)i], pa.ori[3 * (size_t)i + 1], pa.ori[3 * (size_t)i + 2]);
  const f3 e0 = mk3(pa.end[3 * (size_t)i], pa.end[3 * (size_t)i + 1], pa.end[3 * (size_t)i + 2]) - ori;
  const float l1 = pa.len1[i];
  const f3 e1 = mk3(pa.w1[3 * (size_t)i], pa.w1[3 * (size_t)i + 1], pa.w1[3 * (size_t)i + 2]) * l1;
  out[3 * (size_t)i + 0] = make_float4(ori.x, ori.y, ori.z, sqrtf(dot(e0, e0)) * 1.000001f);
  out[3 * (size_t)i + 1] = make_float4(e0.x, e0.y, e0.z, sqrtf(dot(e1, e1)) * 1.000001f);
  out[3 * (size_t)i + 2] = make_float4(e1.x, e1.y, e1.z, 0.f);
}

void launch_plane_records(const PlaneArgs &pa, float4 *out, hipStream_t stream) {
  if (pa.nplanes == 0) return;
  hipLaunchKernelGGL(plane_records_kernel, dim3((pa.nplanes + 255) / 256), dim3(256), 0, stream, pa, out);
}

void launch_gather_planes(const GatherArgs &a, const PlaneArgs &pa, uint32_t ntiles, uint32_t nchunks,
                          hipStream_t stream) {
  if (a.nsets == 0 || pa.nplanes == 0 || ntiles == 0) return;
  hipLaunchKernelGGL(gather_planes_kernel, dim3(ntiles, nchunks), dim3(64), 0, stream, a, pa);
}

}  // namespace gvpm
