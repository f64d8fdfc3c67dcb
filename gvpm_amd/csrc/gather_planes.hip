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
template <int B>
__device__ __forceinline__ bool evaluatePlane(const GatherArgs &a, const PlaneArgs &pa, TileLds<B> &s, uint32_t planeIdx,
                                              uint32_t bIdx, uint32_t &nDiff, uint32_t &nFail) {
  const PlaneD pl = loadPlane(pa, planeIdx);
  const RayReg base = loadRay(s, 0, bIdx);
  const uint32_t edge = s.edge[bIdx];
  const double eps = (double)a.cfg.epsilon;
  const RayD cam{tod(base.o), tod(base.d), eps, (double)base.len - eps};
  PlaneIts bRec;
  if (!intersectPlane0D(pl, cam, bRec)) return false;
  const double g = (double)a.med.g;
  // getContrib0D, pm/plane_struct.h:150-192
  const MRecD mCam = mediumEvalD(a.med, bRec.tCam);
  const MRecD m0 = mediumEvalD(a.med, bRec.t0);
  const MRecD m1 = mediumEvalD(a.med, bRec.t1);
  const double pBase = phaseD(g, pl.w1 * -1.0, cam.d * -1.0);
  const double invJBase = invJacobian(pl, cam.d);
  d3 baseContrib;
  {
    const double k = mCam.tr * pBase;
    baseContrib = mkd(k * (double)a.med.sigmaS[0] * (double)a.med.sigmaS[0] * pl.flux.x,
                      k * (double)a.med.sigmaS[1] * (double)a.med.sigmaS[1] * pl.flux.y,
                      k * (double)a.med.sigmaS[2] * (double)a.med.sigmaS[2] * pl.flux.z);
    baseContrib = baseContrib * (m1.tr * m0.tr);
    baseContrib = baseContrib * (1.0 / m0.pdfFailure);
    baseContrib = baseContrib * (1.0 / m1.pdfFailure);
    baseContrib = baseContrib * invJBase;
  }
  atomicAdd(&s.acc[0][bIdx], (float)baseContrib.x);
  atomicAdd(&s.acc[1][bIdx], (float)baseContrib.y);
  atomicAdd(&s.acc[2][bIdx], (float)baseContrib.z);
  const double w0Dot = dot(pl.w0, pl.w1);
  const double sinW = sqrt(1.0 - w0Dot * w0Dot);
#pragma unroll 1
  for (int i = 0; i < 4; ++i) {
    const RayReg sh = loadRay(s, 1 + i, bIdx);
    double w = 1.0;
    d3 sflux = mkd(0, 0, 0);
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
        const MRecD m1s = mediumEvalD(a.med, t1New);
        const MRecD m0s = mediumEvalD(a.med, t0New);
        const double jShift = fabs(dot(pl.w0, crossd(newW1, shiftRay.d)));
        double f = (m0s.tr * (1.0 / m0.tr)) * (m1s.tr * (1.0 / m1.tr));
        f = f / invJBase;
        f = f * (1.0 / jShift);
        double jac = invJBase * jShift;
        jac /= t1New / bRec.t1;
        if (pl.edgeID != 1) jac /= t0New / bRec.t0;
        const double pNew = phaseD(g, newW1 * -1.0, shiftRay.d * -1.0);
        f = f * pNew / pBase;
        w = 0.5;
        sflux = baseContrib * (f * jac);
        nDiff++;
        if (a.cfg.use_mis) {
          const double basePdf = m0.pdfSuccess * m1.pdfSuccess * pBase;
          const double offsetPdf = m0s.pdfSuccess * m1s.pdfSuccess * pNew;
          if (offsetPdf == 0.0 || basePdf == 0.0) {
            w = 1.0;
          } else {
            w = 1.0 / (1.0 + (double)sensorMIS(sh, base, edge) * jac * offsetPdf / basePdf);
          }
        }
      }
    }
    if (sflux.x != 0 || sflux.y != 0 || sflux.z != 0) {
      atomicAdd(&s.acc[3 + 3 * i + 0][bIdx], (float)(sflux.x * w));
      atomicAdd(&s.acc[3 + 3 * i + 1][bIdx], (float)(sflux.y * w));
      atomicAdd(&s.acc[3 + 3 * i + 2][bIdx], (float)(sflux.z * w));
    }
    atomicAdd(&s.acc[15 + 3 * i + 0][bIdx], (float)(baseContrib.x * w));
    atomicAdd(&s.acc[15 + 3 * i + 1][bIdx], (float)(baseContrib.y * w));
    atomicAdd(&s.acc[15 + 3 * i + 2][bIdx], (float)(baseContrib.z * w));
  }
  return true;
}

// grid: x = image tile, y = plane chunk
__global__ __launch_bounds__(64, 2) void gather_planes_kernel(GatherArgs a, PlaneArgs pa) {
  constexpr int B = 64;
  __shared__ TileLds<B> s;
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
    for (int idx = lane; idx < 27 * B; idx += 64) (&s.acc[0][0])[idx] = 0.f;
    __syncthreads();
    const RayReg base = loadRay(s, 0, lane);
    const bool rayValid = (uint32_t)lane < nb && base.valid;
    if (!__ballot(rayValid)) continue;
    const float mint = eps, maxt = base.len - eps;
    const f3 o = base.o, d = base.d;
    uint32_t qHead = 0, qCount = 0;
    nCand += (unsigned long long)__popcll(__ballot(rayValid)) * (p1 - p0);
    for (uint32_t p = p0; p < p1; ++p) {
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
      const float nT = sqrtf(dot(T, T));
      // conservative acceptance: every quantity carries a relative error <= K of its magnitude bound
      const float K = 4e-6f;
      const float D = fabsf(det);
      const float sgn = det < 0.f ? -1.f : 1.f;
      const float eD = K * n0 * n1;
      const float us = u * sgn, vs = v * sgn, cs = c * sgn;
      const float eU = K * nT * n1 + eD, eV = K * nT * n0 + eD, eC = K * nT * n0 * n1;
      bool hit = rayValid && D + eD >= 0.99999e-5f;
      hit = hit && us >= -eU && us <= D + eU && vs >= -eV && vs <= D + eV;
      hit = hit && cs > mint * D - eC - mint * eD && cs < maxt * D + eC + maxt * eD;
      const unsigned long long m = __ballot(hit);
      if (m) {
        if (hit) {
          const uint32_t off = __popcll(m & ((1ull << lane) - 1ull));
          s.queue[(qHead + qCount + off) % QCAP] = make_uint2(p, (uint32_t)lane);
        }
        qCount += __popcll(m);
        if (qCount >= 64u) {
          __syncthreads();
          const uint2 e = s.queue[(qHead + lane) % QCAP];
          if (evaluatePlane<B>(a, pa, s, e.x, e.y, nDiff, nFail)) nEval++;
          qHead = (qHead + 64u) % QCAP;
          qCount -= 64u;
          __syncthreads();
        }
      }
    }
    __syncthreads();
    if ((uint32_t)lane < qCount) {
      const uint2 e = s.queue[(qHead + lane) % QCAP];
      if (evaluatePlane<B>(a, pa, s, e.x, e.y, nDiff, nFail)) nEval++;
    }
    __syncthreads();
    for (int idx = lane; idx < 27 * B; idx += 64) {
      const int k = idx / B, bb = idx % B;
      if ((uint32_t)bb < nb) {
        const float val = s.acc[k][bb];
        if (val != 0.f) {
          const uint32_t pv = s.pix[bb];
          const size_t pix = (size_t)(pv >> 16) * a.cfg.width + (pv & 0xFFFFu);
          atomicAdd(&a.iter[pix * 27 + k], val);
        }
      }
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
      atomicAdd(&a.stats[0], ev);
      atomicAdd(&a.stats[1], nCand);
      atomicAdd(&a.stats[3], di);
      atomicAdd(&a.stats[4], fa);
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
