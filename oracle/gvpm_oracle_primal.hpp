// ORACLE -- TEST INFRASTRUCTURE ONLY (see gvpm_oracle.hpp header).  parity unpinned: no reference output pins this file.
//
// The PRIMAL beam radiance estimate of the reference's `sppm` integrator (second half), restated
// literally:
//   SPPMIntegrator::volumePhotonPassBRE            src/integrators/photonmapper/sppm.cpp:882-1000
//   BeamRadianceEstimator ctor (cameraHeuristic)   src/integrators/photonmapper/bre.cpp:29-56,128-165
//   BeamRadianceEstimator::query                   src/integrators/photonmapper/bre.cpp:166-254
// The pass builds the estimator with cameraHeuristic = true (sppm.cpp:927): every photon gets the radius
// breInitSize = R * globalScaleVolume * POURCENTAGE_BS, exactly the radius of the gradient pass (gvpm.cpp:989), and the
// hierarchy is the gradient estimator's (a photon's sphere box united with its subtree's).  What differs from the base
// term of VolumeGradientBREQuery (shift_volume_photon.cpp:658-751): the ray is re-based at r(mint) (:168); the photon's
// stored POWER enters without the explicit sigma_s factor (Photon::getPower, :233-235); the 3D kernel draws a random number
// PER HIT (sampler->next1D(), :218) and accepts t' in [0, maxt] of the re-based ray; the 2D kernel HAS a far check
// (:240-242); the photon map's scale factor 1 / shotParticles multiplies every term (:236, sppm.cpp:921); no path-set
// checkerboard, no shifts.
// The photon record is Mitsuba's compressed `Photon` (RGBE power, 8-bit direction angles, include/mitsuba/render/photon.h
// :39-147): the flattened input carries what getPower() / getDirection() / getDepth() RETURN, decoded by the host.
// The per-hit random number of the reference comes from a stateful per-thread sampler, in traversal order: not
// reproducible by any other traversal.  Stand-in specification (as for G-Beams): Philox4x32-10 with
// key = {bits(base ray rand), 0x70726d6c} and counter = {bits(pos.x), bits(pos.y), bits(pos.z), 0} of the photon.
#pragma once
#include "gvpm_oracle.hpp"
#include "gvpm_oracle_beams.hpp"  // philox4x32_10

namespace oracle {

template <typename F> inline F primalHitRandom(float setRand, const Vec3<F> &pos) {
  uint32_t key, c[3];
  std::memcpy(&key, &setRand, 4);
  const float p[3] = {(float)pos.x, (float)pos.y, (float)pos.z};
  std::memcpy(c, p, 12);
  uint32_t o[4];
  philox4x32_10(key, 0x70726d6cu, c[0], c[1], c[2], 0u, o);
  return (F)((float)(o[0] >> 8) * (1.0f / 16777216.0f));
}

// BeamRadianceEstimator::query, bre.cpp:166-254.  useAccel: the stack walk over the hierarchy; else every photon whose own
// box the re-based ray meets (the walk's node test applied to the photon's own box: equal).
template <typename F>
inline Vec3<F> primalQueryBRE(const Gatherer<F> &g, const Ray<F> &r, int maxDepth, bool use3Dkernel, float setRand,
                              bool useAccel, Counters &cnt) {
  typedef Vec3<F> V;
  const PhotonMap<F> &map = g.map;
  const Ray<F> ray(r(r.mint), r.d, (F)0, r.maxt - r.mint);  // :168
  V result((F)0);
  const F m_scaleFactor = (F)1;  // applied by the caller: 1 / shotParticles (sppm.cpp:921)
  auto visit = [&](uint32_t index) {
    const Photon<F> &photon = map.photons[index];
    cnt.candidates++;
    if (maxDepth != -1 && (int)GVPM_PF_DEPTH(photon.flags) > maxDepth) return;  // :193-196
    V originToCenter = photon.pos - ray.o;
    F diskDistance = dot(originToCenter, ray.d), radSqr = map.radius * map.radius;
    F distSqr = (ray(diskDistance) - photon.pos).lengthSquared();
    if (diskDistance > 0 && distSqr < radSqr) {
      if (use3Dkernel) {
        if (diskDistance - (map.radius * 2) > ray.maxt) return;  // :203-206
        F weight = (F)(1 / ((4.0 / 3.0) * 3.14159265358979323846 * std::pow((double)map.radius, 3)));
        F deltaT = std::sqrt(radSqr - distSqr);
        F tminKernel = diskDistance - deltaT;
        F diskDistanceRand = tminKernel + 2 * deltaT * primalHitRandom<F>(setRand, photon.pos);
        if (diskDistanceRand < 0 || diskDistanceRand > ray.maxt) return;  // :220-223
        F invPdfSampling = std::max((F)(2.0f * deltaT), (F)0.0001f);
        V wi = photon.wi;  // -node.photon.getDirection()
        MRec<F> mRecBase;
        Ray<F> baseRay(ray);
        baseRay.maxt = diskDistanceRand;
        g.ctx.medium.eval(baseRay, mRecBase);
        result += mRecBase.transmittance * photon.flux * g.ctx.medium.phase(wi, -ray.d) * (weight * m_scaleFactor) * invPdfSampling;
        cnt.evaluations++;
      } else {
        F weight = (F)(1 / (3.14159265358979323846 * std::pow((double)map.radius, 2)));
        if (diskDistance > ray.maxt) return;  // :240-242
        V wi = photon.wi;
        MRec<F> mRecBase;
        Ray<F> baseRay(ray);
        baseRay.maxt = diskDistance;
        g.ctx.medium.eval(baseRay, mRecBase);
        result += mRecBase.transmittance * photon.flux * g.ctx.medium.phase(wi, -ray.d) * (weight * m_scaleFactor);
        cnt.evaluations++;
      }
    }
  };
  if (map.photons.empty()) return result;
  if (useAccel) {
    std::vector<uint32_t> stackStorage(map.depth + 2);
    uint32_t *stack = stackStorage.data();
    uint32_t index = 0, stackPos = 1;
    while (stackPos > 0) {
      F mint, maxt;
      if (!map.nodeAABB[index].rayIntersect(ray, mint, maxt) || maxt < ray.mint || mint > ray.maxt) {  // :179-183
        index = stack[--stackPos];
        continue;
      }
      const uint32_t cur = index;
      if (!map.isLeaf(cur)) {  // :185-191
        if (map.right[cur] != 0) stack[stackPos++] = map.right[cur];
        index = cur + 1;
      } else {
        index = stack[--stackPos];
      }
      visit(cur);
    }
  } else {
    for (uint32_t i = 0; i < map.photons.size(); ++i) {
      V c = map.photons[i].pos;
      AABB<F> box(c - V(map.radius, map.radius, map.radius), c + V(map.radius, map.radius, map.radius));
      F mint, maxt;
      if (!box.rayIntersect(ray, mint, maxt) || maxt < ray.mint || mint > ray.maxt) continue;
      visit(i);
    }
  }
  return result;
}

// One camera beam of volumePhotonPassBRE's loop, sppm.cpp:949-981: fluxVolIter += query(...) * beam.weight.
// The flattened beam is the base ray of a set: o = beam.p1, d, len = distTotal, eye = beam.weight, edge = beam.depth.
template <typename F>
inline void gatherBeamPrimalBRE(const Gatherer<F> &g, const gvpm_camera_ray &b, bool useAccel, F *iter, Counters &cnt) {
  CamRay<F> base(b);
  const gvpm_params &cfg = g.ctx.cfg;
  const bool use3D = cfg.vol_technique == GVPM_VOL_BRE3D;
  Ray<F> ray(base.o, base.d, g.ctx.Epsilon, base.len - g.ctx.Epsilon);  // :972
  const int maxDepth = cfg.max_depth <= 0 ? -1 : cfg.max_depth - base.edge;  // (m_maxDepth == -1 ? -1 : m_maxDepth - beam.depth)
  const Vec3<F> q = primalQueryBRE<F>(g, ray, maxDepth, use3D, b.rand, useAccel, cnt) * base.eye;
  iter[0] += q.x; iter[1] += q.y; iter[2] += q.z;
}

// ---------------------------------------------------------------------------------------------------------------------
// The PRIMAL point estimate of the sppm integrator's volume pass (SPPMIntegrator::volumePhotonPass with EDistance,
// sppm.cpp:1040-1126) with PhotonMap::estimateVolumeRadiance (src/librender/photonmap.cpp:277-330).  Per camera sample:
// the beam is chosen by the host's CDF (selBeam.sampleReuse, :1088-1089: the flattened sample carries the beam set, the
// re-used random number and selBeam[beamIndex]), the distance is sampled with EDistanceAlwaysValid (:1101-1102), and the
// photons within querySize = BBPourcentageCONST * gp.scaleVol of the sampled point are summed:
//     gp.fluxVol += MCNorm * sum(power * phase) * beam.weight * Tr / (pdfSuccess * selBeam[k])          (:1109-1110)
// between `gp.fluxVol *= kernelVol` and `/= kernelVol` (:1085,1113), i.e. the density accumulates sum / kernelVol.
// M counts the photons that pass BOTH tests of RadianceQueryVolume (radius, then depth -- `maxDepth > 0 &&`: a maxDepth
// of zero or less filters nothing, as written, photonmap.cpp:293).
template <typename F> struct RadianceQueryVolumeO {
  typedef Vec3<F> V;
  const GatherContext<F> &ctx;
  V pos, viewDir, result;
  int maxDepth;
  F searchRadius;
  size_t M;
  Counters cnt;
  RadianceQueryVolumeO(const GatherContext<F> &c, const V &p, const V &vd, int md, F r)
      : ctx(c), pos(p), viewDir(vd), result((F)0), maxDepth(md), searchRadius(r), M(0) {}
  void vpmFunctor(const Photon<F> &photon) {  // operator()(const Photon &), photonmap.cpp:283-308
    V wi = photon.wi;  // -photon.getDirection()
    F lengthSqr = (pos - photon.pos).lengthSquared();
    if ((searchRadius * searchRadius - lengthSqr) < 0) return;
    if (maxDepth > 0 && (int)GVPM_PF_DEPTH(photon.flags) > maxDepth) return;
    M += 1;
    cnt.evaluations++;
    V value = photon.flux * ctx.medium.phase(wi, -viewDir);
    if (value.x == 0 && value.y == 0 && value.z == 0) return;
    result += value;
  }
};

// one camera sample of the loop at sppm.cpp:1087-1112; iter: the pixel's fluxVol increment (3 values); returns M
template <typename F>
inline size_t gatherSamplePrimalVPM(const Gatherer<F> &g, const gvpm_camera_ray *set, F rand, F pdfSel, F querySize, F MCNorm,
                                    bool useAccel, F *iter, Counters &cnt) {
  CamRay<F> base(set[0]);
  Ray<F> ray(base.o, base.d, g.ctx.Epsilon, base.len);  // Ray ray(beam.p1, d, Epsilon, distTotal, 0.f)
  MRec<F> mRec;
  mRec.t = 0;
  if (!g.ctx.medium.sampleDistanceAlwaysValid(ray, mRec, rand, g.ctx.Epsilon)) return 0;
  const int maxDepth = g.ctx.cfg.max_depth <= 0 ? 0x7FFFFFFF : g.ctx.cfg.max_depth - base.edge;  // m_maxDepth == -1 ? INT_MAX : m_maxDepth - beam.depth
  const Vec3<F> p = ray(mRec.t);  // mRec.p
  RadianceQueryVolumeO<F> query(g.ctx, p, ray.d, maxDepth, querySize);
  if (useAccel) g.map.executeQuery(p, querySize, query);
  else g.map.executeQueryBrute(p, querySize, query);
  const F kernelVol = (F)((4.0 / 3.0) * 3.14159265358979323846 * std::pow((double)querySize, 3));
  const Vec3<F> add = query.result * base.eye * (MCNorm * mRec.transmittance.x / (mRec.pdfSuccess * pdfSel));
  iter[0] += add.x / kernelVol;
  iter[1] += add.y / kernelVol;
  iter[2] += add.z / kernelVol;
  cnt.add(query.cnt);
  return query.M;
}

}  // namespace oracle
