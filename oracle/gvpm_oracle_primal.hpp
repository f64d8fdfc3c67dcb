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
    // (one stack per thread, grown on demand: an allocation per query serialised 256 threads in malloc -- 6x on 256)
    static thread_local std::vector<uint32_t> stackStorage;
    if (stackStorage.size() < (size_t)map.depth + 2) stackStorage.resize((size_t)map.depth + 2);
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

// ---------------------------------------------------------------------------------------------------------------------
// The PRIMAL beam x beam estimate of the sppm integrator (volumePhotonPassBeams, sppm.cpp:762-880) with
// BeamRadianceQuery<PhotonBeam>::operator() (src/integrators/photonmapper/beams.h:29-223), kernels EBeamBeam1D and
// EBeamBeam3D_Optimized (the two the gradient pass has; _Naive / _EGSR are not restated).  The gradient pass's
// BeamKernelRecord::eval (shift_volume_beams.h:157-290) is this functor with the random numbers made explicit; what
// differs: the camera transmittance runs over [Epsilon, w] here (rayBeam.mint = Epsilon, :57-61; cameraRay.mint = Epsilon,
// :189-193) against [0, w] there; no checkerboard, no interaction filters; depth filters against maxDepth = m_maxDepth -
// beam.depth and minDepth = max(0, m_minDepth - beam.depth) of the CAMERA beam (sppm.cpp:852-857).  Per-hit random numbers:
// the same Philox stand-in as the gradient oracle (key {bits(ray rand), 0x6265616d}, counter {beam index, 0, 0, 0}).
template <typename F> struct PrimalBeamRadianceQuery {
  typedef Vec3<F> V;
  const GatherContext<F> &ctx;
  Ray<F> baseCameraRay;
  int maxDepth, minDepth, volTechnique;
  F radius;
  float setRand;
  V Li;
  Counters cnt;
  PrimalBeamRadianceQuery(const GatherContext<F> &c, const Ray<F> &ray, int maxD, int minD, int tech, F r, float rnd)
      : ctx(c), baseCameraRay(ray), maxDepth(maxD), minDepth(minD), volTechnique(tech), radius(r), setRand(rnd), Li((F)0) {}

  bool operator()(const Beam<F> &beam, F tmin, F tmax) {
    cnt.candidates++;
    if (tmax > beam.length) tmax = beam.length;
    const int depth = (int)GVPM_PF_DEPTH(beam.ph.flags);
    if (maxDepth != -1 && depth > maxDepth) return false;
    if (minDepth != 0 && depth < minDepth) return false;
    if (volTechnique == GVPM_BEAM_BEAM_1D) {
      F u, v, w, sinTheta;
      if (!BeamKernelRecord<F>::rayIntersect1D(beam, radius, baseCameraRay, tmin, tmax, u, v, w, sinTheta)) return false;
      if (radius <= u) return false;
      Ray<F> rayBeam(baseCameraRay.o, baseCameraRay.d, ctx.Epsilon, w);
      MRec<F> mRec;
      ctx.medium.eval(rayBeam, mRec);
      F weightKernel = 0.5f / radius;
      V bt;
      F pf;
      Li += BeamKernelRecord<F>::getContrib(ctx.medium, beam, v, mRec, rayBeam.d, bt, pf) * weightKernel / sinTheta;
      cnt.evaluations++;
      return true;
    }
    if (volTechnique != GVPM_BEAM_BEAM_3D_OPTIMIZED) return false;
    F uv, uw;
    beamRandoms<F>(setRand, beam.index, uv, uw);
    Ray<F> _cam(baseCameraRay(baseCameraRay.mint), baseCameraRay.d, (F)0, baseCameraRay.maxt - baseCameraRay.mint);
    Ray<F> _beam(beam.ph.parentPos, beam.dir, (F)0, beam.length);
    double tNearBeam, tFarBeam;
    if (!cylinderIntersection(_cam, _beam, radius, tNearBeam, tFarBeam)) return false;
    if (tNearBeam < 0 && tmin <= ctx.Epsilon) {
    } else if (tNearBeam > tmin && tNearBeam < tmax) {
    } else {
      return false;
    }
    F beamSegmentRand = (F)(tNearBeam + (tFarBeam - tNearBeam) * uv);
    F invPDF = (F)std::max(tFarBeam - tNearBeam, 0.0001);
    if (beamSegmentRand < 0 || beamSegmentRand > beam.length) return false;
    V kernelCentroid = beam.getPos(beamSegmentRand);
    F distToProj = dot(kernelCentroid - baseCameraRay.o, baseCameraRay.d);
    F distSqr = (baseCameraRay(distToProj) - kernelCentroid).lengthSquared();
    F radSqr = radius * radius;
    if (distSqr >= radSqr) return false;
    F deltaT = safe_sqrt(radSqr - distSqr);
    F cameraSegmentRand = distToProj - deltaT + 2 * deltaT * uw;
    invPDF *= (F)std::max(2.0 * (double)deltaT, 0.0001);
    if (cameraSegmentRand < baseCameraRay.mint || cameraSegmentRand > baseCameraRay.maxt) return false;
    Ray<F> rayTrans(beam.ph.parentPos, beam.dir, (F)0, beamSegmentRand);
    MRec<F> mRecBeam;
    ctx.medium.eval(rayTrans, mRecBeam);
    V segmentFlux = beam.ph.flux * mRecBeam.transmittance;
    Ray<F> cameraRay(baseCameraRay.o, baseCameraRay.d, ctx.Epsilon, cameraSegmentRand);
    MRec<F> mRecCamera;
    ctx.medium.eval(cameraRay, mRecCamera);
    F phaseTerm = ctx.medium.phase(-beam.dir, -cameraRay.d);
    F kernelVol = (F)((4.0 / 3.0) * 3.14159265358979323846 * std::pow((double)radius, 3));
    V beamContrib = segmentFlux * mRecCamera.sigmaS * mRecCamera.transmittance * phaseTerm * (invPDF / kernelVol);
    beamContrib /= mRecBeam.pdfFailure;  // (!beam->longBeams: the flattened beams are short beams)
    Li += beamContrib;
    cnt.evaluations++;
    return true;
  }
};

// one camera beam of volumePhotonPassBeams' loop, sppm.cpp:838-859: fluxVolIter += bRadQuery.Li * beam.weight
template <typename F>
inline void gatherBeamPrimalBeams(const GatherContext<F> &ctx, const BeamMapO<F> &map, F radius, const gvpm_camera_ray &b,
                                  F subBeamSize, F *iter, Counters &cnt, const SubBeamBVHO<F, Beam<F>> *accel = nullptr) {
  CamRay<F> base(b);
  Ray<F> ray(base.o, base.d, ctx.Epsilon, base.len - ctx.Epsilon);
  const int maxD = ctx.cfg.max_depth <= 0 ? -1 : ctx.cfg.max_depth - base.edge;
  const int minD = std::max(0, ctx.cfg.min_depth - base.edge);
  PrimalBeamRadianceQuery<F> q(ctx, ray, maxD, minD, ctx.cfg.vol_technique, radius, b.rand);
  if (accel) {
    accel->query(map.beams, ray, q);
  } else for (const Beam<F> &bm : map.beams) {
    if (subBeamSize > 0) {
      int nb = (int)std::ceil(bm.length / subBeamSize);
      F ls = bm.length / nb;
      for (int i = 0; i < nb; ++i) q(bm, ls * i, ls * (i + 1));
    } else {
      q(bm, (F)0, std::numeric_limits<F>::infinity());
    }
  }
  const Vec3<F> r = q.Li * base.eye;
  iter[0] += r.x; iter[1] += r.y; iter[2] += r.z;
  cnt.add(q.cnt);
}

}  // namespace oracle
