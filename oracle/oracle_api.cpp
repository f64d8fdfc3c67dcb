// ORACLE -- TEST INFRASTRUCTURE ONLY (see gvpm_oracle.hpp header).
// C entry points used by tests/ and bench.py's cpu_baseline leg via ctypes.
#include <chrono>
#include <cstdio>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

#include "gvpm_oracle.hpp"
#include "gvpm_oracle_beams.hpp"
#include "gvpm_oracle_planes.hpp"
#include "gvpm_oracle_primal.hpp"
#include "poisson_oracle.hpp"
#include "camera_path_oracle.hpp"

using namespace oracle;

namespace {

template <typename F>
int gatherBRE(const gvpm_params *p, const gvpm_medium *m, const gvpm_triangles *t, const gvpm_photon_soa *ph,
              const gvpm_camera_ray *rays, uint64_t nsets, double radius, int it, uint64_t nbPaths, int useAccel,
              int threads, double *accum, uint64_t *counters, double *seconds, double *buildSeconds) {
  Gatherer<F> g;
  g.setup(*p, *m, *t);
  g.map.load(*ph);
  auto t0 = std::chrono::steady_clock::now();
  if (useAccel) g.map.buildBRE((F)radius);  // kd-tree + BRE hierarchy: serial, as the reference (gvpm.cpp:450-454)
  else g.map.radius = (F)radius;
  if (buildSeconds) *buildSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const size_t P = (size_t)p->width * p->height;
  std::vector<F> perSet((size_t)nsets * 27, (F)0);
  Counters total;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
  // BlockScheduler (photonmapper/utilities/block_sched.h:87-113): threads pull
  // work dynamically; here a work item is a run of 16 consecutive beam sets (round 5: runs of 256 left ~4 items a thread on
  // a 256-thread host over a heavy-tailed load -- 7x on 256 threads)
#pragma omp parallel
  {
    Counters local;
#pragma omp for schedule(dynamic, 16)
    for (int64_t s = 0; s < (int64_t)nsets; ++s) g.gatherSetBRE(rays + 5 * s, useAccel != 0, &perSet[(size_t)s * 27], local);
#pragma omp critical
    total.add(local);
  }
  // per pixel: sum over medium edges (gvpm.cpp:1044-1050), normalise (:1055-1059), APA (:1063-1069)
  std::vector<F> iter(P * 27, (F)0);
  for (uint64_t s = 0; s < nsets; ++s) {
    const gvpm_camera_ray &b = rays[5 * s];
    size_t px = b.pixel & 0xFFFFu, py = b.pixel >> 16;
    if (px >= (size_t)p->width || py >= (size_t)p->height) return GVPM_ERR_INVALID_ARG;
    F *dst = &iter[(py * p->width + px) * 27];
    for (int k = 0; k < 27; ++k) dst[k] += perSet[(size_t)s * 27 + k];
  }
  for (size_t i = 0; i < P * 27; ++i) {
    F v = iter[i];
    v /= (F)nbPaths;  // Spectrum /= size_t -> Float
    F prev = (F)accum[i];
    accum[i] = (double)((prev * (F)(it - 1) + v) / (F)it);
  }
  auto t1 = std::chrono::steady_clock::now();
  if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
  if (counters) {
    counters[0] = total.evaluations; counters[1] = total.candidates; counters[2] = total.nullShifts;
    counters[3] = total.diffuseShifts; counters[4] = total.failedShifts;
  }
  return GVPM_OK;
}

// One iteration of the PRIMAL sppm integrator's volumePhotonPassBRE, sppm.cpp:882-1000 (gvpm_oracle_primal.hpp)
template <typename F>
int gatherPrimalBRE(const gvpm_params *p, const gvpm_medium *m, const gvpm_triangles *t, const gvpm_photon_soa *ph,
                    const gvpm_camera_ray *rays, uint64_t nsets, double radius, int it, uint64_t nbPaths, int useAccel,
                    int threads, double *accum, uint64_t *counters) {
  Gatherer<F> g;
  g.setup(*p, *m, *t);
  g.map.load(*ph);
  if (useAccel) g.map.buildBRE((F)radius);
  else g.map.radius = (F)radius;
  const size_t P = (size_t)p->width * p->height;
  std::vector<F> perSet((size_t)nsets * 3, (F)0);
  Counters total;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
  {
    Counters local;
#pragma omp for schedule(dynamic, 16)
    for (int64_t s = 0; s < (int64_t)nsets; ++s) gatherBeamPrimalBRE<F>(g, rays[5 * s], useAccel != 0, &perSet[(size_t)s * 3], local);
#pragma omp critical
    total.add(local);
  }
  std::vector<F> iter(P * 3, (F)0);
  for (uint64_t s = 0; s < nsets; ++s) {
    const gvpm_camera_ray &b = rays[5 * s];
    size_t px = b.pixel & 0xFFFFu, py = b.pixel >> 16;
    if (px >= (size_t)p->width || py >= (size_t)p->height) return GVPM_ERR_INVALID_ARG;
    for (int k = 0; k < 3; ++k) iter[(py * p->width + px) * 3 + k] += perSet[(size_t)s * 3 + k];
  }
  for (size_t i = 0; i < P; ++i)
    for (int k = 0; k < 3; ++k) {
      // photonMap->setScaleFactor(1 / shotParticles) (sppm.cpp:921) multiplies every term of the query: here once per sum
      F v = iter[i * 3 + k] * ((F)1 / (F)nbPaths);
      F prev = (F)accum[i * 27 + k];
      accum[i * 27 + k] = (double)((prev * (F)(it - 1) + v) / (F)it);  // gp.fluxVol APA fold, sppm.cpp:986
    }
  if (counters) {
    counters[0] = total.evaluations; counters[1] = total.candidates; counters[2] = counters[3] = counters[4] = 0;
  }
  return GVPM_OK;
}

// One iteration of computeVolumeGradientPhoton (G-VPM), gvpm.cpp:1081-1203
template <typename F>
int gatherVPM(const gvpm_params *p, const gvpm_medium *m, const gvpm_triangles *t, const gvpm_photon_soa *ph,
              const gvpm_camera_ray *rays, uint64_t nsets, const gvpm_vpm_sample *samples, uint64_t nsamples,
              int useAccel, int threads, double *accum, double *scaleVol, double *nVol, uint64_t *counters,
              double *seconds, double *buildSeconds, int primal = 0) {
  Gatherer<F> g;
  g.setup(*p, *m, *t);
  g.map.load(*ph);
  auto t0 = std::chrono::steady_clock::now();
  if (useAccel) g.map.buildKD();
  if (buildSeconds) *buildSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const size_t P = (size_t)p->width * p->height;
  // BBPourcentageCONST = bsphere radius * POURCENTAGE_BS (gvpm.cpp:1082), Float arithmetic
  const F BBPourcentageCONST = (F)p->bsphere_radius * (F)0.01;
  const F normalization = 1.f / p->nb_camera_samples;
  std::vector<F> perSample((size_t)nsamples * 27, (F)0);
  std::vector<uint32_t> found(nsamples, 0);
  Counters total;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
  int bad = 0;
#pragma omp parallel
  {
    Counters local;
#pragma omp for schedule(dynamic, 16)
    for (int64_t s = 0; s < (int64_t)nsamples; ++s) {
      const gvpm_vpm_sample &sm = samples[s];
      if (sm.set >= nsets) { bad = 1; continue; }
      const gvpm_camera_ray *set = rays + 5 * (size_t)sm.set;
      const size_t px = set[0].pixel & 0xFFFFu, py = set[0].pixel >> 16;
      if (px >= (size_t)p->width || py >= (size_t)p->height) { bad = 1; continue; }
      const F querySize = BBPourcentageCONST * (F)scaleVol[py * p->width + px];  // gvpm.cpp:1132
      // (primal: the sppm integrator's point estimate, sppm.cpp:1087-1112 -- gvpm_oracle_primal.hpp)
      found[s] = primal ? (uint32_t)gatherSamplePrimalVPM<F>(g, set, (F)sm.rand, (F)sm.pdf_sel, querySize, normalization,
                                                             useAccel != 0, &perSample[(size_t)s * 27], local)
                        : (uint32_t)gatherSampleVPM<F>(g, set, (F)sm.rand, (F)sm.pdf_sel, querySize, normalization,
                                                       useAccel != 0, &perSample[(size_t)s * 27], local);
    }
#pragma omp critical
    total.add(local);
  }
  if (bad) return GVPM_ERR_INVALID_ARG;
  // gp.mediumFlux += gRec.mediumFlux * normalization, in sample order (gvpm.cpp:1174-1179)
  std::vector<F> MVol(P, (F)0);
  for (uint64_t s = 0; s < nsamples; ++s) {
    const gvpm_camera_ray &b = rays[5 * (size_t)samples[s].set];
    const size_t pix = (size_t)(b.pixel >> 16) * p->width + (b.pixel & 0xFFFFu);
    for (int k = 0; k < 27; ++k) accum[pix * 27 + k] = (double)((F)accum[pix * 27 + k] + perSample[(size_t)s * 27 + k]);
    MVol[pix] += (F)found[s];
  }
  // SPPM update, gvpm.cpp:1191-1195
  for (size_t i = 0; i < P; ++i) {
    F NV = (F)nVol[i], M = MVol[i], sc = (F)scaleVol[i];
    if (M + NV != 0) {
      F ratioVol = (NV + (F)p->alpha * M) / (NV + M);
      sc = sc * std::cbrt(ratioVol);
      NV = NV + (F)p->alpha * M;
      scaleVol[i] = (double)sc;
      nVol[i] = (double)NV;
    }
  }
  auto t1 = std::chrono::steady_clock::now();
  if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
  if (counters) {
    counters[0] = total.evaluations; counters[1] = total.candidates; counters[2] = total.nullShifts;
    counters[3] = total.diffuseShifts; counters[4] = total.failedShifts;
  }
  return GVPM_OK;
}

// One iteration of computeVolumeGradientBeams, gvpm.cpp:880-986
template <typename F>
int gatherBeams(const gvpm_params *p, const gvpm_medium *m, const gvpm_triangles *t, const gvpm_photon_soa *beams,
                const float *endN, const gvpm_camera_ray *rays, uint64_t nsets, double radius, int it,
                uint64_t nbPaths, double subBeamSize, int useAccel, int threads, double *accum, uint64_t *counters,
                double *seconds, double *buildSeconds, int primal = 0) {
  Gatherer<F> g;
  g.setup(*p, *m, *t);
  BeamMapO<F> map;
  map.load(*beams, endN);
  auto t0 = std::chrono::steady_clock::now();
  // new BeamMap<LTPhotonBeam>(...)->build(EBVHAccel): serial, as the reference (gvpm.cpp:450-454)
  SubBeamBVHO<F, Beam<F>> bvh;
  if (useAccel) bvh.build(map.beams, (F)radius);
  if (buildSeconds) *buildSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const size_t P = (size_t)p->width * p->height;
  std::vector<F> perSet((size_t)nsets * 27, (F)0);
  Counters total;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
  {
    Counters local;
#pragma omp for schedule(dynamic, 16)
    for (int64_t s = 0; s < (int64_t)nsets; ++s)
      if (primal)  // the sppm integrator's beam pass (gvpm_oracle_primal.hpp): fluxVol only
        gatherBeamPrimalBeams<F>(g.ctx, map, (F)radius, rays[5 * s], (F)subBeamSize, &perSet[(size_t)s * 27], local,
                                 useAccel ? &bvh : nullptr);
      else
      gatherSetBeams<F>(g.ctx, map, (F)radius, rays + 5 * s, (F)subBeamSize, &perSet[(size_t)s * 27], local,
                        useAccel ? &bvh : nullptr);
#pragma omp critical
    total.add(local);
  }
  std::vector<F> iter(P * 27, (F)0);
  for (uint64_t s = 0; s < nsets; ++s) {
    const gvpm_camera_ray &b = rays[5 * s];
    size_t px = b.pixel & 0xFFFFu, py = b.pixel >> 16;
    if (px >= (size_t)p->width || py >= (size_t)p->height) return GVPM_ERR_INVALID_ARG;
    F *dst = &iter[(py * p->width + px) * 27];
    for (int k = 0; k < 27; ++k) dst[k] += perSet[(size_t)s * 27 + k];
  }
  // normalisation by nbPathBeams and APA fold, gvpm.cpp:959-975
  for (size_t i = 0; i < P * 27; ++i) {
    F v = iter[i];
    v /= (F)nbPaths;
    F prev = (F)accum[i];
    accum[i] = (double)((prev * (F)(it - 1) + v) / (F)it);
  }
  auto t1 = std::chrono::steady_clock::now();
  if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
  if (counters) {
    counters[0] = total.evaluations; counters[1] = total.candidates; counters[2] = total.nullShifts;
    counters[3] = total.diffuseShifts; counters[4] = total.failedShifts;
  }
  return GVPM_OK;
}

// One iteration of computeVolumeGradientPlanes, gvpm.cpp:782-878
template <typename F>
int gatherPlanes(const gvpm_params *p, const gvpm_medium *m, const gvpm_triangles *t, const gvpm_photon_soa *beams,
                 const float *w1, const float *len1, const gvpm_camera_ray *rays, uint64_t nsets, int it,
                 uint64_t nbPaths, int useAccel, int threads, double *accum, uint64_t *counters, double *seconds,
                 double *buildSeconds) {
  Gatherer<F> g;
  g.setup(*p, *m, *t);
  PlaneMapO<F> map;
  map.load(*beams, w1, len1);
  auto t0 = std::chrono::steady_clock::now();
  // m_planesAccel = new PhotonPlaneBVH<LTPhotonPlane>(m_planes), gvpm.cpp:799
  PhotonPlaneBVHO<F, Plane<F>> bvh;
  if (useAccel) bvh.build(map.planes);
  if (buildSeconds) *buildSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  const size_t P = (size_t)p->width * p->height;
  std::vector<F> perSet((size_t)nsets * 27, (F)0);
  Counters total;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
  {
    Counters local;
#pragma omp for schedule(dynamic, 16)
    for (int64_t s = 0; s < (int64_t)nsets; ++s)
      gatherSetPlanes<F>(g.ctx, map, rays + 5 * s, &perSet[(size_t)s * 27], local, useAccel ? &bvh : nullptr);
#pragma omp critical
    total.add(local);
  }
  std::vector<F> iter(P * 27, (F)0);
  for (uint64_t s = 0; s < nsets; ++s) {
    const gvpm_camera_ray &b = rays[5 * s];
    size_t px = b.pixel & 0xFFFFu, py = b.pixel >> 16;
    if (px >= (size_t)p->width || py >= (size_t)p->height) return GVPM_ERR_INVALID_ARG;
    F *dst = &iter[(py * p->width + px) * 27];
    for (int k = 0; k < 27; ++k) dst[k] += perSet[(size_t)s * 27 + k];
  }
  for (size_t i = 0; i < P * 27; ++i) {  // gvpm.cpp:850-866
    F v = iter[i];
    v /= (F)nbPaths;
    F prev = (F)accum[i];
    accum[i] = (double)((prev * (F)(it - 1) + v) / (F)it);
  }
  auto t1 = std::chrono::steady_clock::now();
  if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
  if (counters) {
    counters[0] = total.evaluations; counters[1] = total.candidates; counters[2] = total.nullShifts;
    counters[3] = total.diffuseShifts; counters[4] = total.failedShifts;
  }
  return GVPM_OK;
}

}  // namespace

extern "C" {
// the BSDF table of the scene's glossy surfaces for the gathers that follow (gvpm_upload_bsdfs on the device side)
int oracle_set_bsdfs(const gvpm_bsdf *table, uint32_t n) {
  if (n && !table) return GVPM_ERR_INVALID_ARG;
  oracle::bsdfTable().assign(table, table + n);
  return GVPM_OK;
}

// One iteration of computeVolumeGradientPlanes (gvpm.cpp:782-878) on the CPU.  use_accel: 1 = through the
// reference's PhotonPlaneBVH (pm/plane_accel.h:85-207), 0 = a loop over all planes.  seconds: build + gather;
// build_seconds (optional): the serial build alone.
int oracle_gather_planes(const gvpm_params *p, const gvpm_medium *m, const gvpm_triangles *t,
                         const gvpm_photon_soa *beams, const float *w1, const float *len1,
                         const gvpm_camera_ray *rays, uint64_t nsets, int it, uint64_t nb_paths, int precision,
                         int use_accel, int threads, double *accum, uint64_t *counters, double *seconds,
                         double *build_seconds) {
  if (!p || !m || !t || !beams || (beams->n && (!w1 || !len1)) || (!rays && nsets) || !accum) return GVPM_ERR_INVALID_ARG;
  if (p->vol_technique != GVPM_VOL_PLANE0D) return GVPM_ERR_INVALID_ARG;
  if (precision == 32)
    return gatherPlanes<float>(p, m, t, beams, w1, len1, rays, nsets, it, nb_paths, use_accel, threads, accum, counters,
                               seconds, build_seconds);
  return gatherPlanes<double>(p, m, t, beams, w1, len1, rays, nsets, it, nb_paths, use_accel, threads, accum, counters,
                              seconds, build_seconds);
}

// One iteration of computeVolumeGradientBeams (gvpm.cpp:880-986) on the CPU.  use_accel: 1 = through the
// reference's SubBeamBVH (pm/beams_accel.h:82-267: what the integrator builds, EBVHAccel), 0 = the reference's
// ENoAccel loop over all beams (pm/beams.h:289-294).  sub_beam_size > 0 (ENoAccel only): every beam is also cut
// into sub-beams of that length, exercising the ownership rule of SubBeamBVH (pm/beams_accel.h:98-131).
// seconds: build + gather; build_seconds (optional): the serial build alone.
int oracle_gather_beams(const gvpm_params *p, const gvpm_medium *m, const gvpm_triangles *t,
                        const gvpm_photon_soa *beams, const float *end_n, const gvpm_camera_ray *rays,
                        uint64_t nsets, double radius, int it, uint64_t nb_paths, int precision,
                        double sub_beam_size, int use_accel, int threads, double *accum, uint64_t *counters,
                        double *seconds, double *build_seconds) {
  if (!p || !m || !t || !beams || (beams->n && !end_n) || (!rays && nsets) || !accum) return GVPM_ERR_INVALID_ARG;
  if (p->vol_technique != GVPM_BEAM_BEAM_1D && p->vol_technique != GVPM_BEAM_BEAM_3D_OPTIMIZED)
    return GVPM_ERR_UNSUPPORTED;  // BeamKernelRecord::eval: SAssert(false) for the other variants
  if (precision == 32)
    return gatherBeams<float>(p, m, t, beams, end_n, rays, nsets, radius, it, nb_paths, sub_beam_size, use_accel, threads,
                              accum, counters, seconds, build_seconds);
  return gatherBeams<double>(p, m, t, beams, end_n, rays, nsets, radius, it, nb_paths, sub_beam_size, use_accel, threads,
                             accum, counters, seconds, build_seconds);
}

// One iteration of computeVolumeGradientPhoton (gvpm.cpp:1081-1203) on the CPU.  accum: P*27
// doubles (in/out, plain sums); scale_vol / n_vol: P doubles each (in/out GatherPoint state).
int oracle_gather_vpm_timed(const gvpm_params *p, const gvpm_medium *m, const gvpm_triangles *t,
                            const gvpm_photon_soa *ph, const gvpm_camera_ray *rays, uint64_t nsets,
                            const gvpm_vpm_sample *samples, uint64_t nsamples, int precision, int use_accel, int threads,
                            double *accum, double *scale_vol, double *n_vol, uint64_t *counters, double *seconds,
                            double *build_seconds);
int oracle_gather_vpm(const gvpm_params *p, const gvpm_medium *m, const gvpm_triangles *t,
                      const gvpm_photon_soa *ph, const gvpm_camera_ray *rays, uint64_t nsets,
                      const gvpm_vpm_sample *samples, uint64_t nsamples, int precision, int use_accel, int threads,
                      double *accum, double *scale_vol, double *n_vol, uint64_t *counters, double *seconds) {
  return oracle_gather_vpm_timed(p, m, t, ph, rays, nsets, samples, nsamples, precision, use_accel, threads, accum,
                                 scale_vol, n_vol, counters, seconds, nullptr);
}
// (seconds: kd-tree build + gather; build_seconds: the serial build alone)
int oracle_gather_vpm_timed(const gvpm_params *p, const gvpm_medium *m, const gvpm_triangles *t,
                            const gvpm_photon_soa *ph, const gvpm_camera_ray *rays, uint64_t nsets,
                            const gvpm_vpm_sample *samples, uint64_t nsamples, int precision, int use_accel, int threads,
                            double *accum, double *scale_vol, double *n_vol, uint64_t *counters, double *seconds,
                            double *build_seconds) {
  if (!p || !m || !t || !ph || (!rays && nsets) || (!samples && nsamples) || !accum || !scale_vol || !n_vol)
    return GVPM_ERR_INVALID_ARG;
  if (p->vol_technique != GVPM_DISTANCE) return GVPM_ERR_INVALID_ARG;
  if (precision == 32)
    return gatherVPM<float>(p, m, t, ph, rays, nsets, samples, nsamples, use_accel, threads, accum, scale_vol, n_vol,
                            counters, seconds, build_seconds);
  return gatherVPM<double>(p, m, t, ph, rays, nsets, samples, nsamples, use_accel, threads, accum, scale_vol, n_vol,
                           counters, seconds, build_seconds);
}

// One iteration of computeVolumeGradientPhotonBRE (gvpm.cpp:988-1079) on the CPU.
// precision: 32 or 64; use_accel: 1 = kd-tree -> BRE BVH stack traversal (reference),
// 0 = brute-force O(B*N) over the same hit predicate.  accum: width*height*27 doubles
// (in/out, the APA running mean).  counters: 5 x uint64 {evaluations, candidates,
// null, diffuse, failed}.
// the stand-in of the host's manifold walk (gvpm_oracle.hpp standinManifoldWalk) applied to downloaded requests: what the
// tests of gvpm_upload_host_shifts hand back to the device
int oracle_standin_host_shifts(const gvpm_photon_soa *ph, const gvpm_shift_request *req, uint64_t n, gvpm_host_shift *out) {
  using R = oracle::VolumeGradientRecord<double>;
  for (uint64_t k = 0; k < n; ++k) {
    const uint64_t i = req[k].photon;
    if (i >= ph->n) return 1;
    auto v3 = [&](const float *a) { return oracle::Vec3<double>(a[3 * i], a[3 * i + 1], a[3 * i + 2]); };
    const oracle::Vec3<double> off(req[k].offset_pos[0], req[k].offset_pos[1], req[k].offset_pos[2]);
    const R::HostShift hs = R::standinManifoldWalk(off, v3(ph->pos), v3(ph->parent_pos), v3(ph->prefix_w),
                                                   (double)ph->parent_pdf[i], (double)ph->edge_pdf[i]);
    out[k].ok = hs.ok ? 1u : 0u;
    for (int c = 0; c < 3; ++c) {
      out[k].throughput[c] = (float)hs.throughput[c];
      out[k].wi[c] = (float)hs.wi[c];
    }
    out[k].pdf = (float)hs.pdf;
    out[k].det_ratio = (float)hs.detRatio;
    out[k].base_pdf = (float)hs.basePdf;
  }
  return 0;
}

// the second stand-in (gvpm_oracle.hpp mirrorManifoldWalk: the image construction through the parent's plane) applied to
// downloaded requests, and the switch that makes the G-BRE gathers that follow answer with it (0: the smooth closed form)
int oracle_mirror_host_shifts(const gvpm_photon_soa *ph, const gvpm_shift_request *req, uint64_t n, gvpm_host_shift *out) {
  using R = oracle::VolumeGradientRecord<double>;
  for (uint64_t k = 0; k < n; ++k) {
    const uint64_t i = req[k].photon;
    if (i >= ph->n) return 1;
    auto v3 = [&](const float *a) { return oracle::Vec3<double>(a[3 * i], a[3 * i + 1], a[3 * i + 2]); };
    const oracle::Vec3<double> off(req[k].offset_pos[0], req[k].offset_pos[1], req[k].offset_pos[2]);
    const R::HostShift hs = R::mirrorManifoldWalk(off, v3(ph->pos), v3(ph->parent_pos), v3(ph->parent_n), v3(ph->parent_wi),
                                                  v3(ph->prefix_w), (double)ph->parent_pdf[i], (double)ph->edge_pdf[i]);
    out[k].ok = hs.ok ? 1u : 0u;
    for (int c = 0; c < 3; ++c) {
      out[k].throughput[c] = (float)hs.throughput[c];
      out[k].wi[c] = (float)hs.wi[c];
    }
    out[k].pdf = (float)hs.pdf;
    out[k].det_ratio = (float)hs.detRatio;
    out[k].base_pdf = (float)hs.basePdf;
  }
  return 0;
}
int oracle_set_manifold_walk(int kind) {
  if (kind != 0 && kind != 1) return GVPM_ERR_INVALID_ARG;
  oracle::VolumeGradientRecord<double>::manifoldWalkKind() = kind;
  oracle::VolumeGradientRecord<float>::manifoldWalkKind() = kind;
  return GVPM_OK;
}

int oracle_gather_bre_timed(const gvpm_params *p, const gvpm_medium *m, const gvpm_triangles *t,
                            const gvpm_photon_soa *ph, const gvpm_camera_ray *rays, uint64_t nsets, double radius, int it,
                            uint64_t nb_paths, int precision, int use_accel, int threads, double *accum,
                            uint64_t *counters, double *seconds, double *build_seconds);
int oracle_gather_bre(const gvpm_params *p, const gvpm_medium *m, const gvpm_triangles *t,
                      const gvpm_photon_soa *ph, const gvpm_camera_ray *rays, uint64_t nsets, double radius, int it,
                      uint64_t nb_paths, int precision, int use_accel, int threads, double *accum,
                      uint64_t *counters, double *seconds) {
  return oracle_gather_bre_timed(p, m, t, ph, rays, nsets, radius, it, nb_paths, precision, use_accel, threads, accum, counters,
                                 seconds, nullptr);
}
// (seconds: kd-tree + BVH build + gather; build_seconds: the serial build alone)
int oracle_gather_bre_timed(const gvpm_params *p, const gvpm_medium *m, const gvpm_triangles *t,
                            const gvpm_photon_soa *ph, const gvpm_camera_ray *rays, uint64_t nsets, double radius, int it,
                            uint64_t nb_paths, int precision, int use_accel, int threads, double *accum,
                            uint64_t *counters, double *seconds, double *build_seconds) {
  if (!p || !m || !t || !ph || (!rays && nsets) || !accum) return GVPM_ERR_INVALID_ARG;
  if (p->vol_technique != GVPM_VOL_BRE2D && p->vol_technique != GVPM_VOL_BRE3D) return GVPM_ERR_INVALID_ARG;
  if (p->use_shift_null && p->vol_technique == GVPM_VOL_BRE2D) return GVPM_ERR_UNSUPPORTED;  // gvpm_struct.h:310-313
  if (precision == 32)
    return gatherBRE<float>(p, m, t, ph, rays, nsets, radius, it, nb_paths, use_accel, threads, accum, counters, seconds,
                            build_seconds);
  return gatherBRE<double>(p, m, t, ph, rays, nsets, radius, it, nb_paths, use_accel, threads, accum, counters, seconds,
                           build_seconds);
}

// the sppm integrator's point estimate (EDistance): accum / scale_vol / n_vol as oracle_gather_vpm; only fluxVol (the first
// three accumulators of a pixel) is written
int oracle_gather_primal_vpm(const gvpm_params *p, const gvpm_medium *m, const gvpm_triangles *t, const gvpm_photon_soa *ph,
                             const gvpm_camera_ray *rays, uint64_t nsets, const gvpm_vpm_sample *samples, uint64_t nsamples,
                             int precision, int use_accel, int threads, double *accum, double *scale_vol, double *n_vol,
                             uint64_t *counters) {
  if (!p || !m || !t || !ph || !accum || !scale_vol || !n_vol || (nsamples && (!samples || !rays))) return GVPM_ERR_INVALID_ARG;
  if (p->vol_technique != GVPM_DISTANCE || p->nb_camera_samples <= 0) return GVPM_ERR_INVALID_ARG;
  if (precision == 32)
    return gatherVPM<float>(p, m, t, ph, rays, nsets, samples, nsamples, use_accel, threads, accum, scale_vol, n_vol, counters,
                            nullptr, nullptr, 1);
  return gatherVPM<double>(p, m, t, ph, rays, nsets, samples, nsamples, use_accel, threads, accum, scale_vol, n_vol, counters,
                           nullptr, nullptr, 1);
}

// the sppm integrator's beam x beam pass: as oracle_gather_beams, fluxVol in the first three accumulators of a pixel
int oracle_gather_primal_beams(const gvpm_params *p, const gvpm_medium *m, const gvpm_triangles *t, const gvpm_photon_soa *beams,
                               const float *end_n, const gvpm_camera_ray *rays, uint64_t nsets, double radius, int it,
                               uint64_t nb_paths, int precision, double sub_beam_size, int use_accel, int threads, double *accum,
                               uint64_t *counters) {
  if (!p || !m || !t || !beams || !end_n || (!rays && nsets) || !accum) return GVPM_ERR_INVALID_ARG;
  if (p->vol_technique != GVPM_BEAM_BEAM_1D && p->vol_technique != GVPM_BEAM_BEAM_3D_OPTIMIZED) return GVPM_ERR_INVALID_ARG;
  if (precision == 32)
    return gatherBeams<float>(p, m, t, beams, end_n, rays, nsets, radius, it, nb_paths, sub_beam_size, use_accel, threads, accum,
                              counters, nullptr, nullptr, 1);
  return gatherBeams<double>(p, m, t, beams, end_n, rays, nsets, radius, it, nb_paths, sub_beam_size, use_accel, threads, accum,
                             counters, nullptr, nullptr, 1);
}

// accum: P*27 doubles, only the first three of a pixel (fluxVol) are read and written
int oracle_gather_primal_bre(const gvpm_params *p, const gvpm_medium *m, const gvpm_triangles *t, const gvpm_photon_soa *ph,
                             const gvpm_camera_ray *rays, uint64_t nsets, double radius, int it, uint64_t nb_paths,
                             int precision, int use_accel, int threads, double *accum, uint64_t *counters) {
  if (!p || !m || !t || !ph || (!rays && nsets) || !accum) return GVPM_ERR_INVALID_ARG;
  if (p->vol_technique != GVPM_VOL_BRE2D && p->vol_technique != GVPM_VOL_BRE3D) return GVPM_ERR_INVALID_ARG;
  if (precision == 32) return gatherPrimalBRE<float>(p, m, t, ph, rays, nsets, radius, it, nb_paths, use_accel, threads, accum, counters);
  return gatherPrimalBRE<double>(p, m, t, ph, rays, nsets, radius, it, nb_paths, use_accel, threads, accum, counters);
}

double oracle_scale_volume_apa(double global_scale, int it, double alpha, int technique) {
  return scaleVolumeAPA(global_scale, it, alpha, technique);
}

// Throughput + gradient assembly: gvpm.cpp:480-532 (normalise, reusePrimal) and
// computeGradient, gvpm.cpp:1205-1306, for an APA volume estimator
// (isAPAVolumeEstimator(): no division by the emitted count).
// accum: P*27 doubles; emission: P*3 doubles or NULL; outputs P*3 doubles each.
// total_emitted > 0: non-APA estimator (G-VPM): volume terms are divided by m_totalEmittedVolume
// (gvpm.cpp:489-492, 526-528, 1252-1253, 1296-1297); 0: APA estimator.
int oracle_assemble_ex(int width, int height, int it, int reuse_primal, double total_emitted, const double *accum,
                       const double *emission, double *throughput, double *dx, double *dy);
int oracle_assemble(int width, int height, int it, int reuse_primal, const double *accum, const double *emission,
                    double *throughput, double *dx, double *dy) {
  return oracle_assemble_ex(width, height, it, reuse_primal, 0.0, accum, emission, throughput, dx, dy);
}
int oracle_assemble_ex(int width, int height, int it, int reuse_primal, double total_emitted, const double *accum,
                       const double *emission, double *throughput, double *dx, double *dy) {
  if (!accum || !throughput || !dx || !dy) return GVPM_ERR_INVALID_ARG;
  const double div = total_emitted > 0 ? total_emitted : 1.0;
  auto A = [&](int x, int y, int k, int c) { return accum[((size_t)y * width + x) * 27 + k * 3 + c]; };
  // k: 0 mediumFlux, 1+i shifted[i], 5+i weighted[i]
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x)
      for (int c = 0; c < 3; ++c) {
        size_t o = ((size_t)y * width + x) * 3 + c;
        double em = emission ? emission[o] / it : 0.0;
        double v = A(x, y, 0, c) / div + 0.0 + em;  // fluxMedia + fluxSurface + emission/it
        if (reuse_primal) {
          double T = 0;
          if (x != width - 1) T += A(x + 1, y, 1 + GVPM_LEFT, c);
          if (x != 0) T += A(x - 1, y, 1 + GVPM_RIGHT, c);
          if (y != height - 1) T += A(x, y + 1, 1 + GVPM_BOTTOM, c);
          if (y != 0) T += A(x, y - 1, 1 + GVPM_TOP, c);
          T += A(x, y, 5 + GVPM_BOTTOM, c) + A(x, y, 5 + GVPM_TOP, c) + A(x, y, 5 + GVPM_RIGHT, c) +
               A(x, y, 5 + GVPM_LEFT, c);
          v = (T / 4.0) / div;
        }
        throughput[o] = v;
        double gx, gy;
        if (x == width - 1) gx = A(x, y, 1 + GVPM_RIGHT, c) - A(x, y, 5 + GVPM_RIGHT, c);
        else gx = (A(x, y, 1 + GVPM_RIGHT, c) - A(x, y, 5 + GVPM_RIGHT, c)) +
                  (A(x + 1, y, 5 + GVPM_LEFT, c) - A(x + 1, y, 1 + GVPM_LEFT, c));
        if (y == height - 1) gy = A(x, y, 1 + GVPM_TOP, c) - A(x, y, 5 + GVPM_TOP, c);
        else gy = (A(x, y, 1 + GVPM_TOP, c) - A(x, y, 5 + GVPM_TOP, c)) +
                  (A(x, y + 1, 5 + GVPM_BOTTOM, c) - A(x, y + 1, 1 + GVPM_BOTTOM, c));
        dx[o] = gx / div;
        dy[o] = gy / div;
      }
  return GVPM_OK;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

// Screened-Poisson reconstruction with one of the reference's presets ("L1D", "L2D", ...); returns -1 for an
// unknown preset.  direct may be NULL.
int oracle_poisson_solve(const char *preset, float alpha, int width, int height, const float *dx, const float *dy,
                         const float *throughput, const float *direct, float *out) {
  oracle::PoissonParams p;
  if (!preset || !oracle::poissonPreset(preset, p)) return -1;
  p.alpha = alpha;
  oracle::poissonSolve(p, width, height, dx, dy, throughput, direct, out);
  return 0;
}
}

// ---- pins ported from the reference's own adjacent tests ------------------------------------------------
// The oracle's kd-tree (PointKDTree, ESlidingMidpoint -- the only heuristic gvpm builds, gvpm_accel.h:96 -- and
// executeQuery, kdtree.h:675-731) queried like src/tests/test_kd.cpp:133-200 queries Mitsuba's: tests compare with a
// brute-force search.  out_idx receives the ORIGINAL indices (path_id carries them) of the points within `radius`.
extern "C" int64_t oracle_kd_radius_query(const float *pos, uint64_t n, const double *query, double radius, int precision,
                                          uint32_t *out_idx, uint64_t cap, uint64_t *visited) {
  if (!pos || !query || !out_idx) return -1;
  std::vector<float> zero3(3 * n, 0.f), zero1(n, 0.f);
  std::vector<uint32_t> zeroU(n, 0u), ids(n);
  for (uint64_t i = 0; i < n; ++i) ids[i] = (uint32_t)i;
  gvpm_photon_soa soa;
  soa.pos = pos; soa.wi = zero3.data(); soa.flux = zero3.data(); soa.parent_pos = zero3.data(); soa.parent_n = zero3.data();
  soa.prefix_w = zero3.data(); soa.parent_scat = zero3.data(); soa.parent_wi = zero3.data();
  soa.parent_pdf = zero1.data(); soa.edge_pdf = zero1.data(); soa.parent_rr = zero1.data(); soa.parent_g = zero1.data();
  soa.flags = zeroU.data(); soa.path_id = ids.data(); soa.n = n;
  auto run = [&](auto tag) -> int64_t {
    using F = decltype(tag);
    PhotonMap<F> map;
    map.load(soa);
    map.buildKD();
    struct Collect {
      Counters cnt;
      std::vector<uint32_t> found;
      void vpmFunctor(const Photon<F> &p) { found.push_back(p.pathID); }
    } c;
    map.executeQuery(Vec3<F>((F)query[0], (F)query[1], (F)query[2]), (F)radius, c);
    if (visited) *visited = c.cnt.candidates;
    for (size_t i = 0; i < c.found.size() && i < cap; ++i) out_idx[i] = c.found[i];
    return (int64_t)c.found.size();
  };
  return precision == 32 ? run(float()) : run(double());
}

// HGPhaseFunction::sample (src/phase/hg.cpp:74-97; Epsilon = 1e-4) and the oracle's phase eval (the pdf): what
// src/tests/test_chisquare.cpp:508-573 checks against each other for data/tests/test_phase.xml's g = 0.9 and -0.3.
// Phong::eval / pdf (phong.cpp:121-186, both components) and Phong::sample (:188-247, bRec.component = -1): what
// src/tests/test_chisquare.cpp (test01_BSDF) checks against each other for the "phong" instance of data/tests/test_bsdf.xml.
extern "C" int oracle_phong_eval_pdf(const gvpm_bsdf *b, const double *kd, const double *n, const double *wi, const double *wo,
                                     double *f, double *pdf) {
  Vec3<double> fv;
  double p = 0;
  const bool ok = phongEvalPdf<double>(*b, Vec3<double>(kd[0], kd[1], kd[2]), Vec3<double>(n[0], n[1], n[2]),
                                       Vec3<double>(wi[0], wi[1], wi[2]), Vec3<double>(wo[0], wo[1], wo[2]), fv, p);
  f[0] = fv.x; f[1] = fv.y; f[2] = fv.z;
  *pdf = p;
  return ok ? 1 : 0;
}
extern "C" int oracle_phong_sample(const gvpm_bsdf *b, const double *n, const double *wiW, double u1, double u2, double *woW) {
  typedef Vec3<double> V;
  const V nn(n[0], n[1], n[2]);
  V s, t;
  coordinateSystem(nn, s, t);
  const V wiV(wiW[0], wiW[1], wiW[2]);
  const V wi(dot(wiV, s), dot(wiV, t), dot(wiV, nn));
  double sx = u1, sy = u2;
  const double w = b->specular_sampling_weight, exponent = b->exponent;
  // (bRec.component = the entry's: with one component there is nothing to choose and the sample is used as it is, :202-211)
  const int component = b->distribution - 1;
  const bool hasSpecular = component == -1 || component == 0, hasDiffuse = component == -1 || component == 1;
  bool choseSpecular = hasSpecular;
  if (hasDiffuse && hasSpecular) {
    if (sx <= w) {
      sx /= w;
    } else {
      sx = (sx - w) / (1 - w);
      choseSpecular = false;
    }
  }
  V wo;
  if (choseSpecular) {
    const V R(-wi.x, -wi.y, wi.z);
    const double sinAlpha = std::sqrt(1 - std::pow(sy, 2 / (exponent + 1)));
    const double cosAlpha = std::pow(sy, 1 / (exponent + 1));
    const double phi = (2.0 * M_PI) * sx;
    const V localDir(sinAlpha * std::cos(phi), sinAlpha * std::sin(phi), cosAlpha);
    V rs, rt;
    coordinateSystem(R, rs, rt);  // Frame(R).toWorld
    wo = rs * localDir.x + rt * localDir.y + R * localDir.z;
    if (wo.z <= 0) return 0;
  } else {
    // warp::squareToCosineHemisphere: squareToUniformDiskConcentric (src/libcore/warp.cpp:81-104) lifted to the hemisphere
    const double r1 = 2.0 * sx - 1.0, r2 = 2.0 * sy - 1.0;
    double phi, r;
    if (r1 == 0 && r2 == 0) {
      r = phi = 0;
    } else if (r1 * r1 > r2 * r2) {
      r = r1;
      phi = (M_PI / 4.0) * (r2 / r1);
    } else {
      r = r2;
      phi = (M_PI / 2.0) - (r1 / r2) * (M_PI / 4.0);
    }
    const double px = r * std::cos(phi), py = r * std::sin(phi);
    wo = V(px, py, std::sqrt(std::max(0.0, 1.0 - px * px - py * py)));
  }
  const V out = s * wo.x + t * wo.y + nn * wo.z;
  woW[0] = out.x; woW[1] = out.y; woW[2] = out.z;
  return 1;
}

// the table entry's eval / pdf whatever its kind (glossyEvalPdf), and MicrofacetDistribution::sampleAll + RoughConductor::sample
// without visible-normal sampling (microfacet.h:287-345, roughconductor.cpp:321-389, isotropic): the half vector, the
// reflected direction, `weight` = F * D G (wi . m) / (pdf_m cos_i), pdf = pdf_m / (4 |wo . m|).
extern "C" int oracle_bsdf_eval_pdf(const gvpm_bsdf *b, const double *kd, const double *n, const double *wi, const double *wo,
                                    double *f, double *pdf) {
  Vec3<double> fv;
  double p = 0;
  const bool ok = glossyEvalPdf<double>(*b, Vec3<double>(kd[0], kd[1], kd[2]), Vec3<double>(n[0], n[1], n[2]),
                                        Vec3<double>(wi[0], wi[1], wi[2]), Vec3<double>(wo[0], wo[1], wo[2]), fv, p);
  f[0] = fv.x; f[1] = fv.y; f[2] = fv.z;
  *pdf = p;
  return ok ? 1 : 0;
}
extern "C" int oracle_roughconductor_sample(const gvpm_bsdf *b, const double *n, const double *wiW, double u1, double u2, double *woW,
                                            double *weight, double *pdf) {
  typedef Vec3<double> V;
  const V nn(n[0], n[1], n[2]);
  V s, t;
  coordinateSystem(nn, s, t);
  const V wiV(wiW[0], wiW[1], wiW[2]);
  const V wi(dot(wiV, s), dot(wiV, t), dot(wiV, nn));
  if (wi.z < 0) return 0;
  const double alpha = b->exponent, alphaSqr = alpha * alpha;
  const double phi = 2.0 * M_PI * u2;
  double tanThetaMSqr, pdfM, cosThetaM;
  if (b->distribution == GVPM_MICROFACET_GGX) {
    tanThetaMSqr = alphaSqr * u1 / (1.0 - u1);
    cosThetaM = 1.0 / std::sqrt(1.0 + tanThetaMSqr);
    const double temp = 1 + tanThetaMSqr / alphaSqr;
    pdfM = (1.0 / M_PI) / (alpha * alpha * cosThetaM * cosThetaM * cosThetaM * temp * temp);
  } else {
    tanThetaMSqr = alphaSqr * -std::log(1.0 - u1);
    cosThetaM = 1.0 / std::sqrt(1.0 + tanThetaMSqr);
    pdfM = (1.0 - u1) / (M_PI * alpha * alpha * cosThetaM * cosThetaM * cosThetaM);
  }
  if (pdfM < 1e-20) pdfM = 0;
  if (pdfM == 0) return 0;
  const double sinThetaM = std::sqrt(std::max(0.0, 1 - cosThetaM * cosThetaM));
  const V m(sinThetaM * std::cos(phi), sinThetaM * std::sin(phi), cosThetaM);
  const V wo = m * (2 * dot(wi, m)) - wi;  // reflect(wi, m)
  if (wo.z <= 0) return 0;
  const double D = microfacetEval<double>(*b, m), G = microfacetSmithG1<double>(*b, wi, m) * microfacetSmithG1<double>(*b, wo, m);
  const double wgt = D * G * dot(wi, m) / (pdfM * wi.z);
  const double wiM = dot(wi, m);
  for (int c = 0; c < 3; ++c) weight[c] = fresnelConductorExact1<double>(wiM, (double)b->eta[c], (double)b->k[c]) * b->specular[c] * wgt;
  *pdf = pdfM / (4.0 * std::abs(dot(wo, m)));
  const V out = s * wo.x + t * wo.y + nn * wo.z;
  woW[0] = out.x; woW[1] = out.y; woW[2] = out.z;
  return 1;
}

// Ward::sample with both components (ward.cpp:268-327, isotropic alpha): the sampled world direction, or 0 (below the horizon)
extern "C" int oracle_ward_sample(const gvpm_bsdf *b, const double *n, const double *wiW, double u1, double u2, double *woW) {
  typedef Vec3<double> V;
  const V nn(n[0], n[1], n[2]);
  V s, t;
  coordinateSystem(nn, s, t);
  const V wiV(wiW[0], wiW[1], wiW[2]);
  const V wi(dot(wiV, s), dot(wiV, t), dot(wiV, nn));
  double sx = u1, sy = u2;
  const double w = b->specular_sampling_weight, alphaU = b->exponent, alphaV = b->exponent;
  bool choseSpecular = true;
  if (sx <= w) {
    sx /= w;
  } else {
    sx = (sx - w) / (1 - w);
    choseSpecular = false;
  }
  V wo;
  if (choseSpecular) {
    double phiH = std::atan(alphaV / alphaU * std::tan(2.0 * M_PI * sy));
    if (sy > 0.5) phiH += M_PI;
    const double cosPhiH = std::cos(phiH), sinPhiH = safe_sqrt(1.0 - cosPhiH * cosPhiH);
    const double thetaH = std::atan(safe_sqrt(-std::log(sx) / ((cosPhiH * cosPhiH) / (alphaU * alphaU) + (sinPhiH * sinPhiH) / (alphaV * alphaV))));
    // sphericalDirection(theta, phi), src/libcore/util.cpp
    const V H(std::sin(thetaH) * std::cos(phiH), std::sin(thetaH) * std::sin(phiH), std::cos(thetaH));
    wo = H * (2.0 * dot(wi, H)) - wi;
    if (wo.z <= 0.0) return 0;
  } else {
    const double r1 = 2.0 * sx - 1.0, r2 = 2.0 * sy - 1.0;
    double phi, r;
    if (r1 == 0 && r2 == 0) {
      r = phi = 0;
    } else if (r1 * r1 > r2 * r2) {
      r = r1;
      phi = (M_PI / 4.0) * (r2 / r1);
    } else {
      r = r2;
      phi = (M_PI / 2.0) - (r1 / r2) * (M_PI / 4.0);
    }
    const double px = r * std::cos(phi), py = r * std::sin(phi);
    wo = V(px, py, std::sqrt(std::max(0.0, 1.0 - px * px - py * py)));
  }
  const V out = s * wo.x + t * wo.y + nn * wo.z;
  woW[0] = out.x; woW[1] = out.y; woW[2] = out.z;
  return 1;
}

extern "C" double oracle_phase_eval(double g, const double *wi, const double *wo) {
  return Medium<double>::phaseEval(g, Vec3<double>(wi[0], wi[1], wi[2]), Vec3<double>(wo[0], wo[1], wo[2]));
}
extern "C" void oracle_hg_sample(double g, const double *wi, double u1, double u2, double *wo) {
  double cosTheta;
  if (std::abs(g) < 1e-4) {
    cosTheta = 1 - 2 * u1;
  } else {
    const double sqrTerm = (1 - g * g) / (1 - g + 2 * g * u1);
    cosTheta = (1 + g * g - sqrTerm * sqrTerm) / (2 * g);
  }
  const double sinTheta = safe_sqrt(1.0 - cosTheta * cosTheta);
  const double sinPhi = std::sin(2 * M_PI * u2), cosPhi = std::cos(2 * M_PI * u2);
  // FrameCoherent(-wi).toWorld
  Vec3<double> n(-wi[0], -wi[1], -wi[2]), s, t;
  coordinateSystemCoherent(n, s, t);
  const Vec3<double> w = s * (sinTheta * cosPhi) + t * (sinTheta * sinPhi) + n * cosTheta;
  wo[0] = w.x; wo[1] = w.y; wo[2] = w.z;
}

// halfVectorShift, gvpm/shift/shift_utilities.h:42-110 (tangent-space vectors; out: wo[3], jacobian)
extern "C" int oracle_half_vector_shift(const double *mainWi, const double *mainWo, const double *shiftedWi, double mainEta,
                                        double shiftedEta, double *out4) {
  using namespace gvpm_oracle;
  const HalfVectorShiftResult r = halfVectorShift(hv(mainWi[0], mainWi[1], mainWi[2]), hv(mainWo[0], mainWo[1], mainWo[2]),
                                                  hv(shiftedWi[0], shiftedWi[1], shiftedWi[2]), mainEta, shiftedEta);
  out4[0] = r.wo.x; out4[1] = r.wo.y; out4[2] = r.wo.z; out4[3] = r.jacobian;
  return r.success ? 1 : 0;
}
// GatherPoint::sensorMIS as written, gvpm_struct.h:608-631
extern "C" double oracle_sensor_mis(unsigned idVertex, double sPdf, double sJac, double sG, double bPdf, double bG, double sDist,
                                    double bDist) {
  return gvpm_oracle::sensorMISRef(idVertex, sPdf, sJac, sG, bPdf, bG, sDist, bDist);
}
