/*
 * gvpm_hip_bridge.h -- the Mitsuba-side half of the drop-in: everything the reference's `gvpm`
 * plugin needs to hand the volume gather of one SPPM iteration to libgvpm_hip.so and to take the result back.
 *
 * How it goes into the reference build.  GPMIntegrator lives in src/integrators/photonmapper/gvpm/gvpm.cpp (a
 * translation unit, not a header), so the drop-in is this header next to it plus the 40-line patch
 * shim/gvpm.cpp.patch: the `switch (m_config.volTechnique)` of photonMapPass (gvpm.cpp:456-474) calls
 * GvpmHipBridge::gather() instead of computeVolumeGradient{Photon,PhotonBRE,Beams,Planes}, and the bridge writes the
 * 27 accumulators back into the GatherPoints, so normalisation, reusePrimal, computeGradient, the Poisson
 * reconstruction and every dump of photonMapPass (gvpm.cpp:476-727) run unchanged.  The plugin stays `gvpm`
 * (MTS_EXPORT_PLUGIN at gvpm.cpp:1530, add_recons(gvpm ...) at src/integrators/CMakeLists.txt:105-146, see
 * shim/CMakeLists.patch); XML scenes do not change; `useHip=false` in the integrator's properties keeps the CPU path.
 *
 * This file cannot be compiled in the build container (Mitsuba needs Boost / Eigen / Xerces / OpenEXR, none present):
 * it is written against the reference headers as they are, every assignment citing the reference expression it
 * copies the VALUE of (not the code: the functors' arithmetic lives in the HIP kernels).
 *
 * Citations are relative to src/integrators/photonmapper/ unless they start with src/ or include/.
 */
#pragma once
#if !defined(__MITSUBA_RENDER_SCENE_H_)
#error "include after the gvpm headers (it is included from gvpm.cpp, below gvpm_struct.h / gvpm_accel.h / gvpm_beams.h / gvpm_plane.h)"
#endif

#include <array>
#include <cctype>
#include <map>
#include <vector>

#include <mitsuba/render/trimesh.h>

#include "../../../bsdfs/ior.h"         /* src/bsdfs/ior.h: lookupIOR, as roughconductor.cpp:22-23 includes them   */
#include "../../../bsdfs/microfacet.h"  /* src/bsdfs/microfacet.h: MicrofacetDistribution(const Properties &)      */

#include "gvpm_hip.h"  /* include/gvpm_hip.h of the gvpm-hip repository */

MTS_NAMESPACE_BEGIN

class GvpmHipBridge {
public:
  GvpmHipBridge() : m_h(nullptr) {}
  ~GvpmHipBridge() {
    if (m_h) gvpm_destroy(m_h);
  }

  /* ---- once per render(): after GPMIntegrator::preprocess and the allocation of the gather points
   *      (gvpm.cpp:126-179,272-291) ------------------------------------------------------------------------------ */
  void create(const Scene *scene, const GPMConfig &config, const AABB &smokeAABB, int device = 0) {
    const Vector2i crop = scene->getFilm()->getCropSize();
    gvpm_params p;
    memset(&p, 0, sizeof(p));
    p.abi_version = GVPM_ABI_VERSION;
    p.width = crop.x;                                               /* film->getCropSize(), gvpm.cpp:396          */
    p.height = crop.y;
    p.vol_technique = (int32_t) config.volTechnique;                /* EVolumeTechnique, same order (volume_utils.h:12-21) */
    p.max_depth = config.maxDepth;                                  /* gvpm_struct.h:107-333, same names          */
    p.min_depth = config.minDepth;
    p.use_mis = config.useMIS ? 1 : 0;                              /* GPMConfig::load: "area" -> true, "none" -> false (:243-256) */
    p.use_shift_null = config.useShiftNull ? 1 : 0;
    p.path_set = config.pathSet ? 1 : 0;
    p.power_heuristic = config.powerHeuristic ? 1 : 0;
    p.no_medium_shift = config.noMediumShift ? 1 : 0;
    p.use_manifold = config.useManifold ? 1 : 0;                    /* the manifold WALK stays on the host: the G-BRE,
                                                                       G-VPM and G-Beams gathers record a request per
                                                                       such shift and answerShiftRequests() /
                                                                       answerBeamShiftRequests() below answer them;
                                                                       G-Planes has no manifold shift             */
    p.debug_shift = (int32_t) config.debugShift;                    /* ELightShiftType values kept                */
    p.lighting_interaction_mode = (int32_t) config.lightingInteractionMode;
    p.bsdf_interaction_mode = (int32_t) config.bsdfInteractionMode;
    p.nb_camera_samples = config.nbCameraSamples;
    p.visibility_as_written = 1;                                    /* shift_volume_photon.cpp:396                */
    p.alpha = (float) config.alpha;
    p.initial_scale_volume = (float) config.initialScaleVolume;
    p.bsphere_radius = (float) smokeAABB.getBSphere().radius;       /* gvpm.cpp:391,881,989,1082                  */
    p.epsilon = (float) Epsilon;                                    /* include/mitsuba/core/constants.h:24-31     */
    p.shadow_epsilon = (float) ShadowEpsilon;
    check(gvpm_create(&p, device, &m_h), "gvpm_create");
    if (config.useManifold) check(gvpm_enable_host_shifts(m_h, kShiftRequestCapacity), "gvpm_enable_host_shifts");
    m_params = p;
    m_config = config;
    uploadScene(scene);
    uploadMedium(scene);
    uploadSensor(scene);
  }

  /* ---- once per SPPM iteration: replaces the switch at gvpm.cpp:456-474 ------------------------------------------
   * gatherBlocks: m_gatherBlocks; threadData: m_threadData[0] (its MemoryPool serves the eager shifted paths);
   * sampler: m_gpManager->getSamplerBlock(0) (the draws of gvpm.cpp:1042 / :1145 / gvpm_plane.h:60-68, made here
   * in block order instead of inside the worker threads: a deterministic stream).
   * Returns after the results are back in the GatherPoints; globalScaleVolume is advanced like scaleVolumeAPA(it). */
  void gather(int it, Scene *scene, std::vector<std::vector<GatherPoint>> &gatherBlocks, GPMThreadData &threadData,
              Sampler *sampler, const GPhotonMap *photonMap, const LTBeamMap *beamMap, size_t nbPaths,
              Float &globalScaleVolume) {
    /* the radius of this pass: R * 0.01 * globalScaleVolume (gvpm.cpp:391,881,989); the handle applies
     * scaleVolumeAPA itself (gvpm.cpp:181-215), set explicitly here so that host and device never drift */
    check(gvpm_set_global_scale(m_h, (float) globalScaleVolume), "gvpm_set_global_scale");
    const EVolumeTechnique tech = (EVolumeTechnique) m_params.vol_technique;
    m_soa.clear();
    switch (tech) {
      case EDistance:
      case EVolBRE2D:
      case EVolBRE3D:
        flattenPhotons(photonMap);
        uploadPhotons();
        break;
      case EBeamBeam1D:
      case EBeamBeam3D_Optimized:
        flattenBeams(beamMap);
        check(gvpm_upload_beams(m_h, &m_soa.view(), m_endN.data()), "gvpm_upload_beams");
        break;
      case EVolPlane0D:
        if (scene->getSensor()->getMedium() == nullptr)             /* gvpm.cpp:784-788                            */
          SLog(EError, "Planes does not support camera outside the medium");
        flattenPlanes(beamMap, sampler);
        check(gvpm_upload_planes(m_h, &m_soa.view(), m_w1.data(), m_len1.data()), "gvpm_upload_planes");
        break;
      default: /* EBeamBeam3D_Naive / _EGSR: SAssert(false) in BeamKernelRecord::eval (shift_volume_beams.h) */
        SLog(EError, "gvpm_hip: volume technique not available (the reference asserts on it too)");
    }
    flattenCameraBeams(scene, gatherBlocks, threadData, sampler, tech == EDistance);
    uploadCameraBeams();   /* (compact sets reorder the upload: m_samples' set indices are remapped there) */
    if (tech == EDistance) {
      uploadVpmState(gatherBlocks);   /* nothing to send: scaleVol / NVol live on the device (see writeBack) */
      check(gvpm_upload_vpm_samples(m_h, m_samples.data(), m_samples.size()), "gvpm_upload_vpm_samples");
    }
    if (m_bsdfsDirty) {   /* a glossy BSDF met for the first time while flattening this iteration's light paths */
      check(gvpm_upload_bsdfs(m_h, m_bsdfs.data(), (uint32_t) m_bsdfs.size()), "gvpm_upload_bsdfs");
      m_bsdfsDirty = false;
    }
    check(gvpm_gather(m_h, it, (uint64_t) nbPaths), "gvpm_gather");
    /* manifold shifts: the gathers record their requests (G-Planes has no such shift) and the walks run here */
    if (m_config.useManifold && (tech == EVolBRE2D || tech == EVolBRE3D || tech == EDistance))
      answerShiftRequests(photonMap, threadData, scene);
    if (m_config.useManifold && (tech == EBeamBeam1D || tech == EBeamBeam3D_Optimized))
      answerBeamShiftRequests(beamMap, threadData);
    writeBack(gatherBlocks, tech == EDistance);
    float r = 0.f;
    check(gvpm_get_radius(m_h, &r), "gvpm_get_radius");
    /* scaleVolumeAPA(it) ran inside gvpm_gather (not for EDistance, gvpm.cpp:456 vs :1081-1203) */
    if (tech != EDistance) globalScaleVolume = (Float) (r / (m_params.bsphere_radius * 0.01f));
  }

  /* GPMIntegrator::render re-initialises the gather points (gvpm.cpp:272-291) */
  void reset() { check(gvpm_reset(m_h), "gvpm_reset"); }

private:
  /* ------------------------------------------------------------------------------------------ manifold shifts -- */
  /* shiftPhotonManifold, shift_volume_photon.cpp:160-295, split where the data lives: the device gathered, found the
   * shifts that need the walk and recorded their inputs (gvpm_shift_request); the walk (generateShiftPathME + ShiftME,
   * shift/operation/shift_ME.cpp:13-142) and the determinants of the specular manifold (:205-214) run here on Mitsuba's
   * Path objects, exactly as the reference runs them; the device then applies :217-279 to what comes back.          */
  static const uint64_t kShiftRequestCapacity = 1u << 22;
  void answerShiftRequests(const GPhotonMap *photonMap, GPMThreadData &thdata, const Scene *scene) {
    m_requests.resize(kShiftRequestCapacity);
    uint64_t n = 0;
    check(gvpm_download_shift_requests(m_h, m_requests.data(), m_requests.size(), &n), "gvpm_download_shift_requests");
    if (n > m_requests.size()) n = m_requests.size();   /* the surplus was written off as failed shifts by the device */
    gvpm_host_shift none;
    memset(&none, 0, sizeof(none));
    m_hostShifts.assign((size_t) n, none);
    const Medium *medium = nullptr;
    for (const auto &m : scene->getMedia()) medium = m.get();
    for (uint64_t k = 0; k < n; ++k) {
      const gvpm_shift_request &rq = m_requests[k];
      gvpm_host_shift &out = m_hostShifts[k];
      const GPhotonNodeData &d = (*photonMap)[rq.photon].getData();
      const Path &source = *d.lightPath;
      const int c = (int) d.vertexId;
      int b = 0;
      getTypeShift(&source, (size_t) c, b);                          /* shift_utilities.h:112-136: where the chain starts */
      const Point offsetPos(rq.offset_pos[0], rq.offset_pos[1], rq.offset_pos[2]);
      Path proposal;
      PathVertex shiftVertex;                                        /* :171-181 */
      memset(&shiftVertex, 0, sizeof(PathVertex));
      MediumSamplingRecord &cacheMRec = shiftVertex.getMediumSamplingRecord();
      cacheMRec.t = 0.f;
      cacheMRec.p = offsetPos;
      cacheMRec.medium = medium;
      shiftVertex.type = PathVertex::EMediumInteraction;
      shiftVertex.measure = EArea;
      shiftVertex.sampledComponentIndex = -1;
      ShiftRecord sRecME;
      bool ok = generateShiftPathME(source, proposal, (size_t) b, (size_t) c, thdata.pool, thdata.offsetGenerator.get(),
                                    shiftVertex, (Float) rq.radius * m_config.relaxME,
                                    Point(rq.base_point[0], rq.base_point[1], rq.base_point[2]),
                                    Point(rq.shift_point[0], rq.shift_point[1], rq.shift_point[2]));
      ok = ok && ShiftME(sRecME, source, proposal, (size_t) b, (size_t) c, false);
      if (ok) {
        SpecularManifold *manifold = thdata.offsetGenerator->getSpecularManifold();
        out.det_ratio = (float) (manifold->det(proposal, b, c) / manifold->det(source, b, c));   /* :212-214 */
        Float r, g, bl;
        sRecME.throughtput.toLinearRGB(r, g, bl);
        out.throughput[0] = (float) r; out.throughput[1] = (float) g; out.throughput[2] = (float) bl;
        const Vector wi = normalize(proposal.vertex(c - 1)->getPosition() - offsetPos);          /* :227 */
        out.wi[0] = (float) wi.x; out.wi[1] = (float) wi.y; out.wi[2] = (float) wi.z;
        out.pdf = (float) sRecME.pdf;
        Float basePdf = 1.f;                                         /* :256-260, without pdfBaseRay (the device's)  */
        for (int i = b; i < c; ++i) basePdf *= source.vertex(i)->pdf[EImportance] * source.edge(i)->pdf[EImportance];
        out.base_pdf = (float) basePdf;
        out.ok = 1;
      }
      for (int i = b; i <= c; ++i) {                                 /* :187-190, :199-202, :287-290 */
        thdata.pool.release(proposal.edge(i - 1));
        thdata.pool.release(proposal.vertex(i));
      }
    }
    check(gvpm_upload_host_shifts(m_h, m_hostShifts.data(), n), "gvpm_upload_host_shifts");
  }
  /* shiftBeamME, shift_volume_beams.cpp:601-746, host half: BeamGradRadianceQuery::cacheSourcePath (:551-599) rebuilds the
   * source path with vertex c moved to the kernel's place on the beam (request: reserved2 = kRec.v, reserved = the bits of
   * kRec.pdf()), then generateShiftPathME + ShiftME + the determinants (:612-679) as for the photons.  The answer's `wi` is
   * the proposal's last edge WHOLE (not normalised): the device's kernelPDF needs its origin and length (:653-656).     */
  void answerBeamShiftRequests(const LTBeamMap *beamMap, GPMThreadData &thdata) {
    m_requests.resize(kShiftRequestCapacity);
    uint64_t n = 0;
    check(gvpm_download_shift_requests(m_h, m_requests.data(), m_requests.size(), &n), "gvpm_download_shift_requests");
    if (n > m_requests.size()) n = m_requests.size();
    gvpm_host_shift none;
    memset(&none, 0, sizeof(none));
    m_hostShifts.assign((size_t) n, none);
    const auto &beams = beamMap->getBeams();
    for (uint64_t k = 0; k < n; ++k) {
      const gvpm_shift_request &rq = m_requests[k];
      gvpm_host_shift &out = m_hostShifts[k];
      const LTPhotonBeam &beam = beams[rq.photon].second;
      const Path &oriSource = *beam.path;
      const int c = (int) beam.edgeID + 1;
      int b = 0;
      getTypeShift(&oriSource, (size_t) c, b);
      Float kpdf;
      memcpy(&kpdf, &rq.reserved, sizeof(float));
      const Float v = (Float) rq.reserved2;
      /* --- cacheSourcePath, :551-599 */
      Path cachePath;
      cachePath.append(oriSource, 0, c - 1);
      cachePath.append(oriSource.edge(c - 2));
      PathVertex *cacheVertexParent = oriSource.vertex(c - 1)->clone(thdata.pool);
      cachePath.append(cacheVertexParent);
      PathEdge *cacheNewEdge = oriSource.edge(c - 1)->clone(thdata.pool);
      MediumSamplingRecord mRecBeam;
      beam.medium->eval(Ray(beam.getOri(), beam.getDir(), 0.f, v, 0.f), mRecBeam);   /* kRec.beamTrans, beams_struct.h:146-152 */
      cacheNewEdge->length = v;
      cacheNewEdge->pdf[EImportance] = kpdf;
      cacheNewEdge->weight[EImportance] = mRecBeam.transmittance / cacheNewEdge->pdf[EImportance];
      cachePath.append(cacheNewEdge);
      PathVertex *cacheVertexVolume = thdata.pool.allocVertex();
      memset(cacheVertexVolume, 0, sizeof(PathVertex));
      MediumSamplingRecord &cacheMRec = cacheVertexVolume->getMediumSamplingRecord();
      cacheMRec.t = v;
      cacheMRec.time = 0.f;
      cacheMRec.p = beam.getPos(v);
      cacheMRec.medium = beam.medium;
      cacheVertexVolume->type = PathVertex::EMediumInteraction;
      cacheVertexVolume->measure = EArea;
      cacheVertexVolume->sampledComponentIndex = -1;
      cachePath.append(cacheVertexVolume);
      if (oriSource.vertex(c - 1)->measure != EDiscrete)
        cacheVertexParent->pdf[EImportance] *= fastGOp(cachePath, c - 1, c) / fastGOp(oriSource, c - 1, c);
      /* --- the proposal, :612-646 */
      const Point newPos(rq.offset_pos[0], rq.offset_pos[1], rq.offset_pos[2]);
      Path proposal;
      PathVertex shiftVertex;
      memset(&shiftVertex, 0, sizeof(PathVertex));
      MediumSamplingRecord &sMRec = shiftVertex.getMediumSamplingRecord();
      sMRec.t = 0.f;
      sMRec.p = newPos;
      sMRec.medium = beam.medium;
      shiftVertex.type = PathVertex::EMediumInteraction;
      shiftVertex.measure = EArea;
      shiftVertex.sampledComponentIndex = -1;
      ShiftRecord sRecME;
      bool ok = generateShiftPathME(cachePath, proposal, (size_t) b, (size_t) c, thdata.pool, thdata.offsetGenerator.get(),
                                    shiftVertex, (Float) rq.radius * m_config.relaxME,
                                    Point(rq.base_point[0], rq.base_point[1], rq.base_point[2]),
                                    Point(rq.shift_point[0], rq.shift_point[1], rq.shift_point[2]));
      ok = ok && ShiftME(sRecME, cachePath, proposal, (size_t) b, (size_t) c, true);
      if (ok) {
        SpecularManifold *manifold = thdata.offsetGenerator->getSpecularManifold();
        out.det_ratio = (float) (manifold->det(proposal, b, c) / manifold->det(cachePath, b, c));   /* :674-679 */
        Float r, g, bl;
        sRecME.throughtput.toLinearRGB(r, g, bl);
        out.throughput[0] = (float) r; out.throughput[1] = (float) g; out.throughput[2] = (float) bl;
        const Vector wi = proposal.vertex(c - 1)->getPosition() - newPos;   /* = -edge(c-1)->d * edge(c-1)->length */
        out.wi[0] = (float) wi.x; out.wi[1] = (float) wi.y; out.wi[2] = (float) wi.z;
        out.pdf = (float) sRecME.pdf;
        Float basePdf = 1.f;                                         /* :712-716 */
        for (int i = b; i < c; ++i) basePdf *= cachePath.vertex(i)->pdf[EImportance] * cachePath.edge(i)->pdf[EImportance];
        out.base_pdf = (float) basePdf;
        out.ok = 1;
      }
      for (int i = b; i <= c; ++i) {
        thdata.pool.release(proposal.edge(i - 1));
        thdata.pool.release(proposal.vertex(i));
      }
      thdata.pool.release(cacheNewEdge);
      thdata.pool.release(cacheVertexVolume);
      thdata.pool.release(cacheVertexParent);
    }
    check(gvpm_upload_host_shifts(m_h, m_hostShifts.data(), n), "gvpm_upload_host_shifts");
  }
  std::vector<gvpm_shift_request> m_requests;
  std::vector<gvpm_host_shift> m_hostShifts;

  /* ------------------------------------------------------------------------------------------- packed uploads -- */
  /* The per-iteration inputs cross PCIe as packed records from pinned memory (include/gvpm_hip.h "packed uploads" and
   * "compact camera-beam sets": 76 bytes a photon instead of 120; 60 bytes for the beam set of a sensor-adjacent edge of a
   * perspective sensor, 272 for any other set, instead of 320; an asynchronous copy instead of a staged one).  The
   * buffers are reused by the next iteration, which starts after writeBack() has waited for this one's results.
   * Both formats are lossy in the last bits of what they re-derive (the header quantifies it); a host that wants the
   * fp32 records moved as they are sets GVPM_HIP_UPLOAD=soa (every record as flattened) or =packed (photons packed,
   * beam sets as the lossless 272-byte records); the default is `compact`.                                          */
  struct Pinned {
    void *p = nullptr;
    size_t cap = 0;
    void *get(size_t bytes) {
      if (bytes > cap) {
        if (p) gvpm_host_free(p);
        p = nullptr;
        cap = 0;
        if (gvpm_host_alloc(bytes + bytes / 4 + 64, &p) != GVPM_OK) SLog(EError, "gvpm_hip: gvpm_host_alloc failed");
        cap = bytes + bytes / 4 + 64;
      }
      return p;
    }
    ~Pinned() { if (p) gvpm_host_free(p); }
  };
  enum EUploadMode { EUploadSoA = 0, EUploadPacked = 1, EUploadCompact = 2 };
  static EUploadMode uploadMode() {
    const char *e = getenv("GVPM_HIP_UPLOAD");
    if (e && strcmp(e, "soa") == 0) return EUploadSoA;
    if (e && strcmp(e, "packed") == 0) return EUploadPacked;
    return EUploadCompact;
  }
  /* The sensor the compact beam sets are rebuilt from (gvpm_sensor): PerspectiveCamera::sampleRay,
   * src/sensors/perspective.cpp:247-269 with m_cameraToSample of :150-157 -- camera space looks along +z and both film
   * axes are flipped, d_cam ~ ((1 - 2u) tan(xfov/2), (1 - 2v) tan(xfov/2) / aspect, 1): the negated rotation of the
   * camera-to-world transform maps the header's (cx, cy, -1) convention onto it.  Any other sensor, or a cropped
   * film, keeps the full records.                                                                                    */
  void uploadSensor(const Scene *scene) {
    m_haveSensor = false;
    const Sensor *sensor = scene->getSensor();
    if (sensor->getClass()->getName() != "PerspectiveCamera") return;
    const PerspectiveCamera *cam = static_cast<const PerspectiveCamera *>(sensor);
    const Film *film = sensor->getFilm();
    if (film->getCropSize() != film->getSize() || film->getCropOffset() != Point2i(0)) return;
    const Transform trafo = cam->getWorldTransform()->eval(0.f);
    const Matrix4x4 &mtx = trafo.getMatrix();
    gvpm_sensor gs;
    memset(&gs, 0, sizeof(gs));
    const Point o = trafo.transformAffine(Point(0.0f));
    gs.pos[0] = o.x; gs.pos[1] = o.y; gs.pos[2] = o.z;
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) gs.to_world[3 * r + c] = -(double) mtx(r, c);
    gs.tan_half_fov_x = std::tan(0.5 * degToRad((double) cam->getXFov()));
    gs.tan_half_fov_y = gs.tan_half_fov_x / (double) cam->getAspect();
    gs.width = film->getSize().x;
    gs.height = film->getSize().y;
    /* (a scaled or sheared world transform is refused by the library: the sets then stay full records) */
    m_haveSensor = gvpm_upload_sensor(m_h, &gs) == GVPM_OK;
    m_sensor = gs;
  }
  void uploadPhotons() {
    const gvpm_photon_soa &v = m_soa.view();
    if (uploadMode() == EUploadSoA) {
      check(gvpm_upload_photons(m_h, &v), "gvpm_upload_photons");
      return;
    }
    if (m_materials.empty()) m_materials.resize(4096);
    gvpm_photon_packed *dst = (gvpm_photon_packed *) m_pinPhotons.get((size_t) v.n * sizeof(gvpm_photon_packed));
    uint32_t nmat = m_nMaterials;
    if (gvpm_pack_photons(&v, dst, m_materials.data(), (uint32_t) m_materials.size(), &nmat) == GVPM_OK) {
      if (nmat != m_nMaterials) {
        check(gvpm_upload_materials(m_h, m_materials.data(), nmat), "gvpm_upload_materials");
        m_nMaterials = nmat;
      }
      check(gvpm_upload_photons_packed(m_h, dst, v.n), "gvpm_upload_photons_packed");
    } else {
      /* more distinct (reflectance, g) pairs than the table holds -- a textured scene: the fp32 SoA upload */
      check(gvpm_upload_photons(m_h, &v), "gvpm_upload_photons");
    }
  }
  void uploadCameraBeams() {
    const size_t nsets = m_rays.size() / 5;
    const EUploadMode mode = uploadMode();
    if (mode == EUploadSoA) {
      check(gvpm_upload_camera_beams(m_h, m_rays.data(), nsets), "gvpm_upload_camera_beams");
      return;
    }
    gvpm_beam_set_packed *full = (gvpm_beam_set_packed *) m_pinRays.get(nsets * sizeof(gvpm_beam_set_packed));
    if (mode == EUploadCompact && m_haveSensor) {
      gvpm_beam_set_compact *compact = (gvpm_beam_set_compact *) m_pinCompact.get(nsets * sizeof(gvpm_beam_set_compact));
      uint64_t nc = 0, nf = 0;
      m_newIndex.resize(nsets);
      if (gvpm_pack_camera_beams_compact(&m_sensor, m_rays.data(), m_jitter.data(), nsets, compact, &nc, full, &nf,
                                         m_newIndex.data()) == GVPM_OK) {
        check(gvpm_upload_camera_beams_compact(m_h, compact, nc, full, nf), "gvpm_upload_camera_beams_compact");
        /* compact sets first, then the full ones: the G-VPM samples name sets by their index in the upload */
        for (gvpm_vpm_sample &sm : m_samples) sm.set = m_newIndex[sm.set];
        return;
      }
    }
    if (gvpm_pack_camera_beams(m_rays.data(), nsets, full) == GVPM_OK)
      check(gvpm_upload_camera_beams_packed(m_h, full, nsets), "gvpm_upload_camera_beams_packed");
    else
      check(gvpm_upload_camera_beams(m_h, m_rays.data(), nsets), "gvpm_upload_camera_beams");
  }
  Pinned m_pinCompact;
  gvpm_sensor m_sensor;
  bool m_haveSensor = false;
  std::vector<float> m_jitter;          /* 2 per beam set: the base sample's film position minus its pixel */
  std::vector<uint32_t> m_newIndex;
  Pinned m_pinPhotons, m_pinRays;
  std::vector<gvpm_material> m_materials;
  uint32_t m_nMaterials = 0;

  /* ---------------------------------------------------------------------------------------------- SoA storage -- */
  struct Soa {
    std::vector<float> pos, wi, flux, parent_pos, parent_n, prefix_w, parent_scat, parent_wi;
    std::vector<float> parent_pdf, edge_pdf, parent_rr, parent_g;
    std::vector<uint32_t> flags, path_id;
    gvpm_photon_soa v;
    void clear() {
      pos.clear(); wi.clear(); flux.clear(); parent_pos.clear(); parent_n.clear(); prefix_w.clear();
      parent_scat.clear(); parent_wi.clear(); parent_pdf.clear(); edge_pdf.clear(); parent_rr.clear();
      parent_g.clear(); flags.clear(); path_id.clear();
    }
    const gvpm_photon_soa &view() {
      v.pos = pos.data(); v.wi = wi.data(); v.flux = flux.data(); v.parent_pos = parent_pos.data();
      v.parent_n = parent_n.data(); v.prefix_w = prefix_w.data(); v.parent_scat = parent_scat.data();
      v.parent_wi = parent_wi.data(); v.parent_pdf = parent_pdf.data(); v.edge_pdf = edge_pdf.data();
      v.parent_rr = parent_rr.data(); v.parent_g = parent_g.data(); v.flags = flags.data(); v.path_id = path_id.data();
      v.n = flags.size();
      return v;
    }
  };
  static void push3(std::vector<float> &v, const Point &p) { v.push_back((float) p.x); v.push_back((float) p.y); v.push_back((float) p.z); }
  static void push3(std::vector<float> &v, const Vector &p) { v.push_back((float) p.x); v.push_back((float) p.y); v.push_back((float) p.z); }
  static void push3(std::vector<float> &v, const Spectrum &s) {
    Float r, g, b;
    s.toLinearRGB(r, g, b);                                          /* RGB build (SPECTRUM_SAMPLES = 3): identity */
    v.push_back((float) r); v.push_back((float) g); v.push_back((float) b);
  }

  /* The part of a record that describes vertex(c-1), the vertex a photon (c = vertexId) or a beam (c = edgeID + 1)
   * is re-connected from -- what shiftPhotonDiffuse / shiftBeamDiffuse and diffuseReconnection read of it
   * (shift/shift_volume_photon.cpp:382-486, shift/shift_volume_beams.cpp:410-539, shift/operation/shift_diffuse.cpp:11-268). */
  void pushParent(const Path *lt, size_t c) {
    const PathVertex *par = lt->vertex(c - 1);
    const PathEdge *e = lt->edge(c - 1);
    push3(m_soa.parent_pos, par->getPosition());                     /* parentVertex->getPosition(), :389-395      */
    /* geometric normal of a surface / emitter parent (:404-412; area.cpp:132-150 for the emitter's cosine);
     * the closed set is Lambertian: shading frame = geometric frame                                                 */
    push3(m_soa.parent_n, par->isMediumInteraction() ? Vector(0.f) : Vector(par->getGeometricNormal()));
    /* prod_{i < c-1} v_i.weight * v_i.rrWeight * e_i.weight, shift_volume_photon.cpp:415-422                        */
    Spectrum prefix(1.f);
    for (size_t i = 0; i + 1 < c; ++i)
      prefix *= lt->vertex(i)->weight[EImportance] * lt->vertex(i)->rrWeight * lt->edge(i)->weight[EImportance];
    push3(m_soa.prefix_w, prefix);
    /* BSDF::eval of the parent = diffuse reflectance * INV_PI * cos (src/bsdfs/diffuse.cpp:110-127);
     * medium parent: sigma_s * phase (shift_diffuse.cpp:54-70)                                                      */
    Spectrum scat(0.f);
    if (par->isSurfaceInteraction()) {
      const Intersection &its = par->getIntersection();
      scat = its.getBSDF()->getDiffuseReflectance(its);
    } else if (par->isMediumInteraction()) {
      scat = par->getMediumSamplingRecord().sigmaS;
    }
    push3(m_soa.parent_scat, scat);
    /* direction parent -> vertex(c-2): bRec.wi / pRec.wi of the re-evaluated parent (shift_diffuse.cpp:30-70)       */
    push3(m_soa.parent_wi, c >= 3 ? normalize(lt->vertex(c - 2)->getPosition() - par->getPosition()) : Vector(1, 0, 0));
    m_soa.parent_pdf.push_back((float) par->pdf[EImportance]);       /* shift_volume_photon.cpp:463-470, area measure */
    m_soa.edge_pdf.push_back((float) e->pdf[EImportance]);           /* parentEdge->pdf[EImportance]                */
    m_soa.parent_rr.push_back((float) par->rrWeight);                /* shift_diffuse.cpp:111-112                   */
    const int glossy = glossyIndex(par);
    m_soa.parent_g.push_back(par->isMediumInteraction()
                                 ? (float) par->getMediumSamplingRecord().getPhaseFunction()->getMeanCosine()
                                 : (glossy >= 0 ? (float) glossy : 0.f));   /* hg.cpp:112-114; a glossy surface parent:
                                                                               its entry in the table of gvpm_upload_bsdfs */
  }

  /* A surface parent whose BSDF is in the device's table of glossy BSDFs (include/gvpm_hip.h, gvpm_bsdf).  Returns its index
   * (appending the entry on first sight), -1 for every other vertex.
   * Phong (src/bsdfs/phong.cpp), not textured: sampled with BOTH components (sampledComponentIndex == -1: what
   * PathVertex::sampleNext records when Phong::sampleComponent finds the lobe's roughness >= 0.05, vertex.cpp:160-165,
   * phong.cpp:311-330) or -- round 5 -- through ONE of them (0 specular, 1 diffuse): an entry per (BSDF, component), its
   * `distribution` field = component + 1 (include/gvpm_hip.h).  The exponent and the sampling weight have no getters: they are read back through getRoughness =
   * sqrt(2 / (2 + exponent)) (phong.cpp:293-300) and pdfComponent(component 0) = m_specularSamplingWeight (:332-343).
   * RoughConductor (src/bsdfs/roughconductor.cpp), not textured, isotropic Beckmann or GGX, whose Properties name `eta` and `k`
   * themselves (a `material` preset keeps its spectra in protected members: such a surface stays outside the closed set):
   * m_eta = eta / extEta, m_k = k / extEta (:181-191), alpha and the distribution through MicrofacetDistribution(props) as the
   * constructor reads them (:193-201).                                                                                    */
  int glossyIndex(const PathVertex *par) {
    if (!par->isSurfaceInteraction()) return -1;
    const Intersection &its = par->getIntersection();
    const BSDF *bsdf = its.getBSDF();
    const std::string cls = bsdf->getClass()->getName();
    if (bsdf->getType() & BSDF::ESpatiallyVarying) return -1;
    if (cls != "Phong" && cls != "RoughConductor" && cls != "Ward") return -1;
    /* Ward (src/bsdfs/ward.cpp, round 5): isotropic (no EAnisotropic component: alphaU == alphaV, ward.cpp:144-146) and sampled with
     * both components (roughness alpha >= 0.05, :370-376); alpha through getRoughness (:360-368), the sampling weight through
     * pdfComponent(component 0) (:391-402), the model variant from the plugin's own property (default "balanced", :103-113) */
    if (cls == "Ward" && ((bsdf->getType() & BSDF::EAnisotropic) || par->sampledComponentIndex != -1)) return -1;
    const int component = cls == "Phong" ? (int) par->sampledComponentIndex : -1;
    if (component < -1 || component > 1) return -1;
    const std::pair<const BSDF *, int> key(bsdf, component);
    auto found = m_bsdfIndex.find(key);
    if (found != m_bsdfIndex.end()) return (int) found->second;
    gvpm_bsdf b;
    memset(&b, 0, sizeof(b));
    Float cr, cg, cb;
    bsdf->getSpecularReflectance(its).toLinearRGB(cr, cg, cb);
    b.specular[0] = (float) cr; b.specular[1] = (float) cg; b.specular[2] = (float) cb;
    if (cls == "Phong") {
      b.kind = GVPM_BSDF_PHONG;
      const Float rough = bsdf->getRoughness(its, 0);
      b.exponent = (float) (2.0 / ((double) rough * (double) rough) - 2.0);
      BSDFSamplingRecord bRec(its, its.wi, its.wi, EImportance);
      bRec.component = 0;
      b.specular_sampling_weight = (float) bsdf->pdfComponent(bRec);
      b.distribution = component + 1;   /* 0: both components; 1: the specular lobe alone; 2: the diffuse one alone */
    } else if (cls == "Ward") {
      b.kind = GVPM_BSDF_WARD;
      b.exponent = (float) bsdf->getRoughness(its, 0);
      BSDFSamplingRecord bRec(its, its.wi, its.wi, EImportance);
      bRec.component = 0;
      b.specular_sampling_weight = (float) bsdf->pdfComponent(bRec);
      std::string variant = bsdf->getProperties().getString("variant", "balanced");
      for (char &ch : variant) ch = (char) std::tolower((unsigned char) ch);   /* (boost::to_lower_copy there) */
      b.sample_visible = variant == "ward" ? GVPM_WARD_WARD : variant == "ward-duer" ? GVPM_WARD_DUER : GVPM_WARD_BALANCED;
    } else {
      const Properties &props = bsdf->getProperties();
      if (!props.hasProperty("eta") || !props.hasProperty("k")) return -1;
      MicrofacetDistribution distr(props);
      if (!distr.isIsotropic() || (distr.getType() != MicrofacetDistribution::EBeckmann && distr.getType() != MicrofacetDistribution::EGGX))
        return -1;
      b.kind = GVPM_BSDF_ROUGHCONDUCTOR;
      b.exponent = (float) distr.getAlphaU();
      b.distribution = distr.getType() == MicrofacetDistribution::EGGX ? GVPM_MICROFACET_GGX : GVPM_MICROFACET_BECKMANN;
      b.sample_visible = distr.getSampleVisible() ? 1 : 0;
      const Float extEta = lookupIOR(props, "extEta", "air");
      const Spectrum eta = props.getSpectrum("eta") / extEta, k = props.getSpectrum("k") / extEta;
      eta.toLinearRGB(cr, cg, cb);
      b.eta[0] = (float) cr; b.eta[1] = (float) cg; b.eta[2] = (float) cb;
      k.toLinearRGB(cr, cg, cb);
      b.k[0] = (float) cr; b.k[1] = (float) cg; b.k[2] = (float) cb;
    }
    const uint32_t idx = (uint32_t) m_bsdfs.size();
    m_bsdfs.push_back(b);
    m_bsdfIndex[key] = idx;
    m_bsdfsDirty = true;
    return (int) idx;
  }
  std::map<std::pair<const BSDF *, int>, uint32_t> m_bsdfIndex;   /* (BSDF, sampled component) -> table entry */
  std::vector<gvpm_bsdf> m_bsdfs;
  bool m_bsdfsDirty = false;

  /* flags: parent type, the result of getTypeShift (a pure function of the light path, shift/shift_utilities.h:112-136),
   * whether edge(c-1) lies in the medium, depth = c - 1, getVertexComponentType(parent) (shift_utilities.h:222-231).
   * A parent outside the device's closed set (a surface that is neither Lambertian nor an untextured Phong sampled with both
   * components -- glossyIndex() --, a heterogeneous medium) is flagged `invalid`:
   * the shift then fails with w = 1, as the reference's non-invertible shifts do.                                    */
  uint32_t makeFlags(const Path *lt, size_t c, size_t depth) const {
    const PathVertex *par = lt->vertex(c - 1);
    int b = -1;
    const ELightShiftType t = getTypeShift(lt, c, b);
    uint32_t st = t == EDiffuseShift ? 1u : t == EMediumShift ? 2u : t == EManifoldShift ? 3u : 0u;
    const BSDF *parBsdf = par->isSurfaceInteraction() ? par->getIntersection().getBSDF() : nullptr;
    /* (pushParent ran first: the entry of this vertex's (BSDF, sampled component) exists if the BSDF is in the closed set;
     * a rough conductor has one component and one entry, keyed with -1) */
    const bool glossy = parBsdf != nullptr &&
                        (m_bsdfIndex.count(std::make_pair(parBsdf, (int) par->sampledComponentIndex)) != 0 ||
                         m_bsdfIndex.count(std::make_pair(parBsdf, -1)) != 0);
    if (st == 1u || st == 2u) {
      if (par->isSurfaceInteraction() && !glossy) {
        const BSDF *bsdf = par->getIntersection().getBSDF();
        if (!(bsdf->getType() & BSDF::EDiffuseReflection) || (bsdf->getType() & ~(BSDF::EDiffuseReflection | BSDF::EFrontSide)) != 0)
          st = 0u;
      } else if (par->isMediumInteraction() && !par->getMediumSamplingRecord().medium->isHomogeneous()) {
        st = 0u;
      }
    }
    const uint32_t ptype = par->isEmitterSample() ? GVPM_PARENT_EMITTER
                         : par->isSurfaceInteraction() ? (glossy ? GVPM_PARENT_SURFACE_BSDF : GVPM_PARENT_SURFACE)
                                                       : GVPM_PARENT_MEDIUM;
    return GVPM_PF_MAKE(ptype, st, lt->edge(c - 1)->medium != nullptr, depth, getVertexComponentType(par));
  }

  /* ------------------------------------------------------------------------------------------- photon map ---- */
  /* One record per node of the volume photon map, in kd-tree storage order (the device builds its own grid):
   * GPhotonNodeData {vertexId, lightPath, weight, pathID}, gvpm_accel.h:17-65, filled by GPhotonMap::tryAppend (:119-199). */
  void flattenPhotons(const GPhotonMap *map) {
    for (size_t k = 0; k < map->size(); ++k) {
      const GPhotonNodeData &d = (*map)[k].getData();
      const Path *lt = d.lightPath;
      const size_t c = d.vertexId;
      push3(m_soa.pos, lt->vertex(c)->getPosition());                /* GPhotonNodeKD position = photon.its.p, :77-84 */
      push3(m_soa.wi, -lt->edge(c - 1)->d);                          /* its.wi = -edge->d, :49-51                  */
      push3(m_soa.flux, d.weight);                                   /* importanceWeights at the append, :134-148  */
      pushParent(lt, c);
      m_soa.flags.push_back(makeFlags(lt, c, c - 1));                /* depth = vID - 1, :43                       */
      m_soa.path_id.push_back((uint32_t) d.pathID);                  /* m_nbLightPathAdded, :165,189-191           */
    }
  }

  /* ----------------------------------------------------------------------------------------- photon beams ---- */
  /* One record per LTPhotonBeam (gvpm_beams.h:18-43): edge i of a light path, origin vertex(i), end vertex(i+1). */
  void pushBeam(const LTPhotonBeam &bm) {
    const Path *lt = bm.path;
    const size_t i = bm.edgeID;
    push3(m_soa.pos, lt->vertex(i + 1)->getPosition());              /* setEndPoint(lastVertex->getPosition()), :35-36 */
    push3(m_soa.wi, -lt->edge(i)->d);
    push3(m_soa.flux, bm.flux);                                      /* without the transmittance of edge i, :26-33 */
    pushParent(lt, i + 1);                                           /* parent = vertex(i): the beam's origin       */
    m_soa.flags.push_back(makeFlags(lt, i + 1, i));                  /* depth = edgeID (PhotonBeam ctor's last argument) */
    m_soa.path_id.push_back((uint32_t) bm.pathID);
    const PathVertex *end = lt->vertex(i + 1);                       /* geometric normal of a surface end point:
                                                                        pdfBasePos conversion, shift_volume_beams.cpp:502-513 */
    push3(m_endN, end->isSurfaceInteraction() ? Vector(end->getGeometricNormal()) : Vector(0.f));
  }
  void flattenBeams(const LTBeamMap *map) {
    m_endN.clear();
    for (const auto &pb : map->getBeams()) pushBeam(pb.second);     /* std::vector<std::pair<int, T>>, beams.h:332-334 */
  }

  /* ---------------------------------------------------------------------------------------- photon planes ---- */
  /* gvpm.cpp:790-797: every beam becomes a plane by LTPhotonPlane::transformBeam (gvpm_plane.h:53-73) -- kept on the
   * host, serial, with the reference's sampler, so the second edges are the reference's.                           */
  void flattenPlanes(const LTBeamMap *map, Sampler *sampler) {
    m_endN.clear(); m_w1.clear(); m_len1.clear();
    for (const auto &pb : map->getBeams()) {
      const LTPhotonPlane pl = LTPhotonPlane::transformBeam(pb.second, sampler);
      pushBeam(pb.second);                                           /* ori, w0 * length0, flux, edgeID: the beam's  */
      push3(m_w1, pl.w1());                                          /* PhotonPlane::w1(), plane_struct.h:205          */
      m_len1.push_back((float) pl.length1());                        /* PhotonPlane::length1(), plane_struct.h:217     */
    }
  }

  /* ------------------------------------------------------------------------------------------ camera beams --- */
  /* The SVertexPDF cache entries of edge e of a (base or shifted) gather point, gvpm_struct.h:361-370,585-631.      */
  static void fillRay(gvpm_camera_ray &r, const GatherPoint &gp, size_t e, bool valid) {
    memset(&r, 0, sizeof(r));
    r.info = GVPM_RAY_INFO(valid ? 1 : 0, e);
    if (!valid) return;
    const PathEdge *ed = gp.path.edge(e);
    const Point o = gp.path.vertex(e)->getPosition();                /* gvpm.cpp:1032-1038, shift_volume_photon.cpp:765-770 */
    const Vector d = -ed->d;                                         /* ERadiance edges point at the camera, edge.cpp:78-81 */
    r.o[0] = (float) o.x; r.o[1] = (float) o.y; r.o[2] = (float) o.z;
    r.d[0] = (float) d.x; r.d[1] = (float) d.y; r.d[2] = (float) d.z;
    r.len = (float) ed->length;
    Float cr, cg, cb;
    (gp.getWeightBeam(e - 1) * gp.getWeightVertex(e)).toLinearRGB(cr, cg, cb);  /* eyeContrib, shift_volume_photon.cpp:741-745 */
    r.eye[0] = (float) cr; r.eye[1] = (float) cg; r.eye[2] = (float) cb;
    r.pdf = (float) gp.getVertexInfo(e).pdf;                         /* sensorMIS operands, gvpm_struct.h:608-631     */
    r.jacobian = (float) gp.getVertexInfo(e).jacobian;
    r.gop = (float) gp.GOp(e);
  }

  /* One beam set per medium edge of every gather point, base + the four offset pixels in EPixel order
   * (gvpm_struct.h:354-359; generateOffsetPos, shift_utilities.h:255-261).  The shifted gather points are generated
   * EAGERLY here (the functors generate them lazily on their first hit, shift_volume_photon.cpp:543,761).
   * vpm: also the camera samples of computeVolumeGradientPhoton (gvpm.cpp:1117-1172).                              */
  void flattenCameraBeams(Scene *scene, std::vector<std::vector<GatherPoint>> &blocks, GPMThreadData &td, Sampler *sampler,
                          bool vpm) {
    m_rays.clear();
    m_samples.clear();
    m_jitter.clear();
    const GPMConfig &cfg = m_config;
    for (auto &block : blocks) {
      for (GatherPoint &gp : block) {
        std::vector<size_t> mediumEdges;
        std::vector<Float> selWeight;
        Spectrum weightBeam(1.f);
        for (size_t e = 1; e < gp.path.edgeCount(); ++e) {
          if (!vpm) {
            if (cfg.minCameraDepth > e) continue;                     /* gvpm.cpp:1020-1025 (BRE), :921-926 (beams)  */
            if (cfg.maxCameraDepth != -1 && (int) e > cfg.maxCameraDepth + 1) break;
          }
          if (gp.path.edge(e)->medium != nullptr) {
            mediumEdges.push_back(e);
            selWeight.push_back(weightBeam.max());                    /* selBeam.append(weightBeam.max()), gvpm.cpp:1119-1122 */
          }
          weightBeam *= gp.path.vertex(e + 1)->weight[EImportance];   /* :1127-1128                                   */
          weightBeam *= gp.path.edge(e)->weight[EImportance];
        }
        if (mediumEdges.empty()) continue;
        gp.haveSmoke = true;                                          /* gvpm.cpp:1030                                */
        std::vector<ShiftGatherPoint> shiftGPs(4);
        const Point2 basePixel = gp.path.vertex(1)->getSamplePosition();
        const std::array<Point2, 4> pixels = generateOffsetPos(basePixel);
        for (int i = 0; i < 4; ++i) shiftGPs[i].generate(scene, td.pool, gp, pixels[i], false);
        const size_t firstSet = m_rays.size() / 5;
        for (size_t e : mediumEdges) {
          gvpm_camera_ray set[5];
          fillRay(set[0], gp, e, true);
          set[0].rand = vpm ? 0.f : (float) sampler->next1D();        /* bre->query(..., sampler->next1D()), gvpm.cpp:1042;
                                                                         G-Beams: the key of the per-hit Philox stream */
          set[0].pixel = (uint32_t) gp.pixel.x | ((uint32_t) gp.pixel.y << 16);
          for (int i = 0; i < 4; ++i)
            fillRay(set[1 + i], shiftGPs[i], e, shiftGPs[i].validVolumeEdge(e, gp.path.edge(e)->medium));
          m_rays.insert(m_rays.end(), set, set + 5);
          /* the film position the base path was sampled at, relative to its pixel: the offset paths keep the fractional
           * part (src/libbidir/vertex.cpp:345-346), which is all a compact set carries of the five directions            */
          m_jitter.push_back((float) (basePixel.x - (Float) gp.pixel.x));
          m_jitter.push_back((float) (basePixel.y - (Float) gp.pixel.y));
        }
        if (vpm) {
          /* DiscreteDistribution selBeam; normalize(); per sample sampleReuse(randSample) (gvpm.cpp:1130,1143-1150) */
          DiscreteDistribution selBeam(selWeight.size());
          for (Float w : selWeight) selBeam.append(w);
          selBeam.normalize();
          const Float normalization = 1.f / cfg.nbCameraSamples;
          for (int s = 0; s < cfg.nbCameraSamples; ++s) {
            Float randSample = sampler->next1D();
            if (cfg.stratified) randSample = s * normalization + randSample * normalization;
            const size_t sampleIndex = selBeam.sampleReuse(randSample);
            const size_t e = mediumEdges[sampleIndex];
            if (cfg.maxCameraDepth != -1 && (int) e > cfg.maxCameraDepth + 1) continue;   /* :1153-1158 */
            if (cfg.minCameraDepth != 0 && e < cfg.minCameraDepth + 1) continue;
            gvpm_vpm_sample sm;
            sm.set = (uint32_t) (firstSet + sampleIndex);
            sm.rand = (float) randSample;                              /* drives sampleDistance(EDistanceAlwaysValid), :1168 */
            sm.pdf_sel = (float) selBeam[sampleIndex];                 /* gRec.changeEdge(idEdge, selBeam[sampleIndex]), :1160 */
            sm.reserved = 0;
            m_samples.push_back(sm);
          }
        }
        for (auto &sgp : shiftGPs) sgp.path.release(td.pool);         /* gvpm.cpp:1071-1074                           */
      }
    }
  }

  /* G-VPM keeps GatherPoint::scaleVol / NVol on the device from gvpm_reset on (initialScaleVolume, 0: what
   * GVPMRadiusInitializer::init writes, gvpm_gatherpoint.h:176-260); they are mirrored back after every pass. */
  void uploadVpmState(std::vector<std::vector<GatherPoint>> &) {}

  /* ---------------------------------------------------------------------------------------------- results ---- */
  /* gp.mediumFlux / shiftedMediumFlux[4] / weightedMediumFlux[4] (gvpm_struct.h:429-441) <- the 27 accumulators;
   * G-VPM also scaleVol / NVol (gvpm.cpp:1191-1195).                                                              */
  void writeBack(std::vector<std::vector<GatherPoint>> &blocks, bool vpm) {
    const size_t P = (size_t) m_params.width * m_params.height;
    m_accum.resize(P * GVPM_ACCUM_FLOATS);
    check(gvpm_download_accum(m_h, m_accum.data()), "gvpm_download_accum");
    if (vpm) {
      m_scaleVol.resize(P);
      m_nVol.resize(P);
      check(gvpm_download_vpm_state(m_h, m_scaleVol.data(), m_nVol.data()), "gvpm_download_vpm_state");
    }
    for (auto &block : blocks)
      for (GatherPoint &gp : block) {
        const size_t px = (size_t) gp.pixel.y * m_params.width + gp.pixel.x;
        const float *a = &m_accum[px * GVPM_ACCUM_FLOATS];
        gp.mediumFlux.fromLinearRGB(a[0], a[1], a[2]);
        for (int k = 0; k < 4; ++k) {
          gp.shiftedMediumFlux[k].fromLinearRGB(a[3 + 3 * k], a[4 + 3 * k], a[5 + 3 * k]);
          gp.weightedMediumFlux[k].fromLinearRGB(a[15 + 3 * k], a[16 + 3 * k], a[17 + 3 * k]);
        }
        if (vpm) {
          gp.scaleVol = m_scaleVol[px];
          gp.NVol = m_nVol[px];
        }
      }
  }

  /* ------------------------------------------------------------------------------------- scene and medium ---- */
  /* scene->rayIntersect(Ray) of the shifts' visibility tests (shift_volume_photon.cpp:398, shift_volume_beams.cpp:421):
   * every triangle of every TriMesh of the scene, world space.                                                    */
  void uploadScene(const Scene *scene) {
    std::vector<float> v0, e1, e2;
    for (const auto &shape : scene->getShapes()) {
      const TriMesh *mesh = dynamic_cast<const TriMesh *>(shape.get());
      if (!mesh) {
        ref<TriMesh> tess = const_cast<Shape *>(shape.get())->createTriMesh();   /* analytic shapes: their tessellation */
        mesh = tess.get();
        if (!mesh) SLog(EError, "gvpm_hip: shape '%s' has no triangle representation", shape->getName().c_str());
        appendMesh(mesh, v0, e1, e2);
        continue;
      }
      appendMesh(mesh, v0, e1, e2);
    }
    gvpm_triangles t;
    t.v0 = v0.data(); t.e1 = e1.data(); t.e2 = e2.data();
    t.n = (uint32_t) (v0.size() / 3);
    check(gvpm_upload_scene(m_h, &t), "gvpm_upload_scene");
  }
  static void appendMesh(const TriMesh *mesh, std::vector<float> &v0, std::vector<float> &e1, std::vector<float> &e2) {
    const Point *pos = mesh->getVertexPositions();
    const Triangle *tri = mesh->getTriangles();
    for (size_t k = 0; k < mesh->getTriangleCount(); ++k) {
      const Point &a = pos[tri[k].idx[0]], &b = pos[tri[k].idx[1]], &c = pos[tri[k].idx[2]];
      push3(v0, a);
      push3(e1, b - a);
      push3(e2, c - a);
    }
  }

  /* The one homogeneous medium of the scene (the reference's own restriction: `Medium *m_smoke`, computeOnlyVolume-
   * Interaction sets its sampling weight to 1, gvpm.cpp:135-142): sigma_a, sigma_s, sigma_t = sigma_a + sigma_s
   * (src/medium/homogeneous.cpp:170-200), the phase function's mean cosine (isotropic: 0).                         */
  void uploadMedium(const Scene *scene) {
    const Medium *med = nullptr;
    for (const auto &m : scene->getMedia()) med = m.get();
    if (!med) SLog(EError, "gvpm_hip: the scene has no participating medium");
    if (!med->isHomogeneous()) SLog(EError, "gvpm_hip: heterogeneous media stay on the CPU path (useHip=false)");
    gvpm_medium gm;
    memset(&gm, 0, sizeof(gm));
    Float r, g, b;
    med->getSigmaA().toLinearRGB(r, g, b);
    gm.sigma_a[0] = (float) r; gm.sigma_a[1] = (float) g; gm.sigma_a[2] = (float) b;
    med->getSigmaS().toLinearRGB(r, g, b);
    gm.sigma_s[0] = (float) r; gm.sigma_s[1] = (float) g; gm.sigma_s[2] = (float) b;
    med->getSigmaT().toLinearRGB(r, g, b);
    gm.sigma_t[0] = (float) r; gm.sigma_t[1] = (float) g; gm.sigma_t[2] = (float) b;
    gm.g = (float) med->getPhaseFunction()->getMeanCosine();
    gm.medium_sampling_weight = 1.f;                                  /* computeOnlyVolumeInteraction(), gvpm.cpp:135-142 */
    check(gvpm_upload_medium(m_h, &gm), "gvpm_upload_medium");
  }

  /* Mitsuba's SLog(EError, ...) throws: a negative status is raised exactly where the reference would have raised   */
  void check(int rc, const char *what) const {
    if (rc < 0) SLog(EError, "gvpm_hip: %s failed (%d): %s", what, rc, m_h ? gvpm_last_error(m_h) : "no handle");
  }

  gvpm_context *m_h;
  gvpm_params m_params;
  GPMConfig m_config;
  Soa m_soa;
  std::vector<float> m_endN, m_w1, m_len1, m_accum, m_scaleVol, m_nVol;
  std::vector<gvpm_camera_ray> m_rays;
  std::vector<gvpm_vpm_sample> m_samples;
};

MTS_NAMESPACE_END
