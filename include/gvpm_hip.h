/*
 * gvpm_hip.h -- C ABI of libgvpm_hip.so: the MI355X (gfx950) photon-gather +
 * gradient-domain shift path of the `gvpm` integrator.
 *
 * This is the seam a Mitsuba-side `gvpm` Integrator shim binds instead of the
 * `switch (m_config.volTechnique)` at
 *   src/integrators/photonmapper/gvpm/gvpm.cpp:456-474
 * (computeVolumeGradientPhoton / ...PhotonBRE / ...Beams / ...Planes), and of
 * the per-pixel assembly that follows it (gvpm.cpp:480-532, 1205-1306).
 * All citations below are relative to the reference tree
 * (gvpm/ = src/integrators/photonmapper/gvpm/).
 *
 * Conventions
 *  - plain C, plain pointers and sizes, no torch / HIP types in any signature;
 *  - every call returns int: 0 = GVPM_OK, negative = gvpm_status; the text of
 *    the last failure of a handle is available through gvpm_last_error();
 *    nothing throws or aborts (Mitsuba's SLog(EError) throws; a shim converts);
 *  - the caller owns every host buffer passed in; the library copies during
 *    the call (uploads are synchronous with respect to the host buffer), with
 *    one exception: a buffer in PINNED host memory (gvpm_host_alloc*) is copied
 *    asynchronously on the handle's copy stream and must stay untouched until
 *    the gvpm_gather that consumes it has returned (G-BRE) or, for the other
 *    techniques, until gvpm_synchronize / a download has returned;
 *  - a handle is single-owner (call it from the RenderJob thread only);
 *    kernels are stream-ordered; gvpm_download_* block until results are ready;
 *  - the `_dev` variants take DEVICE pointers (already resident in HBM, e.g.
 *    a torch tensor's data_ptr()) and do no PCIe traffic.  They are BORROWED:
 *    kernels of up to three consecutive steps are in flight at a time (build of
 *    N+2, traversal of N+1, evaluation of N), so a borrowed buffer must stay
 *    untouched until gvpm_synchronize has returned or the SECOND gvpm_gather
 *    after the one that consumed it has returned.
 *
 * All floating-point payload is fp32 (the reference's SCons default
 * SINGLE_PRECISION build, build/config-linux-gcc.py:7); the device computes in
 * fp32.
 */
#ifndef GVPM_HIP_H
#define GVPM_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GVPM_ABI_VERSION 3  /* 2: gvpm_bsdf grew to 64 bytes (rough conductor), round 4; 3: gvpm_devgen_scene carries the
                              sensor's rotation (cam_to_world), round 5 */

/* ---- status codes --------------------------------------------------------*/
typedef enum gvpm_status {
  GVPM_OK = 0,
  GVPM_ERR_INVALID_ARG = -1,  /* null pointer, bad enum, inconsistent sizes   */
  GVPM_ERR_NO_DEVICE = -2,    /* no gfx950 device / HIP runtime failure       */
  GVPM_ERR_HIP = -3,          /* a HIP call failed (see gvpm_last_error)      */
  GVPM_ERR_STATE = -4,        /* call order violated (e.g. gather w/o photons)*/
  GVPM_ERR_UNSUPPORTED = -5,  /* flag combination the reference itself rejects
                                 (GPMConfig::load SLog(EError) cases,
                                 gvpm/gvpm_struct.h:291-331) or a branch that
                                 is host-only (manifold shift)                */
  GVPM_ERR_COMM = -6          /* RCCL failure                                 */
} gvpm_status;

/* ---- enums mirrored from the reference -----------------------------------*/
/* EVolumeTechnique, src/integrators/volume_utils.h:12-21 (same order).      */
typedef enum gvpm_technique {
  GVPM_VOL_BRE2D = 0,
  GVPM_VOL_BRE3D = 1,
  GVPM_DISTANCE = 2,          /* G-VPM, 3D point kernel                       */
  GVPM_BEAM_BEAM_1D = 3,
  GVPM_BEAM_BEAM_3D_NAIVE = 4,
  GVPM_BEAM_BEAM_3D_EGSR = 5,
  GVPM_BEAM_BEAM_3D_OPTIMIZED = 6,
  GVPM_VOL_PLANE0D = 7
} gvpm_technique;

/* EPixel, gvpm/gvpm_struct.h:354-359; offsets from generateOffsetPos,
 * gvpm/shift/shift_utilities.h:255-261: L=(-1,0) R=(+1,0) T=(0,+1) B=(0,-1). */
enum { GVPM_LEFT = 0, GVPM_RIGHT = 1, GVPM_TOP = 2, GVPM_BOTTOM = 3 };

/* ELightShiftType, gvpm/gvpm_struct.h:30-37 (values kept).                   */
enum {
  GVPM_SHIFT_ALL = 0,
  GVPM_SHIFT_DIFFUSE = 1 << 1,
  GVPM_SHIFT_MANIFOLD = 1 << 2,
  GVPM_SHIFT_NULL = 1 << 4,
  GVPM_SHIFT_MEDIUM = 1 << 5,
  GVPM_SHIFT_INVALID = 1 << 6
};

/* ELightingEffects, src/integrators/volume_utils.h:94-103.                   */
enum {
  GVPM_SURF2SURF = 1 << 1,
  GVPM_SURF2MEDIA = 1 << 2,
  GVPM_MEDIA2SURF = 1 << 3,
  GVPM_MEDIA2MEDIA = 1 << 4
};

/* BSDF::EBSDFType bits used by bsdfInteractionMode
 * (include/mitsuba/render/bsdf.h; parse in volume_utils.h:120-151).          */
enum {
  GVPM_BSDF_DIFFUSE_REFLECTION = 0x00002,
  GVPM_BSDF_ALL = 0x1FFFF
};

/* ---- photon record flags (one uint32 per photon) -------------------------*/
/* parent = light-path vertex (vertexId-1), the vertex the photon is
 * re-connected from (gvpm/shift/shift_volume_photon.cpp:389-390).            */
#define GVPM_PARENT_EMITTER 0u  /* PathVertex::EEmitterSample (area light)    */
#define GVPM_PARENT_SURFACE 1u  /* ESurfaceInteraction, Lambertian closed set */
#define GVPM_PARENT_MEDIUM 2u   /* EMediumInteraction                         */
#define GVPM_PARENT_SURFACE_BSDF 3u /* ESurfaceInteraction whose BSDF is entry
                                   (uint32_t) parent_g of the table of
                                   gvpm_upload_bsdfs (a glossy parent, round 4);
                                   parent_scat = its diffuse reflectance       */
#define GVPM_PF_PARENT_TYPE(f) ((f) & 3u)
/* result of getTypeShift() (gvpm/shift/shift_utilities.h:112-136) -- a pure
 * function of the light path, evaluated by the host at flattening time:
 * 0 invalid, 1 diffuse, 2 medium, 3 manifold                                 */
#define GVPM_PF_SHIFT_TYPE(f) (((f) >> 2) & 7u)
#define GVPM_PF_EDGE_IN_MEDIUM(f) (((f) >> 5) & 1u) /* edge(vertexId-1)->medium != 0 */
#define GVPM_PF_DEPTH(f) (((f) >> 8) & 0xFFu)        /* vertexId - 1            */
/* getVertexComponentType(vPrev) (shift_utilities.h:219-229), low 16 bits of
 * BSDF::EBSDFType                                                            */
#define GVPM_PF_PREV_COMPONENT(f) (((f) >> 16) & 0xFFFFu)
#define GVPM_PF_MAKE(parent, shift, edge_medium, depth, comp)                  \
  (((uint32_t)(parent) & 3u) | (((uint32_t)(shift) & 7u) << 2) |               \
   (((uint32_t)(edge_medium) & 1u) << 5) | (((uint32_t)(depth) & 0xFFu) << 8) | \
   (((uint32_t)(comp) & 0xFFFFu) << 16))

/* ---- configuration: the GPMConfig flags the kernels read -----------------*/
/* gvpm/gvpm_struct.h:107-333 (same names, snake_case).                       */
typedef struct gvpm_params {
  int32_t abi_version;          /* = GVPM_ABI_VERSION                          */
  int32_t width, height;        /* film crop size                              */
  int32_t vol_technique;        /* gvpm_technique                              */
  int32_t max_depth;            /* maxDepth; <= 0: no limit (the functors test
                                   `config.maxDepth > 0`)                      */
  int32_t min_depth;            /* minDepth                                    */
  int32_t use_mis;              /* useMIS: 1 "area", 0 "none"                  */
  int32_t use_shift_null;       /* useShiftNull                                */
  int32_t path_set;             /* pathSet                                     */
  int32_t power_heuristic;      /* powerHeuristic                              */
  int32_t no_medium_shift;      /* noMediumShift (must be 1: shiftPhotonMedium
                                   is SAssert(false), shift_volume_photon.cpp:307) */
  int32_t use_manifold;         /* useManifold: manifold shifts are host-only;
                                   the device treats them as the reference does
                                   with useManifold=false (failed shift)       */
  int32_t debug_shift;          /* debugShift, ELightShiftType value           */
  int32_t lighting_interaction_mode; /* ELightingEffects bits                 */
  int32_t bsdf_interaction_mode;     /* BSDF type mask, GVPM_BSDF_ALL = "all"  */
  int32_t nb_camera_samples;    /* nbCameraSamples (G-VPM)                     */
  int32_t visibility_as_written;/* 1: shadow ray maxt = lProj*ShadowEpsilon as
                                   written at shift_volume_photon.cpp:396;
                                   0: lProj*(1-ShadowEpsilon)                  */
  float alpha;                  /* alpha (radius reduction)                    */
  float initial_scale_volume;   /* initialScaleVolume                          */
  float bsphere_radius;         /* m_smokeAABB.getBSphere().radius             */
  float epsilon;                /* Epsilon, include/mitsuba/core/constants.h   */
  float shadow_epsilon;         /* ShadowEpsilon                               */
  int32_t reserved[8];
} gvpm_params;

/* ---- medium (homogeneous, "balance" strategy) ----------------------------*/
/* src/medium/homogeneous.cpp:432-513, src/phase/{isotropic,hg}.cpp.          */
typedef struct gvpm_medium {
  float sigma_a[3];
  float sigma_s[3];
  float sigma_t[3];             /* must be equal across channels (:196-200)    */
  float g;                      /* HG mean cosine, 0 = isotropic               */
  float medium_sampling_weight; /* 1 after computeOnlyVolumeInteraction()      */
  float reserved[5];
} gvpm_medium;

/* ---- glossy surface parents ------------------------------*/
/* diffuseReconnection re-evaluates the BSDF of the vertex a photon is re-connected from towards the offset position
 * (gvpm/shift/operation/shift_diffuse.cpp:25-47: BSDF::eval, BSDF::pdf * pdfComponent with
 * bRec.component = parent->sampledComponentIndex).  The device's closed set: the Lambertian of GVPM_PARENT_SURFACE
 * (src/bsdfs/diffuse.cpp:110-127) and the BSDFs of this table, named by photons of parent type
 * GVPM_PARENT_SURFACE_BSDF through parent_g (an index, stored as a float).
 *   GVPM_BSDF_PHONG  the modified Phong model of src/bsdfs/phong.cpp with BOTH components, i.e. a vertex whose
 *     sampledComponentIndex is -1 -- what PathVertex::sampleNext records for every Phong surface of roughness
 *     sqrt(2 / (2 + exponent)) >= 0.05 (exponent <= 798; vertex.cpp:160-165 with Phong::sampleComponent, phong.cpp:311-330):
 *       eval(wi, wo) = (specular (exponent + 2) / (2 pi) alpha^exponent [alpha > 0] + diffuse / pi) cos(theta_o)   :121-150
 *       pdf(wi, wo)  = w alpha^exponent (exponent + 1) / (2 pi) [alpha > 0] + (1 - w) cos(theta_o) / pi          :157-186
 *     with alpha = wo . reflect(wi), w = specular_sampling_weight, both zero unless cos(theta_i), cos(theta_o) > 0;
 *     pdfComponent = 1.  Such a vertex classifies as DIFFUSE for every roughness above bounceRoughness (default 0.001,
 *     gvpm_struct.h:66-100,232-236): its photons are re-connected through it like through a Lambertian wall.
 *     ONE component (round 5): below roughness 0.05 (exponent > 798) sampleNext picks a component first
 *     (Phong::sampleComponent, phong.cpp:308-329: 0 = the specular lobe with probability w, 1 = the diffuse one) and the
 *     reconnection evaluates THAT component (bRec.component = sampledComponentIndex): eval = its term alone, pdf = its pdf
 *     times pdfComponent (w or 1 - w, :331-342) -- which is the matching term of the mixture above.  Such a vertex names an
 *     entry whose `distribution` field holds component + 1 (0: both, the entries of round 4; 1: specular only; 2: diffuse
 *     only): a surface that can be met both ways has one entry per way.
 *   GVPM_BSDF_ROUGHCONDUCTOR  src/bsdfs/roughconductor.cpp with an ISOTROPIC Beckmann or GGX distribution (alphaU == alphaV:
 *     the photon record carries the parent's normal, not its tangent frame) -- one component, EGlossyReflection:
 *       H = normalize(wi + wo), D = MicrofacetDistribution::eval (microfacet.h:191-232), G = smithG1(wi, H) smithG1(wo, H)
 *       (:477-522), F = fresnelConductorExact(wi . H, eta, k) * specular (libcore/util.cpp:747-769)
 *       eval(wi, wo) = F D G / (4 cos(theta_i))                                                                 :257-293
 *       pdf(wi, wo)  = D smithG1(wi, H) / (4 cos(theta_i))  [sample_visible]   or   D cos(theta_H) / (4 |wo . H|)  :295-319
 *     both zero unless cos(theta_i), cos(theta_o) > 0; pdfComponent = 1.  `exponent` carries alpha (after the constructor's
 *     clamp to >= 1e-4, microfacet.h:135-136).  Classified like any vertex by its roughness alpha against bounceRoughness.
 *   GVPM_BSDF_WARD (round 5)  src/bsdfs/ward.cpp with alphaU == alphaV (isotropic: the photon record carries the parent's normal,
 *     not its tangent frame) and roughness alpha >= 0.05, i.e. BOTH components (Ward::sampleComponent, ward.cpp:370-389): with
 *     H = wi + wo (NOT normalised, as the reference has it), tan2 = (|H|^2 - H.z^2) / H.z^2, E = exp(-tan2 / alpha^2),
 *       eval = (specular * factor1 * E [if factor1 * E > 1e-10] + diffuse / pi) cos(theta_o)                            :178-228
 *         factor1 = 1 / (4 pi alpha^2 sqrt(cos_i cos_o))      variant 0, "ward"
 *                 = 1 / (4 pi alpha^2 cos_i cos_o)            variant 1, "ward-duer"
 *                 = |H|^2 / (pi alpha^2 H.z^4)                variant 2, "balanced" (the plugin's default)
 *       pdf  = w E / (4 pi alpha^2 (Hn . wi) cos^3(theta_Hn)) + (1 - w) cos(theta_o) / pi,  Hn = H / |H|                :230-266
 *     `exponent` carries alpha, `sample_visible` the variant, `specular_sampling_weight` w; pdfComponent = 1.
 * A surface parent outside the closed set stays what it was: the host flags the photon's shift type 0 (failed shift).   */
enum { GVPM_BSDF_PHONG = 1, GVPM_BSDF_ROUGHCONDUCTOR = 2, GVPM_BSDF_WARD = 3 };
enum { GVPM_WARD_WARD = 0, GVPM_WARD_DUER = 1, GVPM_WARD_BALANCED = 2 };
enum { GVPM_MICROFACET_BECKMANN = 0, GVPM_MICROFACET_GGX = 1 };
typedef struct gvpm_bsdf {    /* 64 bytes */
  int32_t kind;               /* GVPM_BSDF_*                                                            */
  float specular[3];          /* m_specularReflectance (Phong: after ensureEnergyConservation, phong.cpp:86-91) */
  float exponent;             /* Phong: m_exponent; rough conductor, Ward: alpha                        */
  float specular_sampling_weight; /* Phong, Ward: m_specularSamplingWeight, phong.cpp:93-97, ward.cpp:158-162 */
  int32_t distribution;       /* rough conductor: GVPM_MICROFACET_*; Phong: sampled component + 1 (0 = both) */
  int32_t sample_visible;     /* rough conductor: m_sampleVisible (the pdf's form); Ward: GVPM_WARD_* variant */
  float eta[3], k[3];         /* rough conductor: m_eta, m_k (relative to the exterior, roughconductor.cpp:181-191) */
  float reserved[2];
} gvpm_bsdf;

/* ---- scene occluders for scene->rayIntersect(Ray) ------------------------*/
/* call sites shift_volume_photon.cpp:398, shift_volume_beams.cpp:421.
 * Triangle i = (v0, v0+e1, v0+e2); arrays of 3*n floats.                     */
typedef struct gvpm_triangles {
  const float *v0, *e1, *e2;
  uint32_t n;
} gvpm_triangles;

/* ---- volume photons: one record per stored light-path vertex -------------*/
/* Host-side flattening of GPhotonNodeData + the Path it points to
 * (gvpm/gvpm_accel.h:17-65,119-199). SoA, n elements per array, 3-vectors
 * packed xyz.  c = vertexId, parent = vertex(c-1).                            */
typedef struct gvpm_photon_soa {
  const float *pos;         /* 3n  vertex(c) position                          */
  const float *wi;          /* 3n  -edge(c-1)->d  (towards the parent)         */
  const float *flux;        /* 3n  GPhotonNodeData::weight                     */
  const float *parent_pos;  /* 3n  vertex(c-1) position                        */
  const float *parent_n;    /* 3n  geometric (= shading) normal of a surface or
                                   emitter parent; ignored for a medium parent */
  const float *prefix_w;    /* 3n  prod_{i<c-1} v_i.weight*v_i.rrWeight*e_i.weight
                                   (shift_volume_photon.cpp:415-422)           */
  const float *parent_scat; /* 3n  surface: diffuse reflectance; medium: sigma_s */
  const float *parent_wi;   /* 3n  unit direction parent -> vertex(c-2)        */
  const float *parent_pdf;  /* n   vertex(c-1)->pdf[EImportance] (area measure) */
  const float *edge_pdf;    /* n   edge(c-1)->pdf[EImportance]                 */
  const float *parent_rr;   /* n   vertex(c-1)->rrWeight                       */
  const float *parent_g;    /* n   HG g of the parent's medium (medium parent) */
  const uint32_t *flags;    /* n   GVPM_PF_*                                   */
  const uint32_t *path_id;  /* n   GPhotonNodeData::pathID                     */
  uint64_t n;
} gvpm_photon_soa;

/* ---- camera beams ---------------------------------------------------------*/
/* One gvpm_camera_ray = one medium edge of a camera path (base, or the same
 * edge of the path re-traced through an offset pixel by
 * ShiftGatherPoint::generate, gvpm/shift/shift_cameraPath.h:29-133), with the
 * SVertexPDF cache entries the functors read (gvpm/gvpm_struct.h:361-370,
 * 585-631).  64 bytes.  A beam SET = 5 consecutive rays: base, L, R, T, B.   */
typedef struct gvpm_camera_ray {
  float o[3];        /* vertex(e) position                                     */
  float len;         /* edge(e)->length                                        */
  float d[3];        /* -edge(e)->d, unit, pointing away from the camera       */
  float pdf;         /* getVertexInfo(e).pdf                                   */
  float eye[3];      /* getWeightBeam(e-1) * getWeightVertex(e)                */
  float jacobian;    /* getVertexInfo(e).jacobian                              */
  float gop;         /* GOp(e) = geometryOpposingTerm(path, e, e+1)            */
  uint32_t info;     /* bit0: validVolumeEdge(e, medium) (always 1 for base);
                        bits 8..15: edge index e                               */
  float rand;        /* base ray only: the sampler->next1D() of this beam
                        (gvpm.cpp:1042 BRE; :1156 VPM)                         */
  uint32_t pixel;    /* base ray only: px | py << 16                           */
} gvpm_camera_ray;

#define GVPM_RAY_VALID(info) ((info) & 1u)
#define GVPM_RAY_EDGE(info) (((info) >> 8) & 0xFFu)
#define GVPM_RAY_INFO(valid, edge) (((uint32_t)(valid) & 1u) | (((uint32_t)(edge) & 0xFFu) << 8))

/* G-VPM only: one record per camera sample (gvpm.cpp:1143-1180).  The host keeps the
 * per-path part of the loop -- the CDF over the medium edges of the pixel's camera path
 * (selBeam, :1117-1129) and sampleReuse (:1148) -- and hands over, per sample, which beam
 * set was selected, the re-used random number that drives sampleDistance(EDistanceAlwaysValid)
 * (:1168) and the selection probability selBeam[sampleIndex].  Samples of one pixel must be
 * consecutive (they are summed in order).                                                  */
typedef struct gvpm_vpm_sample {
  uint32_t set;      /* index of the beam set (pixel, medium edge) in the uploaded rays      */
  float rand;        /* randSample after sampleReuse                                        */
  float pdf_sel;     /* selBeam[sampleIndex]                                                */
  uint32_t reserved;
} gvpm_vpm_sample;

/* ---- per-pixel accumulators (what GatherPoint keeps across iterations) ---*/
/* gvpm/gvpm_struct.h:429-441: mediumFlux, shiftedMediumFlux[4],
 * weightedMediumFlux[4]; each RGB => 27 floats per pixel, row-major pixels.   */
#define GVPM_ACCUM_FLOATS 27

typedef struct gvpm_stats {
  uint64_t evaluations;   /* functor invocations that produced a base
                             contribution + 4 shift attempts       */
  uint64_t candidates;    /* (beam, photon) pairs tested geometrically         */
  uint64_t null_shifts, diffuse_shifts, failed_shifts;
  uint64_t dropped_pairs; /* (photon, beam) pairs the traversal could not store: must be 0 -- gvpm_get_stats returns
                             GVPM_ERR_STATE otherwise (the image would be biased)  */
  uint64_t reserved[2];   /* [0]: G-BRE, the last build's photon cells: kind << 56 | count (low 32 bits); kind 0 = uniform
                             3D grid, 1 = cells over the (u, v) plane of a single-origin ray bundle;
                             bits 32-55: steps whose traversal + evaluation were queued before the planner's counters were
                             known and had to be queued again (a diagnostic, not an error);
                             [1]: packed photon records whose material index lay beyond the uploaded table (decoded with a
                             black parent): must be 0 -- gvpm_get_stats returns GVPM_ERR_STATE otherwise           */
} gvpm_stats;

typedef struct gvpm_context gvpm_context;

/* ---- lifecycle ------------------------------------------------------------*/
/* device: HIP device ordinal.  *out receives the handle.                     */
int gvpm_create(const gvpm_params *params, int device, gvpm_context **out);
int gvpm_destroy(gvpm_context *h);
const char *gvpm_last_error(const gvpm_context *h);
int gvpm_abi_version(void);

/* Reset per-pixel accumulators and the APA radius scale to
 * initialScaleVolume (GPMIntegrator::render, gvpm.cpp:272-291).              */
int gvpm_reset(gvpm_context *h);

/* Decisions.  Every decision of a shift -- null shift or reconnection, the mirror step of getShiftPos, the triangle tests of
 * the shadow segment, the sign / cosine tests of the reconnection -- is taken in fp32 WITH a rigorous error margin; a shift
 * with a comparison inside its margin (~1e-5 of them; for a parent that position rounding left behind the wall it sits
 * on, the reconnections within a few degrees of grazing) is evaluated by the EXACT PASS instead: the reference's statement
 * in fp64 on the same fp32 inputs (csrc/exact_shift.hip for G-BRE and G-VPM, run when the sums are read and every few
 * gathers; csrc/gather_beams.hip exact_beams_kernel for G-Beams, behind every evaluation), so the shift counters of
 * gvpm_stats equal those of a double-precision evaluation of the same inputs exactly (all four techniques; limits:
 * DESIGN.md section 2).  evaluated: shifts the exact passes have taken since gvpm_reset; lost: shifts that did not fit
 * their list (also reported as dropped_pairs: gvpm_get_stats then fails).                                               */
int gvpm_get_exact_shift_count(gvpm_context *h, uint64_t *evaluated, uint64_t *lost);


/* ---- uploads --------------------------------------------------------------*/
int gvpm_upload_scene(gvpm_context *h, const gvpm_triangles *tris);
int gvpm_upload_medium(gvpm_context *h, const gvpm_medium *medium);
/* the BSDF table of the scene's glossy surfaces (once per scene, before the first gather that meets a photon of parent
 * type GVPM_PARENT_SURFACE_BSDF; such a photon with an index beyond the table is a failed shift)                       */
int gvpm_upload_bsdfs(gvpm_context *h, const gvpm_bsdf *table, uint32_t n);
/* per iteration: replaces the proc->getPhotonVolumeMap() of gvpm.cpp:450-454  */
int gvpm_upload_photons(gvpm_context *h, const gvpm_photon_soa *photons);
/* G-Beams, per iteration: replaces proc->getBeamMap() (gvpm.cpp:449).  One record per LTPhotonBeam
 * (gvpm/gvpm_beams.h:18-43), i.e. per medium edge i of a light path, in the photon SoA re-read as:
 *   pos = vertex(i+1) (beam end), parent_* = vertex(i) (beam origin), wi = -edge(i)->d,
 *   flux = LTPhotonBeam::flux (without the transmittance of edge i), prefix_w = prod_{k<i},
 *   parent_pdf = vertex(i)->pdf[EImportance], flags: depth = i, shift type = getTypeShift(path, i+1),
 *   path_id = LTPhotonBeam::pathID;
 * end_n: 3 floats per beam, geometric normal of vertex(i+1), all zero when it is a medium interaction.
 * The beam radius is R*0.01*globalScaleVolume (gvpm.cpp:391,881), kept by the handle.
 * Per-hit randoms (v and w samples of the 3D kernel) are Philox4x32-10 with
 * key = {bits(base ray rand), 0x6265616d}, counter = {beam index, 0, 0, 0}.                    */
int gvpm_upload_beams(gvpm_context *h, const gvpm_photon_soa *beams, const float *end_n);
int gvpm_upload_beams_dev(gvpm_context *h, const gvpm_photon_soa *beams_dev, const float *end_n_dev);
/* G-Planes (0D kernel), per iteration: replaces the LTPhotonPlane list of gvpm.cpp:790-800.  One
 * plane per photon beam (same SoA reading as gvpm_upload_beams: ori = parent_pos, w0 length0 =
 * pos - parent_pos, flux, flags depth = edgeID) plus the second edge drawn by
 * LTPhotonPlane::transformBeam (gvpm/gvpm_plane.h:53-73): w1 (3 floats, unit) and len1 (1 float)
 * per plane.  Preconditions of the reference hold: sensor inside the medium (gvpm.cpp:784-788,
 * i.e. camera rays with medium edge 1), min_depth >= 2 (gvpm.cpp:163-166), no null shift.       */
int gvpm_upload_planes(gvpm_context *h, const gvpm_photon_soa *beams, const float *w1, const float *len1);
int gvpm_upload_planes_dev(gvpm_context *h, const gvpm_photon_soa *beams_dev, const float *w1_dev,
                           const float *len1_dev);
/* per iteration: n_sets beam sets (5 rays each). Replaces GatherPoint[] +
 * ShiftGatherPoint[4] for the medium edges of every pixel.  Sets may come in
 * any order; several sets may address the same pixel (several medium edges). */
int gvpm_upload_camera_beams(gvpm_context *h, const gvpm_camera_ray *rays,
                             uint64_t n_sets);
/* G-VPM: the camera samples of this iteration (n = pixels_with_medium * nbCameraSamples)   */
int gvpm_upload_vpm_samples(gvpm_context *h, const gvpm_vpm_sample *samples, uint64_t n);
/* the same, source buffers already in device memory                          */
int gvpm_upload_photons_dev(gvpm_context *h, const gvpm_photon_soa *photons_dev);
int gvpm_upload_camera_beams_dev(gvpm_context *h,
                                 const gvpm_camera_ray *rays_dev,
                                 uint64_t n_sets);
int gvpm_upload_vpm_samples_dev(gvpm_context *h, const gvpm_vpm_sample *samples_dev, uint64_t n);

/* ---- pipelined uploads -----------------------------------------------------*/
/* Pinned host memory for producers (hipHostMalloc).  gvpm_host_alloc_photons lays ONE block out as the 14 arrays
 * of a photon SoA in the struct's order and returns the view: an upload from it is a single packed copy.           */
int gvpm_host_alloc(uint64_t bytes, void **out);
int gvpm_host_alloc_photons(uint64_t n, gvpm_photon_soa *view, void **block);
int gvpm_host_free(void *p);
/* Start copying the inputs of the step AFTER the coming gvpm_gather (pinned memory only): the copy runs on the copy
 * stream while that gather's kernels run, and the set becomes the current input when that gather returns -- as if
 * gvpm_upload_* had been called at that point.  One pending set per kind.  Loop of a pipelined host:
 *   upload(1); for N: prefetch(N+1); gather(N).                                                                     */
int gvpm_prefetch_photons(gvpm_context *h, const gvpm_photon_soa *photons);
int gvpm_prefetch_camera_beams(gvpm_context *h, const gvpm_camera_ray *rays, uint64_t n_sets);

/* ---- packed uploads (round 3) ----------------------------------------------*/
/* The SoA entry points above move 120 bytes a photon and 320 a beam set over PCIe -- at C2 2.9 times the duration of the
 * step they feed.  These records carry the same inputs in 76 and 272 bytes:
 *   photons  -- wi is not sent: the device forms normalize(parent_pos - pos) (in fp64, rounded once), which is what
 *               -edge(c-1)->d is -- to the rounding of the two fp32 positions: for a photon at distance L from its parent
 *               the derived direction is off by up to ~1e-7 / L rad (L ~ 1e-4, one photon in a thousand at C2: 1e-3 to
 *               1e-2 rad; tests/test_packed_upload.py bounds it).  A host that cannot accept that keeps the SoA upload
 *               (the shim: GVPM_HIP_UPLOAD=soa); parent_n and parent_wi travel as octahedral 2 x snorm16 (axis-aligned vectors are exact,
 *               others are off by <= 4.8e-5 rad (parent_n; parent_wi 5.8e-5): in general position -- tests/test_rotated_gpu.py -- the film's L2
 *               against the fp64 oracle is 8e-6 with packed records, 9e-8 with SoA; a zero vector stays zero); parent_scat and parent_g become an index into a
 *               table of the scene's materials (gvpm_upload_materials); path_id travels as the one bit the gather reads
 *               of it (checkerboard parity, gvpm.cpp:1020-1030) in bit 7 of flags.  Positions, flux, prefix_w and the three
 *               pdfs / weights stay fp32: evaluation counts are those of the SoA upload, bit for bit.
 *   beam set -- the base ray as it is, the four shifted rays without the fields only the base carries (rand, pixel) and
 *               with `info` folded away: valid = !signbit(len), edge = the base ray's.  Lossless.
 * gvpm_pack_* / gvpm_unpack_* are plain host functions (no GPU): a producer packs straight into pinned memory
 * (gvpm_host_alloc), and what a packed upload means is DEFINED by gvpm_unpack_photons -- the device decodes with the same
 * arithmetic, operation for operation.  Reference seam: gvpm/gvpm_accel.h:31-59,119-199 (the fields read back there).     */
typedef struct gvpm_material {
  float scat[3];            /* parent_scat: diffuse reflectance of a surface, sigma_s of a medium */
  float g;                  /* parent_g                                                            */
} gvpm_material;
typedef struct gvpm_photon_packed { /* 76 bytes */
  float pos[3];
  float parent_pdf;
  float parent_pos[3];
  float edge_pdf;
  float flux[3];
  float parent_rr;
  float prefix_w[3];
  uint32_t parent_n_oct;    /* x | y << 16, snorm16 each; 0x80008000: the zero vector            */
  uint32_t parent_wi_oct;
  uint32_t flags;           /* GVPM_PF_* with bit 7 = path_id & 1                                 */
  uint32_t material;        /* index into the table of gvpm_upload_materials                      */
} gvpm_photon_packed;
typedef struct gvpm_ray_packed {    /* 52 bytes: a shifted ray of a beam set */
  float o[3];
  float len;                /* sign bit set: !validVolumeEdge (the magnitude is still the length) */
  float d[3];
  float pdf;
  float eye[3];
  float jacobian;
  float gop;
} gvpm_ray_packed;
typedef struct gvpm_beam_set_packed { /* 272 bytes */
  gvpm_camera_ray base;
  gvpm_ray_packed shifted[4];
} gvpm_beam_set_packed;
/* table / table_n: the materials met so far (in/out, appended to; at most table_cap <= 65536 entries --
 * GVPM_ERR_INVALID_ARG beyond, e.g. a textured scene: such a host keeps the SoA upload).                                  */
int gvpm_pack_photons(const gvpm_photon_soa *src, gvpm_photon_packed *dst, gvpm_material *table, uint32_t table_cap,
                      uint32_t *table_n);
/* dst: 14 writable arrays of src_n elements (the pointers of a gvpm_photon_soa, cast)                                    */
int gvpm_unpack_photons(const gvpm_photon_packed *src, uint64_t n, const gvpm_material *table, uint32_t table_n,
                        const gvpm_photon_soa *dst);
/* GVPM_ERR_INVALID_ARG when a shifted ray's edge index differs from its base ray's                                       */
int gvpm_pack_camera_beams(const gvpm_camera_ray *rays, uint64_t n_sets, gvpm_beam_set_packed *dst);
int gvpm_unpack_camera_beams(const gvpm_beam_set_packed *src, uint64_t n_sets, gvpm_camera_ray *dst);
/* The table may be re-uploaded at any time with MORE entries behind an unchanged prefix (what gvpm_pack_photons' appending
 * produces): that never waits.  Changing existing entries, or growing beyond the allocated capacity, first waits for all
 * work of the handle.  A gather that consumes packed photons without a table fails with GVPM_ERR_STATE.                   */
int gvpm_upload_materials(gvpm_context *h, const gvpm_material *table, uint32_t n);
/* as gvpm_upload_photons / gvpm_prefetch_photons (pageable or pinned memory; prefetch: pinned only)                       */
int gvpm_upload_photons_packed(gvpm_context *h, const gvpm_photon_packed *photons, uint64_t n);
int gvpm_prefetch_photons_packed(gvpm_context *h, const gvpm_photon_packed *photons, uint64_t n);

/* ---- linked photon records (round 6) ---------------------------------------*/
/* The photons of one light path are CONSECUTIVE vertices (GPhotonMap::tryAppend walks the path, gvpm/gvpm_accel.h:119-199):
 * most of what a 76-byte record carries of its parent is already on the wire.  Three record kinds, chosen per photon by the
 * packer, which VERIFIES each choice against the SoA source (a photon that does not fit a short kind travels as a full one):
 *   chain (40 bytes) -- the parent is the medium vertex that IS the previous photon of the upload (same path, MEDIUM parent):
 *                       parent_pos = pos[i-1], prefix_w = flux[i-1] (the path weight up to the parent, bit for bit),
 *                       parent_wi = the previous photon's wi, parent_n = 0, parent_scat / parent_g = entry `component field of
 *                       flags` of the material table; the record keeps pos, flux, the three pdfs / weights, flags;
 *   emit  (48 bytes) -- the parent is an emitter vertex: prefix_w, parent_rr, parent_n, parent_g from entry `component field
 *                       of flags` of the blob's EMITTER table (32 bytes an entry: one per emitter triangle and power),
 *                       parent_scat = 0, parent_wi = (1, 0, 0) as tryAppend leaves them; the record keeps pos, parent_pos,
 *                       flux, parent_pdf, edge_pdf, flags;
 *   full  (76 bytes) -- gvpm_photon_packed, as above.
 * In both short kinds the component field of flags is free (such a parent's component type is always
 * GVPM_BSDF_DIFFUSE_REFLECTION -- verified) and carries the table index.  wi is derived as in the packed records; a chain
 * record's parent_wi is derived from the two positions before it (to the rounding of fp32 positions: tighter than the
 * octahedral code of a full record).  S-cbox / S-fogroom: 28 % chain, 58 % emit, 14 % full = 49.6 bytes a photon.
 * The records travel as ONE blob (one copy): header, 2-bit kinds, per-64-photon bases, emitter table, the three record
 * arrays.  What a blob means is DEFINED by gvpm_unpack_photons_linked; the device decodes with the same arithmetic.  */
typedef struct gvpm_emitter_entry { /* 32 bytes */
  float prefix_w[3];
  float parent_rr;
  float parent_n[3];
  float parent_g;
} gvpm_emitter_entry;
typedef struct gvpm_photon_emit { /* 48 bytes */
  float pos[3];
  float parent_pdf;
  float parent_pos[3];
  float edge_pdf;
  float flux[3];
  uint32_t flags;             /* GVPM_PF_* with bit 7 = path_id & 1; bits 16-31: index into the blob's emitter table */
} gvpm_photon_emit;
typedef struct gvpm_photon_chain { /* 40 bytes */
  float pos[3];
  float parent_pdf;
  float flux[3];
  float edge_pdf;
  float parent_rr;
  uint32_t flags;             /* ... bits 16-31: index into the table of gvpm_upload_materials (the medium parent's) */
} gvpm_photon_chain;
#define GVPM_LINKED_MAGIC 0x4C4E4B31u /* "LNK1" */
#define GVPM_LINKED_FULL 0u
#define GVPM_LINKED_EMIT 1u
#define GVPM_LINKED_CHAIN 2u
typedef struct gvpm_linked_header { /* 64 bytes, at the start of the blob; offsets in bytes from it, 16-byte aligned */
  uint32_t magic, n, n_full, n_emit, n_chain, n_emitters;
  uint32_t off_kinds;         /* ceil(n / 16) words, photon i: bits 2 (i % 16) .. +1                                    */
  uint32_t off_groups;        /* ceil(n / 64) x {full records, emit records before photon 64 g}                         */
  uint32_t off_emitters, off_full, off_emit, off_chain;
  uint32_t bytes;             /* of the whole blob                                                                      */
  uint32_t reserved[3];
} gvpm_linked_header;
/* bytes a blob of n photons can need at most (every photon full, 65536 emitter entries never reached in practice: 1024) */
size_t gvpm_linked_photons_bound(uint64_t n);
/* packs into `blob` (capacity `cap` bytes, e.g. pinned memory of gvpm_linked_photons_bound(n) bytes); the material table grows
 * as in gvpm_pack_photons; *bytes: the blob's size.  GVPM_ERR_INVALID_ARG: capacity or a table exhausted.               */
int gvpm_pack_photons_linked(const gvpm_photon_soa *src, void *blob, size_t cap, gvpm_material *table, uint32_t table_cap,
                             uint32_t *table_n, size_t *bytes);
int gvpm_unpack_photons_linked(const void *blob, size_t bytes, const gvpm_material *table, uint32_t table_n,
                               const gvpm_photon_soa *dst);
int gvpm_upload_photons_linked(gvpm_context *h, const void *blob, size_t bytes);
int gvpm_prefetch_photons_linked(gvpm_context *h, const void *blob, size_t bytes);
int gvpm_upload_camera_beams_packed(gvpm_context *h, const gvpm_beam_set_packed *sets, uint64_t n_sets);
int gvpm_prefetch_camera_beams_packed(gvpm_context *h, const gvpm_beam_set_packed *sets, uint64_t n_sets);

/* ---- compact camera-beam sets (round 4) ------------------------------------*/
/* The beam sets of a frame are most of a step's upload (272 of every 348 bytes at one set per photon-free pixel; 71 of
 * C2's 143.6 MB).  For the FIRST medium edge of a camera path of a perspective sensor nearly all of a set is a function
 * of the sensor: every ray starts on the line through the sensor's origin and its film position, the five film positions
 * share their fractional offset (src/libbidir/vertex.cpp:345-346), eyeContrib = getWeightBeam(e-1) * getWeightVertex(e)
 * is 1 (importance-sampled pinhole, null boundary), and the SVertexPDF entries enter the gather only through
 * GatherPoint::sensorMIS (gvpm/gvpm_struct.h:608-631) = (pdf_s / pdf_b) * jacobian_s, where ShiftGatherPoint::trace /
 * generate DEFINE jacobian_s as the reciprocal of that pdf ratio (shift_cameraPath.h:76-116,191-242: jacobian = pdf1/pdf2 *
 * GOpBase/GOpNew against pdf = pdf2 * GOpNew; pdf2 == 0 replaces pdf2 by pdf1 in BOTH) -- the product is 1.  Such a set
 * travels as 60 bytes: pixel, the sample's fractional film offset, the base ray's random number, a validity mask, and per
 * ray the distance from the sensor's origin to the start of the edge (0 when the sensor sits in the medium) and the edge's
 * length.  The device rebuilds the five rays from the sensor (gvpm_upload_sensor); deeper edges (behind a mirror or a
 * rough vertex: their eye weight, origin and Jacobians are the path's) keep the full 272-byte records and ride along in the
 * same upload.  What a compact set MEANS is defined by gvpm_unpack_camera_beams_compact, with which the device agrees bit
 * for bit (fp64, no contraction).  Against the producer's own fp32 rays the decoded ones differ by rounding only: d is the
 * sensor direction rounded once; o = fp32(origin + t0 * d) moves along the ray by the rounding of t0 (<= 6e-8 relative)
 * and off it by half an ulp, as the producer's own fp32 o does against the exact ray; pdf = jacobian = gop = 1 carry the
 * product above exactly where the producer's fp32 factors carry it to a few ulp.                                           */
typedef struct gvpm_sensor {
  double pos[3];            /* sensor origin, world space                                                                 */
  double to_world[9];       /* row-major rotation camera -> world; camera space looks along -z, +x to the right of the
                               film, +y towards larger pixel rows (a shim folds Mitsuba's flips into it)                  */
  double tan_half_fov_x, tan_half_fov_y;
  int32_t width, height;    /* film size in pixels                                                                        */
  int32_t reserved[4];
} gvpm_sensor;
/* direction through film position (sx, sy), in pixels:
 *   c = ((2 sx / width - 1) tan_half_fov_x, (2 sy / height - 1) tan_half_fov_y, -1);  d = to_world * (c / |c|)          */
typedef struct gvpm_beam_set_compact { /* 60 bytes */
  uint32_t pixel;           /* px | py << 16 of the base path                                                             */
  float jitter[2];          /* film position of the base sample minus (px, py), in [0, 1); shifted path k samples
                               (px, py) + offset_k + jitter, offsets L R T B = (-1,0) (+1,0) (0,+1) (0,-1)                */
  float rand;               /* the base ray's `rand`                                                                      */
  uint32_t info;            /* bit k (0 base, 1..4 L R T B): validVolumeEdge of ray k; bits 8..15: edge index e           */
  float t0[5];              /* |vertex(e) - sensor origin| per ray                                                        */
  float len[5];             /* edge(e)->length per ray                                                                    */
} gvpm_beam_set_compact;
int gvpm_upload_sensor(gvpm_context *h, const gvpm_sensor *sensor);
/* Splits n_sets beam sets (5 rays each) into compact records and full packed records.  jitter: 2 floats per set.  A set
 * goes to `compact` iff its decode reproduces the five rays (o, d within 4 ulp of their magnitude, len, validity and edge
 * exactly), every eye weight is 1 and every valid shifted ray has |pdf_s / pdf_b * jacobian_s - 1| <= 1e-4; otherwise to
 * `full`.  Both outputs need room for n_sets records.  new_index (n_sets entries, may be NULL): the index a set has in the
 * upload made of the two lists -- compact sets first, in input order, then the full ones (G-VPM samples and shift
 * requests name sets by that index).                                                                                     */
int gvpm_pack_camera_beams_compact(const gvpm_sensor *sensor, const gvpm_camera_ray *rays, const float *jitter,
                                   uint64_t n_sets, gvpm_beam_set_compact *compact, uint64_t *n_compact,
                                   gvpm_beam_set_packed *full, uint64_t *n_full, uint32_t *new_index);
/* dst: 5 * n_compact rays                                                                                                */
int gvpm_unpack_camera_beams_compact(const gvpm_sensor *sensor, const gvpm_beam_set_compact *src, uint64_t n_compact,
                                     gvpm_camera_ray *dst);
/* as gvpm_upload_camera_beams_packed / gvpm_prefetch_camera_beams_packed: the upload holds n_compact + n_full sets       */
int gvpm_upload_camera_beams_compact(gvpm_context *h, const gvpm_beam_set_compact *compact, uint64_t n_compact,
                                     const gvpm_beam_set_packed *full, uint64_t n_full);
int gvpm_prefetch_camera_beams_compact(gvpm_context *h, const gvpm_beam_set_compact *compact, uint64_t n_compact,
                                       const gvpm_beam_set_packed *full, uint64_t n_full);

/* ---- manifold shifts through the host (first slice) ----------------------------------------*/
/* A photon whose shift type is 3 (EManifoldShift: a specular chain between the photon and the vertex it can be re-connected
 * from) needs the manifold walk of shiftPhotonManifold (shift_volume_photon.cpp:160-295: generateShiftPathME + ShiftME,
 * shift/operation/shift_ME.cpp:13-142, SpecularManifold::det, src/libbidir/mut_manifold.cpp:1310-1410) -- Newton iterations
 * over Mitsuba's Path / BSDF objects that stay on the host.  With use_manifold = 0 such a shift is a failed shift (weight 1),
 * as in the reference.  With use_manifold = 1 and gvpm_enable_host_shifts(h, capacity > 0), a G-BRE, (round 4) G-VPM
 * (VolumeGradientPositionQuery reaches the same dispatch, shift_volume_photon.cpp:489-655 -> :49-117) or G-Beams gather
 * (shiftBeamME, shift_volume_beams.cpp:601-746; see the request's and the answer's G-Beams notes) instead RECORDS
 * one request per (photon, beam, shifted pixel) that reaches shiftPhotonManifold -- everything the walk takes as input -- and
 * adds nothing for it; the host runs the walk for each request and hands the results back; the device then applies
 * shift_volume_photon.cpp:217-279 (contribution, Jacobian, MIS weight) and adds the terms to the iteration:
 *     gvpm_gather(it) -> gvpm_download_shift_requests -> [host: manifold walks] -> gvpm_upload_host_shifts
 * Requests the host never answers (the next gvpm_gather or download comes first), and requests beyond `capacity`, count as
 * failed shifts.                                                                                                          */
typedef struct gvpm_shift_request { /* 64 bytes */
  uint32_t photon;          /* index into this iteration's photon (G-Beams: beam) upload: lightPath / vertexId = c of the
                               walk (G-Beams: beam->path, c = edgeID + 1)                                                 */
  uint32_t set;             /* beam set (upload order) and ...                                                            */
  uint32_t shift;           /* ... which of its shifted rays: 0..3 = L R T B                                              */
  uint32_t reserved;        /* G-Beams: the bits of the float kRec.pdf() (pdfEdgeFailure * pdfKernel), the pdf
                               cacheSourcePath gives the re-cut last edge of the source path (shift_volume_beams.cpp:574)  */
  float offset_pos[3];      /* offsetPos (getShiftPos, :858-896; G-Beams: newPos): where vertex c of the proposal lies    */
  float radius;             /* photonRadius (the host multiplies by its config.relaxME); G-VPM: the pixel's own radius   */
  float base_point[3];      /* baseRay(baseRay.maxt); G-Beams: baseCameraRay(shiftW - mint), as generateShiftPathME is
                               handed it (shift_volume_beams.cpp:627)                                                     */
  float t;                  /* baseRay.maxt = shiftRay.maxt = t'; G-Beams: shiftW = kRec.w                                */
  float shift_point[3];     /* shiftRay(shiftRay.maxt); G-Beams: shiftRay(shiftW - mint) (:628)                           */
  float reserved2;          /* G-Beams: kRec.v, the kernel's place on the beam (cacheSourcePath moves vertex c there)      */
} gvpm_shift_request;
typedef struct gvpm_host_shift {    /* 40 bytes */
  uint32_t ok;              /* generateShiftPathME && ShiftME succeeded                                                    */
  float throughput[3];      /* sRecME.throughtput                                                                          */
  float wi[3];              /* normalize(proposal.vertex(c-1)->getPosition() - offsetPos); G-Beams: NOT normalised --
                               proposal.vertex(c-1)->getPosition() - newPos, i.e. -edge(c-1)->d * edge(c-1)->length: kernelPDF
                               (shift_volume_beams.cpp:653-656) needs the proposal's last edge whole                       */
  float pdf;                /* sRecME.pdf                                                                                  */
  float det_ratio;          /* manifold->det(proposal, b, c) / manifold->det(source, b, c) (G-Beams: of the cached source) */
  float base_pdf;           /* prod_{i=b}^{c-1} source.vertex(i)->pdf[EImportance] * source.edge(i)->pdf[EImportance]      */
} gvpm_host_shift;
int gvpm_enable_host_shifts(gvpm_context *h, uint64_t capacity);   /* 0: off (the default)                                 */
/* waits for the gather; *n = requests recorded (at most capacity), the first min(*n, cap) of them copied to `out`        */
int gvpm_download_shift_requests(gvpm_context *h, gvpm_shift_request *out, uint64_t cap, uint64_t *n);
/* results[k] answers request k; n must be the number of recorded requests                                                */
int gvpm_upload_host_shifts(gvpm_context *h, const gvpm_host_shift *results, uint64_t n);

/* ---- the hot path ---------------------------------------------------------*/
/* One SPPM iteration of computeVolumeGradientPhotonBRE (gvpm.cpp:988-1079;
 * vol_technique BRE2D/BRE3D): builds the acceleration structure over the
 * uploaded photons, gathers every uploaded beam set, normalises by nb_paths,
 * folds the result into the APA running mean with iteration `it` (1-based) and
 * applies scaleVolumeAPA(it) (gvpm.cpp:181-215).
 * vol_technique DISTANCE: computeVolumeGradientPhoton (gvpm.cpp:1081-1203):
 * gathers every uploaded camera sample with the per-pixel radius
 * R*0.01*scaleVol, ADDS the result (x 1/nbCameraSamples) to the accumulators
 * (plain sums; gvpm_download_film divides by the total emitted path count as
 * gvpm.cpp:489-492 does) and applies the per-pixel SPPM update of scaleVol /
 * NVol (:1191-1195).
 * vol_technique BEAM_BEAM_1D / BEAM_BEAM_3D_OPTIMIZED: computeVolumeGradientBeams
 * (gvpm.cpp:880-986) over the uploaded photon beams (APA fold as for BRE).
 * Asynchronous on the handle's stream.                                        */
int gvpm_gather(gvpm_context *h, int it, uint64_t nb_paths);

/* ---- primal estimators (second half) -----------------------*/
/* One iteration of the PRIMAL volume pass of the reference's `sppm` integrator (src/integrators/photonmapper/sppm.cpp) for
 * the handle's vol_technique, over the SAME uploads, grids, planners and traversals as gvpm_gather -- the primal estimators
 * are the gradient functors' base terms with the conventions below -- with GatherPoint::fluxVol as the result: the first
 * three of a pixel's 27 accumulators (gvpm_download_accum; the other 24 stay zero).  Inputs re-read as:
 *   photons  pos, wi = -Photon::getDirection(), flux = Photon::getPower() (what Mitsuba's compressed record DEcodes to,
 *            include/mitsuba/render/photon.h:86-135), flags depth = Photon::getDepth(); the parent fields are not read;
 *   beams    (gvpm_upload_beams) as for gvpm_gather: PhotonBeam origin / end / flux / depth;
 *   camera   the BASE ray of each set = one Beam{p1, p2, weight, depth} of a gather point: o = p1, d, len = |p2 - p1|,
 *            eye = beam.weight, edge = beam.depth, rand, pixel (sppm.cpp:949-981); the four shifted rays are ignored.
 * BRE2D / BRE3D  volumePhotonPassBRE (sppm.cpp:882-1000) with BeamRadianceEstimator::query (bre.cpp:166-254) on the
 *   estimator built with cameraHeuristic = true (sppm.cpp:927: the gradient pass's uniform radius): per hit
 *     Tr(t') * power * phase(wi, -d) / kernelVolume * (1 / nb_paths) [* max(2 deltaT, 1e-4), 3D] * beam.weight
 *   on the ray re-based at r(mint), WITHOUT the sigma_s factor of the gradient functor, one random number PER HIT for the
 *   3D kernel -- Philox4x32-10, key = {bits(base ray rand), 0x70726d6c}, counter = {bits(pos.x), bits(pos.y), bits(pos.z),
 *   0} of the photon (the reference draws from a stateful per-thread sampler in traversal order) -- and a far check for the
 *   2D kernel (:240-242).  APA fold of sppm.cpp:986 and scaleVolumeAPA(it) (sppm.cpp:255-285 = gvpm.cpp:181-215).
 * DISTANCE  the point estimate (sppm.cpp:1040-1126) with PhotonMap::estimateVolumeRadiance (librender/photonmap.cpp:277-330)
 *   per camera sample of gvpm_upload_vpm_samples: sum(power * phase) * beam.weight * Tr / (pdfSuccess * sel) * MCNorm /
 *   kernelVolume added to fluxVol (plain sums, as gvpm_gather does for this technique), M = the photons that pass the
 *   radius AND the depth test (`maxDepth > 0 &&`: a bound of zero or less filters nothing, as written), then the SPPM
 *   update of scaleVol / NVol (:1116-1120).
 * BEAM_BEAM_1D / BEAM_BEAM_3D_OPTIMIZED  volumePhotonPassBeams (sppm.cpp:762-880) with BeamRadianceQuery (pm/beams.h:29-223):
 *   the gradient pass's kernel record (shift_volume_beams.h:157-290) with the camera transmittance over [Epsilon, w]
 *   instead of [0, w]; the per-hit random numbers are gvpm_upload_beams' Philox stand-in; APA fold as for BRE.
 * max_depth filters as the queries' maxDepth = m_maxDepth - beam.depth does.  GVPM_ERR_UNSUPPORTED when the handle carries a
 * filter the primal passes do not have (path_set, debug_shift, min_depth, interaction modes) and for PLANE0D / the _NAIVE /
 * _EGSR beam kernels.  gvpm_stats.evaluations counts the accepted pairs.                                              */
int gvpm_gather_primal(gvpm_context *h, int it, uint64_t nb_paths);

/* current kernel radius R*0.01*globalScaleVolume (gvpm.cpp:989)              */
int gvpm_get_radius(gvpm_context *h, float *radius);
int gvpm_set_global_scale(gvpm_context *h, float global_scale_volume);
int gvpm_get_stats(gvpm_context *h, gvpm_stats *out);
/* G-VPM per-pixel SPPM state (GatherPoint::scaleVol / NVol), width*height floats each */
int gvpm_download_vpm_state(gvpm_context *h, float *scale_vol, float *n_vol);
/* average duration in ms of the gather kernel launches since the last call,
 * measured with HIP events on the handle's stream, and their number          */
int gvpm_get_kernel_time(gvpm_context *h, float *avg_ms, uint32_t *launches);
/* the same for one phase of gvpm_gather: 0 = dominant kernel (as above; G-BRE: the evaluation
 * kernel), 1 = G-BRE traversal kernel, 2 = build (photon grid, beam sort, planner)             */
int gvpm_get_phase_time(gvpm_context *h, int phase, float *avg_ms, uint32_t *launches);

/* ---- results --------------------------------------------------------------*/
/* 27 floats per pixel (GVPM_ACCUM_FLOATS), width*height pixels               */
int gvpm_download_accum(gvpm_context *h, float *accum);
/* the same into DEVICE memory (e.g. a tensor that torch.distributed all-reduces) */
int gvpm_download_accum_dev(gvpm_context *h, float *accum_dev);
/* throughput (gvpm.cpp:480-500 with reusePrimal :503-532 when reuse_primal)
 * and the gradient images of computeGradient (gvpm.cpp:1205-1306), each
 * width*height*3 floats; emission may be NULL (else added as emission/it).   */
int gvpm_download_film(gvpm_context *h, int it, int reuse_primal,
                       const float *emission, float *throughput, float *dx,
                       float *dy);
/* the same three images, back to back (throughput | dx | dy, 9*width*height
 * floats) in DEVICE memory on the context's stream: what an image-sharded rank
 * hands to the film all-reduce and gvpm_poisson_solve_dev.  emission_dev may
 * be NULL.  Asynchronous: film_dev must not be in flight on another stream, and
 * gvpm_synchronize orders it before a consumer on one.                         */
int gvpm_download_film_dev(gvpm_context *h, int it, int reuse_primal,
                           const float *emission_dev, float *film_dev);
int gvpm_synchronize(gvpm_context *h);

/* ---- reconstruction -----------------------------------*/
/* Screened-Poisson reconstruction of the final image from throughput + gradients: replaces
 * poisson::Solver as gvpm.cpp:610-690 drives it (importImagesMTS, setupBackend, solveIndirect,
 * exportImagesMTS; src/integrators/poisson_solver/Solver.cpp:376-497, Backend.cpp:154-384).
 * The fields are Solver::Params' solver configuration (Solver.hpp:81-88); cg_precond must be 0
 * (no preset of the reference enables the preconditioner).                                    */
typedef struct gvpm_poisson_params {
  float alpha;               /* weight of the primal image, reconstructAlpha (0.2)   */
  int32_t irls_iter_max;     /* 1 = L2, > 1 = L1 by IRLS                              */
  float irls_reg_init;
  float irls_reg_iter;
  int32_t cg_iter_max;
  int32_t cg_iter_check;
  int32_t cg_precond;
  float cg_tolerance;
} gvpm_poisson_params;
/* Solver::Params::setConfigPreset (Solver.cpp:91-164): "L1D", "L1Q", "L1L", "L2D", "L2Q";
 * alpha is set to the reference's default 0.2.  GVPM_ERR_INVALID_ARG for an unknown preset.   */
int gvpm_poisson_preset(const char *preset, gvpm_poisson_params *out);
/* dx, dy, throughput, direct (may be NULL), out: width*height*3 floats, RGB triplets, row major.
 * out = reconstruction (+ direct), i.e. what exportImagesMTS returns.                          */
int gvpm_poisson_solve(gvpm_context *h, const gvpm_poisson_params *params, int width, int height,
                       const float *dx, const float *dy, const float *throughput, const float *direct,
                       float *out);
/* the same with every image in device memory (a device-side producer)      */
int gvpm_poisson_solve_dev(gvpm_context *h, const gvpm_poisson_params *params, int width, int height,
                           const float *dx, const float *dy, const float *throughput,
                           const float *direct, float *out);

/* ---- device-side producers for closed-form scenes -------*/
/* Photon shooting (GPMIntegrator's photon pass: gvpm_proc.cpp:125-209,278-350 with GPhotonMap::tryAppend /
 * LTBeamMap::tryAppendLT flattening) and camera-beam generation (randomWalkFromPixelToFirstDiffuse +
 * ShiftGatherPoint::generate, gvpm_gatherpoint.h:22-170, shift_cameraPath.h:29-133) on the GPU for scenes the
 * device can intersect itself: a triangle list, Lambertian / index-matched materials, one quad area light, one
 * homogeneous medium, a pinhole sensor.  The outputs are device-resident and go straight into
 * gvpm_upload_photons_dev / gvpm_upload_beams_dev / gvpm_upload_camera_beams_dev: nothing crosses PCIe.
 * Same counter-based streams as the host generators of gvpm_amd/host (keyed by path / pixel index), so the
 * device reproduces the sequential host loop: same photon count, same path count, same order.              */
typedef struct gvpm_devgen_scene {
  uint32_t n_tris, n_mats;
  const double *tris;        /* 12 per triangle: v0, e1, e2, geometric normal   */
  const int32_t *tri_mat;    /* material index per triangle                      */
  const int32_t *mat_kind;   /* 0 Lambertian, 1 index-matched medium boundary    */
  const double *mat_albedo;  /* 3 per material                                   */
  double light_c[3], light_u[3], light_v[3], light_n[3], radiance[3], light_area;
  gvpm_medium medium;
  double cam_pos[3], tan_half_fov_x;
  int32_t width, height;
  uint32_t seed;
  int32_t camera_inside;     /* sensor inside the medium: edge 1 is the medium edge */
  int32_t max_depth, rr_depth, min_depth;  /* GPMConfig maxDepth, rrDepth, minDepth */
  double camera_sphere;      /* world units (gvpm.cpp:162)                        */
  double cam_to_world[9];    /* row-major rotation camera -> world (camera space looks along -z, +x to the right of the
                                film, +y towards larger pixel rows: gvpm_sensor's to_world); ALL ZERO = identity (ABI 3) */
} gvpm_devgen_scene;
typedef struct gvpm_devgen gvpm_devgen;
int gvpm_devgen_create(const gvpm_devgen_scene *scene, int device, gvpm_devgen **out);
int gvpm_devgen_destroy(gvpm_devgen *g);
/* shoots light paths 0, 1, 2, ... of `iteration` until `capacity` photons are stored; *dev_soa receives device
 * pointers owned by the generator, *nb_paths the number of paths shot.  The generator rotates THREE output sets per
 * kind (photons / beams, camera rays): an output stays untouched during the two following calls of its kind, which
 * is what the borrowing rule of the `_dev` uploads needs (kernels of three steps in flight)                       */
int gvpm_devgen_shoot_photons(gvpm_devgen *g, int iteration, uint64_t capacity, gvpm_photon_soa *dev_soa,
                              uint64_t *nb_paths);
/* photon beams (one record per medium edge, see gvpm_upload_beams) + the end normals                           */
int gvpm_devgen_shoot_beams(gvpm_devgen *g, int iteration, uint64_t capacity, gvpm_photon_soa *dev_soa,
                            const float **end_n_dev, uint64_t *nb_paths);
/* beam sets (5 rays each) of the pixels whose 4x4 tile t has t % tile_mod == tile_rem (1, 0: the whole frame),
 * row-major pixel order; *rays_dev is owned by the generator (one of three buffers used in turn, see above)        */
int gvpm_devgen_camera_beams(gvpm_devgen *g, int iteration, int tile_mod, int tile_rem,
                             const gvpm_camera_ray **rays_dev, uint64_t *n_sets);
/* copies `bytes` of a generator output back to the host (inspection, tests)                                     */
int gvpm_devgen_read(gvpm_devgen *g, const void *dev, void *host, uint64_t bytes);

/* ---- multi-GPU (image-tile sharding) --------------------------*/
/* Each rank gathers only the beam sets of its own pixels; before
 * reconstruction the 27-float accumulators (disjoint supports) are summed
 * across ranks.  comm_id is the 128-byte ncclUniqueId produced on rank 0 by
 * gvpm_comm_unique_id and distributed by the host (e.g. torch.distributed).  */
int gvpm_comm_unique_id(void *id128);
int gvpm_comm_init(gvpm_context *h, const void *id128, int rank, int world);
int gvpm_allreduce_accum(gvpm_context *h);
/* SURVEY 8e's collective: sum the 3 film planes (gvpm_download_film_dev's
 * layout, 9*width*height floats) in place.  computeGradient adds one term from
 * the pixel and one from its +x / +y neighbour (gvpm.cpp:1223,1266), so the sum
 * of the ranks' partial films is bit-identical to the single-GPU film for dx, dy
 * and throughput; with reuse_primal the throughput's 8-term sum is associated
 * differently (a few ulp).  A third of gvpm_allreduce_accum's bytes.            */
int gvpm_allreduce_film(gvpm_context *h, float *film_dev);

#ifdef __cplusplus
}
#endif
#endif /* GVPM_HIP_H */
