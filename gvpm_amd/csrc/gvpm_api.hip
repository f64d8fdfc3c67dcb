// C ABI of libgvpm_hip.so (include/gvpm_hip.h): handle, uploads, per-iteration driver.
// Mirrors the driver logic of GPMIntegrator::photonMapPass / computeVolumeGradientPhotonBRE
// (gvpm/gvpm.cpp:383-500, 988-1079) and scaleVolumeAPA (gvpm.cpp:181-215).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "device_types.h"
#include "scene_bvh.h"

namespace gvpm {
hipError_t sortPairsU32(SortTemp &tmp, const uint32_t *kIn, uint32_t *kOut, const uint32_t *vIn, uint32_t *vOut,
                        uint32_t n, int endBit, hipStream_t s);
void launch_bounds(const float *pos, uint32_t n, float *partial, int nblocks, float *out6, float *hostOut,
                   hipStream_t s);
hipError_t reserveScanTemp(SortTemp &tmp, uint32_t n);
void launch_bundle_fit(const gvpm_camera_ray *rays, uint32_t nsets, int pass, const Grid &g, double *out, hipStream_t s);
void launch_export_u32(const uint32_t *a, const uint32_t *b, const uint32_t *c, const uint32_t *d, const uint32_t *e, uint32_t *hostOut,
                       hipStream_t s);
void launch_cell_keys(const float *pos, uint32_t n, const Grid &g, uint32_t *keys, uint32_t *vals, hipStream_t s);
void launch_sat(const uint32_t *cellStart, const Grid &g, uint32_t *sat, hipStream_t s);
void launch_cell_count(const float *pos, uint32_t n, const Grid &g, uint32_t *keys, uint32_t *rank, uint32_t *count,
                       hipStream_t s);
void launch_reorder(const gvpm_photon_soa &raw, const uint32_t *keys, const uint32_t *rank, const uint32_t *cellStart,
                    uint32_t n, const gvpm_params &cfg, const float4 *bvh, const float4 *tri4, uint32_t ntri, float dmax,
                    const NearGrid &ng, uint32_t *nearExt, uint32_t extCap, float4 *hot, float4 *cold, uint32_t *overflow,
                    hipStream_t s);
void launch_near_grid(const float4 *tri4, uint32_t ntri, const NearGrid &g, float reach, uint32_t *counts, uint32_t *tris, int mode,
                      hipStream_t s);
void launch_beam_count(const gvpm_camera_ray *rays, uint32_t nsets, int width, int tw, int th, uint32_t *keys,
                       uint32_t *rank, uint32_t *count, hipStream_t s);
void launch_beam_scatter(const uint32_t *keys, const uint32_t *rank, const uint32_t *start, uint32_t n,
                         uint32_t *setPerm, hipStream_t s);
void launch_tile_start(const uint32_t *start, uint32_t ntiles, uint32_t shift, uint32_t *tileStart, hipStream_t s);
void launch_segment_start(const uint32_t *keys, uint32_t n, uint32_t nseg, uint32_t shift, uint32_t *start,
                          hipStream_t s);
void launch_beam_keys(const gvpm_camera_ray *rays, uint32_t nsets, int width, int tw, int th, uint32_t *keys,
                      uint32_t *vals, hipStream_t s);
void launch_plan_bre(const GatherArgs &a, int beamsPerWave, uint32_t ntiles, uint32_t target, uint4 *items,
                     uint32_t *itemCount, uint2 *itemOff, uint32_t *blockTotal, uint32_t itemCap, hipStream_t stream);
void launch_traverse_bre(const GatherArgs &a, int beamsPerWave, const uint4 *items, const uint2 *itemOff,
                         const uint32_t *itemCount, uint32_t *queueHead, uint32_t *pairs, uint32_t *pairCnt,
                         uint32_t nwaves, bool persistent, hipStream_t stream);
void launch_evaluate_bre(const GatherArgs &a, int beamsPerWave, bool fullVis, const uint4 *items, const uint2 *itemOff,
                         const uint32_t *itemCount, uint32_t *queueHead, const uint32_t *pairs, const uint32_t *pairCnt,
                         uint32_t nwaves, bool persistent, hipStream_t stream);
uint32_t plan_items_capacity(uint32_t nsets, uint32_t ntiles, int beamsPerWave);
struct PoissonGraphCache {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  void *scratch = nullptr;
  int W = 0, H = 0, cgMax = 0;
  float alpha = 0.f;
};
void poisson_graph_release(PoissonGraphCache &c);
hipError_t poisson_solve_device(const gvpm_poisson_params &prm, int W, int H, const float *dx, const float *dy,
                                const float *tp, const float *direct, float *out, void *scratch,
                                PoissonGraphCache &cache, hipStream_t s);
size_t poisson_scratch_bytes(int W, int H);
void launch_finalize(float *accum, const float *iter, size_t n, int it, uint64_t nbPaths, hipStream_t s);
void launch_scale(const float *in, float *out, size_t n, float scale, hipStream_t s);
void launch_film(const float *acc, const float *emission, int w, int h, int it, int reusePrimal, float invDiv,
                 float *thr, float *dx, float *dy, hipStream_t s);
void launch_gather_vpm(const GatherArgs &a, bool fullVis, hipStream_t stream);
void launch_vpm_update(float *scaleVol, float *nVol, const float *mvol, size_t n, float alpha, uint32_t *maxScaleBits,
                       hipStream_t stream);
void launch_accumulate(float *accum, const float *iter, size_t n, hipStream_t stream);
hipError_t exclusiveSumU32(SortTemp &tmp, const uint32_t *in, uint32_t *out, uint32_t n, hipStream_t s);
void launch_shift_extent(const gvpm_camera_ray *rays, uint32_t nsets, uint32_t *extentBits, hipStream_t s);
void launch_beam_near(float4 *cold, uint32_t n, const float4 *tri4, uint32_t ntri, float r, const uint32_t *extentBits, float2 *clear, bool freeCone,
                      hipStream_t s);
void launch_beam_near_hist(const float4 *cold, uint32_t n, uint32_t ntri, uint32_t *hist, hipStream_t s);
void launch_beam_cold(const gvpm_photon_soa &raw, const float *endN, uint32_t n, const gvpm_params &cfg,
                      const uint32_t *subCounts, float4 *cold, float4 *aux, hipStream_t s);
void launch_beam_subcount(const float *p2, const float *p1, uint32_t n, float ls, uint32_t *counts, uint32_t *maxLs,
                          hipStream_t s);
void launch_beam_expand(const float *p2, const float *p1, uint32_t n, const uint32_t *counts, const uint32_t *offsets,
                        float *centres, uint32_t *ids, hipStream_t s);
void launch_sub_hot(const uint32_t *ids, const uint32_t *order, uint32_t n, const float4 *aux, float4 *hot,
                    uint32_t *hotFlags, hipStream_t s);
void launch_traverse_beams(const GatherArgs &a, const uint32_t *hotFlags, int beamsPerWave, const uint4 *items,
                           const uint32_t *itemCount, uint32_t itemCap, uint32_t *queueHead, uint2 *pairs, uint32_t *pairCount,
                           uint32_t pairCap, uint32_t *blockKey, uint32_t *blockVal, uint32_t nwaves, hipStream_t stream);
void launch_evaluate_beams(const GatherArgs &a, int beamsPerWave, bool exact, const uint2 *pairs, const uint32_t *sortedKey,
                           const uint32_t *sortedBlock, uint32_t nBlocks, uint32_t *queueHead, uint32_t nwaves,
                           hipStream_t stream);
struct PlaneArgs {
  const float4 *test;
  const float *ori, *end, *flux, *w1, *len1;
  const uint32_t *flags;
  uint32_t nplanes, planesPerItem;
};
void launch_plane_records(const PlaneArgs &pa, float4 *out, hipStream_t stream);
void launch_gather_planes(const GatherArgs &a, const PlaneArgs &pa, uint32_t ntiles, uint32_t nchunks,
                          hipStream_t stream);
}  // namespace gvpm

using namespace gvpm;

namespace {

template <typename T> struct DevBuf {
  T *p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = n + n / 2 + 64;  // generous: a regrowth is a device-wide sync
    hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  // at least `c` elements, no slack (mirrors another buffer's capacity)
  hipError_t reserveExact(size_t c) {
    if (c <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc((void **)&p, c * sizeof(T));
    if (e == hipSuccess) cap = c;
    return e;
  }
};

struct RcclApi {
  void *dl = nullptr;
  decltype(&ncclGetUniqueId) getUniqueId = nullptr;
  decltype(&ncclCommInitRank) commInitRank = nullptr;
  decltype(&ncclAllReduce) allReduce = nullptr;
  decltype(&ncclCommDestroy) commDestroy = nullptr;
  bool load() {
    if (dl) return true;
    dl = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!dl) dl = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!dl) return false;
    getUniqueId = (decltype(getUniqueId))dlsym(dl, "ncclGetUniqueId");
    commInitRank = (decltype(commInitRank))dlsym(dl, "ncclCommInitRank");
    allReduce = (decltype(allReduce))dlsym(dl, "ncclAllReduce");
    commDestroy = (decltype(commDestroy))dlsym(dl, "ncclCommDestroy");
    return getUniqueId && commInitRank && allReduce && commDestroy;
  }
};
RcclApi g_rccl;

}  // namespace

#define GVPM_PHASES 3
// counters: GVPM_STAT_ROWS rows of 8 (device_types.h), one per persistent wave / a few workgroups each, summed on read

// Everything a gather reads that is rebuilt per photon set / beam set.  Two of them: G-BRE builds
// step N+1 (grid, sorts, planner) on a second stream while the evaluation kernel of step N runs.
struct BuildSet {
  float builtRadius = -1.f;
  DevBuf<float4> hot, cold;
  DevBuf<uint32_t> overflowCtr;  // photons whose near-occluder list overflowed (they need the BVH kernels)
  DevBuf<uint32_t> cellStart, cellCount, sat, keysA, keysB, valsA, valsB;
  DevBuf<uint32_t> beamCount, beamStart;  // counting sort of the beam sets
  DevBuf<float> boundsPartial, bounds6;
  Grid grid;
  SortTemp sortTmp;
  DevBuf<uint32_t> bKeysA, bKeysB, bValsA, setPerm, tileStart;
  uint32_t ntiles = 0;
  int tileW = 4, tileH = 4;
  bool scanSized = false;
  DevBuf<uint4> items;
  DevBuf<uint2> itemOff;
  DevBuf<uint2> planBoxes;  // the planner's slab boxes, read by the traversal (GatherArgs::planBoxes)
  DevBuf<uint32_t> queueCtl;   // [0] itemCount, [1] queueHead, [2] queueHead of the evaluation kernel, [3] pair blocks
  // G-BRE: per-beam photon lists between the traversal and the evaluation kernel
  DevBuf<uint32_t> pairs, pairCnt, nearExt;
  hipEvent_t traversed = nullptr;  // recorded on the build stream after the traversal kernel
  hipEvent_t lastUse = nullptr;  // recorded on the gather stream after the kernels that read this set
  bool used = false;
  // Give this (so far unused) set the capacities of the set that just ran its first step, so that the second
  // step of a run does not stop for gigabytes of hipMalloc in the middle of the pipeline.
  hipError_t mirrorFrom(const BuildSet &o) {
    hipError_t e = hipSuccess;
#define GVPM_MIRROR(X) if (e == hipSuccess) e = X.reserveExact(o.X.cap)
    GVPM_MIRROR(hot); GVPM_MIRROR(cold); GVPM_MIRROR(overflowCtr); GVPM_MIRROR(cellStart); GVPM_MIRROR(cellCount);
    GVPM_MIRROR(sat); GVPM_MIRROR(keysA); GVPM_MIRROR(keysB); GVPM_MIRROR(valsA); GVPM_MIRROR(valsB);
    GVPM_MIRROR(beamCount); GVPM_MIRROR(beamStart); GVPM_MIRROR(boundsPartial); GVPM_MIRROR(bounds6);
    GVPM_MIRROR(bKeysA); GVPM_MIRROR(bKeysB); GVPM_MIRROR(bValsA); GVPM_MIRROR(setPerm); GVPM_MIRROR(tileStart);
    GVPM_MIRROR(items); GVPM_MIRROR(itemOff); GVPM_MIRROR(planBoxes); GVPM_MIRROR(queueCtl); GVPM_MIRROR(pairs); GVPM_MIRROR(pairCnt);
    GVPM_MIRROR(nearExt);
#undef GVPM_MIRROR
    return e;
  }
  void release() {
    hot.release(); cold.release(); overflowCtr.release(); cellStart.release(); cellCount.release(); sat.release();
    beamCount.release(); beamStart.release(); keysA.release(); keysB.release();
    valsA.release(); valsB.release(); boundsPartial.release(); bounds6.release(); bKeysA.release(); bKeysB.release();
    bValsA.release(); setPerm.release(); tileStart.release(); items.release(); itemOff.release(); planBoxes.release(); queueCtl.release();
    if (sortTmp.d) (void)hipFree(sortTmp.d);
    sortTmp.d = nullptr;
    sortTmp.bytes = 0;
    pairs.release(); pairCnt.release(); nearExt.release();
    if (lastUse) (void)hipEventDestroy(lastUse);
    if (traversed) (void)hipEventDestroy(traversed);
    lastUse = traversed = nullptr;
  }
};

struct gvpm_context {
  int device = 0;
  hipStream_t stream = nullptr;   // gather stream: traversal, evaluation, film
  hipStream_t streamB = nullptr;  // build stream of the G-BRE pipeline
  hipStream_t bstream = nullptr;  // where the current gather builds (stream, or streamB for G-BRE)
  hipStream_t streamC = nullptr;  // traversal stream of the three-stage pipeline
  bool travStream = true;         // traversal on its own stream, three build sets (GVPM_TRAV_STREAM=0: two stages)
  BuildSet sets[3];
  BuildSet *bs = &sets[0];
  int setIdx = 0;
  bool travOnBuild = true;        // traversal on the build stream (else on the gather stream)
  bool beamsExact = false;        // G-Beams: the literal fp64 evaluation instead of the local-frame fp32 one
  // G-BRE bundle cells (Grid::mode 1, bundle_grid.h), GVPM_BUNDLE=1.  Off by default: measured on MI355X (round 3,
  // scripts/r03_bundle_ab.sh, r03_bundle_c4.sh) the traversal gains 16 % alone (0.45 -> 0.38 ms at C2) and a rank's step
  // of the 8-GPU C4 run 8 % (3.09 -> 2.85 ms), but the pipelined C2 step is unchanged (1.29 ms: the evaluation kernel
  // paces it) and the whole-frame C4 step loses 4 %.  bundleState: 0 not fitted, 1 frame fitted (bundleGrid holds it;
  // the cell fields are filled per build), -1 the rays are not a bundle (until gvpm_reset).
  bool bundleEnabled = false;
  int bundleState = 0, bundleViolations = 0;
  int lastGridMode = 0;  // of the last G-BRE build (gvpm_stats::reserved[0])
  uint32_t lastGridCells = 0;
  float bundleDiv = 2.f;  // level-0 cells per tile width (GVPM_BUNDLE_DIV)
  Grid bundleGrid{};
  bool planBoxHandOff = true;     // G-BRE: the traversal reads the planner's slab boxes (GVPM_PLAN_BOXES=0: computes its own)
  bool beamsFreeCone = true;      // G-Beams: reconnections inside their beam's free cone skip the any-hit loop (GVPM_BEAMS_FREE_CONE=0: none do)
  size_t beamPairsInit = (size_t)16 << 20;  // G-Beams: first capacity of the pair list (GVPM_BEAM_PAIRS_INIT; tests shrink it)
  uint32_t beamItemsInit = 0;     // G-Beams: first capacity of the item list (GVPM_BEAM_ITEMS_INIT; tests shrink it; 0: the planner's bound)
  uint32_t beamItemCap = 0;       // G-Beams: capacity the item list was regrown to after an overflow
  bool pipeline = true;           // GVPM_PIPELINE=0: everything on the gather stream (isolated kernel timings)
  gvpm_params cfg;
  gvpm_medium medium;
  bool haveMedium = false;
  std::string err;

  // scene
  DevBuf<float4> tri4, bvh;   // packed triangles in BVH leaf order + nodes (scene_bvh.h)
  uint32_t ntri = 0;
  float triMin[3] = {0, 0, 0}, triMax[3] = {0, 0, 0};  // occluder bounds (host side, at upload)
  // occluders by cell of a coarse grid, for the near-occluder lists of scenes with more than 64 of them (built on the
  // first gather of a scene, for 1.5 x the reach it asks for; rebuilt if a later gather asks for more)
  DevBuf<uint32_t> nearGridStart, nearGridTris, nearGridCount;
  NearGrid nearGrid;
  float nearGridReach = -1.f;
  bool useNearGrid = true;  // GVPM_NEAR_GRID=0: the BVH point query instead (kept as the cross-check of the grid)

  // Host uploads land in a ring of three staging slots per kind, through a copy stream of their own: the copy of
  // step N+1 (or, prefetched, N+2) then runs while the kernels of step N still read theirs.  A slot's `copied` event
  // orders its consumers after the copy, `freed` (rays: read until the evaluation kernel ends) the next copy after them.
  struct PhotonSlot {
    DevBuf<uint32_t> raw;   // 30 words per photon: the 8 xyz arrays, the 4 scalars, flags, path_id (the ABI's order)
    gvpm_photon_soa dev;    // device pointers into raw
    hipEvent_t copied = nullptr;
    // recorded, on the gather stream and on the build stream, behind the kernels of every gather that read the slot
    // (builds; the G-Planes gather itself): the next copy into the slot waits for both
    hipEvent_t consumed = nullptr, consumedB = nullptr;
    bool read = false;      // a gather has launched kernels that read it since its last copy
  } phSlot[3];
  struct RaySlot {
    DevBuf<gvpm_camera_ray> rays;
    uint32_t nsets = 0;
    hipEvent_t copied = nullptr, freed = nullptr;
    bool read = false;      // a gather has launched kernels that read it since its last copy
  } raySlot[3];
  int phCur = 0, phPending = -1, rayCur = 0, rayPending = -1;   // pending: prefetched, current after the next gather
  bool phWait = false, rayWait = false;   // the next gather's streams must wait for the current slot's copy
  bool raysOwnedCur = false;              // the current camera rays live in raySlot[rayCur]
  bool photonsOwnedCur = false;           // the current photon map lives in phSlot[phCur]
  hipStream_t copyStream = nullptr;
  // photons: raw upload (owned copies or borrowed device pointers) and the built grid
  gvpm_photon_soa rawDev;  // device pointers
  uint32_t nph = 0;
  bool havePhotons = false, photonsDirty = false;
  bool nearOverflow = false;
  size_t nearExtWant = 0;         // entries the near-occluder extension lists asked for so far
  // G-BRE keeps its per-step host syncs to one: the photon bounds of step N are read back with the
  // planner's counters and size the grid of step N+1 (photons outside the grid sit in its border cells)
  float cachedB6[6] = {0, 0, 0, 0, 0, 0};
  bool haveCachedBounds = false, boundsPending = false;
  float *pinB6 = nullptr;      // pinned host staging: 6 floats + 2 uint32
  uint32_t *pinCtl = nullptr;

  // camera beams
  const gvpm_camera_ray *raysDev = nullptr;
  uint32_t nsets = 0;
  bool haveBeams = false, beamsDirty = false;

  // G-Beams: raw upload shares rawF/rawU/rawDev with the photons; end normals + sub-beam build
  DevBuf<float> endNOwned, subCentres;
  const float *endNDev = nullptr;
  bool haveBeamsMap = false;
  DevBuf<uint32_t> subCounts, subOffsets, subIds, beamCtl;
  DevBuf<float4> beamAux;  // G-Beams: {p1, bits} {direction, sub-beam length} per beam, what sub_hot_kernel gathers
  DevBuf<float2> beamClear;  // G-Beams: {cosA0, M1} per beam, the free cone of its reconnections (beam_near_kernel)
  uint32_t nsub = 0;
  float subLen = 0.f, maxSubLen = 0.f;

  // G-Planes: second edge of every plane + the 48-byte test records
  DevBuf<float> w1Owned, len1Owned;
  const float *w1Dev = nullptr, *len1Dev = nullptr;
  DevBuf<float4> planeTest;
  DevBuf<uint32_t> subFlags;      // G-Beams: filter bits per sorted sub-beam
  DevBuf<uint32_t> shiftExtent;   // G-Beams: max distance between a shifted camera ray and its base ray (float bits)
  bool beamNearStale = true;      // G-Beams: the per-beam near-occluder lists must be rebuilt
  DevBuf<uint32_t> blockKeyA, blockKeyB, blockValA, blockValB;  // G-Beams: pair blocks and their tiles, unsorted / sorted
  DevBuf<uint2> beamPairs;        // G-Beams: (beam | sub << 24, sorted set) pairs between traversal and evaluation
  bool havePlanes = false;

  // G-VPM: camera samples + per-pixel SPPM state
  DevBuf<gvpm_vpm_sample> samplesOwned;
  const gvpm_vpm_sample *samplesDev = nullptr;
  uint32_t nsamples = 0;
  bool haveSamples = false;
  DevBuf<float> scaleVol, nVol, mvol;
  DevBuf<uint32_t> maxScaleBits;
  double totalEmitted = 0;   // m_totalEmittedVolume

  // film
  DevBuf<float> accum, accumAll, iter, filmOut, emission;
  bool useAll = false;  // accumAll holds the all-reduced film until the next gather
  size_t npix = 0;
  float globalScaleVolume = 1.f;
  // G-BRE: the image tiles touched since the last reset (everything outside is exactly zero): the per-iteration
  // buffer is folded and cleared only there -- a rank of an image-sharded run owns a fraction of the frame
  // G-BRE keeps the running SUM of the per-iteration estimates in `accum` (the evaluation kernel adds straight into
  // it): the APA running mean (gvpm.cpp:1055-1069) is sum / it, applied by the readers.  No per-iteration buffer, no
  // fold kernel between two evaluation kernels.
  bool sumMode = false;
  int sumIt = 0;                  // `it` of the last gather (0: nothing accumulated)
  DevBuf<float> accumTmp;         // scaled copy for host downloads

  // stats / timing
  DevBuf<unsigned long long> stats;
  // HIP event brackets per phase: 0 = dominant kernel, 1 = BRE traversal, 2 = build (grid + sorts + plan)
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events[GVPM_PHASES];
  size_t eventsHead[GVPM_PHASES] = {0, 0, 0};   // next slot of the ring
  size_t eventsCount[GVPM_PHASES] = {0, 0, 0};  // launches recorded since the last poll, saturating at the ring size

  int beamsPerWave = 16;
  float cellScale = 0.f;  // GVPM_CELL_SCALE; 0: the technique's default (buildGrid)
  uint32_t planTarget = 1024;  // staged photons per work item
  bool planTargetSet = false;  // GVPM_PLAN_TARGET given (else G-Beams takes its own default)
  uint32_t nwaves = 2048;      // persistent gather waves
  bool nwavesFromEnv = false;
  uint32_t ncu = 256;
  uint32_t nwavesTrav = 4096;  // persistent traversal waves (G-BRE)
  // G-BRE: persistent waves pulling items from a queue, or one item per wave (GVPM_PERSISTENT: bit 0 evaluation, bit 1
  // traversal).  Measured at C2: the evaluation is faster persistent (0.97 against 1.11 ms: its 4-wave workgroups stage
  // the occluders once), the traversal one item per wave (0.57 against 0.78 ms beside the evaluation: the dispatcher
  // slots its workgroups, and the next build's kernels, in as others retire)
  bool persistentEval = true, persistentTrav = false;
  // per item and beam: photon index lists + their lengths

  // reconstruction scratch
  DevBuf<float> poissonScratch, poissonIO;
  PoissonGraphCache poissonGraph;

  // multi-GPU
  ncclComm_t comm = nullptr;
};

#define CHECK_H(h)                                \
  if (!(h)) return GVPM_ERR_INVALID_ARG;          \
  if (hipSetDevice((h)->device) != hipSuccess) {  \
    (h)->err = "hipSetDevice failed";             \
    return GVPM_ERR_HIP;                          \
  }

#define HIP_TRY(h, expr)                                                                      \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      (h)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                           \
      return GVPM_ERR_HIP;                                                                    \
    }                                                                                         \
  } while (0)

static int fail(gvpm_context *h, int code, const char *msg) {
  if (h) h->err = msg;
  return code;
}

// GPMConfig::load SLog(EError) cases that concern this path (gvpm/gvpm_struct.h:291-313)
static const char *validateParams(const gvpm_params *p) {
  if (p->abi_version != GVPM_ABI_VERSION) return "abi_version mismatch";
  if (p->width <= 0 || p->height <= 0 || p->width > 65535 || p->height > 65535) return "bad film size";
  if (p->vol_technique < GVPM_VOL_BRE2D || p->vol_technique > GVPM_VOL_PLANE0D) return "unknown vol_technique";
  const bool use3D = p->vol_technique == GVPM_DISTANCE || p->vol_technique == GVPM_VOL_BRE3D ||
                     (p->vol_technique >= GVPM_BEAM_BEAM_3D_NAIVE && p->vol_technique <= GVPM_BEAM_BEAM_3D_OPTIMIZED);
  if (p->use_shift_null && !use3D && p->vol_technique != GVPM_BEAM_BEAM_1D)
    return "Not possible to shift null without using 3D kernel";
  if (p->max_depth <= 1 && p->max_depth != -1 && p->max_depth != 0) return "Maximum depth must be set to \"2\" or higher!";
  if (!(p->bsphere_radius > 0.f)) return "bsphere_radius must be positive";
  if (!(p->epsilon > 0.f) || !(p->shadow_epsilon > 0.f)) return "epsilon / shadow_epsilon must be positive";
  if (p->vol_technique == GVPM_VOL_PLANE0D && p->min_depth < 2) return "Impossible to use plane with minDepth smaller than 2";
  if (!p->no_medium_shift) return "noMediumShift=false is not supported (shiftPhotonMedium is SAssert(false))";
  return nullptr;
}

extern "C" {

int gvpm_abi_version(void) { return GVPM_ABI_VERSION; }

const char *gvpm_last_error(const gvpm_context *h) { return h ? h->err.c_str() : "null handle"; }

int gvpm_create(const gvpm_params *params, int device, gvpm_context **out) {
  if (!params || !out) return GVPM_ERR_INVALID_ARG;
  *out = nullptr;
  if (validateParams(params)) {
    // parameter errors the reference raises at load time
    const char *m = validateParams(params);
    return (strstr(m, "shift null") || strstr(m, "noMediumShift")) ? GVPM_ERR_UNSUPPORTED : GVPM_ERR_INVALID_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return GVPM_ERR_NO_DEVICE;
  if (hipSetDevice(device) != hipSuccess) return GVPM_ERR_NO_DEVICE;
  gvpm_context *h = new gvpm_context();
  h->device = device;
  h->cfg = *params;
  // stream priorities (lower = more urgent): the traversal of step N+1 is what the next evaluation waits for, the
  // build of step N+2 is two steps ahead.  With equal priorities about one process in four ran the build's small
  // kernels in front of the traversal's workgroups and lost the third stage's gain (5.5 instead of 5.8 G/s);
  // these priorities make that rarer
  int prA = 0, prB = 1, prC = -1;  // gather / build / traversal
  {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess) {
      prB = std::min(prB, least);
      prC = std::max(prC, greatest);
    }
  }
  if (const char *e = getenv("GVPM_STREAM_PRIORITIES")) (void)sscanf(e, "%d,%d,%d", &prA, &prB, &prC);
  if (hipStreamCreateWithPriority(&h->stream, hipStreamNonBlocking, prA) != hipSuccess ||
      hipStreamCreateWithPriority(&h->streamB, hipStreamNonBlocking, prB) != hipSuccess ||
      hipStreamCreateWithPriority(&h->streamC, hipStreamNonBlocking, prC) != hipSuccess ||
      hipEventCreateWithFlags(&h->sets[0].lastUse, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->sets[1].lastUse, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->sets[2].lastUse, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->sets[0].traversed, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->sets[1].traversed, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&h->sets[2].traversed, hipEventDisableTiming) != hipSuccess) {
    gvpm_destroy(h);
    return GVPM_ERR_HIP;
  }
  if (hipStreamCreateWithFlags(&h->copyStream, hipStreamNonBlocking) != hipSuccess) {
    gvpm_destroy(h);
    return GVPM_ERR_HIP;
  }
  for (int k = 0; k < 3; ++k)
    if (hipEventCreateWithFlags(&h->phSlot[k].copied, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->phSlot[k].consumed, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->phSlot[k].consumedB, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->raySlot[k].copied, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->raySlot[k].freed, hipEventDisableTiming) != hipSuccess) {
      gvpm_destroy(h);
      return GVPM_ERR_HIP;
    }
  h->bstream = h->stream;
  if (const char *e = getenv("GVPM_BEAMS_PER_WAVE")) {
    int v = atoi(e);
    if (v == 16 || v == 32 || v == 64) h->beamsPerWave = v;
  }
  // development switch (perf attribution only): bit 0 = skip the evaluations, keep the traversal
  h->cfg.reserved[0] = 0;
  if (const char *e = getenv("GVPM_DEBUG_FLAGS")) h->cfg.reserved[0] = atoi(e);
  if (const char *e = getenv("GVPM_RECORD_PREFETCH")) h->cfg.reserved[0] |= atoi(e) ? 64 : 32;  // (gather_bre.hip: launch_evaluate_bre)
  h->cfg.reserved[1] = 0;  // slab layers per step (0 = default)
  if (const char *e = getenv("GVPM_SLAB_LAYERS")) h->cfg.reserved[1] = atoi(e);
  h->cfg.reserved[2] = 0;  // slab layers per step when x is the major axis (0 = default)
  if (const char *e = getenv("GVPM_SLAB_LAYERS_X")) h->cfg.reserved[2] = atoi(e);
  h->cfg.reserved[3] = 0;  // G-BRE traversal: staged photons per box row set from which the staging is lane-coalesced (0 = default)
  if (const char *e = getenv("GVPM_COALESCE_AT")) h->cfg.reserved[3] = atoi(e);
  if (const char *e = getenv("GVPM_PLAN_TARGET")) {
    int v = atoi(e);
    if (v >= 64 && v <= (1 << 24)) h->planTarget = (uint32_t)v, h->planTargetSet = true;
  }
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
      h->nwaves = (uint32_t)prop.multiProcessorCount * 8u, h->ncu = (uint32_t)prop.multiProcessorCount;
    if (const char *e = getenv("GVPM_WAVES_PER_CU")) {
      int v = atoi(e);
      if (v >= 1 && v <= 32 && prop.multiProcessorCount > 0) h->nwaves = (uint32_t)prop.multiProcessorCount * v, h->nwavesFromEnv = true;
    }
  }
  h->nwavesTrav = h->ncu * 16u;
  if (const char *e = getenv("GVPM_TRAV_WAVES_PER_CU")) {
    int v = atoi(e);
    if (v >= 1 && v <= 32) h->nwavesTrav = h->ncu * (uint32_t)v;
  }
  {
    if (h->nwavesTrav > GVPM_STAT_ROWS) h->nwavesTrav = GVPM_STAT_ROWS;
    if (h->nwaves > GVPM_STAT_ROWS) h->nwaves = GVPM_STAT_ROWS;
  }
  if (const char *e = getenv("GVPM_PIPELINE")) h->pipeline = atoi(e) != 0;
  // alone on the GPU (single stream) the evaluation runs best at its full 3 waves per SIMD; beside the traversal and the
  // build of the following steps 8 per CU leave them room (measured: 12 per CU costs the pipelined step 3-4 %)
  if (!h->pipeline && !getenv("GVPM_WAVES_PER_CU") && h->ncu) h->nwaves = std::min<uint32_t>(h->ncu * 12u, GVPM_STAT_ROWS);
  if (const char *e = getenv("GVPM_PERSISTENT")) h->persistentEval = (atoi(e) & 1) != 0, h->persistentTrav = (atoi(e) & 2) != 0;
  if (const char *e = getenv("GVPM_TRAV_ON_BUILD")) h->travOnBuild = atoi(e) != 0;
  if (const char *e = getenv("GVPM_NEAR_GRID")) h->useNearGrid = atoi(e) != 0;
  if (const char *e = getenv("GVPM_TRAV_STREAM")) h->travStream = atoi(e) != 0;
  if (const char *e = getenv("GVPM_BEAMS_FP64")) h->beamsExact = atoi(e) != 0;
  if (const char *e = getenv("GVPM_BEAMS_FREE_CONE")) h->beamsFreeCone = atoi(e) != 0;
  if (const char *e = getenv("GVPM_PLAN_BOXES")) h->planBoxHandOff = atoi(e) != 0;
  if (const char *e = getenv("GVPM_BUNDLE")) h->bundleEnabled = atoi(e) != 0;
  if (const char *e = getenv("GVPM_BUNDLE_DIV")) {
    const float v = (float)atof(e);
    if (v >= 0.25f && v <= 16.f) h->bundleDiv = v;
  }
  if (const char *e = getenv("GVPM_BEAM_PAIRS_INIT")) {
    const long long v = atoll(e);
    if (v >= 64 && v <= ((long long)1 << 31)) h->beamPairsInit = (size_t)v;
  }
  if (const char *e = getenv("GVPM_BEAM_ITEMS_INIT")) {
    const long long v = atoll(e);
    if (v >= 1 && v <= ((long long)1 << 30)) h->beamItemsInit = (uint32_t)v;
  }
  if (const char *e = getenv("GVPM_CELL_SCALE")) {
    float v = (float)atof(e);
    if (v >= 0.25f && v <= 8.f) h->cellScale = v;
  }
  h->npix = (size_t)params->width * params->height;
  if (h->accum.ensure(h->npix * 27) != hipSuccess || h->iter.ensure(h->npix * 27) != hipSuccess ||
      h->stats.ensure(8 * GVPM_STAT_ROWS) != hipSuccess || h->scaleVol.ensure(h->npix) != hipSuccess ||
      h->nVol.ensure(h->npix) != hipSuccess || h->mvol.ensure(h->npix) != hipSuccess ||
      h->maxScaleBits.ensure(2) != hipSuccess) {
    gvpm_destroy(h);
    return GVPM_ERR_HIP;
  }
  *out = h;
  int rc = gvpm_reset(h);
  if (rc != GVPM_OK) {
    gvpm_destroy(h);
    *out = nullptr;
  }
  return rc;
}

int gvpm_destroy(gvpm_context *h) {
  if (!h) return GVPM_ERR_INVALID_ARG;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->comm && g_rccl.commDestroy) g_rccl.commDestroy(h->comm);
  for (auto &v : h->events)
    for (auto &e : v) {
      (void)hipEventDestroy(e.first);
      (void)hipEventDestroy(e.second);
    }
  if (h->streamB) (void)hipStreamSynchronize(h->streamB);
  if (h->streamC) (void)hipStreamSynchronize(h->streamC);
  for (BuildSet &b : h->sets) b.release();
  h->tri4.release(); h->bvh.release();
  for (auto &ps : h->phSlot) {
    ps.raw.release();
    if (ps.copied) (void)hipEventDestroy(ps.copied);
    if (ps.consumed) (void)hipEventDestroy(ps.consumed);
    if (ps.consumedB) (void)hipEventDestroy(ps.consumedB);
  }
  for (auto &rs : h->raySlot) {
    rs.rays.release();
    if (rs.copied) (void)hipEventDestroy(rs.copied);
    if (rs.freed) (void)hipEventDestroy(rs.freed);
  }
  if (h->copyStream) (void)hipStreamDestroy(h->copyStream);
  h->endNOwned.release(); h->subCentres.release(); h->subCounts.release(); h->subOffsets.release();
  h->subIds.release(); h->beamCtl.release(); h->beamAux.release(); h->beamClear.release();
  h->nearGridStart.release(); h->nearGridTris.release(); h->nearGridCount.release();
  h->w1Owned.release(); h->len1Owned.release(); h->planeTest.release(); h->beamPairs.release(); h->subFlags.release(); h->shiftExtent.release();
  h->blockKeyA.release(); h->blockKeyB.release(); h->blockValA.release(); h->blockValB.release();
  h->samplesOwned.release(); h->scaleVol.release(); h->nVol.release(); h->mvol.release(); h->maxScaleBits.release();
  poisson_graph_release(h->poissonGraph);
  h->accumTmp.release();
  h->poissonScratch.release(); h->poissonIO.release();
  h->accum.release(); h->accumAll.release(); h->iter.release(); h->filmOut.release(); h->emission.release(); h->stats.release();
  if (h->pinB6) (void)hipHostFree(h->pinB6);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  if (h->streamB) (void)hipStreamDestroy(h->streamB);
  if (h->streamC) (void)hipStreamDestroy(h->streamC);
  delete h;
  return GVPM_OK;
}

int gvpm_reset(gvpm_context *h) {
  CHECK_H(h);
  HIP_TRY(h, hipMemsetAsync(h->accum.p, 0, h->npix * 27 * sizeof(float), h->stream));
  HIP_TRY(h, hipMemsetAsync(h->stats.p, 0, 8 * GVPM_STAT_ROWS * sizeof(unsigned long long), h->stream));
  h->globalScaleVolume = h->cfg.initial_scale_volume;  // gvpm.cpp:291
  h->sumIt = 0;
  h->bundleState = 0;
  h->bundleViolations = 0;
  HIP_TRY(h, hipMemsetAsync(h->iter.p, 0, h->npix * 27 * sizeof(float), h->stream));
  for (size_t &u : h->eventsHead) u = 0;
  for (size_t &u : h->eventsCount) u = 0;
  h->useAll = false;
  h->totalEmitted = 0;
  {
    // newGP.scaleVol = initialScaleVolume, NVol = 0 (gvpm.cpp:285-288)
    std::vector<float> init(h->npix, h->cfg.initial_scale_volume);
    HIP_TRY(h, hipMemcpyAsync(h->scaleVol.p, init.data(), h->npix * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemsetAsync(h->nVol.p, 0, h->npix * sizeof(float), h->stream));
    uint32_t bits;
    const float sc = h->cfg.initial_scale_volume;
    memcpy(&bits, &sc, 4);
    HIP_TRY(h, hipMemcpyAsync(h->maxScaleBits.p, &bits, 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  return GVPM_OK;
}

int gvpm_upload_scene(gvpm_context *h, const gvpm_triangles *t) {
  CHECK_H(h);
  if (!t || (t->n && (!t->v0 || !t->e1 || !t->e2))) return fail(h, GVPM_ERR_INVALID_ARG, "null triangle arrays");
  // occluder BVH on the host; triangles packed {v0,n.x} {e1,n.y} {e2,n.z} in leaf order
  BvhBuild bvh;
  buildSceneBvh(t->v0, t->e1, t->e2, t->n, bvh);
  std::vector<float> packed(12 * (size_t)t->n + 12, 0.f);
  for (uint32_t k = 0; k < t->n; ++k) {
    const uint32_t i = bvh.order[k];
    const float *a = t->v0 + 3 * (size_t)i, *b = t->e1 + 3 * (size_t)i, *c = t->e2 + 3 * (size_t)i;
    float n[3] = {b[1] * c[2] - b[2] * c[1], b[2] * c[0] - b[0] * c[2], b[0] * c[1] - b[1] * c[0]};
    const float l = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    for (int q = 0; q < 3; ++q) n[q] = l > 0.f ? n[q] / l : 0.f;
    float *o = packed.data() + 12 * (size_t)k;
    for (int q = 0; q < 3; ++q) {
      o[q] = a[q];
      o[4 + q] = b[q];
      o[8 + q] = c[q];
      o[4 * q + 3] = n[q];
    }
  }
  HIP_TRY(h, h->tri4.ensure(3 * (size_t)t->n + 3));
  HIP_TRY(h, h->bvh.ensure(bvh.nodes.size() / 4 + 2));
  HIP_TRY(h, hipMemcpyAsync(h->tri4.p, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(h->bvh.p, bvh.nodes.data(), bvh.nodes.size() * sizeof(float), hipMemcpyHostToDevice,
                            h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  h->ntri = t->n;
  for (int c = 0; c < 3; ++c) {
    h->triMin[c] = INFINITY;
    h->triMax[c] = -INFINITY;
  }
  for (uint32_t i = 0; i < t->n; ++i)
    for (int c = 0; c < 3; ++c) {
      const float a = t->v0[3 * i + c], b = a + t->e1[3 * i + c], d = a + t->e2[3 * i + c];
      h->triMin[c] = fminf(h->triMin[c], fminf(a, fminf(b, d)));
      h->triMax[c] = fmaxf(h->triMax[c], fmaxf(a, fmaxf(b, d)));
    }
  h->photonsDirty = h->havePhotons;  // the near-occluder lists depend on the scene
  h->nearGridReach = -1.f;
  h->nearGrid = NearGrid();
  return GVPM_OK;
}

int gvpm_upload_medium(gvpm_context *h, const gvpm_medium *m) {
  CHECK_H(h);
  if (!m) return fail(h, GVPM_ERR_INVALID_ARG, "null medium");
  // homogeneous.cpp:196-200: the balance strategy requires equal sigma_t across channels
  if (m->sigma_t[0] != m->sigma_t[1] || m->sigma_t[0] != m->sigma_t[2])
    return fail(h, GVPM_ERR_UNSUPPORTED, "Not possible to have different albedo values...");
  if (!(m->sigma_t[0] > 0.f)) return fail(h, GVPM_ERR_INVALID_ARG, "sigma_t must be positive");
  h->medium = *m;
  h->haveMedium = true;
  return GVPM_OK;
}

static bool isPinnedHost(const void *ptr) {
  hipPointerAttribute_t attr;
  if (!ptr || hipPointerGetAttributes(&attr, ptr) != hipSuccess) {
    (void)hipGetLastError();  // pageable memory is reported as an error: not one of ours
    return false;
  }
  return attr.type == hipMemoryTypeHost;
}

int gvpm_host_alloc(uint64_t bytes, void **out) {
  if (!out || bytes == 0) return GVPM_ERR_INVALID_ARG;
  return hipHostMalloc(out, bytes, hipHostMallocDefault) == hipSuccess ? GVPM_OK : GVPM_ERR_HIP;
}
int gvpm_host_free(void *p) { return hipHostFree(p) == hipSuccess ? GVPM_OK : GVPM_ERR_HIP; }
int gvpm_host_alloc_photons(uint64_t n, gvpm_photon_soa *view, void **block) {
  if (!view || !block || n == 0 || n > 0x7FFFFFF0ull) return GVPM_ERR_INVALID_ARG;
  if (hipHostMalloc(block, (size_t)n * 30 * 4, hipHostMallocDefault) != hipSuccess) return GVPM_ERR_HIP;
  float *f = (float *)*block;
  const float **v3[8] = {&view->pos, &view->wi, &view->flux, &view->parent_pos, &view->parent_n, &view->prefix_w,
                         &view->parent_scat, &view->parent_wi};
  const float **v1[4] = {&view->parent_pdf, &view->edge_pdf, &view->parent_rr, &view->parent_g};
  for (auto q : v3) { *q = f; f += (size_t)n * 3; }
  for (auto q : v1) { *q = f; f += n; }
  view->flags = (const uint32_t *)f;
  view->path_id = (const uint32_t *)f + n;
  view->n = n;
  return GVPM_OK;
}

static int uploadPhotonsCommon(gvpm_context *h, const gvpm_photon_soa *p, bool fromDevice, bool prefetch = false) {
  if (!p) return fail(h, GVPM_ERR_INVALID_ARG, "null photon soa");
  if (p->n > 0x7FFFFFF0ull) return fail(h, GVPM_ERR_INVALID_ARG, "too many photons");
  const uint32_t n = (uint32_t)p->n;
  if (n) {
    const void *ptrs[] = {p->pos, p->wi, p->flux, p->parent_pos, p->parent_n, p->prefix_w, p->parent_scat,
                          p->parent_wi, p->parent_pdf, p->edge_pdf, p->parent_rr, p->parent_g, p->flags, p->path_id};
    for (const void *q : ptrs)
      if (!q) return fail(h, GVPM_ERR_INVALID_ARG, "null photon array");
  }
  if (fromDevice) {
    if (prefetch) return fail(h, GVPM_ERR_INVALID_ARG, "prefetch takes host buffers");
    h->rawDev = *p;
    h->phWait = false;
    h->photonsOwnedCur = false;
  } else {
    // pinned only if EVERY array is: one pageable array makes the runtime stage that copy itself, and the call must then
    // not return before the copy stream has drained (the header's contract: the library copies during the call)
    bool pinned = n > 0;
    {
      const void *all[14] = {p->pos, p->wi, p->flux, p->parent_pos, p->parent_n, p->prefix_w, p->parent_scat, p->parent_wi,
                             p->parent_pdf, p->edge_pdf, p->parent_rr, p->parent_g, p->flags, p->path_id};
      // (one block in the ABI's order -- gvpm_host_alloc_photons -- is one allocation: its first array speaks for all)
      bool oneBlock = n > 0;
      size_t off = 0;
      for (int k = 0; k < 14 && oneBlock; ++k) {
        oneBlock = (const char *)all[k] == (const char *)all[0] + off * 4;
        off += (size_t)n * (k < 8 ? 3 : 1);
      }
      for (int k = 0; k < (oneBlock ? 1 : 14) && pinned; ++k) pinned = isPinnedHost(all[k]);
    }
    if (prefetch && !pinned) return fail(h, GVPM_ERR_INVALID_ARG, "gvpm_prefetch_photons needs pinned host memory (gvpm_host_alloc*)");
    if (prefetch && h->phPending >= 0) return fail(h, GVPM_ERR_STATE, "a prefetched photon set is already pending");
    int slot = (h->phCur + 1) % 3;
    if (slot == h->phPending) slot = (h->phCur + 2) % 3;
    gvpm_context::PhotonSlot &ps = h->phSlot[slot];
    // the kernels of the gathers that last read this slot (its build; the G-Planes gather) may still be running
    if (ps.read) {
      HIP_TRY(h, hipStreamWaitEvent(h->copyStream, ps.consumed, 0));
      HIP_TRY(h, hipStreamWaitEvent(h->copyStream, ps.consumedB, 0));
    }
    ps.read = false;
    if (ps.raw.cap < (size_t)n * 30 + 8) {
      // regrowing frees the old buffer
      HIP_TRY(h, hipStreamSynchronize(h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->streamB));
      HIP_TRY(h, ps.raw.ensure((size_t)n * 30 + 8));
    }
    const void *src[14] = {p->pos, p->wi, p->flux, p->parent_pos, p->parent_n, p->prefix_w, p->parent_scat, p->parent_wi,
                           p->parent_pdf, p->edge_pdf, p->parent_rr, p->parent_g, p->flags, p->path_id};
    const void **dst[14] = {(const void **)&ps.dev.pos, (const void **)&ps.dev.wi, (const void **)&ps.dev.flux,
                            (const void **)&ps.dev.parent_pos, (const void **)&ps.dev.parent_n, (const void **)&ps.dev.prefix_w,
                            (const void **)&ps.dev.parent_scat, (const void **)&ps.dev.parent_wi, (const void **)&ps.dev.parent_pdf,
                            (const void **)&ps.dev.edge_pdf, (const void **)&ps.dev.parent_rr, (const void **)&ps.dev.parent_g,
                            (const void **)&ps.dev.flags, (const void **)&ps.dev.path_id};
    // one packed copy when the host arrays are one block in the ABI's order (gvpm_host_alloc_photons), else one each
    bool packed = n > 0;
    size_t off = 0;
    for (int k = 0; k < 14 && packed; ++k) {
      packed = (const char *)src[k] == (const char *)src[0] + off * 4;
      off += (size_t)n * (k < 8 ? 3 : 1);
    }
    off = 0;
    for (int k = 0; k < 14; ++k) {
      const size_t words = (size_t)n * (k < 8 ? 3 : 1);
      *dst[k] = ps.raw.p + off;
      if (n && !packed) HIP_TRY(h, hipMemcpyAsync(ps.raw.p + off, src[k], words * 4, hipMemcpyHostToDevice, h->copyStream));
      off += words;
    }
    if (packed) HIP_TRY(h, hipMemcpyAsync(ps.raw.p, src[0], off * 4, hipMemcpyHostToDevice, h->copyStream));
    ps.dev.n = n;
    HIP_TRY(h, hipEventRecord(ps.copied, h->copyStream));
    // pageable memory: the caller may reuse its buffers when this returns.  Pinned memory (gvpm_host_alloc*): the copy is
    // left in flight; the buffer must stay untouched until the gather that consumes it has returned (G-BRE) or the
    // handle was synchronised
    if (!pinned) HIP_TRY(h, hipStreamSynchronize(h->copyStream));
    if (prefetch) {
      h->phPending = slot;
      return GVPM_OK;
    }
    h->phCur = slot;
    h->rawDev = ps.dev;
    h->phWait = true;
    h->photonsOwnedCur = true;
  }
  h->nph = n;
  h->havePhotons = true;
  h->photonsDirty = true;
  return GVPM_OK;
}

int gvpm_upload_photons(gvpm_context *h, const gvpm_photon_soa *p) {
  CHECK_H(h);
  return uploadPhotonsCommon(h, p, false);
}
int gvpm_upload_photons_dev(gvpm_context *h, const gvpm_photon_soa *p) {
  CHECK_H(h);
  return uploadPhotonsCommon(h, p, true);
}
int gvpm_prefetch_photons(gvpm_context *h, const gvpm_photon_soa *p) {
  CHECK_H(h);
  return uploadPhotonsCommon(h, p, false, true);
}

static int uploadPhotonBeamsCommon(gvpm_context *h, const gvpm_photon_soa *b, const float *end_n, bool fromDevice) {
  if (b && b->n && !end_n) return fail(h, GVPM_ERR_INVALID_ARG, "null end_n");
  if (b && b->n > 0xFFFFFFull) return fail(h, GVPM_ERR_INVALID_ARG, "too many beams (24-bit beam index)");
  int rc = uploadPhotonsCommon(h, b, fromDevice);
  if (rc != GVPM_OK) return rc;
  if (fromDevice) {
    h->endNDev = end_n;
  } else {
    HIP_TRY(h, h->endNOwned.ensure((size_t)h->nph * 3 + 4));
    if (h->nph) {
      HIP_TRY(h, hipMemcpyAsync(h->endNOwned.p, end_n, (size_t)h->nph * 12, hipMemcpyHostToDevice, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    h->endNDev = h->endNOwned.p;
  }
  h->haveBeamsMap = true;
  return GVPM_OK;
}

static int uploadPlanesCommon(gvpm_context *h, const gvpm_photon_soa *b, const float *w1, const float *len1,
                              bool fromDevice) {
  if (b && b->n && (!w1 || !len1)) return fail(h, GVPM_ERR_INVALID_ARG, "null w1 / len1");
  int rc = uploadPhotonsCommon(h, b, fromDevice);
  if (rc != GVPM_OK) return rc;
  if (fromDevice) {
    h->w1Dev = w1;
    h->len1Dev = len1;
  } else {
    HIP_TRY(h, h->w1Owned.ensure((size_t)h->nph * 3 + 4));
    HIP_TRY(h, h->len1Owned.ensure((size_t)h->nph + 4));
    if (h->nph) {
      HIP_TRY(h, hipMemcpyAsync(h->w1Owned.p, w1, (size_t)h->nph * 12, hipMemcpyHostToDevice, h->stream));
      HIP_TRY(h, hipMemcpyAsync(h->len1Owned.p, len1, (size_t)h->nph * 4, hipMemcpyHostToDevice, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    h->w1Dev = h->w1Owned.p;
    h->len1Dev = h->len1Owned.p;
  }
  h->havePlanes = true;
  return GVPM_OK;
}

int gvpm_upload_planes(gvpm_context *h, const gvpm_photon_soa *beams, const float *w1, const float *len1) {
  CHECK_H(h);
  return uploadPlanesCommon(h, beams, w1, len1, false);
}
int gvpm_upload_planes_dev(gvpm_context *h, const gvpm_photon_soa *beams, const float *w1, const float *len1) {
  CHECK_H(h);
  return uploadPlanesCommon(h, beams, w1, len1, true);
}

int gvpm_upload_beams(gvpm_context *h, const gvpm_photon_soa *beams, const float *end_n) {
  CHECK_H(h);
  return uploadPhotonBeamsCommon(h, beams, end_n, false);
}
int gvpm_upload_beams_dev(gvpm_context *h, const gvpm_photon_soa *beams, const float *end_n) {
  CHECK_H(h);
  return uploadPhotonBeamsCommon(h, beams, end_n, true);
}

static int uploadBeamsCommon(gvpm_context *h, const gvpm_camera_ray *rays, uint64_t nsets, bool fromDevice,
                             bool prefetch = false) {
  if (nsets && !rays) return fail(h, GVPM_ERR_INVALID_ARG, "null camera rays");
  if (nsets > 0x0FFFFFFFull) return fail(h, GVPM_ERR_INVALID_ARG, "too many beam sets");
  if (fromDevice) {
    if (prefetch) return fail(h, GVPM_ERR_INVALID_ARG, "prefetch takes host buffers");
    h->raysDev = rays;
    h->rayWait = false;
    h->raysOwnedCur = false;
  } else {
    const bool pinned = nsets && isPinnedHost(rays);
    if (prefetch && !pinned) return fail(h, GVPM_ERR_INVALID_ARG, "gvpm_prefetch_camera_beams needs pinned host memory (gvpm_host_alloc)");
    if (prefetch && h->rayPending >= 0) return fail(h, GVPM_ERR_STATE, "a prefetched camera-beam list is already pending");
    int slot = (h->rayCur + 1) % 3;
    if (slot == h->rayPending) slot = (h->rayCur + 2) % 3;
    gvpm_context::RaySlot &rs = h->raySlot[slot];
    // the evaluation kernel of the step that last used this slot may still be reading it
    if (rs.read) HIP_TRY(h, hipStreamWaitEvent(h->copyStream, rs.freed, 0));
    rs.read = false;
    if (rs.rays.cap < (size_t)nsets * 5 + 1) {
      HIP_TRY(h, hipStreamSynchronize(h->stream));  // regrowing frees the old buffer
      HIP_TRY(h, rs.rays.ensure((size_t)nsets * 5 + 1));
    }
    if (nsets)
      HIP_TRY(h, hipMemcpyAsync(rs.rays.p, rays, (size_t)nsets * 5 * sizeof(gvpm_camera_ray), hipMemcpyHostToDevice,
                                h->copyStream));
    rs.nsets = (uint32_t)nsets;
    HIP_TRY(h, hipEventRecord(rs.copied, h->copyStream));
    if (!pinned) HIP_TRY(h, hipStreamSynchronize(h->copyStream));
    if (prefetch) {
      h->rayPending = slot;
      return GVPM_OK;
    }
    h->rayCur = slot;
    h->raysDev = rs.rays.p;
    h->rayWait = true;
    h->raysOwnedCur = true;
  }
  h->nsets = (uint32_t)nsets;
  h->haveBeams = true;
  h->beamsDirty = true;
  return GVPM_OK;
}

int gvpm_upload_camera_beams(gvpm_context *h, const gvpm_camera_ray *rays, uint64_t n_sets) {
  CHECK_H(h);
  return uploadBeamsCommon(h, rays, n_sets, false);
}
int gvpm_upload_camera_beams_dev(gvpm_context *h, const gvpm_camera_ray *rays, uint64_t n_sets) {
  CHECK_H(h);
  return uploadBeamsCommon(h, rays, n_sets, true);
}
int gvpm_prefetch_camera_beams(gvpm_context *h, const gvpm_camera_ray *rays, uint64_t n_sets) {
  CHECK_H(h);
  return uploadBeamsCommon(h, rays, n_sets, false, true);
}

static int uploadSamplesCommon(gvpm_context *h, const gvpm_vpm_sample *smp, uint64_t n, bool fromDevice) {
  if (n && !smp) return fail(h, GVPM_ERR_INVALID_ARG, "null vpm samples");
  if (n > 0x7FFFFFF0ull) return fail(h, GVPM_ERR_INVALID_ARG, "too many vpm samples");
  if (fromDevice) {
    h->samplesDev = smp;
  } else {
    HIP_TRY(h, h->samplesOwned.ensure(n + 1));
    if (n) {
      HIP_TRY(h, hipMemcpyAsync(h->samplesOwned.p, smp, n * sizeof(gvpm_vpm_sample), hipMemcpyHostToDevice, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    h->samplesDev = h->samplesOwned.p;
  }
  h->nsamples = (uint32_t)n;
  h->haveSamples = true;
  return GVPM_OK;
}

int gvpm_upload_vpm_samples(gvpm_context *h, const gvpm_vpm_sample *samples, uint64_t n) {
  CHECK_H(h);
  return uploadSamplesCommon(h, samples, n, false);
}
int gvpm_upload_vpm_samples_dev(gvpm_context *h, const gvpm_vpm_sample *samples, uint64_t n) {
  CHECK_H(h);
  return uploadSamplesCommon(h, samples, n, true);
}

static float currentRadius(const gvpm_context *h) {
  // breInitSize = bsphere.radius * globalScaleVolume * POURCENTAGE_BS, gvpm.cpp:989 (Float = float)
  return h->cfg.bsphere_radius * h->globalScaleVolume * 0.01f;
}

static int ilog2ceil(uint32_t v) {
  int b = 0;
  while ((1ull << b) < v) ++b;
  return b;
}

// uniform grid over the photons for kernel radius r.  deferred: use the bounds of the previous
// photon set when there is one and leave this set's bounds in flight (pinB6) for the caller's sync.
// The occluder grid of the near-occluder lists (grid_build.hip: near_grid_kernel, nearVisit): count, scan, fill.  Once per
// scene: the one host read-back (the number of entries) stalls nothing that matters.
static int buildNearGrid(gvpm_context *h, float reach) {
  NearGrid g{};
  float ext[3], vol = 1.f;
  for (int c = 0; c < 3; ++c) {
    ext[c] = std::max(h->triMax[c] - h->triMin[c] + 2.02f * reach, 1e-6f);
    vol *= ext[c];
  }
  // about 64^3 cells, cubic ones: per axis extent / cell, cell = (volume / 64^3)^(1/3), at least 1, at most 256
  const float cell = std::cbrt(vol / 262144.f);
  size_t ncells = 1;
  for (int c = 0; c < 3; ++c) {
    g.dim[c] = std::max(1, std::min(256, (int)std::ceil(ext[c] / std::max(cell, 1e-9f))));
    g.org[c] = h->triMin[c] - 1.01f * reach;
    g.inv[c] = (float)g.dim[c] / ext[c];
    ncells *= (size_t)g.dim[c];
  }
  hipStream_t st = h->bstream;
  HIP_TRY(h, h->nearGridStart.ensure(ncells + 1));
  HIP_TRY(h, h->nearGridCount.ensure(ncells + 1));
  HIP_TRY(h, hipMemsetAsync(h->nearGridCount.p, 0, (ncells + 1) * sizeof(uint32_t), st));
  launch_near_grid(h->tri4.p, h->ntri, g, reach, h->nearGridCount.p, nullptr, 0, st);
  HIP_TRY(h, exclusiveSumU32(h->bs->sortTmp, h->nearGridCount.p, h->nearGridStart.p, (uint32_t)ncells + 1, st));
  uint32_t total = 0;
  HIP_TRY(h, hipMemcpyAsync(&total, h->nearGridStart.p + ncells, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(h, hipStreamSynchronize(st));
  HIP_TRY(h, h->nearGridTris.ensure((size_t)total + 1));
  HIP_TRY(h, hipMemsetAsync(h->nearGridCount.p, 0, (ncells + 1) * sizeof(uint32_t), st));
  g.start = h->nearGridStart.p;
  g.tris = h->nearGridTris.p;
  launch_near_grid(h->tri4.p, h->ntri, g, reach, h->nearGridCount.p, h->nearGridTris.p, 1, st);
  HIP_TRY(h, hipGetLastError());
  h->nearGrid = g;
  h->nearGridReach = reach;
  return GVPM_OK;
}

// The frame of the camera beams' bundle: common origin, mean direction, (u, v) range.  Two small reductions and a host
// wait, once; afterwards every planner run checks its rays against the frame (GatherArgs::bundleFlag).
static int fitBundle(gvpm_context *h) {
  h->bundleState = -1;
  if (!h->nsets || !h->raysDev) return GVPM_OK;
  HIP_TRY(h, h->bs->bounds6.ensure(32));
  double *dev = reinterpret_cast<double *>(h->bs->bounds6.p);
  double o[13];
  Grid g{};
  launch_bundle_fit(h->raysDev, h->nsets, 0, g, dev, h->bstream);
  HIP_TRY(h, hipMemcpyAsync(o, dev, sizeof(o), hipMemcpyDeviceToHost, h->bstream));
  HIP_TRY(h, hipStreamSynchronize(h->bstream));
  for (int k = 0; k < 13; ++k)
    if (!std::isfinite(o[k])) return GVPM_OK;
  const double cnt = o[12];
  if (!(cnt >= 1.0)) return GVPM_OK;
  // C = M^-1 b, M symmetric (adjugate); rays of one direction only (an orthographic sensor) leave M singular
  const double xx = o[0], xy = o[1], xz = o[2], yy = o[3], yz = o[4], zz = o[5];
  const double c00 = yy * zz - yz * yz, c01 = xz * yz - xy * zz, c02 = xy * yz - xz * yy;
  const double c11 = xx * zz - xz * xz, c12 = xy * xz - xx * yz, c22 = xx * yy - xy * xy;
  const double det = xx * c00 + xy * c01 + xz * c02;
  if (!(det > 1e-9 * cnt * cnt * cnt)) return GVPM_OK;
  const double C[3] = {(c00 * o[6] + c01 * o[7] + c02 * o[8]) / det, (c01 * o[6] + c11 * o[7] + c12 * o[8]) / det,
                       (c02 * o[6] + c12 * o[7] + c22 * o[8]) / det};
  double A[3] = {o[9], o[10], o[11]};
  const double len = std::sqrt(A[0] * A[0] + A[1] * A[1] + A[2] * A[2]);
  if (!(len > 0.2 * cnt)) return GVPM_OK;  // directions all over the sphere
  int sm = 0;
  for (int c = 0; c < 3; ++c) {
    A[c] /= len;
    if (fabs(A[c]) < fabs(A[sm])) sm = c;
  }
  double E[3] = {0, 0, 0}, U[3], V[3];
  E[sm] = 1.0;
  U[0] = A[1] * E[2] - A[2] * E[1]; U[1] = A[2] * E[0] - A[0] * E[2]; U[2] = A[0] * E[1] - A[1] * E[0];
  const double ul = std::sqrt(U[0] * U[0] + U[1] * U[1] + U[2] * U[2]);
  for (int c = 0; c < 3; ++c) U[c] /= ul;
  V[0] = A[1] * U[2] - A[2] * U[1]; V[1] = A[2] * U[0] - A[0] * U[2]; V[2] = A[0] * U[1] - A[1] * U[0];
  double c1 = 0;
  for (int c = 0; c < 3; ++c) {
    g.bo[c] = (float)C[c];
    g.ba[c] = (float)A[c];
    g.bu[c] = (float)U[c];
    g.bv[c] = (float)V[c];
    c1 += fabs(C[c]);
  }
  launch_bundle_fit(h->raysDev, h->nsets, 1, g, dev, h->bstream);
  HIP_TRY(h, hipMemcpyAsync(o, dev, sizeof(o), hipMemcpyDeviceToHost, h->bstream));
  HIP_TRY(h, hipStreamSynchronize(h->bstream));
  for (int k = 0; k < 8; ++k)
    if (!std::isfinite(o[k])) return GVPM_OK;
  if (!(o[4] > 0.15)) return GVPM_OK;  // a field of view near 180 degrees: (u, v) is no parametrisation for it
  // the lines must pass the point to fp32 rounding of positions of this size, and the point must not lie ahead of a start
  const double scale = 1.0 + c1 + fabs(o[6]) + fabs(o[7]);
  const double res = std::sqrt(std::max(o[5], 0.0));
  if (res > 4e-6 * scale || o[6] < -4e-6 * scale) return GVPM_OK;
  g.lineTol = (float)(2.0 * res + 4e-6 * scale);
  // the next uploads are the same sensor with other sub-pixel offsets: two pixels of margin
  const double px = std::max(o[1] - o[0], o[3] - o[2]) / (double)std::max(1, std::min(h->cfg.width, h->cfg.height));
  const double pad = 2.0 * px + 1e-5 * (1.0 + fabs(o[0]) + fabs(o[1]) + fabs(o[2]) + fabs(o[3]));
  g.uMin = (float)(o[0] - pad); g.uMax = (float)(o[1] + pad);
  g.vMin = (float)(o[2] - pad); g.vMax = (float)(o[3] + pad);
  g.mode = 1;
  h->bundleGrid = g;
  h->bundleState = 1;
  return GVPM_OK;
}

// cells of the bundle grid for this build: level-0 cells of 1 / bundleDiv of a tile's width
static Grid bundleCells(const gvpm_context *h, float r, int tileW) {
  Grid g = h->bundleGrid;
  const float range = fmaxf(g.uMax - g.uMin, g.vMax - g.vMin);
  const float tile = range * (float)tileW / (float)std::max(1, std::max(h->cfg.width, h->cfg.height));
  int G = 8;
  while (G < 512 && range / (float)G > tile / h->bundleDiv) G *= 2;
  int levels = 1;
  while ((1 << (levels - 1)) < G) ++levels;  // log2(G) + 1: the last level is one cell
  g.dim[0] = g.dim[1] = G;
  g.dim[2] = 2;
  g.levels = levels;
  g.ncells = (uint32_t)G * (uint32_t)G * 2u;
  g.s0 = range / (float)G;
  g.invS0 = 1.f / g.s0;
  g.radius = r;
  g.org[0] = g.org[1] = g.org[2] = 0.f;
  g.cell = g.s0;
  g.invCell = g.invS0;
  return g;
}

static int buildGrid(gvpm_context *h, float r, bool deferred = false, bool force3D = false) {
  const uint32_t n = h->nph;
  h->boundsPending = false;
  if (n == 0) {
    h->bs->grid = Grid{{0, 0, 0}, 1.f, 1.f, {1, 1, 1}, 1};
    HIP_TRY(h, h->bs->cellStart.ensure(2));
    HIP_TRY(h, hipMemsetAsync(h->bs->cellStart.p, 0, 2 * sizeof(uint32_t), h->bstream));
    return GVPM_OK;
  }
  const int nblocks = 1024;
  HIP_TRY(h, h->bs->boundsPartial.ensure(nblocks * 6));
  HIP_TRY(h, h->bs->bounds6.ensure(32));
  float b6[6];
  if (!h->pinB6) {
    HIP_TRY(h, hipHostMalloc((void **)&h->pinB6, 64, hipHostMallocMapped));
    h->pinCtl = reinterpret_cast<uint32_t *>(h->pinB6 + 8);
  }
  const bool defer = deferred && h->haveCachedBounds;
  launch_bounds(h->rawDev.pos, n, h->bs->boundsPartial.p, nblocks, h->bs->bounds6.p, defer ? h->pinB6 : nullptr,
                h->bstream);
  if (defer) {
    h->boundsPending = true;
    memcpy(b6, h->cachedB6, sizeof(b6));
  } else {
    HIP_TRY(h, hipMemcpyAsync(b6, h->bs->bounds6.p, sizeof(b6), hipMemcpyDeviceToHost, h->bstream));
    HIP_TRY(h, hipStreamSynchronize(h->bstream));
    memcpy(h->cachedB6, b6, sizeof(b6));
    h->haveCachedBounds = true;
  }
  float ext = 0.f;
  for (int c = 0; c < 3; ++c) {
    if (!std::isfinite(b6[c]) || !std::isfinite(b6[3 + c])) return fail(h, GVPM_ERR_INVALID_ARG, "non-finite photon position");
    ext = fmaxf(ext, b6[3 + c] - b6[c]);
  }
  Grid g{};
  // cell edge in radii.  G-BRE with maps up to 2 M photons: 1.5 -- the cell arrays (memset, scan, summed-volume table:
  // ~135 of the build's 575 us at C2 with cells of one radius) shrink 3.4x, and there the build is the stage the
  // pipelined step waits for; the traversal tests 1.3x the photons per hit.  Measured at C2: 1.51 -> 1.43-1.445 ms per
  // step.  At C4 (4 M photons) the evaluation is the long stage and the larger cells cost the traversal 1 % of the step.
  const float cellScale = h->cellScale > 0.f ? h->cellScale : (deferred && n <= 2000000u ? 1.5f : 1.0f);
  float cell = fmaxf(cellScale * r, ext / 384.f);
  if (!(cell > 0.f)) cell = 1.f;
  g.cell = cell;
  g.invCell = 1.f / cell;
  uint64_t nc = 1;
  for (int c = 0; c < 3; ++c) {
    g.org[c] = b6[c] - 0.5f * cell;
    g.dim[c] = (int)floorf((b6[3 + c] - g.org[c]) * g.invCell) + 2;
    nc *= (uint64_t)g.dim[c];
  }
  if (nc > 0x7FFFFFFFull) return fail(h, GVPM_ERR_INVALID_ARG, "grid too large");
  g.ncells = (uint32_t)nc;
  if (deferred && h->bundleEnabled && h->bundleState >= 0 && !force3D) {
    // G-BRE: the camera beams of a pinhole sensor leave one point -- cells over the bundle's (u, v) plane instead
    if (h->bundleState == 0) {
      const int rcf = fitBundle(h);
      if (rcf != GVPM_OK) return rcf;
    }
    if (h->bundleState == 1) g = bundleCells(h, r, h->beamsPerWave == 16 ? 4 : 8);
  }
  h->bs->grid = g;
  // counting sort by cell (x fastest): count + rank, exclusive scan, scatter
  HIP_TRY(h, h->bs->keysA.ensure(n));
  HIP_TRY(h, h->bs->valsA.ensure(n));
  HIP_TRY(h, h->bs->hot.ensure(n));
  HIP_TRY(h, h->bs->cold.ensure((size_t)n * GVPM_REC_QUADS));
  // the radius shrinks every iteration, so the grid grows: size the cell arrays (and the scan's temporary)
  // for the finest grid the cell rule allows (386^3 cells, 230 MB each) once -- a regrowth is a hipFree,
  // i.e. a device-wide sync in the middle of the pipeline
  const size_t worstCells = (size_t)386 * 386 * 386 + 2;
  HIP_TRY(h, h->bs->cellCount.ensure(std::max((size_t)g.ncells + 2, worstCells)));
  HIP_TRY(h, h->bs->cellStart.ensure(std::max((size_t)g.ncells + 2, worstCells)));
  if (!h->bs->scanSized) {
    HIP_TRY(h, reserveScanTemp(h->bs->sortTmp, (uint32_t)worstCells));
    h->bs->scanSized = true;
  }
  HIP_TRY(h, hipMemsetAsync(h->bs->cellCount.p, 0, ((size_t)g.ncells + 1) * sizeof(uint32_t), h->bstream));
  launch_cell_count(h->rawDev.pos, n, g, h->bs->keysA.p, h->bs->valsA.p, h->bs->cellCount.p, h->bstream);
  HIP_TRY(h, exclusiveSumU32(h->bs->sortTmp, h->bs->cellCount.p, h->bs->cellStart.p, g.ncells + 1, h->bstream));
  if (deferred) {
    // G-BRE: summed-volume table for the planner (sized once for the finest grid, like the cell arrays)
    const size_t satCells = (size_t)(g.dim[0] + 1) * (g.dim[1] + 1) * (g.dim[2] + 1);
    HIP_TRY(h, h->bs->sat.ensure(std::max(satCells, (size_t)387 * 387 * 387)));
    launch_sat(h->bs->cellStart.p, g, h->bs->sat.p, h->bstream);
  }
  // longest possible reconnection segment: diagonal of (occluders U photons), generously padded
  float diag2 = 0.f;
  for (int c = 0; c < 3; ++c) {
    const float lo = fminf(b6[c], h->ntri ? h->triMin[c] : b6[c]), hi = fmaxf(b6[3 + c], h->ntri ? h->triMax[c] : b6[3 + c]);
    diag2 += (hi - lo) * (hi - lo);
  }
  const float lmax = 1.25f * sqrtf(diag2) + 8.f * r + 1e-3f;
  const float dmax = h->cfg.shadow_epsilon * lmax * 1.01f + 1e-6f;
  if (h->ntri > 64u && h->useNearGrid && !(dmax <= h->nearGridReach)) {
    const int rcg = buildNearGrid(h, dmax * 1.5f);
    if (rcg != GVPM_OK) return rcg;
  }
  HIP_TRY(h, h->bs->overflowCtr.ensure(2));
  HIP_TRY(h, hipMemsetAsync(h->bs->overflowCtr.p, 0, 4, h->bstream));
  // extension lists of the near-occluder lists: sized once per set for the largest photon count (grow only);
  // word 0 is the allocation cursor
  const size_t extWant = std::min<size_t>(std::max<size_t>((size_t)n * 8u + 4096u, h->nearExtWant), 0xFFFFFF00u);
  HIP_TRY(h, h->bs->nearExt.ensure(extWant));
  HIP_TRY(h, hipMemsetD32Async((hipDeviceptr_t)h->bs->nearExt.p, 1, 1, h->bstream));
  launch_reorder(h->rawDev, h->bs->keysA.p, h->bs->valsA.p, h->bs->cellStart.p, n, h->cfg, h->bvh.p, h->tri4.p, h->ntri, dmax,
                 h->nearGrid, h->bs->nearExt.p, (uint32_t)std::min<size_t>(h->bs->nearExt.cap, 0xFFFFFF00u), h->bs->hot.p,
                 h->bs->cold.p, h->bs->overflowCtr.p, h->bstream);
  HIP_TRY(h, hipGetLastError());
  h->nearOverflow = false;
  if (!deferred && h->cfg.visibility_as_written) {
    uint32_t over = 0;
    HIP_TRY(h, hipMemcpyAsync(&over, h->bs->overflowCtr.p, 4, hipMemcpyDeviceToHost, h->bstream));
    HIP_TRY(h, hipStreamSynchronize(h->bstream));
    h->nearOverflow = over != 0;
  }
  return GVPM_OK;
}

static int sortBeams(gvpm_context *h, int beamsPerWave = 0) {
  const uint32_t n = h->nsets;
  if (!beamsPerWave) beamsPerWave = h->beamsPerWave;
  int tw = 8, th = 4;
  if (beamsPerWave == 64) th = 8;
  if (beamsPerWave == 16) tw = 4;
  const uint32_t tilesX = (h->cfg.width + tw - 1) / tw, tilesY = (h->cfg.height + th - 1) / th;
  h->bs->ntiles = tilesX * tilesY;
  h->bs->tileW = tw;
  h->bs->tileH = th;
  HIP_TRY(h, h->bs->tileStart.ensure((size_t)h->bs->ntiles + 2));
  HIP_TRY(h, h->bs->bKeysA.ensure(n + 1));
  HIP_TRY(h, h->bs->bValsA.ensure(n + 1));
  HIP_TRY(h, h->bs->setPerm.ensure(n + 1));
  if (n) {
    // counting sort by (tile, pixel in tile, edge): count + rank, exclusive scan, scatter
    const int tileShift = ilog2ceil(tw * th) + 3;
    const uint64_t nkeys = (uint64_t)h->bs->ntiles << tileShift;
    if (nkeys > 0x7FFFFFF0ull) return fail(h, GVPM_ERR_INVALID_ARG, "film too large for the beam sort");
    HIP_TRY(h, h->bs->beamCount.ensure(nkeys + 2));
    HIP_TRY(h, h->bs->beamStart.ensure(nkeys + 2));
    HIP_TRY(h, hipMemsetAsync(h->bs->beamCount.p, 0, (nkeys + 1) * sizeof(uint32_t), h->bstream));
    launch_beam_count(h->raysDev, n, h->cfg.width, tw, th, h->bs->bKeysA.p, h->bs->bValsA.p, h->bs->beamCount.p, h->bstream);
    HIP_TRY(h, exclusiveSumU32(h->bs->sortTmp, h->bs->beamCount.p, h->bs->beamStart.p, (uint32_t)nkeys + 1, h->bstream));
    launch_beam_scatter(h->bs->bKeysA.p, h->bs->bValsA.p, h->bs->beamStart.p, n, h->bs->setPerm.p, h->bstream);
    launch_tile_start(h->bs->beamStart.p, h->bs->ntiles, (uint32_t)tileShift, h->bs->tileStart.p, h->bstream);
  } else {
    HIP_TRY(h, hipMemsetAsync(h->bs->tileStart.p, 0, ((size_t)h->bs->ntiles + 1) * sizeof(uint32_t), h->bstream));
  }
  HIP_TRY(h, hipGetLastError());
  return GVPM_OK;
}

// shadow rays through the occluder BVH instead of the per-photon near-occluder lists
static bool needFullVis(const gvpm_context *h) {
  return !h->cfg.visibility_as_written || h->nearOverflow;
}

static void fillArgs(const gvpm_context *h, GatherArgs &a, float r) {
  memset(&a, 0, sizeof(a));
  a.hot = h->bs->hot.p;
  a.cold = h->bs->cold.p;
  a.cellStart = h->bs->cellStart.p;
  a.nearExt = h->bs->nearExt.p;
  a.sat = h->bs->sat.p;
  a.nph = h->nph;
  a.grid = h->bs->grid;
  a.rays = h->raysDev;
  a.setPerm = h->bs->setPerm.p;
  a.tileStart = h->bs->tileStart.p;
  a.nsets = h->nsets;
  a.tri4 = h->tri4.p;
  a.bvh = h->bvh.p;
  a.ntri = h->ntri;
  a.triAbs1 = 0.f;
  if (h->ntri)
    for (int c = 0; c < 3; ++c) a.triAbs1 += fmaxf(fabsf(h->triMin[c]), fabsf(h->triMax[c]));
  for (int c = 0; c < 3; ++c) {
    a.med.sigmaS[c] = h->medium.sigma_s[c];
    a.med.sigmaT[c] = h->medium.sigma_t[c];
  }
  a.med.g = h->medium.g;
  a.med.msw = h->medium.medium_sampling_weight;
  a.cfg = h->cfg;
  a.radius = r;
  a.iter = h->iter.p;
  a.stats = h->stats.p;
  a.samples = h->samplesDev;
  a.nsamples = h->nsamples;
  a.scaleVol = h->scaleVol.p;
  a.mvol = h->mvol.p;
  a.kernelRadius = r;
  a.subLen = 0.f;
  a.nbeams = 0;
}

static int nextEvents(gvpm_context *h, std::pair<hipEvent_t, hipEvent_t> **ev, int phase = 0) {
  // HIP events on the handle's stream bracket the dominant kernel (roofline.achieved) and the other phases
  // a ring of at most GVPM_EVENT_RING pairs per phase: a host that never polls gvpm_get_phase_time keeps the timings of
  // its last launches instead of growing the pool by three pairs per step
  constexpr size_t GVPM_EVENT_RING = 256;
  if (h->eventsHead[phase] == h->events[phase].size()) {
    if (h->events[phase].size() >= GVPM_EVENT_RING) {
      h->eventsHead[phase] = 0;  // overwrite the oldest
    } else {
      hipEvent_t e0, e1;
      HIP_TRY(h, hipEventCreate(&e0));
      HIP_TRY(h, hipEventCreate(&e1));
      h->events[phase].emplace_back(e0, e1);
    }
  }
  *ev = &h->events[phase][h->eventsHead[phase]++];
  h->eventsCount[phase] = std::min(h->eventsCount[phase] + 1, GVPM_EVENT_RING);
  return GVPM_OK;
}

// computeVolumeGradientPhotonBRE, gvpm.cpp:988-1079.
// Pipeline: the build of this step (grid, beam sort, planner) runs on streamB into the build set
// the previous step is NOT reading, so it overlaps the previous step's evaluation kernel; the
// host waits once (planner counters) and then queues traversal + evaluation on the gather stream.
static int gatherBRE(gvpm_context *h, int it, uint64_t nb_paths) {
  const float r = currentRadius(h);
  static const bool traceHost = getenv("GVPM_TRACE_HOST") != nullptr;
  auto T0 = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!traceHost) return;
    auto t = std::chrono::steady_clock::now();
    fprintf(stderr, "[host] it=%d %-10s %8.1f us\n", it, what, std::chrono::duration<double, std::micro>(t - T0).count());
    T0 = t;
  };
  std::pair<hipEvent_t, hipEvent_t> *evBuild, *evTrav, *evEval;
  int rc = nextEvents(h, &evBuild, 2);
  if (rc != GVPM_OK) return rc;
  h->bstream = h->pipeline ? h->streamB : h->stream;
  // the evaluation adds this iteration's estimate (1 / nb_paths per partial sum) straight into the running sum
  h->sumMode = true;
  if (h->sumIt != 0 && it - 1 != h->sumIt) {
    // not the successor of the last iteration: the reference's fold (mean * (it - 1) + v) / it then weighs the old
    // mean by (it - 1) / it, i.e. the sum by (it - 1) / last
    launch_scale(h->accum.p, h->accum.p, h->npix * 27, (float)((double)(it - 1) / (double)h->sumIt), h->stream);
  }
  h->sumIt = it;
  bool rebuilt = false;
  GatherArgs a;
  uint32_t itemCap = 0, blocks = 0, nItems = 0;
  bool force3D = false;
  // (a second pass only when the planner met a ray outside the bundle the grid was keyed for: rebuilt in 3D)
  for (int attempt = 0;; ++attempt) {
  rebuilt = false;
  if (h->photonsDirty || h->beamsDirty || r != h->bs->builtRadius) {
    // the other set; wait until the kernels that last read it are done
    h->setIdx = (h->setIdx + 1) % (h->pipeline && h->travStream ? 3 : 2);
    h->bs = &h->sets[h->setIdx];
    if (h->bs->used) HIP_TRY(h, hipStreamWaitEvent(h->bstream, h->bs->lastUse, 0));
    HIP_TRY(h, hipEventRecord(evBuild->first, h->bstream));
    lap("waitevent");
    rc = buildGrid(h, r, true, force3D);
    lap("buildGrid");
    if (rc == GVPM_OK) rc = sortBeams(h);
    lap("sortBeams");
    if (rc != GVPM_OK) {
      h->bstream = h->stream;
      return rc;
    }
    h->photonsDirty = false;
    h->beamsDirty = false;
    h->bs->builtRadius = r;
    rebuilt = h->nph > 0;
  } else {
    // same inputs, same radius: the set is re-planned and re-traversed in place, once the evaluation kernel that still
    // reads its items and pair lists is done
    if (h->bs->used) HIP_TRY(h, hipStreamWaitEvent(h->bstream, h->bs->lastUse, 0));
    if (h->bs->used && h->pipeline && h->travStream) HIP_TRY(h, hipStreamWaitEvent(h->streamC, h->bs->lastUse, 0));
    HIP_TRY(h, hipEventRecord(evBuild->first, h->bstream));
  }
  fillArgs(h, a, r);
  a.iter = h->accum.p;
  a.iterScale = 1.0f / (float)nb_paths;
  itemCap = plan_items_capacity(h->nsets, h->bs->ntiles, h->beamsPerWave);
  HIP_TRY(h, h->bs->items.ensure(itemCap));
  HIP_TRY(h, h->bs->itemOff.ensure(itemCap));
  if (h->planBoxHandOff) {
    // one box per slab step and tile chunk: steps <= dim / (thinnest slab) + 1, chunks < nsets / B + ntiles + 1 (< 2^24)
    const Grid &g = h->bs->grid;
    const int kmin = std::max(1, std::min(a.cfg.reserved[2] ? a.cfg.reserved[2] : 8, a.cfg.reserved[1] ? a.cfg.reserved[1] : 6));
    // (sized for the finest grid the cell rule allows, as the cell arrays: a regrowth is a device-wide sync)
    const uint32_t stride = (uint32_t)(std::max(386, std::max(g.dim[0], std::max(g.dim[1], g.dim[2]))) / kmin + 2);
    const size_t chunks = (size_t)h->nsets / (size_t)h->beamsPerWave + h->bs->ntiles + 2;
    if (chunks < (1u << 24) && std::max(g.dim[0], std::max(g.dim[1], g.dim[2])) < 1024) {
      HIP_TRY(h, h->bs->planBoxes.ensure(chunks * stride));
      a.planBoxes = h->bs->planBoxes.p;
      a.planBoxStride = stride;
    }
  }
  HIP_TRY(h, h->bs->queueCtl.ensure(8));
  HIP_TRY(h, hipMemsetAsync(h->bs->queueCtl.p, 0, 8 * sizeof(uint32_t), h->bstream));
  a.bundleFlag = h->bs->queueCtl.p + 4;
  launch_plan_bre(a, h->beamsPerWave, h->bs->ntiles, h->planTarget, h->bs->items.p, h->bs->queueCtl.p, h->bs->itemOff.p,
                  h->bs->queueCtl.p + 3, itemCap, h->bstream);
  // the planner's bound on (photon, beam) pairs sizes the pair buffer (grow only) ...
  // ... read back in the step's one host sync, with the photon bounds and the near-list overflow count
  if (!h->pinB6) {
    HIP_TRY(h, hipHostMalloc((void **)&h->pinB6, 64, hipHostMallocMapped));
    h->pinCtl = reinterpret_cast<uint32_t *>(h->pinB6 + 8);
  }
  launch_export_u32(h->bs->queueCtl.p + 3, rebuilt ? h->bs->overflowCtr.p : nullptr, rebuilt ? h->bs->nearExt.p : nullptr,
                    h->bs->queueCtl.p, h->bs->queueCtl.p + 4, h->pinCtl, h->bstream);
  HIP_TRY(h, hipEventRecord(evBuild->second, h->bstream));
  lap("plan");
  HIP_TRY(h, hipStreamSynchronize(h->bstream));
  lap("syncB");
  blocks = h->pinCtl[0];
  nItems = h->pinCtl[3];
  if (a.grid.mode == 1 && h->pinCtl[4] != 0u && attempt == 0) {
    // not the bundle the cells were keyed for (another sensor, or later edges of the camera paths among the beams):
    // this step again on the 3D grid; the frame is fitted anew at the next build, a few times
    h->bundleState = ++h->bundleViolations > 3 ? -1 : 0;
    h->photonsDirty = true;
    h->boundsPending = false;
    force3D = true;
    continue;
  }
  break;
  }
  h->lastGridMode = a.grid.mode;
  h->lastGridCells = a.grid.ncells;
  if (nItems > itemCap) {
    h->bstream = h->stream;
    return fail(h, GVPM_ERR_STATE, "G-BRE planner produced more work items than its bound");
  }
  if (getenv("GVPM_TRACE_PLAN")) {
    uint32_t q[4] = {0, 0, 0, 0};
    (void)hipMemcpy(q, h->bs->queueCtl.p, sizeof(q), hipMemcpyDeviceToHost);
    fprintf(stderr, "[plan] items %u staged blocks %u tiles %u sets %u; photons %u, %s %d x %d x %d cells of %g (radius %g)\n", q[0],
            blocks, h->bs->ntiles, h->nsets, h->nph, h->bs->grid.mode == 1 ? "bundle cells" : "grid", h->bs->grid.dim[0],
            h->bs->grid.dim[1], h->bs->grid.dim[2], (double)h->bs->grid.cell, (double)r);
  }
  if (rebuilt) {
    h->nearOverflow = h->cfg.visibility_as_written && h->pinCtl[1] != 0;
    // what the extension lists asked for (the cursor keeps counting past the capacity): sizes the next build's
    h->nearExtWant = std::max<size_t>(h->nearExtWant, (size_t)h->pinCtl[2] + h->pinCtl[2] / 4);
    if (getenv("GVPM_TRACE_VIS"))
      fprintf(stderr, "[vis] ntri %u photons %u: %u lists overflowed, extension cursor %u of %zu, fullvis %d\n", h->ntri, h->nph,
              h->pinCtl[1], h->pinCtl[2], h->bs->nearExt.cap, (int)needFullVis(h));
  }
  if (h->boundsPending) {
    h->boundsPending = false;
    for (int c = 0; c < 6; ++c)
      if (!std::isfinite(h->pinB6[c])) {
        h->bstream = h->stream;
        return fail(h, GVPM_ERR_INVALID_ARG, "non-finite photon position");
      }
    memcpy(h->cachedB6, h->pinB6, sizeof(h->cachedB6));
  }
  // the traversal also runs on the build stream (this set's own pair buffer): only the evaluation
  // kernels of consecutive steps are serialised on the gather stream
  HIP_TRY(h, h->bs->pairs.ensure((size_t)blocks * 64u + 64u));
  HIP_TRY(h, h->bs->pairCnt.ensure((size_t)itemCap * h->beamsPerWave));
  rc = nextEvents(h, &evTrav, 1);
  if (rc == GVPM_OK) rc = nextEvents(h, &evEval, 0);
  if (rc != GVPM_OK) {
    h->bstream = h->stream;
    return rc;
  }
  // three stages: the traversal has its own stream, so that the build of the NEXT step (which starts on the build
  // stream as soon as this call returns) overlaps it; the build stream is idle here, the host has just synchronised it
  hipStream_t ts = h->pipeline && h->travStream ? h->streamC : (h->travOnBuild ? h->bstream : h->stream);
  HIP_TRY(h, hipEventRecord(evTrav->first, ts));
  launch_traverse_bre(a, h->beamsPerWave, h->bs->items.p, h->bs->itemOff.p, h->bs->queueCtl.p, h->bs->queueCtl.p + 1,
                      h->bs->pairs.p, h->bs->pairCnt.p, h->persistentTrav ? h->nwavesTrav : nItems, h->persistentTrav, ts);
  HIP_TRY(h, hipEventRecord(evTrav->second, ts));
  HIP_TRY(h, hipEventRecord(h->bs->traversed, ts));
  h->bstream = h->stream;
  HIP_TRY(h, hipStreamWaitEvent(h->stream, h->bs->traversed, 0));
  // maps beyond 2 M photons: the evaluation is the stage the pipelined step waits for (its records no longer fit the
  // Infinity Cache), so it gets its third wave per SIMD; below, the other stages need the room more (measured: +6 % on a
  // rank's step at C4 with 12 waves per CU, -3 % at C2)
  const uint32_t nwEval = (h->pipeline && !h->nwavesFromEnv && h->ncu && h->nph > 2000000u)
                              ? std::min<uint32_t>(h->ncu * 12u, GVPM_STAT_ROWS) : h->nwaves;
  HIP_TRY(h, hipEventRecord(evEval->first, h->stream));
  launch_evaluate_bre(a, h->beamsPerWave, needFullVis(h), h->bs->items.p, h->bs->itemOff.p, h->bs->queueCtl.p,
                      h->bs->queueCtl.p + 2, h->bs->pairs.p, h->bs->pairCnt.p, h->persistentEval ? nwEval : nItems, h->persistentEval,
                      h->stream);
  HIP_TRY(h, hipEventRecord(evEval->second, h->stream));
  HIP_TRY(h, hipEventRecord(h->bs->lastUse, h->stream));
  if (h->pipeline) {
    for (int k = 0; k < (h->travStream ? 3 : 2); ++k) {
      BuildSet &other = h->sets[k];
      if (&other != h->bs && !other.used && !h->bs->used) HIP_TRY(h, other.mirrorFrom(*h->bs));
    }
  }
  h->bs->used = true;
  HIP_TRY(h, hipGetLastError());
  lap("launchK");
  // scaleVolumeAPA(it), gvpm.cpp:181-215 (m_independentScale = false, forceAPA empty)
  {
    const double ratio = ((it - 1) + (double)h->cfg.alpha) / ((it - 1) + 1);
    double f = ratio;
    if (h->cfg.vol_technique == GVPM_VOL_BRE3D) f = std::cbrt(ratio);
    else if (h->cfg.vol_technique == GVPM_VOL_BRE2D) f = std::sqrt(ratio);
    h->globalScaleVolume = (float)(h->globalScaleVolume * f);
  }
  return GVPM_OK;
}

// sub-beam grid for photon beams of kernel radius r
static int buildBeamGrid(gvpm_context *h, float r) {
  const uint32_t n = h->nph;
  h->nsub = 0;
  h->maxSubLen = 0.f;
  HIP_TRY(h, h->bs->cold.ensure((size_t)(n + 1) * GVPM_REC_QUADS));
  if (n == 0) {
    h->bs->grid = Grid{{0, 0, 0}, 1.f, 1.f, {1, 1, 1}, 1};
    HIP_TRY(h, h->bs->cellStart.ensure(2));
    HIP_TRY(h, hipMemsetAsync(h->bs->cellStart.p, 0, 2 * sizeof(uint32_t), h->stream));
    h->subLen = r;
    return GVPM_OK;
  }
  // bounds of the beam end points and origins
  const int nblocks = 256;
  HIP_TRY(h, h->bs->boundsPartial.ensure(nblocks * 6));
  HIP_TRY(h, h->bs->bounds6.ensure(32));
  launch_bounds(h->rawDev.pos, n, h->bs->boundsPartial.p, nblocks, h->bs->bounds6.p, nullptr, h->stream);
  launch_bounds(h->rawDev.parent_pos, n, h->bs->boundsPartial.p, nblocks, h->bs->bounds6.p + 6, nullptr, h->stream);
  float b12[12];
  HIP_TRY(h, hipMemcpyAsync(b12, h->bs->bounds6.p, sizeof(b12), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  float b6[6], ext = 0.f;
  for (int c = 0; c < 3; ++c) {
    b6[c] = fminf(b12[c], b12[6 + c]);
    b6[3 + c] = fmaxf(b12[3 + c], b12[9 + c]);
    if (!std::isfinite(b6[c]) || !std::isfinite(b6[3 + c])) return fail(h, GVPM_ERR_INVALID_ARG, "non-finite beam position");
    ext = fmaxf(ext, b6[3 + c] - b6[c]);
  }
  Grid g{};
  // sub-beams (and cells) of 3/4 of the kernel radius: the traversal cost follows the number of sphere tests, which
  // shrinks with the cell until the ext/256 floor (measured: 34 ms at 1.5 r, 23.5 ms at 0.75 r and below)
  // (round 3, measured at C3 with the round's evaluation: sub-beams and cells of 1.5 r -- 24 M sub-beams instead of 47 M,
  // build 6.4 -> 4.2 ms, traversal 10.2 -> 9.9 -- and, with them, work items of 4096 staged sub-beams: traversal -> 7.9 ms.
  // Round 2 had measured the opposite (23.5 ms at 0.75 r against 34 at 1.5 r) on a traversal that resolved its candidates
  // one per lane and round: the cost followed the sphere tests then, the walk and the staging now.)
  float cell = fmaxf(0.75f * (h->cellScale > 0.f ? h->cellScale : 2.0f) * r, ext / 256.f);
  if (!(cell > 0.f)) cell = 1.f;
  g.cell = cell;
  g.invCell = 1.f / cell;
  uint64_t nc = 1;
  for (int c = 0; c < 3; ++c) {
    g.org[c] = b6[c] - 0.5f * cell;
    g.dim[c] = (int)floorf((b6[3 + c] - g.org[c]) * g.invCell) + 2;
    nc *= (uint64_t)g.dim[c];
  }
  if (nc > 0x7FFFFFFFull) return fail(h, GVPM_ERR_INVALID_ARG, "grid too large");
  g.ncells = (uint32_t)nc;
  h->bs->grid = g;
  h->subLen = cell;
  // cut the beams into sub-beams of about one cell
  HIP_TRY(h, h->subCounts.ensure(n + 1));
  HIP_TRY(h, h->subOffsets.ensure(n + 1));
  HIP_TRY(h, h->beamCtl.ensure(4));
  HIP_TRY(h, hipMemsetAsync(h->beamCtl.p, 0, 4 * sizeof(uint32_t), h->stream));
  launch_beam_subcount(h->rawDev.pos, h->rawDev.parent_pos, n, h->subLen, h->subCounts.p, h->beamCtl.p, h->stream);
  HIP_TRY(h, exclusiveSumU32(h->bs->sortTmp, h->subCounts.p, h->subOffsets.p, n, h->stream));
  uint32_t lastOff = 0, lastCnt = 0, maxBits = 0;
  HIP_TRY(h, hipMemcpyAsync(&lastOff, h->subOffsets.p + (n - 1), 4, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(&lastCnt, h->subCounts.p + (n - 1), 4, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(&maxBits, h->beamCtl.p, 4, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  const uint64_t S = (uint64_t)lastOff + lastCnt;
  if (S > 0x7FFFFFF0ull) return fail(h, GVPM_ERR_INVALID_ARG, "too many sub-beams");
  h->nsub = (uint32_t)S;
  memcpy(&h->maxSubLen, &maxBits, 4);
  HIP_TRY(h, h->subCentres.ensure(S * 3 + 4));
  HIP_TRY(h, h->subIds.ensure(S + 1));
  HIP_TRY(h, h->bs->keysA.ensure(S));
  HIP_TRY(h, h->bs->keysB.ensure(S));
  HIP_TRY(h, h->bs->valsA.ensure(S));
  HIP_TRY(h, h->bs->valsB.ensure(S));
  HIP_TRY(h, h->bs->hot.ensure(2 * S));
  HIP_TRY(h, h->subFlags.ensure(S + 1));
  HIP_TRY(h, h->bs->cellStart.ensure((size_t)g.ncells + 2));
  launch_beam_expand(h->rawDev.pos, h->rawDev.parent_pos, n, h->subCounts.p, h->subOffsets.p, h->subCentres.p,
                     h->subIds.p, h->stream);
  launch_cell_keys(h->subCentres.p, h->nsub, g, h->bs->keysA.p, h->bs->valsA.p, h->stream);
  HIP_TRY(h, sortPairsU32(h->bs->sortTmp, h->bs->keysA.p, h->bs->keysB.p, h->bs->valsA.p, h->bs->valsB.p, h->nsub,
                          ilog2ceil(g.ncells + 1), h->stream));
  HIP_TRY(h, h->beamAux.ensure(2 * (size_t)n + 2));
  launch_beam_cold(h->rawDev, h->endNDev, n, h->cfg, h->subCounts.p, h->bs->cold.p, h->beamAux.p, h->stream);
  launch_sub_hot(h->subIds.p, h->bs->valsB.p, h->nsub, h->beamAux.p, h->bs->hot.p, h->subFlags.p, h->stream);
  launch_segment_start(h->bs->keysB.p, h->nsub, g.ncells, 0, h->bs->cellStart.p, h->stream);
  {
    const size_t satCells = (size_t)(g.dim[0] + 1) * (g.dim[1] + 1) * (g.dim[2] + 1);
    HIP_TRY(h, h->bs->sat.ensure(satCells));
    launch_sat(h->bs->cellStart.p, g, h->bs->sat.p, h->stream);
  }
  HIP_TRY(h, hipGetLastError());
  return GVPM_OK;
}

// computeVolumeGradientBeams, gvpm.cpp:880-986
static int gatherBeams(gvpm_context *h, int it, uint64_t nb_paths) {
  if (!h->haveBeamsMap) return fail(h, GVPM_ERR_STATE, "G-Beams gather needs gvpm_upload_beams");
  const float r = currentRadius(h);  // beamInitSize, gvpm.cpp:881
  // phases as for G-BRE (gvpm_get_phase_time): 2 = build (sub-beam grid, beam records, camera-beam sort, near lists),
  // 1 = plan + traversal, 0 = the evaluation (block sort + evaluate_beams2_kernel)
  std::pair<hipEvent_t, hipEvent_t> *evBuild, *evTrav;
  {
    int rcE = nextEvents(h, &evBuild, 2);
    if (rcE != GVPM_OK) return rcE;
    rcE = nextEvents(h, &evTrav, 1);
    if (rcE != GVPM_OK) return rcE;
  }
  HIP_TRY(h, hipEventRecord(evBuild->first, h->stream));
  if (h->photonsDirty || r != h->bs->builtRadius) {
    int rc = buildBeamGrid(h, r);
    if (rc != GVPM_OK) return rc;
    h->photonsDirty = false;
    h->bs->builtRadius = r;
    h->beamNearStale = true;  // new records (or a new radius): the near-occluder lists are rebuilt below
  }
  bool nearDirty = false;
  if (h->beamsDirty) {
    int rc = sortBeams(h);
    if (rc != GVPM_OK) return rc;
    h->beamsDirty = false;
    // how far a shifted camera ray strays from its base ray, over all uploaded sets (beam_near_kernel's delta)
    HIP_TRY(h, h->shiftExtent.ensure(1));
    launch_shift_extent(h->raysDev, h->nsets, h->shiftExtent.p, h->stream);
    nearDirty = true;
  }
  if (nearDirty || h->beamNearStale) {
    HIP_TRY(h, h->shiftExtent.ensure(1));
    HIP_TRY(h, h->beamClear.ensure((size_t)h->nph + 1));
    launch_beam_near(h->bs->cold.p, h->nph, h->tri4.p, h->ntri, r, h->shiftExtent.p, h->beamClear.p, h->beamsFreeCone, h->stream);
    if (getenv("GVPM_BEAMS_TRACE")) {
      DevBuf<uint32_t> hist;
      uint32_t hh[21] = {0}, ext = 0;
      if (hist.ensure(24) == hipSuccess && hipMemsetAsync(hist.p, 0, 96, h->stream) == hipSuccess) {
        launch_beam_near_hist(h->bs->cold.p, h->nph, h->ntri, hist.p, h->stream);
        (void)hipMemcpyAsync(hh, hist.p, sizeof(hh), hipMemcpyDeviceToHost, h->stream);
        (void)hipMemcpyAsync(&ext, h->shiftExtent.p, 4, hipMemcpyDeviceToHost, h->stream);
        (void)hipStreamSynchronize(h->stream);
        float extf;
        memcpy(&extf, &ext, 4);
        fprintf(stderr, "[beams] near lists (delta = 4 x %g + %g, %u occluders): lengths 0..19:", (double)r, (double)extf, h->ntri);
        for (int k = 0; k <= 19; ++k) fprintf(stderr, " %u", hh[k]);
        fprintf(stderr, "; overflowed %u\n", hh[20]);
      }
      hist.release();
    }
    h->beamNearStale = false;
  }
  HIP_TRY(h, hipMemsetAsync(h->iter.p, 0, h->npix * 27 * sizeof(float), h->stream));
  HIP_TRY(h, hipEventRecord(evBuild->second, h->stream));
  GatherArgs a;
  fillArgs(h, a, r);
  a.kernelRadius = r;
  // one cell layer per slab step (the box of a thicker slab grows with the tile's perspective spread, and the
  // traversal cost follows the number of sphere tests: 5.1 ms at 1 layer, 6.2 at 2, 8.9 at 4/6)
  if (!a.cfg.reserved[1]) a.cfg.reserved[1] = 1;
  if (!a.cfg.reserved[2]) a.cfg.reserved[2] = 1;
  a.radius = r + 0.5f * h->maxSubLen * 1.001f + 1e-6f;  // traversal radius: sub-beams are binned by their centre
  a.subLen = h->subLen;
  a.nbeams = h->nph;
  a.nph = h->nsub;
  a.beamClear = h->beamClear.p;
  std::pair<hipEvent_t, hipEvent_t> *ev;
  int rc = nextEvents(h, &ev);
  if (rc != GVPM_OK) return rc;
  // The planner splits heavy items into as many as 512 parts (tile_walk.h), so its own bound (items per tile chunk) is
  // not a bound on the list: the list starts at that bound plus room for the parts and, when the planner reports more
  // (it counts what it could not write, and the traversal never reads past the capacity), is regrown to the count and
  // the plan repeated -- as the pair list below.
  uint32_t itemCap = h->beamItemsInit ? h->beamItemsInit
                                      : plan_items_capacity(h->nsets, h->bs->ntiles, h->beamsPerWave) + h->nsub / 256u + 4096u;
  itemCap = std::max(itemCap, h->beamItemCap);
  HIP_TRY(h, h->bs->queueCtl.ensure(8));
  // traversal -> pair list (blocks of 64) -> evaluation.  The list has no useful a-priori bound (the planner's is
  // sub-beams x rays per slab box, ~100x the survivors): it starts at 16 M pairs and, when the traversal reports
  // more than fit, is regrown to what it counted and the traversal repeated (deterministic, first iterations only).
  // queueCtl: [0] items, [1] item queue head, [2] pairs (multiple of 64), [3] block queue head
  if (h->beamPairs.cap == 0) HIP_TRY(h, h->beamPairs.ensure(h->beamPairsInit));
  uint32_t npairs = 0;
  bool planned = false;
  HIP_TRY(h, hipEventRecord(evTrav->first, h->stream));
  for (int attempt = 0;; ++attempt) {
    if (!planned) {
      HIP_TRY(h, h->bs->items.ensure(itemCap));
      HIP_TRY(h, hipMemsetAsync(h->bs->queueCtl.p, 0, 8 * sizeof(uint32_t), h->stream));
      launch_plan_bre(a, h->beamsPerWave, h->bs->ntiles, h->planTargetSet ? h->planTarget : 4096u, h->bs->items.p, h->bs->queueCtl.p, nullptr, nullptr,
                      itemCap, h->stream);
      planned = true;
    }
    const uint32_t cap = (uint32_t)std::min<size_t>(h->beamPairs.cap, 0xFFFFFFC0u);
    const size_t nblkCap = cap / 64u + 1u;
    for (DevBuf<uint32_t> *b : {&h->blockKeyA, &h->blockKeyB, &h->blockValA, &h->blockValB}) HIP_TRY(h, b->ensure(nblkCap));
    launch_traverse_beams(a, h->subFlags.p, h->beamsPerWave, h->bs->items.p, h->bs->queueCtl.p, itemCap, h->bs->queueCtl.p + 1,
                          h->beamPairs.p, h->bs->queueCtl.p + 2, cap, h->blockKeyA.p, h->blockValA.p, h->nwavesTrav, h->stream);
    uint32_t ctl[3] = {0, 0, 0};
    HIP_TRY(h, hipMemcpyAsync(ctl, h->bs->queueCtl.p, sizeof(ctl), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    npairs = ctl[2];
    if (getenv("GVPM_BEAMS_TRACE")) {
      uint32_t q[4];
      (void)hipMemcpy(q, h->bs->queueCtl.p, sizeof(q), hipMemcpyDeviceToHost);
      fprintf(stderr, "[beams] items %u (cap %u) pairs %u (cap %u) nsub %u nsets %u tiles %u cell %.3f r %.3f dims %d %d %d\n", q[0],
              itemCap, npairs, cap, h->nsub, h->nsets, h->bs->ntiles, h->bs->grid.cell, r, h->bs->grid.dim[0], h->bs->grid.dim[1],
              h->bs->grid.dim[2]);
    }
    const bool itemsOver = ctl[0] > itemCap, pairsOver = npairs > cap;
    if (!itemsOver && !pairsOver) break;
    if (attempt >= 3) return fail(h, GVPM_ERR_HIP, "G-Beams item / pair lists overflowed after being regrown");
    if (itemsOver) {
      // (the pass over the truncated list is discarded whole)
      itemCap = ctl[0] + (ctl[0] >> 2) + 64u;
      h->beamItemCap = itemCap;
      planned = false;
    } else {
      HIP_TRY(h, h->beamPairs.ensure((size_t)npairs + (npairs >> 2) + 64));
      HIP_TRY(h, hipMemsetAsync(h->bs->queueCtl.p + 1, 0, 2 * sizeof(uint32_t), h->stream));
    }
    // the candidate count of the discarded pass
    HIP_TRY(h, hipMemset2DAsync(a.stats + 1, 8 * sizeof(unsigned long long), 0, sizeof(unsigned long long), GVPM_STAT_ROWS,
                                h->stream));
  }
  HIP_TRY(h, hipEventRecord(evTrav->second, h->stream));
  HIP_TRY(h, hipEventRecord(ev->first, h->stream));
  // blocks of 64 pairs, sorted by tile: the evaluation loads a tile's rays once per run of its blocks
  const uint32_t nBlocks = npairs / 64u;
  if (nBlocks)
    HIP_TRY(h, sortPairsU32(h->bs->sortTmp, h->blockKeyA.p, h->blockKeyB.p, h->blockValA.p, h->blockValB.p, nBlocks,
                            ilog2ceil(h->nsets + 1), h->stream));
  launch_evaluate_beams(a, h->beamsPerWave, h->beamsExact, h->beamPairs.p, h->blockKeyB.p, h->blockValB.p, nBlocks,
                        h->bs->queueCtl.p + 3, h->nwaves, h->stream);
  HIP_TRY(h, hipEventRecord(ev->second, h->stream));
  launch_finalize(h->accum.p, h->iter.p, h->npix * 27, it, nb_paths, h->stream);
  HIP_TRY(h, hipGetLastError());
  {
    // scaleVolumeAPA(it): cube root for the 3D kernels, linear for the 1D kernel (gvpm.cpp:195-201)
    const double ratio = ((it - 1) + (double)h->cfg.alpha) / ((it - 1) + 1);
    const double f = h->cfg.vol_technique == GVPM_BEAM_BEAM_1D ? ratio : std::cbrt(ratio);
    h->globalScaleVolume = (float)(h->globalScaleVolume * f);
  }
  return GVPM_OK;
}

// computeVolumeGradientPlanes, gvpm.cpp:782-878
static int gatherPlanes(gvpm_context *h, int it, uint64_t nb_paths) {
  if (!h->havePlanes) return fail(h, GVPM_ERR_STATE, "G-Planes gather needs gvpm_upload_planes");
  PlaneArgs pa;
  pa.ori = h->rawDev.parent_pos;
  pa.end = h->rawDev.pos;
  pa.flux = h->rawDev.flux;
  pa.flags = h->rawDev.flags;
  pa.w1 = h->w1Dev;
  pa.len1 = h->len1Dev;
  pa.nplanes = h->nph;
  pa.planesPerItem = h->nph;
  if (h->photonsDirty) {
    HIP_TRY(h, h->planeTest.ensure((size_t)h->nph * 3 + 1));
    pa.test = h->planeTest.p;
    launch_plane_records(pa, h->planeTest.p, h->stream);
    h->photonsDirty = false;
    h->bs->builtRadius = -1.f;
  }
  pa.test = h->planeTest.p;
  if (h->beamsDirty) {
    int rc = sortBeams(h, 64);  // one camera ray per lane: 8x8 pixel tiles
    if (rc != GVPM_OK) return rc;
    h->beamsDirty = false;
  }
  HIP_TRY(h, hipMemsetAsync(h->iter.p, 0, h->npix * 27 * sizeof(float), h->stream));
  GatherArgs a;
  fillArgs(h, a, 0.f);
  // enough (tile, plane chunk) items to fill the chip; chunks of at least 256 planes
  uint32_t nchunks = 1;
  if (h->bs->ntiles && h->nph) {
    nchunks = (4u * h->nwaves + h->bs->ntiles - 1) / h->bs->ntiles;
    nchunks = std::max(1u, std::min(nchunks, (h->nph + 255u) / 256u));
    nchunks = std::min(nchunks, 65535u);
    pa.planesPerItem = (h->nph + nchunks - 1) / nchunks;
    nchunks = (h->nph + pa.planesPerItem - 1) / pa.planesPerItem;
  }
  std::pair<hipEvent_t, hipEvent_t> *ev;
  int rc = nextEvents(h, &ev);
  if (rc != GVPM_OK) return rc;
  HIP_TRY(h, hipEventRecord(ev->first, h->stream));
  launch_gather_planes(a, pa, h->bs->ntiles, nchunks, h->stream);
  HIP_TRY(h, hipEventRecord(ev->second, h->stream));
  launch_finalize(h->accum.p, h->iter.p, h->npix * 27, it, nb_paths, h->stream);
  HIP_TRY(h, hipGetLastError());
  {
    // scaleVolumeAPA(it): the plane estimator takes the linear ratio (gvpm.cpp:195-201)
    const double ratio = ((it - 1) + (double)h->cfg.alpha) / ((it - 1) + 1);
    h->globalScaleVolume = (float)(h->globalScaleVolume * ratio);
  }
  return GVPM_OK;
}

// computeVolumeGradientPhoton (G-VPM), gvpm.cpp:1081-1203
static int gatherVPM(gvpm_context *h, int it, uint64_t nb_paths) {
  (void)it;
  if (!h->haveSamples) return fail(h, GVPM_ERR_STATE, "G-VPM gather needs gvpm_upload_vpm_samples");
  if (h->cfg.nb_camera_samples <= 0) return fail(h, GVPM_ERR_INVALID_ARG, "nb_camera_samples must be positive");
  // grid cell = the largest per-pixel radius R * 0.01 * max(scaleVol)
  uint32_t bits = 0;
  HIP_TRY(h, hipMemcpyAsync(&bits, h->maxScaleBits.p, 4, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  float maxScale;
  memcpy(&maxScale, &bits, 4);
  const float rmax = (h->cfg.bsphere_radius * 0.01f) * maxScale;
  if (h->photonsDirty || rmax != h->bs->builtRadius) {
    int rc = buildGrid(h, rmax);
    if (rc != GVPM_OK) return rc;
    h->photonsDirty = false;
    h->bs->builtRadius = rmax;
  }
  HIP_TRY(h, hipMemsetAsync(h->iter.p, 0, h->npix * 27 * sizeof(float), h->stream));
  HIP_TRY(h, hipMemsetAsync(h->mvol.p, 0, h->npix * sizeof(float), h->stream));
  HIP_TRY(h, hipMemsetAsync(h->maxScaleBits.p, 0, 4, h->stream));
  GatherArgs a;
  fillArgs(h, a, rmax);
  std::pair<hipEvent_t, hipEvent_t> *ev;
  int rc = nextEvents(h, &ev);
  if (rc != GVPM_OK) return rc;
  HIP_TRY(h, hipEventRecord(ev->first, h->stream));
  launch_gather_vpm(a, needFullVis(h), h->stream);
  HIP_TRY(h, hipEventRecord(ev->second, h->stream));
  launch_accumulate(h->accum.p, h->iter.p, h->npix * 27, h->stream);
  launch_vpm_update(h->scaleVol.p, h->nVol.p, h->mvol.p, h->npix, h->cfg.alpha, h->maxScaleBits.p, h->stream);
  HIP_TRY(h, hipGetLastError());
  h->totalEmitted += (double)nb_paths;  // m_totalEmittedVolume, gvpm.cpp:434
  return GVPM_OK;
}

int gvpm_gather(gvpm_context *h, int it, uint64_t nb_paths) {
  CHECK_H(h);
  if (it < 1 || nb_paths == 0) return fail(h, GVPM_ERR_INVALID_ARG, "it must be >= 1 and nb_paths > 0");
  if (!h->haveMedium || !h->havePhotons || !h->haveBeams)
    return fail(h, GVPM_ERR_STATE, "gather needs medium, photons and camera beams uploaded");
  h->useAll = false;
  h->bstream = h->stream;
  // the streams that read this step's host-uploaded inputs wait for their copies (copy stream)
  if (h->phWait) {
    HIP_TRY(h, hipStreamWaitEvent(h->stream, h->phSlot[h->phCur].copied, 0));
    HIP_TRY(h, hipStreamWaitEvent(h->streamB, h->phSlot[h->phCur].copied, 0));
    h->phWait = false;
  }
  if (h->rayWait) {
    HIP_TRY(h, hipStreamWaitEvent(h->stream, h->raySlot[h->rayCur].copied, 0));
    HIP_TRY(h, hipStreamWaitEvent(h->streamB, h->raySlot[h->rayCur].copied, 0));
    h->rayWait = false;
  }
  int rc;
  switch (h->cfg.vol_technique) {
    case GVPM_VOL_BRE2D:
    case GVPM_VOL_BRE3D: rc = gatherBRE(h, it, nb_paths); break;
    case GVPM_DISTANCE: rc = gatherVPM(h, it, nb_paths); break;
    case GVPM_BEAM_BEAM_1D:
    case GVPM_BEAM_BEAM_3D_OPTIMIZED: rc = gatherBeams(h, it, nb_paths); break;
    case GVPM_VOL_PLANE0D: rc = gatherPlanes(h, it, nb_paths); break;
    default: return fail(h, GVPM_ERR_UNSUPPORTED, "vol_technique not built in this library yet");
  }
  if (rc != GVPM_OK) return rc;
  // the kernels just queued on the gather stream are the last readers of this step's camera rays
  if (h->raysOwnedCur) {
    HIP_TRY(h, hipEventRecord(h->raySlot[h->rayCur].freed, h->stream));
    h->raySlot[h->rayCur].read = true;
  }
  // ... and of the staged photon arrays (the build on either stream; G-Planes reads them in the gather itself)
  if (h->photonsOwnedCur) {
    // G-BRE reads them in its build only (reorder_kernel), and gatherBRE has waited for that build on the host (the
    // planner's counters): nothing of it is in flight here.  An event behind the whole gather made the prefetched copy
    // of step N+2 wait for the EVALUATION of step N: the PCIe-inclusive step went from 3.6 to 5.8 ms.  The other
    // techniques read them on the gather stream (their builds; G-Planes in the gather kernel itself).
    gvpm_context::PhotonSlot &ps = h->phSlot[h->phCur];
    const bool bre = h->cfg.vol_technique == GVPM_VOL_BRE2D || h->cfg.vol_technique == GVPM_VOL_BRE3D;
    if (!bre) {
      HIP_TRY(h, hipEventRecord(ps.consumed, h->stream));
      HIP_TRY(h, hipEventRecord(ps.consumedB, h->streamB));
      ps.read = true;
    }
  }
  // prefetched inputs (gvpm_prefetch_*) become the current ones: what an upload at this point would have done
  if (h->phPending >= 0) {
    h->phCur = h->phPending;
    h->phPending = -1;
    h->rawDev = h->phSlot[h->phCur].dev;
    h->nph = (uint32_t)h->rawDev.n;
    h->phWait = true;
    h->photonsOwnedCur = true;
    h->photonsDirty = true;
  }
  if (h->rayPending >= 0) {
    h->rayCur = h->rayPending;
    h->rayPending = -1;
    h->raysDev = h->raySlot[h->rayCur].rays.p;
    h->nsets = h->raySlot[h->rayCur].nsets;
    h->rayWait = true;
    h->raysOwnedCur = true;
    h->beamsDirty = true;
  }
  return GVPM_OK;
}

int gvpm_download_vpm_state(gvpm_context *h, float *scale_vol, float *n_vol) {
  CHECK_H(h);
  if (!scale_vol || !n_vol) return GVPM_ERR_INVALID_ARG;
  HIP_TRY(h, hipMemcpyAsync(scale_vol, h->scaleVol.p, h->npix * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(n_vol, h->nVol.p, h->npix * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return GVPM_OK;
}

int gvpm_get_radius(gvpm_context *h, float *radius) {
  if (!h || !radius) return GVPM_ERR_INVALID_ARG;
  *radius = currentRadius(h);
  return GVPM_OK;
}

int gvpm_set_global_scale(gvpm_context *h, float s) {
  if (!h || !(s > 0.f)) return GVPM_ERR_INVALID_ARG;
  h->globalScaleVolume = s;
  return GVPM_OK;
}

int gvpm_get_stats(gvpm_context *h, gvpm_stats *out) {
  CHECK_H(h);
  if (!out) return GVPM_ERR_INVALID_ARG;
  std::vector<unsigned long long> rows(8 * (size_t)GVPM_STAT_ROWS);
  HIP_TRY(h, hipMemcpyAsync(rows.data(), h->stats.p, rows.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                            h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  unsigned long long v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (size_t r = 0; r < GVPM_STAT_ROWS; ++r)
    for (int k = 0; k < 8; ++k) v[k] += rows[8 * r + k];
  memset(out, 0, sizeof(*out));
  out->evaluations = v[0];
  out->candidates = v[1];
  out->null_shifts = v[2];
  out->diffuse_shifts = v[3];
  out->failed_shifts = v[4];
  out->dropped_pairs = v[7];
  out->reserved[0] = ((uint64_t)h->lastGridMode << 56) | (uint64_t)h->lastGridCells;
  // the planner's bound on an item's pair region is exact: a dropped pair means a biased image, not a slow one
  if (v[7]) return fail(h, GVPM_ERR_STATE, "the G-BRE traversal dropped pairs: planner bound violated");
  return GVPM_OK;
}

int gvpm_get_phase_time(gvpm_context *h, int phase, float *avg_ms, uint32_t *launches) {
  CHECK_H(h);
  if (phase < 0 || phase >= GVPM_PHASES) return fail(h, GVPM_ERR_INVALID_ARG, "unknown phase");
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  // the last `count` launches: the slots behind the head, wrapping (once the ring is full every slot holds one)
  double total = 0;
  const size_t count = h->eventsCount[phase], size = h->events[phase].size();
  for (size_t k = 0; k < count; ++k) {
    const size_t i = (h->eventsHead[phase] + size - 1 - k) % size;
    float ms = 0;
    HIP_TRY(h, hipEventElapsedTime(&ms, h->events[phase][i].first, h->events[phase][i].second));
    total += ms;
  }
  if (avg_ms) *avg_ms = count ? (float)(total / count) : 0.f;
  if (launches) *launches = (uint32_t)count;
  h->eventsHead[phase] = 0;
  h->eventsCount[phase] = 0;
  return GVPM_OK;
}

int gvpm_get_kernel_time(gvpm_context *h, float *avg_ms, uint32_t *launches) {
  return gvpm_get_phase_time(h, 0, avg_ms, launches);
}

// G-BRE: accum holds the sum over iterations, the mean is sum / it
static float accumScale(const gvpm_context *h) { return h->sumMode && h->sumIt > 0 ? (float)(1.0 / (double)h->sumIt) : 1.f; }

int gvpm_download_accum(gvpm_context *h, float *accum) {
  CHECK_H(h);
  if (!accum) return GVPM_ERR_INVALID_ARG;
  const float *src = h->useAll ? h->accumAll.p : h->accum.p;
  if (accumScale(h) != 1.f) {
    HIP_TRY(h, h->accumTmp.ensure(h->npix * 27));
    launch_scale(src, h->accumTmp.p, h->npix * 27, accumScale(h), h->stream);
    src = h->accumTmp.p;
  }
  HIP_TRY(h, hipMemcpyAsync(accum, src, h->npix * 27 * sizeof(float),
                            hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return GVPM_OK;
}

int gvpm_download_accum_dev(gvpm_context *h, float *accum_dev) {
  CHECK_H(h);
  if (!accum_dev) return GVPM_ERR_INVALID_ARG;
  launch_scale(h->useAll ? h->accumAll.p : h->accum.p, accum_dev, h->npix * 27, accumScale(h), h->stream);
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return GVPM_OK;
}

// throughput | dx | dy planes into filmOut (device); emission already on the device or null
static int filmToDevice(gvpm_context *h, int it, int reuse_primal, const float *emissionDev, float *out) {
  const size_t n = h->npix * 3;
  // non-APA estimators are normalised by the emitted path count (gvpm.cpp:489-492)
  float invDiv = 1.f;
  if (h->cfg.vol_technique == GVPM_DISTANCE) invDiv = h->totalEmitted > 0 ? (float)(1.0 / h->totalEmitted) : 0.f;
  invDiv *= accumScale(h);
  launch_film(h->useAll ? h->accumAll.p : h->accum.p, emissionDev, h->cfg.width, h->cfg.height, it, reuse_primal, invDiv,
              out, out + n, out + 2 * n, h->stream);
  HIP_TRY(h, hipGetLastError());
  return GVPM_OK;
}

int gvpm_download_film(gvpm_context *h, int it, int reuse_primal, const float *emission, float *throughput, float *dx,
                       float *dy) {
  CHECK_H(h);
  if (!throughput || !dx || !dy || it < 1) return GVPM_ERR_INVALID_ARG;
  const size_t n = h->npix * 3;
  HIP_TRY(h, h->filmOut.ensure(3 * n));
  const float *em = nullptr;
  if (emission) {
    HIP_TRY(h, h->emission.ensure(n));
    HIP_TRY(h, hipMemcpyAsync(h->emission.p, emission, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
    em = h->emission.p;
  }
  if (int rc = filmToDevice(h, it, reuse_primal, em, h->filmOut.p)) return rc;
  HIP_TRY(h, hipMemcpyAsync(throughput, h->filmOut.p, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(dx, h->filmOut.p + n, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(dy, h->filmOut.p + 2 * n, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return GVPM_OK;
}

int gvpm_download_film_dev(gvpm_context *h, int it, int reuse_primal, const float *emission_dev, float *film_dev) {
  CHECK_H(h);
  if (!film_dev || it < 1) return GVPM_ERR_INVALID_ARG;
  return filmToDevice(h, it, reuse_primal, emission_dev, film_dev);
}

int gvpm_synchronize(gvpm_context *h) {
  CHECK_H(h);
  HIP_TRY(h, hipStreamSynchronize(h->copyStream));  // uploads from pinned memory are left in flight
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return GVPM_OK;
}

// ---- multi-GPU: RCCL all-reduce of the accumulators over xGMI -----------------------------
int gvpm_poisson_preset(const char *preset, gvpm_poisson_params *p) {
  if (!preset || !p) return GVPM_ERR_INVALID_ARG;
  // base config, Solver.cpp:95-101
  p->alpha = 0.2f;
  p->irls_iter_max = 1; p->irls_reg_init = 0.f; p->irls_reg_iter = 0.f;
  p->cg_iter_max = 1; p->cg_iter_check = 100; p->cg_precond = 0; p->cg_tolerance = 0.f;
  if (!strcmp(preset, "L1D")) { p->irls_iter_max = 20; p->irls_reg_init = 0.05f; p->irls_reg_iter = 0.5f; p->cg_iter_max = 50; return GVPM_OK; }
  if (!strcmp(preset, "L1Q")) { p->irls_iter_max = 64; p->irls_reg_init = 1.0f; p->irls_reg_iter = 0.7f; p->cg_iter_max = 1000; return GVPM_OK; }
  if (!strcmp(preset, "L1L")) { p->irls_iter_max = 7; p->irls_reg_init = 1.0e-4f; p->irls_reg_iter = 1.0e-1f; p->cg_iter_max = 20000; p->cg_tolerance = 1.0e-20f; return GVPM_OK; }
  if (!strcmp(preset, "L2D")) { p->cg_iter_max = 50; return GVPM_OK; }
  if (!strcmp(preset, "L2Q")) { p->cg_iter_max = 500; return GVPM_OK; }
  return GVPM_ERR_INVALID_ARG;
}

static int poissonCommon(gvpm_context *h, const gvpm_poisson_params *prm, int W, int H, const float *dx, const float *dy,
                         const float *tp, const float *direct, float *out, bool fromDevice) {
  if (!prm || !dx || !dy || !out) return fail(h, GVPM_ERR_INVALID_ARG, "null argument");
  if (W <= 0 || H <= 0 || (uint64_t)W * H > 0x7FFFFFFull) return fail(h, GVPM_ERR_INVALID_ARG, "bad image size");
  if (prm->cg_precond) return fail(h, GVPM_ERR_UNSUPPORTED, "cgPrecond is not supported (no preset of the reference enables it)");
  const size_t n3 = (size_t)W * H * 3;
  HIP_TRY(h, h->poissonScratch.ensure(poisson_scratch_bytes(W, H) / sizeof(float) + 16));
  const float *ddx = dx, *ddy = dy, *dtp = tp, *ddir = direct;
  float *dout = out;
  if (!fromDevice) {
    HIP_TRY(h, h->poissonIO.ensure(5 * n3));
    float *io = h->poissonIO.p;
    HIP_TRY(h, hipMemcpyAsync(io, dx, n3 * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync(io + n3, dy, n3 * 4, hipMemcpyHostToDevice, h->stream));
    if (tp) HIP_TRY(h, hipMemcpyAsync(io + 2 * n3, tp, n3 * 4, hipMemcpyHostToDevice, h->stream));
    if (direct) HIP_TRY(h, hipMemcpyAsync(io + 3 * n3, direct, n3 * 4, hipMemcpyHostToDevice, h->stream));
    ddx = io; ddy = io + n3; dtp = tp ? io + 2 * n3 : nullptr; ddir = direct ? io + 3 * n3 : nullptr; dout = io + 4 * n3;
  }
  HIP_TRY(h, poisson_solve_device(*prm, W, H, ddx, ddy, dtp, ddir, dout, h->poissonScratch.p, h->poissonGraph, h->stream));
  if (!fromDevice) {
    HIP_TRY(h, hipMemcpyAsync(out, dout, n3 * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
  }
  return GVPM_OK;
}

int gvpm_poisson_solve(gvpm_context *h, const gvpm_poisson_params *params, int width, int height, const float *dx,
                       const float *dy, const float *throughput, const float *direct, float *out) {
  CHECK_H(h);
  return poissonCommon(h, params, width, height, dx, dy, throughput, direct, out, false);
}
int gvpm_poisson_solve_dev(gvpm_context *h, const gvpm_poisson_params *params, int width, int height, const float *dx,
                           const float *dy, const float *throughput, const float *direct, float *out) {
  CHECK_H(h);
  return poissonCommon(h, params, width, height, dx, dy, throughput, direct, out, true);
}

int gvpm_comm_unique_id(void *id128) {
  if (!id128) return GVPM_ERR_INVALID_ARG;
  if (!g_rccl.load()) return GVPM_ERR_COMM;
  ncclUniqueId id;
  if (g_rccl.getUniqueId(&id) != ncclSuccess) return GVPM_ERR_COMM;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  memcpy(id128, &id, 128);
  return GVPM_OK;
}

int gvpm_comm_init(gvpm_context *h, const void *id128, int rank, int world) {
  CHECK_H(h);
  if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(h, GVPM_ERR_INVALID_ARG, "bad comm arguments");
  if (!g_rccl.load()) return fail(h, GVPM_ERR_COMM, "librccl.so not loadable");
  ncclUniqueId id;
  memcpy(&id, id128, 128);
  if (g_rccl.commInitRank(&h->comm, world, id, rank) != ncclSuccess) return fail(h, GVPM_ERR_COMM, "ncclCommInitRank failed");
  return GVPM_OK;
}

int gvpm_allreduce_accum(gvpm_context *h) {
  CHECK_H(h);
  if (!h->comm) return fail(h, GVPM_ERR_STATE, "gvpm_comm_init not called");
  // out of place: the per-rank running means keep their disjoint supports for later iterations
  HIP_TRY(h, h->accumAll.ensure(h->npix * 27));
  if (g_rccl.allReduce(h->accum.p, h->accumAll.p, h->npix * 27, ncclFloat, ncclSum, h->comm, h->stream) != ncclSuccess)
    return fail(h, GVPM_ERR_COMM, "ncclAllReduce failed");
  h->useAll = true;
  return GVPM_OK;
}

int gvpm_allreduce_film(gvpm_context *h, float *film_dev) {
  CHECK_H(h);
  if (!film_dev) return GVPM_ERR_INVALID_ARG;
  if (!h->comm) return fail(h, GVPM_ERR_STATE, "gvpm_comm_init not called");
  if (g_rccl.allReduce(film_dev, film_dev, h->npix * 9, ncclFloat, ncclSum, h->comm, h->stream) != ncclSuccess)
    return fail(h, GVPM_ERR_COMM, "ncclAllReduce failed");
  return GVPM_OK;
}

}  // extern "C"
