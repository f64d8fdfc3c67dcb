// Internal to libgvpm_hip.so: the handle behind the C ABI (include/gvpm_hip.h), its device buffers and the launch functions of
// the kernel files.  Shared by gvpm_api.hip (lifecycle, results, comm), uploads.hip (staging of the per-iteration inputs)
// and gather_drivers.hip (the per-technique drivers behind gvpm_gather).
#pragma once
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
                   hipStream_t s, const uint32_t *word = nullptr, uint32_t *wordOut = nullptr);
hipError_t reserveScanTemp(SortTemp &tmp, uint32_t n);
void launch_bundle_fit(const gvpm_camera_ray *rays, uint32_t nsets, int pass, const Grid &g, double *out, hipStream_t s);
void launch_export_u32(const uint32_t *a, const uint32_t *b, const uint32_t *c, const uint32_t *d, const uint32_t *e, uint32_t *hostOut,
                       hipStream_t s);
void launch_sat(const uint32_t *cellStart, const Grid &g, uint32_t *sat, hipStream_t s);
uint32_t cell_stripes();
void launch_cell_count(const float *pos, uint32_t n, const Grid &g, uint32_t *keys, uint32_t *rank, uint32_t *count, uint32_t *sub,
                       hipStream_t s, uint32_t *zeroWord = nullptr, uint32_t *oneWord = nullptr);
void launch_reorder(const gvpm_photon_soa &raw, const uint32_t *keys, const uint32_t *rank, const uint32_t *cellStart,
                    uint32_t n, const gvpm_params &cfg, const float4 *bvh, const float4 *tri4, uint32_t ntri, float dmax,
                    const NearGrid &ng, uint32_t *nearExt, uint32_t extCap, float4 *hot, float4 *cold, uint32_t *overflow,
                    uint32_t *origIdx, const uint32_t *sub, uint32_t ncells, hipStream_t s);
void launch_apply_host_shifts(const GatherArgs &a, const gvpm_host_shift *results, uint32_t n, hipStream_t s);
// the exact pass over the shifts the evaluation deferred (exact_shift.hip); totals: {evaluated, lost}
void launch_exact_pass(const GatherArgs &a, unsigned long long *totals, uint32_t *hostOut, hipStream_t s);
void launch_capture_notes(const GatherArgs &a, hipStream_t s, uint32_t nblocks = 256);
void launch_exact_beams(const GatherArgs &a, unsigned long long *totals, hipStream_t s);
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
// units / unitCtl / unitCap: the evaluation's work units (gather_bre.hip, EVAL_UNIT) -- two lists of unitCap entries, their
// lengths in unitCtl[0..1] (zero on entry); null: the evaluation takes whole items
uint32_t eval_unit_pairs();
void launch_traverse_bre(const GatherArgs &a, int beamsPerWave, const uint4 *items, const uint2 *itemOff,
                         const uint32_t *itemCount, uint32_t *queueHead, uint32_t *pairs, uint32_t *pairCnt,
                         uint32_t nwaves, bool persistent, hipStream_t stream, uint2 *units = nullptr, uint32_t *unitCtl = nullptr,
                         uint32_t unitCap = 0);
void launch_apply_host_shifts_beams(const GatherArgs &a, const gvpm_host_shift *results, uint32_t n, hipStream_t s);
void launch_evaluate_primal(const GatherArgs &a, int beamsPerWave, const uint4 *items, const uint2 *itemOff,
                            const uint32_t *itemCount, uint32_t *queueHead, const uint32_t *pairs, const uint32_t *pairCnt,
                            uint32_t nwaves, hipStream_t stream);
void launch_evaluate_bre(const GatherArgs &a, int beamsPerWave, bool fullVis, const uint4 *items, const uint2 *itemOff,
                         const uint32_t *itemCount, uint32_t *queueHead, const uint32_t *pairs, const uint32_t *pairCnt,
                         uint32_t nwaves, bool persistent, hipStream_t stream, const uint2 *units = nullptr,
                         const uint32_t *unitCtl = nullptr, uint32_t unitCap = 0);
uint32_t plan_items_capacity(uint32_t nsets, uint32_t ntiles, int beamsPerWave);
// the G-BRE build (cells, beam sort, summed-volume table, planner, scatter) as one chain of six launches (grid_build.hip)
void launch_build_chain(ChainArgs c, const GatherArgs &a, const gvpm_photon_soa &raw, int beamsPerWave, uint32_t target, uint4 *items,
                        uint2 *itemOff, uint32_t itemCap, float dmax, const NearGrid &ng, uint32_t extCap, uint32_t *origIdx,
                        uint32_t *hostOut, bool initBuckets, hipStream_t s, uint32_t pairCapBlocks = 0xFFFFFFFFu, uint32_t unitCap = 0,
                        uint32_t unitPairs = 1, bool fullVis = false);
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
void launch_gather_vpm(const GatherArgs &a, bool fullVis, bool primal, hipStream_t stream);
void launch_vpm_find(const GatherArgs &a, const VpmSplit &sp, hipStream_t stream);
void launch_vpm_redo(const GatherArgs &a, const VpmSplit &sp, bool fullVis, uint32_t nwaves, hipStream_t stream);
void launch_vpm_eval(const GatherArgs &a, const VpmSplit &sp, bool fullVis, uint32_t wavesPerShard, hipStream_t stream);
void launch_vpm_finish(float *accum, float *iter, float *scaleVol, float *nVol, float *mvol, size_t n, float alpha,
                       uint32_t *maxScaleBits, hipStream_t stream);
void launch_accumulate(float *accum, float *iter, size_t n, uint32_t *zeroWord, hipStream_t stream);
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
                        const Grid &g, uint32_t *keys, uint32_t *vals, hipStream_t s);
void launch_sub_hot(const uint32_t *sortedIds, uint32_t n, const float4 *aux, float4 *hot, uint32_t *hotFlags, hipStream_t s);
void launch_traverse_beams(const GatherArgs &a, const uint32_t *hotFlags, int beamsPerWave, const uint4 *items,
                           const uint32_t *itemCount, uint32_t itemCap, uint32_t *queueHead, uint2 *pairs, uint32_t *pairCount,
                           uint32_t pairCap, uint32_t *blockKey, uint32_t *blockVal, uint32_t nwaves, hipStream_t stream);
void launch_evaluate_beams_split(const GatherArgs &a, uint32_t *qId, uint32_t *qMeta, float4 *qK, float *qU, uint32_t *blkCnt,
                                 uint4 *runTab, uint32_t *ctl, const uint2 *pairs, const uint32_t *sortedKey,
                                 const uint32_t *sortedBlock, uint32_t nBlocks, uint32_t *queueHead, uint32_t ncu,
                                 hipStream_t stream);
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
extern RcclApi g_rccl;

#define GVPM_PHASES 3
// counters: GVPM_STAT_ROWS rows of 8 (device_types.h), one per persistent wave / a few workgroups each, summed on read

// Everything a gather reads that is rebuilt per photon set / beam set.  Two of them: G-BRE builds
// step N+1 (grid, sorts, planner) on a second stream while the evaluation kernel of step N runs.
struct BuildSet {
  float builtRadius = -1.f;
  DevBuf<float4> hot, cold;
  DevBuf<uint32_t> overflowCtr;  // photons whose near-occluder list overflowed (they need the BVH kernels)
  DevBuf<uint32_t> origIdx;      // sorted photon -> index in the upload (host-shift requests; filled only when enabled)
  DevBuf<uint32_t> cellStart, cellCount, cellSub, sat, keysA, keysB, valsA, valsB;
  DevBuf<uint32_t> beamCount, beamStart;  // counting sort of the beam sets
  DevBuf<float> boundsPartial, bounds6;
  Grid grid;
  SortTemp sortTmp;
  DevBuf<uint32_t> bKeysA, bKeysB, bValsA, setPerm, tileStart;
  uint32_t ntiles = 0;
  int tileW = 4, tileH = 4;
  bool scanSized = false, scanSizedChain = false;
  bool countsClean = false;  // cellCount is all zero (the build chain hands it back so)
  size_t subCleanWords = 0;  // cellSub's first this-many words (the striped counters of the bundle cells) are zero, likewise
  DevBuf<uint32_t> chainBuckets;  // the build chain's bounds buckets (64 x 6), reset by the chain itself after the first use
  bool bucketsInit = false;
  DevBuf<uint4> items;
  DevBuf<uint2> itemOff;
  DevBuf<uint2> planBoxes;  // the planner's slab boxes, read by the traversal (GatherArgs::planBoxes)
  DevBuf<uint32_t> queueCtl;   // [0] itemCount, [1] queueHead, [2] queueHead of the evaluation kernel, [3] pair blocks
  // G-BRE: per-beam photon lists between the traversal and the evaluation kernel
  DevBuf<uint32_t> pairs, pairCnt, nearExt;
  DevBuf<uint2> units;  // the evaluation's work units: two lists (large parts, small parts), written by the traversal
  // the notes of this set's evaluation ({count, ticket} + list; exact_shift.hip): its own, because the evaluations of two sets
  // may run side by side
  DevBuf<uint4> notes;
  DevBuf<uint32_t> notesCount;
  hipEvent_t traversed = nullptr;  // recorded on the build stream after the traversal kernel
  hipEvent_t lastUse = nullptr;  // recorded on the gather stream after the kernels that read this set
  bool used = false;
  bool lastUseValid = false;  // lastUse has been recorded
  // Give this (so far unused) set the capacities of the set that just ran its first step, so that the second
  // step of a run does not stop for gigabytes of hipMalloc in the middle of the pipeline.
  hipError_t mirrorFrom(const BuildSet &o) {
    hipError_t e = hipSuccess;
#define GVPM_MIRROR(X) if (e == hipSuccess) e = X.reserveExact(o.X.cap)
    GVPM_MIRROR(hot); GVPM_MIRROR(cold); GVPM_MIRROR(overflowCtr); GVPM_MIRROR(origIdx); GVPM_MIRROR(cellStart); GVPM_MIRROR(cellCount); GVPM_MIRROR(cellSub);
    GVPM_MIRROR(sat); GVPM_MIRROR(keysA); GVPM_MIRROR(keysB); GVPM_MIRROR(valsA); GVPM_MIRROR(valsB);
    GVPM_MIRROR(beamCount); GVPM_MIRROR(beamStart); GVPM_MIRROR(boundsPartial); GVPM_MIRROR(bounds6);
    GVPM_MIRROR(bKeysA); GVPM_MIRROR(bKeysB); GVPM_MIRROR(bValsA); GVPM_MIRROR(setPerm); GVPM_MIRROR(tileStart);
    GVPM_MIRROR(items); GVPM_MIRROR(itemOff); GVPM_MIRROR(planBoxes); GVPM_MIRROR(queueCtl); GVPM_MIRROR(pairs); GVPM_MIRROR(pairCnt);
    GVPM_MIRROR(nearExt); GVPM_MIRROR(chainBuckets); GVPM_MIRROR(units);
#undef GVPM_MIRROR
    return e;
  }
  void release() {
    hot.release(); cold.release(); overflowCtr.release(); origIdx.release(); cellStart.release(); cellCount.release(); cellSub.release(); sat.release();
    beamCount.release(); beamStart.release(); keysA.release(); keysB.release();
    valsA.release(); valsB.release(); boundsPartial.release(); bounds6.release(); bKeysA.release(); bKeysB.release();
    bValsA.release(); setPerm.release(); tileStart.release(); items.release(); itemOff.release(); planBoxes.release(); queueCtl.release();
    if (sortTmp.d) (void)hipFree(sortTmp.d);
    sortTmp.d = nullptr;
    sortTmp.bytes = 0;
    pairs.release(); pairCnt.release(); nearExt.release(); chainBuckets.release(); units.release(); notes.release(); notesCount.release();
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
  // G-BRE (round 6): consecutive evaluations on ALTERNATING streams (`stream`, `streamA2`).  They read different build sets and
  // add to the running sums with atomics, so nothing orders them but the stream -- and on one stream the next evaluation
  // waited for the last wave of this one (a third of a launch is its tail) plus two kernel boundaries around capture_notes.
  // Everything that reads or rescales the sums joins both first (gvpm_join_exact).  GVPM_EVAL_ALT=0: one stream.
  hipStream_t streamA2 = nullptr;
  bool evalAlt = false;  // (measured at C2, round 6: 1.10-1.12 ms a step either way -- each evaluation then takes 1.4 ms beside the next instead of 1.0 alone)
  int evalToggle = 0;
  hipEvent_t exactDone = nullptr;  // behind the last exact pass (gather stream)
  bool exactDoneValid = false;
  hipStream_t lastEvalStream = nullptr;  // where the last gather's evaluation went (its camera rays are read until it ends)
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
  bool bundleFromEnv = false;  // GVPM_BUNDLE given: the automatic choice below is off
  bool bundleAuto = false;     // GVPM_BUNDLE_AUTO=1: bundle cells chosen per build for image-sharded input (the default until round 5)
  // Round 4: chosen per build when GVPM_BUNDLE is not set -- ON when the uploaded beam sets cover at most half of the
  // frame's pixels, i.e. the handle is one rank of an image-sharded run (its traversal walks the whole volume for a
  // fraction of the rays: the case the bundle cells measured -8 % on, C4 rank step 3.09 -> 2.85 ms), OFF otherwise.
  int bundleState = 0, bundleViolations = 0;
  int lastGridMode = 0;  // of the last G-BRE build (gvpm_stats::reserved[0])
  uint32_t lastGridCells = 0;
  float bundleDiv = 2.f;  // level-0 cells per tile width (GVPM_BUNDLE_DIV)
  Grid bundleGrid{};
  bool buildChain = true;         // G-BRE: the build as one chain of six launches (GVPM_BUILD_CHAIN=0: the separate launches)
  DevBuf<uint32_t> chainCtl;      // its arrival counters
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
    DevBuf<uint32_t> packed;  // a packed upload lands here and is decoded into raw (uploads.hip) by the consuming gather
    bool needUnpack = false;
    bool linked = false;      // `packed` holds a blob of linked records (gvpm_upload_photons_linked)
    hipEvent_t unpacked = nullptr;
    hipEvent_t copied = nullptr;
    // recorded, on the gather stream and on the build stream, behind the kernels of every gather that read the slot
    // (builds; the G-Planes gather itself): the next copy into the slot waits for both
    hipEvent_t consumed = nullptr, consumedB = nullptr;
    bool read = false;      // a gather has launched kernels that read it since its last copy
  } phSlot[3];
  struct RaySlot {
    DevBuf<gvpm_camera_ray> rays;
    DevBuf<uint32_t> packed;  // a packed upload lands here and is decoded into rays by the consuming gather
    DevBuf<uint32_t> compact; // compact sets (60 bytes each) of a compact upload; its full sets land in `packed`
    uint32_t ncompact = 0;    // the first ncompact sets of the slot are decoded from `compact`, the others from `packed`
    bool needUnpack = false;
    hipEvent_t unpacked = nullptr;
    uint32_t nsets = 0;
    hipEvent_t copied = nullptr, freed = nullptr;
    bool read = false;      // a gather has launched kernels that read it since its last copy
  } raySlot[3];
  int phCur = 0, phPending = -1, rayCur = 0, rayPending = -1;   // pending: prefetched, current after the next gather
  bool phWait = false, rayWait = false;   // the next gather's streams must wait for the current slot's copy
  bool raysOwnedCur = false;              // the current camera rays live in raySlot[rayCur]
  bool photonsOwnedCur = false;           // the current photon map lives in phSlot[phCur]
  hipStream_t copyStream = nullptr;
  // manifold shifts through the host (gvpm_enable_host_shifts): the requests of the last G-BRE gather
  uint64_t reqCap = 0;
  DevBuf<gvpm_shift_request> reqHost;
  DevBuf<float4> reqCtx;
  DevBuf<uint32_t> reqCount;
  DevBuf<gvpm_host_shift> reqResults;
  // The exact pass (exact_shift.hip; ExEntry in device_types.h): the entries outlive the gathers; the pass runs on the
  // gather stream when something reads or rescales the sums (gvpm_join_exact: downloads, statistics, all-reduce, a new
  // scene / medium / BSDF table, another technique) and every exFlushEvery gathers.  exTotals: {evaluated, lost, largest
  // list, -, by cause ...}.
  DevBuf<ExEntry> exPay;
  DevBuf<uint32_t> exPayCount;       // {entries, lost notes, ticket}
  // the notes of a gather ({count, ticket} + list).  Their copy kernel runs on the gather stream behind the evaluation
  // (measured at C2: ~25 us of the step; on a stream of its own, ordered by events, the step LOST 6 %)
  DevBuf<uint4> exOvf;
  DevBuf<uint32_t> exOvfCount;
  // G-VPM leaves its per-iteration buffers zeroed behind it (accumulate_kernel, vpm_update_kernel): the next G-VPM gather then
  // skips three memsets -- 25 us of launches in a 0.6 ms step at C1.  False whenever something else may have written `iter`.
  bool iterClean = false;
  DevBuf<unsigned long long> exTotals;
  uint32_t exPayCap = 1u << 17;      // 64 MB, allocated with the first gather that can defer; regrown x 4 when a pass finds it a quarter full
  uint32_t exOvfCap = 1u << 20;      // 16 MB of notes per gather
  uint32_t exSince = 0;              // gathers since the last pass
  uint32_t exFlushEvery = 8;         // paced by what the last pass found (pinExact[0]): the list is kept below a quarter full
  bool exFlushFixed = false;         // GVPM_EXACT_EVERY
  uint32_t exSinceAtLast = 8;        // gathers the last pass covered
  uint32_t *pinExact = nullptr;      // pinned: {entries the last pass found}
  bool reqOutstanding = false;   // a gather recorded requests that were neither answered nor written off yet
  bool reqBeams = false;         // ... of a G-Beams gather (five float4 of context a request, its own apply kernel)
  GatherArgs reqArgs;            // of that gather (medium, film, iteration scale)
  DevBuf<gvpm_material> materials;  // gvpm_upload_materials: the table the packed photon records index
  uint32_t nmaterials = 0;
  std::vector<gvpm_material> materialsHost;  // what the device table holds (append-only growth needs no stream sync)
  DevBuf<float4> bsdfs;             // gvpm_upload_bsdfs: 4 float4 per glossy surface BSDF (device_types.h, GatherArgs::bsdfs)
  uint32_t nbsdfs = 0;
  gvpm_sensor sensor{};             // gvpm_upload_sensor: what the compact beam sets are decoded with
  bool haveSensor = false;
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
  // G-BRE: the last step's camera beams' bounds (their base rays' segments), which clip the grid (buildGrid; GVPM_CLIP_GRID=0: off)
  float beamB6[6] = {0, 0, 0, 0, 0, 0};
  bool haveBeamBounds = false, clipGrid = true;
  float *pinB6 = nullptr;      // pinned host staging: 6 floats + 2 uint32
  uint32_t *pinCtl = nullptr;
  uint32_t vpmOrderN = 0;      // batches the G-VPM order in blockValB was sorted for (0: none)
  uint32_t vpmLaunches = 0;
  bool vpmNoOrder = false;     // GVPM_VPM_ORDER=0
  // G-VPM: a bound on the largest per-pixel scale (gatherVPM: the initial scale, then what the iterations have exported); the
  // grid of step N + 1 built on the build stream beside the gather of step N (GVPM_VPM_PIPELINE=0: behind it)
  float vpmScaleBound = 0.f;
  bool vpmPipeline = true;
  // G-VPM as walk + evaluation + redo kernels (gather_vpm.hip; GVPM_VPM_SPLIT=0: the fused kernel)
  bool vpmSplit = true;
  uint32_t vpmPoolPerBatch = 4;  // chunks of 64 pairs the pool holds per batch of 64 samples (GVPM_VPM_POOL; tests shrink it so that batches take the redo path)
  uint32_t vpmRedoWaves = 256;   // persistent waves of the redo kernel
  uint32_t vpmEvalWaves = 12288; // ... and of the evaluation (GVPM_VPM_EVAL_WAVES; 3072 / 6144 / 12288 / one a group: 0.413 / 0.388 / 0.362 / 0.362 ms at C1: the waves differ in length)
  DevBuf<uint2> vpmPairs, vpmChunkMeta;
  DevBuf<uint32_t> vpmCtl, vpmStatus, vpmRedo;
  DevBuf<VpmSampleState> vpmState;
  hipEvent_t vpmFound = nullptr, vpmRedone = nullptr;
  // GVPM_BEAMS_SPLIT=1: the evaluation in two kernels (gather_beams.hip); the reconnection entries between them
  bool beamsSplit = false;
  DevBuf<uint32_t> splitId, splitMeta, splitBlkCnt, splitCtl;
  DevBuf<float4> splitK;
  DevBuf<float> splitU;
  DevBuf<uint4> splitRuns;
  float *pinBeams = nullptr;   // the G-Beams driver's: 2 x 6 bounds (floats 0-5, 8-13), counters (uint32 from float 16 on)

  // camera beams
  const gvpm_camera_ray *raysDev = nullptr;
  uint32_t nsets = 0;
  bool haveBeams = false, beamsDirty = false;

  // G-Beams: raw upload shares rawF/rawU/rawDev with the photons; end normals + sub-beam build
  DevBuf<float> endNOwned;
  const float *endNDev = nullptr;
  bool haveBeamsMap = false;
  DevBuf<uint32_t> subCounts, subOffsets, beamCtl;
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
  uint32_t planTarget = 1536;  // staged photons per work item (G-BRE; 1024 until the end of round 4: C2 -0.8 %, C4 whole -1.8 %, a rank of 8 -5 %)
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
  // G-BRE: traversal and evaluation are queued behind the build BEFORE the host has read the planner's counters, sized for
  // what the buffers hold; the build's last block checks (grid_build.hip, TailArgs) and the host queues them again when the
  // guess was wrong (GVPM_OPTIMISTIC=0: always after the host's wait)
  bool optimistic = true;
  uint32_t lastItems = 0;  // the planner's item count of the last G-BRE step (the traversal's grid of an optimistic step)
  uint32_t optRefuseEvery = 0, optSteps = 0, optRefused = 0;  // GVPM_OPTIMISTIC_REFUSE (tests); steps the guard refused
  bool evalUnits = true;  // G-BRE: the evaluation's queue serves work units, large parts first (GVPM_EVAL_UNITS=0: whole items in order)
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

static inline int fail(gvpm_context *h, int code, const char *msg) {
  if (h) h->err = msg;
  return code;
}


namespace gvpm {
void launch_unpack_photons(const uint32_t *packed, uint32_t n, const gvpm_material *table, uint32_t table_n,
                           const gvpm_photon_soa &dst, unsigned long long *bad, hipStream_t s);
void launch_unpack_rays(const uint32_t *packed, uint32_t nsets, gvpm_camera_ray *dst, hipStream_t s);
void launch_unpack_linked(const uint32_t *blob, uint32_t n, const gvpm_material *table, uint32_t table_n, const gvpm_photon_soa &dst,
                          unsigned long long *bad, hipStream_t s);
void launch_unpack_compact_rays(const gvpm_sensor &sensor, const uint32_t *compact, uint32_t ncompact, gvpm_camera_ray *dst,
                                hipStream_t s);
}  // namespace gvpm

// shared between the files above
// runs the exact pass over the deferred shifts, if a gather may have left any (before anything reads or rescales the sums)
int gvpm_join_exact(gvpm_context *h);
int joinEvalStreams(gvpm_context *h);
int flushHostShifts(gvpm_context *h);  // unanswered shift requests become failed shifts (before anything reads the film)
float currentRadius(const gvpm_context *h);

