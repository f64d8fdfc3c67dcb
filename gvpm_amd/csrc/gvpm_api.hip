// C ABI of libgvpm_hip.so (include/gvpm_hip.h): handle, uploads, per-iteration driver.
// Mirrors the driver logic of GPMIntegrator::photonMapPass / computeVolumeGradientPhotonBRE
// (gvpm/gvpm.cpp:383-500, 988-1079) and scaleVolumeAPA (gvpm.cpp:181-215).
#include "context.h"

RcclApi g_rccl;

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
      hipStreamCreateWithPriority(&h->streamA2, hipStreamNonBlocking, prA) != hipSuccess ||
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
  if (const char *e = getenv("GVPM_TRAV_PREFILTER"))
    if (!atoi(e)) h->cfg.reserved[0] |= 128;  // (gather_bre.hip: traverse_bre_kernel's tile cylinder)
  if (const char *e = getenv("GVPM_RECORD_PREFETCH")) h->cfg.reserved[0] |= atoi(e) ? 64 : 32;  // (gather_bre.hip: launch_evaluate_bre)
  h->cfg.reserved[1] = 0;  // slab layers per step (0 = default)
  if (const char *e = getenv("GVPM_SLAB_LAYERS")) h->cfg.reserved[1] = atoi(e);
  h->cfg.reserved[2] = 0;  // slab layers per step when x is the major axis (0 = default)
  if (const char *e = getenv("GVPM_SLAB_LAYERS_X")) h->cfg.reserved[2] = atoi(e);
  // (every reserved word is the library's: a caller's struct that was never zeroed must not switch anything -- reserved[5]
  // is the G-Beams primal pass's flag, set per launch by its driver)
  h->cfg.reserved[4] = h->cfg.reserved[5] = 0;
  if (const char *e = getenv("GVPM_EXACT_ALL")) h->cfg.reserved[4] = atoi(e) ? 1 : 0;  // (tests: every shift through the exact pass)
  if (h->cfg.reserved[4]) h->exPayCap = 1u << 20;  // (... which then holds every shift of a gather, not one in 1e5)
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
  if (const char *e = getenv("GVPM_BUILD_CHAIN")) h->buildChain = atoi(e) != 0;
  if (const char *e = getenv("GVPM_EVAL_UNITS")) h->evalUnits = atoi(e) != 0;
  if (const char *e = getenv("GVPM_OPTIMISTIC")) h->optimistic = atoi(e) != 0;
  if (const char *e = getenv("GVPM_OPTIMISTIC_REFUSE")) h->optRefuseEvery = (uint32_t)std::max(0, atoi(e));
  if (const char *e = getenv("GVPM_CLIP_GRID")) h->clipGrid = atoi(e) != 0;
  if (const char *e = getenv("GVPM_EVAL_ALT")) h->evalAlt = atoi(e) != 0;
  if (const char *e = getenv("GVPM_VPM_ORDER")) h->vpmNoOrder = atoi(e) == 0;
  if (const char *e = getenv("GVPM_VPM_SPLIT")) h->vpmSplit = atoi(e) != 0;
  if (const char *e = getenv("GVPM_VPM_PIPELINE")) h->vpmPipeline = atoi(e) != 0;
  if (const char *e = getenv("GVPM_VPM_POOL")) h->vpmPoolPerBatch = (uint32_t)std::max(0, atoi(e));
  if (const char *e = getenv("GVPM_VPM_EVAL_WAVES")) h->vpmEvalWaves = (uint32_t)std::max(64, atoi(e));
  if (const char *e = getenv("GVPM_VPM_REDO_WAVES")) h->vpmRedoWaves = (uint32_t)std::max(1, atoi(e));
  if (const char *e = getenv("GVPM_BEAMS_SPLIT")) h->beamsSplit = atoi(e) != 0;
  if (const char *e = getenv("GVPM_BUNDLE_AUTO")) h->bundleAuto = atoi(e) != 0;
  if (const char *e = getenv("GVPM_BUNDLE")) {
    h->bundleEnabled = atoi(e) != 0;
    h->bundleFromEnv = true;
  }
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
  if (const char *e = getenv("GVPM_EXACT_EVERY")) {  // gathers between two exact passes (probes)
    const int v = atoi(e);
    if (v >= 1 && v <= 1000000) {
      h->exFlushEvery = (uint32_t)v;
      h->exFlushFixed = true;
    }
  }
  if (const char *e = getenv("GVPM_CELL_SCALE")) {
    float v = (float)atof(e);
    if (v >= 0.25f && v <= 8.f) h->cellScale = v;
  }
  h->npix = (size_t)params->width * params->height;
  if (h->accum.ensure(h->npix * 27) != hipSuccess || h->iter.ensure(h->npix * 27) != hipSuccess ||
      h->stats.ensure(8 * GVPM_STAT_ROWS) != hipSuccess || h->scaleVol.ensure(h->npix) != hipSuccess ||
      h->nVol.ensure(h->npix) != hipSuccess || h->mvol.ensure(h->npix) != hipSuccess ||
      h->maxScaleBits.ensure(2) != hipSuccess || h->exTotals.ensure(32) != hipSuccess ||
      h->exPayCount.ensure(4) != hipSuccess || h->exOvfCount.ensure(4) != hipSuccess) {
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
  if (h->streamA2) (void)hipStreamSynchronize(h->streamA2);
  for (BuildSet &b : h->sets) b.release();
  h->tri4.release(); h->bvh.release();
  for (auto &ps : h->phSlot) {
    ps.raw.release();
    ps.packed.release();
    if (ps.unpacked) (void)hipEventDestroy(ps.unpacked);
    if (ps.copied) (void)hipEventDestroy(ps.copied);
    if (ps.consumed) (void)hipEventDestroy(ps.consumed);
    if (ps.consumedB) (void)hipEventDestroy(ps.consumedB);
  }
  for (auto &rs : h->raySlot) {
    rs.rays.release();
    rs.packed.release();
    if (rs.unpacked) (void)hipEventDestroy(rs.unpacked);
    if (rs.copied) (void)hipEventDestroy(rs.copied);
    if (rs.freed) (void)hipEventDestroy(rs.freed);
  }
  if (h->copyStream) (void)hipStreamDestroy(h->copyStream);
  h->materials.release();
  h->reqHost.release(); h->reqCtx.release(); h->reqCount.release(); h->reqResults.release();
  h->endNOwned.release(); h->subCounts.release(); h->subOffsets.release();
  h->beamCtl.release(); h->beamAux.release(); h->beamClear.release();
  h->nearGridStart.release(); h->nearGridTris.release(); h->nearGridCount.release();
  h->w1Owned.release(); h->len1Owned.release(); h->planeTest.release(); h->beamPairs.release(); h->subFlags.release(); h->shiftExtent.release();
  h->blockKeyA.release(); h->blockKeyB.release(); h->blockValA.release(); h->blockValB.release();
  h->samplesOwned.release(); h->scaleVol.release(); h->nVol.release(); h->mvol.release(); h->maxScaleBits.release();
  poisson_graph_release(h->poissonGraph);
  h->accumTmp.release();
  h->chainCtl.release();
  h->poissonScratch.release(); h->poissonIO.release();
  h->accum.release(); h->accumAll.release(); h->iter.release(); h->filmOut.release(); h->emission.release(); h->stats.release();
  h->exPay.release(); h->exPayCount.release(); h->exOvf.release(); h->exOvfCount.release(); h->exTotals.release();
  if (h->pinB6) (void)hipHostFree(h->pinB6);
  if (h->pinExact) (void)hipHostFree(h->pinExact);
  if (h->pinBeams) (void)hipHostFree(h->pinBeams);
  h->splitId.release(); h->splitMeta.release(); h->splitBlkCnt.release(); h->splitCtl.release(); h->splitK.release();
  h->splitU.release(); h->splitRuns.release();
  if (h->stream) (void)hipStreamDestroy(h->stream);
  if (h->streamB) (void)hipStreamDestroy(h->streamB);
  if (h->streamC) (void)hipStreamDestroy(h->streamC);
  if (h->streamA2) (void)hipStreamDestroy(h->streamA2);
  if (h->exactDone) (void)hipEventDestroy(h->exactDone);
  h->vpmPairs.release(); h->vpmChunkMeta.release(); h->vpmCtl.release(); h->vpmStatus.release(); h->vpmRedo.release(); h->vpmState.release();
  if (h->vpmFound) (void)hipEventDestroy(h->vpmFound);
  if (h->vpmRedone) (void)hipEventDestroy(h->vpmRedone);
  delete h;
  return GVPM_OK;
}

int gvpm_reset(gvpm_context *h) {
  CHECK_H(h);
  // Host-shift requests of a gather that were neither answered nor written off are DISCARDED: their base terms belong to
  // the run that ends here (flushed later they would land in the zeroed accumulators).  The G-VPM batch order is the old
  // run's too.
  {
    const int rcj = joinEvalStreams(h);  // (evaluations still in flight on either stream add to the sums cleared below)
    if (rcj != GVPM_OK) return rcj;
  }
  h->reqOutstanding = false;
  h->reqBeams = false;
  if (h->reqCount.p) HIP_TRY(h, hipMemsetAsync(h->reqCount.p, 0, 8, h->stream));
  h->vpmOrderN = 0;
  h->vpmLaunches = 0;
  // (deferred shifts of the run that ends here are dropped with its sums)
  HIP_TRY(h, hipMemsetAsync(h->exPayCount.p, 0, 4 * sizeof(uint32_t), h->stream));
  HIP_TRY(h, hipMemsetAsync(h->exOvfCount.p, 0, 4 * sizeof(uint32_t), h->stream));
  h->exSince = 0;
  HIP_TRY(h, hipMemsetAsync(h->exTotals.p, 0, 32 * sizeof(unsigned long long), h->stream));
  HIP_TRY(h, hipMemsetAsync(h->accum.p, 0, h->npix * 27 * sizeof(float), h->stream));
  HIP_TRY(h, hipMemsetAsync(h->stats.p, 0, 8 * GVPM_STAT_ROWS * sizeof(unsigned long long), h->stream));
  h->globalScaleVolume = h->cfg.initial_scale_volume;  // gvpm.cpp:291
  h->sumIt = 0;
  h->bundleState = 0;
  h->bundleViolations = 0;
  HIP_TRY(h, hipMemsetAsync(h->iter.p, 0, h->npix * 27 * sizeof(float), h->stream));
  HIP_TRY(h, hipMemsetAsync(h->mvol.p, 0, h->npix * sizeof(float), h->stream));
  h->iterClean = true;
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
    // (nothing of an earlier run is in flight any more: its exports cannot land behind this)
    h->vpmScaleBound = sc;
    if (h->pinCtl) h->pinCtl[32] = 0u;
  }
  return GVPM_OK;
}

int gvpm_upload_scene(gvpm_context *h, const gvpm_triangles *t) {
  CHECK_H(h);
  if (int rcj = gvpm_join_exact(h)) return rcj;  // (deferred shifts are evaluated against the scene they met)
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
  if (int rcj = gvpm_join_exact(h)) return rcj;  // (deferred shifts are evaluated against the scene they met)
  if (!m) return fail(h, GVPM_ERR_INVALID_ARG, "null medium");
  // homogeneous.cpp:196-200: the balance strategy requires equal sigma_t across channels
  if (m->sigma_t[0] != m->sigma_t[1] || m->sigma_t[0] != m->sigma_t[2])
    return fail(h, GVPM_ERR_UNSUPPORTED, "Not possible to have different albedo values...");
  if (!(m->sigma_t[0] > 0.f)) return fail(h, GVPM_ERR_INVALID_ARG, "sigma_t must be positive");
  h->medium = *m;
  h->haveMedium = true;
  return GVPM_OK;
}

int gvpm_upload_bsdfs(gvpm_context *h, const gvpm_bsdf *table, uint32_t n) {
  CHECK_H(h);
  if (int rcj = gvpm_join_exact(h)) return rcj;  // (deferred shifts are evaluated against the scene they met)
  if (n && !table) return fail(h, GVPM_ERR_INVALID_ARG, "null bsdf table");
  if (n > (1u << 24)) return fail(h, GVPM_ERR_INVALID_ARG, "more than 2^24 bsdfs (the index travels as a float)");
  // four quads per entry: {kind, specular} {exponent | alpha, sampling weight, distribution, sample_visible} {eta, k.x} {k.yz}
  std::vector<float4> rows(4 * (size_t)n + 4);
  for (uint32_t i = 0; i < n; ++i) {
    const gvpm_bsdf &b = table[i];
    float kindBits, distBits, visBits;
    memcpy(&kindBits, &b.kind, 4);
    memcpy(&distBits, &b.distribution, 4);
    memcpy(&visBits, &b.sample_visible, 4);
    if (b.kind == GVPM_BSDF_PHONG) {
      if (!(b.exponent >= 0.f) || !(b.specular_sampling_weight >= 0.f && b.specular_sampling_weight <= 1.f))
        return fail(h, GVPM_ERR_INVALID_ARG, "Phong: exponent >= 0 and a sampling weight in [0, 1]");
    } else if (b.kind == GVPM_BSDF_ROUGHCONDUCTOR) {
      if (!(b.exponent >= 1e-4f)) return fail(h, GVPM_ERR_INVALID_ARG, "rough conductor: alpha >= 1e-4 (the reference clamps it)");
      if (b.distribution != GVPM_MICROFACET_BECKMANN && b.distribution != GVPM_MICROFACET_GGX)
        return fail(h, GVPM_ERR_UNSUPPORTED, "rough conductor: Beckmann or GGX");
    } else if (b.kind == GVPM_BSDF_WARD) {
      // (both components: Ward::sampleComponent picks one below roughness 0.05, ward.cpp:370-389 -- such a surface is outside
      // the closed set; isotropic by construction: the entry has one alpha)
      if (!(b.exponent >= 0.05f) || !(b.specular_sampling_weight >= 0.f && b.specular_sampling_weight <= 1.f))
        return fail(h, GVPM_ERR_INVALID_ARG, "Ward: alpha >= 0.05 (both components) and a sampling weight in [0, 1]");
      if (b.sample_visible < GVPM_WARD_WARD || b.sample_visible > GVPM_WARD_BALANCED || b.distribution != 0)
        return fail(h, GVPM_ERR_UNSUPPORTED, "Ward: variant ward / ward-duer / balanced, both components");
    } else {
      return fail(h, GVPM_ERR_UNSUPPORTED, "bsdf kind outside the device's closed set (Phong, rough conductor, Ward)");
    }
    rows[4 * i] = make_float4(kindBits, b.specular[0], b.specular[1], b.specular[2]);
    rows[4 * i + 1] = make_float4(b.exponent, b.specular_sampling_weight, distBits, visBits);
    rows[4 * i + 2] = make_float4(b.eta[0], b.eta[1], b.eta[2], b.k[0]);
    rows[4 * i + 3] = make_float4(b.k[1], b.k[2], 0.f, 0.f);
  }
  // once per scene: waits for whatever still reads the old table (as gvpm_upload_materials does)
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->streamB));
  if (h->streamC) HIP_TRY(h, hipStreamSynchronize(h->streamC));
  if (h->copyStream) HIP_TRY(h, hipStreamSynchronize(h->copyStream));
  HIP_TRY(h, h->bsdfs.ensure(rows.size()));
  HIP_TRY(h, hipMemcpy(h->bsdfs.p, rows.data(), rows.size() * sizeof(float4), hipMemcpyHostToDevice));
  h->nbsdfs = n;
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

int gvpm_get_exact_shift_count(gvpm_context *h, uint64_t *evaluated, uint64_t *lost) {
  CHECK_H(h);
  if (int rcj = gvpm_join_exact(h)) return rcj;
  unsigned long long v[32];
  HIP_TRY(h, hipMemcpyAsync(v, h->exTotals.p, sizeof(v), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  // (the cause slots 4.. are a bit MASK's bits for G-BRE / G-VPM, exact_pass_kernel, and a cause NUMBER 0..9 for
  // G-Beams, exact_beams_kernel -- the same slots, so each technique's line is printed under its own labels)
  const bool beamsTech = h->cfg.vol_technique == GVPM_BEAM_BEAM_1D || h->cfg.vol_technique == GVPM_BEAM_BEAM_3D_OPTIMIZED;
  if (getenv("GVPM_TRACE_EXACT") && beamsTech) {
    fprintf(stderr, "[exact beams] evaluated %llu lost %llu largest list %llu; by cause number 0..9:", v[0], v[1], v[2]);
    for (int k = 0; k < 10; ++k) fprintf(stderr, " %llu", v[4 + k]);
    fprintf(stderr, "\n");
  } else if (getenv("GVPM_TRACE_EXACT"))
    fprintf(stderr, "[exact] evaluated %llu lost %llu largest list %llu; by cause: pair %llu branch %llu mirror %llu visibility %llu cosine %llu\n",
            v[0], v[1], v[2], v[4], v[5], v[7], v[8], v[9]);
  if (getenv("GVPM_TRACE_EXACT") && !beamsTech && (v[16] | v[17] | v[18] | v[19] | v[20]))
    fprintf(stderr, "[exact] visibility-caused, |cos| < .01 / .03 / .1 / .3 / more: surface parents %llu %llu %llu %llu %llu, medium parents %llu %llu %llu %llu %llu\n",
            v[16], v[17], v[18], v[19], v[20], v[21], v[22], v[23], v[24], v[25]);
  if (getenv("GVPM_TRACE_EXACT") && !beamsTech && (v[10] | v[11] | v[12] | v[13]))
    fprintf(stderr, "[exact] pairs: other %llu, rim of the kernel %llu, beyond the beam's end %llu, t' at an end %llu\n", v[10], v[11], v[12], v[13]);
  if (evaluated) *evaluated = v[0];
  if (lost) *lost = v[1];
  return GVPM_OK;
}

int gvpm_get_stats(gvpm_context *h, gvpm_stats *out) {
  CHECK_H(h);
  if (int rcj = gvpm_join_exact(h)) return rcj;
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
  // reserved[0]: kind of the last G-BRE grid << 56 | optimistic steps the build refused (24 bits) << 32 | its cells
  out->reserved[0] = ((uint64_t)h->lastGridMode << 56) | ((uint64_t)(h->optRefused & 0xFFFFFFu) << 32) | (uint64_t)h->lastGridCells;
  out->reserved[1] = v[6];
  if (v[6]) return fail(h, GVPM_ERR_STATE, "packed photon records named materials beyond the uploaded table (decoded as black)");
  // the planner's bound on an item's pair region is exact: a dropped pair means a biased image, not a slow one
  if (v[7]) return fail(h, GVPM_ERR_STATE, "pairs or deferred shifts were dropped (the G-BRE planner's bound violated, or the exact pass's lists full)");
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
  {
    int rcf = flushHostShifts(h);
    if (rcf == GVPM_OK) rcf = gvpm_join_exact(h);
    if (rcf != GVPM_OK) return rcf;
  }
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
  {
    int rcf = flushHostShifts(h);
    if (rcf == GVPM_OK) rcf = gvpm_join_exact(h);
    if (rcf != GVPM_OK) return rcf;
  }
  if (!accum_dev) return GVPM_ERR_INVALID_ARG;
  launch_scale(h->useAll ? h->accumAll.p : h->accum.p, accum_dev, h->npix * 27, accumScale(h), h->stream);
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return GVPM_OK;
}

// throughput | dx | dy planes into filmOut (device); emission already on the device or null
static int filmToDevice(gvpm_context *h, int it, int reuse_primal, const float *emissionDev, float *out) {
  {
    int rcf = flushHostShifts(h);
    if (rcf == GVPM_OK) rcf = gvpm_join_exact(h);
    if (rcf != GVPM_OK) return rcf;
  }
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
  if (int rcj = gvpm_join_exact(h)) return rcj;
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

