// The per-technique drivers behind gvpm_gather (C ABI, include/gvpm_hip.h): device builds (photon grid, beam sort, planner)
// and the kernel sequences of G-BRE, G-Beams, G-Planes and G-VPM.  Mirrors GPMIntegrator::photonMapPass and the
// computeVolumeGradient* drivers (gvpm/gvpm.cpp:383-500, 880-1180).
#include "context.h"

float currentRadius(const gvpm_context *h) {
  // breInitSize = bsphere.radius * globalScaleVolume * POURCENTAGE_BS, gvpm.cpp:989 (Float = float)
  return h->cfg.bsphere_radius * h->globalScaleVolume * 0.01f;
}

static int ilog2ceil(uint32_t v) {
  int b = 0;
  while ((1ull << b) < v) ++b;
  return b;
}

// G-BRE's build as one chain of launches (grid_build.hip, launch_build_chain): buildGrid / sortBeams then only size the
// buffers and the grid and leave here what the chain's launcher needs.
struct ChainPrep {
  bool on = false;
  float dmax = 0.f;
  bool wantOrig = false;
  uint32_t *sub = nullptr;
  uint32_t nkeys = 0, tileShift = 0;  // the beam sort's key space (sortBeams)
  int tw = 4, th = 4;
};
// tile shape and key space of the beam sort for `beamsPerWave` sets per wave
static void beamTiling(const gvpm_context *h, int beamsPerWave, int &tw, int &th, uint32_t &ntiles, int &tileShift) {
  tw = 8;
  th = 4;
  if (beamsPerWave == 64) th = 8;
  if (beamsPerWave == 16) tw = 4;
  const uint32_t tilesX = (h->cfg.width + tw - 1) / tw, tilesY = (h->cfg.height + th - 1) / th;
  ntiles = tilesX * tilesY;
  tileShift = ilog2ceil(tw * th) + 3;
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

// knownB6 (optional): the photons' bounds, already on the host (the caller read them back with something else)
static int buildGrid(gvpm_context *h, float r, bool deferred = false, bool force3D = false, const float *knownB6 = nullptr,
                     ChainPrep *cp = nullptr) {
  const uint32_t n = h->nph;
  h->boundsPending = false;
  const bool chain = cp && cp->on && n > 0;
  if (cp) cp->on = chain;
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
    HIP_TRY(h, hipHostMalloc((void **)&h->pinB6, 256, hipHostMallocMapped));
    memset(h->pinB6, 0, 256);
    h->pinCtl = reinterpret_cast<uint32_t *>(h->pinB6 + 8);
  }
  const bool defer = deferred && h->haveCachedBounds;
  if (!knownB6 && !(chain && defer))  // (the chain reduces the bounds in its own first two launches)
    launch_bounds(h->rawDev.pos, n, h->bs->boundsPartial.p, nblocks, h->bs->bounds6.p, defer ? h->pinB6 : nullptr,
                  h->bstream);
  if (knownB6) {
    memcpy(b6, knownB6, sizeof(b6));
    memcpy(h->cachedB6, b6, sizeof(b6));
    h->haveCachedBounds = true;
  } else if (defer) {
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
  const float cellScale0 = h->cellScale > 0.f ? h->cellScale : (deferred && n <= 2000000u ? 1.5f : 1.0f);
  if (deferred && h->clipGrid && h->haveBeamBounds) {
    // G-BRE (round 6): the grid covers what the camera beams can reach, not what the photons fill.  An open scene's photons
    // fly far (S-cbox: extents of 2-5 times the box from step to step, 1.5-16 M cells), the beams end on its surfaces: the
    // cell arrays, their scan and the summed-volume table shrink to the box.  Photons outside sit (clamped) in the border cells,
    // as photons outside LAST step's bounds always have -- the binning and the footprints clamp alike, so the evaluated set
    // does not depend on where the grid ends; two cells of margin keep the beams' footprints off those border cells.  The
    // beams' bounds are the last step's (reduced by its build), like the photons'.
    float lo[3], hi[3], cell = cellScale0 * r;
    for (int pass = 0; pass < 2; ++pass) {
      const float pad = 1.01f * r + 2.f * cell;
      float e = 0.f;
      for (int c = 0; c < 3; ++c) {
        lo[c] = fmaxf(b6[c], h->beamB6[c] - pad);
        hi[c] = fminf(b6[3 + c], h->beamB6[3 + c] + pad);
        if (!(lo[c] <= hi[c])) lo[c] = hi[c] = fminf(fmaxf(h->beamB6[c], b6[c]), b6[3 + c]);  // (no photon near the beams)
        e = fmaxf(e, hi[c] - lo[c]);
      }
      cell = fmaxf(cellScale0 * r, e / 384.f);
      ext = e;
    }
    for (int c = 0; c < 3; ++c) {
      b6[c] = lo[c];
      b6[3 + c] = hi[c];
    }
  }
  Grid g{};
  // cell edge in radii.  G-BRE with maps up to 2 M photons: 1.5 -- the cell arrays (memset, scan, summed-volume table:
  // ~135 of the build's 575 us at C2 with cells of one radius) shrink 3.4x, and there the build is the stage the
  // pipelined step waits for; the traversal tests 1.3x the photons per hit.  Measured at C2: 1.51 -> 1.43-1.445 ms per
  // step.  At C4 (4 M photons) the evaluation is the long stage and the larger cells cost the traversal 1 % of the step.
  const float cellScale = cellScale0;
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
  // (round 6: the automatic choice is OFF -- with the grid clipped to the beams' reach and the planner walking four chunks a wave
  // the 3D grid is the faster one for a rank's share too: C4, rank 0 of 8 / of 4: 2.36-2.39 / 3.90 ms a step against 2.47 /
  // 4.02 with the bundle cells, alternating runs.  GVPM_BUNDLE=1 forces them, GVPM_BUNDLE_AUTO=1 restores the choice.)
  if (deferred && !h->bundleFromEnv && h->bundleAuto) {
    // "a rank's share of an image-sharded frame": the beam sets cover at most half of the pixels.  With hysteresis (ADVICE
    // round 4): a frame whose medium covers about half the pixels must not flip the cell kind -- and the planner's tuning with
    // it -- from step to step; the kind is left only beyond 0.6 / below 0.4 of the frame.
    const size_t twice = (size_t)h->nsets * 2u;
    const bool sharded = h->nsets > 0 && (h->bundleEnabled ? twice * 5u <= h->npix * 6u : twice * 5u <= h->npix * 4u);
    if (sharded != h->bundleEnabled) {
      h->bundleEnabled = sharded;
      if (sharded && h->bundleState < 0) h->bundleState = 0;  // (a frame that failed earlier is tried again, a few times)
    }
  }
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
  // (the chain counts and scans the beam sets' tile keys behind the cells, in the same two arrays)
  const size_t keyRoom = chain ? (size_t)cp->nkeys + 4u : 0u;
  {
    const uint32_t *was = h->bs->cellCount.p;
    HIP_TRY(h, h->bs->cellCount.ensure(std::max((size_t)g.ncells + 2, worstCells) + keyRoom));
    if (h->bs->cellCount.p != was) h->bs->countsClean = false;
  }
  HIP_TRY(h, h->bs->cellStart.ensure(std::max((size_t)g.ncells + 2, worstCells) + keyRoom));
  if (!h->bs->scanSized || (chain && !h->bs->scanSizedChain)) {
    HIP_TRY(h, reserveScanTemp(h->bs->sortTmp, (uint32_t)std::min<size_t>(worstCells + keyRoom, 0x7FFFFFF0u)));
    h->bs->scanSized = true;
    h->bs->scanSizedChain = chain;
  }
  if (chain) {
    // the chain hands its counters back zeroed: one memset when the array is new, or was last used by the separate launches
    if (!h->bs->countsClean) HIP_TRY(h, hipMemsetAsync(h->bs->cellCount.p, 0, h->bs->cellCount.cap * sizeof(uint32_t), h->bstream));
    h->bs->countsClean = true;
  } else {
    HIP_TRY(h, hipMemsetAsync(h->bs->cellCount.p, 0, ((size_t)g.ncells + 1) * sizeof(uint32_t), h->bstream));
    h->bs->countsClean = false;
  }
  // (bundle cells: striped counters, grid_build.hip cell_count_kernel)
  uint32_t *sub = nullptr;
  if (g.mode == 1) {
    // (the chain: counters + their prefixes, and the counters are handed back zeroed -- one memset when the array is new, has
    // another size or was last used by the separate launches)
    const size_t subWords = (size_t)cell_stripes() * g.ncells;
    const uint32_t *was = h->bs->cellSub.p;
    HIP_TRY(h, h->bs->cellSub.ensure(subWords * (chain ? 2u : 1u)));
    if (!chain || h->bs->cellSub.p != was || h->bs->subCleanWords != subWords)
      HIP_TRY(h, hipMemsetAsync(h->bs->cellSub.p, 0, subWords * sizeof(uint32_t), h->bstream));
    h->bs->subCleanWords = chain ? subWords : 0u;
    sub = h->bs->cellSub.p;
  }
  // extension lists of the near-occluder lists: sized once per set for the largest photon count (grow only); word 0 is the
  // allocation cursor.  The count kernel starts it at 1 and the lists' overflow counter at 0 (two memsets less per build).
  HIP_TRY(h, h->bs->overflowCtr.ensure(2));
  const size_t extWant = std::min<size_t>(std::max<size_t>((size_t)n * 8u + 4096u, h->nearExtWant), 0xFFFFFF00u);
  HIP_TRY(h, h->bs->nearExt.ensure(extWant));
  if (!chain) {
    launch_cell_count(h->rawDev.pos, n, g, h->bs->keysA.p, h->bs->valsA.p, h->bs->cellCount.p, sub, h->bstream, h->bs->overflowCtr.p,
                      h->bs->nearExt.p);
    HIP_TRY(h, exclusiveSumU32(h->bs->sortTmp, h->bs->cellCount.p, h->bs->cellStart.p, g.ncells + 1, h->bstream));
  }
  if (deferred) {
    // G-BRE: summed-volume table for the planner (sized once for the finest grid, like the cell arrays)
    const size_t satCells = (size_t)(g.dim[0] + 1) * (g.dim[1] + 1) * (g.dim[2] + 1);
    HIP_TRY(h, h->bs->sat.ensure(std::max(satCells, (size_t)387 * 387 * 387)));
    if (!chain) launch_sat(h->bs->cellStart.p, g, h->bs->sat.p, h->bstream);
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
    // (the occluder grid belongs to the handle, not to a build set: a gather that still reads it on the other stream ends first)
    if (h->bstream != h->stream) HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->streamA2) HIP_TRY(h, hipStreamSynchronize(h->streamA2));
    const int rcg = buildNearGrid(h, dmax * 1.5f);
    if (rcg != GVPM_OK) return rcg;
  }
  const bool wantOrig = h->reqCap > 0 && h->cfg.use_manifold;  // (host-shift requests name photons by their place in the upload)
  if (wantOrig) HIP_TRY(h, h->bs->origIdx.ensure((size_t)n + 1));
  if (chain) {
    cp->dmax = dmax;
    cp->wantOrig = wantOrig;
    cp->sub = sub;
    h->nearOverflow = false;
    return GVPM_OK;
  }
  launch_reorder(h->rawDev, h->bs->keysA.p, h->bs->valsA.p, h->bs->cellStart.p, n, h->cfg, h->bvh.p, h->tri4.p, h->ntri, dmax,
                 h->nearGrid, h->bs->nearExt.p, (uint32_t)std::min<size_t>(h->bs->nearExt.cap, 0xFFFFFF00u), h->bs->hot.p,
                 h->bs->cold.p, h->bs->overflowCtr.p, wantOrig ? h->bs->origIdx.p : nullptr, sub, g.ncells, h->bstream);
  HIP_TRY(h, hipGetLastError());
  h->nearOverflow = false;
  if (!deferred && h->cfg.visibility_as_written) {
    if (!h->pinB6) {
      HIP_TRY(h, hipHostMalloc((void **)&h->pinB6, 256, hipHostMallocMapped));
    memset(h->pinB6, 0, 256);
      h->pinCtl = reinterpret_cast<uint32_t *>(h->pinB6 + 8);
    }
    launch_export_u32(h->bs->overflowCtr.p, nullptr, nullptr, nullptr, nullptr, h->pinCtl, h->bstream);
    HIP_TRY(h, hipStreamSynchronize(h->bstream));
    h->nearOverflow = h->pinCtl[0] != 0;
  }
  return GVPM_OK;
}

static int sortBeams(gvpm_context *h, int beamsPerWave = 0, const ChainPrep *cp = nullptr) {
  const uint32_t n = h->nsets;
  if (!beamsPerWave) beamsPerWave = h->beamsPerWave;
  int tw, th, tileShift;
  uint32_t ntilesAll;
  beamTiling(h, beamsPerWave, tw, th, ntilesAll, tileShift);
  h->bs->ntiles = ntilesAll;
  h->bs->tileW = tw;
  h->bs->tileH = th;
  HIP_TRY(h, h->bs->tileStart.ensure((size_t)h->bs->ntiles + 2));
  HIP_TRY(h, h->bs->bKeysA.ensure(n + 1));
  HIP_TRY(h, h->bs->bValsA.ensure(n + 1));
  HIP_TRY(h, h->bs->setPerm.ensure(n + 1));
  if (cp && cp->on) return GVPM_OK;  // (counted, scanned and scattered by the build chain)
  if (n) {
    // counting sort by (tile, pixel in tile, edge): count + rank, exclusive scan, scatter
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
  a.bsdfs = h->bsdfs.p;
  a.nbsdfs = h->nbsdfs;
  a.vpmOrder = nullptr;
  a.vpmOrderN = 0;
  a.vpmCostKey = a.vpmCostVal = nullptr;
  a.exPay = h->exPay.p;
  a.exPayCount = h->exPayCount.p;
  a.exPayCap = h->exPay.p ? h->exPayCap : 0u;
  a.exOvf = h->exOvf.p;
  a.exOvfCount = h->exOvfCount.p;
  a.exOvfCap = a.exOvf ? h->exOvfCap : 0u;
}

// The exact pass over the shifts the gathers deferred (exact_shift.hip): on the gather stream, when a reader asks
// (gvpm_join_exact) or every exFlushEvery gathers.
// the evaluations of G-BRE alternate between two streams: whatever reads or rescales the sums on the gather stream waits for both
int joinEvalStreams(gvpm_context *h) {
  if (h->streamA2)
    for (BuildSet &b : h->sets)
      if (b.lastUseValid && b.lastUse) HIP_TRY(h, hipStreamWaitEvent(h->stream, b.lastUse, 0));
  return GVPM_OK;
}
int gvpm_join_exact(gvpm_context *h) {
  {
    const int rcj = joinEvalStreams(h);
    if (rcj != GVPM_OK) return rcj;
  }
  if (h->exSince == 0) return GVPM_OK;
  GatherArgs a;
  fillArgs(h, a, 0.f);
  a.hot = a.cold = nullptr;
  a.rays = nullptr;
  a.iter = h->accum.p;
  launch_exact_pass(a, h->exTotals.p, h->pinExact, h->stream);
  HIP_TRY(h, hipGetLastError());
  // (the pass empties the handle's entry list: the capture kernels of later evaluations, on either stream, come after it)
  if (!h->exactDone) HIP_TRY(h, hipEventCreateWithFlags(&h->exactDone, hipEventDisableTiming));
  HIP_TRY(h, hipEventRecord(h->exactDone, h->stream));
  h->exactDoneValid = true;
  h->exSinceAtLast = h->exSince;
  h->exSince = 0;
  return GVPM_OK;
}
// before a gather that can defer: the lists exist; after its kernels have been queued: the cadence
static int exactPrepare(gvpm_context *h) {
  // (the list starts at 2^17 entries -- 64 MB, not 512 -- and is regrown, with the pass that empties it in front,
  // when a pass has found it more than a quarter full; the measured rate is ~1e-5 of the shifts)
  if (h->exPay.p && h->pinExact && h->pinExact[0] != 0xFFFFFFFFu && h->pinExact[0] > h->exPayCap / 4u && h->exPayCap < (1u << 22)) {
    const int rcj = gvpm_join_exact(h);
    if (rcj != GVPM_OK) return rcj;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->streamA2) HIP_TRY(h, hipStreamSynchronize(h->streamA2));
    h->exPay.release();
    h->exPayCap *= 4u;
    h->pinExact[0] = 0u;
  }
  HIP_TRY(h, h->exPay.reserveExact(h->exPayCap));
  HIP_TRY(h, h->exOvf.reserveExact(h->exOvfCap));
  if (!h->pinExact) {
    HIP_TRY(h, hipHostMalloc((void **)&h->pinExact, 64, hipHostMallocMapped));
    h->pinExact[0] = 0xFFFFFFFFu;  // (no pass has reported yet)
  }
  return GVPM_OK;
}
static int exactAfterGather(gvpm_context *h) {
  if (!h->exFlushFixed && h->pinExact[0] != 0xFFFFFFFFu) {
    // what the last pass found per gather it covered (read without waiting: a stale number only delays the adjustment)
    const uint32_t perGather = h->pinExact[0] / std::max(1u, h->exSinceAtLast) + 1u;
    h->exFlushEvery = std::max(1u, std::min(256u, (h->exPayCap / 4u) / perGather));
    static const bool trace = getenv("GVPM_TRACE_EXACT") != nullptr;
    if (trace) fprintf(stderr, "[exact] last pass found %u entries over %u gathers -> a pass every %u gathers (since %u)\n", h->pinExact[0], h->exSinceAtLast, h->exFlushEvery, h->exSince);
  }
  if (++h->exSince >= h->exFlushEvery) return gvpm_join_exact(h);
  return GVPM_OK;
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
static int gatherBRE(gvpm_context *h, int it, uint64_t nb_paths, bool primal = false) {
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
    rc = gvpm_join_exact(h);
    if (rc != GVPM_OK) return rc;
    launch_scale(h->accum.p, h->accum.p, h->npix * 27, (float)((double)(it - 1) / (double)h->sumIt), h->stream);
  }
  h->sumIt = it;
  rc = exactPrepare(h);
  if (rc != GVPM_OK) return rc;
  bool rebuilt = false;
  GatherArgs a;
  uint32_t itemCap = 0, blocks = 0, nItems = 0;
  bool force3D = false;
  const bool prevOverflow = h->nearOverflow;  // (of the last build: what an optimistic step picks its evaluation kernel by)
  bool chainRan = false;
  bool queued = false;                        // traversal + evaluation of this step are in their streams already

  // Traversal + evaluation of the built set, for `blocksQ` pair blocks and `nItemsQ` work items.  optimisticQ: queued before the
  // host has seen the planner's counters -- nothing may be regrown (a regrowth is a device-wide sync), the traversal's grid is a
  // guess (its waves take what lies beyond it from the queue), and the kernels wait for the build by an event.
  auto queueGather = [&](uint32_t blocksQ, uint32_t nItemsQ, bool optimisticQ, bool fullVisQ) -> int {
    int rq = GVPM_OK;
    if (!optimisticQ) {
      HIP_TRY(h, h->bs->pairs.ensure((size_t)blocksQ * 64u + 64u));
      HIP_TRY(h, h->bs->pairCnt.ensure((size_t)itemCap * h->beamsPerWave));
    }
    rq = nextEvents(h, &evTrav, 1);
    if (rq == GVPM_OK) rq = nextEvents(h, &evEval, 0);
    if (rq != GVPM_OK) return rq;
    // three stages: the traversal has its own stream, so that the build of the NEXT step (which starts on the build
    // stream as soon as this call returns) overlaps it
    hipStream_t ts = h->pipeline && h->travStream ? h->streamC : (h->travOnBuild ? h->streamB : h->stream);
    if (!h->pipeline) ts = h->stream;
    // the evaluation's work units (gather_bre.hip, EVAL_UNIT): an item yields at most staged x beams pairs, i.e. at most
    // blocks_i * 64 / unit + 1 parts; their counters are queueCtl[5..6] (zeroed with the queue heads)
    uint2 *units = nullptr;
    uint32_t unitCap = 0;
    if (h->evalUnits && h->persistentEval && !primal) {
      if (optimisticQ) {
        unitCap = (uint32_t)std::min<size_t>(h->bs->units.cap / 2u, 0x3FFFFFFFu);
        units = unitCap ? h->bs->units.p : nullptr;
      } else {
        const uint64_t capU = (uint64_t)blocksQ * 64u / eval_unit_pairs() + nItemsQ + 64u;
        if (capU < 0x3FFFFFFFull) {
          HIP_TRY(h, h->bs->units.ensure((size_t)capU * 2u));
          unitCap = (uint32_t)std::min<size_t>(h->bs->units.cap / 2u, 0x3FFFFFFFu);
          units = h->bs->units.p;
        }
      }
    }
    // ... and the evaluation alternates between two (context.h, streamA2)
    hipStream_t es = h->stream;
    if (h->evalAlt && h->pipeline && h->streamA2 && !primal && h->reqCap == 0) es = (h->evalToggle++ & 1) ? h->streamA2 : h->stream;
    h->lastEvalStream = es;
    if (!primal) {
      // this set's own note list (exact_shift.hip)
      if (!h->bs->notes.p) {
        HIP_TRY(h, h->bs->notes.reserveExact(h->exOvfCap));
        HIP_TRY(h, h->bs->notesCount.ensure(4));
        HIP_TRY(h, hipMemsetAsync(h->bs->notesCount.p, 0, 4 * sizeof(uint32_t), es));
      }
      a.exOvf = h->bs->notes.p;
      a.exOvfCount = h->bs->notesCount.p;
      a.exOvfCap = h->exOvfCap;
    }
    if (optimisticQ && ts != h->bstream) HIP_TRY(h, hipStreamWaitEvent(ts, evBuild->second, 0));
    HIP_TRY(h, hipEventRecord(evTrav->first, ts));
    launch_traverse_bre(a, h->beamsPerWave, h->bs->items.p, h->bs->itemOff.p, h->bs->queueCtl.p, h->bs->queueCtl.p + 1,
                        h->bs->pairs.p, h->bs->pairCnt.p, h->persistentTrav ? h->nwavesTrav : nItemsQ, h->persistentTrav || optimisticQ, ts,
                        units, h->bs->queueCtl.p + 5, unitCap);
    HIP_TRY(h, hipEventRecord(evTrav->second, ts));
    HIP_TRY(h, hipEventRecord(h->bs->traversed, ts));
    HIP_TRY(h, hipStreamWaitEvent(es, h->bs->traversed, 0));
    // maps beyond 2 M photons: the evaluation is the stage the pipelined step waits for (its records no longer fit the
    // Infinity Cache), so it gets its third wave per SIMD; below, the other stages need the room more (measured: +6 % on a
    // rank's step at C4 with 12 waves per CU, -3 % at C2)
    // (... when the evaluation is the long stage: a rank that holds an eighth of the frame evaluates for 1.3 ms beside a build of
    // 1.0 -- with 12 waves per CU the build starves.  C4, rank 0 of N emulated, 8 / 12 waves: N = 2: 7.14 / 6.91 ms per step,
    // N = 4: 3.94 / 3.93, N = 8: 2.33 / 2.47)
    const bool smallShare = h->nsets > 0 && (size_t)h->nsets * 6u <= h->npix;
    const uint32_t nwEval = (h->pipeline && !h->nwavesFromEnv && h->ncu && h->nph > 2000000u && !smallShare)
                                ? std::min<uint32_t>(h->ncu * 12u, GVPM_STAT_ROWS) : h->nwaves;
    if (!primal && h->reqCap > 0 && h->cfg.use_manifold && h->bs->origIdx.p) {
      // manifold-typed shifts are recorded for the host (gvpm_download_shift_requests) instead of failing
      HIP_TRY(h, h->reqHost.ensure(h->reqCap));
      HIP_TRY(h, h->reqCtx.ensure(4 * h->reqCap));
      HIP_TRY(h, h->reqCount.ensure(2));
      HIP_TRY(h, hipMemsetAsync(h->reqCount.p, 0, 8, es));
      a.reqHost = h->reqHost.p;
      a.reqCtx = h->reqCtx.p;
      a.reqCount = h->reqCount.p;
      a.reqCap = (uint32_t)h->reqCap;
      a.origIdx = h->bs->origIdx.p;
      h->reqArgs = a;
      h->reqOutstanding = true;
      h->reqBeams = false;
    }
    HIP_TRY(h, hipEventRecord(evEval->first, es));
    if (primal)
      // the primal beam radiance estimate over the same items and pair lists (gather_bre.hip, evaluate_primal_kernel)
      launch_evaluate_primal(a, h->beamsPerWave, h->bs->items.p, h->bs->itemOff.p, h->bs->queueCtl.p, h->bs->queueCtl.p + 2,
                             h->bs->pairs.p, h->bs->pairCnt.p, std::max<uint32_t>(1u, std::min<uint32_t>(nItemsQ, h->ncu * 16u)), es);
    else
      launch_evaluate_bre(a, h->beamsPerWave, fullVisQ, h->bs->items.p, h->bs->itemOff.p, h->bs->queueCtl.p,
                          h->bs->queueCtl.p + 2, h->bs->pairs.p, h->bs->pairCnt.p, h->persistentEval ? nwEval : nItemsQ, h->persistentEval,
                          es, units, h->bs->queueCtl.p + 5, unitCap);
    HIP_TRY(h, hipEventRecord(evEval->second, es));
    // the shifts and pairs the evaluation could not decide in fp32: their records and rays into the handle's list, where they
    // wait for the exact pass
    if (!primal) {
      if (h->exactDoneValid && es != h->stream) HIP_TRY(h, hipStreamWaitEvent(es, h->exactDone, 0));
      launch_capture_notes(a, es);
    }
    HIP_TRY(h, hipEventRecord(h->bs->lastUse, es));
    h->bs->lastUseValid = true;
    if (!primal) {
      rq = exactAfterGather(h);  // (a pass it starts waits for this evaluation too: after the event above)
      if (rq != GVPM_OK) return rq;
    }
    return GVPM_OK;
  };
  // (a second pass only when the planner met a ray outside the bundle the grid was keyed for: rebuilt in 3D)
  for (int attempt = 0;; ++attempt) {
  rebuilt = false;
  ChainPrep cp;
  if (h->photonsDirty || h->beamsDirty || r != h->bs->builtRadius) {
    // the other set; wait until the kernels that last read it are done
    h->setIdx = (h->setIdx + 1) % (h->pipeline && h->travStream ? 3 : 2);
    h->bs = &h->sets[h->setIdx];
    if (h->bs->used) HIP_TRY(h, hipStreamWaitEvent(h->bstream, h->bs->lastUse, 0));
    HIP_TRY(h, hipEventRecord(evBuild->first, h->bstream));
    lap("waitevent");
    {
      // the beam sort's key space rides behind the cells in the chain's counter array: known before the grid is sized
      int tileShift;
      uint32_t ntilesAll;
      beamTiling(h, h->beamsPerWave, cp.tw, cp.th, ntilesAll, tileShift);
      tileShift -= 3;  // (the chain's keys carry no edge bits: the sets of one pixel keep no order among themselves)
      const uint64_t nkeys = (uint64_t)ntilesAll << tileShift;
      cp.on = h->buildChain && h->nph > 0 && h->nsets > 0 && nkeys <= 0x3FFFFFF0ull;
      cp.nkeys = (uint32_t)nkeys;
      cp.tileShift = (uint32_t)tileShift;
    }
    rc = buildGrid(h, r, true, force3D, nullptr, &cp);
    lap("buildGrid");
    if (rc == GVPM_OK) rc = sortBeams(h, 0, &cp);
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
  // slab thickness along y / z: 8 layers with the 1.5-radius cells of maps up to 2 M photons (round 3, with the traversal's
  // cylinder filter: -4 % on the C2 step, three alternating runs on one box), 6 with the one-radius cells above (a rank's
  // share of C4: 2.9 ms against 3.1)
  if (!a.cfg.reserved[1] && h->cellScale <= 0.f && h->nph <= 2000000u) a.cfg.reserved[1] = 8;
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
  a.bundleFlag = h->bs->queueCtl.p + 4;
  // the planner's bound on (photon, beam) pairs sizes the pair buffer (grow only) ...
  // ... read back in the step's one host sync, with the photon bounds and the near-list overflow count
  if (!h->pinB6) {
    HIP_TRY(h, hipHostMalloc((void **)&h->pinB6, 256, hipHostMallocMapped));
    memset(h->pinB6, 0, 256);
    h->pinCtl = reinterpret_cast<uint32_t *>(h->pinB6 + 8);
  }
  if (cp.on && rebuilt) {
    // the whole build -- cells, beam sort, summed-volume table, planner beside the photon scatter -- in six launches
    if (!h->chainCtl.p) {
      HIP_TRY(h, h->chainCtl.ensure(192));
      HIP_TRY(h, hipMemsetAsync(h->chainCtl.p, 0, h->chainCtl.cap * sizeof(uint32_t), h->bstream));
    }
    ChainArgs c{};
    c.pos = h->rawDev.pos;
    c.n = h->nph;
    c.g = h->bs->grid;
    c.keys = h->bs->keysA.p;
    c.rank = h->bs->valsA.p;
    c.counts = h->bs->cellCount.p;
    c.starts = h->bs->cellStart.p;
    c.beamOff = h->bs->grid.ncells + 1u;
    c.scanLen = c.beamOff + cp.nkeys + 1u;
    c.sub = cp.sub;
    HIP_TRY(h, h->bs->chainBuckets.ensure(768));
    c.buckets = h->bs->chainBuckets.p;
    c.out6 = h->bs->bounds6.p;
    c.hostB6 = h->pinB6;
    c.rays = h->raysDev;
    c.nsets = h->nsets;
    c.width = h->cfg.width;
    c.tw = cp.tw;
    c.th = cp.th;
    c.tileShift = cp.tileShift;
    c.ntiles = h->bs->ntiles;
    c.bKeys = h->bs->bKeysA.p;
    c.bRank = h->bs->bValsA.p;
    c.setPerm = h->bs->setPerm.p;
    c.tileStart = h->bs->tileStart.p;
    c.blockSum = reinterpret_cast<uint32_t *>(h->bs->sortTmp.d);
    c.ctl = h->chainCtl.p;
    c.queueCtl = h->bs->queueCtl.p;
    c.overflowCtr = h->bs->overflowCtr.p;
    c.nearExt = h->bs->nearExt.p;
    c.sat = h->bs->sat.p;
    h->boundsPending = h->haveCachedBounds;  // (the chain's bounds land in pinB6 with the counters)
    // An OPTIMISTIC step: the pair buffer and the unit lists as the last steps left them (the radius shrinks: what held the
    // last step holds this one), the traversal's grid from the last item count, the evaluation kernel by the last build's
    // near lists.  The build's last block checks all of it against what the planner found.
    const bool optimistic = h->optimistic && h->pipeline && attempt == 0 && !primal && h->persistentEval && h->lastItems > 0 &&
                            h->bs->pairs.cap >= 128u && h->bs->pairCnt.cap >= (size_t)itemCap * h->beamsPerWave &&
                            (!h->evalUnits || h->bs->units.cap >= 2u);
    const bool fullVisOpt = !h->cfg.visibility_as_written || prevOverflow;
    // (tests: GVPM_OPTIMISTIC_REFUSE=n makes the guard refuse every n-th optimistic step -- a pair buffer of zero blocks)
    const bool refuse = optimistic && h->optRefuseEvery > 0 && (++h->optSteps % h->optRefuseEvery) == 0;
    launch_build_chain(c, a, h->rawDev, h->beamsPerWave, h->planTarget, h->bs->items.p, h->bs->itemOff.p, itemCap, cp.dmax, h->nearGrid,
                       (uint32_t)std::min<size_t>(h->bs->nearExt.cap, 0xFFFFFF00u), cp.wantOrig ? h->bs->origIdx.p : nullptr, h->pinCtl,
                       !h->bs->bucketsInit, h->bstream,
                       optimistic ? (refuse ? 0u : (uint32_t)std::min<size_t>((h->bs->pairs.cap - 64u) / 64u, 0xFFFFFFF0u)) : 0xFFFFFFFFu,
                       optimistic && h->evalUnits ? (uint32_t)std::min<size_t>(h->bs->units.cap / 2u, 0x3FFFFFFFu) : 0u, eval_unit_pairs(),
                       fullVisOpt);
    h->bs->bucketsInit = true;
    chainRan = true;
    if (optimistic) {
      HIP_TRY(h, hipEventRecord(evBuild->second, h->bstream));
      const uint32_t guess = std::min<uint32_t>(itemCap, h->lastItems + h->lastItems / 8u + 256u);
      rc = queueGather(0u, guess, true, fullVisOpt);
      if (rc != GVPM_OK) {
        h->bstream = h->stream;
        return rc;
      }
      queued = true;
    }
  } else {
    HIP_TRY(h, hipMemsetAsync(h->bs->queueCtl.p, 0, 8 * sizeof(uint32_t), h->bstream));
    launch_plan_bre(a, h->beamsPerWave, h->bs->ntiles, h->planTarget, h->bs->items.p, h->bs->queueCtl.p, h->bs->itemOff.p,
                    h->bs->queueCtl.p + 3, itemCap, h->bstream);
    launch_export_u32(h->bs->queueCtl.p + 3, rebuilt ? h->bs->overflowCtr.p : nullptr, rebuilt ? h->bs->nearExt.p : nullptr,
                      h->bs->queueCtl.p, h->bs->queueCtl.p + 4, h->pinCtl, h->bstream);
  }
  if (!queued) HIP_TRY(h, hipEventRecord(evBuild->second, h->bstream));
  lap("plan");
  HIP_TRY(h, hipStreamSynchronize(h->bstream));
  lap("syncB");
  blocks = h->pinCtl[0];
  nItems = h->pinCtl[3];
  if (queued && h->pinCtl[5] != 0u) {
    // the guess was wrong: both kernels have returned at once (or will); wait for them, then queue them again below, sized
    if (getenv("GVPM_TRACE_PLAN")) fprintf(stderr, "[plan] optimistic step refused by the build (status %u): queued again\n", h->pinCtl[5]);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->pipeline && h->travStream) HIP_TRY(h, hipStreamSynchronize(h->streamC));
    HIP_TRY(h, hipMemsetAsync(h->bs->queueCtl.p + 1, 0, 2 * sizeof(uint32_t), h->bstream));
    HIP_TRY(h, hipMemsetAsync(h->bs->queueCtl.p + 5, 0, 3 * sizeof(uint32_t), h->bstream));
    HIP_TRY(h, hipStreamSynchronize(h->bstream));
    queued = false;
    h->optRefused++;
  }
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
  if (rebuilt && chainRan) {
    // the camera beams' bounds of this step clip the next step's grid
    bool ok = true;
    for (int c = 0; c < 6; ++c) ok = ok && std::isfinite(h->pinB6[16 + c]);
    h->haveBeamBounds = ok;
    if (ok) memcpy(h->beamB6, h->pinB6 + 16, sizeof(h->beamB6));
  }
  h->lastItems = nItems;
  h->bstream = h->stream;
  if (!queued) {
    rc = queueGather(blocks, nItems, false, needFullVis(h));
    if (rc != GVPM_OK) return rc;
  }
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
  const int nblocks = 1024;  // (four workgroups per CU: 256 left this bandwidth-bound pass latency-bound, 58 us for 24 MB)
  HIP_TRY(h, h->bs->boundsPartial.ensure(nblocks * 6));
  HIP_TRY(h, h->bs->bounds6.ensure(32));
  // (what the host reads back in this build -- bounds, sub-beam count -- and after the traversal is written by the kernels
  // into pinned memory, as in the G-BRE build: a D2H copy is a runtime kernel of its own with ~20 us of latency around it)
  if (!h->pinBeams) HIP_TRY(h, hipHostMalloc((void **)&h->pinBeams, 256, hipHostMallocMapped));
  uint32_t *pinU = reinterpret_cast<uint32_t *>(h->pinBeams + 16);
  launch_bounds(h->rawDev.pos, n, h->bs->boundsPartial.p, nblocks, h->bs->bounds6.p, h->pinBeams, h->stream);
  launch_bounds(h->rawDev.parent_pos, n, h->bs->boundsPartial.p, nblocks, h->bs->bounds6.p + 6, h->pinBeams + 8, h->stream);
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  float b12[12];
  memcpy(b12, h->pinBeams, 6 * sizeof(float));
  memcpy(b12 + 6, h->pinBeams + 8, 6 * sizeof(float));
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
  // (end of round 4, with the build's kernels cheaper per sub-beam and the traversal's parity partition: 2.25 r -- GVPM_CELL_SCALE
  // 2 / 2.5 / 3 / 3.5 / 4 at C3: 23.19 / 22.80 / 22.72 / 23.15 / 24.0 ms per step)
  float cell = fmaxf(0.75f * (h->cellScale > 0.f ? h->cellScale : 3.0f) * r, ext / 256.f);
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
  launch_export_u32(h->subOffsets.p + (n - 1), h->subCounts.p + (n - 1), h->beamCtl.p, nullptr, nullptr, pinU, h->stream);
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  const uint32_t lastOff = pinU[0], lastCnt = pinU[1], maxBits = pinU[2];
  const uint64_t S = (uint64_t)lastOff + lastCnt;
  if (S > 0x7FFFFFF0ull) return fail(h, GVPM_ERR_INVALID_ARG, "too many sub-beams");
  h->nsub = (uint32_t)S;
  memcpy(&h->maxSubLen, &maxBits, 4);
  HIP_TRY(h, h->bs->keysA.ensure(S));
  HIP_TRY(h, h->bs->keysB.ensure(S));
  HIP_TRY(h, h->bs->valsA.ensure(S));
  HIP_TRY(h, h->bs->valsB.ensure(S));
  HIP_TRY(h, h->bs->hot.ensure(2 * S));
  HIP_TRY(h, h->subFlags.ensure(S + 1));
  HIP_TRY(h, h->bs->cellStart.ensure((size_t)g.ncells + 2));
  launch_beam_expand(h->rawDev.pos, h->rawDev.parent_pos, n, h->subCounts.p, h->subOffsets.p, g, h->bs->keysA.p,
                     h->bs->valsA.p, h->stream);
  HIP_TRY(h, sortPairsU32(h->bs->sortTmp, h->bs->keysA.p, h->bs->keysB.p, h->bs->valsA.p, h->bs->valsB.p, h->nsub,
                          ilog2ceil(g.ncells + 1), h->stream));
  HIP_TRY(h, h->beamAux.ensure(2 * (size_t)n + 2));
  launch_beam_cold(h->rawDev, h->endNDev, n, h->cfg, h->subCounts.p, h->bs->cold.p, h->beamAux.p, h->stream);
  launch_sub_hot(h->bs->valsB.p, h->nsub, h->beamAux.p, h->bs->hot.p, h->subFlags.p, h->stream);
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
static int gatherBeams(gvpm_context *h, int it, uint64_t nb_paths, bool primal = false) {
  if (!h->haveBeamsMap) return fail(h, GVPM_ERR_STATE, "G-Beams gather needs gvpm_upload_beams");
  h->vpmOrderN = 0;  // (the block-sort buffers below also hold G-VPM's batch order)
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
  h->iterClean = false;  // (this gather leaves `iter` written)
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
  a.cfg.reserved[5] = primal ? 1 : 0;  // the evaluation stops after the kernel record's base term
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
      // (work items of 6144 staged sub-beams: 4096 / 6144 / 8192 / 12288 at C3 with cells of 2.25 r: 22.73 / 22.45 / 22.43 / 22.51 ms)
      launch_plan_bre(a, h->beamsPerWave, h->bs->ntiles, h->planTargetSet ? h->planTarget : 6144u, h->bs->items.p, h->bs->queueCtl.p, nullptr, nullptr,
                      itemCap, h->stream);
      planned = true;
    }
    const uint32_t cap = (uint32_t)std::min<size_t>(h->beamPairs.cap, 0xFFFFFFC0u);
    const size_t nblkCap = cap / 64u + 1u;
    for (DevBuf<uint32_t> *b : {&h->blockKeyA, &h->blockKeyB, &h->blockValA, &h->blockValB}) HIP_TRY(h, b->ensure(nblkCap));
    launch_traverse_beams(a, h->subFlags.p, h->beamsPerWave, h->bs->items.p, h->bs->queueCtl.p, itemCap, h->bs->queueCtl.p + 1,
                          h->beamPairs.p, h->bs->queueCtl.p + 2, cap, h->blockKeyA.p, h->blockValA.p, h->nwavesTrav, h->stream);
    if (!h->pinBeams) HIP_TRY(h, hipHostMalloc((void **)&h->pinBeams, 256, hipHostMallocMapped));
    uint32_t *pinU = reinterpret_cast<uint32_t *>(h->pinBeams + 16) + 8;
    launch_export_u32(h->bs->queueCtl.p, h->bs->queueCtl.p + 1, h->bs->queueCtl.p + 2, nullptr, nullptr, pinU, h->stream);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const uint32_t ctl[3] = {pinU[0], pinU[1], pinU[2]};
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
  if (!primal && !h->beamsExact) {
    // the notes of the shifts the fp32 evaluation cannot decide (exact_beams_kernel, behind it): ~1.2e-3 of the PAIRS at C3, and
    // a block holds 64 pairs -- room for four notes a block, i.e. 6 % of the pairs, fifty times that rate (a
    // quarter of the block count was three times the rate, not twenty), and never less than the handle's default; S-laser's plate
    // had 2.8 % undecided segments before triHitFine.  Beyond it gvpm_get_stats still fails loudly (the counter keeps counting).
    const size_t need = std::max<size_t>(h->exOvfCap, (size_t)nBlocks * 4u);
    if (need > 0xFFFFFF00ull) return fail(h, GVPM_ERR_INVALID_ARG, "too many pair blocks for the note list");
    HIP_TRY(h, h->exOvf.reserveExact(need));
    a.exOvf = h->exOvf.p;
    a.exOvfCap = (uint32_t)need;
  }
  if (!primal && !h->beamsExact && h->reqCap > 0 && h->cfg.use_manifold) {
    // manifold-typed shifts (shiftBeamME) are recorded for the host instead of failing; the answered terms go straight into
    // the running mean: this iteration's buffer is folded right below, with weight 1 / (nb_paths * it)
    HIP_TRY(h, h->reqHost.ensure(h->reqCap));
    HIP_TRY(h, h->reqCtx.ensure(5 * h->reqCap));
    HIP_TRY(h, h->reqCount.ensure(2));
    HIP_TRY(h, hipMemsetAsync(h->reqCount.p, 0, 8, h->stream));
    a.reqHost = h->reqHost.p;
    a.reqCtx = h->reqCtx.p;
    a.reqCount = h->reqCount.p;
    a.reqCap = (uint32_t)h->reqCap;
    h->reqArgs = a;
    h->reqArgs.iter = h->accum.p;
    h->reqArgs.iterScale = (float)(1.0 / ((double)nb_paths * (double)it));
    h->reqOutstanding = true;
    h->reqBeams = true;
  }
  if (h->beamsSplit && !primal && !h->beamsExact && !a.reqHost && h->beamsPerWave == 16 && nBlocks) {
    // (experiment: phase 1 and phase 2 as two kernels, the reconnection entries through HBM -- gather_beams.hip)
    const size_t nEnt = (size_t)nBlocks * 256u;
    if (nEnt > 0xFFFFFF00ull) return fail(h, GVPM_ERR_INVALID_ARG, "too many blocks for the split evaluation");
    HIP_TRY(h, h->splitId.ensure(nEnt));
    HIP_TRY(h, h->splitMeta.ensure(nEnt));
    HIP_TRY(h, h->splitK.ensure(nEnt));
    HIP_TRY(h, h->splitU.ensure(nEnt));
    HIP_TRY(h, h->splitBlkCnt.ensure((size_t)nBlocks + 1));
    HIP_TRY(h, h->splitRuns.ensure((size_t)nBlocks / 16u + (size_t)h->ncu * 12u + 64u));
    HIP_TRY(h, h->splitCtl.ensure(4));
    HIP_TRY(h, hipMemsetAsync(h->splitCtl.p, 0, 16, h->stream));
    launch_evaluate_beams_split(a, h->splitId.p, h->splitMeta.p, h->splitK.p, h->splitU.p, h->splitBlkCnt.p, h->splitRuns.p,
                                h->splitCtl.p, h->beamPairs.p, h->blockKeyB.p, h->blockValB.p, nBlocks, h->bs->queueCtl.p + 3,
                                h->ncu, h->stream);
  } else
  launch_evaluate_beams(a, h->beamsPerWave, h->beamsExact && !primal, h->beamPairs.p, h->blockKeyB.p, h->blockValB.p, nBlocks,
                        h->bs->queueCtl.p + 3, h->nwaves, h->stream);
  HIP_TRY(h, hipEventRecord(ev->second, h->stream));
  // the shifts the fp32 evaluation left undecided (gather_beams.hip, exact_beams_kernel): in fp64, behind it, every gather
  if (!primal && !h->beamsExact) {
    launch_exact_beams(a, h->exTotals.p, h->stream);
    static const bool trace = getenv("GVPM_TRACE_EXACT") != nullptr;
    if (trace) {
      unsigned long long v[20];
      HIP_TRY(h, hipMemcpyAsync(v, h->exTotals.p, sizeof(v), hipMemcpyDeviceToHost, h->stream));
      HIP_TRY(h, hipStreamSynchronize(h->stream));
      fprintf(stderr, "[exact beams] so far %llu evaluated, %llu lost, longest list %llu; by cause 0..9:", v[0], v[1], v[2]);
      for (int k = 0; k < 10; ++k) fprintf(stderr, " %llu", v[4 + k]);
      fprintf(stderr, "\n");
    }
  }
  if (primal) {
    // BeamRadianceQuery takes the camera ray's transmittance over [Epsilon, w] (pm/beams.h:57-61,189-193) where the gradient
    // pass's kernel record takes it over [0, w] (shift_volume_beams.h:167,262): one factor exp(sigma_t Epsilon) on every term
    // (sigma_t is equal across the channels, gvpm_upload_medium)
    const float f = (float)std::exp((double)h->medium.sigma_t[0] * (double)h->cfg.epsilon);
    launch_scale(h->iter.p, h->iter.p, h->npix * 27, f, h->stream);
  }
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
  h->iterClean = false;
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
static int gatherVPM(gvpm_context *h, int it, uint64_t nb_paths, bool primal = false) {
  (void)it;
  if (!h->haveSamples) return fail(h, GVPM_ERR_STATE, "G-VPM gather needs gvpm_upload_vpm_samples");
  if (h->cfg.nb_camera_samples <= 0) return fail(h, GVPM_ERR_INVALID_ARG, "nb_camera_samples must be positive");
  // grid cell = the largest per-pixel radius R * 0.01 * max(scaleVol) -- or anything above it.
  // Round 6: the step no longer waits for the one before it.  (1) The scale: a pixel's scale only shrinks (ratio <= 1 in the
  // SPPM update), so the largest scale of ANY earlier iteration bounds this one's: the host keeps such a bound -- the initial
  // scale at gvpm_reset, then whatever the iterations' last kernels have exported to pinned memory by now, one or two steps
  // stale -- and the cells are that much larger than they need to be.  (2) The photons' bounds and the grid on the BUILD stream,
  // into the other build set, while the gather of the step before still runs on the gather stream: the host waits for the
  // bounds of THIS upload only (a 10 us reduction that depends on nothing else), then for the near lists' overflow word behind
  // the build.  Until round 6 both waits stood behind the previous gather, the GPU idle for ~25 us of a 0.45 ms step, and the
  // build's nine small launches (~70 us) ran between two gathers instead of beside one.
  if (!h->pinB6) {
    HIP_TRY(h, hipHostMalloc((void **)&h->pinB6, 256, hipHostMallocMapped));
    memset(h->pinB6, 0, 256);
    h->pinCtl = reinterpret_cast<uint32_t *>(h->pinB6 + 8);
  }
  {
    const uint32_t bits = reinterpret_cast<volatile uint32_t *>(h->pinCtl)[32];
    float e;
    memcpy(&e, &bits, 4);
    if (bits != 0u && e > 0.f && e < h->vpmScaleBound) h->vpmScaleBound = e;
  }
  const float maxScale = h->vpmScaleBound;
  if (!(maxScale > 0.f)) return fail(h, GVPM_ERR_STATE, "G-VPM gather: no scale bound (gvpm_reset sets it)");
  const float rmax = (h->cfg.bsphere_radius * 0.01f) * maxScale;
  const bool pipe = h->pipeline && h->vpmPipeline && h->streamB;
  h->bstream = pipe ? h->streamB : h->stream;
  const bool rebuild = h->photonsDirty || rmax != h->bs->builtRadius;
  if (rebuild) {
    if (pipe) {
      // the other set; wait (on the build stream) until the gather that last read it is done
      h->setIdx = (h->setIdx + 1) % 2;
      h->bs = &h->sets[h->setIdx];
      if (h->bs->lastUseValid) HIP_TRY(h, hipStreamWaitEvent(h->bstream, h->bs->lastUse, 0));
    }
    const bool wantBounds = h->nph > 0;
    if (wantBounds) {
      HIP_TRY(h, h->bs->boundsPartial.ensure(1024 * 6));
      HIP_TRY(h, h->bs->bounds6.ensure(32));
      launch_bounds(h->rawDev.pos, h->nph, h->bs->boundsPartial.p, 1024, h->bs->bounds6.p, h->pinB6, h->bstream, nullptr, nullptr);
      HIP_TRY(h, hipStreamSynchronize(h->bstream));
    }
    int rc = buildGrid(h, rmax, false, false, wantBounds ? h->pinB6 : nullptr);
    if (rc != GVPM_OK) {
      h->bstream = h->stream;
      return rc;
    }
    h->photonsDirty = false;
    h->bs->builtRadius = rmax;
    if (pipe) {
      HIP_TRY(h, hipEventRecord(h->bs->traversed, h->bstream));  // (this set's spare event: the build is done)
      HIP_TRY(h, hipStreamWaitEvent(h->stream, h->bs->traversed, 0));
    }
  }
  h->bstream = h->stream;
  // (the previous G-VPM gather zeroed `iter` and `mvol` as it folded them, and zeroes the largest-scale word before its update:
  // accumulate_kernel / vpm_update_kernel; anything else in between -- another technique, a failed gather -- and they are cleared here)
  if (!h->iterClean) {
    HIP_TRY(h, hipMemsetAsync(h->iter.p, 0, h->npix * 27 * sizeof(float), h->stream));
    HIP_TRY(h, hipMemsetAsync(h->mvol.p, 0, h->npix * sizeof(float), h->stream));
  }
  h->iterClean = false;
  {
    const int rcx = exactPrepare(h);
    if (rcx != GVPM_OK) return rcx;
  }
  GatherArgs a;
  fillArgs(h, a, rmax);
  std::pair<hipEvent_t, hipEvent_t> *ev;
  int rc = nextEvents(h, &ev);
  if (rc != GVPM_OK) return rc;
  if (!primal && h->reqCap > 0 && h->cfg.use_manifold && h->bs->origIdx.p) {
    // manifold-typed shifts are recorded for the host (gvpm_download_shift_requests) instead of failing; the answered
    // terms are added straight to the accumulators (plain sums: this iteration's buffer is folded right below)
    HIP_TRY(h, h->reqHost.ensure(h->reqCap));
    HIP_TRY(h, h->reqCtx.ensure(4 * h->reqCap));
    HIP_TRY(h, h->reqCount.ensure(2));
    HIP_TRY(h, hipMemsetAsync(h->reqCount.p, 0, 8, h->stream));
    a.reqHost = h->reqHost.p;
    a.reqCtx = h->reqCtx.p;
    a.reqCount = h->reqCount.p;
    a.reqCap = (uint32_t)h->reqCap;
    a.origIdx = h->bs->origIdx.p;
    h->reqArgs = a;
    h->reqArgs.iter = h->accum.p;
    h->reqArgs.iterScale = 1.f;
    h->reqOutstanding = true;
    h->reqBeams = false;
  }
  // Heaviest batches first.  A wave's time follows its candidate count (correlation 0.99, scripts/vpm_timing.py) and the
  // counts are heavy-tailed (C1: median 159, maximum 6 500 -- the pixels that look at the light): in sample order the last
  // heavy wave started when the others were done, and the kernel ran 160 of its 560 us on a handful of waves.  The batches
  // hold the same pixels every iteration, so the last launch's counts order this one.
  const uint32_t nBatches = (h->nsamples + 63u) / 64u;
  if (h->blockValB.cap < (size_t)nBatches + 1) h->vpmOrderN = 0;  // (a regrown buffer has lost the order)
  for (DevBuf<uint32_t> *b : {&h->blockKeyA, &h->blockKeyB, &h->blockValA, &h->blockValB}) HIP_TRY(h, b->ensure(nBatches + 1));
  a.vpmCostKey = h->blockKeyA.p;
  a.vpmCostVal = h->blockValA.p;
  // (a launch of another size: the permutation's slots, then the batches it does not know, in order; see the kernel)
  const bool haveOrder = h->vpmOrderN != 0 && !h->vpmNoOrder && h->vpmOrderN <= 2u * nBatches && nBatches <= 2u * h->vpmOrderN;
  a.vpmOrder = haveOrder ? h->blockValB.p : nullptr;
  a.vpmOrderN = haveOrder ? h->vpmOrderN : 0u;
  HIP_TRY(h, hipEventRecord(ev->first, h->stream));
  if (h->vpmSplit && !primal) {
    // walk -> (redo of the heavy batches on a second stream) + evaluation (gather_vpm.hip)
    VpmSplit sp;
    // the pool: four chunks a batch (four pairs a sample) unless GVPM_VPM_POOL says otherwise; a step that needs more sends the
    // batches that find it exhausted through the fused code
    const uint32_t shardChunks = std::max<uint32_t>(4u, (uint32_t)(((uint64_t)h->vpmPoolPerBatch * nBatches + VPM_SHARDS - 1) / VPM_SHARDS));
    HIP_TRY(h, h->vpmPairs.ensure((size_t)shardChunks * VPM_SHARDS * 64));
    HIP_TRY(h, h->vpmChunkMeta.ensure((size_t)shardChunks * VPM_SHARDS));
    HIP_TRY(h, h->vpmStatus.ensure(nBatches));
    HIP_TRY(h, h->vpmState.ensure(h->nsamples));
    HIP_TRY(h, h->vpmRedo.ensure(nBatches));
    HIP_TRY(h, h->vpmCtl.ensure(VPM_CTL_REDO + 32));
    HIP_TRY(h, hipMemsetAsync(h->vpmCtl.p, 0, (VPM_CTL_REDO + 32) * sizeof(uint32_t), h->stream));
    sp.pairs = h->vpmPairs.p;
    sp.chunkMeta = h->vpmChunkMeta.p;
    sp.ctl = h->vpmCtl.p;
    sp.status = h->vpmStatus.p;
    sp.redo = h->vpmRedo.p;
    sp.state = h->vpmState.p;
    sp.shardChunks = shardChunks;
    sp.nBatches = nBatches;
    if (!h->vpmFound) {
      HIP_TRY(h, hipEventCreateWithFlags(&h->vpmFound, hipEventDisableTiming));
      HIP_TRY(h, hipEventCreateWithFlags(&h->vpmRedone, hipEventDisableTiming));
    }
    sp.zeroWord = h->maxScaleBits.p;  // (read through the host at the top; the iteration's last kernel reduces the new maximum into it)
    launch_vpm_find(a, sp, h->stream);
    const bool side = h->pipeline && h->streamA2;
    hipStream_t rs = side ? h->streamA2 : h->stream;
    if (side) {
      HIP_TRY(h, hipEventRecord(h->vpmFound, h->stream));
      HIP_TRY(h, hipStreamWaitEvent(rs, h->vpmFound, 0));
    }
    launch_vpm_redo(a, sp, needFullVis(h), h->vpmRedoWaves, rs);
    launch_vpm_eval(a, sp, needFullVis(h), std::max(1u, h->vpmEvalWaves / VPM_SHARDS), h->stream);
    if (side) {
      HIP_TRY(h, hipEventRecord(h->vpmRedone, rs));
      HIP_TRY(h, hipStreamWaitEvent(h->stream, h->vpmRedone, 0));
    }
  } else {
    HIP_TRY(h, hipMemsetAsync(h->maxScaleBits.p, 0, 4, h->stream));
    launch_gather_vpm(a, needFullVis(h), primal, h->stream);
  }
  HIP_TRY(h, hipEventRecord(ev->second, h->stream));
  // the shifts the kernel could not decide in fp32: into the handle's list (before the radii of this iteration are updated),
  // where they wait for the exact pass -- which adds to the plain sums whenever it runs
  if (!primal) {
    launch_capture_notes(a, h->stream, 32);  // (a hundred notes an iteration at C1: 256 workgroups spent 8 us on their own hand-off)
    rc = exactAfterGather(h);
    if (rc != GVPM_OK) return rc;
  }
  // (re-sorted every fourth launch: the heavy pixels stay where they are while the radii shrink)
  if (!h->vpmNoOrder && nBatches > 1024u && (!haveOrder || (h->vpmLaunches & 3u) == 0u)) {
    HIP_TRY(h, sortPairsU32(h->bs->sortTmp, h->blockKeyA.p, h->blockKeyB.p, h->blockValA.p, h->blockValB.p, nBatches, 20, h->stream));
    h->vpmOrderN = nBatches;
  }
  h->vpmLaunches++;
  HIP_TRY(h, hipEventRecord(h->bs->lastUse, h->stream));  // (the kernels above are the last readers of this build set)
  h->bs->lastUseValid = true;
  launch_vpm_finish(h->accum.p, h->iter.p, h->scaleVol.p, h->nVol.p, h->mvol.p, h->npix, h->cfg.alpha, h->maxScaleBits.p, h->stream);
  // the new largest scale to pinned memory: a later gather's bound (above)
  launch_export_u32(h->maxScaleBits.p, nullptr, nullptr, nullptr, nullptr, h->pinCtl + 32, h->stream);
  HIP_TRY(h, hipGetLastError());
  h->iterClean = true;
  h->totalEmitted += (double)nb_paths;  // m_totalEmittedVolume, gvpm.cpp:434
  return GVPM_OK;
}

// ---- manifold shifts through the host (include/gvpm_hip.h) --------------------------------------------------------------
static int requestCount(gvpm_context *h, uint32_t *n) {
  uint32_t c = 0;
  HIP_TRY(h, hipMemcpyAsync(&c, h->reqCount.p, 4, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  *n = (uint32_t)std::min<uint64_t>(c, h->reqCap);
  return GVPM_OK;
}
int flushHostShifts(gvpm_context *h) {
  if (!h->reqOutstanding) return GVPM_OK;
  h->reqOutstanding = false;
  uint32_t n = 0;
  const int rc = requestCount(h, &n);
  if (rc != GVPM_OK) return rc;
  if (h->reqBeams) launch_apply_host_shifts_beams(h->reqArgs, nullptr, n, h->stream);
  else launch_apply_host_shifts(h->reqArgs, nullptr, n, h->stream);
  HIP_TRY(h, hipGetLastError());
  return GVPM_OK;
}

extern "C" {

int gvpm_enable_host_shifts(gvpm_context *h, uint64_t capacity) {
  CHECK_H(h);
  if (capacity > 0x7FFFFFF0ull) return fail(h, GVPM_ERR_INVALID_ARG, "host-shift capacity too large");
  const int rc = flushHostShifts(h);
  if (rc != GVPM_OK) return rc;
  h->reqCap = capacity;
  h->photonsDirty = true;  // (the next build writes the upload index of every photon)
  return GVPM_OK;
}

int gvpm_download_shift_requests(gvpm_context *h, gvpm_shift_request *out, uint64_t cap, uint64_t *n) {
  CHECK_H(h);
  if (!n || (cap && !out)) return GVPM_ERR_INVALID_ARG;
  *n = 0;
  if (!h->reqOutstanding) return GVPM_OK;
  uint32_t c = 0;
  const int rc = requestCount(h, &c);
  if (rc != GVPM_OK) return rc;
  *n = c;
  const size_t m = (size_t)std::min<uint64_t>(c, cap);
  if (m) HIP_TRY(h, hipMemcpy(out, h->reqHost.p, m * sizeof(gvpm_shift_request), hipMemcpyDeviceToHost));
  return GVPM_OK;
}

int gvpm_upload_host_shifts(gvpm_context *h, const gvpm_host_shift *results, uint64_t n) {
  CHECK_H(h);
  if (!h->reqOutstanding) return n == 0 ? GVPM_OK : fail(h, GVPM_ERR_STATE, "no shift requests are waiting for results");
  if (n && !results) return fail(h, GVPM_ERR_INVALID_ARG, "null host shifts");
  uint32_t c = 0;
  const int rc = requestCount(h, &c);
  if (rc != GVPM_OK) return rc;
  if (n != c) return fail(h, GVPM_ERR_INVALID_ARG, "gvpm_upload_host_shifts: one result per recorded request");
  h->reqOutstanding = false;
  if (!c) return GVPM_OK;
  HIP_TRY(h, h->reqResults.ensure(c));
  HIP_TRY(h, hipMemcpyAsync(h->reqResults.p, results, (size_t)c * sizeof(gvpm_host_shift), hipMemcpyHostToDevice, h->stream));
  if (h->reqBeams) launch_apply_host_shifts_beams(h->reqArgs, h->reqResults.p, c, h->stream);
  else launch_apply_host_shifts(h->reqArgs, h->reqResults.p, c, h->stream);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipStreamSynchronize(h->stream));  // (the caller may reuse `results`)
  return GVPM_OK;
}

static int gatherEntry(gvpm_context *h, int it, uint64_t nb_paths, bool primal);
int gvpm_gather(gvpm_context *h, int it, uint64_t nb_paths) {
  CHECK_H(h);
  return gatherEntry(h, it, nb_paths, false);
}
int gvpm_gather_primal(gvpm_context *h, int it, uint64_t nb_paths) {
  CHECK_H(h);
  const gvpm_params &c = h->cfg;
  if (c.vol_technique == GVPM_VOL_PLANE0D || c.vol_technique == GVPM_BEAM_BEAM_3D_NAIVE || c.vol_technique == GVPM_BEAM_BEAM_3D_EGSR)
    return fail(h, GVPM_ERR_UNSUPPORTED, "gvpm_gather_primal: built for BRE 2D / 3D, DISTANCE, BEAM_BEAM_1D and BEAM_BEAM_3D_OPTIMIZED");
  // the primal pass has none of the gradient pass's filters (sppm.cpp:882-1000): the handle must not carry them
  if (c.path_set || c.debug_shift != GVPM_SHIFT_ALL || c.min_depth != 0 || c.bsdf_interaction_mode != GVPM_BSDF_ALL ||
      !((c.lighting_interaction_mode & GVPM_SURF2MEDIA) && (c.lighting_interaction_mode & GVPM_MEDIA2MEDIA)))
    return fail(h, GVPM_ERR_UNSUPPORTED, "gvpm_gather_primal: path_set, debug_shift, min_depth and the interaction modes must be neutral");
  return gatherEntry(h, it, nb_paths, true);
}
static int gatherEntry(gvpm_context *h, int it, uint64_t nb_paths, bool primal) {
  if (it < 1 || nb_paths == 0) return fail(h, GVPM_ERR_INVALID_ARG, "it must be >= 1 and nb_paths > 0");
  if (!h->haveMedium || !h->havePhotons || !h->haveBeams)
    return fail(h, GVPM_ERR_STATE, "gather needs medium, photons and camera beams uploaded");
  h->useAll = false;
  h->bstream = h->stream;
  {
    const int rcf = flushHostShifts(h);
    if (rcf != GVPM_OK) return rcf;
  }
  // the streams that read this step's host-uploaded inputs wait for their copies (copy stream)
  // (packed records are decoded here, at the head of the chain that reads them: G-BRE builds on the build stream)
  const bool breTech = h->cfg.vol_technique == GVPM_VOL_BRE2D || h->cfg.vol_technique == GVPM_VOL_BRE3D;
  {
    // G-BRE and G-VPM keep plain sums: the exact pass may add its terms whenever it runs.  The beam / plane gathers fold
    // iter[] into running means on the gather stream: entries that wait are taken first.
    const bool sums = (breTech || h->cfg.vol_technique == GVPM_DISTANCE) && !primal;
    if (!sums) {
      const int rcj = gvpm_join_exact(h);
      if (rcj != GVPM_OK) return rcj;
    }
  }
  hipStream_t us = breTech && h->pipeline ? h->streamB : h->stream, other = us == h->stream ? h->streamB : h->stream;
  if (h->phWait) {
    gvpm_context::PhotonSlot &ps = h->phSlot[h->phCur];
    HIP_TRY(h, hipStreamWaitEvent(h->stream, ps.copied, 0));
    HIP_TRY(h, hipStreamWaitEvent(h->streamB, ps.copied, 0));
    if (ps.needUnpack) {
      // (a packed record names its parent's material by index: without a table every parent would decode as black)
      if (h->nmaterials == 0)
        return fail(h, GVPM_ERR_STATE, "packed photons were uploaded but no material table (gvpm_upload_materials)");
      if (ps.linked) launch_unpack_linked(ps.packed.p, (uint32_t)ps.dev.n, h->materials.p, h->nmaterials, ps.dev, h->stats.p + 6, us);
      else launch_unpack_photons(ps.packed.p, (uint32_t)ps.dev.n, h->materials.p, h->nmaterials, ps.dev, h->stats.p + 6, us);
      HIP_TRY(h, hipGetLastError());
      if (!ps.unpacked) HIP_TRY(h, hipEventCreateWithFlags(&ps.unpacked, hipEventDisableTiming));
      HIP_TRY(h, hipEventRecord(ps.unpacked, us));
      HIP_TRY(h, hipStreamWaitEvent(other, ps.unpacked, 0));
      ps.needUnpack = false;
    }
    h->phWait = false;
  }
  if (h->rayWait) {
    gvpm_context::RaySlot &rs = h->raySlot[h->rayCur];
    HIP_TRY(h, hipStreamWaitEvent(h->stream, rs.copied, 0));
    HIP_TRY(h, hipStreamWaitEvent(h->streamB, rs.copied, 0));
    if (rs.needUnpack) {
      if (rs.ncompact) {
        if (!h->haveSensor)
          return fail(h, GVPM_ERR_STATE, "compact beam sets were uploaded but no sensor (gvpm_upload_sensor)");
        launch_unpack_compact_rays(h->sensor, rs.compact.p, rs.ncompact, rs.rays.p, us);
      }
      launch_unpack_rays(rs.packed.p, rs.nsets - rs.ncompact, rs.rays.p + (size_t)rs.ncompact * 5, us);
      HIP_TRY(h, hipGetLastError());
      if (!rs.unpacked) HIP_TRY(h, hipEventCreateWithFlags(&rs.unpacked, hipEventDisableTiming));
      HIP_TRY(h, hipEventRecord(rs.unpacked, us));
      HIP_TRY(h, hipStreamWaitEvent(other, rs.unpacked, 0));
      rs.needUnpack = false;
    }
    h->rayWait = false;
  }
  int rc;
  switch (h->cfg.vol_technique) {
    case GVPM_VOL_BRE2D:
    case GVPM_VOL_BRE3D: rc = gatherBRE(h, it, nb_paths, primal); break;
    case GVPM_DISTANCE: rc = gatherVPM(h, it, nb_paths, primal); break;
    case GVPM_BEAM_BEAM_1D:
    case GVPM_BEAM_BEAM_3D_OPTIMIZED: rc = gatherBeams(h, it, nb_paths, primal); break;
    case GVPM_VOL_PLANE0D: rc = gatherPlanes(h, it, nb_paths); break;
    default: return fail(h, GVPM_ERR_UNSUPPORTED, "vol_technique not built in this library yet");
  }
  if (rc != GVPM_OK) return rc;
  // the kernels just queued on the gather stream are the last readers of this step's camera rays
  if (h->raysOwnedCur) {
    HIP_TRY(h, hipEventRecord(h->raySlot[h->rayCur].freed, (breTech && h->lastEvalStream) ? h->lastEvalStream : h->stream));
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
}  // extern "C"
