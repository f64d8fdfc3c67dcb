// Device-side data layout of libgvpm_hip.so (gfx950).  See DESIGN.md "HBM layout".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gvpm_hip.h"

namespace gvpm {

// Photon records after the grid build, sorted by cell (x fastest):
//   hot[i]      = {pos.xyz, bits}         16 B  -- streamed by the hit test
//   cold[i][k], k = 0..7                  one 128 B record -- read once per evaluation
//     0 {pos, bits} 1 {wi, parentPdf} 2 {flux, edgePdf} 3 {parentPos, parentRR} 4 {parentN, parentG}
//     5 {prefixW, nearOccluders} 6 {parentScat, -} 7 {parentWi, -}
// bits = GVPM_PF_* flags of the ABI with bit 7 = pathID & 1.
// (G-Beams keeps one 128-byte record per BEAM in the same buffer -- grid_build.hip, beam_cold_kernel -- and the sorted
// sub-beam centres in `hot`.)
#define GVPM_HOT_PARITY_BIT 7
// per-photon near-occluder lists (grid_build.hip, nearOccluders): 8-bit indices up to NARROW_MAX occluders, 16-bit up
// to WIDE_MAX (the top byte of word 0 must stay below the 0xFD / 0xFE marks), extension lists beyond
#define GVPM_NEAR_NARROW_MAX 253u
#define GVPM_NEAR_WIDE_MAX 64767u
#define GVPM_REC_QUADS 8

struct Grid {
  float org[3];    // world position of cell (0,0,0)'s lower corner
  float cell;      // cell edge length
  float invCell;
  int dim[3];
  uint32_t ncells;
  // mode 1 (G-BRE, round 3): the camera beams' lines pass through ONE point (a pinhole sensor's first medium edges), so a
  // photon can only be met by rays inside a small rectangle of the ray bundle's (u, v) = (d.U, d.V) / d.A plane.  The
  // cells are then (u cell, v cell, level): level l has cells of s0 * 2^l and holds the photons whose rectangle's half
  // extent is at most that; a tile reads, per level, the cells its rays' bounding rectangle reaches when dilated by one
  // cell size.  dim = {G, G, 2}, G a power of two, level l has (G >> l)^2 cells, the last level is one cell; the levels
  // are packed into the two layers as bundle_grid.h lays out.  Everything downstream of the cell key (counting sort, summed-volume table, planner items,
  // staging, tests) is the 3D grid's: a "slab step" of a tile is a level instead of a run of layers, and there are a
  // handful of them instead of dozens (bundle_grid.h).
  int mode, levels;
  float bo[3], ba[3], bu[3], bv[3];  // the point the rays' lines share; A (the bundle's mean direction), U, V orthonormal
  float lineTol;                     // how far from that point a ray's line may pass (fp32 rounding of the host's rays)
  float uMin, uMax, vMin, vMax;      // the rays' (u, v) range, padded
  float s0, invS0;                   // level-0 cell size
  float radius;                      // kernel radius the photons were binned for
};

// Occluders by cell of a coarse uniform grid over the scene (grid_build.hip: near_grid_*): tris[start[c] .. start[c + 1])
// are the triangles a point of cell c can have within `reach`.  start == nullptr: no grid (small scenes scan linearly).
struct NearGrid {
  const uint32_t *start = nullptr;
  const uint32_t *tris = nullptr;
  float org[3] = {0, 0, 0}, inv[3] = {0, 0, 0};  // cell index = floor((p - org) * inv)
  int dim[3] = {0, 0, 0};
};

// A beam's near-occluder list in the three spare words of its record (beam_near_kernel / beamShadowBlocked): a string of
// 96 bits cut into entries of `bits`, all ones = empty.  Small scenes get narrow indices and hence longer lists -- at C3
// (22 occluders) one beam in forty listed more than twelve and, tested against everything, kept its whole wave in a
// 22-trip loop in four drains of five.
//   occluders <= 31:  19 entries of 5 bits, overflow = bit 95
//   occluders <= 63:  15 entries of 6 bits, overflow = bit 95
//   otherwise:        12 entries of 8 bits, overflow = top byte of word 0 is 0xFE  (<= 253 occluders; beyond: all overflow)
#ifdef __HIPCC__
#define GVPM_DT_HD __host__ __device__
#else
#define GVPM_DT_HD
#endif
struct BeamNearFmt {
  uint32_t bits, cap, mask;
};
GVPM_DT_HD inline BeamNearFmt beamNearFmt(uint32_t ntri) {
  return ntri <= 31u ? BeamNearFmt{5u, 19u, 31u} : (ntri <= 63u ? BeamNearFmt{6u, 15u, 63u} : BeamNearFmt{8u, 12u, 255u});
}
GVPM_DT_HD inline bool beamNearOverflow(const BeamNearFmt &f, uint32_t w0, uint32_t w2) {
  return f.bits == 8u ? (w0 >> 24) == 0xFEu : (w2 >> 31) != 0u;
}
// entry k of the list (k < cap)
GVPM_DT_HD inline uint32_t beamNearEntry(const BeamNearFmt &f, uint32_t w0, uint32_t w1, uint32_t w2, uint32_t k) {
  const uint32_t off = k * f.bits, word = off >> 5, sh = off & 31u;
  const uint32_t lo = word == 0u ? w0 : (word == 1u ? w1 : w2), hi = word == 0u ? w1 : (word == 1u ? w2 : 0u);
  const unsigned long long v = ((unsigned long long)hi << 32) | lo;
  return (uint32_t)(v >> sh) & f.mask;
}

#ifdef __HIPCC__
// the grid cell of a coordinate (clamped: positions on the bounds' upper faces, rounding)
__device__ __forceinline__ int cellCoord(float p, float org, float inv, int dim) {
  int c = (int)floorf((p - org) * inv);
  return min(max(c, 0), dim - 1);
}
#endif

struct SortTemp {
  void *d = nullptr;
  size_t bytes = 0;
  uint32_t sortN = 0;   // the largest pair count whose temporary size has been asked for (sortPairsU32) ...
  size_t sortNeed = 0;  // ... and the answer
};

struct MediumDev {
  float sigmaS[3], sigmaT[3];
  float g, msw;
};

// The exact pass (exact_shift.hip).  A shift (or pair) the fp32 kernels cannot decide is NOTED as {set | sample, record,
// meta, -} -- meta = kind | shift << 8 | cause << 16: one atomic and one 16-byte store in the hot loops.  A small copy
// kernel behind the gather's kernels (capture_notes_kernel: 16 registers, one workgroup) turns the notes into
// self-contained 512-byte entries -- everything the evaluation reads of the gather: the record, the five rays, radius,
// output scale -- which outlive the gather's buffers; the pass itself (fp64, 128 registers a lane) runs when something
// reads the sums.  Measured at C2: the pass behind every evaluation waits for registers the other streams' persistent
// kernels hold, +5 % on the step; entries written where the shift is deferred, or through per-wave LDS lists, cost the
// evaluation's loops registers: +3 % and +17 %.
#define GVPM_EX_KIND_BRE 0u        /* one shift of a G-BRE pair */
#define GVPM_EX_KIND_BRE_PAIR 1u   /* a whole G-BRE pair whose HIT the fp32 bands could not decide */
#define GVPM_EX_KIND_VPM 2u
#define GVPM_EX_KIND_BEAMS 3u
struct ExEntry {
  uint32_t meta, pad0;       // kind | shift << 8 | cause << 16
  float outScale;            // what the fast kernel multiplies the shift's two sums by when it adds them (G-BRE: 1 / nb_paths)
  float radius;              // kernel radius of the pair (G-VPM: the pixel's own)
  float4 rec[GVPM_REC_QUADS];  // the photon's (G-Beams: the beam's) 128-byte record
  gvpm_camera_ray rays[5];   // the beam set
  float4 extra[3];           // per technique
};
static_assert(sizeof(ExEntry) == 512, "ExEntry is 512 bytes");

struct GatherArgs {
  // photons
  const float4 *hot;
  const float4 *cold;        // nph records of GVPM_REC_QUADS float4
  const uint32_t *cellStart; // ncells + 1
  const uint32_t *sat;       // summed-volume table of the cell counts (planner), (dimx+1)(dimy+1)(dimz+1)
  uint32_t nph;
  Grid grid;
  // camera beam sets (5 x 64 B each), visited through the tile permutation
  const gvpm_camera_ray *rays;
  const uint32_t *setPerm;   // beam sets ordered by image tile
  const uint32_t *tileStart; // ntiles + 1 offsets into setPerm
  uint32_t nsets;
  // scene
  const float4 *tri4;        // 3 float4 per triangle {v0,n.x} {e1,n.y} {e2,n.z}, BVH leaf order
  const float4 *bvh;         // 2 float4 per node (scene_bvh.h)
  const uint32_t *nearExt;   // extension lists of the per-photon near-occluder lists {count, index...}
  uint32_t ntri;
  float triAbs1;             // sum over the axes of the largest |coordinate| of an occluder vertex (planeSideMargin)
  MediumDev med;
  // config
  gvpm_params cfg;
  float radius;
  // G-Beams only: a.radius is the traversal radius (kernel radius + half a sub-beam),
  // cold holds one 128-byte record per beam, hot holds the sub-beam centres {xyz, beam | sub << 24}
  float kernelRadius;
  float subLen;              // target sub-beam length used by the build
  uint32_t nbeams;
  // G-BRE: the cell box of every slab step of every tile chunk, written by the planner (which needs them to count the
  // staged photons) and read by the traversal instead of being reduced over the tile's beams a second time:
  // planBoxes[chunk * planBoxStride + step], chunk = setBase / B + tile, {x0 | x1 << 10 | y0 << 20, y1 | z0 << 10 | z1 << 20},
  // x = 0xFFFFFFFF: no beam of the chunk reaches the slab.  Null: the traversal computes its boxes (G-Beams).
  uint2 *planBoxes;
  uint32_t planBoxStride;
  // manifold shifts through the host (gvpm_enable_host_shifts): requests, the device's own part of each, their number
  gvpm_shift_request *reqHost;
  float4 *reqCtx;       // 4 per request: {tr, pdfCam, pdfShiftPos, sensorMIS} {scale, bc} {shifted ray d, pixel} {eye, shift}
  uint32_t *reqCount;
  uint32_t reqCap;
  const uint32_t *origIdx;  // sorted photon -> index in the upload
  uint32_t *bundleFlag;  // set by the planner when a valid ray is outside the bundle the grid (mode 1) was built for
  const float2 *beamClear;   // per beam {cosA0, M1}: the free cone of its reconnections (grid_build.hip, beam_near_kernel)
  // glossy surface parents (gvpm_upload_bsdfs): 4 float4 per entry {kind, specular} {exponent | alpha, sampling weight,
  // distribution, sample_visible} {eta, k.x} {k.y, k.z, -, -}
  const float4 *bsdfs;
  uint32_t nbsdfs;
  // G-VPM only
  const gvpm_vpm_sample *samples;
  uint32_t nsamples;
  const float *scaleVol;     // per pixel GatherPoint::scaleVol (read)
  float *mvol;               // per pixel photons found this iteration (MVol, atomically added)
  // the order the waves take the 64-sample batches in: heaviest first, by the candidate counts of the LAST launch (the batches
  // hold the same pixels every iteration); null: in order.  vpmCostKey / vpmCostVal: this launch's {0xFFFFF - candidates, batch}
  const uint32_t *vpmOrder;
  uint32_t vpmOrderN;        // entries of vpmOrder (a permutation of the last launch's batches; the counts may differ a little)
  uint32_t *vpmCostKey, *vpmCostVal;
  // Shifts the fp32 kernels could not DECIDE as the reference does (a comparison inside its rigorous error margin): nothing
  // is added or counted for them by the fast kernels.  exOvf / exOvfCount {count, ticket}: this gather's notes; exPay /
  // exPayCount {entries, lost notes, ticket}: the exact pass's entries (see ExEntry).  The counters keep counting past
  // the capacities: the excess is reported as dropped pairs (gvpm_get_stats fails).
  ExEntry *exPay;
  uint32_t *exPayCount;
  uint32_t exPayCap;
  uint4 *exOvf;
  uint32_t *exOvfCount;
  uint32_t exOvfCap;
  // outputs
  float *iter;               // P * 27: this iteration's un-normalised sums (G-BRE: the running SUM over iterations)
  float iterScale;           // G-BRE: 1 / nb_paths of this iteration, applied when a partial sum is added
  unsigned long long *stats; // GVPM_STAT_ROWS rows of 8 counters (gvpm_stats order), summed on read
};

// G-VPM as three kernels (gather_vpm.hip, launch_gather_vpm_split): what the walk leaves for the evaluation.
// state: what the evaluation needs of a camera sample that found photons (written once per sample, by the walk)
struct alignas(16) VpmSampleState {
  double t;            // sampled camera distance (mRec.t)
  float pdfBase;       // mRec.pdfSuccess * pdfSel
  float trBase;        // exp(-sigma_t (t - mint))
  float radius;        // R * 0.01 * gp.scaleVol
  float pdfSel;
  uint32_t set;        // beam set of the sample
  uint32_t pix;        // y << 16 | x
  uint32_t edge;
  uint32_t pad[3];
};
constexpr uint32_t VPM_SHARDS = 64;       // chunk cursors (a cursor per 128-byte line: ctl[shard * 32])
constexpr uint32_t VPM_CTL_REDO = VPM_SHARDS * 32;  // ctl[VPM_CTL_REDO]: batches in the redo list
struct VpmSplit {
  uint2 *pairs;        // chunks of 64 {photon, camera sample}; chunk c of shard k is chunk k * shardChunks + c of the pool
  uint2 *chunkMeta;    // per chunk {pairs, batch}
  uint32_t *ctl;       // the shards' cursors and the redo count (zeroed before the walk)
  uint32_t *status;    // per batch: 1 = its pairs did not fit the pool, the whole batch is the redo kernel's (its chunks are skipped)
  uint32_t *redo;      // those batches
  VpmSampleState *state;
  uint32_t shardChunks, nBatches;
  uint32_t *zeroWord;  // cleared by the walk (the largest-scale word)
};

// The G-BRE build chain (grid_build.hip, launch_build_chain): what its first five launches share.
struct ChainArgs {
  // photons -> cells
  const float *pos;
  uint32_t n;
  Grid g;
  uint32_t *keys, *rank;      // per photon: cell key, arrival rank within the cell
  // ONE counter array and ONE scan for the cells and the beam sets' tile keys: [0, ncells] the cells (entry ncells stays
  // 0, so starts[ncells] = photons sorted), [beamOff, beamOff + nkeys] the keys.  counts: zero on entry and again on exit.
  uint32_t *counts, *starts;
  uint32_t scanLen;           // beamOff + nkeys + 1
  uint32_t beamOff;           // ncells + 1
  uint32_t *sub;              // bundle cells: the striped counters (cell_count_kernel), zeroed by the caller
  uint32_t *buckets;          // 2 x 64 x 6 order-preserving codes: the photons' bounds, the camera beams' (grid_build.hip, ordCode)
  float *out6, *hostB6;       // the bounds: device {photons 0-5, beams 6-11}, pinned host {photons 0-5, beams 16-21} (optional)
  // camera-beam sets -> tiles
  const gvpm_camera_ray *rays;
  uint32_t nsets;
  int width, tw, th;
  uint32_t tileShift, ntiles;
  uint32_t *bKeys, *bRank, *setPerm, *tileStart;
  uint32_t *blockSum;         // the scan's block sums
  uint32_t *ctl;              // 2 x 65 arrival counters (lastBlockArrives), zero on entry and on exit
  // words the chain initialises for the kernels behind it
  uint32_t *queueCtl;         // 8 words -> 0
  uint32_t *overflowCtr;      // -> 0
  uint32_t *nearExt;          // word 0 (the extension lists' cursor) -> 1
  uint32_t *sat;
  uint32_t nPhBlocks, nBeamBlocks;  // (filled by the launcher)
};

// Counter rows: a workgroup adds into row blockIdx % GVPM_STAT_ROWS.  One shared row would put every workgroup's
// atomics on one cache line, and atomics on one address retire at ~11 ns each (41 k workgroups of G-VPM: ~1 ms).
#define GVPM_STAT_ROWS 8192
__device__ __forceinline__ unsigned long long *statRow(const GatherArgs &a) {
  return a.stats + 8 * (size_t)(blockIdx.x % GVPM_STAT_ROWS);
}

}  // namespace gvpm
