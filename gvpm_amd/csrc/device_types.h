// Device-side data layout of libgvpm_hip.so (gfx950).  See DESIGN.md "HBM layout".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gvpm_hip.h"

namespace gvpm {

// Photon records after the grid build, sorted by cell (x fastest):
//   hot[i]      = {pos.xyz, bits}         16 B  -- streamed by the hit test
//   cold[k][i], k = 0..6                  7 x 16 B planes -- gathered per evaluation
//     0 {wi, parentPdf} 1 {flux, edgePdf} 2 {parentPos, parentRR} 3 {parentN, parentG}
//     4 {prefixW, -}    5 {parentScat, -} 6 {parentWi, -}
// bits = GVPM_PF_* flags of the ABI with bit 7 = pathID & 1.
#define GVPM_HOT_PARITY_BIT 7
#define GVPM_COLD_PLANES 7

struct Grid {
  float org[3];    // world position of cell (0,0,0)'s lower corner
  float cell;      // cell edge length
  float invCell;
  int dim[3];
  uint32_t ncells;
};

struct SortTemp {
  void *d = nullptr;
  size_t bytes = 0;
};

struct MediumDev {
  float sigmaS[3], sigmaT[3];
  float g, msw;
};

struct GatherArgs {
  // photons
  const float4 *hot;
  const float4 *cold;        // GVPM_COLD_PLANES planes of `nph` float4
  const uint32_t *cellStart; // ncells + 1
  uint32_t nph;
  Grid grid;
  // camera beam sets (5 x 64 B each), visited through the tile permutation
  const gvpm_camera_ray *rays;
  const uint32_t *setPerm;   // beam sets ordered by image tile
  const uint32_t *tileStart; // ntiles + 1 offsets into setPerm
  uint32_t nsets;
  // scene
  const float *triV0, *triE1, *triE2;  // 3 * ntri each
  uint32_t ntri;
  MediumDev med;
  // config
  gvpm_params cfg;
  float radius;
  // G-Beams only: a.radius is the traversal radius (kernel radius + half a sub-beam),
  // cold holds 9 planes of `nbeams`, hot holds the sub-beam centres {xyz, beam | sub << 24}
  float kernelRadius;
  float subLen;              // target sub-beam length used by the build
  uint32_t nbeams;
  // G-VPM only
  const gvpm_vpm_sample *samples;
  uint32_t nsamples;
  const float *scaleVol;     // per pixel GatherPoint::scaleVol (read)
  float *mvol;               // per pixel photons found this iteration (MVol, atomically added)
  // outputs
  float *iter;               // P * 27, this iteration's un-normalised sums
  unsigned long long *stats; // 8 counters (gvpm_stats order)
};

}  // namespace gvpm
