// Device-side acceleration-structure build for gfx950.
//
// Replaces GPhotonMap::build() (PointKDTree sliding midpoint, serial on the CPU;
// gvpm/gvpm_accel.h:201-203, include/mitsuba/core/kdtree.h:326-395) and the
// GradientBeamRadianceEstimator constructor (gvpm/gvpm_accel.cpp:10-54).  Every
// photon of a BRE pass has the same radius (gvpm.cpp:989,  gvpm_accel.cpp:27), so a
// uniform grid sorted by cell index (x fastest) gives x-contiguous photon ranges
// for any axis-aligned cell box -- the access pattern the gather kernel streams.
//
// Also orders the camera beam sets by image tile so that one wave works on a
// coherent bundle of beams (the role of BlockScheduler's image blocks,
// photonmapper/utilities/block_sched.h:19-136).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "bundle_grid.h"
#include "device_types.h"
#include "shift_device.h"
#include "tile_walk.h"
#include "vec.h"

namespace gvpm {

// ---- bounds: per-block min/max of the photon positions -----------------------------------
__global__ __launch_bounds__(256) void bounds_kernel(const float *__restrict__ pos, uint32_t n, float *partial) {
  __shared__ float red[6][4];
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float v = pos[3 * (size_t)i + c];
      mn[c] = fminf(mn[c], v);
      mx[c] = fmaxf(mx[c], v);
    }
  }
  const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float a = wave_min(mn[c]), b = wave_max(mx[c]);
    if (lane == 0) {
      red[c][wave] = a;
      red[3 + c][wave] = b;
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int c = threadIdx.x;
    float v = red[c][0];
    for (int w = 1; w < (int)(blockDim.x / 64u); ++w) v = c < 3 ? fminf(v, red[c][w]) : fmaxf(v, red[c][w]);
    partial[blockIdx.x * 6 + c] = v;
  }
}

// hostOut (optional): device-visible pinned host memory -- the result lands there without a copy
// operation in the stream (copy engines and the streams' kernels do not overlap reliably)
// (word / wordOut, optional: one more 32-bit value for the host -- G-VPM's largest scale rides along instead of a launch of its own)
__global__ void bounds_final_kernel(const float *partial, int nblocks, float *out6, float *hostOut, const uint32_t *word,
                                    uint32_t *wordOut) {
  const int lane = threadIdx.x;  // one wave
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int b = lane; b < nblocks; b += 64)
    for (int c = 0; c < 3; ++c) {
      lo[c] = fminf(lo[c], partial[b * 6 + c]);
      hi[c] = fmaxf(hi[c], partial[b * 6 + 3 + c]);
    }
  for (int c = 0; c < 3; ++c) {
    lo[c] = wave_min(lo[c]);
    hi[c] = wave_max(hi[c]);
  }
  if (lane == 0)
    for (int c = 0; c < 3; ++c) {
      out6[c] = lo[c];
      out6[3 + c] = hi[c];
      if (hostOut) {
        hostOut[c] = lo[c];
        hostOut[3 + c] = hi[c];
      }
    }
  if (lane == 0 && word && wordOut) wordOut[0] = word[0];
}

// counters -> pinned host memory (same reason)
__global__ void export_u32_kernel(const uint32_t *a, const uint32_t *b, const uint32_t *c, const uint32_t *d,
                                  const uint32_t *e, uint32_t *hostOut) {
  if (threadIdx.x == 0) {
    hostOut[0] = a ? *a : 0u;
    hostOut[1] = b ? *b : 0u;
    hostOut[2] = c ? *c : 0u;
    hostOut[3] = d ? *d : 0u;
    hostOut[4] = e ? *e : 0u;
    __threadfence_system();
  }
}

// ---- cell keys ---------------------------------------------------------------------------

// counting sort, pass 1: key of every photon, its arrival rank within the cell, photons per cell
// `stripes` (bundle cells): the counter of cell k is split into CELL_STRIPES counters, stripe s at sub[s * ncells + k], and
// photon i counts in stripe i % CELL_STRIPES.  Image-space cells are far from equally filled (every photon around a light
// seen by the camera shares a handful of them) and atomics on one address retire ~11 ns apart: 4 M photons took 0.53 ms
// through one counter per cell (0.21 ms through the 3D grid's, < 0.03 ms with the atomic compiled out).
// cell_stripes_kernel then turns the stripes into exclusive prefixes within the cell and writes the cell's total to count[].
#ifndef GVPM_CELL_STRIPES
#define GVPM_CELL_STRIPES 16
#endif
constexpr uint32_t CELL_STRIPES = GVPM_CELL_STRIPES;  // (C4, a rank of 8: 4 / 8 / 16 / 32 stripes: 2.57 / 2.58 / 2.48 / 2.55 ms per step; one counter: 2.75)
// (zeroWord / oneWord, optional: the near lists' overflow counter and the cursor of their extension lists, which the scatter
// behind this kernel starts from 0 and 1 -- two memsets less in the build chain)
__global__ __launch_bounds__(256) void cell_count_kernel(const float *__restrict__ pos, uint32_t n, Grid g,
                                                         uint32_t *keys, uint32_t *rank, uint32_t *count, uint32_t *sub,
                                                         uint32_t *zeroWord, uint32_t *oneWord) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    if (zeroWord) *zeroWord = 0u;
    if (oneWord) *oneWord = 1u;
  }
  if (i >= n) return;
  uint32_t k;
  if (g.mode == 1) {
    k = bundlePhotonKey(g, pos[3 * (size_t)i + 0], pos[3 * (size_t)i + 1], pos[3 * (size_t)i + 2]);
    keys[i] = k;
    // a photon no ray of the bundle can meet is not sorted at all (no box reaches the dump cell; reorder_kernel skips it)
    if (k == GVPM_BUNDLE_DUMP_CELL(g.dim[0])) rank[i] = 0xFFFFFFFFu;
    else if (sub) rank[i] = atomicAdd(&sub[(size_t)(i % CELL_STRIPES) * g.ncells + k], 1u);
    else rank[i] = atomicAdd(&count[k], 1u);
    return;
  }
  const int cx = cellCoord(pos[3 * (size_t)i + 0], g.org[0], g.invCell, g.dim[0]);
  const int cy = cellCoord(pos[3 * (size_t)i + 1], g.org[1], g.invCell, g.dim[1]);
  const int cz = cellCoord(pos[3 * (size_t)i + 2], g.org[2], g.invCell, g.dim[2]);
  k = ((uint32_t)cz * g.dim[1] + cy) * g.dim[0] + cx;
  keys[i] = k;
  rank[i] = atomicAdd(&count[k], 1u);
}
__global__ __launch_bounds__(256) void cell_stripes_kernel(uint32_t *sub, uint32_t ncells, uint32_t *count) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= ncells) return;
  uint32_t run = 0;
#pragma unroll
  for (uint32_t st = 0; st < CELL_STRIPES; ++st) {
    const uint32_t c = sub[(size_t)st * ncells + k];
    sub[(size_t)st * ncells + k] = run;
    run += c;
  }
  count[k] = run;
}

// ---- summed-volume table over the cell counts: T(x,y,z) = photons in cells {x' < x, y' < y, z' < z},
// (dimx+1)(dimy+1)(dimz+1) entries, x fastest.  The planner counts the photons of a cell box with 8
// reads instead of two per cell row.
// pass Y: S1(x,y,z) = sum_{y' < y} rowPrefix(x,y',z), one thread per (x,z); the row prefixes come
// straight from cellStart (an exclusive prefix in x-fastest order)
__global__ __launch_bounds__(256) void sat_y_kernel(const uint32_t *__restrict__ cellStart, Grid g, uint32_t *sat) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nx1 = g.dim[0] + 1, ny1 = g.dim[1] + 1;
  if (t >= nx1 * (uint32_t)g.dim[2]) return;
  const uint32_t x = t % nx1, z = t / nx1;
  uint32_t run = 0;
  sat[((size_t)z * ny1 + 0) * nx1 + x] = 0u;
#pragma unroll 8
  for (int y = 0; y < g.dim[1]; ++y) {
    const size_t row = ((size_t)z * g.dim[1] + y) * g.dim[0];
    run += cellStart[row + x] - cellStart[row];
    sat[((size_t)z * ny1 + (y + 1)) * nx1 + x] = run;
  }
}
// pass Z (in place): T(x,y,z) = sum_{z' < z} S1(x,y,z'), one thread per (x,y)
__global__ __launch_bounds__(256) void sat_z_kernel(Grid g, uint32_t *sat) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nx1 = g.dim[0] + 1, ny1 = g.dim[1] + 1;
  if (t >= nx1 * ny1) return;
  const size_t layer = (size_t)nx1 * ny1;
  uint32_t run = 0;
#pragma unroll 8
  for (int z = 0; z < g.dim[2]; ++z) {
    const uint32_t v = sat[(size_t)z * layer + t];
    sat[(size_t)z * layer + t] = run;
    run += v;
  }
  sat[(size_t)g.dim[2] * layer + t] = run;
}

// ---- reorder: SoA upload layout -> sorted hot/cold planes ----------------------------------
struct RawPhotons {
  const float *pos, *wi, *flux, *parent_pos, *parent_n, *prefix_w, *parent_scat, *parent_wi;
  const float *parent_pdf, *edge_pdf, *parent_rr, *parent_g;
  const uint32_t *flags, *path_id;
};

__device__ __forceinline__ float4 ld3(const float *p, uint32_t i, float w) {
  return make_float4(p[3 * (size_t)i], p[3 * (size_t)i + 1], p[3 * (size_t)i + 2], w);
}

// Occluders a shadow segment of length <= dmax starting at the photon's parent can reach:
// parent within dmax of the triangle's plane and of its (dmax-inflated) bounding box.
// Packs up to twelve 8-bit indices into three words (0xFF = empty slot); 0xFE in the top byte of the
// first word = overflow / too many occluders for 8-bit indices: such photons need the BVH kernels.
// Three formats, chosen by the occluder count (GVPM_NEAR_* in device_types.h):
//   narrow  (<= 253)    twelve 8-bit indices in the three words, 0xFF = empty
//   wide    (<= 64767)  six 16-bit indices, 0xFFFF = empty
//   ext                 word 0 = 0xFD << 24, word 1 = offset into the extension array {count, index...}: lists too long for
//                       the inline slots, and every non-empty list of a scene too large for 16-bit indices
// 0xFE in the top byte of word 0 = the extension array is full: such photons need the BVH kernels.
// Near = within dmax of the triangle's plane and of its (dmax-inflated) bounding box; found by a linear scan for
// small scenes, else by a point query of the occluder BVH (boxes inflated by dmax): the as-written segment is a
// thousandth of the reconnection distance, so the query touches a handful of leaves whatever the scene size.
__device__ __forceinline__ bool nearTriangle(f3 P, const float4 *tri4, uint32_t i, float dmax) {
  const float4 t0 = tri4[3 * (size_t)i], t1 = tri4[3 * (size_t)i + 1], t2 = tri4[3 * (size_t)i + 2];
  const f3 a = mk3(t0.x, t0.y, t0.z), b = mk3(t1.x, t1.y, t1.z), c = mk3(t2.x, t2.y, t2.z);
  const f3 n = mk3(t0.w, t1.w, t2.w);  // unit normal (zero for a degenerate triangle)
  if (fabsf(dot(n, P - a)) > dmax * 1.0001f) return false;
  bool out = false;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float ak = comp(a, k), bk = comp(b, k), ck = comp(c, k), pk = comp(P, k);
    const float lo = ak + fminf(0.f, fminf(bk, ck)), hi = ak + fmaxf(0.f, fmaxf(bk, ck));
    if (pk < lo - dmax || pk > hi + dmax) out = true;
  }
  return !out;
}
// MODE (one per instantiation of reorder_kernel, chosen at launch: the three walks inlined together cost the small
// scenes' kernel a quarter of its speed): 0 linear scan (<= 64 occluders), 1 BVH point query, 2 occluder grid
template <int MODE, class F>
__device__ __forceinline__ void nearVisit(f3 P, const float4 *bvh, const float4 *tri4, uint32_t ntri, float dmax,
                                          const NearGrid &ng, F f) {
  if (MODE == 0) {
    for (uint32_t i = 0; i < ntri; ++i)
      if (nearTriangle(P, tri4, i, dmax)) f(i);
    return;
  }
  if (MODE == 2) {
    // the cell's occluder list (built for a reach >= dmax): one dependent load instead of a stack walk of the BVH,
    // which was latency-bound per lane -- 2.1 ms of a 4.4 ms step for 4 M photons among 780 occluders (C4)
    const float cx = (P.x - ng.org[0]) * ng.inv[0], cy = (P.y - ng.org[1]) * ng.inv[1], cz = (P.z - ng.org[2]) * ng.inv[2];
    if (!(cx >= 0.f && cy >= 0.f && cz >= 0.f && cx < (float)ng.dim[0] && cy < (float)ng.dim[1] && cz < (float)ng.dim[2])) return;
    const uint32_t c = ((uint32_t)cz * (uint32_t)ng.dim[1] + (uint32_t)cy) * (uint32_t)ng.dim[0] + (uint32_t)cx;
    const uint32_t e0 = ng.start[c], e1 = ng.start[c + 1];
    for (uint32_t e = e0; e < e1; ++e) {
      const uint32_t i = ng.tris[e];
      if (nearTriangle(P, tri4, i, dmax)) f(i);
    }
    return;
  }
  uint32_t stack[32];
  int sp = 0;
  uint32_t cur = 0;
  const float pad = dmax * 1.0001f;
  for (;;) {
    const float4 lo = bvh[2 * (size_t)cur], hi = bvh[2 * (size_t)cur + 1];
    bool descend = false;
    if (P.x >= lo.x - pad && P.x <= hi.x + pad && P.y >= lo.y - pad && P.y <= hi.y + pad && P.z >= lo.z - pad &&
        P.z <= hi.z + pad) {
      const uint32_t first = __float_as_uint(lo.w), count = __float_as_uint(hi.w);
      if (count == 0u) {
        if (sp < 32) stack[sp++] = first + 1u;  // (the host builder's depth is far below 32)
        cur = first;
        descend = true;
      } else {
        for (uint32_t i = first; i < first + count; ++i)
          if (nearTriangle(P, tri4, i, dmax)) f(i);
      }
    }
    if (!descend) {
      if (sp == 0) break;
      cur = stack[--sp];
    }
  }
}
// The wall the parent SITS ON cannot block its reconnections (round 4): the segment starts Epsilon along a direction that
// leaves the wall's plane on the photon's side -- a direction into the wall fails the shift before visibility matters
// (shift_volume_photon.cpp:404-412: the sign test on dot(n_g, dProj) / dot(n_g, edge.d)) -- and moves away from it.  Such
// triangles (coplanar with the parent to position rounding, normal parallel to the parent's) were 95 % of S-cbox's list
// entries: every evaluation wave walked the any-hit loop for tests that cannot succeed.  A medium parent has no normal
// (zero): nothing is skipped for it.
// Round 5 -- the skip derived from the segment instead of a tolerance.  The parent's fp32 position lies delta = N . (P - a)
// off the triangle's plane (rounding: 0 for an axis-aligned wall, a few 1e-7 either way for a tilted one); a segment
// along an admissible direction d (n_parent . d > 0: the sign test and the cosine tests of the reconnection) meets that
// plane at t_self = -delta / (N . d).  With the parent ON the plane or on the side its normal points to, t_self <= 0 for
// every admissible d: the wall can never be hit and is left out.  With the parent BEHIND the plane, directions within
// |delta| / Epsilon of grazing meet it at t_self >= Epsilon -- the reference's rayIntersect reports that self-hit
// (shift_volume_photon.cpp:396-398) -- so the wall STAYS in the list and the evaluation decides (triHitChecked,
// shift_device.h: fp64 where fp32 cannot tell).  delta's sign is pure rounding noise of the inputs: it is taken in fp64
// from the fp32 data, as the oracle's (and a double-precision reference's) Moeller-Trumbore sees it.
// cstar (in / out): for a parent BEHIND the plane the wall is left out all the same and its reach is kept instead --
// max over the parent's own-wall triangles of |delta| / Epsilon, delta the distance to the triangle's plane (unit normal):
// the segment along d meets that plane at t_self >= Epsilon  <=>  n . d <= |delta| / Epsilon.  The evaluation compares the
// reconnection's cosine with it (shiftDiffuse): above, no self-hit is possible; at or below, the shift goes to the exact pass.
__device__ __forceinline__ bool ownWall(f3 P, f3 pn, const float4 *tri4, uint32_t i, float eps, float &cstar) {
  const float4 t0 = tri4[3 * (size_t)i], t1 = tri4[3 * (size_t)i + 1], t2 = tri4[3 * (size_t)i + 2];
  const f3 a = mk3(t0.x, t0.y, t0.z), n = mk3(t0.w, t1.w, t2.w);
  const float tol = 1e-6f * (1.f + fabsf(P.x) + fabsf(P.y) + fabsf(P.z) + fabsf(a.x) + fabsf(a.y) + fabsf(a.z));
  const float align = dot(n, pn);
  // (parallel to 1e-3 rad: the parent's normal is this triangle's to rounding -- 6e-5 rad through the packed upload's
  // octahedral code -- and the evaluation's cosine is taken against the PARENT's normal)
  if (!(fabsf(align) > 0.9999995f && fabsf(dot(n, P - a)) <= tol)) return false;
  // N = e1 x e2 and delta in fp64 (exact products of fp32 data up to the last additions: |error| ~ 1e-16 of O(1) terms)
  const double e1x = t1.x, e1y = t1.y, e1z = t1.z, e2x = t2.x, e2y = t2.y, e2z = t2.z;
  const double nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
  const double delta = nx * ((double)P.x - (double)a.x) + ny * ((double)P.y - (double)a.y) + nz * ((double)P.z - (double)a.z);
  const bool front = align > 0.f ? delta >= 0.0 : delta <= 0.0;
  if (!front) {
    const double nl = sqrt(nx * nx + ny * ny + nz * nz);
    if (nl > 0.0) cstar = fmaxf(cstar, (float)(fabs(delta) / (nl * (double)eps)) * 1.000001f);
  }
  return true;
}

template <int MODE>
// cstarOut > 0: the parent lies behind a wall it sits on (ownWall); word 2 then carries that reach instead of entries (the
// inline lists hold 8 / 4 of them; bit 15 of the record's flags says so)
__device__ __forceinline__ void nearOccluders(f3 P, f3 pn, const float4 *bvh, const float4 *tri4, uint32_t ntri, float dmax,
                                              const NearGrid &ng, uint32_t *ext, uint32_t extCap, float eps, uint32_t &w0, uint32_t &w1,
                                              uint32_t &w2, float &cstarOut) {
  w0 = w1 = w2 = 0xFFFFFFFFu;
  cstarOut = 0.f;
  if (ntri == 0u) return;
  const bool narrow = ntri <= GVPM_NEAR_NARROW_MAX, wide = !narrow && ntri <= GVPM_NEAR_WIDE_MAX;
  const uint32_t cap = narrow ? 12u : (wide ? 6u : 0u);
  uint32_t cnt = 0, a0 = 0xFFFFFFFFu, a1 = 0xFFFFFFFFu, a2 = 0xFFFFFFFFu;
  float cstar = 0.f;
  nearVisit<MODE>(P, bvh, tri4, ntri, dmax, ng, [&](uint32_t i) {
    if (ownWall(P, pn, tri4, i, eps, cstar)) return;
    if (cnt < cap) {
      uint32_t word, sh, m;
      if (narrow) { word = cnt >> 2; sh = 8u * (cnt & 3u); m = ~(0xFFu << sh); }
      else        { word = cnt >> 1; sh = 16u * (cnt & 1u); m = ~(0xFFFFu << sh); }
      const uint32_t v = i << sh;
      if (word == 0u) a0 = (a0 & m) | v;
      else if (word == 1u) a1 = (a1 & m) | v;
      else a2 = (a2 & m) | v;
    }
    cnt++;
  });
  cstarOut = cstar;
  const uint32_t capC = cstar > 0.f ? (cap * 2u) / 3u : cap;  // (word 2 is taken)
  if (cnt <= capC) {
    w0 = a0; w1 = a1; w2 = cstar > 0.f ? __float_as_uint(cstar) : a2;
    return;
  }
  // extension list {count, indices}
  const uint32_t off = atomicAdd(ext, cnt + 1u);  // ext[0] = the allocation cursor (starts at 1)
  if ((unsigned long long)off + cnt + 1u > extCap) {
    w0 = 0xFEFFFFFFu;
    return;
  }
  ext[off] = cnt;
  uint32_t k = 0;
  float dummy = 0.f;
  nearVisit<MODE>(P, bvh, tri4, ntri, dmax, ng, [&](uint32_t i) {
    if (!ownWall(P, pn, tri4, i, eps, dummy)) ext[off + 1u + (k++)] = i;
  });
  w0 = 0xFDFFFFFFu;
  w1 = off;
  if (cstar > 0.f) w2 = __float_as_uint(cstar);
}

// counting sort, pass 3: photon `src` (reads in upload order: coalesced) goes to slot cellStart[key] + rank
// (whole 128-byte records: full-line writes)
struct ReorderLds {
  float4 stg[64][GVPM_REC_QUADS / 2 + 1];  // half a record a pass; +1: odd stride against bank conflicts (5 KB: one wave a block)
  uint32_t dstIdx[64];
};
// bid: which 64 photons of the upload this wave takes (the kernel's block index, or the block's index among the reorder
// blocks of the build chain's tail kernel)
template <int MODE>
__device__ __forceinline__ void reorderBody(const RawPhotons &r, const uint32_t *__restrict__ keys,
                                            const uint32_t *__restrict__ rank,
                                            const uint32_t *__restrict__ cellStart, uint32_t n,
                                            const gvpm_params &cfg, const float4 *bvh, const float4 *tri4,
                                            uint32_t ntri, float dmax, const NearGrid &ng, uint32_t *nearExt,
                                            uint32_t extCap, float4 *hot, float4 *cold, uint32_t *overflow,
                                            uint32_t *origIdx, const uint32_t *__restrict__ sub, uint32_t ncells, uint32_t bid,
                                            uint32_t *counts, uint32_t *subCount, ReorderLds &L) {
  // The record is assembled in LDS and written by EIGHT lanes (one 16-byte quad each): a store instruction then
  // covers whole 64-byte halves of eight records instead of sixty-four 16-byte pieces of sixty-four lines.  Two passes of four
  // quads (round 6: 5 KB of LDS a wave instead of 9.5 -- the kernel runs beside the evaluation and the traversal).
  auto &stg = L.stg;
  auto &dstIdx = L.dstIdx;
  const uint32_t src = bid * 64u + threadIdx.x;
  const int t = threadIdx.x;
  constexpr int HALF = GVPM_REC_QUADS / 2;
  dstIdx[t] = 0xFFFFFFFFu;
  float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f), q5 = q4, q6 = q4, q7 = q4;
  const bool live = src < n && (counts || rank[src] != 0xFFFFFFFFu);  // (0xFFFFFFFF: outside the bundle, cell_count_kernel)
  if (live) {
    // (sub: the photon's rank counts within its stripe of the cell, cell_count_kernel)
    // (counts: the build chain -- no ranks were taken when the cells were counted; the counter is counted back down here)
    // (subCount: the chain with bundle cells -- the stripe's counter is counted back down, its prefix within the cell is in `sub`)
    uint32_t i;
    if (counts) i = cellStart[keys[src]] + (atomicSub(&counts[keys[src]], 1u) - 1u);
    else if (subCount) {
      const size_t sk = (size_t)(src % CELL_STRIPES) * ncells + keys[src];
      i = cellStart[keys[src]] + sub[sk] + (atomicSub(&subCount[sk], 1u) - 1u);
    } else i = cellStart[keys[src]] + rank[src] + (sub ? sub[(size_t)(src % CELL_STRIPES) * ncells + keys[src]] : 0u);
    uint32_t bits = r.flags[src] & ~((1u << 6) | (1u << GVPM_HOT_PARITY_BIT));
    if (photonContributes(bits, cfg)) bits |= 1u << 6;
    bits |= (r.path_id[src] & 1u) << GVPM_HOT_PARITY_BIT;
    const float4 h0 = ld3(r.pos, src, __uint_as_float(bits));
    hot[i] = h0;
    if (origIdx) origIdx[i] = src;  // (host-shift requests name photons by their place in the upload)
    // one 128-byte record per photon: an evaluation touches exactly one cache line
    stg[t][0] = h0;
    stg[t][1] = ld3(r.wi, src, r.parent_pdf[src]);
    stg[t][2] = ld3(r.flux, src, r.edge_pdf[src]);
    stg[t][3] = ld3(r.parent_pos, src, r.parent_rr[src]);
    q4 = ld3(r.parent_n, src, r.parent_g[src]);
    const f3 P = mk3(r.parent_pos[3 * (size_t)src], r.parent_pos[3 * (size_t)src + 1], r.parent_pos[3 * (size_t)src + 2]);
    uint32_t w0, w1, w2;
    const f3 PN = mk3(q4.x, q4.y, q4.z);
    float cstar;
    nearOccluders<MODE>(P, PN, bvh, tri4, ntri, dmax, ng, nearExt, extCap, cfg.epsilon, w0, w1, w2, cstar);
    if ((w0 >> 24) == 0xFEu) atomicAdd(overflow, 1u);
    // (bit 15 of the COLD record's flags only -- the depth field's top bit, which the evaluation does not read: "word 2 of the
    // near list is the reach of the wall the parent sits behind"; the hot record the traversal filters by keeps the flags)
    stg[t][0].w = __uint_as_float((bits & ~(1u << 15)) | (cstar > 0.f ? 1u << 15 : 0u));
    q5 = ld3(r.prefix_w, src, __uint_as_float(w0));
    q6 = ld3(r.parent_scat, src, __uint_as_float(w1));
    q7 = ld3(r.parent_wi, src, __uint_as_float(w2));
    dstIdx[t] = i;
  }
  __syncthreads();
  for (int e = t; e < 64 * HALF; e += 64) {
    const int rec = e / HALF, part = e % HALF;
    const uint32_t i = dstIdx[rec];
    if (i != 0xFFFFFFFFu) cold[(size_t)i * GVPM_REC_QUADS + part] = stg[rec][part];
  }
  __syncthreads();
  if (live) {
    stg[t][0] = q4;
    stg[t][1] = q5;
    stg[t][2] = q6;
    stg[t][3] = q7;
  }
  __syncthreads();
  for (int e = t; e < 64 * HALF; e += 64) {
    const int rec = e / HALF, part = e % HALF;
    const uint32_t i = dstIdx[rec];
    if (i != 0xFFFFFFFFu) cold[(size_t)i * GVPM_REC_QUADS + HALF + part] = stg[rec][part];
  }
}
template <int MODE>
__global__ __launch_bounds__(64) void reorder_kernel(RawPhotons r, const uint32_t *__restrict__ keys,
                                                      const uint32_t *__restrict__ rank,
                                                      const uint32_t *__restrict__ cellStart, uint32_t n,
                                                      gvpm_params cfg, const float4 *bvh, const float4 *tri4,
                                                      uint32_t ntri, float dmax, NearGrid ng, uint32_t *nearExt,
                                                      uint32_t extCap, float4 *hot, float4 *cold, uint32_t *overflow,
                                                      uint32_t *origIdx, const uint32_t *__restrict__ sub, uint32_t ncells) {
  __shared__ ReorderLds L;
  reorderBody<MODE>(r, keys, rank, cellStart, n, cfg, bvh, tri4, ntri, dmax, ng, nearExt, extCap, hot, cold, overflow, origIdx, sub,
                    ncells, blockIdx.x, nullptr, nullptr, L);
}

// ---- the occluder grid of nearVisit: triangle i is listed in every cell its bounding box, grown by `reach`, overlaps and
// whose centre is within reach + the cell's half extent (projected on the normal) of its plane.  One wave per triangle,
// the lanes striding over the cells of its box (built once per scene).  mode 0 counts, mode 1 fills.
__global__ __launch_bounds__(64) void near_grid_kernel(const float4 *__restrict__ tri4, uint32_t ntri, NearGrid g, float reach,
                                                       uint32_t *counts, uint32_t *tris, int mode) {
  const uint32_t i = blockIdx.x;
  if (i >= ntri) return;
  const float4 t0 = tri4[3 * (size_t)i], t1 = tri4[3 * (size_t)i + 1], t2 = tri4[3 * (size_t)i + 2];
  const f3 a = mk3(t0.x, t0.y, t0.z), e1 = mk3(t1.x, t1.y, t1.z), e2 = mk3(t2.x, t2.y, t2.z), n = mk3(t0.w, t1.w, t2.w);
  int lo[3], hi[3];
  float cs[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float ak = comp(a, k), bk = comp(e1, k), ck = comp(e2, k);
    const float l = ak + fminf(0.f, fminf(bk, ck)) - reach * 1.001f, h = ak + fmaxf(0.f, fmaxf(bk, ck)) + reach * 1.001f;
    cs[k] = 1.f / g.inv[k];
    lo[k] = max(0, (int)floorf((l - g.org[k]) * g.inv[k]) - 1);           // (one cell of slack either side: the
    hi[k] = min(g.dim[k] - 1, (int)floorf((h - g.org[k]) * g.inv[k]) + 1);  //  float cell index of a query may round)
  }
  const float slab = reach * 1.001f + 0.5f * 1.001f * (fabsf(n.x) * cs[0] + fabsf(n.y) * cs[1] + fabsf(n.z) * cs[2]) + 1e-6f * (cs[0] + cs[1] + cs[2]);
  const bool flat = n.x == 0.f && n.y == 0.f && n.z == 0.f;  // degenerate: no plane to cull with
  const int nx = hi[0] - lo[0] + 1, ny = hi[1] - lo[1] + 1, nz = hi[2] - lo[2] + 1;
  if (nx <= 0 || ny <= 0 || nz <= 0) return;
  const long long ncell = (long long)nx * ny * nz;
  for (long long q = threadIdx.x; q < ncell; q += blockDim.x) {
        const int x = lo[0] + (int)(q % nx), y = lo[1] + (int)((q / nx) % ny), z = lo[2] + (int)(q / ((long long)nx * ny));
        const f3 c = mk3(g.org[0] + ((float)x + 0.5f) * cs[0], g.org[1] + ((float)y + 0.5f) * cs[1], g.org[2] + ((float)z + 0.5f) * cs[2]);
        if (!flat && fabsf(dot(n, c - a)) > slab + cs[0] + cs[1] + cs[2]) continue;  // (a whole cell of slack again)
        const uint32_t cell = ((uint32_t)z * (uint32_t)g.dim[1] + (uint32_t)y) * (uint32_t)g.dim[0] + (uint32_t)x;
        if (mode == 0) atomicAdd(&counts[cell], 1u);
        else tris[g.start[cell] + atomicAdd(&counts[cell], 1u)] = i;
  }
}
void launch_near_grid(const float4 *tri4, uint32_t ntri, const NearGrid &g, float reach, uint32_t *counts, uint32_t *tris, int mode,
                      hipStream_t s) {
  if (ntri) hipLaunchKernelGGL(near_grid_kernel, dim3(ntri), dim3(64), 0, s, tri4, ntri, g, reach, counts, tris, mode);
}

// ---- segment starts of a sorted key array: start[c] = first i with (key[i] >> shift) >= c ----
// one thread per segment, binary search (segments far outnumber the long empty runs a
// per-element scatter would serialise on)
__global__ __launch_bounds__(256) void segment_start_kernel(const uint32_t *__restrict__ keys, uint32_t n,
                                                            uint32_t nseg, uint32_t shift, uint32_t *start) {
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > nseg) return;
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if ((keys[mid] >> shift) < c) lo = mid + 1;
    else hi = mid;
  }
  start[c] = lo;
}

// ---- camera beam sets: tile keys -------------------------------------------------------------
// key = ((tileIndex * tilePixels + pixelInTile) << 3 | (edge & 7)); tile index = key >> tileShift
__global__ __launch_bounds__(256) void beam_key_kernel(const gvpm_camera_ray *__restrict__ rays, uint32_t nsets,
                                                       int width, int tw, int th, uint32_t *keys, uint32_t *vals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nsets) return;
  const gvpm_camera_ray &b = rays[(size_t)i * 5];
  const uint32_t px = b.pixel & 0xFFFFu, py = b.pixel >> 16;
  const uint32_t tilesX = (width + tw - 1) / tw;
  const uint32_t tile = (py / th) * tilesX + px / tw;
  const uint32_t inTile = (py % th) * tw + px % tw;
  keys[i] = ((tile * (uint32_t)(tw * th) + inTile) << 3) | (GVPM_RAY_EDGE(b.info) & 7u);
  vals[i] = i;
}

// counting sort of the beam sets by the same key
__global__ __launch_bounds__(256) void beam_count_kernel(const gvpm_camera_ray *__restrict__ rays, uint32_t nsets,
                                                         int width, int tw, int th, uint32_t *keys, uint32_t *rank,
                                                         uint32_t *count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nsets) return;
  const gvpm_camera_ray &b = rays[(size_t)i * 5];
  const uint32_t px = b.pixel & 0xFFFFu, py = b.pixel >> 16;
  const uint32_t tilesX = (width + tw - 1) / tw;
  const uint32_t tile = (py / th) * tilesX + px / tw;
  const uint32_t inTile = (py % th) * tw + px % tw;
  const uint32_t k = ((tile * (uint32_t)(tw * th) + inTile) << 3) | (GVPM_RAY_EDGE(b.info) & 7u);
  keys[i] = k;
  rank[i] = atomicAdd(&count[k], 1u);
}

__global__ __launch_bounds__(256) void beam_scatter_kernel(const uint32_t *__restrict__ keys,
                                                           const uint32_t *__restrict__ rank,
                                                           const uint32_t *__restrict__ start, uint32_t n,
                                                           uint32_t *setPerm) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  setPerm[start[keys[i]] + rank[i]] = i;
}

// tileStart[t] = first slot of tile t = start[t << shift]; start has (ntiles << shift) + 1 entries
__global__ __launch_bounds__(256) void tile_start_kernel(const uint32_t *__restrict__ start, uint32_t ntiles,
                                                         uint32_t shift, uint32_t *tileStart) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t > ntiles) return;
  tileStart[t] = start[(size_t)t << shift];
}

// ---- host-side drivers -------------------------------------------------------------------------

static hipError_t ensureTemp(SortTemp &t, size_t need) {
  if (need <= t.bytes) return hipSuccess;
  if (t.d) (void)hipFree(t.d);
  t.d = nullptr;
  t.bytes = 0;
  hipError_t e = hipMalloc(&t.d, need);
  if (e == hipSuccess) t.bytes = need;
  return e;
}

// (The size query is a full host-side pass of the library's dispatch -- ~240 us, device properties included -- and the G-Beams
// step paid it twice per sort with the GPU idle behind it: asked once per high-water mark, for 1.25 x the count.)
hipError_t sortPairsU32(SortTemp &tmp, const uint32_t *kIn, uint32_t *kOut, const uint32_t *vIn, uint32_t *vOut,
                        uint32_t n, int endBit, hipStream_t s) {
  if (n > tmp.sortN || tmp.sortNeed == 0) {
    const uint32_t nq = (uint32_t)std::min<uint64_t>((uint64_t)n + n / 4u + 1024u, 0x7FFFFFF0u);
    size_t need = 0;
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(nullptr, need, kIn, kOut, vIn, vOut, (int)nq, 0, 32, s);
    if (e != hipSuccess) return e;
    tmp.sortN = nq;
    tmp.sortNeed = std::max(need, tmp.sortNeed);
  }
  hipError_t e = ensureTemp(tmp, tmp.sortNeed);
  if (e != hipSuccess) return e;
  size_t have = tmp.bytes;
  return hipcub::DeviceRadixSort::SortPairs(tmp.d, have, kIn, kOut, vIn, vOut, (int)n, 0, endBit, s);
}

// make the temporary large enough for scans of up to n elements (so that no scan of a step allocates)
hipError_t reserveScanTemp(SortTemp &tmp, uint32_t n) {
  return ensureTemp(tmp, ((size_t)n / 2048 + 2) * sizeof(uint32_t) + 256);
}

// Exclusive prefix sum in three plain kernels: block sums, their scan by one block, down-sweep.  (hipCUB's single-pass
// scan makes every block look back at its predecessors' flags: beside the persistent evaluation waves of the previous
// step, which hold most of the chip until they finish, those predecessors are often not resident yet -- the two scans of
// a G-BRE build, 50 us each alone, took 260 us each in the pipelined run.  Nothing here waits for another block.)
constexpr uint32_t SCAN_BLOCK = 256, SCAN_PER_THREAD = 8, SCAN_TILE = SCAN_BLOCK * SCAN_PER_THREAD;
__device__ __forceinline__ uint32_t block_scan_excl(uint32_t v, uint32_t *lds, uint32_t &total) {
  // 256 threads = 4 waves: wave scan, then the 4 wave totals
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t incl = wave_scan_incl(v, lane);
  if (lane == 63) lds[wv] = incl;
  __syncthreads();
  uint32_t base = 0;
  for (int k = 0; k < wv; ++k) base += lds[k];
  total = lds[0] + lds[1] + lds[2] + lds[3];
  __syncthreads();
  return base + incl - v;
}
__global__ __launch_bounds__(256) void scan_reduce_kernel(const uint32_t *__restrict__ in, uint32_t n, uint32_t *blockSum) {
  __shared__ uint32_t lds[4];
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
  uint32_t v = 0;
#pragma unroll
  for (uint32_t k = 0; k < SCAN_PER_THREAD; ++k)
    if (base + k < n) v += in[base + k];
  uint32_t total;
  (void)block_scan_excl(v, lds, total);
  if (threadIdx.x == 0) blockSum[blockIdx.x] = total;
}
__global__ __launch_bounds__(256) void scan_spine_kernel(uint32_t *blockSum, uint32_t nblocks) {
  __shared__ uint32_t lds[4];
  uint32_t carry = 0;
  for (uint32_t b0 = 0; b0 < nblocks; b0 += SCAN_BLOCK) {
    const uint32_t i = b0 + threadIdx.x;
    const uint32_t v = i < nblocks ? blockSum[i] : 0u;
    uint32_t total;
    const uint32_t ex = block_scan_excl(v, lds, total);
    if (i < nblocks) blockSum[i] = carry + ex;
    carry += total;
  }
}
__global__ __launch_bounds__(256) void scan_down_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t n,
                                                        const uint32_t *__restrict__ blockSum) {
  __shared__ uint32_t lds[4];
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
  uint32_t x[SCAN_PER_THREAD], v = 0;
#pragma unroll
  for (uint32_t k = 0; k < SCAN_PER_THREAD; ++k) {
    x[k] = base + k < n ? in[base + k] : 0u;
    v += x[k];
  }
  uint32_t total;
  uint32_t run = blockSum[blockIdx.x] + block_scan_excl(v, lds, total);
#pragma unroll
  for (uint32_t k = 0; k < SCAN_PER_THREAD; ++k) {
    if (base + k < n) out[base + k] = run;  // (in == out is fine: every thread has read its own eight before it writes)
    run += x[k];
  }
}
hipError_t exclusiveSumU32(SortTemp &tmp, const uint32_t *in, uint32_t *out, uint32_t n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  const uint32_t nblocks = (n + SCAN_TILE - 1) / SCAN_TILE;
  hipError_t e = ensureTemp(tmp, (size_t)nblocks * sizeof(uint32_t) + 256);
  if (e != hipSuccess) return e;
  uint32_t *blockSum = reinterpret_cast<uint32_t *>(tmp.d);
  hipLaunchKernelGGL(scan_reduce_kernel, dim3(nblocks), dim3(SCAN_BLOCK), 0, s, in, n, blockSum);
  hipLaunchKernelGGL(scan_spine_kernel, dim3(1), dim3(SCAN_BLOCK), 0, s, blockSum, nblocks);
  hipLaunchKernelGGL(scan_down_kernel, dim3(nblocks), dim3(SCAN_BLOCK), 0, s, in, out, n, blockSum);
  return hipGetLastError();
}

// records of photon beams (indexed by beam, not sorted), 128 bytes = one cache line per evaluation (the nine
// plane-major float4 arrays of round 1 cost nine):
//   0 {flux, parentPdf} 1 {p1, parentRR} 2 {parentN, parentG} 3 {prefixW, near0} 4 {parentScat, near1}
//   5 {parentWi, near2} 6 {p2, bits} 7 {endN, length}   (near0..2: beam_near_kernel)
__global__ __launch_bounds__(64) void beam_cold_kernel(RawPhotons r, const float *__restrict__ endN, uint32_t n,
                                                       gvpm_params cfg, const uint32_t *__restrict__ subCounts, float4 *cold,
                                                       float4 *aux) {
  // The record is assembled in LDS and written by EIGHT lanes (one 16-byte quad each), as reorder_kernel does: a store
  // instruction then covers eight whole 128-byte lines instead of sixty-four 16-byte pieces of sixty-four lines.
  __shared__ float4 stg[64][GVPM_REC_QUADS + 1];  // +1: odd stride against bank conflicts
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = threadIdx.x;
  if (i < n) {
    uint32_t bits = r.flags[i] & ~((1u << 6) | (1u << GVPM_HOT_PARITY_BIT));
    // for beams the debugShift filter only suppresses the shifts (shift_volume_beams.cpp:210-216),
    // so only computeVolumeContribution is folded into bit 6
    gvpm_params c2 = cfg;
    c2.debug_shift = GVPM_SHIFT_ALL;
    if (photonContributes(bits, c2)) bits |= 1u << 6;
    bits |= (r.path_id[i] & 1u) << GVPM_HOT_PARITY_BIT;
    stg[t][0] = ld3(r.flux, i, r.parent_pdf[i]);
    stg[t][1] = ld3(r.parent_pos, i, r.parent_rr[i]);
    stg[t][2] = ld3(r.parent_n, i, r.parent_g[i]);
    stg[t][3] = ld3(r.prefix_w, i, 0.f);
    stg[t][4] = ld3(r.parent_scat, i, 0.f);
    stg[t][5] = ld3(r.parent_wi, i, 0.f);
    stg[t][6] = ld3(r.pos, i, __uint_as_float(bits));
    // PhotonBeam::setEndPoint, pm/beams_struct.h:73-81: the length, once, in fp64 (an fp64 square root and division
    // per evaluation otherwise)
    const double dx = (double)r.pos[3 * (size_t)i] - (double)r.parent_pos[3 * (size_t)i];
    const double dy = (double)r.pos[3 * (size_t)i + 1] - (double)r.parent_pos[3 * (size_t)i + 1];
    const double dz = (double)r.pos[3 * (size_t)i + 2] - (double)r.parent_pos[3 * (size_t)i + 2];
    const double lenD = sqrt(dx * dx + dy * dy + dz * dz);
    stg[t][7] = ld3(endN, i, (float)lenD);
    // what the sorted sub-beam records are made of (sub_hot_kernel), 32 bytes per beam: {p1, bits} {direction, sub-beam
    // length} -- the fp64 norm and division once per beam instead of once per sub-beam (47 M of them at C3)
    if (aux) {
      const double inv = 1.0 / lenD;
      aux[2 * (size_t)i] = ld3(r.parent_pos, i, __uint_as_float(bits));
      aux[2 * (size_t)i + 1] = make_float4((float)(dx * inv), (float)(dy * inv), (float)(dz * inv), (float)lenD / (float)subCounts[i]);
    }
  }
  __syncthreads();
  const uint32_t base = blockIdx.x * blockDim.x;
  for (int e = t; e < 64 * GVPM_REC_QUADS; e += 64) {
    const int rec = e / GVPM_REC_QUADS, part = e % GVPM_REC_QUADS;
    if (base + (uint32_t)rec < n) cold[(size_t)(base + rec) * GVPM_REC_QUADS + part] = stg[rec][part];
  }
}

// ---- occluders near a photon beam -----------------------------------------------------------------------------
// The reconnection of a beam shift tests the whole NEW beam parent -> offset position for occluders
// (shift_volume_beams.cpp:420-426).  The offset position lies within `delta` of the point X of the original beam the
// kernel sits at, so every point of the new segment lies within delta of the original segment [p1, p2] (the two
// segments share p1 and end delta apart): only triangles whose box, inflated by delta, the original segment crosses
// can be hit.  delta = 4 r + S: |offsetPos - X| <= |shiftRay(w) - baseRay(w)| + (kernel offset, replayed in the
// shifted frame: 2 r) + (mirror adjustment of getShiftPos, shift_volume_beams.cpp:120-135: 2 r), and S bounds the
// first term over every uploaded beam set (shift_extent_kernel).  Up to 12 byte indices per beam go into the spare
// words of quads 3-5 of its record; a beam with more (or a scene of more than 253 occluders) is marked 0xFE in the
// top byte of word 0 and takes the full test.
__global__ __launch_bounds__(256) void shift_extent_kernel(const gvpm_camera_ray *__restrict__ rays, uint32_t nsets,
                                                           uint32_t *extentBits) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  float S = 0.f;
  if (i < nsets) {
    const gvpm_camera_ray b = rays[(size_t)i * 5];
    for (int k = 1; k < 5; ++k) {
      const gvpm_camera_ray s = rays[(size_t)i * 5 + k];
      if (!GVPM_RAY_VALID(s.info)) continue;
      const float ox = s.o[0] - b.o[0], oy = s.o[1] - b.o[1], oz = s.o[2] - b.o[2];
      const float dx = s.d[0] - b.d[0], dy = s.d[1] - b.d[1], dz = s.d[2] - b.d[2];
      // w <= min(len_b, len_s) (shift_volume_beams.cpp: the shift is dropped when w exceeds the shifted edge)
      const float e = sqrtf(ox * ox + oy * oy + oz * oz) + fminf(b.len, s.len) * sqrtf(dx * dx + dy * dy + dz * dz);
      S = fmaxf(S, e);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) S = fmaxf(S, __shfl_xor(S, o, 64));
  // S >= 0: bit order = value order; a wave first looks whether it would raise the maximum (same-address atomics retire ~11 ns apart)
  if ((threadIdx.x & 63) == 0 && S > 0.f &&
      __float_as_uint(S * 1.0001f) > __hip_atomic_load(extentBits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(extentBits, __float_as_uint(S * 1.0001f));
}

// The FREE CONE of a beam (round 3).  A new beam of a reconnection is a straight segment from the beam's origin p1: every
// point of it is seen from p1 under ONE angle to the beam's direction, alpha = angle(nd, bd), and lies no farther than
// its length.  For a listed triangle T let cosT = the largest cosine of that angle over the points of T (0 angle: the
// beam's line pierces it) and, when the line pierces T's PLANE at t_p > 0, alongMin = t_p - delta tan(incidence): every
// point of the plane within delta of the line -- all a new beam can reach, it stays within delta of the beam -- lies at
// least that far along the beam.  With M1 = the smallest alongMin of the triangles the beam points at (cosT ~ 1: the
// surface it ends on) and cosA0 = the largest cosT of the others,
//     dot(nd, bd) > cosA0  and  |new beam| < M1   =>   the new beam meets no occluder,
// because it is inside the cone no other triangle enters and too short for the ones in front.  The evaluation skips the
// any-hit loop for such reconnections (85-95 % of them at C3: a beam through the aperture is certified unless the new
// beam grazes the aperture's edge or reaches the floor), and takes the others through it as before.
// Returned in clear[i] = {cosA0 + margin, M1 - margin}; {2, 0} = never certified (list overflowed, degenerate).
struct BeamClearTri {
  float cosT, alongMin;
};
__device__ __forceinline__ BeamClearTri beamClearTri(const float p1[3], const float bd[3], float delta, const float4 t0, const float4 t1,
                                                     const float4 t2) {
  const float a[3] = {t0.x - p1[0], t0.y - p1[1], t0.z - p1[2]};
  const float e1[3] = {t1.x, t1.y, t1.z}, e2[3] = {t2.x, t2.y, t2.z};
  const float b[3] = {a[0] + e1[0], a[1] + e1[1], a[2] + e1[2]}, c[3] = {a[0] + e2[0], a[1] + e2[1], a[2] + e2[2]};
  auto dot3 = [](const float *x, const float *y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; };
  // the largest cosine over an edge P + s E, s in [0, 1]: g(s) = (al + be s) / |P + s E|, stationary at
  // s* = (al d - be c0) / (be d - al e)
  auto edgeMax = [&](const float *P, const float *Q) {
    const float E[3] = {Q[0] - P[0], Q[1] - P[1], Q[2] - P[2]};
    const float al = dot3(P, bd), be = dot3(E, bd), c0 = dot3(P, P), d = dot3(P, E), e = dot3(E, E);
    float m = -1.f;
    auto at = [&](float sp) {
      const float l2 = c0 + 2.f * d * sp + e * sp * sp;
      if (l2 > 1e-30f) m = fmaxf(m, (al + be * sp) * rsqrtf(l2));
      else m = 1.f;  // the beam's origin ON the edge: every direction of the plane is met
    };
    at(0.f);
    at(1.f);
    const float den = be * d - al * e;
    if (fabsf(den) > 1e-30f) {
      const float sp = (al * d - be * c0) / den;
      if (sp > 0.f && sp < 1.f) at(sp);
    }
    return m;
  };
  BeamClearTri o;
  o.cosT = fmaxf(edgeMax(a, b), fmaxf(edgeMax(b, c), edgeMax(c, a)));
  // the beam's line through the triangle (Moeller-Trumbore from p1 along bd, with slack: a near miss is a hit here)
  const float nrm[3] = {t0.w, t1.w, t2.w};
  const float cn = dot3(bd, nrm), dn = dot3(a, nrm);  // plane: nrm . (x - v0) = 0, a = v0 - p1
  // The wall the beam STARTS on (p1 within position rounding of the plane; round 5): p1 may lie a few 1e-7 BEHIND it, and a
  // new beam within |delta| / Epsilon (~ 4e-3 rad) of grazing then meets the plane at t >= Epsilon -- the reference's
  // rayIntersect reports that hit (shift_volume_beams.cpp:420-426).  The cone such a wall leaves free is drawn 0.02 rad
  // inside its plane: reconnections in the sliver take the any-hit loop, where triHitChecked decides them in fp64.
  {
    const float scale = fabsf(p1[0]) + fabsf(p1[1]) + fabsf(p1[2]) + fabsf(t0.x) + fabsf(t0.y) + fabsf(t0.z);
    if (fabsf(dn) <= 2e-6f * (1.f + scale) && o.cosT < 1.f) {
      const float sinT = sqrtf(fmaxf(1.f - o.cosT * o.cosT, 0.f));
      o.cosT = fminf(1.f, o.cosT * 0.9998f + sinT * 0.02f);  // cos(angle - 0.02)
    }
  }
  o.alongMin = -INFINITY;
  // The pierce test runs at ANY incidence: a line that crosses a large triangle at a grazing angle (|cn| <= 0.05) sees it
  // under angle 0 although every edge is far off axis -- filed under "others" with the edges' cosine, such a triangle let
  // reconnections inside that cone skip the any-hit test (a thin plate with medium on both sides).  alongMin is only
  // trusted away from grazing (delta tan(incidence) explodes there): a grazed triangle the beam points at leaves
  // M1 = -inf, i.e. the beam is never certified.
  if (fabsf(cn) > 1e-12f) {
    const float tp = dn / cn;  // p1 + tp bd on the plane
    if (tp > 0.f) {
      if (fabsf(cn) > 0.05f) o.alongMin = tp - delta * sqrtf(fmaxf(1.f - cn * cn, 0.f)) / fabsf(cn);
      // inside (or within slack of) the triangle?  barycentrics of the piercing point
      const float q[3] = {tp * bd[0] - a[0], tp * bd[1] - a[1], tp * bd[2] - a[2]};  // from v0
      const float d11 = dot3(e1, e1), d12 = dot3(e1, e2), d22 = dot3(e2, e2), q1 = dot3(q, e1), q2 = dot3(q, e2);
      const float det = d11 * d22 - d12 * d12;
      if (det > 1e-30f) {
        const float u = (q1 * d22 - q2 * d12) / det, v = (q2 * d11 - q1 * d12) / det;
        if (u > -1e-3f && v > -1e-3f && u + v < 1.001f) o.cosT = 1.f;
      }
    }
  }
  return o;
}

__global__ __launch_bounds__(128) void beam_near_kernel(float4 *cold, uint32_t n, const float4 *__restrict__ tri4, uint32_t ntri,
                                                        float r, const uint32_t *__restrict__ extentBits, float2 *clear, bool freeCone) {
  __shared__ float2 ctab[19][128];  // (19: the longest list, BeamNearFmt)
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t N = GVPM_REC_QUADS;
  uint32_t w[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
  const BeamNearFmt fmt = beamNearFmt(ntri);
  if (fmt.bits != 8u) w[2] &= 0x7FFFFFFFu;  // (bit 95: the narrow formats' overflow flag)
  if (ntri > GVPM_NEAR_NARROW_MAX) {
    w[0] = 0xFE000000u | 0x00FFFFFFu;
  } else {
    const float delta = 4.f * r + __uint_as_float(*extentBits);
    const float4 c2 = cold[N * i + 1], c7 = cold[N * i + 6];
    const float p1[3] = {c2.x, c2.y, c2.z}, p2[3] = {c7.x, c7.y, c7.z};
    uint32_t cnt = 0;
    for (uint32_t t = 0; t < ntri; ++t) {
      const float4 t0 = tri4[3 * t], t1 = tri4[3 * t + 1], t2 = tri4[3 * t + 2];
      const float v0[3] = {t0.x, t0.y, t0.z}, e1[3] = {t1.x, t1.y, t1.z}, e2[3] = {t2.x, t2.y, t2.z};
      // segment [p1, p2] against the triangle's box inflated by delta (slab test, conservative)
      float t0s = 0.f, t1s = 1.f;
      bool hit = true;
      for (int c = 0; c < 3; ++c) {
        const float lo = fminf(v0[c], fminf(v0[c] + e1[c], v0[c] + e2[c])) - delta;
        const float hi = fmaxf(v0[c], fmaxf(v0[c] + e1[c], v0[c] + e2[c])) + delta;
        const float d = p2[c] - p1[c];
        if (fabsf(d) < 1e-20f) {
          hit = hit && p1[c] >= lo && p1[c] <= hi;
        } else {
          const float inv = 1.f / d;
          const float a = (lo - p1[c]) * inv, b = (hi - p1[c]) * inv;
          t0s = fmaxf(t0s, fminf(a, b) - 1e-5f);
          t1s = fminf(t1s, fmaxf(a, b) + 1e-5f);
        }
      }
      hit = hit && t0s <= t1s;
      if (hit) {
        if (cnt < fmt.cap) {
          // entry cnt: fmt.bits bits at bit cnt * fmt.bits of the 96-bit string (it may straddle two words)
          const uint32_t off = cnt * fmt.bits, wi = off >> 5, sh = off & 31u;
          w[wi] = (w[wi] & ~(fmt.mask << sh)) | (t << sh);
          if (sh + fmt.bits > 32u) w[wi + 1] = (w[wi + 1] & ~(fmt.mask >> (32u - sh))) | (t >> (32u - sh));
        }
        cnt++;
      }
    }
    // (index 0xFE / 0xFF cannot occur in the top byte of word 0: ntri <= 253)
    if (cnt > fmt.cap) {
      if (fmt.bits == 8u) w[0] = 0xFE000000u | 0x00FFFFFFu;
      else w[2] |= 0x80000000u;
    }
  }
  cold[N * i + 3].w = __uint_as_float(w[0]);
  cold[N * i + 4].w = __uint_as_float(w[1]);
  cold[N * i + 5].w = __uint_as_float(w[2]);
  // the beam's free cone over its list (see above): two passes over the listed triangles
  float2 cl = make_float2(2.f, 0.f);
  if (freeCone && ntri <= GVPM_NEAR_NARROW_MAX && !beamNearOverflow(fmt, w[0], w[2])) {
    const float delta = 4.f * r + __uint_as_float(*extentBits);
    const float4 c2 = cold[N * i + 1], c7 = cold[N * i + 6];
    const float p1[3] = {c2.x, c2.y, c2.z};
    float bd[3] = {c7.x - c2.x, c7.y - c2.y, c7.z - c2.z};
    const float l2 = bd[0] * bd[0] + bd[1] * bd[1] + bd[2] * bd[2];
    if (l2 > 1e-30f) {
      const float il = rsqrtf(l2);
      bd[0] *= il; bd[1] *= il; bd[2] *= il;
      const float cosSmall = 0.99955f;  // ~0.03 rad: "the beam points at this triangle"
      // (each listed triangle's {cosT, alongMin} once: parked in this lane's LDS column between the two passes)
      float M1 = INFINITY;
      uint32_t nl = 0;
      for (uint32_t q = 0; q < fmt.cap; ++q) {
        const uint32_t t = beamNearEntry(fmt, w[0], w[1], w[2], q);
        if (t == fmt.mask) break;
        const BeamClearTri ct = beamClearTri(p1, bd, delta, tri4[3 * t], tri4[3 * t + 1], tri4[3 * t + 2]);
        ctab[q][threadIdx.x] = make_float2(ct.cosT, ct.alongMin);
        nl = q + 1;
        if (ct.cosT > cosSmall) M1 = fminf(M1, ct.alongMin);
      }
      float cosA0 = -1.f;
      for (uint32_t q = 0; q < nl; ++q) {
        const float2 ct = ctab[q][threadIdx.x];
        if (!(ct.y >= M1)) cosA0 = fmaxf(cosA0, ct.x);
      }
      if (M1 > 0.f) cl = make_float2(cosA0 + 1e-5f, M1 * (1.f - 1e-5f) - 1e-6f);
    }
  }
  if (clear) clear[i] = cl;
}

void launch_shift_extent(const gvpm_camera_ray *rays, uint32_t nsets, uint32_t *extentBits, hipStream_t s) {
  (void)hipMemsetAsync(extentBits, 0, sizeof(uint32_t), s);
  if (nsets) hipLaunchKernelGGL(shift_extent_kernel, dim3((nsets + 255) / 256), dim3(256), 0, s, rays, nsets, extentBits);
}

// GVPM_BEAMS_TRACE: how long the beams' near-occluder lists are -- hist[k] = lists of k entries (k <= 19), hist[20] = overflowed
__global__ __launch_bounds__(256) void beam_near_hist_kernel(const float4 *__restrict__ cold, uint32_t n, uint32_t ntri, uint32_t *hist) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t N = GVPM_REC_QUADS;
  const uint32_t w[3] = {__float_as_uint(cold[N * i + 3].w), __float_as_uint(cold[N * i + 4].w), __float_as_uint(cold[N * i + 5].w)};
  const BeamNearFmt fmt = beamNearFmt(ntri);
  uint32_t k = 20;
  if (!beamNearOverflow(fmt, w[0], w[2])) {
    k = 0;
    for (uint32_t q = 0; q < fmt.cap; ++q)
      if (beamNearEntry(fmt, w[0], w[1], w[2], q) != fmt.mask) k++;
  }
  atomicAdd(&hist[k], 1u);
}
void launch_beam_near_hist(const float4 *cold, uint32_t n, uint32_t ntri, uint32_t *hist, hipStream_t s) {
  if (n) hipLaunchKernelGGL(beam_near_hist_kernel, dim3((n + 255) / 256), dim3(256), 0, s, cold, n, ntri, hist);
}
void launch_beam_near(float4 *cold, uint32_t n, const float4 *tri4, uint32_t ntri, float r, const uint32_t *extentBits,
                      float2 *clear, bool freeCone, hipStream_t s) {
  if (n) hipLaunchKernelGGL(beam_near_kernel, dim3((n + 127) / 128), dim3(128), 0, s, cold, n, tri4, ntri, r, extentBits, clear, freeCone);
}

void launch_beam_cold(const gvpm_photon_soa &raw, const float *endN, uint32_t n, const gvpm_params &cfg,
                      const uint32_t *subCounts, float4 *cold, float4 *aux, hipStream_t s) {
  RawPhotons r{raw.pos,        raw.wi,         raw.flux,     raw.parent_pos, raw.parent_n,
               raw.prefix_w,   raw.parent_scat, raw.parent_wi, raw.parent_pdf, raw.edge_pdf,
               raw.parent_rr,  raw.parent_g,   raw.flags,    raw.path_id};
  hipLaunchKernelGGL(beam_cold_kernel, dim3((n + 63) / 64), dim3(64), 0, s, r, endN, n, cfg, subCounts, cold, aux);
}

void launch_bounds(const float *pos, uint32_t n, float *partial, int nblocks, float *out6, float *hostOut,
                   hipStream_t s, const uint32_t *word, uint32_t *wordOut) {
  // single-wave workgroups: beside the traversal's 22 000 one-wave workgroups a 4-wave workgroup waits for four slots
  // to fall free on ONE CU at the same moment -- this 10 us reduction sat 450 us in front of the next build
  hipLaunchKernelGGL(bounds_kernel, dim3(nblocks), dim3(64), 0, s, pos, n, partial);
  hipLaunchKernelGGL(bounds_final_kernel, dim3(1), dim3(64), 0, s, partial, nblocks, out6, hostOut, word, wordOut);
}

void launch_export_u32(const uint32_t *a, const uint32_t *b, const uint32_t *c, const uint32_t *d, const uint32_t *e,
                       uint32_t *hostOut, hipStream_t s) {
  hipLaunchKernelGGL(export_u32_kernel, dim3(1), dim3(64), 0, s, a, b, c, d, e, hostOut);
}

// ---- the ray bundle of an upload of camera beams (Grid::mode 1, bundle_grid.h) ------------
// One workgroup, run once per fit (the host keeps the frame until a planner reports a ray outside it).  The base rays
// start where their camera path enters the medium, so what they share is a point C on their LINES, the sensor's
// pinhole: the least-squares point of sum |(I - d d^T)(C - o)|^2.
// pass 0 (sums, fp64): out = {M = sum (I - d d^T): xx xy xz yy yz zz; b = sum (I - d d^T) o (3); sum d (3); valid base rays}
// pass 1 (frame in g): out = {uMin, uMax, vMin, vMax, min d.A, max |(I - d d^T)(o - C)|^2, min (o - C).d, max (o - C).d}
__global__ __launch_bounds__(1024) void bundle_fit_kernel(const gvpm_camera_ray *__restrict__ rays, uint32_t nsets, int pass, Grid g,
                                                          double *out) {
  constexpr int NV = 13;
  __shared__ double red[NV][16];
  double v[NV];
  // kind of reduction per slot: 0 sum, 1 min, 2 max
  auto kind = [&](int k) { return pass == 0 ? 0 : ((k == 0 || k == 2 || k == 4 || k == 6) ? 1 : (k < 8 ? 2 : 0)); };
  for (int k = 0; k < NV; ++k) v[k] = kind(k) == 0 ? 0.0 : (kind(k) == 1 ? (double)INFINITY : -(double)INFINITY);
  for (uint32_t i = threadIdx.x; i < nsets; i += blockDim.x) {
    const gvpm_camera_ray &r = rays[(size_t)i * 5];
    if (!GVPM_RAY_VALID(r.info)) continue;
    const double dx = r.d[0], dy = r.d[1], dz = r.d[2];
    if (pass == 0) {
      const double ox = r.o[0], oy = r.o[1], oz = r.o[2];
      const double od = ox * dx + oy * dy + oz * dz;
      v[0] += 1.0 - dx * dx; v[1] -= dx * dy; v[2] -= dx * dz;
      v[3] += 1.0 - dy * dy; v[4] -= dy * dz; v[5] += 1.0 - dz * dz;
      v[6] += ox - dx * od; v[7] += oy - dy * od; v[8] += oz - dz * od;
      v[9] += dx; v[10] += dy; v[11] += dz;
      v[12] += 1.0;
    } else {
      const double dA = dx * g.ba[0] + dy * g.ba[1] + dz * g.ba[2];
      const double inv = 1.0 / fmax(dA, 1e-30);
      const double u = (dx * g.bu[0] + dy * g.bu[1] + dz * g.bu[2]) * inv;
      const double w = (dx * g.bv[0] + dy * g.bv[1] + dz * g.bv[2]) * inv;
      const double wx = (double)r.o[0] - g.bo[0], wy = (double)r.o[1] - g.bo[1], wz = (double)r.o[2] - g.bo[2];
      const double sd = wx * dx + wy * dy + wz * dz;
      const double px = wx - sd * dx, py = wy - sd * dy, pz = wz - sd * dz;
      v[0] = fmin(v[0], u); v[1] = fmax(v[1], u);
      v[2] = fmin(v[2], w); v[3] = fmax(v[3], w);
      v[4] = fmin(v[4], dA);
      v[5] = fmax(v[5], px * px + py * py + pz * pz);
      v[6] = fmin(v[6], sd);
      v[7] = fmax(v[7], sd);
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int k = 0; k < NV; ++k) {
    double x = v[k];
    for (int o = 32; o > 0; o >>= 1) {
      const double y = __shfl_xor(x, o, 64);
      x = kind(k) == 0 ? x + y : (kind(k) == 1 ? fmin(x, y) : fmax(x, y));
    }
    if (lane == 0) red[k][wave] = x;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    const int k = threadIdx.x;
    double x = red[k][0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) x = kind(k) == 0 ? x + red[k][w] : (kind(k) == 1 ? fmin(x, red[k][w]) : fmax(x, red[k][w]));
    out[k] = x;
  }
}
void launch_bundle_fit(const gvpm_camera_ray *rays, uint32_t nsets, int pass, const Grid &g, double *out, hipStream_t s) {
  hipLaunchKernelGGL(bundle_fit_kernel, dim3(1), dim3(1024), 0, s, rays, nsets, pass, g, out);
}


void launch_sat(const uint32_t *cellStart, const Grid &g, uint32_t *sat, hipStream_t s) {
  const uint32_t nx1 = g.dim[0] + 1, ny1 = g.dim[1] + 1;
  hipLaunchKernelGGL(sat_y_kernel, dim3((nx1 * g.dim[2] + 255) / 256), dim3(256), 0, s, cellStart, g, sat);
  hipLaunchKernelGGL(sat_z_kernel, dim3((nx1 * ny1 + 255) / 256), dim3(256), 0, s, g, sat);
}

// sub (optional, bundle cells only): CELL_STRIPES x ncells counters, zeroed by the caller like count[]
uint32_t cell_stripes() { return CELL_STRIPES; }
void launch_cell_count(const float *pos, uint32_t n, const Grid &g, uint32_t *keys, uint32_t *rank, uint32_t *count, uint32_t *sub,
                       hipStream_t s, uint32_t *zeroWord, uint32_t *oneWord) {
  if (g.mode != 1) sub = nullptr;
  hipLaunchKernelGGL(cell_count_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pos, n, g, keys, rank, count, sub, zeroWord, oneWord);
  if (sub) hipLaunchKernelGGL(cell_stripes_kernel, dim3((g.ncells + 255) / 256), dim3(256), 0, s, sub, g.ncells, count);
}

void launch_reorder(const gvpm_photon_soa &raw, const uint32_t *keys, const uint32_t *rank, const uint32_t *cellStart,
                    uint32_t n, const gvpm_params &cfg, const float4 *bvh, const float4 *tri4, uint32_t ntri, float dmax,
                    const NearGrid &ng, uint32_t *nearExt, uint32_t extCap, float4 *hot, float4 *cold, uint32_t *overflow,
                    uint32_t *origIdx, const uint32_t *sub, uint32_t ncells, hipStream_t s) {
  RawPhotons r{raw.pos,        raw.wi,         raw.flux,     raw.parent_pos, raw.parent_n,
               raw.prefix_w,   raw.parent_scat, raw.parent_wi, raw.parent_pdf, raw.edge_pdf,
               raw.parent_rr,  raw.parent_g,   raw.flags,    raw.path_id};
#define GVPM_REORDER(M) \
  hipLaunchKernelGGL(reorder_kernel<M>, dim3((n + 63) / 64), dim3(64), 0, s, r, keys, rank, cellStart, n, cfg, bvh, tri4, ntri, \
                     dmax, ng, nearExt, extCap, hot, cold, overflow, origIdx, sub, ncells)
  if (ntri <= 64u) GVPM_REORDER(0);
  else if (ng.start) GVPM_REORDER(2);
  else GVPM_REORDER(1);
#undef GVPM_REORDER
}

void launch_segment_start(const uint32_t *keys, uint32_t n, uint32_t nseg, uint32_t shift, uint32_t *start,
                          hipStream_t s) {
  hipLaunchKernelGGL(segment_start_kernel, dim3((nseg + 1 + 255) / 256), dim3(256), 0, s, keys, n, nseg, shift, start);
}

void launch_beam_count(const gvpm_camera_ray *rays, uint32_t nsets, int width, int tw, int th, uint32_t *keys,
                       uint32_t *rank, uint32_t *count, hipStream_t s) {
  hipLaunchKernelGGL(beam_count_kernel, dim3((nsets + 255) / 256), dim3(256), 0, s, rays, nsets, width, tw, th, keys,
                     rank, count);
}
void launch_beam_scatter(const uint32_t *keys, const uint32_t *rank, const uint32_t *start, uint32_t n,
                         uint32_t *setPerm, hipStream_t s) {
  hipLaunchKernelGGL(beam_scatter_kernel, dim3((n + 255) / 256), dim3(256), 0, s, keys, rank, start, n, setPerm);
}
void launch_tile_start(const uint32_t *start, uint32_t ntiles, uint32_t shift, uint32_t *tileStart, hipStream_t s) {
  hipLaunchKernelGGL(tile_start_kernel, dim3((ntiles + 1 + 255) / 256), dim3(256), 0, s, start, ntiles, shift, tileStart);
}

void launch_beam_keys(const gvpm_camera_ray *rays, uint32_t nsets, int width, int tw, int th, uint32_t *keys,
                      uint32_t *vals, hipStream_t s) {
  hipLaunchKernelGGL(beam_key_kernel, dim3((nsets + 255) / 256), dim3(256), 0, s, rays, nsets, width, tw, th, keys,
                     vals);
}

// ------------------------------------------------------------------------------------------
// The G-BRE build as ONE chain of six launches (round 6; until round 5: twenty-three -- bounds x 2, four memsets, count,
// scan x 3, summed-volume table x 2, scatter, beam count, scan x 3, beam scatter, tile starts, memset, planner, export --
// 350 us of work that took 860 us of wall beside the other streams' kernels, a launch boundary and a ramp each):
//   K1 count    photons -> cell keys + ranks, per-block bounds | beam sets -> tile keys + ranks | the control words of the
//               kernels behind the chain (queue heads, near-list cursor) initialised
//   K2 reduce   block sums of ONE scan over the concatenated counters [cells | beam keys]; the LAST block to arrive scans
//               the block sums and reduces the photons' bounds (no block waits for another: see exclusiveSumU32)
//   K3 down     the scan; the counters are handed back ZEROED (no memset in the next chain)
//   K4 mid      summed-volume table pass Y | beam scatter | tile starts
//   K5 sat_z    summed-volume table pass Z
//   K6 tail     the planner | the photon scatter (whole records + near-occluder lists) side by side -- neither reads the other's
//               output, the planner is a few thousand latency-bound waves and the scatter is bandwidth: alone 126 + 112 us
//               one after the other; the last block to arrive exports the counters the host sizes the pair buffer with
// ------------------------------------------------------------------------------------------

// True in exactly one block of the launch: the last of `nblocks` to get here.  ctl: 65 words, zero on entry, zero again on
// exit.  Arrivals are spread over up to 64 counters (same-address atomics retire ~11 ns apart: 20 000 blocks on one word
// would take longer than the kernel).  NO cache write-back rides on an arrival (a release fence is a write-back of the
// XCD's whole L2: with one per block the tail kernel, which dirties 144 MB, took 800 us instead of 150): what the last
// block reads of the others must have been written by device-scope ATOMICS (performed at the memory side, past the
// per-XCD L2s) and is read by atomics; every lane waits for its own to be acknowledged before the block arrives.
__device__ __forceinline__ bool lastBlockArrives(uint32_t *ctl, uint32_t blockId, uint32_t nblocks) {
  __shared__ uint32_t isLast;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t S = nblocks < 64u ? nblocks : 64u;
    const uint32_t sl = blockId % S;
    const uint32_t expect = (nblocks - sl + S - 1u) / S;  // blocks with this residue
    uint32_t last = 0u;
    if (atomicAdd(&ctl[1u + sl], 1u) == expect - 1u) {
      atomicExch(&ctl[1u + sl], 0u);
      if (atomicAdd(&ctl[0], 1u) == S - 1u) {
        atomicExch(&ctl[0], 0u);
        last = 1u;
      }
    }
    isLast = last;
  }
  __syncthreads();
  return isLast != 0u;
}
__device__ __forceinline__ uint32_t coherentRead(uint32_t *p) { return atomicAdd(p, 0u); }

// photon bounds through 64 x 6 buckets of order-preserving unsigned codes (min slots start at 0xFFFFFFFF, max slots at 0):
// a block's six numbers go to bucket (block % 64) by atomicMin / atomicMax -- sixty atomics an address -- and whoever reduces
// the buckets at the end reads 384 words instead of six per block, and resets them.
__device__ __forceinline__ uint32_t ordCode(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u >> 31) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordDecode(uint32_t o) { return __uint_as_float((o >> 31) ? (o & 0x7FFFFFFFu) : ~o); }
__global__ void chain_init_buckets_kernel(uint32_t *buckets) {
  const uint32_t t = threadIdx.x;  // 768 threads: the photons' buckets, then the camera beams'
  buckets[t] = (t % 6u) < 3u ? 0xFFFFFFFFu : 0u;
}

// K1.  Mode 0 and the beam sets only COUNT here (atomics without a return value: nothing waits for them); the arrival rank
// is taken where the record is scattered, by counting the cell's counter back down to zero (chain_tail / chain_mid) -- which
// also hands the counters back zeroed.  Bundle cells keep cell_count_kernel's striped ranks.
__global__ __launch_bounds__(256) void chain_count_kernel(ChainArgs c) {
  __shared__ float red[6][4];
  const uint32_t b = blockIdx.x;
  if (b == 0) {
    if (threadIdx.x < 8) c.queueCtl[threadIdx.x] = 0u;
    if (threadIdx.x == 8) *c.overflowCtr = 0u;
    if (threadIdx.x == 9) c.nearExt[0] = 1u;
  }
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  const bool photons = b < c.nPhBlocks;
  if (photons) {
    const uint32_t i = b * 256u + threadIdx.x;
    if (i < c.n) {
      const float px = c.pos[3 * (size_t)i + 0], py = c.pos[3 * (size_t)i + 1], pz = c.pos[3 * (size_t)i + 2];
      mn[0] = px; mn[1] = py; mn[2] = pz;
      mx[0] = px; mx[1] = py; mx[2] = pz;
      const Grid &g = c.g;
      if (g.mode == 1) {
        // (bundle cells: 16 striped counters a cell -- every photon around a light the camera sees shares a handful of
        // cells --, counted here, turned into prefixes by chain_stripes_kernel and counted back down by the scatter)
        const uint32_t k = bundlePhotonKey(g, px, py, pz);
        c.keys[i] = k;
        if (k == GVPM_BUNDLE_DUMP_CELL(g.dim[0])) {
          c.rank[i] = 0xFFFFFFFFu;  // a photon no ray of the bundle can meet is not sorted at all
        } else {
          c.rank[i] = 0u;
          (void)atomicAdd(&c.sub[(size_t)(i % CELL_STRIPES) * g.ncells + k], 1u);
        }
      } else {
        const int cx = cellCoord(px, g.org[0], g.invCell, g.dim[0]);
        const int cy = cellCoord(py, g.org[1], g.invCell, g.dim[1]);
        const int cz = cellCoord(pz, g.org[2], g.invCell, g.dim[2]);
        const uint32_t k = ((uint32_t)cz * g.dim[1] + cy) * g.dim[0] + cx;
        c.keys[i] = k;
        (void)atomicAdd(&c.counts[k], 1u);
      }
    }
  } else {
    // the beam sets: key = tile * tilePixels + pixel in tile, counted behind the cells (the sets of one pixel -- one per
    // medium edge of its camera path -- keep no order among themselves: nothing reads one).  Their base rays' segments give
    // the region the NEXT build's grid is clipped to (gather_drivers.hip, buildGrid).
    const uint32_t i = (b - c.nPhBlocks) * 256u + threadIdx.x;
    if (i < c.nsets) {
      const gvpm_camera_ray &r = c.rays[(size_t)i * 5];
      const uint32_t px = r.pixel & 0xFFFFu, py = r.pixel >> 16;
      const uint32_t tilesX = (c.width + c.tw - 1) / c.tw;
      const uint32_t tile = (py / c.th) * tilesX + px / c.tw;
      const uint32_t inTile = (py % c.th) * c.tw + px % c.tw;
      const uint32_t k = tile * (uint32_t)(c.tw * c.th) + inTile;
      c.bKeys[i] = k;
      (void)atomicAdd(&c.counts[(size_t)c.beamOff + k], 1u);
      if (GVPM_RAY_VALID(r.info)) {
#pragma unroll
        for (int k3 = 0; k3 < 3; ++k3) {
          const float e0 = r.o[k3], e1 = r.o[k3] + r.d[k3] * r.len;
          mn[k3] = fminf(e0, e1);
          mx[k3] = fmaxf(e0, e1);
        }
      }
    }
  }
  const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float lo = wave_min(mn[k]), hi = wave_max(mx[k]);
    if (lane == 0) {
      red[k][wave] = lo;
      red[3 + k][wave] = hi;
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    float v = red[k][0];
    for (int w = 1; w < 4; ++w) v = k < 3 ? fminf(v, red[k][w]) : fmaxf(v, red[k][w]);
    uint32_t *slot = c.buckets + (photons ? 0u : 384u) + (b % 64u) * 6u + k;
    if (k < 3) (void)atomicMin(slot, ordCode(v));
    else (void)atomicMax(slot, ordCode(v));
  }
}

// (bundle cells) the stripes of a cell's counter -> exclusive prefixes within the cell (a second array: the counters stay, the
// scatter counts them back down to zero), the cell's total to counts[]
__global__ __launch_bounds__(256) void chain_stripes_kernel(const uint32_t *__restrict__ sub, uint32_t *__restrict__ prefix, uint32_t ncells,
                                                            uint32_t *__restrict__ count) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= ncells) return;
  uint32_t run = 0;
#pragma unroll
  for (uint32_t st = 0; st < CELL_STRIPES; ++st) {
    prefix[(size_t)st * ncells + k] = run;
    run += sub[(size_t)st * ncells + k];
  }
  count[k] = run;
}

// K2: block sums; the last block to arrive turns them into block offsets and reduces the bounds' buckets
__global__ __launch_bounds__(256) void chain_scan_reduce_kernel(ChainArgs c) {
  __shared__ uint32_t lds[4];
  __shared__ float red[6][4];
  const uint32_t n = c.scanLen;
  {
    const uint32_t *__restrict__ counts = c.counts;
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
    uint32_t v = 0;
#pragma unroll
    for (uint32_t k = 0; k < SCAN_PER_THREAD; ++k)
      if (base + k < n) v += counts[base + k];
    uint32_t total;
    (void)block_scan_excl(v, lds, total);
    if (threadIdx.x == 0) (void)atomicExch(&c.blockSum[blockIdx.x], total);
  }
  if (!lastBlockArrives(c.ctl, blockIdx.x, gridDim.x)) return;
  // the spine, by whoever came last: every thread takes a run of consecutive block sums (all its reads in flight at once)
  const uint32_t nblocks = gridDim.x;
  const uint32_t per = (nblocks + SCAN_BLOCK - 1u) / SCAN_BLOCK;
  const uint32_t i0 = min(nblocks, threadIdx.x * per), i1 = min(nblocks, i0 + per);
  uint32_t mine = 0;
  for (uint32_t i = i0; i < i1; ++i) mine += coherentRead(&c.blockSum[i]);
  uint32_t total;
  uint32_t run = block_scan_excl(mine, lds, total);
  for (uint32_t i = i0; i < i1; ++i) {
    const uint32_t v = coherentRead(&c.blockSum[i]);
    c.blockSum[i] = run;
    run += v;
  }
  // ... and the bounds: the photons', then the camera beams'
  for (uint32_t which = 0; which < 2u; ++which) {
    float v6[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      float v = k < 3 ? INFINITY : -INFINITY;
      if (threadIdx.x < 64u) {
        uint32_t *slot = c.buckets + which * 384u + threadIdx.x * 6u + k;
        v = ordDecode(coherentRead(slot));
        *slot = k < 3 ? 0xFFFFFFFFu : 0u;  // handed back reset
      }
      v6[k] = v;
    }
    const int wave = threadIdx.x / 64, lane = threadIdx.x % 64;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float a = wave_min(v6[k]), b = wave_max(v6[3 + k]);
      if (lane == 0) {
        red[k][wave] = a;
        red[3 + k][wave] = b;
      }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
      const int k = threadIdx.x;
      float v = red[k][0];
      for (int w = 1; w < 4; ++w) v = k < 3 ? fminf(v, red[k][w]) : fmaxf(v, red[k][w]);
      c.out6[which * 6u + k] = v;
      if (c.hostB6) c.hostB6[which * 16u + k] = v;
    }
  }
}

// K3: the scan.  (Bundle cells: the cells' counters, written by cell_stripes_kernel, are zeroed here -- everything else is
// counted back down by the scatters.)
__global__ __launch_bounds__(256) void chain_scan_down_kernel(ChainArgs c) {
  __shared__ uint32_t lds[4];
  const uint32_t n = c.scanLen;
  uint32_t *__restrict__ counts = c.counts;
  uint32_t *__restrict__ starts = c.starts;
  const uint32_t zeroBelow = c.g.mode == 1 ? c.beamOff : 0u;
  const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
  uint32_t x[SCAN_PER_THREAD], v = 0;
#pragma unroll
  for (uint32_t k = 0; k < SCAN_PER_THREAD; ++k) {
    x[k] = base + k < n ? counts[base + k] : 0u;
    v += x[k];
  }
  uint32_t total;
  uint32_t run = c.blockSum[blockIdx.x] + block_scan_excl(v, lds, total);
#pragma unroll
  for (uint32_t k = 0; k < SCAN_PER_THREAD; ++k) {
    if (base + k < n) {
      starts[base + k] = run;
      if (x[k] && base + k < zeroBelow) counts[base + k] = 0u;
    }
    run += x[k];
  }
}

// K4: three small jobs that need nothing but the scan
__device__ __forceinline__ void satYBody(const uint32_t *__restrict__ cellStart, const Grid &g, uint32_t *__restrict__ sat, uint32_t t) {
  const uint32_t nx1 = g.dim[0] + 1, ny1 = g.dim[1] + 1;
  if (t >= nx1 * (uint32_t)g.dim[2]) return;
  const uint32_t x = t % nx1, z = t / nx1;
  uint32_t run = 0;
  sat[((size_t)z * ny1 + 0) * nx1 + x] = 0u;
#pragma unroll 8
  for (int y = 0; y < g.dim[1]; ++y) {
    const size_t row = ((size_t)z * g.dim[1] + y) * g.dim[0];
    run += cellStart[row + x] - cellStart[row];
    sat[((size_t)z * ny1 + (y + 1)) * nx1 + x] = run;
  }
}
__global__ __launch_bounds__(256) void chain_mid_kernel(ChainArgs c, uint32_t nSatBlocks, uint32_t nBeamBlocks) {
  const uint32_t b = blockIdx.x;
  if (b < nSatBlocks) {
    satYBody(c.starts, c.g, c.sat, b * 256u + threadIdx.x);
    return;
  }
  const uint32_t *__restrict__ starts = c.starts;
  const uint32_t base = starts[c.beamOff];  // every sorted photon lies before the first beam key
  if (b < nSatBlocks + nBeamBlocks) {
    const uint32_t i = (b - nSatBlocks) * 256u + threadIdx.x;
    if (i < c.nsets) {
      const size_t k = (size_t)c.beamOff + c.bKeys[i];
      c.setPerm[starts[k] - base + (atomicSub(&c.counts[k], 1u) - 1u)] = i;
    }
    return;
  }
  const uint32_t t = (b - nSatBlocks - nBeamBlocks) * 256u + threadIdx.x;
  if (t <= c.ntiles) c.tileStart[t] = starts[(size_t)c.beamOff + ((size_t)t << c.tileShift)] - base;
}

struct TailArgs {
  // planner
  uint32_t ntiles, target, itemCap, nPlan;
  uint4 *items;
  uint2 *itemOff;
  uint32_t *itemCount, *blockTotal;
  // scatter
  RawPhotons raw;
  const uint32_t *keys, *rank, *cellStart, *sub;
  uint32_t *counts;  // mode 0: the cells' counters, counted back down for the arrival ranks
  uint32_t *subCount;  // bundle cells: the striped counters, counted back down (their prefixes: `sub`)
  uint32_t n, ncells, extCap;
  uint32_t every;    // every `every`-th block of the launch is a planner's, until they are nPlan
  float dmax;
  NearGrid ng;
  uint32_t *nearExt, *overflow, *origIdx;
  float4 *hot, *cold;
  // export
  uint32_t *ctl;
  const uint32_t *bundleFlag;
  uint32_t *hostOut;
  // The guard of an OPTIMISTIC step (gather_drivers.hip, gatherBRE): traversal and evaluation are queued behind this launch
  // before the host has seen its counters, sized for what the buffers hold.  The last block compares and leaves a status in
  // itemCount[7]; non-zero, both kernels return at once and the host, which reads the same word, queues them again.
  uint32_t pairCapBlocks;  // 64-entry blocks the pair buffer holds (0xFFFFFFFF: not optimistic, no guard)
  uint32_t unitCap, unitPairs;  // entries of each unit list (0: none), pairs per unit
  uint32_t fullVis;        // the evaluation queued walks the BVH (near-list overflows do not matter)
  uint32_t planGroups;     // the 3D grid's planner walks 64 / B chunks side by side (GVPM_PLAN_GROUPS=0: one at a time)
};
constexpr uint32_t TAIL_PLAN_STAGE = 128;
template <int B> constexpr size_t tailLdsBytes() {
  constexpr size_t pl = sizeof(PlanLds<B, TAIL_PLAN_STAGE>) > sizeof(PlanLdsQ<B, TAIL_PLAN_STAGE>) ? sizeof(PlanLds<B, TAIL_PLAN_STAGE>)
                                                                                                  : sizeof(PlanLdsQ<B, TAIL_PLAN_STAGE>);
  return pl > sizeof(ReorderLds) ? pl : sizeof(ReorderLds);
}
#ifndef GVPM_TAIL_WPE
#define GVPM_TAIL_WPE 6  // waves per SIMD the tail is compiled for (registers: 512 / this)
#endif
template <int B, int MODE> __global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GVPM_TAIL_WPE, GVPM_TAIL_WPE))) void chain_tail_kernel(GatherArgs a, TailArgs t) {
  __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[tailLdsBytes<B>()];
  // The two roles INTERLEAVED over the launch (workgroups are dispatched in index order): with the planner's blocks first,
  // they alone fill the chip -- a few thousand single-wave workgroups of 11 KB -- and the scatter starts when they end.
  const uint32_t b = blockIdx.x, pslot = b / t.every;
  if (b % t.every == 0u && pslot < t.nPlan) {
    // (the 3D grid: 64 / B chunks side by side, tile_walk.h planBodyQ; bundle cells: one chunk at a time)
    if (a.grid.mode != 1 && t.planGroups)
      planBodyQ<B, TAIL_PLAN_STAGE>(a, t.ntiles, t.target, t.items, t.itemCount, t.itemOff, t.blockTotal, t.itemCap, pslot, t.nPlan,
                                    *reinterpret_cast<PlanLdsQ<B, TAIL_PLAN_STAGE> *>(ldsRaw));
    else
      planBody<B, TAIL_PLAN_STAGE>(a, t.ntiles, t.target, t.items, t.itemCount, t.itemOff, t.blockTotal, t.itemCap, pslot, t.nPlan,
                                   *reinterpret_cast<PlanLds<B, TAIL_PLAN_STAGE> *>(ldsRaw));
  }
  else
    reorderBody<MODE>(t.raw, t.keys, t.rank, t.cellStart, t.n, a.cfg, a.bvh, a.tri4, a.ntri, t.dmax, t.ng, t.nearExt, t.extCap, t.hot,
                      t.cold, t.overflow, t.origIdx, t.sub, t.ncells, b - min(t.nPlan, (b + t.every - 1u) / t.every), t.counts,
                      t.subCount, *reinterpret_cast<ReorderLds *>(ldsRaw));
  if (!lastBlockArrives(t.ctl, blockIdx.x, gridDim.x)) return;
  if (threadIdx.x == 0) {
    // what export_u32_kernel handed the host: pair blocks, overflowed near lists, extension cursor, items, bundle flag
    // (all five are only ever touched by atomics in this launch)
    t.hostOut[0] = coherentRead(t.blockTotal);
    t.hostOut[1] = coherentRead(t.overflow);
    t.hostOut[2] = coherentRead(t.nearExt);
    t.hostOut[3] = coherentRead(t.itemCount);
    t.hostOut[4] = t.bundleFlag ? coherentRead(const_cast<uint32_t *>(t.bundleFlag)) : 0u;
    uint32_t bad = 0u;
    if (t.pairCapBlocks != 0xFFFFFFFFu) {
      const uint32_t blocks = t.hostOut[0], nIt = t.hostOut[3];
      if (blocks > t.pairCapBlocks) bad |= 1u;
      if (nIt > t.itemCap) bad |= 2u;
      if (t.unitCap && (unsigned long long)blocks * 64ull / t.unitPairs + nIt + 64ull > (unsigned long long)t.unitCap) bad |= 4u;
      if (t.hostOut[1] != 0u && !t.fullVis) bad |= 8u;
      if (t.hostOut[4] != 0u) bad |= 16u;
    }
    t.itemCount[7] = bad;
    t.hostOut[5] = bad;
    __threadfence_system();
  }
}

template <int B>
static void launchTail(int mode, uint32_t grid, hipStream_t s, const GatherArgs &a, const TailArgs &t) {
  if (mode == 0) hipLaunchKernelGGL((chain_tail_kernel<B, 0>), dim3(grid), dim3(64), 0, s, a, t);
  else if (mode == 2) hipLaunchKernelGGL((chain_tail_kernel<B, 2>), dim3(grid), dim3(64), 0, s, a, t);
  else hipLaunchKernelGGL((chain_tail_kernel<B, 1>), dim3(grid), dim3(64), 0, s, a, t);
}

void launch_build_chain(ChainArgs c, const GatherArgs &a, const gvpm_photon_soa &raw, int beamsPerWave, uint32_t target, uint4 *items,
                        uint2 *itemOff, uint32_t itemCap, float dmax, const NearGrid &ng, uint32_t extCap, uint32_t *origIdx,
                        uint32_t *hostOut, bool initBuckets, hipStream_t s, uint32_t pairCapBlocks, uint32_t unitCap, uint32_t unitPairs,
                        bool fullVis) {
  c.nPhBlocks = (c.n + 255u) / 256u;
  c.nBeamBlocks = (c.nsets + 255u) / 256u;
  if (initBuckets) hipLaunchKernelGGL(chain_init_buckets_kernel, dim3(1), dim3(768), 0, s, c.buckets);
  hipLaunchKernelGGL(chain_count_kernel, dim3(c.nPhBlocks + c.nBeamBlocks), dim3(256), 0, s, c);
  // (bundle cells: the stripes of a cell's counter -> exclusive prefixes within the cell, its total to counts[])
  if (c.g.mode == 1)
    hipLaunchKernelGGL(chain_stripes_kernel, dim3((c.g.ncells + 255) / 256), dim3(256), 0, s, c.sub, c.sub + (size_t)CELL_STRIPES * c.g.ncells,
                       c.g.ncells, c.counts);
  const uint32_t nScan = (c.scanLen + SCAN_TILE - 1) / SCAN_TILE;
  hipLaunchKernelGGL(chain_scan_reduce_kernel, dim3(nScan), dim3(SCAN_BLOCK), 0, s, c);
  hipLaunchKernelGGL(chain_scan_down_kernel, dim3(nScan), dim3(SCAN_BLOCK), 0, s, c);
  const uint32_t nx1 = c.g.dim[0] + 1, ny1 = c.g.dim[1] + 1;
  const uint32_t nSat = (nx1 * (uint32_t)c.g.dim[2] + 255u) / 256u, nTile = (c.ntiles + 1u + 255u) / 256u;
  hipLaunchKernelGGL(chain_mid_kernel, dim3(nSat + c.nBeamBlocks + nTile), dim3(256), 0, s, c, nSat, c.nBeamBlocks);
  hipLaunchKernelGGL(sat_z_kernel, dim3((nx1 * ny1 + 255) / 256), dim3(256), 0, s, c.g, c.sat);
  TailArgs t{};
  t.ntiles = c.ntiles;
  t.target = target;
  t.itemCap = itemCap;
  t.planGroups = getenv("GVPM_PLAN_GROUPS") ? (atoi(getenv("GVPM_PLAN_GROUPS")) != 0) : 1u;
  {
    // planner waves: a prime stride over the tiles (launch_plan_bre); with 64 / B tiles a wave and trip, as many fewer waves
    const uint32_t per = (t.planGroups && c.g.mode != 1) ? (uint32_t)(64 / (beamsPerWave > 0 ? beamsPerWave : 16)) : 1u;
    const uint32_t want = (c.ntiles + per - 1u) / per;
    t.nPlan = want < 4093u ? std::max(1u, want) : 4093u;
    (void)per;
  }
  t.items = items;
  t.itemOff = itemOff;
  t.itemCount = c.queueCtl;
  t.blockTotal = c.queueCtl + 3;
  t.raw = RawPhotons{raw.pos,       raw.wi,          raw.flux,      raw.parent_pos, raw.parent_n, raw.prefix_w, raw.parent_scat,
                     raw.parent_wi, raw.parent_pdf,  raw.edge_pdf,  raw.parent_rr,  raw.parent_g, raw.flags,    raw.path_id};
  t.keys = c.keys;
  t.rank = c.rank;
  t.cellStart = c.starts;
  t.sub = c.g.mode == 1 ? c.sub + (size_t)CELL_STRIPES * c.g.ncells : nullptr;
  t.subCount = c.g.mode == 1 ? c.sub : nullptr;
  t.counts = c.g.mode == 1 ? nullptr : c.counts;
  t.n = c.n;
  t.ncells = c.g.ncells;
  t.extCap = extCap;
  t.dmax = dmax;
  t.ng = ng;
  t.nearExt = c.nearExt;
  t.overflow = c.overflowCtr;
  t.origIdx = origIdx;
  t.hot = const_cast<float4 *>(a.hot);
  t.cold = const_cast<float4 *>(a.cold);
  t.ctl = c.ctl + 65;
  t.bundleFlag = a.bundleFlag;
  t.hostOut = hostOut;
  t.pairCapBlocks = pairCapBlocks;
  t.unitCap = unitCap;
  t.unitPairs = unitPairs ? unitPairs : 1u;
  t.fullVis = fullVis ? 1u : 0u;
  const int mode = a.ntri <= 64u ? 0 : (ng.start ? 2 : 1);
  const uint32_t grid = t.nPlan + (c.n + 63u) / 64u;
  t.every = std::max(1u, grid / std::max(1u, t.nPlan));
  switch (beamsPerWave) {
    case 64: launchTail<64>(mode, grid, s, a, t); break;
    case 32: launchTail<32>(mode, grid, s, a, t); break;
    default: launchTail<16>(mode, grid, s, a, t); break;
  }
}

}  // namespace gvpm
