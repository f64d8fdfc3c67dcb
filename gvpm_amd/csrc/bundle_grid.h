// The "bundle" grid of the G-BRE build (Grid::mode == 1, device_types.h): cells over the (u, v) plane of a single-origin
// ray bundle, by levels of the photons' angular size.  Two functions define it; both are conservative by construction
// (a photon is in the candidate set of every tile one of whose rays can pass within the radius of it), so the evaluated
// set does not depend on which grid was used -- tests/test_parity_gpu.py runs both.
#pragma once
#include <hip/hip_runtime.h>

#include "device_types.h"

namespace gvpm {

// Cell layout: a dense G x G x 2 array, so that the counting sort, the summed-volume table and the box ranges of the 3D
// grid apply as they are.  Layer 0 is level 0; layer 1 holds the levels l >= 1 side by side along x, level l in columns
// [G - (G >> (l - 1)), +(G >> l)) and rows [0, G >> l); its far corner cell (G - 1, G - 1) is the dump: photons no ray
// of the bundle can meet (no box reaches it).
__host__ __device__ __forceinline__ int bundleLevelX(int G, int l) { return l == 0 ? 0 : G - (G >> (l - 1)); }
__host__ __device__ __forceinline__ uint32_t bundleCell(int G, int l, int cu, int cv) {
  return ((uint32_t)(l ? 1 : 0) * (uint32_t)G + (uint32_t)cv) * (uint32_t)G + (uint32_t)(bundleLevelX(G, l) + cu);
}
#define GVPM_BUNDLE_DUMP_CELL(G) ((((uint32_t)(G) + (uint32_t)(G) - 1u) * (uint32_t)(G)) + (uint32_t)(G) - 1u)

// Cell key of a photon.
// A ray with (u, v) passes within r of p only if some point q = O + t d, t > 0, of its line from the shared point O lies
// in the box p +- r (the ray itself is a piece of that half line): with
// c = p - O in the (A, U, V) frame, q_A in (0, c_A + r] and q_U in [c_U - r, c_U + r], so u = q_U / q_A lies between the
// quotients of the interval ends (q_A >= max(c_A - r, tiny): an origin inside or beside the sphere gives an unbounded,
// i.e. clamped, rectangle).  The rectangle, clipped to the bundle's range, is filed under its midpoint at the level whose
// cell size covers its half extent.
__device__ __forceinline__ uint32_t bundlePhotonKey(const Grid &g, float px, float py, float pz, int *level = nullptr) {
  const float cx = px - g.bo[0], cy = py - g.bo[1], cz = pz - g.bo[2];
  const float cA = cx * g.ba[0] + cy * g.ba[1] + cz * g.ba[2];
  const float cU = cx * g.bu[0] + cy * g.bu[1] + cz * g.bu[2];
  const float cV = cx * g.bv[0] + cy * g.bv[1] + cz * g.bv[2];
  // (radius fattened by the fp32 error of the frame coordinates: a few ulps of |c|)
  // (... and by how far from the shared point the rays' lines may pass)
  const float r = g.radius * 1.0001f + 4e-7f * (fabsf(cx) + fabsf(cy) + fabsf(cz)) + g.lineTol;
  const float a1 = cA + r;
  if (!(a1 > 0.f)) return GVPM_BUNDLE_DUMP_CELL(g.dim[0]);  // behind the origin by more than the radius
  const float a0 = fmaxf(cA - r, 1e-30f);
  const float i0 = 1.f / a0, i1 = 1.f / a1;
  const float bu0 = cU - r, bu1 = cU + r, bv0 = cV - r, bv1 = cV + r;
  float uLo = bu0 >= 0.f ? bu0 * i1 : bu0 * i0, uHi = bu1 >= 0.f ? bu1 * i0 : bu1 * i1;
  float vLo = bv0 >= 0.f ? bv0 * i1 : bv0 * i0, vHi = bv1 >= 0.f ? bv1 * i0 : bv1 * i1;
  // (relative slack for the quotients' rounding)
  const float eu = 2e-6f * (fabsf(uLo) + fabsf(uHi)) + 1e-7f, ev = 2e-6f * (fabsf(vLo) + fabsf(vHi)) + 1e-7f;
  uLo -= eu; uHi += eu; vLo -= ev; vHi += ev;
  if (uHi < g.uMin || uLo > g.uMax || vHi < g.vMin || vLo > g.vMax) return GVPM_BUNDLE_DUMP_CELL(g.dim[0]);
  uLo = fmaxf(uLo, g.uMin); uHi = fminf(uHi, g.uMax);
  vLo = fmaxf(vLo, g.vMin); vHi = fminf(vHi, g.vMax);
  const float mu = 0.5f * (uLo + uHi), mv = 0.5f * (vLo + vHi);
  const float half = fmaxf(uHi - mu, fmaxf(mu - uLo, fmaxf(vHi - mv, mv - vLo))) * 1.00001f;
  const int L = g.levels, G = g.dim[0];
  // the smallest level whose cell size s0 * 2^l covers the half extent
  int l = 0;
  const float q = half * g.invS0;
  if (q > 1.f) l = min(ilogbf(q) + 1, L - 1);
  if (level) *level = l;
  const int Gl = G >> l;
  const float inv = g.invS0 / (float)(1 << l);
  const int cu = min(max((int)floorf((mu - g.uMin) * inv), 0), Gl - 1);
  const int cv = min(max((int)floorf((mv - g.vMin) * inv), 0), Gl - 1);
  return bundleCell(G, l, cu, cv);
}

// (u, v) of a ray of the bundle; ok: it belongs to it (its line passes the shared point, which lies behind its start,
// and it looks along A)
__device__ __forceinline__ bool bundleRayUV(const Grid &g, float ox, float oy, float oz, float dx, float dy, float dz, float &u, float &v) {
  const float dA = dx * g.ba[0] + dy * g.ba[1] + dz * g.ba[2];
  const float wx = ox - g.bo[0], wy = oy - g.bo[1], wz = oz - g.bo[2];
  const float sd = wx * dx + wy * dy + wz * dz;
  const float px = wx - sd * dx, py = wy - sd * dy, pz = wz - sd * dz;
  // (this fp32 evaluation of the distance adds ~1e-7 |w| to it: the host's lineTol has that margin)
  const bool through = px * px + py * py + pz * pz <= g.lineTol * g.lineTol && sd >= -g.lineTol;
  const float inv = 1.f / fmaxf(dA, 1e-30f);
  u = (dx * g.bu[0] + dy * g.bu[1] + dz * g.bu[2]) * inv;
  v = (dx * g.bv[0] + dy * g.bv[1] + dz * g.bv[2]) * inv;
  return through && dA > 0.05f && u >= g.uMin && u <= g.uMax && v >= g.vMin && v <= g.vMax;
}

// Cell box of a tile at level l: the cells that hold the photons of that level within one cell size of the tile's
// rectangle [u0, u1] x [v0, v1] (a photon of level l is filed under a point within s_l of every (u, v) that meets it).
struct BundleBox {
  int x0, x1, y0, y1, z;
};
__device__ __forceinline__ BundleBox bundleTileBox(const Grid &g, int l, float u0, float u1, float v0, float v1) {
  const int G = g.dim[0];
  const int Gl = max(G >> l, 1);
  const float s = g.s0 * (float)(1 << l);
  const float inv = g.invS0 / (float)(1 << l);
  const float pad = s * 1.0001f + 1e-6f * (fabsf(u0) + fabsf(u1) + fabsf(v0) + fabsf(v1) + 1.f);
  BundleBox b;
  b.x0 = min(max((int)floorf((u0 - pad - g.uMin) * inv), 0), Gl - 1);
  b.x1 = min(max((int)floorf((u1 + pad - g.uMin) * inv), 0), Gl - 1);
  b.y0 = min(max((int)floorf((v0 - pad - g.vMin) * inv), 0), Gl - 1);
  b.y1 = min(max((int)floorf((v1 + pad - g.vMin) * inv), 0), Gl - 1);
  const int xo = bundleLevelX(G, l);
  b.x0 += xo;
  b.x1 += xo;
  b.z = l ? 1 : 0;
  return b;
}

}  // namespace gvpm
