// Tile traversal shared by the beam-vs-point gather kernels (gather_bre.hip, gather_beams.hip):
// LDS layout of a tile of camera-beam sets, the slab walk over the sorted uniform grid and the
// planner that cuts tiles into work items of equal candidate count.  See gather_bre.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "bundle_grid.h"
#include "device_types.h"
#include "shift_device.h"
#include "vec.h"

namespace gvpm {

constexpr int STAGE = 256;
constexpr int QCAP = 128;
constexpr int PLAN_MAX_ITEMS = 256;  // per tile chunk

// the camera-beam sets of a tile: base + 4 shifted rays each
template <int B> struct RayTile {
  float4 ray4[5][3][B];  // {o, len (<0: invalid)} {d, pdf} {eye, jacobian}
  float gop[5][B];
  float rnd[B];
  uint32_t pix[B];
  uint32_t edge[B];
};

template <int B> struct TileLds : RayTile<B> {
  double acc[27][B];  // double: ds_add_f64 runs ~25x the rate of ds_add_f32 on gfx950 (scripts/probes/lds_atomics_bench.hip)
  float4 stage[STAGE];
  uint32_t stageIdx[STAGE];
  uint2 queue[QCAP];
};

template <int B> __device__ __forceinline__ RayReg loadRay(const RayTile<B> &s, int k, int b) {
  RayReg r;
  const float4 q0 = s.ray4[k][0][b], q1 = s.ray4[k][1][b], q2 = s.ray4[k][2][b];
  r.o = mk3(q0.x, q0.y, q0.z);
  r.len = fabsf(q0.w);
  r.valid = q0.w >= 0.f;
  r.d = mk3(q1.x, q1.y, q1.z);
  r.pdf = q1.w;
  r.eye = mk3(q2.x, q2.y, q2.z);
  r.jac = q2.w;
  r.gop = s.gop[k][b];
  return r;
}

// the shifted rays of a freshly loaded tile, in place: {o, len} -> {o_s - o_b, len}, {d, pdf} -> {d_s - d_b, sensorMIS},
// {eye, jacobian} stays.  The differences are formed in fp64 and are exact to fp32; the base rays stay absolute.
template <int B> __device__ __forceinline__ void relToBase(RayTile<B> &s, int lane) {
  for (int idx = lane; idx < 4 * B; idx += 64) {
    const int i = idx / B, bb = idx % B;
    const RayReg br = loadRay(s, 0, bb), sr = loadRay(s, 1 + i, bb);
    const f3 dO = tof(tod(sr.o) - tod(br.o)), dD = tof(tod(sr.d) - tod(br.d));
    const float lenSigned = s.ray4[1 + i][0][bb].w;  // the valid bit rides on its sign
    const float sm = sensorMIS(sr, br, s.edge[bb]);
    s.ray4[1 + i][0][bb] = make_float4(dO.x, dO.y, dO.z, lenSigned);
    s.ray4[1 + i][1][bb] = make_float4(dD.x, dD.y, dD.z, sm);
  }
}
struct ShiftRel {
  f3 ro, rd, d, eye;  // o_s - o_b, d_s - d_b, d_s, eyeContrib
  float len, sMIS;
  bool valid;
};
template <int B> __device__ __forceinline__ ShiftRel loadShiftRel(const RayTile<B> &s, int i, int b, f3 baseD) {
  ShiftRel r;
  const float4 q0 = s.ray4[1 + i][0][b], q1 = s.ray4[1 + i][1][b], q2 = s.ray4[1 + i][2][b];
  r.ro = mk3(q0.x, q0.y, q0.z);
  r.len = fabsf(q0.w);
  r.valid = q0.w >= 0.f;
  r.rd = mk3(q1.x, q1.y, q1.z);
  r.sMIS = q1.w;
  r.d = baseD + r.rd;
  r.eye = mk3(q2.x, q2.y, q2.z);
  return r;
}
// ------------------------------------------------------------------------------------------
// Tile traversal state shared by the planner and the gather kernel
// ------------------------------------------------------------------------------------------
struct TileWalk {
  // per lane
  RayReg base;
  bool beamValid;
  float oA, dA, oU, dU, oV, dV, t0, t1;
  // wave-uniform
  int A, cA0, cA1, K;
  float orgA, orgU, orgV, pad;
  int dimA, dimU, dimV;
  bool any;
  // bundle grid (Grid::mode == 1): the (u, v) rectangle of the tile's rays; a "slab step" is a level
  float bu0, bu1, bv0, bv1;
  bool bundleBad;  // a valid ray of the tile is not of the bundle the grid was built for: the host rebuilds in 3D
};

// (no barrier at the end: for kernels whose waves own their LDS tile and order its accesses themselves)
template <int B>
__device__ __forceinline__ void loadTileRaysNoSync(const GatherArgs &a, RayTile<B> &s, uint32_t setBase, uint32_t nb,
                                                   int lane) {
  for (int idx = lane; idx < B * 20; idx += 64) {
    const int b = idx / 20, k = (idx % 20) / 4, q = idx % 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    bool valid = false;
    if ((uint32_t)b < nb) {
      const uint32_t set = a.setPerm[setBase + b];
      const gvpm_camera_ray *ray = a.rays + (size_t)set * 5 + k;
      v = reinterpret_cast<const float4 *>(ray)[q];
      if (q == 0) valid = GVPM_RAY_VALID(ray->info) != 0;
    }
    if (q == 0) {
      // the valid bit rides on the sign of len
      const float l = fabsf(v.w);
      v.w = valid ? l : -fmaxf(l, 1e-30f);
      s.ray4[k][0][b] = v;
    } else if (q < 3) {
      s.ray4[k][q][b] = v;
    } else {
      s.gop[k][b] = v.x;
      if (k == 0) {
        s.rnd[b] = v.z;
        s.pix[b] = __float_as_uint(v.w);
        s.edge[b] = GVPM_RAY_EDGE(__float_as_uint(v.y));
      }
    }
  }
}

template <int B>
__device__ __forceinline__ void loadTileRays(const GatherArgs &a, RayTile<B> &s, uint32_t setBase, uint32_t nb,
                                             int lane) {
  loadTileRaysNoSync<B>(a, s, setBase, nb, lane);
  __syncthreads();
}

// the base ray of beam set (setBase + lane % B), straight from global memory (no LDS tile)
struct BaseInfo {
  float rnd;
  uint32_t pix, edge;
};
template <int B>
__device__ __forceinline__ RayReg loadBaseDirect(const GatherArgs &a, uint32_t setBase, uint32_t nb, int lane,
                                                 BaseInfo &bi) {
  RayReg r;
  r.o = r.d = r.eye = mk3(0.f);
  r.len = r.pdf = r.jac = r.gop = 0.f;
  r.valid = false;
  bi.rnd = 0.f;
  bi.pix = bi.edge = 0u;
  const int b = lane % B;
  if ((uint32_t)b < nb) {
    const uint32_t set = a.setPerm[setBase + b];
    const float4 *ray = reinterpret_cast<const float4 *>(a.rays + (size_t)set * 5);
    const float4 q0 = ray[0], q1 = ray[1], q2 = ray[2], q3 = ray[3];
    r.o = mk3(q0.x, q0.y, q0.z);
    r.len = fabsf(q0.w);
    r.d = mk3(q1.x, q1.y, q1.z);
    r.pdf = q1.w;
    r.eye = mk3(q2.x, q2.y, q2.z);
    r.jac = q2.w;
    r.gop = q3.x;
    const uint32_t info = __float_as_uint(q3.y);
    r.valid = GVPM_RAY_VALID(info) != 0;
    bi.edge = GVPM_RAY_EDGE(info);
    bi.rnd = q3.z;
    bi.pix = __float_as_uint(q3.w);
  }
  return r;
}

__device__ __forceinline__ void tileSetupFrom(const GatherArgs &a, const RayReg &base, bool valid, TileWalk &w);

template <int B>
__device__ __forceinline__ void tileSetup(const GatherArgs &a, const RayTile<B> &s, uint32_t nb, int lane,
                                          TileWalk &w) {
  const int b = lane % B;
  const RayReg base = loadRay(s, 0, b);
  tileSetupFrom(a, base, (uint32_t)b < nb && base.valid, w);
}

__device__ __forceinline__ void tileSetupFrom(const GatherArgs &a, const RayReg &base, bool valid, TileWalk &w) {
  w.base = base;
  w.beamValid = valid;
  w.bundleBad = false;
  if (a.grid.mode == 1) {
    float u = 0.f, v = 0.f;
    const bool ok = bundleRayUV(a.grid, base.o.x, base.o.y, base.o.z, base.d.x, base.d.y, base.d.z, u, v);
    w.bundleBad = __ballot(valid && !ok) != 0ull;
    w.bu0 = wave_min(valid ? u : INFINITY); w.bu1 = wave_max(valid ? u : -INFINITY);
    w.bv0 = wave_min(valid ? v : INFINITY); w.bv1 = wave_max(valid ? v : -INFINITY);
    w.any = w.bu0 <= w.bu1 && a.nph > 0 && !w.bundleBad;
    w.A = 0;
    w.cA0 = 0;
    w.cA1 = w.any ? a.grid.levels - 1 : -1;
    w.K = 1;
    w.dimA = a.grid.levels;
    w.pad = 0.f;
    return;
  }
  const float r = a.radius, eps = a.cfg.epsilon;
  const float mint = eps, maxt = w.base.len - eps;
  int A;
  {
    const float ax = fabsf(w.base.d.x), ay = fabsf(w.base.d.y), az = fabsf(w.base.d.z);
    const int my = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
    const int n0 = __popcll(__ballot(w.beamValid && my == 0));
    const int n1 = __popcll(__ballot(w.beamValid && my == 1));
    const int n2 = __popcll(__ballot(w.beamValid && my == 2));
    A = (n0 >= n1 && n0 >= n2) ? 0 : (n1 >= n2 ? 1 : 2);
  }
  w.A = A;
  const int U = (A + 1) % 3, V = (A + 2) % 3;
  w.oA = comp(w.base.o, A); w.dA = comp(w.base.d, A);
  w.oU = comp(w.base.o, U); w.dU = comp(w.base.d, U);
  w.oV = comp(w.base.o, V); w.dV = comp(w.base.d, V);
  const f3 org = mk3(a.grid.org[0], a.grid.org[1], a.grid.org[2]);
  w.orgA = comp(org, A); w.orgU = comp(org, U); w.orgV = comp(org, V);
  const int dimA = A == 0 ? a.grid.dim[0] : (A == 1 ? a.grid.dim[1] : a.grid.dim[2]);
  w.dimU = U == 0 ? a.grid.dim[0] : (U == 1 ? a.grid.dim[1] : a.grid.dim[2]);
  w.dimV = V == 0 ? a.grid.dim[0] : (V == 1 ? a.grid.dim[1] : a.grid.dim[2]);
  w.pad = r * 1.01f + 1e-6f;
  // fattened parameter range: photons may hit with diskDistance up to ~maxt + sqrt(3) r
  w.t0 = mint - 2.f * r;
  w.t1 = maxt + 2.f * r;
  float aLo = INFINITY, aHi = -INFINITY;
  if (w.beamValid) {
    const float e0 = w.oA + w.dA * w.t0, e1 = w.oA + w.dA * w.t1;
    aLo = fminf(e0, e1) - w.pad;
    aHi = fmaxf(e0, e1) + w.pad;
  }
  aLo = wave_min(aLo);
  aHi = wave_max(aHi);
  w.any = aLo <= aHi && a.nph > 0;
  w.cA0 = 1;
  w.cA1 = 0;
  w.dimA = dimA;
  if (w.any) {
    // photons outside the grid bounds sit (clamped) in its border cells: clamp both ends alike
    w.cA0 = min(max(0, (int)floorf((aLo - w.orgA) * a.grid.invCell)), dimA - 1);
    w.cA1 = min(max(0, (int)floorf((aHi - w.orgA) * a.grid.invCell)), dimA - 1);
  }
  // layers per step: thicker slabs when the contiguous (x) axis is the slab axis
  // (development overrides: reserved[1] = layers for A != 0, reserved[2] = layers for A == 0)
  w.K = (A == 0) ? (a.cfg.reserved[2] ? a.cfg.reserved[2] : 8) : (a.cfg.reserved[1] ? a.cfg.reserved[1] : 6);
}

// The tile's rays in one cylinder: axis = the mean ray, radius = how far the segment [tLo, tHi] of any ray strays from it
// -- the distance from the points of a segment to a line is convex along the segment, so its end points bound it.  A
// point within `rad` of some ray's segment then lies within rho + rad of the axis and inside the segments' axial range
// padded by rad.  `slack`: the caller's bound on the fp32 error of ITS test of that point; the error of the cylinder test
// itself is added here.  ok == false (no valid ray, or rays that look every way): no cylinder, everything passes.
struct TileCyl {
  f3 o, d;
  float R2, s0, s1;
  bool ok;
};
__device__ __forceinline__ TileCyl tileCylinder(const RayReg &base, bool valid, float tLo, float tHi, float rad, float slack) {
  TileCyl c;
  c.o = mk3(0.f);
  c.d = mk3(0.f, 0.f, 1.f);
  c.R2 = INFINITY;
  c.s0 = -INFINITY;
  c.s1 = INFINITY;
  c.ok = false;
  const float nv = (float)__popcll(__ballot(valid));
  f3 so = valid ? base.o : mk3(0.f), sd = valid ? base.d : mk3(0.f);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    so.x += __shfl_xor(so.x, o, 64); so.y += __shfl_xor(so.y, o, 64); so.z += __shfl_xor(so.z, o, 64);
    sd.x += __shfl_xor(sd.x, o, 64); sd.y += __shfl_xor(sd.y, o, 64); sd.z += __shfl_xor(sd.z, o, 64);
  }
  const float dl = fsqrt(dot(sd, sd));
  if (!(nv > 0.f && dl > 0.5f * nv)) return c;
  c.o = so * frcp(nv);
  c.d = sd * frcp(dl);
  float rho2 = 0.f, s0 = INFINITY, s1 = -INFINITY;
  if (valid) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const f3 q = (base.o - c.o) + base.d * (e ? tHi : tLo);
      const float sq = dot(q, c.d);
      const f3 pq = q - c.d * sq;
      rho2 = fmaxf(rho2, dot(pq, pq));
      s0 = fminf(s0, sq);
      s1 = fmaxf(s1, sq);
    }
  }
  const float rho = fsqrt(wave_max(rho2));
  s0 = wave_min(s0);
  s1 = wave_max(s1);
  const float own = 4e-6f * (fabsf(c.o.x) + fabsf(c.o.y) + fabsf(c.o.z) + fabsf(s0) + fabsf(s1) + rho + rad) + slack;
  const float R = (rad + rho) * 1.0001f + own;
  c.R2 = R * R;
  c.s0 = s0 - rad * 1.0001f - own;
  c.s1 = s1 + rad * 1.0001f + own;
  c.ok = true;
  return c;
}
__device__ __forceinline__ bool insideCylinder(const TileCyl &c, f3 p) {
  const f3 wv = p - c.o;
  const float sq = dot(wv, c.d);
  return dot(wv, wv) - sq * sq < c.R2 && sq > c.s0 && sq < c.s1;
}

struct CellBox {
  int bx0, bx1, by0, by1, bz0, bz1;
};

// the (U,V) footprint of ONE beam inside the slab [lo, hi] along A; false when it does not reach it
struct BeamSlab {
  float oA, dA, oU, dU, oV, dV, t0, t1;
};
__device__ __forceinline__ bool beamFootprint(const BeamSlab &q, float lo, float hi, float pad, float &uLo, float &uHi,
                                              float &vLo, float &vHi) {
  float ta = q.t0, tb = q.t1;
  bool act = true;
  if (fabsf(q.dA) > 1e-12f) {
    const float inv = 1.f / q.dA;
    const float s0 = (lo - q.oA) * inv, s1 = (hi - q.oA) * inv;
    ta = fmaxf(ta, fminf(s0, s1));
    tb = fminf(tb, fmaxf(s0, s1));
    act = ta <= tb;
  } else {
    act = q.oA >= lo && q.oA <= hi;
  }
  if (!act) return false;
  const float u0 = q.oU + q.dU * ta, u1 = q.oU + q.dU * tb, v0 = q.oV + q.dV * ta, v1 = q.oV + q.dV * tb;
  uLo = fminf(u0, u1) - pad; uHi = fmaxf(u0, u1) + pad;
  vLo = fminf(v0, v1) - pad; vHi = fmaxf(v0, v1) + pad;
  return true;
}

// slab [cA, cAe] along A -> its [lo, hi] in world units; border layers extend to infinity (they hold
// the clamped photons)
__device__ __forceinline__ void slabRange(const GatherArgs &a, const TileWalk &w, int cA, int cAe, float &lo, float &hi) {
  lo = cA == 0 ? -INFINITY : w.orgA + cA * a.grid.cell - w.pad;
  hi = cAe == w.dimA - 1 ? INFINITY : w.orgA + (cAe + 1) * a.grid.cell + w.pad;
}

// cell box from the union footprint of the tile's beams
__device__ __forceinline__ bool boxFromFootprint(const GatherArgs &a, const TileWalk &w, int cA, int cAe, float uLo,
                                                 float uHi, float vLo, float vHi, CellBox &bx) {
  if (!(uLo <= uHi)) return false;
  const int cU0 = min(max(0, (int)floorf((uLo - w.orgU) * a.grid.invCell)), w.dimU - 1);
  const int cU1 = min(max(0, (int)floorf((uHi - w.orgU) * a.grid.invCell)), w.dimU - 1);
  const int cV0 = min(max(0, (int)floorf((vLo - w.orgV) * a.grid.invCell)), w.dimV - 1);
  const int cV1 = min(max(0, (int)floorf((vHi - w.orgV) * a.grid.invCell)), w.dimV - 1);
  // (A,U,V) -> (x,y,z): A=0: x=A y=U z=V; A=1: x=V y=A z=U; A=2: x=U y=V z=A
  const int A = w.A;
  bx.bx0 = A == 0 ? cA : (A == 1 ? cV0 : cU0); bx.bx1 = A == 0 ? cAe : (A == 1 ? cV1 : cU1);
  bx.by0 = A == 0 ? cU0 : (A == 1 ? cA : cV0); bx.by1 = A == 0 ? cU1 : (A == 1 ? cAe : cV1);
  bx.bz0 = A == 0 ? cV0 : (A == 1 ? cU0 : cA); bx.bz1 = A == 0 ? cV1 : (A == 1 ? cU1 : cAe);
  return true;
}

// cell box of the slab layers [cA, cAe] (every lane holds one beam; wave-uniform result);
// false when no beam of the tile reaches the slab
__device__ __forceinline__ bool slabBox(const GatherArgs &a, const TileWalk &w, int cA, int cAe, CellBox &bx) {
  if (a.grid.mode == 1) {
    const BundleBox b = bundleTileBox(a.grid, cA, w.bu0, w.bu1, w.bv0, w.bv1);
    bx.bx0 = b.x0; bx.bx1 = b.x1; bx.by0 = b.y0; bx.by1 = b.y1; bx.bz0 = bx.bz1 = b.z;
    return true;
  }
  float lo, hi;
  slabRange(a, w, cA, cAe, lo, hi);
  float uLo = INFINITY, uHi = -INFINITY, vLo = INFINITY, vHi = -INFINITY;
  if (w.beamValid) {
    const BeamSlab q{w.oA, w.dA, w.oU, w.dU, w.oV, w.dV, w.t0, w.t1};
    float a0, a1, b0, b1;
    if (beamFootprint(q, lo, hi, w.pad, a0, a1, b0, b1)) {
      uLo = a0; uHi = a1; vLo = b0; vHi = b1;
    }
  }
  uLo = wave_min(uLo); uHi = wave_max(uHi);
  vLo = wave_min(vLo); vHi = wave_max(vHi);
  return boxFromFootprint(a, w, cA, cAe, uLo, uHi, vLo, vHi, bx);
}

__device__ __forceinline__ uint2 packCellBox(const CellBox &bx) {
  return make_uint2((uint32_t)bx.bx0 | ((uint32_t)bx.bx1 << 10) | ((uint32_t)bx.by0 << 20),
                    (uint32_t)bx.by1 | ((uint32_t)bx.bz0 << 10) | ((uint32_t)bx.bz1 << 20));
}
__device__ __forceinline__ bool unpackCellBox(uint2 p, CellBox &bx) {
  bx.bx0 = (int)(p.x & 1023u); bx.bx1 = (int)((p.x >> 10) & 1023u); bx.by0 = (int)((p.x >> 20) & 1023u);
  bx.by1 = (int)(p.y & 1023u); bx.bz0 = (int)((p.y >> 10) & 1023u); bx.bz1 = (int)((p.y >> 20) & 1023u);
  return p.x != 0xFFFFFFFFu;
}

// x-contiguous photon range of this lane for range index ri of the box
__device__ __forceinline__ void boxRange(const GatherArgs &a, const CellBox &bx, int ri, int nranges, uint32_t &start,
                                         uint32_t &count) {
  start = 0;
  count = 0;
  if (ri < nranges) {
    const int nyr = bx.by1 - bx.by0 + 1;
    const int y = bx.by0 + ri % nyr, z = bx.bz0 + ri / nyr;
    const uint32_t row = ((uint32_t)z * a.grid.dim[1] + y) * a.grid.dim[0];
    start = a.cellStart[row + bx.bx0];
    count = a.cellStart[row + bx.bx1 + 1] - start;
  }
}

// number of photons staged for a slab step (sum over the box's ranges), wave-uniform
__device__ __forceinline__ uint32_t boxCount(const GatherArgs &a, const CellBox &bx, int lane) {
  const int nranges = (bx.by1 - bx.by0 + 1) * (bx.bz1 - bx.bz0 + 1);
  uint32_t c = 0;
  for (int rbase = 0; rbase < nranges; rbase += 64) {
    uint32_t st, cnt;
    boxRange(a, bx, rbase + lane, nranges, st, cnt);
    c += cnt;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  return c;
}

// ------------------------------------------------------------------------------------------
// plan: cut every tile chunk into work items of ~target staged photons
// item = {setBase, nb, firstLayer, lastLayer}
// itemOff (optional): per item {first 64-entry block of its region in the pair buffer, staged photons};
// the region holds one list of up to `staged` photon indices for each of the item's beams.
// One wave per tile; the slab steps of a chunk are spread over the LANES (each lane loops over the
// tile's beams, held in LDS, and over the cell rows of its own slab box), so the whole walk of a
// chunk is a handful of dependent steps instead of one per slab.
// ------------------------------------------------------------------------------------------
// The planner's LDS (one wave): handed in by the kernel, so that a kernel with other roles beside the planner's (the tail of
// the G-BRE build chain, grid_build.hip) can lay them over one another.
// (PLAN_STAGE: the items staged between two flushes, and the most parts a heavy item is split into; 512 in the planner's
// own kernel, 128 -- 2.5 KB instead of 10 -- where it shares a launch and a CU with everything else: the build chain's tail)
template <int B, uint32_t PLAN_STAGE = 512> struct PlanLds {
  uint4 stItem[PLAN_STAGE];
  uint32_t stStaged[PLAN_STAGE];
  float pb[8][B];
  uint32_t pvalid[B];
};
// bid / nblk: this wave's index among the planner's waves and their number (tiles are taken with stride nblk)
template <int B, uint32_t PLAN_STAGE>
__device__ __forceinline__ void planBody(const GatherArgs &a, uint32_t ntiles, uint32_t target, uint4 *items, uint32_t *itemCount,
                                         uint2 *itemOff, uint32_t *blockTotal, uint32_t itemCap, uint32_t bid, uint32_t nblk,
                                         PlanLds<B, PLAN_STAGE> &L) {
  // The finished items are staged in LDS and flushed ~PLAN_STAGE at a time: one pair of global atomics
  // per flush (same-address returning atomics retire one per ~10 ns: one per tile cost 0.35 ms at C2).
  // (no pair regions to size -- G-Beams -- means heavy items may be split into parts, see below)
  const bool splitHeavy = itemOff == nullptr;
  auto &pb = L.pb;
  auto &pvalid = L.pvalid;
  auto &stItem = L.stItem;
  auto &stStaged = L.stStaged;
  const int lane = threadIdx.x;
  uint32_t nStaged = 0;  // wave-uniform
  auto flush = [&]() {
    __syncthreads();
    for (uint32_t base = 0; base < nStaged; base += 64) {
      const uint32_t k = base + lane;
      const bool live = k < nStaged;
      const uint4 itv = live ? stItem[k] : make_uint4(0u, 0u, 0u, 0u);
      const uint32_t stg = live ? stStaged[k] : 0u;
      const uint32_t blocks = live ? (uint32_t)(((unsigned long long)stg * (itv.y & 0xFFu) + 63ull) / 64ull) : 0u;
      const uint32_t bIncl = wave_scan_incl(blocks, lane);
      const uint32_t bTotal = __shfl(bIncl, 63, 64);
      const uint32_t n = min(64u, nStaged - base);
      uint32_t slot0 = 0, blk0 = 0;
      if (lane == 0) {
        slot0 = atomicAdd(itemCount, n);
        if (itemOff) blk0 = atomicAdd(blockTotal, bTotal);
      }
      slot0 = __shfl(slot0, 0, 64);
      blk0 = __shfl(blk0, 0, 64);
      if (live && slot0 + lane < itemCap) {  // (past the capacity: counted, not written -- the host checks the count)
        items[slot0 + lane] = itv;
        if (itemOff) itemOff[slot0 + lane] = make_uint2(blk0 + (bIncl - blocks), stg);
      }
    }
    nStaged = 0;
    __syncthreads();
  };
  for (uint32_t tile = bid; tile < ntiles; tile += nblk) {
  const uint32_t tileBeg = a.tileStart[tile], tileEnd = a.tileStart[tile + 1];
  for (uint32_t setBase = tileBeg; setBase < tileEnd; setBase += B) {
    const uint32_t nb = min((uint32_t)B, tileEnd - setBase);
    BaseInfo bi;
    const RayReg base = loadBaseDirect<B>(a, setBase, nb, lane, bi);
    TileWalk w;
    tileSetupFrom(a, base, base.valid, w);
    if (w.bundleBad && lane == 0 && a.bundleFlag) atomicOr(a.bundleFlag, 1u);
    if (!w.any) continue;
    __syncthreads();
    if (lane < B && a.grid.mode != 1) {
      pb[0][lane] = w.oA; pb[1][lane] = w.dA; pb[2][lane] = w.oU; pb[3][lane] = w.dU;
      pb[4][lane] = w.oV; pb[5][lane] = w.dV; pb[6][lane] = w.t0; pb[7][lane] = w.t1;
      pvalid[lane] = w.beamValid ? 1u : 0u;
    }
    __syncthreads();
    const int nsteps = (w.cA1 - w.cA0 + w.K) / w.K;
    for (int sbase = 0; sbase < nsteps; sbase += 64) {
      const int step = sbase + lane;
      const bool live = step < nsteps;
      const int cA = w.cA0 + step * w.K, cAe = min(cA + w.K - 1, w.cA1);
      uint32_t cnt = 0;
      if (live) {
        CellBox bx;
        bool haveBox;
        if (a.grid.mode == 1) {
          haveBox = slabBox(a, w, cA, cAe, bx);  // (no wave operation inside in this mode)
        } else {
          float lo, hi;
          slabRange(a, w, cA, cAe, lo, hi);
          float uLo = INFINITY, uHi = -INFINITY, vLo = INFINITY, vHi = -INFINITY;
          // (unrolled by 8 the compiler holds 64 prefetched LDS words and the kernel needs 107 VGPRs: beside the persistent
          // evaluation waves and a traversal wave a SIMD has 48 left)
#pragma unroll 1
          for (uint32_t j = 0; j < nb; ++j) {
            if (!pvalid[j]) continue;
            const BeamSlab q{pb[0][j], pb[1][j], pb[2][j], pb[3][j], pb[4][j], pb[5][j], pb[6][j], pb[7][j]};
            float a0, a1, b0, b1;
            if (beamFootprint(q, lo, hi, w.pad, a0, a1, b0, b1)) {
              uLo = fminf(uLo, a0); uHi = fmaxf(uHi, a1);
              vLo = fminf(vLo, b0); vHi = fmaxf(vHi, b1);
            }
          }
          haveBox = boxFromFootprint(a, w, cA, cAe, uLo, uHi, vLo, vHi, bx);
        }
        if (a.planBoxes && (uint32_t)step < a.planBoxStride)
          a.planBoxes[(size_t)(setBase / B + tile) * a.planBoxStride + step] = haveBox ? packCellBox(bx) : make_uint2(0xFFFFFFFFu, 0u);
        if (haveBox) {
          // photons in the box from the summed-volume table: 8 reads
          const uint32_t nx1 = a.grid.dim[0] + 1, ny1 = a.grid.dim[1] + 1;
          const uint32_t *T = a.sat;
          const size_t z0 = (size_t)bx.bz0 * ny1, z1 = (size_t)(bx.bz1 + 1) * ny1;
          const uint32_t y0 = bx.by0, y1 = bx.by1 + 1, x0 = bx.bx0, x1 = bx.bx1 + 1;
          cnt = (T[(z1 + y1) * nx1 + x1] - T[(z1 + y1) * nx1 + x0] - T[(z1 + y0) * nx1 + x1] + T[(z1 + y0) * nx1 + x0]) -
                (T[(z0 + y1) * nx1 + x1] - T[(z0 + y1) * nx1 + x0] - T[(z0 + y0) * nx1 + x1] + T[(z0 + y0) * nx1 + x0]);
        }
      }
      // greedy cut, in parallel: step s belongs to item floor(exclusive_cumulative(s) / target); items do
      // not span groups of 64 steps (tiles rarely have that many)
      const uint32_t incl = wave_scan_incl(cnt, lane);
      const uint32_t excl = incl - cnt;
      const uint32_t id = excl / target;
      const uint32_t idPrev = __shfl_up(id, 1, 64), idNext = __shfl_down(id, 1, 64);
      const bool first = live && (lane == 0 || id != idPrev);
      const bool liveNext = lane < 63 && step + 1 < nsteps;
      const bool closes = live && (!liveNext || idNext != id);
      const unsigned long long firstMask = __ballot(first);
      const unsigned long long below = firstMask & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
      const int fl = below ? 63 - __clzll(below) : 0;
      const uint32_t exclFirst = __shfl(excl, fl, 64);
      const int cAFirst = __shfl(cA, fl, 64);
      // stage the finished items
      const uint32_t staged = incl - exclFirst;
      const bool emit = closes && staged > 0u;
      const unsigned long long emitMask = __ballot(emit);
      if (emitMask && !splitHeavy && a.grid.mode == 1) {
        // Bundle cells: a tile has a handful of steps (levels), and one of them usually holds most of its photons -- the
        // greedy rule cannot cut there, and the evaluation then meets items twenty times the average (measured at C2:
        // its kernel took twice as long).  A heavy item becomes `parts` items that share its walk and take its staging
        // windows round-robin (as G-Beams' do): item.z = first level | part << 8 | parts << 20; the pair region of a
        // part is sized for its share of the windows.
        for (unsigned long long em = emitMask; em; em &= em - 1ull) {
          const int src = __ffsll((long long)em) - 1;
          const uint32_t stg = __shfl(staged, src, 64);
          const uint32_t cF = (uint32_t)__shfl(cAFirst, src, 64), cE = (uint32_t)__shfl(cAe, src, 64);
          const uint32_t wmax = stg / (uint32_t)STAGE + (cE - cF + 1u);  // windows of the walk, at most
          uint32_t parts = stg > 2u * target ? min(PLAN_STAGE, (stg + target - 1u) / target) : 1u;
          parts = max(1u, min(parts, wmax));
          const uint32_t capP = parts > 1u ? min(stg, ((wmax + parts - 1u) / parts) * (uint32_t)STAGE) : stg;
          if (nStaged + parts > PLAN_STAGE) flush();
          for (uint32_t pp = (uint32_t)lane; pp < parts; pp += 64u) {
            stItem[nStaged + pp] = make_uint4(setBase, nb | ((setBase / B + tile) << 8), cF | (parts > 1u ? (pp << 8) | (parts << 20) : 0u), cE);
            stStaged[nStaged + pp] = capP;
          }
          nStaged += parts;
        }
      } else if (emitMask && !splitHeavy) {
        if (nStaged + 64u > PLAN_STAGE) flush();
        if (emit) {
          const uint32_t k = nStaged + (uint32_t)__popcll(emitMask & ((1ull << lane) - 1ull));
          // (G-BRE: the chunk's ordinal rides on nb, for the traversal to find the chunk's boxes)
          stItem[k] = make_uint4(setBase, nb | ((setBase / B + tile) << 8), (uint32_t)cAFirst, (uint32_t)cAe);
          stStaged[k] = staged;
        }
        nStaged += (uint32_t)__popcll(emitMask);
      } else if (emitMask) {
        // A single slab step cannot be cut by the greedy rule above, and a dense spot of the map (the shaft of
        // S-laser: half a million sub-beams in one box) then is ONE item that a single wave walks long after every
        // other wave has finished (measured at C3: the traversal's waves were resident 30 % of its duration).  Such an
        // item is emitted as `parts` items that share its walk and take its staging windows round-robin:
        // item.y = nb | part << 8 | parts << 20.
        for (unsigned long long em = emitMask; em; em &= em - 1ull) {
          const int src = __ffsll((long long)em) - 1;
          const uint32_t stg = __shfl(staged, src, 64);
          const uint32_t cF = (uint32_t)__shfl(cAFirst, src, 64), cE = (uint32_t)__shfl(cAe, src, 64);
          const uint32_t parts = stg > 2u * target ? min(PLAN_STAGE, (stg + target - 1u) / target) : 1u;
          if (nStaged + parts > PLAN_STAGE) flush();
          for (uint32_t pp = (uint32_t)lane; pp < parts; pp += 64u) {
            stItem[nStaged + pp] = make_uint4(setBase, nb | (parts > 1u ? (pp << 8) | (parts << 20) : 0u), cF, cE);
            stStaged[nStaged + pp] = (stg + parts - 1u) / parts;
          }
          nStaged += parts;
        }
      }
    }
  }
  }
  if (nStaged) flush();
}
// ------------------------------------------------------------------------------------------
// The same plan with 64 / B tile chunks walked SIDE BY SIDE by one wave (round 6): B lanes a chunk -- its beams during
// the set-up, its slab steps afterwards.  A chunk of 16 beams has ~13 slab steps at C2: the planner above keeps 13 of 64
// lanes busy for a chain of dependent loads per chunk, one chunk after the other, in 107 registers.  Everything that was
// a wave-wide collective there is one over the B lanes of a group here (xor shuffles below B stay inside a group), what was
// wave-uniform is group-uniform.  3D grid and pair regions only (G-BRE): the bundle cells and G-Beams keep planBody.
// Same items, same boxes -- but an item does not span groups of B steps (there: of 64).
// ------------------------------------------------------------------------------------------
template <int B, uint32_t PLAN_STAGE> struct PlanLdsQ {
  uint4 stItem[PLAN_STAGE];
  uint32_t stStaged[PLAN_STAGE];
  float pb[64 / B][8][B];
  uint32_t pvalid[64 / B][B];
};
template <int B> __device__ __forceinline__ float group_min(float v) {
#pragma unroll
  for (int o = B / 2; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
template <int B> __device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = B / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
template <int B, uint32_t PLAN_STAGE>
__device__ __forceinline__ void planBodyQ(const GatherArgs &a, uint32_t ntiles, uint32_t target, uint4 *items, uint32_t *itemCount,
                                          uint2 *itemOff, uint32_t *blockTotal, uint32_t itemCap, uint32_t bid, uint32_t nblk,
                                          PlanLdsQ<B, PLAN_STAGE> &L) {
  constexpr int G = 64 / B;
  auto &stItem = L.stItem;
  auto &stStaged = L.stStaged;
  const int lane = threadIdx.x, q = lane / B, bl = lane % B;
  const unsigned long long gm = (B == 64 ? ~0ull : ((1ull << (B & 63)) - 1ull) << (q * B));  // my group's lanes
  uint32_t nStaged = 0;  // wave-uniform
  auto flush = [&]() {
    __syncthreads();
    for (uint32_t base = 0; base < nStaged; base += 64) {
      const uint32_t k = base + lane;
      const bool live = k < nStaged;
      const uint4 itv = live ? stItem[k] : make_uint4(0u, 0u, 0u, 0u);
      const uint32_t stg = live ? stStaged[k] : 0u;
      const uint32_t blocks = live ? (uint32_t)(((unsigned long long)stg * (itv.y & 0xFFu) + 63ull) / 64ull) : 0u;
      const uint32_t bIncl = wave_scan_incl(blocks, lane);
      const uint32_t bTotal = __shfl(bIncl, 63, 64);
      const uint32_t n = min(64u, nStaged - base);
      uint32_t slot0 = 0, blk0 = 0;
      if (lane == 0) {
        slot0 = atomicAdd(itemCount, n);
        blk0 = atomicAdd(blockTotal, bTotal);
      }
      slot0 = __shfl(slot0, 0, 64);
      blk0 = __shfl(blk0, 0, 64);
      if (live && slot0 + lane < itemCap) {  // (past the capacity: counted, not written -- the host checks the count)
        items[slot0 + lane] = itv;
        itemOff[slot0 + lane] = make_uint2(blk0 + (bIncl - blocks), stg);
      }
    }
    nStaged = 0;
    __syncthreads();
  };
  const float r = a.radius, eps = a.cfg.epsilon;
  // (group q of wave `bid` takes the tiles bid + (trip * G + q) * nblk: the tiles one wave of planBody takes one after the
  // other -- the items then come out in the order they always had.  With ADJACENT tiles side by side the item list runs over the
  // image row by row and the traversal, whose workgroup i takes item i, lost a fifth of its speed: 0.33 -> 0.40 ms alone.)
  for (uint32_t t0 = bid; t0 < ntiles; t0 += nblk * G) {
    const uint32_t tile = t0 + (uint32_t)q * nblk;
    const bool haveTile = tile < ntiles;
    const uint32_t tileBeg = haveTile ? a.tileStart[tile] : 0u, tileEnd = haveTile ? a.tileStart[tile + 1] : 0u;
    for (uint32_t setBase = tileBeg; __ballot(setBase < tileEnd) != 0ull; setBase += B) {
      const uint32_t nb = setBase < tileEnd ? min((uint32_t)B, tileEnd - setBase) : 0u;
      BaseInfo bi;
      const RayReg base = loadBaseDirect<B>(a, setBase, nb, lane, bi);
      // ---- tileSetupFrom over the group ----
      const bool beamValid = base.valid;
      int A;
      {
        const float ax = fabsf(base.d.x), ay = fabsf(base.d.y), az = fabsf(base.d.z);
        const int my = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
        const int n0 = __popcll(__ballot(beamValid && my == 0) & gm);
        const int n1 = __popcll(__ballot(beamValid && my == 1) & gm);
        const int n2 = __popcll(__ballot(beamValid && my == 2) & gm);
        A = (n0 >= n1 && n0 >= n2) ? 0 : (n1 >= n2 ? 1 : 2);
      }
      TileWalk w;
      w.base = base;
      w.beamValid = beamValid;
      w.bundleBad = false;
      w.A = A;
      const int U = (A + 1) % 3, V = (A + 2) % 3;
      w.oA = comp(base.o, A); w.dA = comp(base.d, A);
      w.oU = comp(base.o, U); w.dU = comp(base.d, U);
      w.oV = comp(base.o, V); w.dV = comp(base.d, V);
      const f3 org = mk3(a.grid.org[0], a.grid.org[1], a.grid.org[2]);
      w.orgA = comp(org, A); w.orgU = comp(org, U); w.orgV = comp(org, V);
      const int dimA = A == 0 ? a.grid.dim[0] : (A == 1 ? a.grid.dim[1] : a.grid.dim[2]);
      w.dimU = U == 0 ? a.grid.dim[0] : (U == 1 ? a.grid.dim[1] : a.grid.dim[2]);
      w.dimV = V == 0 ? a.grid.dim[0] : (V == 1 ? a.grid.dim[1] : a.grid.dim[2]);
      w.dimA = dimA;
      w.pad = r * 1.01f + 1e-6f;
      w.t0 = eps - 2.f * r;
      w.t1 = (base.len - eps) + 2.f * r;
      float aLo = INFINITY, aHi = -INFINITY;
      if (beamValid) {
        const float e0 = w.oA + w.dA * w.t0, e1 = w.oA + w.dA * w.t1;
        aLo = fminf(e0, e1) - w.pad;
        aHi = fmaxf(e0, e1) + w.pad;
      }
      aLo = group_min<B>(aLo);
      aHi = group_max<B>(aHi);
      w.any = aLo <= aHi && a.nph > 0;
      w.cA0 = 1;
      w.cA1 = 0;
      if (w.any) {
        w.cA0 = min(max(0, (int)floorf((aLo - w.orgA) * a.grid.invCell)), dimA - 1);
        w.cA1 = min(max(0, (int)floorf((aHi - w.orgA) * a.grid.invCell)), dimA - 1);
      }
      w.K = (A == 0) ? (a.cfg.reserved[2] ? a.cfg.reserved[2] : 8) : (a.cfg.reserved[1] ? a.cfg.reserved[1] : 6);
      if (__ballot(w.any) == 0ull) continue;  // (wave-uniform)
      __syncthreads();
      L.pb[q][0][bl] = w.oA; L.pb[q][1][bl] = w.dA; L.pb[q][2][bl] = w.oU; L.pb[q][3][bl] = w.dU;
      L.pb[q][4][bl] = w.oV; L.pb[q][5][bl] = w.dV; L.pb[q][6][bl] = w.t0; L.pb[q][7][bl] = w.t1;
      L.pvalid[q][bl] = beamValid ? 1u : 0u;
      __syncthreads();
      const int nsteps = w.any ? (w.cA1 - w.cA0 + w.K) / w.K : 0;
      int maxSteps = nsteps;
#pragma unroll
      for (int o = 32; o >= B; o >>= 1) maxSteps = max(maxSteps, __shfl_xor(maxSteps, o, 64));
      const uint32_t chunkOrd = setBase / B + tile;
      for (int sbase = 0; sbase < maxSteps; sbase += B) {
        const int step = sbase + bl;
        const bool live = step < nsteps;
        const int cA = w.cA0 + step * w.K, cAe = min(cA + w.K - 1, w.cA1);
        uint32_t cnt = 0;
        if (live) {
          CellBox bx;
          float lo, hi;
          slabRange(a, w, cA, cAe, lo, hi);
          float uLo = INFINITY, uHi = -INFINITY, vLo = INFINITY, vHi = -INFINITY;
#pragma unroll 1
          for (uint32_t j = 0; j < nb; ++j) {
            if (!L.pvalid[q][j]) continue;
            const BeamSlab qs{L.pb[q][0][j], L.pb[q][1][j], L.pb[q][2][j], L.pb[q][3][j], L.pb[q][4][j], L.pb[q][5][j], L.pb[q][6][j], L.pb[q][7][j]};
            float a0, a1, b0, b1;
            if (beamFootprint(qs, lo, hi, w.pad, a0, a1, b0, b1)) {
              uLo = fminf(uLo, a0); uHi = fmaxf(uHi, a1);
              vLo = fminf(vLo, b0); vHi = fmaxf(vHi, b1);
            }
          }
          const bool haveBox = boxFromFootprint(a, w, cA, cAe, uLo, uHi, vLo, vHi, bx);
          if (a.planBoxes && (uint32_t)step < a.planBoxStride)
            a.planBoxes[(size_t)chunkOrd * a.planBoxStride + step] = haveBox ? packCellBox(bx) : make_uint2(0xFFFFFFFFu, 0u);
          if (haveBox) {
            // photons in the box from the summed-volume table: 8 reads
            const uint32_t nx1 = a.grid.dim[0] + 1, ny1 = a.grid.dim[1] + 1;
            const uint32_t *T = a.sat;
            const size_t z0 = (size_t)bx.bz0 * ny1, z1 = (size_t)(bx.bz1 + 1) * ny1;
            const uint32_t y0 = bx.by0, y1 = bx.by1 + 1, x0 = bx.bx0, x1 = bx.bx1 + 1;
            cnt = (T[(z1 + y1) * nx1 + x1] - T[(z1 + y1) * nx1 + x0] - T[(z1 + y0) * nx1 + x1] + T[(z1 + y0) * nx1 + x0]) -
                  (T[(z0 + y1) * nx1 + x1] - T[(z0 + y1) * nx1 + x0] - T[(z0 + y0) * nx1 + x1] + T[(z0 + y0) * nx1 + x0]);
          }
        }
        // greedy cut over the group's B steps
        uint32_t incl = cnt;
#pragma unroll
        for (int o = 1; o < B; o <<= 1) {
          const uint32_t nv = __shfl_up(incl, o, 64);
          if (bl >= o) incl += nv;
        }
        const uint32_t excl = incl - cnt;
        const uint32_t id = excl / target;
        const uint32_t idPrev = __shfl_up(id, 1, 64), idNext = __shfl_down(id, 1, 64);
        const bool first = live && (bl == 0 || id != idPrev);
        const bool liveNext = bl < B - 1 && step + 1 < nsteps;
        const bool closes = live && (!liveNext || idNext != id);
        const unsigned long long firstMask = __ballot(first) & gm;
        const unsigned long long below = firstMask & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
        const int fl = below ? 63 - __clzll(below) : lane;
        const uint32_t exclFirst = __shfl(excl, fl, 64);
        const int cAFirst = __shfl(cA, fl, 64);
        const uint32_t staged = incl - exclFirst;
        const bool emit = closes && staged > 0u;
        const unsigned long long emitMask = __ballot(emit);
        if (emitMask) {
          if (nStaged + 64u > PLAN_STAGE) flush();
          if (emit) {
            const uint32_t k = nStaged + (uint32_t)__popcll(emitMask & ((1ull << lane) - 1ull));
            stItem[k] = make_uint4(setBase, nb | (chunkOrd << 8), (uint32_t)cAFirst, (uint32_t)cAe);
            stStaged[k] = staged;
          }
          nStaged += (uint32_t)__popcll(emitMask);
        }
      }
    }
  }
  if (nStaged) flush();
}

template <int B>
__global__ __launch_bounds__(64) void plan_kernel(GatherArgs a, uint32_t ntiles, uint32_t target, uint4 *items,
                                                  uint32_t *itemCount, uint2 *itemOff, uint32_t *blockTotal,
                                                  uint32_t itemCap) {
  __shared__ PlanLds<B, 512> L;
  planBody<B, 512>(a, ntiles, target, items, itemCount, itemOff, blockTotal, itemCap, blockIdx.x, gridDim.x, L);
}

}  // namespace gvpm
