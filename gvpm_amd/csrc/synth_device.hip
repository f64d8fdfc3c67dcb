// Device-side producers for the closed-form synthetic scenes: photon shooting and camera-beam
// generation on the GPU, so that an iteration's inputs never cross PCIe (204 MB per iteration at C2, about twice
// the gather's step time).  The per-path / per-pixel logic is the host generator's, compiled from the same source
// (host/synth_core.h); here it runs one light path or one pixel per lane.
//
// Order.  The host loop appends the photons of path 0, 1, 2, ... until `capacity` is reached (GPhotonMap::tryAppend
// semantics, gvpm_proc.cpp:278-350) and reports how many paths it shot.  The device reproduces exactly that: a batch
// of paths is walked once to COUNT its photons, an exclusive scan gives every path its slot, a one-lane search finds
// the path at which the capacity is reached, and the batch is walked again to WRITE (recomputing a path is cheaper
// than parking ~4 KB of records per lane).  Camera beam sets are compacted the same way (pixels in row-major order).
// The output is the C ABI's device-resident SoA (gvpm_upload_photons_dev / gvpm_upload_camera_beams_dev).
#include <hip/hip_runtime.h>

#include <new>
#include <vector>

#include "../../include/gvpm_hip.h"
#include "../host/synth_core.h"
#include "device_types.h"

namespace gvpm {

hipError_t exclusiveSumU32(SortTemp &tmp, const uint32_t *in, uint32_t *out, uint32_t n, hipStream_t s);

namespace {

template <typename T> struct Buf {
  T *p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t n) {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc((void **)&p, n * sizeof(T));
    if (e == hipSuccess) cap = n;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct PhotonOut {
  float *v3[8];  // pos wi flux parent_pos parent_n prefix_w parent_scat parent_wi
  float *f1[4];  // parent_pdf edge_pdf parent_rr parent_g
  uint32_t *flags, *pathId;
  float *endN;
};

// the sinks of flattenPath / flattenBeams (synth_core.h): the count pass keeps a number, the write pass stores each record
// where it belongs -- neither holds the path's records in (scratch) memory
struct CountSink {
  int n;
  __device__ CountSink() : n(0) {}
  __device__ void clear() { n = 0; }
  __device__ bool empty() const { return n == 0; }
  __device__ void push_back(const PhotonRec &) {
    if (n < GVPM_SYNTH_MAXV) ++n;
  }
};
// The walk runs ONCE (round 3; it ran twice: count, then write): its records are parked at a slot that depends on the path
// alone -- path k, record n at (k * stride + n) -- and a copy kernel moves them to their place in the output once the
// scans have said where that is.  32 words a record; with 288 GB of HBM the 1-2 GB of parking space are affordable, a
// second walk of 0.85 M light paths in fp64 (1.5 ms) was not.
constexpr int PARK_WORDS = 32;
struct ParkSink {
  float *park;  // this path's slots
  int n, stride;
  __device__ void clear() { n = 0; }
  __device__ bool empty() const { return n == 0; }
  __device__ void push_back(const PhotonRec &r) {
    if (n >= stride) return;
    float *d = park + (size_t)n * PARK_WORDS;
    ++n;
    const V3 v[9] = {r.pos, r.wi, r.flux, r.parentPos, r.parentN, r.prefixW, r.parentScat, r.parentWi, r.endN};
#pragma unroll
    for (int a = 0; a < 9; ++a) {
      d[3 * a] = (float)v[a].x;
      d[3 * a + 1] = (float)v[a].y;
      d[3 * a + 2] = (float)v[a].z;
    }
    d[27] = r.parentPdf;
    d[28] = r.edgePdf;
    d[29] = r.parentRR;
    d[30] = r.parentG;
    d[31] = __uint_as_float(r.flags);
  }
};

// A wave owns `chunk` consecutive paths of the batch and its lanes REFILL: a lane whose path has ended takes the wave's next
// path while its neighbours walk on (walkBegin / walkStep, synth_core.h).  With one path per lane a wave ran until its
// longest path ended -- twelve bounces with the mean near five: less than half of the lane-steps did work.  Where a path's
// records are parked depends on the path alone, so which lane walks it changes nothing.
template <bool BEAMS>
__device__ __forceinline__ void walkChunk(const SceneView &sc, int iteration, uint64_t base, uint32_t k0, uint32_t k1, float *park,
                                          int stride, uint32_t *counts, uint32_t *counted, uint32_t *nonEmpty) {
  const int lane = threadIdx.x & 63;
  ParkSink recs;
  recs.park = park;
  recs.stride = stride;
  recs.n = 0;
  StreamPath<ParkSink, BEAMS> path(sc, recs);
  Philox rng(sc.seed, 0x11ffu, (uint32_t)iteration, 0u, 0u);
  V3 throughput(1.0);
  uint32_t next = k0;  // wave-uniform
  uint32_t k = 0;
  int i = 0;
  bool active = false;
  for (;;) {
    // refill: the idle lanes take the next paths, in lane order
    const unsigned long long idle = __ballot(!active);
    if (idle && next < k1) {
      const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
      if (!active && next + rank < k1) {
        k = next + rank;
        const uint64_t idx = base + k;
        rng = Philox(sc.seed, 0x11ffu, (uint32_t)iteration, (uint32_t)idx, (uint32_t)(idx >> 32));
        recs.park = park + (size_t)k * stride * PARK_WORDS;
        walkBegin(sc, rng, path, throughput);
        i = 1;
        active = true;
      }
      next = min(k1, next + (uint32_t)__popcll(idle));
    }
    if (!__ballot(active)) break;
    if (active) {
      if (i >= sc.maxDepth || !walkStep(sc, rng, path, throughput, i)) {
        const bool c = path.finish();
        counts[k] = (uint32_t)recs.n;
        counted[k] = c ? 1u : 0u;
        nonEmpty[k] = recs.n ? 1u : 0u;
        active = false;
      } else {
        ++i;
      }
    }
  }
}

__global__ __launch_bounds__(64) void synth_walk_kernel(SceneView sc, int iteration, uint64_t base, uint32_t m, uint32_t chunk, int beams,
                                                        float *park, int stride, uint32_t *counts, uint32_t *counted,
                                                        uint32_t *nonEmpty) {
  const uint32_t k0 = blockIdx.x * chunk;
  if (k0 >= m) return;
  const uint32_t k1 = min(m, k0 + chunk);
  if (beams) walkChunk<true>(sc, iteration, base, k0, k1, park, stride, counts, counted, nonEmpty);
  else walkChunk<false>(sc, iteration, base, k0, k1, park, stride, counts, counted, nonEmpty);
}

// ctl: [0] photons stored before this batch, [1] paths counted before, [2] paths with photons before,
//      [3] (out) paths of this batch that the host loop would have processed
__global__ void synth_stop_kernel(const uint32_t *counts, const uint32_t *offs, const uint32_t *countedOffs,
                                  const uint32_t *counted, const uint32_t *neOffs, const uint32_t *nonEmpty, uint32_t m,
                                  uint64_t capacity, uint32_t *ctl) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const uint64_t before = ctl[0];
  // first k with before + offs[k] + counts[k] >= capacity
  uint32_t lo = 0, hi = m;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (before + offs[mid] + counts[mid] >= capacity) hi = mid;
    else lo = mid + 1;
  }
  const uint32_t processed = lo < m ? lo + 1u : m;
  const uint32_t last = processed - 1u;
  ctl[3] = processed;
  const uint64_t stored = before + offs[last] + counts[last];
  ctl[4] = (uint32_t)(stored < capacity ? stored : capacity);
  ctl[5] = ctl[1] + countedOffs[last] + counted[last];
  ctl[6] = ctl[2] + neOffs[last] + nonEmpty[last];
}

// one lane per (path, record slot): the parked records of the paths the host loop would have processed -> the output
__global__ __launch_bounds__(256) void synth_place_kernel(const float *__restrict__ park, int stride, const uint32_t *__restrict__ counts,
                                                          const uint32_t *__restrict__ offs, const uint32_t *__restrict__ neOffs,
                                                          const uint32_t *__restrict__ ctl, uint32_t m, uint64_t capacity, PhotonOut o) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t k = t / (uint32_t)stride, q = t % (uint32_t)stride;
  if (k >= m || k >= ctl[3] || q >= counts[k]) return;
  const uint64_t i = (uint64_t)ctl[0] + offs[k] + q;
  if (i >= capacity) return;
  const float *d = park + ((size_t)k * stride + q) * PARK_WORDS;
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    o.v3[a][3 * i] = d[3 * a];
    o.v3[a][3 * i + 1] = d[3 * a + 1];
    o.v3[a][3 * i + 2] = d[3 * a + 2];
  }
  if (o.endN) {
    o.endN[3 * i] = d[24];
    o.endN[3 * i + 1] = d[25];
    o.endN[3 * i + 2] = d[26];
  }
  o.f1[0][i] = d[27];
  o.f1[1][i] = d[28];
  o.f1[2][i] = d[29];
  o.f1[3][i] = d[30];
  o.flags[i] = __float_as_uint(d[31]);
  o.pathId[i] = ctl[2] + neOffs[k];
}

// ---- camera beams ----
__device__ __forceinline__ bool ownedPixel(const SceneView &sc, uint32_t p, int tileMod, int tileRem, int &px, int &py) {
  px = (int)(p % (uint32_t)sc.width);
  py = (int)(p / (uint32_t)sc.width);
  const int tilesX = (sc.width + 3) / 4;
  return !(tileMod > 1 && ((py / 4) * tilesX + px / 4) % tileMod != tileRem);
}
__global__ __launch_bounds__(64) void synth_beam_flag_kernel(SceneView sc, int iteration, int tileMod, int tileRem,
                                                             uint32_t npix, uint32_t *flag) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  int px, py;
  uint32_t f = 0;
  if (ownedPixel(sc, p, tileMod, tileRem, px, py)) {
    // the number of sets (medium edges of the base path: 0, 1, or 2 behind a mirror)
    gvpm_camera_ray sets[2][5];
    float w[2];
    f = (uint32_t)cameraBeamSets(sc, iteration, px, py, sets, w);
  }
  flag[p] = f;
}
__global__ __launch_bounds__(64) void synth_beam_write_kernel(SceneView sc, int iteration, uint32_t npix,
                                                              const uint32_t *flag, const uint32_t *offs,
                                                              gvpm_camera_ray *out) {
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix || !flag[p]) return;
  const int px = (int)(p % (uint32_t)sc.width), py = (int)(p / (uint32_t)sc.width);
  gvpm_camera_ray sets[2][5];
  float w[2];
  const int n = cameraBeamSets(sc, iteration, px, py, sets, w);
  gvpm_camera_ray *dst = out + (size_t)offs[p] * 5;
  for (int q = 0; q < n; ++q)
    for (int k = 0; k < 5; ++k) dst[5 * q + k] = sets[q][k];
}

}  // namespace
}  // namespace gvpm

using namespace gvpm;

struct gvpm_devgen {
  int device = 0;
  hipStream_t stream = nullptr;
  SceneView view;
  Buf<SynthTri> tris;
  Buf<SynthMat> mats;
  Buf<uint32_t> counts, counted, nonEmpty, offs, countedOffs, neOffs, ctl;
  Buf<float> park;  // parked records of a batch's paths (synth_walk_kernel)
  // Outputs are BORROWED by the gather context (gvpm_upload_*_dev), whose kernels of up to three consecutive steps are
  // in flight: three output sets per kind, used in turn, so that a call never writes what the two steps before it read.
  struct PhotonOut {
    Buf<float> f3[8], f1[4], endN;
    Buf<uint32_t> flags, pathId;
  } pout[3];
  int poutIdx = 0, raysIdx = 0;
  PhotonOut &po() { return pout[poutIdx]; }
  Buf<uint32_t> pixFlag, pixOffs;
  Buf<gvpm_camera_ray> raysOut[3];
  SortTemp scanTmp;
  uint32_t *pinned = nullptr;  // 8 words, mapped host memory
  double pathsPerPhoton = 0.0;  // light paths walked per stored photon in the last shoot
};

#define SY_TRY(expr)                         \
  do {                                       \
    hipError_t _e = (expr);                  \
    if (_e != hipSuccess) return GVPM_ERR_HIP; \
  } while (0)

extern "C" {

int gvpm_devgen_create(const gvpm_devgen_scene *sc, int device, gvpm_devgen **out) {
  if (!sc || !out) return GVPM_ERR_INVALID_ARG;
  if (sc->n_tris && (!sc->tris || !sc->tri_mat)) return GVPM_ERR_INVALID_ARG;
  if (sc->n_mats && (!sc->mat_kind || !sc->mat_albedo)) return GVPM_ERR_INVALID_ARG;
  if (sc->max_depth + 1 > GVPM_SYNTH_MAXV || sc->width <= 0 || sc->height <= 0) return GVPM_ERR_INVALID_ARG;
  if (hipSetDevice(device) != hipSuccess) return GVPM_ERR_NO_DEVICE;
  gvpm_devgen *g = new (std::nothrow) gvpm_devgen();
  if (!g) return GVPM_ERR_HIP;
  g->device = device;
  auto bail = [&](int rc) {
    gvpm_devgen_destroy(g);
    return rc;
  };
  if (hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking) != hipSuccess) return bail(GVPM_ERR_HIP);
  std::vector<SynthTri> tris(sc->n_tris);
  for (uint32_t i = 0; i < sc->n_tris; ++i) {
    const double *t = sc->tris + 12 * (size_t)i;
    tris[i].v0 = V3(t[0], t[1], t[2]);
    tris[i].e1 = V3(t[3], t[4], t[5]);
    tris[i].e2 = V3(t[6], t[7], t[8]);
    tris[i].n = V3(t[9], t[10], t[11]);
    tris[i].mat = sc->tri_mat[i];
    if (tris[i].mat < 0 || (uint32_t)tris[i].mat >= sc->n_mats) return bail(GVPM_ERR_INVALID_ARG);
  }
  std::vector<SynthMat> mats(sc->n_mats);
  for (uint32_t i = 0; i < sc->n_mats; ++i) {
    mats[i].kind = sc->mat_kind[i];
    mats[i].albedo = V3(sc->mat_albedo[3 * i], sc->mat_albedo[3 * i + 1], sc->mat_albedo[3 * i + 2]);
    mats[i].spec = V3(0.0);
    mats[i].exponent = mats[i].specWeight = 0.0;
    mats[i].bsdf = -1;
    // (the device generators' closed set: Lambertian, index-matched boundary, mirror -- a Phong wall's parameters do not
    // travel in gvpm_devgen_scene: such scenes are shot on the host)
    if (mats[i].kind < 0 || mats[i].kind > MAT_MIRROR) return bail(GVPM_ERR_UNSUPPORTED);
  }
  if (g->tris.ensure(tris.size() + 1) != hipSuccess || g->mats.ensure(mats.size() + 1) != hipSuccess) return bail(GVPM_ERR_HIP);
  if (!tris.empty() && hipMemcpy(g->tris.p, tris.data(), tris.size() * sizeof(SynthTri), hipMemcpyHostToDevice) != hipSuccess)
    return bail(GVPM_ERR_HIP);
  if (!mats.empty() && hipMemcpy(g->mats.p, mats.data(), mats.size() * sizeof(SynthMat), hipMemcpyHostToDevice) != hipSuccess)
    return bail(GVPM_ERR_HIP);
  if (hipHostMalloc((void **)&g->pinned, 64, hipHostMallocMapped) != hipSuccess) return bail(GVPM_ERR_HIP);
  SceneView &v = g->view;
  v.tris = g->tris.p;
  v.ntri = (int)sc->n_tris;
  v.mats = g->mats.p;
  v.nmats = (int)sc->n_mats;
  v.lightC = V3(sc->light_c[0], sc->light_c[1], sc->light_c[2]);
  v.lightU = V3(sc->light_u[0], sc->light_u[1], sc->light_u[2]);
  v.lightV = V3(sc->light_v[0], sc->light_v[1], sc->light_v[2]);
  v.lightN = V3(sc->light_n[0], sc->light_n[1], sc->light_n[2]);
  v.radiance = V3(sc->radiance[0], sc->radiance[1], sc->radiance[2]);
  v.lightArea = sc->light_area;
  v.medium = sc->medium;
  v.camPos = V3(sc->cam_pos[0], sc->cam_pos[1], sc->cam_pos[2]);
  {
    const double *m = sc->cam_to_world;
    bool zero = true;
    for (int k = 0; k < 9; ++k) zero = zero && m[k] == 0.0;
    v.camX = zero ? V3(1, 0, 0) : V3(m[0], m[3], m[6]);
    v.camY = zero ? V3(0, 1, 0) : V3(m[1], m[4], m[7]);
    v.camZ = zero ? V3(0, 0, 1) : V3(m[2], m[5], m[8]);
  }
  v.tanHalfFovX = sc->tan_half_fov_x;
  v.width = sc->width;
  v.height = sc->height;
  v.seed = sc->seed;
  v.cameraInside = sc->camera_inside != 0;
  v.maxDepth = sc->max_depth;
  v.rrDepth = sc->rr_depth;
  v.minDepth = sc->min_depth;
  v.cameraSphere = sc->camera_sphere;
  *out = g;
  return GVPM_OK;
}

int gvpm_devgen_destroy(gvpm_devgen *g) {
  if (!g) return GVPM_ERR_INVALID_ARG;
  (void)hipSetDevice(g->device);
  if (g->stream) (void)hipStreamSynchronize(g->stream);
  g->tris.release(); g->mats.release();
  g->counts.release(); g->counted.release(); g->nonEmpty.release(); g->offs.release(); g->countedOffs.release();
  g->neOffs.release(); g->ctl.release(); g->park.release();
  for (auto &o : g->pout) {
    for (auto &b : o.f3) b.release();
    for (auto &b : o.f1) b.release();
    o.endN.release(); o.flags.release(); o.pathId.release();
  }
  g->pixFlag.release(); g->pixOffs.release();
  for (auto &b : g->raysOut) b.release();
  if (g->scanTmp.d) (void)hipFree(g->scanTmp.d);
  if (g->pinned) (void)hipHostFree(g->pinned);
  if (g->stream) (void)hipStreamDestroy(g->stream);
  delete g;
  return GVPM_OK;
}

// beams != 0: photon beams (one record per medium edge, LTBeamMap::tryAppendLT) + end normals
static int shootCommon(gvpm_devgen *g, int iteration, uint64_t capacity, int beams, gvpm_photon_soa *soa, const float **endN,
                       uint64_t *nbPaths) {
  if (!g || !soa || !nbPaths || capacity == 0 || capacity > 0x7FFFFFF0ull) return GVPM_ERR_INVALID_ARG;
  (void)hipSetDevice(g->device);
  hipStream_t s = g->stream;
  g->poutIdx = (g->poutIdx + 1) % 3;
  for (auto &b : g->po().f3) SY_TRY(b.ensure(capacity * 3 + 4));
  for (auto &b : g->po().f1) SY_TRY(b.ensure(capacity + 4));
  SY_TRY(g->po().flags.ensure(capacity + 4));
  SY_TRY(g->po().pathId.ensure(capacity + 4));
  if (beams) SY_TRY(g->po().endN.ensure(capacity * 3 + 4));
  // paths per batch: what the previous shoot needed per stored photon plus a margin, so that one batch (one
  // count pass + one write pass) usually suffices; `capacity` paths the first time
  const double perPhoton = g->pathsPerPhoton > 0.0 ? g->pathsPerPhoton * 1.06 : 1.0;
  const uint32_t m = (uint32_t)std::min<uint64_t>(std::max<uint64_t>((uint64_t)((double)capacity * perPhoton) + 64, 4096), 1u << 22);
  Buf<uint32_t> *scratch[] = {&g->counts, &g->counted, &g->nonEmpty, &g->offs, &g->countedOffs, &g->neOffs};
  for (auto *b : scratch) SY_TRY(b->ensure((size_t)m + 1));
  SY_TRY(g->ctl.ensure(8));
  SY_TRY(hipMemsetAsync(g->ctl.p, 0, 8 * sizeof(uint32_t), s));
  // a path stores at most one record per vertex beyond the emitter sample
  const int stride = std::max(1, std::min(GVPM_SYNTH_MAXV, g->view.maxDepth));
  SY_TRY(g->park.ensure((size_t)m * stride * PARK_WORDS + 64));
  PhotonOut o;
  for (int a = 0; a < 8; ++a) o.v3[a] = g->po().f3[a].p;
  for (int a = 0; a < 4; ++a) o.f1[a] = g->po().f1[a].p;
  o.flags = g->po().flags.p;
  o.pathId = g->po().pathId.p;
  o.endN = beams ? g->po().endN.p : nullptr;
  uint64_t base = 0, stored = 0, paths = 0;
  for (int batch = 0; stored < capacity; ++batch) {
    if (batch > 4096) return GVPM_ERR_STATE;  // a scene that stores nothing
    // one round of waves (two per SIMD at this kernel's registers), each walking its share of the batch with refill
    const uint32_t chunk = std::max(64u, (((m + 2047u) / 2048u) + 63u) & ~63u);  // (1024 / 4096 / 8192 shares: the same)
    const unsigned nb = (m + chunk - 1) / chunk;
    hipLaunchKernelGGL(synth_walk_kernel, dim3(nb), dim3(64), 0, s, g->view, iteration, base, m, chunk, beams, g->park.p, stride,
                       g->counts.p, g->counted.p, g->nonEmpty.p);
    SY_TRY(exclusiveSumU32(g->scanTmp, g->counts.p, g->offs.p, m, s));
    SY_TRY(exclusiveSumU32(g->scanTmp, g->counted.p, g->countedOffs.p, m, s));
    SY_TRY(exclusiveSumU32(g->scanTmp, g->nonEmpty.p, g->neOffs.p, m, s));
    hipLaunchKernelGGL(synth_stop_kernel, dim3(1), dim3(1), 0, s, g->counts.p, g->offs.p, g->countedOffs.p, g->counted.p,
                       g->neOffs.p, g->nonEmpty.p, m, capacity, g->ctl.p);
    {
      const uint64_t slots = (uint64_t)m * (uint64_t)stride;
      hipLaunchKernelGGL(synth_place_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, s, g->park.p, stride, g->counts.p,
                         g->offs.p, g->neOffs.p, g->ctl.p, m, capacity, o);
    }
    uint32_t c[8];
    SY_TRY(hipMemcpyAsync(c, g->ctl.p, sizeof(c), hipMemcpyDeviceToHost, s));
    SY_TRY(hipStreamSynchronize(s));
    SY_TRY(hipGetLastError());
    base += c[3];
    stored = c[4];
    paths = c[5];
    // next batch continues from the totals
    const uint32_t next[3] = {c[4], c[5], c[6]};
    SY_TRY(hipMemcpyAsync(g->ctl.p, next, sizeof(next), hipMemcpyHostToDevice, s));
    SY_TRY(hipStreamSynchronize(s));
  }
  soa->pos = g->po().f3[0].p; soa->wi = g->po().f3[1].p; soa->flux = g->po().f3[2].p; soa->parent_pos = g->po().f3[3].p;
  soa->parent_n = g->po().f3[4].p; soa->prefix_w = g->po().f3[5].p; soa->parent_scat = g->po().f3[6].p; soa->parent_wi = g->po().f3[7].p;
  soa->parent_pdf = g->po().f1[0].p; soa->edge_pdf = g->po().f1[1].p; soa->parent_rr = g->po().f1[2].p; soa->parent_g = g->po().f1[3].p;
  soa->flags = g->po().flags.p;
  soa->path_id = g->po().pathId.p;
  soa->n = stored;
  if (endN) *endN = beams ? g->po().endN.p : nullptr;
  *nbPaths = paths;
  if (stored) g->pathsPerPhoton = (double)base / (double)stored;
  return GVPM_OK;
}

int gvpm_devgen_shoot_photons(gvpm_devgen *g, int iteration, uint64_t capacity, gvpm_photon_soa *dev_soa, uint64_t *nb_paths) {
  return shootCommon(g, iteration, capacity, 0, dev_soa, nullptr, nb_paths);
}

int gvpm_devgen_shoot_beams(gvpm_devgen *g, int iteration, uint64_t capacity, gvpm_photon_soa *dev_soa, const float **end_n_dev,
                           uint64_t *nb_paths) {
  if (!end_n_dev) return GVPM_ERR_INVALID_ARG;
  return shootCommon(g, iteration, capacity, 1, dev_soa, end_n_dev, nb_paths);
}

int gvpm_devgen_read(gvpm_devgen *g, const void *dev, void *host, uint64_t bytes) {
  if (!g || (bytes && (!dev || !host))) return GVPM_ERR_INVALID_ARG;
  (void)hipSetDevice(g->device);
  if (bytes) SY_TRY(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
  return GVPM_OK;
}

int gvpm_devgen_camera_beams(gvpm_devgen *g, int iteration, int tile_mod, int tile_rem, const gvpm_camera_ray **rays_dev,
                            uint64_t *n_sets) {
  if (!g || !rays_dev || !n_sets || tile_mod < 1 || tile_rem < 0 || tile_rem >= tile_mod) return GVPM_ERR_INVALID_ARG;
  (void)hipSetDevice(g->device);
  hipStream_t s = g->stream;
  const uint32_t npix = (uint32_t)g->view.width * (uint32_t)g->view.height;
  SY_TRY(g->pixFlag.ensure((size_t)npix + 1));
  SY_TRY(g->pixOffs.ensure((size_t)npix + 1));
  const unsigned nb = (npix + 63) / 64;
  SY_TRY(hipMemsetAsync(g->pixFlag.p + npix, 0, 4, s));
  hipLaunchKernelGGL(synth_beam_flag_kernel, dim3(nb), dim3(64), 0, s, g->view, iteration, tile_mod, tile_rem, npix,
                     g->pixFlag.p);
  SY_TRY(exclusiveSumU32(g->scanTmp, g->pixFlag.p, g->pixOffs.p, npix + 1, s));
  uint32_t total = 0;
  SY_TRY(hipMemcpyAsync(&total, g->pixOffs.p + npix, 4, hipMemcpyDeviceToHost, s));
  SY_TRY(hipStreamSynchronize(s));
  g->raysIdx = (g->raysIdx + 1) % 3;
  SY_TRY(g->raysOut[g->raysIdx].ensure((size_t)total * 5 + 5));
  hipLaunchKernelGGL(synth_beam_write_kernel, dim3(nb), dim3(64), 0, s, g->view, iteration, npix, g->pixFlag.p, g->pixOffs.p,
                     g->raysOut[g->raysIdx].p);
  SY_TRY(hipStreamSynchronize(s));
  SY_TRY(hipGetLastError());
  *rays_dev = g->raysOut[g->raysIdx].p;
  *n_sets = total;
  return GVPM_OK;
}

}  // extern "C"
