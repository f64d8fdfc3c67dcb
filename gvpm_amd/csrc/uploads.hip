// Staging of the per-iteration inputs (C ABI, include/gvpm_hip.h): photon maps, photon beams / planes, camera beams and
// G-VPM samples, from pageable or pinned host memory (copy stream, three slots each, prefetch) or borrowed device memory.
// Reference seam: the flattening of GPhotonNodeData + Path at gvpm/gvpm_accel.h:31-59,119-199.
#include "context.h"
#include "pack_codec.h"

namespace {

// ---- packed records -> what the SoA uploads put in the slots (pack_codec.h) ----------------
__global__ __launch_bounds__(256) void unpack_photons_kernel(const gvpm_photon_packed *__restrict__ src, uint32_t n,
                                                             const gvpm_material *__restrict__ table, uint32_t table_n,
                                                             gvpm_photon_soa dst, unsigned long long *bad) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    // gvpm_unpack_photons refuses such a record; here it decodes as a black parent and is REPORTED (gvpm_get_stats fails)
    if (src[i].material >= table_n) atomicAdd(bad, 1ull);
    unpackPhoton(src[i], table, table_n, dst, i);
  }
}
__global__ __launch_bounds__(256) void unpack_rays_kernel(const gvpm_beam_set_packed *__restrict__ src, uint32_t nsets,
                                                          gvpm_camera_ray *__restrict__ dst) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nsets * 5u) dst[i] = unpackRay(src[i / 5u], (int)(i % 5u));
}

__global__ __launch_bounds__(256) void unpack_compact_rays_kernel(gvpm_sensor sensor, const gvpm_beam_set_compact *__restrict__ src,
                                                                  uint32_t nsets, gvpm_camera_ray *__restrict__ dst) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nsets * 5u) dst[i] = unpackCompactRay(sensor, src[i / 5u], (int)(i % 5u));
}

// linked records (pack_codec.h): first pass -- a wave takes 64 consecutive photons; the index of a photon's record within
// its kind's array is the group's base + the lanes of that kind below it
__global__ __launch_bounds__(256) void unpack_linked_kernel(const unsigned char *__restrict__ blob, const gvpm_material *__restrict__ table,
                                                            uint32_t table_n, gvpm_photon_soa dst, unsigned long long *bad) {
  const gvpm_linked_header hd = *reinterpret_cast<const gvpm_linked_header *>(blob);
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < hd.n;
  const uint32_t *kinds = reinterpret_cast<const uint32_t *>(blob + hd.off_kinds);
  const uint32_t kind = live ? linkedKind(kinds, i) : 3u;
  const int lane = threadIdx.x & 63;
  const unsigned long long below = (1ull << lane) - 1ull;
  const unsigned long long mF = __ballot(kind == GVPM_LINKED_FULL), mE = __ballot(kind == GVPM_LINKED_EMIT), mC = __ballot(kind == GVPM_LINKED_CHAIN);
  if (!live) return;
  const uint2 base = reinterpret_cast<const uint2 *>(blob + hd.off_groups)[i >> 6];
  if (kind == GVPM_LINKED_FULL) {
    const gvpm_photon_packed &r = reinterpret_cast<const gvpm_photon_packed *>(blob + hd.off_full)[base.x + (uint32_t)__popcll(mF & below)];
    if (r.material >= table_n) atomicAdd(bad, 1ull);
    unpackPhoton(r, table, table_n, dst, i);
  } else if (kind == GVPM_LINKED_EMIT) {
    const gvpm_photon_emit &r = reinterpret_cast<const gvpm_photon_emit *>(blob + hd.off_emit)[base.y + (uint32_t)__popcll(mE & below)];
    if ((r.flags >> 16) >= hd.n_emitters) atomicAdd(bad, 1ull);
    unpackEmit(r, reinterpret_cast<const gvpm_emitter_entry *>(blob + hd.off_emitters), hd.n_emitters, dst, i);
  } else {
    const uint32_t cbase = (i & ~63u) - base.x - base.y;
    const gvpm_photon_chain &r = reinterpret_cast<const gvpm_photon_chain *>(blob + hd.off_chain)[cbase + (uint32_t)__popcll(mC & below)];
    if ((r.flags >> 16) >= table_n || i == 0u) atomicAdd(bad, 1ull);
    unpackChainOwn(r, table, table_n, dst, i);
  }
}
// ... second pass: the chain records' links
__global__ __launch_bounds__(256) void link_linked_kernel(const unsigned char *__restrict__ blob, gvpm_photon_soa dst) {
  const gvpm_linked_header hd = *reinterpret_cast<const gvpm_linked_header *>(blob);
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0u || i >= hd.n) return;
  const uint32_t *kinds = reinterpret_cast<const uint32_t *>(blob + hd.off_kinds);
  if (linkedKind(kinds, i) != GVPM_LINKED_CHAIN) return;
  linkChain(dst, i, i >= 2u && linkedKind(kinds, i - 1u) == GVPM_LINKED_CHAIN);
}

}  // namespace

namespace gvpm {
void launch_unpack_linked(const uint32_t *blob, uint32_t n, const gvpm_material *table, uint32_t table_n, const gvpm_photon_soa &dst,
                          unsigned long long *bad, hipStream_t s) {
  if (!n) return;
  hipLaunchKernelGGL(unpack_linked_kernel, dim3((n + 255) / 256), dim3(256), 0, s, reinterpret_cast<const unsigned char *>(blob), table,
                     table_n, dst, bad);
  hipLaunchKernelGGL(link_linked_kernel, dim3((n + 255) / 256), dim3(256), 0, s, reinterpret_cast<const unsigned char *>(blob), dst);
}
void launch_unpack_compact_rays(const gvpm_sensor &sensor, const uint32_t *compact, uint32_t ncompact, gvpm_camera_ray *dst,
                                hipStream_t s) {
  if (ncompact)
    hipLaunchKernelGGL(unpack_compact_rays_kernel, dim3((ncompact * 5u + 255) / 256), dim3(256), 0, s, sensor,
                       reinterpret_cast<const gvpm_beam_set_compact *>(compact), ncompact, dst);
}
void launch_unpack_photons(const uint32_t *packed, uint32_t n, const gvpm_material *table, uint32_t table_n,
                           const gvpm_photon_soa &dst, unsigned long long *bad, hipStream_t s) {
  if (n)
    hipLaunchKernelGGL(unpack_photons_kernel, dim3((n + 255) / 256), dim3(256), 0, s,
                       reinterpret_cast<const gvpm_photon_packed *>(packed), n, table, table_n, dst, bad);
}
void launch_unpack_rays(const uint32_t *packed, uint32_t nsets, gvpm_camera_ray *dst, hipStream_t s) {
  if (nsets)
    hipLaunchKernelGGL(unpack_rays_kernel, dim3((nsets * 5u + 255) / 256), dim3(256), 0, s,
                       reinterpret_cast<const gvpm_beam_set_packed *>(packed), nsets, dst);
}
}  // namespace gvpm

namespace {

// unit vector -> octahedral 2 x snorm16 (host)
uint32_t octEncode(const float v[3]) {
  const double x = v[0], y = v[1], z = v[2];
  const double l1 = std::fabs(x) + std::fabs(y) + std::fabs(z);
  if (!(l1 > 0.0) || !std::isfinite(l1)) return GVPM_OCT_ZERO;
  double px = x / l1, py = y / l1;
  if (z < 0.0) {
    const double fx = (1.0 - std::fabs(py)) * (px >= 0.0 ? 1.0 : -1.0), fy = (1.0 - std::fabs(px)) * (py >= 0.0 ? 1.0 : -1.0);
    px = fx;
    py = fy;
  }
  const long ix = std::lrint(px * 32767.0), iy = std::lrint(py * 32767.0);
  return ((uint32_t)(int16_t)ix & 0xFFFFu) | ((uint32_t)(int16_t)iy << 16);
}

}  // namespace

extern "C" {

int gvpm_pack_photons(const gvpm_photon_soa *src, gvpm_photon_packed *dst, gvpm_material *table, uint32_t table_cap,
                      uint32_t *table_n) {
  if (!src || !table_n || (src->n && (!dst || !table))) return GVPM_ERR_INVALID_ARG;
  if (table_cap > 65536u) table_cap = 65536u;
  uint32_t nt = *table_n, last = 0;
  if (nt > table_cap) return GVPM_ERR_INVALID_ARG;
  for (uint64_t i = 0; i < src->n; ++i) {
    gvpm_photon_packed &r = dst[i];
    for (int c = 0; c < 3; ++c) {
      r.pos[c] = src->pos[3 * i + c];
      r.parent_pos[c] = src->parent_pos[3 * i + c];
      r.flux[c] = src->flux[3 * i + c];
      r.prefix_w[c] = src->prefix_w[3 * i + c];
    }
    r.parent_pdf = src->parent_pdf[i];
    r.edge_pdf = src->edge_pdf[i];
    r.parent_rr = src->parent_rr[i];
    r.parent_n_oct = octEncode(src->parent_n + 3 * i);
    r.parent_wi_oct = octEncode(src->parent_wi + 3 * i);
    r.flags = (src->flags[i] & ~(1u << 7)) | ((src->path_id[i] & 1u) << 7);
    gvpm_material m;
    m.scat[0] = src->parent_scat[3 * i];
    m.scat[1] = src->parent_scat[3 * i + 1];
    m.scat[2] = src->parent_scat[3 * i + 2];
    m.g = src->parent_g[i];
    // (consecutive photons of a light path mostly share their parent's material: the last hit first)
    uint32_t k = last;
    if (!(k < nt && memcmp(&table[k], &m, sizeof(m)) == 0)) {
      for (k = 0; k < nt; ++k)
        if (memcmp(&table[k], &m, sizeof(m)) == 0) break;
      if (k == nt) {
        if (nt >= table_cap) return GVPM_ERR_INVALID_ARG;
        table[nt++] = m;
      }
    }
    last = k;
    r.material = k;
  }
  *table_n = nt;
  return GVPM_OK;
}

int gvpm_unpack_photons(const gvpm_photon_packed *src, uint64_t n, const gvpm_material *table, uint32_t table_n,
                        const gvpm_photon_soa *dst) {
  if (!dst || (n && (!src || !dst->pos || !dst->wi || !dst->flux || !dst->parent_pos || !dst->parent_n || !dst->prefix_w ||
                     !dst->parent_scat || !dst->parent_wi || !dst->parent_pdf || !dst->edge_pdf || !dst->parent_rr ||
                     !dst->parent_g || !dst->flags || !dst->path_id)))
    return GVPM_ERR_INVALID_ARG;
  for (uint64_t i = 0; i < n; ++i) {
    if (src[i].material >= table_n) return GVPM_ERR_INVALID_ARG;
    unpackPhoton(src[i], table, table_n, *dst, i);
  }
  return GVPM_OK;
}

// ---- linked photon records (include/gvpm_hip.h) ----
static size_t align16(size_t x) { return (x + 15u) & ~(size_t)15u; }
static void linkedLayout(gvpm_linked_header &hd) {
  size_t off = sizeof(gvpm_linked_header);
  hd.off_kinds = (uint32_t)off;
  off = align16(off + (((size_t)hd.n + 15u) / 16u) * 4u);
  hd.off_groups = (uint32_t)off;
  off = align16(off + (((size_t)hd.n + 63u) / 64u) * 8u);
  hd.off_emitters = (uint32_t)off;
  off = align16(off + (size_t)hd.n_emitters * sizeof(gvpm_emitter_entry));
  hd.off_full = (uint32_t)off;
  off = align16(off + (size_t)hd.n_full * sizeof(gvpm_photon_packed));
  hd.off_emit = (uint32_t)off;
  off = align16(off + (size_t)hd.n_emit * sizeof(gvpm_photon_emit));
  hd.off_chain = (uint32_t)off;
  off = align16(off + (size_t)hd.n_chain * sizeof(gvpm_photon_chain));
  hd.bytes = (uint32_t)off;
}
size_t gvpm_linked_photons_bound(uint64_t n) {
  gvpm_linked_header hd{};
  hd.n = hd.n_full = (uint32_t)std::min<uint64_t>(n, 0x3FFFFFFFull);
  hd.n_emitters = 1024u;
  linkedLayout(hd);
  return (size_t)hd.bytes + 64u;
}

int gvpm_pack_photons_linked(const gvpm_photon_soa *src, void *blob, size_t cap, gvpm_material *table, uint32_t table_cap,
                             uint32_t *table_n, size_t *bytes) {
  if (!src || !table_n || !bytes || !blob || (src->n && !table)) return GVPM_ERR_INVALID_ARG;
  const uint64_t n = src->n;
  // 76 B x 2^25 photons keeps every offset below 2^32
  if (n > (1ull << 25)) return GVPM_ERR_INVALID_ARG;
  if (table_cap > 65536u) table_cap = 65536u;
  uint32_t nt = *table_n, lastM = 0, lastE = 0;
  if (nt > table_cap) return GVPM_ERR_INVALID_ARG;
  // pass 1: the kind of every photon (every short kind is VERIFIED against the source), the tables
  std::vector<uint8_t> kind(n, (uint8_t)GVPM_LINKED_FULL);
  std::vector<uint16_t> idx(n, 0);
  std::vector<gvpm_emitter_entry> emitters;
  auto material = [&](uint64_t i, uint32_t &k) -> bool {
    gvpm_material m;
    m.scat[0] = src->parent_scat[3 * i];
    m.scat[1] = src->parent_scat[3 * i + 1];
    m.scat[2] = src->parent_scat[3 * i + 2];
    m.g = src->parent_g[i];
    k = lastM;
    if (!(k < nt && memcmp(&table[k], &m, sizeof(m)) == 0)) {
      for (k = 0; k < nt; ++k)
        if (memcmp(&table[k], &m, sizeof(m)) == 0) break;
      if (k == nt) {
        if (nt >= table_cap) return false;
        table[nt++] = m;
      }
    }
    lastM = k;
    return true;
  };
  auto same3 = [](const float *a, const float *b) { return memcmp(a, b, 12) == 0; };
  const float zero3[3] = {0.f, 0.f, 0.f}, ex3[3] = {1.f, 0.f, 0.f};
  uint64_t nf = 0, ne = 0, nc = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const uint32_t fl = src->flags[i];
    const uint32_t ptype = GVPM_PF_PARENT_TYPE(fl), comp = GVPM_PF_PREV_COMPONENT(fl);
    bool done = false;
    if (i > 0 && ptype == GVPM_PARENT_MEDIUM && comp == (uint32_t)GVPM_BSDF_DIFFUSE_REFLECTION && src->path_id[i] == src->path_id[i - 1] &&
        same3(src->parent_pos + 3 * i, src->pos + 3 * (i - 1)) && same3(src->prefix_w + 3 * i, src->flux + 3 * (i - 1)) &&
        same3(src->parent_n + 3 * i, zero3)) {
      // the link's parent_wi (derived from the two positions before the photon) against the source's: within the octahedral
      // code's error of a full record, or the photon travels as one
      const float *pPrev = src->pos + 3 * (i - 1);
      const float *ppPrev = kind[i - 1] == GVPM_LINKED_CHAIN ? src->pos + 3 * (i - 2) : src->parent_pos + 3 * (i - 1);
      float w[3];
      gvpm::deriveWi(pPrev, ppPrev, w);
      const float *t = src->parent_wi + 3 * i;
      const double cx = (double)w[1] * t[2] - (double)w[2] * t[1], cy = (double)w[2] * t[0] - (double)w[0] * t[2], cz = (double)w[0] * t[1] - (double)w[1] * t[0];
      const double dt = (double)w[0] * t[0] + (double)w[1] * t[1] + (double)w[2] * t[2];
      uint32_t k;
      if (std::sqrt(cx * cx + cy * cy + cz * cz) <= 6e-5 && dt > 0.0) {
        if (!material(i, k)) return GVPM_ERR_INVALID_ARG;
        if (k < 65536u) {
          kind[i] = (uint8_t)GVPM_LINKED_CHAIN;
          idx[i] = (uint16_t)k;
          ++nc;
          done = true;
        }
      }
    }
    if (!done && ptype == GVPM_PARENT_EMITTER && comp == (uint32_t)GVPM_BSDF_DIFFUSE_REFLECTION && same3(src->parent_scat + 3 * i, zero3) &&
        same3(src->parent_wi + 3 * i, ex3)) {
      gvpm_emitter_entry e;
      memcpy(e.prefix_w, src->prefix_w + 3 * i, 12);
      e.parent_rr = src->parent_rr[i];
      memcpy(e.parent_n, src->parent_n + 3 * i, 12);
      e.parent_g = src->parent_g[i];
      uint32_t k = lastE;
      if (!(k < emitters.size() && memcmp(&emitters[k], &e, sizeof(e)) == 0)) {
        for (k = 0; k < emitters.size(); ++k)
          if (memcmp(&emitters[k], &e, sizeof(e)) == 0) break;
        if (k == emitters.size() && k < 1024u) emitters.push_back(e);
      }
      if (k < emitters.size()) {
        lastE = k;
        kind[i] = (uint8_t)GVPM_LINKED_EMIT;
        idx[i] = (uint16_t)k;
        ++ne;
        done = true;
      }
    }
    if (!done) ++nf;
  }
  gvpm_linked_header hd{};
  hd.magic = GVPM_LINKED_MAGIC;
  hd.n = (uint32_t)n;
  hd.n_full = (uint32_t)nf;
  hd.n_emit = (uint32_t)ne;
  hd.n_chain = (uint32_t)nc;
  hd.n_emitters = (uint32_t)emitters.size();
  linkedLayout(hd);
  if ((size_t)hd.bytes > cap) return GVPM_ERR_INVALID_ARG;
  unsigned char *B = static_cast<unsigned char *>(blob);
  memset(B, 0, hd.off_full);
  memcpy(B, &hd, sizeof(hd));
  uint32_t *kinds = reinterpret_cast<uint32_t *>(B + hd.off_kinds);
  uint32_t *groups = reinterpret_cast<uint32_t *>(B + hd.off_groups);
  if (!emitters.empty()) memcpy(B + hd.off_emitters, emitters.data(), emitters.size() * sizeof(gvpm_emitter_entry));
  gvpm_photon_packed *F = reinterpret_cast<gvpm_photon_packed *>(B + hd.off_full);
  gvpm_photon_emit *E = reinterpret_cast<gvpm_photon_emit *>(B + hd.off_emit);
  gvpm_photon_chain *Cn = reinterpret_cast<gvpm_photon_chain *>(B + hd.off_chain);
  uint64_t jf = 0, je = 0, jc = 0;
  for (uint64_t i = 0; i < n; ++i) {
    if ((i & 63u) == 0u) {
      groups[2 * (i >> 6)] = (uint32_t)jf;
      groups[2 * (i >> 6) + 1] = (uint32_t)je;
    }
    kinds[i >> 4] |= (uint32_t)kind[i] << (2u * (uint32_t)(i & 15u));
    const uint32_t lo = (src->flags[i] & 0xFF7Fu) | ((src->path_id[i] & 1u) << 7);
    if (kind[i] == GVPM_LINKED_CHAIN) {
      gvpm_photon_chain &r = Cn[jc++];
      memcpy(r.pos, src->pos + 3 * i, 12);
      memcpy(r.flux, src->flux + 3 * i, 12);
      r.parent_pdf = src->parent_pdf[i];
      r.edge_pdf = src->edge_pdf[i];
      r.parent_rr = src->parent_rr[i];
      r.flags = lo | ((uint32_t)idx[i] << 16);
    } else if (kind[i] == GVPM_LINKED_EMIT) {
      gvpm_photon_emit &r = E[je++];
      memcpy(r.pos, src->pos + 3 * i, 12);
      memcpy(r.parent_pos, src->parent_pos + 3 * i, 12);
      memcpy(r.flux, src->flux + 3 * i, 12);
      r.parent_pdf = src->parent_pdf[i];
      r.edge_pdf = src->edge_pdf[i];
      r.flags = lo | ((uint32_t)idx[i] << 16);
    } else {
      gvpm_photon_packed &r = F[jf++];
      for (int c = 0; c < 3; ++c) {
        r.pos[c] = src->pos[3 * i + c];
        r.parent_pos[c] = src->parent_pos[3 * i + c];
        r.flux[c] = src->flux[3 * i + c];
        r.prefix_w[c] = src->prefix_w[3 * i + c];
      }
      r.parent_pdf = src->parent_pdf[i];
      r.edge_pdf = src->edge_pdf[i];
      r.parent_rr = src->parent_rr[i];
      r.parent_n_oct = octEncode(src->parent_n + 3 * i);
      r.parent_wi_oct = octEncode(src->parent_wi + 3 * i);
      r.flags = (src->flags[i] & ~(1u << 7)) | ((src->path_id[i] & 1u) << 7);
      uint32_t k;
      if (!material(i, k)) return GVPM_ERR_INVALID_ARG;
      r.material = k;
    }
  }
  *table_n = nt;
  *bytes = hd.bytes;
  return GVPM_OK;
}

static bool linkedHeaderOk(const gvpm_linked_header &hd, size_t bytes) {
  gvpm_linked_header want = hd;
  linkedLayout(want);
  return hd.magic == GVPM_LINKED_MAGIC && (size_t)hd.n_full + hd.n_emit + hd.n_chain == hd.n && want.bytes == hd.bytes &&
         (size_t)hd.bytes <= bytes && want.off_kinds == hd.off_kinds && want.off_groups == hd.off_groups &&
         want.off_emitters == hd.off_emitters && want.off_full == hd.off_full && want.off_emit == hd.off_emit &&
         want.off_chain == hd.off_chain;
}

int gvpm_unpack_photons_linked(const void *blob, size_t bytes, const gvpm_material *table, uint32_t table_n, const gvpm_photon_soa *dst) {
  if (!blob || bytes < sizeof(gvpm_linked_header) || !dst) return GVPM_ERR_INVALID_ARG;
  const unsigned char *B = static_cast<const unsigned char *>(blob);
  gvpm_linked_header hd;
  memcpy(&hd, B, sizeof(hd));
  if (!linkedHeaderOk(hd, bytes)) return GVPM_ERR_INVALID_ARG;
  const uint32_t *kinds = reinterpret_cast<const uint32_t *>(B + hd.off_kinds);
  const gvpm_emitter_entry *emitters = reinterpret_cast<const gvpm_emitter_entry *>(B + hd.off_emitters);
  const gvpm_photon_packed *F = reinterpret_cast<const gvpm_photon_packed *>(B + hd.off_full);
  const gvpm_photon_emit *E = reinterpret_cast<const gvpm_photon_emit *>(B + hd.off_emit);
  const gvpm_photon_chain *Cn = reinterpret_cast<const gvpm_photon_chain *>(B + hd.off_chain);
  uint64_t jf = 0, je = 0, jc = 0;
  for (uint64_t i = 0; i < hd.n; ++i) {
    const uint32_t k = gvpm::linkedKind(kinds, i);
    if (k == GVPM_LINKED_FULL) {
      if (jf >= hd.n_full || F[jf].material >= table_n) return GVPM_ERR_INVALID_ARG;
      gvpm::unpackPhoton(F[jf++], table, table_n, *dst, i);
    } else if (k == GVPM_LINKED_EMIT) {
      if (je >= hd.n_emit || (E[je].flags >> 16) >= hd.n_emitters) return GVPM_ERR_INVALID_ARG;
      gvpm::unpackEmit(E[je++], emitters, hd.n_emitters, *dst, i);
    } else if (k == GVPM_LINKED_CHAIN) {
      if (jc >= hd.n_chain || i == 0 || (Cn[jc].flags >> 16) >= table_n) return GVPM_ERR_INVALID_ARG;
      gvpm::unpackChainOwn(Cn[jc++], table, table_n, *dst, i);
    } else {
      return GVPM_ERR_INVALID_ARG;
    }
  }
  for (uint64_t i = 1; i < hd.n; ++i)
    if (gvpm::linkedKind(kinds, i) == GVPM_LINKED_CHAIN) gvpm::linkChain(*dst, i, i >= 2 && gvpm::linkedKind(kinds, i - 1) == GVPM_LINKED_CHAIN);
  return GVPM_OK;
}

int gvpm_pack_camera_beams(const gvpm_camera_ray *rays, uint64_t n_sets, gvpm_beam_set_packed *dst) {
  if (n_sets && (!rays || !dst)) return GVPM_ERR_INVALID_ARG;
  for (uint64_t i = 0; i < n_sets; ++i) {
    const gvpm_camera_ray *s = rays + 5 * i;
    dst[i].base = s[0];
    for (int k = 1; k < 5; ++k) {
      if (GVPM_RAY_EDGE(s[k].info) != GVPM_RAY_EDGE(s[0].info)) return GVPM_ERR_INVALID_ARG;
      gvpm_ray_packed &q = dst[i].shifted[k - 1];
      for (int c = 0; c < 3; ++c) {
        q.o[c] = s[k].o[c];
        q.d[c] = s[k].d[c];
        q.eye[c] = s[k].eye[c];
      }
      const float l = std::fabs(s[k].len);
      q.len = GVPM_RAY_VALID(s[k].info) ? l : -l;  // (the sign BIT: -0 is an invalid ray of length 0)
      q.pdf = s[k].pdf;
      q.jacobian = s[k].jacobian;
      q.gop = s[k].gop;
    }
  }
  return GVPM_OK;
}

int gvpm_unpack_camera_beams(const gvpm_beam_set_packed *src, uint64_t n_sets, gvpm_camera_ray *dst) {
  if (n_sets && (!src || !dst)) return GVPM_ERR_INVALID_ARG;
  for (uint64_t i = 0; i < n_sets; ++i)
    for (int k = 0; k < 5; ++k) dst[5 * i + k] = unpackRay(src[i], k);
  return GVPM_OK;
}

int gvpm_unpack_camera_beams_compact(const gvpm_sensor *sensor, const gvpm_beam_set_compact *src, uint64_t n_compact,
                                     gvpm_camera_ray *dst) {
  if (!sensor || (n_compact && (!src || !dst))) return GVPM_ERR_INVALID_ARG;
  for (uint64_t i = 0; i < n_compact; ++i)
    for (int k = 0; k < 5; ++k) dst[5 * i + k] = unpackCompactRay(*sensor, src[i], k);
  return GVPM_OK;
}

int gvpm_pack_camera_beams_compact(const gvpm_sensor *sensor, const gvpm_camera_ray *rays, const float *jitter,
                                   uint64_t n_sets, gvpm_beam_set_compact *compact, uint64_t *n_compact,
                                   gvpm_beam_set_packed *full, uint64_t *n_full, uint32_t *new_index) {
  if (!sensor || !n_compact || !n_full || (n_sets && (!rays || !jitter || !compact || !full))) return GVPM_ERR_INVALID_ARG;
  uint64_t nc = 0, nf = 0;
  std::vector<uint8_t> isFull(new_index ? n_sets : 0);
  for (uint64_t i = 0; i < n_sets; ++i) {
    const gvpm_camera_ray *s = rays + 5 * i;
    gvpm_beam_set_compact c;
    memset(&c, 0, sizeof(c));
    c.pixel = s[0].pixel;
    c.jitter[0] = jitter[2 * i];
    c.jitter[1] = jitter[2 * i + 1];
    c.rand = s[0].rand;
    const uint32_t edge = GVPM_RAY_EDGE(s[0].info);
    c.info = edge << 8;
    bool ok = c.jitter[0] >= 0.f && c.jitter[0] < 1.f && c.jitter[1] >= 0.f && c.jitter[1] < 1.f;
    for (int k = 0; k < 5; ++k) {
      if (GVPM_RAY_EDGE(s[k].info) != edge) return GVPM_ERR_INVALID_ARG;
      if (!GVPM_RAY_VALID(s[k].info)) continue;
      c.info |= 1u << k;
      const double ox = (double)s[k].o[0] - sensor->pos[0], oy = (double)s[k].o[1] - sensor->pos[1], oz = (double)s[k].o[2] - sensor->pos[2];
      // (a ray that starts AT the sensor -- the sensor sits in the medium -- has t0 = 0 whatever the rounding of pos)
      const bool atSensor = (float)sensor->pos[0] == s[k].o[0] && (float)sensor->pos[1] == s[k].o[1] && (float)sensor->pos[2] == s[k].o[2];
      c.t0[k] = atSensor ? 0.f : (float)std::sqrt(ox * ox + oy * oy + oz * oz);
      c.len[k] = s[k].len;
    }
    // eligible iff the decode gives the rays back (to rounding) and the fields the format drops hold what it assumes
    for (int k = 0; k < 5 && ok; ++k) {
      const gvpm_camera_ray u = unpackCompactRay(*sensor, c, k);
      if (u.info != s[k].info) ok = false;
      if (!GVPM_RAY_VALID(s[k].info)) continue;
      for (int a = 0; a < 3 && ok; ++a) {
        const double tolD = 4.0 * 1.1920929e-7, tolO = 4.0 * 1.1920929e-7 * std::fmax(1.0, std::fabs((double)s[k].o[a]));
        if (std::fabs((double)u.d[a] - (double)s[k].d[a]) > tolD || std::fabs((double)u.o[a] - (double)s[k].o[a]) > tolO) ok = false;
        if (s[k].eye[a] != 1.f) ok = false;
      }
      if (u.len != s[k].len) ok = false;
      if (k > 0 && ok) {
        const double prod = (double)s[k].pdf / (double)s[0].pdf * (double)s[k].jacobian;
        if (!(std::fabs(prod - 1.0) <= 1e-4)) ok = false;
      }
    }
    if (ok && (!GVPM_RAY_VALID(s[0].info))) ok = false;  // (a set without a base ray is not a set)
    if (ok) {
      compact[nc++] = c;
    } else {
      const int rc = gvpm_pack_camera_beams(s, 1, full + nf);
      if (rc != GVPM_OK) return rc;
      ++nf;
    }
    if (new_index) isFull[i] = !ok;
  }
  if (new_index) {
    uint64_t a = 0, b = nc;
    for (uint64_t i = 0; i < n_sets; ++i) new_index[i] = (uint32_t)(isFull[i] ? b++ : a++);
  }
  *n_compact = nc;
  *n_full = nf;
  return GVPM_OK;
}

int gvpm_upload_materials(gvpm_context *h, const gvpm_material *table, uint32_t n) {
  CHECK_H(h);
  if (n && !table) return fail(h, GVPM_ERR_INVALID_ARG, "null material table");
  if (n > 65536u) return fail(h, GVPM_ERR_INVALID_ARG, "more than 65536 materials");
  // The unpack kernels that read the table run at the head of a gather's build chain (gather stream or build stream) and
  // may be in flight.  A table that only GROWS (the usual case: gvpm_pack_photons appends) is safe to extend under them --
  // no record in flight names an entry beyond the old count; an edit of existing entries, or a regrowth (which frees the
  // old buffer), first waits for every stream of the handle.
  const bool prefixSame = n >= h->nmaterials && h->nmaterials == (uint32_t)h->materialsHost.size() &&
                          (h->nmaterials == 0 || memcmp(h->materialsHost.data(), table, (size_t)h->nmaterials * sizeof(gvpm_material)) == 0);
  const bool regrow = h->materials.cap < (size_t)n + 1;
  if (regrow || !prefixSame) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->streamB));
    HIP_TRY(h, hipStreamSynchronize(h->copyStream));
  }
  if (regrow) HIP_TRY(h, h->materials.ensure(std::max<size_t>((size_t)n + 1, 64)));
  const uint32_t first = (regrow || !prefixSame) ? 0u : h->nmaterials;
  if (n > first)
    HIP_TRY(h, hipMemcpy(h->materials.p + first, table + first, (size_t)(n - first) * sizeof(gvpm_material), hipMemcpyHostToDevice));
  h->materialsHost.assign(table, table + n);
  h->nmaterials = n;
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
    ps.needUnpack = false;
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

// the packed twin of uploadPhotonsCommon: one copy of 76 n bytes into the slot.  The decode into the slot's SoA arrays is
// left to the gather that consumes the slot, at the head of its build (gvpm_gather): a kernel on the copy stream would
// sit between two copies and wait for compute units behind the gather kernels of the step in flight -- measured at C2,
// 4.7 ms a step that way against 3.6 for the SoA upload it was to beat
// (linkedBytes != 0: `src` is a blob of linked records of that size, n64 its photon count)
static int uploadPhotonsPacked(gvpm_context *h, const gvpm_photon_packed *src, uint64_t n64, bool prefetch, size_t linkedBytes = 0) {
  if (n64 && !src) return fail(h, GVPM_ERR_INVALID_ARG, "null packed photons");
  if (n64 > 0x7FFFFFF0ull) return fail(h, GVPM_ERR_INVALID_ARG, "too many photons");
  const uint32_t n = (uint32_t)n64;
  const bool pinned = n > 0 && isPinnedHost(src);
  if (prefetch && !pinned) return fail(h, GVPM_ERR_INVALID_ARG, "gvpm_prefetch_photons_packed needs pinned host memory (gvpm_host_alloc)");
  if (prefetch && h->phPending >= 0) return fail(h, GVPM_ERR_STATE, "a prefetched photon set is already pending");
  int slot = (h->phCur + 1) % 3;
  if (slot == h->phPending) slot = (h->phCur + 2) % 3;
  gvpm_context::PhotonSlot &ps = h->phSlot[slot];
  if (ps.read) {
    HIP_TRY(h, hipStreamWaitEvent(h->copyStream, ps.consumed, 0));
    HIP_TRY(h, hipStreamWaitEvent(h->copyStream, ps.consumedB, 0));
  }
  ps.read = false;
  constexpr size_t RW = sizeof(gvpm_photon_packed) / 4;
  const size_t packedWords = linkedBytes ? (linkedBytes + 3u) / 4u + 8u : (size_t)n * RW + 8;
  if (ps.raw.cap < (size_t)n * 30 + 8 || ps.packed.cap < packedWords) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->streamB));
    HIP_TRY(h, hipStreamSynchronize(h->copyStream));
    HIP_TRY(h, ps.raw.ensure((size_t)n * 30 + 8));
    // (a blob of linked records is never larger than the packed records of its photons + its tables: sized for those, so that
    // blobs of varying size do not regrow the slot)
    HIP_TRY(h, ps.packed.ensure(std::max(packedWords, (size_t)n * RW + 8 + 16384u)));
  }
  const void **dst[14] = {(const void **)&ps.dev.pos, (const void **)&ps.dev.wi, (const void **)&ps.dev.flux,
                          (const void **)&ps.dev.parent_pos, (const void **)&ps.dev.parent_n, (const void **)&ps.dev.prefix_w,
                          (const void **)&ps.dev.parent_scat, (const void **)&ps.dev.parent_wi, (const void **)&ps.dev.parent_pdf,
                          (const void **)&ps.dev.edge_pdf, (const void **)&ps.dev.parent_rr, (const void **)&ps.dev.parent_g,
                          (const void **)&ps.dev.flags, (const void **)&ps.dev.path_id};
  size_t off = 0;
  for (int k = 0; k < 14; ++k) {
    *dst[k] = ps.raw.p + off;
    off += (size_t)n * (k < 8 ? 3 : 1);
  }
  ps.dev.n = n;
  if (n) HIP_TRY(h, hipMemcpyAsync(ps.packed.p, src, linkedBytes ? linkedBytes : (size_t)n * sizeof(gvpm_photon_packed), hipMemcpyHostToDevice, h->copyStream));
  ps.needUnpack = n > 0;
  ps.linked = linkedBytes != 0;
  HIP_TRY(h, hipEventRecord(ps.copied, h->copyStream));
  if (!pinned) HIP_TRY(h, hipStreamSynchronize(h->copyStream));
  if (prefetch) {
    h->phPending = slot;
    return GVPM_OK;
  }
  h->phCur = slot;
  h->rawDev = ps.dev;
  h->phWait = true;
  h->photonsOwnedCur = true;
  h->nph = n;
  h->havePhotons = true;
  h->photonsDirty = true;
  return GVPM_OK;
}

int gvpm_upload_photons_packed(gvpm_context *h, const gvpm_photon_packed *photons, uint64_t n) {
  CHECK_H(h);
  return uploadPhotonsPacked(h, photons, n, false);
}
int gvpm_prefetch_photons_packed(gvpm_context *h, const gvpm_photon_packed *photons, uint64_t n) {
  CHECK_H(h);
  return uploadPhotonsPacked(h, photons, n, true);
}
static int uploadLinked(gvpm_context *h, const void *blob, size_t bytes, bool prefetch) {
  if (!blob || bytes < sizeof(gvpm_linked_header)) return fail(h, GVPM_ERR_INVALID_ARG, "null or short blob of linked photon records");
  gvpm_linked_header hd;
  memcpy(&hd, blob, sizeof(hd));
  if (!linkedHeaderOk(hd, bytes)) return fail(h, GVPM_ERR_INVALID_ARG, "not a blob of gvpm_pack_photons_linked (header / sizes)");
  if (hd.n == 0u) return uploadPhotonsPacked(h, nullptr, 0, prefetch);
  return uploadPhotonsPacked(h, static_cast<const gvpm_photon_packed *>(blob), hd.n, prefetch, hd.bytes);
}
int gvpm_upload_photons_linked(gvpm_context *h, const void *blob, size_t bytes) {
  CHECK_H(h);
  return uploadLinked(h, blob, bytes, false);
}
int gvpm_prefetch_photons_linked(gvpm_context *h, const void *blob, size_t bytes) {
  CHECK_H(h);
  return uploadLinked(h, blob, bytes, true);
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
    rs.needUnpack = false;
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

static int uploadBeamsPacked(gvpm_context *h, const gvpm_beam_set_packed *src, uint64_t nsets, bool prefetch) {
  if (nsets && !src) return fail(h, GVPM_ERR_INVALID_ARG, "null packed beam sets");
  if (nsets > 0x0FFFFFFFull) return fail(h, GVPM_ERR_INVALID_ARG, "too many beam sets");
  const bool pinned = nsets && isPinnedHost(src);
  if (prefetch && !pinned) return fail(h, GVPM_ERR_INVALID_ARG, "gvpm_prefetch_camera_beams_packed needs pinned host memory (gvpm_host_alloc)");
  if (prefetch && h->rayPending >= 0) return fail(h, GVPM_ERR_STATE, "a prefetched camera-beam list is already pending");
  int slot = (h->rayCur + 1) % 3;
  if (slot == h->rayPending) slot = (h->rayCur + 2) % 3;
  gvpm_context::RaySlot &rs = h->raySlot[slot];
  if (rs.read) HIP_TRY(h, hipStreamWaitEvent(h->copyStream, rs.freed, 0));
  rs.read = false;
  constexpr size_t SW = sizeof(gvpm_beam_set_packed) / 4;
  if (rs.rays.cap < (size_t)nsets * 5 + 1 || rs.packed.cap < (size_t)nsets * SW + 8) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));  // regrowing frees the old buffers
    HIP_TRY(h, hipStreamSynchronize(h->copyStream));
    HIP_TRY(h, rs.rays.ensure((size_t)nsets * 5 + 1));
    HIP_TRY(h, rs.packed.ensure((size_t)nsets * SW + 8));
  }
  if (nsets)
    HIP_TRY(h, hipMemcpyAsync(rs.packed.p, src, (size_t)nsets * sizeof(gvpm_beam_set_packed), hipMemcpyHostToDevice, h->copyStream));
  rs.needUnpack = nsets > 0;
  rs.nsets = (uint32_t)nsets;
  rs.ncompact = 0;
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
  h->nsets = (uint32_t)nsets;
  h->haveBeams = true;
  h->beamsDirty = true;
  return GVPM_OK;
}

// compact sets (60 bytes, rebuilt from the sensor) + the full packed sets of deeper edges: two copies into the slot,
// decoded by the consuming gather like the packed records
static int uploadBeamsCompact(gvpm_context *h, const gvpm_beam_set_compact *compact, uint64_t ncompact,
                              const gvpm_beam_set_packed *full, uint64_t nfull, bool prefetch) {
  if ((ncompact && !compact) || (nfull && !full)) return fail(h, GVPM_ERR_INVALID_ARG, "null compact / full beam sets");
  const uint64_t nsets = ncompact + nfull;
  if (nsets > 0x0FFFFFFFull) return fail(h, GVPM_ERR_INVALID_ARG, "too many beam sets");
  if (ncompact && !h->haveSensor) return fail(h, GVPM_ERR_STATE, "compact beam sets need gvpm_upload_sensor first");
  const bool pinned = nsets && (!ncompact || isPinnedHost(compact)) && (!nfull || isPinnedHost(full));
  if (prefetch && !pinned) return fail(h, GVPM_ERR_INVALID_ARG, "gvpm_prefetch_camera_beams_compact needs pinned host memory (gvpm_host_alloc)");
  if (prefetch && h->rayPending >= 0) return fail(h, GVPM_ERR_STATE, "a prefetched camera-beam list is already pending");
  int slot = (h->rayCur + 1) % 3;
  if (slot == h->rayPending) slot = (h->rayCur + 2) % 3;
  gvpm_context::RaySlot &rs = h->raySlot[slot];
  if (rs.read) HIP_TRY(h, hipStreamWaitEvent(h->copyStream, rs.freed, 0));
  rs.read = false;
  constexpr size_t SW = sizeof(gvpm_beam_set_packed) / 4, CW = sizeof(gvpm_beam_set_compact) / 4;
  if (rs.rays.cap < (size_t)nsets * 5 + 1 || rs.packed.cap < (size_t)nfull * SW + 8 || rs.compact.cap < (size_t)ncompact * CW + 8) {
    HIP_TRY(h, hipStreamSynchronize(h->stream));  // regrowing frees the old buffers
    HIP_TRY(h, hipStreamSynchronize(h->copyStream));
    HIP_TRY(h, rs.rays.ensure((size_t)nsets * 5 + 1));
    HIP_TRY(h, rs.packed.ensure((size_t)nfull * SW + 8));
    HIP_TRY(h, rs.compact.ensure((size_t)ncompact * CW + 8));
  }
  if (ncompact)
    HIP_TRY(h, hipMemcpyAsync(rs.compact.p, compact, (size_t)ncompact * sizeof(gvpm_beam_set_compact), hipMemcpyHostToDevice, h->copyStream));
  if (nfull)
    HIP_TRY(h, hipMemcpyAsync(rs.packed.p, full, (size_t)nfull * sizeof(gvpm_beam_set_packed), hipMemcpyHostToDevice, h->copyStream));
  rs.needUnpack = nsets > 0;
  rs.nsets = (uint32_t)nsets;
  rs.ncompact = (uint32_t)ncompact;
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
  h->nsets = (uint32_t)nsets;
  h->haveBeams = true;
  h->beamsDirty = true;
  return GVPM_OK;
}

int gvpm_upload_camera_beams_compact(gvpm_context *h, const gvpm_beam_set_compact *compact, uint64_t n_compact,
                                     const gvpm_beam_set_packed *full, uint64_t n_full) {
  CHECK_H(h);
  return uploadBeamsCompact(h, compact, n_compact, full, n_full, false);
}
int gvpm_prefetch_camera_beams_compact(gvpm_context *h, const gvpm_beam_set_compact *compact, uint64_t n_compact,
                                       const gvpm_beam_set_packed *full, uint64_t n_full) {
  CHECK_H(h);
  return uploadBeamsCompact(h, compact, n_compact, full, n_full, true);
}

int gvpm_upload_sensor(gvpm_context *h, const gvpm_sensor *sensor) {
  CHECK_H(h);
  if (!sensor) return fail(h, GVPM_ERR_INVALID_ARG, "null sensor");
  if (sensor->width < 1 || sensor->height < 1 || sensor->width > 65535 || sensor->height > 65535 ||
      !(sensor->tan_half_fov_x > 0.0) || !(sensor->tan_half_fov_y > 0.0))
    return fail(h, GVPM_ERR_INVALID_ARG, "sensor: film size in 1..65535 and positive field of view");
  for (int k = 0; k < 3; ++k) {
    // rows of a rotation: unit length, mutually orthogonal (a sheared or scaled transform is not a pinhole this format covers)
    const double *r = sensor->to_world + 3 * k, *q = sensor->to_world + 3 * ((k + 1) % 3);
    if (std::fabs(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] - 1.0) > 1e-9 || std::fabs(r[0] * q[0] + r[1] * q[1] + r[2] * q[2]) > 1e-9)
      return fail(h, GVPM_ERR_INVALID_ARG, "sensor: to_world must be a rotation");
  }
  // (the decode kernel takes the sensor by value at launch: nothing on the device to order against)
  h->sensor = *sensor;
  h->haveSensor = true;
  return GVPM_OK;
}

int gvpm_upload_camera_beams_packed(gvpm_context *h, const gvpm_beam_set_packed *sets, uint64_t n_sets) {
  CHECK_H(h);
  return uploadBeamsPacked(h, sets, n_sets, false);
}
int gvpm_prefetch_camera_beams_packed(gvpm_context *h, const gvpm_beam_set_packed *sets, uint64_t n_sets) {
  CHECK_H(h);
  return uploadBeamsPacked(h, sets, n_sets, true);
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

}  // extern "C"
