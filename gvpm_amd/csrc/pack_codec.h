// The packed upload records (include/gvpm_hip.h, "packed uploads"): the decode that DEFINES them, compiled for the host
// (gvpm_unpack_*) and for the device (unpack kernels, uploads.hip) from this one text.  fp64, no contraction: the two
// sides produce the same bits.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/gvpm_hip.h"

#define GVPM_HD __host__ __device__ __forceinline__

namespace gvpm {

constexpr uint32_t GVPM_OCT_ZERO = 0x80008000u;

// unit vector <- octahedral 2 x snorm16
GVPM_HD void octDecode(uint32_t w, float out[3]) {
#pragma clang fp contract(off)
  if (w == GVPM_OCT_ZERO) {
    out[0] = out[1] = out[2] = 0.f;
    return;
  }
  double x = (double)(int16_t)(w & 0xFFFFu) / 32767.0, y = (double)(int16_t)(w >> 16) / 32767.0;
  const double z = 1.0 - fabs(x) - fabs(y);
  if (z < 0.0) {
    const double fx = (1.0 - fabs(y)) * (x >= 0.0 ? 1.0 : -1.0), fy = (1.0 - fabs(x)) * (y >= 0.0 ? 1.0 : -1.0);
    x = fx;
    y = fy;
  }
  const double len = sqrt(x * x + y * y + z * z);
  out[0] = (float)(x / len);
  out[1] = (float)(y / len);
  out[2] = (float)(z / len);
}

// -edge(c-1)->d: from the photon towards its parent vertex
GVPM_HD void deriveWi(const float pos[3], const float parent[3], float out[3]) {
#pragma clang fp contract(off)
  const double x = (double)parent[0] - (double)pos[0], y = (double)parent[1] - (double)pos[1], z = (double)parent[2] - (double)pos[2];
  const double len = sqrt(x * x + y * y + z * z);
  if (!(len > 0.0)) {
    out[0] = out[1] = out[2] = 0.f;
    return;
  }
  out[0] = (float)(x / len);
  out[1] = (float)(y / len);
  out[2] = (float)(z / len);
}

// one photon: record -> the 14 SoA fields (element i of each destination array)
GVPM_HD void unpackPhoton(const gvpm_photon_packed &r, const gvpm_material *table, uint32_t table_n, const gvpm_photon_soa &d,
                          uint64_t i) {
  float *pos = const_cast<float *>(d.pos) + 3 * i, *wi = const_cast<float *>(d.wi) + 3 * i;
  float *flux = const_cast<float *>(d.flux) + 3 * i, *pp = const_cast<float *>(d.parent_pos) + 3 * i;
  float *pn = const_cast<float *>(d.parent_n) + 3 * i, *pw = const_cast<float *>(d.prefix_w) + 3 * i;
  float *sc = const_cast<float *>(d.parent_scat) + 3 * i, *pwi = const_cast<float *>(d.parent_wi) + 3 * i;
  for (int c = 0; c < 3; ++c) {
    pos[c] = r.pos[c];
    pp[c] = r.parent_pos[c];
    flux[c] = r.flux[c];
    pw[c] = r.prefix_w[c];
  }
  deriveWi(r.pos, r.parent_pos, wi);
  octDecode(r.parent_n_oct, pn);
  octDecode(r.parent_wi_oct, pwi);
  const gvpm_material m = r.material < table_n ? table[r.material] : gvpm_material{{0.f, 0.f, 0.f}, 0.f};
  sc[0] = m.scat[0];
  sc[1] = m.scat[1];
  sc[2] = m.scat[2];
  const_cast<float *>(d.parent_g)[i] = m.g;
  const_cast<float *>(d.parent_pdf)[i] = r.parent_pdf;
  const_cast<float *>(d.edge_pdf)[i] = r.edge_pdf;
  const_cast<float *>(d.parent_rr)[i] = r.parent_rr;
  const_cast<uint32_t *>(d.flags)[i] = r.flags & ~(1u << 7);
  const_cast<uint32_t *>(d.path_id)[i] = (r.flags >> 7) & 1u;
}

// ---- linked photon records (include/gvpm_hip.h) ----
GVPM_HD uint32_t linkedKind(const uint32_t *kinds, uint64_t i) { return (kinds[i >> 4] >> (2u * (uint32_t)(i & 15u))) & 3u; }
// an emit record: everything of the parent but its position and pdfs comes from the blob's emitter table
GVPM_HD void unpackEmit(const gvpm_photon_emit &r, const gvpm_emitter_entry *emitters, uint32_t n_emitters, const gvpm_photon_soa &d,
                        uint64_t i) {
  float *pos = const_cast<float *>(d.pos) + 3 * i, *wi = const_cast<float *>(d.wi) + 3 * i;
  float *flux = const_cast<float *>(d.flux) + 3 * i, *pp = const_cast<float *>(d.parent_pos) + 3 * i;
  float *pn = const_cast<float *>(d.parent_n) + 3 * i, *pw = const_cast<float *>(d.prefix_w) + 3 * i;
  float *sc = const_cast<float *>(d.parent_scat) + 3 * i, *pwi = const_cast<float *>(d.parent_wi) + 3 * i;
  const uint32_t e = r.flags >> 16;
  const gvpm_emitter_entry t = e < n_emitters ? emitters[e] : gvpm_emitter_entry{{0.f, 0.f, 0.f}, 0.f, {0.f, 0.f, 0.f}, 0.f};
  for (int c = 0; c < 3; ++c) {
    pos[c] = r.pos[c];
    pp[c] = r.parent_pos[c];
    flux[c] = r.flux[c];
    pw[c] = t.prefix_w[c];
    pn[c] = t.parent_n[c];
    sc[c] = 0.f;
  }
  pwi[0] = 1.f;
  pwi[1] = pwi[2] = 0.f;
  deriveWi(r.pos, r.parent_pos, wi);
  const_cast<float *>(d.parent_g)[i] = t.parent_g;
  const_cast<float *>(d.parent_pdf)[i] = r.parent_pdf;
  const_cast<float *>(d.edge_pdf)[i] = r.edge_pdf;
  const_cast<float *>(d.parent_rr)[i] = t.parent_rr;
  const_cast<uint32_t *>(d.flags)[i] = (r.flags & 0xFF7Fu) | ((uint32_t)GVPM_BSDF_DIFFUSE_REFLECTION << 16);
  const_cast<uint32_t *>(d.path_id)[i] = (r.flags >> 7) & 1u;
}
// a chain record, first pass: what the record itself and the material table hold
GVPM_HD void unpackChainOwn(const gvpm_photon_chain &r, const gvpm_material *table, uint32_t table_n, const gvpm_photon_soa &d,
                            uint64_t i) {
  float *pos = const_cast<float *>(d.pos) + 3 * i, *flux = const_cast<float *>(d.flux) + 3 * i;
  float *pn = const_cast<float *>(d.parent_n) + 3 * i, *sc = const_cast<float *>(d.parent_scat) + 3 * i;
  const uint32_t mi = r.flags >> 16;
  const gvpm_material m = mi < table_n ? table[mi] : gvpm_material{{0.f, 0.f, 0.f}, 0.f};
  for (int c = 0; c < 3; ++c) {
    pos[c] = r.pos[c];
    flux[c] = r.flux[c];
    pn[c] = 0.f;
    sc[c] = m.scat[c];
  }
  const_cast<float *>(d.parent_g)[i] = m.g;
  const_cast<float *>(d.parent_pdf)[i] = r.parent_pdf;
  const_cast<float *>(d.edge_pdf)[i] = r.edge_pdf;
  const_cast<float *>(d.parent_rr)[i] = r.parent_rr;
  const_cast<uint32_t *>(d.flags)[i] = (r.flags & 0xFF7Fu) | ((uint32_t)GVPM_BSDF_DIFFUSE_REFLECTION << 16);
  const_cast<uint32_t *>(d.path_id)[i] = (r.flags >> 7) & 1u;
}
// ... second pass (every position and flux of the upload is decoded by then): the link to the previous photon.
// prevChain: photon i - 1 is a chain record itself -- its parent position is pos[i - 2] (its own link may not be written yet)
GVPM_HD void linkChain(const gvpm_photon_soa &d, uint64_t i, bool prevChain) {
  float *pp = const_cast<float *>(d.parent_pos) + 3 * i, *pw = const_cast<float *>(d.prefix_w) + 3 * i;
  float *pwi = const_cast<float *>(d.parent_wi) + 3 * i, *wi = const_cast<float *>(d.wi) + 3 * i;
  const float *prevPos = d.pos + 3 * (i - 1), *prevFlux = d.flux + 3 * (i - 1);
  const float *prevParent = prevChain ? d.pos + 3 * (i - 2) : d.parent_pos + 3 * (i - 1);
  float ppos[3];
  for (int c = 0; c < 3; ++c) {
    ppos[c] = prevPos[c];
    pp[c] = ppos[c];
    pw[c] = prevFlux[c];
  }
  deriveWi(ppos, prevParent, pwi);
  deriveWi(d.pos + 3 * i, ppos, wi);
}

// ray k (0: base, 1..4: shifted) of a packed beam set -> a full camera ray
GVPM_HD gvpm_camera_ray unpackRay(const gvpm_beam_set_packed &s, int k) {
  if (k == 0) return s.base;
  const gvpm_ray_packed &q = s.shifted[k - 1];
  gvpm_camera_ray r;
  for (int c = 0; c < 3; ++c) {
    r.o[c] = q.o[c];
    r.d[c] = q.d[c];
    r.eye[c] = q.eye[c];
  }
  uint32_t lenBits;
  memcpy(&lenBits, &q.len, 4);
  const uint32_t lenSign = lenBits >> 31;
  lenBits &= 0x7FFFFFFFu;
  memcpy(&r.len, &lenBits, 4);
  r.pdf = q.pdf;
  r.jacobian = q.jacobian;
  r.gop = q.gop;
  r.info = GVPM_RAY_INFO(lenSign ? 0u : 1u, GVPM_RAY_EDGE(s.base.info));
  r.rand = 0.f;
  r.pixel = 0u;
  return r;
}

// ray k (0: base, 1..4: L R T B) of a compact beam set: the first medium edge of the camera path through film position
// (px, py) + offset_k + jitter of a perspective sensor (shift_cameraPath.h:29-133 re-traces the base path through the
// offset pixel with the base sample's fractional position, vertex.cpp:345-346).  pdf = jacobian = gop = 1: see the header
// (sensorMIS, gvpm_struct.h:608-631, is their only reader and its value for such an edge is 1 by construction).
GVPM_HD gvpm_camera_ray unpackCompactRay(const gvpm_sensor &s, const gvpm_beam_set_compact &c, int k) {
#pragma clang fp contract(off)
  gvpm_camera_ray r;
  memset(&r, 0, sizeof(r));
  const uint32_t edge = (c.info >> 8) & 0xFFu;
  if (!((c.info >> k) & 1u)) {
    r.info = GVPM_RAY_INFO(0u, edge);
    return r;
  }
  const double offX = k == 1 ? -1.0 : (k == 2 ? 1.0 : 0.0), offY = k == 3 ? 1.0 : (k == 4 ? -1.0 : 0.0);
  const double sx = (double)(c.pixel & 0xFFFFu) + offX + (double)c.jitter[0];
  const double sy = (double)(c.pixel >> 16) + offY + (double)c.jitter[1];
  const double cx = (2.0 * sx / (double)s.width - 1.0) * s.tan_half_fov_x;
  const double cy = (2.0 * sy / (double)s.height - 1.0) * s.tan_half_fov_y;
  const double cz = -1.0;
  const double len = sqrt(cx * cx + cy * cy + cz * cz);
  const double ux = cx / len, uy = cy / len, uz = cz / len;
  const double dx = s.to_world[0] * ux + s.to_world[1] * uy + s.to_world[2] * uz;
  const double dy = s.to_world[3] * ux + s.to_world[4] * uy + s.to_world[5] * uz;
  const double dz = s.to_world[6] * ux + s.to_world[7] * uy + s.to_world[8] * uz;
  const double t0 = (double)c.t0[k];
  r.o[0] = (float)(s.pos[0] + dx * t0);
  r.o[1] = (float)(s.pos[1] + dy * t0);
  r.o[2] = (float)(s.pos[2] + dz * t0);
  r.d[0] = (float)dx;
  r.d[1] = (float)dy;
  r.d[2] = (float)dz;
  r.len = c.len[k];
  r.eye[0] = r.eye[1] = r.eye[2] = 1.f;
  r.pdf = r.jacobian = r.gop = 1.f;
  r.info = GVPM_RAY_INFO(1u, edge);
  if (k == 0) {
    r.rand = c.rand;
    r.pixel = c.pixel;
  }
  return r;
}

}  // namespace gvpm
