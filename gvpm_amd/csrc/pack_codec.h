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

}  // namespace gvpm
