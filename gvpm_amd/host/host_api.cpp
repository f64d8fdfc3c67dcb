#include "host_api.h"

#include <vector>

#include "synth.h"

struct gvpm_synth {
  gvpm::SynthScene scene;
  gvpm::PhotonBuffers photons;
  std::vector<gvpm_camera_ray> rays;
  std::vector<float> v0, e1, e2;
  std::vector<gvpm_vpm_sample> samples;
  std::vector<float> selW;  // per beam set of `rays`: its weight in the G-VPM edge selection
  std::vector<float> endN, w1, len1;
  std::vector<double> dgTris, dgAlbedo;  // gvpm_devgen_scene arrays
  std::vector<int32_t> dgTriMat, dgMatKind;
};

extern "C" {

gvpm_synth *gvpm_synth_create(const char *scene, int width, int height, uint32_t seed) {
  if (!scene || width <= 0 || height <= 0 || width > 65535 || height > 65535) return nullptr;
  gvpm_synth *s = new gvpm_synth();
  if (!gvpm::makeScene(scene, width, height, seed, s->scene)) {
    delete s;
    return nullptr;
  }
  for (const auto &t : s->scene.tris) {
    for (int k = 0; k < 3; ++k) {
      s->v0.push_back((float)t.v0[k]);
      s->e1.push_back((float)t.e1[k]);
      s->e2.push_back((float)t.e2[k]);
    }
  }
  return s;
}

void gvpm_synth_destroy(gvpm_synth *s) { delete s; }

int gvpm_synth_devgen_scene(gvpm_synth *s, gvpm_devgen_scene *out) {
  if (!s || !out) return GVPM_ERR_INVALID_ARG;
  const gvpm::SynthScene &sc = s->scene;
  s->dgTris.clear(); s->dgAlbedo.clear(); s->dgTriMat.clear(); s->dgMatKind.clear();
  for (const auto &t : sc.tris) {
    const gvpm::V3 v[4] = {t.v0, t.e1, t.e2, t.n};
    for (const auto &q : v) { s->dgTris.push_back(q.x); s->dgTris.push_back(q.y); s->dgTris.push_back(q.z); }
    s->dgTriMat.push_back(t.mat);
  }
  for (const auto &m : sc.mats) {
    s->dgMatKind.push_back(m.kind);
    s->dgAlbedo.push_back(m.albedo.x); s->dgAlbedo.push_back(m.albedo.y); s->dgAlbedo.push_back(m.albedo.z);
  }
  out->n_tris = (uint32_t)sc.tris.size();
  out->n_mats = (uint32_t)sc.mats.size();
  out->tris = s->dgTris.data();
  out->tri_mat = s->dgTriMat.data();
  out->mat_kind = s->dgMatKind.data();
  out->mat_albedo = s->dgAlbedo.data();
  auto put = [](double *d, gvpm::V3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; };
  put(out->light_c, sc.lightC); put(out->light_u, sc.lightU); put(out->light_v, sc.lightV); put(out->light_n, sc.lightN);
  put(out->radiance, sc.radiance);
  out->light_area = sc.lightArea;
  out->medium = sc.medium;
  put(out->cam_pos, sc.camPos);
  out->tan_half_fov_x = sc.tanHalfFovX;
  out->width = sc.width; out->height = sc.height;
  out->seed = sc.seed;
  out->camera_inside = sc.cameraInside ? 1 : 0;
  out->max_depth = sc.maxDepth; out->rr_depth = sc.rrDepth; out->min_depth = sc.minDepth;
  out->camera_sphere = sc.cameraSphere;
  const gvpm::V3 col[3] = {sc.camX, sc.camY, sc.camZ};
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) out->cam_to_world[3 * r + c] = col[c][r];
  return GVPM_OK;
}

int gvpm_synth_params(const gvpm_synth *s, gvpm_params *out) {
  if (!s || !out) return GVPM_ERR_INVALID_ARG;
  gvpm::defaultParams(s->scene, *out);
  return GVPM_OK;
}

int gvpm_synth_medium(const gvpm_synth *s, gvpm_medium *out) {
  if (!s || !out) return GVPM_ERR_INVALID_ARG;
  *out = s->scene.medium;
  return GVPM_OK;
}

int gvpm_synth_triangles(gvpm_synth *s, gvpm_triangles *out) {
  if (!s || !out) return GVPM_ERR_INVALID_ARG;
  out->v0 = s->v0.data();
  out->e1 = s->e1.data();
  out->e2 = s->e2.data();
  out->n = (uint32_t)s->scene.tris.size();
  return GVPM_OK;
}

uint64_t gvpm_synth_shoot(gvpm_synth *s, int it, uint64_t capacity, gvpm_photon_soa *out,
                          uint64_t *nb_paths) {
  if (!s || !out) return 0;
  uint64_t np = gvpm::shootPhotons(s->scene, it, capacity, s->photons);
  if (nb_paths) *nb_paths = np;
  s->photons.view(*out);
  return s->photons.n;
}

uint64_t gvpm_synth_shoot_beams(gvpm_synth *s, int it, uint64_t capacity, gvpm_photon_soa *out,
                                const float **end_n, uint64_t *nb_paths) {
  if (!s || !out || !end_n) return 0;
  uint64_t np = gvpm::shootBeams(s->scene, it, capacity, s->photons, s->endN);
  if (nb_paths) *nb_paths = np;
  s->photons.view(*out);
  *end_n = s->endN.data();
  return s->photons.n;
}

uint64_t gvpm_synth_planes(gvpm_synth *s, int it, const float **w1, const float **len1) {
  if (!s || !w1 || !len1) return 0;
  gvpm::planesFromBeams(s->scene, it, s->photons, s->w1, s->len1);
  *w1 = s->w1.data();
  *len1 = s->len1.data();
  return s->photons.n;
}

uint64_t gvpm_synth_beams(gvpm_synth *s, int it, int x0, int y0, int x1, int y1,
                          const gvpm_camera_ray **out) {
  if (!s || !out) return 0;
  gvpm::cameraBeams(s->scene, it, x0, y0, x1, y1, s->rays, 1, 0, &s->selW);
  *out = s->rays.data();
  return s->rays.size() / 5;
}

uint64_t gvpm_synth_beams_interleaved(gvpm_synth *s, int it, int tile_mod, int tile_rem, const gvpm_camera_ray **out) {
  if (!s || !out || tile_mod < 1 || tile_rem < 0 || tile_rem >= tile_mod) return 0;
  gvpm::cameraBeams(s->scene, it, 0, 0, s->scene.width, s->scene.height, s->rays, tile_mod, tile_rem, &s->selW);
  *out = s->rays.data();
  return s->rays.size() / 5;
}

uint32_t gvpm_synth_bsdfs(const gvpm_synth *s, gvpm_bsdf *out, uint32_t cap) {
  if (!s) return 0;
  uint32_t n = 0;
  for (const auto &m : s->scene.mats) {
    const int entries = gvpm::bsdfEntries(m.kind, m.exponent);
    for (int c = 0; c < entries; ++c) {
      if (out && (uint32_t)(m.bsdf + c) < cap) {
        gvpm_bsdf &b = out[m.bsdf + c];
        memset(&b, 0, sizeof(b));
        b.specular[0] = (float)m.spec.x; b.specular[1] = (float)m.spec.y; b.specular[2] = (float)m.spec.z;
        b.exponent = (float)m.exponent;
        if (m.kind == gvpm::MAT_PHONG) {
          b.kind = GVPM_BSDF_PHONG;
          b.specular_sampling_weight = (float)m.specWeight;
          b.distribution = entries == 2 ? c + 1 : 0;  // (sampled component + 1; 0: both components, include/gvpm_hip.h)
        } else if (m.kind == gvpm::MAT_WARD) {
          b.kind = GVPM_BSDF_WARD;
          b.specular_sampling_weight = (float)m.specWeight;
          b.sample_visible = m.distribution;  // (the model variant, include/gvpm_hip.h)
        } else {
          b.kind = GVPM_BSDF_ROUGHCONDUCTOR;
          b.distribution = m.distribution;
          b.sample_visible = 0;  // (the host walk samples all normals, synth_core.h)
          b.eta[0] = (float)m.eta.x; b.eta[1] = (float)m.eta.y; b.eta[2] = (float)m.eta.z;
          b.k[0] = (float)m.k.x; b.k[1] = (float)m.k.y; b.k[2] = (float)m.k.z;
        }
      }
      ++n;
    }
  }
  return n;
}

uint64_t gvpm_synth_stream_check(gvpm_synth *s, int iteration, uint64_t n_paths, int beams, uint64_t *n_records) {
  using namespace gvpm;
  if (!s) return ~0ull;
  const SceneView sc = s->scene.view();
  uint64_t differ = 0, nrec = 0;
  LPath path;
  RecList a, b;
  for (uint64_t idx = 0; idx < n_paths; ++idx) {
    Philox r1(sc.seed, 0x11ffu, (uint32_t)iteration, (uint32_t)idx, (uint32_t)(idx >> 32)), r2 = r1;
    randomWalk(sc, r1, path);
    bool ca = true, cb;
    if (beams) ca = flattenBeams(sc, path, a);
    else flattenPath(sc, path, a);
    if (beams) {
      StreamPath<RecList, true> sp(sc, b);
      randomWalk(sc, r2, sp);
      cb = sp.finish();
    } else {
      StreamPath<RecList, false> sp(sc, b);
      randomWalk(sc, r2, sp);
      cb = sp.finish();
    }
    bool same = a.n == b.n && ca == cb;
    for (int q = 0; same && q < a.n; ++q) {
      const PhotonRec &x = a.r[q], &y = b.r[q];
      const V3 *vx[9] = {&x.pos, &x.wi, &x.flux, &x.parentPos, &x.parentN, &x.prefixW, &x.parentScat, &x.parentWi, &x.endN};
      const V3 *vy[9] = {&y.pos, &y.wi, &y.flux, &y.parentPos, &y.parentN, &y.prefixW, &y.parentScat, &y.parentWi, &y.endN};
      for (int k = 0; k < (beams ? 9 : 8); ++k) same = same && memcmp(vx[k], vy[k], sizeof(V3)) == 0;  // (endN: beams only)
      same = same && memcmp(&x.parentPdf, &y.parentPdf, 4) == 0 && memcmp(&x.edgePdf, &y.edgePdf, 4) == 0 &&
             memcmp(&x.parentRR, &y.parentRR, 4) == 0 && memcmp(&x.parentG, &y.parentG, 4) == 0 && x.flags == y.flags;
    }
    nrec += (uint64_t)a.n;
    if (!same) differ++;
  }
  if (n_records) *n_records = nrec;
  return differ;
}

int gvpm_synth_sensor(const gvpm_synth *s, gvpm_sensor *out) {
  if (!s || !out) return GVPM_ERR_INVALID_ARG;
  memset(out, 0, sizeof(*out));
  const gvpm::SynthScene &sc = s->scene;
  out->pos[0] = sc.camPos.x;
  out->pos[1] = sc.camPos.y;
  out->pos[2] = sc.camPos.z;
  const gvpm::V3 col[3] = {sc.camX, sc.camY, sc.camZ};  // identity for the axis-aligned scenes (they look along -z)
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) out->to_world[3 * r + c] = col[c][r];
  out->tan_half_fov_x = sc.tanHalfFovX;
  out->tan_half_fov_y = sc.tanHalfFovX * sc.height / sc.width;   // traceCamera's `ty`, same operations
  out->width = sc.width;
  out->height = sc.height;
  return GVPM_OK;
}

int gvpm_synth_jitter(const gvpm_synth *s, int it, const gvpm_camera_ray *rays, uint64_t n_sets, float *out) {
  if (!s || (n_sets && (!rays || !out))) return GVPM_ERR_INVALID_ARG;
  const gvpm::SynthScene &sc = s->scene;
  for (uint64_t i = 0; i < n_sets; ++i) {
    const uint32_t px = rays[5 * i].pixel & 0xFFFFu, py = rays[5 * i].pixel >> 16;
    gvpm::Philox rng(sc.seed, 0xca3eu, (uint32_t)it, (uint32_t)(py * sc.width + px));
    out[2 * i] = rng.next1D();
    out[2 * i + 1] = rng.next1D();
  }
  return GVPM_OK;
}

uint64_t gvpm_synth_vpm_samples(gvpm_synth *s, int it, int nb_camera_samples, const gvpm_vpm_sample **out) {
  if (!s || !out || nb_camera_samples <= 0) return 0;
  gvpm::cameraSamplesVPM(s->scene, it, s->rays, s->selW, nb_camera_samples, s->samples);
  *out = s->samples.data();
  return s->samples.size();
}
}
