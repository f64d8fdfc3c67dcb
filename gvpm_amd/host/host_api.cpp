#include "host_api.h"

#include <vector>

#include "synth.h"

struct gvpm_synth {
  gvpm::SynthScene scene;
  gvpm::PhotonBuffers photons;
  std::vector<gvpm_camera_ray> rays;
  std::vector<float> v0, e1, e2;
  std::vector<gvpm_vpm_sample> samples;
  std::vector<float> endN, w1, len1;
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
  gvpm::cameraBeams(s->scene, it, x0, y0, x1, y1, s->rays);
  *out = s->rays.data();
  return s->rays.size() / 5;
}

uint64_t gvpm_synth_beams_interleaved(gvpm_synth *s, int it, int tile_mod, int tile_rem, const gvpm_camera_ray **out) {
  if (!s || !out || tile_mod < 1 || tile_rem < 0 || tile_rem >= tile_mod) return 0;
  gvpm::cameraBeams(s->scene, it, 0, 0, s->scene.width, s->scene.height, s->rays, tile_mod, tile_rem);
  *out = s->rays.data();
  return s->rays.size() / 5;
}

uint64_t gvpm_synth_vpm_samples(gvpm_synth *s, int it, int nb_camera_samples, const gvpm_vpm_sample **out) {
  if (!s || !out || nb_camera_samples <= 0) return 0;
  gvpm::cameraSamplesVPM(s->scene, it, s->rays, nb_camera_samples, s->samples);
  *out = s->samples.data();
  return s->samples.size();
}
}
