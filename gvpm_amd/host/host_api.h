/* C ABI of libgvpm_host.so: synthetic hosts (scene, light paths, camera beams)
 * and the GPMIntegrator mirror.  Used by bench.py / tests through ctypes and
 * by C++ callers directly. */
#ifndef GVPM_HOST_API_H
#define GVPM_HOST_API_H
#include "../../include/gvpm_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct gvpm_synth gvpm_synth;
gvpm_synth *gvpm_synth_create(const char *scene, int width, int height, uint32_t seed);
void gvpm_synth_destroy(gvpm_synth *s);
/* the scene as the device-side generators take it (gvpm_devgen_create, include/gvpm_hip.h); the arrays stay owned by `s` */
int gvpm_synth_devgen_scene(gvpm_synth *s, gvpm_devgen_scene *out);
int gvpm_synth_params(const gvpm_synth *s, gvpm_params *out);
int gvpm_synth_medium(const gvpm_synth *s, gvpm_medium *out);
int gvpm_synth_triangles(gvpm_synth *s, gvpm_triangles *out);
/* shoots light paths of iteration `it` until `capacity` photons are stored;
 * *out points into buffers owned by `s` (valid until the next shoot) */
uint64_t gvpm_synth_shoot(gvpm_synth *s, int it, uint64_t capacity, gvpm_photon_soa *out,
                          uint64_t *nb_paths);
/* photon beams of iteration `it` (see gvpm_upload_beams); *end_n: 3 floats per beam */
uint64_t gvpm_synth_shoot_beams(gvpm_synth *s, int it, uint64_t capacity, gvpm_photon_soa *out,
                                const float **end_n, uint64_t *nb_paths);
/* the beam sets of the 4x4-pixel tiles t with t % tile_mod == tile_rem, whole frame (image-sharded ranks) */
uint64_t gvpm_synth_beams_interleaved(gvpm_synth *s, int it, int tile_mod, int tile_rem, const gvpm_camera_ray **out);
/* photon planes for the beams of the LAST gvpm_synth_shoot_beams call (see gvpm_upload_planes) */
uint64_t gvpm_synth_planes(gvpm_synth *s, int it, const float **w1, const float **len1);
/* camera beam sets of the pixel rectangle; returns the number of sets */
uint64_t gvpm_synth_beams(gvpm_synth *s, int it, int x0, int y0, int x1, int y1,
                          const gvpm_camera_ray **out);
/* the BSDF table of the scene's glossy (Phong) walls, in the order the photons' parent_g name them (gvpm_upload_bsdfs);
 * returns the number of entries (at most cap are written) */
uint32_t gvpm_synth_bsdfs(const gvpm_synth *s, gvpm_bsdf *out, uint32_t cap);
/* self-check of the streaming flattening the device generator uses (StreamPath, synth_core.h) against flattenPath /
 * flattenBeams on the same `n_paths` light paths of `iteration`: the number of paths whose records differ in any bit
 * (0 = identical); *n_records: records compared */
uint64_t gvpm_synth_stream_check(gvpm_synth *s, int iteration, uint64_t n_paths, int beams, uint64_t *n_records);
/* the scene's pinhole sensor as the compact beam sets take it (gvpm_upload_sensor, include/gvpm_hip.h) */
int gvpm_synth_sensor(const gvpm_synth *s, gvpm_sensor *out);
/* the fractional film offsets (2 floats per set) the base paths of `rays` (5 per set) of iteration `it` were sampled at:
 * the first two draws of the pixel's stream (cameraBeamSets, synth_core.h) -- what a Mitsuba host reads off its sample
 * position; the second argument of gvpm_pack_camera_beams_compact */
int gvpm_synth_jitter(const gvpm_synth *s, int it, const gvpm_camera_ray *rays, uint64_t n_sets, float *out);
/* G-VPM camera samples for the beam sets of the LAST gvpm_synth_beams call */
uint64_t gvpm_synth_vpm_samples(gvpm_synth *s, int it, int nb_camera_samples, const gvpm_vpm_sample **out);
#ifdef __cplusplus
}
#endif
#endif
