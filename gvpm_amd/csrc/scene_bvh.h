// Occluder BVH of the scene triangles (scene->rayIntersect any-hit call sites,
// shift_volume_photon.cpp:398, shift_volume_beams.cpp:421, gvpm_geoOps.h:59-68; the reference walks
// Mitsuba's SAH kd-tree, skdtree.h).  Built on the host at gvpm_upload_scene: median split on the
// largest centroid axis, leaves of <= 4 triangles, triangles reordered into leaf order.
// Node = 2 float4: {min.xyz, bits(first)} {max.xyz, bits(count)}; count == 0: children first, first+1.
#pragma once
#include <stdint.h>

#include <vector>

namespace gvpm {

struct BvhBuild {
  std::vector<float> nodes;      // 8 floats per node
  std::vector<uint32_t> order;   // order[k] = original index of the k-th triangle in leaf order
};

void buildSceneBvh(const float *v0, const float *e1, const float *e2, uint32_t n, BvhBuild &out);

}  // namespace gvpm
