#include "scene_bvh.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <numeric>

namespace gvpm {

namespace {

struct Builder {
  const float *v0, *e1, *e2;
  std::vector<float> cx, lo, hi;  // centroids 3n, bounds 3n each
  std::vector<uint32_t> idx;
  std::vector<float> nodes;

  static void putU(float &f, uint32_t u) { memcpy(&f, &u, 4); }

  uint32_t alloc() {
    nodes.resize(nodes.size() + 8);
    return (uint32_t)(nodes.size() / 8 - 1);
  }

  void bounds(uint32_t b, uint32_t e, float *mn, float *mx, float *cmn, float *cmx) const {
    for (int c = 0; c < 3; ++c) {
      mn[c] = cmn[c] = INFINITY;
      mx[c] = cmx[c] = -INFINITY;
    }
    for (uint32_t k = b; k < e; ++k) {
      const uint32_t t = idx[k];
      for (int c = 0; c < 3; ++c) {
        mn[c] = fminf(mn[c], lo[3 * t + c]);
        mx[c] = fmaxf(mx[c], hi[3 * t + c]);
        cmn[c] = fminf(cmn[c], cx[3 * t + c]);
        cmx[c] = fmaxf(cmx[c], cx[3 * t + c]);
      }
    }
  }

  void build(uint32_t node, uint32_t b, uint32_t e) {
    float mn[3], mx[3], cmn[3], cmx[3];
    bounds(b, e, mn, mx, cmn, cmx);
    // boxes are inflated a little so that fp32 slab tests stay conservative
    for (int c = 0; c < 3; ++c) {
      const float pad = 1e-5f * (fabsf(mn[c]) + fabsf(mx[c]) + (mx[c] - mn[c])) + 1e-7f;
      nodes[8 * node + c] = mn[c] - pad;
      nodes[8 * node + 4 + c] = mx[c] + pad;
    }
    const uint32_t n = e - b;
    int axis = 0;
    for (int c = 1; c < 3; ++c)
      if (cmx[c] - cmn[c] > cmx[axis] - cmn[axis]) axis = c;
    if (n <= 4 || !(cmx[axis] > cmn[axis])) {
      putU(nodes[8 * node + 3], b);
      putU(nodes[8 * node + 7], n);
      return;
    }
    const uint32_t mid = b + n / 2;
    std::nth_element(idx.begin() + b, idx.begin() + mid, idx.begin() + e,
                     [&](uint32_t x, uint32_t y) { return cx[3 * x + axis] < cx[3 * y + axis]; });
    const uint32_t left = alloc();
    alloc();
    putU(nodes[8 * node + 3], left);
    putU(nodes[8 * node + 7], 0u);
    build(left, b, mid);
    build(left + 1, mid, e);
  }
};

}  // namespace

void buildSceneBvh(const float *v0, const float *e1, const float *e2, uint32_t n, BvhBuild &out) {
  Builder B;
  B.v0 = v0; B.e1 = e1; B.e2 = e2;
  B.cx.resize(3 * (size_t)n); B.lo.resize(3 * (size_t)n); B.hi.resize(3 * (size_t)n);
  B.idx.resize(n);
  std::iota(B.idx.begin(), B.idx.end(), 0u);
  for (uint32_t t = 0; t < n; ++t)
    for (int c = 0; c < 3; ++c) {
      const float a = v0[3 * t + c], b = a + e1[3 * t + c], d = a + e2[3 * t + c];
      B.lo[3 * t + c] = fminf(a, fminf(b, d));
      B.hi[3 * t + c] = fmaxf(a, fmaxf(b, d));
      B.cx[3 * t + c] = (a + b + d) * (1.f / 3.f);
    }
  B.nodes.reserve(16 * (size_t)n + 16);
  const uint32_t root = B.alloc();
  if (n == 0) {
    for (int c = 0; c < 3; ++c) {
      B.nodes[c] = INFINITY;       // empty box: every slab test fails
      B.nodes[4 + c] = -INFINITY;
    }
    Builder::putU(B.nodes[3], 0u);  // never traversed: the device returns early when ntri == 0
    Builder::putU(B.nodes[7], 0u);
  } else {
    B.build(root, 0, n);
  }
  out.nodes.swap(B.nodes);
  out.order.swap(B.idx);
}

}  // namespace gvpm
