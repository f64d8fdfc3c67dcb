// ORACLE -- TEST INFRASTRUCTURE ONLY (see gvpm_oracle.hpp).
//
// The reference's acceleration structures for photon beams and photon planes, restated literally:
//   PointKDTree::build, EBalanced heuristic          include/mitsuba/core/kdtree.h:326-395, 925-1037
//   SimpleKDNode (left child = index + 1)            include/mitsuba/core/kdtree.h:44-112
//   SubBeamBVH (cut, build, buildHierarchy, query)   pm/beams_accel.h:82-267   (pm/ = src/integrators/photonmapper/)
//   PhotonBeam::nbSubBeams                           pm/beams_struct.h:316-318
//   PhotonPlaneBVH (build, buildHierarchy, query)    pm/plane_accel.h:85-207
//   PhotonPlane::getCenter / getAABB                 pm/plane_struct.h:56-66
// Both are a kd-tree over representative points (sub-beam centres / plane centres), whose nodes then receive a
// bounding box = own primitive's box united with the children's (bottom-up), walked with a stack and
// AABB::rayIntersect; the functor is invoked for EVERY node whose box the ray meets, inner nodes included -- it
// carries the whole intersection test, so the structure only prunes (tests/test_oracle_accel.py: the walk and the
// loop over all primitives give the same evaluations).
#pragma once

#include <algorithm>
#include <cstdint>
#include <vector>

#include "gvpm_oracle.hpp"

namespace oracle {

// PointKDTree<SimpleKDNode<...>>::build(): indirection table, recursive build(depth, base, rangeStart, rangeEnd),
// then permute_inplace.  After the build slot i holds the point order[i]; the left child of an inner node is the next
// slot, the right child right[i] (0: none).  MTS_PHOTONMAP_LEFT_BALANCED is 0 (include/mitsuba/render/photon.h:33).
template <typename F> struct PointKD {
  enum Heuristic { EBalanced = 0, ESlidingMidpoint = 2 };
  std::vector<uint32_t> order, right;
  std::vector<uint8_t> leaf;
  size_t depth = 0;

  void build(const std::vector<Vec3<F>> &pts, Heuristic h) {
    const size_t n = pts.size();
    order.resize(n);
    right.assign(n, 0);
    leaf.assign(n, 0);
    depth = 0;
    if (n == 0) return;
    AABB<F> box;
    for (const Vec3<F> &p : pts) box.expandBy(p);  // push_back, kdtree.h:305-308
    std::vector<uint32_t> ind(n), rightTmp(n, 0);
    std::vector<uint8_t> leafTmp(n, 0);
    for (size_t i = 0; i < n; ++i) ind[i] = (uint32_t)i;
    rec(1, pts, ind, 0, n, box, h, rightTmp, leafTmp);
    for (size_t i = 0; i < n; ++i) {  // permute_inplace(&m_nodes[0], indirection)
      order[i] = ind[i];
      right[i] = rightTmp[ind[i]];
      leaf[i] = leafTmp[ind[i]];
    }
  }

 private:
  void rec(size_t d, const std::vector<Vec3<F>> &pts, std::vector<uint32_t> &ind, size_t rangeStart, size_t rangeEnd,
           AABB<F> &box, Heuristic h, std::vector<uint32_t> &rightTmp, std::vector<uint8_t> &leafTmp) {
    depth = std::max(d, depth);
    const size_t count = rangeEnd - rangeStart;
    if (count == 1) {
      leafTmp[ind[rangeStart]] = 1;
      return;
    }
    const int axis = box.getLargestAxis();
    size_t split;
    if (h == EBalanced) {
      split = rangeStart + count / 2;  // kdtree.h:940-945
    } else {
      const F midpoint = (F)0.5f * (box.max[axis] + box.min[axis]);
      size_t nLT = 0;
      for (size_t i = rangeStart; i < rangeEnd; ++i)
        if (pts[ind[i]][axis] <= midpoint) nLT++;
      split = rangeStart + nLT;
      if (split == rangeStart) ++split;
      else if (split == rangeEnd) --split;
    }
    std::nth_element(ind.begin() + rangeStart, ind.begin() + split, ind.begin() + rangeEnd,
                     [&](uint32_t a, uint32_t b) { return pts[a][axis] < pts[b][axis]; });
    const uint32_t splitNode = ind[split];
    leafTmp[splitNode] = 0;
    rightTmp[splitNode] = split + 1 != rangeEnd ? (uint32_t)(split + 1) : 0u;
    std::swap(ind[rangeStart], ind[split]);
    F temp = box.max[axis];
    const F splitPos = pts[splitNode][axis];
    box.max.at(axis) = splitPos;
    rec(d + 1, pts, ind, rangeStart + 1, split + 1, box, h, rightTmp, leafTmp);
    box.max.at(axis) = temp;
    if (split + 1 != rangeEnd) {
      temp = box.min[axis];
      box.min.at(axis) = splitPos;
      rec(d + 1, pts, ind, split + 1, rangeEnd, box, h, rightTmp, leafTmp);
      box.min.at(axis) = temp;
    }
  }
};

// The stack walk both structures share (beams_accel.h:170-203, plane_accel.h:138-167): the query ray is re-based at
// r(r.mint) with extent [0, r.maxt - r.mint]; visit(slot) is called for every node whose box is met.
template <typename F, typename Visit>
inline void bvhWalk(const PointKD<F> &kd, const std::vector<AABB<F>> &boxes, const Ray<F> &r, Visit &&visit) {
  if (boxes.empty()) return;
  const Ray<F> ray(r(r.mint), r.d, (F)0, r.maxt - r.mint);
  // (one stack per thread, grown on demand: an allocation per query serialised 256 threads in malloc -- 6x on 256)
    static thread_local std::vector<uint32_t> stackStorage;
    if (stackStorage.size() < (size_t)kd.depth + 2) stackStorage.resize((size_t)kd.depth + 2);
  uint32_t *stack = stackStorage.data();
  uint32_t index = 0, stackPos = 1;
  stack[0] = 0;
  while (stackPos > 0) {
    F mint, maxt;
    if (!boxes[index].rayIntersect(ray, mint, maxt) || maxt < ray.mint || mint > ray.maxt) {
      index = stack[--stackPos];
      continue;
    }
    const uint32_t cur = index;
    if (!kd.leaf[cur]) {
      if (kd.right[cur] != 0) stack[stackPos++] = kd.right[cur];
      index = cur + 1;
    } else {
      index = stack[--stackPos];
    }
    visit(cur);
  }
}

// bottom-up boxes: own box united with the children's (buildHierarchy of either structure)
template <typename F, typename OwnBox>
inline AABB<F> bvhFit(const PointKD<F> &kd, std::vector<AABB<F>> &boxes, uint32_t index, OwnBox &&own) {
  AABB<F> box = own(index);
  if (!kd.leaf[index]) {
    const uint32_t left = index + 1, right = kd.right[index];
    if (left) box.expandBy(bvhFit(kd, boxes, left, own));
    if (right) box.expandBy(bvhFit(kd, boxes, right, own));
  }
  boxes[index] = box;
  return box;
}

// SubBeamBVH<LTPhotonBeam>, pm/beams_accel.h:82-267.  `Beam` needs getPos(v), dir, length.
template <typename F, typename Beam> struct SubBeamBVHO {
  struct Sub {
    uint32_t beam;
    F t1, t2;
  };
  PointKD<F> kd;
  std::vector<Sub> subs;  // in kd order
  std::vector<AABB<F>> boxes;
  F subbeamSize = 0;

  void build(const std::vector<Beam> &beams, F radius) {
    subs.clear();
    boxes.clear();
    if (beams.empty()) return;
    // the size of the cut: a tenth of the average beam length (:94-104)
    F avgSize = 0;
    for (const Beam &b : beams) avgSize += b.length;
    avgSize /= (F)beams.size();
    subbeamSize = avgSize / 10;
    std::vector<Sub> raw;
    std::vector<Vec3<F>> centres;
    for (size_t j = 0; j < beams.size(); ++j) {
      const int nbSBeams = (int)std::ceil(beams[j].length / subbeamSize);  // nbSubBeams, beams_struct.h:316-318
      const F lengthSubBeams = beams[j].length / nbSBeams;
      for (int i = 0; i < nbSBeams; ++i) {
        centres.push_back(beams[j].getPos(0) + beams[j].dir * lengthSubBeams * (F)(i + 0.5));
        raw.push_back(Sub{(uint32_t)j, lengthSubBeams * i, lengthSubBeams * (i + 1)});
      }
    }
    kd.build(centres, PointKD<F>::EBalanced);
    subs.resize(raw.size());
    for (size_t i = 0; i < raw.size(); ++i) subs[i] = raw[kd.order[i]];
    boxes.assign(subs.size(), AABB<F>());
    if (subs.empty()) return;
    // buildHierarchy (:208-243): the sub-beam's end points, each inflated by the beam radius
    bvhFit(kd, boxes, 0, [&](uint32_t index) {
      const Sub &s = subs[index];
      const Beam &b = beams[s.beam];
      const Vec3<F> p1 = b.getPos(0) + b.dir * s.t1, p2 = b.getPos(0) + b.dir * s.t2, rv(radius, radius, radius);
      AABB<F> box(p1 - rv, p1 + rv);
      box.expandBy(AABB<F>(p2 - rv, p2 + rv));
      return box;
    });
  }

  // query(bRadQuery): functor(beam, t1, t2) for every node met (:170-203)
  template <typename Q> void query(const std::vector<Beam> &beams, const Ray<F> &baseCameraRay, Q &q) const {
    bvhWalk(kd, boxes, baseCameraRay, [&](uint32_t cur) { q(beams[subs[cur].beam], subs[cur].t1, subs[cur].t2); });
  }
};

// PhotonPlaneBVH<LTPhotonPlane>, pm/plane_accel.h:85-207.  `Plane` needs ori, w0, w1, length0, length1.
template <typename F, typename Plane> struct PhotonPlaneBVHO {
  PointKD<F> kd;
  std::vector<uint32_t> planeOf;  // kd slot -> plane
  std::vector<AABB<F>> boxes;

  void build(const std::vector<Plane> &planes) {
    boxes.clear();
    planeOf.clear();
    if (planes.empty()) return;
    std::vector<Vec3<F>> centres(planes.size());
    for (size_t j = 0; j < planes.size(); ++j) {
      const Plane &p = planes[j];  // getCenter, plane_struct.h:56-58
      centres[j] = p.ori + p.w0 * p.length0 * (F)0.5 + p.w1 * p.length1 * (F)0.5;
    }
    kd.build(centres, PointKD<F>::EBalanced);
    planeOf = kd.order;
    boxes.assign(planes.size(), AABB<F>());
    bvhFit(kd, boxes, 0, [&](uint32_t index) {
      const Plane &p = planes[planeOf[index]];  // getAABB, plane_struct.h:60-66
      AABB<F> box(p.ori, p.ori);
      box.expandBy(p.ori + p.w0 * p.length0);
      box.expandBy(p.ori + p.w1 * p.length1);
      box.expandBy(p.ori + p.w1 * p.length1 + p.w0 * p.length0);
      return box;
    });
  }

  template <typename Q> void query(const std::vector<Plane> &planes, const Ray<F> &baseCameraRay, Q &q) const {
    bvhWalk(kd, boxes, baseCameraRay, [&](uint32_t cur) { q(planes[planeOf[cur]]); });
  }
};

}  // namespace oracle
