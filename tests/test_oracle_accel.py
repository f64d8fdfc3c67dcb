"""The reference's beam and plane accelerators, restated in oracle/gvpm_oracle_accel.hpp (SubBeamBVH,
pm/beams_accel.h:82-267; PhotonPlaneBVH, pm/plane_accel.h:85-207), against the loops over all primitives
(BeamMap::query's ENoAccel branch, pm/beams.h:289-294): both structures only prune -- the functor carries the whole
intersection test and, for beams, the ownership rule that makes one sub-beam evaluate a (ray, beam) pair -- so the
evaluated set, hence every counter and every sum, must be the loop's."""
import numpy as np
import pytest

import oracle_lib as O
from gvpm_amd import abi
from test_oracle_beams import make_beam_case, TECHS
from test_oracle_planes import make_plane_case

COUNTERS = ("evaluations", "null_shifts", "diffuse_shifts", "failed_shifts")


@pytest.mark.parametrize("tech", TECHS)
@pytest.mark.parametrize("scene", ["cbox", "cbox_hg", "laser"])
@pytest.mark.parametrize("precision", [64, 32])
def test_sub_beam_bvh_walk_evaluates_what_the_loop_over_all_beams_does(tech, scene, precision):
    c = make_beam_case(scene, 20, 16, 4000, 3.0, technique=tech)
    args = (c.p, c.m, c.tris, c.beams, c.end_n, c.rays, c.r, 1, c.nb, precision)
    loop, lcnt, _ = O.gather_beams(*args)
    tm = {}
    walk, wcnt, _ = O.gather_beams(*args, use_accel=True, timing=tm)
    assert lcnt["evaluations"] > 1500
    # the walk visits a fraction of what the loop does (it prunes), and evaluates the same pairs
    assert wcnt["candidates"] < lcnt["candidates"]
    for k in COUNTERS:
        assert wcnt[k] == lcnt[k], (k, wcnt, lcnt)
    # the sums differ by their order only (the walk meets the beams in tree order)
    lum = loop[..., 0:3].mean()
    assert np.abs(walk - loop).max() <= (1e-12 if precision == 64 else 2e-4) * max(np.abs(loop).max(), lum)
    assert tm["build_s"] >= 0


def test_sub_beam_cut_is_a_tenth_of_the_average_length_and_ownership_is_unique():
    # the same pairs again with the beams cut at another size through the ENoAccel path: the ownership rule
    # (3D: tNear in (t1, t2), shift_volume_beams.h:213-220) makes the result independent of the cut
    c = make_beam_case("cbox", 20, 16, 4000, 3.0)
    args = (c.p, c.m, c.tris, c.beams, c.end_n, c.rays, c.r, 1, c.nb, 64)
    _, wcnt, _ = O.gather_beams(*args, use_accel=True)
    ln = np.linalg.norm(c.beams.pos.astype(np.float64) - c.beams.parent_pos.astype(np.float64), axis=1)
    _, ccnt, _ = O.gather_beams(*args, sub_beam_size=float(ln.mean() / 10))
    for k in COUNTERS:
        assert wcnt[k] == ccnt[k]


@pytest.mark.parametrize("precision", [64, 32])
@pytest.mark.parametrize("g", [0.0, 0.7])
def test_photon_plane_bvh_walk_evaluates_what_the_loop_over_all_planes_does(precision, g):
    c = make_plane_case(W=20, H=16, nplanes=3000)
    c.m.g = g
    loop, lcnt, _ = O.gather_planes(c.p, c.m, c.tris, c.beams, c.w1, c.len1, c.rays, 1, c.nb, precision)
    tm = {}
    walk, wcnt, _ = O.gather_planes(c.p, c.m, c.tris, c.beams, c.w1, c.len1, c.rays, 1, c.nb, precision, use_accel=True,
                                    timing=tm)
    assert lcnt["evaluations"] > 3000
    assert wcnt["candidates"] < lcnt["candidates"]
    # A plane's box is flat (zero thickness across the plane): AABB::rayIntersect meets it where the ray pierces the
    # plane, and the rounding of its slab parameters can leave out a pierce point that lies on the parallelogram's
    # boundary to the last bit -- a tree-dependent difference of measure zero, bounded here
    for k in COUNTERS:
        assert abs(wcnt[k] - lcnt[k]) <= max(2, 1e-5 * lcnt[k]), (k, wcnt, lcnt)
    if wcnt["evaluations"] == lcnt["evaluations"]:
        lum = loop[..., 0:3].mean()
        assert np.abs(walk - loop).max() <= (1e-12 if precision == 64 else 2e-4) * max(np.abs(loop).max(), lum)


def test_empty_maps_walk_nothing():
    c = make_beam_case("cbox", 8, 6, 200, 3.0)
    c.beams = c.beams.subset(np.zeros(0, np.int64))
    acc, cnt, _ = O.gather_beams(c.p, c.m, c.tris, c.beams, c.end_n[:0], c.rays, c.r, 1, c.nb, 64, use_accel=True)
    assert cnt["evaluations"] == 0 and not acc.any()
    c = make_plane_case(W=8, H=6, nplanes=200)
    acc, cnt, _ = O.gather_planes(c.p, c.m, c.tris, c.beams.subset(np.zeros(0, np.int64)), c.w1[:0], c.len1[:0], c.rays, 1,
                                  c.nb, 64, use_accel=True)
    assert cnt["evaluations"] == 0 and not acc.any()
