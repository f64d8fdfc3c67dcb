"""G-BRE's ray-bundle cells (Grid::mode 1, gvpm_amd/csrc/bundle_grid.h): for camera beams that leave one point the photons
are binned over the bundle's (u, v) plane by levels of angular size instead of a 3D grid.  The cells only select
candidates -- the hit test, hence the evaluated set, is the same -- so every counter but `candidates` must equal the 3D
grid's and the oracle's, bit for bit, and the sums may differ by their order only."""
import os

import numpy as np
import pytest

import cases
import oracle_lib as O
from gvpm_amd import abi, hip

pytestmark = pytest.mark.gpu
COUNTERS = ("evaluations", "null_shifts", "diffuse_shifts", "failed_shifts")


def run(c, bundle, div=None, rays=None, iters=1, p=None):
    env = {"GVPM_BUNDLE": "1" if bundle else "0"}
    if div is not None:
        env["GVPM_BUNDLE_DIV"] = str(div)
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ctx = hip.Context(c.p if p is None else p, device=0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    kinds = []
    for it in range(1, iters + 1):
        ctx.upload_photons(c.ph)
        ctx.upload_camera_beams(c.rays if rays is None else rays)
        ctx.gather(it, c.nb)
        kinds.append(ctx.grid_info()[0])
    acc = ctx.download_accum()
    st = ctx.stats()
    ctx.close()
    return acc, st, kinds


@pytest.mark.parametrize("scene", ["cbox", "cbox_hg", "laser"])
@pytest.mark.parametrize("div", [None, 0.5, 4])
def test_bundle_cells_select_what_the_3d_grid_selects(scene, div):
    c = cases.make_case(scene, 48, 40, 30000, 3.0)
    a3, s3, k3 = run(c, False)
    ab, sb, kb = run(c, True, div)
    assert k3 == [0] and kb == [1]
    assert s3["evaluations"] > 10000
    for k in COUNTERS:
        assert sb[k] == s3[k], (k, sb, s3)
    lum = max(a3[..., 0:3].mean(), 1e-30)
    assert np.abs(ab.astype(np.float64) - a3).max() <= 2e-4 * max(np.abs(a3).max(), lum)
    ref, cnt, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, c.it, c.nb, 64)
    for k in COUNTERS:
        assert sb[k] == cnt[k], (k, sb, cnt)


def test_radius_schedule_rebins_every_iteration_under_one_frame():
    c = cases.make_case("cbox", 40, 32, 20000, 3.0)
    a3, s3, k3 = run(c, False, iters=3)
    ab, sb, kb = run(c, True, iters=3)
    assert kb == [1, 1, 1] and k3 == [0, 0, 0]
    for k in COUNTERS:
        assert sb[k] == s3[k]
    assert np.abs(ab.astype(np.float64) - a3).max() <= 2e-4 * np.abs(a3).max()


def test_rays_outside_the_fitted_frame_fall_back_to_the_3d_grid_for_that_step():
    """the frame is fitted on the first upload; a later upload from another point (a moved sensor) is detected by the
    planner and that step is redone on the 3D grid, then the frame is fitted anew"""
    c = cases.make_case("cbox", 40, 32, 20000, 3.0)
    moved = c.rays.copy()
    moved["o"] += np.float32(0.01)
    os.environ["GVPM_BUNDLE"] = "1"
    try:
        ctx = hip.Context(c.p, device=0)
    finally:
        os.environ.pop("GVPM_BUNDLE")
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    ctx.upload_photons(c.ph)
    ctx.upload_camera_beams(c.rays)
    ctx.gather(1, c.nb)
    assert ctx.grid_info()[0] == 1
    first = ctx.stats()
    ctx.upload_camera_beams(moved)
    ctx.set_global_scale(c.p.initial_scale_volume)
    ctx.gather(2, c.nb)
    assert ctx.grid_info()[0] == 0          # redone on the 3D grid
    second = ctx.stats()
    ctx.upload_camera_beams(moved)
    ctx.set_global_scale(c.p.initial_scale_volume)
    ctx.gather(3, c.nb)
    assert ctx.grid_info()[0] == 1          # new frame
    third = ctx.stats()
    ctx.close()
    _, cnt, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, moved, c.r, 2, c.nb, 64)
    for k in COUNTERS:
        assert second[k] - first[k] == cnt[k]
        assert third[k] - second[k] == cnt[k]


def test_beams_that_are_no_bundle_keep_the_3d_grid():
    c = cases.make_case("cbox", 32, 24, 10000, 3.0)
    rays = c.rays.copy()
    rng = np.random.default_rng(5)
    # every beam set from elsewhere
    rays["o"] += rng.normal(0, 0.05, (rays.shape[0], 1, 3)).astype(np.float32)
    ab, sb, kb = run(c, True, rays=rays)
    a3, s3, k3 = run(c, False, rays=rays)
    assert kb == [0]
    for k in COUNTERS:
        assert sb[k] == s3[k]
    assert np.abs(ab.astype(np.float64) - a3).max() <= 2e-4 * np.abs(a3).max()


def test_c2_size_bundle_cells_and_packed_records_give_the_plain_runs_counters():
    """BASELINE configs[1] at its size (512x512, 1 M photons): the ray-bundle cells and the packed upload records only
    change HOW candidates are found and how the inputs travel -- all four counters equal the plain run's, the sums to
    the order of their atomics / the records' stated losses."""
    c = cases.make_case("cbox", 512, 512, 1000000, 1.0)
    a0, s0, k0 = run(c, False)
    a1, s1, k1 = run(c, True)
    assert k0 == [0] and k1 == [1] and s0["evaluations"] > 10_000_000
    for k in COUNTERS:
        assert s1[k] == s0[k], (k, s1, s0)
    assert s1["candidates"] != s0["candidates"]
    lum = max(a0[..., 0:3].mean(), 1e-30)
    assert np.sqrt(((a1.astype(np.float64) - a0) ** 2).mean()) / lum < 1e-6
    t = hip.MaterialTable()
    pk = hip.pack_photons(c.ph, t)
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    ctx.upload_materials(t)
    ctx.upload_photons_packed(pk)
    ctx.upload_camera_beams_packed(hip.pack_camera_beams(c.rays))
    ctx.gather(1, c.nb)
    a2 = ctx.download_accum().astype(np.float64)
    s2 = ctx.stats()
    ctx.close()
    assert s2["evaluations"] == s0["evaluations"]
    for k in ("null_shifts", "diffuse_shifts", "failed_shifts"):
        assert abs(s2[k] - s0[k]) <= max(2, 1e-5 * s0[k]), (k, s2[k], s0[k])
    assert np.sqrt(((a2 - a0) ** 2).mean()) / lum < 2e-5
