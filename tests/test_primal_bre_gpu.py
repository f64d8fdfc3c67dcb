"""The primal beam radiance estimate on the device (gvpm_gather_primal; gather_bre.hip evaluate_primal_kernel) against the
fp64 oracle's literal restatement of the reference's sppm pass (oracle/gvpm_oracle_primal.hpp): accepted pairs equal,
fluxVol to the parity bar, over the APA schedule, at C2's size on windows, and the refusals of the entry point."""
import numpy as np
import pytest

import cases
import oracle_lib as O
from gvpm_amd import abi, hip
from test_configs_gpu import pick_windows, window_of
from test_primal_bre import primal_case

pytestmark = pytest.mark.gpu


def device(c, iters=1, rays=None):
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    ref = None
    total = 0
    for it in range(1, iters + 1):
        ph, nb = (c.ph, c.nb) if it == 1 else c.sc.shoot_photons(it, c.ph.n)
        r = (c.rays if rays is None else rays) if it == 1 else c.sc.camera_beams(it)
        rad = ctx.radius()
        ctx.upload_photons(ph)
        ctx.upload_camera_beams(r)
        ctx.gather_primal(it, nb)
        ref, cnt = O.gather_primal_bre(c.p, c.m, c.tris, ph, r, rad, it, nb, 64, accum=ref)
        total += cnt["evaluations"]
    acc = ctx.download_accum().astype(np.float64)
    st = ctx.stats()
    ctx.close()
    return acc, st, ref, total


@pytest.mark.parametrize("tech", [abi.GVPM_VOL_BRE3D, abi.GVPM_VOL_BRE2D])
@pytest.mark.parametrize("scene", ["cbox", "cbox_hg", "cbox_mirror"])
def test_primal_matches_fp64_oracle(tech, scene):
    c = primal_case(scene, 40, 36, 30000, 2.5, vol_technique=tech, use_shift_null=0)
    acc, st, ref, total = device(c)
    assert st["evaluations"] == total > 10000
    assert st["null_shifts"] == st["diffuse_shifts"] == st["failed_shifts"] == 0
    lum = ref[..., 0:3].mean()
    assert np.sqrt(((acc - ref) ** 2).mean()) / lum < 1e-4
    assert not acc[..., 3:].any()


def test_three_iterations_radius_schedule_and_max_depth():
    c = primal_case("cbox", 32, 32, 20000, 3.0, max_depth=4)
    acc, st, ref, total = device(c, iters=3)
    assert st["evaluations"] == total
    assert np.sqrt(((acc - ref) ** 2).mean()) / ref[..., 0:3].mean() < 1e-4


def test_c2_size_windows():
    """BASELINE configs[1]'s inputs (512 x 512, 1 M photons) through the primal pass: two 32 x 32 windows against the oracle"""
    W = H = 512
    sc = cases.SynthScene("cbox", W, H)
    p = sc.params()
    p.initial_scale_volume = 1.0
    p.path_set = 0
    m, tris = sc.medium(), sc.triangles()
    ph, nb = sc.shoot_photons(1, 1_000_000)
    rays = sc.camera_beams(1)
    c = cases.Case()
    c.p, c.m, c.tris, c.ph, c.nb, c.rays, c.sc = p, m, tris, ph, nb, rays, sc
    acc, st, _, _ = device_noref(c)
    assert st["evaluations"] > 15_000_000
    for (x0, y0) in pick_windows(acc, 32):
        sel = window_of(rays, x0, y0, 32, 32)
        wr = np.ascontiguousarray(rays[sel])
        c.rays = wr
        wacc, wst, ref, total = device(c)
        win = (slice(y0, y0 + 32), slice(x0, x0 + 32))
        lum = ref[win][..., 0:3].mean()
        assert wst["evaluations"] == total > 20000
        assert np.sqrt(((wacc[win] - ref[win]) ** 2).mean()) / lum < 1e-4
        assert np.allclose(acc[win], wacc[win], rtol=2e-5, atol=1e-9 * lum)


def device_noref(c):
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    ctx.upload_photons(c.ph)
    ctx.upload_camera_beams(c.rays)
    ctx.gather_primal(1, c.nb)
    acc = ctx.download_accum().astype(np.float64)
    st = ctx.stats()
    ctx.close()
    return acc, st, None, None


def test_refusals():
    c = cases.make_case("cbox", 16, 12, 500, 3.0)  # path_set = 1: a filter the primal pass does not have
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    ctx.upload_photons(c.ph)
    ctx.upload_camera_beams(c.rays)
    with pytest.raises(hip.GvpmError) as e:
        ctx.gather_primal(1, c.nb)
    assert e.value.code == abi.GVPM_ERR_UNSUPPORTED
    ctx.close()


# ------------------------------------------------------------------------------------------------ the primal point estimate
from test_oracle_vpm import make_vpm_case  # noqa: E402


def device_vpm_primal(c, iters=1):
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    ref = sv = nv = None
    total = 0
    for it in range(1, iters + 1):
        if it == 1:
            ph, nb, r, smp = c.ph, c.nb, c.rays, c.samples
        else:
            ph, nb = c.sc.shoot_photons(it, c.ph.n)
            r, smp = c.sc.camera_beams_and_vpm_samples(it, c.p.nb_camera_samples)
        ctx.upload_photons(ph)
        ctx.upload_camera_beams(r)
        ctx.upload_vpm_samples(smp)
        ctx.gather_primal(it, nb)
        ref, sv, nv, cnt = O.gather_primal_vpm(c.p, c.m, c.tris, ph, r, smp, 64, use_accel=False, accum=ref, scale_vol=sv, n_vol=nv)
        total += cnt["evaluations"]
    acc = ctx.download_accum().astype(np.float64)
    st = ctx.stats()
    dsv, dnv = ctx.download_vpm_state()
    ctx.close()
    return acc, st, ref, total, (dsv, dnv, sv, nv)


@pytest.mark.parametrize("scene,kw", [("cbox", dict()), ("cbox_hg", dict()), ("cbox_mirror", dict(max_depth=3)), ("cbox", dict(max_depth=2))])
def test_primal_point_estimate_matches_fp64_oracle(scene, kw):
    c = make_vpm_case(scene, 32, 28, 40000, 5.0, nb=10, path_set=0, **kw)
    acc, st, ref, total, (dsv, dnv, sv, nv) = device_vpm_primal(c)
    assert st["evaluations"] == total > 5000
    assert st["null_shifts"] == st["diffuse_shifts"] == st["failed_shifts"] == 0
    assert np.sqrt(((acc - ref) ** 2).mean()) / ref[..., 0:3].mean() < 1e-4 and not acc[..., 3:].any()
    assert np.allclose(dsv, sv, rtol=1e-6) and np.allclose(dnv, nv, rtol=1e-6)  # M counts what passed radius AND depth


def test_primal_point_estimate_three_iterations_sppm_state():
    c = make_vpm_case("cbox", 24, 20, 30000, 5.0, nb=8, path_set=0)
    acc, st, ref, total, (dsv, dnv, sv, nv) = device_vpm_primal(c, iters=3)
    assert st["evaluations"] == total
    assert np.sqrt(((acc - ref) ** 2).mean()) / ref[..., 0:3].mean() < 1e-4
    assert np.allclose(dsv, sv, rtol=1e-6) and np.allclose(dnv, nv, rtol=1e-6)


# ------------------------------------------------------------------------------------------------ the primal beam x beam estimate
from test_oracle_beams import make_beam_case, TECHS  # noqa: E402


@pytest.mark.parametrize("tech", TECHS)
@pytest.mark.parametrize("scene", ["cbox", "cbox_hg", "laser"])
def test_primal_beams_match_fp64_oracle(tech, scene):
    c = make_beam_case(scene, 32, 28, 12000, 2.5, technique=tech, path_set=0)
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    ref = None
    total = 0
    for it in (1, 2):
        beams, en, nb = (c.beams, c.end_n, c.nb) if it == 1 else c.sc.shoot_beams(it, c.beams.n)
        r = c.rays if it == 1 else c.sc.camera_beams(it)
        rad = ctx.radius()
        ctx.upload_beams(beams, en)
        ctx.upload_camera_beams(r)
        ctx.gather_primal(it, nb)
        ref, cnt = O.gather_primal_beams(c.p, c.m, c.tris, beams, en, r, rad, it, nb, 64, accum=ref)
        total += cnt["evaluations"]
    acc = ctx.download_accum().astype(np.float64)
    st = ctx.stats()
    ctx.close()
    assert st["evaluations"] == total > 20000
    assert st["null_shifts"] == st["diffuse_shifts"] == st["failed_shifts"] == 0
    assert np.sqrt(((acc - ref) ** 2).mean()) / ref[..., 0:3].mean() < 2e-4 and not acc[..., 3:].any()
