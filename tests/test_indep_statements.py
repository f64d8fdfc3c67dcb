"""The oracle's SHIFTED terms against a second, independent statement (tests/indep_statements.py: numpy, written from the
reference sources): base term, null shift, diffuse reconnection (emitter / Lambertian / medium parents, visibility,
pdfs, transmittance), MIS weights with sensorMIS, border rule -- all 27 accumulators and the shift counters."""
import numpy as np
import pytest

import cases
import indep_statements as I
import oracle_lib as O
from gvpm_amd import abi


def compare(c, tol=1e-9):
    ref, cnt, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 64, use_accel=False)
    acc, icnt = I.bre3d_full(c)
    assert icnt["evaluations"] == cnt["evaluations"] > 100
    for k in ("null_shifts", "diffuse_shifts", "failed_shifts"):
        assert icnt[k] == cnt[k], (k, icnt, cnt)
    lum = ref[..., 0:3].mean()
    names = ["flux"] + [f"shifted[{i}]" for i in range(4)] + [f"weighted[{i}]" for i in range(4)]
    for j, name in enumerate(names):
        err = np.abs(acc[..., 3 * j:3 * j + 3] - ref[..., 3 * j:3 * j + 3]).max() / lum
        assert err < tol, (name, err)
    return cnt


@pytest.mark.parametrize("scene", ["cbox", "cbox_hg", "cbox_in", "cbox_mirror", "cbox_rot", "cbox_hg_rot", "cbox_mirror_rot"])
def test_bre3d_all_27_accumulators(scene):
    c = cases.make_case(scene, 20, 16, 4000, 4.0)
    cnt = compare(c)
    assert cnt["null_shifts"] > 100 and cnt["diffuse_shifts"] > 100


@pytest.mark.parametrize("kw", [dict(use_mis=0), dict(power_heuristic=1), dict(use_shift_null=0), dict(path_set=0),
                                dict(visibility_as_written=0), dict(max_depth=3), dict(min_depth=3),
                                dict(debug_shift=abi.GVPM_SHIFT_NULL), dict(debug_shift=abi.GVPM_SHIFT_DIFFUSE),
                                dict(lighting_interaction_mode=abi.GVPM_MEDIA2MEDIA)])
def test_bre3d_flags(kw):
    c = cases.make_case("cbox", 16, 12, 3000, 4.0, **kw)
    compare(c)


def test_fine_image_mostly_null_shifts_and_coarse_image_mostly_reconnections():
    fine = cases.make_case("cbox", 160, 160, 3000, 5.0)
    fine.rays = fine.rays[::97]
    cnt = compare(fine)
    assert cnt["null_shifts"] > 2 * cnt["diffuse_shifts"]
    coarse = cases.make_case("cbox", 8, 6, 20000, 2.0)
    cnt = compare(coarse)
    assert cnt["diffuse_shifts"] > cnt["null_shifts"]


# ---------------------------------------------------------------------------------------------------------------------
# G-Beams: base term, null shift (3D), reconnection with visibility over the whole new beam, kernelPDF, MIS
from test_oracle_beams import make_beam_case, TECHS  # noqa: E402


def compare_beams(c, tol=1e-9):
    ref, cnt, _ = O.gather_beams(c.p, c.m, c.tris, c.beams, c.end_n, c.rays, c.r, 1, c.nb, 64)
    acc, icnt = I.beams_full(c)
    for k in ("evaluations", "null_shifts", "diffuse_shifts", "failed_shifts"):
        assert icnt[k] == cnt[k], (k, icnt, cnt)
    assert cnt["evaluations"] > 100
    lum = ref[..., 0:3].mean()
    names = ["flux"] + [f"shifted[{i}]" for i in range(4)] + [f"weighted[{i}]" for i in range(4)]
    for j, name in enumerate(names):
        err = np.abs(acc[..., 3 * j:3 * j + 3] - ref[..., 3 * j:3 * j + 3]).max() / lum
        assert err < tol, (name, err)
    return cnt


@pytest.mark.parametrize("tech", TECHS)
@pytest.mark.parametrize("scene", ["cbox", "cbox_hg", "laser", "cbox_rot", "laser_rot"])
def test_beams_all_27_accumulators(tech, scene):
    c = make_beam_case(scene, 12, 10, 1500, 4.0, technique=tech)
    cnt = compare_beams(c)
    assert cnt["diffuse_shifts"] > 50


@pytest.mark.parametrize("kw", [dict(use_mis=0), dict(power_heuristic=1), dict(use_shift_null=0), dict(path_set=0), dict(max_depth=3),
                                dict(debug_shift=abi.GVPM_SHIFT_NULL), dict(debug_shift=abi.GVPM_SHIFT_DIFFUSE),
                                dict(lighting_interaction_mode=abi.GVPM_MEDIA2MEDIA)])
def test_beams_flags(kw):
    c = make_beam_case("cbox", 10, 8, 1200, 4.0, **kw)
    compare_beams(c)


def test_beams_fine_image_takes_the_null_shift():
    c = make_beam_case("cbox", 96, 96, 1200, 5.0)
    c.rays = c.rays[::61]
    cnt = compare_beams(c)
    assert cnt["null_shifts"] > 100


# ---------------------------------------------------------------------------------------------------------------------
# G-Planes 0D: base term, specularShift (re-intersection, throughput ratio, Jacobian, MIS)
from test_oracle_planes import make_plane_case  # noqa: E402


@pytest.mark.parametrize("scene,g", [("cbox_in", 0.0), ("cbox_in", 0.7), ("laser_in", 0.0), ("laser_in_hg", 0.7), ("cbox_in_rot", 0.0),
                                     ("laser_in_hg_rot", 0.7)])
@pytest.mark.parametrize("kw", [dict(), dict(use_mis=0)])
def test_planes_all_27_accumulators(scene, g, kw):
    c = make_plane_case(scene, 16, 12, 1500, **kw)
    c.m.g = g
    ref, cnt, _ = O.gather_planes(c.p, c.m, c.tris, c.beams, c.w1, c.len1, c.rays, 1, c.nb, 64)
    acc, icnt = I.planes_full(c)
    assert cnt["evaluations"] > 1000
    for k in ("evaluations", "diffuse_shifts", "failed_shifts"):
        assert icnt[k] == cnt[k], (k, icnt, cnt)
    lum = ref[..., 0:3].mean()
    for j in range(9):
        assert np.abs(acc[..., 3 * j:3 * j + 3] - ref[..., 3 * j:3 * j + 3]).max() / lum < 1e-9, j


# ---------------------------------------------------------------------------------------------------------------------
# G-VPM: distance sampling, radius query, null shift / reconnection at the sampled distance, pdfs of the distance estimator
from test_oracle_vpm import make_vpm_case  # noqa: E402


@pytest.mark.parametrize("scene,kw", [(s, k) for s in ("cbox", "cbox_hg", "cbox_mirror") for k in (dict(), dict(use_mis=0), dict(use_shift_null=0), dict(max_depth=3))]
                         + [("cbox_rot", dict()), ("cbox_rot", dict(visibility_as_written=0))])
def test_vpm_all_27_accumulators(scene, kw):
    c = make_vpm_case(scene, 12, 10, 6000, 8.0, 6, **kw)
    ref, rsv, rnv, cnt, _ = O.gather_vpm(c.p, c.m, c.tris, c.ph, c.rays, c.samples, 64, use_accel=True)
    acc, icnt, mvol = I.vpm_full(c)
    assert cnt["evaluations"] > 300
    for k in ("evaluations", "null_shifts", "diffuse_shifts", "failed_shifts"):
        assert icnt[k] == cnt[k], (k, icnt, cnt)
    lum = ref[..., 0:3].mean()
    for j in range(9):
        assert np.abs(acc[..., 3 * j:3 * j + 3] - ref[..., 3 * j:3 * j + 3]).max() / lum < 1e-9, j
    # the SPPM update of gvpm.cpp:1191-1195 from the photon counts
    N = mvol * c.p.alpha
    with np.errstate(invalid="ignore", divide="ignore"):
        sv = np.where(mvol > 0, c.p.initial_scale_volume * np.cbrt((c.p.alpha * mvol) / mvol), c.p.initial_scale_volume)
    assert np.allclose(rnv, N, rtol=1e-12) and np.allclose(rsv, sv, rtol=1e-12)
