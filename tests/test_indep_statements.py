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


@pytest.mark.parametrize("scene", ["cbox", "cbox_hg", "cbox_in", "cbox_mirror"])
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
