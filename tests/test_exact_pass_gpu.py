"""The exact passes (csrc/exact_shift.hip for G-BRE / G-VPM, exact_beams_kernel in csrc/gather_beams.hip) against the oracle on
WHOLE frames.  In a normal gather a pass sees one shift in 10^4 .. 10^5 -- the ones whose decisions fp32 cannot take -- so the
parity suite says little about it.  GVPM_EXACT_ALL=1 widens the
fast kernels' ambiguity band to everything: they add the base terms only, and every one of the four shifts of every pair is
evaluated by the pass -- the reference's statement in uncontracted fp64.  Bars: counters == the fp64 oracle's exactly, the 27
accumulators to fp32 accumulation noise (1e-6 of the mean luminance: tighter than the fast path's, whose radiometry is fp32)."""
import numpy as np
import pytest

import cases
import oracle_lib as O
from gvpm_amd import abi, hip
from test_oracle_beams import make_beam_case, TECHS
from test_oracle_vpm import make_vpm_case
from test_parity_gpu import l2

pytestmark = pytest.mark.gpu
COUNTERS = ("evaluations", "null_shifts", "diffuse_shifts", "failed_shifts")


def run_bre(c, monkeypatch):
    monkeypatch.setenv("GVPM_EXACT_ALL", "1")
    ctx = hip.Context(c.p, device=0)
    monkeypatch.delenv("GVPM_EXACT_ALL")
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    cases.upload_bsdfs(ctx, c)
    ctx.upload_photons(c.ph)
    ctx.upload_camera_beams(c.rays)
    ctx.gather(c.it, c.nb)
    acc = ctx.download_accum().astype(np.float64)
    st = ctx.stats()
    taken, lost = ctx.exact_shifts()
    ctx.close()
    return acc, st, taken, lost


@pytest.mark.parametrize("scene,kw", [("cbox", dict()), ("cbox_hg", dict()), ("cbox_rot", dict(visibility_as_written=0)),
                                      ("fogroom_rot", dict()), ("cbox_phong_rot", dict()), ("cbox_conductor", dict(power_heuristic=1)),
                                      ("cbox_mirror_rot", dict()), ("cbox_ward_rot", dict()), ("cbox", dict(use_mis=0, path_set=0)), ("cbox", dict(use_shift_null=0)),
                                      ("cbox_rot", dict(vol_technique=abi.GVPM_VOL_BRE2D, use_shift_null=0))])
def test_every_bre_shift_through_the_exact_pass(scene, kw, monkeypatch):
    c = cases.make_case(scene, 40, 36, 30000, 1.6 if scene.endswith("_rot") else 2.5, **kw)
    acc, st, taken, lost = run_bre(c, monkeypatch)
    ref, cnt, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, c.it, c.nb, 64, use_accel=False)
    for k in COUNTERS:
        assert st[k] == cnt[k], (k, st, cnt)
    # every shift of every valid shifted ray went through the pass (invalid rays add their weight-1 term in the fast kernel)
    assert lost == 0 and taken >= st["null_shifts"] + st["diffuse_shifts"] + st["failed_shifts"] > 30000
    lum = ref[..., 0:3].mean()
    # glossy parents: the table's closed forms are evaluated in fp32 by the pass too (values, not decisions)
    assert l2(acc, ref, lum) < (2e-6 if "phong" not in scene and "conductor" not in scene and "ward" not in scene else 2e-5), l2(acc, ref, lum)


@pytest.mark.parametrize("scene,kw", [("cbox", dict()), ("cbox_hg", dict(use_mis=0)), ("cbox_rot", dict()), ("fogroom_rot", dict(visibility_as_written=0))])
def test_every_vpm_shift_through_the_exact_pass(scene, kw, monkeypatch):
    c = make_vpm_case(scene, 32, 28, 40000, 3.1 if scene.endswith("_rot") else 5.0, nb=10, **kw)
    monkeypatch.setenv("GVPM_EXACT_ALL", "1")
    ctx = hip.Context(c.p, device=0)
    monkeypatch.delenv("GVPM_EXACT_ALL")
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    ctx.upload_photons(c.ph)
    ctx.upload_camera_beams(c.rays)
    ctx.upload_vpm_samples(c.samples)
    ctx.gather(1, c.nb)
    acc = ctx.download_accum().astype(np.float64)
    st = ctx.stats()
    taken, lost = ctx.exact_shifts()
    ctx.close()
    ref, sv, nv, cnt, _ = O.gather_vpm(c.p, c.m, c.tris, c.ph, c.rays, c.samples, 64, use_accel=False)
    for k in COUNTERS:
        assert st[k] == cnt[k], (k, st, cnt)
    assert lost == 0 and taken > 10000
    lum = ref[..., 0:3].mean()
    # (the pass takes the pixel's radius as the fp32 state the device carries: R * 0.01 * scaleVol differs by 1e-7 from the oracle's
    # double product, i.e. 3e-7 in the kernel volume)
    assert l2(acc, ref, lum) < 2e-5, l2(acc, ref, lum)


@pytest.mark.parametrize("tech", TECHS)
@pytest.mark.parametrize("scene,kw", [("cbox", dict()), ("cbox_rot", dict()), ("cbox_hg_rot", dict(use_mis=0)),
                                      ("cbox_conductor_rot", dict(power_heuristic=1)), ("laser_rot", dict(use_shift_null=0))])
def test_every_beam_shift_through_the_exact_pass(scene, kw, tech, monkeypatch):
    """G-Beams (gather_beams.hip, exact_beams_kernel): behind the evaluation, every gather; with GVPM_EXACT_ALL every shift of
    every evaluated pair is its to decide and to add -- the fp64 transcription with the shadow segment's triangle tests in fp64."""
    c = make_beam_case(scene, 32, 28, 12000, 1.6 if scene.endswith("_rot") else 2.5, technique=tech, **kw)
    monkeypatch.setenv("GVPM_EXACT_ALL", "1")
    ctx = hip.Context(c.p, device=0)
    monkeypatch.delenv("GVPM_EXACT_ALL")
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    cases.upload_bsdfs(ctx, c)
    rad = ctx.radius()
    ctx.upload_beams(c.beams, c.end_n)
    ctx.upload_camera_beams(c.rays)
    ctx.gather(1, c.nb)
    acc = ctx.download_accum().astype(np.float64)
    st = ctx.stats()
    taken, lost = ctx.exact_shifts()
    ctx.close()
    ref, cnt, _ = O.gather_beams(c.p, c.m, c.tris, c.beams, c.end_n, c.rays, rad, 1, c.nb, 64)
    for k in COUNTERS:
        assert st[k] == cnt[k], (k, st, cnt)
    assert lost == 0 and taken == 4 * st["evaluations"] > 20000
    lum = ref[..., 0:3].mean()
    # (the transcription keeps the reference's float intermediates and rounds every term to float: measured <= 2.2e-5)
    assert l2(acc, ref, lum) < 5e-5, l2(acc, ref, lum)
