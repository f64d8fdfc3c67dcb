"""Committed fixtures of the G-VPM / G-Beams / G-Planes paths (tests/golden/make_golden.py: seeded inputs + fp64
oracle outputs): the oracle must reproduce them (CPU), the HIP path must match them (GPU)."""
import os

import numpy as np
import pytest

import golden_io
import oracle_lib as O
from gvpm_amd import abi, hip

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return golden_io.load(os.path.join(GOLD, name + ".npz"))


def l2(a, ref):
    return float(np.sqrt(((np.asarray(a, np.float64) - ref) ** 2).mean()) / max(ref[..., 0:3].mean(), 1e-30))


def test_oracle_reproduces_vpm_fixture():
    g = load("cbox_vpm")
    samples = g.extra["samples"].view(abi.VPM_SAMPLE_DTYPE).reshape(-1)
    acc, sv, nv, cnt, _ = O.gather_vpm(g.p, g.m, g.tris, g.ph, g.rays, samples, 64, use_accel=False)
    assert cnt["evaluations"] == g.evaluations
    assert np.allclose(acc, g.accum, rtol=1e-12, atol=0) and np.allclose(sv, g.extra["scale_vol"], rtol=1e-12)


@pytest.mark.parametrize("name", ["cbox_beams3d", "cbox_beams1d"])
def test_oracle_reproduces_beam_fixtures(name):
    g = load(name)
    acc, cnt, _ = O.gather_beams(g.p, g.m, g.tris, g.ph, g.extra["end_n"], g.rays, g.r, g.it, g.nb, 64)
    assert cnt["evaluations"] == g.evaluations
    assert np.allclose(acc, g.accum, rtol=1e-12, atol=0)


def test_oracle_reproduces_plane_fixture():
    g = load("cbox_in_planes0d")
    acc, cnt, _ = O.gather_planes(g.p, g.m, g.tris, g.ph, g.extra["w1"], g.extra["len1"], g.rays, g.it, g.nb, 64)
    assert cnt["evaluations"] == g.evaluations
    assert np.allclose(acc, g.accum, rtol=1e-12, atol=0)


@pytest.mark.gpu
def test_device_matches_vpm_fixture():
    g = load("cbox_vpm")
    samples = g.extra["samples"].view(abi.VPM_SAMPLE_DTYPE).reshape(-1)
    ctx = hip.Context(g.p, device=0)
    ctx.upload_scene(*g.tris)
    ctx.upload_medium(g.m)
    ctx.upload_photons(g.ph)
    ctx.upload_camera_beams(g.rays)
    ctx.upload_vpm_samples(samples)
    ctx.gather(1, g.nb)
    acc, st = ctx.download_accum(), ctx.stats()
    sv, nv = ctx.download_vpm_state()
    ctx.close()
    assert st["evaluations"] == g.evaluations
    assert l2(acc, g.accum) < 1e-4 and np.allclose(sv, g.extra["scale_vol"], rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cbox_beams3d", "cbox_beams1d"])
def test_device_matches_beam_fixtures(name):
    g = load(name)
    ctx = hip.Context(g.p, device=0)
    ctx.upload_scene(*g.tris)
    ctx.upload_medium(g.m)
    ctx.upload_beams(g.ph, g.extra["end_n"])
    ctx.upload_camera_beams(g.rays)
    ctx.gather(g.it, g.nb)
    acc, st = ctx.download_accum(), ctx.stats()
    ctx.close()
    assert abs(st["evaluations"] - g.evaluations) <= 2
    assert l2(acc, g.accum) < 1e-3


@pytest.mark.gpu
def test_device_matches_plane_fixture():
    g = load("cbox_in_planes0d")
    ctx = hip.Context(g.p, device=0)
    ctx.upload_scene(*g.tris)
    ctx.upload_medium(g.m)
    ctx.upload_planes(g.ph, g.extra["w1"], g.extra["len1"])
    ctx.upload_camera_beams(g.rays)
    ctx.gather(g.it, g.nb)
    acc, st = ctx.download_accum(), ctx.stats()
    ctx.close()
    assert st["evaluations"] == g.evaluations
    assert l2(acc, g.accum) < 1e-5
