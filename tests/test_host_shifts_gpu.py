"""Manifold shifts through the host (include/gvpm_hip.h gvpm_enable_host_shifts ... gvpm_upload_host_shifts; SURVEY 8 row
f4, first slice): the G-BRE gather records a request for every shift that reaches shiftPhotonManifold, the host answers,
the device finishes the shift (shift_volume_photon.cpp:205-279).  The host's walk is Mitsuba's; here it is replaced, on
both sides, by the oracle's stand-in (a smooth function of the request and the photon's parent) -- what is verified is the
content of the requests and the arithmetic the device applies to the answers."""
import numpy as np
import pytest

import cases
import oracle_lib as O
from gvpm_amd import abi, hip

pytestmark = pytest.mark.gpu
COUNTERS = ("evaluations", "null_shifts", "diffuse_shifts", "failed_shifts")


def mirror_case(**over):
    c = cases.make_case("cbox_mirror", 48, 40, 40000, 4.0, use_manifold=1, **over)
    st = (c.ph.flags >> 2) & 7
    assert (st == 3).sum() > 500          # light paths through the mirror
    return c


def device(c, answer, cap=1 << 20):
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    ctx.enable_host_shifts(cap)
    ctx.upload_photons(c.ph)
    ctx.upload_camera_beams(c.rays)
    ctx.gather(c.it, c.nb)
    req, n = ctx.download_shift_requests(cap)
    if answer:
        ctx.upload_host_shifts(O.standin_host_shifts(c.ph, req))
    acc = ctx.download_accum().astype(np.float64)
    st = ctx.stats()
    ctx.close()
    return acc, st, req, n


@pytest.mark.parametrize("over", [dict(), dict(use_mis=0), dict(power_heuristic=1), dict(vol_technique=abi.GVPM_VOL_BRE2D, use_shift_null=0)])
def test_answered_requests_give_the_oracles_manifold_shifts(over):
    c = mirror_case(**over)
    acc, st, req, n = device(c, True)
    assert n == req.size and n > 1000
    # (BRE-2D: the device's hit set is the own-box one, tests/test_parity_gpu.py::test_bre2d_matches_own_box_oracle)
    accel = c.p.vol_technique != abi.GVPM_VOL_BRE2D
    ref, cnt, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, c.it, c.nb, 64, use_accel=accel)
    assert st["evaluations"] == cnt["evaluations"] and st["null_shifts"] == cnt["null_shifts"]
    # (a stand-in walk "fails" where its reach test flips: smooth inputs, the device's are fp32)
    assert abs(st["diffuse_shifts"] - cnt["diffuse_shifts"]) <= 2 and abs(st["failed_shifts"] - cnt["failed_shifts"]) <= 2
    lum = max(ref[..., 0:3].mean(), 1e-30)
    assert np.sqrt(((acc - ref) ** 2).mean()) / lum < 1e-4
    # every request names a manifold-typed photon, a beam set of the upload and one of its four shifted rays; the points
    # are the base and shifted rays at the same t', the offset position is within the kernel of the shifted point
    assert (((c.ph.flags[req["photon"]] >> 2) & 7) == 3).all()
    assert (req["set"] < c.rays.shape[0]).all() and (req["shift"] < 4).all()
    base = c.rays[req["set"], 0]
    sh = c.rays[req["set"], 1 + req["shift"].astype(np.int64)]
    bp = base["o"].astype(np.float64) + base["d"] * req["t"][:, None].astype(np.float64)
    sp = sh["o"].astype(np.float64) + sh["d"] * req["t"][:, None].astype(np.float64)
    assert np.abs(bp - req["base_point"]).max() < 1e-5 and np.abs(sp - req["shift_point"]).max() < 1e-5
    assert (np.linalg.norm(req["offset_pos"] - req["shift_point"], axis=1) <= 3.01 * req["radius"]).all()
    assert np.array_equal(req["radius"], np.full(n, np.float32(c.r)))
    # the answered terms matter: without them the film differs by far more than the parity bar
    c0 = mirror_case(**over)
    c0.p.use_manifold = 0
    ref0, _, _ = O.gather_bre(c0.p, c.m, c.tris, c.ph, c.rays, c.r, c.it, c.nb, 64, use_accel=accel)
    assert np.sqrt(((ref0 - ref) ** 2).mean()) / lum > 1e-3


def test_unanswered_requests_and_requests_beyond_the_capacity_are_failed_shifts():
    c = mirror_case()
    c0 = mirror_case()
    c0.p.use_manifold = 0
    ref0, cnt0, _ = O.gather_bre(c0.p, c.m, c.tris, c.ph, c.rays, c.r, c.it, c.nb, 64)
    lum = max(ref0[..., 0:3].mean(), 1e-30)
    for cap in (1 << 20, 100):
        acc, st, req, n = device(c, False, cap)
        assert n == min(cap, n) or cap == 100
        for k in COUNTERS:
            assert st[k] == cnt0[k], (cap, k, st, cnt0)
        assert np.sqrt(((acc - ref0) ** 2).mean()) / lum < 1e-4


def test_off_by_default_and_wrong_result_count():
    c = mirror_case()
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    ctx.upload_photons(c.ph)
    ctx.upload_camera_beams(c.rays)
    ctx.gather(c.it, c.nb)
    req, n = ctx.download_shift_requests(16)
    assert n == 0 and req.size == 0
    ctx.enable_host_shifts(1 << 20)
    ctx.gather(c.it + 1, c.nb)
    req, n = ctx.download_shift_requests(1 << 20)
    assert n > 1000
    with pytest.raises(hip.GvpmError):
        ctx.upload_host_shifts(np.zeros(n - 1, abi.HOST_SHIFT_DTYPE))
    ctx.upload_host_shifts(O.standin_host_shifts(c.ph, req))
    with pytest.raises(hip.GvpmError):
        ctx.upload_host_shifts(np.zeros(3, abi.HOST_SHIFT_DTYPE))
    ctx.close()
