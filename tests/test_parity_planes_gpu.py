"""G-Planes parity: HIP path vs the fp64 oracle (computeVolumeGradientPlanes)."""
import numpy as np
import pytest

import cases
import oracle_lib as O
from gvpm_amd import abi, hip
from test_oracle_planes import make_plane_case
from test_parity_gpu import l2

pytestmark = pytest.mark.gpu


def device_planes(c, p=None, rays=None, iters=1, exact=True):
    p = c.p if p is None else p
    ctx = hip.Context(p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    ref = None
    total = dict(evaluations=0, diffuse_shifts=0, failed_shifts=0)
    for it in range(1, iters + 1):
        if it == 1:
            beams, w1, l1, nb, r = c.beams, c.w1, c.len1, c.nb, (c.rays if rays is None else rays)
        else:
            beams, _, w1, l1, nb = c.sc.shoot_planes(it, c.beams.n)
            r = c.sc.camera_beams(it)
        ctx.upload_planes(beams, w1, l1)
        ctx.upload_camera_beams(r)
        ctx.gather(it, nb)
        ref, cnt, _ = O.gather_planes(p, c.m, c.tris, beams, w1, l1, r, it, nb, 64, accum=ref)
        for k in total:
            total[k] += cnt[k]
    acc = ctx.download_accum()
    st = ctx.stats()
    film = ctx.download_film(iters, True)
    ctx.close()
    lum = max(ref[..., 0:3].mean(), 1e-30)
    # the pierce decision is the oracle's own fp64 test: the evaluated pairs are identical
    assert st["evaluations"] == total["evaluations"], (st, total)
    tol = 0 if exact else 2
    assert abs(st["diffuse_shifts"] - total["diffuse_shifts"]) <= tol and abs(st["failed_shifts"] - total["failed_shifts"]) <= tol, (st, total)
    assert st["null_shifts"] == 0
    assert l2(acc, ref, lum) < 1e-5
    rfilm = O.assemble(ref, iters, True)
    for a, b in zip(film, rfilm):
        assert l2(a, b, lum) < 1e-5
    return acc, ref, st


@pytest.mark.parametrize("scene,W,H,n", [("cbox_in", 32, 28, 6000), ("cbox_in", 70, 50, 3000)])
def test_planes_match_fp64_oracle(scene, W, H, n):
    c = make_plane_case(scene, W, H, n)
    acc, ref, st = device_planes(c)
    assert st["evaluations"] > 20000


def test_planes_two_iterations_and_radius_ratio():
    c = make_plane_case("cbox_in", 24, 20, 3000)
    ctx = hip.Context(c.p, device=0)
    r0 = ctx.radius()
    ctx.close()
    device_planes(c, iters=2)
    # linear APA ratio for the 0D kernel
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    ctx.upload_planes(c.beams, c.w1, c.len1)
    ctx.upload_camera_beams(c.rays)
    ctx.gather(1, c.nb)
    assert abs(ctx.radius() / r0 - c.p.alpha) < 1e-6
    ctx.close()


@pytest.mark.parametrize("kw", [dict(use_mis=0), dict(power_heuristic=1), dict(path_set=0), dict(max_depth=3)])
def test_planes_flag_sweep(kw):
    c = make_plane_case("cbox_in", 24, 20, 3000)
    p = c.p.copy()
    for k, v in kw.items():
        setattr(p, k, v)
    device_planes(c, p=p)


def test_planes_identical_shift_empty_and_preconditions():
    c = make_plane_case("cbox_in", 16, 12, 2000)
    acc, ref, st = device_planes(c, rays=cases.rays_shift_equals_base(c.rays))
    assert st["failed_shifts"] == 0
    f = acc[..., 0:3]
    m = f[..., 0] > 0
    assert np.allclose(acc[..., 15:18][m], 0.5 * f[m], rtol=1e-4)
    c2 = make_plane_case("cbox_in", 16, 12, 300)
    c2.beams = c2.beams.subset(np.zeros(0, np.int64))
    c2.w1, c2.len1 = c2.w1[:0], c2.len1[:0]
    acc, ref, st = device_planes(c2)
    assert st["evaluations"] == 0 and not acc.any()
    # gvpm.cpp:163-166 and gvpm_struct.h:310-313
    p = c.p.copy()
    p.min_depth = 1
    with pytest.raises(hip.GvpmError):
        hip.Context(p, device=0)
    p = c.p.copy()
    p.use_shift_null = 1
    with pytest.raises(hip.GvpmError):
        hip.Context(p, device=0)
    # gather without planes uploaded
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    ctx.upload_photons(c.beams)
    ctx.upload_camera_beams(c.rays)
    with pytest.raises(hip.GvpmError):
        ctx.gather(1, c.nb)
    ctx.close()


def test_planes_without_a_common_ray_origin():
    """The tile-frustum cull needs the tile's rays to share their origin; when they do not (camera edges beyond the
    first) the kernel tests every plane against every ray.  Jittered origins: same result as the oracle."""
    c = make_plane_case("cbox_in", 32, 28, 4000)
    rng = np.random.default_rng(5)
    rays = c.rays.copy()
    rays["o"] += rng.uniform(-2e-3, 2e-3, size=rays["o"].shape[:1] + (1, 3)).astype(np.float32)
    acc, ref, st = device_planes(c, rays=rays)
    assert st["evaluations"] > 10000
