"""Screened-Poisson reconstruction: HIP path vs the oracle (= the reference's naive backend, bit for bit) and
vs the vectors the reference itself produced.  The device reduces its dot products in fp64 in a fixed order,
the reference sums sequentially in fp32, so the comparison carries the reference's own rounding noise."""
import os

import numpy as np
import pytest

import cases
import oracle_lib as O
from gvpm_amd import hip
from test_oracle_poisson import load

pytestmark = pytest.mark.gpu


def ctx_small():
    c = cases.make_case("cbox", 8, 8, 50, 3.0)
    return hip.Context(c.p, device=0)


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("name,tol", [("poisson_L2D", 2e-5), ("poisson_L1D", 2e-3), ("poisson_L1D_wide", 2e-3)])
def test_reference_vectors(name, tol):
    dx, dy, tp, di, preset, alpha, out = load(name)
    ctx = ctx_small()
    got = ctx.poisson_solve(dx, dy, tp, di, preset, alpha)
    ctx.close()
    assert rel(got, out) < tol


@pytest.mark.parametrize("preset,tol", [("L2D", 2e-5), ("L1D", 2e-3), ("L2Q", 1e-4)])
def test_random_images_and_edge_sizes(preset, tol):
    rng = np.random.default_rng(7)
    ctx = ctx_small()
    for (W, H) in ((1, 1), (1, 9), (11, 1), (33, 20), (128, 96)):
        dx, dy = (rng.standard_normal((H, W, 3)).astype(np.float32) * 0.1 for _ in range(2))
        tp = rng.random((H, W, 3)).astype(np.float32)
        di = rng.random((H, W, 3)).astype(np.float32)
        for d in (None, di):
            got = ctx.poisson_solve(dx, dy, tp, d, preset, 0.2)
            ref = O.poisson_solve(dx, dy, tp, d, preset, 0.2)
            assert rel(got, ref) < tol, (W, H, rel(got, ref))
    # without a primal image alpha is forced to 0 and x starts at 0 (Solver.cpp:323, 338-340)
    got = ctx.poisson_solve(dx, dy, None, None, preset, 0.2)
    assert np.isfinite(got).all()
    ctx.close()


def test_film_of_a_gather_reconstructs_like_the_oracle():
    c = cases.make_case("cbox", 48, 40, 30000, 3.0)
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    ctx.upload_photons(c.ph)
    ctx.upload_camera_beams(c.rays)
    ctx.gather(1, c.nb)
    thr, dx, dy = ctx.download_film(1, True)
    for preset, tol in (("L2D", 5e-5), ("L1D", 5e-3)):
        got = ctx.poisson_solve(dx, dy, thr, None, preset, 0.2)
        ref = O.poisson_solve(dx, dy, thr, None, preset, 0.2)
        assert rel(got, ref) < tol, (preset, rel(got, ref))
    ctx.close()


def test_full_size_properties_and_errors():
    # 512^2: consistent gradients are a fixed point; the solve is deterministic run to run
    rng = np.random.default_rng(9)
    img = rng.random((512, 512, 3)).astype(np.float32)
    dx = np.zeros_like(img); dx[:, :-1] = img[:, 1:] - img[:, :-1]
    dy = np.zeros_like(img); dy[:-1] = img[1:] - img[:-1]
    ctx = ctx_small()
    out = ctx.poisson_solve(dx, dy, img, None, "L1D", 0.2)
    assert np.abs(out - img).max() < 1e-4
    noisy = (img + 0.2 * rng.standard_normal(img.shape)).astype(np.float32)
    a = ctx.poisson_solve(dx, dy, noisy, None, "L2D", 0.2)
    b = ctx.poisson_solve(dx, dy, noisy, None, "L2D", 0.2)
    assert np.array_equal(a, b)
    assert np.abs(a - img).mean() < 0.5 * np.abs(noisy - img).mean()
    p = hip.poisson_preset("L2D")
    p.cg_precond = 1
    with pytest.raises(hip.GvpmError):
        ctx.poisson_solve(dx, dy, img, None, params=p)
    ctx.close()
