"""Pins the CPU oracle (oracle/gvpm_oracle.hpp) on the analytic invariants of SURVEY 8c.

The reference has no golden vectors for this path ("parity unpinned"), so the restatement is
checked against (a) an independent numpy statement of the base estimator, (b) the closed-form
chord integral, (c) invariants that follow from the cited code (w = 1/2 for identical beams,
0 <= w <= 1, border rule, accel independence) and (d) committed fp64 fixtures (regressions)."""
import os

import numpy as np
import pytest

import cases
import oracle_lib as O
from gvpm_amd import abi

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def split(acc):
    """accum[H,W,27] -> flux[H,W,3], shifted[H,W,4,3], weighted[H,W,4,3]"""
    H, W = acc.shape[:2]
    return acc[..., 0:3], acc[..., 3:15].reshape(H, W, 4, 3), acc[..., 15:27].reshape(H, W, 4, 3)


@pytest.fixture(scope="module")
def case():
    return cases.make_case("cbox", 24, 20, 6000, 4.0)


def test_base_flux_matches_numpy_bruteforce(case):
    c = case
    acc, cnt, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 64, use_accel=False)
    ref, evals = cases.numpy_base_flux(c)
    assert cnt["evaluations"] == evals and evals > 2000
    flux = split(acc)[0] * c.nb
    assert np.allclose(flux, ref, rtol=1e-10, atol=1e-12 * ref.max())


def test_base_flux_independent_of_shift_flags(case):
    c = case
    base = split(O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 64)[0])[0]
    for kw in (dict(use_shift_null=0), dict(use_mis=0), dict(power_heuristic=1), dict(visibility_as_written=0),
               dict(debug_shift=abi.GVPM_SHIFT_NULL)):
        p = c.p.copy()
        for k, v in kw.items():
            setattr(p, k, v)
        f = split(O.gather_bre(p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 64)[0])[0]
        assert np.array_equal(f, base), kw


def test_accel_independence_3d(case):
    """kd-tree -> BVH stack walk and the own-box brute force give the same BRE-3D result."""
    c = case
    a, ca, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 64, use_accel=True)
    b, cb, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 64, use_accel=False)
    for k in ("evaluations", "null_shifts", "diffuse_shifts", "failed_shifts"):
        assert ca[k] == cb[k]
    assert np.allclose(a, b, rtol=1e-12, atol=1e-14 * a.max())  # only the summation order differs


def test_bvh_visits_every_photon_once():
    c = cases.make_case("cbox", 4, 4, 3000, 1.0)
    # a radius larger than the scene: every own box contains every ray -> all photons are candidates
    _, cnt, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, 10.0, 1, c.nb, 64, use_accel=True)
    assert cnt["candidates"] == c.ph.n * c.rays.shape[0]


def test_bre2d_bvh_only_adds_photons_beyond_the_beam_end():
    """BRE-2D has no far bound (shift_volume_photon.cpp:726-731): the reference BVH accepts extra
    inner-node photons beyond the beam end, the own-box definition does not (see oracle header)."""
    c = cases.make_case("cbox", 24, 20, 6000, 4.0, vol_technique=abi.GVPM_VOL_BRE2D, use_shift_null=0)
    a, ca, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 64, use_accel=True)
    b, cb, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 64, use_accel=False)
    assert ca["evaluations"] >= cb["evaluations"]
    assert ca["evaluations"] - cb["evaluations"] <= 0.01 * cb["evaluations"]
    assert (a[..., 0:3] >= b[..., 0:3] - 1e-12).all()


def test_identical_shifted_beams_give_half_weight_and_zero_gradient(case):
    """shifted camera beam == base beam => w = 1/2, shifted == weighted, gradient 0."""
    c = case
    rays = cases.rays_shift_equals_base(c.rays)
    for null in (1, 0):
        p = c.p.copy()
        p.use_shift_null = null
        acc, cnt, _ = O.gather_bre(p, c.m, c.tris, c.ph, rays, c.r, 1, c.nb, 64)
        flux, sh, wt = split(acc)
        px, py = cases.pixels_of(rays)
        assert cnt["failed_shifts"] == 0
        for i in range(4):
            border = ((i == abi.GVPM_RIGHT) & (px == p.width - 1)) | ((i == abi.GVPM_TOP) & (py == p.height - 1))
            inner = ~border
            f = flux[py, px]
            # null shift: exact; reconnection: the fp32-rounded path records reproduce the photon
            # flux and the pdf ratio to ~1e-6
            tol = 1e-9 if null else 2e-5
            assert np.allclose(wt[py[inner], px[inner], i], 0.5 * f[inner], rtol=tol, atol=1e-12)
            assert np.allclose(sh[py[inner], px[inner], i], 0.5 * f[inner], rtol=2e-5, atol=1e-12)
            # border rule (vi): w = 1
            assert np.allclose(wt[py[border], px[border], i], f[border], rtol=1e-9, atol=1e-12)
        _, dx, dy = O.assemble(acc, 1, False)
        inner_img = np.zeros(flux.shape[:2], bool)
        inner_img[py, px] = True
        lum = flux.mean()
        # interior gradient: (S_R - W_R)(x) + (W_L - S_L)(x+1) = 0
        assert np.abs(dx[:, :-1]).max() < 1e-4 * lum and np.abs(dy[:-1]).max() < 1e-4 * lum


def test_no_mis_gives_half_weight_for_successful_shifts(case):
    """useMIS = none => successful shift w = 1/2, failed w = 1."""
    c = case
    p = c.p.copy()
    p.use_mis = 0
    rays = cases.rays_shift_equals_base(c.rays)
    acc, cnt, _ = O.gather_bre(p, c.m, c.tris, c.ph, rays, c.r, 1, c.nb, 64)
    flux, sh, wt = split(acc)
    px, py = cases.pixels_of(rays)
    inner = (px < p.width - 1) & (py < p.height - 1)
    for i in range(4):
        assert np.allclose(wt[py[inner], px[inner], i], 0.5 * flux[py[inner], px[inner]], rtol=1e-9, atol=1e-12)
        assert np.allclose(sh[py[inner], px[inner], i], 0.5 * flux[py[inner], px[inner]], rtol=2e-5, atol=1e-12)


def test_weights_between_zero_and_one(case):
    """0 <= w <= 1 => 0 <= weighted[i] <= mediumFlux per pixel."""
    c = case
    for scene in ("cbox", "cbox_hg"):
        cc = c if scene == "cbox" else cases.make_case(scene, 24, 20, 6000, 4.0)
        acc, _, _ = O.gather_bre(cc.p, cc.m, cc.tris, cc.ph, cc.rays, cc.r, 1, cc.nb, 64)
        flux, sh, wt = split(acc)
        assert (wt >= 0).all() and (sh >= 0).all()
        assert (wt <= flux[:, :, None, :] * (1 + 1e-9) + 1e-15).all()


def test_border_pixels_have_unit_weight(case):
    c = case
    acc, _, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 64)
    flux, sh, wt = split(acc)
    W, H = c.p.width, c.p.height
    assert np.allclose(wt[:, W - 1, abi.GVPM_RIGHT], flux[:, W - 1], rtol=1e-9, atol=1e-14)
    assert np.allclose(wt[H - 1, :, abi.GVPM_TOP], flux[H - 1, :], rtol=1e-9, atol=1e-14)


def test_bre3d_random_resample_averages_to_the_chord_integral():
    """mean over randValue of contrib/pdfCameraPos = integral over the kernel chord."""
    c = cases.make_case("cbox", 8, 8, 20000, 6.0, path_set=0, max_depth=0)
    # one photon well inside the kernel of one beam
    b = c.rays[c.rays.shape[0] // 2].copy()
    o, d = b[0]["o"].astype(np.float64), b[0]["d"].astype(np.float64)
    w = c.ph.pos.astype(np.float64) - o
    disk = w @ d
    d2 = ((o + np.outer(disk, d) - c.ph.pos) ** 2).sum(1)
    cand = np.nonzero((disk > 0.2) & (disk < b[0]["len"] - 0.2) & (d2 < 0.5 * c.r ** 2))[0]
    assert cand.size > 0
    one = c.ph.subset(cand[:1])
    n = 4001
    sets = np.repeat(b[None], n, axis=0)
    sets[:, 0]["rand"] = (np.arange(n) + 0.5) / n
    # spread the sets over distinct pixels is unnecessary: the oracle sums sets of one pixel
    acc, cnt, _ = O.gather_bre(c.p, c.m, c.tris, one, sets, c.r, 1, 1, 64, use_accel=False)
    px, py = int(b[0]["pixel"]) & 0xFFFF, int(b[0]["pixel"]) >> 16
    mean_flux = acc[py, px, 0:3] / n
    # analytic: sigma_s * flux * phase / kernelVol * int_{chord} exp(-sigma_t (t - eps)) dt
    st = float(c.m.sigma_t[0])
    dT = np.sqrt(c.r ** 2 - d2[cand[0]])
    t0, t1 = disk[cand[0]] - dT, disk[cand[0]] + dT
    eps = c.p.epsilon
    integral = (np.exp(-st * (t0 - eps)) - np.exp(-st * (t1 - eps))) / st
    kv = 4.0 / 3.0 * np.pi * c.r ** 3
    expect = np.array(list(c.m.sigma_s)) * one.flux[0] / (4 * np.pi) / kv * integral
    assert cnt["evaluations"] == n
    assert np.allclose(mean_flux, expect, rtol=2e-6)


def test_path_set_keeps_the_expectation():
    """the checkerboard path-set filter (x2 weight) keeps the image mean."""
    c = cases.make_case("cbox", 24, 20, 20000, 5.0)
    on = split(O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 64)[0])[0].mean()
    p = c.p.copy()
    p.path_set = 0
    off = split(O.gather_bre(p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 64)[0])[0].mean()
    assert abs(on - off) / off < 0.05


def test_float_and_double_builds_agree(case):
    c = case
    a64, c64, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 64)
    a32, c32, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 32)
    assert abs(c64["evaluations"] - c32["evaluations"]) <= 3
    lum = a64[..., 0:3].mean()
    assert np.sqrt(((a64 - a32) ** 2).mean()) / lum < 1e-3


def test_apa_running_mean_and_radius_schedule():
    c = cases.make_case("cbox", 12, 10, 3000, 4.0)
    acc = None
    scale = c.p.initial_scale_volume
    per_it = []
    for it in (1, 2, 3):
        ph, nb = c.sc.shoot_photons(it, 3000)
        rays = c.sc.camera_beams(it)
        r = cases.radius_of(c.p, scale)
        one, _, _ = O.gather_bre(c.p, c.m, c.tris, ph, rays, r, 1, nb, 64)
        per_it.append(one)
        acc, _, _ = O.gather_bre(c.p, c.m, c.tris, ph, rays, r, it, nb, 64, accum=acc)
        scale = O.scale_volume_apa(scale, it, c.p.alpha, c.p.vol_technique)
    assert np.allclose(acc, np.mean(per_it, axis=0), rtol=1e-12, atol=1e-18)
    # gvpm.cpp:181-215: ratio = (it-1+alpha)/it, cube root for 3D kernels
    al = float(c.p.alpha)  # fp32 0.7
    expect = c.p.initial_scale_volume * np.cbrt(al / 1) * np.cbrt((1 + al) / 2) * np.cbrt((2 + al) / 3)
    assert abs(scale - expect) < 1e-12
    assert abs(O.scale_volume_apa(1.0, 1, 0.7, abi.GVPM_VOL_BRE2D) - np.sqrt(0.7)) < 1e-15
    assert abs(O.scale_volume_apa(1.0, 4, 0.7, abi.GVPM_BEAM_BEAM_1D) - 3.7 / 4) < 1e-15


def test_assemble_matches_numpy(case):
    c = case
    acc, _, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 64)
    flux, sh, wt = split(acc)
    thr, dx, dy = O.assemble(acc, 2, True, emission=np.ones(flux.shape))
    L, R, T, B = 0, 1, 2, 3
    gx = sh[:, :, R] - wt[:, :, R]
    gx[:, :-1] += wt[:, 1:, L] - sh[:, 1:, L]
    gy = sh[:, :, T] - wt[:, :, T]
    gy[:-1] += wt[1:, :, B] - sh[1:, :, B]
    assert np.allclose(dx, gx) and np.allclose(dy, gy)
    Tsum = wt.sum(2)
    Tsum[:, :-1] += sh[:, 1:, L]
    Tsum[:, 1:] += sh[:, :-1, R]
    Tsum[:-1] += sh[1:, :, B]
    Tsum[1:] += sh[:-1, :, T]
    assert np.allclose(thr, Tsum / 4.0)
    thr2, _, _ = O.assemble(acc, 2, False, emission=np.ones(flux.shape))
    assert np.allclose(thr2, flux + 0.5)


def test_empty_inputs():
    c = cases.make_case("cbox", 8, 8, 500, 4.0)
    none = c.ph.subset(np.zeros(0, np.int64))
    acc, cnt, _ = O.gather_bre(c.p, c.m, c.tris, none, c.rays, c.r, 1, c.nb, 64)
    assert cnt["evaluations"] == 0 and not acc.any()
    acc, cnt, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays[:0], c.r, 1, c.nb, 64)
    assert cnt["evaluations"] == 0 and not acc.any()


@pytest.mark.parametrize("name", ["cbox_bre3d", "cbox_hg_bre3d", "cbox_bre2d"])
def test_golden_fixture(name):
    """Committed fp64 fixtures (tests/golden/make_golden.py): regression pin for the oracle."""
    import golden_io
    g = golden_io.load(os.path.join(GOLD, name + ".npz"))
    acc, cnt, _ = O.gather_bre(g.p, g.m, g.tris, g.ph, g.rays, g.r, g.it, g.nb, 64, use_accel=False)
    assert cnt["evaluations"] == g.evaluations
    assert np.allclose(acc, g.accum, rtol=1e-9, atol=1e-12 * g.accum.max())
