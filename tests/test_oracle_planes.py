"""Pins the G-Planes part of the oracle (computeVolumeGradientPlanes, 0D kernel, specularShift)."""
import numpy as np
import pytest

import cases
import oracle_lib as O
from gvpm_amd import abi
from test_oracle_beams import split


def make_plane_case(scene="cbox_in", W=16, H=12, nplanes=2000, it=1, **kw):
    kw.setdefault("use_shift_null", 0)
    kw.setdefault("min_depth", 2)
    c = cases.make_case(scene, W, H, 10, 1.0, it=it, vol_technique=abi.GVPM_VOL_PLANE0D, **kw)
    c.beams, c.end_n, c.w1, c.len1, c.nb = c.sc.shoot_planes(it, nplanes)
    return c


def gather(c, p=None, rays=None, precision=64, **kw):
    return O.gather_planes(c.p if p is None else p, c.m, c.tris, c.beams, c.w1, c.len1,
                           c.rays if rays is None else rays, 1, c.nb, precision, **kw)


def numpy_plane0d_base(c):
    """Independent numpy statement of the 0D plane estimator: ray/parallelogram intersection
    (pm/plane_struct.h:104-135) and getContrib0D (:150-192)."""
    p = c.p
    out = np.zeros((p.height, p.width, 3))
    ori = c.beams.parent_pos.astype(np.float64)
    e0 = c.beams.pos.astype(np.float64) - ori
    l0 = np.linalg.norm(e0, axis=1)
    w0 = e0 / l0[:, None]
    w1 = c.w1.astype(np.float64)
    l1 = c.len1.astype(np.float64)
    e1 = w1 * l1[:, None]
    flux = c.beams.flux.astype(np.float64)
    st = float(c.m.sigma_t[0])
    ss = np.array(list(c.m.sigma_s), np.float64)
    msw = float(c.m.medium_sampling_weight)
    g = float(c.m.g)
    eps = float(p.epsilon)
    n = 0
    for s in range(c.rays.shape[0]):
        b = c.rays[s, 0]
        if not (int(b["info"]) & 1):
            continue
        o, d, L = b["o"].astype(np.float64), b["d"].astype(np.float64), float(b["len"])
        px, py = int(b["pixel"]) & 0xFFFF, int(b["pixel"]) >> 16
        P = np.cross(d, e1)
        det = (e0 * P).sum(1)
        ok = np.abs(det) >= 1e-5
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / det
            T = o - ori
            t0 = (T * P).sum(1) * inv
            Q = np.cross(T, e0)
            t1 = (Q @ d) * inv
            tc = (e1 * Q).sum(1) * inv
        ok &= (t0 >= 0) & (t0 <= 1) & (t1 >= 0) & (t1 <= 1) & (tc > eps) & (tc < L - 2 * eps + eps)
        idx = np.nonzero(ok)[0]
        if idx.size == 0:
            continue
        n += idx.size
        T0, T1, TC = t0[idx] * l0[idx], t1[idx] * l1[idx], tc[idx]
        cosT = (w1[idx] @ d)
        if g == 0:
            ph = np.full(idx.size, 1 / (4 * np.pi))
        else:
            tmp = 1 + g * g + 2 * g * cosT
            ph = (1 - g * g) / (4 * np.pi * tmp * np.sqrt(tmp))
        pf0 = np.exp(-st * T0) * msw + (1 - msw)
        pf1 = np.exp(-st * T1) * msw + (1 - msw)
        invj = 1.0 / np.abs((w0[idx] * np.cross(w1[idx], d)).sum(1))
        k = np.exp(-st * (TC + T0 + T1)) * ph / pf0 / pf1 * invj
        out[py, px] += (flux[idx] * k[:, None]).sum(0) * ss * ss
    return out / c.nb, n


def test_base_flux_matches_numpy_statement():
    c = make_plane_case("cbox_in", 16, 12, 1500)
    acc, cnt, _ = gather(c)
    ref, n = numpy_plane0d_base(c)
    assert cnt["evaluations"] > 3000
    assert abs(cnt["evaluations"] - n) <= 2  # float det / reciprocal of the reference vs plain doubles
    f = acc[..., 0:3]
    assert np.abs(f - ref).max() < 1e-5 * ref.max()


def test_identical_shift_gives_half_weights():
    c = make_plane_case("cbox_in", 16, 12, 1500)
    acc, cnt, _ = gather(c, rays=cases.rays_shift_equals_base(c.rays))
    f, sh, wt = split(acc)
    m = f[..., 0] > 0
    assert m.sum() > 50 and cnt["failed_shifts"] == 0
    for k in range(4):
        assert np.allclose(wt[:, :, k][m], 0.5 * f[m], rtol=2e-5)
        assert np.allclose(sh[:, :, k][m], 0.5 * f[m], rtol=2e-5)


def test_weights_bounded_and_counters():
    c = make_plane_case("cbox_in", 20, 16, 2000)
    acc, cnt, _ = gather(c)
    f, sh, wt = split(acc)
    assert (wt <= f[:, :, None, :] * (1 + 1e-9) + 1e-30).all() and (wt >= 0).all() and (sh >= 0).all()
    assert cnt["diffuse_shifts"] + cnt["failed_shifts"] <= 4 * cnt["evaluations"]
    assert cnt["diffuse_shifts"] > 3 * cnt["evaluations"] and cnt["null_shifts"] == 0
    # no MIS: weights are exactly 1/2 wherever a shift succeeded, 1 where it failed
    p = c.p.copy()
    p.use_mis = 0
    acc2, cnt2, _ = gather(c, p=p)
    f2, sh2, wt2 = split(acc2)
    assert np.allclose(f2, f) and cnt2["evaluations"] == cnt["evaluations"]
    assert (wt2 >= 0.5 * f2[:, :, None, :] * (1 - 1e-9)).all()


def test_fp32_agrees_with_fp64():
    c = make_plane_case("cbox_in", 16, 12, 1500)
    a64, c64, _ = gather(c)
    a32, c32, _ = gather(c, precision=32)
    assert abs(c64["evaluations"] - c32["evaluations"]) <= max(3, 1e-3 * c64["evaluations"])
    lum = a64[..., 0:3].mean()
    assert np.sqrt(((a64 - a32) ** 2).mean()) < 2e-3 * lum


def test_apa_fold_and_linear_radius_ratio():
    c = make_plane_case("cbox_in", 12, 10, 800)
    a1, _, _ = gather(c)
    a2, _, _ = O.gather_planes(c.p, c.m, c.tris, c.beams, c.w1, c.len1, c.rays, 2, c.nb, 64, accum=a1)
    assert np.allclose(a2, a1, rtol=1e-12)  # the same pass folded twice: mean unchanged
    s = O.scale_volume_apa(1.0, 1, c.p.alpha, abi.GVPM_VOL_PLANE0D)
    assert abs(s - c.p.alpha) < 1e-7  # linear ratio for the 0D kernel, gvpm.cpp:195-201


def test_empty_and_invalid_rays():
    c = make_plane_case("cbox_in", 12, 10, 500)
    rays = c.rays.copy()
    rays["info"][:, 0] &= ~np.uint32(1)
    rays["len"][:, 0] = 0
    acc, cnt, _ = gather(c, rays=rays)
    assert cnt["evaluations"] == 0 and not acc.any()
    c.beams = c.beams.subset(np.zeros(0, np.int64))
    acc, cnt, _ = O.gather_planes(c.p, c.m, c.tris, c.beams, c.w1[:0], c.len1[:0], c.rays, 1, c.nb, 64)
    assert cnt["evaluations"] == 0 and not acc.any()
