"""Pins the G-VPM part of the oracle (computeVolumeGradientPhoton + VolumeGradientPositionQuery)."""
import numpy as np
import pytest

import cases
import oracle_lib as O
from gvpm_amd import abi


def make_vpm_case(scene="cbox", W=20, H=16, nph=20000, scale=6.0, nb=8, it=1, **kw):
    c = cases.make_case(scene, W, H, nph, scale, it=it, vol_technique=abi.GVPM_DISTANCE, nb_camera_samples=nb, **kw)
    c.rays, c.samples = c.sc.camera_beams_and_vpm_samples(it, nb)
    return c


def numpy_vpm_base(c):
    """Independent numpy statement of the G-VPM base estimator (gvpm.cpp:1143-1180,
    shift_volume_photon.cpp:489-531, homogeneous.cpp:293-430)."""
    p = c.p
    H, W = p.height, p.width
    out = np.zeros((H, W, 3))
    mvol = np.zeros((H, W))
    pos = c.ph.pos.astype(np.float64)
    flux = c.ph.flux.astype(np.float64)
    depth = ((c.ph.flags >> 8) & 0xFF).astype(np.int64)
    st = float(c.m.sigma_t[1])
    ss = np.array(list(c.m.sigma_s), np.float64)
    eps = float(p.epsilon)
    r = float(np.float32(p.bsphere_radius)) * 0.01 * float(p.initial_scale_volume)
    kv = 4.0 / 3.0 * np.pi * r ** 3
    evals = 0
    for sm in c.samples:
        b = c.rays[sm["set"], 0]
        o, d, ln = b["o"].astype(np.float64), b["d"].astype(np.float64), float(b["len"])
        px, py = int(b["pixel"]) & 0xFFFF, int(b["pixel"]) >> 16
        edge = (int(b["info"]) >> 8) & 0xFF
        max_dist = max((ln - eps) - eps, 0.0)
        nrm = 1 - np.exp(-st * max_dist)
        sd = -np.log(1 - float(sm["rand"]) * nrm) / st
        t = sd + eps
        nrm2 = 1 - np.exp(-st * (ln - eps))
        pdf = st / nrm2 * np.exp(-st * sd) * float(sm["pdf_sel"])
        tr = np.exp(-st * sd)
        q = o + d * t
        d2 = ((pos - q) ** 2).sum(1)
        inside = d2 < r * r
        mvol[py, px] += inside.sum()
        keep = inside.copy()
        if p.max_depth > 0:
            keep &= (depth + edge) <= p.max_depth
        evals += keep.sum()
        contrib = (flux[keep] * ss / (4 * np.pi)).sum(0) * tr * b["eye"].astype(np.float64)
        out[py, px] += contrib / (kv * pdf) / p.nb_camera_samples
    return out, mvol, evals


@pytest.fixture(scope="module")
def case():
    return make_vpm_case()


def test_vpm_base_matches_numpy(case):
    c = case
    acc, sv, nv, cnt, _ = O.gather_vpm(c.p, c.m, c.tris, c.ph, c.rays, c.samples, 64, use_accel=False)
    ref, mvol, evals = numpy_vpm_base(c)
    assert cnt["evaluations"] == evals and evals > 1000
    assert np.allclose(acc[..., 0:3], ref, rtol=1e-9, atol=1e-12 * ref.max())
    # SPPM statistics, gvpm.cpp:1191-1195 (first iteration: N = 0 -> ratio = alpha)
    al = float(c.p.alpha)
    has = mvol > 0
    assert np.allclose(nv[has], al * mvol[has])
    assert np.allclose(sv[has], c.p.initial_scale_volume * np.cbrt(al))
    assert np.allclose(sv[~has], c.p.initial_scale_volume) and not nv[~has].any()


def test_vpm_kdtree_query_equals_bruteforce(case):
    c = case
    a = O.gather_vpm(c.p, c.m, c.tris, c.ph, c.rays, c.samples, 64, use_accel=True)
    b = O.gather_vpm(c.p, c.m, c.tris, c.ph, c.rays, c.samples, 64, use_accel=False)
    for k in ("evaluations", "null_shifts", "diffuse_shifts", "failed_shifts"):
        assert a[3][k] == b[3][k]
    assert np.allclose(a[0], b[0], rtol=1e-12, atol=1e-14 * a[0].max())
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_vpm_identical_shifted_beams_half_weight(case):
    c = case
    rays = cases.rays_shift_equals_base(c.rays)
    for null in (1, 0):
        p = c.p.copy()
        p.use_shift_null = null
        acc = O.gather_vpm(p, c.m, c.tris, c.ph, rays, c.samples, 64)[0]
        H, W = acc.shape[:2]
        flux, sh, wt = acc[..., 0:3], acc[..., 3:15].reshape(H, W, 4, 3), acc[..., 15:27].reshape(H, W, 4, 3)
        inner = np.s_[:-1, :-1]
        # not exactly 1/2: the base distance pdf / transmittance use t - mint (sampleDistance) while the
        # shifted ones use t (eval with EDistanceAlwaysValid) -- a sigma_t * Epsilon = 1e-4 asymmetry of the
        # reference (homogeneous.cpp:350-354 vs :473-476), restated literally
        for i in range(4):
            assert np.allclose(wt[inner][:, :, i], 0.5 * flux[inner], rtol=2e-4, atol=1e-12)
            assert np.allclose(sh[inner][:, :, i], 0.5 * flux[inner], rtol=2e-4, atol=1e-12)


def test_vpm_accumulates_sums_and_shrinks_radii():
    c = make_vpm_case(nph=15000)
    acc = sv = nv = None
    singles = []
    for it in (1, 2):
        ph, nb = c.sc.shoot_photons(it, 15000)
        rays, smp = c.sc.camera_beams_and_vpm_samples(it, c.p.nb_camera_samples)
        acc, sv, nv, cnt, _ = O.gather_vpm(c.p, c.m, c.tris, ph, rays, smp, 64, accum=acc, scale_vol=sv, n_vol=nv)
        singles.append(cnt["evaluations"])
    assert (sv <= c.p.initial_scale_volume).all() and (sv < c.p.initial_scale_volume).any()
    assert singles[1] < singles[0]  # smaller radii gather fewer photons
    thr, dx, dy = O.assemble(acc, 2, False, total_emitted=1000.0)
    assert np.allclose(thr, acc[..., 0:3] / 1000.0)


def test_vpm_float_and_double_agree(case):
    c = case
    a64 = O.gather_vpm(c.p, c.m, c.tris, c.ph, c.rays, c.samples, 64)
    a32 = O.gather_vpm(c.p, c.m, c.tris, c.ph, c.rays, c.samples, 32)
    assert abs(a64[3]["evaluations"] - a32[3]["evaluations"]) <= 3
    lum = a64[0][..., 0:3].mean()
    assert np.sqrt(((a64[0] - a32[0]) ** 2).mean()) / lum < 1e-3
