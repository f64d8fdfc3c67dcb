"""The PRIMAL beam radiance estimate (second half; include/gvpm_hip.h gvpm_gather_primal): the oracle's
literal restatement of BeamRadianceEstimator::query / volumePhotonPassBRE (oracle/gvpm_oracle_primal.hpp) against an
independent numpy statement of the estimator, its relation to the gradient functor's base term, and its closed-form limit."""
import numpy as np
import pytest

import cases
import indep_statements as I
import oracle_lib as O
from gvpm_amd import abi


def primal_case(scene="cbox", W=20, H=16, nph=6000, scale=4.0, **kw):
    kw.setdefault("path_set", 0)
    return cases.make_case(scene, W, H, nph, scale, **kw)


@pytest.mark.parametrize("tech", [abi.GVPM_VOL_BRE3D, abi.GVPM_VOL_BRE2D])
@pytest.mark.parametrize("scene", ["cbox", "cbox_hg", "cbox_in"])
def test_oracle_equals_the_independent_statement(tech, scene):
    c = primal_case(scene, vol_technique=tech, use_shift_null=0)
    ref, cnt = O.gather_primal_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 64, use_accel=True)
    acc, n = I.primal_bre_full(c)
    assert n == cnt["evaluations"] > 500
    lum = ref[..., 0:3].mean()
    assert np.abs(acc - ref[..., 0:3]).max() / lum < 1e-9
    assert not ref[..., 3:].any()
    # the walk over the hierarchy evaluates what the loop over all photons does (bre.cpp:179-191: pruning only)
    brute, cntb = O.gather_primal_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 64, use_accel=False)
    assert cntb["evaluations"] == cnt["evaluations"] and np.abs(brute - ref).max() / lum < 1e-12


def test_max_depth_and_the_apa_fold():
    c = primal_case("cbox", max_depth=3)
    ref, cnt = O.gather_primal_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 64)
    acc, n = I.primal_bre_full(c)
    assert n == cnt["evaluations"] and 0 < n
    call = primal_case("cbox")
    assert n < O.gather_primal_bre(call.p, call.m, call.tris, call.ph, call.rays, call.r, 1, call.nb, 64)[1]["evaluations"]
    assert np.abs(acc - ref[..., 0:3]).max() / ref[..., 0:3].mean() < 1e-9
    # gp.fluxVol = (gp.fluxVol * (it - 1) + fluxVolIter) / it, sppm.cpp:986
    two, _ = O.gather_primal_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 2, c.nb, 64, accum=ref)
    assert np.allclose(two, ref, rtol=1e-12)  # the same iteration twice: the mean of two equal estimates
    three, _ = O.gather_primal_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 3, c.nb, 64, accum=np.zeros_like(ref))
    assert np.allclose(three, ref / 3, rtol=1e-12)


def test_2d_primal_is_the_gradient_base_term_without_sigma_s():
    """The 2D kernels of the two estimators differ by the explicit sigma_s of the gradient functor
    (shift_volume_photon.h:78-85 against bre.cpp:246-248), by the gradient functor's missing far check (:726-731, an empty
    block: photons just beyond the beam's end) and by transmittance over (t - eps) in both: with the far-end photons
    masked, primal * sigma_s == base."""
    c = primal_case("cbox", vol_technique=abi.GVPM_VOL_BRE2D, use_shift_null=0)
    prim, cp = O.gather_primal_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 64, use_accel=False)
    grad, cg, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 64, use_accel=False)
    sig_s = np.array(list(c.m.sigma_s))
    extra = cg["evaluations"] - cp["evaluations"]
    assert 0 <= extra <= 0.02 * cp["evaluations"]
    lum = grad[..., 0:3].mean()
    # pixels whose beams met no far-end photon agree to rounding; the others differ by those photons' terms
    d = np.abs(prim[..., 0:3] * sig_s - grad[..., 0:3]).max(-1) / lum
    assert (d < 1e-9).mean() > 0.8 and np.median(d) < 1e-11


def test_3d_primal_averages_to_the_chord_integral():
    """E over the per-hit random number of the 3D term = the chord integral: with a constant integrand (no extinction,
    isotropic phase) every fully contained chord contributes power * phase * chord / (4/3 pi r^3) exactly."""
    c = primal_case("cbox_in", 24, 20, 3000, 5.0)
    for k in range(3):
        c.m.sigma_t[k] = 1e-12  # (sigma_t > 0 is required; its effect vanishes)
    r = float(c.r)
    tot = np.zeros(3)
    for rep in range(24):
        rays = c.rays.copy()
        rays["rand"][:, 0] = (rays["rand"][:, 0] + np.float32(0.61803398875 * (rep + 1))) % np.float32(1.0)
        cc = cases.Case()
        cc.__dict__.update(c.__dict__)
        cc.rays = rays
        acc, n = I.primal_bre_full(cc)
        tot += acc.sum((0, 1))
    mean = tot / 24
    # closed form: every photon x beam with the whole chord inside the usable segment
    pos, power = c.ph.pos.astype(np.float64), c.ph.flux.astype(np.float64)
    eps = float(np.float32(c.p.epsilon))
    want = np.zeros(3)
    part = np.zeros(3)
    for s in c.rays:
        b = s[0]
        o, d, ln = b["o"].astype(np.float64), b["d"].astype(np.float64), float(b["len"])
        a0, L = o + d * eps, (ln - eps) - eps
        t = (pos - a0) @ d
        perp2 = ((a0 + t[:, None] * d - pos) ** 2).sum(1)
        ok = (t > 0) & (perp2 < r * r)
        half = np.sqrt(np.maximum(r * r - perp2, 0))
        lo, hi = np.maximum(t - half, 0), np.minimum(t + half, L)
        inside = ok & (t - 2 * r <= L) & (hi > lo)
        # the estimator draws on the WHOLE chord and drops draws outside [0, L]: expectation = clipped chord length
        chord = np.where(inside, hi - lo, 0.0)
        want += (power * (chord / (4 * np.pi) / (4.0 / 3.0 * np.pi * r ** 3))[:, None]).sum(0) * b["eye"]
    want /= c.nb
    assert np.allclose(mean, want, rtol=0.03)


# ---------------------------------------------------------------------------------------------------------------------
# the primal POINT estimate (sppm.cpp:1040-1126, photonmap.cpp:277-330)
from test_oracle_vpm import make_vpm_case  # noqa: E402


@pytest.mark.parametrize("scene", ["cbox", "cbox_hg", "cbox_mirror"])
@pytest.mark.parametrize("kw", [dict(), dict(max_depth=3), dict(max_depth=2)])
def test_primal_point_estimate_oracle_equals_the_independent_statement(scene, kw):
    c = make_vpm_case(scene, 12, 10, 6000, 8.0, 6, **kw)
    ref, sv, nv, cnt = O.gather_primal_vpm(c.p, c.m, c.tris, c.ph, c.rays, c.samples, 64, use_accel=True)
    acc, mvol, n = I.primal_vpm_full(c)
    assert cnt["evaluations"] == n > 200
    lum = ref[..., 0:3].mean()
    assert np.abs(acc - ref[..., 0:3]).max() / lum < 1e-9 and not ref[..., 3:].any()
    # the kd-tree query evaluates what the loop over all photons does
    brute = O.gather_primal_vpm(c.p, c.m, c.tris, c.ph, c.rays, c.samples, 64, use_accel=False)
    assert brute[3]["evaluations"] == n and np.abs(brute[0] - ref).max() / lum < 1e-12
    # the SPPM update from M (sppm.cpp:1116-1120)
    al = float(c.p.alpha)
    has = mvol > 0
    assert np.allclose(nv[has], al * mvol[has]) and np.allclose(sv[has], c.p.initial_scale_volume * np.cbrt(al))
    assert np.allclose(sv[~has], c.p.initial_scale_volume) and not nv[~has].any()


def test_primal_point_estimate_is_the_gradient_base_term_without_sigma_s():
    c = make_vpm_case("cbox", 12, 10, 6000, 8.0, 6)
    prim = O.gather_primal_vpm(c.p, c.m, c.tris, c.ph, c.rays, c.samples, 64)[0]
    grad = O.gather_vpm(c.p, c.m, c.tris, c.ph, c.rays, c.samples, 64)[0]
    sig_s = np.array(list(c.m.sigma_s))
    assert np.allclose(prim[..., 0:3] * sig_s, grad[..., 0:3], rtol=1e-9, atol=1e-14)


# ---------------------------------------------------------------------------------------------------------------------
# the primal BEAM x BEAM estimate (sppm.cpp:762-880, pm/beams.h:29-223)
from test_oracle_beams import make_beam_case, TECHS  # noqa: E402


@pytest.mark.parametrize("tech", TECHS)
@pytest.mark.parametrize("scene", ["cbox", "cbox_hg", "laser"])
def test_primal_beams_are_the_kernel_records_base_term_over_eps_to_w(tech, scene):
    """BeamRadianceQuery (the literal restatement) against the INDEPENDENT numpy statement of the gradient pass's kernel
    record: the same term but for the camera transmittance, taken over [Epsilon, w] instead of [0, w] -- one factor
    exp(sigma_t Epsilon) -- through the reference's SubBeamBVH and through the loop over all beams."""
    c = make_beam_case(scene, 12, 10, 600, 5.0, technique=tech, path_set=0)
    prim, cnt = O.gather_primal_beams(c.p, c.m, c.tris, c.beams, c.end_n, c.rays, c.r, 1, c.nb, 64, use_accel=True)
    loop, cntl = O.gather_primal_beams(c.p, c.m, c.tris, c.beams, c.end_n, c.rays, c.r, 1, c.nb, 64, use_accel=False)
    acc, icnt = I.beams_full(c)
    assert cnt["evaluations"] == cntl["evaluations"] == icnt["evaluations"] > 100
    f = np.exp(float(c.m.sigma_t[0]) * float(np.float32(c.p.epsilon)))
    lum = prim[..., 0:3].mean()
    assert np.abs(prim[..., 0:3] - acc[..., 0:3] * f).max() / lum < 1e-9 and not prim[..., 3:].any()
    assert np.abs(loop - prim).max() / lum < 1e-12


def test_primal_beams_depth_filters():
    c = make_beam_case("cbox", 12, 10, 600, 5.0, path_set=0, max_depth=3)
    prim, cnt = O.gather_primal_beams(c.p, c.m, c.tris, c.beams, c.end_n, c.rays, c.r, 1, c.nb, 64)
    acc, icnt = I.beams_full(c)
    call = make_beam_case("cbox", 12, 10, 600, 5.0, path_set=0)
    assert 0 < cnt["evaluations"] == icnt["evaluations"] < O.gather_primal_beams(call.p, call.m, call.tris, call.beams, call.end_n, call.rays, call.r, 1, call.nb, 64)[1]["evaluations"]
