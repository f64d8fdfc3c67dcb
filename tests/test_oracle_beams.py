"""Pins the G-Beams part of the oracle (computeVolumeGradientBeams, BeamKernelRecord, beam shifts)."""
import numpy as np
import pytest

import cases
import oracle_lib as O
from gvpm_amd import abi


def make_beam_case(scene="cbox", W=16, H=12, nbeams=3000, scale=3.0, technique=abi.GVPM_BEAM_BEAM_3D_OPTIMIZED,
                   it=1, **kw):
    if technique == abi.GVPM_BEAM_BEAM_1D:
        kw.setdefault("use_shift_null", 0)
    c = cases.make_case(scene, W, H, 10, scale, it=it, vol_technique=technique, **kw)
    c.beams, c.end_n, c.nb = c.sc.shoot_beams(it, nbeams)
    return c


TECHS = [abi.GVPM_BEAM_BEAM_3D_OPTIMIZED, abi.GVPM_BEAM_BEAM_1D]


def split(acc):
    H, W = acc.shape[:2]
    return acc[..., 0:3], acc[..., 3:15].reshape(H, W, 4, 3), acc[..., 15:27].reshape(H, W, 4, 3)


def numpy_beam1d_base(c):
    """Independent numpy statement of the 1D beam x beam base estimator: line-line closest
    approach (pm/beams_struct.h:250-311), contrib of beams_struct.h:136-185, kernel 0.5/r."""
    p = c.p
    out = np.zeros((p.height, p.width, 3))
    p1 = c.beams.parent_pos.astype(np.float64)
    p2 = c.beams.pos.astype(np.float64)
    dv = p2 - p1
    ln = np.linalg.norm(dv, axis=1)
    bd = dv / ln[:, None]
    flux = c.beams.flux.astype(np.float64)
    depth = ((c.beams.flags >> 8) & 0xFF).astype(np.int64)
    parity = (c.beams.path_id & 1).astype(np.int64)
    st = float(c.m.sigma_t[0])
    ss = np.array(list(c.m.sigma_s), np.float64)
    r = c.r
    eps = float(p.epsilon)
    n = 0
    for s in range(c.rays.shape[0]):
        b = c.rays[s, 0]
        o, d, L = b["o"].astype(np.float64), b["d"].astype(np.float64), float(b["len"])
        px, py = int(b["pixel"]) & 0xFFFF, int(b["pixel"]) >> 16
        edge = (int(b["info"]) >> 8) & 0xFF
        cr = np.cross(d, bd)
        sin2 = (cr * cr).sum(1)
        ad = ((p1 - o) * cr).sum(1)
        ok = ad * ad < r * r * sin2
        c12 = bd @ d
        den = c12 * c12 - 1.0
        ok &= np.abs(den) >= 1e-5
        with np.errstate(divide="ignore", invalid="ignore"):
            w = ((d @ o) - (p1 @ d) - c12 * ((bd @ o) - (bd * p1).sum(1))) / den
            v = (w + (d @ o) - (p1 @ d)) / c12
        ok &= (w > eps) & (w < L - eps) & (v > 0) & (v < ln)
        if p.max_depth > 0:
            ok &= (depth + edge) <= p.max_depth
        rr = 1.0
        if p.path_set:
            ok &= parity == ((px + py) % 2)
            rr = 2.0
        idx = np.nonzero(ok)[0]
        n += idx.size
        if idx.size == 0:
            continue
        sinT = np.sqrt(sin2[idx])
        trb, trc = np.exp(-st * v[idx]), np.exp(-st * w[idx])
        contrib = (trb * trc / (4 * np.pi) / trb / sinT)[:, None] * flux[idx] * ss  # / pdfFailure (= Tr) / sin
        out[py, px] += contrib.sum(0) * (0.5 / r) * rr * b["eye"].astype(np.float64)
    return out, n


def test_beam1d_base_matches_numpy():
    c = make_beam_case(technique=abi.GVPM_BEAM_BEAM_1D)
    acc, cnt, _ = O.gather_beams(c.p, c.m, c.tris, c.beams, c.end_n, c.rays, c.r, 1, c.nb, 64)
    ref, n = numpy_beam1d_base(c)
    assert abs(cnt["evaluations"] - n) <= 2 and n > 1000   # float intermediates of the reference at the boundaries
    flux = split(acc)[0] * c.nb
    assert np.abs(flux - ref).max() < 2e-3 * ref.max()
    assert abs(flux.sum() - ref.sum()) < 1e-4 * ref.sum()


@pytest.mark.parametrize("tech", TECHS)
def test_sub_beam_ownership_is_cut_independent(tech):
    """SubBeamBVH cuts beams into avgLen/10 pieces (pm/beams_accel.h:98-131); the functor's
    ownership rule makes the result independent of the cut."""
    c = make_beam_case(technique=tech)
    a, ca, _ = O.gather_beams(c.p, c.m, c.tris, c.beams, c.end_n, c.rays, c.r, 1, c.nb, 64, sub_beam_size=0.0)
    for sub in (0.31, 0.07):
        b, cb, _ = O.gather_beams(c.p, c.m, c.tris, c.beams, c.end_n, c.rays, c.r, 1, c.nb, 64, sub_beam_size=sub)
        assert ca["evaluations"] == cb["evaluations"]
        assert np.allclose(a, b, rtol=1e-12, atol=1e-15 * a.max())


@pytest.mark.parametrize("tech", TECHS)
def test_beams_identical_shifted_rays(tech):
    c = make_beam_case(technique=tech)
    rays = cases.rays_shift_equals_base(c.rays)
    acc, cnt, _ = O.gather_beams(c.p, c.m, c.tris, c.beams, c.end_n, rays, c.r, 1, c.nb, 64)
    flux, sh, wt = split(acc)
    inner = np.s_[:-1, :-1]
    # 3D: w = 1/2 up to the sigma_t*Epsilon asymmetry of the ray minima; 1D: the float line-line
    # intersection and the asin-based getShiftPos1D reproduce the base point only to ~1e-2
    tol = 3e-4 if tech == abi.GVPM_BEAM_BEAM_3D_OPTIMIZED else 3e-2
    for i in range(4):
        assert np.allclose(wt[inner][:, :, i], 0.5 * flux[inner], rtol=tol, atol=1e-12)
        assert np.allclose(sh[inner][:, :, i], 0.5 * flux[inner], rtol=tol, atol=1e-12)
    assert cnt["failed_shifts"] == 0


@pytest.mark.parametrize("tech", TECHS)
def test_beams_weights_and_flags(tech):
    c = make_beam_case(technique=tech)
    acc, cnt, _ = O.gather_beams(c.p, c.m, c.tris, c.beams, c.end_n, c.rays, c.r, 1, c.nb, 64)
    flux, sh, wt = split(acc)
    assert (wt >= 0).all() and (sh >= 0).all() and (wt <= flux[:, :, None, :] * (1 + 1e-9) + 1e-15).all()
    W, H = c.p.width, c.p.height
    assert np.allclose(wt[H - 1, :, abi.GVPM_TOP], flux[H - 1], rtol=1e-9, atol=1e-14)
    p = c.p.copy()
    p.use_mis = 0
    f2 = split(O.gather_beams(p, c.m, c.tris, c.beams, c.end_n, c.rays, c.r, 1, c.nb, 64)[0])[0]
    assert np.array_equal(f2, flux)
    a32, c32, _ = O.gather_beams(c.p, c.m, c.tris, c.beams, c.end_n, c.rays, c.r, 1, c.nb, 32)
    assert abs(c32["evaluations"] - cnt["evaluations"]) <= 3
    assert np.sqrt(((a32 - acc) ** 2).mean()) / flux.mean() < 1e-3


def test_beam_records_follow_the_path_conventions():
    c = make_beam_case()
    b = c.beams
    ptype = b.flags & 3
    depth = (b.flags >> 8) & 0xFF
    assert (depth[ptype == 0] == 1).all() and depth.min() >= 1
    # beam flux excludes the transmittance of its own edge: flux = prefix * v.weight * rr
    med = ptype == 2
    assert np.allclose(b.flux[med], b.prefix_w[med] * 0.5 * b.parent_rr[med, None], rtol=1e-5)
    on_surface = np.abs(c.end_n).sum(1) > 0
    end = b.pos[on_surface]
    assert (np.abs(end).max(axis=1) > 0.99).all()          # surface ends lie on the walls / the light
    assert 0.2 < on_surface.mean() < 0.5
