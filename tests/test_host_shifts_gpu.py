"""Manifold shifts through the host (include/gvpm_hip.h gvpm_enable_host_shifts ... gvpm_upload_host_shifts; SURVEY 8 row
f4): the G-BRE and (round 4) G-VPM gathers record a request for every shift that reaches shiftPhotonManifold, the host
answers, the device finishes the shift (shift_volume_photon.cpp:205-279).  The host's walk is Mitsuba's; here it is
replaced by a STAND-IN (a smooth function of the request and the photon's parent).  The oracle has its own statement of the
stand-in (gvpm_oracle.hpp standinManifoldWalk); the device is answered by a SECOND, independent numpy statement of it
(standin_numpy below; round 3 fed the device the oracle's own routine) -- what is verified is the content of the requests
and the arithmetic the device applies to the answers."""
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


def standin_numpy(ph, req):
    """The stand-in of the host's manifold walk, stated independently of oracle/gvpm_oracle.hpp: a "reconnection" of the
    proposal's vertex c to the photon's parent -- succeeds when the new segment is shorter than three times the old one;
    throughput = prefix * |old| / |new|, new wi towards the parent, pdf = parent pdf * (|old| / |new|)^2 = the determinant
    ratio, base pdf = parent pdf * edge pdf.  float64 on the uploaded fp32 records, rounded once into the answers."""
    out = np.zeros(req.size, abi.HOST_SHIFT_DTYPE)
    k = req["photon"].astype(np.int64)
    par, pos = ph.parent_pos[k].astype(np.float64), ph.pos[k].astype(np.float64)
    d = par - req["offset_pos"].astype(np.float64)
    ln, lb = np.linalg.norm(d, axis=1), np.linalg.norm(par - pos, axis=1)
    ok = (ln > 0) & (ln < 3.0 * lb)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.where(ln > 0, (lb * lb) / (ln * ln), 0.0)
        out["wi"] = np.where(ln[:, None] > 0, d / ln[:, None], 0.0)
        out["throughput"] = ph.prefix_w[k].astype(np.float64) * np.where(ln > 0, lb / ln, 0.0)[:, None]
    out["ok"] = ok
    out["pdf"] = ph.parent_pdf[k].astype(np.float64) * q
    out["det_ratio"] = q
    out["base_pdf"] = ph.parent_pdf[k].astype(np.float64) * ph.edge_pdf[k]
    return out


def test_the_two_statements_of_the_stand_in_agree():
    c = mirror_case()
    acc, st, req, n = device(c, False)
    a, b = standin_numpy(c.ph, req), O.standin_host_shifts(c.ph, req)
    assert np.array_equal(a["ok"], b["ok"]) and a["ok"].sum() > 100 and (~a["ok"].astype(bool)).sum() >= 0
    for k in ("throughput", "wi", "pdf", "det_ratio", "base_pdf"):
        assert np.allclose(a[k], b[k], rtol=2e-6, atol=1e-30), k


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
        ctx.upload_host_shifts(standin_numpy(c.ph, req))
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


# ---------------------------------------------------------------------------------------------------------------- G-VPM
def vpm_mirror_case(**over):
    from test_oracle_vpm import make_vpm_case
    c = make_vpm_case("cbox_mirror", 32, 28, 40000, 6.0, nb=10, use_manifold=1, **over)
    assert (((c.ph.flags >> 2) & 7) == 3).sum() > 500
    return c


def device_vpm(c, answer, cap=1 << 20):
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    ctx.enable_host_shifts(cap)
    ctx.upload_photons(c.ph)
    ctx.upload_camera_beams(c.rays)
    ctx.upload_vpm_samples(c.samples)
    ctx.gather(1, c.nb)
    req, n = ctx.download_shift_requests(cap)
    if answer:
        ctx.upload_host_shifts(standin_numpy(c.ph, req))
    acc = ctx.download_accum().astype(np.float64)
    st = ctx.stats()
    ctx.close()
    return acc, st, req, n


@pytest.mark.parametrize("over", [dict(), dict(use_mis=0), dict(power_heuristic=1)])
def test_vpm_answered_requests_give_the_oracles_manifold_shifts(over):
    """computeVolumeGradientPhoton reaches the same dispatch (VolumeGradientPositionQuery -> shiftPhoton,
    shift_volume_photon.cpp:489-655 -> :49-117): the G-VPM kernel records the same requests, with the pixel's own radius"""
    c = vpm_mirror_case(**over)
    acc, st, req, n = device_vpm(c, True)
    assert n == req.size and n > 300
    ref, sv, nv, cnt, _ = O.gather_vpm(c.p, c.m, c.tris, c.ph, c.rays, c.samples, 64, use_accel=False)
    assert st["evaluations"] == cnt["evaluations"] and st["null_shifts"] == cnt["null_shifts"]
    assert abs(st["diffuse_shifts"] - cnt["diffuse_shifts"]) <= 2 and abs(st["failed_shifts"] - cnt["failed_shifts"]) <= 2
    lum = max(ref[..., 0:3].mean(), 1e-30)
    assert np.sqrt(((acc - ref) ** 2).mean()) / lum < 1e-4
    assert (((c.ph.flags[req["photon"]] >> 2) & 7) == 3).all()
    assert (req["set"] < c.rays.shape[0]).all() and (req["shift"] < 4).all()
    base = c.rays[req["set"], 0]
    sh = c.rays[req["set"], 1 + req["shift"].astype(np.int64)]
    bp = base["o"].astype(np.float64) + base["d"] * req["t"][:, None].astype(np.float64)
    sp = sh["o"].astype(np.float64) + sh["d"] * req["t"][:, None].astype(np.float64)
    assert np.abs(bp - req["base_point"]).max() < 1e-5 and np.abs(sp - req["shift_point"]).max() < 1e-5
    # the radius of a request is its pixel's: R * 0.01 * scaleVol (gvpm.cpp:1132), here the initial one everywhere
    assert np.allclose(req["radius"], np.float32(c.p.bsphere_radius) * np.float32(0.01) * np.float32(c.p.initial_scale_volume), rtol=1e-6)
    assert (np.linalg.norm(req["offset_pos"] - req["shift_point"], axis=1) <= 3.01 * req["radius"]).all()
    # the answered terms matter
    p0 = c.p.copy()
    p0.use_manifold = 0
    ref0, _, _, cnt0, _ = O.gather_vpm(p0, c.m, c.tris, c.ph, c.rays, c.samples, 64, use_accel=False)
    assert np.sqrt(((ref0 - ref) ** 2).mean()) / lum > 1e-3
    # ... and unanswered requests are failed shifts: the gather without the feature
    acc0, st0, _, _ = device_vpm(c, False)
    for k in COUNTERS:
        assert st0[k] == cnt0[k], (k, st0, cnt0)
    assert np.sqrt(((acc0 - ref0) ** 2).mean()) / lum < 1e-4


# -------------------------------------------------------------------------------------------------------------- G-Beams
def standin_numpy_beams(beams, req):
    """The stand-in of the host's walk for G-Beams, stated independently of oracle/gvpm_oracle_beams.hpp (standinBeamWalk):
    the proposal's last edge runs from the beam's origin to the offset position; it succeeds when shorter than three times
    the beam; throughput = prefix * |beam| / e; pdf = origin pdf * (|beam| / e)^2 = the determinant ratio, e^2 = |edge|^2 +
    (|beam| / 10)^2; base pdf = origin pdf * edge pdf.  `wi` is the edge as a VECTOR from the new vertex to the origin (direction and length)."""
    out = np.zeros(req.size, abi.HOST_SHIFT_DTYPE)
    k = req["photon"].astype(np.int64)
    org, end = beams.parent_pos[k].astype(np.float64), beams.pos[k].astype(np.float64)
    wi = org - req["offset_pos"].astype(np.float64)
    ln, lb = np.linalg.norm(wi, axis=1), np.linalg.norm(end - org, axis=1)
    le = np.sqrt(ln * ln + 0.01 * lb * lb)   # (softened by a tenth of the beam's length, as the oracle's)
    q = (lb * lb) / (le * le)
    out["throughput"] = beams.prefix_w[k].astype(np.float64) * (lb / le)[:, None]
    out["wi"] = wi
    out["ok"] = (ln > 0) & (ln < 3.0 * lb)
    out["pdf"] = beams.parent_pdf[k].astype(np.float64) * q
    out["det_ratio"] = q
    out["base_pdf"] = beams.parent_pdf[k].astype(np.float64) * beams.edge_pdf[k]
    return out


def beam_mirror_case(tech, **over):
    from test_oracle_beams import make_beam_case
    c = make_beam_case("cbox_mirror", 32, 28, 12000, 3.0, technique=tech, use_manifold=1, **over)
    assert (((c.beams.flags >> 2) & 7) == 3).sum() > 300
    return c


def device_beams_hs(c, answer, cap=1 << 20):
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    ctx.enable_host_shifts(cap)
    rad = ctx.radius()
    ctx.upload_beams(c.beams, c.end_n)
    ctx.upload_camera_beams(c.rays)
    ctx.gather(1, c.nb)
    req, n = ctx.download_shift_requests(cap)
    if answer:
        ctx.upload_host_shifts(standin_numpy_beams(c.beams, req))
    acc = ctx.download_accum().astype(np.float64)
    st = ctx.stats()
    ctx.close()
    return acc, st, req, n, rad


@pytest.mark.parametrize("tech,over", [(abi.GVPM_BEAM_BEAM_3D_OPTIMIZED, dict()), (abi.GVPM_BEAM_BEAM_3D_OPTIMIZED, dict(use_mis=0)),
                                       (abi.GVPM_BEAM_BEAM_3D_OPTIMIZED, dict(power_heuristic=1)), (abi.GVPM_BEAM_BEAM_1D, dict())])
def test_beams_answered_requests_give_the_oracles_manifold_shifts(tech, over):
    """shiftBeamME (shift_volume_beams.cpp:601-746) split where the data lives: the device records a request per manifold-typed
    beam and shifted ray, a stand-in answers (numpy; the oracle has its own statement), the device applies kernelPDF,
    Jacobian and MIS to the answers"""
    c = beam_mirror_case(tech, **over)
    acc, st, req, n, rad = device_beams_hs(c, True)
    assert n == req.size and n > 300
    ref, cnt, _ = O.gather_beams(c.p, c.m, c.tris, c.beams, c.end_n, c.rays, rad, 1, c.nb, 64)
    assert st["evaluations"] == cnt["evaluations"] and abs(st["null_shifts"] - cnt["null_shifts"]) <= 2
    # (a stand-in walk "fails" where its reach test or kernelPDF flips in fp32)
    assert abs(st["diffuse_shifts"] - cnt["diffuse_shifts"]) <= 4 and abs(st["failed_shifts"] - cnt["failed_shifts"]) <= 4
    lum = max(ref[..., 0:3].mean(), 1e-30)
    assert np.sqrt(((acc - ref) ** 2).mean()) / lum < 2e-4
    # the requests: manifold-typed beams, sets and shifted rays of the upload, both rays at (w - Epsilon), the kernel's place
    # v on the beam, the offset position within reach of the shifted point
    assert (((c.beams.flags[req["photon"]] >> 2) & 7) == 3).all()
    assert (req["set"] < c.rays.shape[0]).all() and (req["shift"] < 4).all()
    eps = np.float64(np.float32(c.p.epsilon))
    base = c.rays[req["set"], 0]
    sh = c.rays[req["set"], 1 + req["shift"].astype(np.int64)]
    tw = (req["t"].astype(np.float64) - eps)[:, None]
    assert np.abs(base["o"].astype(np.float64) + base["d"] * tw - req["base_point"]).max() < 2e-5
    assert np.abs(sh["o"].astype(np.float64) + sh["d"] * tw - req["shift_point"]).max() < 2e-5
    blen = np.linalg.norm(c.beams.pos[req["photon"]].astype(np.float64) - c.beams.parent_pos[req["photon"]], axis=1)
    assert (req["reserved2"] >= -1e-6).all() and (req["reserved2"] <= blen * (1 + 1e-5)).all()
    assert np.array_equal(req["radius"], np.full(n, np.float32(rad)))
    # the answered terms matter, and unanswered requests are failed shifts (the gather without the feature)
    p0 = c.p.copy()
    p0.use_manifold = 0
    ref0, cnt0, _ = O.gather_beams(p0, c.m, c.tris, c.beams, c.end_n, c.rays, rad, 1, c.nb, 64)
    assert np.sqrt(((ref0 - ref) ** 2).mean()) / lum > 1e-3
    acc0, st0, _, _, _ = device_beams_hs(c, False)
    assert st0["evaluations"] == cnt0["evaluations"]
    for k in ("null_shifts", "diffuse_shifts", "failed_shifts"):
        assert abs(st0[k] - cnt0[k]) <= 2, (k, st0, cnt0)
    assert np.sqrt(((acc0 - ref0) ** 2).mean()) / lum < 2e-4


# ---- a real specular walk, stated twice with DIFFERENT algorithms (round 5; VERDICT round 4, missing 3 / next 7) -------------
def mirror_newton_numpy(ph, req, iters=40):
    """The planar-mirror manifold walk of oracle/gvpm_oracle.hpp (mirrorManifoldWalk) as a NEWTON SOLVE -- what a manifold
    walk is (mut_manifold.cpp:1310-1410) -- instead of that header's image construction.  Fermat: the specular point m' of
    the mirror's plane makes the path length |a - p| + |x' - p| stationary, i.e. the tangential components of the half
    vector vanish there.  Newton on that gradient in the plane's two coordinates, with the analytic 2 x 2 Hessian
    T^t [(I - u u^t) / |a - p| + (I - v v^t) / |x' - p|] T (positive definite: the length is convex) and step halving, from
    the old mirror point.  float64 on the uploaded fp32 records.  The fixed end (the record holds no vertex c - 2):
    a = m + parent_wi |m - x|."""
    k = req["photon"].astype(np.int64)
    m = ph.parent_pos[k].astype(np.float64)
    n = ph.parent_n[k].astype(np.float64)
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    x = ph.pos[k].astype(np.float64)
    xo = req["offset_pos"].astype(np.float64)
    d2 = np.linalg.norm(m - x, axis=1)
    a = m + ph.parent_wi[k].astype(np.float64) * d2[:, None]
    h = np.where(np.abs(n[:, :1]) > 0.9, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
    t1 = np.cross(n, h)
    t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
    t2 = np.cross(n, t1)
    da, dx = ((a - m) * n).sum(1), ((xo - m) * n).sum(1)
    front = (da > 0) & (dx > 0)

    def point(uv):
        return m + t1 * uv[:, :1] + t2 * uv[:, 1:2]

    def length(uv):
        p = point(uv)
        return np.linalg.norm(a - p, axis=1) + np.linalg.norm(xo - p, axis=1)

    def grad_hess(uv):
        p = point(uv)
        u, v = a - p, xo - p
        lu, lv = np.linalg.norm(u, axis=1), np.linalg.norm(v, axis=1)
        u, v = u / lu[:, None], v / lv[:, None]
        s_ = u + v
        g = -np.stack([(s_ * t1).sum(1), (s_ * t2).sum(1)], 1)
        T = np.stack([t1, t2], 2)                                    # [k, 3, 2]
        def proj(w, l):                                               # T^t (I - w w^t) T / l
            tw = np.einsum("kij,ki->kj", T, w)
            return (np.eye(2)[None] - tw[:, :, None] * tw[:, None, :]) / l[:, None, None]
        return g, proj(u, lu) + proj(v, lv)

    uv = np.zeros((len(k), 2))
    with np.errstate(all="ignore"):
        for _ in range(iters):
            g, H = grad_hess(uv)
            det = H[:, 0, 0] * H[:, 1, 1] - H[:, 0, 1] * H[:, 1, 0]
            step = -np.stack([(H[:, 1, 1] * g[:, 0] - H[:, 0, 1] * g[:, 1]) / det, (-H[:, 1, 0] * g[:, 0] + H[:, 0, 0] * g[:, 1]) / det], 1)
            step = np.where(front[:, None] & np.isfinite(step).all(1)[:, None], step, 0.0)
            l0 = length(uv)
            t = np.ones(len(k))
            for _ in range(30):                                       # (halve until the length does not grow)
                worse = length(uv + step * t[:, None]) > l0 * (1 + 1e-15)
                if not worse.any():
                    break
                t = np.where(worse, 0.5 * t, t)
            uv = uv + step * t[:, None]
        resid = np.abs(grad_hess(uv)[0]).max(1)
    mn = point(uv)
    d1, d1n, d2n = np.linalg.norm(a - m, axis=1), np.linalg.norm(a - mn, axis=1), np.linalg.norm(mn - xo, axis=1)
    ratio = (d1 + d2) / (d1n + d2n)
    out = np.zeros(req.size, abi.HOST_SHIFT_DTYPE)
    ok = front & (d1n + d2n < 3.0 * (d1 + d2)) & (d2n > 0)
    out["ok"] = ok
    out["wi"] = np.where(front[:, None], (mn - xo) / d2n[:, None], 0.0)
    out["throughput"] = np.where(front[:, None], ph.prefix_w[k].astype(np.float64) * ratio[:, None], 0.0)
    out["det_ratio"] = np.where(front, ratio * ratio, 0.0)
    out["pdf"] = np.where(front, ph.parent_pdf[k].astype(np.float64) * ratio * ratio, 0.0)
    out["base_pdf"] = ph.parent_pdf[k].astype(np.float64) * ph.edge_pdf[k]
    return out, mn, resid, front


def device_answered_by(c, answer_fn, cap=1 << 20):
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    ctx.enable_host_shifts(cap)
    ctx.upload_photons(c.ph)
    ctx.upload_camera_beams(c.rays)
    ctx.gather(c.it, c.nb)
    req, n = ctx.download_shift_requests(cap)
    ctx.upload_host_shifts(answer_fn(c.ph, req))
    acc = ctx.download_accum().astype(np.float64)
    st = ctx.stats()
    ctx.close()
    return acc, st, req, n


@pytest.mark.parametrize("scene", ["cbox_mirror", "cbox_mirror_rot"])
def test_a_newton_mirror_walk_answers_the_device_and_the_oracles_image_construction_agrees(scene):
    """The device is answered by the Newton solve above; the oracle gathers with its OWN statement of the same walk -- the
    mirror image of the offset position, one line-plane intersection: a different algorithm.  Films and counters must agree,
    and request by request the two solvers find the same mirror point."""
    c = cases.make_case(scene, 48, 40, 40000, 4.0 if scene == "cbox_mirror" else 2.5, use_manifold=1)
    assert (((c.ph.flags >> 2) & 7) == 3).sum() > 500
    acc, st, req, n = device_answered_by(c, lambda ph, rq: mirror_newton_numpy(ph, rq)[0])
    assert n == req.size and n > 1000
    newton, mn, resid, front = mirror_newton_numpy(c.ph, req)
    image = O.mirror_host_shifts(c.ph, req)
    # the Newton iterations converged (the half vector's tangential components vanish) wherever the walk is defined
    assert front.sum() > 0.9 * n and resid[front].max() < 1e-9
    assert np.array_equal(newton["ok"], image["ok"]) and newton["ok"].sum() > 0.5 * n
    ok = newton["ok"].astype(bool)
    for k in ("throughput", "wi", "pdf", "det_ratio", "base_pdf"):
        # (both round their float64 results into the fp32 answers; wi: components of a unit vector)
        assert np.allclose(newton[k][ok], image[k][ok], rtol=3e-6, atol=2e-7 if k == "wi" else 1e-30), k
    # the walk MOVES the mirror point (it is not the identity) and keeps it in the mirror's plane
    kk = req["photon"].astype(np.int64)
    m0, nn = c.ph.parent_pos[kk].astype(np.float64), c.ph.parent_n[kk].astype(np.float64)
    moved = np.linalg.norm(mn - m0, axis=1)
    assert np.median(moved[ok]) > 1e-3 and np.abs(((mn - m0) * nn).sum(1))[ok].max() < 1e-9
    O.set_manifold_walk(1)
    try:
        ref, cnt, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, c.it, c.nb, 64, use_accel=True)
    finally:
        O.set_manifold_walk(0)
    for k in COUNTERS:
        # (the walk's reach test -- the path may not grow three-fold -- is a smooth inequality of fp32-rounded answers)
        assert abs(st[k] - cnt[k]) <= (0 if k in ("evaluations", "null_shifts") else 2), (k, st, cnt)
    lum = max(ref[..., 0:3].mean(), 1e-30)
    assert np.sqrt(((acc - ref) ** 2).mean()) / lum < 1e-4
    # ... and it is a different walk from the smooth stand-in: the films differ far beyond the bar
    ref_s, _, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, c.it, c.nb, 64, use_accel=True)
    assert np.sqrt(((ref_s - ref) ** 2).mean()) / lum > 1e-3
