"""Camera paths with more than one medium edge: a mirror wall puts a second medium edge on the paths
that meet it.  The synthetic host's shifted paths (half-vector copy at the Dirac vertex, SVertexPDF caches) are checked
against the oracle-side restatements of halfVectorShift (shift_utilities.h:42-110) and GatherPoint::sensorMIS
(gvpm_struct.h:608-631), and the oracle gathers them."""
import numpy as np
import pytest

import cases
import oracle_lib as O
from gvpm_amd import abi

SCENES = ["cbox_mirror", "cbox_mirror_side"]
SIZE = {"cbox_mirror": (64, 48), "cbox_mirror_side": (128, 96)}  # (the side wall fills a narrow strip of the frame)
NORMAL = {"cbox_mirror": np.array([0.0, 0.0, 1.0]), "cbox_mirror_side": np.array([1.0, 0.0, 0.0])}
RHO = np.array([0.9, 0.85, 0.8])


def edges_of(rays):
    return (rays["info"][:, 0] >> 8) & 0xFF


def frame(n):
    """coordinateSystem (util.cpp:600-609): any orthonormal frame around n does for a Dirac vertex."""
    a = np.array([1.0, 0, 0]) if abs(n[0]) < 0.9 else np.array([0, 1.0, 0])
    s = np.cross(n, a)
    s /= np.linalg.norm(s)
    return s, np.cross(n, s)


@pytest.mark.parametrize("scene", SCENES)
def test_two_edge_paths_exist_and_carry_the_mirror(scene):
    sc = cases.SynthScene(scene, *SIZE[scene])
    m = sc.medium()
    rays = sc.camera_beams(1)
    e = edges_of(rays)
    assert set(int(v) for v in np.unique(e)) == {1, 2} and (e == 2).sum() > 50
    px = rays["pixel"][:, 0]
    # the sets of a pixel are consecutive, edge 1 then edge 2, sharing origin -> end point
    i2 = np.nonzero(e == 2)[0]
    assert (e[i2 - 1] == 1).all() and (px[i2 - 1] == px[i2]).all()
    b1, b2 = rays[i2 - 1, 0], rays[i2, 0]
    end1 = b1["o"].astype(np.float64) + b1["d"].astype(np.float64) * b1["len"][:, None].astype(np.float64)
    assert np.allclose(end1, b2["o"], atol=1e-5)
    # eyeContrib(e = 2) = getWeightBeam(1) * getWeightVertex(2) = Tr(len1) * rho * rrWeight, rrWeight = 1 / min(Tr max(rho), 0.95)
    tr = np.exp(-float(m.sigma_t[0]) * b1["len"].astype(np.float64))
    rr = 1.0 / np.minimum(tr * RHO.max(), 0.95)
    assert np.allclose(b2["eye"], (tr * rr)[:, None] * RHO, rtol=1e-5)
    assert np.allclose(b1["eye"], 1.0)
    # the base caches: a Dirac vertex multiplies the pdf by 1 and leaves the Jacobian (generateVertexInfo)
    assert np.allclose(b2["pdf"], b1["pdf"]) and (b2["jacobian"] == 1).all() and (b1["jacobian"] == 1).all()
    # GOp(2) is the second edge's own geometry term: not the first edge's
    assert not np.allclose(b2["gop"], b1["gop"])


@pytest.mark.parametrize("scene", SCENES)
def test_shifted_second_edge_is_the_half_vector_copy(scene):
    sc = cases.SynthScene(scene, *SIZE[scene])
    rays = sc.camera_beams(2)
    e = edges_of(rays)
    i2 = np.nonzero(e == 2)[0]
    n = NORMAL[scene]
    s, t = frame(n)
    to_local = lambda v: np.array([v @ s, v @ t, v @ n])
    to_world = lambda v: s * v[0] + t * v[1] + n * v[2]
    checked = 0
    for i in i2:
        b1, b2 = rays[i - 1, 0], rays[i, 0]
        base_wi, base_wo = -b1["d"].astype(np.float64), b2["d"].astype(np.float64)
        for k in range(1, 5):
            s1, s2 = rays[i - 1, k], rays[i, k]
            if not (s2["info"] & 1):
                continue
            assert s1["info"] & 1  # an edge behind a valid one
            ok, wo, jac = O.half_vector_shift(to_local(base_wi), to_local(base_wo), to_local(-s1["d"].astype(np.float64)))
            assert ok
            assert np.allclose(to_world(wo), s2["d"], atol=2e-6)
            # the reflection Jacobian |wo'.h / wo.h| -- which trace() then overrides with 1 for a Dirac component
            # (shift_cameraPath.h:317-318): the cache entry carries no trace of it
            assert jac > 0 and s2["jacobian"] == s1["jacobian"] and s2["pdf"] == s1["pdf"]
            # the second edge starts where the first one of the same (shifted) path ends
            end = s1["o"].astype(np.float64) + s1["d"].astype(np.float64) * float(s1["len"])
            assert np.allclose(end, s2["o"], atol=1e-5)
            checked += 1
    assert checked > 200
    # refraction branch of the same function: eta = 1 refuses, a real interface gives Snell's direction
    ok, _, _ = O.half_vector_shift([0.3, 0, 0.954], [-0.2, 0, -0.98], [0.31, 0, 0.951], 1.0, 1.0)
    assert not ok
    wi = np.array([np.sin(0.4), 0, np.cos(0.4)])
    eta = 1.5
    wo = np.array([-np.sin(0.4) / eta, 0, -np.sqrt(1 - (np.sin(0.4) / eta) ** 2)])
    ok, got, jac = O.half_vector_shift(wi, wo, wi, eta, eta)
    assert ok and np.allclose(got, wo, atol=1e-12) and abs(jac - 1) < 1e-9  # the identity shift


@pytest.mark.parametrize("scene", SCENES)
def test_sensor_mis_as_written_equals_the_cancelled_form(scene):
    """sensorMIS multiplies the Jacobian by G_s/G_b and the pdf ratio by G_b/G_s for idVertex != 1: the product the
    device evaluates (shift_device.h sensorMIS) is the literal one."""
    sc = cases.SynthScene(scene, *SIZE[scene])
    rays = sc.camera_beams(1)
    e = edges_of(rays)
    n = 0
    for i in np.nonzero(e == 2)[0][:200]:
        b = rays[i, 0]
        for k in range(1, 5):
            sft = rays[i, k]
            if not (sft["info"] & 1):
                continue
            lit = O.sensor_mis(2, float(sft["pdf"]), float(sft["jacobian"]), float(sft["gop"]), float(b["pdf"]), float(b["gop"]))
            assert abs(lit - float(sft["pdf"]) / float(b["pdf"]) * float(sft["jacobian"])) <= 1e-12 * abs(lit)
            n += 1
    assert n > 100
    if scene == "cbox_mirror_side":
        sh = rays[e == 2][:, 1:]
        v = (sh["info"] & 1) != 0
        assert (np.abs(sh["jacobian"][v] - 1) > 1e-4).any()  # grazing incidence: a real Jacobian


@pytest.mark.parametrize("scene", SCENES)
def test_oracle_gathers_two_edge_paths(scene):
    c = cases.make_case(scene, *((40, 32) if scene == "cbox_mirror" else (96, 72)), 15000, 3.0)
    e = edges_of(c.rays)
    assert (e == 2).sum() > 30
    ref, cnt, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 64, use_accel=False)
    assert cnt["evaluations"] > 5000 and cnt["failed_shifts"] > 0   # (light paths through the mirror: manifold-type shifts)
    # the edge-2 sets contribute to their pixels: gathering them alone gives a part of the full estimate
    only2, cnt2, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, np.ascontiguousarray(c.rays[e == 2]), c.r, 1, c.nb, 64, use_accel=False)
    only1, cnt1, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, np.ascontiguousarray(c.rays[e == 1]), c.r, 1, c.nb, 64, use_accel=False)
    assert cnt2["evaluations"] > 100 and cnt1["evaluations"] + cnt2["evaluations"] == cnt["evaluations"]
    assert np.allclose(only1 + only2, ref, rtol=1e-12, atol=1e-300)
    # shifted rays equal to the base ray: w = 1/2 and zero gradient on every edge (the oracle invariant (i) of SURVEY 8c)
    same = cases.rays_shift_equals_base(c.rays)
    acc, _, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, same, c.r, 1, c.nb, 64, use_accel=False)
    H, W = acc.shape[:2]
    flux, shifted, weighted = acc[..., 0:3], acc[..., 3:15].reshape(H, W, 4, 3), acc[..., 15:27].reshape(H, W, 4, 3)
    lit = flux.sum(-1) > 0
    assert np.allclose(shifted[lit], weighted[lit], rtol=1e-9)
    # accel independence
    ref2, cntb, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 64, use_accel=True)
    assert cntb["evaluations"] == cnt["evaluations"] and np.allclose(ref2, ref, rtol=1e-10)


def test_vpm_edge_selection_follows_the_cdf():
    """gvpm.cpp:1117-1172: the samples of a two-edge pixel pick an edge with probability weightBeam.max() / sum and
    carry that probability; sampleReuse re-stretches the random number."""
    sc = cases.SynthScene("cbox_mirror", 64, 48)
    m = sc.medium()
    nb = 64
    rays, smp = sc.camera_beams_and_vpm_samples(1, nb)
    e = edges_of(rays)
    px = rays["pixel"][:, 0]
    assert smp.shape[0] == np.unique(px).size * nb      # nbCameraSamples per PIXEL, not per set
    assert ((smp["rand"] >= 0) & (smp["rand"] < 1)).all()
    i2 = np.nonzero(e == 2)[0]
    hits2, exp2 = 0, 0.0
    for i in i2[:40]:
        tr = np.exp(-float(m.sigma_t[0]) * float(rays[i - 1, 0]["len"]))
        a = RHO.max() * tr
        mine = smp[(smp["set"] == i - 1) | (smp["set"] == i)]
        assert mine.shape[0] == nb
        on1, on2 = mine[mine["set"] == i - 1], mine[mine["set"] == i]
        assert np.allclose(on1["pdf_sel"], 1 / (1 + a), rtol=1e-5) and np.allclose(on2["pdf_sel"], a / (1 + a), rtol=1e-5)
        hits2 += on2.shape[0]
        exp2 += a / (1 + a)
    # the empirical frequency of the second edge over 40 pixels x 64 samples against its mean probability
    assert abs(hits2 / (40 * nb) - exp2 / 40) < 0.03 and hits2 > 100
    single = smp[np.isin(smp["set"], np.nonzero(np.bincount(np.unique(px, return_inverse=True)[1])[np.unique(px, return_inverse=True)[1]] == 1)[0])]
    assert (single["pdf_sel"] == 1).all()
    # the oracle takes them
    p = sc.params()
    p.vol_technique = abi.GVPM_DISTANCE
    p.nb_camera_samples = nb
    p.initial_scale_volume = 5.0
    ph, nbp = sc.shoot_photons(1, 20000)
    acc, sv, nv, cnt, _ = O.gather_vpm(p, m, sc.triangles(), ph, rays, smp, 64, use_accel=True)
    assert cnt["evaluations"] > 2000
