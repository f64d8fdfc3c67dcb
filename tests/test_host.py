"""Host-side logic (synthetic Mitsuba stand-ins): light-path flattening conventions, camera beams."""
import numpy as np

import cases
from gvpm_amd import abi
from gvpm_amd.host import SynthScene


def test_photon_records_follow_the_path_conventions():
    sc = SynthScene("cbox", 16, 16)
    ph, nb = sc.shoot_photons(1, 5000)
    assert ph.n == 5000 and nb > 0
    ptype = ph.flags & 3
    shift = (ph.flags >> 2) & 7
    depth = (ph.flags >> 8) & 0xFF
    assert set(np.unique(ptype)) <= {0, 1, 2}
    assert (shift == 1).all()                       # isotropic fog + Lambertian walls: diffuse parents
    assert ((ph.flags >> 5) & 1).all()               # every light-path edge is inside the medium
    assert depth.min() >= 1 and depth.max() <= 11   # vertexId - 1, maxDepth = 12
    assert (depth[ptype == 0] == 1).all()            # parent = emitter sample <=> photon is vertex 2
    assert np.allclose(np.linalg.norm(ph.wi, axis=1), 1, atol=1e-5)
    # wi points from the photon to its parent
    d = (ph.parent_pos - ph.pos).astype(np.float64)
    ln = np.linalg.norm(d, axis=1)
    far = ln > 1e-2   # fp32-rounded endpoints: only well-separated pairs give an accurate direction
    assert ((d[far] / ln[far, None] * ph.wi[far]).sum(1) > 1 - 1e-6).all()
    # flux = prefix * parent.weight * rr * edge.weight; medium parent: sigma_s * rr / sigma_t
    med = ptype == 2
    assert np.allclose(ph.flux[med], ph.prefix_w[med] * 0.5 * ph.parent_rr[med, None], rtol=1e-5)
    emi = ptype == 0
    assert np.allclose(ph.flux[emi], ph.prefix_w[emi] * ph.parent_rr[emi, None], rtol=1e-5)
    assert np.allclose(ph.prefix_w[emi], 15 * np.pi * 0.25, rtol=1e-5)   # radiance * pi * area
    sur = ptype == 1
    assert np.allclose(ph.flux[sur], ph.prefix_w[sur] * ph.parent_scat[sur] * ph.parent_rr[sur, None], rtol=1e-5)
    # path ids are non-decreasing and dense
    assert (np.diff(ph.path_id.astype(np.int64)) >= 0).all() and ph.path_id[0] == 0
    assert (np.abs(ph.pos) <= 1.0 + 1e-6).all()
    assert (ph.parent_pdf > 0).all() and (ph.edge_pdf > 0).all()


def test_photon_shooting_is_deterministic_and_iteration_keyed():
    sc = SynthScene("cbox", 8, 8)
    a, na = sc.shoot_photons(3, 4000)
    b, nb = sc.shoot_photons(3, 4000)
    c, nc = sc.shoot_photons(4, 4000)
    assert na == nb and np.array_equal(a.pos, b.pos) and np.array_equal(a.flux, b.flux)
    assert not np.array_equal(a.pos, c.pos)
    # capacity semantics (gvpm_proc.cpp:278-350): a prefix of the same path sequence
    d, nd = sc.shoot_photons(3, 1000)
    assert nd <= na and np.array_equal(d.pos, a.pos[:1000])


def test_hg_scene_produces_medium_shift_types():
    sc = SynthScene("cbox_hg", 8, 8)
    ph, _ = sc.shoot_photons(1, 5000)
    shift = (ph.flags >> 2) & 7
    ptype = ph.flags & 3
    # g = 0.7 > 0.5: medium vertices are "glossy" (gvpm_struct.h:73-76) -> medium parents give
    # EMediumShift, surface / emitter parents EDiffuseShift
    assert (shift[ptype == 2] == 2).all() and (shift[ptype != 2] == 1).all()
    assert (shift == 2).any()


def test_camera_beam_sets():
    sc = SynthScene("cbox", 32, 24)
    rays = sc.camera_beams(1)
    assert rays.shape[1] == 5 and 0 < rays.shape[0] <= 32 * 24
    base = rays[:, 0]
    assert ((base["info"] & 1) == 1).all() and (((base["info"] >> 8) & 0xFF) == 2).all()
    assert np.allclose(np.linalg.norm(base["d"], axis=1), 1, atol=1e-5)
    assert np.allclose(base["o"][:, 2], 1.0, atol=1e-5)          # enters through the front face
    end = base["o"] + base["d"] * base["len"][:, None]
    assert (np.abs(end).max(axis=1) <= 1 + 1e-4).all() and (np.abs(end).max(axis=1) >= 0.997).all()
    assert (base["rand"] >= 0).all() and (base["rand"] < 1).all()
    assert np.allclose(base["jacobian"], 1)
    # sensorMIS == 1 analytically for primary beams of a pinhole
    for k in range(1, 5):
        s = rays[:, k]
        v = (s["info"] & 1) == 1
        mis = (s["pdf"][v] / base["pdf"][v]) * s["jacobian"][v]
        assert np.allclose(mis, 1, rtol=1e-4)
    # pixel windows tile the frame
    a = sc.camera_beams(1, 0, 0, 16, 24)
    b = sc.camera_beams(1, 16, 0, 32, 24)
    assert a.shape[0] + b.shape[0] == rays.shape[0]
    px, py = cases.pixels_of(rays)
    assert px.max() < 32 and py.max() < 24
    # shifted rays go through the neighbouring pixels with the same sub-pixel offset
    tx = np.tan(np.radians(39) / 2)
    for k, (dx, dy) in enumerate([(-1, 0), (1, 0), (0, 1), (0, -1)], start=1):
        s = rays[:, k]
        v = (s["info"] & 1) == 1
        bx = base["d"][v, 0] / -base["d"][v, 2]
        sx = s["d"][v, 0] / -s["d"][v, 2]
        assert np.allclose(sx - bx, dx * 2 * tx / 32, atol=1e-5)


def test_unknown_scene_is_rejected():
    import pytest
    with pytest.raises(ValueError):
        SynthScene("nope", 8, 8)
