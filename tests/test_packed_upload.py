"""The packed upload records (include/gvpm_hip.h "packed uploads"; gvpm_amd/csrc/pack_codec.h): host-side pack / unpack
(plain C, no GPU).  What a packed upload MEANS is defined by gvpm_unpack_*; here that definition is pinned by an
independent numpy decode, and the losses of the format are bounded."""
import numpy as np
import pytest

import cases
from gvpm_amd import abi, hip


def angle(a, b):
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    return np.linalg.norm(np.cross(a, b), axis=1) / np.maximum(np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1), 1e-300)


def oct_decode_numpy(w):
    """octahedral 2 x snorm16 -> unit vector, in float64, rounded once"""
    ix = (w & 0xFFFF).astype(np.uint16).view(np.int16).astype(np.float64) / 32767.0
    iy = (w >> 16).astype(np.uint16).view(np.int16).astype(np.float64) / 32767.0
    z = 1.0 - np.abs(ix) - np.abs(iy)
    fx = np.where(z < 0, (1.0 - np.abs(iy)) * np.where(ix >= 0, 1.0, -1.0), ix)
    fy = np.where(z < 0, (1.0 - np.abs(ix)) * np.where(iy >= 0, 1.0, -1.0), iy)
    ln = np.sqrt(fx * fx + fy * fy + z * z)
    v = np.stack([fx / ln, fy / ln, z / ln], 1).astype(np.float32)
    v[w == 0x80008000] = 0
    return v


@pytest.mark.parametrize("scene", ["cbox", "cbox_hg", "laser"])
def test_unpack_is_what_the_header_says(scene):
    c = cases.make_case(scene, 24, 20, 30000, 3.0)
    t = hip.MaterialTable()
    pk = hip.pack_photons(c.ph, t)
    assert pk.dtype.itemsize == 76 and 1 <= t.n <= 16
    got = hip.unpack_photons(pk, t)
    # carried as they are
    for k in ("pos", "parent_pos", "flux", "prefix_w", "parent_pdf", "edge_pdf", "parent_rr", "parent_scat", "parent_g", "flags"):
        assert np.array_equal(getattr(got, k), getattr(c.ph, k)), k
    assert np.array_equal(got.path_id, c.ph.path_id & 1)
    # derived / quantised, against an independent statement
    d = c.ph.parent_pos.astype(np.float64) - c.ph.pos.astype(np.float64)
    ln = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])
    wi = (d / ln[:, None]).astype(np.float32)
    assert np.array_equal(got.wi, wi)
    assert np.array_equal(got.parent_n, oct_decode_numpy(pk["parent_n_oct"]))
    assert np.array_equal(got.parent_wi, oct_decode_numpy(pk["parent_wi_oct"]))
    # the losses: the walls of these scenes are axis aligned (exact); other unit vectors within 7e-5 rad; the derived
    # wi is the stored one to rounding except where the photon sits within ~1e-4 of its parent (position rounding)
    assert np.array_equal(got.parent_n, c.ph.parent_n)
    assert angle(got.parent_wi, c.ph.parent_wi).max() < 7e-5
    e = angle(got.wi, c.ph.wi)
    far = ln > 1e-2
    assert e[far].max() < 2e-5 and np.sqrt((e ** 2).mean()) < 2e-4 and (e > 1e-3).mean() < 1e-3


def test_zero_vectors_parity_bit_and_material_table():
    ph = abi.Photons(5)
    rng = np.random.default_rng(1)
    ph.pos[:] = rng.random((5, 3))
    ph.parent_pos[:] = rng.random((5, 3))
    ph.parent_pos[4] = ph.pos[4]           # degenerate edge: wi = 0
    ph.parent_n[1] = (0, 0, -1)
    ph.parent_n[2] = (-1, 0, 0)
    ph.parent_n[3] = (0.6, 0, -0.8)
    ph.parent_wi[3] = (0, -1, 0)
    ph.flags[:] = [abi.pf_make(1, 1, 1, d, 0x1234) for d in range(5)] if hasattr(abi, "pf_make") else np.arange(5) << 8
    ph.path_id[:] = [10, 11, 12, 13, 0xFFFFFFFF]
    ph.parent_scat[:] = [(0.5, 0.5, 0.5), (0.5, 0.5, 0.5), (1, 2, 3), (0.5, 0.5, 0.5), (1, 2, 3)]
    ph.parent_g[:] = [0, 0, 0.3, 0.1, 0.3]
    t = hip.MaterialTable(cap=8)
    pk = hip.pack_photons(ph, t)
    assert t.n == 3 and list(pk["material"]) == [0, 0, 1, 2, 1]
    got = hip.unpack_photons(pk, t)
    assert np.array_equal(got.parent_n[0], (0, 0, 0)) and np.array_equal(got.parent_wi[0], (0, 0, 0))
    assert np.array_equal(got.parent_n[1], (0, 0, -1)) and np.array_equal(got.parent_n[2], (-1, 0, 0))
    assert np.array_equal(got.parent_wi[3], (0, -1, 0))
    assert angle(got.parent_n[3:4], ph.parent_n[3:4])[0] < 7e-5
    assert np.array_equal(got.wi[4], (0, 0, 0))
    assert list(got.path_id) == [0, 1, 0, 1, 1] and np.array_equal(got.flags, ph.flags)
    assert np.array_equal(got.parent_scat, ph.parent_scat) and np.array_equal(got.parent_g, ph.parent_g)
    # the table persists across iterations and refuses to overflow
    pk2 = hip.pack_photons(ph, t)
    assert t.n == 3 and np.array_equal(pk2, pk)
    small = hip.MaterialTable(cap=2)
    with pytest.raises(hip.GvpmError):
        hip.pack_photons(ph, small)


def test_beam_sets_are_lossless_and_reject_mixed_edges():
    c = cases.make_case("cbox", 24, 20, 100, 3.0)
    rays = c.rays.copy()
    # invalid shifted rays, one of them of length 0
    rays["info"][3, 2] &= ~np.uint32(1)
    rays["info"][5, 4] &= ~np.uint32(1)
    rays["len"][5, 4] = 0.0
    pk = hip.pack_camera_beams(rays)
    assert pk.shape == (rays.shape[0], 272)
    back = hip.unpack_camera_beams(pk)
    for k in ("o", "len", "d", "pdf", "eye", "jacobian", "gop", "info"):
        assert np.array_equal(back[k], rays[k]), k
    assert np.array_equal(back["rand"][:, 0], rays["rand"][:, 0]) and np.array_equal(back["pixel"][:, 0], rays["pixel"][:, 0])
    assert not back["rand"][:, 1:].any() and not back["pixel"][:, 1:].any()
    bad = rays.copy()
    bad["info"][7, 1] = bad["info"][7, 1] + np.uint32(1 << 8)
    with pytest.raises(hip.GvpmError):
        hip.pack_camera_beams(bad)
    assert hip.pack_camera_beams(rays[:0]).shape == (0, 272)
