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


# ---- linked photon records (round 6) -------------------------------------------------------------------------------------

def linked_decode_numpy(blob, table):
    """an independent decode of a blob of gvpm_pack_photons_linked, from the header's description of the three kinds"""
    hd = hip.linked_header(blob)
    n = hd["n"]
    kw = np.frombuffer(blob[hd["off_kinds"]:hd["off_kinds"] + 4 * ((n + 15) // 16)].tobytes(), np.uint32)
    kind = (kw[np.arange(n) // 16] >> (2 * (np.arange(n) % 16)).astype(np.uint32)) & 3
    F = np.frombuffer(blob[hd["off_full"]:hd["off_full"] + 76 * hd["n_full"]].tobytes(), abi.PHOTON_PACKED_DTYPE)
    E = np.frombuffer(blob[hd["off_emit"]:hd["off_emit"] + 48 * hd["n_emit"]].tobytes(), abi.PHOTON_EMIT_DTYPE)
    Cn = np.frombuffer(blob[hd["off_chain"]:hd["off_chain"] + 40 * hd["n_chain"]].tobytes(), abi.PHOTON_CHAIN_DTYPE)
    em = np.frombuffer(blob[hd["off_emitters"]:hd["off_emitters"] + 32 * hd["n_emitters"]].tobytes(), abi.EMITTER_ENTRY_DTYPE)
    groups = np.frombuffer(blob[hd["off_groups"]:hd["off_groups"] + 8 * ((n + 63) // 64)].tobytes(), np.uint32).reshape(-1, 2)
    isF, isE, isC = kind == 0, kind == 1, kind == 2
    assert isF.sum() == hd["n_full"] and isE.sum() == hd["n_emit"] and isC.sum() == hd["n_chain"]
    # the per-64-photon bases are the running counts
    assert np.array_equal(groups[:, 0], np.concatenate([[0], np.cumsum(isF)])[0:n:64])
    assert np.array_equal(groups[:, 1], np.concatenate([[0], np.cumsum(isE)])[0:n:64])
    out = abi.Photons(n)
    full = hip.unpack_photons(F.copy(), table)
    for k in abi.PHOTON_VEC3 + abi.PHOTON_F1 + abi.PHOTON_U1:
        getattr(out, k)[isF] = getattr(full, k)
    DIFF = abi.GVPM_BSDF_DIFFUSE_REFLECTION if hasattr(abi, "GVPM_BSDF_DIFFUSE_REFLECTION") else 0x2
    # emit records
    ei = E["flags"] >> 16
    out.pos[isE], out.parent_pos[isE], out.flux[isE] = E["pos"], E["parent_pos"], E["flux"]
    out.parent_pdf[isE], out.edge_pdf[isE] = E["parent_pdf"], E["edge_pdf"]
    out.prefix_w[isE], out.parent_rr[isE], out.parent_n[isE], out.parent_g[isE] = em["prefix_w"][ei], em["parent_rr"][ei], em["parent_n"][ei], em["parent_g"][ei]
    out.parent_scat[isE] = 0
    out.parent_wi[isE] = np.array([1, 0, 0], np.float32)
    out.flags[isE] = (E["flags"] & 0xFF7F) | (DIFF << 16)
    out.path_id[isE] = (E["flags"] >> 7) & 1
    # chain records: own fields, then the links
    mi = Cn["flags"] >> 16
    out.pos[isC], out.flux[isC] = Cn["pos"], Cn["flux"]
    out.parent_pdf[isC], out.edge_pdf[isC], out.parent_rr[isC] = Cn["parent_pdf"], Cn["edge_pdf"], Cn["parent_rr"]
    out.parent_scat[isC], out.parent_g[isC] = table.table["scat"][mi], table.table["g"][mi]
    out.parent_n[isC] = 0
    out.flags[isC] = (Cn["flags"] & 0xFF7F) | (DIFF << 16)
    out.path_id[isC] = (Cn["flags"] >> 7) & 1
    ic = np.nonzero(isC)[0]
    out.parent_pos[ic] = out.pos[ic - 1]
    out.prefix_w[ic] = out.flux[ic - 1]

    def derive(a, b):  # normalize(b - a) in float64, rounded once
        d = b.astype(np.float64) - a.astype(np.float64)
        ln = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])
        return (d / ln[:, None]).astype(np.float32)
    prev_parent = np.where(isC[ic - 1][:, None], out.pos[ic - 2], out.parent_pos[ic - 1])
    out.parent_wi[ic] = derive(out.pos[ic - 1], prev_parent)
    for sel in (isE, isC):
        out.wi[sel] = derive(out.pos[sel], out.parent_pos[sel])
    return out, kind


@pytest.mark.parametrize("scene", ["cbox", "cbox_hg", "cbox_mirror", "cbox_phong", "cbox_rot", "fogroom"])
def test_linked_records_decode_as_the_header_says(scene):
    c = cases.make_case(scene, 24, 20, 30000, 3.0)
    t = hip.MaterialTable()
    blob = hip.pack_photons_linked(c.ph, t)
    hd = hip.linked_header(blob)
    assert hd["n"] == c.ph.n and hd["bytes"] == blob.size and hd["n_emitters"] >= 1
    # a photon map in a participating medium: most parents are the emitter or the previous photon
    assert blob.size < 56 * c.ph.n and hd["n_chain"] > 0.2 * c.ph.n and hd["n_emit"] > 0.4 * c.ph.n
    got = hip.unpack_photons_linked(blob, t)
    want, kind = linked_decode_numpy(blob, t)
    for k in abi.PHOTON_VEC3 + abi.PHOTON_F1 + abi.PHOTON_U1:
        assert np.array_equal(getattr(got, k), getattr(want, k)), k
    # what the format carries as it is (a chain record's parent_pos / prefix_w ARE the previous photon's pos / flux)
    for k in ("pos", "parent_pos", "flux", "prefix_w", "parent_pdf", "edge_pdf", "parent_rr", "parent_scat", "parent_g", "flags"):
        assert np.array_equal(getattr(got, k), getattr(c.ph, k)), k
    assert np.array_equal(got.path_id, c.ph.path_id & 1)
    # the short kinds carry parent_n exactly (the full records: octahedral); parent_wi within the octahedral code's error
    short = kind != 0
    assert np.array_equal(got.parent_n[short], c.ph.parent_n[short])
    assert angle(got.parent_wi, c.ph.parent_wi).max() < 7e-5
    # ... and as a whole the blob is no worse than the packed records of the same photons
    t2 = hip.MaterialTable()
    pk = hip.unpack_photons(hip.pack_photons(c.ph, t2), t2)
    assert np.array_equal(got.wi, pk.wi)
    assert angle(got.parent_wi, c.ph.parent_wi).max() <= angle(pk.parent_wi, c.ph.parent_wi).max() + 1e-12


def test_linked_records_fall_back_to_full_ones_and_reject_bad_blobs():
    c = cases.make_case("cbox", 16, 12, 3000, 3.0)
    ph = c.ph
    # photons shuffled: no photon follows its parent any more -- nothing may chain, the blob still decodes to the same photons
    perm = np.random.default_rng(5).permutation(ph.n)
    sh = ph.subset(perm)
    t = hip.MaterialTable()
    blob = hip.pack_photons_linked(sh, t)
    hd = hip.linked_header(blob)
    got = hip.unpack_photons_linked(blob, t)
    for k in ("pos", "parent_pos", "flux", "prefix_w", "parent_pdf", "edge_pdf", "parent_rr", "flags"):
        assert np.array_equal(getattr(got, k), getattr(sh, k)), k
    chained = np.nonzero(np.all(sh.parent_pos[1:] == sh.pos[:-1], 1))[0]
    assert hd["n_chain"] <= chained.size
    # empty map
    e = hip.pack_photons_linked(ph.subset(np.zeros(0, np.int64)), hip.MaterialTable())
    assert hip.linked_header(e)["n"] == 0 and hip.unpack_photons_linked(e, hip.MaterialTable()).n == 0
    # a truncated blob, a wrong magic, a material index beyond the table
    with pytest.raises(hip.GvpmError):
        hip.unpack_photons_linked(blob[:-16].copy(), t)
    bad = blob.copy()
    bad[0] ^= 1
    with pytest.raises(hip.GvpmError):
        hip.unpack_photons_linked(bad, t)
    with pytest.raises(hip.GvpmError):
        hip.unpack_photons_linked(blob, hip.MaterialTable())
