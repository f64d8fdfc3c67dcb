"""Packed uploads on the device (gvpm_upload_*_packed, uploads.hip): the device decodes a record exactly as
gvpm_unpack_photons does, so a packed upload is an SoA upload of the unpacked arrays -- same counters, and the oracle fed
with the unpacked inputs must be matched to the usual bars; against the ORIGINAL inputs the evaluation count is unchanged
(positions travel as fp32) and the film moves far less than the parity bar."""
import numpy as np
import pytest

import cases
import oracle_lib as O
from gvpm_amd import abi, hip

pytestmark = pytest.mark.gpu
COUNTERS = ("evaluations", "null_shifts", "diffuse_shifts", "failed_shifts")


def run(c, mode, table=None, pk=None, rp=None, ph=None, iters=1):
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    pinned = None
    for it in range(1, iters + 1):
        if mode == "soa":
            ctx.upload_photons(c.ph if ph is None else ph)
            ctx.upload_camera_beams(c.rays)
        elif mode == "packed":
            ctx.upload_materials(table)
            ctx.upload_photons_packed(pk)
            ctx.upload_camera_beams_packed(rp)
        else:  # pinned + prefetch
            if it == 1:
                ctx.upload_materials(table)
                pinned = hip.PinnedPacked(c.ph, c.rays, table)
                ctx.upload_pinned_packed(pinned)
            if it < iters:
                ctx.prefetch_packed(pinned)
        ctx.gather(it, c.nb)
    acc = ctx.download_accum().astype(np.float64)
    st = ctx.stats()
    film = ctx.download_film(iters, True)
    ctx.close()
    if pinned is not None:
        pinned.close()
    return acc, st, film


@pytest.mark.parametrize("scene", ["cbox", "cbox_hg", "laser"])
def test_packed_upload_is_the_soa_upload_of_the_unpacked_arrays(scene):
    c = cases.make_case(scene, 48, 40, 30000, 3.0)
    t = hip.MaterialTable()
    pk = hip.pack_photons(c.ph, t)
    rp = hip.pack_camera_beams(c.rays)
    unp = hip.unpack_photons(pk, t)
    a_p, s_p, f_p = run(c, "packed", t, pk, rp)
    a_u, s_u, f_u = run(c, "soa", ph=unp)
    a_o, s_o, f_o = run(c, "soa")
    assert s_p["evaluations"] > 10000
    for k in COUNTERS:
        assert s_p[k] == s_u[k], (k, s_p, s_u)
    # (the sums of the two runs differ by the order of their atomics only)
    assert np.abs(a_p - a_u).max() <= 2e-5 * np.abs(a_u).max()
    # the oracle on what the device decoded
    ref, cnt, _ = O.gather_bre(c.p, c.m, c.tris, unp, c.rays, c.r, c.it, c.nb, 64)
    lum = max(ref[..., 0:3].mean(), 1e-30)
    for k in COUNTERS:
        assert s_p[k] == cnt[k], (k, s_p, cnt)
    assert np.sqrt(((a_p - ref) ** 2).mean()) / lum < 1e-4
    # against the original inputs: the same pairs are evaluated; the film is within the parity bar with room
    assert s_p["evaluations"] == s_o["evaluations"]
    for k in ("null_shifts", "diffuse_shifts", "failed_shifts"):
        assert abs(s_p[k] - s_o[k]) <= max(2, 1e-5 * s_o[k])
    assert np.sqrt(((a_p - a_o) ** 2).mean()) / lum < 2e-5
    rfilm = O.assemble(O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, c.it, c.nb, 64)[0], c.it, True)
    for g, r in zip(f_p, rfilm):
        assert np.sqrt(((g.astype(np.float64) - r) ** 2).mean()) / lum < 1e-4


def test_prefetched_packed_records_from_pinned_memory():
    c = cases.make_case("cbox", 40, 32, 20000, 3.0)
    t = hip.MaterialTable()
    pk = hip.pack_photons(c.ph, t)
    rp = hip.pack_camera_beams(c.rays)
    a1, s1, _ = run(c, "packed", t, pk, rp, iters=3)
    a2, s2, _ = run(c, "pinned", t, iters=3)
    for k in COUNTERS:
        assert s1[k] == s2[k]
    assert np.abs(a1 - a2).max() <= 2e-5 * np.abs(a1).max()


def test_empty_and_unknown_material():
    c = cases.make_case("cbox", 16, 12, 500, 3.0)
    t = hip.MaterialTable()
    pk = hip.pack_photons(c.ph, t)
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    ctx.upload_materials(t)
    ctx.upload_photons_packed(pk[:0])
    ctx.upload_camera_beams_packed(hip.pack_camera_beams(c.rays))
    ctx.gather(1, c.nb)
    assert ctx.stats()["evaluations"] == 0
    with pytest.raises(hip.GvpmError):
        lib = hip.lib()
        ctx._check(lib.gvpm_upload_photons_packed(ctx._h, None, 5))
    ctx.close()


def test_packed_records_feed_g_vpm_too():
    """the packed photon / beam-set uploads fill the same staging slots as the SoA ones: G-VPM on packed records evaluates
    what the oracle evaluates on the unpacked arrays"""
    from test_oracle_vpm import make_vpm_case
    c = make_vpm_case("cbox_hg", 32, 28, 40000, 5.0, nb=10)
    t = hip.MaterialTable()
    pk = hip.pack_photons(c.ph, t)
    unp = hip.unpack_photons(pk, t)
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    ctx.upload_materials(t)
    ctx.upload_photons_packed(pk)
    ctx.upload_camera_beams_packed(hip.pack_camera_beams(c.rays))
    ctx.upload_vpm_samples(c.samples)
    ctx.gather(1, c.nb)
    acc = ctx.download_accum().astype(np.float64)
    st = ctx.stats()
    ctx.close()
    ref, sv, nv, cnt, _ = O.gather_vpm(c.p, c.m, c.tris, unp, c.rays, c.samples, 64, use_accel=False)
    assert st["evaluations"] == cnt["evaluations"] > 5000
    lum = max(ref[..., 0:3].mean(), 1e-30)
    assert np.sqrt(((acc - ref) ** 2).mean()) / lum < 1e-4


# ---- linked photon records (round 6): 40 / 48 / 76 bytes a photon ---------------------------------------------------------

@pytest.mark.parametrize("scene", ["cbox", "cbox_hg", "cbox_mirror_rot", "cbox_phong"])
def test_linked_upload_is_the_soa_upload_of_the_unpacked_arrays(scene):
    c = cases.make_case(scene, 48, 40, 30000, 3.0)
    t = hip.MaterialTable()
    blob = hip.pack_photons_linked(c.ph, t)
    assert blob.size < 56 * c.ph.n
    unp = hip.unpack_photons_linked(blob, t)

    def run_linked():
        ctx = hip.Context(c.p, device=0)
        ctx.upload_scene(*c.tris)
        ctx.upload_medium(c.m)
        cases.upload_bsdfs(ctx, c)
        ctx.upload_materials(t)
        ctx.upload_photons_linked(blob)
        ctx.upload_camera_beams(c.rays)
        ctx.gather(c.it, c.nb)
        acc, st = ctx.download_accum().astype(np.float64), ctx.stats()
        ctx.close()
        return acc, st

    def run_soa(ph):
        ctx = hip.Context(c.p, device=0)
        ctx.upload_scene(*c.tris)
        ctx.upload_medium(c.m)
        cases.upload_bsdfs(ctx, c)
        ctx.upload_photons(ph)
        ctx.upload_camera_beams(c.rays)
        ctx.gather(c.it, c.nb)
        acc, st = ctx.download_accum().astype(np.float64), ctx.stats()
        ctx.close()
        return acc, st

    a_l, s_l = run_linked()
    a_u, s_u = run_soa(unp)
    a_o, s_o = run_soa(c.ph)
    assert s_l["evaluations"] > 10000
    # the device decodes a blob exactly as gvpm_unpack_photons_linked does: same counters, same sums up to atomics' order
    for k in COUNTERS:
        assert s_l[k] == s_u[k], (k, s_l, s_u)
    assert np.abs(a_l - a_u).max() <= 2e-5 * np.abs(a_u).max()
    # against the original inputs: the same pairs; the film moves far less than the parity bar
    lum = max(a_o[..., 0:3].mean(), 1e-30)
    assert s_l["evaluations"] == s_o["evaluations"]
    for k in ("null_shifts", "diffuse_shifts", "failed_shifts"):
        assert abs(s_l[k] - s_o[k]) <= max(2, 1e-5 * s_o[k])
    assert np.sqrt(((a_l - a_o) ** 2).mean()) / lum < 3e-5


def test_prefetched_linked_blobs_from_pinned_memory():
    c = cases.make_case("cbox", 40, 32, 20000, 3.0)
    t = hip.MaterialTable()
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    pinned = [hip.PinnedPacked(*c.sc.shoot_photons(it, 20000)[:1], c.sc.camera_beams(it), t, linked=True) for it in (1, 2, 3)]
    assert all(p.linked and p.photon_bytes < 56 * p.n for p in pinned)
    ctx.upload_materials(t)
    ctx.upload_pinned_packed(pinned[0])
    ref, total = None, 0
    for it in (1, 2, 3):
        if it < 3:
            ctx.prefetch_packed(pinned[it])
        ph, nb = c.sc.shoot_photons(it, 20000)
        r = ctx.radius()
        ctx.gather(it, nb)
        ref, cnt, _ = O.gather_bre(c.p, c.m, c.tris, ph, c.sc.camera_beams(it), r, it, nb, 64, use_accel=False, accum=ref)
        total += cnt["evaluations"]
    acc = ctx.download_accum().astype(np.float64)
    st = ctx.stats()
    ctx.close()
    for p in pinned:
        p.close()
    assert st["evaluations"] == total
    assert np.sqrt(((acc - ref) ** 2).mean()) / max(ref[..., 0:3].mean(), 1e-30) < 1e-4
