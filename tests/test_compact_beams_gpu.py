"""Compact camera-beam sets on the device (gvpm_upload_camera_beams_compact, uploads.hip): the device rebuilds the five rays
of a sensor-adjacent set exactly as gvpm_unpack_camera_beams_compact does, so a compact upload IS an SoA upload of the
unpacked rays -- same counters -- and the oracle fed with the unpacked rays must be matched to the usual bars; against the
producer's ORIGINAL fp32 rays (whose origins differ in the last bit, see the header) the film stays far inside the bar."""
import numpy as np
import pytest

import cases
import oracle_lib as O
from gvpm_amd import abi, hip

pytestmark = pytest.mark.gpu
COUNTERS = ("evaluations", "null_shifts", "diffuse_shifts", "failed_shifts")


def gather(c, rays=None, compact=None, full=None, sensor=None, pinned=None, iters=1):
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    if sensor is not None:
        ctx.upload_sensor(sensor)
    for it in range(1, iters + 1):
        if pinned is not None:
            if it == 1:
                ctx.upload_materials(pinned[1])
                ctx.upload_pinned_packed(pinned[0])
            if it < iters:
                ctx.prefetch_packed(pinned[0])
        else:
            ctx.upload_photons(c.ph)
            if compact is not None:
                ctx.upload_camera_beams_compact(compact, full)
            else:
                ctx.upload_camera_beams(rays)
        ctx.gather(it, c.nb)
    acc = ctx.download_accum().astype(np.float64)
    st = ctx.stats()
    ctx.close()
    return acc, st


@pytest.mark.parametrize("scene,nph", [("cbox", 30000), ("laser_in", 20000), ("cbox_mirror", 30000)])
def test_compact_upload_is_the_soa_upload_of_the_unpacked_rays(scene, nph):
    c = cases.make_case(scene, 48, 40, nph, 3.0)
    sensor = c.sc.sensor()
    comp, full, idx = hip.pack_camera_beams_compact(sensor, c.rays, c.sc.jitter(c.it, c.rays))
    assert comp.size > 1000 and (full.shape[0] > 0) == (scene == "cbox_mirror")
    unp = np.concatenate([hip.unpack_camera_beams_compact(sensor, comp), hip.unpack_camera_beams(full)])
    a_c, s_c = gather(c, compact=comp, full=full, sensor=sensor)
    a_u, s_u = gather(c, rays=unp)
    a_o, s_o = gather(c, rays=c.rays)
    assert s_c["evaluations"] > 10000
    for k in COUNTERS:
        assert s_c[k] == s_u[k], (k, s_c, s_u)
    assert np.abs(a_c - a_u).max() <= 2e-5 * np.abs(a_u).max()  # (the order of the atomics)
    # the oracle on what the device decoded
    ref, cnt, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, unp, c.r, c.it, c.nb, 64)
    lum = max(ref[..., 0:3].mean(), 1e-30)
    for k in COUNTERS:
        assert s_c[k] == cnt[k], (k, s_c, cnt)
    assert np.sqrt(((a_c - ref) ** 2).mean()) / lum < 1e-4
    # against the producer's own rays: the same estimate (origins differ by rounding: a pair on the rim of a kernel may
    # flip, with a contribution that vanishes there), sensorMIS = 1 instead of 1 +- ulps
    assert abs(s_c["evaluations"] - s_o["evaluations"]) <= max(2, 1e-5 * s_o["evaluations"])
    for k in ("null_shifts", "diffuse_shifts", "failed_shifts"):
        assert abs(s_c[k] - s_o[k]) <= max(4, 2e-5 * s_o[k])
    assert np.sqrt(((a_c - a_o) ** 2).mean()) / lum < 2e-5
    ref_o = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, c.it, c.nb, 64)[0]
    assert np.sqrt(((a_c - ref_o) ** 2).mean()) / lum < 1e-4


def test_prefetched_compact_sets_from_pinned_memory():
    c = cases.make_case("cbox", 40, 32, 20000, 3.0)
    sensor = c.sc.sensor()
    jit = c.sc.jitter(c.it, c.rays)
    comp, full, _ = hip.pack_camera_beams_compact(sensor, c.rays, jit)
    t = hip.MaterialTable()
    pk = hip.PinnedPacked(c.ph, c.rays, t, sensor=sensor, jitter=jit)
    assert pk.ncompact == comp.size and pk.nfull == 0 and pk.nbytes == c.ph.n * 76 + comp.size * 60
    a1, s1 = gather(c, compact=comp, full=full, sensor=sensor, iters=3)
    unp = hip.unpack_photons(hip.pack_photons(c.ph, hip.MaterialTable()), t)
    a2, s2 = gather(c, pinned=(pk, t), sensor=sensor, iters=3)
    pk.close()
    # (the pinned path also packs the photons: wi is re-derived, so the shift counters may move by the packed format's
    # own tolerance; the evaluated set is the same)
    assert s1["evaluations"] == s2["evaluations"]
    for k in ("null_shifts", "diffuse_shifts", "failed_shifts"):
        assert abs(s1[k] - s2[k]) <= max(2, 1e-5 * s1[k])
    assert np.abs(a1 - a2).max() <= 1e-4 * np.abs(a1).max()


def test_missing_sensor_and_missing_material_table_are_errors():
    c = cases.make_case("cbox", 16, 12, 500, 3.0)
    sensor = c.sc.sensor()
    comp, full, _ = hip.pack_camera_beams_compact(sensor, c.rays, c.sc.jitter(c.it, c.rays))
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    ctx.upload_photons(c.ph)
    with pytest.raises(hip.GvpmError) as e:
        ctx.upload_camera_beams_compact(comp, full)
    assert e.value.code == abi.GVPM_ERR_STATE
    bad = abi.Sensor()
    with pytest.raises(hip.GvpmError):
        ctx.upload_sensor(bad)  # no film, no rotation
    ctx.upload_sensor(sensor)
    ctx.upload_camera_beams_compact(comp, full)
    # packed photons without a material table: the gather refuses instead of decoding black parents
    t = hip.MaterialTable()
    ctx.upload_photons_packed(hip.pack_photons(c.ph, t))
    with pytest.raises(hip.GvpmError) as e:
        ctx.gather(1, c.nb)
    assert e.value.code == abi.GVPM_ERR_STATE
    ctx.upload_materials(t)
    ctx.gather(1, c.nb)
    assert ctx.stats()["evaluations"] > 0
    # a record that names a material beyond the table is reported by gvpm_get_stats
    pk = hip.pack_photons(c.ph, t)
    pk["material"][3] = 77
    ctx.upload_photons_packed(pk)
    ctx.gather(2, c.nb)
    with pytest.raises(hip.GvpmError) as e:
        ctx.stats()
    assert e.value.code == abi.GVPM_ERR_STATE
    ctx.close()


def test_g_vpm_samples_follow_the_new_set_order():
    """G-VPM samples name beam sets by their index in the upload: with compact + full sets that is new_index"""
    from test_oracle_vpm import make_vpm_case
    c = make_vpm_case("cbox_mirror", 32, 28, 40000, 5.0, nb=10)
    sensor = c.sc.sensor()
    comp, full, idx = hip.pack_camera_beams_compact(sensor, c.rays, c.sc.jitter(c.it if hasattr(c, "it") else 1, c.rays))
    assert comp.size > 0 and full.shape[0] > 0
    smp = c.samples.copy()
    smp["set"] = idx[c.samples["set"]]
    unp = np.concatenate([hip.unpack_camera_beams_compact(sensor, comp), hip.unpack_camera_beams(full)])
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    ctx.upload_sensor(sensor)
    ctx.upload_photons(c.ph)
    ctx.upload_camera_beams_compact(comp, full)
    ctx.upload_vpm_samples(smp)
    ctx.gather(1, c.nb)
    acc = ctx.download_accum().astype(np.float64)
    st = ctx.stats()
    ctx.close()
    ref, sv, nv, cnt, _ = O.gather_vpm(c.p, c.m, c.tris, c.ph, unp, smp, 64, use_accel=False)
    assert st["evaluations"] == cnt["evaluations"] > 5000
    lum = max(ref[..., 0:3].mean(), 1e-30)
    assert np.sqrt(((acc - ref) ** 2).mean()) / lum < 1e-4
