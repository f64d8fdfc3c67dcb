"""The compact camera-beam sets (include/gvpm_hip.h "compact camera-beam sets"; gvpm_amd/csrc/pack_codec.h): host-side
pack / unpack (plain C, no GPU).  What a compact set MEANS is defined by gvpm_unpack_camera_beams_compact; here that
definition is pinned by an independent numpy statement of the sensor model, the losses of the format are bounded, and
the claim the format rests on -- sensorMIS of a sensor-adjacent edge is 1 -- is checked on the producers' own records."""
import numpy as np
import pytest

import cases
from gvpm_amd import abi, hip

OFFS = np.array([(0, 0), (-1, 0), (1, 0), (0, 1), (0, -1)], np.float64)  # base L R T B (shift_utilities.h:255-261)


def sensor_rays_numpy(sensor, compact):
    """independent statement: the five rays of every compact set from the sensor (float64, rounded once)"""
    n = compact.size
    px = (compact["pixel"] & 0xFFFF).astype(np.float64)
    py = (compact["pixel"] >> 16).astype(np.float64)
    M = np.array(list(sensor.to_world), np.float64).reshape(3, 3)
    pos = np.array(list(sensor.pos), np.float64)
    o = np.zeros((n, 5, 3), np.float32)
    d = np.zeros((n, 5, 3), np.float32)
    for k in range(5):
        sx = px + OFFS[k, 0] + compact["jitter"][:, 0].astype(np.float64)
        sy = py + OFFS[k, 1] + compact["jitter"][:, 1].astype(np.float64)
        cx = (2.0 * sx / sensor.width - 1.0) * sensor.tan_half_fov_x
        cy = (2.0 * sy / sensor.height - 1.0) * sensor.tan_half_fov_y
        cz = -np.ones(n)
        ln = np.sqrt(cx * cx + cy * cy + cz * cz)
        u = np.stack([cx / ln, cy / ln, cz / ln], 1)
        dd = np.stack([M[r, 0] * u[:, 0] + M[r, 1] * u[:, 1] + M[r, 2] * u[:, 2] for r in range(3)], 1)
        t0 = compact["t0"][:, k].astype(np.float64)
        o[:, k] = (pos[None, :] + dd * t0[:, None]).astype(np.float32)
        d[:, k] = dd.astype(np.float32)
    return o, d


@pytest.mark.parametrize("scene", ["cbox", "laser", "laser_in", "fogroom", "cbox_rot", "laser_in_rot"])
def test_first_edges_travel_compact_and_decode_as_the_header_says(scene):
    c = cases.make_case(scene, 40, 32, 100, 3.0)
    sensor = c.sc.sensor()
    jit = c.sc.jitter(c.it, c.rays)
    comp, full, idx = hip.pack_camera_beams_compact(sensor, c.rays, jit)
    n = c.rays.shape[0]
    assert comp.dtype.itemsize == 60 and comp.size == n > 500 and full.shape[0] == 0
    assert np.array_equal(idx, np.arange(n))
    back = hip.unpack_camera_beams_compact(sensor, comp)
    valid = (c.rays["info"] & 1) != 0
    # carried as they are
    assert np.array_equal(back["info"], c.rays["info"])
    assert np.array_equal(back["len"][valid], c.rays["len"][valid])
    assert np.array_equal(back["rand"], c.rays["rand"]) and np.array_equal(back["pixel"], c.rays["pixel"])
    assert not back["len"][~valid].any() and not back["o"][~valid].any() and not back["d"][~valid].any()
    # rebuilt from the sensor: the independent statement, bit for bit
    o, d = sensor_rays_numpy(sensor, comp)
    assert np.array_equal(back["o"][valid], o[valid]) and np.array_equal(back["d"][valid], d[valid])
    # against the producer's own fp32 rays: the direction is the sensor's, exactly (the synthetic host rounds the same
    # float64 direction); the origin moves by the rounding of t0 along the ray (sensor inside the medium: t0 = 0, exact)
    assert np.array_equal(back["d"][valid], c.rays["d"][valid])
    do = np.abs(back["o"][valid].astype(np.float64) - c.rays["o"][valid].astype(np.float64))
    if scene == "laser_in":
        assert do.max() == 0.0 and not comp["t0"].any()
    else:
        assert do.max() <= 2.5e-7
    # what the format drops: every eye weight is 1, and sensorMIS = pdf_s / pdf_b * jacobian_s is 1 (gvpm_struct.h:608-631
    # with shift_cameraPath.h:76-116,191-242) on the producer's records; the decode writes pdf = jacobian = gop = 1
    assert np.all(c.rays["eye"][valid] == 1.0)
    sm = c.rays["pdf"][:, 1:].astype(np.float64) / c.rays["pdf"][:, :1].astype(np.float64) * c.rays["jacobian"][:, 1:]
    assert np.abs(sm[valid[:, 1:]] - 1.0).max() < 1e-6
    for k in ("pdf", "jacobian", "gop"):
        assert np.all(back[k][valid] == 1.0) and not back[k][~valid].any()
    assert np.all(back["eye"][valid] == 1.0)


def test_deeper_edges_keep_their_full_records():
    """behind the mirror the eye weight, the origin and the direction are the path's: those sets stay 272-byte records"""
    c = cases.make_case("cbox_mirror", 48, 40, 100, 3.0)
    sensor = c.sc.sensor()
    jit = c.sc.jitter(c.it, c.rays)
    comp, full, idx = hip.pack_camera_beams_compact(sensor, c.rays, jit)
    n = c.rays.shape[0]
    edge = (c.rays["info"][:, 0] >> 8) & 0xFF
    first = edge == edge.min()
    assert 0 < (~first).sum() == full.shape[0] and comp.size == first.sum() and comp.size + full.shape[0] == n
    # the upload order: compact sets first, in input order, then the full ones
    order = np.concatenate([np.flatnonzero(first), np.flatnonzero(~first)])
    assert np.array_equal(np.argsort(idx, kind="stable"), order) and sorted(idx) == list(range(n))
    back_full = hip.unpack_camera_beams(full)
    for k in ("o", "len", "d", "pdf", "eye", "jacobian", "gop", "info"):
        assert np.array_equal(back_full[k], c.rays[~first][k]), k
    back = hip.unpack_camera_beams_compact(sensor, comp)
    assert np.array_equal(back["pixel"][:, 0], c.rays[first]["pixel"][:, 0])
    assert np.array_equal(back["d"], c.rays[first]["d"])


def test_sets_the_format_cannot_carry_fall_back_and_bad_input_is_refused():
    c = cases.make_case("cbox", 24, 20, 100, 3.0)
    sensor = c.sc.sensor()
    jit = c.sc.jitter(c.it, c.rays)
    rays = c.rays.copy()
    rays["eye"][3, 2] = (0.5, 0.5, 0.5)          # an eye weight that is not 1
    rays["jacobian"][5, 4] *= 1.01               # a sensorMIS that is not 1
    rays["d"][7, 0] = (0.0, 0.6, -0.8)           # a ray that does not come from the sensor
    rays["info"][9, 1] &= ~np.uint32(1)          # an invalid shifted ray: still compact
    rays["len"][9, 1] = 0
    comp, full, idx = hip.pack_camera_beams_compact(sensor, rays, jit)
    n = rays.shape[0]
    assert full.shape[0] == 3 and comp.size == n - 3
    assert sorted(idx[[3, 5, 7]]) == [n - 3, n - 2, n - 1]
    back = hip.unpack_camera_beams_compact(sensor, comp)
    assert (back["info"][idx[9], 1] & 1) == 0 and (back["info"][idx[9], 2] & 1) == 1
    # a wrong jitter does not reproduce the rays: everything falls back, nothing is silently bent
    comp2, full2, _ = hip.pack_camera_beams_compact(sensor, c.rays, (jit + 0.25) % 1.0)
    assert comp2.size == 0 and full2.shape[0] == n
    bad = c.rays.copy()
    bad["info"][2, 3] += np.uint32(1 << 8)       # a shifted ray on another edge than its base
    with pytest.raises(hip.GvpmError):
        hip.pack_camera_beams_compact(sensor, bad, jit)
    c0, f0, i0 = hip.pack_camera_beams_compact(sensor, c.rays[:0], jit[:0])
    assert c0.size == 0 and f0.shape[0] == 0 and i0.size == 0


def test_a_rotated_sensor_round_trips():
    """to_world is a general rotation: rays generated by the numpy statement for a rotated sensor pack as compact sets"""
    sensor = abi.Sensor()
    a, b = 0.3, -0.7
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
    Rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
    M = Rz @ Rx
    for i in range(9):
        sensor.to_world[i] = float(M.reshape(-1)[i])
    sensor.pos[0], sensor.pos[1], sensor.pos[2] = 0.5, -2.0, 1.25
    sensor.tan_half_fov_x, sensor.tan_half_fov_y = 0.4, 0.3
    sensor.width, sensor.height = 64, 48
    rng = np.random.default_rng(5)
    n = 200
    comp = np.zeros(n, abi.BEAM_SET_COMPACT_DTYPE)
    comp["pixel"] = rng.integers(1, 47, n).astype(np.uint32) << 16 | rng.integers(1, 63, n).astype(np.uint32)
    comp["jitter"] = (rng.integers(0, 1 << 24, (n, 2)) / float(1 << 24)).astype(np.float32)
    comp["rand"] = rng.random(n).astype(np.float32)
    comp["info"] = 0x1F | (1 << 8)
    comp["len"] = rng.random((n, 5)).astype(np.float32) + 0.5
    rays = hip.unpack_camera_beams_compact(sensor, comp)
    o, d = sensor_rays_numpy(sensor, comp)
    assert np.array_equal(rays["o"], o) and np.array_equal(rays["d"], d)
    assert np.abs(np.linalg.norm(rays["d"].astype(np.float64), axis=2) - 1).max() < 1e-7
    c2, f2, _ = hip.pack_camera_beams_compact(sensor, rays, comp["jitter"])
    assert f2.shape[0] == 0 and np.array_equal(c2, comp)
