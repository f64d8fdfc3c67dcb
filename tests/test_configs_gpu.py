"""The five workloads of BASELINE.json `configs`, each AT ITS STATED SIZE on the device, against the fp64 oracle.

The oracle cannot run a full frame at these sizes in seconds, so every test follows one shape:
  (1) the device gathers the full frame with the full photon / beam / plane map;
  (2) the device gathers a pixel WINDOW of the same frame (same map) and the fp64 oracle gathers the same window:
      evaluation counts must agree (exactly where the hit decision is the oracle's own fp64 predicate; to the
      stated tolerance for G-Beams) and the 27 accumulators to the parity bar;
  (3) the window of the full-frame run must equal the window-only run (the estimate of a pixel does not depend
      on which other pixels are in the launch): this carries the oracle comparison over to the full frame;
  (4) size-independent properties on the full frame (0 <= weighted <= flux, the border rule, linearity in the
      flux, run-to-run reproducibility).

C1 (configs[0]) S-cbox, G-VPM 3D point kernel, 256x256, 100k photons, 40 camera samples, 4 SPPM iterations
C2 (configs[1]) is tests/test_parity_gpu.py::test_full_size_properties (512x512, 1M photons)
C3 (configs[2]) S-laser, G-Beams 3D-optimised (and the 1D kernel), 512x512, 2M beam segments
C4 (configs[3]) S-fogroom, G-BRE 3D, 1024x1024, 4M photons (one GPU holds the whole frame; the sharded run is
                tests/test_multi_rank_gpu.py)
C5 (configs[4]) S-laser with the sensor inside the medium (gvpm.cpp:784-788), g = 0 and HG g = 0.7, G-Planes 0D,
                256x256, 50k planes
"""
import numpy as np
import pytest

import cases
import oracle_lib as O
from gvpm_amd import abi, hip
from test_parity_gpu import l2, TOL

pytestmark = pytest.mark.gpu


def window_of(rays, x0, y0, w, h):
    px, py = cases.pixels_of(rays)
    return (px >= x0) & (px < x0 + w) & (py >= y0) & (py < y0 + h)


def pick_windows(acc, w):
    """Two w x w pixel blocks of a gathered frame: the brightest and the median one among the lit blocks."""
    H, W = acc.shape[:2]
    lum = acc[: H // w * w, : W // w * w, 0:3].sum(-1).reshape(H // w, w, W // w, w).sum((1, 3))
    order = np.argsort(lum.ravel())
    lit = order[lum.ravel()[order] > 0]
    out = []
    for k in (lit[-1], lit[len(lit) // 2]):
        out.append((int(k % lum.shape[1]) * w, int(k // lum.shape[1]) * w))
    return out


def check_weights(acc, border=True):
    H, W = acc.shape[:2]
    flux, wt = acc[..., 0:3], acc[..., 15:27].reshape(H, W, 4, 3)
    assert np.isfinite(acc).all()
    assert (wt >= 0).all() and (wt <= flux[:, :, None, :] * (1 + 1e-4) + 1e-12).all()
    if border:  # w = 1 for ERight at x = W-1 and ETop at y = H-1 (shift_volume_photon.cpp:843-846)
        assert np.allclose(wt[H - 1, :, abi.GVPM_TOP], flux[H - 1], rtol=1e-4, atol=1e-12)
        assert np.allclose(wt[:, W - 1, abi.GVPM_RIGHT], flux[:, W - 1], rtol=1e-4, atol=1e-12)


# --------------------------------------------------------------------------- C1: G-VPM
def run_vpm(p, m, tris, data, sel_sets=None):
    ctx = hip.Context(p, device=0)
    ctx.upload_scene(*tris)
    ctx.upload_medium(m)
    emitted = 0
    for it, (ph, nb, rays, smp) in enumerate(data, 1):
        if sel_sets is not None:
            rays, smp = sub_vpm(rays, smp, sel_sets[it - 1])
        ctx.upload_photons(ph)
        ctx.upload_camera_beams(rays)
        ctx.upload_vpm_samples(smp)
        ctx.gather(it, nb)
        emitted += nb
    acc, st = ctx.download_accum(), ctx.stats()
    sv, nv = ctx.download_vpm_state()
    ctx.close()
    return acc, st, sv, nv, emitted


def sub_vpm(rays, smp, sel):
    remap = np.full(len(rays), -1, np.int64)
    remap[np.nonzero(sel)[0]] = np.arange(int(sel.sum()))
    keep = remap[smp["set"]] >= 0
    s2 = smp[keep].copy()
    s2["set"] = remap[s2["set"]]
    return np.ascontiguousarray(rays[sel]), s2


def test_c1_vpm_256x256_100k_photons_40_samples_4_iterations():
    W = H = 256
    sc = cases.SynthScene("cbox", W, H)
    p = sc.params()
    p.vol_technique = abi.GVPM_DISTANCE
    p.nb_camera_samples = 40
    p.initial_scale_volume = 2.0
    m, tris = sc.medium(), sc.triangles()
    data = []
    for it in range(1, 5):
        ph, nb = sc.shoot_photons(it, 100000)
        rays, smp = sc.camera_beams_and_vpm_samples(it, 40)
        data.append((ph, nb, rays, smp))
    acc, st, sv, nv, emitted = run_vpm(p, m, tris, data)
    assert st["evaluations"] > 2_000_000
    x0, y0, w, h = 112, 120, 32, 32
    sels = [window_of(d[2], x0, y0, w, h) for d in data]
    wacc, wst, wsv, wnv, _ = run_vpm(p, m, tris, data, sels)
    ref = rsv = rnv = None
    total = 0
    for (ph, nb, rays, smp), sel in zip(data, sels):
        r2, s2 = sub_vpm(rays, smp, sel)
        ref, rsv, rnv, cnt, _ = O.gather_vpm(p, m, tris, ph, r2, s2, 64, use_accel=True, accum=ref, scale_vol=rsv, n_vol=rnv)
        total += cnt["evaluations"]
    win = (slice(y0, y0 + h), slice(x0, x0 + w))
    lum = ref[win][..., 0:3].mean()
    assert wst["evaluations"] == total and total > 20000
    assert l2(wacc[win], ref[win], lum) < TOL
    assert np.allclose(wsv[win], rsv[win], rtol=1e-6) and np.allclose(wnv[win], rnv[win], rtol=1e-6)
    # the window of the full-frame run is the window-only run (per-pixel SPPM state included)
    assert np.allclose(acc[win], wacc[win], rtol=2e-5, atol=1e-9 * lum)
    assert np.allclose(sv[win], wsv[win], rtol=1e-6) and np.allclose(nv[win], wnv[win], rtol=1e-6)
    check_weights(acc)
    # the radius of every pixel that met photons shrank (gvpm.cpp:1191-1195), the others kept theirs
    assert (sv <= p.initial_scale_volume * (1 + 1e-6)).all() and (sv[nv > 0] < p.initial_scale_volume).all()


# --------------------------------------------------------------------------- C3: G-Beams
def run_beams(p, m, tris, beams, en, nb, rays):
    ctx = hip.Context(p, device=0)
    ctx.upload_scene(*tris)
    ctx.upload_medium(m)
    rad = ctx.radius()
    ctx.upload_beams(beams, en)
    ctx.upload_camera_beams(rays)
    ctx.gather(1, nb)
    acc, st = ctx.download_accum(), ctx.stats()
    ctx.close()
    return acc, st, rad


@pytest.mark.parametrize("tech", [abi.GVPM_BEAM_BEAM_3D_OPTIMIZED, abi.GVPM_BEAM_BEAM_1D])
def test_c3_laser_beams_512x512_2m_segments(tech):
    W = H = 512
    sc = cases.SynthScene("laser", W, H)
    p = sc.params()
    p.vol_technique = tech
    if tech == abi.GVPM_BEAM_BEAM_1D:
        p.use_shift_null = 0
    p.initial_scale_volume = 1.0
    m, tris = sc.medium(), sc.triangles()
    beams, en, nb = sc.shoot_beams(1, 2_000_000)
    assert beams.n == 2_000_000
    rays = sc.camera_beams(1)
    acc, st, rad = run_beams(p, m, tris, beams, en, nb, rays)
    assert st["evaluations"] > 20_000_000
    # the shaft of the laser crosses the middle of the frame: a window on it and one beside it
    for (x0, y0) in ((240, 200), (60, 330)):
        w = h = 24
        sel = window_of(rays, x0, y0, w, h)
        wr = np.ascontiguousarray(rays[sel])
        wacc, wst, _ = run_beams(p, m, tris, beams, en, nb, wr)
        ref, cnt, _ = O.gather_beams(p, m, tris, beams, en, wr, rad, 1, nb, 64)
        win = (slice(y0, y0 + h), slice(x0, x0 + w))
        lum = max(ref[win][..., 0:3].mean(), 1e-30)
        # the evaluated set is the oracle's exactly (every decision of the kernel record is banded and settled in fp64
        # inside the band) -- and, round 5, so are the shift counters: the shifts' own decisions are banded too and the
        # undecided ones go to exact_beams_kernel (until round 4: 5 of 6.05 M reconnections on the other side)
        assert wst["evaluations"] == cnt["evaluations"], (wst, cnt)
        for k in ("null_shifts", "diffuse_shifts", "failed_shifts"):
            assert wst[k] == cnt[k], (k, wst, cnt)
        assert cnt["evaluations"] > 10000
        assert l2(wacc[win], ref[win], lum) < 1e-3
        assert np.allclose(acc[win], wacc[win], rtol=1e-4, atol=1e-7 * lum)
    check_weights(acc)
    # linearity in the beam flux
    b2 = beams.subset(np.arange(beams.n))
    b2.flux = b2.flux * 2
    b2.prefix_w = b2.prefix_w * 2
    acc2, st2, _ = run_beams(p, m, tris, b2, en, nb, rays)
    assert st2["evaluations"] == st["evaluations"]
    assert np.allclose(acc2, 2 * acc, rtol=1e-4, atol=1e-9)


# --------------------------------------------------------------------------- C4: G-BRE on S-fogroom
def run_bre(p, m, tris, ph, nb, rays):
    ctx = hip.Context(p, device=0)
    ctx.upload_scene(*tris)
    ctx.upload_medium(m)
    ctx.upload_photons(ph)
    ctx.upload_camera_beams(rays)
    rad = ctx.radius()
    ctx.gather(1, nb)
    acc, st = ctx.download_accum(), ctx.stats()
    ctx.close()
    return acc, st, rad


@pytest.mark.parametrize("scene", ["cbox", "cbox_rot"])
def test_c2_cbox_bre3d_512x512_1m_photons(scene):
    """BASELINE configs[1], the bench line's workload, at its stated size: the full frame on the device, two 32x32 windows
    of it against the fp64 oracle through the reference's kd-tree -> BVH walk (until round 4 only bench.py's parity leg
    held C2 at size against the oracle).  Round 5: also in general position (`cbox_rot`), shift counters EXACT."""
    W = H = 512
    sc = cases.SynthScene(scene, W, H)
    p = sc.params()
    p.initial_scale_volume = 1.0
    m, tris = sc.medium(), sc.triangles()
    ph, nb = sc.shoot_photons(1, 1_000_000)
    assert ph.n == 1_000_000
    rays = sc.camera_beams(1)
    acc, st, rad = run_bre(p, m, tris, ph, nb, rays)
    assert st["evaluations"] > 10_000_000
    for (x0, y0) in pick_windows(acc, 32):
        w = h = 32
        sel = window_of(rays, x0, y0, w, h)
        wr = np.ascontiguousarray(rays[sel])
        wacc, wst, _ = run_bre(p, m, tris, ph, nb, wr)
        ref, cnt, _ = O.gather_bre(p, m, tris, ph, wr, rad, 1, nb, 64, use_accel=True)
        win = (slice(y0, y0 + h), slice(x0, x0 + w))
        lum = ref[win][..., 0:3].mean()
        assert wst["evaluations"] == cnt["evaluations"] and cnt["evaluations"] > 20000
        for k in ("null_shifts", "diffuse_shifts", "failed_shifts"):
            assert wst[k] == cnt[k], (k, wst, cnt)   # (round 5: exact -- banded decisions + the exact pass)
        assert l2(wacc[win], ref[win], lum) < TOL
        assert np.allclose(acc[win], wacc[win], rtol=2e-5, atol=1e-9 * lum)
    check_weights(acc)


@pytest.mark.parametrize("scene", ["fogroom", "fogroom_rot"])
def test_c4_fogroom_bre3d_1024x1024_4m_photons(scene):
    W = H = 1024
    sc = cases.SynthScene(scene, W, H)
    p = sc.params()
    p.initial_scale_volume = 1.0
    m, tris = sc.medium(), sc.triangles()
    assert tris[0].shape[0] > 700  # the box + 64 inner boxes: interior occluders for the shadow rays
    ph, nb = sc.shoot_photons(1, 4_000_000)
    assert ph.n == 4_000_000
    rays = sc.camera_beams(1)
    acc, st, rad = run_bre(p, m, tris, ph, nb, rays)
    assert st["evaluations"] > 100_000_000
    assert st["diffuse_shifts"] > 10_000_000 and st["failed_shifts"] > 0
    # the brightest 32x32 block of the frame and a median one
    for (x0, y0) in pick_windows(acc, 32):
        w = h = 32
        sel = window_of(rays, x0, y0, w, h)
        wr = np.ascontiguousarray(rays[sel])
        wacc, wst, _ = run_bre(p, m, tris, ph, nb, wr)
        ref, cnt, _ = O.gather_bre(p, m, tris, ph, wr, rad, 1, nb, 64, use_accel=True)
        win = (slice(y0, y0 + h), slice(x0, x0 + w))
        lum = ref[win][..., 0:3].mean()
        assert wst["evaluations"] == cnt["evaluations"] and cnt["evaluations"] > 20000
        for k in ("null_shifts", "diffuse_shifts", "failed_shifts"):
            assert wst[k] == cnt[k], (k, wst, cnt)   # (round 5: exact -- banded decisions + the exact pass)
        assert l2(wacc[win], ref[win], lum) < TOL
        assert np.allclose(acc[win], wacc[win], rtol=2e-5, atol=1e-9 * lum)
    check_weights(acc)


# --------------------------------------------------------------------------- C5: G-Planes, sensor inside
def run_planes(p, m, tris, beams, w1, l1, nb, rays):
    ctx = hip.Context(p, device=0)
    ctx.upload_scene(*tris)
    ctx.upload_medium(m)
    ctx.upload_planes(beams, w1, l1)
    ctx.upload_camera_beams(rays)
    ctx.gather(1, nb)
    acc, st = ctx.download_accum(), ctx.stats()
    ctx.close()
    return acc, st


@pytest.mark.parametrize("scene", ["laser_in", "laser_in_hg"])
def test_c5_laser_planes0d_256x256_50k_planes(scene):
    W = H = 256
    sc = cases.SynthScene(scene, W, H)
    p = sc.params()
    p.vol_technique = abi.GVPM_VOL_PLANE0D
    p.use_shift_null = 0   # GPMConfig::load rejects useShiftNull for planes (gvpm_struct.h:310-313)
    p.min_depth = 2        # gvpm.cpp:164-175
    m, tris = sc.medium(), sc.triangles()
    assert abs(m.g - (0.7 if scene.endswith("hg") else 0.0)) < 1e-6
    beams, en, w1, l1, nb = sc.shoot_planes(1, 50000)
    assert beams.n == 50000
    rays = sc.camera_beams(1)
    assert rays.shape[0] == W * H and (rays["info"][:, 0] >> 8 & 0xFF == 1).all()  # medium edge 1: sensor inside
    acc, st = run_planes(p, m, tris, beams, w1, l1, nb, rays)
    assert st["evaluations"] > 10_000_000 and st["null_shifts"] == 0
    for (x0, y0) in ((112, 100), (20, 200)):
        w = h = 24
        sel = window_of(rays, x0, y0, w, h)
        wr = np.ascontiguousarray(rays[sel])
        wacc, wst = run_planes(p, m, tris, beams, w1, l1, nb, wr)
        ref, cnt, _ = O.gather_planes(p, m, tris, beams, w1, l1, wr, 1, nb, 64)
        win = (slice(y0, y0 + h), slice(x0, x0 + w))
        lum = max(ref[win][..., 0:3].mean(), 1e-30)
        assert wst["evaluations"] == cnt["evaluations"] and cnt["evaluations"] > 10000
        assert wst["diffuse_shifts"] == cnt["diffuse_shifts"]  # (exact, as DESIGN section 2 claims for all four techniques)
        assert l2(wacc[win], ref[win], lum) < 1e-5
        assert np.allclose(acc[win], wacc[win], rtol=2e-5, atol=1e-9 * lum)
    check_weights(acc, border=False)
