"""G-VPM parity: HIP path vs the fp64 oracle (computeVolumeGradientPhoton)."""
import numpy as np
import pytest

import cases
import oracle_lib as O
from gvpm_amd import abi, hip
from test_oracle_vpm import make_vpm_case
from test_parity_gpu import l2, TOL

pytestmark = pytest.mark.gpu


def device_vpm(c, iters=1, p=None, rays=None, exact=True):
    p = c.p if p is None else p
    ctx = hip.Context(p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    cases.upload_bsdfs(ctx, c)
    ref = sv = nv = None
    total = 0
    shifts = {k: 0 for k in ("null_shifts", "diffuse_shifts", "failed_shifts")}
    emitted = 0
    for it in range(1, iters + 1):
        if it == 1:
            ph, nb, r, smp = c.ph, c.nb, (c.rays if rays is None else rays), c.samples
        else:
            ph, nb = c.sc.shoot_photons(it, c.ph.n)
            r, smp = c.sc.camera_beams_and_vpm_samples(it, p.nb_camera_samples)
        ctx.upload_photons(ph)
        ctx.upload_camera_beams(r)
        ctx.upload_vpm_samples(smp)
        ctx.gather(it, nb)
        emitted += nb
        ref, sv, nv, cnt, _ = O.gather_vpm(p, c.m, c.tris, ph, r, smp, 64, use_accel=False, accum=ref, scale_vol=sv,
                                           n_vol=nv)
        total += cnt["evaluations"]
        for k in shifts:
            shifts[k] += cnt[k]
    acc = ctx.download_accum()
    st = ctx.stats()
    dsv, dnv = ctx.download_vpm_state()
    film = ctx.download_film(iters, True)
    ctx.close()
    lum = max(ref[..., 0:3].mean(), 1e-30)
    assert st["evaluations"] == total
    for k in shifts:
        assert abs(st[k] - shifts[k]) <= (0 if exact else max(2, 2e-6 * 4 * total)), (k, st, shifts)
    assert l2(acc, ref, lum) < TOL
    assert np.allclose(dsv, sv, rtol=1e-6) and np.allclose(dnv, nv, rtol=1e-6)
    rfilm = O.assemble(ref, iters, True, total_emitted=emitted)
    for a, b in zip(film, rfilm):
        assert l2(a, b, lum / emitted) < TOL
    return acc, ref, st


@pytest.mark.parametrize("scene", ["cbox", "cbox_hg"])
def test_vpm_matches_fp64_oracle(scene):
    c = make_vpm_case(scene, 32, 28, 40000, 5.0, nb=10)
    acc, ref, st = device_vpm(c)
    assert st["evaluations"] > 5000


def test_vpm_thin_medium():
    """sigma_t d ~ 1e-3: the distance pdf's normalisation 1 - exp(-sigma_t d) must not be formed as a difference in fp32
    (1e-4 relative error at sigma_t d = 1e-3, an infinite pdf below 6e-8)"""
    c = make_vpm_case("cbox", 32, 28, 40000, 5.0, nb=10)
    for k in range(3):
        c.m.sigma_s[k] = 2.5e-4
        c.m.sigma_a[k] = 2.5e-4
        c.m.sigma_t[k] = 5e-4
    acc, ref, st = device_vpm(c)
    assert st["evaluations"] > 5000


def test_vpm_three_iterations_sppm_state():
    c = make_vpm_case("cbox", 24, 20, 30000, 5.0, nb=8)
    device_vpm(c, iters=3)


@pytest.mark.parametrize("kw", [dict(use_mis=0), dict(use_shift_null=0), dict(power_heuristic=1), dict(max_depth=3),
                                dict(debug_shift=abi.GVPM_SHIFT_NULL), dict(visibility_as_written=0),
                                dict(lighting_interaction_mode=abi.GVPM_MEDIA2MEDIA)])
def test_vpm_flag_sweep(kw):
    c = make_vpm_case("cbox", 24, 20, 30000, 5.0, nb=8)
    p = c.p.copy()
    for k, v in kw.items():
        setattr(p, k, v)
    device_vpm(c, p=p)


def test_vpm_paper_radius_and_empty_inputs():
    # initialScaleVolume 0.15 (paper setting): almost every query is empty
    c = make_vpm_case("cbox", 32, 32, 50000, 0.15, nb=40)
    device_vpm(c)
    c2 = make_vpm_case("cbox", 16, 16, 5000, 5.0, nb=4)
    c2.samples = c2.samples[:0]
    device_vpm(c2)
    ctx = hip.Context(c2.p, device=0)
    ctx.upload_scene(*c2.tris); ctx.upload_medium(c2.m); ctx.upload_photons(c2.ph); ctx.upload_camera_beams(c2.rays)
    with pytest.raises(hip.GvpmError) as e:
        ctx.gather(1, 10)   # samples not uploaded
    assert e.value.code == abi.GVPM_ERR_STATE
    ctx.close()


def test_vpm_sample_orderings():
    # The kernel combines the sums of the samples of one pixel inside a wave (runs of consecutive lanes) before the
    # global atomics: one sample per pixel (runs of length 1), a shuffled order (a pixel's samples in many runs and
    # waves) and samples naming a beam set that does not exist (skipped) must all give the per-sample sums.
    c = make_vpm_case("cbox", 24, 20, 30000, 5.0, nb=1)
    device_vpm(c)
    c = make_vpm_case("cbox", 24, 20, 30000, 5.0, nb=6)
    rng = np.random.default_rng(7)
    c.samples = c.samples[rng.permutation(len(c.samples))]
    device_vpm(c)
    c = make_vpm_case("cbox", 24, 20, 30000, 5.0, nb=6)
    smp = c.samples.copy()
    bad = smp[::5].copy()
    bad["set"] = len(c.rays) + 3
    mixed = np.concatenate([smp[:100], bad, smp[100:]])
    ctx_ref = device_vpm(c)                      # all valid samples
    c.samples = mixed
    p = c.p
    ctx = hip.Context(p, device=0)
    ctx.upload_scene(*c.tris); ctx.upload_medium(c.m); ctx.upload_photons(c.ph); ctx.upload_camera_beams(c.rays)
    ctx.upload_vpm_samples(mixed)
    ctx.gather(1, c.nb)
    acc = ctx.download_accum()
    st = ctx.stats()
    ctx.close()
    lum = max(ctx_ref[1][..., 0:3].mean(), 1e-30)
    assert st["evaluations"] == ctx_ref[2]["evaluations"]
    assert l2(acc, ctx_ref[1], lum) < TOL


def test_heaviest_first_batch_order_changes_nothing(monkeypatch):
    """Round 4: the waves take their 64-sample batches in the order of the LAST launch's candidate counts (gatherVPM,
    re-sorted every fourth launch; above 1024 batches).  Sample sets of different sizes from one iteration to the next --
    more, fewer, and a regrown buffer -- must give what sample order gives: the oracle's evaluations, the same sums."""
    c = make_vpm_case("cbox", 64, 48, 30000, 4.0, nb=36)
    assert c.samples.shape[0] > 1024 * 64
    n = c.samples.shape[0]

    def run():
        ctx = hip.Context(c.p, device=0)
        ctx.upload_scene(*c.tris)
        ctx.upload_medium(c.m)
        ctx.upload_photons(c.ph)
        ctx.upload_camera_beams(c.rays)
        out = []
        for k, m in enumerate([n * 2 // 3, n * 2 // 3 + 64 * 7 + 5, n // 2, n, n - 1000, n, n, n, n]):
            ctx.upload_vpm_samples(c.samples[:m])
            ctx.gather(k + 1, c.nb)
            out.append(ctx.stats()["evaluations"])
        acc = ctx.download_accum().astype(np.float64)
        st = ctx.stats()
        ctx.close()
        return acc, st, out

    a1, s1, e1 = run()
    monkeypatch.setenv("GVPM_VPM_ORDER", "0")
    a0, s0, e0 = run()
    assert e1 == e0 and s1["evaluations"] > 50000
    for k in ("evaluations", "null_shifts", "diffuse_shifts", "failed_shifts"):
        assert s1[k] == s0[k], (k, s1, s0)
    assert np.abs(a1 - a0).max() <= 2e-5 * np.abs(a0).max()  # (the order of the atomics)


def test_a_reconnection_shorter_than_the_as_written_visibility_interval():
    """Found by tests/stress_vpm.py (round 5): a medium parent 1e-4 from a wall of S-cbox rotated, its reconnection 0.056 long --
    the as-written visibility test runs over [Epsilon, 0.056 ShadowEpsilon] = [1e-4, 5.6e-5], an EMPTY interval: no t satisfies
    the reference's mint <= t <= maxt (skdtree.h:318-320).  The three-state test phrased the interval as "the ends lie on
    different sides of the plane", which is symmetric in the two, and called the wall between them a certain hit."""
    c = make_vpm_case("cbox_rot", 36, 30, 30000, 5.0, nb=4, use_shift_null=0)
    device_vpm(c, iters=2)


def test_a_null_shift_test_at_equality_takes_the_double_radius():
    """Found by tests/stress_vpm.py (round 5): |y|^2 = 0.019781068 against r^2 = 0.0197810698 -- inside the band, so the exact pass
    decides; it had the fast kernel's fp32 radius (6e-8 off the double product R * 0.01 * scaleVol, gvpm.cpp:1082,1132) and
    decided the other way.  The note carries the pixel's scale now and the pass forms the product in double."""
    c = make_vpm_case("cbox_mirror_rot", 36, 30, 30000, 5.0, nb=10)
    device_vpm(c)


def test_a_photon_at_the_radius_takes_the_double_radius():
    """Found by tests/stress_vpm.py on iteration-3 inputs (round 5): the pair test pointDistSquared < distSquared (kdtree.h:722) is
    settled in fp64 inside its band -- against the fp32 radius squared, 6e-8 off the reference's double product: one pair in
    3 10^7 went the other way (`evaluations` off by one).  The in-band test forms R * 0.01 * scaleVol in double now."""
    c = make_vpm_case("cbox_conductor_rot", 36, 30, 30000, 5.0, nb=4, it=3)
    device_vpm(c, iters=2)


@pytest.mark.parametrize("knob", ["GVPM_VPM_POOL=0", "GVPM_VPM_SPLIT=0", "GVPM_VPM_PIPELINE=0", "GVPM_PIPELINE=0"])
def test_vpm_kernel_paths_agree(knob, monkeypatch):
    """Round 6: the gather is three kernels (the walk, the evaluation over the pool's chunks, the fused code for batches that
    find the pool exhausted).  With a pool of four chunks per shard nearly every batch takes the fallback; with the split off
    the fused kernel does everything; with the pipeline off the grid is built behind the previous gather, on its stream, instead
    of beside it into the other build set: counters equal the default path's -- and all of them the oracle's (device_vpm
    asserts) -- and the sums agree to the atomics' order.  Four iterations: both build sets are used twice, and the cell size
    comes from the host's bound on the largest scale, one or two iterations stale."""
    c = make_vpm_case("cbox_hg", 32, 28, 40000, 5.0, nb=10)
    acc0, _, st0 = device_vpm(c, iters=4)
    name, val = knob.split("=")
    monkeypatch.setenv(name, val)
    acc1, ref, st1 = device_vpm(c, iters=4)
    for k in ("evaluations", "candidates", "null_shifts", "diffuse_shifts", "failed_shifts"):
        assert st0[k] == st1[k], (k, st0[k], st1[k])
    lum = max(ref[..., 0:3].mean(), 1e-30)
    assert l2(acc1, acc0, lum) < 1e-6


def test_vpm_reset_and_same_photons_again():
    """The pipelined step keeps state on the host (the bound on the largest scale, the build set in use): a handle that is
    reset in the middle of a run, and one that gathers the SAME photons twice (no rebuild, no rotation), must give what a
    fresh handle gives."""
    c = make_vpm_case("cbox", 24, 20, 30000, 5.0, nb=8)
    p = c.p

    def run(ctx, iters):
        for it in range(1, iters + 1):
            if it == 1:
                ph, nb, r, smp = c.ph, c.nb, c.rays, c.samples
            else:
                ph, nb = c.sc.shoot_photons(it, c.ph.n)
                r, smp = c.sc.camera_beams_and_vpm_samples(it, p.nb_camera_samples)
            ctx.upload_photons(ph)
            ctx.upload_camera_beams(r)
            ctx.upload_vpm_samples(smp)
            ctx.gather(it, nb)
        return ctx.download_accum(), ctx.stats()

    ctx = hip.Context(p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    acc_a, st_a = run(ctx, 3)
    ctx.reset()
    acc_b, st_b = run(ctx, 3)
    for k in ("evaluations", "candidates", "null_shifts", "diffuse_shifts", "failed_shifts"):
        assert st_a[k] == st_b[k], (k, st_a[k], st_b[k])
    lum = max(acc_a[..., 0:3].mean(), 1e-30)
    assert l2(acc_b, acc_a, lum) < 1e-6
    # the same photons and samples again, twice: iterations 4 and 5 of this handle against iteration 4 twice on another
    ph, nb = c.sc.shoot_photons(4, c.ph.n)
    r, smp = c.sc.camera_beams_and_vpm_samples(4, p.nb_camera_samples)
    ctx.upload_photons(ph)
    ctx.upload_camera_beams(r)
    ctx.upload_vpm_samples(smp)
    ctx.gather(4, nb)
    s4 = ctx.stats()
    ctx.gather(5, nb)  # (nothing uploaded in between: the grid is kept unless the bound on the radius has moved)
    s5 = ctx.stats()
    ctx.close()
    ref = sv = nv = None
    total = 0
    for it, (pp, rr, ss) in enumerate([(c.ph, c.rays, c.samples)] + [(c.sc.shoot_photons(i, c.ph.n)[0],) + c.sc.camera_beams_and_vpm_samples(i, p.nb_camera_samples) for i in (2, 3)] + [(ph, r, smp), (ph, r, smp)], 1):
        ref, sv, nv, cnt, _ = O.gather_vpm(p, c.m, c.tris, pp, rr, ss, 64, use_accel=False, accum=ref, scale_vol=sv, n_vol=nv)
        total += cnt["evaluations"]
        if it == 4:
            assert s4["evaluations"] == total
    assert s5["evaluations"] == total
