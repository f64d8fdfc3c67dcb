"""G-Beams parity: HIP path vs the fp64 oracle (computeVolumeGradientBeams)."""
import numpy as np
import pytest

import cases
import oracle_lib as O
from gvpm_amd import abi, hip
from test_oracle_beams import make_beam_case, TECHS
from test_parity_gpu import l2, TOL

pytestmark = pytest.mark.gpu
SHIFT_COUNTERS = ("null_shifts", "diffuse_shifts", "failed_shifts")


def device_beams(c, p=None, rays=None, iters=1, exact=True):
    p = c.p if p is None else p
    ctx = hip.Context(p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    cases.upload_bsdfs(ctx, c)
    ref = None
    total = {k: 0 for k in SHIFT_COUNTERS + ("evaluations",)}
    for it in range(1, iters + 1):
        if it == 1:
            beams, en, nb, r = c.beams, c.end_n, c.nb, (c.rays if rays is None else rays)
        else:
            beams, en, nb = c.sc.shoot_beams(it, c.beams.n)
            r = c.sc.camera_beams(it)
        rad = ctx.radius()
        ctx.upload_beams(beams, en)
        ctx.upload_camera_beams(r)
        ctx.gather(it, nb)
        ref, cnt, _ = O.gather_beams(p, c.m, c.tris, beams, en, r, rad, it, nb, 64, accum=ref)
        for k in total:
            total[k] += cnt[k]
    acc = ctx.download_accum()
    st = ctx.stats()
    film = ctx.download_film(iters, True)
    ctx.close()
    lum = max(ref[..., 0:3].mean(), 1e-30)
    # "count = device atomic, must equal the oracle's count exactly" -- on the default (fp32 local-frame)
    # path too: every validity decision of the kernel record is banded and settled in fp64 inside the band
    # (gather_beams.hip beamBase / beamKernelExact).  Round 5: the shifts' own decisions (null shift or reconnection, the
    # visibility of a reconnection) are banded as well and the undecided shifts evaluated in fp64 behind the kernel
    # (exact_beams_kernel): the shift counters are exact too (exact=False: the +-2 of rounds 1-4, for callers that feed
    # the device something the oracle does not see bit for bit).
    assert st["evaluations"] == total["evaluations"], (st, total)
    for k in SHIFT_COUNTERS:
        assert abs(st[k] - total[k]) <= (0 if exact else 2), (k, st, total)
    assert l2(acc, ref, lum) < 2e-4   # (measured 6e-6 .. 6e-5; SURVEY's bar is 1e-3)
    rfilm = O.assemble(ref, iters, True)
    for a, b in zip(film, rfilm):
        assert l2(a, b, lum) < 2e-4
    return acc, ref, st


@pytest.mark.parametrize("tech", TECHS)
@pytest.mark.parametrize("scene", ["cbox", "cbox_hg"])
def test_beams_match_fp64_oracle(tech, scene):
    c = make_beam_case(scene, 32, 28, 12000, 2.5, technique=tech)
    acc, ref, st = device_beams(c)
    assert st["evaluations"] > 20000


@pytest.mark.parametrize("tech", TECHS)
def test_beams_two_iterations(tech):
    c = make_beam_case("cbox", 24, 20, 6000, 3.0, technique=tech)
    device_beams(c, iters=2)


@pytest.mark.parametrize("kw", [dict(use_mis=0), dict(use_shift_null=0), dict(power_heuristic=1), dict(max_depth=3),
                                dict(path_set=0), dict(debug_shift=abi.GVPM_SHIFT_NULL),
                                dict(debug_shift=abi.GVPM_SHIFT_MANIFOLD),
                                dict(lighting_interaction_mode=abi.GVPM_MEDIA2MEDIA)])
def test_beams_flag_sweep(kw):
    c = make_beam_case("cbox", 24, 20, 6000, 3.0)
    p = c.p.copy()
    for k, v in kw.items():
        setattr(p, k, v)
    device_beams(c, p=p)


def test_beams_fine_image_null_shifts_and_empty():
    # pixel spacing below the kernel radius: the null-shift branch (shiftNull3D) is exercised
    c = make_beam_case("cbox", 96, 96, 4000, 4.0)
    c.rays = c.rays[(cases.pixels_of(c.rays)[0] < 40) & (cases.pixels_of(c.rays)[1] < 60)]
    acc, ref, st = device_beams(c)
    assert st["null_shifts"] > 1000
    c2 = make_beam_case("cbox", 16, 12, 500, 3.0)
    c2.beams = c2.beams.subset(np.zeros(0, np.int64))
    c2.end_n = c2.end_n[:0]
    acc, ref, st = device_beams(c2)
    assert st["evaluations"] == 0 and not acc.any()


def test_beams_item_list_regrows(monkeypatch):
    # an item list far too small for the planner's output (heavy items are split into parts, so its size has no
    # a-priori bound): the planner counts what it could not write, the traversal never reads past the capacity, the host
    # regrows the list to the count and repeats plan + traversal
    monkeypatch.setenv("GVPM_BEAM_ITEMS_INIT", "7")
    c = make_beam_case("cbox", 32, 28, 12000, 2.5)
    acc, ref, st = device_beams(c, iters=2)
    assert st["evaluations"] > 20000


def test_beams_pair_list_regrows(monkeypatch):
    # a pair list far too small for the first pass: the traversal only counts, the host regrows the list to the count
    # and repeats the pass (gather_drivers.hip gatherBeams); blocks are reserved eight at a time, so the count includes the
    # empty blocks waves had left over
    monkeypatch.setenv("GVPM_BEAM_PAIRS_INIT", "1024")
    c = make_beam_case("cbox", 32, 28, 12000, 2.5)
    acc, ref, st = device_beams(c, iters=2)
    assert st["evaluations"] > 20000


@pytest.mark.parametrize("tech", TECHS)
@pytest.mark.parametrize("scene", ["cbox", "cbox_hg"])
def test_literal_fp64_path_has_the_oracles_hit_set_exactly(tech, scene, monkeypatch):
    """What separates "float intermediates" from "an ownership bug" in the looser bar of the tests above: the device's
    literal transcription (GVPM_BEAMS_FP64=1: the reference's statements with its float intermediates, the same
    traversal, sub-beam ownership, pair lists and tile order as the fast path) must reproduce the oracle's counters
    EXACTLY -- evaluations, null shifts, reconnections, failures -- and its sums to fp32 accumulation noise.  The fast
    path differs from it only by its fp32 local-frame arithmetic (ownership decided in fp64 inside the error band)."""
    c = make_beam_case(scene, 32, 28, 12000, 2.5, technique=tech)
    monkeypatch.setenv("GVPM_BEAMS_FP64", "1")
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    rad = ctx.radius()
    ctx.upload_beams(c.beams, c.end_n)
    ctx.upload_camera_beams(c.rays)
    ctx.gather(1, c.nb)
    st, acc = ctx.stats(), ctx.download_accum()
    ctx.close()
    ref, cnt, _ = O.gather_beams(c.p, c.m, c.tris, c.beams, c.end_n, c.rays, rad, 1, c.nb, 64)
    for k in ("evaluations", "null_shifts", "diffuse_shifts", "failed_shifts"):
        assert st[k] == cnt[k], (k, st, cnt)
    assert cnt["evaluations"] > 20000
    assert l2(acc, ref, max(ref[..., 0:3].mean(), 1e-30)) < 2e-5


@pytest.mark.parametrize("levels", [0, 1, 2, 3])
def test_beam_near_list_formats(levels):
    """The per-beam near-occluder lists by scene size (device_types.h: BeamNearFmt): 14 occluders -> 19 x 5 bits, 56 ->
    15 x 6 bits, 224 -> 12 x 8 bits read from global memory (more than the 128 LDS holds), 896 -> every list overflowed,
    visibility through the occluder BVH.  The same surface in every case: the oracle, which tests every triangle of the
    untessellated scene, must be met at the usual bar, and the shift counters must not move with the tessellation."""
    c = make_beam_case("cbox", 24, 20, 6000, 3.0)
    fine = cases.tessellate(c.tris, levels) if levels else c.tris
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*fine)
    ctx.upload_medium(c.m)
    rad = ctx.radius()
    ctx.upload_beams(c.beams, c.end_n)
    ctx.upload_camera_beams(c.rays)
    ctx.gather(1, c.nb)
    st, acc = ctx.stats(), ctx.download_accum()
    ctx.close()
    ref, cnt, _ = O.gather_beams(c.p, c.m, c.tris, c.beams, c.end_n, c.rays, rad, 1, c.nb, 64)
    assert st["evaluations"] == cnt["evaluations"], (st, cnt)
    for k in ("diffuse_shifts", "failed_shifts"):
        assert abs(st[k] - cnt[k]) <= 2, (k, st, cnt)
    assert cnt["failed_shifts"] > 0
    assert l2(acc, ref, max(ref[..., 0:3].mean(), 1e-30)) < 1e-3


@pytest.mark.parametrize("tech", TECHS)
@pytest.mark.parametrize("scene,scale", [("fogroom", 2.5), ("laser", 2.0), ("laser_hg", 2.0)])
def test_reconnections_among_occluders(tech, scene, scale):
    """Scenes whose new beams ARE blocked (64 boxes standing in the fog; the aperture plate of S-laser): the evaluation
    skips the any-hit loop for a reconnection inside its beam's free cone (grid_build.hip, beam_near_kernel) and takes
    the others through it -- a cone that certified a blocked beam would turn failed shifts into reconnections."""
    c = make_beam_case(scene, 32, 28, 12000, scale, technique=tech)
    acc, ref, st = device_beams(c)
    assert st["evaluations"] > 2000 and st["failed_shifts"] > (200 if scene == "fogroom" else 20)


@pytest.mark.parametrize("tilt_deg", [1.5, 0.4])
def test_a_thin_plate_grazed_by_the_beams_blocks_their_reconnections(tilt_deg, monkeypatch):
    """S-laser's shaft runs down the y axis.  A large thin plate is stood IN the shaft, a degree or so off the beams'
    direction: the beams' lines pierce it at grazing incidence (|cos| <= 0.05), far from all of its edges.  The free cone
    of such a beam must not certify reconnections (beamClearTri: the pierce test runs at any incidence) -- every counter
    equals the oracle's, which tests every new beam against every triangle, and the gather with the cone switched off."""
    c = make_beam_case("laser", 32, 28, 12000, 2.0)
    v0, e1, e2 = (np.array(a, np.float32) for a in c.tris)
    t = np.tan(np.radians(tilt_deg))
    # the plane x = 0.004 - t * y, for y in [-0.95, 0.85], z in [-0.6, 0.6]: one quad, two triangles
    a = np.array([0.004 + t * 0.95, -0.95, -0.6]); b = np.array([0.004 - t * 0.85, 0.85, -0.6])
    cc = np.array([0.004 - t * 0.85, 0.85, 0.6]); d = np.array([0.004 + t * 0.95, -0.95, 0.6])
    v0 = np.vstack([v0, a, a]).astype(np.float32)
    e1 = np.vstack([e1, b - a, cc - a]).astype(np.float32)
    e2 = np.vstack([e2, cc - a, d - a]).astype(np.float32)
    c.tris = (v0, e1, e2)
    acc1, ref, st1 = device_beams(c)
    assert st1["failed_shifts"] > 200 and st1["diffuse_shifts"] > 200
    monkeypatch.setenv("GVPM_BEAMS_FREE_CONE", "0")
    acc0, _, st0 = device_beams(c)
    assert st0 == st1


def test_free_cone_off_equals_free_cone_on(monkeypatch):
    # the same gather with every reconnection sent through the any-hit loop (GVPM_BEAMS_FREE_CONE=0): same counters,
    # same sums to float-atomic ordering
    c = make_beam_case("fogroom", 32, 28, 12000, 2.5)
    acc1, ref, st1 = device_beams(c)
    monkeypatch.setenv("GVPM_BEAMS_FREE_CONE", "0")
    acc0, _, st0 = device_beams(c)
    assert st0 == st1
    assert np.allclose(acc0, acc1, rtol=1e-5, atol=1e-7 * max(ref[..., 0:3].mean(), 1e-30))


@pytest.mark.parametrize("scene,kw", [("laser", dict()), ("cbox", dict(use_shift_null=0)), ("fogroom", dict(path_set=0))])
def test_split_evaluation_equals_the_fused_one(monkeypatch, scene, kw):
    """GVPM_BEAMS_SPLIT=1 (round 4, opt-in): phase 1 and phase 2 of the evaluation as two kernels with the reconnection
    entries in HBM between them -- the same parity bars, the same counters as the fused kernel"""
    c = make_beam_case(scene, 40, 32, 9000, 3.0, **kw)
    acc0, ref, st0 = device_beams(c)
    monkeypatch.setenv("GVPM_BEAMS_SPLIT", "1")
    acc1, _, st1 = device_beams(c)
    for k in ("evaluations", "null_shifts", "diffuse_shifts", "failed_shifts"):
        assert st1[k] == st0[k], (k, st1, st0)
    assert np.abs(acc1.astype(np.float64) - acc0).max() <= 2e-5 * np.abs(acc0).max()


def test_the_1d_shift_with_its_kernel_at_the_beams_origin(monkeypatch):
    """Found by tests/stress_beams.py (round 5): in S-cbox rotated one pair has its 1D kernel 1e-4 from the beam's origin, where
    shift()'s sine sqrt(1 - (u / ly)^2) (shift_volume_beams.cpp:47-79) is all rounding in fp32 -- the flip of getShiftPos1D
    (:81-91) came out wrong and a reconnection the reference makes was counted as failed.  The sine's error is part of the
    flip's band now (beams_eval_f32.h shiftSinErr): counters exact (device_beams asserts them)."""
    monkeypatch.setenv("GVPM_BEAMS_FREE_CONE", "1")
    c = make_beam_case("cbox_rot", 40, 32, 9000, 3.0, technique=abi.GVPM_BEAM_BEAM_1D)
    device_beams(c)


def test_the_1d_intersection_is_uncontracted(monkeypatch):
    """Found by tests/stress_beams.py on iteration-3 inputs (round 5): rayIntersectInternal1D (pm/beams_struct.h:250-311) rounds
    its double dot products to float and divides by d1.d2; the device's transcription was compiled with FMA contraction, a
    last-bit difference of a double moved a float rounding and 1 / d1.d2 made it a different v -- a pair 2e-5 from the beam's
    origin was evaluated that the oracle rejects.  dotU / crossU: the oracle's operations to the bit."""
    monkeypatch.setenv("GVPM_BEAMS_FREE_CONE", "1")
    c = make_beam_case("cbox_rot", 40, 32, 9000, 3.0, technique=abi.GVPM_BEAM_BEAM_1D, it=3, path_set=0)
    device_beams(c)


def test_the_1d_kernels_float_division_is_correctly_rounded(monkeypatch):
    """Found by tests/stress_beams.py on iteration-5 inputs (round 5): u = |ad| / sqrt(sin^2) is a FLOAT division in the reference
    (pm/beams_struct.h:250-311) and the library is built with fast fp32 division; one ulp of u moved shift()'s sine by 8 % on a
    pair whose kernel sits 7e-5 from the beam's origin and the exact pass flipped where the oracle does not."""
    monkeypatch.setenv("GVPM_BEAMS_FREE_CONE", "1")
    c = make_beam_case("cbox_hg_rot", 40, 32, 9000, 3.0, technique=abi.GVPM_BEAM_BEAM_1D, it=5, path_set=0)
    device_beams(c)
