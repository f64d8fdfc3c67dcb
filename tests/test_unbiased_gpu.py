"""E[dx] = E[throughput(x + 1)] - E[throughput(x)] on the DEVICE, per technique (tests/unbiased.py has the statement; VERDICT
round 4, next 2).  S-cbox (convex), the intended visibility segment, alpha = 1 (a stationary radius), thousands of one-iteration
renders of a 32 x 24 frame with fresh photons and camera samples each -- device-generated where the technique allows, so
that the whole test is a few seconds of GPU time per configuration.  Bars: no pixel of D = dx - finite difference beyond 4.5
standard errors (2 200 tests a plane: 4.5 sigma = 7e-6 two-sided), the regression slope of mean dx on the mean finite difference
within 1 % of one, and the relative L2 of mean D below 1.5 x the noise floor its standard errors give (and below 2 %)."""
import numpy as np
import pytest

import unbiased
from gvpm_amd import abi, hip
from gvpm_amd.host import SynthScene

pytestmark = pytest.mark.gpu
W, H = 32, 24
TECH = dict(bre3d=abi.GVPM_VOL_BRE3D, bre2d=abi.GVPM_VOL_BRE2D, vpm=abi.GVPM_DISTANCE, beams3d=abi.GVPM_BEAM_BEAM_3D_OPTIMIZED,
            beams1d=abi.GVPM_BEAM_BEAM_1D)


def estimate(tech, n, nph, scale, **kw):
    sc = SynthScene("cbox", W, H)
    p = sc.params()
    p.initial_scale_volume = scale
    p.alpha = 1.0
    p.visibility_as_written = 0
    p.vol_technique = TECH[tech]
    if tech == "bre2d":
        p.use_shift_null = 0  # GPMConfig::load rejects useShiftNull for the 2D kernel (gvpm_struct.h:310-313)
    if tech == "vpm":
        p.nb_camera_samples = 8
    for k, v in kw.items():
        setattr(p, k, v)
    ctx = hip.Context(p, device=0)
    ctx.upload_scene(*sc.triangles())
    ctx.upload_medium(sc.medium())
    gen = hip.DeviceGenerator(sc) if tech != "vpm" else None

    def step(k):
        it = k + 1
        ctx.reset()
        if tech == "vpm":
            # (the camera samples of G-VPM are the host's)
            ph, nb = sc.shoot_photons(it, nph)
            rays, smp = sc.camera_beams_and_vpm_samples(it, p.nb_camera_samples)
            ctx.upload_photons(ph)
            ctx.upload_camera_beams(rays)
            ctx.upload_vpm_samples(smp)
        else:
            rptr, nsets = gen.camera_beams(it)
            if tech.startswith("beams"):
                soa, en, nb = gen.shoot_beams(it, nph)
                ctx.upload_beams_dev(soa, en)
            else:
                soa, nb = gen.shoot_photons(it, nph)
                ctx.upload_photons_dev(soa)
            ctx.upload_camera_beams_dev(rptr, nsets)
        ctx.gather(1, nb)
        return ctx.download_film(1, False)

    out = unbiased.run(step, n)
    st = ctx.stats()
    ctx.close()
    if gen:
        gen.close()
    return out, st


def check(out, st, what, slope_tol=0.01, kernel3d=True):
    for key in ("dx", "dy"):
        o = out[key]
        print(f"{what} {key}: slope {o['slope']:.4f}  rel L2 {o['rel_l2']:.4f} (noise floor {o['noise_l2']:.4f})  "
              f"max |z| {o['zmax']:.2f} over {o['n_tests']} pixels x channels, {o['n_over4']} beyond 4 sigma; "
              f"|gradient| / throughput {o['grad_over_thr']:.2f}")
        assert o["n_tests"] > 1500
        assert abs(o["slope"] - 1.0) < (slope_tol if kernel3d else 0.03), (what, key, o["slope"])
        if kernel3d:
            assert o["zmax"] < 4.5, (what, key, o["zmax"])
            assert o["rel_l2"] < max(1.5 * o["noise_l2"], 1e-3), (what, key, o["rel_l2"], o["noise_l2"])
        else:
            # The 2D / 1D kernels are biased AT BEAM ENDS in the reference's own statement -- the fp64 oracle through this
            # very test (scripts/probes_py/unbiased_oracle.py, 1500 iterations): the column of pixels beside the frame's first beams
            # |z| 3 - 5.7, the pixels under the light 4 - 6, everything else below 4 -- the 2D kernel accepts photons beyond
            # the beam's end (the empty far check, shift_volume_photon.cpp:726-731) where the neighbour's beam is shorter or
            # absent.  Here: the same pixels, and nothing else.
            # the same pixels (and their neighbours), a few per cent of the frame -- not a frame-wide effect:
            assert o["n_over4"] <= 0.1 * o["n_tests"] and o["rel_l2"] < 0.05, (what, key, o["n_over4"], o["rel_l2"])
            z = o["z"]
            inner = z[2:-3, 3:-3]
            assert (inner > 4.5).sum() <= 0.01 * inner.size, (what, key, np.argwhere(z > 4.5)[:20])


@pytest.mark.parametrize("kw", [dict(), dict(use_mis=0, path_set=0), dict(use_shift_null=0), dict(power_heuristic=1)])
def test_bre3d_gradient_is_the_finite_difference_of_the_throughput(kw):
    out, st = estimate("bre3d", 3000, 60000, 3.0, **kw)
    if kw.get("use_shift_null", 1):
        assert st["null_shifts"] > 5000   # the MIXED shift: null shifts and reconnections both in play
    assert st["diffuse_shifts"] > 100000
    check(out, st, f"G-BRE 3D {kw}")


def test_bre2d():
    out, st = estimate("bre2d", 3000, 60000, 3.0)
    check(out, st, "G-BRE 2D", kernel3d=False)


def test_vpm():
    out, st = estimate("vpm", 1000, 40000, 6.0)
    check(out, st, "G-VPM", slope_tol=0.03)


@pytest.mark.parametrize("tech,kw", [("beams3d", dict()), ("beams3d", dict(use_mis=0, path_set=0)), ("beams1d", dict())])
def test_beams(tech, kw):
    out, st = estimate(tech, 3000, 20000, 2.0, **kw)
    check(out, st, f"G-Beams {tech} {kw}", kernel3d=tech == "beams3d")
