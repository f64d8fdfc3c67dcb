"""E[dx] = E[throughput(x + 1)] - E[throughput(x)] through the ORACLE (tests/unbiased.py; the device's version with the statistics
that make it sharp is tests/test_unbiased_gpu.py): a few hundred one-iteration renders of a small frame.  At this size the
test sees a sign error, a factor of two or a missing weight -- slope and z-scores -- not a per-cent effect."""
import numpy as np
import pytest

import cases
import oracle_lib as O
import unbiased
from gvpm_amd import abi
from gvpm_amd.host import SynthScene


def oracle_estimate(n, nph, W=16, H=12, **kw):
    sc = SynthScene("cbox", W, H)
    p = sc.params()
    p.initial_scale_volume = 4.0
    p.alpha = 1.0
    p.visibility_as_written = 0
    for k, v in kw.items():
        setattr(p, k, v)
    m, tris = sc.medium(), sc.triangles()
    r = cases.radius_of(p)

    def step(k):
        ph, nb = sc.shoot_photons(k + 1, nph)
        ref, cnt, _ = O.gather_bre(p, m, tris, ph, sc.camera_beams(k + 1), r, 1, nb, 64, use_accel=True)
        return O.assemble(ref, 1, False)

    return unbiased.run(step, n)


@pytest.mark.parametrize("kw", [dict(), dict(use_mis=0, path_set=0)])
def test_oracle_bre3d_gradient_is_the_finite_difference_of_the_throughput(kw):
    out = oracle_estimate(250, 6000, **kw)
    for key in ("dx", "dy"):
        o = out[key]
        assert o["n_tests"] > 300 and o["grad_over_thr"] > 0.2
        assert abs(o["slope"] - 1.0) < 0.05, (key, o["slope"])
        assert o["zmax"] < 4.5, (key, o["zmax"])
        assert o["rel_l2"] < max(1.6 * o["noise_l2"], 1e-3), (key, o["rel_l2"], o["noise_l2"])


def test_the_statistic_sees_a_missing_weight():
    """the test's own power: the same renders with the reverse shift's term dropped from the assembly (what a pair of weights
    that does not sum to one amounts to) must fail by a wide margin"""
    sc = SynthScene("cbox", 16, 12)
    p = sc.params()
    p.initial_scale_volume = 4.0
    p.alpha = 1.0
    p.visibility_as_written = 0
    m, tris = sc.medium(), sc.triangles()
    r = cases.radius_of(p)

    def step(k):
        ph, nb = sc.shoot_photons(k + 1, 6000)
        ref, cnt, _ = O.gather_bre(p, m, tris, ph, sc.camera_beams(k + 1), r, 1, nb, 64, use_accel=True)
        thr, dx, dy = O.assemble(ref, 1, False)
        a = ref.reshape(ref.shape[0], ref.shape[1], 9, 3)
        dx_half = a[:, :, 1 + abi.GVPM_RIGHT] - a[:, :, 5 + abi.GVPM_RIGHT]   # the forward shift alone
        return thr, dx_half, dy

    out = unbiased.run(step, 120)
    assert abs(out["dx"]["slope"] - 1.0) > 0.2 or out["dx"]["zmax"] > 6
