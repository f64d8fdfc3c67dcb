"""What a DOUBLE-built host would see through the shim.

The boundary rounds every record to fp32, and every other parity test hands the oracle and the device the SAME rounded
inputs -- the rounding of a double host's positions is never measured there.  Here the photons and the camera beams exist in
double first (the generated fp32 records, each float displaced by a uniform fraction of half its ulp and the directions
re-normalised in double: records a double-built Mitsuba would hold), the reference statement runs on the UNROUNDED doubles
(tests/indep_statements.py: numpy, fp64, written from the reference sources) and the device on their fp32 rounding, as the
shim uploads them.  Bars: per-pixel L2 of the 27 accumulators over the mean luminance < 1e-3, evaluation
count within 1e-4 of the pairs (a pair on the kernel's rim may fall either side of a rounded position) -- measured: a few
1e-7 and 0-2 pairs."""
import copy
import types

import numpy as np
import pytest

import cases
import indep_statements as I
from gvpm_amd import abi, hip

pytestmark = pytest.mark.gpu


def _displace(a32, rng):
    """fp32 array -> doubles that ROUND to it: every value moved by U(-0.45, 0.45) of its ulp"""
    a = a32.astype(np.float64)
    ulp = np.spacing(np.abs(a32)).astype(np.float64)
    return a + rng.uniform(-0.45, 0.45, a.shape) * ulp


def _unit(v):
    return v / np.sqrt((v * v).sum(-1, keepdims=True))


def double_case(c, seed):
    """(case with float64 records, case with their fp32 rounding)"""
    rng = np.random.default_rng(seed)
    phD, ph32 = types.SimpleNamespace(n=c.ph.n), abi.Photons(c.ph.n)
    for k in abi.PHOTON_VEC3 + abi.PHOTON_F1:
        d = _displace(getattr(c.ph, k), rng)
        if k in ("wi", "parent_wi", "parent_n"):
            nz = (np.abs(d).sum(-1) > 0)
            d[nz] = _unit(d[nz])  # (a medium parent carries a zero normal)
        setattr(phD, k, d)
        setattr(ph32, k, np.ascontiguousarray(d.astype(np.float32)))
    for k in abi.PHOTON_U1:
        setattr(phD, k, getattr(c.ph, k).copy())
        setattr(ph32, k, getattr(c.ph, k).copy())
    # camera beams: the same fields in double
    fD = np.dtype([(n, np.float64 if c.rays.dtype[n].base == np.float32 else c.rays.dtype[n].base, c.rays.dtype[n].shape)
                   for n in c.rays.dtype.names])
    raysD = np.zeros(c.rays.shape, fD)
    rays32 = c.rays.copy()
    for n in c.rays.dtype.names:
        if c.rays.dtype[n].base != np.float32 or n == "rand":
            raysD[n] = c.rays[n]
            continue
        d = _displace(c.rays[n], rng)
        if n == "d":
            d = _unit(d)
        raysD[n] = d
        rays32[n] = d.astype(np.float32)
    cD, c32 = copy.copy(c), copy.copy(c)
    cD.ph, cD.rays = phD, raysD
    c32.ph, c32.rays = ph32, rays32
    return cD, c32


@pytest.mark.parametrize("scene,seed", [("cbox", 1), ("cbox_hg", 2), ("cbox_rot", 3)])
def test_double_built_records_through_the_fp32_boundary(scene, seed):
    c = cases.make_case(scene, 20, 16, 4000, 4.0)
    cD, c32 = double_case(c, seed)
    # the double records round to the uploaded ones, and differ from them
    assert np.array_equal(cD.ph.pos.astype(np.float32), c32.ph.pos) and np.abs(cD.ph.pos - c32.ph.pos).max() > 0
    ref, cnt = I.bre3d_full(cD)
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris)
    ctx.upload_medium(c.m)
    cases.upload_bsdfs(ctx, c)
    ctx.upload_photons(c32.ph)
    ctx.upload_camera_beams(c32.rays)
    ctx.gather(1, c.nb)
    acc = ctx.download_accum().astype(np.float64)
    st = ctx.stats()
    ctx.close()
    assert cnt["evaluations"] > 1000
    assert abs(st["evaluations"] - cnt["evaluations"]) <= max(2, 1e-4 * cnt["evaluations"]), (st["evaluations"], cnt["evaluations"])
    lum = ref[..., 0:3].mean()
    err = float(np.sqrt(((acc - ref) ** 2).mean()) / lum)
    assert err < 1e-3, err
