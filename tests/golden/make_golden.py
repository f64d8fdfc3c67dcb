#!/usr/bin/env python3
"""Generates the fixtures of tests/golden/*.npz: seeded synthetic inputs (gvpm_amd.host) and the
outputs of the fp64 CPU oracle (own-box brute-force mode) on them.

The reference cannot be built in this image and holds no vectors for this path,
so these fixtures pin the ORACLE against regressions and give the GPU tests committed
input/output pairs; they are not outputs of the reference binary ("parity unpinned").

    python tests/golden/make_golden.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cases  # noqa: E402
import golden_io  # noqa: E402
import oracle_lib as O  # noqa: E402
from gvpm_amd import abi  # noqa: E402

SPECS = {
    "cbox_bre3d": dict(scene="cbox", W=16, H=12, nph=2500, scale=5.0),
    "cbox_hg_bre3d": dict(scene="cbox_hg", W=16, H=12, nph=2500, scale=5.0),
    "cbox_bre2d": dict(scene="cbox", W=16, H=12, nph=2500, scale=5.0, vol_technique=abi.GVPM_VOL_BRE2D,
                       use_shift_null=0),
}

def other_techniques():
    """G-VPM, G-Beams (3D and 1D) and G-Planes fixtures: same idea, technique-specific inputs in `extra`."""
    from test_oracle_vpm import make_vpm_case
    from test_oracle_beams import make_beam_case
    from test_oracle_planes import make_plane_case
    c = make_vpm_case("cbox", 14, 10, 4000, 6.0, nb=4)
    acc, sv, nv, cnt, _ = O.gather_vpm(c.p, c.m, c.tris, c.ph, c.rays, c.samples, 64, use_accel=False)
    yield "cbox_vpm", c, c.ph, acc, cnt, dict(samples=c.samples.view("u1"), scale_vol=sv, n_vol=nv)
    for name, tech in (("cbox_beams3d", abi.GVPM_BEAM_BEAM_3D_OPTIMIZED), ("cbox_beams1d", abi.GVPM_BEAM_BEAM_1D)):
        c = make_beam_case("cbox", 14, 10, 1200, 3.0, technique=tech)
        acc, cnt, _ = O.gather_beams(c.p, c.m, c.tris, c.beams, c.end_n, c.rays, c.r, 1, c.nb, 64)
        yield name, c, c.beams, acc, cnt, dict(end_n=c.end_n)
    c = make_plane_case("cbox_in", 14, 10, 800)
    acc, cnt, _ = O.gather_planes(c.p, c.m, c.tris, c.beams, c.w1, c.len1, c.rays, 1, c.nb, 64)
    yield "cbox_in_planes0d", c, c.beams, acc, cnt, dict(end_n=c.end_n, w1=c.w1, len1=c.len1)


if __name__ == "__main__":
    for name, c, ph, acc, cnt, extra in other_techniques():
        path = os.path.join(HERE, name + ".npz")
        golden_io.save(path, c.p, c.m, c.tris, ph, c.rays, c.r, 1, c.nb, acc, cnt["evaluations"], extra)
        print(name, cnt, os.path.getsize(path), "bytes")
    for name, spec in SPECS.items():
        c = cases.make_case(**spec)
        acc, cnt, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 64, use_accel=False)
        path = os.path.join(HERE, name + ".npz")
        golden_io.save(path, c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, acc, cnt["evaluations"])
        print(name, cnt, os.path.getsize(path), "bytes")
