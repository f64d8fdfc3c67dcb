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

if __name__ == "__main__":
    for name, spec in SPECS.items():
        c = cases.make_case(**spec)
        acc, cnt, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 64, use_accel=False)
        path = os.path.join(HERE, name + ".npz")
        golden_io.save(path, c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, acc, cnt["evaluations"])
        print(name, cnt, os.path.getsize(path), "bytes")
