"""Stress: G-BRE device == fp64 oracle over scenes x flags x sharded beam sets (bundle cells with striped counters for the
shards, 3D grid for the whole frame), several radii.  python tests/stress_bre.py [scene ...]   (on the GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # (the repository: this file lives in tests/)
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import cases
from test_parity_gpu import check
from gvpm_amd import abi
n = 0
IT = int(os.environ.get("STRESS_IT", "1"))  # (the iteration the inputs are generated for: other random streams, another radius)
SCENES = ("cbox", "cbox_hg", "fogroom", "cbox_mirror", "laser", "cbox_phong", "cbox_conductor", "cbox_phong1",
              # general position (round 5): shift counters exact there too (check() asserts them exactly by default)
              "cbox_rot", "fogroom_rot", "cbox_mirror_rot", "cbox_phong1_rot", "cbox_conductor_rot")
for scene in (sys.argv[1:] or SCENES):
    for kw in (dict(), dict(vol_technique=abi.GVPM_VOL_BRE2D, use_shift_null=0), dict(path_set=0), dict(use_mis=0, max_depth=4)):
        for scale in (1.5, 4.0):
            c = cases.make_case(scene, 48, 40, 25000, scale, it=IT, **kw)
            for world in (1, 2, 8):
                for rank in ((0,) if world == 1 else (0, world - 1)):
                    rays = c.sc.camera_beams_interleaved(c.it, world, rank) if world > 1 else c.rays
                    for bundle in ("0", "1"):
                        os.environ["GVPM_BUNDLE"] = bundle
                        acc, ref, st = check(c, rays=rays)
                        n += 1
            print(scene, kw, scale, st["evaluations"], flush=True)
print("cases", n)
