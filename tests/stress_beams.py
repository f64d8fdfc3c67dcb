"""Stress: G-Beams device == fp64 oracle on worst cases for the evaluation's queues (no null shifts, free cone off, both kernels,
axis-aligned and general-position scenes, two radii; the shift counters asserted exactly -- device_beams' default):
python tests/stress_beams.py [scene ...]   (on the GPU box; not collected by pytest)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from test_oracle_beams import make_beam_case
from test_parity_beams_gpu import device_beams
from gvpm_amd import abi
n = 0
IT = int(os.environ.get("STRESS_IT", "1"))  # (the iteration the inputs are generated for: other random streams, another radius)
SCENES = ("laser", "cbox", "fogroom", "laser_rot", "cbox_rot", "fogroom_rot", "cbox_hg_rot", "cbox_conductor_rot", "cbox_phong1_rot",
          "cbox_ward_rot")
for scene in (sys.argv[1:] or SCENES):
    for tech in (abi.GVPM_BEAM_BEAM_3D_OPTIMIZED, abi.GVPM_BEAM_BEAM_1D):
        for kw in (dict(use_shift_null=0), dict(use_shift_null=0, path_set=0), dict(path_set=0, max_depth=4), dict()):
            for fc in ("1", "0"):
                os.environ["GVPM_BEAMS_FREE_CONE"] = fc
                if tech == abi.GVPM_BEAM_BEAM_1D and "use_shift_null" in kw:
                    kw = {k: v for k, v in kw.items() if k != "use_shift_null"}
                for scale in ((3.0,) if not scene.endswith("_rot") else (1.6, 3.0)):
                    c = make_beam_case(scene, 40, 32, 9000, scale, technique=tech, it=IT, **kw)
                    acc, ref, st = device_beams(c)
                    n += 1
                    print(scene, tech, kw, fc, scale, st["evaluations"], st["diffuse_shifts"], st["failed_shifts"], flush=True)
print("cases", n)
