"""Stress: G-VPM device == fp64 oracle over scenes x flags x radii x iterations, shift counters asserted exactly (device_vpm's
default): python tests/stress_vpm.py [scene ...]   (on the GPU box; not collected by pytest)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from test_oracle_vpm import make_vpm_case
from test_parity_vpm_gpu import device_vpm
n = 0
IT = int(os.environ.get("STRESS_IT", "1"))  # (the iteration the inputs are generated for: other random streams, another radius)
SCENES = ("cbox", "cbox_hg", "fogroom", "cbox_mirror", "cbox_phong", "cbox_rot", "fogroom_rot", "cbox_hg_rot", "cbox_mirror_rot",
          "cbox_conductor_rot", "cbox_phong1_rot", "cbox_ward_rot")
for scene in (sys.argv[1:] or SCENES):
    for kw in (dict(), dict(use_mis=0), dict(path_set=0, max_depth=4), dict(visibility_as_written=0), dict(use_shift_null=0)):
        for scale in (2.0, 5.0):
            for nb in (4, 10):
                c = make_vpm_case(scene, 36, 30, 30000, scale, nb=nb, it=IT, **kw)
                res = device_vpm(c, iters=2 if nb == 4 else 1)
                n += 1
                st = res[2] if isinstance(res, tuple) and len(res) > 2 and isinstance(res[2], dict) else {}
                print(scene, kw, scale, nb, st.get("evaluations"), st.get("diffuse_shifts"), st.get("failed_shifts"), flush=True)
print("cases", n)
