"""Stress: G-Planes 0D device == fp64 oracle over scenes x flags x plane counts x iterations, shift counters asserted exactly
(device_planes' default): python tests/stress_planes.py [scene ...]   (on the GPU box; not collected by pytest)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from test_oracle_planes import make_plane_case
from test_parity_planes_gpu import device_planes
n = 0
IT = int(os.environ.get("STRESS_IT", "1"))  # (the iteration the inputs are generated for: other random streams, another radius)
for scene in (sys.argv[1:] or ("cbox_in", "cbox_in_rot", "laser_in", "laser_in_hg")):
    for kw in (dict(), dict(use_mis=0), dict(power_heuristic=1), dict(path_set=0), dict(max_depth=3)):
        for W, H, npl, iters in ((32, 28, 6000, 1), (70, 50, 3000, 1), (40, 32, 4000, 3)):
            c = make_plane_case(scene, W, H, npl, it=IT, **kw)
            res = device_planes(c, iters=iters)
            n += 1
            st = res[2] if isinstance(res, tuple) and len(res) > 2 and isinstance(res[2], dict) else {}
            print(scene, kw, W, H, npl, iters, st.get("evaluations"), st.get("diffuse_shifts"), st.get("failed_shifts"), flush=True)
print("cases", n)
