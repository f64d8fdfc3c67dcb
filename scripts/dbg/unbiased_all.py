import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
from test_unbiased_gpu import estimate
cfgs = [("bre3d", 3000, 60000, 3.0, {}), ("bre3d", 3000, 60000, 3.0, dict(use_mis=0, path_set=0)), ("bre3d", 3000, 60000, 3.0, dict(use_shift_null=0)),
        ("bre3d", 3000, 60000, 3.0, dict(power_heuristic=1)),
        ("bre2d", 3000, 60000, 3.0, {}), ("vpm", 1500, 40000, 6.0, {}), ("beams3d", 3000, 20000, 2.0, {}), ("beams3d", 3000, 20000, 2.0, dict(use_mis=0, path_set=0)),
        ("beams1d", 3000, 20000, 2.0, {})]
for tech, n, nph, scale, kw in cfgs:
    t0 = time.time()
    out, st = estimate(tech, n, nph, scale, **kw)
    print(tech, kw, "N", n, "%.1fs" % (time.time() - t0), {k: st[k] for k in ("evaluations", "null_shifts", "diffuse_shifts", "failed_shifts")}, flush=True)
    for key in ("dx", "dy"):
        o = out[key]
        z = o["z"]
        bad = np.argwhere(z > 4.5)
        print(f"   {key}: slope {o['slope']:.4f} rel L2 {o['rel_l2']:.4f} (noise {o['noise_l2']:.4f}) max|z| {o['zmax']:.2f} n>4 {o['n_over4']}/{o['n_tests']} grad/thr {o['grad_over_thr']:.2f}; z>4.5 at (y,x,c): {bad[:8].tolist()}", flush=True)
