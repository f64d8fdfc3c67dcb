import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import cases, unbiased, oracle_lib as O
from gvpm_amd import abi
from gvpm_amd.host import SynthScene
tech, N, nph = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
W, H = 32, 24
sc = SynthScene("cbox", W, H)
p = sc.params(); p.initial_scale_volume = 3.0; p.alpha = 1.0; p.visibility_as_written = 0
p.vol_technique = abi.GVPM_VOL_BRE2D if tech == "bre2d" else abi.GVPM_VOL_BRE3D
if tech == "bre2d": p.use_shift_null = 0
m, tris = sc.medium(), sc.triangles()
r = cases.radius_of(p)
def step(k):
    it = k + 1
    ph, nb = sc.shoot_photons(it, nph); rays = sc.camera_beams(it)
    ref, cnt, _ = O.gather_bre(p, m, tris, ph, rays, r, 1, nb, 64, use_accel=True)
    return O.assemble(ref, 1, False)
t0 = time.time()
out = unbiased.run(step, N)
print(tech, "oracle N", N, "%.1fs" % (time.time() - t0))
for key in ("dx", "dy"):
    o = out[key]; z = o["z"]; bad = np.argwhere(z > 4.0)
    print(f"   {key}: slope {o['slope']:.4f} rel L2 {o['rel_l2']:.4f} (noise {o['noise_l2']:.4f}) max|z| {o['zmax']:.2f} n>4 {o['n_over4']}/{o['n_tests']}; z>4 at (y,x,c): {bad[:12].tolist()}")
    if key == "dy": print("   z[22, 10:20, 1]", np.round(z[22, 10:20, 1], 1))
    else: print("   z[:, 0, 0]", np.round(z[:, 0, 0], 1), "z[:, 30, 0]", np.round(z[:, 30, 0], 1))
