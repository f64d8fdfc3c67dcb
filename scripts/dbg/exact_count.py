import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, time
import cases
from gvpm_amd import abi, hip
scene, W, H, nph, scale = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
kw = {}
for a in sys.argv[6:]:
    k, v = a.split("="); kw[k] = int(v)
c = cases.make_case(scene, W, H, nph, scale, **kw)
ctx = hip.Context(c.p, device=0)
ctx.upload_scene(*c.tris); ctx.upload_medium(c.m); cases.upload_bsdfs(ctx, c)
ctx.upload_photons(c.ph); ctx.upload_camera_beams(c.rays)
for it in range(1, 4):
    ctx.gather(it, c.nb)
st = ctx.stats()
print(scene, st, "exact (evaluated, lost):", ctx.exact_shifts(), "per step", ctx.exact_shifts()[0] / 3, "shifts/step", 4 * st["evaluations"] / 3)
ctx.close()
