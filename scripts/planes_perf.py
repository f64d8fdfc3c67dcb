"""G-Planes kernel timing on a synthetic inside-camera scene (not the bench line; a sizing probe)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from gvpm_amd import abi, hip
from gvpm_amd.host import SynthScene

W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
sc = SynthScene("cbox_in", W, W)
p = sc.params(); p.vol_technique = abi.GVPM_VOL_PLANE0D; p.use_shift_null = 0; p.min_depth = 2
ctx = hip.Context(p, 0)
ctx.upload_scene(*sc.triangles()); ctx.upload_medium(sc.medium())
beams, en, w1, l1, nb = sc.shoot_planes(1, N)
rays = sc.camera_beams(1)
ctx.upload_planes(beams, w1, l1); ctx.upload_camera_beams(rays)
for it in range(1, 4):
    ctx.gather(it, nb)
ctx.synchronize()
ms, n = ctx.kernel_time()
st = ctx.stats()
pairs = rays.shape[0] * beams.n
print(f"planes W={W} planes={beams.n} kernel {ms:.3f} ms  pairs/launch {pairs:.3e}  {pairs/ms/1e6:.1f} Gpairs/s  "
      f"evals/launch {st['evaluations']/n:.0f}  {st['evaluations']/n/ms/1e3:.1f} Mevals/s")
