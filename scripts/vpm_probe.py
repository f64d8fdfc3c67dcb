"""G-VPM at the C1 shape (256^2, 100k photons, 40 camera samples per pixel): kernel time and counters of one context.
A/B probes select another build of the library with GVPM_HIP_LIB."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.cuda.init()
from gvpm_amd import abi, hip
from gvpm_amd.host import SynthScene

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sc = SynthScene("cbox", 256, 256)
p = sc.params(); p.vol_technique = abi.GVPM_DISTANCE; p.nb_camera_samples = 40; p.initial_scale_volume = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
m, tris = sc.medium(), sc.triangles()
data = {it: (sc.shoot_photons(it, 100000), sc.camera_beams_and_vpm_samples(it, 40)) for it in range(1, iters + 2)}
ctx = hip.Context(p, 0); ctx.upload_scene(*tris); ctx.upload_medium(m)
for it in range(1, iters + 2):
    if it == 2:
        ctx.synchronize(); ctx.kernel_time(); s0 = ctx.stats(); t0 = time.perf_counter()
    (ph, nb), (rays, smp) = data[it]
    ctx.upload_photons(ph); ctx.upload_camera_beams(rays); ctx.upload_vpm_samples(smp)
    ctx.gather(it, nb)
ctx.synchronize()
dt = time.perf_counter() - t0
s1 = ctx.stats()
ms, n = ctx.kernel_time()
d = {k: (s1[k] - s0[k]) // iters for k in s1 if isinstance(s1[k], int)}
print(json.dumps(dict(lib=os.path.basename(hip.LIB_PATH), kernel_ms=round(ms, 3), ms_per_iter=round(dt / iters * 1e3, 3), per_iter=d)))
