import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import cases, unbiased
from gvpm_amd import abi, hip
from gvpm_amd.host import SynthScene
tech, N, nph, W, H, scale = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6])
kw = {}
for a in sys.argv[7:]:
    k, v = a.split("="); kw[k] = int(v)
sc = SynthScene("cbox", W, H)
p = sc.params()
p.initial_scale_volume = scale
p.alpha = 1.0
p.visibility_as_written = 0
p.vol_technique = dict(bre3d=abi.GVPM_VOL_BRE3D, bre2d=abi.GVPM_VOL_BRE2D, vpm=abi.GVPM_DISTANCE, beams3d=abi.GVPM_BEAM_BEAM_3D_OPTIMIZED, beams1d=abi.GVPM_BEAM_BEAM_1D)[tech]
if tech == "bre2d": p.use_shift_null = 0
if tech == "vpm": p.nb_camera_samples = 8
for k, v in kw.items(): setattr(p, k, v)
ctx = hip.Context(p, device=0)
ctx.upload_scene(*sc.triangles()); ctx.upload_medium(sc.medium())
def step(k):
    it = k + 1
    ctx.reset()
    if tech.startswith("beams"):
        b, en, nb = sc.shoot_beams(it, nph); ctx.upload_beams(b, en); ctx.upload_camera_beams(sc.camera_beams(it))
    elif tech == "vpm":
        ph, nb = sc.shoot_photons(it, nph); r, smp = sc.camera_beams_and_vpm_samples(it, p.nb_camera_samples)
        ctx.upload_photons(ph); ctx.upload_camera_beams(r); ctx.upload_vpm_samples(smp)
    else:
        ph, nb = sc.shoot_photons(it, nph); ctx.upload_photons(ph); ctx.upload_camera_beams(sc.camera_beams(it))
    ctx.gather(1, nb)
    return ctx.download_film(1, False)
t0 = time.time()
out = unbiased.run(step, N)
print(tech, kw, "N", N, "%.1fs" % (time.time() - t0), ctx.stats())
np.set_printoptions(linewidth=250, precision=1, suppress=True)
for k, v in out.items():
    print("  ", k, {a: b for a, b in v.items() if not isinstance(b, np.ndarray)})
    print("   z (green), rows = y:"); print(v["z"][..., 1])
    print("   mean D / |ref| max:"); print((v["mean"][..., 1] / max(np.abs(v["ref"]).max(), 1e-30) * 100))
