"""Throughput of the other gather techniques (parity cases of BASELINE.json, not the bench.py line): G-VPM at the
C1 shape, G-Beams 3D at a C3-like shape, G-Planes 0D at a C5-like shape; each with the CPU oracle (fp32, fast
build, all host cores) on a bounded sample.  Prints one JSON line per technique."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.cuda.init()
from gvpm_amd import abi, hip
from gvpm_amd.host import SynthScene
import oracle_lib as O


def run(name, ctx, uploads, gather_args, iters=4):
    for it in range(1, iters + 2):
        if it == 2:
            ctx.synchronize(); ctx.kernel_time(); ev0 = ctx.stats()["evaluations"]; t0 = time.perf_counter()
        uploads(it)
        ctx.gather(it, gather_args(it))
    ctx.synchronize()
    dt = time.perf_counter() - t0
    ev = ctx.stats()["evaluations"] - ev0
    ms, n = ctx.kernel_time()
    return dict(technique=name, mevals_per_s=round(ev / dt / 1e6, 1), evals_per_iter=ev // iters,
                ms_per_iter=round(dt / iters * 1e3, 3), kernel_ms=round(ms, 3))


res = []
# ---- G-VPM, C1: 256x256, 100k photons, 40 camera samples per pixel
sc = SynthScene("cbox", 256, 256)
p = sc.params(); p.vol_technique = abi.GVPM_DISTANCE; p.nb_camera_samples = 40; p.initial_scale_volume = 4.0
m, tris = sc.medium(), sc.triangles()
data = {it: (sc.shoot_photons(it, 100000), sc.camera_beams_and_vpm_samples(it, 40)) for it in range(1, 6)}
ctx = hip.Context(p, 0); ctx.upload_scene(*tris); ctx.upload_medium(m)
def up(it):
    (ph, nb), (rays, smp) = data[it]
    ctx.upload_photons(ph); ctx.upload_camera_beams(rays); ctx.upload_vpm_samples(smp)
r = run("G-VPM (C1: 256^2, 100k photons, 40 samples/pixel)", ctx, up, lambda it: data[it][0][1])
(ph, nb), (rays, smp) = data[1]
sel = smp[smp["set"] < 64 * 256]  # bounded sample: the first 64 rows of pixels
t0 = time.perf_counter(); _, _, _, cnt, secs = O.gather_vpm(p, m, tris, ph, rays, sel, 32, use_accel=True, fast=True)
r["cpu_oracle_mevals_per_s"] = round(cnt["evaluations"] / secs / 1e6, 2); r["cpu_cores"] = os.cpu_count()
res.append(r); ctx.close()

# ---- G-Beams 3D: 256x256, 200k beam segments
sc = SynthScene("cbox", 256, 256)
p = sc.params(); p.vol_technique = abi.GVPM_BEAM_BEAM_3D_OPTIMIZED; p.initial_scale_volume = 1.0
data = {it: (sc.shoot_beams(it, 200000), sc.camera_beams(it)) for it in range(1, 6)}
ctx = hip.Context(p, 0); ctx.upload_scene(*tris); ctx.upload_medium(m)
def up(it):
    (beams, en, nb), rays = data[it]
    ctx.upload_beams(beams, en); ctx.upload_camera_beams(rays)
r = run("G-Beams 3D (256^2, 200k beam segments)", ctx, up, lambda it: data[it][0][2])
(beams, en, nb), rays = data[1]
rad = float(np.float32(p.bsphere_radius) * np.float32(p.initial_scale_volume) * np.float32(0.01))
sub = rays[:16 * 256]
_, cnt, secs = O.gather_beams(p, m, tris, beams.subset(np.arange(0, beams.n, 8)), en[::8], sub, rad, 1, nb, 32, fast=True)
r["cpu_oracle_mevals_per_s"] = round(cnt["evaluations"] / secs / 1e6, 3); r["cpu_note"] = "ENoAccel loop (pm/beams.h:289-294), 1/8 of the beams x 16 pixel rows"
res.append(r); ctx.close()

# ---- G-Planes 0D: 256x256, camera inside the medium, 50k planes
sc = SynthScene("cbox_in", 256, 256)
p = sc.params(); p.vol_technique = abi.GVPM_VOL_PLANE0D; p.use_shift_null = 0; p.min_depth = 2
m, tris = sc.medium(), sc.triangles()
data = {it: (sc.shoot_planes(it, 50000), sc.camera_beams(it)) for it in range(1, 6)}
ctx = hip.Context(p, 0); ctx.upload_scene(*tris); ctx.upload_medium(m)
def up(it):
    (beams, en, w1, l1, nb), rays = data[it]
    ctx.upload_planes(beams, w1, l1); ctx.upload_camera_beams(rays)
r = run("G-Planes 0D (256^2, 50k planes)", ctx, up, lambda it: data[it][0][4])
(beams, en, w1, l1, nb), rays = data[1]
_, cnt, secs = O.gather_planes(p, m, tris, beams, w1, l1, rays[:8 * 256], 1, nb, 32, fast=True)
r["cpu_oracle_mevals_per_s"] = round(cnt["evaluations"] / secs / 1e6, 2)
res.append(r); ctx.close()
for r in res:
    print(json.dumps(r), flush=True)
