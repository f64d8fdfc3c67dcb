"""G-Beams throughput probe (C3-like shape): python scripts/beams_bench.py [--tech 3d|1d] [--size 256] [--beams 200000]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
torch.cuda.init()
from gvpm_amd import abi, hip
from gvpm_amd.host import SynthScene

ap = argparse.ArgumentParser()
ap.add_argument("--tech", default="3d")
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--beams", type=int, default=200000)
ap.add_argument("--iters", type=int, default=4)
ap.add_argument("--scale", type=float, default=1.0)
ap.add_argument("--scene", default="cbox")
ap.add_argument("--phases", action="store_true", help="also report the traversal and build phases (gvpm_get_phase_time)")
args = ap.parse_args()
sc = SynthScene(args.scene, args.size, args.size)
p = sc.params()
p.vol_technique = abi.GVPM_BEAM_BEAM_3D_OPTIMIZED if args.tech == "3d" else abi.GVPM_BEAM_BEAM_1D
p.initial_scale_volume = args.scale
m, tris = sc.medium(), sc.triangles()
data = {it: (sc.shoot_beams(it, args.beams), sc.camera_beams(it)) for it in range(1, args.iters + 2)}
ctx = hip.Context(p, 0); ctx.upload_scene(*tris); ctx.upload_medium(m)
for it in range(1, args.iters + 2):
    if it == 2:
        ctx.synchronize(); ctx.kernel_time(); ctx.phase_time(1); ctx.phase_time(2); s0 = ctx.stats(); t0 = time.perf_counter()
    (beams, en, nb), rays = data[it]
    ctx.upload_beams(beams, en); ctx.upload_camera_beams(rays)
    ctx.gather(it, nb)
ctx.synchronize()
dt = time.perf_counter() - t0
s1 = ctx.stats()
d = {k: (s1[k] - s0[k]) // args.iters for k in s1 if isinstance(s1[k], int)}
ms, n = ctx.kernel_time()
out = dict(tech=args.tech, nbeams=int(data[1][0][0].n), mevals_per_s=round(d["evaluations"] * args.iters / dt / 1e6, 1),
           ms_per_iter=round(dt / args.iters * 1e3, 3), kernel_ms=round(ms, 3), per_iter=d)
if args.phases:
    out["trav_ms"] = round(ctx.phase_time(1)[0], 3)
    out["build_ms"] = round(ctx.phase_time(2)[0], 3)
print(json.dumps(out))
ctx.close()
