"""Parity stress over many independent inputs: every SPPM iteration number keys its own photon map and camera jitter,
so sweeping `it` gives fresh random cases.  Device vs fp64 oracle: evaluation counts must be equal (G-BRE, G-Planes)
or within 2e-4 (G-Beams), L2/lum below the bars of the test suite.   python scripts/iteration_sweep.py [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
torch.cuda.init()
import cases, oracle_lib as O
from gvpm_amd import abi, hip

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
l2 = lambda a, r, lum: float(np.sqrt(((a.astype(np.float64) - r) ** 2).mean()) / lum)
worst = dict(bre=0.0, beams=0.0, planes=0.0)
for it in range(2, 2 + n):
    scene = ("cbox", "cbox_hg", "fogroom")[it % 3]
    # G-BRE 3D
    c = cases.make_case(scene, 28, 24, 12000, 3.0, it=it)
    ctx = hip.Context(c.p, 0); ctx.upload_scene(*c.tris); ctx.upload_medium(c.m); ctx.upload_photons(c.ph); ctx.upload_camera_beams(c.rays)
    ctx.gather(1, c.nb); acc = ctx.download_accum(); st = ctx.stats(); ctx.close()
    ref, cnt, _ = O.gather_bre(c.p, c.m, c.tris, c.ph, c.rays, c.r, 1, c.nb, 64, use_accel=False)
    assert st["evaluations"] == cnt["evaluations"], ("bre", it, st, cnt)
    e = l2(acc, ref, max(ref[..., 0:3].mean(), 1e-30)); worst["bre"] = max(worst["bre"], e); assert e < 1e-4, ("bre", it, e)
    # G-Beams 3D
    p = c.p.copy(); p.vol_technique = abi.GVPM_BEAM_BEAM_3D_OPTIMIZED
    beams, en, nb = c.sc.shoot_beams(it, 3000)
    ctx = hip.Context(p, 0); ctx.upload_scene(*c.tris); ctx.upload_medium(c.m); ctx.upload_beams(beams, en); ctx.upload_camera_beams(c.rays)
    rad = ctx.radius(); ctx.gather(1, nb); acc = ctx.download_accum(); st = ctx.stats(); ctx.close()
    ref, cnt, _ = O.gather_beams(p, c.m, c.tris, beams, en, c.rays, rad, 1, nb, 64)
    assert abs(st["evaluations"] - cnt["evaluations"]) <= max(2, 2e-4 * cnt["evaluations"]), ("beams", it, st, cnt)
    e = l2(acc, ref, max(ref[..., 0:3].mean(), 1e-30)); worst["beams"] = max(worst["beams"], e); assert e < 1e-3, ("beams", it, e)
    # G-Planes 0D
    ci = cases.make_case("cbox_in", 28, 24, 10, 1.0, it=it, vol_technique=abi.GVPM_VOL_PLANE0D, use_shift_null=0, min_depth=2)
    pb, pen, w1, l1, pnb = ci.sc.shoot_planes(it, 2500)
    ctx = hip.Context(ci.p, 0); ctx.upload_scene(*ci.tris); ctx.upload_medium(ci.m); ctx.upload_planes(pb, w1, l1); ctx.upload_camera_beams(ci.rays)
    ctx.gather(1, pnb); acc = ctx.download_accum(); st = ctx.stats(); ctx.close()
    ref, cnt, _ = O.gather_planes(ci.p, ci.m, ci.tris, pb, w1, l1, ci.rays, 1, pnb, 64)
    assert st["evaluations"] == cnt["evaluations"], ("planes", it, st, cnt)
    e = l2(acc, ref, max(ref[..., 0:3].mean(), 1e-30)); worst["planes"] = max(worst["planes"], e); assert e < 1e-5, ("planes", it, e)
    print(it, scene, "ok", flush=True)
print("worst L2/lum:", worst)
