"""Quick device-vs-oracle check on the GPU box (development aid; the judged tests live in tests/)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from gvpm_amd import abi, hip
from gvpm_amd.host import SynthScene
import oracle_lib as O


def run(scene="cbox", W=32, H=32, nph=20000, scale=3.0, technique=abi.GVPM_VOL_BRE3D, shift_null=1, iters=1):
    sc = SynthScene(scene, W, H)
    p = sc.params()
    p.vol_technique = technique
    p.use_shift_null = shift_null
    p.initial_scale_volume = scale
    m, tris = sc.medium(), sc.triangles()
    ctx = hip.Context(p)
    ctx.upload_scene(*tris)
    ctx.upload_medium(m)
    acc_ref = None
    gs = scale
    for it in range(1, iters + 1):
        ph, nb = sc.shoot_photons(it, nph)
        rays = sc.camera_beams(it)
        r = ctx.radius()
        ctx.upload_photons(ph)
        ctx.upload_camera_beams(rays)
        t0 = time.time()
        ctx.gather(it, nb)
        ctx.synchronize()
        t1 = time.time()
        acc_ref, cnt, secs = O.gather_bre(p, m, tris, ph, rays, r, it, nb, 64, True, threads=0, accum=acc_ref)
        st = ctx.stats()
        acc = ctx.download_accum().astype(np.float64)
        lum = acc_ref[..., 0:3].mean()
        print(f"it {it} r={r:.6f} sets={rays.shape[0]} photons={ph.n} gpu {1e3*(t1-t0):.2f} ms; oracle {secs:.3f}s")
        print("  oracle counters", cnt)
        print("  device counters", st)
        for name, sl in (("flux", slice(0, 3)), ("shifted", slice(3, 15)), ("weighted", slice(15, 27))):
            d = acc[..., sl] - acc_ref[..., sl]
            l2 = np.sqrt((d ** 2).mean()) / lum
            print(f"  {name}: L2/lum = {l2:.3e}  max|d|/lum = {np.abs(d).max()/lum:.3e}  sum dev {acc[...,sl].sum():.6f} ref {acc_ref[...,sl].sum():.6f}")
    ms, n = ctx.kernel_time()
    print("kernel avg ms", ms, n)
    ctx.close()


if __name__ == "__main__":
    args = sys.argv[1:]
    run()
    run(W=48, H=40, nph=50000, scale=2.0, iters=2)
    run(scene="cbox_hg", W=32, H=32, nph=20000, scale=3.0)
    run(technique=abi.GVPM_VOL_BRE2D, shift_null=0)
