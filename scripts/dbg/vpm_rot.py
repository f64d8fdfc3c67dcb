import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import cases, oracle_lib as O
from gvpm_amd import abi, hip
from test_oracle_vpm import make_vpm_case
for scene in sys.argv[1:]:
    c = make_vpm_case(scene, 32, 28, 40000, 3.0, nb=10)
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris); ctx.upload_medium(c.m); cases.upload_bsdfs(ctx, c)
    ctx.upload_photons(c.ph); ctx.upload_camera_beams(c.rays); ctx.upload_vpm_samples(c.samples)
    ctx.gather(1, c.nb)
    acc = ctx.download_accum().astype(np.float64)
    film = ctx.download_film(1, True)
    st = ctx.stats()
    ctx.close()
    ref, sv, nv, cnt, _ = O.gather_vpm(c.p, c.m, c.tris, c.ph, c.rays, c.samples, 64, use_accel=False)
    lum = ref[..., :3].mean()
    print(scene, st["evaluations"], cnt["evaluations"], "acc L2", np.sqrt(((acc-ref)**2).mean())/lum)
    for j in range(9):
        d = np.abs(acc[..., 3*j:3*j+3]-ref[..., 3*j:3*j+3])
        print("  acc", j, np.sqrt((d**2).mean())/lum, d.max()/lum, np.unravel_index(d.argmax(), d.shape))
    rfilm = O.assemble(ref, 1, True, total_emitted=c.nb)
    dfilm = O.assemble(acc, 1, True, total_emitted=c.nb)
    for name, a, b, d in zip(("thr", "dx", "dy"), film, rfilm, dfilm):
        e = np.abs(a.astype(np.float64) - b)
        print("  film", name, np.sqrt((e**2).mean())/(lum/c.nb), e.max()/(lum/c.nb), np.unravel_index(e.argmax(), e.shape),
              "| oracle-assemble(device acc) vs oracle:", np.sqrt(((d-b)**2).mean())/(lum/c.nb),
              "| device film vs assemble(device acc):", np.sqrt(((a-d)**2).mean())/(lum/c.nb))
