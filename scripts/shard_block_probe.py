"""Probe (C4, 8-way strong sharding on ONE GPU): how large should the round-robin image blocks be?  4x4-pixel tiles balance
the evaluations to 0.99 but make every rank touch most of the photon map (its rays cross the whole volume): a rank's
evaluation costs twice what it costs on one GPU per evaluation.  Larger blocks keep a rank's photons together.
For block sizes S: evaluations per rank (balance) and the time per step of every rank's shard run alone on the GPU."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
torch.cuda.init()
from gvpm_amd import abi, hip
from gvpm_amd.host import SynthScene

N = 8
W = H = int(os.environ.get("FRAME", 1024))
NPH = int(os.environ.get("PHOTONS", 4000000))
sc = SynthScene("fogroom", W, H)
p = sc.params(); p.vol_technique = abi.GVPM_VOL_BRE3D; p.initial_scale_volume = 1.0
m, tris = sc.medium(), sc.triangles()
ndist = 3
data = []
keep = []
for i in range(ndist):
    ph, nb = sc.shoot_photons(i + 1, NPH)
    rays = sc.camera_beams(i + 1)
    soa = abi.PhotonSoA()
    for k in abi.PHOTON_VEC3 + abi.PHOTON_F1 + abi.PHOTON_U1:
        a = getattr(ph, k)
        t = torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else a).cuda(); keep.append(t)
        setattr(soa, k, t.data_ptr())
    soa.n = ph.n
    data.append((soa, nb, rays))


def owner(rays, S, skew):
    px = (rays["pixel"][:, 0] & 0xFFFF).astype(np.int64); py = (rays["pixel"][:, 0] >> 16).astype(np.int64)
    bx, by = px // S, py // S
    nbx = (W + S - 1) // S
    return ((by * nbx + bx) % N) if not skew else ((bx + by * 3) % N)


for S, skew in ((4, 0), (16, 1), (32, 1), (64, 1), (128, 1)):
    res = []
    for r in range(N if os.environ.get("ALL_RANKS") else 2):
        rdev = []
        for i in range(ndist):
            rr = np.ascontiguousarray(data[i][2][owner(data[i][2], S, skew) == r])
            t = torch.from_numpy(rr.view(np.uint8).reshape(-1)).cuda(); keep.append(t)
            rdev.append((t.data_ptr(), rr.shape[0]))
        ctx = hip.Context(p, 0); ctx.upload_scene(*tris); ctx.upload_medium(m)
        K = 8
        for it in range(1, K + 3):
            if it == 3:
                ctx.synchronize(); ev0 = ctx.stats()["evaluations"]; t0 = time.perf_counter()
            soa, nb, _ = data[(it - 1) % ndist]
            ctx.upload_photons_dev(soa); ctx.upload_camera_beams_dev(*rdev[(it - 1) % ndist]); ctx.gather(it, nb)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / K
        ev = (ctx.stats()["evaluations"] - ev0) / K
        ctx.close()
        res.append((ev, dt * 1e3))
    e = np.array([x[0] for x in res]); t = np.array([x[1] for x in res])
    print(f"blocks of {S:3d} px ({'skewed' if skew else 'row-major'} round-robin): evals/rank {e.mean() / 1e6:.1f} M, balance over the measured ranks (mean/max) {e.mean() / e.max():.3f}, "
          f"ms/step per rank: mean {t.mean():.2f} max {t.max():.2f}", flush=True)
