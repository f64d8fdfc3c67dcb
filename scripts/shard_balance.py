"""How evenly do image shards split the G-BRE work?  Evaluations per rank for contiguous tiles vs interleaved
row strips, 8 ranks, S-cbox (a sizing probe for bench.py's sharding)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
torch.cuda.init()
from gvpm_amd import abi, hip
from gvpm_amd.host import SynthScene

N, T = 8, 128
tx, ty = 4, 2
W, H = T * tx, T * ty
sc = SynthScene("cbox", W, H)
p = sc.params(); p.vol_technique = abi.GVPM_VOL_BRE3D; p.initial_scale_volume = 1.0
ph, nb = sc.shoot_photons(1, 1000000)
rays = sc.camera_beams(1)
px = (rays["pixel"][:, 0] & 0xFFFF).astype(np.int64); py = (rays["pixel"][:, 0] >> 16).astype(np.int64)

def evals(sel):
    ctx = hip.Context(p, 0)
    ctx.upload_scene(*sc.triangles()); ctx.upload_medium(sc.medium()); ctx.upload_photons(ph)
    ctx.upload_camera_beams(np.ascontiguousarray(rays[sel]))
    ctx.gather(1, nb)
    e = ctx.stats()["evaluations"]
    ctx.close()
    return e

for name, owner in (("contiguous 4x2 tiles", (py // T) * tx + px // T),
                    ("column strips", px // (W // N)),
                    ("interleaved 4-row strips", (py // 4) % N),
                    ("interleaved 4x4 pixel tiles", ((py // 4) * (W // 4) + px // 4) % N)):
    e = np.array([evals(owner == r) for r in range(N)], np.float64)
    print(f"{name:30s} evals/rank min {e.min():.3g} max {e.max():.3g}  balance (mean/max) {e.mean() / e.max():.3f}")
