import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from gvpm_amd import abi, hip
from test_oracle_beams import make_beam_case
import cases
for scene, n in (("laser", 60000), ("cbox", 12000), ("laser_rot", 12000)):
    c = make_beam_case(scene, 96, 96, n, 1.0, technique=abi.GVPM_BEAM_BEAM_1D)
    ctx = hip.Context(c.p, device=0)
    ctx.upload_scene(*c.tris); ctx.upload_medium(c.m); cases.upload_bsdfs(ctx, c)
    ctx.upload_beams(c.beams, c.end_n); ctx.upload_camera_beams(c.rays)
    ctx.gather(1, c.nb)
    print(scene, ctx.stats(), ctx.exact_shifts(), flush=True)
    ctx.close()
