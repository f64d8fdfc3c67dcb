import os, sys, ctypes as C
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from gvpm_amd import abi, hip
from test_oracle_beams import make_beam_case
import cases
c = make_beam_case("laser", 96, 96, 60000, 1.0)
ctx = hip.Context(c.p, device=0)
ctx.upload_scene(*c.tris); ctx.upload_medium(c.m); cases.upload_bsdfs(ctx, c)
ctx.upload_beams(c.beams, c.end_n); ctx.upload_camera_beams(c.rays)
ctx.gather(1, c.nb)
print(ctx.stats(), ctx.exact_shifts())
out = (C.c_uint * 512)()
hip.lib().gvpm_debug_vis(out)
h = np.array(out[:])
print("by triangle:", {i: int(v) for i, v in enumerate(h[:256]) if v})
print("by reason (1 cross, 2 start in band, 4 end in band):", {i: int(v) for i, v in enumerate(h[256:264]) if v})
v0, e1, e2 = c.tris
for i in np.nonzero(h[:256])[0][:6]:
    print(i, v0[i], e1[i], e2[i])
