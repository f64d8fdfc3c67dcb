"""Find the (camera set, beam) pair on which the device's G-Beams shift counters differ from the oracle's: bisection over the
camera sets, then over the beams.  python scripts/probes_py/beams_bisect.py scene tech scale [free_cone]   (GPU box)"""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from gvpm_amd import abi, hip
from test_oracle_beams import make_beam_case
import cases, oracle_lib as O
scene, tech, scale = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
os.environ["GVPM_BEAMS_FREE_CONE"] = sys.argv[4] if len(sys.argv) > 4 else "1"
kw = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[5:] if "=" in a}
c = make_beam_case(scene, 40, 32, 9000, scale, technique=tech, it=int(os.environ.get("STRESS_IT", "1")), **kw)
KEYS = ("evaluations", "null_shifts", "diffuse_shifts", "failed_shifts")

def run(rays, beams, en, exact_all=False):
    if exact_all: os.environ["GVPM_EXACT_ALL"] = "1"
    ctx = hip.Context(c.p, device=0)
    os.environ.pop("GVPM_EXACT_ALL", None)
    ctx.upload_scene(*c.tris); ctx.upload_medium(c.m); cases.upload_bsdfs(ctx, c)
    rad = ctx.radius()
    ctx.upload_beams(beams, en); ctx.upload_camera_beams(rays)
    ctx.gather(1, c.nb)
    st = ctx.stats(); ex = ctx.exact_shifts()
    ctx.close()
    ref, cnt, _ = O.gather_beams(c.p, c.m, c.tris, beams, en, rays, rad, 1, c.nb, 64)
    return tuple(st[k] for k in KEYS), tuple(cnt[k] for k in KEYS), ex

rays, beams, en = c.rays, c.beams, c.end_n
if "--pair" in sys.argv:  # (second stage, under a probe library: the one pair, with the library's prints)
    k = sys.argv.index("--pair"); si, bi = int(sys.argv[k + 1]), int(sys.argv[k + 2])
    print("pair", si, bi, run(np.ascontiguousarray(rays[si:si + 1]), beams.subset(np.array([bi])), np.ascontiguousarray(en[bi:bi + 1])))
    sys.exit(0)
sidx = np.arange(rays.shape[0])
d, o, ex = run(rays, beams, en)
print("all:", d, o, ex, flush=True)
if d == o: sys.exit(0)
while rays.shape[0] > 1:
    h = rays.shape[0] // 2
    for part, pi in ((rays[:h], sidx[:h]), (rays[h:], sidx[h:])):
        d, o, _ = run(np.ascontiguousarray(part), beams, en)
        if d != o:
            rays = np.ascontiguousarray(part); sidx = pi; break
    else:
        print("difference vanished when the sets were split at", rays.shape[0]); break
print("sets left:", rays.shape[0], "pixel", int(rays[0, 0]["pixel"]) & 0xFFFF, int(rays[0, 0]["pixel"]) >> 16, flush=True)
idx = np.arange(beams.n)
while idx.size > 1:
    h = idx.size // 2
    for part in (idx[:h], idx[h:]):
        d, o, _ = run(rays, beams.subset(part), np.ascontiguousarray(en[part]))
        if d != o:
            idx = part; break
    else:
        print("difference vanished when the beams were split at", idx.size); break
print("beams left:", idx, flush=True)
b = beams.subset(idx); e = np.ascontiguousarray(en[idx])
d, o, ex = run(rays, b, e)
print("pair: device", d, "oracle", o, "exact (taken, lost)", ex)
d2, o2, ex2 = run(rays, b, e, exact_all=True)
print("pair, everything through the exact pass: device", d2, "oracle", o2, ex2)
np.set_printoptions(precision=9, suppress=False)
for k in ("pos", "parent_pos", "parent_n", "parent_wi", "flags", "parent_pdf", "parent_g"):
    print(k, getattr(b, k))
print("end_n", e)
print("rays", rays[0])

dbg = os.path.join("build", "variants", "libgvpm_hip_dbg2.so")
if os.path.exists(dbg):
    import subprocess
    sys.stdout.flush()
    subprocess.run([sys.executable, __file__] + [a for a in sys.argv[1:] if a != "--pair"] + ["--pair", str(int(sidx[0])), str(int(idx[0]))],
                   env=dict(os.environ, GVPM_HIP_LIB=os.path.abspath(dbg)))
