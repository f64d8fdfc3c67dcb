"""Find the (photon, camera sample) pair on which the device's G-VPM shift counters differ from the oracle's: the iteration first,
then bisection over the photons, then over the samples.  python scripts/probes_py/vpm_bisect.py scene scale nb iters key=value...  (GPU box)"""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from gvpm_amd import abi, hip
from test_oracle_vpm import make_vpm_case
import cases, oracle_lib as O
scene, scale, nb, iters = sys.argv[1], float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
kw = {a.split("=")[0]: int(a.split("=")[1]) for a in sys.argv[5:] if "=" in a}
c = make_vpm_case(scene, 36, 30, 30000, scale, nb=nb, it=int(os.environ.get("STRESS_IT", "1")), **kw)
KEYS = ("evaluations", "null_shifts", "diffuse_shifts", "failed_shifts")
inputs = {1: (c.ph, c.nb, c.rays, c.samples)}
for it in range(2, iters + 1):
    ph, nbp = c.sc.shoot_photons(it, c.ph.n)
    r, smp = c.sc.camera_beams_and_vpm_samples(it, c.p.nb_camera_samples)
    inputs[it] = (ph, nbp, r, smp)
state = {1: (None, None)}
sv = nv = None
for it in range(1, iters):
    ph, nbp, r, smp = inputs[it]
    _, sv, nv, _, _ = O.gather_vpm(c.p, c.m, c.tris, ph, r, smp, 64, use_accel=False, scale_vol=sv, n_vol=nv)
    state[it + 1] = (sv.copy(), nv.copy())

def run(target, ph, smp, exact_all=False):
    if exact_all: os.environ["GVPM_EXACT_ALL"] = "1"
    ctx = hip.Context(c.p, device=0)
    os.environ.pop("GVPM_EXACT_ALL", None)
    ctx.upload_scene(*c.tris); ctx.upload_medium(c.m); cases.upload_bsdfs(ctx, c)
    before = {k: 0 for k in KEYS}
    for it in range(1, target + 1):
        p0, nbp, r, s0 = inputs[it]
        if it == target:
            before = dict(ctx.stats()) if it > 1 else before
            p0, s0 = ph, smp
        ctx.upload_photons(p0); ctx.upload_camera_beams(r); ctx.upload_vpm_samples(s0)
        ctx.gather(it, nbp)
    st = ctx.stats(); ex = ctx.exact_shifts()
    ctx.close()
    sv0, nv0 = state[target]
    _, _, _, cnt, _ = O.gather_vpm(c.p, c.m, c.tris, ph, inputs[target][2], smp, 64, use_accel=False, scale_vol=sv0, n_vol=nv0)
    return tuple(st[k] - before[k] for k in KEYS), tuple(cnt[k] for k in KEYS), ex

if "--pair" in sys.argv:  # (second stage, under a probe library: the one pair, with the library's prints)
    k = sys.argv.index("--pair"); target, pi, si = int(sys.argv[k + 1]), int(sys.argv[k + 2]), int(sys.argv[k + 3])
    print("pair", run(target, inputs[target][0].subset(np.array([pi])), np.ascontiguousarray(inputs[target][3][si:si + 1])))
    sys.exit(0)
target = None
for it in range(1, iters + 1):
    d, o, ex = run(it, inputs[it][0], inputs[it][3])
    print("iteration", it, d, o, ex, flush=True)
    if d != o and target is None: target = it
if target is None: sys.exit(0)
ph, smp = inputs[target][0], inputs[target][3]
idx = np.arange(ph.n)
while idx.size > 1:
    h = idx.size // 2
    for part in (idx[:h], idx[h:]):
        d, o, _ = run(target, ph.subset(part), smp)
        if d != o:
            idx = part; break
    else:
        print("difference vanished when the photons were split at", idx.size); break
print("photons left:", idx, flush=True)
one = ph.subset(idx)
sidx = np.arange(smp.shape[0])
while sidx.size > 1:
    h = sidx.size // 2
    for part in (sidx[:h], sidx[h:]):
        d, o, _ = run(target, one, np.ascontiguousarray(smp[part]))
        if d != o:
            sidx = part; break
    else:
        print("difference vanished when the samples were split at", sidx.size); break
print("samples left:", sidx, smp[sidx], flush=True)
s1 = np.ascontiguousarray(smp[sidx])
print("pair: device / oracle / exact", run(target, one, s1))
print("pair, everything through the exact pass:", run(target, one, s1, exact_all=True))
np.set_printoptions(precision=9)
for k in ("pos", "parent_pos", "parent_n", "parent_wi", "flags", "parent_pdf", "parent_g", "wi"):
    print(k, getattr(one, k))
print("set", inputs[target][2][int(s1[0]["set"])])

dbg = os.path.join("build", "variants", "libgvpm_hip_dbgv.so")
if os.path.exists(dbg):
    import subprocess
    sys.stdout.flush()
    subprocess.run([sys.executable, __file__] + [a for a in sys.argv[1:] if a != "--pair"] + ["--pair", str(target), str(int(idx[0])), str(int(sidx[0]))],
                   env=dict(os.environ, GVPM_HIP_LIB=os.path.abspath(dbg)))
