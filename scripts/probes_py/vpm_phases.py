"""Probe: where a wave of vpm_find_kernel spends its life (last launch; variant
  bash scripts/build_variant.sh vtiming2 gather_vpm.hip -DGVPM_VPM_TIMING -DGVPM_VPM_TIMING2).  python scripts/probes_py/vpm_phases.py"""
import ctypes, os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = os.path.join(ROOT, "build", "variants", "libgvpm_hip_vtiming2.so")
os.environ["GVPM_HIP_LIB"] = lib
sys.argv = ["vpm_probe.py", "3"] + sys.argv[1:2]
runpy.run_path(os.path.join(ROOT, "scripts", "vpm_probe.py"), run_name="__main__")
import numpy as np
h = ctypes.CDLL(lib)
out = (ctypes.c_ulonglong * (4 * 65536))()
h.gvpm_debug_vpm_timing(out)
log = np.array(out[:], dtype=np.uint64).reshape(-1, 4)
ev = log[32768:]
log = log[:32768]
log = log[log[:, 1] > 0]
log = log[log[:, 0] > log[:, 0].max() - np.uint64(100000)]
st = log[:, 0].astype(np.float64)
life = (log[:, 1].astype(np.float64) - st) / 100.0
setup = (log[:, 2].astype(np.float64) - st) / 100.0
rows = (log[:, 3] & np.uint64(0xFFFFFFFF)).astype(np.float64) / 100.0
trips = (log[:, 3] >> np.uint64(32)).astype(np.float64) / 100.0
ok = rows > 0
for name, v in (("life", life), ("to end of set-up", setup), ("to end of the first pass's row ranges + scan", rows[ok]), ("to end of the trips", trips)):
    print("%-48s mean %6.2f p50 %6.2f p90 %6.2f us" % (name, v.mean(), np.percentile(v, 50), np.percentile(v, 90)))

# the evaluation's waves
ev = ev[ev[:, 1] > 0]
ev = ev[ev[:, 0] > ev[:, 0].max() - np.uint64(100000)]
life = (ev[:, 1] - ev[:, 0]).astype(np.float64) / 100.0
load = (ev[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.float64) / 100.0
p1 = (ev[:, 2] >> np.uint64(32)).astype(np.float64) / 100.0
p2 = (ev[:, 3] & np.uint64(0xFFFFF)).astype(np.float64) / 100.0
fl = ((ev[:, 3] >> np.uint64(20)) & np.uint64(0xFFFFF)).astype(np.float64) / 100.0
nb = (ev[:, 3] >> np.uint64(40)).astype(np.float64)
act = nb > 0
span = (ev[:, 1].max() - ev[:, 0].min()) / 100.0
print("evaluation: %d waves logged, %d with batches; span %.1f us; batches %d (mean %.2f a wave, max %d)" % (len(ev), act.sum(), span, nb.sum(), nb[act].mean(), nb.max()))
print("  life of a wave with batches: mean %.1f p50 %.1f p90 %.1f max %.1f us" % (life[act].mean(), np.percentile(life[act], 50), np.percentile(life[act], 90), life[act].max()))
tot = nb.sum()
print("  per batch: load %.2f  phase 1 %.2f  phase 2 %.2f  flush %.2f us (sum %.2f)" % (load.sum() / tot, p1.sum() / tot, p2.sum() / tot, fl.sum() / tot, (load + p1 + p2 + fl).sum() / tot))
print("  wave-time: %.1f ms*wave in batches, %.1f outside" % ((load + p1 + p2 + fl).sum() / 1e3, (life.sum() - (load + p1 + p2 + fl).sum()) / 1e3))
