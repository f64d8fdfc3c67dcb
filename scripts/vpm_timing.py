"""Probe: the waves of gather_vpm_kernel in time (last launch; needs the variant
  bash scripts/build_variant.sh vtiming gather_vpm.hip -DGVPM_VPM_TIMING).  python scripts/vpm_timing.py"""
import ctypes, os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "build", "variants", "libgvpm_hip_vtiming.so")
os.environ["GVPM_HIP_LIB"] = lib
sys.argv = ["vpm_probe.py", "3"] + sys.argv[1:2]
runpy.run_path(os.path.join(ROOT, "scripts", "vpm_probe.py"), run_name="__main__")
import numpy as np
h = ctypes.CDLL(lib)
out = (ctypes.c_ulonglong * (4 * 65536))()
h.gvpm_debug_vpm_timing(out)
log = np.array(out[:], dtype=np.float64).reshape(-1, 4)
log = log[log[:, 1] > 0]
log = log[log[:, 0] > log[:, 0].max() - 1e5]  # (rows of the last launch only: a millisecond back from the latest start)
t0 = log[:, 0].min()
st, en = (log[:, 0] - t0) / 100.0, (log[:, 1] - t0) / 100.0  # microseconds
life = en - st
print("waves %d; kernel span %.1f us; lifetime us: mean %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f" % (
    len(log), en.max(), life.mean(), np.percentile(life, 50), np.percentile(life, 90), np.percentile(life, 99), life.max()))
print("start us: p10 %.1f p50 %.1f p90 %.1f max %.1f" % tuple(np.percentile(st, [10, 50, 90, 100])))
print("candidates per wave: mean %.0f p50 %.0f p99 %.0f max %.0f; evaluations per wave mean %.1f max %.0f" % (
    log[:, 2].mean(), np.percentile(log[:, 2], 50), np.percentile(log[:, 2], 99), log[:, 2].max(), log[:, 3].mean(), log[:, 3].max()))
# concurrency over time
edges = np.linspace(0, en.max(), 21)
for a, b in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (a + b)
    print("t %6.1f us: resident %5d  started so far %6d" % (mid, int(((st <= mid) & (en > mid)).sum()), int((st <= mid).sum())))
order = np.argsort(log[:, 0])
wid = np.flatnonzero(np.ones(len(log)))
print("correlation of lifetime with candidates: %.3f" % np.corrcoef(life, log[:, 2])[0, 1])
