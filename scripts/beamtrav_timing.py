"""Probe: the persistent waves of traverse_beams_kernel in time (last launch; needs the variant
  bash scripts/build_variant.sh ttiming gather_beams.hip -DGVPM_TRAV_TIMING).  python scripts/beamtrav_timing.py [beams_bench args]"""
import ctypes, os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "build", "variants", "libgvpm_hip_ttiming.so")
os.environ["GVPM_HIP_LIB"] = lib
sys.argv = ["beams_bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "scripts", "beams_bench.py"), run_name="__main__")
import numpy as np
h = ctypes.CDLL(lib)
out = (ctypes.c_ulonglong * (4 * 8192))()
h.gvpm_debug_beamtrav_timing(out)
log = np.array(out[:], dtype=np.float64).reshape(-1, 4)
log = log[log[:, 1] > 0]
t0 = log[:, 0].min()
st, en = (log[:, 0] - t0) / 100.0, (log[:, 1] - t0) / 100.0
print("waves %d; span %.0f us; start us p50 %.0f max %.0f; end us: min %.0f p10 %.0f p50 %.0f p90 %.0f max %.0f" % (
    len(log), en.max(), np.percentile(st, 50), st.max(), en.min(), *np.percentile(en, [10, 50, 90, 100])))
print("items per wave: min %d mean %.1f max %d; candidates per wave: min %.0f mean %.0f max %.0f" % (
    log[:, 2].min(), log[:, 2].mean(), log[:, 2].max(), log[:, 3].min(), log[:, 3].mean(), log[:, 3].max()))
h_, e_ = np.histogram(en, bins=10)
print("end-time histogram (us):", [int(x) for x in e_], list(h_))
