"""Probe: where the waves of evaluate_beams2_kernel spend their time (last launch; needs the variant
  bash scripts/build_variant.sh btiming gather_beams.hip -DGVPM_EVAL_TIMING).  python scripts/beams_timing.py [beams_bench args]"""
import ctypes
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "build", "variants", "libgvpm_hip_btiming.so")
os.environ["GVPM_HIP_LIB"] = lib
sys.argv = ["beams_bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "scripts", "beams_bench.py"), run_name="__main__")
import numpy as np  # noqa: E402

h = ctypes.CDLL(lib)
out = (ctypes.c_ulonglong * (8 * 16384))()
h.gvpm_debug_beams_timing(out)
log = np.array(out[:], dtype=np.float64).reshape(-1, 8)
log = log[log[:, 7] > 0]
tot = log[:, 7].sum()
names = ["beamBase (block load, filters, kernel record, base)", "4 x beamShift1 + queue push", "phase 2 (drain)", "tile change (flush, rays)"]
for k, n in enumerate(names):
    print("%-52s %5.1f %%" % (n, 100 * log[:, k].sum() / tot))
print("waves %d, blocks %d, pairs alive after beamBase %.1f %% of the block slots, reconnections per block %.1f" % (
    len(log), log[:, 4].sum(), 100 * log[:, 5].sum() / (64 * log[:, 4].sum()), log[:, 6].sum() / log[:, 4].sum()))
print("wave lifetime ticks: mean %.0f max %.0f" % (log[:, 7].mean(), log[:, 7].max()))
