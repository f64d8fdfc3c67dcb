"""Probe: where the waves of evaluate_bre_kernel spend their time (shader-clock ticks summed over waves).
Needs the -DGVPM_EVAL_TIMING variant: bash scripts/build_variant.sh timing gather_bre.hip -DGVPM_EVAL_TIMING
  python scripts/eval_timing.py [bench args]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "build", "variants", "libgvpm_hip_timing.so")
os.environ["GVPM_HIP_LIB"] = lib
os.environ.setdefault("GVPM_PIPELINE", "0")
sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--only-timed", "--steps", "8", "--warmup", "2"] + sys.argv[1:]
import bench  # noqa: E402

bench.main()
h = ctypes.CDLL(lib)
out = (ctypes.c_ulonglong * (16 + 16 * 16384))()
h.gvpm_debug_eval_timing(out, 0)
names = ["wave lifetime", "item header + LDS setup", "decision + phase 1", "phase 2", "late pass", "write-out", "items", "p1: issue loads", "p1: flush on beam change", "p1: wait record + decide", "p1: evalPhase1 + queue", "p1 steps", "longest wave (max over launches)", "longest item", "waves"]
tot = out[0] or 1
for n, v in zip(names, out):
    print("%-26s %16d  %5.1f %%" % (n, v, 100.0 * v / tot))
print("ticks per item: %.0f; mean wave lifetime %.0f, longest %d, longest item %d" % (out[0] / max(out[6], 1), out[0] / max(out[14], 1), out[12], out[13]))

import numpy as np  # noqa: E402
log = np.array(out[16:], dtype=np.uint64).reshape(-1, 16)
log = log[log[:, 0] > 0].astype(np.float64)
t0 = log[:, 0].min()
st, en = (log[:, 0] - t0) / 100.0, (log[:, 1] - t0) / 100.0  # microseconds
print("last launch: %d waves; start us: mean %.1f p50 %.1f p90 %.1f max %.1f; end us: min %.1f p10 %.1f p50 %.1f mean %.1f max %.1f" % (
    len(log), st.mean(), np.percentile(st, 50), np.percentile(st, 90), st.max(), en.min(), np.percentile(en, 10),
    np.percentile(en, 50), en.mean(), en.max()))
print("units per wave: min %d mean %.1f max %d" % (log[:, 2].min(), log[:, 2].mean(), log[:, 2].max()))
h, e = np.histogram(en, bins=10)
print("end-time histogram (us):", [int(x) for x in e], list(h))
h, e = np.histogram(st, bins=10)
print("start-time histogram (us):", [int(x) for x in e], list(h))

ls = (log[:, 3] - t0) / 100.0
dur = en - ls
print("last unit of each wave: start us p10 %.1f p50 %.1f p90 %.1f max %.1f; duration us p10 %.1f p50 %.1f p90 %.1f max %.1f; pairs p10 %d p50 %d p90 %d max %d" % (
    np.percentile(ls, 10), np.percentile(ls, 50), np.percentile(ls, 90), ls.max(), np.percentile(dur, 10), np.percentile(dur, 50),
    np.percentile(dur, 90), dur.max(), np.percentile(log[:, 4], 10), np.percentile(log[:, 4], 50), np.percentile(log[:, 4], 90), log[:, 4].max()))
o = np.argsort(en)[-8:]
for i in o:
    print("  wave %5d: units %2d, last unit #%d of n=%d pairs started %.1f us, took %.1f us (to unit end %.1f); ticks setup %d p1 %d p2 %d late %d wout %d" % (
        i, log[i, 2], log[i, 5], log[i, 4], ls[i], dur[i], (log[i, 11] - t0) / 100.0 - ls[i], log[i, 6], log[i, 7], log[i, 8], log[i, 9], log[i, 10]))
    print("        setup parts (ticks): pop+header %d, itemOff+boff %d, rays %d, zero+relToBase %d" % (log[i, 12], log[i, 13], log[i, 14], log[i, 15]))
print("corr(duration, pairs) of last units: %.2f" % np.corrcoef(dur, log[:, 4])[0, 1])
