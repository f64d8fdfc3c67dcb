"""Probe: audit of the fp32 kernel record of the G-Beams evaluation against the fp64 transcription, pair by pair
(needs the variant: bash scripts/build_variant.sh baudit gather_beams.hip -DGVPM_BEAMS_AUDIT).
python scripts/beams_audit.py [beams_bench args]"""
import ctypes
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "build", "variants", "libgvpm_hip_baudit.so")
os.environ["GVPM_HIP_LIB"] = lib
sys.argv = ["beams_bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "scripts", "beams_bench.py"), run_name="__main__")
import numpy as np  # noqa: E402

h = ctypes.CDLL(lib)
cnt = ctypes.c_uint(0)
log = (ctypes.c_float * (256 * 16))()
ratio = (ctypes.c_float * 8)()
assert h.gvpm_debug_beams_audit(ctypes.byref(cnt), log, ratio) == 0
print("sure fp32 decisions that differ from the fp64 transcription:", cnt.value)
L = np.array(log[:], np.float32).reshape(256, 16)
names = "id pix exact stage tN tN64 tF tF64 v v64 d2/r2 d2/r2_64 w w64 sin2 bandT".split()
for row in L[:min(cnt.value, 40)]:
    ids = row[:2].view(np.uint32)
    print("  beam %d sub %d pix (%d,%d) exact=%d stage=%d | " % (ids[0] & 0xFFFFFF, ids[0] >> 24, ids[1] & 0xFFFF, ids[1] >> 16, row[2], row[3]) +
          " ".join("%s=%.9g" % (n, v) for n, v in zip(names[4:], row[4:])))
print("largest |fp32 - fp64| / band over the accepted pairs: tN %.3g  tF %.3g  v %.3g  w %.3g  distSqr %.3g;  pdfKernel relative %.3g" % (
    ratio[0], ratio[5], ratio[1], ratio[2], ratio[4], ratio[3]))
