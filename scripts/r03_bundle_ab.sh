#!/bin/bash
# bundle grid: isolated phases (GVPM_PIPELINE=0) and pipelined step for a few settings
mkdir -p gpurun_out/r03_bundle
run() {  # label, env...
  label=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-parity --no-upload-inclusive --steps 32 --warmup 4 > gpurun_out/r03_bundle/ab_$label.json 2> gpurun_out/r03_bundle/ab_$label.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r03_bundle/ab_$label.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("%-22s step %.3f ms  eval %.3f trav %.3f build %.3f  eval_iso %.3f  cand %.0f M" % ("$label", d["ms_per_step"], r["kernel_avg_ms"], r["traverse_avg_ms"], r["build_avg_ms"], r.get("kernel_isolated_ms") or 0, d["stats"]["candidates"] / 32e6))
except Exception as e:
    print("$label failed", e); print(open("gpurun_out/r03_bundle/ab_$label.err").read()[-800:])
PY
}
run 3d_serial GVPM_BUNDLE=0 GVPM_PIPELINE=0
run b2_serial GVPM_BUNDLE=1 GVPM_PIPELINE=0
run b1_serial GVPM_BUNDLE=1 GVPM_BUNDLE_DIV=1 GVPM_PIPELINE=0
run b4_serial GVPM_BUNDLE=1 GVPM_BUNDLE_DIV=4 GVPM_PIPELINE=0
run b8_serial GVPM_BUNDLE=1 GVPM_BUNDLE_DIV=8 GVPM_PIPELINE=0
run 3d GVPM_BUNDLE=0
run b1 GVPM_BUNDLE=1 GVPM_BUNDLE_DIV=1
run b2 GVPM_BUNDLE=1
run b4 GVPM_BUNDLE=1 GVPM_BUNDLE_DIV=4
run b8 GVPM_BUNDLE=1 GVPM_BUNDLE_DIV=8
run b4_t2048 GVPM_BUNDLE=1 GVPM_BUNDLE_DIV=4 GVPM_PLAN_TARGET=2048
run b4_t8192 GVPM_BUNDLE=1 GVPM_BUNDLE_DIV=4 GVPM_PLAN_TARGET=8192
