#!/bin/bash
# bundle grid: parity tests, then A/B of the C2 step
mkdir -p gpurun_out/r03_bundle
timeout 900 python -m pytest tests/test_bundle_grid_gpu.py tests/test_parity_gpu.py -x -q -m gpu > gpurun_out/r03_bundle/pytest.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r03_bundle/pytest.log
for b in 0 1; do
  GVPM_BUNDLE=$b GVPM_TRACE_PLAN=1 timeout 300 python bench.py --no-cpu-baseline --no-parity --no-upload-inclusive --steps 2 --warmup 1 2>&1 | grep "\[plan\]" | tail -1
  GVPM_BUNDLE=$b timeout 300 python bench.py --no-cpu-baseline --no-parity --no-upload-inclusive --steps 32 --warmup 4 > gpurun_out/r03_bundle/c2_b$b.json 2> gpurun_out/r03_bundle/c2_b$b.err
  python - <<PY
import json
try:
    d = json.loads(open("gpurun_out/r03_bundle/c2_b$b.json").read().strip().splitlines()[-1])
    print("bundle=$b", d["value"], d["ms_per_step"], d.get("phases"), d["roofline"]["achieved"])
except Exception as e:
    print("bundle=$b failed", e); print(open("gpurun_out/r03_bundle/c2_b$b.err").read()[-1500:])
PY
done
