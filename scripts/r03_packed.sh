#!/bin/bash
mkdir -p gpurun_out/r03_packed
timeout 900 python -m pytest tests/test_packed_upload_gpu.py tests/test_parity_gpu.py -x -q -m gpu > gpurun_out/r03_packed/pytest.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r03_packed/pytest.log
timeout 600 python bench.py --no-cpu-baseline --no-parity --steps 32 --warmup 4 > gpurun_out/r03_packed/c2.json 2> gpurun_out/r03_packed/c2.err
python - <<'PY'
import json
try:
    d = json.loads(open("gpurun_out/r03_packed/c2.json").read().strip().splitlines()[-1])
    print("step", d["ms_per_step"], "value", d["value"]); print(json.dumps(d["upload_inclusive"], indent=1))
except Exception as e:
    print("failed", e); print(open("gpurun_out/r03_packed/c2.err").read()[-2000:])
PY
