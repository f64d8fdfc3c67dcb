#!/bin/bash
mkdir -p gpurun_out/r03_hs
timeout 900 python -m pytest tests/test_host_shifts_gpu.py tests/test_parity_gpu.py -x -q -m gpu > gpurun_out/r03_hs/pytest.log 2>&1
echo "pytest rc=$?"; tail -25 gpurun_out/r03_hs/pytest.log
timeout 300 python bench.py --no-cpu-baseline --no-parity --no-upload-inclusive --steps 32 --warmup 4 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 step', d['ms_per_step'], d['value'], d['roofline']['kernel_avg_ms'], d['roofline'].get('kernel_isolated_ms'))"
