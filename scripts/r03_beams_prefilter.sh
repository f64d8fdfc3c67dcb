#!/bin/bash
timeout 900 python -m pytest tests/test_parity_beams_gpu.py tests/test_configs_gpu.py -x -q -m gpu 2>&1 | tail -3
p() { python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']
print('$1 step %.2f ms  eval %.2f trav %.2f build %.2f  cand %.0f M/step' % (d['ms_per_step'], r['kernel_avg_ms'], r['traverse_avg_ms'], r['build_avg_ms'], d['stats']['candidates'] / 8e6))"; }
for i in 1 2; do
python bench.py --workload c3 --only-timed --steps 8 --warmup 2 2>/dev/null | p "prefilter  "
GVPM_TRAV_PREFILTER=0 python bench.py --workload c3 --only-timed --steps 8 --warmup 2 2>/dev/null | p "prefilter 0"
done
python bench.py --workload c3 --technique beams1d --only-timed --steps 8 --warmup 2 2>/dev/null | p "1d prefilter  "
GVPM_TRAV_PREFILTER=0 python bench.py --workload c3 --technique beams1d --only-timed --steps 8 --warmup 2 2>/dev/null | p "1d prefilter 0"
