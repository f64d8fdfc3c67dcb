#!/bin/bash
# rocprofv3 summaries of the four single-GPU bench lines -> gpurun_out/prof_r03_<wl>/ (copied to profiles/ by hand)
bash scripts/profile.sh r03_c2 > /dev/null 2>&1
bash scripts/profile.sh r03_c3 --workload c3 --steps 8 > /dev/null 2>&1
bash scripts/profile.sh r03_c1 --workload c1 > /dev/null 2>&1
bash scripts/profile.sh r03_c5 --workload c5 > /dev/null 2>&1
for w in c2 c3 c1 c5; do echo "== $w"; head -12 gpurun_out/prof_r03_$w/summary.txt | cut -c1-200; tail -6 gpurun_out/prof_r03_$w/summary.txt | cut -c1-200; done
python bench.py --no-cpu-baseline --no-parity --no-isolated 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('c2 step', d['ms_per_step'], 'upload_inclusive', d['upload_inclusive'])"
