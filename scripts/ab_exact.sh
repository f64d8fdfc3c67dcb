#!/bin/bash
# A/B of the exact pass's cadence and of library variants on ONE box: bash scripts/ab_exact.sh <rounds>
rounds=${1:-2}
line() { python bench.py --no-cpu-baseline --no-parity --no-upload-inclusive --no-isolated 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%.4f ms/step  %.0f Mev/s  kernel %.4f ms' % (d['ms_per_step'], d['value'], r['kernel_avg_ms']))"; }
for r in $(seq 1 $rounds); do
  for v in r4 default; do
    if [ $v == default ]; then unset GVPM_HIP_LIB; else export GVPM_HIP_LIB=$PWD/build/variants/libgvpm_hip_$v.so; fi
    for e in 1 8 1000; do
      [ $v == r4 ] && [ $e != 8 ] && continue
      echo -n "round $r $v every=$e: "; GVPM_EXACT_EVERY=$e line
    done
  done
done
