#!/bin/bash
# bash scripts/ab_variants.sh <rounds> VARIANT...: alternating C2 lines with library variants (default = the in-tree library)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    if [ $v == default ]; then unset GVPM_HIP_LIB; else export GVPM_HIP_LIB=$PWD/build/variants/libgvpm_hip_$v.so; fi
    echo -n "round $r $v: "
    python bench.py --no-cpu-baseline --no-parity --no-upload-inclusive --no-isolated $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%.4f ms/step  %.0f Mev/s  kernel %.4f ms' % (d['ms_per_step'], d['value'], r['kernel_avg_ms']))"
  done
done
