#!/bin/bash
# bash scripts/ab_lines.sh "<bench.py args>" <rounds> default VARIANT...: alternating runs of a bench line on ONE box with the
# default library and with build/variants/libgvpm_hip_VARIANT.so (scripts/build_variant.sh): ms/step and the kernel's duration
args=$1; rounds=$2; shift; shift
for r in $(seq 1 $rounds); do
  for v in "$@"; do
    if [ $v == default ]; then unset GVPM_HIP_LIB; else export GVPM_HIP_LIB=$PWD/build/variants/libgvpm_hip_$v.so; fi
    echo -n "round $r $v: "
    python bench.py $args --no-cpu-baseline --no-parity --no-upload-inclusive --no-isolated 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%.4f ms/step  %.0f Mev/s  kernel %.4f ms  traverse %s  build %s' % (d['ms_per_step'], d['value'], r['kernel_avg_ms'], r.get('traverse_avg_ms'), r.get('build_avg_ms')))"
  done
done
