#!/bin/bash
for v in default "$@"; do
  if [ $v == default ]; then unset GVPM_HIP_LIB; else export GVPM_HIP_LIB=$PWD/build/variants/libgvpm_hip_$v.so; fi
  echo -n "$v: "; python bench.py --workload c1 --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('c1 kernel %.3f' % r['kernel_avg_ms'], end='  ')"
  python bench.py --workload c1 --scale 4.0 --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('scale 4: kernel %.3f  %.0f Mev/s' % (r['kernel_avg_ms'], d['value']))"
done
