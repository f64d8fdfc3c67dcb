#!/bin/bash
# A/B of gather_beams variants at C3: bash scripts/r03_ab.sh v1 v2 ...
for v in default "$@"; do
  if [ $v == default ]; then unset GVPM_HIP_LIB; else export GVPM_HIP_LIB=$PWD/build/variants/libgvpm_hip_$v.so; fi
  for t in ${TECHS:-3d}; do
    echo -n "$v $t: "; python scripts/beams_bench.py --scene laser --size 512 --beams 2000000 --iters 4 --tech $t 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('kernel_ms', d['kernel_ms'], 'ms_per_iter', d['ms_per_iter'], 'evals', d['per_iter']['evaluations'], 'diff', d['per_iter']['diffuse_shifts'], 'fail', d['per_iter']['failed_shifts'])"
  done
done
