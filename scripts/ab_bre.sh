#!/bin/bash
# A/B of G-BRE variants on the GPU box: bash scripts/ab_bre.sh OUTDIR
out=${1:-gpurun_out/ab}; mkdir -p $out
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-parity --no-upload-inclusive --distinct 4 --steps 16 --warmup 3 2> $out/$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-28s %8.1f Mev/s  step %.3f ms  eval %.3f  iso %.3f  trav %.3f  build %.3f' % ('$name', d['value'], d['ms_per_step'], r['kernel_avg_ms'], r.get('kernel_isolated_ms',0), r['traverse_avg_ms'], r['build_avg_ms']))" | tee -a $out/summary.txt
}
run default A=1
run default_single GVPM_PIPELINE=0
for e in $ENVS; do
  run single_$e GVPM_PIPELINE=0 $e
  run pipe_$e $e
done
IFS=';' read -ra CB <<< "$COMBOS"
for c in "${CB[@]}"; do
  [ -z "$c" ] && continue
  n=$(echo $c | sed 's#[^ ]*libgvpm_hip_##g' | tr ' =/' '___')
  run pipe_$n $c
  run single_$n GVPM_PIPELINE=0 $c
done
for v in $VARIANTS; do
  run $v GVPM_HIP_LIB=$PWD/build/variants/libgvpm_hip_$v.so
  run ${v}_single GVPM_HIP_LIB=$PWD/build/variants/libgvpm_hip_$v.so GVPM_PIPELINE=0
done
