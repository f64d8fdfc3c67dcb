#!/bin/bash
# A/B of G-BRE variants on the GPU box: bash scripts/ab_bre.sh OUTDIR
out=${1:-gpurun_out/ab}; mkdir -p $out
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --no-cpu-baseline --distinct 4 --steps 16 --warmup 3 2> $out/$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-28s %8.1f Mev/s  step %.3f ms  eval %.3f  iso %.3f  trav %.3f  build %.3f' % ('$name', d['value'], d['ms_per_step'], r['kernel_avg_ms'], r.get('kernel_isolated_ms',0), r['traverse_avg_ms'], r['build_avg_ms']))" | tee -a $out/summary.txt
}
run mix_default A=1
run both_pers GVPM_PERSISTENT=3
run mix_w10 GVPM_WAVES_PER_CU=10
run mix_w12 GVPM_WAVES_PER_CU=12
run mix_prio000 GVPM_STREAM_PRIORITIES=0,0,0
run mix_buildhigh GVPM_STREAM_PRIORITIES=0,-1,0
run mix_default_b A=1
