#!/bin/bash
# bash scripts/shard_probe.sh OUTDIR [which...]: the single-GPU probes of DESIGN.md section 6 (C2, C4 whole, rank 0's share of the 8-GPU runs)
out=${1:-gpurun_out/shard}; mkdir -p $out; shift
which=${*:-c2 c2_weak8_rank0 c4_one_gpu c4_strong8_rank0 c4_strong2_rank0 c4_strong4_rank0}
run() { name=$1; shift
  python bench.py --no-cpu-baseline --no-parity --no-upload-inclusive ${ISO---no-isolated} --steps 8 --warmup 2 "$@" 2> $out/$name.err | grep '^{' > $out/$name.json
  python - $out/$name.json $name <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']; c=d['config']
print('%-22s %7.2f ms/step  %7.1f M evals/step/GPU  %7.0f M evals/s  eval %.2f (alone %.2f) trav %.2f build %.2f  | %s' % (sys.argv[2], d['ms_per_step'], c['evals_per_iter_per_gpu']/1e6, d['value'], r['kernel_avg_ms'], r.get('kernel_isolated_ms', 0), r['traverse_avg_ms'], r['build_avg_ms'], c['workload'][:60]))
PY
}
for w in $which; do
  case $w in
    c2) run c2 ;;
    c2_weak8_rank0) run c2_weak8_rank0 --workload c2 --weak --emulate-gpus 8 ;;
    c4_one_gpu) run c4_one_gpu --workload c4 ;;
    c4_strong8_rank0) run c4_strong8_rank0 --workload c4 --emulate-gpus 8 ;;
    c4_strong2_rank0) run c4_strong2_rank0 --workload c4 --emulate-gpus 2 ;;
    c4_strong4_rank0) run c4_strong4_rank0 --workload c4 --emulate-gpus 4 ;;
  esac
done
