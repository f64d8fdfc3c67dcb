#!/bin/bash
# bash ab_env.sh "<bench args>" rounds "ENV1=.." "ENV2=.." ...   ("-" = no extra env)
args=$1; rounds=$2; shift; shift
for r in $(seq 1 $rounds); do
  for e in "$@"; do
    echo -n "round $r [$e]: "
    if [ "$e" == "-" ]; then envs=""; else envs="$e"; fi
    env $envs python bench.py $args --no-cpu-baseline --no-parity --no-upload-inclusive --no-isolated 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%.4f ms/step  %.0f Mev/s  kernel %.4f ms  traverse %s  build %s' % (d['ms_per_step'], d['value'], r['kernel_avg_ms'], r.get('traverse_avg_ms'), r.get('build_avg_ms')))"
  done
done
