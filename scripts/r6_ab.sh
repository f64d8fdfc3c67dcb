#!/bin/bash
# A/B of an environment knob on the default bench line, alternating runs on one box:
#   bash scripts/r6_ab.sh <tag> "<env A>" "<env B>" [rounds] [bench args...]
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; A=$2; B=$3; R=${4:-3}; shift 4
O=gpurun_out/$tag; mkdir -p $O
for i in $(seq 1 $R); do
  for v in A B; do
    if [ $v == A ]; then e="$A"; else e="$B"; fi
    env $e python3 bench.py --steps 32 --warmup 4 --only-timed "$@" 2> $O/err_$v$i.txt | tail -1 > $O/line_$v$i.json
    python3 - <<PY
import json
d=json.load(open("$O/line_$v$i.json"))
r=d.get("roofline",{})
print("$v$i [$e]", "ms/step %.4f" % d["ms_per_step"], "value %.1f" % d["value"], "kernel %.3f trav %.3f build %.3f" % (r.get("kernel_avg_ms",0), r.get("traverse_avg_ms",0), r.get("build_avg_ms",0)), "evals", d.get("stats",{}).get("evaluations"))
PY
  done
done
