#!/bin/bash
# one bench line (32 timed steps) per environment setting, in turn, ROUNDS times: bash scripts/r6_sweep.sh <tag> <rounds> "<env 1>" "<env 2>" ...
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; R=$2; shift 2
O=gpurun_out/$tag; mkdir -p $O
for i in $(seq 1 $R); do
  k=0
  for e in "$@"; do
    k=$((k+1))
    env $e python3 bench.py --steps 32 --warmup 4 --only-timed $BENCH_ARGS 2> $O/err_$k_$i.txt | tail -1 > $O/line_${k}_$i.json
    python3 - <<PY
import json
d=json.load(open("$O/line_${k}_$i.json"))
r=d.get("roofline",{})
print("r$i [$e]", "ms/step %.4f" % d["ms_per_step"], "value %.1f" % d["value"], "kernel %.3f trav %.3f build %.3f" % (r.get("kernel_avg_ms",0), r.get("traverse_avg_ms",0), r.get("build_avg_ms",0)), "evals", d.get("stats",{}).get("evaluations"))
PY
  done
done
