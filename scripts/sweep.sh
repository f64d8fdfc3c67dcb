#!/bin/bash
# bash scripts/sweep.sh "VAR=a,b,c" "VAR2=x,y" ...  -> runs bench.py for the cartesian product
run() {
  out=$(env "$@" timeout 300 python bench.py ${SWEEP_ARGS:---steps 4 --warmup 1 --distinct 2} --no-cpu-baseline 2>/dev/null | tail -1)
  echo "$* :: $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("%.0f Mev/s step %.2f ms eval %.2f (recon %.2f) trav %.2f build %.2f" % (d["value"], d["ms_per_step"], r["kernel_avg_ms"], r.get("reconnect_avg_ms",0), r["traverse_avg_ms"], r["build_avg_ms"]))')"
}
combos=("")
for spec in "$@"; do
  var=${spec%%=*}; vals=${spec#*=}
  new=()
  IFS=',' read -ra arr <<< "$vals"
  for c in "${combos[@]}"; do for v in "${arr[@]}"; do new+=("$c $var=$v"); done; done
  combos=("${new[@]}")
done
for c in "${combos[@]}"; do run $c; done
