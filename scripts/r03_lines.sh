#!/bin/bash
# the four single-GPU bench lines (BASELINE configs[0], [1], [2], [4]) as the driver would run them
out=gpurun_out/r03_lines; mkdir -p $out
for wl in c1 c5 c3; do
  SECONDS=0; python bench.py --workload $wl > $out/bench_$wl.json 2> $out/bench_$wl.err
  echo "$wl wall $SECONDS s"; tail -1 $out/bench_$wl.err
  python - $out/bench_$wl.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]; c=d["cpu_baseline"]
print("%s | %.0f Mevals/s  %.3f ms/step | kernel %s %.3f ms frac %.3f (trav %.3f build %.3f) | parity_l2 %.2e evals %d/%d | cpu %.2f Mevals/s on %d cores (build %.2f s gather %.2f s)" % (
  d["config"]["workload"][:40], d["value"], d["ms_per_step"], r["kernel"], r["kernel_avg_ms"], r["frac"], r["traverse_avg_ms"], r["build_avg_ms"],
  d["parity_l2"], d["parity"]["evaluations_device"], d["parity"]["evaluations_oracle"], c["value"], c["cores"], c["build_s"], c["gather_s"]))
PY
done
SECONDS=0; python bench.py > $out/bench_c2.json 2> $out/bench_c2.err; echo "c2 wall $SECONDS s"; tail -1 $out/bench_c2.err
python - $out/bench_c2.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r=d["roofline"]; c=d["cpu_baseline"]
print("C2 | %.0f Mevals/s  %.3f ms/step | eval %.3f ms (alone %.3f) frac %.3f trav %.3f build %.3f | upload-inclusive %.3f ms | cpu %.2f (build %.2f s gather %.2f s)" % (
  d["value"], d["ms_per_step"], r["kernel_avg_ms"], r.get("kernel_isolated_ms",0), r["frac"], r["traverse_avg_ms"], r["build_avg_ms"], d["upload_inclusive"]["ms_per_step"], c["value"], c["build_s"], c["gather_s"]))
PY
