#!/bin/bash
# bash scripts/collect_profiles.sh <round tag, e.g. r04> [workloads: c2 c1 c3 c5]  (on the GPU box, from the repo root)
# For every workload: scripts/profile.sh (kernel statistics pipelined + single stream, four PMC passes, traffic), then the
# plain bench line with the fresh traffic file in place -> gpurun_out/<round>/, named as profiles/ keeps them.
R=$1; shift
WL=${*:-c2 c1 c3 c5}
D=$PWD/gpurun_out/$R
mkdir -p $D
for w in $WL; do
  if [ $w == c2 ]; then args=""; tj=${R}_traffic.json; else args="--workload $w"; tj=${R}_traffic_$w.json; fi
  bash scripts/profile.sh ${R}_$w $args > $D/${R}_${w}_profile.log 2>&1
  P=$PWD/gpurun_out/prof_${R}_$w
  cp $P/kernel_stats.csv $D/${R}_${w}_kernel_stats.csv
  cp $P/kernel_stats_single_stream.csv $D/${R}_${w}_kernel_stats_single_stream.csv
  cp $P/summary.txt $D/${R}_${w}_summary.txt
  cp $P/bench_line.json $D/${R}_${w}_profiled_bench_line.json
  cp $P/traffic.json $D/$tj
  cp $P/traffic.json profiles/$tj   # (this box's copy: the bench line below reports `traffic` from it)
  python3 bench.py $args > $D/${R}_${w}_bench_line.json 2> $D/${R}_${w}_bench.err
  tail -c 600 $D/${R}_${w}_bench_line.json
  rm -rf $P/trace $P/trace1 $P/pmc1 $P/pmc2 $P/pmc3 $P/pmc4
done
