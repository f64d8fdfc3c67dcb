#!/bin/bash
# copy what scripts/r03_final.sh left under gpurun_out/ into profiles/ (tracked): per bench line the rocprofv3 kernel
# statistics (pipelined and single stream), the PMC summary, the traffic file bench.py looks for, the line itself
set -e
cd "$(dirname "$0")/.."
for w in c1 c2 c3 c5; do
  d=gpurun_out/prof_r03_$w
  [ -d $d ] || { echo "missing $d"; continue; }
  cp $d/kernel_stats.csv profiles/r03_${w}_kernel_stats.csv
  cp $d/kernel_stats_single_stream.csv profiles/r03_${w}_kernel_stats_single_stream.csv
  cp $d/summary.txt profiles/r03_${w}_summary.txt
  if [ $w == c2 ]; then cp $d/traffic.json profiles/r03_traffic.json; else cp $d/traffic.json profiles/r03_traffic_$w.json; fi
  cp $d/bench_line.json profiles/r03_${w}_profiled_bench_line.json
  [ -f gpurun_out/r03_lines/bench_$w.json ] && grep '^{' gpurun_out/r03_lines/bench_$w.json | tail -1 > profiles/r03_${w}_bench_line.json
done
cp gpurun_out/r03_lines.log profiles/r03_bench_lines.txt
cp gpurun_out/r03_shard_probe.txt profiles/r03_shard_probe.txt
[ -f gpurun_out/r03_shard_blocks.txt ] && cp gpurun_out/r03_shard_blocks.txt profiles/r03_shard_blocks.txt
[ -f gpurun_out/pmc_beams3.txt ] && cp gpurun_out/pmc_beams3.txt profiles/r03_c3_pmc.txt
ls -la profiles/ | grep r03
