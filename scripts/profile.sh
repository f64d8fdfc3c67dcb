#!/bin/bash
# Usage (on the GPU box, from the repo root): bash scripts/profile.sh <tag> [bench args...]
# Writes rocprofv3 kernel-trace stats and PMC passes under gpurun_out/prof_<tag>/ .
set -u
TAG=${1:-run}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 4 --warmup 1 --distinct 2 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
# PMC passes (own runs, no tracing domains besides kernel-trace)
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU -d $OUT/pmc1 -o pmc -- python3 bench.py $ARGS > $OUT/bench_pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA -d $OUT/pmc2 -o pmc -- python3 bench.py $ARGS > $OUT/bench_pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc -- python3 bench.py $ARGS > $OUT/bench_pmc3.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc4 -o pmc -- python3 bench.py $ARGS > $OUT/bench_pmc4.log 2>&1
python3 scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
