#!/bin/bash
# Usage (on the GPU box, from the repo root): bash scripts/profile.sh <tag> [bench args...]
# rocprofv3 kernel-trace stats (pipelined and single-stream) and PMC passes (own runs, --kernel-trace only)
# under gpurun_out/prof_<tag>/ ; summary.txt and traffic.json are what gets copied to profiles/.
set -u
TAG=${1:-run}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 16 --warmup 2 --only-timed $*"   # nothing but the warm-up and the timed steps runs on the GPU
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py $ARGS > $OUT/bench_trace.log 2>&1
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats.csv; done
GVPM_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -o trace -- python3 bench.py $ARGS > $OUT/bench_trace1.log 2>&1
for f in $(find $OUT/trace1 -name "*kernel_stats.csv"); do cp $f $OUT/kernel_stats_single_stream.csv; done
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU -d $OUT/pmc1 -o pmc -- python3 bench.py $ARGS > $OUT/bench_pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA -d $OUT/pmc2 -o pmc -- python3 bench.py $ARGS > $OUT/bench_pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc -- python3 bench.py $ARGS > $OUT/bench_pmc3.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc4 -o pmc -- python3 bench.py $ARGS > $OUT/bench_pmc4.log 2>&1
python3 scripts/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
grep "^{" $OUT/bench_trace.log | tail -1 > $OUT/bench_line.json
cat $OUT/summary.txt | head -80
