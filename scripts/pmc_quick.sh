#!/bin/bash
# bash scripts/pmc_quick.sh <tag> [bench args]: kernel stats + two SQ counter passes for the current env
set -u
TAG=${1:-run}; shift || true
OUT=$PWD/gpurun_out/pq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 4 --warmup 1 --distinct 2 --no-cpu-baseline $*"
python3 bench.py $ARGS 2>/dev/null | tail -1 > $OUT/bench.json
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU -d $OUT/pmc1 -o pmc -- python3 bench.py $ARGS > $OUT/bench_pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA -d $OUT/pmc2 -o pmc -- python3 bench.py $ARGS > $OUT/bench_pmc2.log 2>&1
python3 scripts/pmc_summary.py $OUT 2>&1 | grep -A9 "evaluate_bre\|traverse_bre\|plan_kernel" > $OUT/summary.txt
echo "== $TAG: $(cat $OUT/bench.json | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"].get("kernel_avg_ms"), d.get("stats"))')"
cat $OUT/summary.txt
