#!/bin/bash
# bash scripts/pmc_any.sh <tag> <kernel-name filter> <script> [args]: three SQ counter passes of one python script
set -u
TAG=$1; FILT=$2; shift; shift
OUT=$PWD/gpurun_out/pa_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU -d $OUT/pmc1 -o pmc -- python3 "$@" > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA -d $OUT/pmc2 -o pmc -- python3 "$@" > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_GDS -d $OUT/pmc3 -o pmc -- python3 "$@" > $OUT/pmc3.log 2>&1
python3 scripts/pmc_summary.py $OUT 2>&1 | grep -A9 "$FILT"
