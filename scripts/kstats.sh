#!/bin/bash
# per-kernel rocprofv3 stats of one bench.py command:  bash scripts/kstats.sh <tag> <bench args...>
TAG=$1; shift
OUT=$PWD/gpurun_out/ks_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 ${KS_SCRIPT:-bench.py} "$@" ${KS_EXTRA---no-cpu-baseline} > $OUT/bench.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:24]:
    print(r["Name"][:64].ljust(64), r["Calls"].rjust(6), r["AverageNs"].rjust(12), r["Percentage"])
PY
