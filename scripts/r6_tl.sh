#!/bin/bash
# kernel timeline + stats of the default bench (pipelined) and the single-stream stats: bash scripts/r6_tl.sh <tag> [bench args]
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
O=gpurun_out/$tag; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 bench.py --steps 16 --warmup 2 --only-timed "$@" > $O/trace.log 2>&1
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 scripts/timeline2.py $f > $O/timeline.txt 2>&1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/trace
GVPM_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace1 -o t -- python3 bench.py --steps 16 --warmup 2 --only-timed "$@" > $O/trace1.log 2>&1
cp $(find $O/trace1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_single.csv
rm -rf $O/trace1
python3 - <<PY
import csv
for f in ["$O/kernel_stats.csv","$O/kernel_stats_single.csv"]:
    print(f)
    for r in list(csv.DictReader(open(f)))[:16]:
        print(" ", r["Name"][:56].ljust(56), r["Calls"].rjust(5), "%10.1f us" % (float(r["AverageNs"])/1e3), r["Percentage"])
PY
