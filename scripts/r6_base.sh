#!/bin/bash
# round 6 baseline: default bench line, a kernel timeline of the C2 step (every launch), kernel statistics
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r6base; mkdir -p $O
python3 bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 bench.py --steps 16 --warmup 2 --only-timed > $O/trace.log 2>&1
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 scripts/timeline2.py $f > $O/timeline.txt 2>&1
f2=$(find $O/trace -name "*kernel_stats.csv" | head -1); cp $f2 $O/kernel_stats.csv
rm -rf $O/trace
tail -c 1500 $O/bench.json
