#!/bin/bash
run() { echo -n "$1: "; shift; env "$@" GVPM_BENCH_UPLOAD_TRACE=1 python bench.py --no-cpu-baseline --no-parity --steps 32 --warmup 4 $EXTRA 2>&1 | grep "^\[upload\]" | tr '\n' ' '; echo; }
EXTRA=""             run "A default order, isolated leg" X=1
EXTRA=""             run "A again" X=1
EXTRA="--no-isolated" run "B default order, no isolated" X=1
EXTRA="--no-isolated" run "B again" X=1
EXTRA=""             run "C prefetch first, isolated" GVPM_BENCH_UPLOAD_MODES=prefetch,packed,serial
EXTRA=""             run "C again" GVPM_BENCH_UPLOAD_MODES=prefetch,packed,serial
