#!/bin/bash
# round 3, first GPU call: the whole -m gpu suite + C3 timing with kernel stats
out=gpurun_out/r03_first; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
python scripts/beams_bench.py --scene laser --size 512 --beams 2000000 --iters 4 > $out/c3_3d.json 2> $out/c3_3d.err
cat $out/c3_3d.json
python scripts/beams_bench.py --scene laser --size 512 --beams 2000000 --iters 4 --tech 1d > $out/c3_1d.json 2> $out/c3_1d.err
cat $out/c3_1d.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/ks_c3 -o t -- python3 $GRAFT_REPO_ROOT/scripts/beams_bench.py --scene laser --size 512 --beams 2000000 --iters 4 > $GRAFT_REPO_ROOT/$out/ks_c3.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv,glob
f=glob.glob("$out/ks_c3/**/*kernel_stats.csv",recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:16]:
        print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), r["AverageNs"].rjust(12), r["Percentage"])
PY
