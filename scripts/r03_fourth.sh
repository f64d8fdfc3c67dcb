#!/bin/bash
out=gpurun_out/r03_fourth; mkdir -p $out
python -m pytest tests/test_configs_gpu.py tests/test_parity_beams_gpu.py -q -k "beams or c3" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -12 $out/pytest.log
python scripts/beams_audit.py --scene laser --size 512 --beams 2000000 --iters 1 > $out/audit_c3.txt 2>&1
tail -2 $out/audit_c3.txt | cut -c1-600
for t in 3d 1d; do
python scripts/beams_bench.py --scene laser --size 512 --beams 2000000 --iters 4 --tech $t 2>/dev/null | tail -1
done
python scripts/beams_timing.py --scene laser --size 512 --beams 2000000 --iters 2 2>/dev/null | tail -6
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/ks_c3 -o t -- python3 $GRAFT_REPO_ROOT/scripts/beams_bench.py --scene laser --size 512 --beams 2000000 --iters 4 > $GRAFT_REPO_ROOT/$out/ks_c3.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv,glob
f=glob.glob("$out/ks_c3/**/*kernel_stats.csv",recursive=True)
if f:
    for r in list(csv.DictReader(open(f[0])))[:8]:
        print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), r["AverageNs"].rjust(12), r["Percentage"])
PY
