#!/bin/bash
out=gpurun_out/r03_second; mkdir -p $out
python scripts/beams_audit.py --scene laser --size 512 --beams 2000000 --iters 1 > $out/audit_c3.txt 2>&1
cat $out/audit_c3.txt | cut -c1-400
python scripts/beams_audit.py --scene cbox --size 256 --beams 200000 --iters 1 > $out/audit_cbox.txt 2>&1
cat $out/audit_cbox.txt | cut -c1-400
python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -15 $out/pytest.log
