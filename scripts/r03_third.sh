#!/bin/bash
out=gpurun_out/r03_third; mkdir -p $out
python -m pytest tests/test_configs_gpu.py tests/test_parity_beams_gpu.py tests/test_parity_planes_gpu.py -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -12 $out/pytest.log
python scripts/beams_audit.py --scene laser --size 512 --beams 2000000 --iters 1 > $out/audit_c3.txt 2>&1
tail -3 $out/audit_c3.txt | cut -c1-600
python scripts/beams_bench.py --scene laser --size 512 --beams 2000000 --iters 4 2>/dev/null | tail -1
python scripts/beams_bench.py --scene laser --size 512 --beams 2000000 --iters 4 --tech 1d 2>/dev/null | tail -1
