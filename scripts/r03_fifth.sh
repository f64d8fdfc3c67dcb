#!/bin/bash
out=gpurun_out/r03_fifth; mkdir -p $out
python -m pytest tests/test_configs_gpu.py tests/test_parity_beams_gpu.py -q -k "beams or c3" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -5 $out/pytest.log
TECHS="3d 1d" bash scripts/r03_ab.sh
python scripts/beams_timing.py --scene laser --size 512 --beams 2000000 --iters 2 2>/dev/null | tail -6
python scripts/beams_bench.py --scene cbox --size 256 --beams 200000 --iters 4 2>/dev/null | tail -1
python scripts/beams_bench.py --scene fogroom --size 256 --beams 200000 --iters 4 2>/dev/null | tail -1
