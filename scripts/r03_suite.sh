#!/bin/bash
out=gpurun_out/r03_suite; mkdir -p $out
python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -15 $out/pytest.log
