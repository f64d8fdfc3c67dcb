python -m pytest tests/test_parity_beams_gpu.py tests/test_rotated_gpu.py -k "beams" -x -q 2>&1 | tail -3
BENCH_ARGS="--workload c3 --steps 8" bash scripts/ab_variants.sh 2 head default
