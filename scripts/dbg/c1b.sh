python -m pytest tests/test_parity_vpm_gpu.py tests/test_rotated_gpu.py tests/test_exact_pass_gpu.py tests/test_configs_gpu.py tests/test_primal_bre_gpu.py tests/test_host_shifts_gpu.py tests/test_unbiased_gpu.py -k "vpm or c1" -x -q 2>&1 | tail -3
python -m pytest tests/test_parity_beams_gpu.py tests/test_parity_planes_gpu.py -x -q 2>&1 | tail -2
BENCH_ARGS="--workload c1 --steps 16" bash scripts/ab_variants.sh 3 r4 default
