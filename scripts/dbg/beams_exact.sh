python -m pytest tests/test_exact_pass_gpu.py -k beam -x -q 2>&1 | tail -3
python -m pytest tests/test_rotated_gpu.py -k beams -x -q 2>&1 | tail -3
python -m pytest tests/test_parity_beams_gpu.py tests/test_glossy_parents_gpu.py tests/test_compact_beams.py tests/test_configs_gpu.py tests/test_host_shifts_gpu.py tests/test_primal_gpu.py -x -q 2>&1 | tail -5
GVPM_TRACE_EXACT=1 python scripts/dbg/beams1d.py 2>&1 | grep "exact beams" | cut -c1-200
BENCH_ARGS="--workload c3 --steps 8" bash scripts/ab_variants.sh 2 head default
