GVPM_TRACE_EXACT=1 python bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-upload-inclusive --no-isolated 2>&1 | grep "exact beams" | head -2
BENCH_ARGS="--workload c3 --steps 8" bash scripts/ab_variants.sh 2 head default
