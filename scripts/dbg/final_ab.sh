python -m pytest tests -m gpu -x -q 2>&1 | tail -8
bash scripts/ab_variants.sh 3 r4 default
BENCH_ARGS="--workload c4 --emulate-gpus 8 --steps 8" bash scripts/ab_variants.sh 2 r4 default
BENCH_ARGS="--workload c4 --steps 6" bash scripts/ab_variants.sh 1 r4 default
BENCH_ARGS="--workload c3 --steps 8" bash scripts/ab_variants.sh 1 r4 default
BENCH_ARGS="--workload c1 --steps 16" bash scripts/ab_variants.sh 2 r4 default
