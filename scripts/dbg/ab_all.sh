bash scripts/ab_variants.sh 3 r4 default
BENCH_ARGS="--scene cbox_rot" bash scripts/ab_variants.sh 2 r4 default
BENCH_ARGS="--workload c4 --steps 6" bash scripts/ab_variants.sh 2 r4 default
bash scripts/dbg/lines.sh
