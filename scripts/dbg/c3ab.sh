BENCH_ARGS="--workload c3 --steps 8" bash scripts/ab_variants.sh 1 head default bPV bNC bNS bALL
