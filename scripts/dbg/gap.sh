cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaptrace2 -- python3 bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-parity --no-upload-inclusive --no-isolated > /dev/null 2>&1
python scripts/dbg/gap.py gpurun_out/gaptrace2
python -m pytest tests/test_exact_pass_gpu.py tests/test_rotated_gpu.py tests/test_parity_gpu.py tests/test_parity_vpm_gpu.py -x -q 2>&1 | tail -3
bash scripts/ab_variants.sh 3 r4 default
BENCH_ARGS="--workload c1 --steps 16" bash scripts/ab_variants.sh 2 r4 default
BENCH_ARGS="--workload c4 --emulate-gpus 8 --steps 8" bash scripts/ab_variants.sh 2 r4 default
