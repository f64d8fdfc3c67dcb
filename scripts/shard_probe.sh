f() { python bench.py "$@" --steps 16 --warmup 2 --no-cpu-baseline | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']), round(d['ms_per_step'],3), round(d['config']['evals_per_iter_per_gpu']), 'eval',round(r['kernel_avg_ms'],3), 'trav',round(r['traverse_avg_ms'],3), 'build',round(r['build_avg_ms'],3))"; }
echo "C2 x8 pipelined"; f --emulate-gpus 8
echo "C2 x8 single stream"; GVPM_PIPELINE=0 f --emulate-gpus 8
echo "C4 x8 pipelined"; f --emulate-gpus 8 --tile 362 --photons 4000000
echo "C4 x8 single"; GVPM_PIPELINE=0 f --emulate-gpus 8 --tile 362 --photons 4000000
echo "C2 x2"; f --emulate-gpus 2
