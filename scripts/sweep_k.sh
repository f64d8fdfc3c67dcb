f() { python bench.py --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']), round(d['ms_per_step'],3), round(r['kernel_avg_ms'],3), round(r['traverse_avg_ms'],3), round(r['build_avg_ms'],3), d['stats']['candidates']//16)"; }
for k in "2 2" "3 4" "4 4" "4 6" "6 8" "8 8"; do set -- $k; echo "layers $1 $2: $(GVPM_SLAB_LAYERS=$1 GVPM_SLAB_LAYERS_X=$2 f)"; done
for c in 0.7 0.85 1.2; do echo "cell $c: $(GVPM_CELL_SCALE=$c f)"; echo "cell $c K 3/4: $(GVPM_CELL_SCALE=$c GVPM_SLAB_LAYERS=3 GVPM_SLAB_LAYERS_X=4 f)"; done
