#!/bin/bash
p() { python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']
print('$1 step %.3f ms  eval %.3f trav %.3f build %.3f' % (d['ms_per_step'], r['kernel_avg_ms'], r['traverse_avg_ms'], r['build_avg_ms']))"; }
A="--no-cpu-baseline --no-parity --no-upload-inclusive --no-isolated --steps 48 --warmup 6"
for i in 1 2 3; do
python bench.py $A 2>/dev/null | p "default (1.5, 8/6, 4096)   "
GVPM_SLAB_LAYERS=8 GVPM_SLAB_LAYERS_X=8 python bench.py $A 2>/dev/null | p "1.5, 8/8, 4096             "
GVPM_CELL_SCALE=2.0 GVPM_SLAB_LAYERS=8 GVPM_SLAB_LAYERS_X=8 python bench.py $A 2>/dev/null | p "2.0, 8/8, 4096             "
GVPM_PLAN_TARGET=3072 GVPM_CELL_SCALE=1.75 GVPM_SLAB_LAYERS=7 GVPM_SLAB_LAYERS_X=7 python bench.py $A 2>/dev/null | p "1.75, 7/7, 3072            "
GVPM_PLAN_TARGET=3072 python bench.py $A 2>/dev/null | p "1.5, 8/6, 3072             "
done
