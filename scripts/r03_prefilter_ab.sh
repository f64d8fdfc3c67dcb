#!/bin/bash
p() { python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = d['roofline']
print('$1 step %.3f ms  eval %.3f trav %.3f build %.3f  cand %.0f M/step' % (d['ms_per_step'], r['kernel_avg_ms'], r['traverse_avg_ms'], r['build_avg_ms'], d['stats']['candidates'] / 32e6))"; }
A="--no-cpu-baseline --no-parity --no-upload-inclusive --no-isolated --steps 32 --warmup 4"
for i in 1 2; do
GVPM_HIP_LIB=$PWD/build/variants/libgvpm_hip_minw4.so python bench.py $A 2>/dev/null | p "w4        "
GVPM_HIP_LIB=$PWD/build/variants/libgvpm_hip_minw5.so python bench.py $A 2>/dev/null | p "w5        "
GVPM_BUNDLE=1 GVPM_HIP_LIB=$PWD/build/variants/libgvpm_hip_minw4.so python bench.py $A 2>/dev/null | p "w4 bundle "
done
GVPM_PIPELINE=0 GVPM_HIP_LIB=$PWD/build/variants/libgvpm_hip_minw5.so python bench.py $A 2>/dev/null | p "w5 serial"
GVPM_PIPELINE=0 GVPM_BUNDLE=1 GVPM_HIP_LIB=$PWD/build/variants/libgvpm_hip_minw4.so python bench.py $A 2>/dev/null | p "w4 bundle serial"
