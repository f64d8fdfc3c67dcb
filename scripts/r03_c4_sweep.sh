#!/bin/bash
# C4 (4 M photons): cell size and slab thickness again, now that the traversal filters its windows
for cs in 1.0 1.25 1.5; do for k in 6 8; do
echo "== cell $cs layers $k"
GVPM_CELL_SCALE=$cs GVPM_SLAB_LAYERS=$k bash scripts/shard_probe.sh gpurun_out/r03_c4s c4_strong8_rank0 c4_one_gpu
done; done
