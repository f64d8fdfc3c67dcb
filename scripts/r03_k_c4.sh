#!/bin/bash
for k in 6 8; do echo "== y/z slab layers $k"; GVPM_SLAB_LAYERS=$k bash scripts/shard_probe.sh gpurun_out/r03_k$k c2 c2_weak8_rank0 c4_one_gpu c4_strong8_rank0; done
for k in 6 8; do echo "== y/z slab layers $k (again)"; GVPM_SLAB_LAYERS=$k bash scripts/shard_probe.sh gpurun_out/r03_k$k c4_one_gpu c4_strong8_rank0; done
