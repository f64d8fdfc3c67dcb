#!/bin/bash
# bundle cells at the other G-BRE shapes (DESIGN.md section 6 probes)
for b in 0 1; do
  echo "== GVPM_BUNDLE=$b"
  GVPM_BUNDLE=$b bash scripts/shard_probe.sh gpurun_out/r03_bundle/shard_b$b c2_weak8_rank0 c4_one_gpu c4_strong8_rank0
done
