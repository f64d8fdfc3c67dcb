#!/bin/bash
# A/B of environment settings at C3: bash scripts/r03_env_ab.sh "A=1" "GVPM_X=2 GVPM_Y=3" ...
for v in "$@"; do
  echo -n "[$v] "; env $v python scripts/beams_bench.py --scene laser --size 512 --beams 2000000 --iters 4 --tech ${TECH:-3d} --phases 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('eval', d['kernel_ms'], 'trav', d.get('trav_ms'), 'build', d.get('build_ms'), 'ms_per_iter', d['ms_per_iter'], 'evals', d['per_iter']['evaluations'])"
done
