#!/bin/bash
# bash scripts/round_profiles.sh rNN: everything the round's figures come from, in one GPU call (run it through gpurun):
# the GPU suite, smoke, the default bench line, the rotated-scene stress, the rocprof summaries of C2 / C1 / C3 / C5 and the
# single-GPU shard probes.  Outputs under gpurun_out/rNN/.  Afterwards, here: cp gpurun_out/rNN/rNN_* profiles/ ;
# python scripts/docs/fill.py (DESIGN.md / README.md figures) ; update profiles/rNN_tests.txt (csrc sha, counts).
tag=${1:-r06}
cd /tmp && export TMPDIR=/tmp; cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/$tag
python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/$tag/tests_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$tag/smoke.txt 2>&1
( time python bench.py ) > gpurun_out/$tag/bench_default.json 2> gpurun_out/$tag/bench_default.err
python tests/stress_bre.py cbox_rot fogroom_rot cbox_mirror_rot cbox_phong1_rot cbox_conductor_rot cbox_phong1 > gpurun_out/$tag/stress.txt 2>&1
python tests/stress_vpm.py > gpurun_out/$tag/stress_vpm.txt 2>&1
python tests/stress_beams.py > gpurun_out/$tag/stress_beams.txt 2>&1
python tests/stress_planes.py > gpurun_out/$tag/stress_planes.txt 2>&1
bash scripts/collect_profiles.sh $tag c2 c1 c3 c5 > gpurun_out/collect_$tag.log 2>&1
bash scripts/shard_probe.sh gpurun_out/$tag > gpurun_out/$tag/${tag}_shard_probe.txt 2>&1
cp profiles/${tag}_traffic*.json gpurun_out/$tag/ 2>/dev/null
true
