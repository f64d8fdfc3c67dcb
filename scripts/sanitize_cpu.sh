#!/bin/bash
# AddressSanitizer + UBSan over the CPU-side code (oracle restatement, synthetic hosts): builds instrumented copies
# of liboracle.so / libgvpm_host.so, runs the CPU tests that exercise them, restores the normal builds.
# (GPU sanitizers are not available on the pool: the HIP library is covered by the parity tests instead.)
set -e
cd "$(dirname "$0")/.."
T=$(mktemp -d)
FL="-O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-sanitize-recover=undefined"
g++ $FL -ffp-contract=off -fopenmp -o $T/liboracle.so oracle/oracle_api.cpp
g++ $FL -pthread -o $T/libgvpm_host.so gvpm_amd/host/synth.cpp gvpm_amd/host/host_api.cpp
cp oracle/liboracle.so $T/liboracle.orig.so; cp gvpm_amd/host/libgvpm_host.so $T/libgvpm_host.orig.so
restore() { cp $T/liboracle.orig.so oracle/liboracle.so; cp $T/libgvpm_host.orig.so gvpm_amd/host/libgvpm_host.so; rm -rf $T; }
trap restore EXIT
cp $T/liboracle.so oracle/liboracle.so; cp $T/libgvpm_host.so gvpm_amd/host/libgvpm_host.so
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
  python -m pytest tests/test_oracle.py tests/test_oracle_beams.py tests/test_oracle_planes.py tests/test_oracle_vpm.py \
  tests/test_host.py tests/test_camera_paths.py tests/test_oracle_pins.py tests/test_indep_statements.py \
  tests/test_indep_lightpaths.py tests/test_glossy_parents.py tests/test_mirror_walk.py tests/test_primal_bre.py tests/test_oracle_accel.py -x -q -m "not gpu"
