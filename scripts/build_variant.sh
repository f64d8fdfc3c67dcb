#!/bin/bash
# bash scripts/build_variant.sh NAME FILE.hip "<extra hipcc flags>": an A/B copy of libgvpm_hip.so with ONE translation unit
# rebuilt with extra flags -> build/variants/libgvpm_hip_NAME.so (run with GVPM_HIP_LIB=<that path>; probes only)
set -e
name=$1; file=$2; shift; shift
root=$(cd "$(dirname "$0")/.." && pwd)
cs=$root/gvpm_amd/csrc
out=$root/build/variants
mkdir -p $out
make -s -C $cs
obj=$out/${file%.hip}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-hip-fp32-correctly-rounded-divide-sqrt \
  -Wno-unused-function -Wno-unused-variable -Wno-unused-result "$@" -I$cs -c $cs/$file -o $obj
objs=""
for f in gvpm_api uploads gather_drivers gather_bre gather_vpm gather_beams gather_planes grid_build assemble poisson synth_device exact_shift; do
  if [ "$f.hip" == "$file" ]; then objs="$objs $obj"; else objs="$objs $cs/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libgvpm_hip_$name.so $objs $cs/scene_bvh.o -ldl
echo $out/libgvpm_hip_$name.so
