#!/bin/bash
# bash scripts/regs.sh <file.hip> [name filter] [extra hipcc flags]: register / scratch / occupancy report per kernel
f=$1; filt=${2:-.}; shift; shift
cd $(dirname $f)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fno-hip-fp32-correctly-rounded-divide-sqrt \
  -Rpass-analysis=kernel-resource-usage "$@" -c $(basename $f) -o /tmp/regs.o 2>&1 |
  awk '/Function Name/{name=$0; sub(/.*Function Name: /,"",name); sub(/ \[.*/,"",name)} /VGPRs:|AGPRs:|ScratchSize|Occupancy|VGPRs Spill|SGPRs Spill/{v=$0; sub(/.*remark: +/,"",v); sub(/ \[.*/,"",v); out[name]=out[name] " | " v} END{for(n in out) print n out[n]}' | grep -E "$filt" | sed 's/_ZN4gvpm//'
