#!/bin/bash
# end of round 3: the rocprofv3 summaries of the four bench lines, the four lines as the driver would run them, the shard probes
bash scripts/r03_profiles.sh > gpurun_out/r03_profiles.log 2>&1
bash scripts/r03_lines.sh > gpurun_out/r03_lines.log 2>&1
bash scripts/shard_probe.sh gpurun_out/r03_shard > gpurun_out/r03_shard_probe.txt 2>&1
tail -5 gpurun_out/r03_lines.log; cat gpurun_out/r03_shard_probe.txt
