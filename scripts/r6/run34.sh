#!/bin/bash
# soak: the step repeated from a saved state in THREE processes at once (bench shape, then 256 and 512), and the pipelined-input check
mkdir -p gpurun_out/r6soak
for spec in "300 400" "400 256" "150 512"; do
  set -- $spec
  for p in 1 2 3; do timeout 900 python3 scripts/step_repeat_stress.py $1 $2 > gpurun_out/r6soak/p${p}_$2.txt 2>&1 & done
  wait
  for p in 1 2 3; do tail -2 gpurun_out/r6soak/p${p}_$2.txt | grep -v amdgpu; done
done
timeout 600 python3 scripts/r6/pipelined_x_check.py 500 2>&1 | tail -2
