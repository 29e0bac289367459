#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6j; mkdir -p $O
for k in 1 2 3; do (STRESS_DETAIL=1 timeout 900 python3 scripts/step_repeat_stress.py 120 400 > $O/detail_$k.txt 2>&1) & done; wait
for k in 1 2 3; do echo "== detail $k"; grep -v "amdgpu.ids" $O/detail_$k.txt | grep -v "seg.keep\[[1-9]\|rec.keep\|wrong pixels" | cut -c1-600 | head -40; done
