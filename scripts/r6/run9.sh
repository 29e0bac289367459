#!/bin/bash
# load-dependent differences at 400 x 400: which tensor moves first (STRESS_DETAIL), and does it need the RAM kernels (STRESS_NORAM)?
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6i; mkdir -p $O
for k in 1 2 3; do (STRESS_DETAIL=1 timeout 900 python3 scripts/step_repeat_stress.py 60 400 > $O/detail_$k.txt 2>&1) & done; wait
for k in 1 2 3; do echo "== detail $k"; grep -v "amdgpu.ids" $O/detail_$k.txt | cut -c1-400 | head -40; done
for k in 1 2 3; do (STRESS_NORAM=1 timeout 900 python3 scripts/step_repeat_stress.py 100 400 > $O/noram_$k.txt 2>&1) & done; wait
for k in 1 2 3; do echo "== noram $k"; tail -3 $O/noram_$k.txt | cut -c1-300; done
