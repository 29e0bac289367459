#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6k; mkdir -p $O
echo "== RAM alone, one process"; timeout 300 python3 scripts/r6/ram_stress.py 3000 2>&1 | grep -v amdgpu.ids | tail -3
echo "== three RAM processes at once"
for k in 1 2 3; do (timeout 600 python3 scripts/r6/ram_stress.py 3000 > $O/ram3_$k.txt 2>&1) & done; wait
for k in 1 2 3; do grep -v amdgpu.ids $O/ram3_$k.txt | tail -4; done
echo "== one RAM process beside two conv-step processes (no RAM in them)"
for k in 1 2; do (STRESS_NORAM=1 timeout 600 python3 scripts/step_repeat_stress.py 300 400 > $O/load_$k.txt 2>&1) & done
timeout 600 python3 scripts/r6/ram_stress.py 3000 > $O/ram_beside.txt 2>&1; wait
grep -v amdgpu.ids $O/ram_beside.txt | tail -4; tail -1 $O/load_1.txt $O/load_2.txt
