#!/bin/bash
# robustness battery with the final library
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6n; mkdir -p $O
for c in C2 C3 C5 F256; do echo "pipelined $c: $(timeout 400 python3 scripts/r6/pipelined_x_check.py 1500 $c 2>&1 | grep -v amdgpu.ids | tail -1)"; done
for sz in 256 512; do
  for k in 1 2 3; do (timeout 900 python3 scripts/step_repeat_stress.py 120 $sz > $O/stress_${sz}_$k.txt 2>&1) & done; wait
  for k in 1 2 3; do echo "load test $sz: $(grep -v amdgpu.ids $O/stress_${sz}_$k.txt | tail -1)"; done
done
for k in 1 2 3; do (timeout 900 python3 scripts/step_repeat_stress.py 400 400 > $O/stress_400_$k.txt 2>&1) & done; wait
for k in 1 2 3; do echo "load test 400 x 400 reps: $(grep -v amdgpu.ids $O/stress_400_$k.txt | tail -1)"; done
for k in 1 2; do (timeout 200 scripts/probe/mfma_spin.bin mfma16 60 > /dev/null 2>&1) & done; sleep 2
echo "pipelined C2 beside two MFMA spinners: $(timeout 400 python3 scripts/r6/pipelined_x_check.py 1000 C2 2>&1 | grep -v amdgpu.ids | tail -1)"
echo "RAM beside two MFMA spinners: $(timeout 300 python3 scripts/r6/ram_stress.py 2000 2>&1 | grep -v amdgpu.ids | tail -1)"; wait
