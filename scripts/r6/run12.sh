#!/bin/bash
# the whole-CU-LDS RAM kernels: (1) pipelined step of one process, x bit for bit; (2) RAM beside in-process conv aggressors; (3) RAM beside two
# conv-step processes; (4) the three-process load test at the bench shape; (5) RAM timing per kernel, geometry switches (debug library)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6l; mkdir -p $O
echo "(1) $(timeout 300 python3 scripts/r6/pipelined_x_check.py 2000 2>&1 | grep -v amdgpu.ids | tail -2 | tr '\n' ' ')"
for fam in "fwd enc.convd5.conv2" "fwd dec.convu1.conv2" "rd_conv conv_small_kernel" "rd_conv conv_kernel"; do
  echo "(2) $fam: $(RAM_STRESS_AGG_REPEAT=10 RAM_STRESS_INPROC="$fam" timeout 300 python3 scripts/r6/ram_stress.py 300 4 2>&1 | grep -v amdgpu.ids | tail -1)"
done
for k in 1 2; do (STRESS_NORAM=1 timeout 600 python3 scripts/step_repeat_stress.py 300 400 > $O/load_$k.txt 2>&1) & done
sleep 12
echo "(3) $(timeout 300 python3 scripts/r6/ram_stress.py 3000 2>&1 | grep -v amdgpu.ids | tail -1)"; wait
for k in 1 2 3; do (timeout 900 python3 scripts/step_repeat_stress.py 150 400 > $O/stress_$k.txt 2>&1) & done; wait
for k in 1 2 3; do echo "(4) $(grep -v amdgpu.ids $O/stress_$k.txt | tail -2 | tr '\n' ' ' | cut -c1-260)"; done
echo "(5)"; bash scripts/ram_prof.sh u8 400 7 2>&1 | tail -6
RD_RAM_KT=4 RD_RAM_ROWS_INV=4 bash scripts/ram_prof.sh u8 400 7 2>&1 | tail -5
