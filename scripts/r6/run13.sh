#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6m; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_ram.py -m gpu -x -q 2>&1 | tail -3
echo "(1) $(timeout 300 python3 scripts/r6/pipelined_x_check.py 2000 2>&1 | grep -v amdgpu.ids | tail -2 | tr '\n' ' ')"
for fam in "fwd enc.convd5.conv2" "rd_conv conv_small_kernel"; do
  echo "(2) $fam: $(RAM_STRESS_AGG_REPEAT=10 RAM_STRESS_INPROC="$fam" timeout 300 python3 scripts/r6/ram_stress.py 300 4 2>&1 | grep -v amdgpu.ids | tail -1)"
done
echo "(5) default (KT 4 / ROWS 8 at 1024 threads)"; bash scripts/ram_prof.sh u8 400 7 2>&1 | tail -5
echo "(5b) 256-thread forms"; RD_RAM_KT=2 RD_RAM_ROWS_INV=2 bash scripts/ram_prof.sh u8 400 7 2>&1 | tail -4
bash scripts/ram_prof.sh f32 384 0 2>&1 | tail -5
