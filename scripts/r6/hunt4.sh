#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6hunt4; mkdir -p $O
echo "== in-process aggressors on a second stream (one process)"
for fam in "rd_wgrad" "rd_conv conv_small_kernel" "rd_conv conv_kernel"; do
  echo "$fam: $(RAM_STRESS_INPROC="$fam" timeout 200 python3 scripts/r6/ram_stress.py 600 8 2>&1 | grep -v amdgpu.ids | tail -1)"
done
for v in ram81 ram54; do
  for fam in "rd_wgrad" "rd_conv conv_small_kernel"; do
    tag=$(echo "$v $fam" | tr -c 'A-Za-z0-9' '_')
    for k in 1 2; do (timeout 120 python3 scripts/r6/aggressor.py "$fam" 25 > $O/agg_${tag}_$k.txt 2>&1) & done
    sleep 9
    RD_LIB_OVERRIDE=$PWD/ab/$v.so timeout 100 python3 scripts/r6/ram_stress.py 1500 > $O/victim_$tag.txt 2>&1
    wait
    echo "$v | $fam: $(grep -v amdgpu.ids $O/victim_$tag.txt | tail -1)"
  done
done
