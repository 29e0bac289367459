#!/bin/bash
# victim library variants (only the victim process loads the variant; aggressors use the product library)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6hunt2; mkdir -p $O
for v in product rampad ramexcl; do
  for fam in "rd_wgrad" "rd_conv conv_small_kernel"; do
    tag=$(echo "$v $fam" | tr -c 'A-Za-z0-9' '_')
    for k in 1 2; do (timeout 120 python3 scripts/r6/aggressor.py "$fam" 25 > $O/agg_${tag}_$k.txt 2>&1) & done
    sleep 9
    if [ $v = product ]; then
      timeout 100 python3 scripts/r6/ram_stress.py 1500 > $O/victim_$tag.txt 2>&1
    else
      RD_LIB_OVERRIDE=$PWD/ab/$v.so timeout 100 python3 scripts/r6/ram_stress.py 1500 > $O/victim_$tag.txt 2>&1
    fi
    wait
    echo "$v | $fam: $(grep -v amdgpu.ids $O/victim_$tag.txt | tail -1)"
  done
done
