#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6hunt3; mkdir -p $O
echo "== canary alone"; timeout 60 scripts/probe/lds_canary.bin 5
for fam in "rd_wgrad" "rd_conv conv_small_kernel" "rd_conv conv_kernel" "rd_conv_bwd_fused"; do
  tag=$(echo "$fam" | tr -c 'A-Za-z0-9' '_')
  for k in 1 2; do (timeout 120 python3 scripts/r6/aggressor.py "$fam" 25 > $O/agg_${tag}_$k.txt 2>&1) & done
  sleep 9
  echo "== canary beside $fam"; timeout 60 scripts/probe/lds_canary.bin 12 > $O/canary_$tag.txt 2>&1; head -30 $O/canary_$tag.txt
  wait
done
