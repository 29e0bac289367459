#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6hunt5; mkdir -p $O
echo "== alone"; timeout 60 scripts/probe/barrier_canary.bin 6 28800
for fam in "rd_conv conv_small_kernel" "rd_conv conv_kernel"; do
  tag=$(echo "$fam" | tr -c 'A-Za-z0-9' '_')
  for k in 1 2; do (timeout 120 python3 scripts/r6/aggressor.py "$fam" 30 > $O/agg_${tag}_$k.txt 2>&1) & done
  sleep 9
  echo "== beside $fam"; timeout 60 scripts/probe/barrier_canary.bin 18 28800
  wait
done
