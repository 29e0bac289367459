#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6hunt10; mkdir -p $O
echo "== alone"; timeout 60 scripts/probe/pk_canary.bin 5
for fam in "fwd enc.convd5.conv2" "fwd dec.convu1.conv2" "rd_conv conv_kernel"; do
  tag=$(echo "$fam" | tr -c 'A-Za-z0-9' '_')
  for k in 1 2; do (timeout 120 python3 scripts/r6/aggressor.py "$fam" 25 > $O/agg_${tag}_$k.txt 2>&1) & done
  sleep 9
  echo "== beside $fam"; timeout 60 scripts/probe/pk_canary.bin 12
  wait
done
