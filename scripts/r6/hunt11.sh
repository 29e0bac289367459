#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for agg in mfma mfma16 valu pkswap dpp lds trans; do
  for k in 1 2; do (timeout 60 scripts/probe/mfma_spin.bin $agg 14 > /dev/null 2>&1) & done
  sleep 2
  echo "== pk_canary beside 2 x mfma_spin $agg"; timeout 60 scripts/probe/pk_canary.bin 8 | cut -c1-420 | head -4
  wait
done
echo "== beside the step's conv_kernel family (samples)"
for k in 1 2; do (timeout 120 python3 scripts/r6/aggressor.py "rd_conv conv_kernel" 22 > /dev/null 2>&1) & done
sleep 9; timeout 60 scripts/probe/pk_canary.bin 8 | cut -c1-420 | head -10; wait
