#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for agg in "scripts/probe/math_canary.bin 28" "scripts/probe/lds_canary.bin 28 28800" "scripts/probe/barrier_canary.bin 28 28800" "scripts/probe/swap_probe.bin 400 20000"; do
  for k in 1 2; do (timeout 120 $agg > /dev/null 2>&1) & done
  sleep 2
  echo "victim beside [$agg] x2: $(timeout 100 python3 scripts/r6/ram_stress.py 1500 2>&1 | grep -v amdgpu.ids | tail -1)"
  wait
done
