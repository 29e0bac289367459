#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export RAM_STRESS_AGG_REPEAT=10
for v in v_base v_noslp v_vol v_both; do
  echo "$v: $(RD_LIB_OVERRIDE=$PWD/ab/$v.so RAM_STRESS_INPROC="fwd enc.convd5.conv2" timeout 200 python3 scripts/r6/ram_stress.py 300 4 2>&1 | grep -v amdgpu.ids | tail -1)"
done
