#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export RAM_STRESS_AGG_REPEAT=10
echo "== in-process, single layers x 10 per repetition (product library)"
for fam in "fwd enc.convd1.conv2" "fwd enc.convd1.conv1" "fwd dec.convu1.conv3" "fwd dec.out1" "fwd enc.convd2.conv2" "fwd dec.convu2.conv3" "dgrad dec.convu2.conv3" "fwd enc.convd3.conv2" "dgrad dec.convu1.conv2" "fwd dec.convu1.conv2" "fwd rec.convu1.conv3" "fwd enc.convd5.conv2"; do
  echo "$fam: $(RAM_STRESS_INPROC="$fam" timeout 200 python3 scripts/r6/ram_stress.py 300 4 2>&1 | grep -v amdgpu.ids | tail -2 | tr '\n' ' ')"
done
