#!/bin/bash
# ON THE GPU BOX: victim = rd_ram_mix repeated (ram_stress.py), aggressors = two processes looping one family of the step's launches.
# usage: hunt.sh "family1" "family2" ...
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6hunt; mkdir -p $O
for fam in "$@"; do
  tag=$(echo "$fam" | tr -c 'A-Za-z0-9' '_')
  for k in 1 2; do (timeout 120 python3 scripts/r6/aggressor.py "$fam" 25 > $O/agg_${tag}_$k.txt 2>&1) & done
  sleep 9
  timeout 100 python3 scripts/r6/ram_stress.py 1500 > $O/victim_$tag.txt 2>&1
  wait
  echo "$fam: $(grep -v amdgpu.ids $O/victim_$tag.txt | tail -1)   [$(grep 'launches match' $O/agg_${tag}_1.txt | head -1)]"
done
