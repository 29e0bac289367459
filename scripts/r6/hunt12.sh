#!/bin/bash
# round 5's reproducer (three processes, the forward segment of a 64 x 64 step, no RAM) under three builds
cd ${GRAFT_REPO_ROOT:-.}
export RD_FORK=0 STRESS_NORAM=1 STRESS_HEAD=1
L=ram-dsir_amd/ramdsir/libramdsir_hip.so
cp $L /tmp/keep12.so
for v in ab/swap_slp.so ab/swap_noslp.so ab/product.so; do
  cp $v $L
  for k in 1 2 3; do (timeout 400 python3 scripts/step_repeat_stress.py 2500 64 > /tmp/h12_$k.txt 2>&1) & done; wait
  echo "$v: $(for k in 1 2 3; do tail -1 /tmp/h12_$k.txt | sed 's/.*step: //'; done | tr '\n' ';')"
done
cp /tmp/keep12.so $L
bash scripts/r6/ab_many.sh 3 ab/product.so ab/swap_noslp.so
