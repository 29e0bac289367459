#!/bin/bash
# ON THE GPU BOX: the forward segment of a 64 x 64 step repeated under load (three processes at once, one stream, no RAM kernels;
# scripts/step_repeat_stress.py) with several builds of the library in turn: how many repetitions differ from the first?
# usage: bash scripts/load_stress_ab.sh reps ab/a.so ab/b.so ...
R=${GRAFT_REPO_ROOT:-.}
REPS=$1; shift
L=$R/ram-dsir_amd/ramdsir/libramdsir_hip.so
cp $L /tmp/keep.so
export RD_FORK=0 STRESS_NORAM=1 STRESS_HEAD=1
for v in "$@"; do
  cp $R/$v $L
  for k in 1 2 3; do (timeout 400 python3 $R/scripts/step_repeat_stress.py $REPS 64 > /tmp/ls_$k.txt 2>&1) & done; wait
  echo "$v: $(for k in 1 2 3; do tail -1 /tmp/ls_$k.txt | sed 's/.*step: //'; done | tr '\n' ';')"
done
cp /tmp/keep.so $L
