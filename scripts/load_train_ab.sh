#!/bin/bash
# ON THE GPU BOX: the 300-step Dice-proxy training (scripts/determinism_probe.py) from THREE processes at once, per library build: how many
# distinct results?  usage: bash scripts/load_train_ab.sh runs ab/a.so ab/b.so ...
R=${GRAFT_REPO_ROOT:-.}
RUNS=$1; shift
L=$R/ram-dsir_amd/ramdsir/libramdsir_hip.so
cp $L /tmp/keep.so
for v in "$@"; do
  cp $R/$v $L
  for k in 1 2 3; do (python3 $R/scripts/determinism_probe.py bf16 $RUNS 2>/dev/null | grep -E "run [0-9]+: digest" | sed "s/run [0-9]*: //" | cut -c1-30 > /tmp/lt_$k.txt) & done; wait
  echo "$v: $(cat /tmp/lt_1.txt /tmp/lt_2.txt /tmp/lt_3.txt | sort -u | wc -l) distinct result(s) in $(cat /tmp/lt_1.txt /tmp/lt_2.txt /tmp/lt_3.txt | wc -l) trainings"
done
cp /tmp/keep.so $L
