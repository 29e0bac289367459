#!/bin/bash
# ON THE GPU BOX: run-to-run repeatability of the 300-step Dice-proxy training (scripts/determinism_probe.py) under several builds of the
# library, alternating, `runs` runs each per round.  usage: bash scripts/determinism_ab.sh runs rounds ab/a.so ab/b.so ...
R=${GRAFT_REPO_ROOT:-.}
RUNS=$1; ROUNDS=$2; shift; shift
L=$R/ram-dsir_amd/ramdsir/libramdsir_hip.so
cp $L /tmp/keep.so
for round in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    cp $R/$v $L
    echo "== $v (round $round): distinct digests and how often"
    python3 $R/scripts/determinism_probe.py bf16 $RUNS 2>/dev/null | grep -E "run [0-9]+: digest" | sed 's/run [0-9]*: //' | cut -c1-30 | sort | uniq -c
  done
done
cp /tmp/keep.so $L
