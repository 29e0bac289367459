#!/bin/bash
# ON THE GPU BOX: the bench step (no CPU baseline / fp32 leg / ablation / counters) under several builds of the library, alternating.
# usage: bash scripts/ab_bench_lib.sh rounds ab/a.so ab/b.so ...
R=${GRAFT_REPO_ROOT:-.}
ROUNDS=$1; shift
L=$R/ram-dsir_amd/ramdsir/libramdsir_hip.so
cp $L /tmp/keep.so
for round in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    cp $R/$v $L
    echo -n "$v (round $round): "
    python3 $R/bench.py --no-cpu-baseline --no-fp32-leg --no-saturation --no-ablation --no-live-pmc 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
cp /tmp/keep.so $L
