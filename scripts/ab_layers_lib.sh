#!/bin/bash
# ON THE GPU BOX: per-layer timing under several builds of the library: scripts/layer_bench.py (every launch of one step alone, HIP events)
# with each build in turn, twice; prints the rows that match the pattern (scripts/ab_layers_cmp.py compares them).
# usage: bash scripts/ab_layers_lib.sh 'pattern' ab/a.so ab/b.so ...
R=${GRAFT_REPO_ROOT:-.}
PAT=$1; shift
L=$R/ram-dsir_amd/ramdsir/libramdsir_hip.so
cp $L /tmp/keep.so
for round in 1 2; do
  for v in "$@"; do
    cp $R/$v $L
    echo "== $v (round $round)"
    python3 $R/scripts/layer_bench.py bf16 400 200 2>/dev/null | grep -E "$PAT" | head -60
  done
done
cp /tmp/keep.so $L
