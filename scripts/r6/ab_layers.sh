#!/bin/bash
# ON THE GPU BOX: per-layer timing (scripts/layer_bench.py) under several builds of the library.  usage: ab_layers.sh "grep pattern" ab/a.so ...
R=${GRAFT_REPO_ROOT:-.}
PAT=$1; shift
L=$R/ram-dsir_amd/ramdsir/libramdsir_hip.so
cp $L /tmp/keep_layers.so
for v in "$@"; do
  cp $R/$v $L
  echo "== $v"
  timeout 300 python3 $R/scripts/layer_bench.py bf16 400 250 2>/dev/null | grep "$PAT" | head -${ROWS:-16}
done
cp /tmp/keep_layers.so $L
