#!/bin/bash
# Runs ON THE GPU BOX: per-launch timings (scripts/layer_bench.py, one stream) of dispatch variants selected through the
# debug library's switches.  Usage: bash scripts/ab_layers.sh "NAME:VAR=VAL,VAR=VAL" ...   -> gpurun_out/ab_<NAME>.log
cd $GRAFT_REPO_ROOT
export RAMDSIR_DEBUG_LIB=1
for spec in "$@"; do
  name=${spec%%:*}; vars=${spec#*:}
  ( IFS=','; for kv in $vars; do [ -n "$kv" ] && export "$kv"; done; python scripts/layer_bench.py bf16 400 400 > gpurun_out/ab_$name.log 2>&1 )
  echo "== $name ($vars)"; grep -E "^total us|rd_wgrad wgrad|conv_small_kernel<bf16,9|rd_pool|pool " gpurun_out/ab_$name.log | head -8
done
