#!/bin/bash
# Runs ON THE GPU BOX: timing builds of conv_ws_kernel (ab/exp_dbg.so = the debug library with conv_pp.hip compiled -DRD_WS_EXP; results are
# wrong, durations right).  usage: bash scripts/ws_exp.sh <RD_CONV_WS value: 1 forward trace, 3 gradient trace> bits...
# bits (csrc/conv_pp.hip WS_EXP): 1 no weight loads, 2 no item loads, 4 no MFMAs, 8 loader waves idle, 16 no output stores, 32 no BatchNorm sums, 64 no fragment reads
R=${GRAFT_REPO_ROOT:-.}
cp $R/ab/exp_dbg.so $R/ram-dsir_amd/ramdsir/libramdsir_hip_dbg.so
WS=$1; shift
for x in "$@"; do
  echo "== RD_CONV_WS_EXP=$x"
  RAMDSIR_DEBUG_LIB=1 RD_CONV_WS=$WS RD_CONV_WS_EXP=$x RD_CONV_WS_TRACE_MIN=2000 python3 $R/scripts/ws_trace.py 2>/dev/null | sed -n '1,10p;$p'
done
