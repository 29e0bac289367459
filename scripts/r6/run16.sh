#!/bin/bash
# fine-grained stamps inside conv_ws_kernel steps (debug library)
mkdir -p gpurun_out/r6i
export RAMDSIR_DEBUG_LIB=1 RD_CONV_WS_TRACE_MIN=1
for spec in "fwd dec.convu2.conv3" "dgrad dec.convu2.conv3" "fwd dec.convu3.conv3" "dgrad dec.convu3.conv3"; do
  set -- $spec
  timeout 300 python3 scripts/r6/ws_trace2.py $1 $2 10 > gpurun_out/r6i/fine_$1_$2.txt 2>&1
done
tail -n 40 gpurun_out/r6i/fine_dgrad_dec.convu2.conv3.txt
