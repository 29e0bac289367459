#!/bin/bash
# timing experiments (results wrong) inside conv_ws_kernel's gradient mode: which memory operation the loader waits for
mkdir -p gpurun_out/r6l
R=${GRAFT_REPO_ROOT:-.}
D=$R/ram-dsir_amd/ramdsir/libramdsir_hip_dbg.so
cp $D /tmp/keep_dbg.so; cp $R/ab/exp_dbg.so $D
export RAMDSIR_DEBUG_LIB=1 RD_CONV_WS_TRACE_MIN=1
for e in 0 128 256 384 512 1024 1536 2 258 1 8; do
  RD_CONV_WS_EXP=$e timeout 300 python3 scripts/r6/ws_trace2.py dgrad dec.convu2.conv3 8 > gpurun_out/r6l/dgrad_exp$e.txt 2>&1
  echo "exp $e: $(grep alone gpurun_out/r6l/dgrad_exp$e.txt) | $(grep 'kernel body' gpurun_out/r6l/dgrad_exp$e.txt)"
done
for e in 0 16 32 48 1 2 8; do
  RD_CONV_WS_EXP=$e timeout 300 python3 scripts/r6/ws_trace2.py fwd dec.convu2.conv3 8 > gpurun_out/r6l/fwd_exp$e.txt 2>&1
  echo "fwd exp $e: $(grep alone gpurun_out/r6l/fwd_exp$e.txt) | $(grep 'kernel body' gpurun_out/r6l/fwd_exp$e.txt)"
done
cp /tmp/keep_dbg.so $D
