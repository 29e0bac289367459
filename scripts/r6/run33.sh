#!/bin/bash
mkdir -p gpurun_out/r6z
export RAMDSIR_DEBUG_LIB=1
RD_CONV_WS_TRACE_MIN=2560 RD_CONV_WS_TRACE_MODE=2 RD_CONV_WS_TRACE_CIN=64 timeout 300 python3 scripts/r6/wg_times.py 2>&1 | grep -v amdgpu | tee gpurun_out/r6z/wg_dgrad64.txt
RD_CONV_WS_TRACE_MIN=600 RD_CONV_WS_TRACE_MODE=2 RD_CONV_WS_TRACE_CIN=256 timeout 300 python3 scripts/r6/wg_times.py 2>&1 | grep -v amdgpu | tee gpurun_out/r6z/wg_dgrad256.txt
RD_CONV_WS_TRACE_MIN=2560 RD_CONV_WS_TRACE_MODE=1 RD_CONV_WS_TRACE_CIN=64 timeout 300 python3 scripts/r6/wg_times.py 2>&1 | grep -v amdgpu | tee gpurun_out/r6z/wg_fwd64.txt
