#!/bin/bash
mkdir -p gpurun_out/r6q
RAMDSIR_DEBUG_LIB=1 RD_CONV_WS_TRACE_MIN=1 timeout 600 python3 scripts/r6/ws_prologue.py > gpurun_out/r6q/prologue.txt 2>&1
cat gpurun_out/r6q/prologue.txt | tail -60
