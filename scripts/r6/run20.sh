#!/bin/bash
# gradient launches of conv_ws_kernel alone at their step budget (160 CUs) and on the whole device (256), beside the forward launches
mkdir -p gpurun_out/r6m
export RAMDSIR_DEBUG_LIB=1
timeout 300 python3 scripts/layer_bench.py bf16 400 250 2>/dev/null | grep "conv_kernel<bf16,9" > gpurun_out/r6m/layers_dgrad160.txt
RD_DGRAD_CUS=0 timeout 300 python3 scripts/layer_bench.py bf16 400 250 2>/dev/null | grep "conv_kernel<bf16,9" > gpurun_out/r6m/layers_dgrad256.txt
head -8 gpurun_out/r6m/layers_dgrad256.txt
