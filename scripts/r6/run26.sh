#!/bin/bash
# deferred bias-table stores in conv_ws + unrolled seg-loss final: parity, per-layer and step A/B, prologue stamps
mkdir -p gpurun_out/r6s
timeout 1200 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_bitwise_golden.py tests/test_gpu_blocks.py -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r6s/pytest.txt
cat gpurun_out/r6s/pytest.txt
ROWS=60 bash scripts/r6/ab_layers.sh "conv_kernel<bf16,9\|seg_loss" ab/head.so ab/tabdefer.so > gpurun_out/r6s/layers.txt 2>&1
bash scripts/r6/ab_many.sh 3 ab/head.so ab/tabdefer.so 2>&1 | tee gpurun_out/r6s/step.txt
RAMDSIR_DEBUG_LIB=1 RD_CONV_WS_TRACE_MIN=1 timeout 600 python3 scripts/r6/ws_prologue.py > gpurun_out/r6s/prologue.txt 2>&1
