#!/bin/bash
# incremental tile advance in conv_ws: parity, per-layer and step A/B
mkdir -p gpurun_out/r6j
timeout 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_bitwise_golden.py -m gpu -x -q -k "conv or golden or bitwise" 2>&1 | tail -3 > gpurun_out/r6j/pytest.txt
cat gpurun_out/r6j/pytest.txt
ROWS=40 bash scripts/r6/ab_layers.sh "conv_kernel<bf16,9" ab/base.so ab/inc.so ab/off32.so > gpurun_out/r6j/layers.txt 2>&1
bash scripts/r6/ab_many.sh 3 ab/base.so ab/inc.so ab/off32.so 2>&1 | tee gpurun_out/r6j/step.txt
