#!/bin/bash
mkdir -p gpurun_out/r6y
timeout 600 python3 -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "rd_zero" 2>&1 | tail -3
bash scripts/r6/ab_many.sh 3 ab/fills.so ab/zero1.so 2>&1 | tee gpurun_out/r6y/step.txt
timeout 900 python3 -m pytest tests/test_gpu_step.py tests/test_gpu_bitwise_golden.py -m gpu -x -q 2>&1 | tail -3
