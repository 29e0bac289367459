#!/bin/bash
mkdir -p gpurun_out/r6x
bash scripts/r6/ab_many.sh 3 ab/record.so ab/bound.so 2>&1 | tee gpurun_out/r6x/step.txt
timeout 1200 python3 -m pytest tests/test_gpu_step.py tests/test_gpu_bitwise_golden.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r6x/pytest.txt
