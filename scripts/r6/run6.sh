#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6f; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_blocks.py tests/test_gpu_fold.py -m gpu -x -q 2>&1 | tail -4
ROWS=20 bash scripts/r6/ab_layers.sh "conv_small_kernel<bf16,9" ab/regroup.so ab/direct.so > $O/ab_layers.txt 2>&1
cat $O/ab_layers.txt
bash scripts/r6/ab_many.sh 3 ab/regroup.so ab/direct.so > $O/ab_many.txt 2>&1; cat $O/ab_many.txt
