#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6g; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_ops.py -m gpu -x -q -k wgrad 2>&1 | tail -4
ROWS=22 bash scripts/r6/ab_layers.sh "rd_wgrad" ab/sym0.so ab/sym1.so ab/sym2.so > $O/ab_layers.txt 2>&1
cat $O/ab_layers.txt
bash scripts/r6/ab_many.sh 3 ab/sym0.so ab/sym1.so ab/sym2.so > $O/ab_many.txt 2>&1; cat $O/ab_many.txt
