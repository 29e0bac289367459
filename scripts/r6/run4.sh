#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6d; mkdir -p $O
bash scripts/r6/ab_many.sh 3 ab/sym_off_fast.so ab/sym_on_fast.so ab/sym_off.so ab/sym_on.so > $O/ab_many.txt 2>&1
cat $O/ab_many.txt
timeout 300 python3 scripts/layer_bench.py bf16 400 200 > $O/layer_new.txt 2>&1
grep "rd_wgrad" $O/layer_new.txt | head -24
timeout 600 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_step.py -m gpu -q -k "wgrad or share_the_gpu" 2>&1 | tail -5
