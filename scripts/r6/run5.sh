#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6e; mkdir -p $O
scripts/probe/simd_probe.bin > $O/simd_probe.txt 2>&1; cat $O/simd_probe.txt
ROWS=9 bash scripts/r6/ab_layers.sh "rd_wgrad" ab/sym_off.so ab/sym_w4.so ab/sym_w1.so ab/sym_none.so ab/sym_hwid.so > $O/ab_layers.txt 2>&1
cat $O/ab_layers.txt
