#!/bin/bash
# wave priority in conv_ws (loader half / MFMA half at s_setprio 1): per-layer and step A/B
mkdir -p gpurun_out/r6k
ROWS=40 bash scripts/r6/ab_layers.sh "conv_kernel<bf16,9" ab/off32.so ab/prio_ld.so ab/prio_mf.so > gpurun_out/r6k/layers.txt 2>&1
bash scripts/r6/ab_many.sh 3 ab/off32.so ab/prio_ld.so ab/prio_mf.so 2>&1 | tee gpurun_out/r6k/step.txt
