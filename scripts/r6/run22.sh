#!/bin/bash
# fragment prefetch distance / burst reads in conv_ws: per-layer and step A/B
mkdir -p gpurun_out/r6o
ROWS=60 bash scripts/r6/ab_layers.sh "conv_kernel<bf16,9" ab/rdord.so ab/ahead2.so ab/burst.so ab/ahead2burst.so > gpurun_out/r6o/layers.txt 2>&1
bash scripts/r6/ab_many.sh 2 ab/rdord.so ab/ahead2.so ab/burst.so ab/ahead2burst.so 2>&1 | tee gpurun_out/r6o/step.txt
