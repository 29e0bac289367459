#!/bin/bash
mkdir -p gpurun_out/r6u
bash scripts/r6/ab_many.sh 3 ab/evsys.so ab/evnofence.so 2>&1 | tee gpurun_out/r6u/step.txt
