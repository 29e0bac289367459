#!/bin/bash
mkdir -p gpurun_out/r6v
timeout 300 python3 scripts/join_probe.py > gpurun_out/r6v/join.txt 2>&1; cat gpurun_out/r6v/join.txt | tail -14
bash scripts/r6/run27.sh 2>&1 | grep -A 40 "last launches" | head -50
