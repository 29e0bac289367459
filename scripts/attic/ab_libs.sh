#!/bin/bash
# Runs ON THE GPU BOX: A/B of two builds of the library in the step.  usage: bash scripts/ab_libs.sh ab/old.so ab/new.so [rounds]
# Each round copies one build over ram-dsir_amd/ramdsir/libramdsir_hip.so and runs the bench (no CPU baseline, no fp32 leg, no ablation);
# rounds on one box with alternating order, ms/step per run.
R=${GRAFT_REPO_ROOT:-.}
A=$1; B=$2; N=${3:-3}
L=$R/ram-dsir_amd/ramdsir/libramdsir_hip.so
for i in $(seq $N); do
  if [ $((i % 2)) = 1 ]; then order="$A $B"; else order="$B $A"; fi      # alternate who runs first (the second run of a pair is not the same box state)
  for v in $order; do
    cp $R/$v $L
    ms=$(python3 $R/bench.py --no-cpu-baseline --no-fp32-leg --no-ablation --steps 100 --warmup 10 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'])")
    echo "$v $ms ms/step"
  done
done
