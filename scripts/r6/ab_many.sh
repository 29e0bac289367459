#!/bin/bash
# ON THE GPU BOX: same-box A/B of several builds of the library in the step.  usage: ab_many.sh rounds ab/a.so ab/b.so ...
R=${GRAFT_REPO_ROOT:-.}
N=$1; shift
L=$R/ram-dsir_amd/ramdsir/libramdsir_hip.so
cp $L /tmp/keep_many.so
for i in $(seq $N); do
  for v in "$@"; do
    cp $R/$v $L
    ms=$(python3 $R/bench.py --no-cpu-baseline --no-fp32-leg --no-ablation --no-live-pmc --no-saturation --no-box --steps 100 --warmup 10 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'])")
    echo "round $i $v $ms ms/step"
  done
done
cp /tmp/keep_many.so $L
