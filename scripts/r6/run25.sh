#!/bin/bash
# kernel-argument placement (HIP_FORCE_DEV_KERNARG) A/B on the step and on the forward chain
mkdir -p gpurun_out/r6r
B="python3 bench.py --no-cpu-baseline --no-fp32-leg --no-ablation --no-live-pmc --no-saturation --no-box --steps 100 --warmup 10"
for i in 1 2 3; do
  for v in unset 0 1; do
    if [ $v = unset ]; then ms=$($B 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'])");
    else ms=$(HIP_FORCE_DEV_KERNARG=$v $B 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readlines()[-1])['ms_per_step'])"); fi
    echo "round $i HIP_FORCE_DEV_KERNARG=$v $ms ms/step"
  done
done | tee gpurun_out/r6r/kernarg.txt
