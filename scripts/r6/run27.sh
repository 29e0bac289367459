#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6t; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o p -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-fp32-leg --no-ablation --no-live-pmc --no-saturation --no-box > $O/log.txt 2>&1
f=$(ls $O/kt/*kernel_trace.csv | head -1); ls -la $f
python3 $R/scripts/r6/lane_gaps.py $f > $O/gaps.txt 2>&1; cat $O/gaps.txt
rm -f $O/kt/*kernel_trace.csv
