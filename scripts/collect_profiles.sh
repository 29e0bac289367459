#!/bin/bash
# Runs ON THE GPU BOX (gpurun -- 'bash scripts/collect_profiles.sh'): the bench line, the rocprofv3 kernel statistics
# of the same command, and the HBM counters (FETCH_SIZE / WRITE_SIZE in separate passes, kernel-trace only).
# Raw output goes to gpurun_out/profiles_run/; scripts/pmc_traffic.py turns it into the files under profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_run
rm -rf $O && mkdir -p $O
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -M --output-format csv -d $O/stats -o p -- python3 $R/bench.py --no-cpu-baseline --no-fp32-leg --no-ablation --no-live-pmc > $O/stats.log 2>&1
rocprofv3 -i $R/scripts/pmc_hbm.txt --kernel-trace -M --output-format csv -d $O/pmc -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fp32-leg --no-ablation --no-live-pmc > $O/pmc.log 2>&1
# the kernel trace is large: keep the statistics, drop the per-dispatch trace of the stats run
rm -f $O/stats/*kernel_trace.csv
ls -la $O $O/stats $O/pmc/* | head -40
