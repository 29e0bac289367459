#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of one eager step (single stream) -> gpurun_out/sq_run/, summarised by scripts/pmc_agg.py
# (per kernel) and scripts/mfma_busy.py (per bench.py family -> profiles/<tag>_mfma_busy.json).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sq_run
rm -rf $O && mkdir -p $O
rocprofv3 -i $R/scripts/pmc_sq.txt --kernel-trace --output-format csv -d $O -o p -- python3 $R/scripts/pmc_step.py bf16 400 2 > $O/run.log 2>&1
python3 $R/scripts/pmc_agg.py $O '' 30 > $O/sq_counters.txt 2>&1
python3 $R/scripts/mfma_busy.py $O ${1:-r04} > $O/mfma_busy.txt 2>&1
cp $R/profiles/${1:-r04}_mfma_busy.json $O/
# keep the summaries, drop the raw per-dispatch tables (hundreds of MB)
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
tail -5 $O/mfma_busy.txt
