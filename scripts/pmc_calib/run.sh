#!/bin/bash
# ON THE GPU BOX: build the calibration kernels, run them under the two counter passes, print true bytes / counter per kernel.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_calib
rm -rf $O && mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/scripts/pmc_calib/pmc_calib.hip -o $O/pmc_calib || exit 1
rocprofv3 -i $R/scripts/pmc_hbm.txt --kernel-trace --output-format csv -d $O/pmc -o p -- $O/pmc_calib > $O/run.log 2>&1
python3 $R/scripts/pmc_calib/summarize.py $O/pmc > $O/summary.txt
cat $O/summary.txt
