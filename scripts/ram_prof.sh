#!/bin/bash
# ON THE GPU BOX: per-kernel GPU time of rd_ram_mix alone (rocprofv3 --kernel-trace --stats around scripts/ram_bench.py) under the debug
# library's switches: RD_RAM_DFT bits 0 / 1 / 2 = pass A / B / C on the matrix cores (csrc/ram_dft.hip), 0 = the FFT kernels.
# usage: bash scripts/ram_prof.sh [u8|f32] [S] [masks...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export RAMDSIR_DEBUG_LIB=1
K=${1:-u8}; S=${2:-400}; shift; shift
for w in ${@:-7 0}; do
  export RD_RAM_DFT=$w
  rm -rf /tmp/rp_$w
  rocprofv3 --kernel-trace --stats -M --output-format csv -d /tmp/rp_$w -o rp -- python3 $R/scripts/ram_bench.py $K $S > /tmp/rp_$w.log 2>&1
  echo "== RD_RAM_DFT=$w: $(grep rd_ram_mix /tmp/rp_$w.log || tail -5 /tmp/rp_$w.log)"
  f=$(find /tmp/rp_$w -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'ram_' in r['Name']:
        print('   %-70s calls %4s  avg %8.1f us  min %8.1f  max %8.1f' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
done
