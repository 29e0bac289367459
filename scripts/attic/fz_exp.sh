#!/bin/bash
# Runs ON THE GPU BOX: timing builds of conv_small_bwd_fused_kernel (ab/fz_<bits>.so = the library with conv_fused.hip compiled
# -DRD_FZ_EXP=<bits>: 1 no weight-gradient phase, 2 no dgrad MFMAs, 4 no gradient stores, 8 no global loads in the loader, 16 no
# epilogue-operand loads, 32 no `a` tile writes; results wrong, durations right).  Per-launch times of the fused backward launches alone.
R=${GRAFT_REPO_ROOT:-.}
L=$R/ram-dsir_amd/ramdsir/libramdsir_hip.so
cp $L /tmp/lib_keep.so
for x in cur "$@"; do
  if [ $x = cur ]; then cp /tmp/lib_keep.so $L; else cp $R/ab/fz_$x.so $L; fi
  echo "== RD_FZ_EXP=$x: $(python3 $R/scripts/layer_bench.py bf16 400 400 2>/dev/null | grep -E 'rd_conv_bwd_fused conv_small_bwd_fused +dgrad' | sort -k8,8 | awk '{printf "%s %s  ", $8, $1}')"
done
cp /tmp/lib_keep.so $L
