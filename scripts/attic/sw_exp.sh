#!/bin/bash
# timing experiments on conv_small_fwd_kernel (debug library; csrc/conv_small_fwd.hip XP bits): per-layer times of three forward launches
export RAMDSIR_DEBUG_LIB=1
for x in "$@"; do
  echo "RD_SW_EXP=$x: $(RD_SW_EXP=$x python scripts/layer_bench.py bf16 400 300 2>/dev/null | grep -E 'fwd +(enc.convd1.conv2|enc.convd2.conv2|dec.convu1.conv3) ' | sort -k8,8 | awk '{printf "%s %s us   ", $8, $1}')"
done
