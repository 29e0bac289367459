#!/bin/bash
# timing experiments of conv_small_bwd_fused_kernel (debug library, RD_FZ_EXP bits: csrc/conv_fused.hip)
export RAMDSIR_DEBUG_LIB=1
for x in 0 1 2 4 8 16 3 12 28 31; do
  echo "== RD_FZ_EXP=$x"
  RD_FZ_EXP=$x python scripts/layer_bench.py bf16 400 400 2>/dev/null | grep -E "conv_small_bwd_fused +dgrad\+wgrad (dec.convu1.conv3|enc.convd1.conv2|dec.out1|enc.convd2.conv2) "
done
