#!/bin/bash
# timing experiments of wgrad_ws_kernel (debug library, RD_WGWS_EXP bits: csrc/wgrad.hip), side budget 128 CUs and full width
export RAMDSIR_DEBUG_LIB=1
for cus in 128 0; do
for x in ${WGWS_X:-0 1 4 5}; do
  echo "== RD_SIDE_CUS=$cus RD_WGWS_EXP=$x"
  RD_SIDE_CUS=$cus RD_WGWS_EXP=$x python scripts/layer_bench.py bf16 400 400 2>/dev/null | grep -E "rd_wgrad +wgrad +(dec.convu4.conv3|dec.convu3.conv3|dec.convu2.conv3|enc.convd5.conv2|enc.convd3.conv2) "
done
done
