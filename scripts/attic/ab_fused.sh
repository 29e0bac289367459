#!/bin/bash
# A/B of the fused small-channel backward (csrc/conv_fused.hip) in the step context: bench.py with tuning option fused_bwd on / off
# (debug library: the host reads RD_FUSED_BWD_HOST only there), alternating, plus the per-launch table of the fused configuration.
mkdir -p gpurun_out/r3
export RAMDSIR_DEBUG_LIB=1
for rep in 1 2; do
  for f in 1 0; do
    RD_FUSED_BWD_HOST=$f python bench.py --no-cpu-baseline --no-fp32-leg --steps 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fused_bwd=$f rep $rep: %.3f ms/step  %.1f images/s  final_loss %s' % (d['ms_per_step'], d['value'], d['config']['final_loss']))"
  done
done
RD_FUSED_BWD_HOST=1 python scripts/layer_bench.py bf16 400 60 > gpurun_out/r3/layer_fused.txt 2>&1
grep -E "total us|conv_small_bwd_fused|rd_wgrad  |rd_conv_bwd" gpurun_out/r3/layer_fused.txt | head -40
