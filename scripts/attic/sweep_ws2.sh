#!/bin/bash
# step time with the >= 64-channel gradient launches on the persistent conv_ws_kernel<2> (debug library: RD_CONV_WS_MIN2 = tile threshold),
# under compute-unit budgets for those launches (tuning.py dgrad_cus).  Alternating rounds.
export RAMDSIR_DEBUG_LIB=1
for r in 1 2; do
for spec in "RD_NONE=0" "RD_CONV_WS_MIN2=0" "RD_CONV_WS_MIN2=0 RD_DGRAD_CUS=224" "RD_CONV_WS_MIN2=0 RD_DGRAD_CUS=192" "RD_CONV_WS_MIN2=0 RD_DGRAD_CUS=160" "RD_CONV_WS_MIN2=0 RD_DGRAD_CUS=128"; do
  out=$(env $spec python bench.py --no-cpu-baseline --no-fp32-leg --no-ablation --steps 60 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r=[d[k] for k in d if k.startswith('roofline') and isinstance(d[k],dict) and d[k].get('family')=='conv64'][0]
print('%.3f ms/step  conv64 in-step %.1f us avg, alone %.1f us' % (d['ms_per_step'], r['avg_launch_us'], r['alone']['avg_launch_us']))")
  echo "$spec: $out"
done; done
