#!/bin/bash
# step time under tuning-option overrides (debug library: ramdsir/tuning.py reads RD_* only there).  usage: sweep_opts.sh "VAR=VAL[,VAR=VAL]" ...
export RAMDSIR_DEBUG_LIB=1
for spec in "$@"; do
  envs=$(echo "$spec" | tr ',' ' ')
  out=$(env $envs python bench.py --no-cpu-baseline --no-fp32-leg --steps 30 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.3f ms/step  %.1f images/s' % (d['ms_per_step'], d['value']))")
  echo "$spec: $out"
done
