#!/bin/bash
mkdir -p gpurun_out/r6p
for i in 1 2 3; do
timeout 600 python3 bench.py --no-cpu-baseline --no-fp32-leg --no-ablation --no-live-pmc --no-saturation > gpurun_out/r6p/bench.json 2> gpurun_out/r6p/bench.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r6p/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], json.dumps(d['box']['shader_ghz'])[:120])"
done
timeout 600 python3 -m pytest tests/test_gpu_bench.py -m gpu -x -q 2>&1 | tail -3
export RAMDSIR_DEBUG_LIB=1 RD_CONV_WS_TRACE_MIN=1
for l in dec.convu2.conv3 enc.convd5.conv3 enc.convd4.conv2 dec.convu4.conv3; do timeout 300 python3 scripts/r6/ws_trace2.py fwd $l 4 2>&1 | grep -E "alone|kernel body|before the first" ; done
