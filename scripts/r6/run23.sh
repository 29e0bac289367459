#!/bin/bash
mkdir -p gpurun_out/r6p
timeout 600 python3 bench.py --no-cpu-baseline --no-fp32-leg --no-ablation --no-live-pmc --no-saturation > gpurun_out/r6p/bench.json 2> gpurun_out/r6p/bench.err
python3 -c "
import json; d=json.loads(open('gpurun_out/r6p/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], json.dumps(d['box']))"
tail -3 gpurun_out/r6p/bench.err
