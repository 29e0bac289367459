#!/bin/bash
# round 6, final collection: GPU suite, bench lines of every config, rocprof stats + PMC of the bench command, per-layer and per-lane timing
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6h; mkdir -p $O
timeout 2400 python3 -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; tail -6 $O/pytest_gpu.txt
bash scripts/collect_profiles.sh > $O/collect.log 2>&1
cp gpurun_out/profiles_run/bench.json $O/bench.json
for c in C3 C5 F256; do timeout 600 python3 bench.py --config $c --no-cpu-baseline --no-ablation --no-saturation > $O/bench_$c.json 2> $O/bench_$c.err; done
timeout 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-leg --no-ablation --no-live-pmc --no-saturation > $O/bench_driver_form.json 2>/dev/null
timeout 300 python3 scripts/layer_bench.py bf16 400 250 > $O/layer_bench.txt 2>&1
timeout 300 python3 scripts/contended_ops.py 30 > $O/contended_ops.txt 2>&1
# stamps inside conv_ws_kernel's steps (debug library), forward and gradient launch of one layer; the gradient launches on the whole device
for w in fwd dgrad; do RAMDSIR_DEBUG_LIB=1 RD_CONV_WS_TRACE_MIN=1 timeout 300 python3 scripts/r6/ws_trace2.py $w dec.convu2.conv3 10 > $O/ws_fine_$w.txt 2>&1; done
RAMDSIR_DEBUG_LIB=1 RD_DGRAD_CUS=0 timeout 300 python3 scripts/layer_bench.py bf16 400 250 2>/dev/null | grep "conv_kernel<bf16,9" > $O/layer_bench_dgrad256.txt
python3 -c "
import json
for f in ('bench', 'bench_C3', 'bench_C5', 'bench_F256', 'bench_driver_form'):
    try:
        d = json.loads(open('$O/%s.json' % f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d.get('box', {}).get('copy_gbs'), d.get('box', {}).get('mfma_tflops'), d.get('value_normalised'))
    except Exception as e: print(f, 'ERR', e)
"
head -5 $O/contended_ops.txt
