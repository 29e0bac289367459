#!/bin/bash
# round 6, GPU call 3: with -ffp-contract=on the old-swap and the new build must agree bit for bit; the whole GPU suite; regenerate the
# bitwise golden; a bench line
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6c; mkdir -p $O
L=ram-dsir_amd/ramdsir/libramdsir_hip.so
cp $L /tmp/keep.so
timeout 300 python3 scripts/r6/ab_bits.py dump $O/bits_new.json 6 > $O/ab_bits.txt 2>&1
cp ab/swap_old.so $L
timeout 300 python3 scripts/r6/ab_bits.py dump $O/bits_old.json 6 >> $O/ab_bits.txt 2>&1
cp /tmp/keep.so $L
python3 scripts/r6/ab_bits.py cmp $O/bits_old.json $O/bits_new.json >> $O/ab_bits.txt 2>&1
cat $O/ab_bits.txt
RD_REGEN_BITWISE=1 timeout 600 python3 -m pytest tests/test_gpu_bitwise_golden.py -m gpu -q > $O/regen.txt 2>&1; tail -2 $O/regen.txt
cp gpurun_out/hip_bitwise.json $O/hip_bitwise.json
timeout 2400 python3 -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -15 $O/pytest_gpu.txt
timeout 600 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
python3 -c "
import json
d = json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('box'), d.get('value_normalised'))
for k in d:
    if k.startswith('roofline'):
        r = d[k]; print(k, {x: r.get(x) for x in ('family','frac','step_kernel_us','avg_launch_us','traffic_over_algorithmic','step_cost_ms','alone')})
print(d.get('dominant_by_step_cost'))
"
