#!/bin/bash
# round 6, GPU call 1: swap probe (alone, then three at once), driver-form vs long-form bench, per-layer baseline, the GPU test suite
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r6a; mkdir -p $O
P=scripts/probe/swap_probe.bin
( echo "== alone"; timeout 120 $P 20 20000
  echo "== three processes at once"
  for k in 1 2 3; do (timeout 200 $P 20 20000 > $O/swap_$k.txt 2>&1) & done; wait
  cat $O/swap_1.txt $O/swap_2.txt $O/swap_3.txt ) > $O/swap_probe.txt 2>&1
Q="--no-cpu-baseline --no-fp32-leg --no-ablation --no-live-pmc --no-saturation"
for i in 1 2; do
  timeout 300 python3 bench.py --steps 20 --warmup 5 $Q > $O/bench_short_$i.json 2> $O/bench_short_$i.err
  timeout 300 python3 bench.py --steps 100 --warmup 10 $Q > $O/bench_long_$i.json 2> $O/bench_long_$i.err
done
timeout 300 python3 scripts/layer_bench.py bf16 400 80 > $O/layer_bench.txt 2>&1
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
grep -h '"value"' $O/bench_*.json | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['steps'], d['value'], d['ms_per_step'], d.get('box'), d.get('value_normalised'))
"
cat $O/swap_probe.txt
